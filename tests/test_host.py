"""Host-side logic of the product library, no GPU needed: the library loads and exports every
symbol its headers declare, the parser twin behaves like the reference's parser (unit cases of
src/raytracer/config/config_builder.rs:464-500 and its error strings), flattening agrees with the
oracle's independent parser, the comparator twin passes the reference's comparator tests
(src/imgcomparator/mod.rs:170-244), and the tree builder produces a valid, conservative tree.
"""
import ctypes as C
import glob
import os
import re
import struct

import numpy as np
import pytest

import __graft_entry__ as ge
import oracle as O

from conftest import GOLDEN, ROOT

rt = ge.load_package()


# ------------------------------------------------------------------ library surface
def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(rth?_[a-z_0-9]+)\s*\(", text))


def test_library_exports_every_declared_symbol():
    names = _declared("rt_amd.h") | _declared("rt_host.h")
    assert {"rt_init", "rt_upload_scene", "rt_render", "rt_render_tiles", "rt_last_error", "rt_shutdown"} <= names
    L = C.CDLL(rt.LIB_PATH)
    for n in sorted(names):
        assert hasattr(L, n), f"{n} declared in include/ but not exported"
    assert names == set(rt._SIGNATURES), "python binding table and headers disagree"


def test_render_without_init_fails_loudly():
    # No CPU fallback: without a bound GPU the render entry point reports an error.
    f = rt.frame_from_camera((0, 0, 0), (0, 0, 1), (0, 1, 0), 60, 8, 8, 1, (0, 0, 0))
    out = np.zeros((8, 8), np.uint32)
    if rt._bound_device is None:
        rc = rt.lib().rt_render(C.byref(f), out.ctypes.data, None, 0)
        assert rc != 0 and b"rt_init" in rt.lib().rt_last_error()


# ------------------------------------------------------------------ parser twin
def _parse(text):
    return rt.ParsedConfigState().parse_text(text)


def test_parse_size():
    c = _parse("size 1920 1080\n")
    assert (c.width, c.height) == (1920, 1080)


def test_parse_camera():
    c = _parse("camera 0.0 0.0 150.0 0.0 0.0 5.0 0.0 1.0 0.0 60\n")
    cam = c.camera
    assert cam["position"] == (0.0, 0.0, 150.0) and cam["look_at"] == (0.0, 0.0, 5.0) and cam["up"] == (0.0, 1.0, 0.0) and cam["fov"] == 60.0


def test_parse_ambient():
    c = _parse("ambient 0.2 0.3 0.4\n")
    assert c.ambient == tuple(np.float32([0.2, 0.3, 0.4]))


def test_parse_output():
    assert _parse("output final.png\n").output_file == "final.png"


def test_defaults():
    c = _parse("")
    assert (c.width, c.height, c.maxdepth, c.output_file) == (800, 600, 1, "output.png")
    assert c.camera == {"position": (0, 0, 0), "look_at": (0, 0, 1), "up": (0, 1, 0), "fov": 60.0}


@pytest.mark.parametrize("text,msg", [
    ("bogus 1 2\n", "Unknown configuration key: bogus"),
    ("size 640\n", "Invalid size format"),
    ("size 0 480\n", "Width and height must be greater than zero"),
    ("size 640 x\n", "invalid digit found in string"),
    ("camera 0 0 0 0 0 1 0 1 0 180\n", "Field of view (fov) must be between 1 and 179 degrees"),
    ("camera 0 0 0 0 0 1 0 1 0\n", "Invalid camera format"),
    ("ambient 0.5 0.5 1.5\n", "RGB values must be between 0.0 and 1.0"),
    ("ambient .5 .5 .5\ndiffuse .6 .1 .1\n", "Sum of diffuse color and ambient light components must not exceed 1.0"),
    ("specular -1 0 0\n", "Specular color components must be non-negative"),
    ("shininess -2\n", "Shininess must be non-negative"),
    ("sphere 0 0 0 0\n", "Sphere radius must be greater than zero"),
    ("sphere 0 0 0\n", "Invalid sphere format"),
    ("vertex 0 0 0\n", "Exceeded maximum number of vertices (maxverts)"),
    ("maxverts 3\nvertex 0 0 0\ntri 0 1 2\n", "Triangle vertex index out of bounds"),
    ("plane 0 0 0 0 1\n", "Invalid plane format"),
    ("point 0 0 0 2 0 0\n", "RGB values must be between 0.0 and 1.0"),
    ("directional 0 0 1 1 1\n", "Invalid directional light format"),
    ("size 640  480\n", "Invalid size format"),                       # double space between values -> empty token
    (" size 640 480\n", "Unknown configuration key: "),               # leading space -> empty keyword
    ("sphere 0 0 0 1e\n", "invalid float literal"),
    ("output \n", None),                                               # single token after trim? "output " splits in two
])
def test_parser_errors_match_reference_strings(text, msg):
    if msg is None:
        with pytest.raises(rt.RtError, match="Output file name cannot be empty"):
            _parse(text)
        return
    with pytest.raises(rt.RtError) as e:
        _parse(text)
    assert str(e.value) == msg
    with pytest.raises(ValueError) as e2:       # the oracle's independent parser says the same
        O.OracleScene.parse(text)
    assert str(e2.value) == msg


def test_comments_blank_and_single_token_lines_are_skipped():
    c = _parse("# comment\n\n   \nfoo\nsize 10 20 \n   # indented comment\n")
    assert (c.width, c.height) == (10, 20)


def test_material_state_is_captured_per_shape():
    c = _parse("diffuse .1 .2 .3\nsphere 0 0 0 1\nspecular .4 .5 .6\nshininess 7\nsphere 1 0 0 1\ndiffuse 0 0 0\nplane 0 0 0 0 2 0\n")
    s = c.scene()
    sm = np.ctypeslib.as_array(s.sph_mat, shape=(2, 7))
    assert np.array_equal(sm, np.float32([[.1, .2, .3, 0, 0, 0, 0], [.1, .2, .3, .4, .5, .6, 7]]))
    pm = np.ctypeslib.as_array(s.pl_mat, shape=(1, 7))
    assert np.array_equal(pm, np.float32([[0, 0, 0, .4, .5, .6, 7]]))
    assert list(np.ctypeslib.as_array(s.pl_normal, shape=(3,))) == [0.0, 1.0, 0.0]     # normalised at parse time
    assert [s.sph_order[0], s.sph_order[1], s.pl_order[0]] == [0, 1, 2]


def _flatten_from_oracle(path):
    shapes, lights, cam = O.OracleScene.load(path).export()
    return shapes, lights, cam


ALL_SCENES = sorted(glob.glob(os.path.join(GOLDEN, "test_file", "*", "*.test"))) + [
    os.path.join(GOLDEN, "test.scene"), os.path.join(GOLDEN, "final_avec_bonus.scene")]


@pytest.mark.parametrize("path", ALL_SCENES, ids=[os.path.basename(p) for p in ALL_SCENES])
def test_flattened_scene_equals_oracle_parse(path):
    cfg = rt.ParsedConfigState().load_config_file(path)
    shapes, lights, cam = _flatten_from_oracle(path)
    s = cfg.scene()
    kinds = shapes[:, 0].astype(int)
    tri, sph, pl = shapes[kinds == 1], shapes[kinds == 0], shapes[kinds == 2]
    assert (s.ntri, s.nsph, s.npl, s.nlights) == (len(tri), len(sph), len(pl), len(lights))

    def arr(p, n, k):
        return np.ctypeslib.as_array(p, shape=(n, k)) if n else np.zeros((0, k), np.float32)

    assert np.array_equal(arr(s.tri_v0, s.ntri, 3), tri[:, 1:4]) and np.array_equal(arr(s.tri_v1, s.ntri, 3), tri[:, 4:7])
    assert np.array_equal(arr(s.tri_v2, s.ntri, 3), tri[:, 7:10]) and np.array_equal(arr(s.tri_mat, s.ntri, 7), tri[:, 10:17])
    assert np.array_equal(arr(s.sph_center, s.nsph, 3), sph[:, 1:4]) and np.array_equal(arr(s.sph_radius, s.nsph, 1)[:, 0], sph[:, 4])
    assert np.array_equal(arr(s.sph_mat, s.nsph, 7), sph[:, 10:17])
    assert np.array_equal(arr(s.pl_point, s.npl, 3), pl[:, 1:4]) and np.array_equal(arr(s.pl_normal, s.npl, 3), pl[:, 4:7])
    assert np.array_equal(arr(s.lights, s.nlights, 7), lights)
    order = np.concatenate([np.ctypeslib.as_array(p, shape=(n,)) if n else np.zeros(0, np.uint32)
                            for p, n in ((s.tri_order, s.ntri), (s.sph_order, s.nsph), (s.pl_order, s.npl))])
    file_pos = np.concatenate([np.nonzero(kinds == 1)[0], np.nonzero(kinds == 0)[0], np.nonzero(kinds == 2)[0]])
    assert np.array_equal(order, file_pos)
    # frame constants (raytracer.rs:46-55) agree bit for bit with the oracle's
    f = cfg.frame()
    want = O.OracleScene.load(path).frame_setup()
    got = np.float32(list(f.pos) + list(f.cv) + list(f.right) + list(f.up) + [f.pixel_w, f.pixel_h, f.half_w, f.half_h])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


# ------------------------------------------------------------------ comparator twin
def test_compare_identical_images():
    data = [0xFF0000, 0x00FF00, 0x0000FF]
    n, d = rt.Image.compare(rt.Image(3, 1, data), rt.Image(3, 1, data))
    assert n == 0 and list(d.data) == [0, 0, 0] and d.width == 3


def test_compare_dimension_mismatch():
    with pytest.raises(rt.RtError, match="Images have different dimensions"):
        rt.Image.compare(rt.Image(2, 2, [0] * 4), rt.Image(3, 3, [0] * 9))


def test_compare_calculates_difference_correctly():
    _, d = rt.Image.compare(rt.Image(2, 1, [0xFF0000, 0xFFFFFF]), rt.Image(2, 1, [0x000000, 0x0000FF]))
    assert list(d.data) == [0xFF0000, 0xFFFF00]


def test_compare_channel_borrowing():
    _, d = rt.Image.compare(rt.Image(1, 1, [0x010000]), rt.Image(1, 1, [0x00FFFF]))
    assert d.data[0] == 0x01FFFF


def test_compare_ignores_alpha_and_tolerates_one():
    n, d = rt.Image.compare(rt.Image(2, 1, [0xFF102030, 0xFF102030]), rt.Image(2, 1, [0x00112131, 0x00122030]))
    assert n == 1 and list(d.data) == [0, 0x020000]


def test_compare_fixture_images():
    a = rt.file_to_image(os.path.join(GOLDEN, "test_compare_image1.png"))
    b = rt.file_to_image(os.path.join(GOLDEN, "test_compare_image2.png"))
    n, _ = rt.Image.compare(a, b)
    assert n == 879


# ------------------------------------------------------------------ tree builder / blob
HDR = struct.Struct("<IIQ5Q5II3IfI3f3f")


def _read_blob(blob):
    (magic, ver, total, o_nodes, o_tree, o_list, o_mat, o_light, n_pairs, n_tree, n_list, n_mat, n_light, root,
     max_depth, max_leaf, n_leaves, sah, finite, gx, gy, gz, sx, sy, sz) = HDR.unpack_from(blob.tobytes()[:HDR.size])
    assert magic == 0x52544231 and ver == 2 and total == len(blob)
    nodes = blob[o_nodes:o_nodes + 32 * n_pairs].view(np.uint32).reshape(n_pairs, 8)
    tree = blob[o_tree:o_tree + 48 * n_tree].view(np.uint32).reshape(n_tree, 12)
    lst = blob[o_list:o_list + 48 * n_list].view(np.uint32).reshape(n_list, 12)
    return dict(nodes=nodes, tree=tree, list=lst, root=root, max_depth=max_depth, max_leaf=max_leaf, n_leaves=n_leaves,
                n_mat=n_mat, n_light=n_light, origin=np.float64([gx, gy, gz]), scale=np.float64([sx, sy, sz]))


def _check_tree(b):
    nodes, tree = b["nodes"], b["tree"]
    tf = tree.view(np.float32)

    def child_box(ref, c):
        # 16-bit grid coordinates -> the exact planes they stand for (float64: exact for these products)
        w = nodes[ref, 4 * c:4 * c + 3]
        q = np.float64([w[0] & 0xFFFF, w[0] >> 16, w[1] & 0xFFFF, w[1] >> 16, w[2] & 0xFFFF, w[2] >> 16])
        return b["origin"] + q[:3] * b["scale"], b["origin"] + q[3:] * b["scale"]
    seen = np.zeros(len(tree), dtype=np.int32)
    max_depth = 0
    leaves = 0

    def shape_box(rec_u, rec_f):
        kind = rec_u[3] & 3
        if kind == 1:
            v = np.stack([rec_f[0:3], rec_f[4:7], rec_f[8:11]])
            return v.min(0), v.max(0)
        assert kind == 0
        c, r = rec_f[0:3], rec_f[4]
        return c - r, c + r

    stack = [(b["root"], 0, None, None)]
    while stack:
        ref, depth, lo, hi = stack.pop()
        if ref == 0xFFFFFFFF:
            continue
        if ref & 0x80000000:
            first, cnt = (ref & 0x7FFFFFFF) >> 3, (ref & 7) + 1
            leaves += 1
            max_depth = max(max_depth, depth)
            orders = []
            for k in range(first, first + cnt):
                seen[k] += 1
                slo, shi = shape_box(tree[k], tf[k])
                if lo is not None:
                    assert (slo >= lo).all() and (shi <= hi).all(), "leaf box does not contain its shape"
                orders.append(int(tree[k][7]))
            assert orders == sorted(orders)
            continue
        for c in (0, 1):
            clo, chi = child_box(ref, c)
            if lo is not None:      # a child may stick out of its parent by the padding and one grid step, not more
                slack = 2 * b["scale"] + 1e-5 * (np.abs(lo) + np.abs(hi)) + 1e-4 * (hi - lo).max()
                assert (clo >= lo - slack).all() and (chi <= hi + slack).all()
            cref = int(nodes[ref, 4 * c + 3])
            stack.append((cref, depth + 1, clo, chi))
    assert (seen == 1).all(), "every tree shape must sit in exactly one leaf"
    assert max_depth == b["max_depth"] <= 64 and leaves == b["n_leaves"]


def test_tree_of_test_scene_is_valid():
    cfg = rt.ParsedConfigState().load_config_file(os.path.join(GOLDEN, "test.scene"))
    blob, info = rt.pack_scene_host(cfg.scene())
    b = _read_blob(blob)
    # the ground sphere of radius 10000 is kept out of the tree (it would stretch the node grid)
    assert info["bvh_prims"] == 265 and info["list_prims"] == 1 and len(b["tree"]) == 265
    assert sorted(b["tree"][:, 7].tolist() + b["list"][:, 7].tolist()) == list(range(266))
    assert (b["list"][0][3] & 3) == 0 and b["list"].view(np.float32)[0][4] == 10000.0
    _check_tree(b)


def test_tree_of_dragon_is_valid():
    cfg = rt.ParsedConfigState().load_config_file(os.path.join(GOLDEN, "final_avec_bonus.scene"))
    blob, info = rt.pack_scene_host(cfg.scene())
    b = _read_blob(blob)
    assert info["bvh_prims"] == 100000 and info["list_prims"] == 1 and info["max_leaf"] <= 8
    assert (b["list"][0][3] & 3) == 2 and b["n_mat"] == 2 and b["n_light"] == 2
    assert sorted(b["tree"][:, 7].tolist()) == list(range(100000))
    _check_tree(b)


def test_small_scenes_skip_the_tree():
    cfg = rt.ParsedConfigState().load_config_file(os.path.join(GOLDEN, "test_file", "jalon6", "tp63.test"))
    blob, info = rt.pack_scene_host(cfg.scene())
    b = _read_blob(blob)
    assert b["root"] == 0xFFFFFFFF and info["list_prims"] == 4 and info["bvh_prims"] == 0
    assert b["list"][:, 7].tolist() == [0, 1, 2, 3]          # file order


def test_degenerate_inputs_build():
    rng = np.random.default_rng(1)
    # identical centroids (cannot be binned), a single shape, an empty scene
    tri = np.tile(rng.random(9, dtype=np.float32), (50, 1))
    for s in (rt.SceneArrays(tri=tri, tri_mat=np.zeros(7)), rt.SceneArrays(spheres=[[0, 0, 0, 1]] * 9 + [[5, 5, 5, 1]], sph_mat=np.zeros(7)),
              rt.SceneArrays()):
        blob, info = rt.pack_scene_host(s.c)
        b = _read_blob(blob)
        if info["bvh_prims"]:
            _check_tree(b)
    with pytest.raises(rt.RtError, match="non-finite"):
        rt.pack_scene_host(rt.SceneArrays(spheres=[[0, 0, 0, 1]] * 9 + [[np.inf, 0, 0, 1]], sph_mat=np.zeros(7)).c)


def test_depth_limit_holds_on_adversarial_input():
    # Exponentially growing spheres make SAH peel one shape per level; the builder must fall back
    # to median splits before the tree outgrows the traversal stack.
    n = 3000
    r = np.float32(1.02) ** np.arange(n, dtype=np.float32)
    sph = np.stack([r * 2, np.zeros(n), np.zeros(n), r], axis=1).astype(np.float32)
    sph = sph[np.isfinite(sph).all(1)]
    blob, info = rt.pack_scene_host(rt.SceneArrays(spheres=sph, sph_mat=np.zeros(7)).c)
    assert info["max_depth"] <= 64
    _check_tree(_read_blob(blob))


# ------------------------------------------------------------------ sharding arithmetic
@pytest.mark.parametrize("height,nshards,tile", [(480, 1, 8), (2160, 8, 8), (2161, 8, 8), (7, 4, 8), (100, 3, 16), (4320, 8, 8)])
def test_shard_rows_partition_the_frame(height, nshards, tile):
    rows = [rt.shard_rows(height, s, nshards, tile) for s in range(nshards)]
    assert sum(rows) == height
    owner = (np.arange(height) // tile) % nshards
    assert rows == [int((owner == s).sum()) for s in range(nshards)]
