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


def test_headers_are_plain_c(tmp_path):
    """The boundary is a C ABI: both headers compile as strict C99, and the calling sequence a C host
    would write type-checks against them."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("needs gcc")
    src = tmp_path / "host.c"
    src.write_text("""
#include <stdio.h>
#include <stdlib.h>
#include "rt_amd.h"
#include "rt_host.h"
int main(void) {
    rth_config* cfg = rth_load_config_file("x.scene");
    rt_scene scene; rt_frame frame; rt_stats stats; rt_bvh_info info;
    if (!cfg) { fprintf(stderr, "%s\\n", rt_last_error()); return 1; }
    rth_config_scene(cfg, &scene); rth_config_frame(cfg, &frame);
    rt_init(0); rt_upload_scene_ex(&scene, RT_BUILD_ON_GPU); rt_get_bvh_info(&info);
    { uint32_t* pixels = malloc(4ull * frame.width * frame.height);
      if (rt_render(&frame, pixels, &stats, 0)) fprintf(stderr, "%s\\n", rt_last_error());
      free(pixels); }
    rt_select_lane(RT_LANES - 1);
    rt_shutdown(); rth_config_free(cfg);
    return 0;
}
""")
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-fsyntax-only", str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


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


def _assert_same_scene(cfg, oracle_scene):
    shapes, lights, cam = oracle_scene.export()
    s = cfg.scene()
    kinds = shapes[:, 0].astype(int)
    tri, sph, pl = shapes[kinds == 1], shapes[kinds == 0], shapes[kinds == 2]
    assert (s.ntri, s.nsph, s.npl, s.nlights) == (len(tri), len(sph), len(pl), len(lights))

    def arr(p, n, k):
        return np.ctypeslib.as_array(p, shape=(n, k)) if n else np.zeros((0, k), np.float32)

    def same(a, b):       # bit patterns, so that NaN == NaN
        return np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32), np.ascontiguousarray(b, np.float32).view(np.uint32))

    assert same(arr(s.tri_v0, s.ntri, 3), tri[:, 1:4]) and same(arr(s.tri_v1, s.ntri, 3), tri[:, 4:7])
    assert same(arr(s.tri_v2, s.ntri, 3), tri[:, 7:10]) and same(arr(s.tri_mat, s.ntri, 7), tri[:, 10:17])
    assert same(arr(s.sph_center, s.nsph, 3), sph[:, 1:4]) and same(arr(s.sph_radius, s.nsph, 1)[:, 0], sph[:, 4])
    assert same(arr(s.sph_mat, s.nsph, 7), sph[:, 10:17])
    assert same(arr(s.pl_point, s.npl, 3), pl[:, 1:4]) and same(arr(s.pl_normal, s.npl, 3), pl[:, 4:7])
    assert same(arr(s.pl_mat, s.npl, 7), pl[:, 10:17])
    assert same(arr(s.lights, s.nlights, 7), lights)
    order = np.concatenate([np.ctypeslib.as_array(p, shape=(n,)) if n else np.zeros(0, np.uint32)
                            for p, n in ((s.tri_order, s.ntri), (s.sph_order, s.nsph), (s.pl_order, s.npl))])
    file_pos = np.concatenate([np.nonzero(kinds == 1)[0], np.nonzero(kinds == 0)[0], np.nonzero(kinds == 2)[0]])
    assert np.array_equal(order, file_pos)
    # frame constants (raytracer.rs:46-55) agree bit for bit with the oracle's
    f = cfg.frame()
    want = oracle_scene.frame_setup()
    got = np.float32(list(f.pos) + list(f.cv) + list(f.right) + list(f.up) + [f.pixel_w, f.pixel_h, f.half_w, f.half_h])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    i = oracle_scene.info()
    assert (f.width, f.height, f.maxdepth) == (i["width"], i["height"], i["maxdepth"])


@pytest.mark.parametrize("path", ALL_SCENES, ids=[os.path.basename(p) for p in ALL_SCENES])
def test_flattened_scene_equals_oracle_parse(path):
    _assert_same_scene(rt.ParsedConfigState().load_config_file(path), O.OracleScene.load(path))


def _big_scene_text(seed, lines=160_000):
    """A few MiB of scene text (the parser works in chunks of 1 MiB) with every kind of line
    scattered through it, so that vertex numbering, the current material, ambient and maxverts
    all have to be carried across chunk borders."""
    rng = np.random.default_rng(seed)
    out = ["size 320 200", "camera 0 -30 12 0 0 0 0 0 1 50", "maxdepth 3", "maxverts 1000", "ambient 0.1 0.1 0.1"]
    nverts, maxverts = 0, 1000
    for i in range(lines):
        r = rng.random()
        if r < 0.45 and nverts < maxverts:
            x, y, z = rng.uniform(-20, 20, 3)
            out.append(f"vertex {float(x)!r} {y:.6e} {z:.3f}")
            nverts += 1
        elif r < 0.80 and nverts >= 3:
            a, b, c = rng.integers(0, nverts, 3)
            out.append(f"tri {a} {b} +{c}")
        elif r < 0.84:
            x, y, z, rad = rng.uniform(-20, 20, 3).tolist() + [rng.uniform(0.1, 2)]
            out.append(f"sphere {x:.4f} {y:.4f} {z:.4f} {rad:.4f}")
        elif r < 0.88:
            d = rng.uniform(0, 0.8, 3)
            out.append(f"diffuse {d[0]:.3f} {d[1]:.3f} {d[2]:.3f}")
        elif r < 0.91:
            d = rng.uniform(0, 1, 3)
            out.append(f"specular {d[0]:.3f} {d[1]:.3f} {d[2]:.3f}")
        elif r < 0.93:
            out.append(f"shininess {rng.integers(0, 50)}")
        elif r < 0.94:
            out.append("# a comment line\r")
        elif r < 0.945:
            n = rng.uniform(-1, 1, 3)
            out.append(f"plane 0 0 {rng.uniform(-30, -20):.2f} {n[0]:.3f} {n[1]:.3f} {abs(n[2]) + 0.1:.3f}")
        elif r < 0.95:
            out.append(f"directional {rng.uniform(-1, 1):.3f} 0.5 1 0.2 0.2 0.2" if rng.random() < 0.5 else f"point {rng.uniform(-30, 30):.3f} 10 40 .3 .3 .3")
        elif r < 0.955:
            maxverts += 500
            out.append(f"maxverts {maxverts}")
        elif r < 0.956:
            out.append("ambient 0.1 0.15 0.2")
        else:
            out.append("")
    return "\n".join(out) + "\n"


def test_multi_chunk_text_equals_oracle_parse(tmp_path):
    text = _big_scene_text(1)
    assert len(text) > 3 << 20
    _assert_same_scene(_parse(text), O.OracleScene.parse(text))
    path = tmp_path / "big.scene"                  # and through the memory-mapped file route
    path.write_text(text)
    _assert_same_scene(rt.ParsedConfigState().load_config_file(str(path)), O.OracleScene.parse(text))


@pytest.mark.parametrize("first,second,msg", [
    ("tri 0 1 99999999", "bogus 1 2", "Triangle vertex index out of bounds"),
    ("bogus 1 2", "tri 0 1 99999999", "Unknown configuration key: bogus"),
    ("diffuse 0.95 0 0", "sphere 0 0 0", "Sum of diffuse color and ambient light components must not exceed 1.0"),
    ("sphere 0 0 0 -1", "diffuse 0.95 0 0", "Sphere radius must be greater than zero"),
    ("tri 0 1 18446744073709551616", "vertex 0 0", "number too large to fit in target type"),
])
def test_earliest_failing_line_wins_across_chunks(first, second, msg):
    text = _big_scene_text(2, lines=120_000)
    lines = text.split("\n")
    lines.insert(len(lines) // 3, first)           # lands in an earlier chunk than `second`
    lines.insert(2 * len(lines) // 3, second)
    bad = "\n".join(lines)
    with pytest.raises(rt.RtError) as e:
        _parse(bad)
    assert str(e.value) == msg
    with pytest.raises(ValueError) as e2:
        O.OracleScene.parse(bad)
    assert str(e2.value) == msg


def test_vertex_pool_limit_is_checked_where_the_line_stands():
    text = "maxverts 2\n" + "#" * (2 << 20) + "\nvertex 0 0 0\nvertex 1 0 0\nvertex 2 0 0\nmaxverts 10\n"
    with pytest.raises(rt.RtError, match="Exceeded maximum number of vertices"):
        _parse(text)
    ok = "maxverts 2\nvertex 0 0 0\n" + "#" * (2 << 20) + "\nvertex 1 0 0\nmaxverts 3\nvertex 2 0 0\ntri 0 1 2\n"
    assert _parse(ok).scene().ntri == 1


def test_missing_file_reports_the_os_error():
    with pytest.raises(rt.RtError, match=r"No such file or directory \(os error 2\)"):
        rt.ParsedConfigState().load_config_file("/nonexistent/x.scene")


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
HDR = struct.Struct("<IIQ5Q5II3IfI3f3fI")


def _read_blob(blob):
    (magic, ver, total, o_nodes, o_tree, o_list, o_mat, o_light, n_pairs, n_tree, n_list, n_mat, n_light, root,
     max_depth, max_leaf, n_leaves, sah, finite, gx, gy, gz, sx, sy, sz, tree_kinds) = HDR.unpack_from(blob.tobytes()[:HDR.size])
    assert magic == 0x52544231 and ver == 7 and total == len(blob)
    nodes = blob[o_nodes:o_nodes + 32 * n_pairs].view(np.uint32).reshape(n_pairs, 8)
    tree = blob[o_tree:o_tree + 48 * n_tree].view(np.uint32).reshape(n_tree, 12)
    lst = blob[o_list:o_list + 48 * n_list].view(np.uint32).reshape(n_list, 12)
    assert tree_kinds == int(np.bitwise_or.reduce(np.uint32(1) << (tree[:, 3] & 3))) if n_tree else tree_kinds == 0      # one bit per kind of shape in the tree
    return dict(nodes=nodes, tree=tree, list=lst, root=root, max_depth=max_depth, max_leaf=max_leaf, n_leaves=n_leaves, tree_kinds=tree_kinds,
                n_mat=n_mat, n_light=n_light, origin=np.float64([gx, gy, gz]), scale=np.float64([sx, sy, sz]))


def _check_tree(b):
    nodes, tree = b["nodes"], b["tree"]
    tf = tree.view(np.float32)

    def child_box(ref, c):
        # 16-bit grid coordinates (low | high << 16 per axis) -> the exact planes they stand for: q units below 32768,
        # 2 q - 32768 from there on (float64: exact for these products)
        w = nodes[ref, 4 * c:4 * c + 3]
        units = lambda q: np.where(q < 32768, q, 2.0 * q - 32768.0)
        lo, hi = units(np.float64(w & 0xFFFF)), units(np.float64(w >> 16))
        assert (lo <= hi).all()
        return b["origin"] + lo * b["scale"], b["origin"] + hi * b["scale"]
    seen = np.zeros(len(tree), dtype=np.int32)
    max_depth = 0
    leaves = 0

    def shape_box(rec_u, rec_f):
        kind = rec_u[3] & 3
        if kind == 1:         # a triangle is stored as v0 and its two edges; v0 + edge is the vertex to within one rounding
            v0, e1, e2 = np.float64(rec_f[0:3]), np.float64(rec_f[4:7]), np.float64(rec_f[8:11])
            v = np.stack([v0, v0 + e1, v0 + e2])
            ulp = np.abs(v).max(0) * 2.0 ** -23
            return v.min(0) + ulp, v.max(0) - ulp
        assert kind == 0
        c, r = rec_f[0:3], rec_f[4]
        return c - r, c + r

    stack = [(b["root"], 0, None, None)]
    while stack:
        ref, depth, lo, hi = stack.pop()
        if ref == 0xFFFFFFFF:
            continue
        if not ref & 0x80000000:          # an inner reference is the byte offset of a 32-byte pair
            assert ref % 32 == 0
            ref //= 32
        if ref & 0x80000000:
            first, cnt = (ref & 0x7FFFFFFF) >> 3, (ref & 7) + 1
            leaves += 1
            max_depth = max(max_depth, depth)
            for k in range(first, first + cnt):
                seen[k] += 1
                slo, shi = shape_box(tree[k], tf[k])
                if lo is not None:
                    assert (slo >= lo).all() and (shi <= hi).all(), "leaf box does not contain its shape"
            continue
        for c in (0, 1):
            clo, chi = child_box(ref, c)
            if lo is not None:      # a child may stick out of its parent by the padding and one grid step, not more
                slack = 4 * b["scale"] + 1e-5 * (np.abs(lo) + np.abs(hi)) + 1e-4 * (hi - lo).max()
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


def test_section_offsets_near_2_to_64_are_refused():
    """offset + count * size must not be computed in a way that wraps round (a crafted header would pass and the check of
    the nodes would then read far outside the buffer)."""
    cfg = rt.ParsedConfigState().load_config_file(os.path.join(GOLDEN, "test.scene"))
    blob, _ = rt.pack_scene_host(cfg.scene())
    rt.validate_blob(blob)
    for field in (16, 24, 32, 40, 48):                                     # off_nodes .. off_lights
        for off in (0xFFFFFFFFFFFFFF00, 0xFFFFFFFFFFFFFFF0, 0x8000000000000000, len(blob) + 16):
            bad = blob.copy()
            bad[field:field + 8] = np.frombuffer(struct.pack("<Q", off), dtype=np.uint8)
            with pytest.raises(rt.RtError):
                rt.validate_blob(bad)
    bad = blob.copy()                                                      # a count so large that count * size wraps
    bad[56:60] = np.frombuffer(struct.pack("<I", 0xFFFFFFFF), dtype=np.uint8)
    with pytest.raises(rt.RtError):
        rt.validate_blob(bad)


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
# ------------------------------------------------------------------ PNG files
@pytest.mark.parametrize("size", [(1, 1), (7, 5), (640, 480), (2000, 1500)])
def test_png_round_trip_and_independent_decoder(size, tmp_path):
    from PIL import Image as PILImage
    w, h = size
    rng = np.random.default_rng(w + h)
    px = rng.integers(0, 1 << 32, size=(h, w), dtype=np.uint32)
    if w > 1000:                              # smooth content, several bands of rows
        yy, xx = np.mgrid[0:h, 0:w]
        px = (((xx * 255 // w) << 16) | ((yy * 255 // h) << 8) | ((xx ^ yy) & 0xFF)).astype(np.uint32)
    path = str(tmp_path / "a.png")
    rt.save_image(rt.Image(w, h, px), path)
    back = rt.file_to_image(path)
    assert (back.width, back.height) == (w, h) and np.array_equal(back.pixels(), px & 0xFFFFFF)
    pil = np.asarray(PILImage.open(path).convert("RGB"), dtype=np.uint32)       # written file is a valid PNG for others
    assert np.array_equal((pil[..., 0] << 16) | (pil[..., 1] << 8) | pil[..., 2], px & 0xFFFFFF)


@pytest.mark.parametrize("mode", ["L", "LA", "RGB", "RGBA", "P", "I;16"])
def test_png_reader_handles_every_colour_type(mode, tmp_path):
    from PIL import Image as PILImage
    rng = np.random.default_rng(5)
    w, h = 37, 23
    if mode == "I;16":
        im = PILImage.fromarray(rng.integers(0, 1 << 16, size=(h, w), dtype=np.uint16))
        want_rgb = np.asarray(im, dtype=np.uint32) >> 8
        want = (want_rgb << 16) | (want_rgb << 8) | want_rgb
    else:
        base = PILImage.fromarray(rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8), "RGBA")
        im = base.convert(mode) if mode != "P" else base.convert("RGB").quantize(64)
        rgb = np.asarray(im.convert("RGB"), dtype=np.uint32)
        want = (rgb[..., 0] << 16) | (rgb[..., 1] << 8) | rgb[..., 2]
    path = str(tmp_path / "m.png")
    im.save(path)
    assert np.array_equal(rt.file_to_image(path).pixels(), want)


def test_png_reader_matches_pil_on_every_fixture():
    from PIL import Image as PILImage
    files = glob.glob(os.path.join(GOLDEN, "**", "*.png"), recursive=True)
    assert len(files) >= 23
    for f in files:
        ref = np.asarray(PILImage.open(f).convert("RGB"), dtype=np.uint32)
        assert np.array_equal(rt.file_to_image(f).pixels(), (ref[..., 0] << 16) | (ref[..., 1] << 8) | ref[..., 2]), f


def test_png_errors(tmp_path):
    with pytest.raises(rt.RtError, match="os error 2"):
        rt.file_to_image("/nonexistent/a.png")
    p = tmp_path / "x.png"
    p.write_bytes(b"not a png at all")
    with pytest.raises(rt.RtError, match="not a PNG file"):
        rt.file_to_image(str(p))
    good = tmp_path / "g.png"
    rt.save_image(rt.Image(4, 4, np.arange(16)), str(good))
    raw = bytearray(good.read_bytes())
    raw[40] ^= 0x55
    p.write_bytes(bytes(raw))
    with pytest.raises(rt.RtError, match="PNG"):
        rt.file_to_image(str(p))
    with pytest.raises(rt.RtError):
        rt.save_image(rt.Image(4, 4, np.arange(16)), "/nonexistent/dir/a.png")


# ------------------------------------------------------------------ blob validation and the scene cache file
def _dragon_blob():
    cfg = rt.ParsedConfigState().load_config_file(os.path.join(GOLDEN, "final_avec_bonus.scene"))
    blob, _ = rt.pack_scene_host(cfg.scene())
    return cfg, blob


def test_builder_output_passes_the_blob_check():
    for name in ("final_avec_bonus.scene", "test.scene", os.path.join("test_file", "jalon5", "tp55.test")):
        cfg = rt.ParsedConfigState().load_config_file(os.path.join(GOLDEN, name))
        blob, _ = rt.pack_scene_host(cfg.scene())
        rt.validate_blob(blob)


@pytest.mark.parametrize("damage,msg", [
    ("child_out_of_range", "child outside the node array"),
    ("child_misaligned", "child outside the node array"),
    ("cycle", "node pair referenced twice"),
    ("leaf_out_of_range", "leaf outside the shape array"),
    ("material", "unknown kind or material"),
    ("truncated", "size does not match its header"),
    ("magic", "bad magic or version"),
    ("orphan", "unreachable node pairs"),
    ("sphere_among_triangles", "kinds of the tree's shapes differ from the header"),
])
def test_damaged_blobs_are_rejected_before_they_reach_a_gpu(damage, msg):
    _, blob = _dragon_blob()
    blob = blob.copy()
    b = _read_blob(blob)
    nodes = b["nodes"]                       # views into `blob`: [lo/hi words x3, ref] per child
    inner = [i for i in range(len(nodes)) if not (nodes[i, 3] & 0x80000000)]
    if damage == "child_out_of_range":
        nodes[inner[0], 3] = 32 * (len(nodes) + 5)
    elif damage == "child_misaligned":
        nodes[inner[0], 3] = nodes[inner[0], 3] + 16
    elif damage == "cycle":
        nodes[inner[5], 3] = b["root"]
    elif damage == "leaf_out_of_range":
        leafy = next(i for i in range(len(nodes)) if nodes[i, 3] & 0x80000000 and nodes[i, 3] != 0xFFFFFFFF)
        nodes[leafy, 3] = 0x80000000 | ((len(b["tree"]) - 1) << 3) | 7
    elif damage == "material":
        b["tree"][10, 3] = 1 | (b["n_mat"] << 2)
    elif damage == "truncated":
        blob = blob[:-256]
    elif damage == "magic":
        blob[0] ^= 0xFF
    elif damage == "orphan":
        nodes[inner[0], 3] = 0xFFFFFFFF      # drops a whole subtree
    elif damage == "sphere_among_triangles":
        assert b["tree_kinds"] == 2          # the header promises triangles only, which selects the kernel that tests nothing else
        b["tree"][10, 3] &= ~np.uint32(3)    # kind 0: a sphere
    with pytest.raises(rt.RtError, match=msg):
        rt.validate_blob(blob)


def test_scene_cache_round_trip(tmp_path):
    src = os.path.join(GOLDEN, "final_avec_bonus.scene")
    cfg, blob = _dragon_blob()
    path = str(tmp_path / "dragon.rtcache")
    rt.SceneCache.write(path, cfg, blob, source_path=src)
    cache = rt.SceneCache(path)
    assert np.array_equal(cache.blob(), blob)
    c2 = cache.config()
    assert (c2.width, c2.height, c2.maxdepth, c2.output_file) == (cfg.width, cfg.height, cfg.maxdepth, cfg.output_file)
    assert c2.camera == cfg.camera and c2.ambient == cfg.ambient
    f1, f2 = cfg.frame(), c2.frame()
    assert bytes(f1) == bytes(f2)
    assert cache.is_fresh(src) and not cache.is_fresh(os.path.join(GOLDEN, "test.scene")) and not cache.is_fresh(str(tmp_path / "none"))
    cache.close()


def test_scene_cache_goes_stale_when_the_scene_file_changes(tmp_path):
    scene = tmp_path / "s.scene"
    scene.write_text("size 8 8\nsphere 0 0 5 1\n")
    cfg = rt.ParsedConfigState().load_config_file(str(scene))
    blob, _ = rt.pack_scene_host(cfg.scene())
    path = str(tmp_path / "s.rtcache")
    rt.SceneCache.write(path, cfg, blob, source_path=str(scene))
    assert rt.SceneCache(path).is_fresh(str(scene))
    scene.write_text("size 8 8\nsphere 0 0 5 1.5\n")
    os.utime(scene, ns=(1, 1))
    assert not rt.SceneCache(path).is_fresh(str(scene))


@pytest.mark.parametrize("damage,msg", [("flip", "checksum mismatch"), ("truncate", "truncated"), ("magic", "not a cache file"),
                                        ("short", "not a cache file"), ("version", "another version")])
def test_damaged_cache_files_are_rejected(tmp_path, damage, msg):
    cfg, blob = _dragon_blob()
    path = tmp_path / "d.rtcache"
    rt.SceneCache.write(str(path), cfg, blob)
    raw = bytearray(path.read_bytes())
    if damage == "flip":
        raw[4096 + len(blob) // 2] ^= 1
    elif damage == "truncate":
        raw = raw[:-64]
    elif damage == "magic":
        raw[0] ^= 0xFF
    elif damage == "short":
        raw = raw[:100]
    elif damage == "version":
        raw[8] += 1
    path.write_bytes(bytes(raw))
    with pytest.raises(rt.RtError, match=msg):
        rt.SceneCache(str(path))


def test_missing_cache_file_is_an_error():
    with pytest.raises(rt.RtError, match="os error 2"):
        rt.SceneCache("/nonexistent/x.rtcache")


@pytest.mark.parametrize("height,nshards,tile", [(480, 1, 8), (2160, 8, 8), (2161, 8, 8), (7, 4, 8), (100, 3, 16), (4320, 8, 8)])
def test_shard_rows_partition_the_frame(height, nshards, tile):
    rows = [rt.shard_rows(height, s, nshards, tile) for s in range(nshards)]
    assert sum(rows) == height
    owner = (np.arange(height) // tile) % nshards
    assert rows == [int((owner == s).sum()) for s in range(nshards)]
