"""GPU parity: the HIP render path, called through the C ABI, against the CPU oracle, the
reference's golden images and size-independent properties.

Tolerance (the render is binary32 on both sides, with the C library's powf algorithm on both): at least 99.9 % of
the pixels identical after 8-bit quantisation and no channel off by more than 2/255.  Where the
reference holds a golden image its own criterion is applied too: Image::compare must count zero
pixels (src/raytracer/raytracer.rs:369).
"""
import glob
import os
import sys

import numpy as np
import pytest

import __graft_entry__ as ge
import oracle as O

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

rt = ge.load_package()

MIN_IDENTICAL = 0.999
MAX_CHANNEL_DIFF = 2

SCENES = sorted(glob.glob(os.path.join(GOLDEN, "test_file", "*", "*.test")))


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    rt.init(0)
    yield


def _assert_parity(got, want, min_identical=MIN_IDENTICAL):
    c, _ = O.compare(got, want)
    assert c["identical_frac"] >= min_identical and c["max_diff"] <= MAX_CHANNEL_DIFF, c
    return c


@pytest.mark.parametrize("scene", SCENES, ids=[os.path.basename(s)[:-5] for s in SCENES])
def test_reference_test_scenes(scene):
    cfg = rt.ParsedConfigState().load_config_file(scene)
    tracer = rt.RayTracer.new(cfg)
    img = tracer.render(count_work=True)
    want, st = O.OracleScene.load(scene).render(mode=O.OracleScene.MODE_ALL)
    c = _assert_parity(img.pixels(), want)
    # these scenes have no tree: nothing is culled, so the match should be exact up to powf
    assert c["identical_frac"] >= 0.99999, c
    # the reference's own pass criterion against its golden image
    diff, _ = rt.Image.compare(img, rt.file_to_image(scene[:-5] + ".png"))
    assert diff == 0
    # ray accounting equals the oracle's
    got = tracer.last_stats
    assert (got["primary"], got["shadow"], got["secondary"]) == (st["primary"], st["shadow"], st["secondary"])


def test_dragon_native_size_against_reference_render():
    cfg = rt.ParsedConfigState().load_config_file(os.path.join(GOLDEN, "final_avec_bonus.scene"))
    tracer = rt.RayTracer.new(cfg)
    img = tracer.render(count_work=True)
    gold = rt.file_to_image(os.path.join(GOLDEN, "dragon3.png"))      # the reference's own output of this scene
    c = _assert_parity(img.pixels(), gold.pixels())
    diff, _ = rt.Image.compare(img, gold)
    assert diff == 0
    st = tracer.last_stats
    assert (st["primary"], st["shadow"], st["secondary"]) == (307200, 1098150, 532878)
    print("dragon 640x480 vs dragon3.png:", c, st)


def test_dragon_4k_against_oracle():
    path = os.path.join(GOLDEN, "final_avec_bonus.scene")
    cfg = rt.ParsedConfigState().load_config_file(path)
    cfg.set_size(3840, 2160)
    tracer = rt.RayTracer.new(cfg)
    img = tracer.render()
    want, st = O.OracleScene.load(path).render(width=3840, height=2160, mode=O.OracleScene.MODE_TREE_PRUNED)
    c = _assert_parity(img.pixels(), want)
    print("dragon 3840x2160 vs oracle:", c, "kernel ms", tracer.last_stats["kernel_ms"])


def test_dragon_from_random_cameras_against_oracle():
    """The dragon's tree and rays from viewpoints the reference's scene file does not have: around, above, close to and inside the
    dragon, from below the mirror plane, with odd frame sizes, fields of view from 5 to 120 degrees and maxdepth 1 to 7.  Each
    frame against the oracle's render from the same camera (frame set-up raytracer.rs:41-55 on both sides)."""
    path = os.path.join(GOLDEN, "final_avec_bonus.scene")
    cfg = rt.ParsedConfigState().load_config_file(path)
    rt.RayTracer.new(cfg)
    osc = O.OracleScene.load(path)
    rng = np.random.default_rng(20261112)
    worst = 1.0
    for k in range(10):
        if k < 6:
            pos = rng.uniform(-1, 1, 3) * (25.0, 20.0, 25.0) + (0.0, 12.0, 0.0)
        elif k < 8:
            pos = rng.uniform(-1, 1, 3) * (2.0, 3.0, 2.0) + (0.0, 5.0, -1.0)       # among the dragon's triangles
        else:
            pos = rng.uniform(-1, 1, 3) * (15.0, 15.0, 3.0) + (0.0, 5.0, -9.0)      # behind the plane z = -5
        look = rng.uniform(-1, 1, 3) * (3.0, 3.0, 3.0) + (0.0, 5.0, -1.0)
        up = (0.0, 1.0, 0.0) if k % 3 else tuple(rng.uniform(-1, 1, 3))
        fov = float(rng.choice([5.0, 20.0, 45.0, 90.0, 120.0]))
        w, h, depth = int(rng.integers(40, 400)), int(rng.integers(30, 300)), int(rng.integers(1, 8))
        frame = rt.frame_from_camera(tuple(pos), tuple(look), up, fov, w, h, depth, cfg.ambient)
        got, st = rt.render(frame, count_work=True)
        osc.set_frame(w, h, depth, list(pos) + list(look) + list(up) + [fov], cfg.ambient)
        want, ost = osc.render(mode=O.OracleScene.MODE_TREE_PRUNED)
        c = _assert_parity(got, want)
        assert (st["primary"], st["shadow"], st["secondary"]) == (ost["primary"], ost["shadow"], ost["secondary"]), (k, st, ost)
        worst = min(worst, c["identical_frac"])
    print("dragon from 10 random cameras vs oracle: least identical fraction", worst)


def test_test_scene_full_size_against_oracle():
    path = os.path.join(GOLDEN, "test.scene")
    cfg = rt.ParsedConfigState().load_config_file(path)
    tracer = rt.RayTracer.new(cfg)
    img = tracer.render(count_work=True)
    want, st = O.OracleScene.load(path).render(mode=O.OracleScene.MODE_TREE_PRUNED)
    c = _assert_parity(img.pixels(), want)
    got = tracer.last_stats
    assert (got["primary"], got["shadow"], got["secondary"]) == (st["primary"], st["shadow"], st["secondary"]) == (2073600, 2073600, 0)
    print("test.scene 1920x1080 vs oracle:", c)


def _random_scene(rng, ntri, nsph, npl, nlights, maxdepth, size=(96, 64)):
    lines = [f"size {size[0]} {size[1]}", "camera 0 1 9 0 0 0 0 1 0 50", "ambient .05 .05 .05", f"maxdepth {maxdepth}"]
    for _ in range(nlights):
        p = rng.uniform(-8, 8, 3)
        col = rng.uniform(.2, .6, 3)
        kind = "point" if rng.random() < .6 else "directional"
        lines.append(f"{kind} {p[0]:.4f} {abs(p[1]) + 3:.4f} {p[2] + 8:.4f} {col[0]:.3f} {col[1]:.3f} {col[2]:.3f}")
    nv = 3 * ntri
    lines.append(f"maxverts {nv}")

    def material():
        d, s = rng.uniform(0, .9, 3), rng.uniform(0, .6, 3) * (rng.random() < .7)
        sh = rng.choice([0, 1, 5, 20, 77.5])
        return [f"diffuse {d[0]:.3f} {d[1]:.3f} {d[2]:.3f}", f"specular {s[0]:.3f} {s[1]:.3f} {s[2]:.3f}", f"shininess {sh}"]

    for i in range(ntri):
        c = rng.uniform(-4, 4, 3)
        for _ in range(3):
            v = c + rng.uniform(-1.2, 1.2, 3)
            lines.append(f"vertex {v[0]:.5f} {v[1]:.5f} {v[2]:.5f}")
    for i in range(ntri):
        if i % 7 == 0:
            lines += material()
        lines.append(f"tri {3 * i} {3 * i + 1} {3 * i + 2}")
    for i in range(nsph):
        if i % 3 == 0:
            lines += material()
        c = rng.uniform(-4, 4, 3)
        lines.append(f"sphere {c[0]:.4f} {c[1]:.4f} {c[2]:.4f} {rng.uniform(.2, 1.2):.4f}")
    for i in range(npl):
        lines += material()
        lines.append(f"plane 0 {-4 - i} 0 {rng.uniform(-.2, .2):.3f} 1 {rng.uniform(-.2, .2):.3f}")
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("seed,ntri,nsph,npl,nlights,maxdepth", [
    (1, 0, 0, 0, 1, 1),          # empty scene: black frame
    (2, 40, 10, 1, 2, 4),
    (3, 300, 60, 2, 3, 6),       # tree with mixed shapes, two planes in the always-tested list
    (4, 9, 0, 0, 0, 3),          # just above the small-scene threshold, no lights
    (5, 0, 200, 0, 2, 10),
    (6, 2000, 0, 1, 1, 5),
])
def test_random_scenes_against_exhaustive_oracle(seed, ntri, nsph, npl, nlights, maxdepth):
    text = _random_scene(np.random.default_rng(seed), ntri, nsph, npl, nlights, maxdepth)
    cfg = rt.ParsedConfigState().parse_text(text)
    tracer = rt.RayTracer.new(cfg)
    img = tracer.render(count_work=True)
    want, st = O.OracleScene.parse(text).render(mode=O.OracleScene.MODE_ALL)
    c = _assert_parity(img.pixels(), want)
    got = tracer.last_stats
    assert (got["primary"], got["shadow"], got["secondary"]) == (st["primary"], st["shadow"], st["secondary"])
    if ntri + nsph + npl == 0:
        assert (img.data == 0xFF000000).all()


@pytest.mark.parametrize("gpu_build", [False, True], ids=["host_tree", "gpu_tree"])
@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16])
def test_random_scenes_from_random_cameras_against_exhaustive_oracle(seed, gpu_build):
    """Mixed trees (triangles, spheres; planes in the always-tested list), point and directional lights, seen from a random
    viewpoint that is as likely inside the cloud of shapes as outside it, through both tree builders: every pixel within
    tolerance of the oracle that tests every shape for every ray, and the same number of rays of each kind."""
    rng = np.random.default_rng(seed)
    ntri, nsph, npl = int(rng.integers(0, 400)), int(rng.integers(0, 80)), int(rng.integers(0, 3))
    text = _random_scene(rng, ntri, nsph, npl, int(rng.integers(1, 4)), int(rng.integers(1, 8)), size=(int(rng.integers(30, 130)), int(rng.integers(20, 90))))
    pos, look = rng.uniform(-7, 7, 3), rng.uniform(-2, 2, 3)
    up = rng.uniform(-1, 1, 3)
    camera = "camera " + " ".join(f"{v:.4f}" for v in (*pos, *look, *up)) + f" {rng.choice([10, 35, 60, 100])}"
    text = text.replace("camera 0 1 9 0 0 0 0 1 0 50", camera)
    assert camera in text
    cfg = rt.ParsedConfigState().parse_text(text)
    tracer = rt.RayTracer(cfg, gpu_build=gpu_build)
    img = tracer.render(count_work=True)
    want, st = O.OracleScene.parse(text).render(mode=O.OracleScene.MODE_ALL)
    _assert_parity(img.pixels(), want)
    got = tracer.last_stats
    assert (got["primary"], got["shadow"], got["secondary"]) == (st["primary"], st["shadow"], st["secondary"])


@pytest.mark.parametrize("size", [(1, 1), (7, 5), (8, 8), (9, 17), (333, 3)])
def test_ragged_frame_sizes(size):
    path = os.path.join(GOLDEN, "test_file", "jalon6", "tp64.test")
    cfg = rt.ParsedConfigState().load_config_file(path)
    cfg.set_size(*size)
    img = rt.RayTracer.new(cfg).render()
    want, _ = O.OracleScene.load(path).render(width=size[0], height=size[1])
    assert img.pixels().shape == want.shape
    # tp64's frame is identical to the oracle's at its own size; so it must be at every other (a single pixel may sit on a
    # powf rounding, which is the one case where a channel may differ by one)
    if size == (1, 1):
        _assert_parity(img.pixels(), want, min_identical=0.0)
    else:
        assert np.array_equal(img.pixels(), want)


def _render_both(text, gpu_build=False):
    cfg = rt.ParsedConfigState().parse_text(text)
    img = rt.RayTracer(cfg, gpu_build=gpu_build).render()
    want, _ = O.OracleScene.parse(text).render(mode=O.OracleScene.MODE_ALL)
    return img.pixels(), want


def test_ray_starting_inside_a_sphere_sees_nothing_of_it():
    """shape.rs:80-83: only the near root counts, and a negative one is no hit: from inside a sphere the sphere is
    invisible, so the frame is black (no ambient on a miss, raytracer.rs:199-201) although the camera is enclosed."""
    text = "size 40 30\ncamera 0 0 0 0 0 -1 0 1 0 60\nambient .3 .3 .3\ndiffuse .5 .5 .5\nsphere 0 0 0 5\npoint 0 0 0 .5 .5 .5\n"
    got, want = _render_both(text)
    assert np.array_equal(got, want) and (got == 0xFF000000).all()
    # a second sphere outside the first one is seen through it, lit by a light that sits inside the first sphere:
    # the enclosing sphere does not shadow it either (its near root lies behind the shadow ray's origin)
    text += "sphere 0 0 -9 1\n"
    got, want = _render_both(text)
    assert np.array_equal(got, want) and (got != 0xFF000000).any()


def test_rays_nearly_parallel_to_a_plane_miss_it():
    """shape.rs:113: |normal . direction| < 1e-6 is a miss, whatever the distance would be.  The camera looks along a
    plane from a hair above it; the rays of the middle row of an odd-height frame have direction.y == 0, so normal .
    direction is the plane's tilt: tilted by 5e-7 the row sees nothing although its rays would reach the plane at
    t = 2000, tilted by 2e-6 it sees the plane, tilted away it misses because t < 0."""
    middle = {}
    for tilt in ("0", "5e-7", "-5e-7", "2e-6", "-2e-6"):
        text = (f"size 64 47\ncamera 0 1e-3 0 0 1e-3 -1 0 1 0 1\nambient .2 .1 .1\ndiffuse .5 .6 .7\n"
                f"plane 0 0 0 0 1 {tilt}\npoint 0 5 0 .4 .4 .4\n")
        got, want = _render_both(text)
        assert np.array_equal(got, want), tilt
        assert (got[22] == 0xFF000000).all() and (got[24] != 0xFF000000).all(), tilt      # above: t < 0; below: a hit
        middle[tilt] = int((got[23] != 0xFF000000).sum())
    assert middle == {"0": 0, "5e-7": 0, "-5e-7": 0, "2e-6": 64, "-2e-6": 0}


@pytest.mark.parametrize("gpu_build", [False, True])
def test_coincident_shapes_tie_on_distance_goes_to_the_first_in_the_file(gpu_build):
    """raytracer.rs:116-120 keeps the first of equal minima in candidate order; the candidate order of the bvh crate is
    not part of the reference tree, so this build (and its oracle) let the shape that comes first in the scene file win.
    Pins GPU and oracle to each other on exact ties, with and without a tree, for both tree builders."""
    rng = np.random.default_rng(5)
    head = ["size 96 64", "camera 0 0 9 0 0 0 0 1 0 50", "ambient .1 .1 .1", "maxdepth 2", "point 3 5 9 .7 .7 .7", "maxverts 400"]
    tris = []
    for i in range(40):
        c = rng.uniform(-3, 3, 3)
        tris.append([c + rng.uniform(-1.5, 1.5, 3) for _ in range(3)])
    for t in tris + tris:                               # every triangle twice, same vertices
        for v in t:
            head.append(f"vertex {v[0]:.5f} {v[1]:.5f} {v[2]:.5f}")

    def scene(copies):
        lines = list(head)
        for rep, (dif, spec) in enumerate(((".9 .1 .1", "0 0 0"), (".1 .9 .1", ".5 .5 .5"))[:copies]):
            lines += [f"diffuse {dif}", f"specular {spec}", "shininess 10"]
            for i in range(40):
                base = 3 * (rep * 40 + i)
                lines.append(f"tri {base} {base + 1} {base + 2}")
        for dif in (".2 .2 .8", ".8 .8 .2")[:copies]:                                      # and a sphere twice
            lines += [f"diffuse {dif}", "specular 0 0 0", "sphere 0 0 -2 1.5"]
        return "\n".join(lines) + "\n"

    got, want = _render_both(scene(2), gpu_build=gpu_build)
    c = _assert_parity(got, want)
    assert c["identical_frac"] >= 0.9999, c
    # The copies that come second in the file never win a tie and occlude exactly what the first ones occlude, so the frame
    # is the frame of the scene without them (the first copies do not reflect, so nothing ever looks at a second copy).
    alone, _ = _render_both(scene(1), gpu_build=gpu_build)
    assert np.array_equal(got, alone)


def test_two_tracers_each_render_their_own_scene():
    """The reference's RayTracer owns its Config and Bvh (raytracer.rs:16-38); the library holds one resident scene,
    so a tracer whose scene was replaced by a later one makes it resident again before it renders."""
    a_path = os.path.join(GOLDEN, "test_file", "jalon5", "tp55.test")
    b_path = os.path.join(GOLDEN, "test_file", "jalon6", "tp63.test")
    a = rt.RayTracer.new(rt.ParsedConfigState().load_config_file(a_path))
    b = rt.RayTracer.new(rt.ParsedConfigState().load_config_file(b_path))
    for tracer, path in ((a, a_path), (b, b_path), (a, a_path)):
        diff, _ = rt.Image.compare(tracer.render(), rt.file_to_image(path[:-5] + ".png"))
        assert diff == 0, path
    # a failed upload and a shutdown both leave no scene (or another one) resident: the tracers notice and upload again
    with pytest.raises(rt.RtError):
        rt.upload_scene_blob(np.zeros(1000, dtype=np.uint8))
    diff, _ = rt.Image.compare(b.render(), rt.file_to_image(b_path[:-5] + ".png"))
    assert diff == 0
    rt.shutdown()
    try:
        diff, _ = rt.Image.compare(a.render(), rt.file_to_image(a_path[:-5] + ".png"))
        assert diff == 0
    finally:
        rt.init(0)


def test_kernel_for_trees_of_triangles_equals_the_generic_one():
    """A tree that holds nothing but triangles is rendered by a build of the kernel without the dispatch on the kind of shape
    and with the triangle test as straight-line code (all four rejections of shape.rs:152-176 combined at the end).  Same
    frame as the generic build, pixel for pixel, on the dragon at its native size and from a camera inside the tree."""
    cfg = rt.ParsedConfigState().load_config_file(os.path.join(GOLDEN, "final_avec_bonus.scene"))
    scene = cfg.scene()
    cam = cfg.camera
    frames = [cfg.frame(), rt.frame_from_camera((0.3, 5.0, 0.2), cam["look_at"], cam["up"], 70.0, 800, 600, 6, cfg.ambient)]
    got = {}
    try:
        for which in ("", "generic"):
            rt.shutdown()
            os.environ["RT_RENDER_KERNEL"] = which
            rt.init(0)
            rt.upload_scene(scene)
            got[which] = [rt.render(f, count_work=True) for f in frames]
    finally:
        os.environ.pop("RT_RENDER_KERNEL", None)
        rt.shutdown()
        rt.init(0)
    for (a, sa), (b, sb) in zip(got[""], got["generic"]):
        assert np.array_equal(a, b)
        assert sa["nodes_visited"] == sb["nodes_visited"] and sa["prims_tested"] == sb["prims_tested"] and sa["shadow"] == sb["shadow"]
    golden = rt.file_to_image(os.path.join(GOLDEN, "dragon3.png"))
    assert rt.Image.compare(rt.Image(640, 480, got[""][0][0]), golden)[0] == 0


def test_first_frame_is_ordered_by_where_the_shapes_are_on_screen():
    """The first launch after an upload has no measured costs; its regions go by a guess: twice the tree shapes seen in a
    region plus their mirror images in the specular planes, summed over the five by five regions around it.  Checked
    against the same count in numpy (projection as raytracer.rs:61-66 inverted), for a whole frame and for one of three
    shards; and the order changes no pixel."""
    import struct
    cfg = rt.ParsedConfigState().load_config_file(os.path.join(GOLDEN, "final_avec_bonus.scene"))
    cfg.set_size(1920, 1080)                                  # 240 x 135 tiles: regions of 8 x 8 tiles, 30 x 17 of them
    scene, frame = cfg.scene(), cfg.frame()
    blob, _ = rt.pack_scene_host(scene)
    hdr = struct.unpack_from("<IIQ5Q5I", blob, 0)
    shapes = np.frombuffer(blob, dtype=np.float32, count=hdr[9] * 12, offset=hdr[4]).reshape(-1, 12)
    plane = np.frombuffer(blob, dtype=np.float32, count=12, offset=hdr[5])      # the floor: the one always-tested shape, specular
    assert hdr[10] == 1 and (plane[3:4].view(np.uint32)[0] & 3) == 2
    pos, cv, right, up = (np.array(v[:], dtype=np.float64) for v in (frame.pos, frame.cv, frame.right, frame.up))

    def guess(nshards):
        rows = rt.shard_rows(frame.height, 0, nshards)
        nx, ny = -(-frame.width // 64), -(-rows // 64)
        seen = np.zeros((ny, nx))
        v = shapes[:, 0:3].astype(np.float64)
        p0, n = plane[0:3].astype(np.float64), plane[8:11].astype(np.float64)
        for pts, weight in ((v, 2), (v - 2 * ((v - p0) @ n)[:, None] * n[None, :], 1)):
            q = pts - pos
            depth = q @ cv
            px = (q @ right) / depth / frame.pixel_w * frame.half_w + frame.half_w
            py = frame.half_h - (q @ up) / depth / frame.pixel_h * frame.half_h
            ok = (depth > 0) & (px >= 0) & (px < frame.width) & (py >= 0) & (py < frame.height)
            y = py[ok].astype(np.int64)
            local = (y // 8 // nshards) * 8 + y % 8
            keep = local // 64 < ny
            np.add.at(seen, ((local // 64)[keep], (px[ok].astype(np.int64) // 64)[keep]), weight)
        padded = np.pad(seen, 2)
        return sum(padded[a:a + ny, b:b + nx] for a in range(5) for b in range(5)).ravel()

    rt.upload_scene(scene)
    got, want = rt.tile_costs_guess(frame).astype(np.float64), guess(1)
    assert got.shape == want.shape == (510,)
    # binary32 on the device against binary64 here: a shape on a region's border may fall on either side
    assert abs(got.sum() - want.sum()) <= 1e-3 * want.sum() and np.abs(got - want).max() <= 0.01 * want.max(), (got.sum(), want.sum(), np.abs(got - want).max())
    order_got, order_want = np.argsort(-got, kind="stable"), np.argsort(-want, kind="stable")
    assert len(set(order_got[:100]) & set(order_want[:100])) >= 95
    with pytest.raises(rt.RtError, match="has not been rendered"):           # asking for the guess starts nothing
        rt.tile_costs_export(frame)
    first = rt.render(frame)[0].copy()                        # ordered by the guess
    measured = rt.tile_costs_export(frame)
    assert np.corrcoef(got, measured.astype(np.float64))[0, 1] > 0.5, "the guess says little about the measured costs"
    again = rt.render(frame)[0]                               # ordered by the measured costs
    assert np.array_equal(first, again)
    # one of three shards, as a process per GPU renders it: the guess is made for the rows that shard owns
    torch = pytest.importorskip("torch")
    rows = rt.shard_rows(frame.height, 0, 3)
    buf = torch.zeros((rows, frame.width), dtype=torch.int32, device="cuda")
    hold = rt.after_torch()
    rt.upload_scene(scene)
    rt.render_tiles(frame, 0, 3, buf.data_ptr())
    rt.synchronize()
    del hold
    assert np.array_equal(buf.cpu().numpy().view(np.uint32), np.concatenate([first[t * 8:(t + 1) * 8] for t in range(0, 135, 3)]))


def test_tile_costs_carry_over_to_a_new_upload_and_change_no_pixel(tmp_path):
    """The measured costs of the tile regions can be taken out after a frame and handed back after the scene has been
    uploaded again (a later run of the same scene file): they set the order of the work of the first launch, nothing
    else.  Arbitrary costs must give the same frame too."""
    cfg = rt.ParsedConfigState().load_config_file(os.path.join(GOLDEN, "final_avec_bonus.scene"))
    cfg.set_size(1280, 720)                                   # 160 x 90 tiles: regions of 8 x 8 tiles, 20 x 12 of them
    scene, frame = cfg.scene(), cfg.frame()
    rt.upload_scene(scene)
    assert rt.tile_costs_count(frame) == 240
    with pytest.raises(rt.RtError, match="has not been rendered"):
        rt.tile_costs_export(frame)
    first, _ = rt.render(frame)
    first = first.copy()
    costs = rt.tile_costs_export(frame)
    assert costs.shape == (240,) and costs.max() > 0 and (costs > 0).sum() > 120
    rt.upload_scene(scene)
    with pytest.raises(rt.RtError, match="has not been rendered"):
        rt.tile_costs_export(frame)
    rt.tile_costs_import(frame, costs)
    assert np.array_equal(rt.render(frame)[0], first)
    rt.upload_scene(scene)
    rt.tile_costs_import(frame, np.random.default_rng(3).integers(0, 2 ** 32, 240, dtype=np.uint64).astype(np.uint32))
    assert np.array_equal(rt.render(frame)[0], first)
    with pytest.raises(rt.RtError, match="count differs"):
        rt.tile_costs_import(frame, costs[:-1])
    with pytest.raises(rt.RtError, match="no such GPU"):
        rt.tile_costs_count(frame, gpu=1)
    cfg.set_size(64, 48)                                      # too few tiles to be ordered: nothing to save, nothing to do
    small = cfg.frame()
    assert rt.tile_costs_count(small) == 0
    rt.tile_costs_import(small, np.zeros(0, np.uint32))
    assert rt.tile_costs_export(small).shape == (0,)

    # through the scene cache: the first start leaves the costs beside the cache, the second one picks them up
    cfg.set_size(1280, 720)
    src = tmp_path / "dragon.scene"
    text = open(os.path.join(GOLDEN, "final_avec_bonus.scene")).read().replace("size 640 480", "size 1280 720")
    assert "size 1280 720" in text
    src.write_text(text)
    one = rt.RayTracer.from_scene_file(str(src))
    a = one.render().data.copy()
    tiles = str(src) + ".rtcache.tiles"
    assert not one.from_cache and os.path.getsize(tiles) > 240 * 4
    two = rt.RayTracer.from_scene_file(str(src))
    assert two.from_cache and rt.load_tile_costs(tiles, two.config.frame())
    assert np.array_equal(two.render().data, a)
    os.truncate(tiles, 100)                                   # a damaged file is ignored
    three = rt.RayTracer.from_scene_file(str(src))
    assert not rt.load_tile_costs(tiles, three.config.frame())
    assert np.array_equal(three.render().data, a)


def test_render_is_deterministic_and_instrumentation_does_not_change_pixels():
    cfg = rt.ParsedConfigState().load_config_file(os.path.join(GOLDEN, "final_avec_bonus.scene"))
    tracer = rt.RayTracer.new(cfg)
    a = tracer.render().data.copy()
    b = tracer.render(count_work=True).data.copy()
    c = tracer.render().data.copy()
    assert np.array_equal(a, b) and np.array_equal(a, c)
    st = tracer.last_stats


@pytest.mark.parametrize("nshards,tile_rows", [(2, 8), (8, 8), (3, 16), (5, 7)])
def test_sharded_tiles_reassemble_to_the_full_frame(nshards, tile_rows):
    import torch
    path = os.path.join(GOLDEN, "final_avec_bonus.scene")
    cfg = rt.ParsedConfigState().load_config_file(path)
    cfg.set_size(400, 301)
    tracer = rt.RayTracer.new(cfg)
    full = tracer.render().pixels()
    frame = cfg.frame()
    W, H = frame.width, frame.height
    stride = max(rt.shard_rows(H, s, nshards, tile_rows) for s in range(nshards))
    gathered = torch.zeros((nshards, stride, W), dtype=torch.int32, device="cuda:0")
    rt.after_torch()                 # the fill runs on torch's stream: the library's lane waits for it (an event, no wait on the host)
    for s in range(nshards):
        rt.render_tiles(frame, s, nshards, gathered[s].data_ptr(), tile_rows=tile_rows)
    out = torch.zeros((H, W), dtype=torch.int32, device="cuda:0")
    rt.after_torch()                 # the fill runs on torch's stream: the library's lane waits for it (an event, no wait on the host)
    rt.assemble_tiles(gathered.data_ptr(), stride, nshards, W, H, out.data_ptr(), tile_rows=tile_rows)
    rt.synchronize()
    got = out.cpu().numpy().view(np.uint32)
    assert np.array_equal(got, full)


def test_scene_blob_round_trip_replicates_the_scene():
    import torch
    path = os.path.join(GOLDEN, "test_file", "jalon6", "tp64.test")
    cfg = rt.ParsedConfigState().load_config_file(path)
    tracer = rt.RayTracer.new(cfg)
    a = tracer.render().data.copy()
    ptr, nbytes = rt.scene_blob()
    copy = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
    rt.after_torch()                 # the fill runs on torch's stream: the library's lane waits for it (an event, no wait on the host)
    rt.copy_scene_blob(copy.data_ptr(), nbytes)
    rt.adopt_scene_blob(copy.data_ptr(), nbytes)
    b, _ = rt.render(cfg.frame())
    assert np.array_equal(a, b.reshape(-1))
    with pytest.raises(rt.RtError, match="scene blob"):
        rt.adopt_scene_blob(copy.data_ptr(), nbytes - 256)
    rt.adopt_scene_blob(copy.data_ptr(), nbytes)      # leave a valid scene behind


def test_scene_uploaded_as_a_blob_or_from_a_cache_file_renders_the_same_frame(tmp_path):
    import shutil
    src = str(tmp_path / "dragon.scene")
    shutil.copy(os.path.join(GOLDEN, "final_avec_bonus.scene"), src)
    gold = rt.file_to_image(os.path.join(GOLDEN, "dragon3.png"))
    first = rt.RayTracer.from_scene_file(src)                      # parses, builds, writes the cache
    assert not first.from_cache and os.path.exists(src + ".rtcache")
    a = first.render()
    second = rt.RayTracer.from_scene_file(src)                     # maps the cache, no parse, no build
    assert second.from_cache and second.get_output_path() == "dragon3.png"
    assert second.bvh_info["node_pairs"] == first.bvh_info["node_pairs"] and second.bvh_info["build_ms"] == 0
    b = second.render()
    assert np.array_equal(a.pixels(), b.pixels())
    assert rt.Image.compare(b, gold)[0] == 0 and np.array_equal(b.pixels() & 0xFFFFFF, gold.pixels())
    # host-built blob, uploaded directly
    cfg = rt.ParsedConfigState().load_config_file(src)
    blob, _ = rt.pack_scene_host(cfg.scene())
    rt.upload_scene_blob(blob)
    c, _ = rt.render(cfg.frame())
    assert np.array_equal(c & 0xFFFFFF, gold.pixels())
    assert np.array_equal(rt.download_scene_blob(), blob)
    # a changed scene file invalidates the cache
    with open(src, "a") as f:
        f.write("sphere 0 5 -1 2\n")
    third = rt.RayTracer.from_scene_file(src)
    assert not third.from_cache
    assert not np.array_equal(third.render().pixels() & 0xFFFFFF, gold.pixels())


def test_damaged_blob_is_refused_and_the_resident_scene_survives():
    cfg = rt.ParsedConfigState().load_config_file(os.path.join(GOLDEN, "final_avec_bonus.scene"))
    tracer = rt.RayTracer.new(cfg)
    before = tracer.render().pixels()
    blob, _ = rt.pack_scene_host(cfg.scene())
    bad = blob.copy()
    hdr_nodes = int(np.frombuffer(bad[16:24].tobytes(), np.uint64)[0])
    bad[hdr_nodes + 12:hdr_nodes + 16] = np.frombuffer(np.uint32(0x7FFFFFF0).tobytes(), np.uint8)   # first child -> far outside
    with pytest.raises(rt.RtError, match="scene blob"):
        rt.upload_scene_blob(bad)
    assert np.array_equal(tracer.render().pixels(), before)


def _frames_on_device(rng, w, h):
    import torch
    a = rng.integers(0, 1 << 32, size=(h, w), dtype=np.uint32)
    b = a.copy()
    n = a.size
    # a third untouched, a third nudged by one in some channel (tolerated), a third changed freely
    idx = rng.permutation(n)
    flat = b.reshape(-1)
    for j in idx[: n // 3]:
        ch = 8 * int(rng.integers(0, 3))
        v = (int(flat[j]) >> ch) & 0xFF
        flat[j] = (int(flat[j]) & ~(0xFF << ch)) | (min(v + 1, 255) << ch)
    flat[idx[n // 3: 2 * n // 3]] = rng.integers(0, 1 << 32, size=len(idx[n // 3: 2 * n // 3]), dtype=np.uint32)
    ta = torch.from_numpy(a.view(np.int32)).to("cuda:0")
    tb = torch.from_numpy(b.view(np.int32)).to("cuda:0")
    return a, b, ta, tb


@pytest.mark.parametrize("size", [(1, 1), (3, 1), (64, 64), (333, 7), (1921, 1081)])
def test_device_compare_equals_image_compare(size):
    import torch
    w, h = size
    a, b, ta, tb = _frames_on_device(np.random.default_rng(w * 31 + h), w, h)
    stats, oracle_diff = O.compare(a, b)                                                # the oracle's restatement of mod.rs:54-89
    want_count = stats["ref_diff"]
    td = torch.full((h, w), -1, dtype=torch.int32, device="cuda:0")
    rt.after_torch()                 # the fill runs on torch's stream: the library's lane waits for it (an event, no wait on the host)
    got = rt.compare_frames(ta.data_ptr(), (w, h), tb.data_ptr(), (w, h), td.data_ptr())
    assert got == want_count
    assert np.array_equal(td.cpu().numpy().view(np.uint32), oracle_diff)
    host_count, host_diff = rt.Image.compare(rt.Image(w, h, a), rt.Image(w, h, b))      # and the library's host twin says the same
    assert host_count == want_count and np.array_equal(host_diff.pixels(), oracle_diff)
    assert rt.compare_frames(ta.data_ptr(), (w, h), tb.data_ptr(), (w, h)) == want_count     # without a difference image
    assert rt.compare_frames(ta.data_ptr(), (w, h), ta.data_ptr(), (w, h)) == 0
    with pytest.raises(rt.RtError, match="Images have different dimensions"):
        rt.compare_frames(ta.data_ptr(), (w, h), tb.data_ptr(), (h, w + 1))


def test_device_compare_of_a_render_against_its_golden_image():
    import torch
    cfg = rt.ParsedConfigState().load_config_file(os.path.join(GOLDEN, "final_avec_bonus.scene"))
    rt.RayTracer.new(cfg)
    frame = cfg.frame()
    out = torch.zeros((frame.height, frame.width), dtype=torch.int32, device="cuda:0")
    rt.after_torch()                 # the fill runs on torch's stream: the library's lane waits for it (an event, no wait on the host)
    rt.render_tiles(frame, 0, 1, out.data_ptr())
    gold = rt.file_to_image(os.path.join(GOLDEN, "dragon3.png"))
    tg = torch.from_numpy(gold.pixels().view(np.int32)).to("cuda:0")
    assert rt.compare_frames(out.data_ptr(), (frame.width, frame.height), tg.data_ptr(), (gold.width, gold.height)) == 0


@pytest.mark.parametrize("size", [(1, 1), (5, 3), (640, 480), (1001, 11)])
def test_device_rgb_packing_feeds_the_png_writer(size, tmp_path):
    import torch
    w, h = size
    px = np.random.default_rng(w).integers(0, 1 << 32, size=(h, w), dtype=np.uint32)
    tp = torch.from_numpy(px.view(np.int32)).to("cuda:0")
    rgb = torch.zeros(3 * w * h + 8, dtype=torch.uint8, device="cuda:0")
    rt.after_torch()                 # the fill runs on torch's stream: the library's lane waits for it (an event, no wait on the host)
    rt.pack_rgb(tp.data_ptr(), w, h, rgb.data_ptr())
    rt.synchronize()
    got = rgb.cpu().numpy()
    want = np.stack([(px >> 16) & 0xFF, (px >> 8) & 0xFF, px & 0xFF], axis=-1).astype(np.uint8).reshape(-1)
    assert np.array_equal(got[:3 * w * h], want) and not got[3 * w * h:].any()
    path = str(tmp_path / "f.png")
    rt.save_image_rgb(got[:3 * w * h], w, h, path)
    assert np.array_equal(rt.file_to_image(path).pixels(), px & 0xFFFFFF)


def test_collective_path_on_a_single_rank_group():
    """The broadcast / gather / reassembly path of a multi-GPU render, rehearsed with an RCCL group of
    one rank: frames pipelined two deep must equal the plain render."""
    import importlib
    import torch
    import torch.distributed as dist
    shard_mod = importlib.import_module("raytracer_rust_amd.distributed")
    cfg = rt.ParsedConfigState().load_config_file(os.path.join(GOLDEN, "final_avec_bonus.scene"))
    cfg.set_size(801, 453)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        renderer, info = shard_mod.gpu_renderer(rt, 0, scene=cfg.scene(), always_gather=True)
        want, _ = rt.render(cfg.frame())
        frame = cfg.frame()
        for _ in range(3):
            out = renderer.render_frame(frame)
        renderer.flush()
        rt.synchronize()
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint32), want)
        small = cfg.frame()
        small.width, small.height = 160, 90          # another frame size through the same renderer
        want_small, _ = rt.render(small)
        out = renderer.render_frame(small)
        renderer.flush()
        rt.synchronize()
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint32), want_small)
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------ tree built on the GPU
def test_frames_through_the_in_library_multi_gpu_path_on_one_gpu():
    """rt_init_multi(1): one process, one RCCL communicator of size one.  Uploads go through the broadcast, frames through
    tile rendering into the gather buffer and the reassembly kernel (the send/receive pairs need a second GPU), and must
    come out as they do on the plain single-GPU path.  More than one GPU was never available to these tests."""
    import torch
    dragon = os.path.join(GOLDEN, "final_avec_bonus.scene")
    tp64 = os.path.join(GOLDEN, "test_file", "jalon6", "tp64.test")
    cfg = rt.ParsedConfigState().load_config_file(dragon)
    single = rt.RayTracer.new(cfg)
    single.render(count_work=True)
    want_stats = single.last_stats
    rt.shutdown()
    try:
        rt.init_multi(1)
        assert rt.device_count() == 1
        with pytest.raises(rt.RtError, match="already initialised"):
            rt.init_multi(2)
        tracer = rt.RayTracer.new(cfg)
        img = tracer.render(count_work=True)
        gold = rt.file_to_image(os.path.join(GOLDEN, "dragon3.png"))
        assert np.array_equal(img.pixels() & 0xFFFFFF, gold.pixels())
        got = tracer.last_stats
        for k in ("primary", "shadow", "secondary", "nodes_visited", "prims_tested"):
            assert got[k] == want_stats[k], k
        # the frame left on the device, two lanes in flight, compared there
        frame = cfg.frame()
        tg = torch.from_numpy(gold.pixels().view(np.int32)).to("cuda:0")
        ptrs = []
        for lane in (0, 1, 0, 1):
            rt.select_lane(lane)
            ptr, _ = rt.render_frame(frame)
            ptrs.append(ptr)
        for lane in (0, 1):
            rt.select_lane(lane)
            rt.synchronize()
            assert rt.compare_frames(ptrs[lane], (frame.width, frame.height), tg.data_ptr(), (gold.width, gold.height)) == 0
        assert ptrs[0] == ptrs[2] and ptrs[0] != ptrs[1]
        rt.select_lane(0)
        # a small scene without a tree, at a height that is no multiple of the tile rows
        cfg2 = rt.ParsedConfigState().load_config_file(tp64)
        cfg2.set_size(97, 61)
        img2 = rt.RayTracer.new(cfg2).render()
        want2, _ = O.OracleScene.load(tp64).render(width=97, height=61)
        assert np.array_equal(img2.pixels(), want2)
    finally:
        rt.shutdown()
        rt.init(0)


def _assert_gpu_built_scene_is_sound(info):
    blob = rt.download_scene_blob()
    rt.validate_blob(blob)                       # every reference in range, a proper tree, depth as declared
    assert 0 < info["max_depth"] <= 63 and 1 <= info["max_leaf"] <= 8 and info["leaves"] == info["node_pairs"] + 1


def test_gpu_built_tree_renders_the_reference_frames():
    cfg = rt.ParsedConfigState().load_config_file(os.path.join(GOLDEN, "final_avec_bonus.scene"))
    tracer = rt.RayTracer(cfg, gpu_build=True)
    _assert_gpu_built_scene_is_sound(tracer.bvh_info)
    assert tracer.bvh_info["bvh_prims"] == 100000 and tracer.bvh_info["list_prims"] == 1
    img = tracer.render(count_work=True)
    gold = rt.file_to_image(os.path.join(GOLDEN, "dragon3.png"))
    assert rt.Image.compare(img, gold)[0] == 0 and np.array_equal(img.pixels() & 0xFFFFFF, gold.pixels())
    st = tracer.last_stats
    assert (st["primary"], st["shadow"], st["secondary"]) == (307200, 1098150, 532878)
    # test.scene: 266 spheres, one of them (the ground, radius 10000) kept out of the tree
    path = os.path.join(GOLDEN, "test.scene")
    cfg = rt.ParsedConfigState().load_config_file(path)
    host = rt.RayTracer(cfg).render().pixels()
    tracer = rt.RayTracer(cfg, gpu_build=True)
    _assert_gpu_built_scene_is_sound(tracer.bvh_info)
    assert np.array_equal(tracer.render().pixels(), host)


def test_gpu_built_blob_is_the_same_every_time():
    """Atomics only ever add integers or take minima and maxima, and node numbers come from prefix sums:
    two builds of one scene give the same bytes (what a scene cache file is keyed on)."""
    cfg = rt.ParsedConfigState().load_config_file(os.path.join(GOLDEN, "final_avec_bonus.scene"))
    blobs = []
    for _ in range(3):
        rt.upload_scene(cfg.scene(), gpu_build=True)
        blobs.append(rt.download_scene_blob().copy())
    assert np.array_equal(blobs[0], blobs[1]) and np.array_equal(blobs[0], blobs[2])


@pytest.mark.parametrize("seed,ntri,nsph,npl,nlights,maxdepth", [(11, 300, 40, 1, 2, 4), (12, 2000, 0, 0, 1, 3), (13, 9, 0, 2, 1, 2), (14, 0, 150, 1, 3, 5)])
def test_gpu_built_tree_against_exhaustive_oracle(seed, ntri, nsph, npl, nlights, maxdepth):
    text = _random_scene(np.random.default_rng(seed), ntri, nsph, npl, nlights, maxdepth)
    cfg = rt.ParsedConfigState().parse_text(text)
    tracer = rt.RayTracer(cfg, gpu_build=True)
    _assert_gpu_built_scene_is_sound(tracer.bvh_info)
    img = tracer.render(count_work=True)
    want, st = O.OracleScene.parse(text).render(mode=O.OracleScene.MODE_ALL)
    _assert_parity(img.pixels(), want)
    got = tracer.last_stats
    assert (got["primary"], got["shadow"], got["secondary"]) == (st["primary"], st["shadow"], st["secondary"])


def test_gpu_tree_build_on_degenerate_and_adversarial_input():
    frame = rt.frame_from_camera((0, 0, 9), (0, 0, 0), (0, 1, 0), 50, 64, 48, 2, (0.1, 0.1, 0.1))
    light = [(0.0, 3.0, 4.0, 9.0, 0.5, 0.5, 0.5)]
    mat = (0.5, 0.5, 0.5, 0.2, 0.2, 0.2, 10.0)
    tri = np.float32([[-1, -1, 0, 1, -1, 0, 0, 1, 0]])
    cases = {
        "identical": np.repeat(tri, 500, axis=0),                                   # a large node that no split separates: halved by position
        "collinear centres": tri + np.float32([[1, 0, 0] * 3]) * np.arange(40, dtype=np.float32)[:, None] * 0.01,
        "points": np.repeat(np.float32([[0.5, 0.5, 0.5] * 3]), 20, axis=0),         # zero-extent boxes
    }
    # points at powers of two on the three axes, over 21 octaves, plus many copies of the smallest: most of
    # every node's shapes fall into its first bin, level after level
    pts = np.zeros((63, 3))
    for b in range(63):
        pts[b, b % 3] = 1e-5 * 2.0 ** (b // 3)
    chain = (tri[0][None, :] * 1e-9 + np.tile(pts, (1, 3))).astype(np.float32)
    cases["deep"] = np.concatenate([chain, np.repeat(chain[:1], 4000, axis=0)])
    for name, t in cases.items():
        scene = rt.SceneArrays(tri=t, tri_mat=mat, lights=light)
        info_h = rt.upload_scene(scene.c)
        host, _ = rt.render(frame)
        info_g = rt.upload_scene(scene.c, gpu_build=True)
        rt.validate_blob(rt.download_scene_blob())
        assert info_g["max_depth"] <= 63, name
        got, _ = rt.render(frame)
        assert np.array_equal(got, host), name
        assert info_g["bvh_prims"] == info_h["bvh_prims"] and info_g["leaves"] >= 1, name
    # non-finite coordinates are refused by both builders with the same message
    bad = tri.copy(); bad = np.repeat(bad, 20, axis=0); bad[3, 4] = np.inf
    for gpu in (False, True):
        with pytest.raises(rt.RtError, match="non-finite shape coordinates"):
            rt.upload_scene(rt.SceneArrays(tri=bad, tri_mat=mat, lights=light).c, gpu_build=gpu)


def test_gpu_built_tree_of_the_large_mesh_matches_the_host_built_one():
    import importlib
    syn = importlib.import_module("raytracer_rust_amd.synthetic")
    scene = syn.product_scene(rt, n=708)                      # about a million triangles
    frame = syn.product_frame(rt, 1920, 1080)
    rt.upload_scene(scene.c)
    host, _ = rt.render(frame)
    info = rt.upload_scene(scene.c, gpu_build=True)
    _assert_gpu_built_scene_is_sound(info)
    got, _ = rt.render(frame)
    assert np.array_equal(got, host)
    print("1M-triangle mesh, GPU tree:", info)


@pytest.mark.parametrize("text", [
    "size 33 17\n",                                                                    # nothing at all: every ray misses
    "size 33 17\npoint 0 5 5 1 1 1\n",                                                 # lights but no shapes
    "size 33 17\nmaxdepth 0\ndiffuse .5 .5 .5\nspecular .5 .5 .5\nplane 0 -1 0 0 1 0\npoint 0 5 5 1 1 1\n",   # a plane only, maxdepth 0
    "size 33 17\nmaxdepth 7\ndiffuse .2 .2 .2\nspecular .9 .9 .9\nshininess 1\nplane 0 -1 0 0 1 0\nplane 0 3 0 0 -1 0\nsphere 0 0 4 1\n"
    "directional 0 1 1 .6 .6 .6\npoint 0 0 0 .4 .4 .4\n",                               # facing mirrors, shininess 1, both light kinds
], ids=["empty", "lights-only", "plane-maxdepth0", "mirrors"])
def test_tiny_and_empty_scenes(text):
    cfg = rt.ParsedConfigState().parse_text(text)
    for gpu_build in (False, True):
        tracer = rt.RayTracer(cfg, gpu_build=gpu_build)
        img = tracer.render(count_work=True)
        want, st = O.OracleScene.parse(text).render(mode=O.OracleScene.MODE_ALL)
        c = _assert_parity(img.pixels(), want)
        assert c["identical_frac"] == 1.0, c
        got = tracer.last_stats
        assert (got["primary"], got["shadow"], got["secondary"]) == (st["primary"], st["shadow"], st["secondary"])


def test_lanes_render_different_frames_at_once():
    """Launches on different lanes overlap on the GPU and must not share work buffers: different frames
    enqueued back to back on all lanes, several times over, equal their renders one at a time."""
    import torch
    cfg = rt.ParsedConfigState().load_config_file(os.path.join(GOLDEN, "final_avec_bonus.scene"))
    cfg.set_size(1283, 721)
    rt.RayTracer(cfg)
    frames = [cfg.frame()]
    for k in range(1, rt.LANES):
        frames.append(rt.frame_from_camera((9.0 - 3.0 * k, 13.0, 12.0), (0.0, 5.0, -1.0), (0.0, 1.0, 0.0), 50.0, 1283, 721, 2 + k, (0.0, 0.0, 0.0)))
    rt.select_lane(0)
    want = [rt.render(f)[0] for f in frames]
    assert all(not np.array_equal(want[0], w) for w in want[1:])
    bufs = [torch.zeros((721, 1283), dtype=torch.int32, device="cuda:0") for _ in range(rt.LANES)]
    try:
        for turn in range(4):
            for lane in range(rt.LANES):
                rt.select_lane(lane)
                if turn == 0:
                    rt.after_torch()     # the fills run on torch's stream: every lane waits for them (an event, no wait on the host)
                rt.render_tiles(frames[lane], 0, 1, bufs[lane].data_ptr())
        for lane in range(rt.LANES):
            rt.select_lane(lane)
            rt.before_torch()            # and torch's stream, which copies the frames back below, waits for every lane
        for lane in range(rt.LANES):
            assert np.array_equal(bufs[lane].cpu().numpy().view(np.uint32), want[lane]), lane
        with pytest.raises(rt.RtError, match="lanes are 0 .. 3"):
            rt.select_lane(rt.LANES)
    finally:
        rt.select_lane(0)


def test_errors_are_reported_not_fatal():
    cfg = rt.ParsedConfigState().load_config_file(os.path.join(GOLDEN, "test_file", "jalon3", "tp31.test"))
    rt.RayTracer.new(cfg)
    f = cfg.frame()
    f.width = 0
    with pytest.raises(rt.RtError, match="greater than zero"):
        rt.render(f)
    with pytest.raises(rt.RtError, match="bad shard"):
        rt.render_tiles(cfg.frame(), 3, 2, 0)


def test_large_procedural_mesh_8k():
    """About 10 M triangles at 7680x4320 with 4 point lights and maxdepth 3 (synthetic.py).
    The whole frame is far beyond what the oracle renders in seconds, so parity is checked on
    every 8th row tile against the oracle (4.1 M pixels) and the rest through determinism and ray accounting; three other viewpoints
    as whole small frames against the oracle; and the GPU-built tree of the same shapes must give the same 8K frame as the host-built one."""
    import importlib
    syn = importlib.import_module("raytracer_rust_amd.synthetic")
    from raytracer_rust_amd.distributed import shard_row_indices
    W, H = 7680, 4320
    scene = syn.product_scene(rt)
    assert scene.c.ntri == 2 * 2236 * 2236
    info = rt.upload_scene(scene.c)
    assert info["bvh_prims"] == scene.c.ntri and info["list_prims"] == 1 and info["max_depth"] <= 64
    frame = syn.product_frame(rt, W, H)
    a, st = rt.render(frame, count_work=True)
    b, st2 = rt.render(frame)
    assert np.array_equal(a, b)
    assert st["primary"] == W * H and st["shadow"] % 4 == 0 and st["shadow"] // 4 >= st["secondary"]
    osc = syn.oracle_scene(O, W, H)
    want, ost = osc.render_tiles(5, 8, 8, width=W, height=H, mode=O.OracleScene.MODE_TREE_PRUNED)
    rows = shard_row_indices(H, 5, 8, 8)
    c = _assert_parity(a[rows], want)
    print("10M-triangle mesh 7680x4320, rows of every 8th tile vs oracle:", c, "kernel ms", st2["kernel_ms"], "bvh", info,
          "rays", {k: st[k] for k in ("primary", "shadow", "secondary", "nodes_visited", "prims_tested")})
    # other viewpoints of the same tree, whole (small) frames against the oracle: low over the surface, straight down, from below it
    rng = np.random.default_rng(7)
    for k, (pos, look, up) in enumerate([((80.0, 5.0, 3.0), (0.0, 0.0, 0.0), (0.0, 0.0, 1.0)), ((3.0, -2.0, 90.0), (3.0, -2.0, 0.0), (0.0, 1.0, 0.0)),
                                         ((-40.0, 30.0, -25.0), (0.0, 0.0, 0.0), (0.0, 0.0, 1.0))]):
        w, h, depth, fov = int(rng.integers(100, 500)), int(rng.integers(100, 400)), 1 + k, (30.0, 75.0, 110.0)[k]
        f = rt.frame_from_camera(pos, look, up, fov, w, h, depth, syn.AMBIENT)
        got, gst = rt.render(f, count_work=True)
        osc.set_frame(w, h, depth, list(pos) + list(look) + list(up) + [fov], syn.AMBIENT)
        want, ost = osc.render(mode=O.OracleScene.MODE_TREE_PRUNED)
        _assert_parity(got, want)
        assert (gst["primary"], gst["shadow"], gst["secondary"]) == (ost["primary"], ost["shadow"], ost["secondary"]), (k, gst, ost)
    # the tree the GPU builds of the same shapes (what bench.py --config mesh renders with) gives the same 8K frame, pixel for pixel
    info_gpu = rt.upload_scene(scene.c, gpu_build=True)
    assert info_gpu["bvh_prims"] == scene.c.ntri
    g, _ = rt.render(frame)
    assert np.array_equal(g, a)


def test_process_exits_cleanly_when_torch_is_imported_after_the_library():
    """A process that binds the library to the GPU first and imports torch later used to abort at exit (two sets of ROCm
    libraries loaded in the wrong order); the package now brings torch in before its own library."""
    import subprocess
    code = ("import os, sys; sys.path.insert(0, %r); import __graft_entry__ as ge; rt = ge.load_package(); rt.init(0); import torch; "
            "cfg = rt.ParsedConfigState().load_config_file(%r); rt.upload_scene(cfg.scene()); "
            "t = torch.zeros(16, device='cuda:0'); torch.cuda.synchronize(); print('done')") % (os.path.dirname(os.path.abspath(ge.__file__)), os.path.join(GOLDEN, "test_file", "jalon6", "tp64.test"))
    done = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert done.returncode == 0 and "done" in done.stdout, (done.returncode, done.stderr[-500:])
