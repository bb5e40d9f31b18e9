"""First frame of a newly resident scene (4K dragon): variants of the guess at the tile order (made in numpy and imported) against
the library's own guess, index order being RT_FIRST_ORDER=raster, and against measured costs; for two cameras."""
import os, sys, struct
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd())
import __graft_entry__ as ge, numpy as np
rt = ge.load_package(); rt.init(0)
root = os.path.dirname(ge.__file__)
cfg = rt.ParsedConfigState().load_config_file(os.path.join(root, "tests", "golden", "final_avec_bonus.scene"))
cfg.set_size(3840, 2160)
blob, info = rt.pack_scene_host(cfg.scene())
hdr = struct.unpack_from("<IIQ5Q5I", blob, 0)
off_tree, off_list = hdr[4], hdr[5]
n_tree, n_list = hdr[9], hdr[10]
shapes = np.frombuffer(blob, dtype=np.float32, count=n_tree * 12, offset=off_tree).reshape(-1, 12)
lst = np.frombuffer(blob, dtype=np.float32, count=n_list * 12, offset=off_list).reshape(-1, 12)
NX, NY, SIDE = 60, 34, 64
cam = cfg.camera
frames = {"file": cfg.frame()}
p = np.array(cam["position"]); l = np.array(cam["look_at"])
frames["moved"] = rt.frame_from_camera(l + (p - l) * 0.8 + np.array([0.7, 0.3, 0.2]), l, cam["up"], cam["fov"], 3840, 2160, cfg.maxdepth, cfg.ambient)
def box(g, r):
    p = np.pad(g, r); return np.sum([p[a:a + NY, b:b + NX] for a in range(2 * r + 1) for b in range(2 * r + 1)], axis=0)
def tent(g):
    return box(box(g, 1), 1)
V0 = shapes[:, 0:3].astype(np.float64)
VC = V0 + (shapes[:, 4:7].astype(np.float64) + shapes[:, 8:11]) / 3
pl = lst[0]; p0 = pl[0:3].astype(float); n = pl[8:11].astype(float)
mirror = lambda V: V - 2 * ((V - p0) @ n)[:, None] * n[None, :]
for fname, f in frames.items():
    pos = np.array(f.pos[:]); cv = np.array(f.cv[:]); right = np.array(f.right[:]); up = np.array(f.up[:])
    pw, ph, w2, h2 = f.pixel_w, f.pixel_h, f.half_w, f.half_h
    def splat(P):
        q = P - pos
        z = q @ cv; x = (q @ right) / z; y = (q @ up) / z
        px = x / pw * w2 + w2; py = h2 - y / ph * h2
        ok = (z > 0) & (px >= 0) & (px < f.width) & (py >= 0) & (py < f.height)
        g = np.zeros((NY, NX)); np.add.at(g, ((py[ok] // SIDE).astype(int), (px[ok] // SIDE).astype(int)), 1); return g
    variants = {"library": None}
    for aname, V in (("v0", V0), ("cen", VC)):
        d, m = splat(V), splat(mirror(V))
        for wname, g in (("1:1", d + m), ("2:1", 2 * d + m)):
            variants[f"{aname} {wname} box1"] = box(g, 1)
            variants[f"{aname} {wname} box2"] = box(g, 2)
            variants[f"{aname} {wname} tent"] = tent(g)
        variants[f"{aname} binary"] = (box(d + m, 1) > 0).astype(float)
    rt.upload_scene(cfg.scene()); rt.render_frame(f, want_stats=True); measured = rt.tile_costs_export(f).astype(np.float64)
    variants["measured"] = measured
    res = {k: [] for k in variants}
    for rep in range(3):
        for name, costs in variants.items():
            rt.upload_scene(cfg.scene())
            if costs is not None:
                c = np.asarray(costs, dtype=np.float64).ravel()
                rt.tile_costs_import(f, (c / max(c.max(), 1e-30) * 1e8).astype(np.uint32))
            ts = [rt.render_frame(f, want_stats=True)[1]["kernel_ms"] for _ in range(3)]
            res[name].append((ts[0], ts[2]))
    for name, r in res.items():
        print("%-6s %-16s first %s   third %.3f" % (fname, name, " ".join("%.3f" % a for a, _ in r), np.mean([b for _, b in r])), flush=True)
