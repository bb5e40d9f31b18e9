#!/usr/bin/env python3
"""Headline benchmark: the dragon scene (final_avec_bonus.scene, 100 000 triangles + a mirror
plane, 2 point lights, maxdepth 5) rendered at 3840x2160 on N MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)
  python bench.py --config mesh     the 10 M-triangle procedural mesh at 7680x4320 (secondary workload, one GPU)

One step = one frame, left in device memory (the frame rendered into host memory, what
RayTracer::render returns, is timed separately as rt_render_ms).  With N GPUs the frame is cut into
interleaved 8-row tiles (tile t -> GPU t % N) with the scene replicated by one RCCL broadcast
(outside the timed region, like the reference's BVH build, src/main.rs:11-15), and each frame ends
with one RCCL gather of the compact row buffers on GPU 0, which restores frame order.  Two ways to
drive them:
  --multi library (default)  rank 0 drives all N GPUs through the C ABI (rt_init_multi: RCCL inside the
                             library, what a single-process host like the reference's main gets); the
                             other ranks only take part in the barriers.  Rank 0 first sends a child process
                             through that path (communicators, scene broadcast, one gathered frame, compared with the
                             frame of one GPU); if the child fails, hangs (it is killed after 7 minutes) or gets a wrong
                             frame, or the same set-up then fails in rank 0 itself, the line says so
                             (`config.library_path_error`) and the ranks path below is used instead.
  --multi ranks              one process per GPU, torch.distributed ("nccl" = RCCL) for broadcast and gather.
Frames are pipelined over the library's lanes (two deep up to four GPUs, four deep beyond): the gather of
one frame overlaps the kernels of the next.  The total work is fixed as N grows: "scaling": "strong".

With more than one GPU the frame that comes out of the gather is first compared, pixel for pixel, with the same
frame rendered by GPU 0 alone (`multi_gpu_frame_equal`, both checksums in the line; a difference voids the line and
the exit status is 4), and every step that can block inside a collective runs under a watchdog that ends the process
with exit status 3 and a message naming the step.

Rank 0 prints ONE JSON line.  `value` counts primary + shadow rays per second (the wording of
the metric); reflection rays are traced too and reported in `mrays_all_per_s`.  Ray counts are
deterministic and come from an instrumented render outside the timed region.
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import __graft_entry__ as ge  # noqa: E402

SCENE = os.path.join(ROOT, "tests", "golden", "final_avec_bonus.scene")
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak
L2_PEAK_GBS = 34500.0      # aggregate L2 bandwidth of MI355X (the dragon's 7 MB scene is cache resident: L2 is the memory-side bound)
SIMDS, XCDS = 1024, 8


class Watchdog:
    """Ends the process (exit status 3, a message on stderr saying which step hung) when a step that can block inside a
    collective does not finish in time: a hung ncclGroupEnd must end the run with a diagnosis, not with silence."""

    def __init__(self):
        self._timer = None

    def arm(self, seconds, what):
        self.disarm()
        self._timer = threading.Timer(seconds, self._fire, (seconds, what))
        self._timer.daemon = True
        self._timer.start()

    def disarm(self):
        if self._timer is not None:
            self._timer.cancel()
            self._timer = None

    @staticmethod
    def _fire(seconds, what):
        sys.stderr.write(f"bench.py: watchdog: {what} did not finish within {seconds:.0f} s; giving up (exit status 3)\n")
        sys.stderr.flush()
        os._exit(3)


def frame_checksum(frame_host):
    return int(np.bitwise_xor.reduce(frame_host.reshape(-1).astype(np.uint64) * np.arange(1, frame_host.size + 1, dtype=np.uint64)))


def counters_commit(entry):
    """The commit the counter passes of profiles/traffic.json were taken at (the canned figures of the line come from that file)."""
    if entry.get("profiled_at_commit"):
        return entry["profiled_at_commit"]
    try:
        return subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%h", "--", "profiles/traffic.json"], capture_output=True, text=True,
                              timeout=10).stdout.strip() or "unknown"
    except (OSError, subprocess.SubprocessError):
        return "unknown"


def cpu_baseline_dragon(oracle, width, height):
    """The oracle timed on this host: the reference's algorithm (collect every candidate of the
    tree, test them all; raytracer.rs:110-159) and the pruned variant, on whole frames of the same
    workload, repeated until about ten seconds of CPU work have been timed."""
    sc = oracle.OracleScene.load(SCENE)
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or (os.cpu_count() or 1))
    out = {}
    for name, mode, budget in (("reference_faithful", oracle.OracleScene.MODE_TREE_COLLECT, 10.0), ("pruned", oracle.OracleScene.MODE_TREE_PRUNED, 3.0)):
        sc.render_tiles(0, 64, width=width, height=height, mode=mode, threads=threads)      # builds the tree, warms up
        sec, rays_ps, rays_all, frames = 0.0, 0, 0, 0
        while sec < budget and frames < 8:
            _, st = sc.render_tiles(0, 1, width=width, height=height, mode=mode, threads=threads)
            sec += st["render_ms"] / 1e3
            rays_ps += st["primary"] + st["shadow"]; rays_all += st["primary"] + st["shadow"] + st["secondary"]; frames += 1
        out[name] = {"mrays_primary_shadow_per_s": rays_ps / sec / 1e6, "mrays_all_per_s": rays_all / sec / 1e6, "seconds": sec, "frames": frames}
    return {"value": out["reference_faithful"]["mrays_primary_shadow_per_s"], "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{out['reference_faithful']['frames']} whole {width}x{height} dragon frames ({out['reference_faithful']['seconds']:.1f} s), C++ restatement of the "
                      f"reference, bit-exact vs dragon3.png; candidate collection as raytracer.rs:110-159",
            "all_rays_value": out["reference_faithful"]["mrays_all_per_s"], "seconds": out["reference_faithful"]["seconds"],
            "pruned_variant": out["pruned"]}


def cpu_baseline_mesh(oracle, syn, width, height):
    """The same for the large mesh, on every 4th row tile of the frame (a quarter of the frame, some 15 s; a whole frame takes a minute)."""
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or (os.cpu_count() or 1))
    sc = syn.oracle_scene(oracle, width, height)
    sc.render_tiles(0, 4096, width=width, height=height, mode=oracle.OracleScene.MODE_TREE_COLLECT, threads=threads)      # builds the tree
    _, st = sc.render_tiles(0, 4, width=width, height=height, mode=oracle.OracleScene.MODE_TREE_COLLECT, threads=threads)
    sec = st["render_ms"] / 1e3
    return {"value": (st["primary"] + st["shadow"]) / sec / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"every 4th 8-row tile of one {width}x{height} frame ({sec:.1f} s), C++ restatement of the reference with candidate collection as raytracer.rs:110-159",
            "all_rays_value": (st["primary"] + st["shadow"] + st["secondary"]) / sec / 1e6, "seconds": sec}


def load_counters(config):
    """Hardware-counter figures per launch from the committed rocprofv3 passes (profiles/traffic.json), for this workload."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            return json.load(fh).get(config) or {}
    except (OSError, ValueError):
        return {}


def make_workload(rt, syn, config, W, H, with_scene):
    """(frame, scene or None, what keeps the scene's arrays alive, workload name, data description) of a configuration."""
    if config == "dragon":
        cfg = rt.ParsedConfigState().load_config_file(SCENE)
        cfg.set_size(W, H)
        workload = f"final_avec_bonus.scene (dragon, 100000 triangles + plane, 2 point lights, maxdepth 5) at {W}x{H}"
        data = ("reference scene file final_avec_bonus.scene with the size line overridden to 3840 2160" if (W, H) == (3840, 2160)
                else f"reference scene file final_avec_bonus.scene at {W}x{H}")
        return cfg.frame(), (cfg.scene() if with_scene else None), cfg, workload, data
    scene_arrays = syn.product_scene(rt) if with_scene else None
    workload = f"procedural height-field mesh, 9998592 triangles + plane, 4 point lights, maxdepth 3, at {W}x{H}"
    return (syn.product_frame(rt, W, H), (scene_arrays.c if with_scene else None), scene_arrays, workload,
            "synthetic mesh generated in memory (raytracer-rust_amd/synthetic.py)")


def device_frame_to_host(ptr, H, W):
    """A frame in device memory of the first GPU (raw pointer) copied to the host."""
    import ctypes
    import torch
    host = np.empty((H, W), dtype=np.uint32)
    bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")      # the HIP runtime already loaded into the process (torch's copy)
    hip = ctypes.CDLL(bundled if os.path.exists(bundled) else "libamdhip64.so")
    assert hip.hipMemcpy(ctypes.c_void_p(host.ctypes.data), ctypes.c_void_p(ptr), ctypes.c_size_t(host.nbytes), 2) == 0      # device to host
    return host


def library_rehearsal(args):
    """`--library-rehearsal`: what rank 0 runs as a child process before it commits itself to the library path on N GPUs.
    Communicators, scene broadcast, one frame through the gather, compared with the frame of the first GPU alone; one JSON
    object on stdout says how it went.  A hang in here costs the parent a time-out and the child its life, not the run."""
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    trouble = os.environ.get("RT_BENCH_TEST_REHEARSAL")       # for testing the parent's handling: hang | error | differ
    try:
        if trouble == "hang":
            time.sleep(1e6)
        import torch
        rt = ge.load_package()
        syn = importlib.import_module("raytracer_rust_amd.synthetic")
        W, H = (args.width or (3840 if args.config == "dragon" else 7680)), (args.height or (2160 if args.config == "dragon" else 4320))
        frame, scene, _keep, _, _ = make_workload(rt, syn, args.config, W, H, with_scene=True)
        rt.init_multi(args.gpus)
        if trouble == "error":
            raise RuntimeError("asked to fail (RT_BENCH_TEST_REHEARSAL=error)")
        rt.upload_scene(scene, gpu_build=args.config == "mesh")
        ptr, _ = rt.render_frame(frame, want_stats=True)
        gathered = device_frame_to_host(ptr, H, W)
        ref = torch.zeros((H, W), dtype=torch.int32, device=torch.device("cuda", 0))
        rt.after_torch()
        rt.render_tiles(frame, 0, 1, ref.data_ptr(), want_stats=True)
        alone = ref.cpu().numpy().view(np.uint32)
        if trouble == "differ":
            alone[H // 2, W // 2] ^= 1
        equal = bool(np.array_equal(gathered, alone))
        rt.shutdown()
        report = {"ok": equal, "error": None if equal else "the frame gathered from the GPUs differs from the frame of one GPU",
                  "frame_checksum": frame_checksum(gathered), "frame_checksum_one_gpu": frame_checksum(alone)}
    except Exception as e:      # noqa: BLE001  (the parent reads it from the report)
        report = {"ok": False, "error": f"{type(e).__name__}: {e}"}
    print(json.dumps(report), file=real_stdout, flush=True)
    return 0


def rehearse_library_in_child(args, world, W, H):
    """None if a child process got a correct frame through the library path on `world` GPUs, else what went wrong there."""
    seconds = float(os.environ.get("RT_BENCH_REHEARSAL_SECONDS", "420"))
    launcher_keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "ROLE_NAME",
                     "MASTER_ADDR", "MASTER_PORT")
    env = {k: v for k, v in os.environ.items() if k not in launcher_keys and not k.startswith("TORCHELASTIC_")}
    cmd = [sys.executable, "-u", os.path.abspath(__file__), "--library-rehearsal", "--gpus", str(world), "--config", args.config, "--width", str(W), "--height", str(H)]
    try:
        child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    except OSError as e:
        return f"rehearsal in a child process: could not start it ({e})"
    try:
        out, _ = child.communicate(timeout=seconds)
    except subprocess.TimeoutExpired:
        child.kill()                    # this child, by its process id
        child.communicate()
        return f"rehearsal in a child process: no answer within {seconds:.0f} s (hung; the child was killed)"
    lines = [l for l in out.splitlines() if l.startswith("{")]
    try:
        report = json.loads(lines[-1])
    except (IndexError, ValueError):
        return f"rehearsal in a child process: exit status {child.returncode}, no report"
    return None if report.get("ok") else f"rehearsal in a child process: {report.get('error')}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=("dragon", "mesh"), default="dragon")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lanes", type=int, default=0, help="frames in flight per GPU (default: 1 on one GPU, 2 on up to four, 4 beyond)")
    ap.add_argument("--multi", choices=("library", "ranks"), default="library")
    ap.add_argument("--library-rehearsal", action="store_true", help=argparse.SUPPRESS)      # rank 0 starts this as a child process (library_rehearsal)
    args = ap.parse_args()
    if args.library_rehearsal:
        return library_rehearsal(args)
    if args.config == "mesh" and args.steps == 200:
        args.steps, args.warmup = 20, 3

    # Libraries underneath (RCCL prints a version banner) write to file descriptor 1; the contract is
    # ONE JSON line on stdout, so everything else is sent to stderr and the line goes to the real stdout.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # RT_BENCH_REHEARSE_COLLECTIVES=1 sends a single GPU through the multi-GPU code as well (communicator of one,
    # broadcast, tile buffers, reassembly: a rehearsal on one GPU, never set for a reported number).
    rehearse = bool(os.environ.get("RT_BENCH_REHEARSE_COLLECTIVES"))
    if world > 1 or (rehearse and args.multi == "ranks"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)       # host-side barriers and flags; GPU collectives are RCCL

    rt = ge.load_package()
    shard_mod = importlib.import_module("raytracer_rust_amd.distributed")
    syn = importlib.import_module("raytracer_rust_amd.synthetic")

    def agree(ok):
        """True on every rank iff rank 0 says so."""
        if not dist.is_initialized():
            return ok
        flag = torch.tensor([1 if ok else 0], dtype=torch.int64)
        dist.broadcast(flag, src=0)
        return bool(flag.item())

    # ------------------------------------------------------------------ the workload
    W, H = (args.width or (3840 if args.config == "dragon" else 7680)), (args.height or (2160 if args.config == "dragon" else 4320))
    mode, library_error = ("single" if world == 1 and not rehearse else args.multi), None
    gpu_build = args.config == "mesh"
    frame, scene, _keep, workload, data = make_workload(rt, syn, args.config, W, H, with_scene=rank == 0)

    # ------------------------------------------------------------------ how the GPUs are driven
    # Library path: rank 0 sets up everything that can fail or hang inside RCCL (communicators, the broadcast of the scene,
    # one whole frame through the gather) under a watchdog before anything is timed.  An error there is reported in the line
    # and the ranks path takes over in the same processes; a hang ends the run with a message that names the step.
    watchdog = Watchdog()
    t_up = time.perf_counter()
    bvh_info = None
    if mode == "library":
        ok = True
        if rank == 0:
            # First in a child process, which a hang costs nothing but a time-out; only then in this one.
            library_error = rehearse_library_in_child(args, world, W, H)
            ok = library_error is None
        if rank == 0 and ok:
            try:
                watchdog.arm(180, f"rt_init_multi({world}) (ncclCommInitAll over {world} GPUs)")
                rt.init_multi(world)
                watchdog.arm(600, "rt_upload_scene on the library path (tree build, upload and ncclBroadcast of the scene)")
                bvh_info = rt.upload_scene(scene, gpu_build=gpu_build)
                watchdog.arm(180, "the first frame on the library path (tile kernels, grouped ncclSend/ncclRecv gather, reassembly)")
                rt.render_frame(frame, want_stats=True)
            except Exception as e:      # noqa: BLE001  (reported in the line; the ranks path takes over)
                ok, library_error = False, f"{type(e).__name__}: {e}"
                watchdog.arm(120, "rt_shutdown after a failure on the library path")
                rt.shutdown()
            watchdog.disarm()
        if not agree(ok):
            mode = "ranks"
    elif mode == "single":
        rt.init(0)
        bvh_info = rt.upload_scene(scene, gpu_build=gpu_build)
    drives_gpus = mode == "ranks" or rank == 0

    renderer = group = None
    if mode == "ranks":
        torch.cuda.set_device(local_rank)
        watchdog.arm(900, "setting up the ranks path (process group, scene broadcast)")
        if dist.is_initialized():
            group = dist.new_group(backend="nccl")
        renderer, bvh_info = shard_mod.gpu_renderer(rt, local_rank, scene=scene, always_gather=dist.is_initialized(), lanes=args.lanes or None,
                                                    gpu_build=gpu_build, group=group)
        watchdog.disarm()
        lanes = renderer.lanes
        device = renderer.device
    else:
        lanes = args.lanes or (1 if world == 1 else shard_mod.default_lanes(world))
        device = torch.device("cuda", 0)
    upload_s = time.perf_counter() - t_up

    # Work of the frame (deterministic): instrumented kernel, outside the timed region.
    if mode == "ranks":
        stride, local, gathered, full = renderer.buffers(W, H)
        target = local[0] if dist.is_initialized() else full[0]
        st = rt.render_tiles(frame, rank, world, target.data_ptr(), count_work=True)
        counts = torch.tensor([st["primary"], st["shadow"], st["secondary"], st["nodes_visited"], st["prims_tested"]], dtype=torch.int64)
        mine = counts.clone()
        if dist.is_initialized():
            dist.all_reduce(counts)
    elif rank == 0:
        _, st = rt.render_frame(frame, count_work=True)
        counts = mine = torch.tensor([st["primary"], st["shadow"], st["secondary"], st["nodes_visited"], st["prims_tested"]], dtype=torch.int64)
    else:
        counts = mine = torch.zeros(5, dtype=torch.int64)
    primary, shadow, secondary, nodes, prims = (int(x) for x in counts.tolist())

    def sync_all():
        if drives_gpus:
            if mode != "ranks":
                for lane in range(lanes):
                    rt.select_lane(lane)
                    rt.synchronize()
                rt.select_lane(0)
            for d in range(world if mode == "library" else 1):
                torch.cuda.synchronize(torch.device("cuda", d) if mode == "library" else device)
        if dist.is_initialized():
            dist.barrier()

    # ------------------------------------------------------------------ the frame of N GPUs must be the frame of one
    # Before anything is timed: the same frame rendered by the first GPU alone (one shard of one) and through the path
    # the timed loop uses; they must agree in every pixel, and the line carries both checksums.
    one_gpu_checksum = multi_equal = None
    if world > 1 or rehearse:
        watchdog.arm(600, f"the check frame on the {mode} path (N GPUs against one GPU)")
        alone = None
        if rank == 0:
            rt.select_lane(0)
            ref = torch.zeros((H, W), dtype=torch.int32, device=device)
            rt.after_torch()
            rt.render_tiles(frame, 0, 1, ref.data_ptr(), want_stats=True)
            alone = ref.cpu().numpy().view(np.uint32)
            one_gpu_checksum = frame_checksum(alone)
            del ref
        if mode == "ranks":
            out = renderer.render_frame(frame)
            renderer.flush()
            if rank == 0:
                torch.cuda.synchronize(device)
                multi_equal = bool(np.array_equal(out.cpu().numpy().view(np.uint32), alone))
        elif rank == 0:
            ptr, _ = rt.render_frame(frame, want_stats=True)
            multi_equal = bool(np.array_equal(device_frame_to_host(ptr, H, W), alone))
        watchdog.disarm()

    # ------------------------------------------------------------------ timed region
    ext = None
    if mode != "ranks" and rank == 0:
        ext = []
        for lane in range(lanes):
            rt.select_lane(lane)
            ext.append(torch.cuda.ExternalStream(rt.stream(), device=device))
        rt.select_lane(0)

    slot = [0]

    def one_frame(events=None):
        """Enqueues one frame on the next lane; HIP events on the stream the frame ends on."""
        if mode == "ranks":
            lane_stream = renderer.ext_streams[renderer.next_lane()]
            if events: events[0].record(lane_stream)
            out = renderer.render_frame(frame)
            if events: events[1].record(lane_stream)
            return out
        if rank != 0:
            return None
        lane = slot[0]; slot[0] = (lane + 1) % lanes
        rt.select_lane(lane)
        if events: events[0].record(ext[lane])
        ptr, _ = rt.render_frame(frame)
        if events: events[1].record(ext[lane])
        return ptr

    watchdog.arm(1200, f"warm-up and timed frames on the {mode} path")
    for _ in range(args.warmup):
        one_frame()
    if mode == "ranks":
        renderer.flush()
    sync_all()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)] if drives_gpus else []
    t0 = time.perf_counter()
    out = None
    for i in range(args.steps):
        out = one_frame(ev[i] if drives_gpus else None)
    if mode == "ranks":
        renderer.flush()          # the last frame's gather and reassembly belong to the timed region
    sync_all()
    elapsed = time.perf_counter() - t0
    watchdog.disarm()
    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        if mode == "ranks":
            frame_host = out.cpu().numpy().view(np.uint32)
        else:
            frame_host = device_frame_to_host(out, H, W)
        checksum = frame_checksum(frame_host)
        if multi_equal is not None:
            multi_equal = multi_equal and checksum == one_gpu_checksum       # the last timed frame too

    # Kernel-only duration (events bracketing just the kernel; the loop above brackets kernel + gather + reassembly),
    # the first frame of a freshly uploaded scene (no tile order from a previous launch yet), and the frame delivered to
    # host memory (the raytracer.rs:40 contract).
    kernel_ms, cold_ms, host_ms, seeded_ms, cold_host_ms, idle_ms = [], None, None, None, None, None
    if mode == "ranks":
        for _ in range(min(args.steps, 10)):
            kernel_ms.append(rt.render_tiles(frame, rank, world, target.data_ptr(), want_stats=True)["kernel_ms"])
    elif rank == 0:
        rt.select_lane(0)
        for _ in range(min(args.steps, 10)):
            kernel_ms.append(rt.render_frame(frame, want_stats=True)[1]["kernel_ms"])
        host_times, host_frame = [], np.empty((H, W), dtype=np.uint32)      # the caller's buffer, kept across frames (first-touch page faults are the allocation's cost)
        for _ in range(7):
            _, st_host = rt.render(frame, out=host_frame)
            host_times.append(st_host["total_ms"])
        host_ms = float(np.mean(host_times[2:]))
        assert np.array_equal(host_frame, frame_host), "rt_render and rt_render_frame disagree"
        # a warm frame (same scene, same measured order) that finds the GPU idle: what an idle GPU costs by itself, which every
        # first frame below pays too (its upload leaves the GPU idle for longer than this)
        idle_times = []
        for _ in range(3):
            time.sleep(0.1)
            idle_times.append(rt.render_frame(frame, want_stats=True)[1]["kernel_ms"])
        idle_ms = float(np.mean(idle_times))
        rt.upload_scene(scene, gpu_build=gpu_build)                       # a new resident scene: the tile order starts from scratch
        cold_ms = rt.render_frame(frame, want_stats=True)[1]["kernel_ms"]     # (includes the kernels that guess the order)
        # the reference's own timed region (main.rs:13-15): ONE render() of a scene that has just been built, frame delivered to the host
        rt.upload_scene(scene, gpu_build=gpu_build)
        cold_host_ms = rt.render(frame, out=host_frame)[1]["total_ms"]
        # ... and the first frame of a later run that kept the tile costs of this one beside its scene cache
        try:
            saved = [rt.tile_costs_export(frame, gpu) for gpu in range(rt.device_count())]
            rt.upload_scene(scene, gpu_build=gpu_build)
            for gpu, costs in enumerate(saved):
                rt.tile_costs_import(frame, costs, gpu)
            seeded_ms = rt.render_frame(frame, want_stats=True)[1]["kernel_ms"]
        except rt.RtError as e:                                           # an extra figure: the line is printed without it
            seeded_ms = f"failed: {e}"
    step_ms_events = [a.elapsed_time(b) for a, b in ev] if drives_gpus else []

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        rays_ps = primary + shadow
        rays_all = primary + shadow + secondary
        k_sync_ms = float(np.mean(kernel_ms))
        # The kernel's duration: on one GPU the HIP events of the timed region itself, recorded on the lane's stream around every
        # step (render kernel plus the 5 us kernel behind it that orders the next launch's tiles, so a little more than the kernel);
        # on several GPUs a step also holds the gather, and the library's events around the kernel of separately launched frames
        # stand in (those launches find the GPU idle after the host's wait and run about 2 % longer).
        k_ms = float(np.mean(step_ms_events)) if mode == "single" and lanes == 1 and step_ms_events else k_sync_ms
        my_primary, _, _, my_nodes, my_prims = (int(x) for x in mine.tolist())
        alg_bytes = 32 * my_nodes + 48 * my_prims + 4 * my_primary      # per launch (all GPUs together on the library path)
        requested = 16 * my_nodes + 48 * my_prims + 4 * my_primary      # what the kernel really asks the caches for: a box is 16 bytes in the node layout
        launches = world if mode == "library" else 1                    # the counts above are spread over this many concurrent launches
        achieved = alg_bytes / launches / (k_ms * 1e-3) / 1e9
        pmc = load_counters(args.config)
        # Neither workload is carried by HBM: measured HBM traffic per launch (traffic) is 1.5 % of the algorithmic bytes on the
        # dragon (7 MB scene, L2 hit rate 94 %) and 4 % on the mesh (641 MB scene, L2 hit rate 87 %, the rest mostly from the
        # 256 MB memory-side cache), so the memory-side bound of the traversal is the aggregate L2 bandwidth for both.
        bound, peak = "l2", L2_PEAK_GBS
        note = ("achieved and frac are measured in this run: algorithmic bytes (32 B x boxes tested + 48 B x shapes tested + 4 B x pixels, "
                "counted by the instrumented kernel) over the kernel's duration by HIP events on its stream around every step of the timed "
                "region (kernel_ms; kernel_ms_synchronous_launches: the library's own events around the kernel of frames launched one at a "
                "time with a wait in between, which find the GPU idle), against the aggregate L2 "
                "bandwidth, because the caches serve these bytes (traffic is what reaches HBM).  A box is 16 B in the node layout, so the "
                "kernel requests requested_bytes_per_launch; frac_of_requested_bytes is the same fraction with those.  The fields named in "
                "counters_source come from committed rocprofv3 --pmc passes, not from this run.  What limits the kernel is vector ALU issue "
                "(valu_issue_frac of the SIMDs' issue slots, 5 wavefronts per SIMD; instructions and cycles come from two counter passes, so a "
                "full unit reads as 1.0 give or take a percent) spent at valu_lane_use of the lanes: a wavefront runs as long as its slowest ray")
        roofline = {"bound": bound, "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak,
                    "traffic": pmc.get("hbm_bytes_per_launch"),
                    "kernel": "render_kernel", "kernel_ms": k_ms, "kernel_ms_synchronous_launches": k_sync_ms,
                    "algorithmic_bytes_per_launch": alg_bytes / launches,
                    "requested_bytes_per_launch": requested / launches,
                    "frac_of_requested_bytes": requested / launches / (k_ms * 1e-3) / 1e9 / peak,
                    "frac_of_hbm_peak": achieved / HBM_PEAK_GBS, "note": note}
        canned = []
        if pmc.get("hbm_bytes_per_launch"):
            roofline["hbm_traffic_frac_of_peak"] = pmc["hbm_bytes_per_launch"] / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            canned += ["traffic", "hbm_traffic_frac_of_peak (canned bytes over this run's kernel time)"]
        if pmc.get("valu_insts") and pmc.get("gui_active_cycles_all_xcds"):
            # issue slots: one vector instruction per SIMD every 4 cycles, over the cycles the same profiled launch took
            roofline["valu_issue_frac"] = pmc["valu_insts"] * 4 / (SIMDS * pmc["gui_active_cycles_all_xcds"] / XCDS)
            roofline["valu_lane_use"] = pmc.get("valu_lane_utilisation")
            roofline["l2_hit_rate"] = pmc.get("l2_hit_rate")
            canned += ["valu_issue_frac", "valu_lane_use", "l2_hit_rate"]
        roofline["counters_source"] = (f"profiles/traffic.json @ {counters_commit(pmc)} ({pmc.get('source', 'no entry for this workload')}): " + ", ".join(canned)
                                       if canned else None)
        result = {
            "metric": "Mrays/sec (primary+shadow) at 3840x2160 dragon scene" if args.config == "dragon" else "Mrays/sec (primary+shadow), 10M-triangle mesh at 7680x4320",
            "value": rays_ps * args.steps / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": data,
            "config": {"workload": workload, "tile_rows": rt.TILE_ROWS,
                       "sharding": f"interleaved row tiles over {world} GPU(s), gather on GPU 0" if world > 1 else "one GPU",
                       "multi_gpu_path": {"single": "none (one GPU)", "library": "one process, RCCL inside the library (rt_init_multi)",
                                          "ranks": "one process per GPU, torch.distributed over RCCL"}[mode],
                       "frames_in_flight_per_gpu": lanes},
            "frames_per_s": args.steps / elapsed,
            "mrays_all_per_s": rays_all * args.steps / elapsed / 1e6,
            "rays_per_frame": {"primary": primary, "shadow": shadow, "reflection": secondary},
            "work_per_frame": {"boxes_tested": nodes, "shapes_tested": prims},
            "bvh": bvh_info,
            "scene_upload_s": upload_s,
            "step_ms_hip_events_mean": float(np.mean(step_ms_events)),
            "cold_first_frame_ms": cold_ms,
            "warm_frame_on_idle_gpu_ms": idle_ms,
            "first_frame_with_saved_tile_costs_ms": seeded_ms,
            "rt_render_ms": host_ms,
            "cold_rt_render_ms": cold_host_ms,
            "frame_checksum": checksum,
            "roofline": roofline,
        }
        if multi_equal is not None:
            result["multi_gpu_frame_equal"] = multi_equal
            result["frame_checksum_one_gpu"] = one_gpu_checksum
            if not multi_equal:
                result["error"] = "the frame gathered from the GPUs differs from the frame of one GPU: the numbers of this line are void"
        if library_error:
            result["config"]["library_path_error"] = library_error
        if world == 1 and not args.no_cpu_baseline:
            oracle = ge.load_oracle()
            result["cpu_baseline"] = cpu_baseline_dragon(oracle, W, H) if args.config == "dragon" else cpu_baseline_mesh(oracle, syn, W, H)
        print(json.dumps(result), file=real_stdout, flush=True)

    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and multi_equal is False:
        sys.exit(4)


if __name__ == "__main__":
    main()
