#!/usr/bin/env python3
"""Where the render kernel's vector instructions go, phase by phase.

Static side: compiles csrc/render_kernels.hip to gfx950 assembly with -DRT_MARKS (the MARK() comments of the source
survive into the instruction stream) and counts, for render_kernel<false>, the instructions that follow each mark
up to the next one, by unit (vector ALU, scalar ALU, vector memory, LDS, scalar memory).  The compiler emits some
phases more than once (it peels the first light of the light loop, for instance); a phase's figure is the average
over its copies.  The marks are comments: instructions the scheduler moved across one are charged to the neighbour, so
figures of adjacent phases can be off by a few instructions, their sum is not.

Dynamic side (optional, `--counts file.json`: the statistics of one instrumented frame as scripts/work_counts.py
prints them): multiplies each phase by the number of wavefront-level passes through it and prints the share of the
frame's vector instructions, next to the hardware total if `--valu-insts N` is given.

  python3 scripts/phase_table.py [--counts gpurun_out/work_counts.json] [--valu-insts 1.64e9] [-D...]
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "raytracer-rust_amd", "csrc", "render_kernels.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wno-unused-function", "-Wno-pass-failed", "-DRT_MARKS", "-S", "--cuda-device-only"]

# phase -> which pass count multiplies it (keys of passes_of); kind-specific phases count only for the kinds the scene has
PHASE_PASSES = collections.OrderedDict([
    ("prologue", "waves"), ("raygen", "waves"),
    ("closest_list", "closest"), ("plane_0", "closest:list:plane"), ("triangle_0", "closest:list:triangle"), ("sphere_0", "closest:list:sphere"),
    ("tested_0", "closest"), ("closest_raybox", "closest"),
    ("closest_box", "closest_box"),
    ("closest_shape", "closest_shape"), ("triangle_1", "closest_shape:tree:triangle"), ("sphere_1", "closest_shape:tree:sphere"),
    ("plane_1", "closest_shape:tree:plane"), ("tested_1", "closest_shape"),
    ("hit", "hit"), ("hit_triangle", "hit:hit:triangle"), ("hit_sphere", "hit:hit:sphere"), ("hit_plane", "hit:hit:plane"), ("hit_material", "hit"),
    ("light", "light"), ("light_point", "light:light:point"), ("light_directional", "light:light:directional"), ("light_horizon", "light"),
    ("shadow_list", "shadow"), ("plane_2", "shadow:list:plane"), ("triangle_2", "shadow:list:triangle"), ("sphere_2", "shadow:list:sphere"),
    ("tested_2", "shadow"), ("occluder_back_face_2", "never"), ("guards_2", "shadow"), ("shadow_raybox", "shadow"),
    ("shadow_box", "shadow_box"),
    ("shadow_shape", "shadow_shape"), ("triangle_3", "shadow_shape:tree:triangle"), ("sphere_3", "shadow_shape:tree:sphere"),
    ("plane_3", "shadow_shape:tree:plane"), ("tested_3", "shadow_shape"), ("occluder_back_face_3", "never"), ("guards_3", "shadow_shape"),
    ("round_setup", "rounds"), ("round_list", "rounds"), ("plane_4", "rounds:list:plane"), ("triangle_4", "rounds:list:triangle"),
    ("sphere_4", "rounds:list:sphere"), ("tested_4", "rounds"), ("occluder_back_face_4", "never"), ("guards_4", "rounds"), ("round_raybox", "rounds"), ("round_box", "round_box"),
    ("round_shape", "round_shape"), ("triangle_5", "round_shape:tree:triangle"), ("sphere_5", "round_shape:tree:sphere"),
    ("plane_5", "round_shape:tree:plane"), ("tested_5", "round_shape"), ("occluder_back_face_5", "never"), ("guards_5", "round_shape"), ("round_collect", "rounds"),
    ("shade", "shade"), ("shade_pow", "shade"), ("shade_pow_main_path", "shade"), ("shade_pow_off_the_main_path", "never"), ("shade_sum", "shade"),
    ("level_end", "hit"), ("fold", "waves"), ("quantise", "waves"), ("epilogue", "waves"),
])
# what the headline scene (final_avec_bonus.scene) holds: a plane in the always-tested list, triangles in the tree, point lights
KINDS = {"list": {"plane"}, "tree": {"triangle"}, "hit": {"triangle", "plane"}, "light": {"point"}}


def classify(mnemonic):
    if mnemonic.startswith("v_"):
        return "valu"
    if mnemonic.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if mnemonic.startswith(("s_waitcnt", "s_nop")):
        return "wait"
    if mnemonic.startswith("s_"):
        return "salu"
    if mnemonic.startswith("ds_"):
        return "lds"
    if mnemonic.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def static_table(extra_flags):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + extra_flags + [SRC, "-o", out], stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    start = next(i for i, l in enumerate(text) if re.match(r"^_Z\d+render_kernelILb0ELb1E.*:", l))     # not instrumented, tree of triangles only (the headline scene)
    end = next(i for i in range(start, len(text)) if text[i].strip().startswith("s_endpgm"))
    phase, counts, copies = "prologue", collections.defaultdict(collections.Counter), collections.Counter({"prologue": 1})
    # The box and shape loops are what follows their mark up to the next mark; a loop's per-pass figure is the body
    # between the mark and the backward branch, which here is the whole marked stretch (the marks sit at loop heads).
    for line in text[start + 1:end]:
        m = re.search(r"; MARK (\w+)", line)
        if m:
            phase = m.group(1); copies[phase] += 1
            continue
        m = re.match(r"^\t([a-z_0-9]+)", line)
        if m:
            counts[phase][classify(m.group(1))] += 1
    return counts, copies


def passes_of(st, n_lights):
    """Wavefront-level pass counts per phase from the statistics of an instrumented frame."""
    ph = st["phase_passes"]
    waves = st["primary"] / 64.0
    return {"waves": waves, "closest": ph[0], "hit": ph[1], "light": ph[1] * n_lights, "shadow": ph[2], "shade": ph[3],
            "closest_box": st["box_passes"] - ph[4] - ph[7], "shadow_box": ph[4], "closest_shape": st["shape_passes"] - ph[5] - ph[8], "shadow_shape": ph[5],
            "rounds": ph[6], "round_box": ph[7], "round_shape": ph[8]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--counts")
    ap.add_argument("--valu-insts", type=float, default=0.0)
    ap.add_argument("--lights", type=int, default=2)
    args, extra = ap.parse_known_args()
    counts, copies = static_table(extra)
    units = ("valu", "salu", "vmem", "lds", "smem", "wait")
    passes = passes_of(json.load(open(args.counts)), args.lights) if args.counts else None
    print(f"{'phase':28s} {'copies':>6s} " + " ".join(f"{u:>6s}" for u in units) + ("   passes(M)  valu(M)  share" if passes else ""))
    total, rows = 0.0, []
    for phase in list(PHASE_PASSES) + [p for p in counts if p not in PHASE_PASSES]:
        if phase not in counts:
            continue
        n = max(copies[phase], 1)
        per = {u: counts[phase][u] / n for u in units}
        row = f"{phase:28s} {n:6d} " + " ".join(f"{per[u]:6.1f}" for u in units)
        dyn = None
        if passes and phase in PHASE_PASSES:
            # loops: the marked stretch is one pass of the loop body (plus the few instructions between loop exit and the next mark)
            key = PHASE_PASSES[phase].split(":")
            n_pass = 0.0 if key[0] == "never" or (len(key) == 3 and key[2] not in KINDS[key[1]]) else passes[key[0]]
            dyn = (per["valu"] * n_pass, n_pass)
            total += dyn[0]
        rows.append((row, phase, dyn))
    for row, phase, dyn in rows:
        if dyn is not None:
            row += f"   {dyn[1] / 1e6:9.3f} {dyn[0] / 1e6:8.1f} {dyn[0] / total * 100:5.1f} %"
        print(row)
    if passes:
        print(f"sum of phases: {total / 1e9:.3f} G vector instructions per frame" + (f" (hardware counter: {args.valu_insts / 1e9:.3f} G)" if args.valu_insts else ""))


if __name__ == "__main__":
    sys.exit(main())
