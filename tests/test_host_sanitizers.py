"""The host-side C ABI (parser, scene packer and tree builder, blob validation, scene cache, PNG reader and writer)
built with AddressSanitizer + UndefinedBehaviorSanitizer and driven through good and damaged inputs
(tests/sanitize/host_driver.cpp).  The reference gets its memory safety from Rust; this is the C++ side's
counterpart.  No GPU involved: only the sources that g++ compiles go into the binary."""
import os
import shutil
import subprocess

import pytest

from conftest import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raytracer-rust_amd", "csrc")
HOST_SOURCES = ["errors.cpp", "frame.cpp", "bvh_build.cpp", "scene_pack.cpp", "host_config.cpp", "png_io.cpp"]


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_code_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    exe = tmp_path / "host_driver"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fopenmp",
           "-ffp-contract=off", os.path.join(ROOT, "tests", "sanitize", "host_driver.cpp")] + [os.path.join(CSRC, s) for s in HOST_SOURCES] + ["-lz", "-o", str(exe)]
    build = subprocess.run(cmd, capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-4000:]
    scratch = tmp_path / "scratch"
    scratch.mkdir()
    env = dict(os.environ, OMP_NUM_THREADS="4", ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)          # the sanitizer runtime has to come first in the program's own link order
    run = subprocess.run([str(exe), GOLDEN, str(scratch)], capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-6000:])
    assert "no findings" in run.stdout
