"""The specular power of the render kernel (csrc/specular_pow.h) equals the C library's powf bit for bit (host side of the
same source; the GPU side is covered by the frames of tests/test_gpu_parity.py, which are compared with an oracle that calls powf)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_specular_power_equals_libm_powf(tmp_path):
    exe = str(tmp_path / "pow_check")
    flags = ["-O2", "-std=c++17", "-ffp-contract=off"]
    if " fma " in open("/proc/cpuinfo").read():
        flags.append("-mfma")          # hardware fused multiply-add; without it __builtin_fma is the (slower, equally exact) library call
    subprocess.check_call(["g++"] + flags + [os.path.join(ROOT, "tests", "native", "pow_check.cpp"), "-o", exe])
    out = subprocess.check_output([exe, "20000000"], text=True)
    domain, mismatches = (int(out.split()[1]), int(out.split()[3]))
    # The C library has a build without fused multiply-add for old CPUs whose last binary64 bits can differ; after the final
    # rounding to binary32 that shows in about one result in 10^8.  Every CPU this is expected to run on takes the fused build: 0.
    assert domain > 15_000_000 and mismatches <= 2, out
