#!/bin/bash
# Builds a side-by-side variant of the library with extra kernel defines, for A/B timing on one GPU box:
#   scripts/build_variant.sh <name> [-DRT_WAVES_PER_EU=4 ...]   ->  raytracer-rust_amd/variants/librt_amd_<name>.so
# (select it with RT_AMD_LIBRARY=<path>).  The host objects of the regular build are reused.
set -e
cd "$(dirname "$0")/../raytracer-rust_amd"
name=$1; shift
mkdir -p variants
make -s -j8 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize \
    -Wno-unused-function -Wno-pass-failed "$@" -c csrc/render_kernels.hip -o variants/render_kernels_$name.o
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I/opt/rocm/include "$@" -c csrc/rt_api.cpp -o variants/rt_api_$name.o
objs=$(ls csrc/*.o | grep -v -e render_kernels.o -e rt_api.o)
/opt/rocm/bin/hipcc -shared -o variants/librt_amd_$name.so variants/render_kernels_$name.o variants/rt_api_$name.o $objs -fopenmp=libgomp -lz -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
rm -f variants/render_kernels_$name.o variants/rt_api_$name.o
echo variants/librt_amd_$name.so
