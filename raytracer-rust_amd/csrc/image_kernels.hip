// image_kernels.hip -- what happens to a finished frame while it is still in HBM.
//
//   compare   Image::compare (src/imgcomparator/mod.rs:54-89): per pixel the absolute difference
//             of each colour channel; a pixel counts as different when any channel differs by more
//             than 1; the difference image holds (dr << 16 | dg << 8 | db) for those and 0 elsewhere.
//   pack_rgb  the per-pixel loop of save_image (src/imgcomparator/mod.rs:156-162): 0x??RRGGBB words
//             to the R, G, B byte triples a PNG encoder wants (a quarter less to copy to the host).
//
// Both are pure streaming kernels bounded by HBM: 8 B read (+ 4 B written) and 4 B read + 3 B
// written per pixel.  Threads take four pixels each with 16-byte accesses.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/rt_amd.h"
#include "device_types.h"
#include "errors.h"

namespace {

__device__ __forceinline__ uint32_t channel_diff(uint32_t a, uint32_t b, bool* differs) {
    if (a == b) return 0u;                                                                   // mod.rs:63
    const int dr = abs((int)((a >> 16) & 0xFFu) - (int)((b >> 16) & 0xFFu));
    const int dg = abs((int)((a >> 8) & 0xFFu) - (int)((b >> 8) & 0xFFu));
    const int db = abs((int)(a & 0xFFu) - (int)(b & 0xFFu));
    if (dr <= 1 && dg <= 1 && db <= 1) return 0u;                                            // mod.rs:70
    *differs = true;
    return ((uint32_t)dr << 16) | ((uint32_t)dg << 8) | (uint32_t)db;
}

// n4 groups of four pixels through 16-byte accesses, then the (up to three) pixels left over.
__global__ __launch_bounds__(256) void compare_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, size_t n,
                                                      uint32_t* __restrict__ diff, unsigned long long* __restrict__ count) {
    const size_t n4 = n >> 2;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    uint32_t mine = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const uint4 pa = ((const uint4*)a)[i], pb = ((const uint4*)b)[i];
        bool d0 = false, d1 = false, d2 = false, d3 = false;
        uint4 d;
        d.x = channel_diff(pa.x, pb.x, &d0); d.y = channel_diff(pa.y, pb.y, &d1);
        d.z = channel_diff(pa.z, pb.z, &d2); d.w = channel_diff(pa.w, pb.w, &d3);
        mine += (uint32_t)d0 + (uint32_t)d1 + (uint32_t)d2 + (uint32_t)d3;
        if (diff) ((uint4*)diff)[i] = d;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3u)) {
        const size_t i = (n4 << 2) + threadIdx.x;
        bool d0 = false;
        const uint32_t d = channel_diff(a[i], b[i], &d0);
        mine += (uint32_t)d0;
        if (diff) diff[i] = d;
    }
    // One atomic per block: every update of the one counter word is serialised in L2.
    __shared__ uint32_t per_wave[4];
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
    if ((threadIdx.x & 63u) == 0) per_wave[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = per_wave[0] + per_wave[1] + per_wave[2] + per_wave[3];
        if (total) atomicAdd(count, (unsigned long long)total);
    }
}

__global__ __launch_bounds__(256) void pack_rgb_kernel(const uint32_t* __restrict__ pixels, size_t n, uint8_t* __restrict__ rgb) {
    const size_t n4 = n >> 2;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const uint4 p = ((const uint4*)pixels)[i];
        // bytes in memory order: R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
        const uint32_t r0 = (p.x >> 16) & 0xFFu, g0 = (p.x >> 8) & 0xFFu, b0 = p.x & 0xFFu;
        const uint32_t r1 = (p.y >> 16) & 0xFFu, g1 = (p.y >> 8) & 0xFFu, b1 = p.y & 0xFFu;
        const uint32_t r2 = (p.z >> 16) & 0xFFu, g2 = (p.z >> 8) & 0xFFu, b2 = p.z & 0xFFu;
        const uint32_t r3 = (p.w >> 16) & 0xFFu, g3 = (p.w >> 8) & 0xFFu, b3 = p.w & 0xFFu;
        uint32_t* out = (uint32_t*)rgb + 3 * i;
        out[0] = r0 | (g0 << 8) | (b0 << 16) | (r1 << 24);
        out[1] = g1 | (b1 << 8) | (r2 << 16) | (g2 << 24);
        out[2] = b2 | (r3 << 8) | (g3 << 16) | (b3 << 24);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3u)) {
        const size_t i = (n4 << 2) + threadIdx.x;
        const uint32_t p = pixels[i];
        rgb[3 * i] = (uint8_t)(p >> 16); rgb[3 * i + 1] = (uint8_t)(p >> 8); rgb[3 * i + 2] = (uint8_t)p;
    }
}

uint32_t streaming_blocks(size_t n) {
    // Enough 256-thread blocks for every CU to hold several, capped: the kernels stride over the rest.
    size_t want = ((n >> 2) + 255) / 256;
    if (want < 1) want = 1;
    if (want > 256u * 8u) want = 256u * 8u;
    return (uint32_t)want;
}

}  // namespace

extern "C" hipError_t rtk_launch_compare(const uint32_t* d_a, const uint32_t* d_b, size_t n, uint32_t* d_diff, unsigned long long* d_count,
                                         hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(compare_kernel, dim3(streaming_blocks(n)), dim3(256), 0, stream, d_a, d_b, n, d_diff, d_count);
    return hipGetLastError();
}

extern "C" hipError_t rtk_launch_pack_rgb(const uint32_t* d_pixels, size_t n, uint8_t* d_rgb, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_rgb_kernel, dim3(streaming_blocks(n)), dim3(256), 0, stream, d_pixels, n, d_rgb);
    return hipGetLastError();
}

// ------------------------------------------------------------------ comparison that allocates nothing
// rt_compare_frames keeps its pixel counter in device memory the library allocated at rt_init.  Callers that must not
// allocate (HIP graph capture, serving loops with a fixed memory budget) get the same kernel with the counter in this
// static device array instead: 256 bytes per GPU for the life of the process, of which the kernel uses the first word.
// In host code the array's name is only a handle; its address on each GPU comes from hipGetSymbolAddress, once per device.
static __device__ __align__(256) unsigned long long scratch[32];

namespace {
constexpr int MAX_DEVICES = 64;
unsigned long long* scratch_ptr[MAX_DEVICES] = {};
}

// Called by rt_init with `device` current.
extern "C" hipError_t rtk_noalloc_setup(int device) {
    if (device < 0 || device >= MAX_DEVICES) return hipErrorInvalidDevice;
    if (scratch_ptr[device]) return hipSuccess;
    void* p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    scratch_ptr[device] = (unsigned long long*)p;
    return hipSuccess;
}

extern "C" int rt_compare_frames_noalloc(const uint32_t* d_a, uint32_t width_a, uint32_t height_a, const uint32_t* d_b, uint32_t width_b,
                                         uint32_t height_b, uint32_t* d_diff, uint64_t* d_count) {
    auto fail = [](const char* m) { rte::set_error(m); return 1; };
    const int device = rti_bound_device();
    if (device < 0) return fail("rt_init has not been called");
    if (height_a != height_b || width_a != width_b) return fail("Images have different dimensions");           // mod.rs:55-57
    if (!d_a || !d_b || !d_count) return fail("rt_compare_frames_noalloc: NULL image or counter");
    if (((uintptr_t)d_a | (uintptr_t)d_b | (uintptr_t)d_diff) & 15u) return fail("rt_compare_frames_noalloc: buffers must be 16-byte aligned");
    if ((uintptr_t)d_count & 7u) return fail("rt_compare_frames_noalloc: the counter must be 8-byte aligned");
    const uint64_t n = (uint64_t)width_a * height_a;
    if (n > RT_NOALLOC_MAX_PIXELS) return fail("rt_compare_frames_noalloc: more than 2^31 pixels");
    unsigned long long* const ptr = device < MAX_DEVICES ? scratch_ptr[device] : nullptr;
    if (!ptr) return fail("rt_compare_frames_noalloc: no scratch array on this device");
    hipStream_t stream = (hipStream_t)rt_stream();
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMemsetAsync(ptr, 0, sizeof(unsigned long long), stream);        // the last call's count is still there
    if (e == hipSuccess) e = rtk_launch_compare(d_a, d_b, (size_t)n, d_diff, ptr, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_count, ptr, sizeof(unsigned long long), hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) { rte::set_error(std::string("rt_compare_frames_noalloc: ") + hipGetErrorString(e)); return 1; }
    return 0;
}
