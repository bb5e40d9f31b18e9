// Does a wave64 vector-ALU instruction cost less when whole 16-lane quarters of the wavefront are inactive?
// One workgroup of 64 threads per CU-ish, a long chain of dependent FMAs under different lane masks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void chain(float* out, unsigned long long mask, int iters) {
    const unsigned lane = threadIdx.x & 63u;
    float a = 1.0f + lane, b = 0.5f, c = 0.25f, d = 0.125f;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; i++) {
            a = fmaf(a, 1.0001f, 0.5f); b = fmaf(b, 0.9999f, 0.25f); c = fmaf(c, 1.0002f, 0.125f); d = fmaf(d, 0.9998f, 0.0625f);
            a = fmaf(a, 1.0001f, 0.5f); b = fmaf(b, 0.9999f, 0.25f); c = fmaf(c, 1.0002f, 0.125f); d = fmaf(d, 0.9998f, 0.0625f);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}

int main() {
    float* d_out; (void)hipMalloc(&d_out, 1 << 24);
    const unsigned long long masks[] = {~0ull, 0xFFFFull, 0xFFFFFFFFull, 1ull, 0x0001000100010001ull, 0xFFFF0000FFFFull};
    const char* names[] = {"all 64", "lanes 0-15", "lanes 0-31", "lane 0", "one lane per quarter", "quarters 0 and 2"};
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 4)
        for (int m = 0; m < 6; m++) {
            const int blocks = 256 * 4 * waves_per_simd;
            hipLaunchKernelGGL(chain, dim3(blocks), dim3(64), 0, 0, d_out, masks[m], 1000);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(chain, dim3(blocks), dim3(64), 0, 0, d_out, masks[m], 200000);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%d wave(s)/SIMD, %-22s %8.3f ms\n", waves_per_simd, names[m], ms);
        }
    return 0;
}
