// Micro-benchmark: cost of a global_load_dwordx4 wave-instruction as a function of how many lanes
// are active and how many distinct 64-byte lines they touch (data resident in L1/L2).
// Build: hipcc --offload-arch=gfx950 -O3 -o ta_lanes ta_lanes.hip ; run: ./ta_lanes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(64) void probe(const float4* __restrict__ buf, uint32_t mask_n, uint32_t stride, uint32_t iters, float* out) {
    const uint32_t lane = threadIdx.x;
    float acc = 0.0f;
    if (lane < mask_n) {
        // every lane reads 4 x 16 B of "its" 64-byte record; records are `stride` records apart per lane
        uint32_t idx = (blockIdx.x * 977u + lane * stride) & 4095u;
        for (uint32_t i = 0; i < iters; i++) {
            const float4* p = buf + 4 * (size_t)idx;
            float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
            idx = (idx + __float_as_uint(a.w)) & 4095u;     // dependent chain, like a traversal
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main() {
    const int nrec = 4096;                    // 256 KiB: L2 resident, largely L1 resident per CU
    std::vector<float4> h(4 * nrec);
    for (int i = 0; i < 4 * nrec; i++) { h[i] = make_float4(1, 2, 3, 0); uint32_t step = 1 + (i * 2654435761u >> 20) % 7; h[i].w = *(float*)&step; }
    float4* d; float* out;
    hipMalloc(&d, h.size() * sizeof(float4)); hipMalloc(&out, 4);
    hipMemcpy(d, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const uint32_t iters = 2000, blocks = 256 * 20;
    for (uint32_t stride : {0u, 1u, 37u}) {
        for (uint32_t n : {64u, 32u, 16u, 8u, 4u, 1u}) {
            probe<<<blocks, 64>>>(d, n, stride, 10, out);
            hipEventRecord(e0);
            probe<<<blocks, 64>>>(d, n, stride, iters, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double wave_instr = (double)blocks * iters * 4;
            printf("stride %2u active lanes %2u: %.3f ms  -> %.1f ns per load wave-instruction per CU (%.1f cycles at 2.4 GHz)\n", stride, n, ms,
                   ms * 1e6 / (wave_instr / 256), ms * 1e6 / (wave_instr / 256) * 2.4);
        }
    }
    return 0;
}
