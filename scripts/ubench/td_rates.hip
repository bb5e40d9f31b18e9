// Throughput of the vector memory path of one CU for the traversal's access shape: every active lane
// loads "its" 32-byte record (two 16-byte loads) from an array of the size of the dragon's node array,
// addresses independent of the loaded data (throughput, not latency), 16 wavefronts per CU.
// Prints cycles per wavefront-level load instruction per CU, by number of active lanes and record spread.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

template <int SHAPE>   // 0: 2 x dwordx4 of one 32-byte record; 1: 1 x dwordx4; 2: 2 x dwordx2 (16 bytes); 3: 4 x dwordx4 of a 64-byte record
__global__ __launch_bounds__(64) void probe(const uint4* __restrict__ buf, unsigned long long mask, uint32_t nrec, uint32_t coherent, uint32_t iters, uint32_t* out) {
    const uint32_t lane = threadIdx.x;
    uint32_t acc = 0;
    if ((mask >> lane) & 1ull) {
        uint32_t s = blockIdx.x * 9781u + (coherent ? 0u : lane * 7919u);
        for (uint32_t i = 0; i < iters; i++) {
            s = s * 1664525u + 1013904223u;
            uint32_t idx = ((s >> 8) + (coherent ? lane : 0u)) % nrec;
            if (SHAPE == 0) { const uint4* p = buf + 2 * (size_t)idx; uint4 a = p[0], b = p[1]; acc += a.x ^ b.w; }
            if (SHAPE == 1) { const uint4* p = buf + 2 * (size_t)idx; uint4 a = p[0]; acc += a.x ^ a.w; }
            if (SHAPE == 2) { const uint2* p = (const uint2*)(buf + 2 * (size_t)idx); uint2 a = p[0], b = p[1]; acc += a.x ^ b.y; }
            if (SHAPE == 3) { const uint4* p = buf + 4 * (size_t)(idx >> 1); uint4 a = p[0], b = p[1], c = p[2], d = p[3]; acc += a.x ^ b.w ^ c.y ^ d.z; }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int SHAPE>
void sweep(const char* name, int loads, const uint4* d, uint32_t nrec, uint32_t* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const uint32_t iters = 4000, blocks = 256 * 16;
    const unsigned long long masks[] = {~0ull, 0xFFFFFFFFull, 0x5555555555555555ull, 0x00000000001FFFFFull, 0x1111111111111111ull, 0xFFFFull, 0x0101010101010101ull};
    const char* mnames[] = {"64", "32 (low half)", "32 (every other)", "21 (low)", "16 (every 4th)", "16 (low quarter)", "8 (every 8th)"};
    for (uint32_t coherent = 0; coherent < 2; coherent++)
        for (int m = 0; m < 7; m++) {
            hipLaunchKernelGGL(probe<SHAPE>, dim3(blocks), dim3(64), 0, 0, d, masks[m], nrec, coherent, 10u, out);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(probe<SHAPE>, dim3(blocks), dim3(64), 0, 0, d, masks[m], nrec, coherent, iters, out);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double per_cu = (double)blocks / 256 * iters * loads;
            printf("%-28s %-10s lanes %-18s %7.3f ms  %6.1f cycles per load instruction per CU (2.4 GHz)\n", name, coherent ? "coherent" : "scattered", mnames[m], ms,
                   ms * 1e-3 * 2.4e9 / per_cu);
        }
}

int main(int argc, char** argv) {
    const uint32_t nrec = argc > 1 ? (uint32_t)atoi(argv[1]) : 55167;     // default: node pairs of the dragon
    uint4* d; uint32_t* out;
    (void)hipMalloc(&d, (size_t)nrec * 64); (void)hipMemset(d, 1, (size_t)nrec * 64); (void)hipMalloc(&out, 4);
    sweep<0>("2 x dwordx4 (32 B record)", 2, d, nrec, out);
    sweep<1>("1 x dwordx4", 1, d, nrec, out);
    sweep<2>("2 x dwordx2 (16 B record)", 2, d, nrec, out);
    sweep<3>("4 x dwordx4 (64 B record)", 4, d, nrec, out);
    return 0;
}
