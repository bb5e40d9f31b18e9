// Issue cost of the vector instructions the box test is made of, on one SIMD: 8 independent chains of one
// instruction each per loop pass, 4 wavefronts per SIMD, every CU busy.  Prints cycles per wavefront-level
// instruction (4 = full rate for wave64).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(64) void rate(float* out, int iters, uint32_t sel) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    float v[8]; f2 p[8]; uint32_t u[8];
    for (int k = 0; k < 8; k++) { v[k] = 1.0f + threadIdx.x + k; p[k] = f2{v[k], v[k] + 1.0f}; u[k] = threadIdx.x * 977u + k; }
    const float a = 1.0001f, b = 0.5f; const f2 pa = {1.0001f, 0.9999f}, pb = {0.5f, 0.25f};
    for (int i = 0; i < iters; i++) {
#define FMA(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
#define PKFMA(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(pa), "v"(pb));
#define PERM(k) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[k]) : "v"(u[(k + 1) & 7]), "v"(sel));
#define CVT(k) asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(v[k]) : "v"(u[k]));
#define MAX3(k) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
#define MINF(k) asm volatile("v_min_f32 %0, %0, %1" : "+v"(v[k]) : "v"(a));
#define PKMUL(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pa));
#define CNDMASK(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[k]) : "v"(a) : );
        if (OP == 0) { REP8(FMA) }
        if (OP == 1) { REP8(PKFMA) }
        if (OP == 2) { REP8(PERM) }
        if (OP == 3) { REP8(CVT) }
        if (OP == 4) { REP8(MAX3) }
        if (OP == 5) { REP8(MINF) }
        if (OP == 6) { REP8(PKMUL) }
    }
    float s = 0; for (int k = 0; k < 8; k++) s += v[k] + p[k].x + p[k].y + (float)u[k];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, float* d_out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * 4 * 4, iters = 100000;
    hipLaunchKernelGGL(rate<OP>, dim3(blocks), dim3(64), 0, 0, d_out, 100, 0x07010002u);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(rate<OP>, dim3(blocks), dim3(64), 0, 0, d_out, iters, 0x07010002u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    int clk_khz = 0; (void)hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    // per SIMD: 4 waves x iters x 8 instructions
    const double insts = 4.0 * iters * 8.0, cycles = ms * 1e-3 * clk_khz * 1e3;
    printf("%-14s %8.3f ms   %.2f cycles per wave-level instruction (clock %d MHz)\n", name, ms, cycles / insts, clk_khz / 1000);
}

int main() {
    float* d_out; (void)hipMalloc(&d_out, 1 << 24);
    for (int round = 0; round < 3; round++) {
    run<0>("v_fma_f32", d_out); run<1>("v_pk_fma_f32", d_out); run<2>("v_perm_b32", d_out); run<3>("v_cvt_sdwa", d_out);
    run<4>("v_max3_f32", d_out); run<5>("v_min_f32", d_out); run<6>("v_pk_mul_f32", d_out); run<0>("v_fma_f32", d_out);
    }
    return 0;
}
