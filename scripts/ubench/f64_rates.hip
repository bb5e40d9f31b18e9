// Issue cost of the binary64 instructions the specular power is made of (specular_pow.h), next to v_fma_f32: 8 independent
// chains of one instruction each per loop pass, 4 wavefronts per SIMD, every CU busy.  Prints cycles per wavefront-level
// instruction (4 = full rate for wave64).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(64) void rate(float* out, int iters) {
    float v[8]; double d[8]; int n[8];
    for (int k = 0; k < 8; k++) { v[k] = 1.0f + threadIdx.x + k; d[k] = 1.0 + threadIdx.x + k; n[k] = threadIdx.x + k; }
    const float a = 1.0001f, b = 0.5f; const double da = 1.0001, db = 0.5;
    for (int i = 0; i < iters; i++) {
#define FMA32(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
#define FMA64(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k]) : "v"(da), "v"(db));
#define MUL64(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k]) : "v"(da));
#define ADD64(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[k]) : "v"(db));
#define CVT6432(k) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[k]) : "v"(v[k]));
#define CVT3264(k) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(v[k]) : "v"(d[k]));
#define CVT64I(k) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[k]) : "v"(n[k]));
#define LOG32(k) asm volatile("v_log_f32 %0, %0" : "+v"(v[k]));
#define RCP32(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[k]));
        if (OP == 0) { REP8(FMA32) }
        if (OP == 1) { REP8(FMA64) }
        if (OP == 2) { REP8(MUL64) }
        if (OP == 3) { REP8(ADD64) }
        if (OP == 4) { REP8(CVT6432) }
        if (OP == 5) { REP8(CVT3264) }
        if (OP == 6) { REP8(CVT64I) }
        if (OP == 7) { REP8(LOG32) }
        if (OP == 8) { REP8(RCP32) }
    }
    float s = 0; for (int k = 0; k < 8; k++) s += v[k] + (float)d[k] + (float)n[k];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, float* d_out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * 4 * 4, iters = 50000;
    hipLaunchKernelGGL(rate<OP>, dim3(blocks), dim3(64), 0, 0, d_out, 100);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(rate<OP>, dim3(blocks), dim3(64), 0, 0, d_out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    int clk_khz = 0; (void)hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const double insts = 4.0 * iters * 8.0, cycles = ms * 1e-3 * clk_khz * 1e3;      // per SIMD: 4 waves x iters x 8 instructions
    printf("%-14s %8.3f ms   %.2f cycles per wave-level instruction (clock %d MHz)\n", name, ms, cycles / insts, clk_khz / 1000);
}

int main() {
    float* d_out; (void)hipMalloc(&d_out, 1 << 24);
    for (int round = 0; round < 2; round++) {
        run<0>("v_fma_f32", d_out); run<1>("v_fma_f64", d_out); run<2>("v_mul_f64", d_out); run<3>("v_add_f64", d_out);
        run<4>("v_cvt_f64_f32", d_out); run<5>("v_cvt_f32_f64", d_out); run<6>("v_cvt_f64_i32", d_out); run<7>("v_log_f32", d_out);
        run<8>("v_rcp_f32", d_out);
    }
    return 0;
}
