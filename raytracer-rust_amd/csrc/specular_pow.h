// specular_pow.h -- x^y for the Blinn-Phong term (src/raytracer/raytracer.rs:173 of the reference).
//
// The reference calls f32::powf, which on Linux is the C library's powf.  glibc (2.28 and later) computes it with the
// algorithm of ARM's optimized-routines (Szabolcs Nagy): log2(x) in binary64 from a 16-entry table {1/c, log2(c)} and a
// degree-5 polynomial, times y, then 2^(that) from a 32-entry table and a degree-3 polynomial, rounded to binary32 once at
// the end.  This file is that algorithm with its published constants, written for both sides: the render kernel runs it
// in binary64 on the GPU (some 55 instruction slots where the device library's float-float powf takes 175), and a host
// test runs the same source against the C library.  Every binary64 operation below is an IEEE operation with one
// rounding, in the order of the C library's build for CPUs with fused multiply-add (a*b+c fused where the source has
// it), so the result is that library's, bit for bit, wherever it takes its main path: x a positive normal number and y
// neither zero, infinite nor NaN.  pow_specular_domain() says whether an argument pair is on that path; everything else
// (zero, negative, subnormal, infinite x ...) is left to the caller's general powf.
#ifndef RT_SPECULAR_POW_H
#define RT_SPECULAR_POW_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define RT_POW_FN __host__ __device__ inline
#else
#define RT_POW_FN inline
#endif

namespace rtpow {

// 64 words of 8 bytes: the {1/c, log2(c)} pairs of the 16 subintervals of [0x1.66p-1, 0x1.66p0), then
// 2^(i/32) for i = 0 .. 31 with i << 47 subtracted from the bit pattern (the exponent of the result is added back in
// the same place).  One word per lane of a wavefront when the kernel copies them into LDS.
constexpr int TABLE_WORDS = 64;
constexpr uint64_t TABLES[TABLE_WORDS] = {
    0x3ff661ec79f8f3beull, 0xbfdefec65b963019ull, 0x3ff571ed4aaf883dull, 0xbfdb0b6832d4fca4ull,
    0x3ff49539f0f010b0ull, 0xbfd7418b0a1fb77bull, 0x3ff3c995b0b80385ull, 0xbfd39de91a6dcf7bull,
    0x3ff30d190c8864a5ull, 0xbfd01d9bf3f2b631ull, 0x3ff25e227b0b8ea0ull, 0xbfc97c1d1b3b7af0ull,
    0x3ff1bb4a4a1a343full, 0xbfc2f9e393af3c9full, 0x3ff12358f08ae5baull, 0xbfb960cbbf788d5cull,
    0x3ff0953f419900a7ull, 0xbfaa6f9db6475fceull, 0x3ff0000000000000ull, 0x0000000000000000ull,
    0x3fee608cfd9a47acull, 0x3fb338ca9f24f53dull, 0x3feca4b31f026aa0ull, 0x3fc476a9543891baull,
    0x3feb2036576afce6ull, 0x3fce840b4ac4e4d2ull, 0x3fe9c2d163a1aa2dull, 0x3fd40645f0c6651cull,
    0x3fe886e6037841edull, 0x3fd88e9c2c1b9ff8ull, 0x3fe767dcf5534862ull, 0x3fdce0a44eb17bccull,
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
};

RT_POW_FN uint32_t bits_of(float v) { uint32_t u; memcpy(&u, &v, 4); return u; }
RT_POW_FN float float_of(uint32_t u) { float v; memcpy(&v, &u, 4); return v; }
RT_POW_FN uint64_t bits_of(double v) { uint64_t u; memcpy(&u, &v, 8); return u; }
RT_POW_FN double double_of(uint64_t u) { double v; memcpy(&v, &u, 8); return v; }

// The polynomial coefficients.  In device code each is put together from two 32-bit literals in scalar registers right
// where it is used (the empty asm keeps the compiler from loading all of them into vector registers once per thread and
// carrying them through the whole path loop, which costs the render kernel spills); the host takes the literal.
#if defined(__HIP_DEVICE_COMPILE__)
template <uint64_t BITS>
__device__ __forceinline__ double coefficient() {
    uint32_t lo = (uint32_t)BITS, hi = (uint32_t)(BITS >> 32);
    asm volatile("" : "+s"(lo), "+s"(hi));
    return double_of(((uint64_t)hi << 32) | lo);
}
#define RT_POW_COEFF(x) rtpow::coefficient<__builtin_bit_cast(uint64_t, (double)(x))>()
#else
#define RT_POW_COEFF(x) (x)
#endif

// x a positive normal number, y not zero, not infinite, not NaN: the C library's main path.
RT_POW_FN bool pow_specular_domain(float x, float y) {
    const uint32_t ix = bits_of(x), iy = bits_of(y);
    return ix - 0x00800000u < 0x7f800000u - 0x00800000u && 2u * iy - 1u < 2u * 0x7f800000u - 1u;
}

// x^y on that path.  `tables` points at the TABLE_WORDS words above (the kernel's copy in LDS).
RT_POW_FN float pow_specular(float x, float y, const uint64_t* tables) {
    const uint32_t ix = bits_of(x);
    // x = 2^k z with z in [0x1.66p-1, 0x1.66p0), split into 16 subintervals with centres c:  log2(x) = log2(z/c) + log2(c) + k
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) & 15u;
    const uint32_t top = tmp & 0xff800000u;
    const int32_t k = (int32_t)top >> 23;
    const double invc = double_of(tables[2u * i]), logc = double_of(tables[2u * i + 1u]);
    const double z = (double)float_of(ix - top);
    const double r = __builtin_fma(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double r2 = r * r;
    double p1 = __builtin_fma(RT_POW_COEFF(0x1.27616c9496e0bp-2), r, RT_POW_COEFF(-0x1.71969a075c67ap-2));
    const double p2 = __builtin_fma(RT_POW_COEFF(0x1.ec70a6ca7baddp-2), r, RT_POW_COEFF(-0x1.7154748bef6c8p-1));
    const double r4 = r2 * r2;
    double q = __builtin_fma(RT_POW_COEFF(0x1.71547652ab82bp0), r, y0);
    q = __builtin_fma(p2, r2, q);
    const double log2x = __builtin_fma(p1, r4, q);
    const double ylogx = (double)y * log2x;
    if (((bits_of(ylogx) >> 47) & 0xffffu) >= (bits_of(126.0) >> 47)) {      // |y log2 x| >= 126 (or NaN)
        if (ylogx > 0x1.fffffffd1d571p+6) return __builtin_huge_valf();      // overflow
        if (ylogx <= -150.0) return 0.0f;                                    // underflow
    }
    // 2^ylogx = 2^(kd) 2^r with kd a multiple of 1/32 and |r| <= 1/64
    const double shift = RT_POW_COEFF(0x1.8p+47);
    double kd = ylogx + shift;
    const uint64_t ki = bits_of(kd);
    kd -= shift;
    const double rr = ylogx - kd;
    const uint64_t t = tables[32u + (uint32_t)(ki & 31u)] + (ki << 47);
    const double s = double_of(t);
    const double zz = __builtin_fma(RT_POW_COEFF(0x1.c6af84b912394p-5), rr, RT_POW_COEFF(0x1.ebfce50fac4f3p-3));
    const double rr2 = rr * rr;
    double v = __builtin_fma(RT_POW_COEFF(0x1.62e42ff0c52d6p-1), rr, 1.0);
    v = __builtin_fma(zz, rr2, v);
    v = v * s;
    return (float)v;
}

}  // namespace rtpow
#endif
