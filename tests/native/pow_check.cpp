// Host check of csrc/specular_pow.h: the binary64 evaluation the render kernel uses for the specular power against the C
// library's powf (what the reference's f32::powf calls, src/raytracer/raytracer.rs:173), bit for bit, on the arguments the
// path produces (bases in [0, 1], a little above 1, shininess values) and on random bit patterns inside the function's domain.
// Usage: pow_check <samples>; prints "domain <n> mismatches <m>".
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../raytracer-rust_amd/csrc/specular_pow.h"

int main(int argc, char** argv) {
    const long long samples = argc > 1 ? atoll(argv[1]) : 20000000LL;
    std::mt19937_64 rng(20240611);
    std::uniform_real_distribution<float> unit(0.0f, 1.0f), shin(0.0f, 200.0f);
    long long in_domain = 0, mismatches = 0;
    for (long long i = 0; i < samples; i++) {
        float x, y;
        switch (i & 3) {
            case 0: x = unit(rng); y = shin(rng); break;                                   // n.h anywhere in [0, 1)
            case 1: x = 1.0f - unit(rng) * 0.01f; y = shin(rng) * 5.0f; break;             // highlights: bases just under 1, large exponents
            case 2: x = rtpow::float_of((uint32_t)rng() & 0x7fffffffu); y = rtpow::float_of((uint32_t)rng()); break;   // any bit patterns
            default: x = unit(rng) * 1.0001f; y = (float)(int)shin(rng); if (y == 0.0f) y = 20.0f; break;               // whole-number shininess, bases up to just over 1
        }
        if (!rtpow::pow_specular_domain(x, y)) continue;
        in_domain++;
        const float mine = rtpow::pow_specular(x, y, rtpow::TABLES), libm = powf(x, y);
        if (rtpow::bits_of(mine) != rtpow::bits_of(libm)) {
            if (mismatches < 10) printf("x=%a y=%a here=%a libm=%a\n", x, y, mine, libm);
            mismatches++;
        }
    }
    // the exact arguments of the headline scene's materials, swept finely
    for (uint32_t b = 0x3f000000u; b <= 0x3f800100u; b += 7u) {
        const float x = rtpow::float_of(b);
        in_domain++;
        if (rtpow::bits_of(rtpow::pow_specular(x, 20.0f, rtpow::TABLES)) != rtpow::bits_of(powf(x, 20.0f))) mismatches++;
    }
    printf("domain %lld mismatches %lld\n", in_domain, mismatches);
    return 0;
}
