// Dev checker for k_events.hip's div_w: x / 3 and x / 6 computed as  q = x * y;  r = fma(-d, q, x);  q' = fma(r, y, q)  with
// y = RN(1 / d) against the division itself -- every float in the guarded range (and that the guard sends the rest to the
// division), doubles on 2^33 random significands x the exponents the guard admits at its edges and in the middle.
//   gcc -O2 -ffp-contract=off -o /tmp/check_div_const tests/dev/check_div_const.c -lm && /tmp/check_div_const
// (software fma from libm unless built with -mfma: both are correctly rounded)
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static float div_f(float x, float d, float y) { float q = x * y; float r = fmaf(-d, q, x); return fmaf(r, y, q); }
static double div_d(double x, double d, double y) { double q = x * y; double r = fma(-d, q, x); return fma(r, y, q); }

int main(int argc, char **argv) {
    const int quick = argc > 1;       // any argument: a sampled run of a few seconds (the test suite's)
    unsigned long long bad = 0, n = 0;
    const float dfs[2] = {3.0f, 6.0f};
    for (int k = 0; k < 2; ++k) {
        const float d = dfs[k], y = 1.0f / d;
        for (uint64_t u = 0; u < (1ull << 32); u += quick ? 97 : 1) {
            uint32_t b = (uint32_t)u; float x; memcpy(&x, &b, 4);
            const float ax = fabsf(x);
            if (!(ax >= 0x1p-100f && ax <= 0x1p100f)) continue;
            volatile float want = x / d;
            const float got = div_f(x, d, y);
            if (memcmp(&got, (const void *)&want, 4)) { if (bad++ < 10) printf("float %a / %g: %a != %a\n", x, d, got, want); }
            ++n;
        }
    }
    printf("float: %llu quotients checked, %llu differ\n", n, bad);
    unsigned long long badd = 0, nd = 0;
    const double dds[2] = {3.0, 6.0};
    const int exps[5] = {-900, -899, 0, 7, 899};
    uint64_t s = 0x9E3779B97F4A7C15ull;
    const uint64_t reps = quick ? (1ull << 22) : (1ull << 33) / 10;
    for (int k = 0; k < 2; ++k) {
        const double d = dds[k], y = 1.0 / d;
        for (uint64_t i = 0; i < reps; ++i) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            const double m = 1.0 + (double)(s >> 12) * 0x1p-52;       // every significand is reachable
            for (int e = 0; e < 5; ++e) {
                const double x = ldexp((s & 1) ? -m : m, exps[e]);
                volatile double want = x / d;
                const double got = div_d(x, d, y);
                if (memcmp(&got, (const void *)&want, 8)) { if (badd++ < 10) printf("double %a / %g: %a != %a\n", x, d, got, want); }
                ++nd;
            }
        }
        // the significands next to the powers of two and the all-ones one, every exponent of the range
        for (int e = -900; e < 900; ++e)
            for (int j = -4; j <= 4; ++j) {
                const double x = ldexp(j < 0 ? 2.0 + j * 0x1p-52 : 1.0 + j * 0x1p-52, e);
                volatile double want = x / d;
                const double got = div_d(x, d, y);
                if (memcmp(&got, (const void *)&want, 8)) { if (badd++ < 10) printf("double %a / %g: %a != %a\n", x, d, got, want); }
                ++nd;
            }
    }
    printf("double: %llu quotients checked, %llu differ\n", nd, badd);
    return (bad || badd) ? 1 : 0;
}
