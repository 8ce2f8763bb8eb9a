// Host check of atracdenc_amd/csrc/at3_libm64.hpp against this machine's libm (glibc 2.35 on an FMA-capable x86-64:
// the build the reference's std::log / std::exp calls resolve to). Prints the number of inputs tried and of
// mismatching bit patterns per function; exit code 0 when there are none. argv[1] = inputs per family.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../atracdenc_amd/csrc/at3_libm64.hpp"
#include "../../atracdenc_amd/csrc/at3_libm64.inc"

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint64_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}
static double unit() { return (double)(rnd() >> 11) * 0x1p-53; }

int main(int argc, char** argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 2000000;
    if (!__builtin_cpu_supports("fma")) {
        printf("skip: no FMA on this CPU (libm runs its non-FMA variants)\n");
        return 77;
    }
    static at3::Libm64 L;
    memcpy(L.log_c, kLogData, sizeof(L.log_c));
    memcpy(L.log_tab, kLogData + 18, sizeof(L.log_tab));
    memcpy(L.exp_c, kExpData, sizeof(L.exp_c));
    memcpy(L.exp_tab, kExpTab, sizeof(L.exp_tab));
    long bad_log = 0, bad_exp = 0, n_log = 0, n_exp = 0;
    volatile double sink;
    auto chk_log = [&](double x) {
        const double a = at3::at3_log(&L, x), b = log(x);
        ++n_log;
        if (at3::l64_bits(a) != at3::l64_bits(b)) {
            if (bad_log++ < 5) printf("log(%a): %a vs libm %a\n", x, a, b);
        }
        sink = a;
    };
    auto chk_exp = [&](double x) {
        const double a = at3::at3_exp(&L, x), b = exp(x);
        ++n_exp;
        if (at3::l64_bits(a) != at3::l64_bits(b)) {
            if (bad_exp++ < 5) printf("exp(%a): %a vs libm %a\n", x, a, b);
        }
        sink = a;
    };
    for (long i = 0; i < n; ++i) {
        // what the flatness measure feeds log: squares of floats, floored at 1e-12f
        const float f = (float)(unit() * 2.0 - 1.0) * (float)ldexp(1.0, -(int)(rnd() % 40));
        const float e = f * f;
        chk_log((double)(e > 1e-12f ? e : 1e-12f));
        chk_log(ldexp(0.5 + 0.5 * unit(), (int)(rnd() % 200) - 100));    // any normal magnitude
        chk_log(0.9375 + unit() * (1.0 + 0x1.09p-4 - 0.9375));          // the polynomial branch around 1
        chk_log(1.0 + (unit() - 0.5) * ldexp(1.0, -(int)(rnd() % 50)));  // ever closer to 1
        chk_log((double)(1.0f + (float)(unit() - 0.5) * 0.1f));
        chk_exp(-27.7 * unit());                                          // mean of logs of values in [1e-12, 1]
        chk_exp((unit() - 0.5) * 1000.0);                                 // the whole |x| < 512 domain
        chk_exp((unit() - 0.5) * ldexp(1.0, -(int)(rnd() % 60)));        // towards and below 2^-54
    }
    const double edge[] = {1.0, 0.9375, 0x1.fffffffffffffp-1, 1.0 + 0x1p-52, 0x1.109p0, 0x1.108ffffffffffp0, 0x1.dffffffffffffp-1,
                           (double)1e-12f, 0x1p-1022, 0x1.fffffffffffffp1023, 2.0, 0.5};
    for (double x : edge) chk_log(x);
    const double eedge[] = {0.0, -0.0, 0x1p-54, -0x1p-54, 0x1.fffffffffffffp-55, 511.999, -511.999, -27.631021115928547, 1.0, -1.0};
    for (double x : eedge) chk_exp(x);
    printf("log: %ld inputs, %ld mismatches; exp: %ld inputs, %ld mismatches\n", n_log, bad_log, n_exp, bad_exp);
    return (bad_log || bad_exp) ? 1 : 0;
}
