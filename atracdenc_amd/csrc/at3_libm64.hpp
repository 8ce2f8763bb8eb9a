// glibc 2.35's f64 log() and exp() restated operation by operation, for CalcSpectralFlatnessPerBfu
// (atrac/atrac_psy_common.cpp:184 std::log, :194 std::exp): the reference's `flat < 0.01f` decision
// (atrac3denc.cpp:606) is made on values these two library routines produce, so "same result" means their bits.
//
// What is restated is the code x86-64 hosts with FMA execute (the ifunc variants __ieee754_log_fma /
// __ieee754_exp_fma of sysdeps/ieee754/dbl-64/e_log.c, e_exp.c - ARM optimized-routines): the placement of every fused
// multiply-add is read from the disassembly of e_log-fma.o / e_exp-fma.o in libm-2.35.a (GCC contracts `a * b + c`
// there), the constant data comes from the same archive (tools/gen_libm_f64.py -> at3_libm64.inc).
// tests/test_libm64.py compiles this header on the host and compares it with libm bit for bit.
//
// Domain: what the flatness measure can feed them. log: finite x >= 2^-1022 (its inputs are >= 1e-12f);
// exp: |x| < 512 (its input is a mean of logarithms of floats). Outside of it the functions return NaN instead of
// reproducing libm's special cases.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIP__)
#include <hip/hip_runtime.h>
#define AT3_HD __host__ __device__ __forceinline__
#else
#define AT3_HD static inline
#endif

namespace at3 {

struct Libm64 {            // constant data, see at3_libm64.inc
    double log_c[18];      // ln2hi, ln2lo, A[0..4], B[0..10]
    double log_tab[128][2];   // {invc, logc}
    double exp_c[8];       // invln2N, shift, negln2hiN, negln2loN, C2, C3, C4, C5
    uint64_t exp_tab[128][2];   // {tail bits, scale bits}
};

AT3_HD double l64_from_bits(uint64_t u)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __longlong_as_double((long long)u);
#else
    double d;
    memcpy(&d, &u, 8);
    return d;
#endif
}
AT3_HD uint64_t l64_bits(double d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint64_t)__double_as_longlong(d);
#else
    uint64_t u;
    memcpy(&u, &d, 8);
    return u;
#endif
}
AT3_HD double l64_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// __ieee754_log_fma
AT3_HD double at3_log(const Libm64* L, double x)
{
    const uint64_t ix = l64_bits(x);
    const double* A = L->log_c + 2;
    const double* B = L->log_c + 7;
    // 1 - 2^-4 <= x < 1 + 0x1.09p-4: a degree-11 polynomial in r = x - 1 with the leading terms in double-double
    if (ix - 0x3fee000000000000ull < 0x3ff1090000000000ull - 0x3fee000000000000ull) {
        if (ix == 0x3ff0000000000000ull) return 0.0;
        const double r = x - 1.0;
        const double r2 = r * r;
        const double r3 = r * r2;
        const double q0 = l64_fma(r2, B[3], l64_fma(r, B[2], B[1]));
        const double q1 = l64_fma(r2, B[6], l64_fma(r, B[5], B[4]));
        double q2 = l64_fma(r2, B[9], l64_fma(r, B[8], B[7]));
        q2 = l64_fma(r3, B[10], q2);
        double y = l64_fma(q2, r3, q1);
        y = l64_fma(y, r3, q0);
        const double t = l64_fma(r, 0x1p27, r);
        const double rhi = l64_fma(-0x1p27, r, t);
        const double rlo = r - rhi;
        const double rhi2 = rhi * rhi;
        const double hi = l64_fma(rhi2, B[0], r);
        double lo = l64_fma(rhi2, B[0], r - hi);
        lo = l64_fma(B[0] * rlo, r + rhi, lo);
        y = l64_fma(y, r3, lo);
        return hi + y;
    }
    if ((ix >> 48) - 0x0010u >= 0x7ff0u - 0x0010u) return l64_from_bits(0x7ff8000000000000ull);   // outside the domain
    // x = 2^k z, z in [0x1.69555p-1, 0x1.69555p0): log(x) = log1p(z / c - 1) + log(c) + k ln2, c near the centre of z's
    // one of 128 subintervals
    const uint64_t tmp = ix - 0x3fe6000000000000ull;
    const int i = (int)((tmp >> 45) & 127u);
    const int k = (int)((int64_t)tmp >> 52);
    const uint64_t iz = ix - (tmp & (0xfffull << 52));
    const double invc = L->log_tab[i][0], logc = L->log_tab[i][1];
    const double z = l64_from_bits(iz);
    const double r = l64_fma(z, invc, -1.0);
    const double kd = (double)k;
    const double w = l64_fma(kd, L->log_c[0], logc);
    const double hi = w + r;
    const double lo = l64_fma(kd, L->log_c[1], (w - hi) + r);
    const double r2 = r * r;
    const double p = l64_fma(l64_fma(r, A[4], A[3]), r2, l64_fma(r, A[2], A[1]));
    const double y = l64_fma(r * r2, p, l64_fma(r2, A[0], lo));
    return y + hi;
}

// __ieee754_exp_fma
AT3_HD double at3_exp(const Libm64* L, double x)
{
    const uint32_t abstop = (uint32_t)(l64_bits(x) >> 52) & 0x7ffu;
    if (abstop - 0x3c9u >= 0x3fu) {
        if (abstop < 0x3c9u) return 1.0 + x;   // |x| < 2^-54
        return l64_from_bits(0x7ff8000000000000ull);   // |x| >= 512, inf, nan: outside the domain
    }
    // x = k ln2 / 128 + r: exp(x) = 2^(k / 128) exp(r)
    const double kd0 = l64_fma(x, L->exp_c[0], L->exp_c[1]);
    const uint64_t ki = l64_bits(kd0);
    const double kd = kd0 - L->exp_c[1];
    const double r = l64_fma(kd, L->exp_c[3], l64_fma(kd, L->exp_c[2], x));
    const int idx = (int)(ki & 127u);
    const double tail = l64_from_bits(L->exp_tab[idx][0]);
    const uint64_t sbits = L->exp_tab[idx][1] + (ki << 45);
    const double r2 = r * r;
    double tmp = l64_fma(l64_fma(r, L->exp_c[5], L->exp_c[4]), r2, tail + r);
    tmp = l64_fma(r2 * r2, l64_fma(r, L->exp_c[7], L->exp_c[6]), tmp);
    const double scale = l64_from_bits(sbits);
    return l64_fma(scale, tmp, scale);
}

}  // namespace at3
