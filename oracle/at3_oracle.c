/*
 * TEST INFRASTRUCTURE ONLY - see at3_oracle.h.
 *
 * Scalar C restatement of the ATRAC3 encode hot path of dcherednik/atracdenc.
 * Every function cites the reference location (relative to /root/reference/src) whose
 * arithmetic it restates. The arithmetic contract: IEEE fp32, no FMA
 * contraction, sequential accumulation order as in the reference, tables from the host libm.
 * Build: gcc -std=c11 -O2 -ffp-contract=off (oracle/Makefile).
 */
#define _GNU_SOURCE
#include "at3_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef struct { float r, i; } cpx;

/* ------------------------------------------------------------------------------------------
 * Constant tables
 * ---------------------------------------------------------------------------------------- */

/* BFU layout: atrac/at3/atrac3.h:83-105 */
static const uint16_t kBfuStart[33] = {
    0, 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160, 176, 192, 224, 256, 288, 320, 352,
    384, 416, 448, 480, 512, 576, 640, 704, 768, 896, 1024};
static const uint8_t kBlocksPerBand[5] = {0, 18, 26, 30, 32};
static const uint8_t kClcLen[8] = {0, 4, 3, 3, 4, 4, 5, 6};          /* atrac3.h:93 */
static const float kMaxQuant[8] = {0.0f, 1.5f, 2.5f, 3.5f, 4.5f, 7.5f, 15.5f, 31.5f}; /* atrac3.h:79-82 */
/* atrac3_bitstream.cpp:44-49 */
static const uint8_t kFixedAlloc[32] = {6, 6, 5, 4, 4, 4, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3,
                                        2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 0, 0, 0};

/* Huffman tables (code, bits) per selector 1..7: atrac3.h:106-176. Flattened. */
typedef struct { uint8_t code, bits; } huff_t;
static const huff_t kHuff1[9] = {{0x0, 1}, {0x4, 3}, {0x5, 3}, {0xC, 4}, {0xD, 4}, {0x1C, 5}, {0x1D, 5}, {0x1E, 5}, {0x1F, 5}};
static const huff_t kHuff2[5] = {{0x0, 1}, {0x4, 3}, {0x5, 3}, {0x6, 3}, {0x7, 3}};
static const huff_t kHuff3[7] = {{0x0, 1}, {0x4, 3}, {0x5, 3}, {0xC, 4}, {0xD, 4}, {0xE, 4}, {0xF, 4}};
static const huff_t kHuff5[15] = {{0x0, 2}, {0x2, 3}, {0x3, 3}, {0x8, 4}, {0x9, 4}, {0xA, 4}, {0xB, 4}, {0x1C, 5},
                                  {0x1D, 5}, {0x3C, 6}, {0x3D, 6}, {0x3E, 6}, {0x3F, 6}, {0xC, 4}, {0xD, 4}};
static const huff_t kHuff6[31] = {{0x0, 3}, {0x2, 4}, {0x3, 4}, {0x4, 4}, {0x5, 4}, {0x6, 4}, {0x7, 4}, {0x14, 5},
                                  {0x15, 5}, {0x16, 5}, {0x17, 5}, {0x18, 5}, {0x19, 5}, {0x34, 6}, {0x35, 6}, {0x36, 6},
                                  {0x37, 6}, {0x38, 6}, {0x39, 6}, {0x3A, 6}, {0x3B, 6}, {0x78, 7}, {0x79, 7}, {0x7A, 7},
                                  {0x7B, 7}, {0x7C, 7}, {0x7D, 7}, {0x7E, 7}, {0x7F, 7}, {0x8, 4}, {0x9, 4}};
static const huff_t kHuff7[63] = {
    {0x0, 3},  {0x8, 5},  {0x9, 5},  {0xA, 5},  {0xB, 5},  {0xC, 5},  {0xD, 5},  {0xE, 5},  {0xF, 5},  {0x10, 5}, {0x11, 5},
    {0x24, 6}, {0x25, 6}, {0x26, 6}, {0x27, 6}, {0x28, 6}, {0x29, 6}, {0x2A, 6}, {0x2B, 6}, {0x2C, 6}, {0x2D, 6}, {0x2E, 6},
    {0x2F, 6}, {0x30, 6}, {0x31, 6}, {0x32, 6}, {0x33, 6}, {0x68, 7}, {0x69, 7}, {0x6A, 7}, {0x6B, 7}, {0x6C, 7}, {0x6D, 7},
    {0x6E, 7}, {0x6F, 7}, {0x70, 7}, {0x71, 7}, {0x72, 7}, {0x73, 7}, {0x74, 7}, {0x75, 7}, {0xEC, 8}, {0xED, 8}, {0xEE, 8},
    {0xEF, 8}, {0xF0, 8}, {0xF1, 8}, {0xF2, 8}, {0xF3, 8}, {0xF4, 8}, {0xF5, 8}, {0xF6, 8}, {0xF7, 8}, {0xF8, 8}, {0xF9, 8},
    {0xFA, 8}, {0xFB, 8}, {0xFC, 8}, {0xFD, 8}, {0xFE, 8}, {0xFF, 8}, {0x2, 4},  {0x3, 4}};
static const huff_t* const kHuffTab[7] = {kHuff1, kHuff2, kHuff3, kHuff1, kHuff5, kHuff6, kHuff7};

/* QMF prototype half (qmf/qmf.cpp:25-32) */
static const float kTapHalf[24] = {
    -0.00001461907,  -0.00009205479, -0.000056157569, 0.00030117269, 0.0002422519,  -0.00085293897,
    -0.0005205574,   0.0020340169,   0.00078333891,   -0.0042153862, -0.00075614988, 0.0078402944,
    -0.000061169922, -0.01344162,    0.0024626821,    0.021736089,   -0.007801671,   -0.034090221,
    0.01880949,      0.054326009,    -0.043596379,    -0.099384367,  0.13207909,     0.46424159};

/* Absolute threshold of hearing, millibel table (Musepack), atrac/atrac_psy_common.cpp:43-83 */
static const short kAthTab[] = {
    9669, 9669, 9626, 9512, 9353, 9113, 8882, 8676, 8469, 8243, 7997, 7748, 7492, 7239, 7000, 6762, 6529, 6302, 6084, 5900,
    5717, 5534, 5351, 5167, 5004, 4812, 4638, 4466, 4310, 4173, 4050, 3922, 3723, 3577, 3451, 3281, 3132, 3036, 2902, 2760,
    2658, 2591, 2441, 2301, 2212, 2125, 2018, 1900, 1770, 1682, 1594, 1512, 1430, 1341, 1260, 1198, 1136, 1057, 998,  943,
    887,  846,  744,  712,  693,  668,  637,  606,  580,  555,  529,  502,  475,  448,  422,  398,  375,  351,  327,  322,
    312,  301,  291,  268,  246,  215,  182,  146,  107,  61,   13,   -35,  -96,  -156, -179, -235, -295, -350, -401, -421,
    -446, -499, -532, -535, -513, -476, -431, -313, -179, 8,    203,  403,  580,  736,  881,  1022, 1154, 1251, 1348, 1421,
    1479, 1399, 1285, 1193, 1287, 1519, 1914, 2369, 3352, 4352, 5352, 6352, 7352, 8352, 9352, 9999, 9999, 9999, 9999, 9999};

static struct {
    int ready;
    float qmf_win[48];
    float scale[64];
    float enc_win[256];
    float gain_level[16];
    float gain_interp[31];
    float mdct_sincos[256];
    cpx tw128[128];     /* forward 128-pt (MDCT) */
    cpx tw256[256];     /* forward 256-pt (rfft-512 core) */
    cpx stw256[128];    /* rfft-512 super twiddles */
    cpx tw2048[2048];   /* inverse 2048-pt (irfft-4096 core) */
    cpx stw2048[1024];  /* irfft-4096 super twiddles */
    float planck[512];
    float hpf_w[3];     /* raised-cosine transition weights i=0..2 */
    float loud_curve[1024];
    float ath_spec[1024];
    float ath_bfu[32];
} T;

/* atrac_psy_common.cpp:33-95 */
static float ath_formula_frank(float freq)
{
    if (freq < 10.) freq = 10.;
    if (freq > 29853.) freq = 29853.;
    const double freq_log = 40. * log10(0.1 * freq);
    const unsigned index = (unsigned)freq_log;
    return 0.01 * (kAthTab[index] * (1 + index - freq_log) + kAthTab[index + 1] * (freq_log - index));
}

static void init_tables(void)
{
    if (T.ready) return;
    /* qmf.cpp:41-44 */
    for (int i = 0; i < 24; ++i) T.qmf_win[i] = T.qmf_win[47 - i] = kTapHalf[i] * 2.0;
    /* atrac3.h:178-198 */
    for (uint32_t i = 0; i < 64; ++i) T.scale[i] = pow(2.0, (double)(i / 3.0 - 21.0));
    for (int i = 0; i < 256; ++i) T.enc_win[i] = (sin(((i + 0.5) / 256.0 - 0.5) * M_PI) + 1.0);
    for (int i = 0; i < 16; ++i) T.gain_level[i] = pow(2.0, 4 - i);
    for (int i = 0; i < 31; ++i) T.gain_interp[i] = pow(2.0, -1.0 / 8 * (i - 15));
    /* lib/mdct/mdct.cpp:25-36 with n=512, scale=1 (float overloads of sqrt/cos/sin) */
    {
        const size_t n = 512;
        const float alpha = 2.0 * M_PI / (8.0 * n);
        const float omiga = 2.0 * M_PI / n;
        float scale = 1.0f;
        scale = sqrtf(scale / n);
        for (size_t i = 0; i < (n >> 2); ++i) {
            T.mdct_sincos[2 * i + 0] = scale * cosf(omiga * i + alpha);
            T.mdct_sincos[2 * i + 1] = scale * sinf(omiga * i + alpha);
        }
    }
    /* kiss_fft.c:357-363 */
    const double pi = 3.141592653589793238462643383279502884197169399375105820974944;
    for (int i = 0; i < 128; ++i) {
        const double ph = -2 * pi * i / 128;
        T.tw128[i].r = (float)cos(ph); T.tw128[i].i = (float)sin(ph);
    }
    for (int i = 0; i < 256; ++i) {
        const double ph = -2 * pi * i / 256;
        T.tw256[i].r = (float)cos(ph); T.tw256[i].i = (float)sin(ph);
    }
    for (int i = 0; i < 2048; ++i) {
        double ph = -2 * pi * i / 2048;
        ph *= -1;
        T.tw2048[i].r = (float)cos(ph); T.tw2048[i].i = (float)sin(ph);
    }
    /* tools/kiss_fftr.c:51-57 */
    for (int i = 0; i < 128; ++i) {
        const double ph = -3.14159265358979323846264338327 * ((double)(i + 1) / 256 + .5);
        T.stw256[i].r = (float)cos(ph); T.stw256[i].i = (float)sin(ph);
    }
    for (int i = 0; i < 1024; ++i) {
        double ph = -3.14159265358979323846264338327 * ((double)(i + 1) / 2048 + .5);
        ph *= -1;
        T.stw2048[i].r = (float)cos(ph); T.stw2048[i].i = (float)sin(ph);
    }
    /* transient_spectral_upsampler.cpp:52-68 (Planck taper, eps = 0.15) */
    {
        const float eN = 0.15f * 512.0f;
        const float fN = 512.0f;
        for (int n = 0; n < 512; ++n) {
            const float fn = (float)n;
            if (n == 0) {
                T.planck[n] = 0.0f;
            } else if (fn < eN) {
                const float Zp = eN * (1.0f / fn + 1.0f / (fn - eN));
                T.planck[n] = 1.0f / (1.0f + expf(Zp));
            } else if (fn <= fN - eN) {
                T.planck[n] = 1.0f;
            } else {
                const float m = fN - fn;
                const float Zp = eN * (1.0f / m + 1.0f / (m - eN));
                T.planck[n] = 1.0f / (1.0f + expf(Zp));
            }
        }
        for (int i = 0; i < 3; ++i) T.hpf_w[i] = 0.5f * (1.0f - cosf((float)M_PI * i / 2.0f));
    }
    /* atrac_psy_common.cpp:142-156 */
    for (size_t i = 0; i < 1024; ++i) {
        float f = (float)(i + 3) * 0.5 * 44100 / (float)1024;
        float t = log10f(f) - 3.5;
        t = -10 * t * t + 3 - f / 3000;
        t = pow(10, (0.1 * t));
        T.loud_curve[i] = t;
    }
    /* atrac_psy_common.cpp:126-140 */
    {
        const float mf = (float)44100 / 2000.0;
        for (size_t i = 0; i < 1024; ++i) {
            const float f = (float)(i + 1) * mf / 1024;
            float trh = ath_formula_frank(1.e3 * f) - 100;
            trh -= f * f * 0.015;
            T.ath_spec[i] = trh;
        }
    }
    /* atrac3_bitstream.cpp:705-717 */
    for (int b = 0; b < 32; ++b) {
        float x = 999;
        for (int line = kBfuStart[b]; line < kBfuStart[b + 1]; ++line) x = fminf(x, T.ath_spec[line]);
        x = pow(10, 0.1f * x);
        T.ath_bfu[b] = x;
    }
    T.ready = 1;
}

void at3o_tables(float* scale64, float* encwin256, float* gainlevel16, float* gaininterp31, float* qmfwin48,
                 float* loud1024, float* ath1024)
{
    init_tables();
    if (scale64) memcpy(scale64, T.scale, sizeof(T.scale));
    if (encwin256) memcpy(encwin256, T.enc_win, sizeof(T.enc_win));
    if (gainlevel16) memcpy(gainlevel16, T.gain_level, sizeof(T.gain_level));
    if (gaininterp31) memcpy(gaininterp31, T.gain_interp, sizeof(T.gain_interp));
    if (qmfwin48) memcpy(qmfwin48, T.qmf_win, sizeof(T.qmf_win));
    if (loud1024) memcpy(loud1024, T.loud_curve, sizeof(T.loud_curve));
    if (ath1024) memcpy(ath1024, T.ath_spec, sizeof(T.ath_spec));
}

/* ------------------------------------------------------------------------------------------
 * log2f: glibc 2.35 (sysdeps/ieee754/flt-32/e_log2f.c, from ARM optimized-routines) restated,
 * in the FMA-contracted form its x86-64 `__log2f_fma` ifunc variant executes on FMA-capable
 * hosts (verified instruction by instruction against libm-2.35.a and exhaustively against
 * log2f() on this container's CPU by tools/check_log2f.c). Used for the reference's
 * std::log2(float) calls: atrac3_bitstream.cpp:264-270, atrac3denc.cpp:277,285-286,526-527.
 * ---------------------------------------------------------------------------------------- */
static const double kLog2fTab[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
    {0x1.49539f0f010b0p+0, -0x1.7418b0a1fb77bp-2}, {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8ea0p+0, -0x1.97c1d1b3b7af0p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1.0000000000000p+0, 0x0.0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aa0p-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
    {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};
static const double kLog2fPoly[4] = {-0x1.712b6f70a7e4dp-2, 0x1.ecabf496832e0p-2, -0x1.715479ffae3dep-1,
                                     0x1.715475f35c8b8p+0};

float at3o_log2f(float x)
{
    uint32_t ix;
    memcpy(&ix, &x, 4);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2 == 0) return -INFINITY;
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return NAN;
        const float xs = x * 0x1p23f;
        memcpy(&ix, &xs, 4);
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) % 16;
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int32_t)tmp >> 23;
    float zf;
    memcpy(&zf, &iz, 4);
    const double z = (double)zf;
    const double r = fma(z, kLog2fTab[i][0], -1.0);
    const double y0 = kLog2fTab[i][1] + (double)k;
    const double r2 = r * r;
    double y = fma(kLog2fPoly[1], r, kLog2fPoly[2]);
    y = fma(kLog2fPoly[0], r2, y);
    const double p = fma(kLog2fPoly[3], r, y0);
    y = fma(y, r2, p);
    return (float)y;
}

/* ------------------------------------------------------------------------------------------
 * kissfft-order complex FFT (lib/fft/kissfft_impl/kiss_fft.c:21-90, 238-302) for sizes whose
 * factorisation is 4,4,...,[2] (kf_factor :309-330): decimation in time, butterflies
 * recombine from the leaves up; arithmetic per butterfly exactly as kf_bfly4 / kf_bfly2.
 * ---------------------------------------------------------------------------------------- */
static inline cpx cmul(cpx a, cpx b)
{
    cpx m;
    m.r = a.r * b.r - a.i * b.i;
    m.i = a.r * b.i + a.i * b.r;
    return m;
}

static void fft_combine2(cpx* F, int m, int fstride, const cpx* tw)
{
    for (int k = 0; k < m; ++k) {
        const cpx t = cmul(F[m + k], tw[k * fstride]);
        F[m + k].r = F[k].r - t.r;
        F[m + k].i = F[k].i - t.i;
        F[k].r += t.r;
        F[k].i += t.i;
    }
}

static void fft_combine4(cpx* F, int m, int fstride, const cpx* tw, int inverse)
{
    for (int k = 0; k < m; ++k) {
        const cpx s0 = cmul(F[m + k], tw[k * fstride]);
        const cpx s1 = cmul(F[2 * m + k], tw[2 * k * fstride]);
        const cpx s2 = cmul(F[3 * m + k], tw[3 * k * fstride]);
        cpx s5, s3, s4;
        s5.r = F[k].r - s1.r; s5.i = F[k].i - s1.i;
        F[k].r += s1.r; F[k].i += s1.i;
        s3.r = s0.r + s2.r; s3.i = s0.i + s2.i;
        s4.r = s0.r - s2.r; s4.i = s0.i - s2.i;
        F[2 * m + k].r = F[k].r - s3.r; F[2 * m + k].i = F[k].i - s3.i;
        F[k].r += s3.r; F[k].i += s3.i;
        if (inverse) {
            F[m + k].r = s5.r - s4.i; F[m + k].i = s5.i + s4.r;
            F[3 * m + k].r = s5.r + s4.i; F[3 * m + k].i = s5.i - s4.r;
        } else {
            F[m + k].r = s5.r + s4.i; F[m + k].i = s5.i - s4.r;
            F[3 * m + k].r = s5.r - s4.i; F[3 * m + k].i = s5.i + s4.r;
        }
    }
}

/* n = product of radices; recursion mirrors kf_work: p sub-transforms of length m on
 * inputs decimated by fstride*p, then one combine pass. */
static void fft_rec(cpx* out, const cpx* in, int n, int fstride, const cpx* tw, int inverse)
{
    const int p = (n % 4 == 0) ? 4 : 2;
    const int m = n / p;
    if (m == 1) {
        for (int q = 0; q < p; ++q) out[q] = in[q * fstride];
    } else {
        for (int q = 0; q < p; ++q) fft_rec(out + q * m, in + q * fstride, m, fstride * p, tw, inverse);
    }
    if (p == 4) fft_combine4(out, m, fstride, tw, inverse);
    else fft_combine2(out, m, fstride, tw);
}

/* ------------------------------------------------------------------------------------------
 * QMF analysis (qmf/qmf.h:47-64) and the ATRAC3 three-filter tree (atrac/at3/atrac3_qmf.h:37-41)
 * ---------------------------------------------------------------------------------------- */
typedef struct { float hist[46]; } qmf_state;

static void qmf_analysis(qmf_state* st, const float* in, int n_in, float* lower, float* upper)
{
    float buf[1024 + 46];
    memcpy(buf, st->hist, sizeof(st->hist));
    memcpy(buf + 46, in, sizeof(float) * n_in);
    for (int j = 0; j < n_in; j += 2) {
        float lo = 0.0f, hi = 0.0f;
        for (int i = 0; i < 24; ++i) {
            lo += T.qmf_win[2 * i] * buf[48 - 1 + j - (2 * i)];
            hi += T.qmf_win[2 * i + 1] * buf[48 - 1 + j - (2 * i) - 1];
        }
        upper[j / 2] = lo - hi;
        lower[j / 2] = lo + hi;
    }
    memcpy(st->hist, buf + n_in, sizeof(st->hist));
}

typedef struct { qmf_state q1, q2, q3; } qmf_tree;

static void qmf_tree_analysis(qmf_tree* t, const float* pcm1024, float* subs[4])
{
    float lo[512], hi[512];
    qmf_analysis(&t->q1, pcm1024, 1024, lo, hi);
    qmf_analysis(&t->q2, lo, 512, subs[0], subs[1]);
    qmf_analysis(&t->q3, hi, 512, subs[3], subs[2]);
}

void at3o_qmf(const float* pcm, int nblocks, float* sub)
{
    init_tables();
    qmf_tree t;
    memset(&t, 0, sizeof(t));
    for (int b = 0; b < nblocks; ++b) {
        float* p[4];
        for (int k = 0; k < 4; ++k) p[k] = sub + (size_t)k * nblocks * 256 + (size_t)b * 256;
        qmf_tree_analysis(&t, pcm + (size_t)b * 1024, p);
    }
}

/* ------------------------------------------------------------------------------------------
 * MDCT-512 (lib/mdct/mdct.h:51-104)
 * ---------------------------------------------------------------------------------------- */
static void mdct512(const float* in, float* out)
{
    cpx fin[128], fout[128];
    const float* cs = T.mdct_sincos;
    for (int n = 0; n < 128; n += 2) {
        const float r0 = in[383 - n] + in[384 + n];
        const float i0 = in[128 + n] - in[127 - n];
        const float c = cs[n], s = cs[n + 1];
        fin[n / 2].r = r0 * c + i0 * s;
        fin[n / 2].i = i0 * c - r0 * s;
    }
    for (int n = 128; n < 256; n += 2) {
        const float r0 = in[383 - n] - in[n - 128];
        const float i0 = in[128 + n] + in[639 - n];
        const float c = cs[n], s = cs[n + 1];
        fin[n / 2].r = r0 * c + i0 * s;
        fin[n / 2].i = i0 * c - r0 * s;
    }
    fft_rec(fout, fin, 128, 1, T.tw128, 0);
    for (int n = 0; n < 256; n += 2) {
        const float r0 = fout[n / 2].r, i0 = fout[n / 2].i;
        const float c = cs[n], s = cs[n + 1];
        out[n] = -r0 * c - i0 * s;
        out[255 - n] = -r0 * s + i0 * c;
    }
}

void at3o_mdct512(const float* in512, float* out256)
{
    init_tables();
    mdct512(in512, out256);
}

/* ------------------------------------------------------------------------------------------
 * Gain modulation (gain_processor.h:87-121) and the windowed MDCT wrapper (atrac3denc.cpp:33-58)
 * ---------------------------------------------------------------------------------------- */
typedef struct { int n; uint32_t level[8]; uint32_t loc[8]; } curve_t;

/* Per-sample divisor of the *new* half for a curve; positions past the last ramp are left
 * untouched by Modulate (marked by 0 here). Same running product as gain_processor.h:93-112. */
static void build_mod_levels(const curve_t* c, float lev[256], uint8_t touched[256])
{
    memset(touched, 0, 256);
    uint32_t pos = 0;
    for (int i = 0; i < c->n; ++i) {
        const uint32_t lastPos = c->loc[i] << 3;
        float level = T.gain_level[c->level[i]];
        const int incPos = ((i + 1) < c->n ? (int)c->level[i + 1] : 4) - (int)c->level[i] + 15;
        const float gainInc = T.gain_interp[incPos];
        for (; pos < lastPos; pos++) { lev[pos] = level; touched[pos] = 1; }
        for (; pos < lastPos + 8; pos++) { lev[pos] = level; touched[pos] = 1; level *= gainInc; }
    }
}

static void mdct_band(float* spec256, float* band512, const curve_t* c, int band)
{
    float tmp[512];
    memcpy(tmp, band512, 256 * sizeof(float));
    if (c && c->n > 0) {
        float lev[256];
        uint8_t touched[256];
        const float scale = T.gain_level[c->level[0]];
        build_mod_levels(c, lev, touched);
        for (int i = 0; i < 256; ++i) {
            tmp[i] /= scale;
            if (touched[i]) band512[256 + i] /= lev[i];
        }
    }
    for (int i = 0; i < 256; ++i) {
        const float x = band512[256 + i];
        band512[i] = T.enc_win[i] * x;
        tmp[256 + i] = T.enc_win[255 - i] * x;
    }
    mdct512(tmp, spec256);
    if (band & 1) {
        for (int i = 0, j = 255; i < 128; ++i, --j) {
            const float t = spec256[i]; spec256[i] = spec256[j]; spec256[j] = t;
        }
    }
}

void at3o_mdct(float* specs, float* bands, const int32_t* n_points, const int32_t* level, const int32_t* loc)
{
    init_tables();
    for (int b = 0; b < 4; ++b) {
        curve_t c;
        c.n = n_points ? n_points[b] : 0;
        for (int i = 0; i < c.n; ++i) { c.level[i] = level[b * 8 + i]; c.loc[i] = loc[b * 8 + i]; }
        mdct_band(specs + 256 * b, bands + 512 * b, &c, b);
    }
}

/* atrac3denc.cpp:143-152 */
static float safe_energy_scale(float orig, float mod)
{
    const float eps = 1.0e-20f;
    if (orig <= eps || mod <= eps || !isfinite(orig) || !isfinite(mod)) return 1.0f;
    const float scale = orig / mod;
    return (isfinite(scale) && scale > 0.0f) ? scale : 1.0f;
}

/* atrac3denc.cpp:154-173 */
static void build_sample_divisors(const curve_t* c, float outDiv[256])
{
    for (int i = 0; i < 256; ++i) outDiv[i] = 1.0f;
    uint32_t pos = 0;
    for (int i = 0; i < c->n; ++i) {
        const uint32_t lastPos = c->loc[i] << 3;
        float level = T.gain_level[c->level[i]];
        const int incPos = ((i + 1) < c->n ? (int)c->level[i + 1] : 4) - (int)c->level[i] + 15;
        const float gainInc = T.gain_interp[incPos];
        for (; pos < lastPos && pos < 256; ++pos) outDiv[pos] = level;
        for (; pos < lastPos + 8 && pos < 256; ++pos) { outDiv[pos] = level; level *= gainInc; }
    }
}

typedef struct { float prev_half, cur_half, frame, next_overlap; } ges_t;

/* atrac3denc.cpp:175-224 */
static ges_t calc_gain_energy_scale(const float* prevOverlap, const float* cur, const curve_t* c, float prevScale)
{
    ges_t res;
    if (!isfinite(prevScale) || prevScale <= 0.0f) prevScale = 1.0f;
    const float prevDiv = (c->n == 0) ? 1.0f : T.gain_level[c->level[0]];
    float prevStored = 0.0f;
    for (int i = 0; i < 256; ++i) prevStored += prevOverlap[i] * prevOverlap[i];
    const float prevOrig = prevStored * prevScale;
    const float prevMod = prevStored / (prevDiv * prevDiv);
    float div[256];
    build_sample_divisors(c, div);
    float curO = 0.0f, curM = 0.0f, nextO = 0.0f, nextM = 0.0f;
    for (int i = 0; i < 256; ++i) {
        const float x = cur[i];
        const float mod = x / div[i];
        const float winCur = T.enc_win[255 - i];
        const float winNext = T.enc_win[i];
        const float curWin = x * winCur;
        const float modCurWin = mod * winCur;
        const float nextWin = x * winNext;
        const float modNextWin = mod * winNext;
        curO += curWin * curWin;
        curM += modCurWin * modCurWin;
        nextO += nextWin * nextWin;
        nextM += modNextWin * modNextWin;
    }
    res.prev_half = safe_energy_scale(prevOrig, prevMod);
    res.cur_half = safe_energy_scale(curO, curM);
    res.frame = safe_energy_scale(prevOrig + curO, prevMod + curM);
    res.next_overlap = safe_energy_scale(nextO, nextM);
    return res;
}

void at3o_gain_energy_scale(const float* prevOverlap, const float* cur, int n_points, const int32_t* level,
                            const int32_t* loc, float prevScale, float* out)
{
    init_tables();
    curve_t c;
    c.n = n_points;
    for (int i = 0; i < n_points; ++i) { c.level[i] = level[i]; c.loc[i] = loc[i]; }
    const ges_t r = calc_gain_energy_scale(prevOverlap, cur, &c, prevScale);
    out[0] = r.prev_half; out[1] = r.cur_half; out[2] = r.frame; out[3] = r.next_overlap;
}

/* ------------------------------------------------------------------------------------------
 * Spectral upsampler (transient_spectral_upsampler.cpp:77-180) on kiss_fftr / kiss_fftri
 * (tools/kiss_fftr.c:61-153)
 * ---------------------------------------------------------------------------------------- */
#define LOW_CUT_BIN 38 /* ceil(800 * 512 / 11025), transient_spectral_upsampler.cpp:33 */

static void rfft512(const float* timedata, cpx* freq /* 257 */)
{
    cpx tmp[256];
    fft_rec(tmp, (const cpx*)timedata, 256, 1, T.tw256, 0);
    const float tr = tmp[0].r, ti = tmp[0].i;
    freq[0].r = tr + ti;
    freq[256].r = tr - ti;
    freq[256].i = freq[0].i = 0;
    for (int k = 1; k <= 128; ++k) {
        const cpx fpk = tmp[k];
        cpx fpnk; fpnk.r = tmp[256 - k].r; fpnk.i = -tmp[256 - k].i;
        cpx f1k, f2k;
        f1k.r = fpk.r + fpnk.r; f1k.i = fpk.i + fpnk.i;
        f2k.r = fpk.r - fpnk.r; f2k.i = fpk.i - fpnk.i;
        const cpx tw = cmul(f2k, T.stw256[k - 1]);
        freq[k].r = (f1k.r + tw.r) * .5;
        freq[k].i = (f1k.i + tw.i) * .5;
        freq[256 - k].r = (f1k.r - tw.r) * .5;
        freq[256 - k].i = (tw.i - f1k.i) * .5;
    }
}

static void irfft4096(const cpx* freq /* 2049 */, float* timedata /* 4096 */)
{
    static _Thread_local cpx tmp[2048];
    tmp[0].r = freq[0].r + freq[2048].r;
    tmp[0].i = freq[0].r - freq[2048].r;
    for (int k = 1; k <= 1024; ++k) {
        const cpx fk = freq[k];
        cpx fnkc; fnkc.r = freq[2048 - k].r; fnkc.i = -freq[2048 - k].i;
        cpx fek, t, fok;
        fek.r = fk.r + fnkc.r; fek.i = fk.i + fnkc.i;
        t.r = fk.r - fnkc.r; t.i = fk.i - fnkc.i;
        fok = cmul(t, T.stw2048[k - 1]);
        tmp[k].r = fek.r + fok.r; tmp[k].i = fek.i + fok.i;
        tmp[2048 - k].r = fek.r - fok.r; tmp[2048 - k].i = fek.i - fok.i;
        tmp[2048 - k].i *= -1;
    }
    fft_rec((cpx*)timedata, tmp, 2048, 1, T.tw2048, 1);
}

static float upsample(const float* in, float* out4096)
{
    float windowed[512];
    cpx fwd[257];
    static _Thread_local cpx inv[2049];
    for (int n = 0; n < 512; ++n) windowed[n] = in[n] * T.planck[n];
    rfft512(windowed, fwd);

    double totalE = 0.0, filtHighE = 0.0;
    for (int k = 0; k <= 256; ++k) {
        const double e = (double)fwd[k].r * fwd[k].r + (double)fwd[k].i * fwd[k].i;
        totalE += e;
        float H = 0.0f;
        if (k >= LOW_CUT_BIN + 2) H = 1.0f;
        else if (k >= LOW_CUT_BIN) H = T.hpf_w[k - LOW_CUT_BIN + 1];
        filtHighE += e * H * H;
    }
    const float hfr = (totalE > 0.0) ? (float)(filtHighE / totalE) : 0.0f;

    memset(inv, 0, sizeof(inv));
    const float scale = 8.0f;
    for (int k = LOW_CUT_BIN + 2; k < 256; ++k) { inv[k].r = fwd[k].r * scale; inv[k].i = fwd[k].i * scale; }
    for (int i = 1; i < 3; ++i) {
        const int k = LOW_CUT_BIN - 1 + i;
        const float w = T.hpf_w[i];
        inv[k].r = fwd[k].r * scale * w;
        inv[k].i = fwd[k].i * scale * w;
    }
    inv[256].r = fwd[256].r * scale * 0.5f;
    inv[256].i = 0.0f;
    irfft4096(inv, out4096);
    const float norm = 1.0f / 4096.0f;
    for (int i = 0; i < 4096; ++i) out4096[i] *= norm;
    return hfr;
}

void at3o_upsample(const float* in512, float* out4096, float* hfr)
{
    init_tables();
    *hfr = upsample(in512, out4096);
}

/* ------------------------------------------------------------------------------------------
 * AnalyzeGain / CalcCurve (transient_detector.cpp:33-40, 95-136, 141-482)
 * ---------------------------------------------------------------------------------------- */
static float rms(const float* in, uint32_t n)
{
    float s = 0;
    for (uint32_t i = 0; i < n; i++) s += (in[i] * in[i]);
    s /= n;
    return sqrtf(s);
}

static void sort_floats(float* a, int n)
{
    for (int i = 1; i < n; ++i) {
        const float v = a[i];
        int j = i - 1;
        while (j >= 0 && a[j] > v) { a[j + 1] = a[j]; --j; }
        a[j + 1] = v;
    }
}

static void analyze_gain(const float* in, uint32_t len, uint32_t maxPoints, float* gain, float* lo, float* hi)
{
    const uint32_t step = len / maxPoints;
    uint32_t idx = 0;
    for (uint32_t pos = 0; pos < len; pos += step, ++idx) {
        gain[idx] = rms(in + pos, step);
        if (lo || hi) {
            const uint32_t chunkSz = (step / 8) > 1u ? (step / 8) : 1u;
            float micro[64];
            int nm = 0;
            for (uint32_t off = 0; off < step; off += chunkSz) {
                const uint32_t n = chunkSz < (step - off) ? chunkSz : (step - off);
                micro[nm++] = rms(in + pos + off, n);
            }
            sort_floats(micro, nm);
            if (lo) lo[idx] = micro[nm / 4];
            if (hi) hi[idx] = micro[(nm * 3) / 4];
        }
    }
}

void at3o_analyze_gain(const float* in, int len, int maxPoints, float* gain, float* lo, float* hi)
{
    analyze_gain(in, (uint32_t)len, (uint32_t)maxPoints, gain, lo, hi);
}

/* util.h:65-76: index of highest set bit, 0 for x == 0 */
static uint16_t first_set_bit(uint32_t x)
{
    uint16_t r = 0;
    while (x >>= 1) ++r;
    return r;
}

/* transient_detector.cpp:141-149 (the static variant used inside CalcCurve) */
static uint16_t relation_to_idx(float x)
{
    if (x <= 0.5f) {
        x = 1.0f / fmaxf(x, 0.00048828125f);
        return 4u + first_set_bit((uint32_t)x);
    } else {
        x = fminf(x, 16.0f);
        return 4u - first_set_bit((uint32_t)x);
    }
}

/* atrac3denc.h:44-52 (the header variant used for point 0) */
static uint16_t relation_to_idx_hdr(float x)
{
    if (x <= 0.5) {
        x = 1.0 / fmaxf(x, (float)0.00048828125);
        return 4 + first_set_bit((uint32_t)(int32_t)truncf(x));
    } else {
        x = fminf(x, (float)16.0);
        return 4 - first_set_bit((uint32_t)(int32_t)truncf(x));
    }
}

int at3o_relation_to_idx_hdr(float x) { return relation_to_idx_hdr(x); }

/* transient_detector.cpp:151-166, Radius = 1 */
static void median3(const float* in, float* out, int n)
{
    for (int i = 0; i < n; ++i) {
        const int lo = i - 1 > 0 ? i - 1 : 0;
        const int hi = i + 1 < n - 1 ? i + 1 : n - 1;
        float w[3];
        int wn = 0;
        for (int j = lo; j <= hi; ++j) w[wn++] = in[j];
        sort_floats(w, wn);
        out[i] = w[wn / 2];
    }
}

typedef struct { float last_level, last_hpf, last_target; } curve_ctx;

/* transient_detector.cpp:255-274 */
static float boundary_score(const float* env, int n, int loc, int win)
{
    const int leftStart = loc - win > 0 ? loc - win : 0;
    const int rightEnd = loc + win < n ? loc + win : n;
    float leftMax = 0.0f, rightMax = 0.0f;
    for (int i = leftStart; i < loc; ++i) leftMax = fmaxf(leftMax, env[i]);
    for (int i = loc; i < rightEnd; ++i) rightMax = fmaxf(rightMax, env[i]);
    const float eps = 1e-9f;
    const float attack = (rightMax + eps) / (leftMax + eps);
    const float release = (leftMax + eps) / (rightMax + eps);
    return fmaxf(attack, release);
}

/* transient_detector.cpp:276-482 (in.size() == 32) */
static int calc_curve(const float* in, curve_ctx* ctx, float minScore, const float* lo, const float* hi, curve_t* out)
{
    enum { n = 32 };
    out->n = 0;
    float filtered[n];

    /* FindPlateau(in, 3): :178-238 */
    float maxRaw = 0.0f;
    for (int i = 0; i < n; ++i) maxRaw = fmaxf(maxRaw, in[i]);
    median3(in, filtered, n);
    float bestLevel = 0.0f;
    int bestEnd = -1;
    for (int j = 0; j + 3 <= n; ++j) {
        float minVal = filtered[j];
        for (int k = 1; k < 3; ++k) minVal = fminf(minVal, filtered[j + k]);
        if (minVal > bestLevel) { bestLevel = minVal; bestEnd = j + 2; }
    }
    float plateau = 0.0f;
    int releaseAtEnd = 0;
    if (!(bestLevel < 1e-6f)) {
        plateau = bestLevel;
        while (bestEnd + 1 < n && filtered[bestEnd + 1] >= bestLevel) ++bestEnd;
        if (bestEnd < n - 1) {
            if (in[n - 1] < bestLevel * 0.1f) {
                releaseAtEnd = 1;
            } else {
                int anyHighAfter = 0;
                for (int i = bestEnd + 1; i < n; ++i)
                    if (in[i] >= bestLevel * 0.7f) { anyHighAfter = 1; break; }
                releaseAtEnd = !anyHighAfter && (in[n - 1] < bestLevel * 0.5f);
            }
        }
    }
    const int usePlateau = plateau > 1e-6f && !releaseAtEnd && plateau >= maxRaw * 0.4f;
    const float target = usePlateau ? plateau : in[n - 1];

    const float savedLastLevel = ctx->last_level;
    const float savedLastTarget = ctx->last_target;
    ctx->last_level = in[n - 1];
    ctx->last_target = target;
    if (target < 1e-6f) return 0;
    if (savedLastLevel < 1e-6f) return 0;

    float maxGain = 0.0f;
    for (int i = 0; i < n; ++i) maxGain = fmaxf(maxGain, in[i]);
    const float intraRatio = maxGain / fmaxf(target, 1e-9f);
    float interRatio = 1.0f;
    if (savedLastTarget > 1e-6f) {
        const float h = fmaxf(savedLastTarget, target);
        const float l = fminf(savedLastTarget, target);
        interRatio = h / fmaxf(l, 1e-9f);
    }
    const int sticky = lo && hi && intraRatio <= 7.0f && interRatio <= 10.0f;

    uint16_t sfLevel[n];
    for (int i = 0; i < n; ++i) {
        const float ratioCenter = filtered[i] / target;
        uint16_t level = relation_to_idx(ratioCenter);
        if (i > 0 && sticky) {
            float ratioLo = lo[i] / target;
            float ratioHi = hi[i] / target;
            if (ratioLo > ratioHi) { const float t = ratioLo; ratioLo = ratioHi; ratioHi = t; }
            const uint16_t idxLo = relation_to_idx(ratioLo);
            const uint16_t idxHi = relation_to_idx(ratioHi);
            const uint16_t minIdx = idxLo < idxHi ? idxLo : idxHi;
            const uint16_t maxIdx = idxLo < idxHi ? idxHi : idxLo;
            const uint16_t prev = sfLevel[i - 1];
            const uint16_t span = maxIdx - minIdx;
            if (span <= 1u && abs((int)level - (int)prev) == 1 && prev >= minIdx && prev <= maxIdx) level = prev;
        }
        sfLevel[i] = level;
    }

    int targetSf = 0;
    for (int sf = n - 2; sf >= 0; --sf)
        if (sfLevel[sf] != 4u) { targetSf = sf + 1; break; }
    if (targetSf == 0) return 0;

    float bscore[n + 1];
    for (int i = 0; i <= n; ++i) bscore[i] = 1.0f;
    for (int loc = 1; loc <= targetSf; ++loc) bscore[loc] = boundary_score(filtered, n, loc, 3);

    struct { int loc; uint16_t level; int delta; } trans[n], tmp;
    int nt = 0;
    {
        uint16_t prev = 4u;
        for (int sf = targetSf - 1; sf >= 0; --sf) {
            const uint16_t lev = sfLevel[sf];
            if (lev != prev) {
                const int loc = sf + 1;
                const int delta = abs((int)lev - (int)prev);
                const int keep = (loc == targetSf) || (delta >= 2) || (bscore[loc] >= minScore);
                if (keep) {
                    trans[nt].loc = loc; trans[nt].level = lev; trans[nt].delta = delta; ++nt;
                    prev = lev;
                }
            }
        }
        for (int i = 0, j = nt - 1; i < j; ++i, --j) { tmp = trans[i]; trans[i] = trans[j]; trans[j] = tmp; }
    }
    if (nt == 0) return 0;

    if (nt > 6) {
        /* stable sort by (delta desc, loc desc) - insertion sort is stable */
        for (int i = 1; i < nt; ++i) {
            tmp = trans[i];
            int j = i - 1;
            while (j >= 0 && ((tmp.delta != trans[j].delta) ? (tmp.delta > trans[j].delta) : (tmp.loc > trans[j].loc))) {
                trans[j + 1] = trans[j]; --j;
            }
            trans[j + 1] = tmp;
        }
        nt = 6;
        for (int i = 1; i < nt; ++i) {
            tmp = trans[i];
            int j = i - 1;
            while (j >= 0 && tmp.loc < trans[j].loc) { trans[j + 1] = trans[j]; --j; }
            trans[j + 1] = tmp;
        }
    }
    out->n = nt;
    for (int i = 0; i < nt; ++i) { out->level[i] = trans[i].level; out->loc[i] = (uint32_t)trans[i].loc; }
    return nt;
}

int at3o_calc_curve(const float* gain32, float* ctx, float minScore, const float* lo, const float* hi,
                    int32_t* level, int32_t* loc)
{
    init_tables();
    curve_ctx c = {ctx[0], ctx[1], ctx[2]};
    curve_t out;
    calc_curve(gain32, &c, minScore, lo, hi, &out);
    ctx[0] = c.last_level; ctx[1] = c.last_hpf; ctx[2] = c.last_target;
    for (int i = 0; i < out.n; ++i) { level[i] = out.level[i]; loc[i] = out.loc[i]; }
    return out.n;
}

/* atrac3denc.cpp:228-255 */
static void build_subframe_divisors(const curve_t* c, float outDiv[32])
{
    float sampleDiv[256];
    build_sample_divisors(c, sampleDiv);
    for (uint32_t sf = 0; sf < 32; ++sf) {
        float sum = 0.0f;
        for (uint32_t s = 0; s < 8; ++s) sum += sampleDiv[sf * 8 + s];
        outDiv[sf] = sum / 8.0f;
    }
}

/* atrac3denc.cpp:259-297 */
static float early_mismatch_score(const float* gain, float target, const curve_t* c)
{
    if (target <= 1e-9f) return 0.0f;
    float div[32];
    build_subframe_divisors(c, div);
    uint32_t maxLoc = 0;
    for (int i = 0; i < c->n; ++i) maxLoc = c->loc[i] > maxLoc ? c->loc[i] : maxLoc;
    uint32_t evalSf = maxLoc + 3 > 3 ? maxLoc + 3 : 3;
    if (evalSf > 32) evalSf = 32;
    const float eps = 1e-9f;
    float fit = 0.0f;
    for (uint32_t sf = 0; sf < evalSf; ++sf) {
        const float mod = gain[sf] / fmaxf(div[sf], eps);
        const float e = at3o_log2f(fmaxf(mod, eps) / fmaxf(target, eps));
        fit += e * e;
    }
    fit /= evalSf;
    float leak = 0.0f, wsum = 0.0f;
    for (uint32_t sf = 0; sf + 1 < evalSf; ++sf) {
        const float a = at3o_log2f(fmaxf(div[sf], eps));
        const float b = at3o_log2f(fmaxf(div[sf + 1], eps));
        const float d = b - a;
        const float w = 0.5f * (gain[sf] + gain[sf + 1]);
        leak += d * d * w;
        wsum += w;
    }
    if (wsum > eps) leak /= wsum;
    return fit + 0.25f * leak;
}

/* atrac3denc.cpp:299-579, one band. `up` = 512 subband samples [prev128|cur256|next128]. */
static void create_band_curve(const float* up, int band, curve_ctx* ctx, curve_t* out)
{
    static _Thread_local float sig[4096];
    out->n = 0;
    const float hfr = upsample(up, sig);
    if (hfr < 0.05f) { ctx->last_level = 0.0f; return; }

    float gain[32], gl[32], gh[32];
    analyze_gain(sig + 1024, 2048, 32, gain, gl, gh);

    float curHpf = 0.0f;
    for (int i = 0; i < 32; ++i) curHpf += gain[i];
    curHpf /= 32.0f;
    const float prevHpf = ctx->last_hpf;
    ctx->last_hpf = curHpf;
    const float hpfRatio = (curHpf > 1e-9f && prevHpf > 1e-9f) ? (prevHpf / curHpf) : 1.0f;
    const float overlapFactor = fminf(1.5f, fmaxf(1.0f, hpfRatio));
    const float dynMinScore = 1.9f * overlapFactor;

    const float prevTarget = ctx->last_target;
    curve_t pts;
    calc_curve(gain, ctx, dynMinScore, gl, gh, &pts);
    const float curTarget = ctx->last_target;
    if (pts.n == 0) return;

    float maxGain = 0.0f;
    for (int i = 0; i < 32; ++i) maxGain = fmaxf(maxGain, gain[i]);
    if (maxGain < 1e-4f) pts.n = 0;
    if (hfr < 0.3f) pts.n = 0;
    if (band >= 3) pts.n = 0;

    if (band < 3) {
        const curve_t before = pts;
        int changed = 0;
        float hpfRmsNextMod = 0.0f;
        int valid = 0;
        if (pts.n > 0 && pts.loc[0] > 0) {
            const uint32_t nBefore = pts.loc[0];
            const float divisor = T.gain_level[pts.level[0]];
            float sum = 0.0f;
            for (uint32_t sf = 0; sf < nBefore; ++sf) sum += gain[sf];
            hpfRmsNextMod = (sum / nBefore) / divisor;
            valid = 1;
        } else if (pts.n == 0) {
            float sum = 0.0f;
            for (int i = 0; i < 32; ++i) sum += gain[i];
            hpfRmsNextMod = sum / 32;
            valid = 1;
        }
        if (valid && prevTarget > 1e-6f && hpfRmsNextMod > 1e-6f) {
            const uint16_t p0 = relation_to_idx_hdr(prevTarget / hpfRmsNextMod);
            int it = -1;
            for (int i = 0; i < pts.n; ++i)
                if (pts.loc[i] == 0) { it = i; break; }
            if (it >= 0) {
                if (pts.level[it] != p0) { pts.level[it] = p0; changed = 1; }
            } else if (p0 != 4 || pts.n > 0) {
                for (int i = pts.n; i > 0; --i) { pts.level[i] = pts.level[i - 1]; pts.loc[i] = pts.loc[i - 1]; }
                pts.level[0] = p0; pts.loc[0] = 0; pts.n++;
                changed = 1;
            }
        }
        if (changed) {
            const float scoreBefore = early_mismatch_score(gain, curTarget, &before);
            const float scoreAfter = early_mismatch_score(gain, curTarget, &pts);
            int keepByBoundary = 0;
            if (valid && prevTarget > 1e-6f && hpfRmsNextMod > 1e-6f) {
                const float x = prevTarget / hpfRmsNextMod;
                const float desired = fminf(fmaxf(x, T.gain_level[15]), T.gain_level[0]);
                const float scaleBefore = T.gain_level[before.n == 0 ? 4 : before.level[0]];
                const float scaleAfter = T.gain_level[pts.n == 0 ? 4 : pts.level[0]];
                const float eps = 1e-9f;
                const float errBefore = fabsf(at3o_log2f(fmaxf(scaleBefore, eps) / fmaxf(desired, eps)));
                const float errAfter = fabsf(at3o_log2f(fmaxf(scaleAfter, eps) / fmaxf(desired, eps)));
                keepByBoundary = (errAfter + 0.20f < errBefore);
            }
            if (!keepByBoundary && scoreAfter > scoreBefore * (1.0f + 0.02f)) pts = before;
        }
    }
    if (pts.n >= 2 && pts.loc[0] == 0 && pts.level[0] == pts.level[1]) {
        for (int i = 1; i < pts.n; ++i) { pts.level[i - 1] = pts.level[i]; pts.loc[i - 1] = pts.loc[i]; }
        pts.n--;
    }
    *out = pts;
}

/* ------------------------------------------------------------------------------------------
 * Scaling and quantisation (atrac/atrac_scale.cpp)
 * ---------------------------------------------------------------------------------------- */
/* atrac_scale.cpp:141-172; lower_bound on the (strictly increasing) ScaleTable */
/* what the reference prints to stderr here, as counts (process-wide; read and reset through at3o_diag_counts):
 * [0] "Scale error: absSpec > MAX_SCALE" (:150-153), [1] "clipping, scaled value" (:163-167) */
static unsigned long long g_diag[2];
void at3o_diag_counts(unsigned long long* out2, int reset)
{
    if (out2) { out2[0] = g_diag[0]; out2[1] = g_diag[1]; }
    if (reset) g_diag[0] = g_diag[1] = 0;
}

static int scale_block(const float* in, int len, float* values, float* energy)
{
    float maxAbs = 0;
    for (int i = 0; i < len; ++i) {
        const float a = fabsf(in[i]);
        if (a > maxAbs) maxAbs = a;
    }
    if (maxAbs > 1.0f) {
        g_diag[0]++;
        maxAbs = 1.0f;
    }
    int sfi = 0;
    while (sfi < 63 && T.scale[sfi] < maxAbs) ++sfi;
    const float sf = T.scale[sfi];
    float e = 0.0f;
    for (int i = 0; i < len; ++i) {
        float v = in[i] / sf;
        e += in[i] * in[i];
        if (fabsf(v) >= 1.0) {
            if (fabsf(v) > 1.0) g_diag[1]++;
            v = (v > 0) ? 0.99999 : -0.99999;
        }
        values[i] = v;
    }
    *energy = e;
    return sfi;
}

void at3o_scale_frame(const float* specs, int32_t* sfi, float* energy, float* values)
{
    init_tables();
    for (int b = 0; b < 32; ++b)
        sfi[b] = scale_block(specs + kBfuStart[b], kBfuStart[b + 1] - kBfuStart[b], values + kBfuStart[b], &energy[b]);
}

/* --- libstdc++ (GCC 11) std::sort order, restated: introsort loop (median-of-3 pivot,
 * unguarded partition, depth limit 2*floor(log2 n) with heap-sort fallback) followed by the
 * final insertion sort with threshold 16. Needed because QuantMantisas (atrac_scale.cpp:79-83)
 * sorts candidates by |delta| only, so the visiting order of equal keys is defined by the
 * library algorithm. Checked against std::sort by tests/test_oracle_sort.py. */
typedef struct { float key; int32_t val; } sitem;
static inline int sless(const sitem* a, const sitem* b) { return fabsf(a->key) < fabsf(b->key); }
static inline void sswap(sitem* a, sitem* b) { const sitem t = *a; *a = *b; *b = t; }

static void s_push_heap(sitem* first, int hole, int top, sitem value)
{
    int parent = (hole - 1) / 2;
    while (hole > top && sless(&first[parent], &value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

static void s_adjust_heap(sitem* first, int hole, int len, sitem value)
{
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (sless(&first[child], &first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    s_push_heap(first, hole, top, value);
}

static void s_heap_sort(sitem* first, int len)
{
    if (len >= 2) {
        int parent = (len - 2) / 2;
        for (;;) {
            const sitem v = first[parent];
            s_adjust_heap(first, parent, len, v);
            if (parent == 0) break;
            parent--;
        }
    }
    while (len > 1) {
        --len;
        const sitem v = first[len];
        first[len] = first[0];
        s_adjust_heap(first, 0, len, v);
    }
}

static void s_unguarded_linear_insert(sitem* last)
{
    const sitem v = *last;
    sitem* next = last - 1;
    while (sless(&v, next)) { *last = *next; last = next; --next; }
    *last = v;
}

static void s_insertion_sort(sitem* first, sitem* last)
{
    if (first == last) return;
    for (sitem* i = first + 1; i != last; ++i) {
        if (sless(i, first)) {
            const sitem v = *i;
            memmove(first + 1, first, (size_t)(i - first) * sizeof(sitem));
            *first = v;
        } else {
            s_unguarded_linear_insert(i);
        }
    }
}

static void s_introsort_loop(sitem* first, sitem* last, int depth)
{
    while (last - first > 16) {
        if (depth == 0) { s_heap_sort(first, (int)(last - first)); return; }
        --depth;
        sitem* mid = first + (last - first) / 2;
        sitem *a = first + 1, *b = mid, *c = last - 1;
        if (sless(a, b)) {
            if (sless(b, c)) sswap(first, b);
            else if (sless(a, c)) sswap(first, c);
            else sswap(first, a);
        } else if (sless(a, c)) sswap(first, a);
        else if (sless(b, c)) sswap(first, c);
        else sswap(first, b);
        sitem* lo = first + 1;
        sitem* hi = last;
        for (;;) {
            while (sless(lo, first)) ++lo;
            --hi;
            while (sless(first, hi)) --hi;
            if (!(lo < hi)) break;
            sswap(lo, hi);
            ++lo;
        }
        s_introsort_loop(lo, last, depth);
        last = lo;
    }
}

static void std_sort_abs(sitem* a, int n)
{
    if (n <= 0) return;
    int lg = 0;
    for (int t = n; t > 1; t >>= 1) ++lg;
    s_introsort_loop(a, a + n, lg * 2);
    if (n > 16) {
        s_insertion_sort(a, a + 16);
        for (sitem* i = a + 16; i != a + n; ++i) s_unguarded_linear_insert(i);
    } else {
        s_insertion_sort(a, a + n);
    }
}

void at3o_sort_abs(float* key, int32_t* payload, int n)
{
    sitem* a = (sitem*)malloc(sizeof(sitem) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) { a[i].key = key[i]; a[i].val = payload[i]; }
    std_sort_abs(a, n);
    for (int i = 0; i < n; ++i) { key[i] = a[i].key; payload[i] = a[i].val; }
    free(a);
}

/* atrac_scale.cpp:40-130 (first = 0) */
static float quant_mantisas(const float* in, int n, float mul, int ea, int* mant)
{
    float e1 = 0.0f, e2 = 0.0f;
    const float inv2 = 1.0 / (mul * mul);
    sitem cand[128];
    int nc = 0;
    for (int j = 0; j < n; ++j) {
        const float t = in[j] * mul;
        e1 += in[j] * in[j];
        mant[j] = (int)lrintf(t);
        e2 += mant[j] * mant[j] * inv2;
        if (ea) {
            const float delta = t - (truncf(t) + 0.5f);
            if (fabsf(delta) < 0.25f) { cand[nc].key = delta; cand[nc].val = j; ++nc; }
        }
    }
    if (!ea || nc == 0) return e1 / e2;
    std_sort_abs(cand, nc);
    if (e2 < e1) {
        for (int c = 0; c < nc; ++c) {
            const int j = cand[c].val;
            const float t = in[j] * mul;
            const float am = (float)abs(mant[j]);
            if (am < fabsf(t) && am < (mul - 1)) {
                int m = mant[j];
                if (m > 0) m++;
                if (m < 0) m--;
                if (m == 0) m = t > 0 ? 1 : -1;
                float ex = e2;
                ex -= mant[j] * mant[j] * inv2;
                ex += m * m * inv2;
                if (fabsf(ex - e1) < fabsf(e2 - e1)) { mant[j] = m; e2 = ex; }
            }
        }
        return e1 / e2;
    }
    if (e2 > e1) {
        for (int c = 0; c < nc; ++c) {
            const int j = cand[c].val;
            const float t = in[j] * mul;
            if ((float)abs(mant[j]) > fabsf(t)) {
                int m = mant[j];
                if (m > 0) m--;
                if (m < 0) m++;
                float ex = e2;
                ex -= mant[j] * mant[j] * inv2;
                ex += m * m * inv2;
                if (fabsf(ex - e1) < fabsf(e2 - e1)) { mant[j] = m; e2 = ex; }
            }
        }
        return e1 / e2;
    }
    return e1 / e2;
}

float at3o_quant_mantisas(const float* in, int n, float mul, int ea, int32_t* mant)
{
    return quant_mantisas(in, n, mul, ea, mant);
}

/* atrac_psy_common.cpp:158-199 */
static void spectral_flatness(const float* energy, float* flat)
{
    const float floor_ = fmaxf(1e-12f, 1e-20f);
    for (int b = 0; b < 32; ++b) {
        const int start = kBfuStart[b], end = kBfuStart[b + 1], len = end - start;
        double arith = 0.0, meanLog = 0.0;
        for (int i = start; i < end; ++i) {
            const double e = fmaxf(0.0f, energy[i]);
            arith += e;
            meanLog += log(e > (double)floor_ ? e : (double)floor_);
        }
        arith /= (double)len;
        meanLog /= (double)len;
        if (arith <= floor_) { flat[b] = 1.0f; continue; }
        const double geom = exp(meanLog);
        const double ratio = geom / arith;
        flat[b] = (float)fmin(1.0, fmax(0.0, ratio));
    }
}

void at3o_flatness(const float* energy1024, float* flat32) { spectral_flatness(energy1024, flat32); }

/* ------------------------------------------------------------------------------------------
 * Bit writer (lib/bitstream/bitstream.cpp:40-63): MSB first
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint8_t buf[4096]; int bits; } bitw;

static void bw_write(bitw* w, uint32_t val, int n)
{
    for (int i = n - 1; i >= 0; --i) {
        if (w->bits < (int)sizeof(w->buf) * 8) {
            if ((val >> i) & 1u) w->buf[w->bits >> 3] |= (uint8_t)(0x80u >> (w->bits & 7));
        }
        w->bits++;
    }
}

/* ------------------------------------------------------------------------------------------
 * Sound-unit encoder (atrac/at3/atrac3_bitstream.cpp)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint16_t pos;       /* position of first component */
    uint8_t bfu;        /* BFU of first component */
    uint8_t len;
    uint8_t sfi;
    float values[8];
} tonal_block;

typedef struct {
    curve_t curve[4];
    float ges_frame[4];
    float loudness;
    int sfi[32];
    float energy[32];
    float values[1024];
    int n_tonal;
    tonal_block tonal[64];
} sce_t;

/* atrac3_bitstream.cpp:92-113 */
static uint32_t clc_enc(uint32_t sel, const int* m, uint32_t n, bitw* w)
{
    const uint32_t nb = kClcLen[sel];
    const uint32_t used = (sel > 1) ? nb * n : nb * n / 2;
    if (!w) return used;
    if (sel > 1) {
        for (uint32_t i = 0; i < n; ++i) bw_write(w, (uint32_t)m[i] & ((1u << nb) - 1u), (int)nb);
    } else {
        static const uint32_t rtab[4] = {2, 3, 0, 1};
        for (uint32_t i = 0; i < n / 2; ++i) {
            uint32_t code = rtab[m[i * 2] + 2] << 2;
            code |= rtab[m[i * 2 + 1] + 2];
            bw_write(w, code, 4);
        }
    }
    return used;
}

/* atrac3_bitstream.cpp:115-149 */
static uint32_t vlc_enc(uint32_t sel, const int* m, uint32_t n, bitw* w)
{
    const huff_t* tab = kHuffTab[sel - 1];
    uint32_t used = 0;
    if (sel > 1) {
        for (uint32_t i = 0; i < n; ++i) {
            const int v = m[i];
            uint32_t h = (v < 0) ? (((uint32_t)(-v)) << 1) | 1 : ((uint32_t)v) << 1;
            if (h) h -= 1;
            used += tab[h].bits;
            if (w) bw_write(w, tab[h].code, tab[h].bits);
        }
    } else {
        static const uint32_t rtab[9] = {8, 4, 7, 2, 0, 1, 6, 3, 5};
        for (uint32_t i = 0; i < n / 2; ++i) {
            const uint32_t h = rtab[3 * (m[i * 2] + 1) + (m[i * 2 + 1] + 1)];
            used += tab[h].bits;
            if (w) bw_write(w, tab[h].code, tab[h].bits);
        }
    }
    return used;
}

/* Quantised-unit cache keyed by (bfu, wordlen): atrac3_bitstream.cpp:154-173 */
typedef struct {
    uint8_t valid;
    float energy_err;
    uint32_t clc_bits, vlc_bits;
    int mant[128];
} spec_unit;

typedef struct {
    const sce_t* sce;
    uint16_t target_bits;
    uint32_t bfu_idx_const;
    float loudness;
    int init_done;
    float spread;
    uint16_t num_bfu;
    uint8_t coding_mode;
    uint32_t prec[32];
    int n_prec;
    float energy_err[32];
    int mant[1024];
    spec_unit cache[32][8];
} enc_ctx;

static const spec_unit* get_unit(enc_ctx* c, int bfu, uint32_t wl)
{
    spec_unit* u = &c->cache[bfu][wl];
    if (!u->valid) {
        const int first = kBfuStart[bfu], n = kBfuStart[bfu + 1] - first;
        const float mul = kMaxQuant[wl < 7 ? wl : 7];
        u->energy_err = quant_mantisas(c->sce->values + first, n, mul, bfu > 18, u->mant);
        u->clc_bits = clc_enc(wl, u->mant, (uint32_t)n, NULL);
        u->vlc_bits = vlc_enc(wl, u->mant, (uint32_t)n, NULL);
        u->valid = 1;
    }
    return u;
}

/* atrac3_bitstream.cpp:190-227 */
static uint32_t specs_bits_consumption(enc_ctx* c, const uint32_t* prec, int n, uint8_t* mode)
{
    uint32_t used = (uint32_t)n * 3, clc = 0, vlc = 0;
    for (int i = 0; i < n; ++i) {
        if (prec[i] == 0) continue;
        used += 6;
        const spec_unit* u = get_unit(c, i, prec[i]);
        memcpy(c->mant + kBfuStart[i], u->mant, sizeof(int) * (size_t)(kBfuStart[i + 1] - kBfuStart[i]));
        c->energy_err[i] = u->energy_err;
        clc += u->clc_bits;
        vlc += u->vlc_bits;
    }
    *mode = clc <= vlc;
    return used + (*mode ? clc : vlc);
}

/* atrac3_bitstream.cpp:241-257 */
static int consider_energy_err(const float* err, uint32_t* bits, int n)
{
    int adjusted = 0;
    const int lim = n < 10 ? n : 10;
    for (int i = 0; i < lim; ++i) {
        const float e = err[i];
        if (((e > 0 && e < 0.7f) || e > 1.2f) & (bits[i] < 7)) { bits[i]++; adjusted = 1; }
    }
    return adjusted;
}

static float sanitize_ges(float s) { return (isfinite(s) && s > 0.0f) ? s : 1.0f; }

/* atrac3_bitstream.cpp:272-336 */
static void calc_bits_allocation(const enc_ctx* c, int n, float spread, float shift, float loudness, uint32_t* bits)
{
    const sce_t* sce = c->sce;
    for (int i = 0; i < n; ++i) {
        uint32_t band = 0;
        for (uint32_t b = 1; b < 4; ++b)
            if ((uint32_t)i >= kBlocksPerBand[b]) band = b;
        const float ges = sanitize_ges(sce->ges_frame[band]);
        const float corrected = sce->energy[i] * ges;
        const float ath = T.ath_bfu[i] * loudness;
        if (corrected < ath) {
            bits[i] = 0;
        } else {
            const uint32_t fix = kFixedAlloc[i];
            float x = 6;
            if (i < 3) x = 2.8;
            else if (i < 10) x = 2.6;
            else if (i < 15) x = 3.3;
            else if (i <= 20) x = 3.6;
            else if (i <= 28) x = 4.2;
            const float csfi = fmaxf(0.0f, fminf(63.0f, (float)sce->sfi[i] + 1.5f * at3o_log2f(ges)));
            const int tmp = spread * (csfi / x) + (1.0f - spread) * fix - shift;
            if (tmp > 7) bits[i] = 7;
            else if (tmp < 0) bits[i] = 0;
            else if (tmp == 0) bits[i] = 1;
            else bits[i] = (uint32_t)tmp;
        }
    }
    for (int t = 0; t < sce->n_tonal; ++t) {
        const int bfu = sce->tonal[t].bfu;
        if (bfu < n && bits[bfu] > 2) bits[bfu] -= 1;
    }
}

/* atrac3_bitstream.cpp:338-524: grouping + (cost | emission) of tonal components */
static uint16_t encode_tonal(const sce_t* sce, const uint32_t* alloc, int n_alloc, bitw* w)
{
    int grp[64][64];
    int gsz[64];
    int sgmap[64][64];
    int sgn[64];
    memset(gsz, 0, sizeof(gsz));
    memset(sgn, 0, sizeof(sgn));
    for (int t = 0; t < sce->n_tonal; ++t) {
        const tonal_block* tb = &sce->tonal[t];
        if ((int)tb->bfu >= n_alloc) continue;
        uint32_t quant = alloc[tb->bfu] + 4;
        if (quant > 7) quant = 7;
        if (quant < 2) quant = 2;
        const int g = (int)quant * 8 + tb->len;
        grp[g][gsz[g]++] = t;
    }
    uint32_t tcsgn = 0;
    for (int i = 0; i < 64; ++i) {
        int startPos, curPos = 0;
        while (curPos < gsz[i]) {
            startPos = curPos;
            ++tcsgn;
            sgmap[i][sgn[i]++] = curPos;
            uint32_t limiter = 0;
            do {
                ++curPos;
                if (curPos == gsz[i]) break;
                if ((int)sce->tonal[grp[i][curPos]].pos - (int)(sce->tonal[grp[i][startPos]].pos & ~63) < 64) {
                    ++limiter;
                } else {
                    limiter = 0;
                    startPos = curPos;
                }
            } while (limiter < 7u);
        }
    }

    uint16_t used = 5;
    if (w) bw_write(w, tcsgn, 5);
    if (tcsgn == 0) return used;
    used += 2;
    if (w) bw_write(w, 0, 2);

    for (int i = 0; i < 64; ++i) {
        if (gsz[i] == 0) continue;
        for (int sg = 0; sg < sgn[i]; ++sg) {
            const int sgStart = sgmap[i][sg];
            const int sgEnd = (sg < sgn[i] - 1) ? sgmap[i][sg + 1] : gsz[i];
            const int codedValues = sce->tonal[grp[i][0]].len;
            uint8_t cnt[16];
            memset(cnt, 0, sizeof(cnt));
            for (int j = sgStart; j < sgEnd; ++j) cnt[sce->tonal[grp[i][j]].pos >> 6]++;
            int bandFlag[4];
            for (int b = 0; b < 4; ++b) bandFlag[b] = cnt[4 * b] | cnt[4 * b + 1] | cnt[4 * b + 2] | cnt[4 * b + 3];
            used += 4;
            if (w) for (int b = 0; b < 4; ++b) bw_write(w, bandFlag[b] != 0, 1);
            used += 3;
            if (w) bw_write(w, (uint32_t)codedValues - 1, 3);
            used += 3;
            if (w) bw_write(w, (uint32_t)i >> 3, 3);
            int lastPos = sgStart;
            for (int j = 0; j < 16; ++j) {
                if (!bandFlag[j >> 2]) continue;
                const int coded = cnt[j];
                used += 3;
                if (w) bw_write(w, (uint32_t)coded, 3);
                int k = lastPos;
                for (; k < lastPos + coded; ++k) {
                    const tonal_block* tb = &sce->tonal[grp[i][k]];
                    const uint32_t relPos = (uint32_t)tb->pos - (uint32_t)j * 64;
                    used += 6;
                    if (w) bw_write(w, tb->sfi, 6);
                    used += 6;
                    if (w) bw_write(w, relPos, 6);
                    int mant[8];
                    const uint32_t q = (uint32_t)i >> 3;
                    const float mul = kMaxQuant[q < 7 ? q : 7];
                    for (int z = 0; z < tb->len; ++z) mant[z] = (int)lrintf(tb->values[z] * mul);
                    used += vlc_enc(q, mant, tb->len, w);
                }
                lastPos = k;
            }
        }
    }
    return used;
}

/* atrac3_bitstream.cpp:526-565 */
static void encode_specs(const enc_ctx* c, bitw* w)
{
    const sce_t* sce = c->sce;
    const int n = c->n_prec;
    encode_tonal(sce, c->prec, n, w);
    bw_write(w, (uint32_t)n - 1, 5);
    bw_write(w, c->coding_mode, 1);
    for (int i = 0; i < n; ++i) bw_write(w, c->prec[i], 3);
    for (int i = 0; i < n; ++i)
        if (c->prec[i]) bw_write(w, (uint32_t)sce->sfi[i], 6);
    for (int i = 0; i < n; ++i) {
        if (!c->prec[i]) continue;
        const int first = kBfuStart[i], len = kBfuStart[i + 1] - first;
        if (c->coding_mode == 1) clc_enc(c->prec[i], c->mant + first, (uint32_t)len, w);
        else vlc_enc(c->prec[i], c->mant + first, (uint32_t)len, w);
    }
}

/* atrac_psy_common.cpp:105-124 */
static float scale_factor_spread(const int* sfi)
{
    float s = 0.0f;
    for (int i = 0; i < 32; ++i) s += sfi[i];
    s /= 32;
    float sigma = 0.0f;
    for (int i = 0; i < 32; ++i) {
        float t = (sfi[i] - s);
        t *= t;
        sigma += t;
    }
    sigma /= 32;
    sigma = sqrtf(sigma);
    if (sigma > 14.0) sigma = 14.0;
    return sigma / 14.0;
}

/* atrac3_bitstream.cpp:567-585 */
static uint16_t initial_num_bfu(uint32_t bfuIdxConst, uint16_t targetBits)
{
    uint16_t numBfu = bfuIdxConst ? (uint16_t)bfuIdxConst : 32;
    if (targetBits < 101) {
        uint16_t lim = 1;
        if (targetBits > 5) lim = (targetBits - 5) / 3;
        if (lim < 1) lim = 1;
        if (numBfu > lim) numBfu = lim;
    }
    return numBfu < 1 ? 1 : numBfu;
}

/* One channel: TConfigure/TAlloc driven by the bisection state machine of
 * lib/bs_encode/encode.cpp:57-129 (Start/Continue/Submit/Repeat semantics), then Dump. */
static void encode_channel(enc_ctx* c, bitw* w)
{
    float minL = 0, maxL = 0, curL = 0, lastL = 0;
    uint32_t alloc[32];
    uint8_t mode = 1;
    for (;;) { /* TConfigure::Encode :589-610 */
        if (!c->init_done) {
            c->spread = scale_factor_spread(c->sce->sfi);
            c->num_bfu = initial_num_bfu(c->bfu_idx_const, c->target_bits);
            c->init_done = 1;
        }
        minL = -8.0f; maxL = 20.0f; lastL = 20.0f;
        int restart = 0;
        for (;;) { /* TAlloc::Encode :621-659 */
            float shift;
            const int exhausted = (maxL <= minL);
            if (exhausted) {
                shift = lastL;
            } else {
                curL = (maxL + minL) / 2.0;
                shift = curL;
            }
            const int n = c->num_bfu;
            calc_bits_allocation(c, n, c->spread, shift, c->loudness, alloc);
            for (int i = 0; i < n; ++i) c->energy_err[i] = 0.0f;
            uint32_t bits;
            do {
                bits = specs_bits_consumption(c, alloc, n, &mode);
            } while (consider_energy_err(c->energy_err, alloc, n));
            const uint32_t total = bits + encode_tonal(c->sce, alloc, n, NULL);
            int done;
            if (exhausted) {
                done = 1;
            } else if (total < c->target_bits) {
                lastL = curL; maxL = curL - 0.01f; done = 0;
            } else if (total > c->target_bits) {
                minL = curL + 0.01f; done = 0;
            } else {
                done = 1;
            }
            if (!done) continue;
            if (!c->bfu_idx_const && c->num_bfu > 1 && alloc[c->num_bfu - 1] == 0) {
                c->num_bfu--;
                restart = 1;
            }
            break;
        }
        if (!restart) break;
    }
    c->n_prec = c->num_bfu;
    memcpy(c->prec, alloc, sizeof(uint32_t) * c->num_bfu);
    c->coding_mode = mode;
    encode_specs(c, w);
}

/* ------------------------------------------------------------------------------------------
 * Encoder object (atrac3denc.cpp:679-867 lambda + atrac3_bitstream.cpp:759-847 WriteSoundUnit)
 * ---------------------------------------------------------------------------------------- */
struct at3o_encoder {
    int bitrate, frame_sz, js, nch, no_gain, no_tonal, bfu_idx_const;
    int look_ahead_pending;
    qmf_tree qmf[2];
    float look_ahead[2][4][640];
    float overlap[2][4][512]; /* [overlap 256 | new 256], TDelayBuffer slot ch + 2*band */
    curve_ctx cctx[2][4];
    float prev_overlap_scale[2][4];
    float loudness;
    sce_t sce[2];
    enc_ctx ectx;
};

static const struct { uint32_t bitrate; uint16_t frame_sz; uint8_t js; } kContainer[8] = {
    {66150, 192, 1}, {93713, 272, 1}, {104738, 304, 0}, {132300, 384, 0},
    {146081, 424, 0}, {176400, 512, 0}, {264600, 768, 0}, {352800, 1024, 0}}; /* atrac3.h:211-220 */

at3o_encoder* at3o_create(int bitrate, int nch, int no_gain, int no_tonal, int bfu_idx_const)
{
    init_tables();
    if (nch != 2 && nch != 1) return NULL;
    at3o_encoder* e = (at3o_encoder*)calloc(1, sizeof(*e));
    if (!e) return NULL;
    uint32_t br = bitrate == 0 ? 132300u : (uint32_t)bitrate;
    int idx = 0;
    while (idx < 7 && kContainer[idx].bitrate < br) ++idx; /* lower_bound, atrac3.cpp:47-53 */
    e->bitrate = (int)kContainer[idx].bitrate;
    e->frame_sz = kContainer[idx].frame_sz;
    e->js = kContainer[idx].js;
    e->nch = nch;
    e->no_gain = no_gain;
    e->no_tonal = no_tonal;
    e->bfu_idx_const = bfu_idx_const;
    e->look_ahead_pending = 1;
    for (int c = 0; c < 2; ++c)
        for (int b = 0; b < 4; ++b) e->prev_overlap_scale[c][b] = 1.0f;
    e->loudness = 0.006f;
    return e;
}

void at3o_destroy(at3o_encoder* e) { free(e); }
int at3o_frame_size(const at3o_encoder* e) { return e->frame_sz; }
int at3o_joint_stereo(const at3o_encoder* e) { return e->js; }

/* atrac3denc.cpp:581-643 */
typedef struct { uint16_t pos; float val; uint8_t bfu; } tonal_val;
static int extract_tonal(float* specs, const float* flat, tonal_val* out)
{
    int n = 0;
    for (uint32_t b = 8; b < 29u; ++b) {
        if (flat[b] >= 0.01f) continue;
        const uint32_t start = kBfuStart[b], end = kBfuStart[b + 1], len = end - start;
        const uint32_t maxLen = 5u < len ? 5u : len;
        float bestScore = -1.0f;
        uint32_t bestStart = start, bestLen = 1;
        for (uint32_t s = start; s < end; ++s) {
            const uint32_t ml = maxLen < end - s ? maxLen : end - s;
            float score = 0.0f;
            for (uint32_t l = 1; l <= ml; ++l) {
                score += fabsf(specs[s + l - 1]);
                if (score > bestScore) { bestScore = score; bestStart = s; bestLen = l; }
            }
        }
        if (bestScore <= 0.0f) continue;
        for (uint32_t k = 0; k < bestLen; ++k) {
            const uint32_t pos = bestStart + k;
            out[n].pos = (uint16_t)pos; out[n].val = specs[pos]; out[n].bfu = (uint8_t)b; ++n;
            specs[pos] = 0.0f;
        }
    }
    return n;
}

/* atrac3denc.cpp:646-662 */
static int map_tonal(const tonal_val* tv, int n, tonal_block* out)
{
    int nb = 0;
    for (int i = 0; i < n;) {
        const int startPos = i;
        uint32_t curPos;
        do {
            curPos = tv[i].pos;
            ++i;
        } while (i < n && tv[i].pos == curPos + 1 && i - startPos < 7);
        const int len = i - startPos;
        float tmp[8], e;
        for (int j = 0; j < len; ++j) tmp[j] = tv[startPos + j].val;
        tonal_block* tb = &out[nb++];
        tb->pos = tv[startPos].pos;
        tb->bfu = tv[startPos].bfu;
        tb->len = (uint8_t)len;
        tb->sfi = (uint8_t)scale_block(tmp, len, tb->values, &e);
    }
    return nb;
}

/* atrac3_bitstream.cpp:732-757 */
static int32_t ms_bytes_shift(uint32_t frameSz, const sce_t* sce, const int32_t b[2])
{
    const int32_t totalUsedBits = 0 - b[0] - b[1];
    const int32_t maxAllowedShift = (int32_t)(frameSz / 2 - (1 + ((uint32_t)totalUsedBits - 1) / 8));
    const float m = sce[0].loudness, s = sce[1].loudness;
    const float total = s + m;
    float ratio = 0;
    if (total > 0) ratio = m / total - 0.5;
    int32_t v = (int32_t)lrintf(frameSz * ratio);
    if (v > maxAllowedShift) v = maxAllowedShift;
    if (v < -maxAllowedShift) v = -maxAllowedShift;
    return v;
}

static void write_sound_unit(at3o_encoder* e, float loudness, unsigned char* out)
{
    const int half = e->frame_sz >> 1;
    static _Thread_local bitw bs[2];
    int32_t bitsToAlloc[2] = {-6, -6};
    /* One input channel in a joint-stereo container: the lambda appends an empty second element with ONE subband
     * and no scaled blocks (atrac3denc.cpp:843-849) so that the frame still carries two sound units. */
    const int mono_js = e->js && e->nch == 1;
    const int nsce = mono_js ? 2 : e->nch;
    memset(bs, 0, sizeof(bs));
    for (int ch = 0; ch < nsce; ++ch) {
        const sce_t* sce = &e->sce[ch];
        bitw* w = &bs[ch];
        const int nqmf = (mono_js && ch == 1) ? 1 : 4;   /* SubbandInfo.GetQmfNum() */
        if (e->js && ch == 1) {
            bw_write(w, 0, 1); bw_write(w, 7, 3);
            for (int i = 0; i < 4; ++i) bw_write(w, 3, 2);
            bw_write(w, 3, 2);
        } else {
            bw_write(w, 0x28, 6);
        }
        bw_write(w, (uint32_t)nqmf - 1, 2);
        for (int band = 0; band < nqmf; ++band) {
            const curve_t* c = &sce->curve[band];
            const int n = (mono_js && ch == 1) ? 0 : c->n;
            bw_write(w, (uint32_t)n, 3);
            for (int i = 0; i < n; ++i) { bw_write(w, c->level[i], 4); bw_write(w, c->loc[i], 5); }
        }
        bitsToAlloc[ch] -= (int16_t)w->bits;
    }
    int32_t shift = 0;
    if (mono_js) {   /* CalcMSBytesShift with elements[1].ScaledBlocks.empty(): the maximum (atrac3_bitstream.cpp:745-747) */
        const int32_t totalUsedBits = 0 - bitsToAlloc[0] - bitsToAlloc[1];
        shift = (int32_t)((uint32_t)e->frame_sz / 2 - (1 + ((uint32_t)totalUsedBits - 1) / 8));
    } else if (e->js) {
        shift = ms_bytes_shift((uint32_t)e->frame_sz, e->sce, bitsToAlloc);
    }
    bitsToAlloc[0] += 8 * (half + shift);
    bitsToAlloc[1] += 8 * (half - shift);

    int outPos = 0;
    for (int ch = 0; ch < nsce; ++ch) {
        enc_ctx* c = &e->ectx;
        memset(c, 0, sizeof(*c));
        c->sce = &e->sce[ch];
        c->target_bits = (uint16_t)(bitsToAlloc[ch] > 1 ? bitsToAlloc[ch] : 1);
        c->bfu_idx_const = (uint32_t)e->bfu_idx_const;
        c->loudness = loudness;
        c->num_bfu = 1;
        c->coding_mode = 1;
        if (mono_js && ch == 1) {
            /* TConfigure / TAlloc with empty ScaledBlocks (atrac3_bitstream.cpp:590-597, 623-626): one BFU of
             * precision 0, coding mode 1, no tonal components - EncodeSpecs writes 5 + 5 + 1 + 3 bits. */
            bitw* w = &bs[ch];
            bw_write(w, 0, 5);       /* tonal sub-group count */
            bw_write(w, 1 - 1, 5);   /* numBlocks - 1 */
            bw_write(w, 1, 1);       /* coding mode */
            bw_write(w, 0, 3);       /* precision of the one block */
        } else {
            encode_channel(c, &bs[ch]);
        }
        if (e->js && ch == 1) {
            const int n = half - shift;
            for (int i = 0; i < n; ++i) out[outPos + i] = bs[ch].buf[n - 1 - i];
            outPos += n;
        } else {
            const int n = half + shift;
            memcpy(out + outPos, bs[ch].buf, (size_t)n);
            outPos += n;
        }
    }
    if (nsce == 1 && !e->js) memcpy(out + half, out, (size_t)half);
}

int at3o_process(at3o_encoder* e, const float* pcm, unsigned char* out, at3o_tap* taps)
{
    const int nch = e->nch;
    const int qmfOffset = e->look_ahead_pending ? 128 : 384;
    for (int ch = 0; ch < nch; ++ch) {
        float src[1024];
        for (int i = 0; i < 1024; ++i) src[i] = pcm[i * nch + ch] / 4.0;
        float* p[4] = {&e->look_ahead[ch][0][qmfOffset], &e->look_ahead[ch][1][qmfOffset],
                       &e->look_ahead[ch][2][qmfOffset], &e->look_ahead[ch][3][qmfOffset]};
        qmf_tree_analysis(&e->qmf[ch], src, p);
    }
    if (e->look_ahead_pending) { e->look_ahead_pending = 0; return 0; }

    for (int ch = 0; ch < nch; ++ch)
        for (int b = 0; b < 4; ++b) memcpy(&e->overlap[ch][b][256], &e->look_ahead[ch][b][128], 256 * sizeof(float));

    const int js = e->js && nch == 2;
    static _Thread_local float jsGain[2][4][512];
    if (js) {
        for (int b = 0; b < 4; ++b) {
            for (int i = 0; i < 512; ++i) {
                const float l = e->look_ahead[0][b][i], r = e->look_ahead[1][b][i];
                jsGain[0][b][i] = (l + r) * 0.5f;
                jsGain[1][b][i] = (l - r) * 0.5f;
            }
            /* Matrixing(): atrac3denc.cpp:665-677 */
            float* p0 = &e->overlap[0][b][256];
            float* p1 = &e->overlap[1][b][256];
            for (int i = 0; i < 256; ++i) {
                const float t0 = p0[i], t1 = p1[i];
                p0[i] = (t0 + t1) / 2.0;
                p1[i] = (t0 - t1) / 2.0;
            }
        }
    }

    for (int ch = 0; ch < nch; ++ch) {
        sce_t* sce = &e->sce[ch];
        float specs[1024];
        sce->n_tonal = 0;
        for (int b = 0; b < 4; ++b) { sce->curve[b].n = 0; sce->ges_frame[b] = 1.0f; }
        if (!e->no_gain) {
            for (int b = 0; b < 4; ++b) {
                const float* up = js ? jsGain[ch][b] : e->look_ahead[ch][b];
                create_band_curve(up, b, &e->cctx[ch][b], &sce->curve[b]);
            }
        }
        for (int b = 0; b < 4; ++b) {
            const ges_t g = calc_gain_energy_scale(&e->overlap[ch][b][0], &e->overlap[ch][b][256], &sce->curve[b],
                                                   e->prev_overlap_scale[ch][b]);
            sce->ges_frame[b] = g.frame;
            e->prev_overlap_scale[ch][b] = g.next_overlap;
        }
        for (int b = 0; b < 4; ++b) mdct_band(specs + 256 * b, e->overlap[ch][b], &sce->curve[b], b);

        float energy[1024];
        float l = 0;
        for (int i = 0; i < 1024; ++i) {
            const float en = specs[i] * specs[i];
            energy[i] = en;
            l += en * sce->ges_frame[i / 256] * T.loud_curve[i];
        }
        sce->loudness = l;
        if (!e->no_tonal) {
            float flat[32];
            tonal_val tv[128];
            spectral_flatness(energy, flat);
            const int ntv = extract_tonal(specs, flat, tv);
            sce->n_tonal = map_tonal(tv, ntv, sce->tonal);
        }
        for (int b = 0; b < 32; ++b)
            sce->sfi[b] = scale_block(specs + kBfuStart[b], kBfuStart[b + 1] - kBfuStart[b], sce->values + kBfuStart[b],
                                      &sce->energy[b]);
    }

    /* atrac_psy_common.h:46-54 */
    if (nch == 2 && !e->js) e->loudness = 0.98 * e->loudness + 0.01 * (e->sce[0].loudness + e->sce[1].loudness);
    else e->loudness = 0.98 * e->loudness + 0.02 * e->sce[0].loudness;

    write_sound_unit(e, e->loudness / 0.006f, out);

    if (taps) {
        for (int ch = 0; ch < nch; ++ch) {
            at3o_tap* t = &taps[ch];
            const sce_t* sce = &e->sce[ch];
            memset(t, 0, sizeof(*t));
            for (int b = 0; b < 4; ++b) {
                t->n_points[b] = sce->curve[b].n;
                for (int i = 0; i < sce->curve[b].n; ++i) { t->level[b][i] = sce->curve[b].level[i]; t->loc[b][i] = sce->curve[b].loc[i]; }
                t->ges_frame[b] = sce->ges_frame[b];
            }
            t->loudness_ch = sce->loudness;
            t->loudness_track = e->loudness;
            for (int b = 0; b < 32; ++b) { t->sfi[b] = sce->sfi[b]; t->energy[b] = sce->energy[b]; }
            memcpy(t->values, sce->values, sizeof(t->values));
            t->n_tonal = sce->n_tonal;
            for (int i = 0; i < sce->n_tonal && i < 64; ++i) {
                t->tonal_pos[i] = sce->tonal[i].pos;
                t->tonal_len[i] = sce->tonal[i].len;
                t->tonal_sfi[i] = sce->tonal[i].sfi;
                for (int j = 0; j < sce->tonal[i].len; ++j) t->tonal_values[i][j] = sce->tonal[i].values[j];
            }
        }
    }

    for (int ch = 0; ch < nch; ++ch)
        for (int b = 0; b < 4; ++b) memmove(e->look_ahead[ch][b], e->look_ahead[ch][b] + 256, 384 * sizeof(float));
    return 1;
}

int at3o_encode(int bitrate, int nch, int no_gain, int no_tonal, int bfu_idx_const, const float* pcm, int nblocks,
                unsigned char* out, int* frame_sz, at3o_tap* taps)
{
    at3o_encoder* e = at3o_create(bitrate, nch, no_gain, no_tonal, bfu_idx_const);
    if (!e) return -1;
    int nf = 0;
    for (int b = 0; b < nblocks; ++b) {
        const int r = at3o_process(e, pcm + (size_t)b * 1024 * nch, out + (size_t)nf * e->frame_sz,
                                   taps ? taps + (size_t)nf * nch : NULL);
        nf += r;
    }
    if (frame_sz) *frame_sz = e->frame_sz;
    at3o_destroy(e);
    return nf;
}
