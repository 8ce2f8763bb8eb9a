// Host-side constant tables (see at3_tables.hpp). Plain C++; compiled with contraction off so the
// float expressions round exactly like the reference's x86-64 baseline build.
#include "at3_tables.hpp"
#include "at1_tables.hpp"
#include "at3p_tables.hpp"

#include <cmath>
#include <cstring>

// The table builders must CALL libm at run time, like the reference's static initialisers do: at -O3 clang folds
// sinf / cosf / powf of constant arguments through double precision, which is one ulp off glibc's float routines for
// some entries (seen on the 64-point MDCT's table once its 16-iteration loop got unrolled).
#define AT3_RUNTIME_LIBM __attribute__((optnone, noinline))

namespace at3 {

// QMF prototype, first half (qmf/qmf.cpp:25-32).
extern const float kTapHalf[24];
const float kTapHalf[24] = {
    -0.00001461907,  -0.00009205479, -0.000056157569, 0.00030117269, 0.0002422519,  -0.00085293897,
    -0.0005205574,   0.0020340169,   0.00078333891,   -0.0042153862, -0.00075614988, 0.0078402944,
    -0.000061169922, -0.01344162,    0.0024626821,    0.021736089,   -0.007801671,   -0.034090221,
    0.01880949,      0.054326009,    -0.043596379,    -0.099384367,  0.13207909,     0.46424159};

namespace {

#include "at3_libm64.inc"

// Threshold in quiet, millibel re 20 uPa, 4 steps per third starting at 10 Hz (Musepack table used by
// atrac/atrac_psy_common.cpp:43-83).
const short kAthMilliBel[] = {
    9669, 9669, 9626, 9512, 9353, 9113, 8882, 8676, 8469, 8243, 7997, 7748, 7492, 7239, 7000, 6762, 6529, 6302, 6084, 5900,
    5717, 5534, 5351, 5167, 5004, 4812, 4638, 4466, 4310, 4173, 4050, 3922, 3723, 3577, 3451, 3281, 3132, 3036, 2902, 2760,
    2658, 2591, 2441, 2301, 2212, 2125, 2018, 1900, 1770, 1682, 1594, 1512, 1430, 1341, 1260, 1198, 1136, 1057, 998,  943,
    887,  846,  744,  712,  693,  668,  637,  606,  580,  555,  529,  502,  475,  448,  422,  398,  375,  351,  327,  322,
    312,  301,  291,  268,  246,  215,  182,  146,  107,  61,   13,   -35,  -96,  -156, -179, -235, -295, -350, -401, -421,
    -446, -499, -532, -535, -513, -476, -431, -313, -179, 8,    203,  403,  580,  736,  881,  1022, 1154, 1251, 1348, 1421,
    1479, 1399, 1285, 1193, 1287, 1519, 1914, 2369, 3352, 4352, 5352, 6352, 7352, 8352, 9352, 9999, 9999, 9999, 9999, 9999};

const uint16_t kBfuStart[33] = {0,   8,   16,  24,  32,  40,  48,  56,  64,  80,  96,  112, 128, 144, 160, 176, 192,
                                224, 256, 288, 320, 352, 384, 416, 448, 480, 512, 576, 640, 704, 768, 896, 1024};

}  // namespace

AT3_RUNTIME_LIBM float ath_db(float freq)
{
    if (freq < 10.) freq = 10.;
    if (freq > 29853.) freq = 29853.;
    const double fl = 40. * log10(0.1 * freq);
    const unsigned idx = (unsigned)fl;
    return 0.01 * (kAthMilliBel[idx] * (1 + idx - fl) + kAthMilliBel[idx + 1] * (fl - idx));
}

AT3_RUNTIME_LIBM void fill_twiddles(cpx* tw, int n, bool inverse)
{
    const double pi = 3.141592653589793238462643383279502884197169399375105820974944;
    for (int i = 0; i < n; ++i) {
        double phase = -2 * pi * i / n;
        if (inverse) phase *= -1;
        tw[i].r = (float)cos(phase);
        tw[i].i = (float)sin(phase);
    }
}

namespace {

AT3_RUNTIME_LIBM void fill_super_twiddles(cpx* tw, int ncfft, bool inverse)
{
    for (int i = 0; i < ncfft / 2; ++i) {
        double phase = -3.14159265358979323846264338327 * ((double)(i + 1) / ncfft + .5);
        if (inverse) phase *= -1;
        tw[i].r = (float)cos(phase);
        tw[i].i = (float)sin(phase);
    }
}

}  // namespace

AT3_RUNTIME_LIBM void build_tables(Tables* t)
{
    memset(t, 0, sizeof(*t));
    for (int i = 0; i < 24; ++i) t->qmf_win[i] = t->qmf_win[47 - i] = kTapHalf[i] * 2.0;
    for (uint32_t i = 0; i < 64; ++i) t->scale[i] = pow(2.0, (double)(i / 3.0 - 21.0));
    for (int i = 0; i < 256; ++i) t->enc_win[i] = (sin(((i + 0.5) / 256.0 - 0.5) * M_PI) + 1.0);
    for (int i = 0; i < 16; ++i) t->gain_level[i] = pow(2.0, 4 - i);
    for (int i = 0; i < 31; ++i) t->gain_interp[i] = pow(2.0, -1.0 / 8 * (i - 15));

    {   // MDCT-512 pre/post rotation, scale 1 -> sqrt(1/512) folded in
        const size_t n = 512;
        const float alpha = 2.0 * M_PI / (8.0 * n);
        const float omiga = 2.0 * M_PI / n;
        float scale = 1.0f;
        scale = sqrtf(scale / n);
        for (size_t i = 0; i < (n >> 2); ++i) {
            t->mdct_sincos[2 * i + 0] = scale * cosf(omiga * i + alpha);
            t->mdct_sincos[2 * i + 1] = scale * sinf(omiga * i + alpha);
        }
    }
    fill_twiddles(t->tw128, 128, false);
    fill_twiddles(t->tw256, 256, false);
    fill_twiddles(t->tw2048, 2048, true);
    fill_super_twiddles(t->stw256, 256, false);
    fill_super_twiddles(t->stw2048, 2048, true);
    for (int L = 0; L < 16; ++L) {   // see at3_k_gain.hpp: k_gain_spec (a 256-point transform in a 16-lane row)
        for (int q = 1; q <= 3; ++q) t->spec16_tw[q - 1][L] = t->tw256[4 * q * L];
        for (int j = 0; j < 4; ++j)
            for (int q = 1; q <= 3; ++q) t->spec16_tw[3 + 3 * j + q - 1][L] = t->tw256[q * (L + 16 * j)];
        for (int jj = 0; jj < 9; ++jj) {
            const int k = L + 16 * jj;
            cpx z = {0.0f, 0.0f};
            if (k >= 1 && k <= 128 && (jj < 8 || L == 0)) z = t->stw256[k - 1];
            t->spec16_stw[jj][L] = z;
        }
    }
    for (int tid = 0; tid < 128; ++tid) {   // see at3_k_gain.hpp: irfft_pass_32_128 / irfft_pass_512
        const int k = tid & 31;
        for (int q = 0; q < 3; ++q) t->gain_tw[q][tid] = t->tw2048[16 * (q + 1) * k];
        for (int j = 0; j < 4; ++j)
            for (int q = 0; q < 3; ++q) t->gain_tw[3 + 3 * j + q][tid] = t->tw2048[4 * (q + 1) * (k + 32 * j)];
        for (int u = 0; u < 4; ++u)
            for (int q = 0; q < 3; ++q) t->gain_tw[15 + 3 * u + q][tid] = t->tw2048[(q + 1) * (tid + 128 * u)];
    }

    for (int u = 0; u < 2; ++u)   // see at3_k_gain.hpp: k_gain_analysis1
        for (int j = 0; j < 16; ++j) {
            const int k = 2 * j + u;
            for (int q = 0; q < 3; ++q) t->ga1_twb[u][q][j] = t->tw2048[16 * (q + 1) * k];
            for (int jj = 0; jj < 4; ++jj)
                for (int q = 0; q < 3; ++q) t->ga1_twb[u][3 + 3 * jj + q][j] = t->tw2048[4 * (q + 1) * (k + 32 * jj)];
        }
    for (int lane = 0; lane < 64; ++lane)
        for (int u = 0; u < 2; ++u)
            for (int tt = 0; tt < 4; ++tt) {
                const int kk = 2 * (lane & 15) + u + 32 * (4 * (lane >> 4) + tt);
                for (int q = 0; q < 3; ++q) t->ga1_twc[4 * u + tt][q][lane] = t->tw2048[(q + 1) * kk];
            }

    for (int en = 0; en < 18; ++en)   // see MdctTab (at3_k_front2.hpp); needs enc_win, mdct_sincos and tw128 from above
        for (int L = 0; L < 16; ++L) {
            const int q1 = L >> 2, q2 = L & 3, b = q1 + 4 * q2;
            float* v = t->mdct_tab[en][L];
            if (en < 4) {
                const int e = 2 * b + 32 * en;
                v[0] = t->enc_win[e]; v[1] = t->enc_win[128 + e]; v[2] = t->enc_win[127 - e]; v[3] = t->enc_win[255 - e];
            } else if (en < 8) {
                const int n = 2 * (b + 16 * (en - 4));
                v[0] = t->mdct_sincos[n]; v[1] = t->mdct_sincos[n + 1]; v[2] = t->mdct_sincos[n + 128]; v[3] = t->mdct_sincos[n + 129];
            } else if (en < 12) {
                const int n = 2 * (8 * q1 + 2 * q2 + 32 * (en - 8));
                v[0] = t->mdct_sincos[n]; v[1] = t->mdct_sincos[n + 1]; v[2] = t->mdct_sincos[n + 2]; v[3] = t->mdct_sincos[n + 3];
            } else if (en < 15) {
                const int j = en - 11, k = 2 * q2;
                v[0] = t->tw128[4 * j * k].r; v[1] = t->tw128[4 * j * k].i; v[2] = t->tw128[4 * j * (k + 1)].r; v[3] = t->tw128[4 * j * (k + 1)].i;
            } else {
                const int j = en - 14, k = 8 * q1 + 2 * q2;
                v[0] = t->tw128[j * k].r; v[1] = t->tw128[j * k].i; v[2] = t->tw128[j * (k + 1)].r; v[3] = t->tw128[j * (k + 1)].i;
            }
        }

    {   // Planck taper, epsilon 0.15, N = 512
        const float eN = 0.15f * 512.0f;
        const float fN = 512.0f;
        for (int n = 0; n < 512; ++n) {
            const float fn = (float)n;
            if (n == 0) {
                t->planck[n] = 0.0f;
            } else if (fn < eN) {
                const float z = eN * (1.0f / fn + 1.0f / (fn - eN));
                t->planck[n] = 1.0f / (1.0f + expf(z));
            } else if (fn <= fN - eN) {
                t->planck[n] = 1.0f;
            } else {
                const float m = fN - fn;
                const float z = eN * (1.0f / m + 1.0f / (m - eN));
                t->planck[n] = 1.0f / (1.0f + expf(z));
            }
        }
        for (int i = 0; i < 3; ++i) t->hpf_w[i] = 0.5f * (1.0f - cosf((float)M_PI * i / 2.0f));
    }
    for (int L = 0; L < 16; ++L)   // the window in the order k_gain_spec's lanes consume it
        for (int tt = 0; tt < 16; ++tt) {
            const int i = L + 16 * tt;
            t->spec16_win[tt][L].r = t->planck[2 * i];
            t->spec16_win[tt][L].i = t->planck[2 * i + 1];
        }

    for (size_t i = 0; i < 1024; ++i) {
        float f = (float)(i + 3) * 0.5 * 44100 / (float)1024;
        float v = log10f(f) - 3.5;
        v = -10 * v * v + 3 - f / 3000;
        v = pow(10, (0.1 * v));
        t->loud_curve[i] = v;
    }
    {
        float ath_line[1024];
        const float mf = (float)44100 / 2000.0;
        for (size_t i = 0; i < 1024; ++i) {
            const float f = (float)(i + 1) * mf / 1024;
            float trh = ath_db(1.e3 * f) - 100;
            trh -= f * f * 0.015;
            ath_line[i] = trh;
        }
        for (int b = 0; b < 32; ++b) {
            float x = 999;
            for (int line = kBfuStart[b]; line < kBfuStart[b + 1]; ++line) x = fminf(x, ath_line[line]);
            x = pow(10, 0.1f * x);
            t->ath_bfu[b] = x;
        }
    }
    // log2f of glibc 2.35 (ARM optimized-routines): 16-entry {1/c, log2 c} table + degree-4 polynomial.
    static const double tab[16][2] = {
        {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
        {0x1.49539f0f010b0p+0, -0x1.7418b0a1fb77bp-2}, {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
        {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8ea0p+0, -0x1.97c1d1b3b7af0p-3},
        {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
        {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1.0000000000000p+0, 0x0.0p+0},
        {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aa0p-1, 0x1.476a9543891bap-3},
        {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
        {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};
    static const double poly[4] = {-0x1.712b6f70a7e4dp-2, 0x1.ecabf496832e0p-2, -0x1.715479ffae3dep-1,
                                   0x1.715475f35c8b8p+0};
    memcpy(t->log2f_tab, tab, sizeof(tab));
    memcpy(t->log2f_poly, poly, sizeof(poly));
    // f64 log / exp of the same glibc (std::log, std::exp in CalcSpectralFlatnessPerBfu, atrac_psy_common.cpp:184,194)
    memcpy(t->libm.log_c, kLogData, sizeof(t->libm.log_c));
    memcpy(t->libm.log_tab, kLogData + 18, sizeof(t->libm.log_tab));
    memcpy(t->libm.exp_c, kExpData, sizeof(t->libm.exp_c));
    memcpy(t->libm.exp_tab, kExpTab, sizeof(t->libm.exp_tab));
}

}  // namespace at3

// ---- ATRAC1 -----------------------------------------------------------------------------------------------------
namespace at1 {

namespace {

const uint16_t kSpecsStartLong[kMaxBfus] = {0,   8,   16,  24,  32,  36,  40,  44,  48,  56,  64,  72,  80,  86,  92,  98,  104, 110,
                                            116, 122, 128, 134, 140, 146, 152, 159, 166, 173, 180, 189, 198, 207, 216, 226, 236, 246,
                                            256, 268, 280, 292, 304, 316, 328, 340, 352, 372, 392, 412, 432, 452, 472, 492};
const uint8_t kSpecsPerBlock[kMaxBfus] = {8,  8,  8,  8,  4,  4,  4,  4,  8,  8,  8,  8,  6,  6,  6,  6,  6,  6,
                                          6,  6,  6,  6,  6,  6,  7,  7,  7,  7,  9,  9,  9,  9,  10, 10, 10, 10,
                                          12, 12, 12, 12, 12, 12, 12, 12, 20, 20, 20, 20, 20, 20, 20, 20};

// CalcSinCos(n, scale), lib/mdct/mdct.cpp:25-36: float overloads of sqrt / cos / sin are the ones selected
AT3_RUNTIME_LIBM void mdct_sincos(float* dst, size_t n, float scale)
{
    const float alpha = 2.0 * M_PI / (8.0 * n);
    const float omiga = 2.0 * M_PI / n;
    scale = sqrtf(scale / n);
    for (size_t i = 0; i < (n >> 2); ++i) {
        dst[2 * i + 0] = scale * cosf(omiga * i + alpha);
        dst[2 * i + 1] = scale * sinf(omiga * i + alpha);
    }
}

}  // namespace

AT3_RUNTIME_LIBM void build_tables(Tables* t)
{
    memset(t, 0, sizeof(*t));
    for (int i = 0; i < 24; ++i) t->qmf_win[i] = t->qmf_win[47 - i] = at3::kTapHalf[i] * 2.0;
    for (uint32_t i = 0; i < 64; ++i) t->scale[i] = pow(2.0, (double)(i / 3.0 - 21.0));
    for (uint32_t i = 0; i < 32; ++i) t->sine[i] = sin((i + 0.5) * (M_PI / (2.0 * 32.0)));
    mdct_sincos(t->sc512, 512, 1.0f);
    mdct_sincos(t->sc256, 256, 0.5f);
    mdct_sincos(t->sc64, 64, 0.5f);
    at3::fill_twiddles(t->tw128, 128, false);
    at3::fill_twiddles(t->tw64, 64, false);
    at3::fill_twiddles(t->tw16, 16, false);
    for (size_t i = 0; i < 512; ++i) {
        float f = (float)(i + 3) * 0.5 * 44100 / (float)512;
        float v = log10f(f) - 3.5;
        v = -10 * v * v + 3 - f / 3000;
        v = pow(10, (0.1 * v));
        t->loud[i] = v;
    }
    {
        float ath_line[512];
        const float mf = (float)44100 / 2000.0;
        for (size_t i = 0; i < 512; ++i) {
            const float f = (float)(i + 1) * mf / 512;
            float trh = at3::ath_db(1.e3 * f) - 100;
            trh -= f * f * 0.015;
            ath_line[i] = trh;
        }
        for (int b = 0; b < kMaxBfus; ++b) {
            float x = 999;
            for (int line = kSpecsStartLong[b]; line < kSpecsStartLong[b] + kSpecsPerBlock[b]; ++line) x = fmin(x, ath_line[line]);
            x = pow(10, 0.1 * x);
            t->ath_bfu[b] = x;
        }
    }
    static const float fir[10] = {-8.65163e-18 * 2.0, -0.00851586 * 2.0, -6.74764e-18 * 2.0, 0.0209036 * 2.0, -3.36639e-17 * 2.0,
                                  -0.0438162 * 2.0,   -1.54175e-17 * 2.0, 0.0931738 * 2.0,   -5.52212e-17 * 2.0, -0.313819 * 2.0};
    memcpy(t->fir, fir, sizeof(fir));
    static const float fix_long[kMaxBfus] = {7, 7, 7, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 5, 5, 5, 5,
                                             5, 5, 5, 5, 5, 5, 5, 5, 5, 4, 4, 4, 3, 3, 3, 3, 3, 3, 2, 1, 1, 1, 1, 0, 0, 0};
    static const float fix_short[kMaxBfus] = {6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 5, 5,
                                              5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 4, 4, 4, 4, 4, 4, 4, 4, 0, 0, 0, 0, 0, 0, 0, 0};
    memcpy(t->fix_long, fix_long, sizeof(fix_long));
    memcpy(t->fix_short, fix_short, sizeof(fix_short));
    static const double tab[16][2] = {
        {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
        {0x1.49539f0f010b0p+0, -0x1.01eae7f513a67p-2}, {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
        {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8ea0p+0, -0x1.1aa2bc79c8100p-3},
        {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
        {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1.0000000000000p+0, 0x0.0p+0},
        {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aa0p-1, 0x1.c5e53aa362eb4p-4},
        {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d224770p-3},
        {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
    memcpy(t->logf_tab, tab, sizeof(tab));
    t->logf_ln2 = 0x1.62e42fefa39efp-1;
    t->logf_poly[0] = -0x1.00ea348b88334p-2;
    t->logf_poly[1] = 0x1.5575b0be00b6ap-2;
    t->logf_poly[2] = -0x1.ffffef20a4123p-2;
}

}  // namespace at1

// ---- ATRAC3plus front end ----------------------------------------------------------------------------------------
namespace at3p {

namespace {

const float kFir[384] = {
#include "at3p_fir.inc"
};

AT3_RUNTIME_LIBM void mdct_sincos(float* dst, size_t n, float scale)   // CalcSinCos, lib/mdct/mdct.cpp:25-36
{
    const float alpha = 2.0 * M_PI / (8.0 * n);
    const float omiga = 2.0 * M_PI / n;
    scale = sqrtf(scale / n);
    for (size_t i = 0; i < (n >> 2); ++i) {
        dst[2 * i + 0] = scale * cosf(omiga * i + alpha);
        dst[2 * i + 1] = scale * sinf(omiga * i + alpha);
    }
}

}  // namespace

AT3_RUNTIME_LIBM void build_tables(Tables* t)
{
    memset(t, 0, sizeof(*t));
    memcpy(t->fir, kFir, sizeof(kFir));
    const float dct_scale = 32.0 * (float)(128 * 512.0);   // atde_create_dct4_16(128 * 512.0) -> TMIDCT<32>(32.0 * scale)
    mdct_sincos(t->sc32, 32, dct_scale / 2);               // TMIDCT(float scale) : TMDCTBase(TN, scale / 2)
    mdct_sincos(t->sc256, 256, 1.0f);
    at3::fill_twiddles(t->tw8, 8, false);
    at3::fill_twiddles(t->tw64, 64, false);
    for (size_t i = 0; i < 128; i++) t->sine128[i] = 2.0 * sinf((i + 0.5) * (M_PI / (2.0 * 128)));
    for (size_t i = 0; i < 64; i++) t->sine64[i] = 2.0 * sinf((i + 0.5) * (M_PI / (2.0 * 64)));
}

}  // namespace at3p
