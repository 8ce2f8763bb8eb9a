/* TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's ATRAC1 encode path.
 *
 * Plain C, scalar, written from the behaviour of dcherednik/atracdenc; every function cites the reference
 * file:line it follows (paths relative to the reference's src/). Pinned against the real reference encoder
 * (oracle/_ref, at1ref_encode) by tests/test_at1_oracle.py. libm calls (log10f, sqrtf, sin, cos, pow) are the
 * container's glibc, the same the reference executes - the device code restates them where it needs them.
 *
 * Scope: TAtrac1Encoder::GetLambda (atrac1denc.cpp:180-255): 2-stage QMF with the 39-sample delay on the high band,
 * transient detection per band, windowed MDCT 512/256/64 with block switching, loudness tracking, scale factors,
 * the shift bisection of TAt1BitAlloc incl. BFU-count reduction and the bit boost, and the 212-byte sound unit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float r, i; } cpx;

/* ---- tables --------------------------------------------------------------------------------------------------- */
static const float kTapHalf[24] = { /* qmf/qmf.cpp:25-32 */
    -0.00001461907,  -0.00009205479, -0.000056157569, 0.00030117269, 0.0002422519,  -0.00085293897,
    -0.0005205574,   0.0020340169,   0.00078333891,   -0.0042153862, -0.00075614988, 0.0078402944,
    -0.000061169922, -0.01344162,    0.0024626821,    0.021736089,   -0.007801671,   -0.034090221,
    0.01880949,      0.054326009,    -0.043596379,    -0.099384367,  0.13207909,     0.46424159};
static const short kAthTab[] = { /* atrac/atrac_psy_common.cpp:43-83 */
    9669, 9669, 9626, 9512, 9353, 9113, 8882, 8676, 8469, 8243, 7997, 7748, 7492, 7239, 7000, 6762, 6529, 6302, 6084, 5900,
    5717, 5534, 5351, 5167, 5004, 4812, 4638, 4466, 4310, 4173, 4050, 3922, 3723, 3577, 3451, 3281, 3132, 3036, 2902, 2760,
    2658, 2591, 2441, 2301, 2212, 2125, 2018, 1900, 1770, 1682, 1594, 1512, 1430, 1341, 1260, 1198, 1136, 1057, 998,  943,
    887,  846,  744,  712,  693,  668,  637,  606,  580,  555,  529,  502,  475,  448,  422,  398,  375,  351,  327,  322,
    312,  301,  291,  268,  246,  215,  182,  146,  107,  61,   13,   -35,  -96,  -156, -179, -235, -295, -350, -401, -421,
    -446, -499, -532, -535, -513, -476, -431, -313, -179, 8,    203,  403,  580,  736,  881,  1022, 1154, 1251, 1348, 1421,
    1479, 1399, 1285, 1193, 1287, 1519, 1914, 2369, 3352, 4352, 5352, 6352, 7352, 8352, 9352, 9999, 9999, 9999, 9999, 9999};

/* atrac/at1/atrac1.h:83-109 */
#define AT1_MAX_BFUS 52
static const uint8_t kSpecsPerBlock[AT1_MAX_BFUS] = {
    8,  8,  8,  8,  4,  4,  4,  4,  8,  8,  8,  8,  6,  6,  6,  6,  6,  6,  6,  6,
    6,  6,  6,  6,  7,  7,  7,  7,  9,  9,  9,  9,  10, 10, 10, 10,
    12, 12, 12, 12, 12, 12, 12, 12, 20, 20, 20, 20, 20, 20, 20, 20};
static const uint8_t kBlocksPerBand[4] = {0, 20, 36, 52};
static const uint16_t kSpecsStartLong[AT1_MAX_BFUS] = {
    0,   8,   16,  24,  32,  36,  40,  44,  48,  56,  64,  72,  80,  86,  92,  98,  104, 110, 116, 122,
    128, 134, 140, 146, 152, 159, 166, 173, 180, 189, 198, 207, 216, 226, 236, 246,
    256, 268, 280, 292, 304, 316, 328, 340, 352, 372, 392, 412, 432, 452, 472, 492};
static const uint16_t kSpecsStartShort[AT1_MAX_BFUS] = {
    0,   32,  64,  96,  8,   40,  72,  104, 12,  44,  76,  108, 20,  52,  84,  116, 26,  58,  90,  122,
    128, 160, 192, 224, 134, 166, 198, 230, 141, 173, 205, 237, 150, 182, 214, 246,
    256, 288, 320, 352, 384, 416, 448, 480, 268, 300, 332, 364, 396, 428, 460, 492};
static const uint8_t kBfuAmountTab[8] = {20, 28, 32, 36, 40, 44, 48, 52};
/* atrac/at1/atrac1_bitalloc.cpp:38-75 */
static const float kFixLong[AT1_MAX_BFUS] = {7, 7, 7, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6,
                                             6, 6, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 4,
                                             4, 4, 3, 3, 3, 3, 3, 3, 2, 1, 1, 1, 1, 0, 0, 0};
static const float kFixShort[AT1_MAX_BFUS] = {6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6,
                                              6, 6, 6, 6, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5,
                                              4, 4, 4, 4, 4, 4, 4, 4, 0, 0, 0, 0, 0, 0, 0, 0};
/* BitBoostMask -> the (bits per BFU, BFU) pairs in std::multimap order (:77-92): keys ascending, insertion order kept */
static const uint8_t kBoostBits[12] = {6, 6, 6, 6, 6, 10, 10, 10, 10, 12, 12, 12};
static const uint8_t kBoostPos[12] = {18, 19, 20, 21, 22, 32, 33, 34, 35, 36, 37, 38};

static struct {
    int ready;
    float qmf_win[48];
    float scale[64];
    float sine[32];
    float sc512[256], sc256[128], sc64[32];
    cpx tw128[128], tw64[64], tw16[16];
    float loud[512];
    float ath_bfu[AT1_MAX_BFUS];
} T;

static int bfu_band(int i) { return i < 20 ? 0 : i < 36 ? 1 : 2; }

static float ath_formula_frank(float freq) /* atrac_psy_common.cpp:33-95 */
{
    if (freq < 10.) freq = 10.;
    if (freq > 29853.) freq = 29853.;
    const double freq_log = 40. * log10(0.1 * freq); /* 4 steps per third, starting at 10 Hz */
    const unsigned index = (unsigned)freq_log;
    return 0.01 * (kAthTab[index] * (1 + index - freq_log) + kAthTab[index + 1] * (freq_log - index));
}

static void calc_sincos(float* dst, size_t n, float scale) /* lib/mdct/mdct.cpp:25-36 (float overloads) */
{
    const float alpha = 2.0 * M_PI / (8.0 * n);
    const float omiga = 2.0 * M_PI / n;
    scale = sqrtf(scale / n);
    for (size_t i = 0; i < (n >> 2); ++i) {
        dst[2 * i + 0] = scale * cosf(omiga * i + alpha);
        dst[2 * i + 1] = scale * sinf(omiga * i + alpha);
    }
}

static void init_tables(void)
{
    if (T.ready) return;
    for (int i = 0; i < 24; ++i) T.qmf_win[i] = T.qmf_win[47 - i] = kTapHalf[i] * 2.0;                 /* qmf.cpp:41-44 */
    for (uint32_t i = 0; i < 64; ++i) T.scale[i] = pow(2.0, (double)(i / 3.0 - 21.0));               /* atrac1.h:124-128 */
    for (uint32_t i = 0; i < 32; ++i) T.sine[i] = sin((i + 0.5) * (M_PI / (2.0 * 32.0)));           /* atrac1.h:129-133 */
    calc_sincos(T.sc512, 512, 1.0f);                                                                  /* atrac1denc.h:48-50 */
    calc_sincos(T.sc256, 256, 0.5f);
    calc_sincos(T.sc64, 64, 0.5f);
    const double pi = 3.141592653589793238462643383279502884197169399375105820974944;                /* kiss_fft.c:357-363 */
    for (int i = 0; i < 128; ++i) { const double ph = -2 * pi * i / 128; T.tw128[i].r = (float)cos(ph); T.tw128[i].i = (float)sin(ph); }
    for (int i = 0; i < 64; ++i) { const double ph = -2 * pi * i / 64; T.tw64[i].r = (float)cos(ph); T.tw64[i].i = (float)sin(ph); }
    for (int i = 0; i < 16; ++i) { const double ph = -2 * pi * i / 16; T.tw16[i].r = (float)cos(ph); T.tw16[i].i = (float)sin(ph); }
    for (size_t i = 0; i < 512; i++) {                                                                /* atrac_psy_common.cpp:142-156 */
        float f = (float)(i + 3) * 0.5 * 44100 / (float)512;
        float t = log10f(f) - 3.5;
        t = -10 * t * t + 3 - f / 3000;
        t = pow(10, (0.1 * t));
        T.loud[i] = t;
    }
    {   /* CalcATH(512, 44100) (:126-140) then CalcAt1ATH (atrac1_bitalloc.cpp:130-149) */
        float spec[512];
        const float mf = (float)44100 / 2000.0;
        for (size_t i = 0; i < 512; i++) {
            const float f = (float)(i + 1) * mf / 512;
            float trh = ath_formula_frank(1.e3 * f) - 100;
            trh -= f * f * 0.015;
            spec[i] = trh;
        }
        for (int b = 0; b < AT1_MAX_BFUS; ++b) {
            float x = 999;
            for (int line = kSpecsStartLong[b]; line < kSpecsStartLong[b] + kSpecsPerBlock[b]; line++) x = fmin(x, spec[line]);
            x = pow(10, 0.1 * x);
            T.ath_bfu[b] = x;
        }
    }
    T.ready = 1;
}

/* ---- kissfft-order FFT (kiss_fft.c:21-90, 238-302), sizes 4^k ------------------------------------------------- */
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
        F[m + k].r = F[k].r - t.r; F[m + k].i = F[k].i - t.i;
        F[k].r += t.r; F[k].i += t.i;
    }
}
static void fft_combine4(cpx* F, int m, int fstride, const cpx* tw)
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
        F[m + k].r = s5.r + s4.i; F[m + k].i = s5.i - s4.r;
        F[3 * m + k].r = s5.r - s4.i; F[3 * m + k].i = s5.i + s4.r;
    }
}
static void fft_rec(cpx* out, const cpx* in, int n, int fstride, const cpx* tw)
{
    const int p = (n % 4 == 0) ? 4 : 2;
    const int m = n / p;
    if (m == 1) {
        for (int q = 0; q < p; ++q) out[q] = in[q * fstride];
    } else {
        for (int q = 0; q < p; ++q) fft_rec(out + q * m, in + q * fstride, m, fstride * p, tw);
    }
    if (p == 4) fft_combine4(out, m, fstride, tw);
    else fft_combine2(out, m, fstride, tw);
}

/* TMDCT<N>::operator() (lib/mdct/mdct.h:51-104): N in -> N/2 out */
static void mdct_n(const float* in, float* out, int N, const float* cs, const cpx* tw)
{
    const int n2 = N >> 1, n4 = N >> 2, n34 = 3 * n4, n54 = 5 * n4;
    cpx fin[128], fout[128];
    int n;
    for (n = 0; n < n4; n += 2) {
        const float r0 = in[n34 - 1 - n] + in[n34 + n];
        const float i0 = in[n4 + n] - in[n4 - 1 - n];
        const float c = cs[n], s = cs[n + 1];
        fin[n / 2].r = r0 * c + i0 * s;
        fin[n / 2].i = i0 * c - r0 * s;
    }
    for (; n < n2; n += 2) {
        const float r0 = in[n34 - 1 - n] - in[n - n4];
        const float i0 = in[n4 + n] + in[n54 - 1 - n];
        const float c = cs[n], s = cs[n + 1];
        fin[n / 2].r = r0 * c + i0 * s;
        fin[n / 2].i = i0 * c - r0 * s;
    }
    fft_rec(fout, fin, n4, 1, tw);
    for (n = 0; n < n2; n += 2) {
        const float r0 = fout[n / 2].r, i0 = fout[n / 2].i;
        const float c = cs[n], s = cs[n + 1];
        out[n] = -r0 * c - i0 * s;
        out[n2 - 1 - n] = -r0 * s + i0 * c;
    }
}

/* ---- QMF (qmf/qmf.h:47-64) and the ATRAC1 filter bank (atrac/at1/atrac1_qmf.h:25-45) -------------------------- */
typedef struct { float hist[46]; } qmf_state;
static void qmf_analysis(qmf_state* st, const float* in, int n_in, float* lower, float* upper)
{
    float buf[512 + 46];
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
typedef struct { qmf_state q1, q2; float delay[39 + 512]; } at1_bank;
static void bank_analysis(at1_bank* b, const float* pcm, float* low, float* mid, float* hi)
{
    float midlow[512];
    memcpy(&b->delay[0], &b->delay[256], sizeof(float) * 39);
    qmf_analysis(&b->q1, pcm, 512, midlow, &b->delay[39]);
    qmf_analysis(&b->q2, midlow, 256, low, mid);
    memcpy(hi, &b->delay[0], sizeof(float) * 256);
}

/* ---- transient detector (transient_detector.cpp:48-95; the float overloads of sqrt / log10 are selected) ------ */
typedef struct { int short_sz, block_sz; float hpf[256 + 21]; float last_energy; } tdet;
static void hp_filter(tdet* d, const float* in, float* out)
{
    static const float fircoef[] = {-8.65163e-18 * 2.0, -0.00851586 * 2.0, -6.74764e-18 * 2.0, 0.0209036 * 2.0, -3.36639e-17 * 2.0,
                                    -0.0438162 * 2.0,   -1.54175e-17 * 2.0, 0.0931738 * 2.0,   -5.52212e-17 * 2.0, -0.313819 * 2.0};
    memcpy(d->hpf + 20, in, d->block_sz * sizeof(float));
    const float* b = d->hpf;
    for (int i = 0; i < d->block_sz; ++i) {
        float s = b[i + 10];
        float s2 = 0;
        for (int j = 0; j < ((21 - 1) / 2) - 1; j += 2) {
            s += fircoef[j] * (b[i + j] + b[i + 21 - j]);
            s2 += fircoef[j + 1] * (b[i + j + 1] + b[i + 21 - j - 1]);
        }
        out[i] = (s + s2) / 2;
    }
    memcpy(d->hpf, in + (d->block_sz - 20), 20 * sizeof(float));
}
static float calc_rms(const float* in, uint32_t n)
{
    float s = 0;
    for (uint32_t i = 0; i < n; i++) s += (in[i] * in[i]);
    s /= n;
    return sqrtf(s);
}
static int detect(tdet* d, const float* buf)
{
    const int nshort = d->block_sz / d->short_sz;
    float rms[17], filtered[256];
    hp_filter(d, buf, filtered);
    int trans = 0;
    rms[0] = d->last_energy;
    for (int i = 1; i < nshort + 1; ++i) {
        rms[i] = 19.0 * log10f(calc_rms(&filtered[(size_t)(i - 1) * d->short_sz], d->short_sz));
        if (rms[i] - rms[i - 1] > 16) trans = 1;
        if (rms[i - 1] - rms[i] > 20) trans = 1;
    }
    d->last_energy = rms[nshort];
    return trans;
}

/* ---- windowed MDCT with block switching (atrac1denc.cpp:70-102) ------------------------------------------------ */
static void swap_array(float* p, size_t len)
{
    for (size_t i = 0, j = len - 1; i < len / 2; ++i, --j) { const float t = p[i]; p[i] = p[j]; p[j] = t; }
}
static void at1_mdct(float specs[512], float* low, float* mid, float* hi, const int log_count[3])
{
    uint32_t pos = 0;
    for (uint32_t band = 0; band < 3; band++) {
        const uint32_t num = 1u << log_count[band];
        float* src = (band == 0) ? low : (band == 1) ? mid : hi;
        const uint32_t buf_sz = (band == 2) ? 256 : 128;
        const uint32_t block_sz = (num == 1) ? buf_sz : 32;
        const uint32_t win_start = (num == 1) ? ((band == 2) ? 112 : 48) : 0;
        const float multiple = (num != 1 && band == 2) ? 2.0 : 1.0;
        float tmp[512];
        memset(tmp, 0, sizeof(tmp));
        uint32_t block_pos = 0;
        for (uint32_t k = 0; k < num; ++k) {
            memcpy(&tmp[win_start], &src[buf_sz], 32 * sizeof(float));
            for (size_t i = 0; i < 32; i++) {
                src[buf_sz + i] = T.sine[i] * src[block_pos + block_sz - 32 + i];
                src[block_pos + block_sz - 32 + i] = T.sine[31 - i] * src[block_pos + block_sz - 32 + i];
            }
            memcpy(&tmp[win_start + 32], &src[block_pos], block_sz * sizeof(float));
            float sp[256];
            int n_sp;
            if (num == 1) {
                if (band == 2) { mdct_n(tmp, sp, 512, T.sc512, T.tw128); n_sp = 256; }
                else { mdct_n(tmp, sp, 256, T.sc256, T.tw64); n_sp = 128; }
            } else {
                mdct_n(tmp, sp, 64, T.sc64, T.tw16);
                n_sp = 32;
            }
            for (int i = 0; i < n_sp; i++) specs[block_pos + pos + i] = sp[i] * multiple;
            if (band) swap_array(&specs[block_pos + pos], n_sp);
            block_pos += 32;
        }
        pos += buf_sz;
    }
}

/* ---- scale factors (atrac/atrac_scale.cpp:141-188) -------------------------------------------------------------- */
typedef struct { int sfi; float energy; float values[20]; } sblock;
static void scale_block(const float* in, int len, sblock* out)
{
    float max_abs = 0;
    for (int i = 0; i < len; ++i) {
        const float a = fabsf(in[i]);
        if (a > max_abs) max_abs = a;
    }
    if (max_abs > 1.0f) max_abs = 1.0f;
    int sfi = 0;
    while (sfi < 63 && T.scale[sfi] < max_abs) ++sfi; /* std::map::lower_bound on the increasing table */
    const float sf = T.scale[sfi];
    out->sfi = sfi;
    out->energy = 0.0;
    for (int i = 0; i < len; ++i) {
        float v = in[i] / sf;
        const float e = in[i] * in[i];
        out->energy += e;
        if (fabsf(v) >= 1.0) v = (v > 0) ? 0.99999 : -0.99999;
        out->values[i] = v;
    }
}
static void scale_frame(const float* specs, const int log_count[3], sblock* blocks)
{
    for (int band = 0; band < 3; ++band) {
        const int short_win = log_count[band] != 0;
        for (int b = kBlocksPerBand[band]; b < kBlocksPerBand[band + 1]; ++b)
            scale_block(&specs[short_win ? kSpecsStartShort[b] : kSpecsStartLong[b]], kSpecsPerBlock[b], &blocks[b]);
    }
}

/* ---- bit writer (lib/bitstream/bitstream.cpp:40-63): MSB first ------------------------------------------------- */
typedef struct { uint8_t buf[512]; int bits; } bitw;
static void bw_write(bitw* w, uint32_t val, int n)
{
    for (int i = n - 1; i >= 0; --i) {
        if (w->bits < (int)sizeof(w->buf) * 8 && ((val >> i) & 1u)) w->buf[w->bits >> 3] |= (uint8_t)(0x80u >> (w->bits & 7));
        w->bits++;
    }
}

/* ---- bit allocation (atrac/at1/atrac1_bitalloc.cpp) -------------------------------------------------------------- */
static float low_to_mid_tilt(const sblock* b, uint32_t n) /* :158-174 */
{
    float sum_low = 0.0f, sum_mid = 0.0f;
    uint32_t n_low = 0, n_mid = 0;
    for (size_t i = 0; i < n; ++i) {
        switch (bfu_band((int)i)) {
            case 0: sum_low += b[i].sfi; n_low++; break;
            case 1: sum_mid += b[i].sfi; n_mid++; break;
            default: break;
        }
    }
    if (!n_low || !n_mid) return 0.0f;
    return sum_low / n_low - sum_mid / n_mid;
}
static void calc_bits_allocation(const sblock* b, uint32_t n, float spread, float shift, const int log_count[3], float loudness,
                                 uint32_t* bits) /* :176-226 */
{
    const float tilt = low_to_mid_tilt(b, n);
    const float mid_bias = fminf(1.5f, 0.3f * fmaxf(0.0f, tilt - 7.0f));
    const float band_bias[3] = {0.0f, mid_bias, mid_bias * 0.5f};
    for (size_t i = 0; i < n; ++i) {
        const int short_block = log_count[bfu_band((int)i)] != 0;
        const float fix = short_block ? kFixShort[i] : kFixLong[i];
        const float ath = T.ath_bfu[i] * loudness;
        if (!short_block && b[i].energy < ath) {
            bits[i] = 0;
        } else {
            const int tmp = spread * ((float)b[i].sfi / 3.2f) + (1.0f - spread) * fix - shift + band_bias[bfu_band((int)i)];
            if (tmp > 16) bits[i] = 16;
            else if (tmp < 2) bits[i] = 0;
            else bits[i] = tmp;
        }
    }
}
static uint32_t max_used_bfu_id(const uint32_t* bits, uint32_t size) /* :228-252 */
{
    uint32_t idx = 7;
    for (;;) {
        uint32_t n = kBfuAmountTab[idx];
        if (n > size) {
            idx--;
        } else if (idx != 0) {
            uint32_t i = 0;
            while (idx && bits[n - 1 - i] == 0) {
                if (++i >= (uint32_t)(kBfuAmountTab[idx] - kBfuAmountTab[idx - 1])) {
                    idx--;
                    n -= i;
                    i = 0;
                }
            }
            break;
        } else {
            break;
        }
    }
    return idx;
}
static uint32_t avail_bits(size_t n) { return 212 * 8 - 3 - 32 - 2 - 3 - (uint32_t)n * (4 + 6); } /* :272-276 */
static uint32_t apply_boost(uint32_t* bits, uint32_t size, uint32_t cur, uint32_t target) /* :94-128 */
{
    uint32_t surplus = target - cur;
    const uint32_t key = (surplus > 12) ? 12 : surplus;
    int max_it = 0;
    while (max_it < 12 && kBoostBits[max_it] <= key) ++max_it; /* multimap::upper_bound(key) */
    if (max_it == 0) return surplus;
    while (surplus >= 6) {
        int done = 1;
        for (int it = 0; it < max_it; ++it) {
            const uint32_t cur_bits = kBoostBits[it], cur_pos = kBoostPos[it];
            if (cur_pos >= size) break;
            if (bits[cur_pos] == 16u) continue;
            const uint32_t per_spec = bits[cur_pos] ? 1 : 2;
            if (bits[cur_pos] == 0u && cur_bits * 2 > surplus) continue;
            if (cur_bits * per_spec > surplus) continue;
            bits[cur_pos] += per_spec;
            surplus -= cur_bits * per_spec;
            done = 0;
        }
        if (done) break;
    }
    return surplus;
}
static int make_sign(int val, unsigned bits) /* lib/bitstream/bitstream.h:27-31 */
{
    const unsigned shift = 8 * sizeof(int) - bits;
    union { unsigned u; int s; } v = {(unsigned)val << shift};
    return v.s >> shift;
}
/* TAt1BitAlloc::Write (:385-405) with the driver of lib/bs_encode/encode.cpp:57-129 unrolled */
static void write_frame(const sblock* blocks, const int log_count[3], float loudness, int bfu_idx_const, uint8_t out[212])
{
    uint32_t bfu_idx = bfu_idx_const ? (uint32_t)bfu_idx_const - 1 : 7;
    const int auto_bfu = !bfu_idx_const;
    const float spread = 0.4f;
    uint32_t bits[AT1_MAX_BFUS];
    uint32_t n = 0;
    for (;;) { /* TConfigure::Encode */
        n = kBfuAmountTab[bfu_idx];
        const uint32_t target = avail_bits(n);
        float min_l = -3, max_l = 15, cur_l = 0, last_l = 15;
        int repeat = 0;
        for (;;) { /* TBfuAlloc::Encode */
            const int exhausted = (max_l <= min_l);
            float shift;
            if (exhausted) {
                shift = last_l;
            } else {
                cur_l = (max_l + min_l) / 2.0;
                shift = cur_l;
            }
            calc_bits_allocation(blocks, n, spread, shift, log_count, loudness, bits);
            uint32_t used = 0;
            for (uint32_t i = 0; i < n; i++) used += kSpecsPerBlock[i] * bits[i];
            int done;
            if (exhausted) done = 1;
            else if (used < target) { last_l = cur_l; max_l = cur_l - 0.01f; done = 0; }
            else if (used > target) { min_l = cur_l + 0.01f; done = 0; }
            else done = 1;
            if (!done) continue;
            if (auto_bfu && max_used_bfu_id(bits, n) < bfu_idx) {
                bfu_idx--;
                repeat = 1;
            } else {
                apply_boost(bits, n, used, avail_bits(n));
            }
            break;
        }
        if (!repeat) break;
    }
    /* TBfuAlloc::Dump (:297-338) */
    bitw w;
    memset(&w, 0, sizeof(w));
    bw_write(&w, 0x2 - log_count[0], 2);
    bw_write(&w, 0x2 - log_count[1], 2);
    bw_write(&w, 0x3 - log_count[2], 2);
    bw_write(&w, 0, 2);
    bw_write(&w, bfu_idx, 3);
    bw_write(&w, 0, 2);
    bw_write(&w, 0, 3);
    for (uint32_t i = 0; i < n; ++i) bw_write(&w, bits[i] ? (bits[i] - 1) : 0, 4);
    for (uint32_t i = 0; i < n; ++i) bw_write(&w, (uint32_t)blocks[i].sfi, 6);
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t wl = bits[i];
        if (wl == 0 || wl == 1) continue;
        const float multiple = ((1 << (wl - 1)) - 1);
        for (int k = 0; k < kSpecsPerBlock[i]; ++k) {
            const int tmp = (int)lrint(blocks[i].values[k] * multiple);
            bw_write(&w, (uint32_t)make_sign(tmp, wl), (int)wl);
        }
    }
    bw_write(&w, 0, 8);
    bw_write(&w, 0, 8);
    bw_write(&w, 0, 8);
    memcpy(out, w.buf, 212);
}

/* Table dump for the tests: the layout of atracdenc_amd/csrc/at1_tables.hpp up to ath_bfu (all float32). */
int at1o_tables(float* dst, int n_floats)
{
    init_tables();
    float* p = dst;
    const int need = 48 + 64 + 32 + 256 + 128 + 32 + 2 * (128 + 64 + 16) + 512 + 52;
    if (n_floats != need) return -1;
    memcpy(p, T.qmf_win, sizeof(T.qmf_win)); p += 48;
    memcpy(p, T.scale, sizeof(T.scale)); p += 64;
    memcpy(p, T.sine, sizeof(T.sine)); p += 32;
    memcpy(p, T.sc512, sizeof(T.sc512)); p += 256;
    memcpy(p, T.sc256, sizeof(T.sc256)); p += 128;
    memcpy(p, T.sc64, sizeof(T.sc64)); p += 32;
    memcpy(p, T.tw128, sizeof(T.tw128)); p += 256;
    memcpy(p, T.tw64, sizeof(T.tw64)); p += 128;
    memcpy(p, T.tw16, sizeof(T.tw16)); p += 32;
    memcpy(p, T.loud, sizeof(T.loud)); p += 512;
    memcpy(p, T.ath_bfu, sizeof(T.ath_bfu)); p += 52;
    return need;
}

/* ---- encoder object (atrac1denc.cpp:180-255) ------------------------------------------------------------------- */
typedef struct {
    int nch, window_auto, window_mask, bfu_idx_const;
    at1_bank bank[2];
    tdet det[2][3];
    float low[2][256 + 16], mid[2][256 + 16], hi[2][512 + 16];
    float loudness;
} at1o_encoder;

void* at1o_create(int nch, int window_auto, int window_mask, int bfu_idx_const)
{
    init_tables();
    if (nch != 1 && nch != 2) return NULL;
    at1o_encoder* e = (at1o_encoder*)calloc(1, sizeof(*e));
    if (!e) return NULL;
    e->nch = nch;
    e->window_auto = window_auto;
    e->window_mask = window_mask;
    e->bfu_idx_const = bfu_idx_const;
    e->loudness = 0.006f;
    for (int c = 0; c < 2; ++c) {
        e->det[c][0].short_sz = 16; e->det[c][0].block_sz = 128;   /* atrac1denc.h:69-72 */
        e->det[c][1].short_sz = 16; e->det[c][1].block_sz = 128;
        e->det[c][2].short_sz = 16; e->det[c][2].block_sz = 256;
    }
    return e;
}
void at1o_destroy(void* e) { free(e); }

/* One 512-sample block: pcm [512][nch] -> out [nch][212]. taps (optional): specs [nch][512], window masks [nch],
 * loudness after tracking [1]. */
void at1o_process(void* ep, const float* pcm, uint8_t* out, float* tap_specs, int32_t* tap_masks, float* tap_loud)
{
    at1o_encoder* e = (at1o_encoder*)ep;
    float specs[2][512], l[2] = {0, 0};
    int log_count[2][3];
    uint32_t masks[2] = {0, 0};
    sblock blocks[2][AT1_MAX_BFUS];
    for (int ch = 0; ch < e->nch; ++ch) {
        float src[512];
        for (int i = 0; i < 512; ++i) src[i] = pcm[i * e->nch + ch];
        bank_analysis(&e->bank[ch], src, e->low[ch], e->mid[ch], e->hi[ch]);
        uint32_t mask = 0;
        if (e->window_auto) {
            float inv[256];
            mask |= (uint32_t)detect(&e->det[ch][0], e->low[ch]);
            memcpy(inv, e->mid[ch], 128 * sizeof(float));
            for (int i = 0; i < 128; i += 2) inv[i] *= -1;                        /* InvertSpectr<128>, util.h:51-63 */
            mask |= (uint32_t)detect(&e->det[ch][1], inv) << 1;
            memcpy(inv, e->hi[ch], 256 * sizeof(float));
            for (int i = 0; i < 256; i += 2) inv[i] *= -1;
            mask |= (uint32_t)detect(&e->det[ch][2], inv) << 2;
        } else {
            mask = (uint32_t)e->window_mask;
        }
        masks[ch] = mask;
        log_count[ch][0] = (mask & 1) ? 2 : 0;   /* TBlockSizeMod::Create, atrac1.h:62-68 */
        log_count[ch][1] = (mask & 2) ? 2 : 0;
        log_count[ch][2] = (mask & 4) ? 3 : 0;
        at1_mdct(specs[ch], e->low[ch], e->mid[ch], e->hi[ch], log_count[ch]);
        float acc = 0.0;
        for (int i = 0; i < 512; i++) {
            const float en = specs[ch][i] * specs[ch][i];
            acc += en * T.loud[i];
        }
        l[ch] = acc;
    }
    if (e->nch == 2 && masks[0] == 0 && masks[1] == 0) e->loudness = 0.98 * e->loudness + 0.01 * (l[0] + l[1]);   /* atrac_psy_common.h:46-54 */
    else if (masks[0] == 0) e->loudness = 0.98 * e->loudness + 0.02 * l[0];
    for (int ch = 0; ch < e->nch; ++ch) {
        scale_frame(specs[ch], log_count[ch], blocks[ch]);
        write_frame(blocks[ch], log_count[ch], e->loudness / 0.006f, e->bfu_idx_const, out + 212 * ch);
    }
    if (tap_specs) memcpy(tap_specs, specs, sizeof(float) * 512 * e->nch);
    if (tap_masks) for (int ch = 0; ch < e->nch; ++ch) tap_masks[ch] = (int32_t)masks[ch];
    if (tap_loud) *tap_loud = e->loudness;
}

/* pcm [n_blocks][512][nch] -> out [n_blocks][nch][212]; taps per block as in at1o_process (may be NULL) */
int at1o_encode(const float* pcm, int nch, int n_blocks, int window_auto, int window_mask, int bfu_idx_const, uint8_t* out,
                float* tap_specs, int32_t* tap_masks, float* tap_loud)
{
    void* e = at1o_create(nch, window_auto, window_mask, bfu_idx_const);
    if (!e) return -1;
    for (int b = 0; b < n_blocks; ++b)
        at1o_process(e, pcm + (size_t)b * 512 * nch, out + (size_t)b * 212 * nch, tap_specs ? tap_specs + (size_t)b * 512 * nch : NULL,
                     tap_masks ? tap_masks + (size_t)b * nch : NULL, tap_loud ? tap_loud + b : NULL);
    at1o_destroy(e);
    return n_blocks * nch * 212;
}
