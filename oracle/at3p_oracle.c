/* TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's ATRAC3plus front end:
 * at3plus_pqf_do_analyse (atrac/atrac3plus_pqf/atrac3plus_pqf.c:81-147, with the 16-point DCT-IV of lib/mdct/mdct.cpp:55-80
 * built on TMIDCT<32>, lib/mdct/mdct.h:107-180) and TAt3pMDCT::Do (atrac/at3p/at3p_mdct.cpp:33-96, TMDCT<256>).
 * Plain C, scalar; pinned against the reference functions compiled into oracle/_ref by tests/test_at3p_oracle.py. */
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct { float r, i; } cpx;

static const float kFir[384] = {
#include "at3p_fir.inc"
};

static struct {
    int ready;
    float sc32[16];    /* TMIDCT<32>(32 * 128 * 512): CalcSinCos(32, scale / 2) */
    float sc256[128];  /* TMDCT<256>(1) */
    cpx tw8[8], tw64[64];
    float sine128[128], sine64[64];
} P;

static void calc_sincos(float* dst, size_t n, float scale) /* lib/mdct/mdct.cpp:25-36 */
{
    const float alpha = 2.0 * M_PI / (8.0 * n);
    const float omiga = 2.0 * M_PI / n;
    scale = sqrtf(scale / n);
    for (size_t i = 0; i < (n >> 2); ++i) {
        dst[2 * i + 0] = scale * cosf(omiga * i + alpha);
        dst[2 * i + 1] = scale * sinf(omiga * i + alpha);
    }
}

static void init(void)
{
    if (P.ready) return;
    const float dct_scale = 32.0 * (float)(128 * 512.0);   /* atde_create_dct4_16(128 * 512.0) -> 32.0 * scale (mdct.cpp:63-66) */
    calc_sincos(P.sc32, 32, dct_scale / 2);                /* TMIDCT(float scale): TMDCTBase(TN, scale / 2) */
    calc_sincos(P.sc256, 256, 1.0f);
    const double pi = 3.141592653589793238462643383279502884197169399375105820974944; /* kiss_fft.c:357-363 */
    for (int i = 0; i < 8; ++i) { const double ph = -2 * pi * i / 8; P.tw8[i].r = (float)cos(ph); P.tw8[i].i = (float)sin(ph); }
    for (int i = 0; i < 64; ++i) { const double ph = -2 * pi * i / 64; P.tw64[i].r = (float)cos(ph); P.tw64[i].i = (float)sin(ph); }
    for (size_t i = 0; i < 128; i++) P.sine128[i] = 2.0 * sinf((i + 0.5) * (M_PI / (2.0 * 128)));   /* at3p_mdct.cpp:33-46 */
    for (size_t i = 0; i < 64; i++) P.sine64[i] = 2.0 * sinf((i + 0.5) * (M_PI / (2.0 * 64)));
    P.ready = 1;
}

/* kissfft-order FFT (kiss_fft.c:21-90, 238-302) */
static inline cpx cmul(cpx a, cpx b)
{
    cpx m;
    m.r = a.r * b.r - a.i * b.i;
    m.i = a.r * b.i + a.i * b.r;
    return m;
}
static void combine2(cpx* F, int m, int fstride, const cpx* tw)
{
    for (int k = 0; k < m; ++k) {
        const cpx t = cmul(F[m + k], tw[k * fstride]);
        F[m + k].r = F[k].r - t.r; F[m + k].i = F[k].i - t.i;
        F[k].r += t.r; F[k].i += t.i;
    }
}
static void combine4(cpx* F, int m, int fstride, const cpx* tw)
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
    if (p == 4) combine4(out, m, fstride, tw);
    else combine2(out, m, fstride, tw);
}

/* atde_do_dct4_16 (mdct.cpp:73-80): TMIDCT<32> of the 16 inputs, outputs x[8..23] negated */
static void dct4_16(const float* in, float* out)
{
    cpx fin[8], fout[8];
    float buf[32];
    const float* cs = P.sc32;
    for (int n = 0; n < 16; n += 2) {
        const float r0 = in[n], i0 = in[15 - n];
        const float c = cs[n], s = cs[n + 1];
        fin[n / 2].r = -2.0 * (i0 * s + r0 * c);
        fin[n / 2].i = -2.0 * (i0 * c - r0 * s);
    }
    fft_rec(fout, fin, 8, 1, P.tw8);
    int n;
    for (n = 0; n < 8; n += 2) {
        const float r0 = fout[n / 2].r, i0 = fout[n / 2].i;
        const float c = cs[n], s = cs[n + 1];
        const float r1 = r0 * c + i0 * s, i1 = r0 * s - i0 * c;
        buf[23 - n] = r1; buf[24 + n] = r1; buf[8 + n] = i1; buf[7 - n] = -i1;
    }
    for (; n < 16; n += 2) {
        const float r0 = fout[n / 2].r, i0 = fout[n / 2].i;
        const float c = cs[n], s = cs[n + 1];
        const float r1 = r0 * c + i0 * s, i1 = r0 * s - i0 * c;
        buf[23 - n] = r1; buf[n - 8] = -r1; buf[8 + n] = i1; buf[39 - n] = i1;
    }
    for (int i = 0; i < 16; i++) out[i] = buf[i + 8] * -1.0;
}

/* in [n_frames][2048] -> out [n_frames][16][128]; start-of-stream state first (atrac3plus_pqf.c:107-147) */
void at3po_pqf_analyse(const float* in, int n_frames, float* out)
{
    init();
    float buf[2048 + 368];
    memset(buf, 0, sizeof(buf));
    for (int f = 0; f < n_frames; ++f) {
        memcpy(buf + 368, in + (size_t)f * 2048, sizeof(float) * 2048);
        const float* x = buf;
        float* o = out + (size_t)f * 2048;
        for (int i = 0; i < 128; i++) {
            double y[32];
            for (int r = 0; r < 32; r++) {          /* vectoring (:62-70): float products, double running sums */
                y[r] = 0;
                for (int j = 0; j < 12; j++) y[r] += kFir[r * 12 + j] * x[j * 32 + r];
            }
            float yy[16], res[16];
            for (int k = 0; k < 8; k++) {           /* matrixing (:72-89) */
                yy[k] = y[k + 8] + y[7 - k];
                yy[k + 8] = y[k + 16] + y[31 - k];
            }
            dct4_16(yy, res);
            for (int k = 0; k < 16; k++) o[k * 128 + i] = res[15 - k];
            x += 16;
        }
        memmove(buf, buf + 2048, sizeof(float) * 368);
    }
}

/* TMDCT<256>::operator() (lib/mdct/mdct.h:51-104) */
static void mdct256(const float* in, float* out)
{
    cpx fin[64], fout[64];
    const float* cs = P.sc256;
    int n;
    for (n = 0; n < 64; n += 2) {
        const float r0 = in[191 - n] + in[192 + n];
        const float i0 = in[64 + n] - in[63 - n];
        fin[n / 2].r = r0 * cs[n] + i0 * cs[n + 1];
        fin[n / 2].i = i0 * cs[n] - r0 * cs[n + 1];
    }
    for (; n < 128; n += 2) {
        const float r0 = in[191 - n] - in[n - 64];
        const float i0 = in[64 + n] + in[319 - n];
        fin[n / 2].r = r0 * cs[n] + i0 * cs[n + 1];
        fin[n / 2].i = i0 * cs[n] - r0 * cs[n + 1];
    }
    fft_rec(fout, fin, 64, 1, P.tw64);
    for (n = 0; n < 128; n += 2) {
        const float r0 = fout[n / 2].r, i0 = fout[n / 2].i;
        out[n] = -r0 * cs[n] - i0 * cs[n + 1];
        out[127 - n] = -r0 * cs[n + 1] + i0 * cs[n];
    }
}

/* bands [n_frames][16][128], steep-window flags per frame (or NULL) -> specs [n_frames][2048]; zeroed history first */
void at3po_mdct(const float* bands, const uint16_t* win_flags, int n_frames, float* specs)
{
    init();
    float work[16][256];
    memset(work, 0, sizeof(work));
    for (int f = 0; f < n_frames; ++f) {
        const unsigned flags = win_flags ? win_flags[f] : 0;
        for (int b = 0; b < 16; b++) {
            const float* src = bands + (size_t)f * 2048 + b * 128;
            float* cur = specs + (size_t)f * 2048 + b * 128;
            float* tmp = work[b];
            const int steep = (flags >> b) & 1;
            if (steep) {
                for (int i = 0; i < 64; i++) tmp[128 + i] = src[i] * 2.0;
                for (int i = 0; i < 64; i++) tmp[160 + i] = P.sine64[63 - i] * src[32 + i];
                memset(&tmp[224], 0, sizeof(float) * 32);
            } else {
                for (int i = 0; i < 128; i++) tmp[128 + i] = P.sine128[127 - i] * src[i];
            }
            mdct256(tmp, cur);
            if (b & 1)
                for (int i = 0, j = 127; i < 64; ++i, --j) { const float t = cur[i]; cur[i] = cur[j]; cur[j] = t; }
            if (steep) {
                memset(&tmp[0], 0, sizeof(float) * 32);
                for (int i = 0; i < 64; i++) tmp[i + 32] = P.sine64[i] * src[i + 32];
                for (int i = 0; i < 32; i++) tmp[i + 96] = src[i + 96] * 2.0;
            } else {
                for (int i = 0; i < 128; i++) tmp[i] = P.sine128[i] * src[i];
            }
        }
    }
}

/* tables for the product-vs-oracle table test: sc32[16] sc256[128] tw8[16] tw64[128] sine128[128] sine64[64] fir[384] */
int at3po_tables(float* dst, int n_floats)
{
    init();
    if (n_floats != 16 + 128 + 16 + 128 + 128 + 64 + 384) return -1;
    float* p = dst;
    memcpy(p, P.sc32, sizeof(P.sc32)); p += 16;
    memcpy(p, P.sc256, sizeof(P.sc256)); p += 128;
    memcpy(p, P.tw8, sizeof(P.tw8)); p += 16;
    memcpy(p, P.tw64, sizeof(P.tw64)); p += 128;
    memcpy(p, P.sine128, sizeof(P.sine128)); p += 128;
    memcpy(p, P.sine64, sizeof(P.sine64)); p += 64;
    memcpy(p, kFir, sizeof(kFir));
    return n_floats;
}
