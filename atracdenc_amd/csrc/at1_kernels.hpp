// ATRAC1 encode kernels (gfx950), SURVEY.md 8(f) row f3: the body of TAtrac1Encoder::GetLambda (atrac1denc.cpp:180-255)
// for a batch of streams. Nothing in the path carries a recursion across sound units except the loudness tracker, so
// the work splits into
//   k_at1_front      one workgroup per (stream, sound unit, channel): QMF tree, transient detection, block-switched
//                    MDCT, per-channel loudness, scale factors
//   k_at1_loud_scan  one lane per stream: TrackLoudness over the call's sound units
//   k_at1_alloc_pack one wave per (stream, sound unit, channel): shift bisection, BFU-count reduction, bit boost, packing
// Every float operation is the reference's, in its order, without contraction; integer work is free to reassociate.
#pragma once
#include "at1_tables.hpp"
#include "at3_common.hpp"

namespace at1 {

using at3::f2;
using at3::fft_leaf_pos;
using at3::fft_lds;
using at3::wave_inclusive_scan;
using at3::wave_sync;

// atrac/at1/atrac1.h:83-109
__device__ static const uint8_t c_spb[kMaxBfus] = {8,  8,  8,  8,  4,  4,  4,  4,  8,  8,  8,  8,  6,  6,  6,  6,  6,  6,
                                                   6,  6,  6,  6,  6,  6,  7,  7,  7,  7,  9,  9,  9,  9,  10, 10, 10, 10,
                                                   12, 12, 12, 12, 12, 12, 12, 12, 20, 20, 20, 20, 20, 20, 20, 20};
__device__ static const uint16_t c_start_long[kMaxBfus] = {
    0,   8,   16,  24,  32,  36,  40,  44,  48,  56,  64,  72,  80,  86,  92,  98,  104, 110, 116, 122, 128, 134, 140, 146, 152, 159,
    166, 173, 180, 189, 198, 207, 216, 226, 236, 246, 256, 268, 280, 292, 304, 316, 328, 340, 352, 372, 392, 412, 432, 452, 472, 492};
__device__ static const uint16_t c_start_short[kMaxBfus] = {
    0,   32,  64,  96,  8,   40,  72,  104, 12,  44,  76,  108, 20,  52,  84,  116, 26,  58,  90,  122, 128, 160, 192, 224, 134, 166,
    198, 230, 141, 173, 205, 237, 150, 182, 214, 246, 256, 288, 320, 352, 384, 416, 448, 480, 268, 300, 332, 364, 396, 428, 460, 492};
__device__ __forceinline__ int bfu_amount(int idx)  // BfuAmountTab
{
    return idx == 0 ? 20 : 24 + 4 * idx;
}
__device__ __forceinline__ int bfu_band(int i) { return i < 20 ? 0 : i < 36 ? 1 : 2; }

struct LogfTab {
    double tab[16][2];
    double ln2;
    double poly[3];
};
static_assert(offsetof(Tables, logf_poly) - offsetof(Tables, logf_tab) == sizeof(double) * 33, "Tables keeps the logf data together");

// logf of glibc 2.35, FMA build (sysdeps/ieee754/flt-32/e_logf.c with the multiarch -mfma variant): normal x only.
__device__ __forceinline__ float at1_logf(const LogfTab* L, float x)
{
    const uint32_t ix = __float_as_uint(x);
    if (ix == 0x3f800000u) return 0.0f;
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) % 16;
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    const double z = (double)__uint_as_float(iz);
    const double r = fma(z, L->tab[i][0], -1.0);
    const double y0 = fma((double)k, L->ln2, L->tab[i][1]);
    const double r2 = r * r;
    double y = fma(L->poly[1], r, L->poly[2]);
    y = fma(L->poly[0], r2, y);
    y = fma(y, r2, y0 + r);
    return (float)y;
}
// log10f of glibc 2.35 (sysdeps/ieee754/flt-32/e_log10f.c): x >= 0 or NaN-free input as calculateRMS produces it.
__device__ __forceinline__ float at1_log10f(const LogfTab* L, float x)
{
    int32_t hx = (int32_t)__float_as_uint(x);
    int k = 0;
    if (hx < 0x00800000) {
        if ((hx & 0x7fffffff) == 0) return -__builtin_inff();
        if (hx < 0) return __builtin_nanf("");
        k -= 25;
        x *= 3.3554432000e+07f;
        hx = (int32_t)__float_as_uint(x);
    }
    if (hx >= 0x7f800000) return x + x;
    k += (hx >> 23) - 127;
    const int i = (int)(((uint32_t)k & 0x80000000u) >> 31);
    hx = (hx & 0x007fffff) | ((0x7f - i) << 23);
    const float y = (float)(k + i);
    const float m = __uint_as_float((uint32_t)hx);
    const float z = y * __uint_as_float(0x355427dbu) + __uint_as_float(0x3ede5bd9u) * at1_logf(L, m);
    return z + y * __uint_as_float(0x3e9a2080u);
}

struct FrontParams {
    const Tables* T;
    const float* pcm;    // [S][n_frames * 512][nch]
    const float* hist;   // [S][512][nch]: the PCM block before this call's first one
    int32_t n_frames, nch;
    int32_t first;       // call starts at the stream start: the detectors' LastEnergy is 0.0
    int32_t window_auto, window_mask;
    int32_t debug;       // AT1HIP_DEBUG_STOP: leave the kernel after phase n (timing experiments only)
    float* specs;        // [S][F][nch][512] MDCT spectrum (kept for the tap interface)
    float* values;       // [S][F][nch][512] scaled mantissa sources, BFU after BFU
    float* energy;       // [S][F][nch][52]
    uint8_t* sfi;        // [S][F][nch][64]
    int32_t* mask;       // [S][F][nch] window mask (bit 0 low, 1 mid, 2 high band short)
    float* loud_ch;      // [S][F][nch]
};

// TQmf<N>::Analysis (qmf/qmf.h:47-64) for one output pair; b points at in[j] of the reference's loop (j = 2 m), 8-byte
// aligned. The two 24-tap running sums are independent, so they ride in the two halves of packed fp32 operations:
// Wp[i] = (QmfWindow[2i+1], QmfWindow[2i]) against the naturally ordered sample pair (in[j-2i], in[j-2i+1]).
__device__ __forceinline__ void qmf_pair(const f2* Wp, const float* b, float& lower, float& upper)
{
    f2 acc = at3::mk2(0.0f, 0.0f);   // (upper-tap sum, lower-tap sum)
#pragma unroll
    for (int i = 0; i < 24; ++i) acc += Wp[i] * *reinterpret_cast<const f2*>(b - 2 * i);
    upper = acc.y - acc.x;
    lower = acc.y + acc.x;
}

// The windowed MDCT input buffer of TAtrac1MDCT::Mdct (atrac1denc.cpp:83-90) at offset o: src = the band's samples with
// index 0 at the unit's first one (the previous unit's tail at negative indices), B = 128 / 256 samples per unit,
// k = short block number. Long: [zeros | sine-rising overlap (32) | B samples, the last 32 sine-falling | zeros].
__device__ __forceinline__ float mdct_in(const float* src, const float* sine, int B, bool sh, int k, int o)
{
    if (sh) return o < 32 ? sine[o] * src[32 * (k - 1) + o] : sine[63 - o] * src[32 * k + o - 32];
    const int i = o - (B == 256 ? 112 : 48) - 32;
    if (i < -32 || i >= B) return 0.0f;
    const float v = src[i];
    if (i < 0) return sine[i + 32] * v;
    if (i >= B - 32) return sine[B - 1 - i] * v;
    return v;
}

__global__ __launch_bounds__(256) void k_at1_front(FrontParams p)
{
    // window of band signals kept per workgroup, indices relative to the sound unit's first sample of each rate
    __shared__ __attribute__((aligned(16))) float s_pcm[800];   // t  in [-288, 512)
    __shared__ __attribute__((aligned(16))) float s_lo1[376];                             // m  in [-118, 256)   first-stage lower band
    __shared__ float s_up1[332];                                // m  in [-75, 256)    first-stage upper band; hi[i] = up1[i - 39]
    __shared__ float s_low[166], s_mid[164];                    // q  in [-36, 128); s_low[164] = 0 for the detector
    __shared__ float s_dmid[166], s_dhi[294];                   // InvertSpectr'ed mid [-36,128] / high [-36,256] band, last = 0
    __shared__ float s_filt[560];                               // detector high-pass output: low/mid [-16,128), hi [-16,256)
    __shared__ float s_rms[3][17];
    __shared__ __attribute__((aligned(16))) float s_tmp[512];   // e * LoudnessCurve
    __shared__ __attribute__((aligned(16))) at3::cpx s_f[256];
    __shared__ __attribute__((aligned(16))) float s_specs[512];
    __shared__ __attribute__((aligned(16))) f2 s_win[24];
    __shared__ float s_scale[64], s_sine[32];
    __shared__ __attribute__((aligned(8))) float s_fir[10];
    __shared__ LogfTab s_logf;
    __shared__ int s_mask;

    const Tables* T = p.T;
    const int f = blockIdx.x, sc = blockIdx.y;
    const int nch = p.nch, s = sc / nch, ch = sc - s * nch;
    const int tid = threadIdx.x;
    const size_t item = ((size_t)s * p.n_frames + f) * nch + ch;

    for (int j = tid; j < 800; j += 256) {
        const int t = 512 * f - 288 + j;
        s_pcm[j] = t >= 0 ? p.pcm[((size_t)s * p.n_frames * 512 + t) * nch + ch] : p.hist[((size_t)s * 512 + (512 + t)) * nch + ch];
    }
    if (tid < 48) reinterpret_cast<float*>(s_win)[tid] = T->qmf_win[tid ^ 1];
    else if (tid < 112) s_scale[tid - 48] = T->scale[tid - 48];
    else if (tid < 144) s_sine[tid - 112] = T->sine[tid - 112];
    else if (tid < 154) s_fir[tid - 144] = T->fir[tid - 144];
    else if (tid < 154 + 36) (&s_logf.tab[0][0])[tid - 154] = (&T->logf_tab[0][0])[tid - 154];
    if (tid == 255) {
        s_mask = 0;
        s_low[164] = 0.0f;   // HPFBuffer[BlockSz + 20] is never written: the sample after the block reads as 0
        s_dmid[164] = 0.0f;
        s_dhi[292] = 0.0f;
    }
    __syncthreads();

    if (p.debug == 1) return;
    // Atrac1AnalysisFilterBank::Analysis (atrac/at1/atrac1_qmf.h:37-43): Qmf1 over the PCM ...
    for (int j = tid; j < 374; j += 256) {
        const int m = j - 118;
        float lo, up;
        qmf_pair(s_win, s_pcm + (2 * m + 288), lo, up);
        s_lo1[j] = lo;
        if (m >= -75) {
            s_up1[m + 75] = up;
            if (m < 217) s_dhi[m + 75] = (m & 1) ? -up : up;   // hi[i] = up1[i - 39]: i even <=> m odd (InvertSpectr, util.h:51-63)
        }
    }
    __syncthreads();
    if (p.debug == 2) return;
    // ... Qmf2 over its lower half; the upper half is delayed by 39 samples
    if (tid < 164) {
        const int q = tid - 36;
        float lo, up;
        qmf_pair(s_win, s_lo1 + (2 * q + 118), lo, up);
        s_low[tid] = lo;
        s_mid[tid] = up;
        s_dmid[tid] = (q & 1) ? up : -up;
    }
    __syncthreads();

    if (p.debug == 3) return;
    int mask = p.window_mask;
    if (p.window_auto) {
        // TTransientDetector::HPFilter (transient_detector.cpp:48-66) for this unit (512 outputs: each wave stays inside one
        // band) and for the last short block of the previous unit (its LastEnergy, 48 outputs). The two running sums of the
        // reference loop are independent and ride in the halves of packed fp32 operations.
        const f2* firp = reinterpret_cast<const f2*>(s_fir);
        for (int j = tid; j < 560; j += 256) {
            int b, i;
            if (j < 512) {
                b = j < 128 ? 0 : j < 256 ? 1 : 2;
                i = j - (b == 0 ? 0 : b == 1 ? 128 : 256);
            } else {
                if (j >= 512 + 48) break;
                b = (j - 512) >> 4;
                i = ((j - 512) & 15) - 16;
            }
            const float* d = (b == 0 ? s_low : b == 1 ? s_dmid : s_dhi) + 36;
            // the previous unit's own last output saw 0 after its block, not this unit's first sample
            const float nxt = i == -1 ? 0.0f : d[i + 1];
            f2 acc = at3::mk2(d[i - 10], 0.0f);
            acc += firp[0] * (at3::mk2(d[i - 20], d[i - 19]) + at3::mk2(nxt, d[i]));
#pragma unroll
            for (int jj = 2; jj < 9; jj += 2)
                acc += firp[jj >> 1] * (at3::mk2(d[i - 20 + jj], d[i - 19 + jj]) + at3::mk2(d[i + 1 - jj], d[i - jj]));
            s_filt[(b == 0 ? 0 : b == 1 ? 144 : 288) + 16 + i] = (acc.x + acc.y) / 2;
        }
        __syncthreads();
        // calculateRMS over the 16-sample short blocks, 19 log10 (transient_detector.cpp:40-46, 76)
        if (tid < 35) {
            const int b = tid < 9 ? 0 : tid < 18 ? 1 : 2;
            const int k = tid - (b == 0 ? 0 : b == 1 ? 9 : 18);
            const float* fl = s_filt + (b == 0 ? 0 : b == 1 ? 144 : 288) + 16 * k;
            float acc = 0.0f;
            for (int i = 0; i < 16; ++i) acc += fl[i] * fl[i];
            acc /= 16.0f;
            float r = (float)(19.0 * (double)at1_log10f(&s_logf, sqrtf(acc)));
            if (k == 0 && f == 0 && p.first) r = 0.0f;
            s_rms[b][k] = r;
        }
        __syncthreads();
        if (tid < 35) {
            const int b = tid < 9 ? 0 : tid < 18 ? 1 : 2;
            const int k = tid - (b == 0 ? 0 : b == 1 ? 9 : 18);
            if (k > 0) {
                const float r1 = s_rms[b][k], r0 = s_rms[b][k - 1];
                if (r1 - r0 > 16 || r0 - r1 > 20) atomicOr(&s_mask, 1 << b);
            }
        }
        __syncthreads();
        mask = s_mask;
    }

    if (p.debug == 4) return;
    // TAtrac1MDCT::Mdct (atrac1denc.cpp:70-102): TMDCT<N>::operator() pre-rotation (lib/mdct/mdct.h:51-87) straight into
    // the FFT's leaf order; the windowed input buffer of the reference is evaluated where it is read (mdct_in).
    {
        const int b = tid < 64 ? 0 : tid < 128 ? 1 : 2;
        const int c = tid - (b == 0 ? 0 : b == 1 ? 64 : 128);
        const bool sh = (mask >> b) & 1;
        const int B = b == 2 ? 256 : 128;
        const int N = sh ? 64 : 2 * B;
        const int n4 = N >> 2, n34 = 3 * n4, n54 = 5 * n4;
        const int k = sh ? c >> 4 : 0;
        const int pidx = sh ? (c & 15) : c;
        const int n = 2 * pidx;
        const float* src = (b == 0 ? s_low : b == 1 ? s_mid : s_up1) + 36;
        const float* cs = sh ? T->sc64 : (b == 2 ? T->sc512 : T->sc256);
        auto in = [&](int o) { return mdct_in(src, s_sine, B, sh, k, o); };
        float r0, i0;
        if (n < n4) {
            r0 = in(n34 - 1 - n) + in(n34 + n);
            i0 = in(n4 + n) - in(n4 - 1 - n);
        } else {
            r0 = in(n34 - 1 - n) - in(n - n4);
            i0 = in(n4 + n) + in(n54 - 1 - n);
        }
        const float cc = cs[n], ss = cs[n + 1];
        at3::cpx v;
        v.r = r0 * cc + i0 * ss;
        v.i = i0 * cc - r0 * ss;
        const int leaf = sh ? fft_leaf_pos<16>(pidx) : (b == 2 ? fft_leaf_pos<128>(pidx) : fft_leaf_pos<64>(pidx));
        s_f[(b == 0 ? 0 : b == 1 ? 64 : 128) + 16 * k + leaf] = v;
    }
    __syncthreads();
    if (p.debug == 5) return;
    {
        // one band per wave: the stages of a band's transforms only need that wave's own lanes
        const int wave = tid >> 6, lane = tid & 63;
        if (wave == 0) {
            if (mask & 1) fft_lds<16, false, false, true>(s_f, 16, 4, T->tw16, lane, 64);
            else fft_lds<64, false, false, true>(s_f, 64, 1, T->tw64, lane, 64);
        } else if (wave == 1) {
            if (mask & 2) fft_lds<16, false, false, true>(s_f + 64, 16, 4, T->tw16, lane, 64);
            else fft_lds<64, false, false, true>(s_f + 64, 64, 1, T->tw64, lane, 64);
        } else if (wave == 2) {
            if (mask & 4) fft_lds<16, false, false, true>(s_f + 128, 16, 8, T->tw16, lane, 64);
            else fft_lds<128, false, false, true>(s_f + 128, 128, 1, T->tw128, lane, 64);
        }
    }
    __syncthreads();
    if (p.debug == 6) return;
    // post-rotation (mdct.h:89-101), the high band's short-window gain and the mirrored bands (atrac1denc.cpp:92-97)
    {
        const int b = tid < 64 ? 0 : tid < 128 ? 1 : 2;
        const int c = tid - (b == 0 ? 0 : b == 1 ? 64 : 128);
        const bool sh = (mask >> b) & 1;
        const int n2 = sh ? 32 : (b == 2 ? 256 : 128);
        const int k = sh ? c >> 4 : 0;
        const int n = 2 * (sh ? (c & 15) : c);
        const float* cs = sh ? T->sc64 : (b == 2 ? T->sc512 : T->sc256);
        const at3::cpx v = s_f[(b == 0 ? 0 : b == 1 ? 64 : 128) + c];
        const float cc = cs[n], ss = cs[n + 1];
        float o1 = -v.r * cc - v.i * ss;
        float o2 = -v.r * ss + v.i * cc;
        if (sh && b == 2) {
            o1 *= 2.0f;
            o2 *= 2.0f;
        }
        float* dst = s_specs + (b == 0 ? 0 : b == 1 ? 128 : 256) + 32 * k;
        if (b) {
            dst[n2 - 1 - n] = o1;
            dst[n] = o2;
        } else {
            dst[n] = o1;
            dst[n2 - 1 - n] = o2;
        }
    }
    __syncthreads();

    if (p.debug == 7) return;
    for (int j = tid; j < 512; j += 256) {
        const float v = s_specs[j];
        const float e = v * v;
        s_tmp[j] = e * T->loud[j];
        p.specs[item * 512 + j] = v;
    }
    __syncthreads();
    if (tid == 0) {
        // per-channel loudness (atrac1denc.cpp:235-240): one running sum over the 512 lines
        float l = 0.0f;
        const float4* q = reinterpret_cast<const float4*>(s_tmp);
        for (int i = 0; i < 128; ++i) {
            const float4 v = q[i];
            l += v.x;
            l += v.y;
            l += v.z;
            l += v.w;
        }
        p.loud_ch[item] = l;
        p.mask[item] = mask;
    }
    if (tid >= 64 && tid < 64 + kMaxBfus) {
        // TScaler<TAtrac1Data>::Scale / ScaleFrame (atrac/atrac_scale.cpp:141-188)
        const int bfu = tid - 64;
        const bool sh = (mask >> bfu_band(bfu)) & 1;
        const float* in = s_specs + (sh ? c_start_short[bfu] : c_start_long[bfu]);
        const int len = c_spb[bfu];
        float max_abs = 0.0f;
        for (int i = 0; i < len; ++i) {
            const float a = fabsf(in[i]);
            if (a > max_abs) max_abs = a;
        }
        if (max_abs > 1.0f) max_abs = 1.0f;
        int lo = 0, hi = 63;
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const int mid = (lo + hi) >> 1;
            if (s_scale[mid] < max_abs) lo = mid + 1;
            else hi = mid;
        }
        const float sf = s_scale[lo];
        float e = 0.0f;
        float* vals = p.values + item * 512 + c_start_long[bfu];
        for (int i = 0; i < len; ++i) {
            const float xv = in[i];
            float v = xv / sf;
            e += xv * xv;
            if (fabsf(v) >= 1.0f) v = (v > 0) ? 0.99999f : -0.99999f;
            vals[i] = v;
        }
        p.sfi[item * 64 + bfu] = (uint8_t)lo;
        p.energy[item * kMaxBfus + bfu] = e;
    }
}

struct ScanParams {
    const int32_t* mask;
    const float* loud_ch;
    float* loud_state;   // [S] carried between calls
    float* loud_track;   // [S][F] Loudness after the unit's update
    int32_t n_streams, n_frames, nch;
};

// TrackLoudness (atrac/atrac_psy_common.h:46-54) as atrac1denc.cpp:243-247 applies it. One wave per stream: the lanes
// stage 256 sound units' masks and loudness sums in LDS with coalesced loads, lane 0 runs the recursion out of LDS (a
// dependent global load per step would cost far more than the three double operations of the step itself), the lanes
// store the tracked values.
__global__ __launch_bounds__(64) void k_at1_loud_scan(ScanParams p)
{
    __shared__ float s_l0[256], s_l1[256], s_track[256];
    __shared__ uint8_t s_long[256];   // bit 0: channel 0 all-long, bit 1: both channels all-long
    const int s = blockIdx.x, lane = threadIdx.x;
    float L = p.loud_state[s];
    for (int base = 0; base < p.n_frames; base += 256) {
        const int cnt = p.n_frames - base < 256 ? p.n_frames - base : 256;
        for (int j = lane; j < cnt; j += 64) {
            const size_t it = ((size_t)s * p.n_frames + base + j) * p.nch;
            const int m0 = p.mask[it];
            const int m1 = p.nch == 2 ? p.mask[it + 1] : 1;
            s_l0[j] = p.loud_ch[it];
            s_l1[j] = p.nch == 2 ? p.loud_ch[it + 1] : 0.0f;
            s_long[j] = (uint8_t)((m0 == 0 ? 1 : 0) | ((m0 == 0 && m1 == 0) ? 2 : 0));
        }
        wave_sync();
        if (lane == 0) {
            for (int j = 0; j < cnt; ++j) {
                const int fl = s_long[j];
                if (fl & 2) L = (float)(0.98 * (double)L + 0.01 * (double)(s_l0[j] + s_l1[j]));
                else if (fl & 1) L = (float)(0.98 * (double)L + 0.02 * (double)s_l0[j]);
                s_track[j] = L;
            }
        }
        wave_sync();
        for (int j = lane; j < cnt; j += 64) p.loud_track[(size_t)s * p.n_frames + base + j] = s_track[j];
        wave_sync();
    }
    if (lane == 0) p.loud_state[s] = L;
}

struct PackParams {
    const Tables* T;
    const float* values;
    const float* energy;
    const uint8_t* sfi;
    const int32_t* mask;
    const float* loud_track;
    uint8_t* out;          // [S][F][nch][212]
    int32_t n_items;       // S * F * nch
    int32_t nch;
    int32_t bfu_idx_const;
};

__device__ __forceinline__ int wave_sum_i32(int v, int lane)
{
    return __builtin_amdgcn_readlane(wave_inclusive_scan(v, lane), 63);
}

// TAt1BitAlloc::Write (atrac/at1/atrac1_bitalloc.cpp:385-405): the part encoders driven by
// TBitStreamEncoder::DoRun (lib/bs_encode/encode.cpp:57-129), then TBfuAlloc::Dump (:297-338). lane = BFU.
__global__ __launch_bounds__(256) void k_at1_alloc_pack(PackParams p)
{
    __shared__ uint32_t s_words[4][56];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + wave;
    if (item >= p.n_items) return;
    const Tables* T = p.T;
    uint32_t* W = s_words[wave];
    if (lane < 56) W[lane] = 0;

    const int mask = p.mask[item];
    const bool in_tab = lane < kMaxBfus;
    const int bl = in_tab ? lane : 0;
    const int band = bfu_band(bl);
    const bool sh = (mask >> band) & 1;
    const int sfi = in_tab ? p.sfi[(size_t)item * 64 + lane] : 0;
    const float energy = p.energy[(size_t)item * kMaxBfus + bl];
    const int spb = in_tab ? c_spb[bl] : 0;
    const float loudness = p.loud_track[item / p.nch] / 0.006f;
    const float fix = sh ? T->fix_short[bl] : T->fix_long[bl];
    const bool gate = !sh && energy < T->ath_bfu[bl] * loudness;
    const float spread = 0.4f;
    const float base = spread * ((float)sfi / 3.2f) + (1.0f - spread) * fix;

    const int sum_low = wave_sum_i32(lane < 20 ? sfi : 0, lane);
    int bfu_idx = p.bfu_idx_const ? p.bfu_idx_const - 1 : 7;
    const bool auto_bfu = !p.bfu_idx_const;
    int bits = 0, n = 0;
    for (;;) {
        n = bfu_amount(bfu_idx);
        const int target = kFrame * 8 - 3 - 32 - 2 - 3 - n * 10;
        // CalcLowToMidTilt (:146-161): sums of small integers are exact in float
        const int n_mid = (n < 36 ? n : 36) - 20;
        const int sum_mid = wave_sum_i32((lane >= 20 && lane < 20 + n_mid) ? sfi : 0, lane);
        const float tilt = n_mid ? (float)sum_low / (float)20 - (float)sum_mid / (float)n_mid : 0.0f;
        const float mid_bias = fminf(1.5f, 0.3f * fmaxf(0.0f, tilt - 7.0f));
        const float bias = band == 0 ? 0.0f : band == 1 ? mid_bias : mid_bias * 0.5f;
        float min_l = -3, max_l = 15, cur_l = 0, last_l = 15;
        int used;
        for (;;) {
            const bool exhausted = max_l <= min_l;
            if (!exhausted) cur_l = (float)((double)(max_l + min_l) / 2.0);
            const float shift = exhausted ? last_l : cur_l;
            const int tmp = (int)(base - shift + bias);
            bits = (lane >= n || gate) ? 0 : tmp > 16 ? 16 : tmp < 2 ? 0 : tmp;
            used = wave_sum_i32(spb * bits, lane);
            if (exhausted || used == target) break;
            if (used < target) {
                last_l = cur_l;
                max_l = cur_l - 0.01f;
            } else {
                min_l = cur_l + 0.01f;
            }
        }
        if (auto_bfu) {
            // GetMaxUsedBfuId (:228-252): drop trailing BFU groups that got no bits at all
            const uint64_t nz = __ballot(bits != 0);
            int idx = bfu_idx;
            while (idx) {
                const int lo = bfu_amount(idx - 1), hi = bfu_amount(idx);
                const uint64_t grp = ((1ull << hi) - 1) & ~((1ull << lo) - 1);
                if (nz & grp) break;
                --idx;
            }
            if (idx < bfu_idx) {
                --bfu_idx;
                continue;
            }
        }
        // TBitsBooster::ApplyBoost (:94-128) on wave-uniform copies of the twelve candidates
        int surplus = target - used;
        const int key = surplus > 12 ? 12 : surplus;
        const int max_it = key >= 12 ? 12 : key >= 10 ? 9 : key >= 6 ? 5 : 0;
        if (max_it) {
            int cb[12];
#pragma unroll
            for (int j = 0; j < 12; ++j) cb[j] = __builtin_amdgcn_readlane(bits, j < 5 ? 18 + j : 27 + j);
            while (surplus >= 6) {
                bool done = true;
#pragma unroll
                for (int j = 0; j < 12; ++j) {
                    const int cur_bits = j < 5 ? 6 : j < 9 ? 10 : 12;
                    const int pos = j < 5 ? 18 + j : 27 + j;
                    if (j >= max_it || pos >= n) break;
                    if (cb[j] == 16) continue;
                    const int per_spec = cb[j] ? 1 : 2;
                    if (cb[j] == 0 && cur_bits * 2 > surplus) continue;
                    if (cur_bits * per_spec > surplus) continue;
                    cb[j] += per_spec;
                    surplus -= cur_bits * per_spec;
                    done = false;
                }
                if (done) break;
            }
#pragma unroll
            for (int j = 0; j < 12; ++j)
                if (lane == (j < 5 ? 18 + j : 27 + j)) bits = cb[j];
        }
        break;
    }

    // Dump: MSB-first bit string into big-endian words
    auto put = [&](int off, uint32_t val, int nb) {
        const int w = off >> 5, sft = off & 31;
        const int room = 32 - sft;
        if (nb <= room) {
            atomicOr(&W[w], val << (room - nb));
        } else {
            atomicOr(&W[w], val >> (nb - room));
            atomicOr(&W[w + 1], val << (32 - (nb - room)));
        }
    };
    wave_sync();
    if (lane == 0) {
        const int lc0 = (mask & 1) ? 2 : 0, lc1 = (mask & 2) ? 2 : 0, lc2 = (mask & 4) ? 3 : 0;
        put(0, (uint32_t)(((2 - lc0) << 14) | ((2 - lc1) << 12) | ((3 - lc2) << 10) | (bfu_idx << 5)), 16);
    }
    const int incl = wave_inclusive_scan(spb * bits, lane);
    if (lane < n) {
        put(16 + 4 * lane, (uint32_t)(bits ? bits - 1 : 0), 4);
        put(16 + 4 * n + 6 * lane, (uint32_t)sfi, 6);
        if (bits > 1) {
            int off = 16 + 10 * n + incl - spb * bits;
            const float multiple = (float)((1 << (bits - 1)) - 1);
            const float* vals = p.values + (size_t)item * 512 + c_start_long[lane];
            const uint32_t field = (1u << bits) - 1;
            for (int k = 0; k < spb; ++k) {
                const int q = __float2int_rn(vals[k] * multiple);
                put(off, (uint32_t)q & field, bits);
                off += bits;
            }
        }
    }
    wave_sync();
    if (lane < 53) {
        const uint32_t w = W[lane];
        reinterpret_cast<uint32_t*>(p.out + (size_t)item * kFrame)[lane] = __builtin_bswap32(w);
    }
}

// carry the last PCM block of the call
__global__ void k_at1_state(const float* pcm, float* hist, int n_frames, int nch, int n_streams)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = 512 * nch;
    if (i >= n_streams * per) return;
    const int s = i / per, r = i - s * per;
    hist[i] = pcm[((size_t)s * n_frames + (n_frames - 1)) * per + r];
}

}  // namespace at1
