// ATRAC1 encode kernels (gfx950), SURVEY.md 8(f) row f3: the body of TAtrac1Encoder::GetLambda (atrac1denc.cpp:180-255)
// for a batch of streams. Nothing in the path carries a recursion across sound units except the loudness tracker, so
// the work splits into
//   k_at1_front      one wavefront per (stream, sound unit, channel): QMF tree, transient detection, block-switched
//                    MDCT, scale factors
//   k_at1_loud       per-channel loudness: the ordered 512-line sums, a lane per sound unit
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
// BFU that owns position i of the BFU-ordered value array (the long-window line layout, SpecsStartLong)
__device__ static const uint8_t c_bfu_of_pos[512] __attribute__((aligned(16))) = {
    0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3,
    4, 4, 4, 4, 5, 5, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8, 8, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9, 9, 9,
    10, 10, 10, 10, 10, 10, 10, 10, 11, 11, 11, 11, 11, 11, 11, 11, 12, 12, 12, 12, 12, 12, 13, 13, 13, 13, 13, 13, 14, 14, 14, 14,
    14, 14, 15, 15, 15, 15, 15, 15, 16, 16, 16, 16, 16, 16, 17, 17, 17, 17, 17, 17, 18, 18, 18, 18, 18, 18, 19, 19, 19, 19, 19, 19,
    20, 20, 20, 20, 20, 20, 21, 21, 21, 21, 21, 21, 22, 22, 22, 22, 22, 22, 23, 23, 23, 23, 23, 23, 24, 24, 24, 24, 24, 24, 24, 25,
    25, 25, 25, 25, 25, 25, 26, 26, 26, 26, 26, 26, 26, 27, 27, 27, 27, 27, 27, 27, 28, 28, 28, 28, 28, 28, 28, 28, 28, 29, 29, 29,
    29, 29, 29, 29, 29, 29, 30, 30, 30, 30, 30, 30, 30, 30, 30, 31, 31, 31, 31, 31, 31, 31, 31, 31, 32, 32, 32, 32, 32, 32, 32, 32,
    32, 32, 33, 33, 33, 33, 33, 33, 33, 33, 33, 33, 34, 34, 34, 34, 34, 34, 34, 34, 34, 34, 35, 35, 35, 35, 35, 35, 35, 35, 35, 35,
    36, 36, 36, 36, 36, 36, 36, 36, 36, 36, 36, 36, 37, 37, 37, 37, 37, 37, 37, 37, 37, 37, 37, 37, 38, 38, 38, 38, 38, 38, 38, 38,
    38, 38, 38, 38, 39, 39, 39, 39, 39, 39, 39, 39, 39, 39, 39, 39, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 41, 41, 41, 41,
    41, 41, 41, 41, 41, 41, 41, 41, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 43, 43, 43, 43, 43, 43, 43, 43, 43, 43, 43, 43,
    44, 44, 44, 44, 44, 44, 44, 44, 44, 44, 44, 44, 44, 44, 44, 44, 44, 44, 44, 44, 45, 45, 45, 45, 45, 45, 45, 45, 45, 45, 45, 45,
    45, 45, 45, 45, 45, 45, 45, 45, 46, 46, 46, 46, 46, 46, 46, 46, 46, 46, 46, 46, 46, 46, 46, 46, 46, 46, 46, 46, 47, 47, 47, 47,
    47, 47, 47, 47, 47, 47, 47, 47, 47, 47, 47, 47, 47, 47, 47, 47, 48, 48, 48, 48, 48, 48, 48, 48, 48, 48, 48, 48, 48, 48, 48, 48,
    48, 48, 48, 48, 49, 49, 49, 49, 49, 49, 49, 49, 49, 49, 49, 49, 49, 49, 49, 49, 49, 49, 49, 49, 50, 50, 50, 50, 50, 50, 50, 50,
    50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 51, 51, 51, 51, 51, 51, 51, 51, 51, 51, 51, 51, 51, 51, 51, 51, 51, 51, 51, 51,};
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
static_assert(512 + 166 + 164 + 166 <= 1008, "the second QMF stage's outputs fit behind the spectrum");
static_assert(offsetof(Tables, sc256) - offsetof(Tables, sc512) == 256 * 4 && offsetof(Tables, sc64) - offsetof(Tables, sc512) == 384 * 4,
              "the three MDCT rotation tables are staged as one block");

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

// LDS layout of the two QMF inputs: one extra pair of floats after every 8 (index p lives at p + 2 (p >> 3)). A thread
// that produces four adjacent output pairs walks samples 8 floats apart from its neighbour's; with the padding the lanes'
// 8-byte reads start 10 floats apart and 16 consecutive lanes cover all 32 banks exactly once.
__device__ __forceinline__ constexpr int qmf_pad(int p) { return p + 2 * (p >> 3); }

// TQmf<N>::Analysis (qmf/qmf.h:47-64) for FOUR adjacent output pairs m0 .. m0+3 of one thread: the 4 x 48 taps touch only
// 27 distinct sample pairs, read once (the one-pair-per-thread form re-read every sample 24 times and was bound by LDS
// bandwidth). `base` = padded address of sample pair (in[2 m0 - 46], in[2 m0 - 45]) for a thread whose unpadded float
// index of that pair is = PHASE (mod 8), so every later offset is a compile-time constant. The two 24-tap running sums
// of an output are independent and ride in the halves of packed fp32 operations: W[i] = (QmfWindow[2i+1], QmfWindow[2i])
// (wave-uniform, scalar registers) against the naturally ordered sample pair (in[j-2i], in[j-2i+1]).
template <int PHASE>
__device__ __forceinline__ void qmf_quad(const float* __restrict__ win, const float* base, float (&lower)[4], float (&upper)[4])
{
    f2 P[27];
#pragma unroll
    for (int j = 0; j < 27; ++j) P[j] = *reinterpret_cast<const f2*>(base + (qmf_pad(PHASE + 2 * j) - qmf_pad(PHASE)));
    f2 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = at3::mk2(0.0f, 0.0f);   // (upper-tap sum, lower-tap sum)
#pragma unroll
    for (int i = 0; i < 24; ++i) {
        const f2 w = at3::mk2(win[2 * i + 1], win[2 * i]);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += w * P[23 + r - i];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        upper[r] = acc[r].y - acc[r].x;
        lower[r] = acc[r].y + acc[r].x;
    }
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

// One WAVEFRONT per (stream, sound unit, channel). The first version gave a sound unit 256 threads and separated its
// phases by workgroup barriers; no phase has 256 independent jobs (the first QMF stage has 94, the second 41, the
// detector's level comparison 35), so most of a workgroup's wavefronts spent most of its life parked at a barrier while
// occupying registers and LDS. Here the 64 lanes walk every phase's jobs in rounds, phases are separated by wave-level
// rendezvous only (the LDS pipeline executes one wavefront's instructions in order), the rotation tables, the window and
// the logarithm's table are read where they are used (they stay in the L1 of a CU that runs nothing else) and the
// scale-factor table lives in the lanes: 11 KB of LDS per unit, fourteen units per CU.
__global__ __launch_bounds__(64) void k_at1_front(FrontParams p)
{
    // Two regions are reused once their first tenant is dead: the PCM window becomes the spectrum; the first QMF stage's
    // lower band becomes the detector's filter output and then the FFT buffer.
    __shared__ __attribute__((aligned(16))) float s_region_a[1008];
    __shared__ __attribute__((aligned(16))) float s_region_b[560];
    float* const s_pcm = s_region_a;                                     // t  in [-288, 512), padded (qmf_pad)
    float* const s_specs = s_region_a;                                   // after the first QMF stage
    float* const s_lo1 = s_region_b;                                     // m  in [-118, 256)   first-stage lower band, padded (qmf_pad): 480 floats
    float* const s_filt = s_region_b;                                    // detector high-pass output: low/mid [-16,128), hi [-16,256)
    at3::cpx* const s_f = reinterpret_cast<at3::cpx*>(s_region_b);       // 256 points, after the detector
    __shared__ float s_up1[332];                                         // m  in [-75, 256)    first-stage upper band; hi[i] = up1[i - 39]
    // the second stage's outputs live behind the spectrum's 512 floats in the PCM window's storage (dead after the first stage)
    float* const s_low = s_region_a + 512;                               // q  in [-36, 128); s_low[164] = 0 for the detector
    float* const s_mid = s_region_a + 512 + 166;                         // 164 floats
    float* const s_dmid = s_region_a + 512 + 166 + 164;                  // InvertSpectr'ed mid band [-36,128], last = 0: 166 floats
    __shared__ float s_dhi[294];                                         // InvertSpectr'ed high band [-36,256], last = 0
    __shared__ float s_rms[3][17];
    __shared__ float s_sf[kMaxBfus];
    __shared__ int s_srcoff[kMaxBfus];
    __shared__ float s_sine[32];
    __shared__ LogfTab s_logf;

    const Tables* T = p.T;
    const int f = blockIdx.x, sc = blockIdx.y;
    const int nch = p.nch, s = sc / nch, ch = sc - s * nch;
    const int lane = threadIdx.x;
    const size_t item = ((size_t)s * p.n_frames + f) * nch + ch;

    // (asked for by every lane at clamped indices, AHEAD of the PCM window - loads return in issue order - and stored below: as
    // load-store pairs under lane conditions behind the window each was a round trip of its own)
    const float my_scale = T->scale[lane];   // ScaleTable has 64 entries: looked up across the lanes
    const float sine_v = T->sine[lane & 31];
    const double logf_v = (&T->logf_tab[0][0])[lane < 36 ? lane : 35];   // 16 x 2 table entries, ln 2, three coefficients
    __builtin_amdgcn_sched_barrier(0);
    {
        float v[13];
#pragma unroll
        for (int r = 0; r < 13; ++r) {
            const int j = lane + 64 * r;
            const int t = 512 * f - 288 + j;
            v[r] = j >= 800 ? 0.0f
                 : t >= 0   ? p.pcm[((size_t)s * p.n_frames * 512 + t) * nch + ch]
                            : p.hist[((size_t)s * 512 + (512 + t)) * nch + ch];
        }
#pragma unroll
        for (int r = 0; r < 13; ++r)
            if (lane + 64 * r < 800) s_pcm[qmf_pad(lane + 64 * r)] = v[r];
    }
    // everything the later phases read from tables at indices that are known now is requested now, behind the PCM window:
    // a global load issued where its value is needed costs a wavefront a microsecond of its 24
    // rotation factors of the lane's point in the four pre- / post-rotation rounds, long window (index 2 c), and of its
    // point in a short block (index 2 (c & 15), the same in every round); sc512 | sc256 | sc64 are contiguous
    f2 cs_long[4];
#pragma unroll
    for (int round = 0; round < 4; ++round)
        cs_long[round] = *reinterpret_cast<const f2*>(&T->sc512[0] + (round < 2 ? 256 : 0) + 2 * (lane + (round == 3 ? 64 : 0)));
    const f2 cs_short = *reinterpret_cast<const f2*>(&T->sc512[0] + 384 + 2 * (lane & 15));
    const uint32_t pos_bfu4[2] = {*reinterpret_cast<const uint32_t*>(c_bfu_of_pos + 4 * lane), *reinterpret_cast<const uint32_t*>(c_bfu_of_pos + 4 * (lane + 64))};
    if (lane < 32) s_sine[lane] = sine_v;
    if (lane < 36) (&s_logf.tab[0][0])[lane] = logf_v;
    int bf_long = 0, bf_short = 0, bf_len = 0;
    if (lane < kMaxBfus) {
        bf_long = c_start_long[lane];
        bf_short = c_start_short[lane];
        bf_len = c_spb[lane];
    }
    if (lane == 63) s_dhi[292] = 0.0f;   // HPFBuffer[BlockSz + 20] is never written: the sample after the block reads as 0
    wave_sync();

    if (p.debug == 1) return;
    // Atrac1AnalysisFilterBank::Analysis (atrac/at1/atrac1_qmf.h:37-43): Qmf1 over the PCM ...
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        const int tt = lane + 64 * round;
        if (tt < 94) {
            // outputs m = -118 + 4 tt + r; the first pair read is PCM index 2 m - 46 = 8 tt - 282, window index 8 tt + 6
            float lo[4], up[4];
            qmf_quad<6>(T->qmf_win, s_pcm + qmf_pad(8 * tt + 6), lo, up);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 4 * tt + r, m = j - 118;
                if (j < 374) {
                    s_lo1[qmf_pad(j)] = lo[r];
                    if (m >= -75) {
                        s_up1[m + 75] = up[r];
                        if (m < 217) s_dhi[m + 75] = (m & 1) ? -up[r] : up[r];   // hi[i] = up1[i - 39]: i even <=> m odd (InvertSpectr, util.h:51-63)
                    }
                }
            }
        }
    }
    wave_sync();
    if (p.debug == 2) return;
    // ... Qmf2 over its lower half; the upper half is delayed by 39 samples
    if (lane < 41) {
        // outputs q = -36 + 4 lane + r; the first pair read is lower-band index 2 q - 46 = 8 lane - 118, buffer index 8 lane
        float lo[4], up[4];
        qmf_quad<0>(T->qmf_win, s_lo1 + qmf_pad(8 * lane), lo, up);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 4 * lane + r, q = j - 36;
            s_low[j] = lo[r];
            s_mid[j] = up[r];
            s_dmid[j] = (q & 1) ? up[r] : -up[r];
        }
    }
    if (lane == 63) {
        s_low[164] = 0.0f;   // (as s_dhi[292] above)
        s_dmid[164] = 0.0f;
    }
    wave_sync();
    if (p.debug == 3) return;
    int mask = p.window_mask;
    if (p.window_auto) {
        // TTransientDetector::HPFilter (transient_detector.cpp:48-66) for this unit (512 outputs: a round of 64 stays inside
        // one band) and for the last short block of the previous unit (its LastEnergy, 48 outputs). The two running sums
        // of the reference loop are independent and ride in the halves of packed fp32 operations.
        f2 fir[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) fir[q] = at3::mk2(T->fir[2 * q], T->fir[2 * q + 1]);   // wave-uniform: scalar registers
#pragma unroll
        for (int round = 0; round < 9; ++round) {
            const int j = lane + 64 * round;
            int b, i;
            if (round < 8) {
                b = round < 2 ? 0 : round < 4 ? 1 : 2;
                i = j - (b == 0 ? 0 : b == 1 ? 128 : 256);
            } else {
                b = lane >> 4;
                i = (lane & 15) - 16;
            }
            if (round < 8 || lane < 48) {
                const float* d = (b == 0 ? s_low : b == 1 ? s_dmid : s_dhi) + 36;
                // the previous unit's own last output saw 0 after its block, not this unit's first sample
                const float nxt = i == -1 ? 0.0f : d[i + 1];
                f2 acc = at3::mk2(d[i - 10], 0.0f);
                acc += fir[0] * (at3::mk2(d[i - 20], d[i - 19]) + at3::mk2(nxt, d[i]));
#pragma unroll
                for (int jj = 2; jj < 9; jj += 2)
                    acc += fir[jj >> 1] * (at3::mk2(d[i - 20 + jj], d[i - 19 + jj]) + at3::mk2(d[i + 1 - jj], d[i - jj]));
                s_filt[(b == 0 ? 0 : b == 1 ? 144 : 288) + 16 + i] = (acc.x + acc.y) / 2;
            }
        }
        wave_sync();
        // calculateRMS over the 16-sample short blocks, 19 log10, the +16 / -20 jumps (transient_detector.cpp:40-46, 76-88):
        // 35 lanes, exchanged inside the wave
        {
            const int b = lane < 9 ? 0 : lane < 18 ? 1 : 2;
            const int k = lane - (b == 0 ? 0 : b == 1 ? 9 : 18);
            float r = 0.0f;
            if (lane < 35) {
                const float* fl = s_filt + (b == 0 ? 0 : b == 1 ? 144 : 288) + 16 * k;
                float acc = 0.0f;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc += fl[i] * fl[i];
                acc /= 16.0f;
                r = (float)(19.0 * (double)at1_log10f(&s_logf, sqrtf(acc)));
                if (k == 0 && f == 0 && p.first) r = 0.0f;
                s_rms[b][k] = r;
            }
            wave_sync();
            bool jump = false;
            if (lane < 35 && k > 0) {
                const float r0 = s_rms[b][k - 1];
                jump = r - r0 > 16 || r0 - r > 20;
            }
            const unsigned long long jl = __ballot(jump && b == 0), jm = __ballot(jump && b == 1), jh = __ballot(jump && b == 2);
            mask = (jl ? 1 : 0) | (jm ? 2 : 0) | (jh ? 4 : 0);
        }
        wave_sync();   // the filter output is dead: its storage becomes the FFT buffer
    }

    if (p.debug == 4) return;
    // the lane's FFT twiddles (their indices depend on the window mask only) are requested before the pre-rotation
    const at3::cpx* tw128 = T->tw128;   // tw128 | tw64 | tw16 are contiguous
    const int f_bsel = lane < 32 ? 2 : lane < 48 ? 0 : 1;
    const int f_j = lane < 32 ? lane : (lane - 32) & 15;
    const bool f_sh = (mask >> f_bsel) & 1;
    const int f_N = f_sh ? 16 : (f_bsel == 2 ? 128 : 64);
    const int f_r = f_j % (f_N >> 2);
    f2 f_tw[3][3];
    const f2 f_tw0 = at3::ld2(tw128);
    {
        const at3::cpx* tw = tw128 + (f_sh ? 192 : f_bsel == 2 ? 0 : 128);
        int m = f_N == 128 ? 2 : 1;
#pragma unroll
        for (int st = 0; st < 3; ++st) {
            const int mm = m < f_N ? m : 1;   // (a stage the transform does not have: any valid index)
            const int fstride = f_N / (4 * mm), k = f_r % mm;
#pragma unroll
            for (int q = 0; q < 3; ++q) f_tw[st][q] = at3::ld2(tw + (q + 1) * k * fstride);
            m <<= 2;
        }
    }
    // TAtrac1MDCT::Mdct (atrac1denc.cpp:70-102): TMDCT<N>::operator() pre-rotation (lib/mdct/mdct.h:51-87) straight into
    // the FFT's leaf order; the windowed input buffer of the reference is evaluated where it is read (mdct_in).
    // Round 0 is the low band's 64 points, round 1 the middle band's, rounds 2 and 3 the high band's 128.
#pragma unroll
    for (int round = 0; round < 4; ++round) {
        const int b = round < 2 ? round : 2;
        const int c = lane + (round == 3 ? 64 : 0);
        const bool sh = (mask >> b) & 1;
        const int B = b == 2 ? 256 : 128;
        const int N = sh ? 64 : 2 * B;
        const int n4 = N >> 2, n34 = 3 * n4, n54 = 5 * n4;
        const int k = sh ? c >> 4 : 0;
        const int pidx = sh ? (c & 15) : c;
        const int n = 2 * pidx;
        const float* src = (b == 0 ? s_low : b == 1 ? s_mid : s_up1) + 36;
        auto in = [&](int o) { return mdct_in(src, s_sine, B, sh, k, o); };
        float r0, i0;
        if (n < n4) {
            r0 = in(n34 - 1 - n) + in(n34 + n);
            i0 = in(n4 + n) - in(n4 - 1 - n);
        } else {
            r0 = in(n34 - 1 - n) - in(n - n4);
            i0 = in(n4 + n) + in(n54 - 1 - n);
        }
        const f2 csv = sh ? cs_short : cs_long[round];
        const float cc = csv.x, ss = csv.y;
        at3::cpx v;
        v.r = r0 * cc + i0 * ss;
        v.i = i0 * cc - r0 * ss;
        const int leaf = sh ? fft_leaf_pos<16>(pidx) : (b == 2 ? fft_leaf_pos<128>(pidx) : fft_leaf_pos<64>(pidx));
        s_f[(b == 0 ? 0 : b == 1 ? 64 : 128) + 16 * k + leaf] = v;
    }
    wave_sync();
    if (p.debug == 5) return;
    {
        // All three bands' transforms at once: a radix-4 stage has 32 butterflies in the high band (one 128-point or
        // eight 16-point transforms) and 16 each in the low and middle bands (one 64-point or four 16-point) - 64 lanes.
        // Lanes 0..31 own the high band, 32..47 the low, 48..63 the middle band; the 128-point transform's radix-2
        // leaves (64 butterflies) take all lanes first.
        if (!(mask & 4)) {
            at3::cpx* a = s_f + 128 + 2 * lane;
            f2 a0 = at3::ld2(a), a1 = at3::ld2(a + 1);
            at3::bfly2(a0, a1, f_tw0);
            at3::st2(a, a0);
            at3::st2(a + 1, a1);
        }
        wave_sync();
        at3::cpx* F = s_f + (f_bsel == 0 ? 0 : f_bsel == 1 ? 64 : 128);
        const int ft = f_j / (f_N >> 2);   // transform inside the band; f_r: butterfly inside the transform
        int m = f_N == 128 ? 2 : 1;
#pragma unroll
        for (int st = 0; st < 3; ++st) {
            if (m < f_N) {
                const int g = f_r / m, k = f_r % m;
                at3::cpx* B = F + ft * f_N + g * 4 * m + k;
                f2 x0 = at3::ld2(B), x1 = at3::ld2(B + m), x2 = at3::ld2(B + 2 * m), x3 = at3::ld2(B + 3 * m);
                at3::bfly4<false>(x0, x1, x2, x3, f_tw[st][0], f_tw[st][1], f_tw[st][2]);
                at3::st2(B, x0);
                at3::st2(B + m, x1);
                at3::st2(B + 2 * m, x2);
                at3::st2(B + 3 * m, x3);
            }
            m <<= 2;
            wave_sync();
        }
    }
    if (p.debug == 6) return;
    // post-rotation (mdct.h:89-101), the high band's short-window gain and the mirrored bands (atrac1denc.cpp:92-97)
#pragma unroll
    for (int round = 0; round < 4; ++round) {
        const int b = round < 2 ? round : 2;
        const int c = lane + (round == 3 ? 64 : 0);
        const bool sh = (mask >> b) & 1;
        const int n2 = sh ? 32 : (b == 2 ? 256 : 128);
        const int k = sh ? c >> 4 : 0;
        const int n = 2 * (sh ? (c & 15) : c);
        const at3::cpx v = s_f[(b == 0 ? 0 : b == 1 ? 64 : 128) + c];
        const f2 csv = sh ? cs_short : cs_long[round];
        const float cc = csv.x, ss = csv.y;
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
    wave_sync();

    if (p.debug == 7) return;
    // the spectrum leaves for HBM; the per-channel loudness (an ordered sum over its 512 lines) is k_at1_loud's
#pragma unroll
    for (int r = 0; r < 2; ++r)
        *reinterpret_cast<float4*>(p.specs + item * 512 + 4 * (lane + 64 * r)) = *reinterpret_cast<const float4*>(s_specs + 4 * (lane + 64 * r));
    if (lane == 0) p.mask[item] = mask;
    {
        // TScaler<TAtrac1Data>::Scale / ScaleFrame (atrac/atrac_scale.cpp:141-188), one lane per BFU: the scale factor and
        // the in-order energy sum; the divisions are spread over the whole wavefront below
        const int bfu = lane < kMaxBfus ? lane : 0;
        const bool sh = (mask >> bfu_band(bfu)) & 1;
        const int src0 = sh ? bf_short : bf_long;
        const float* in = s_specs + src0;
        float max_abs = 0.0f, e = 0.0f;
        {
            float xv[20];   // a BFU has 4 .. 20 lines: all of them requested before the ordered sum starts
#pragma unroll
            for (int i = 0; i < 20; ++i) xv[i] = i < bf_len ? in[i] : 0.0f;
#pragma unroll
            for (int i = 0; i < 20; ++i) {
                if (i < bf_len) {
                    const float a = fabsf(xv[i]);
                    if (a > max_abs) max_abs = a;
                    e += xv[i] * xv[i];
                }
            }
        }
        if (max_abs > 1.0f) max_abs = 1.0f;
        // smallest table entry >= max_abs (std::map::lower_bound in the reference): every lane takes part in the cross-lane
        // look-ups, lanes without a BFU search for 0
        int lo = 0, hi = 63;
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const int mid = (lo + hi) >> 1;
            const float sm = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(4 * mid, (int)__float_as_uint(my_scale)));
            if (sm < max_abs) lo = mid + 1;
            else hi = mid;
        }
        const float sfv = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(4 * lo, (int)__float_as_uint(my_scale)));
        if (lane < kMaxBfus) {
            s_sf[bfu] = sfv;
            s_srcoff[bfu] = src0 - bf_long;
            p.sfi[item * 64 + bfu] = (uint8_t)lo;
            p.energy[item * kMaxBfus + bfu] = e;
        }
    }
    wave_sync();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i0 = 4 * (lane + 64 * r);
        const uint32_t pb4 = pos_bfu4[r];
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pb = (int)((pb4 >> (8 * q)) & 0xffu);
            float x = s_specs[s_srcoff[pb] + i0 + q] / s_sf[pb];
            if (fabsf(x) >= 1.0f) x = (x > 0) ? 0.99999f : -0.99999f;
            v[q] = x;
        }
        *reinterpret_cast<float4*>(p.values + item * 512 + i0) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// Per-channel loudness of a sound unit (atrac1denc.cpp:235-240): l = sum over the 512 lines, IN ORDER, of specs[i]^2 *
// LoudnessCurve[i]. A 512-step chain whatever the hardware: in k_at1_front it was one lane of 256 holding the workgroup's
// slot for the length of the chain. Here a lane runs one unit's chain and a wavefront kAt1LoudUnits of them side by side,
// fed by the workgroup's three other wavefronts (coalesced 16-byte loads of the spectra k_at1_front has just written,
// products laid out per unit in LDS one chunk ahead of the chains) - the ATRAC3 k_loud_sum scheme.
constexpr int kAt1LoudUnits = 32;                       // units per workgroup
constexpr int kAt1LoudLines = 128;                      // lines per chunk
constexpr int kAt1LoudRow = kAt1LoudLines + 4;          // floats per unit and chunk: rows 4 banks apart, 16-byte aligned
constexpr int kAt1LoudChunks = 512 / kAt1LoudLines;
constexpr int kAt1LoudWords = kAt1LoudUnits * kAt1LoudLines / 4;   // 16-byte words of a chunk
constexpr int kAt1LoudPer = (kAt1LoudWords + 191) / 192;           // words per producer thread
static_assert(192 % (kAt1LoudLines / 4) == 0, "a producer thread keeps its place in the row from word to word");

struct LoudParams {
    const Tables* T;
    const float* specs;   // [units][512]
    float* loud_ch;       // [units]
    int32_t n_units;
};

__device__ __forceinline__ void at1_loud_request(const float* specs, int u0, int n_units, int u, int k, float4 (&xr)[kAt1LoudPer])
{
#pragma unroll
    for (int i = 0; i < kAt1LoudPer; ++i) {
        const int w = u + 192 * i, unit = u0 + w / (kAt1LoudLines / 4);
        xr[i] = (w < kAt1LoudWords && unit < n_units)
                    ? *reinterpret_cast<const float4*>(specs + (size_t)unit * 512 + kAt1LoudLines * k + 4 * (w % (kAt1LoudLines / 4)))
                    : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
}
__device__ __forceinline__ void at1_loud_produce(float* tile, const float* s_curve, int u, int k, const float4 (&xr)[kAt1LoudPer])
{
    const float4 cv = *reinterpret_cast<const float4*>(s_curve + kAt1LoudLines * k + 4 * (u % (kAt1LoudLines / 4)));
#pragma unroll
    for (int i = 0; i < kAt1LoudPer; ++i) {
        const int w = u + 192 * i;
        if (w < kAt1LoudWords) {
            const float4 x = xr[i];
            *reinterpret_cast<float4*>(tile + (w / (kAt1LoudLines / 4)) * kAt1LoudRow + 4 * (w % (kAt1LoudLines / 4))) =
                make_float4((x.x * x.x) * cv.x, (x.y * x.y) * cv.y, (x.z * x.z) * cv.z, (x.w * x.w) * cv.w);
        }
    }
}
__device__ __forceinline__ void at1_loud_consume(const float* row, float& l)
{
    const float4* r4 = reinterpret_cast<const float4*>(row);
#pragma unroll
    for (int q = 0; q < kAt1LoudLines / 4; q += 4) {
        const float4 a = r4[q], b = r4[q + 1], c = r4[q + 2], d = r4[q + 3];
        l += a.x; l += a.y; l += a.z; l += a.w;
        l += b.x; l += b.y; l += b.z; l += b.w;
        l += c.x; l += c.y; l += c.z; l += c.w;
        l += d.x; l += d.y; l += d.z; l += d.w;
    }
}

__global__ __launch_bounds__(256) void k_at1_loud(LoudParams p)
{
    __shared__ __attribute__((aligned(16))) float s_t[2][kAt1LoudUnits * kAt1LoudRow];   // [buffer][unit][line in chunk]
    __shared__ __attribute__((aligned(16))) float s_curve[512];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid < 128) *reinterpret_cast<float4*>(s_curve + 4 * tid) = *reinterpret_cast<const float4*>(p.T->loud + 4 * tid);
    const int u0 = blockIdx.x * kAt1LoudUnits;
    const int u = tid - 64;   // producer index 0..191
    const bool chain = wave == 0 && lane < kAt1LoudUnits;
    float4 xa[kAt1LoudPer], xb[kAt1LoudPer];
    if (wave > 0) {
        at1_loud_request(p.specs, u0, p.n_units, u, 0, xa);
        at1_loud_request(p.specs, u0, p.n_units, u, 1, xb);
    }
    __syncthreads();   // the curve is in LDS
    if (wave > 0) {
        at1_loud_produce(s_t[0], s_curve, u, 0, xa);
        at1_loud_request(p.specs, u0, p.n_units, u, 2, xa);
    }
    __syncthreads();
    float l = 0.0f;
    for (int k = 0; k < kAt1LoudChunks; k += 2) {   // chunk k + 1 comes from xb, chunk k + 2 from xa
        if (wave > 0) {
            at1_loud_produce(s_t[(k + 1) & 1], s_curve, u, k + 1, xb);
            if (k + 3 < kAt1LoudChunks) at1_loud_request(p.specs, u0, p.n_units, u, k + 3, xb);
        } else if (chain) {
            at1_loud_consume(s_t[k & 1] + lane * kAt1LoudRow, l);
        }
        __syncthreads();
        if (wave > 0) {
            if (k + 2 < kAt1LoudChunks) {
                at1_loud_produce(s_t[k & 1], s_curve, u, k + 2, xa);
                if (k + 4 < kAt1LoudChunks) at1_loud_request(p.specs, u0, p.n_units, u, k + 4, xa);
            }
        } else if (chain) {
            at1_loud_consume(s_t[(k + 1) & 1] + lane * kAt1LoudRow, l);
        }
        __syncthreads();
    }
    if (chain && u0 + lane < p.n_units) p.loud_ch[u0 + lane] = l;
}

struct ScanParams {
    const int32_t* mask;
    const float* loud_ch;
    float* loud_state;   // [S] carried between calls
    float* loud_track;   // [S][F] Loudness after the unit's update
    int32_t n_streams, n_frames, nch;
};

// TrackLoudness (atrac/atrac_psy_common.h:46-54) as atrac1denc.cpp:243-247 applies it. One wave per stream: every lane
// loads four sound units' masks and loudness sums (coalesced), then the recursion walks the units in order with the
// operands fetched by readlane into scalar registers - no memory access inside the dependent chain (a global load per
// step cost 53 us for 128 units, LDS 19 us) - and with everything that does not depend on the tracked value prepared by
// the lanes in parallel (13 -> 9 us). All lanes compute the same value; lane j keeps the result of "its" unit.
__global__ __launch_bounds__(64) void k_at1_loud_scan(ScanParams p)
{
    const int s = blockIdx.x, lane = threadIdx.x;
    float L = p.loud_state[s];
    for (int base = 0; base < p.n_frames; base += 256) {
        const int cnt = p.n_frames - base < 256 ? p.n_frames - base : 256;
        // what a step adds to 0.98 L - 0.01 (l0 + l1) or 0.02 l0, in double as the reference forms it - does not depend on L:
        // every lane prepares it for its four units, the chain itself is convert, multiply, add, convert per unit
        double add[4];
        float tr[4];
        unsigned long long upd[4];   // units that update the tracker (the others leave it alone)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = 64 * c + lane;
            add[c] = 0.0;
            tr[c] = 0.0f;
            int fl = 0;
            if (j < cnt) {
                const size_t it = ((size_t)s * p.n_frames + base + j) * p.nch;
                const int m0 = p.mask[it];
                const int m1 = p.nch == 2 ? p.mask[it + 1] : 1;
                const float l0 = p.loud_ch[it];
                const float l1 = p.nch == 2 ? p.loud_ch[it + 1] : 0.0f;
                fl = (m0 == 0 ? 1 : 0) | ((m0 == 0 && m1 == 0) ? 2 : 0);   // bit 0: channel 0 all-long, bit 1: both all-long
                add[c] = (fl & 2) ? 0.01 * (double)(l0 + l1) : 0.02 * (double)l0;
            }
            upd[c] = __ballot(fl != 0);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int lim = cnt - 64 * c < 64 ? cnt - 64 * c : 64;
            const long long abits = __double_as_longlong(add[c]);
            const int alo = (int)(uint32_t)abits, ahi = (int)(uint32_t)((unsigned long long)abits >> 32);
#pragma unroll 8
            for (int jj = 0; jj < lim; ++jj) {
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane(alo, jj), hi = (uint32_t)__builtin_amdgcn_readlane(ahi, jj);
                const double a = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
                if ((upd[c] >> jj) & 1ull) L = (float)(0.98 * (double)L + a);
                if (lane == jj) tr[c] = L;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (64 * c + lane < cnt) p.loud_track[(size_t)s * p.n_frames + base + 64 * c + lane] = tr[c];
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

// Sum over the 64 lanes, wave-uniform result: Hillis-Steele inside each 16-lane row (row_shr), then the row totals travel
// down with row_bcast:15 / row_bcast:31 so that lane 63 holds the total - seven DPP additions and one readlane, the
// whole cost of a bisection step's "bits used" apart from the per-lane expression.
__device__ __forceinline__ int wave_sum_i32(int v, int lane)
{
    (void)lane;
    v += AT3_DPP(v, 0x111, true);   // row_shr:1
    v += AT3_DPP(v, 0x112, true);   // row_shr:2
    v += AT3_DPP(v, 0x114, true);   // row_shr:4
    v += AT3_DPP(v, 0x118, true);   // row_shr:8   -> lane 15 of each row = row total
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
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

    // Everything the wavefront reads from global memory is requested here, in one batch, at clamped indices and without conditions
    // (both fixed-allocation rows: which one counts depends on the window mask that is still on its way); written where they are
    // used the mask, the loudness and the table rows were three round trips one after the other.
    const bool in_tab = lane < kMaxBfus;
    const int bl = in_tab ? lane : 0;
    const int mask = p.mask[item];
    const int sfi_raw = p.sfi[(size_t)item * 64 + lane];
    const float energy = p.energy[(size_t)item * kMaxBfus + bl];
    const float loud_raw = p.loud_track[item / p.nch];
    const float fix_s = T->fix_short[bl], fix_l = T->fix_long[bl];
    const float ath_v = T->ath_bfu[bl];
    const int spb_raw = c_spb[bl];
    // the lane's share of the mantissa sources for the packing step at the end: eight consecutive positions of the
    // BFU-ordered value array, their owning BFUs, and (lane = BFU) where each BFU's run starts. Fetched now, used last.
    const uint2 pos_bfu = *reinterpret_cast<const uint2*>(c_bfu_of_pos + 8 * lane);
    const float4 val_a = *reinterpret_cast<const float4*>(p.values + (size_t)item * 512 + 8 * lane);
    const float4 val_b = *reinterpret_cast<const float4*>(p.values + (size_t)item * 512 + 8 * lane + 4);
    const int run_raw = c_start_long[bl];
    __builtin_amdgcn_sched_barrier(0);
    const int band = bfu_band(bl);
    const bool sh = (mask >> band) & 1;
    const int sfi = in_tab ? sfi_raw : 0;
    const int spb = in_tab ? spb_raw : 0;
    const float loudness = loud_raw / 0.006f;
    const float fix = sh ? fix_s : fix_l;
    const bool gate = !sh && energy < ath_v * loudness;
    const float spread = 0.4f;
    const float base = spread * ((float)sfi / 3.2f) + (1.0f - spread) * fix;
    const int run_start = in_tab ? run_raw : 512;

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
    }
    {
        // mantissas, eight consecutive positions per lane (a per-BFU loop would run 20 rounds for the widest BFUs): word
        // length, run start and bit offset of the owning BFU come from that BFU's lane
        const int excl = incl - spb * bits;
        const float v[8] = {val_a.x, val_a.y, val_a.z, val_a.w, val_b.x, val_b.y, val_b.z, val_b.w};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int bfu = (int)(((t < 4 ? pos_bfu.x : pos_bfu.y) >> (8 * (t & 3))) & 0xffu);
            const int wl = __builtin_amdgcn_ds_bpermute(4 * bfu, bits);
            const int st = __builtin_amdgcn_ds_bpermute(4 * bfu, run_start);
            const int ex = __builtin_amdgcn_ds_bpermute(4 * bfu, excl);
            if (wl > 1) {
                const float multiple = (float)((1 << (wl - 1)) - 1);
                const int q = __float2int_rn(v[t] * multiple);
                put(16 + 10 * n + ex + (8 * lane + t - st) * wl, (uint32_t)q & ((1u << wl) - 1), wl);
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
