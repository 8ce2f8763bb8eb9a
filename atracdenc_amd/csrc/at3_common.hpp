// Shared device-side definitions of the ATRAC3 encode kernels (gfx950).
#pragma once
#include <cstddef>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "at3_pk.hpp"
#include "at3_tables.hpp"

namespace at3 {

// PCM history kept per stream between calls: two full blocks (MDCT overlap source + look-behind of the
// gain analysis) plus the 138-sample reach of the two-stage 48-tap QMF, rounded up.
constexpr int kHist = 2304;
constexpr int kMaxTonal = 24;

struct Curve {          // one band's gain curve (TAtrac3Data::SubbandInfo::TGainPoint list)
    uint8_t n;
    uint8_t level[7];
    uint8_t loc[7];
    uint8_t pad;
};
static_assert(sizeof(Curve) == 16, "Curve layout");

struct GainRec {        // per (stream, frame, channel, band<3): gain-analysis results
    float hfr;
    float target;       // CalcCurve's target for this frame (context independent)
    float cur_hpf;      // mean(gain)
    float ctx_level;    // context *before* this frame (resolved by the scan kernel)
    float ctx_target;
    float ctx_hpf;
    float pad[2];
    float gain[32];
    float lo[32];
    float hi[32];
};
static_assert(sizeof(GainRec) == 104 * 4, "GainRec layout");

struct BandState {      // carried per (stream, channel, band) between calls
    float last_level, last_target, last_hpf;
    float pad;
    Curve prev_curve;
};

struct TonalBlock {
    uint16_t pos;
    uint8_t bfu, len, sfi, pad[3];
    float values[7];
    uint8_t pad2[4];
};
static_assert(sizeof(TonalBlock) == 40, "TonalBlock layout");

struct PsyRec {         // per (stream, frame, channel)
    float loud_ch;
    int32_t n_tonal;
    uint8_t sfi[32];
    float energy[32];
    TonalBlock tonal[kMaxTonal];
    float flat[32];     // spectral flatness of BFUs 8..28 (the ones ExtractTonalComponents looks at; 0 elsewhere and with NoTonalComponents)
};
static_assert(sizeof(PsyRec) == 1256, "PsyRec layout (AT3HIP_TAP_PSY)");

// ---- integer constant tables (atrac/at3/atrac3.h:79-176, atrac3_bitstream.cpp:44-49) ----
__device__ static const uint16_t c_bfu_start[33] = {
    0,   8,   16,  24,  32,  40,  48,  56,  64,  80,  96,  112, 128, 144, 160, 176, 192,
    224, 256, 288, 320, 352, 384, 416, 448, 480, 512, 576, 640, 704, 768, 896, 1024};
__device__ static const uint8_t c_clc_len[8] = {0, 4, 3, 3, 4, 4, 5, 6};
__device__ static const float c_max_quant[8] = {0.0f, 1.5f, 2.5f, 3.5f, 4.5f, 7.5f, 15.5f, 31.5f};
__device__ static const uint8_t c_fixed_alloc[32] = {6, 6, 5, 4, 4, 4, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3,
                                                     3, 3, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 0, 0, 0};
// Huffman tables flattened: entry = code | bits << 8; table of selector s (1..7) starts at c_huff_off[s-1].
__device__ static const uint16_t c_huff_off[7] = {0, 9, 14, 0, 21, 36, 67};
#define HE(code, bits) (uint16_t)((code) | ((bits) << 8))
__device__ static const uint16_t c_huff[130] = {
    // table 1 (selectors 1 and 4): 9 entries
    HE(0x0, 1), HE(0x4, 3), HE(0x5, 3), HE(0xC, 4), HE(0xD, 4), HE(0x1C, 5), HE(0x1D, 5), HE(0x1E, 5), HE(0x1F, 5),
    // table 2: 5
    HE(0x0, 1), HE(0x4, 3), HE(0x5, 3), HE(0x6, 3), HE(0x7, 3),
    // table 3: 7
    HE(0x0, 1), HE(0x4, 3), HE(0x5, 3), HE(0xC, 4), HE(0xD, 4), HE(0xE, 4), HE(0xF, 4),
    // table 5: 15
    HE(0x0, 2), HE(0x2, 3), HE(0x3, 3), HE(0x8, 4), HE(0x9, 4), HE(0xA, 4), HE(0xB, 4), HE(0x1C, 5), HE(0x1D, 5),
    HE(0x3C, 6), HE(0x3D, 6), HE(0x3E, 6), HE(0x3F, 6), HE(0xC, 4), HE(0xD, 4),
    // table 6: 31
    HE(0x0, 3), HE(0x2, 4), HE(0x3, 4), HE(0x4, 4), HE(0x5, 4), HE(0x6, 4), HE(0x7, 4), HE(0x14, 5), HE(0x15, 5),
    HE(0x16, 5), HE(0x17, 5), HE(0x18, 5), HE(0x19, 5), HE(0x34, 6), HE(0x35, 6), HE(0x36, 6), HE(0x37, 6),
    HE(0x38, 6), HE(0x39, 6), HE(0x3A, 6), HE(0x3B, 6), HE(0x78, 7), HE(0x79, 7), HE(0x7A, 7), HE(0x7B, 7),
    HE(0x7C, 7), HE(0x7D, 7), HE(0x7E, 7), HE(0x7F, 7), HE(0x8, 4), HE(0x9, 4),
    // table 7: 63
    HE(0x0, 3), HE(0x8, 5), HE(0x9, 5), HE(0xA, 5), HE(0xB, 5), HE(0xC, 5), HE(0xD, 5), HE(0xE, 5), HE(0xF, 5),
    HE(0x10, 5), HE(0x11, 5), HE(0x24, 6), HE(0x25, 6), HE(0x26, 6), HE(0x27, 6), HE(0x28, 6), HE(0x29, 6),
    HE(0x2A, 6), HE(0x2B, 6), HE(0x2C, 6), HE(0x2D, 6), HE(0x2E, 6), HE(0x2F, 6), HE(0x30, 6), HE(0x31, 6),
    HE(0x32, 6), HE(0x33, 6), HE(0x68, 7), HE(0x69, 7), HE(0x6A, 7), HE(0x6B, 7), HE(0x6C, 7), HE(0x6D, 7),
    HE(0x6E, 7), HE(0x6F, 7), HE(0x70, 7), HE(0x71, 7), HE(0x72, 7), HE(0x73, 7), HE(0x74, 7), HE(0x75, 7),
    HE(0xEC, 8), HE(0xED, 8), HE(0xEE, 8), HE(0xEF, 8), HE(0xF0, 8), HE(0xF1, 8), HE(0xF2, 8), HE(0xF3, 8),
    HE(0xF4, 8), HE(0xF5, 8), HE(0xF6, 8), HE(0xF7, 8), HE(0xF8, 8), HE(0xF9, 8), HE(0xFA, 8), HE(0xFB, 8),
    HE(0xFC, 8), HE(0xFD, 8), HE(0xFE, 8), HE(0xFF, 8), HE(0x2, 4), HE(0x3, 4)};
#undef HE

// Arithmetic forms of the small tables: a dependent global-memory look-up inside a loop costs hundreds of
// cycles per iteration, these cost a few VALU ops.
__device__ __forceinline__ int bfu_start(int b)   // == c_bfu_start[b], 0 <= b <= 32
{
    return b < 8 ? 8 * b : b < 16 ? 64 + 16 * (b - 8) : b < 26 ? 192 + 32 * (b - 16) : b < 30 ? 512 + 64 * (b - 26) : 768 + 128 * (b - 30);
}
__device__ __forceinline__ int bfu_of_line(int i)  // BFU that holds spectral line i, 0 <= i < 1024
{
    return i < 64 ? i >> 3 : i < 192 ? 8 + ((i - 64) >> 4) : i < 512 ? 16 + ((i - 192) >> 5) : i < 768 ? 26 + ((i - 512) >> 6) : 30 + ((i - 768) >> 7);
}
__device__ __forceinline__ float max_quant(int wl)  // == c_max_quant[wl]
{
    return wl <= 4 ? (wl == 0 ? 0.0f : (float)wl + 0.5f) : wl == 5 ? 7.5f : wl == 6 ? 15.5f : 31.5f;
}
__device__ __forceinline__ int clc_len(int wl)      // == c_clc_len[wl]
{
    return wl == 0 ? 0 : wl == 1 ? 4 : wl <= 3 ? 3 : wl <= 5 ? 4 : wl == 6 ? 5 : 6;
}
__device__ __forceinline__ int huff_off(int sel)    // == c_huff_off[sel - 1]
{
    return sel == 2 ? 9 : sel == 3 ? 14 : sel == 5 ? 21 : sel == 6 ? 36 : sel == 7 ? 67 : 0;
}

__device__ __forceinline__ cpx cmul(cpx a, cpx b)
{
    cpx m;
    m.r = a.r * b.r - a.i * b.i;
    m.i = a.r * b.i + a.i * b.r;
    return m;
}

__device__ __forceinline__ f2 ld2(const cpx* q) { return *reinterpret_cast<const f2*>(q); }
__device__ __forceinline__ void st2(cpx* q, f2 v) { *reinterpret_cast<f2*>(q) = v; }

// kf_bfly4 (kiss_fft.c:42-90) on four points held in registers; w1..w3 = tw[k fstride], tw[2k fstride], tw[3k fstride]
template <bool INVERSE>
__device__ __forceinline__ void bfly4(f2& x0, f2& x1, f2& x2, f2& x3, f2 w1, f2 w2, f2 w3)
{
    const f2 s0 = pk_cmul(x1, w1), s1 = pk_cmul(x2, w2), s2 = pk_cmul(x3, w3);
    const f2 s5 = x0 - s1;
    f2 f0 = x0 + s1;
    const f2 s3 = s0 + s2, s4 = s0 - s2;
    x2 = f0 - s3;
    x0 = f0 + s3;
    if (INVERSE) {
        x1 = pk_add_ib(s5, s4);
        x3 = pk_sub_ib(s5, s4);
    } else {
        x1 = pk_sub_ib(s5, s4);
        x3 = pk_add_ib(s5, s4);
    }
}
// kf_bfly2 (kiss_fft.c:21-40)
__device__ __forceinline__ void bfly2(f2& x0, f2& x1, f2 w)
{
    const f2 t = pk_cmul(x1, w);
    x1 = x0 - t;
    x0 = x0 + t;
}

// log2f with the exact operation sequence of glibc 2.35's FMA build (see at3_tables.cpp); x > 0, finite.
// L: the 36 doubles {log2f_tab[16][2], log2f_poly[4]} (Tables layout), preferably staged in LDS by the caller.
struct Log2fTab {
    double tab[16][2];
    double poly[4];
};
static_assert(offsetof(Tables, log2f_poly) - offsetof(Tables, log2f_tab) == sizeof(double) * 32, "Tables keeps tab and poly together");
__device__ __forceinline__ float at3_log2f(const Log2fTab* L, float x)
{
    uint32_t ix = __float_as_uint(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {  // subnormal (other specials never reach here)
        ix = __float_as_uint(x * 0x1p23f);
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) % 16;
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int32_t)tmp >> 23;
    const double z = (double)__uint_as_float(iz);
    const double r = fma(z, L->tab[i][0], -1.0);
    const double y0 = L->tab[i][1] + (double)k;
    const double r2 = r * r;
    double y = fma(L->poly[1], r, L->poly[2]);
    y = fma(L->poly[0], r2, y);
    const double p = fma(L->poly[3], r, y0);
    y = fma(y, r2, p);
    return (float)y;
}
__device__ __forceinline__ float at3_log2f(const Tables* T, float x)   // tables straight from global memory
{
    return at3_log2f(reinterpret_cast<const Log2fTab*>(&T->log2f_tab[0][0]), x);
}

__device__ __forceinline__ void wave_sync_fwd()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- kissfft-order in-LDS FFT -------------------------------------------------------------------
// F holds `nfft` arrays of N complex points (stride NS) whose inputs were already stored in the
// decimation-in-time leaf order (see fft_leaf_pos). Recombination runs from the leaves up with the
// butterfly arithmetic of kf_bfly2 / kf_bfly4 (kiss_fft.c:21-90). Uniform call: every thread of the
// workgroup must enter; ends with a barrier.
template <int N>
__device__ __forceinline__ int fft_leaf_pos(int i)
{
    // input index i = q1 + 4 q2 + 16 q3 + ... (radix-4 digits, optional final radix-2 digit)
    // -> output slot o = q1*N/4 + q2*N/16 + ...
    int o = 0;
    int m = N;
    int rem = i;
    while (m >= 4 && (m % 4) == 0) {
        m >>= 2;
        o += (rem & 3) * m;
        rem >>= 2;
    }
    if (m == 2) o += (rem & 1);
    return o;
}

template <int N, bool INVERSE, bool LEAF_DONE = false, bool WAVE = false>
__device__ __forceinline__ void fft_lds(cpx* F, int NS, int nfft, const cpx* tw, int tid, int nthr)
{
    // WAVE: all participating threads are the lanes of ONE wavefront (nthr = 64, tid = lane): the stages are separated by
    // wave-level rendezvous instead of workgroup barriers
    auto stage_sync = [] {
        if (WAVE) wave_sync_fwd();
        else __syncthreads();
    };
    // number of radix-4 stages and whether a radix-2 leaf stage exists; LEAF_DONE: the caller already stored
    // the outputs of the radix-2 leaf butterflies (used when the leaf inputs are mostly exact zeros)
    int lg = 0;
    for (int t = N; t > 1; t >>= 1) ++lg;
    int m = 1;
    if ((lg & 1) && LEAF_DONE) m = 2;
    if ((lg & 1) && !LEAF_DONE) {
        // radix-2 leaves: m = 1, fstride = N/2, twiddle index 0
        const f2 w = ld2(tw);
        for (int j = tid; j < nfft * (N / 2); j += nthr) {
            const int f = j / (N / 2), p = j % (N / 2);
            cpx* a = F + f * NS + 2 * p;
            f2 a0 = ld2(a), a1 = ld2(a + 1);
            bfly2(a0, a1, w);
            st2(a, a0);
            st2(a + 1, a1);
        }
        stage_sync();
        m = 2;
    }
    for (; m < N; m <<= 2) {
        const int fstride = N / (4 * m);
        for (int j = tid; j < nfft * (N / 4); j += nthr) {
            const int f = j / (N / 4), r = j % (N / 4);
            const int g = r / m, k = r % m;
            cpx* B = F + f * NS + g * 4 * m + k;
            f2 x0 = ld2(B), x1 = ld2(B + m), x2 = ld2(B + 2 * m), x3 = ld2(B + 3 * m);
            bfly4<INVERSE>(x0, x1, x2, x3, ld2(tw + k * fstride), ld2(tw + 2 * k * fstride), ld2(tw + 3 * k * fstride));
            st2(B, x0);
            st2(B + m, x1);
            st2(B + 2 * m, x2);
            st2(B + 3 * m, x3);
        }
        stage_sync();
    }
}

// ---- wavefront (64 lanes) cross-lane helpers on DPP / readlane; call from wave-uniform code only ----
#define AT3_DPP(v, ctrl, bc) __builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), 0xf, 0xf, (bc))

// Sum over the 16-lane row that contains the lane (rotate-and-add: every lane of the row gets the total).
__device__ __forceinline__ uint32_t row_allreduce_add(uint32_t v)
{
    v += (uint32_t)AT3_DPP(v, 0x128, false);  // row_ror:8
    v += (uint32_t)AT3_DPP(v, 0x124, false);  // row_ror:4
    v += (uint32_t)AT3_DPP(v, 0x122, false);  // row_ror:2
    v += (uint32_t)AT3_DPP(v, 0x121, false);  // row_ror:1
    return v;
}

// Inclusive prefix sum over the 64 lanes (Hillis-Steele inside each row with row_shr + bound_ctrl,
// then the three row totals are added through readlane).
__device__ __forceinline__ int wave_inclusive_scan(int v, int lane)
{
    v += AT3_DPP(v, 0x111, true);  // row_shr:1
    v += AT3_DPP(v, 0x112, true);  // row_shr:2
    v += AT3_DPP(v, 0x114, true);  // row_shr:4
    v += AT3_DPP(v, 0x118, true);  // row_shr:8
    const int r0 = __builtin_amdgcn_readlane(v, 15);
    const int r1 = __builtin_amdgcn_readlane(v, 31);
    const int r2 = __builtin_amdgcn_readlane(v, 47);
    if (lane >= 16) v += r0;
    if (lane >= 32) v += r1;
    if (lane >= 48) v += r2;
    return v;
}

// Wave-level rendezvous for data exchanged through LDS between lanes of ONE wavefront: the LDS pipeline
// executes a wave's DS instructions in issue order, so only compiler reordering has to be fenced.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// GainLevel[l] = 2^(4 - l) (atrac3.h:192-194): exact powers of two, formed from the exponent field.
__device__ __forceinline__ float gain_level_of(int l) { return __uint_as_float((uint32_t)(127 + 4 - l) << 23); }

// TGainProcessor::Modulate (gain_processor.h:93-112) / BuildSampleDivisors (atrac3denc.cpp:154-173).
// Returns 1.0f for samples the curve does not touch (x / 1.0f == x, so dividing is a no-op there).
// gain_interp: the 31-entry GainInterpolation table, preferably staged in LDS by the caller.
__device__ __forceinline__ float curve_divisor(const float* gain_interp, const Curve& c, int i)
{
    int pos = 0;
    for (int p = 0; p < c.n; ++p) {
        const int lastPos = (int)c.loc[p] << 3;
        float level = gain_level_of(c.level[p]);
        if (i >= pos && i < lastPos) return level;
        if (lastPos > pos) pos = lastPos;
        if (pos < lastPos + 8) {
            if (i >= pos && i < lastPos + 8) {
                const int incPos = ((p + 1) < c.n ? (int)c.level[p + 1] : 4) - (int)c.level[p] + 15;
                const float inc = gain_interp[incPos];
                for (int q = pos; q < i; ++q) level *= inc;
                return level;
            }
            pos = lastPos + 8;
        }
    }
    return 1.0f;
}

// atrac3denc.cpp:143-152
__device__ __forceinline__ float safe_energy_scale(float orig, float mod)
{
    const float eps = 1.0e-20f;
    if (orig <= eps || mod <= eps || !isfinite(orig) || !isfinite(mod)) return 1.0f;
    const float scale = orig / mod;
    return (isfinite(scale) && scale > 0.0f) ? scale : 1.0f;
}

}  // namespace at3
