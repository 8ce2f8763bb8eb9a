// Gain-control kernels: spectral upsampler + AnalyzeGain, context scan, CalcCurve + point-0 logic.
//
// Reference path replaced (paths relative to the reference's src/):
//   transient_spectral_upsampler.cpp:77-180  TSpectralUpsampler::Process (Planck window, rFFT-512,
//                                            high-frequency ratio, raised-cosine HPF, x8 zero-padding, irFFT-4096)
//   tools/kiss_fftr.c:61-153                 kiss_fftr / kiss_fftri packing around the complex FFTs
//   transient_detector.cpp:33-40, 95-136     calculateRMS, AnalyzeGain (+ micro-chunk quartiles)
//   transient_detector.cpp:141-482           RelationToIdx, MedianFilter, FindPlateau, BoundaryTransientScore, CalcCurve
//   atrac3denc.cpp:259-297, 299-579          CalcCurveEarlyMismatchScore, CreateSubbandInfo
//
// Band 3 never yields a curve (atrac3denc.cpp:444-450) and its context feeds nothing else, so only
// bands 0..2 are analysed. The look-ahead "nextLevel" is computed by the reference but never read inside
// CalcCurve, so it is not computed here.
#pragma once
#include "at3_common.hpp"

namespace at3 {

struct GainParams {
    const float* sub;     // [S][2][4][(n_blocks+2)*256] raw L/R subbands, block b at (b+2)*256
    GainRec* rec;         // [S][n_blocks][2][3] by frame index
    cpx* bins;            // [items][kGainBins] rfft-512 bins 38 .. 256 of every item, k_gain_spec -> k_gain_analysis
    BandState* state;     // [S][2][4]
    Curve* curves;        // [S][n_blocks][2][4]
    int n_blocks;
    int f0;
    int js;
    int n_streams;
    int debug;   // profiling aid (env AT3HIP_DEBUG_GAIN): k_gain_curve returns early at stage N
    int literal; // AT3HIP_OPT_LITERAL_FORMS: k_gain_spec's energy sums as the reference's two 257-term chains for every item
    unsigned long long* clk;   // profiling builds: 256 rows of 12 per-phase cycle counters of k_gain_analysis1 (tools/gain_phase_cycles.sh)
};

constexpr int kLowCutBin = 38;  // ceil(800 * 512 / 11025)

// FindPlateau + target selection of CalcCurve (transient_detector.cpp:178-238, 284-297), in[32].
__device__ inline void median3_32(const float* in, float* out)
{
    for (int i = 0; i < 32; ++i) {
        const int lo = i > 0 ? i - 1 : 0;
        const int hi = i < 31 ? i + 1 : 31;
        if (hi - lo == 1) {
            // two elements: sorted ascending, element [1] is the larger
            out[i] = fmaxf(in[lo], in[hi]);
        } else {
            const float a = in[lo], b = in[lo + 1], c = in[hi];
            out[i] = fmaxf(fminf(a, b), fminf(fmaxf(a, b), c));
        }
    }
}

__device__ inline float curve_target(const float* in, const float* filtered)
{
    float maxRaw = 0.0f;
    for (int i = 0; i < 32; ++i) maxRaw = fmaxf(maxRaw, in[i]);
    float bestLevel = 0.0f;
    int bestEnd = -1;
    for (int j = 0; j + 3 <= 32; ++j) {
        const float minVal = fminf(fminf(filtered[j], filtered[j + 1]), filtered[j + 2]);
        if (minVal > bestLevel) {
            bestLevel = minVal;
            bestEnd = j + 2;
        }
    }
    float plateau = 0.0f;
    bool release = false;
    if (!(bestLevel < 1e-6f)) {
        plateau = bestLevel;
        while (bestEnd + 1 < 32 && filtered[bestEnd + 1] >= bestLevel) ++bestEnd;
        if (bestEnd < 31) {
            if (in[31] < bestLevel * 0.1f) {
                release = true;
            } else {
                bool anyHigh = false;
                for (int i = bestEnd + 1; i < 32; ++i)
                    if (in[i] >= bestLevel * 0.7f) {
                        anyHigh = true;
                        break;
                    }
                release = !anyHigh && (in[31] < bestLevel * 0.5f);
            }
        }
    }
    const bool usePlateau = plateau > 1e-6f && !release && plateau >= maxRaw * 0.4f;
    return usePlateau ? plateau : in[31];
}

// ---- 2048-point inverse kissfft core of the irfft-4096, one wavefront, points in LDS -----------------------------
// Logical slot i lives at i + i/32: one pad entry per 32 keeps every access pattern below free of bank conflicts.
// The five radix-4 passes of kissfft (m = 2, 8, 32, 128, 512; the radix-2 leaves are stored by the caller) run as
// register-resident pairs: a work unit loads 16 points, applies the four butterflies of pass m and the four of
// pass 4m that combine exactly those points, and stores them - same butterflies, same operands, half the LDS traffic.
__device__ __forceinline__ int irfft_pad(int i) { return i + (i >> 5); }

// kf_bfly4 (inverse) whose third input is an exact zero and - with Z1 - whose second one is too: the products and sums a
// zero takes part in are dropped (x + 0 w = x exactly; a zero result keeps its magnitude, only its sign may differ, and no
// sign of a zero survives the squares the transform's output ends in).
template <bool Z1>
__device__ __forceinline__ void bfly4_inv_sparse(f2& x0, f2& x1, f2& x2, f2& x3, f2 w1, f2 w3)
{
    const f2 s2 = pk_cmul(x3, w3);
    const f2 f0 = x0;
    if (Z1) {
        x2 = f0 - s2;
        x0 = f0 + s2;
        x1 = pk_sub_ib(f0, s2);   // s5 + i s4 with s4 = -s2
        x3 = pk_add_ib(f0, s2);
    } else {
        const f2 s0 = pk_cmul(x1, w1);
        const f2 s3 = s0 + s2, s4 = s0 - s2;
        x2 = f0 - s3;
        x0 = f0 + s3;
        x1 = pk_add_ib(f0, s4);
        x3 = pk_sub_ib(f0, s4);
    }
}

template <int KAPPA>   // passes m = 2 and m = 8: unit = (32-block G = lane, parity KAPPA), points 32 G + KAPPA + 2 i
__device__ __forceinline__ void irfft_pass_2_8(cpx* F, const cpx* tw, int lane)
{
    cpx* B = F + 33 * lane + KAPPA;
    f2 x[16];
    // Slot 32 G + KAPPA + 2 i is the leaf of input index r + 256 (i % 4) + 1024 KAPPA with r = the digit reversal of G plus
    // 64 (i / 4). The caller stores leaves only for inputs 38..256 and 1792..2010 (and their radix-2 partners), i.e. for
    // r >= 38 at i % 4 == 0, r == 0 at i % 4 == 1 and r <= 218 at i % 4 == 3: every other slot is an exact zero that is
    // neither stored nor read (the buffer is not cleared between items).
    const int r0 = (lane >> 4) + 4 * ((lane >> 2) & 3) + 16 * (lane & 3);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + 64 * (i >> 2);
        const bool nz = (i & 3) == 0 ? r >= 38 : (i & 3) == 1 ? r == 0 : (i & 3) == 3 ? r <= 218 : false;
        x[i] = f2{0.0f, 0.0f};
        if ((i & 3) != 2 && nz) x[i] = ld2(B + 2 * i);
    }
    const f2 a1 = ld2(tw + 256 * KAPPA), a3 = ld2(tw + 768 * KAPPA);   // k = KAPPA, fstride 256 (tw[512 KAPPA] only ever meets a zero)
    // pass m = 2: the third input of every butterfly is a zero leaf, the second one too except for input 256 (g = 0, lane 0)
    bfly4_inv_sparse<false>(x[0], x[1], x[2], x[3], a1, a3);
#pragma unroll
    for (int g = 1; g < 4; ++g) bfly4_inv_sparse<true>(x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3], a1, a3);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = KAPPA + 2 * j;   // fstride 64
        bfly4<true>(x[j], x[j + 4], x[j + 8], x[j + 12], ld2(tw + 64 * k), ld2(tw + 128 * k), ld2(tw + 192 * k));
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) st2(B + 2 * i, x[i]);
}

// passes m = 32 and m = 128: unit v = (512-block G2 = v / 32, k = v % 32), points 512 G2 + k + 32 i.
// The 15 twiddles of a unit depend on v only; they are fetched long before the pass runs (a dependent global load
// costs a microsecond, the passes themselves a few hundred cycles).
struct Tw32_128 {
    f2 a[3];
    f2 w[4][3];
};
__device__ __forceinline__ Tw32_128 irfft_tw_32_128(const cpx (*g)[128], int v)   // g = Tables::gain_tw
{
    Tw32_128 t;
    t.a[0] = ld2(&g[0][v]);   // tw[16 k], tw[32 k], tw[48 k] with k = v % 32 (fstride 16)
    t.a[1] = ld2(&g[1][v]);
    t.a[2] = ld2(&g[2][v]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {   // tw[4 kk], tw[8 kk], tw[12 kk] with kk = k + 32 j (fstride 4)
        t.w[j][0] = ld2(&g[3 + 3 * j][v]);
        t.w[j][1] = ld2(&g[4 + 3 * j][v]);
        t.w[j][2] = ld2(&g[5 + 3 * j][v]);
    }
    return t;
}
__device__ __forceinline__ void irfft_pass_32_128(cpx* F, const Tw32_128& t, int v)
{
    const int k = v & 31, G2 = v >> 5;
    cpx* B = F + 528 * G2 + k;
    f2 x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = ld2(B + 33 * i);
#pragma unroll
    for (int g = 0; g < 4; ++g) bfly4<true>(x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3], t.a[0], t.a[1], t.a[2]);
#pragma unroll
    for (int j = 0; j < 4; ++j) bfly4<true>(x[j], x[j + 4], x[j + 8], x[j + 12], t.w[j][0], t.w[j][1], t.w[j][2]);
#pragma unroll
    for (int i = 0; i < 16; ++i) st2(B + 33 * i, x[i]);
}

// pass m = 512: butterfly k = tid + NT u on points k + 512 q (NT work-items share the pass); twiddles prefetched
template <int NT>
struct Tw512 {
    f2 w[512 / NT][3];
};
template <int NT>
__device__ __forceinline__ Tw512<NT> irfft_tw_512(const cpx (*g)[128], int tid)   // g = Tables::gain_tw; NT == 128
{
    static_assert(NT == 128, "Tables::gain_tw is laid out for 128 work-items");
    Tw512<NT> t;
#pragma unroll
    for (int u = 0; u < 512 / NT; ++u) {   // tw[k], tw[2 k], tw[3 k] with k = tid + NT u
        t.w[u][0] = ld2(&g[15 + 3 * u][tid]);
        t.w[u][1] = ld2(&g[16 + 3 * u][tid]);
        t.w[u][2] = ld2(&g[17 + 3 * u][tid]);
    }
    return t;
}
template <int NT>
__device__ __forceinline__ void irfft_pass_512(cpx* F, const Tw512<NT>& t, int tid)
{
#pragma unroll
    for (int u0 = 0; u0 < 512 / NT; u0 += 4) {
        f2 x[4][4];
        cpx* B[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            B[u] = F + irfft_pad(tid + NT * (u0 + u));
#pragma unroll
            for (int q = 0; q < 4; ++q) x[u][q] = ld2(B[u] + 528 * q);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) bfly4<true>(x[u][0], x[u][1], x[u][2], x[u][3], t.w[u0 + u][0], t.w[u0 + u][1], t.w[u0 + u][2]);
        // only the middle half of the upsampled frame is analysed (samples 1024..3071 = outputs k + 512, k + 1024)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            st2(B[u] + 528, x[u][1]);
            st2(B[u] + 1056, x[u][2]);
        }
    }
}

// The analysis of one (stream, frame, channel, band < 3) item is two kernels:
//   k_gain_spec      Planck window, rFFT-512, the two ordered f64 energy sums and the high-frequency ratio; the 219 bins
//                    that survive the high-pass go to HBM. The 256-point complex core of an item lives in the SIXTEEN LANES
//                    of one DPP row with sixteen points per lane (below): four items per wavefront.
//   k_gain_analysis  x8 zero-padded irFFT-4096 (2048-point complex core, 17 KB of LDS), AnalyzeGain, plateau target - only for
//                    items whose ratio reaches 5 % (atrac3denc.cpp:319-327), two wavefronts per item.
// (One kernel did both at first: its cheap first half then ran at the 8-items-per-CU occupancy of the LDS-hungry second
// half, with its sequential chains on two lanes of 128.)
// Inclusive scan of an f64 over the 16-lane row (row_shr with zero fill: lane 15 ends up with the row's sum)
template <int CTRL>
__device__ __forceinline__ double dpp_f64_zero_fill(double v)
{
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)AT3_DPP((uint32_t)b, CTRL, true), hi = (uint32_t)AT3_DPP((uint32_t)(b >> 32), CTRL, true);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
__device__ __forceinline__ double row_scan_add_f64(double v)
{
    v += dpp_f64_zero_fill<0x111>(v);   // row_shr:1
    v += dpp_f64_zero_fill<0x112>(v);   // row_shr:2
    v += dpp_f64_zero_fill<0x114>(v);   // row_shr:4
    v += dpp_f64_zero_fill<0x118>(v);   // row_shr:8
    return v;
}
constexpr int kGainBins = 220;   // bins 38 .. 256 (219), padded to an even count

// ---- rfft-512 of one item in a 16-lane row --------------------------------------------------------------------------
// kissfft factors 256 as 4 x 4 x 4 x 4 (kiss_fft.c:357-363): input i = q1 + 4 q2 + 16 q3 + 64 q4 enters the recombination at
// position 64 q1 + 16 q2 + 4 q3 + q4, pass m (= 1, 4, 16, 64) combines positions m apart with twiddles tw[q k 64 / m].
// Lane L = q1 + 4 q2 of the row loads its sixteen inputs i = L + 16 t straight from the subbands (for a given t the row reads
// 128 contiguous bytes) and runs passes m = 1 and m = 4 on them in registers - the sixteen positions 64 q1 + 16 q2 + r are
// exactly its own. One transposition through LDS (sixteen 8-byte stores, eight 16-byte loads per lane, conflict free) hands
// lane K position r = K of every sixteen-block; passes m = 16 (butterfly k = K) and m = 64 (k = K + 16 j) are again
// in-lane and leave F[K + 16 jj] in it. kiss_fftr's post-processing (tools/kiss_fftr.c:61-100) pairs F[k] with F[256 - k],
// which sits in lane (16 - K) % 16: sixteen cross-lane reads (ds_bpermute: the LDS crossbar, no storage). The first
// version walked all four passes through LDS with one butterfly per lane and pass: 4.3 k lane accesses per item, now 1.4 k.
constexpr int kSpecRowStride = 18;                    // 8-byte slots per transposed position: 144 bytes, 16-byte aligned rows
constexpr int kSpecItemBytes = 2432;                  // per item: the transposition (16 x 18 x 8 = 2304 B), later 300 f64 energies (2400 B)
struct SpecItemLds {
    union {
        cpx x[16 * kSpecRowStride];
        double e[304];
    };
};
static_assert(sizeof(SpecItemLds) == kSpecItemBytes, "SpecItemLds layout");

__global__ __launch_bounds__(64) void k_gain_spec(GainParams p, const Tables* T, int n_items)
{
    __shared__ __attribute__((aligned(16))) SpecItemLds s_item[4];
    const int lane = threadIdx.x, row = lane >> 4, L = lane & 15;
    SpecItemLds& S = s_item[row];
    const int nfr = p.n_blocks - p.f0;
    // The four rows of a wavefront take four CONSECUTIVE frames of one (stream, channel, band): an item's 512 samples are its frame's
    // block and the next one, so row r's second half is row r + 1's first - five blocks per wavefront instead of eight come from
    // memory (items in index order put the two readers of a block in different workgroups, mostly on different XCDs' L2s).
    // workgroup = ((stream * quads + quad) * 2 + channel) * 3 + band, quads = ceil(frames / 4); a stream's last quad may be ragged
    const int quads = (nfr + 3) >> 2;
    int wg = blockIdx.x;
    const int band = wg % 3; wg /= 3;
    const int ch = wg % 2; wg /= 2;
    const int fr = 4 * (wg % quads) + row;
    const int s = wg / quads;
    const bool valid = fr < nfr;
    const int f = p.f0 + (valid ? fr : nfr - 1);   // (a row past the end repeats the last frame: the wavefront stays uniform, nothing is stored)
    const int item = (((s * nfr) + (f - p.f0)) * 2 + ch) * 3 + band;
    (void)n_items;
    const int cb = f - 1;  // current block
    const size_t sublen = (size_t)(p.n_blocks + 2) * 256;
    const float* sb0 = p.sub + ((size_t)s * 8 + 0 * 4 + band) * sublen + (size_t)(cb + 2) * 256 - 128;
    const float* sb1 = p.sub + ((size_t)s * 8 + 1 * 4 + band) * sublen + (size_t)(cb + 2) * 256 - 128;
    GainRec* rec = p.rec + (((size_t)s * p.n_blocks + f) * 2 + ch) * 3 + band;
    cpx* bins = p.bins + (size_t)item * kGainBins;

    // 1. window and pack: complex input i = L + 16 t is samples (2 i, 2 i + 1); a[4 q3 + q4] = input t = q3 + 4 q4
    // Table values are requested a stage AHEAD of their use, in batches (a request made where the value is wanted costs the
    // wavefront an L2 round trip right there - three of them in a ten-microsecond life, with three wavefronts per SIMD to cover):
    // the window with the samples, the lane's fifteen twiddles while passes m = 1 and 4 run.
    f2 a[16];
    {
        f2 raw[16], rb[16], win[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int i = L + 16 * t;
            raw[t] = *reinterpret_cast<const f2*>((p.js || ch == 0 ? sb0 : sb1) + 2 * i);
            if (p.js) rb[t] = *reinterpret_cast<const f2*>(sb1 + 2 * i);
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) win[t] = ld2(&T->spec16_win[t][L]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            f2 v = raw[t];
            if (p.js) v = (ch == 0) ? (v + rb[t]) * mk2(0.5f, 0.5f) : (v - rb[t]) * mk2(0.5f, 0.5f);
            a[4 * (t & 3) + (t >> 2)] = v * win[t];
        }
    }
    f2 tw_l[15];
#pragma unroll
    for (int q = 0; q < 15; ++q) tw_l[q] = ld2(&T->spec16_tw[q][L]);
    __builtin_amdgcn_sched_barrier(0);
#ifdef AT3HIP_DEBUG_KNOBS
    if (p.debug == 21) return;
#endif
    // passes m = 1 (all twiddles tw[0]) and m = 4 (butterfly k: tw[16 k], tw[32 k], tw[48 k], the same in every lane)
    {
        const f2 w0 = ld2(&T->tw256[0]);
#pragma unroll
        for (int q3 = 0; q3 < 4; ++q3) bfly4<false>(a[4 * q3], a[4 * q3 + 1], a[4 * q3 + 2], a[4 * q3 + 3], w0, w0, w0);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            bfly4<false>(a[k], a[k + 4], a[k + 8], a[k + 12], ld2(&T->tw256[16 * k]), ld2(&T->tw256[32 * k]), ld2(&T->tw256[48 * k]));
    }
    // transposition: lane K takes position r = K of every lane's block; b[l] = position 64 q1 + 16 q2 + K of lane l = q1 + 4 q2
    f2 b[16];
    {
#pragma unroll
        for (int r = 0; r < 16; ++r) st2(&S.x[r * kSpecRowStride + L], a[r]);
        wave_sync();
        const float4* src = reinterpret_cast<const float4*>(&S.x[L * kSpecRowStride]);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const float4 v = src[m];
            b[2 * m] = mk2(v.x, v.y);
            b[2 * m + 1] = mk2(v.z, v.w);
        }
        wave_sync();   // (the item's storage is reused for the energies)
    }
    // passes m = 16 (butterfly k = L over q2, for every q1) and m = 64 (butterfly k = L + 16 j over q1)
    {
        const f2 u1 = tw_l[0], u2 = tw_l[1], u3 = tw_l[2];
#pragma unroll
        for (int q1 = 0; q1 < 4; ++q1) bfly4<false>(b[q1], b[q1 + 4], b[q1 + 8], b[q1 + 12], u1, u2, u3);
#pragma unroll
        for (int j = 0; j < 4; ++j) bfly4<false>(b[4 * j], b[4 * j + 1], b[4 * j + 2], b[4 * j + 3], tw_l[3 + 3 * j], tw_l[4 + 3 * j], tw_l[5 + 3 * j]);
    }
    // now b[q + 4 j] = F[L + 16 (j + 4 q)]: F[L + 16 jj] = b[(jj >> 2) + 4 (jj & 3)]
#ifdef AT3HIP_DEBUG_KNOBS
    if (p.debug == 22) return;
#endif
    // 2. kiss_fftr post-processing -> 257 bins: this lane's k = L + 16 jj, jj = 0..7, each with its partner 256 - k, whose F
    // is element 15 - jj of lane (16 - L) % 16 - for L = 0 element 16 - jj of the lane itself
    f2 fa[8], fb[8];   // freq[k], freq[256 - k]
    f2 f128 = mk2(0.0f, 0.0f);
    {
        const int partner = 4 * ((lane & 48) | ((16 - L) & 15));
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int je = 15 - jj;                                  // partner's element (L != 0)
            const f2 pe = b[(je >> 2) + 4 * (je & 3)];
            f2 other;
            other.x = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(partner, (int)__float_as_uint(pe.x)));
            other.y = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(partner, (int)__float_as_uint(pe.y)));
            if (jj > 0) {
                const int j0 = 16 - jj;                              // L == 0: own element 16 - jj
                const f2 own = b[(j0 >> 2) + 4 * (j0 & 3)];
                if (L == 0) other = own;
            }
            const f2 fpk = b[(jj >> 2) + 4 * (jj & 3)];
            const cpx stw = T->spec16_stw[jj][L];
            const float fnr = other.x, fni = -other.y;               // fpnk = conj(F[256 - k])
            const float f1r = fpk.x + fnr, f1i = fpk.y + fni;
            const float f2r = fpk.x - fnr, f2i = fpk.y - fni;
            const float twr = f2r * stw.r - f2i * stw.i, twi = f2r * stw.i + f2i * stw.r;
            fa[jj] = mk2((f1r + twr) * 0.5f, (f1i + twi) * 0.5f);
            fb[jj] = mk2((f1r - twr) * 0.5f, (twi - f1i) * 0.5f);
        }
        // k = 0 (lane 0, jj = 0): freq[0] = (F0.r + F0.i, 0), freq[256] = (F0.r - F0.i, 0)
        if (L == 0) {
            const f2 f0 = b[0];
            fa[0] = mk2(f0.x + f0.y, 0.0f);
            fb[0] = mk2(f0.x - f0.y, 0.0f);
        }
        // k = 128 (lane 0, element 8): paired with itself; the reference's second store (freq[256 - k]) is the one that stays
        {
            const f2 fpk = b[2];   // jj = 8 -> (8 >> 2) + 4 (8 & 3) = 2
            const cpx stw = T->spec16_stw[8][L];
            const float fnr = fpk.x, fni = -fpk.y;
            const float f1r = fpk.x + fnr, f1i = fpk.y + fni;
            const float f2r = fpk.x - fnr, f2i = fpk.y - fni;
            const float twr = f2r * stw.r - f2i * stw.i, twi = f2r * stw.i + f2i * stw.r;
            f128 = mk2((f1r - twr) * 0.5f, (twi - f1i) * 0.5f);
        }
    }
    // the bins that survive the high-pass (38 .. 256) go to HBM for k_gain_analysis; whether it will want them is known
    // only after the energy sums, and the stores cost less than waiting for that
    if (valid) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int k = L + 16 * jj;
            if (k >= kLowCutBin) st2(&bins[k - kLowCutBin], fa[jj]);
            st2(&bins[256 - k - kLowCutBin], fb[jj]);
        }
        if (L == 0) st2(&bins[128 - kLowCutBin], f128);
    }
#ifdef AT3HIP_DEBUG_KNOBS
    if (p.debug == 23) return;
#endif
    // highFreqRatio (transient_spectral_upsampler.cpp:99-118): two ordered f64 sums over the 257 bin energies, the second
    // one weighted with the squared high-pass response (0 below bin 38, 1 from bin 40 on), and the f32 of their quotient.
    //
    // SHORT FORM. The reference adds each sum up as one chain of 257 dependent f64 additions; eight lanes of the wavefront doing
    // that were a fifth of this kernel's life. The terms are non-negative, so ANY order of adding n <= 257 of them lands within
    // gamma = 256 * 2^-53 (relative) of their exact sum; the quotient of two such sums is then within 4 gamma + 2 * 2^-53 =
    // 1.15e-13 of the chain's quotient. Every lane adds its 17 energies, a 16-lane row scan adds the lanes, and the f32 of the
    // quotient is taken as it stands whenever the quotient is further than 4e-13 (relative) from both rounding boundaries of that
    // f32 - the chain's quotient then rounds to the same f32, bit for bit. Otherwise (about one item in 10^5), for quotients
    // below the normal f32 range, and under AT3HIP_OPT_LITERAL_FORMS, the wavefront walks the chains.
    bool fast_done = false;
    {
        const double h1 = (double)T->hpf_w[1], h2 = (double)T->hpf_w[2];
        double et = 0.0, ef = 0.0;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int k = L + 16 * jj;   // <= 127; its partner 256 - k >= 129 passes the filter unweighted
            const double ea = (double)fa[jj].x * fa[jj].x + (double)fa[jj].y * fa[jj].y;
            const double eb = (double)fb[jj].x * fb[jj].x + (double)fb[jj].y * fb[jj].y;
            et += ea;
            et += eb;
            const double wa = (k < kLowCutBin) ? 0.0 : (k == kLowCutBin) ? ea * h1 * h1 : (k == kLowCutBin + 1) ? ea * h2 * h2 : ea;
            ef += wa;
            ef += eb;
        }
        if (L == 0) {
            const double e128 = (double)f128.x * f128.x + (double)f128.y * f128.y;
            et += e128;
            ef += e128;
        }
        et = row_scan_add_f64(et);
        ef = row_scan_add_f64(ef);   // lane 15 of the row: the item's two sums
        bool need = false;
        float hfr = 0.0f;
        if (L == 15) {
            if (et > 0.0 && ef > 0.0) {
                const double r = ef / et;
                hfr = (float)r;
                if (!(hfr >= 1e-30f)) {
                    need = true;
                } else {
                    const uint32_t hb = __float_as_uint(hfr);
                    const double below = ((double)__uint_as_float(hb - 1u) + (double)hfr) * 0.5, above = ((double)hfr + (double)__uint_as_float(hb + 1u)) * 0.5;
                    const double margin = r * 4e-13;
                    need = !(r - below > margin && above - r > margin);
                }
            }   // (no energy at all, or none behind the filter: every term of that chain is an exact zero too)
        }
        if (p.literal || __ballot(need && valid) != 0ull) {
            fast_done = false;
        } else {
            if (valid && L == 15) rec->hfr = hfr;
            fast_done = true;
        }
    }
    if (fast_done) return;
    // The chains. The energies are parked (as f64) in the item's storage; lanes 0..7 of the wavefront add them up - item
    // lane / 2: even lanes e[0..256], odd lanes the two weighted terms followed by e[40..256] (the skipped terms of the
    // reference are exact zeros and the padding read past bin 256 is zero).
    {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int k = L + 16 * jj;
            S.e[k] = (double)fa[jj].x * fa[jj].x + (double)fa[jj].y * fa[jj].y;
            S.e[256 - k] = (double)fb[jj].x * fb[jj].x + (double)fb[jj].y * fb[jj].y;
        }
        if (L == 0) S.e[128] = (double)f128.x * f128.x + (double)f128.y * f128.y;
        for (int k = 257 + L; k < 304; k += 16) S.e[k] = 0.0;
    }
    wave_sync();
#ifdef AT3HIP_DEBUG_KNOBS
    if (p.debug == 24) return;
#endif
    double acc = 0.0;
    if (lane < 8) {
        const int it = lane >> 1, kind = lane & 1;
        const double* E = s_item[it].e;
        const double h1 = (double)T->hpf_w[1], h2 = (double)T->hpf_w[2];
        if (kind == 1) {
            acc += E[kLowCutBin] * h1 * h1;
            acc += E[kLowCutBin + 1] * h2 * h2;
        }
        const double* src = E + (kind ? kLowCutBin + 2 : 0);
        for (int k0 = 0; k0 < 256; k0 += 16) {
            double v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = src[k0 + i];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc += v[i];
        }
        acc += src[256];
    }
    // lane 2 it holds the item's total energy, lane 2 it + 1 the filtered one
    const uint64_t ab = (uint64_t)__double_as_longlong(acc);
    const int alo = (int)(uint32_t)ab, ahi = (int)(uint32_t)(ab >> 32);
    const uint64_t fb64 = (uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(4 * (2 * row + 1), alo) |
                          ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(4 * (2 * row + 1), ahi) << 32);
    const uint64_t tb64 = (uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(4 * (2 * row), alo) |
                          ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(4 * (2 * row), ahi) << 32);
    if (valid && L == 0) {
        const double totalE = __longlong_as_double((long long)tb64), filtE = __longlong_as_double((long long)fb64);
        rec->hfr = (totalE > 0.0) ? (float)(filtE / totalE) : 0.0f;
    }
}

// Second half of the analysis of one item (see k_gain_spec): one 128-thread workgroup (two wavefronts). Every pass of the
// inverse transform gives each work-item one register-resident radix-16 unit or four butterflies; the 17 KB working set
// allows eight items = 16 wavefronts per CU.
struct GainLds {
    cpx f[2048 + 64];   // the irfft-4096 core / upsampled samples, padded (irfft_pad)
};

__global__ __launch_bounds__(128) void k_gain_analysis(GainParams p, const Tables* __restrict__ T)
{
    __shared__ __attribute__((aligned(16))) GainLds s_item[1];
    // (readfirstlane: the wavefront index as a scalar - the two halves of pass m = 2 / 8 then branch on the scalar unit and fetch
    // their wavefront-wide twiddles through it instead of as 13 vector loads behind the first barrier)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    GainLds& L = s_item[0];
    const int nfr = p.n_blocks - p.f0;
    const bool valid = true;
    int wg = blockIdx.x;
    const size_t item = blockIdx.x;
    const int band = wg % 3; wg /= 3;
    const int ch = wg % 2; wg /= 2;
    const int f = p.f0 + wg % nfr;
    const int s = wg / nfr;
    GainRec* rec = p.rec + (((size_t)s * p.n_blocks + f) * 2 + ch) * 3 + band;
    // below 5 % high-band energy the reference drops the band before the upsampler output is looked at
    // (atrac3denc.cpp:319-327): nothing downstream reads the other fields of such a record
#ifdef AT3HIP_DEBUG_KNOBS
    if (p.debug == 11) return;   // launch cost alone
#endif
    // Loads return in issue order. The item's gate goes first and the bins and input twiddles right behind it, in the same round
    // trip: a workgroup that passes the gate (they all do on busy material) has its leaves' inputs arriving while it tests the
    // ratio, one that fails leaves after the one latency it always paid. The 27 twiddles of the later passes are requested after
    // the gate and BEHIND what the leaves need (sched_barrier: see the leaves below), out of L2, with a whole stage to arrive in.
    const float hfr = rec->hfr;
    const cpx* bins = p.bins + item * kGainBins;
    const int k0 = kLowCutBin + tid, k1 = kLowCutBin + 128 + tid;
    const cpx bin0 = bins[tid];
    const cpx bin1 = bins[k1 <= 256 ? 128 + tid : tid];
    const cpx stw_in0 = T->stw2048[k0 - 1];
    const cpx stw_in1 = T->stw2048[k1 - 1 < 1024 ? k1 - 1 : 1023];
    const float hpf1 = T->hpf_w[1], hpf2 = T->hpf_w[2];
    __builtin_amdgcn_sched_barrier(0);
    if (hfr < 0.05f) return;
    __builtin_amdgcn_sched_barrier(0);   // (requests the compiler sinks below the gate still stay in front of the twiddles')
    const Tw32_128 tw_b = irfft_tw_32_128(T->gain_tw, tid);
    const Tw512<128> tw_c = irfft_tw_512<128>(T->gain_tw, tid);
    __builtin_amdgcn_sched_barrier(0);
#ifdef AT3HIP_DEBUG_KNOBS
    if (p.debug == 12) {   // launch + the fetches
        float sink = bin0.r + bin1.i + stw_in0.r + stw_in1.i + hpf1 + hpf2 + tw_b.a[0].x + tw_b.w[3][2].y + tw_c.w[0][0].x + tw_c.w[3][2].y;
        if (sink == 12345.678f) rec->target = sink;
        return;
    }
#endif
    // 3. kiss_fftri input. Only bins 38..256 survive the high-pass, so tmpbuf is non-zero at k in [38,256] and
    //    [1792,2010]; each of those meets an exact zero in its radix-2 leaf butterfly (x +- 0*w), whose two outputs
    //    are therefore stored directly.
    // (Loads return in issue order. The leaf values are formed for both bins unconditionally - k0 <= 165 always, only the second
    // bin's stores depend on k1 <= 256: with the arithmetic inside an `if` the compiler sank the two input twiddles' requests
    // into it, BEHIND the 27 twiddle requests above, and the leaves then waited for every one of those.)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int k = q ? k1 : k0;
        const cpx fr = q ? bin1 : bin0;
        cpx fk;
        const float scale = 8.0f;
        if (k == 256) {
            fk.r = fr.r * scale * 0.5f;
            fk.i = 0.0f;
        } else if (k >= kLowCutBin + 2) {
            fk.r = fr.r * scale;
            fk.i = fr.i * scale;
        } else {
            const float w = (k == kLowCutBin) ? hpf1 : hpf2;
            fk.r = fr.r * scale * w;
            fk.i = fr.i * scale * w;
        }
        // fnkc = conj(freq[2048 - k]) = (0, -0): fek = fk, tmp = fk
        const cpx fok = cmul(fk, q ? stw_in1 : stw_in0);
        cpx a, b, nb;
        a.r = fk.r + fok.r; a.i = fk.i + fok.i;
        b.r = fk.r - fok.r; b.i = -(fk.i - fok.i);
        nb.r = 0.0f - b.r; nb.i = 0.0f - b.i;
        if (k <= 256) {
            const int pa = fft_leaf_pos<2048>(k);          // even slot: partner input k + 1024 is zero
            L.f[irfft_pad(pa)] = a;
            L.f[irfft_pad(pa) + 1] = a;
            const int pb = fft_leaf_pos<2048>(2048 - k);   // odd slot: partner input 1024 - k is zero
            L.f[irfft_pad(pb) - 1] = b;
            L.f[irfft_pad(pb)] = nb;
        }
    }
    __syncthreads();
    // inverse transform: 128 radix-16 units per pass pair, one per work-item
#ifdef AT3HIP_DEBUG_KNOBS
    if (p.debug == 13) return;
#endif

    if (wave == 0) irfft_pass_2_8<0>(L.f, T->tw2048, lane);
    else irfft_pass_2_8<1>(L.f, T->tw2048, lane);
    __syncthreads();
#ifdef AT3HIP_DEBUG_KNOBS
    if (p.debug == 14) return;
#endif

    irfft_pass_32_128(L.f, tw_b, tid);
    __syncthreads();
#ifdef AT3HIP_DEBUG_KNOBS
    if (p.debug == 15) return;
#endif

    irfft_pass_512<128>(L.f, tw_c, tid);
    __syncthreads();
#ifdef AT3HIP_DEBUG_KNOBS
    if (p.debug == 16) return;
#endif


    // 4. AnalyzeGain over the upsampled samples [1024, 3072): 256 micro-chunks of 8 (4 per lane), 32 sub-frames of 64
    // complex output j holds the real samples 2j, 2j+1; sample 1024 is complex slot 512 = padded slot 528. The sub-frame RMS
    // values and the quartiles leave for HBM; the plateau target made of them is a short serial affair of 32 lanes per item and
    // runs as k_gain_tail on the light stage's stream instead of holding this kernel's 17 KB of LDS for a third of its life.
    const float norm = 1.0f / 4096.0f;
    // the 256 micro-chunk values stay on the chip: they go to the dead first quarter of the transform buffer, and while the first
    // wavefront sums the sub-frames the second one sorts every sub-frame's eight values for the quartiles
    // (transient_detector.cpp:113-133) - what used to travel to k_gain_tail as 1 KB per item is 256 bytes of results now
    float* s_micro = reinterpret_cast<float*>(L.f);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        // which chunk a work-item sums is free: chunk c starts 8 c + 2 (c / 8) dwords into the buffer, so the sixteen lanes an
        // 8-byte LDS read serves together take chunks {0..3} + 8 {0..3} (+ 4 for the next sixteen) - sixteen different bank
        // pairs - instead of sixteen neighbours, which share four (the reads were 4-way conflicts: a fifth of the kernel's LDS cycles)
        const int n = tid + 128 * q;
        const int c = (n & 3) | ((n & 0xc) << 1) | ((n & 0x10) >> 2) | (n & 0xe0);
        cpx* src = L.f + 528 + 4 * c + (c >> 3);
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f2 v = ld2(src + i) * norm;
            const f2 sq = v * v;
            st2(src + i, sq);   // the sub-frame sums below add the same squares: formed here by 128 work-items, not there by 32
            acc += sq.x;
            acc += sq.y;
        }
        acc /= 8;
        s_micro[c] = sqrtf(acc);
    }
    __syncthreads();
    if (tid < 32) {
        const cpx* src = L.f + 528 + 33 * lane;
        f2 x[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) x[i] = ld2(src + i);
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            acc += x[i].x;
            acc += x[i].y;
        }
        acc /= 64;
        rec->gain[lane] = sqrtf(acc);
    } else if (wave == 1 && lane < 32) {
        const float4 ma = *reinterpret_cast<const float4*>(s_micro + 8 * lane), mb = *reinterpret_cast<const float4*>(s_micro + 8 * lane + 4);
        float m[8] = {ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z, mb.w};
#pragma unroll
        for (int i = 1; i < 8; ++i) {   // insertion sort, ascending (static indices)
#pragma unroll
            for (int k = i; k > 0; --k) {
                const float lo = fminf(m[k - 1], m[k]), hi = fmaxf(m[k - 1], m[k]);
                m[k - 1] = lo;
                m[k] = hi;
            }
        }
        rec->lo[lane] = m[2];
        rec->hi[lane] = m[6];
    }
}

#ifdef AT3HIP_DEBUG_KNOBS
#define AT3_GPH_END(k)                                                                   \
    do {                                                                                 \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                      \
        if (ph_slots && threadIdx.x == 0) atomicAdd(ph_slots + (k), t_ - ph_last);       \
        ph_last = t_;                                                                    \
    } while (0)
#else
#define AT3_GPH_END(k) ((void)0)
#endif

// ---- the same second half as ONE wavefront per item (round 4) ---------------------------------------------------------
// The 2048-point inverse core lives in the registers of the 64 lanes, 32 points per lane; what the passes exchange crosses
// the wavefront three times:
//   phase A  lane G forms the leaves of 32-block G from the item's bins (nine 8-byte loads, a row of lanes reads 512 contiguous
//            bytes) and runs passes m = 2 and m = 8 on its two radix-16 units (parity 0 / 1) in registers;
//   exchange 1 (LDS, row-local transposition, 9.5 KB): lane (row R, j) gets offsets 2 j and 2 j + 1 of the sixteen blocks of
//            its row = the inputs of units (G2 = R, k = 2 j + u) of passes m = 32 and m = 128, one parity at a time;
//   exchange 2 (no LDS: v_permlane32_swap / v_permlane16_swap, gfx950): the butterflies of pass m = 512 combine the same
//            element of the four rows, so a 4 x 4 block transposition ACROSS the rows hands lane (R, j) elements i = 4 R .. 4 R + 3
//            of every row;
//   exchange 3 (LDS, 8.3 KB): the kept outputs (samples 1024 .. 3071) in sub-frame order for AnalyzeGain, whose ordered sums,
//            micro-chunk values and quartiles are formed by the same wavefront.
// No workgroup barrier, 9.5 KB of LDS per item: sixteen items per CU. Same butterflies on the same operands in the same order
// as the two-wavefront version (kf_bfly4 / kf_bfly2, kiss_fft.c:21-90; kiss_fftri, tools/kiss_fftr.c:117-153).
constexpr int kGa1RowStride = 18;                       // 8-byte slots per transposed position (16 + 2: conflict-free 16-byte reads)
constexpr int kGa1RowSlots = 16 * kGa1RowStride + 16;   // per row of lanes: 2432 bytes (rows half a bank round apart)
constexpr int kGa1SubStride = 33;                       // 8-byte slots per sub-frame of 32 complex outputs
struct Gain1Lds {
    union {
        cpx x[4 * kGa1RowSlots];          // exchange 1
        struct {
            cpx out[32 * kGa1SubStride];  // exchange 3: outputs 512 .. 1535 of the core, sub-frame sf at 33 sf
            float micro[256];             // micro-chunk RMS values, chunk c at c
        };
    };
};
static_assert(sizeof(Gain1Lds) <= 10240, "sixteen items per CU");

struct PermPair {
    int a, b;
};
// v_permlane32_swap: lanes 32..63 of a <-> lanes 0..31 of b; v_permlane16_swap: the odd rows (of 16 lanes) of a <-> the even rows of b
__device__ __forceinline__ PermPair perm32_swap(int a, int b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    return PermPair{(int)r[0], (int)r[1]};
}
__device__ __forceinline__ PermPair perm16_swap(int a, int b)
{
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    return PermPair{(int)r[0], (int)r[1]};
}
template <bool WIDE>
__device__ __forceinline__ void swap_f2(f2& a, f2& b)
{
    const PermPair x = WIDE ? perm32_swap((int)__float_as_uint(a.x), (int)__float_as_uint(b.x)) : perm16_swap((int)__float_as_uint(a.x), (int)__float_as_uint(b.x));
    const PermPair y = WIDE ? perm32_swap((int)__float_as_uint(a.y), (int)__float_as_uint(b.y)) : perm16_swap((int)__float_as_uint(a.y), (int)__float_as_uint(b.y));
    a = mk2(__uint_as_float((uint32_t)x.a), __uint_as_float((uint32_t)y.a));
    b = mk2(__uint_as_float((uint32_t)x.b), __uint_as_float((uint32_t)y.b));
}

__global__ __launch_bounds__(64) AT3_WAVES_PER_EU(4) void k_gain_analysis1(GainParams p, const Tables* T)
{
    __shared__ __attribute__((aligned(16))) Gain1Lds L;
    const int lane = threadIdx.x, R = lane >> 4, j = lane & 15;
    const int nfr = p.n_blocks - p.f0;
    int wg = blockIdx.x;
    const size_t item = blockIdx.x;
    const int band = wg % 3; wg /= 3;
    const int ch = wg % 2; wg /= 2;
    const int f = p.f0 + wg % nfr;
    const int s = wg / nfr;
    GainRec* rec = p.rec + (((size_t)s * p.n_blocks + f) * 2 + ch) * 3 + band;
    // below 5 % high-band energy the reference drops the band before the upsampler output is looked at (atrac3denc.cpp:319-327)
#ifdef AT3HIP_DEBUG_KNOBS
    unsigned long long ph_last = __builtin_amdgcn_s_memtime();
    unsigned long long* ph_slots = p.clk ? p.clk + (blockIdx.x & 255u) * 12u : nullptr;
#endif
    const float hfr = rec->hfr;
    if (hfr < 0.05f) return;
    const cpx* bins = p.bins + item * kGainBins;
    const cpx* tw = T->tw2048;
    AT3_GPH_END(0);

    // Every table value a wavefront needs is requested one stage AHEAD of its use (a stage computes for a few thousand cycles,
    // a fetch that is asked for where it is needed costs the wavefront about as much again: with four wavefronts per SIMD in the
    // same stage nobody covers it). AT3_STAGE() keeps the compiler from sinking the requests back to their uses.
#define AT3_STAGE() __builtin_amdgcn_sched_barrier(0)
    // ---- phase A: leaves of 32-block G = lane. Slot 32 G + kappa + 2 i' is the leaf of input n + 1024 kappa,
    // n = r0 + 64 (i' / 4) + 256 (i' % 4), r0 = the digit reversal of G; the radix-2 leaf butterfly of the pair (n, n + 1024) has one
    // non-zero input at most: bin k = n in [38, 256] gives (a, a), input 2048 - k = n + 1024 (n in [768, 986]) gives (b, -b).
    const int r0 = (lane >> 4) + 4 * ((lane >> 2) & 3) + 16 * (lane & 3);
    cpx ba[4], bb[4], sa[4], sb[4], b256 = {0.0f, 0.0f}, s256 = {0.0f, 0.0f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ka = r0 + 64 * q, kb = 256 - r0 - 64 * q;
        const int kac = ka >= kLowCutBin ? ka : kLowCutBin, kbc = kb >= kLowCutBin ? kb : kLowCutBin;   // (clamped: dropped below)
        ba[q] = bins[kac - kLowCutBin];
        sa[q] = T->stw2048[kac - 1];
        bb[q] = bins[kbc - kLowCutBin];
        sb[q] = T->stw2048[kbc - 1];
    }
    if (lane == 0) {
        b256 = bins[256 - kLowCutBin];
        s256 = T->stw2048[255];
    }
    const float hpf1 = T->hpf_w[1], hpf2 = T->hpf_w[2];
    // the twiddles of passes m = 2 / 8 are the same in every lane (scalar registers): both parities' now
    f2 a1[2], a3[2], w8[2][4][3];
#pragma unroll
    for (int kappa = 0; kappa < 2; ++kappa) {
        a1[kappa] = ld2(tw + 256 * kappa);   // pass m = 2: fstride 256, k = kappa (tw[512 kappa] only ever meets a zero)
        a3[kappa] = ld2(tw + 768 * kappa);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {     // pass m = 8: fstride 64, k = kappa + 2 jj
            const int k = kappa + 2 * jj;
            w8[kappa][jj][0] = ld2(tw + 64 * k);
            w8[kappa][jj][1] = ld2(tw + 128 * k);
            w8[kappa][jj][2] = ld2(tw + 192 * k);
        }
    }
    // the fifteen twiddles of a unit of passes m = 32 / 128: one 128-byte run per slot (k = 2 j + kappa)
    f2 twb[15];
#pragma unroll
    for (int sl = 0; sl < 15; ++sl) twb[sl] = ld2(&T->ga1_twb[0][sl][j]);
    AT3_STAGE();
    f2 la[4], lb[4], la256 = mk2(0.0f, 0.0f);   // a of bins r0 + 64 q4; b of bins 256 - r0 - 64 q4; a of bin 256 (lane 0)
    {
        auto leaf = [&](cpx fr, cpx stw, int k, bool want_b) -> f2 {
            cpx fk;
            const float scale = 8.0f;
            if (k == 256) {
                fk.r = fr.r * scale * 0.5f;
                fk.i = 0.0f;
            } else if (k >= kLowCutBin + 2) {
                fk.r = fr.r * scale;
                fk.i = fr.i * scale;
            } else {
                const float w = (k == kLowCutBin) ? hpf1 : hpf2;
                fk.r = fr.r * scale * w;
                fk.i = fr.i * scale * w;
            }
            const cpx fok = cmul(fk, stw);   // fnkc = conj(freq[2048 - k]) = (0, -0): fek = fk, tmp = fk
            return want_b ? mk2(fk.r - fok.r, -(fk.i - fok.i)) : mk2(fk.r + fok.r, fk.i + fok.i);
        };
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ka = r0 + 64 * q, kb = 256 - r0 - 64 * q;
            la[q] = (ka >= kLowCutBin) ? leaf(ba[q], sa[q], ka, false) : mk2(0.0f, 0.0f);   // (ka <= 255 always)
            lb[q] = (kb >= kLowCutBin) ? leaf(bb[q], sb[q], kb, true) : mk2(0.0f, 0.0f);    // (kb = 256 for r0 = 0, q = 0: bin 256's b)
        }
        if (lane == 0) la256 = leaf(b256, s256, 256, false);
    }
    AT3_GPH_END(1);
    // the units' results after passes m = 32 / 128, element i of unit u at y[u][i]
    f2 y[2][16];
    f2 twc[2][4][3];   // pass m = 512: tw[(q + 1) kk] of butterfly kk = 2 j + u + 32 (4 R + t), 512 contiguous bytes per fetch
#pragma unroll
    for (int kappa = 0; kappa < 2; ++kappa) {
        f2 x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = mk2(0.0f, 0.0f);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            x[4 * q] = la[q];                                                        // (a, a)
            x[4 * q + 3] = kappa ? mk2(0.0f - lb[q].x, 0.0f - lb[q].y) : lb[q];      // (b, -b)
        }
        x[1] = la256;
        // pass m = 2: the third input of every butterfly is a zero leaf, the second one too except for input 256
        bfly4_inv_sparse<false>(x[0], x[1], x[2], x[3], a1[kappa], a3[kappa]);
#pragma unroll
        for (int g = 1; g < 4; ++g) bfly4_inv_sparse<true>(x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3], a1[kappa], a3[kappa]);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) bfly4<true>(x[jj], x[jj + 4], x[jj + 8], x[jj + 12], w8[kappa][jj][0], w8[kappa][jj][1], w8[kappa][jj][2]);
        AT3_GPH_END(2);
        // exchange 1: x[i'] is slot 32 G + kappa + 2 i'; unit (G2 = R, k = 2 jk + kappa) of the next passes wants offset k of blocks 16 R + i
        cpx* row = L.x + R * kGa1RowSlots;
#pragma unroll
        for (int i = 0; i < 16; ++i) st2(row + i * kGa1RowStride + j, x[i]);
        wave_sync();
        f2 b[16];
        {
            const float4* src = reinterpret_cast<const float4*>(row + j * kGa1RowStride);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const float4 v = src[m];
                b[2 * m] = mk2(v.x, v.y);
                b[2 * m + 1] = mk2(v.z, v.w);
            }
        }
        wave_sync();   // (the buffer takes the other parity next, then the outputs)
        AT3_GPH_END(3);
        // passes m = 32 (fstride 16, butterfly k) and m = 128 (fstride 4, butterflies k + 32 jj) of unit (R, k = 2 j + kappa)
#pragma unroll
        for (int g = 0; g < 4; ++g) bfly4<true>(b[4 * g], b[4 * g + 1], b[4 * g + 2], b[4 * g + 3], twb[0], twb[1], twb[2]);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) bfly4<true>(b[jj], b[jj + 4], b[jj + 8], b[jj + 12], twb[3 + 3 * jj], twb[4 + 3 * jj], twb[5 + 3 * jj]);
#pragma unroll
        for (int i = 0; i < 16; ++i) y[kappa][i] = b[i];
        AT3_STAGE();
        // requested now, used a stage later: the other unit's twiddles, then the last pass'
        if (kappa == 0) {
#pragma unroll
            for (int sl = 0; sl < 15; ++sl) twb[sl] = ld2(&T->ga1_twb[1][sl][j]);
        } else {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int q = 0; q < 3; ++q) twc[u][t][q] = ld2(&T->ga1_twc[4 * u + t][q][lane]);
        }
        AT3_STAGE();
        AT3_GPH_END(4);
    }
    // ---- exchange 2: y[u][i] is slot 512 R + (2 j + u) + 32 i; butterfly kk = 2 j + u + 32 i of pass m = 512 wants it from every row.
    // Rows R and R ^ 2 trade their halves i >= 8 / i < 8, then rows R and R ^ 1 the quarters: afterwards y[u][4 q + t] is
    // element i = 4 R + t of row q.
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 8; ++t) swap_f2<true>(y[u][t], y[u][8 + t]);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            swap_f2<false>(y[u][t], y[u][4 + t]);
            swap_f2<false>(y[u][8 + t], y[u][12 + t]);
        }
    AT3_GPH_END(5);
    // ---- pass m = 512 (fstride 1) and exchange 3: only the middle half of the upsampled frame is analysed (outputs kk + 512, kk + 1024)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int i = 4 * R + t;   // butterfly kk = 2 j + u + 32 i
            bfly4<true>(y[u][t], y[u][4 + t], y[u][8 + t], y[u][12 + t], twc[u][t][0], twc[u][t][1], twc[u][t][2]);
            st2(L.out + kGa1SubStride * i + 2 * j + u, y[u][4 + t]);          // slot 512 + kk: sub-frame i, position 2 j + u
            st2(L.out + kGa1SubStride * (16 + i) + 2 * j + u, y[u][8 + t]);   // slot 1024 + kk: sub-frame 16 + i
        }
    wave_sync();
    AT3_GPH_END(6);
    // ---- AnalyzeGain over the upsampled samples [1024, 3072) (transient_detector.cpp:95-136): 256 micro-chunks of 8 samples (four
    // per lane), 32 sub-frames of 64; complex output jj holds the real samples 2 jj, 2 jj + 1
    const float norm = 1.0f / 4096.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        // chunk c starts 4 c + c / 8 slots into the buffer; the sixteen lanes an 8-byte LDS read serves together take chunks
        // {0..3} + 8 {0..3} (+ 4 for the next sixteen): sixteen different bank pairs
        const int n = lane + 64 * q;
        const int c = (n & 3) | ((n & 0xc) << 1) | ((n & 0x10) >> 2) | (n & 0xe0);
        const cpx* src = L.out + 4 * c + (c >> 3);
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f2 v = ld2(src + i) * norm;
            const f2 sq = v * v;
            acc += sq.x;
            acc += sq.y;
        }
        acc /= 8;
        L.micro[c] = sqrtf(acc);
    }
    wave_sync();
    AT3_GPH_END(7);
    if (lane < 32) {
        const cpx* src = L.out + kGa1SubStride * lane;
        f2 x[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) x[i] = ld2(src + i);
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const f2 v = x[i] * norm;
            const f2 sq = v * v;
            acc += sq.x;
            acc += sq.y;
        }
        acc /= 64;
        rec->gain[lane] = sqrtf(acc);
    } else {
        const int sf = lane - 32;
        const float4 ma = *reinterpret_cast<const float4*>(L.micro + 8 * sf), mb = *reinterpret_cast<const float4*>(L.micro + 8 * sf + 4);
        float m[8] = {ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z, mb.w};
#pragma unroll
        for (int i = 1; i < 8; ++i) {   // insertion sort, ascending (static indices)
#pragma unroll
            for (int k = i; k > 0; --k) {
                const float lo = fminf(m[k - 1], m[k]), hi = fmaxf(m[k - 1], m[k]);
                m[k - 1] = lo;
                m[k] = hi;
            }
        }
        rec->lo[sf] = m[2];
        rec->hi[sf] = m[6];
    }
#ifdef AT3HIP_DEBUG_KNOBS
    AT3_GPH_END(8);
    if (ph_slots && lane == 0) atomicAdd(ph_slots + 11, 1ull);
#endif
}

// CalcCurve's target for one item: the plateau target (transient_detector.cpp:178-238, 284-297), the mean gain (the quartiles
// of the micro-chunk values are k_gain_analysis' since round 3: the 256 values no longer cross HBM). 32 lanes per item,
// eight items per workgroup; items below the 5 % gate have no data and are skipped like k_gain_analysis skipped them.
__global__ __launch_bounds__(256) void k_gain_tail(GainParams p, int n_items)
{
    __shared__ float s_g[8][32], s_f[8][32], s_m[8][32];
    const int tid = threadIdx.x, grp = tid >> 5, j = tid & 31, half = grp & 1;
    const int nfr = p.n_blocks - p.f0;
    int item = blockIdx.x * 8 + grp;
    const bool valid = item < n_items;
    if (!valid) item = n_items - 1;
    int wg = item;
    const int band = wg % 3; wg /= 3;
    const int ch = wg % 2; wg /= 2;
    const int f = p.f0 + wg % nfr;
    const int s = wg / nfr;
    GainRec* rec = p.rec + (((size_t)s * p.n_blocks + f) * 2 + ch) * 3 + band;
    // (the item's values are requested together with its gate: an item below the gate has stale numbers there, which are
    // fetched and dropped - two dependent global-memory latencies would cost this short kernel more)
    const float hfr = rec->hfr;
    const float g_j = rec->gain[j];
    asm volatile("" ::"v"(g_j));   // (keeps the second request in front of the gate: the compiler sinks it behind the branch otherwise)
    const bool active = valid && !(hfr < 0.05f);
    if (__ballot(active) == 0ull) return;
    float* s_gain = s_g[grp];
    float* s_filt = s_f[grp];
    float* s_minv = s_m[grp];
    const float in_j = active ? g_j : 0.0f;
    s_gain[j] = in_j;
    wave_sync();
    // plateau target of CalcCurve (transient_detector.cpp:178-238, 284-297) with ballots
    float filt_j;
    {
        const float a = s_gain[j > 0 ? j - 1 : 0], c = s_gain[j < 31 ? j + 1 : 31];
        if (j == 0) filt_j = fmaxf(in_j, c);
        else if (j == 31) filt_j = fmaxf(a, in_j);
        else filt_j = fmaxf(fminf(a, in_j), fminf(fmaxf(a, in_j), c));
    }
    s_filt[j] = filt_j;
    wave_sync();
    s_minv[j] = (j <= 29) ? fminf(fminf(filt_j, s_filt[j < 30 ? j + 1 : 31]), s_filt[j < 30 ? j + 2 : 31]) : -1.0f;
    wave_sync();
    float maxRaw = 0.0f, sum = 0.0f, bestLevel = 0.0f;
    int bestEnd = -1;
    for (int k = 0; k < 32; ++k) {
        const float g = s_gain[k];
        maxRaw = fmaxf(maxRaw, g);
        sum += g;
        const float mv = s_minv[k];
        if (k <= 29 && mv > bestLevel) {
            bestLevel = mv;
            bestEnd = k + 2;
        }
    }
    const uint32_t ge = (uint32_t)(__ballot(filt_j >= bestLevel) >> (32 * half));
    const uint32_t high = (uint32_t)(__ballot(in_j >= bestLevel * 0.7f) >> (32 * half));
    const float last = s_gain[31];
    float plateau = 0.0f;
    bool release = false;
    if (!(bestLevel < 1e-6f)) {
        plateau = bestLevel;
        const uint32_t x = (bestEnd + 1 < 32) ? (ge >> (bestEnd + 1)) : 0u;
        bestEnd += (x == 0xffffffffu) ? 32 : __builtin_ctz(~x);   // extend while filtered stays at plateau level
        if (bestEnd < 31) {
            if (last < bestLevel * 0.1f) release = true;
            else release = ((high >> (bestEnd + 1)) == 0u) && (last < bestLevel * 0.5f);
        }
    }
    const bool usePlateau = plateau > 1e-6f && !release && plateau >= maxRaw * 0.4f;
    if (active && j == 0) {
        rec->cur_hpf = sum / 32.0f;
        rec->target = usePlateau ? plateau : last;
    }
}

// Context chain (TCurveBuilderCtx): one wavefront per (stream, channel, band<3). Frames with hfr < 0.05 reset
// LastLevel and leave LastTarget / LastHpfEnergy untouched (atrac3denc.cpp:319-327); otherwise
// LastHpfEnergy = mean(gain) (:342-346), LastLevel = gain[31], LastTarget = target (transient_detector.cpp:307-310).
// The lanes fetch the per-frame inputs in parallel, lane 0 walks the chain out of LDS, all lanes store the
// context each frame starts from.
constexpr int kScanChunk = 512;
__global__ __launch_bounds__(64) void k_gain_scan(GainParams p, int n_streams)
{
    __shared__ float s_hfr[kScanChunk], s_g31[kScanChunk], s_tgt[kScanChunk], s_hpf[kScanChunk];
    const int lane = threadIdx.x;
    const int idx = blockIdx.x;
    if (idx >= n_streams * 6) return;
    const int band = idx % 3, ch = (idx / 3) % 2, s = idx / 6;
    BandState* st = p.state + (size_t)s * 8 + ch * 4 + band;
    float lvl = st->last_level, tgt = st->last_target, hpf = st->last_hpf;
    const int nfr = p.n_blocks - p.f0;
    for (int base = 0; base < nfr; base += kScanChunk) {
        const int cnt = (nfr - base < kScanChunk) ? nfr - base : kScanChunk;
        for (int i = lane; i < cnt; i += 64) {
            const GainRec* rec = p.rec + (((size_t)s * p.n_blocks + p.f0 + base + i) * 2 + ch) * 3 + band;
            s_hfr[i] = rec->hfr;
            s_g31[i] = rec->gain[31];
            s_tgt[i] = rec->target;
            s_hpf[i] = rec->cur_hpf;
        }
        __syncthreads();
        if (lane == 0) {
            for (int i = 0; i < cnt; ++i) {
                const float h = s_hfr[i], g = s_g31[i], t = s_tgt[i], e = s_hpf[i];
                s_g31[i] = lvl;    // context before this frame
                s_tgt[i] = tgt;
                s_hpf[i] = hpf;
                if (h < 0.05f) {
                    lvl = 0.0f;
                } else {
                    lvl = g;
                    tgt = t;
                    hpf = e;
                }
            }
        }
        __syncthreads();
        for (int i = lane; i < cnt; i += 64) {
            GainRec* rec = p.rec + (((size_t)s * p.n_blocks + p.f0 + base + i) * 2 + ch) * 3 + band;
            rec->ctx_level = s_g31[i];
            rec->ctx_target = s_tgt[i];
            rec->ctx_hpf = s_hpf[i];
        }
        __syncthreads();
    }
    if (lane == 0) {
        st->last_level = lvl;
        st->last_target = tgt;
        st->last_hpf = hpf;
    }
}

__device__ inline uint32_t first_set_bit(uint32_t x)   // index of the highest set bit, 0 for x < 2 (the reference's shift loop)
{
    return x ? 31u - (uint32_t)__builtin_clz(x) : 0u;
}

// transient_detector.cpp:141-149
__device__ inline int relation_to_idx(float x)
{
    if (x <= 0.5f) {
        x = 1.0f / fmaxf(x, 0.00048828125f);
        return 4 + (int)first_set_bit((uint32_t)x);
    }
    x = fminf(x, 16.0f);
    return 4 - (int)first_set_bit((uint32_t)x);
}

// atrac3denc.h:44-52 (1.0 / x in double then narrowed == correctly rounded float quotient)
__device__ inline int relation_to_idx_hdr(float x)
{
    if (x <= 0.5f) {
        x = 1.0f / fmaxf(x, 0.00048828125f);
        return 4 + (int)first_set_bit((uint32_t)(int32_t)truncf(x));
    }
    x = fminf(x, 16.0f);
    return 4 - (int)first_set_bit((uint32_t)(int32_t)truncf(x));
}


// ---- CalcCurve + CreateSubbandInfo tail, 32 lanes per (stream, frame, channel, band<3) item ------------------
// Lane j of a half-wavefront owns sub-frame j. Everything that is order-free (median filter, level
// quantisation, boundary scores, log2 terms) runs on all 32 lanes; the order-dependent parts (sticky level chain,
// right-to-left transition scan, float sums) are wave-uniform loops that broadcast one lane per step with
// readlane, so both items of a wavefront advance in lockstep without LDS round trips or barriers.
__device__ __forceinline__ float grp_readlane_f(float v, int idx, int half)
{
    const int a = __builtin_amdgcn_readlane((int)__float_as_uint(v), idx);
    const int b = __builtin_amdgcn_readlane((int)__float_as_uint(v), 32 + idx);
    return __uint_as_float((uint32_t)(half ? b : a));
}
__device__ __forceinline__ int grp_readlane_i(int v, int idx, int half)
{
    const int a = __builtin_amdgcn_readlane(v, idx);
    const int b = __builtin_amdgcn_readlane(v, 32 + idx);
    return half ? b : a;
}

struct CurvePts {   // register-resident curve (redundant in every lane of the item)
    int n;
    int level[7];
    int loc[7];
};

// CalcCurveEarlyMismatchScore (atrac3denc.cpp:228-297); in_j / in_next are this lane's gain[j], gain[j+1].

// CalcCurveEarlyMismatchScore (atrac3denc.cpp:259-297) for TWO curves of one item at once (the curve before and after the point-0
// logic: the reference calls it twice):
// the two evaluations share every LDS rendezvous and their chains interleave - the kernel's slowest wavefronts are the ones that
// score, and a wavefront that waits on one dependent chain at a time is what sets the kernel's duration. A's lists live in
// s_tmpA (128 floats) / s_ptsA, B's in aB (32 floats) + sqB / ltB (32 floats each) / s_ptsB; the weights list is A's.
__device__ __forceinline__ void early_mismatch_score_pair(const Log2fTab* L2, const float* gain_interp, float in_j, float in_next, float target,
                                                          const CurvePts& cpA, const CurvePts& cpB, float* s_tmpA, uint8_t* s_ptsA,
                                                          float* aB, float* sqB, float* ltB, uint8_t* s_ptsB, int j, float& scoreA, float& scoreB)
{
    // BuildSampleDivisors restricted to sub-frame j (samples 8j .. 8j+7): level boundaries and ramps are aligned to these cells, so
    // the eight divisors are all 1, all one level, or one running-product ramp (same values as curve_divisor sample by sample). The
    // point list is packed into two 8-byte words from its registers (static indices) and walked by selects (cell_divisors_packed,
    // at3_k_frontend.hpp): as a loop with lane conditions over a list in LDS this walk was the slowest part of a scoring wavefront.
    auto cell_div = [&](const CurvePts& cp) -> float {
        uint64_t lo = (uint64_t)(uint32_t)cp.n, hi = 0;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            lo |= (uint64_t)((uint32_t)cp.level[i] & 0xffu) << (8 * (i + 1));
            hi |= (uint64_t)((uint32_t)cp.loc[i] & 0xffu) << (8 * i);
        }
        float d[8];
        cell_divisors_packed(lo, hi, gain_interp, 8 * j, d);
        float dsum = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) dsum += d[k];
        return dsum / 8.0f;
    };
    (void)s_ptsA; (void)s_ptsB;
    const float divA = cell_div(cpA), divB = cell_div(cpB);
    int maxLocA = 0, maxLocB = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        if (i < cpA.n && cpA.loc[i] > maxLocA) maxLocA = cpA.loc[i];
        if (i < cpB.n && cpB.loc[i] > maxLocB) maxLocB = cpB.loc[i];
    }
    int evalA = maxLocA + 3 > 3 ? maxLocA + 3 : 3, evalB = maxLocB + 3 > 3 ? maxLocB + 3 : 3;
    if (evalA > 32) evalA = 32;
    if (evalB > 32) evalB = 32;
    const float eps = 1e-9f;
    const float modA = in_j / fmaxf(divA, eps), modB = in_j / fmaxf(divB, eps);
    const float eA = at3_log2f(L2, fmaxf(modA, eps) / fmaxf(target, eps)), eB = at3_log2f(L2, fmaxf(modB, eps) / fmaxf(target, eps));
    const float sqA = eA * eA, sqvB = eB * eB;
    const float aA = at3_log2f(L2, fmaxf(divA, eps)), avB = at3_log2f(L2, fmaxf(divB, eps));
    s_tmpA[j] = aA;
    aB[j] = avB;
    wave_sync();
    const float aA_next = s_tmpA[j < 31 ? j + 1 : 31], aB_next = aB[j < 31 ? j + 1 : 31];
    wave_sync();
    const float dA = aA_next - aA, dB = aB_next - avB;
    const float w = 0.5f * (in_j + in_next);
    const float ltermA = dA * dA * w, ltermB = dB * dB * w;
    // The reference's sums stop at the evaluated sub-frames (fit: sf < eval, leak and weight: sf + 1 < eval). The terms behind that point are
    // stored as +0.0f instead and every list is summed to its end: x + 0.0f is x for every x a sum that started at +0.0f can hold (never
    // -0.0f; NaN and inf stay what they are), and six conditions per sub-frame - 192 per scoring wavefront - become plain adds. B's weights
    // get a list of their own for it (the first 32 floats of the item's scratch: the a-list, dead by now).
    float* st = s_tmpA + 32;   // A: [3][32] behind the first 32 floats of the item's scratch
    st[j] = j < evalA ? sqA : 0.0f;
    st[32 + j] = j + 1 < evalA ? ltermA : 0.0f;
    st[64 + j] = j + 1 < evalA ? w : 0.0f;
    sqB[j] = j < evalB ? sqvB : 0.0f;
    ltB[j] = j + 1 < evalB ? ltermB : 0.0f;
    s_tmpA[j] = j + 1 < evalB ? w : 0.0f;
    wave_sync();
    float fitA = 0.0f, leakA = 0.0f, wsumA = 0.0f, fitB = 0.0f, leakB = 0.0f, wsumB = 0.0f;
    {
        const float4* a4 = reinterpret_cast<const float4*>(st);
        const float4* s4 = reinterpret_cast<const float4*>(sqB);
        const float4* l4 = reinterpret_cast<const float4*>(ltB);
        const float4* w4 = reinterpret_cast<const float4*>(s_tmpA);
        // (sixteen terms of each of the six lists per step, the next step's requested ahead by the compiler's schedule: holding all
        // forty-eight 16-byte words at once cost the kernel a wavefront per SIMD)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 va = a4[q], vl = a4[8 + q], vw = a4[16 + q], vb = s4[q], vm = l4[q], vx = w4[q];
            const float v[4] = {va.x, va.y, va.z, va.w};
            const float lt[4] = {vl.x, vl.y, vl.z, vl.w};
            const float ww[4] = {vw.x, vw.y, vw.z, vw.w};
            const float vB[4] = {vb.x, vb.y, vb.z, vb.w};
            const float lB[4] = {vm.x, vm.y, vm.z, vm.w};
            const float wB[4] = {vx.x, vx.y, vx.z, vx.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fitA += v[t];
                leakA += lt[t];
                wsumA += ww[t];
                fitB += vB[t];
                leakB += lB[t];
                wsumB += wB[t];
            }
        }
    }
    wave_sync();   // the scratch is rewritten by the next user
    fitA /= (float)evalA;
    fitB /= (float)evalB;
    if (wsumA > eps) leakA /= wsumA;
    if (wsumB > eps) leakB /= wsumB;
    scoreA = (target <= 1e-9f) ? 0.0f : fitA + 0.25f * leakA;
    scoreB = (target <= 1e-9f) ? 0.0f : fitB + 0.25f * leakB;
}

// 256 threads = 8 items per workgroup.
__global__ __launch_bounds__(256) void k_gain_curve(GainParams p, const Tables* T, int n_streams)
{
    __shared__ float s_in[8][32];
    __shared__ __attribute__((aligned(16))) float s_filt[8][32];   // (after the boundary scores: the paired evaluation's second a-list)
    __shared__ __attribute__((aligned(16))) float s_tmp[8][128];   // per item: 32 floats + three 32-term lists of the early-mismatch score
    __shared__ __attribute__((aligned(16))) int s_tloc[8][32];     // (after the curve points are formed: the paired evaluation's
    __shared__ __attribute__((aligned(16))) int s_tdelta[8][32];   //  second term lists)
    __shared__ int s_tlev[8][32];
    __shared__ uint8_t s_pts[8][16];  // per item: levels [0..6], locations [8..14] of the curve being scored
    __shared__ uint8_t s_pts2[8][16]; // the same for the second curve of the paired evaluation
    __shared__ Log2fTab s_l2[4];      // per wavefront: the log2f tables and GainInterpolation, so that the rare long
    __shared__ float s_gi4[4][32];    // path (a few items with curves set the kernel's duration) has no dependent global loads
    const int tid = threadIdx.x;
    const int grp = tid >> 5, j = tid & 31, half = grp & 1;
    const Log2fTab* L2 = &s_l2[tid >> 6];
    const float* gi = s_gi4[tid >> 6];
    const int nfr = p.n_blocks - p.f0;
    const int n_items = n_streams * nfr * 6;
    int item = blockIdx.x * 8 + grp;
    const bool valid = item < n_items;
    if (!valid) item = n_items - 1;   // keep the wavefront uniform; results are discarded
    int idx = item;
    const int band = idx % 3; idx /= 3;
    const int ch = idx % 2; idx /= 2;
    const int f = p.f0 + idx % nfr;
    const int s = idx / nfr;
    const GainRec* rec = p.rec + (((size_t)s * p.n_blocks + f) * 2 + ch) * 3 + band;
    Curve* dst = p.curves + ((size_t)s * p.n_blocks + f) * 8 + ch * 4 + band;

    const float hfr = rec->hfr;
    const float in_j = rec->gain[j], lo_j = rec->lo[j], hi_j = rec->hi[j];
    const float curHpf = rec->cur_hpf, prevHpf = rec->ctx_hpf;
    const float prevTarget = rec->ctx_target, savedLastLevel = rec->ctx_level;
    const float target = rec->target;   // == ctx.LastTarget after CalcCurve
    const float hpfRatio = (curHpf > 1e-9f && prevHpf > 1e-9f) ? (prevHpf / curHpf) : 1.0f;
    const float minScore = 1.9f * fminf(1.5f, fmaxf(1.0f, hpfRatio));

    s_in[grp][j] = in_j;
    wave_sync();
    // 3-point median (transient_detector.cpp:151-166); the 2-element edge windows return the larger value
    float filt_j;
    {
        const float a = s_in[grp][j > 0 ? j - 1 : 0], c = s_in[grp][j < 31 ? j + 1 : 31];
        if (j == 0) filt_j = fmaxf(in_j, c);
        else if (j == 31) filt_j = fmaxf(a, in_j);
        else filt_j = fmaxf(fminf(a, in_j), fminf(fmaxf(a, in_j), c));
    }
    const float in_next = s_in[grp][j < 31 ? j + 1 : 31];
    s_filt[grp][j] = filt_j;
    float maxGain = fmaxf(0.0f, in_j);   // max over the item's 32 lanes (order-free)
#pragma unroll
    for (int m = 1; m < 32; m <<= 1) maxGain = fmaxf(maxGain, __shfl_xor(maxGain, m, 64));
    wave_sync();

    // ---- CalcCurve (transient_detector.cpp:299-482) ----
    const bool active = valid && !(hfr < 0.05f) && !(target < 1e-6f) && !(savedLastLevel < 1e-6f);
    if (p.debug == 1) return;
    // An item without curve points after CalcCurve ends as "no_curve" whatever the later stages say
    // (atrac3denc.cpp:395-400), and most items are like that: a wavefront whose two items are both out stops here.
    if (__ballot(active) == 0ull) {
        if (valid && j == 0) {
            Curve out;
            out.pad = 0;
            out.n = 0;
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                out.level[i] = 0;
                out.loc[i] = 0;
            }
            *dst = out;
        }
        return;
    }
    const float intraRatio = maxGain / fmaxf(target, 1e-9f);
    float interRatio = 1.0f;
    if (prevTarget > 1e-6f) interRatio = fmaxf(prevTarget, target) / fmaxf(fminf(prevTarget, target), 1e-9f);
    const bool sticky = intraRatio <= 7.0f && interRatio <= 10.0f;
    const int raw = relation_to_idx(filt_j / target);
    // Every sub-frame of both items at the unit level: the sticky chain below can only copy a left neighbour's level or keep
    // the raw one, so every level stays 4, no transition exists and the items end as "no_curve" (the exit further down) -
    // the rule on stationary material, taken here before the quartile ratios, the chain and the boundary scores are formed.
    if (__ballot(active && j <= 30 && raw != 4) == 0ull) {
        if (valid && j == 0) {
            Curve out;
            out.pad = 0;
            out.n = 0;
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                out.level[i] = 0;
                out.loc[i] = 0;
            }
            *dst = out;
        }
        return;
    }
    int minIdx = 0, maxIdx = 0;
    {
        float ratioLo = lo_j / target, ratioHi = hi_j / target;
        if (ratioLo > ratioHi) {
            const float t = ratioLo;
            ratioLo = ratioHi;
            ratioHi = t;
        }
        const int idxLo = relation_to_idx(ratioLo), idxHi = relation_to_idx(ratioHi);
        minIdx = idxLo < idxHi ? idxLo : idxHi;
        maxIdx = idxLo < idxHi ? idxHi : idxLo;
    }
    // sticky quantisation chain (:360-383): L[j] = f_j(L[j-1]) with f_j(x) = x when sub-frame j may stick to its left
    // neighbour's level, raw[j] otherwise. Solved by relaxation from L = raw: every round re-evaluates all sub-frames
    // against the current left neighbour; a round without change is the forward-substitution result.
    int L = raw;
    {
        const bool may_stick = j > 0 && sticky && maxIdx - minIdx <= 1;
        if (__ballot(may_stick) != 0ull) {
            for (int round = 0; round < 32; ++round) {
                const int prev = __builtin_amdgcn_update_dpp(0, L, 0x138, 0xf, 0xf, false);   // wave_shr:1 = lane j-1
                const int d = raw - prev;
                const int nl = (may_stick && (d == 1 || d == -1) && prev >= minIdx && prev <= maxIdx) ? prev : raw;
                const bool moved = nl != L;
                L = nl;
                if (__ballot(moved) == 0ull) break;
            }
        }
    }
    const unsigned long long nzm = __ballot(active && j <= 30 && L != 4);
    const uint32_t gm = (uint32_t)(nzm >> (32 * half));
    const int targetSf = gm ? 32 - __builtin_clz(gm) : 0;
    // BoundaryTransientScore for loc = j (:255-274)
    float bs = 1.0f;
    if (j >= 1) {
        float leftMax = 0.0f, rightMax = 0.0f;
        for (int i = (j - 3 > 0 ? j - 3 : 0); i < j; ++i) leftMax = fmaxf(leftMax, s_filt[grp][i]);
        for (int i = j; i < (j + 3 < 32 ? j + 3 : 32); ++i) rightMax = fmaxf(rightMax, s_filt[grp][i]);
        const float eps = 1e-9f;
        bs = fmaxf((rightMax + eps) / (leftMax + eps), (leftMax + eps) / (rightMax + eps));
    }
    // right-to-left transition scan (:404-450). Sub-frames whose level equals the running `prev` are no-ops, so the
    // scan jumps from one differing sub-frame to the next lower one (ballot + find-first-set) instead of visiting all 31.
    int nt = 0;
    {
        int prev = 4;
        uint32_t remaining = targetSf >= 31 ? 0x7fffffffu : ((1u << targetSf) - 1u);   // sf < targetSf, sf <= 30
        for (int it = 0; it < 32; ++it) {
            const unsigned long long m = __ballot(L != prev);
            const uint32_t cand = (uint32_t)(m >> (32 * half)) & remaining;
            const bool done = cand == 0u;
            if (__ballot(!done) == 0ull) break;
            const int sf = done ? 0 : 31 - __builtin_clz(cand);
            const int lev = __builtin_amdgcn_ds_bpermute(4 * (32 * half + sf), L);
            const float sc = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(4 * (32 * half + sf + 1), (int)__float_as_uint(bs)));
            if (!done) {
                const int loc = sf + 1;
                const int delta = lev > prev ? lev - prev : prev - lev;
                const bool keep = (loc == targetSf) || (delta >= 2) || (sc >= minScore);
                if (keep) {
                    if (j == 0) {
                        s_tloc[grp][nt] = loc;
                        s_tlev[grp][nt] = lev;
                        s_tdelta[grp][nt] = delta;
                    }
                    ++nt;
                    prev = lev;
                }
                remaining &= (1u << sf) - 1u;
            }
        }
    }
    wave_sync();
    // The transitions sit in s_t*[grp][0 .. nt) in DESCENDING location order. The reference reverses them, and when there are more
    // than six keeps the six largest |delta| (ties: the rightmost location) and puts those back in ascending location order
    // (transient_detector.cpp:446-475; locations are distinct, so (delta desc, loc desc) is a strict order and the stable sort has
    // nothing to decide). Lane i of the group takes transition i: its place in that order is the number of transitions ahead of it,
    // its place in the result the number of kept transitions with a smaller location = kept lanes above it. (The first version
    // was one lane swapping and insertion-sorting through LDS: quadratic in nt with an LDS round trip per step - on material
    // whose levels flicker, 22 of the kernel's 56 us.)
    {
        const bool has_t = j < nt;
        int my_loc = 0, my_lev = 0, my_delta = 0;
        if (has_t) {
            my_loc = s_tloc[grp][j];
            my_lev = s_tlev[grp][j];
            my_delta = s_tdelta[grp][j];
        }
        int ahead = 0;
        if (nt > 6) {
#pragma unroll 4
            for (int k = 0; k < nt; ++k) {
                const int dk = s_tdelta[grp][k], lk = s_tloc[grp][k];   // (the same address in every lane of the group: a broadcast)
                ahead += (dk > my_delta || (dk == my_delta && lk > my_loc)) ? 1 : 0;
            }
        }
        const bool keep_t = has_t && ahead < 6;
        const uint32_t kept = (uint32_t)(__ballot(keep_t) >> (32 * half));
        const int pos = j < 31 ? __popc(kept >> (j + 1)) : 0;
        wave_sync();   // every lane has read its transition: the lists are rewritten in place
        if (keep_t) {
            s_tlev[grp][pos] = my_lev;
            s_tloc[grp][pos] = my_loc;
        }
        nt = __popc(kept);
    }
    wave_sync();
    if (nt > 6) nt = 6;
    CurvePts pts;
    pts.n = (active && targetSf > 0) ? nt : 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        pts.level[i] = (i < pts.n) ? s_tlev[grp][i] : 0;
        pts.loc[i] = (i < pts.n) ? s_tloc[grp][i] : 0;
    }
    const bool have = pts.n > 0;   // else "skip: no_curve" (atrac3denc.cpp:395-400)
    if (p.debug == 2) return;
    if (__ballot(have) == 0ull) {
        if (valid && j == 0) {
            Curve out;
            out.pad = 0;
            out.n = 0;
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                out.level[i] = 0;
                out.loc[i] = 0;
            }
            *dst = out;
        }
        return;
    }

    // ---- CreateSubbandInfo tail (atrac3denc.cpp:410-577), band < 3 ----
    {   // (the tables are staged only by the few wavefronts that get here)
        const int wv = tid >> 6, ln = tid & 63;
        const double l2v = (&T->log2f_tab[0][0])[ln < 36 ? ln : 35];   // (both asked for by every lane before either is stored: one round trip)
        const float giv = T->gain_interp[(ln & 31) < 31 ? (ln & 31) : 30];
        if (ln < 36) reinterpret_cast<double*>(&s_l2[wv])[ln] = l2v;
        if (ln < 32) s_gi4[wv][ln] = giv;
        wave_sync();
    }
    if (maxGain < 1e-4f) pts.n = 0;
    if (hfr < 0.3f) pts.n = 0;
    const CurvePts before = pts;
    bool changed = false;
    float hpfRmsNextMod = 0.0f;
    bool validMod = false;
    {
        const int nBefore = (pts.n > 0) ? pts.loc[0] : 32;
        float sum = 0.0f;
        for (int sf = 0; sf < 32; ++sf) {   // ordered sum of the first nBefore gains
            const float v = s_in[grp][sf];
            if (sf < nBefore) sum += v;
        }
        if (pts.n > 0 && pts.loc[0] > 0) {
            hpfRmsNextMod = (sum / (float)nBefore) / gain_level_of(pts.level[0]);
            validMod = true;
        } else if (pts.n == 0) {
            hpfRmsNextMod = sum / 32;
            validMod = true;
        }
    }
    const bool p0ok = validMod && prevTarget > 1e-6f && hpfRmsNextMod > 1e-6f;
    if (p0ok) {
        const int p0 = relation_to_idx_hdr(prevTarget / hpfRmsNextMod);
        int it = -1;
#pragma unroll
        for (int i = 6; i >= 0; --i)
            if (i < pts.n && pts.loc[i] == 0) it = i;
        if (it >= 0) {
#pragma unroll
            for (int i = 0; i < 7; ++i)
                if (i == it && pts.level[i] != p0) {
                    pts.level[i] = p0;
                    changed = true;
                }
        } else if (p0 != 4 || pts.n > 0) {
#pragma unroll
            for (int i = 6; i > 0; --i) {
                pts.level[i] = pts.level[i - 1];
                pts.loc[i] = pts.loc[i - 1];
            }
            pts.level[0] = p0;
            pts.loc[0] = 0;
            pts.n++;
            changed = true;
        }
    }
    if (p.debug == 3) return;
    // both scores are evaluated unconditionally (wave-uniform control flow) and TOGETHER; used only when `changed`
    float scoreBefore, scoreAfter;
    early_mismatch_score_pair(L2, gi, in_j, in_next, target, before, pts, s_tmp[grp], s_pts[grp], s_filt[grp], reinterpret_cast<float*>(s_tloc[grp]),
                              reinterpret_cast<float*>(s_tdelta[grp]), s_pts2[grp], j, scoreBefore, scoreAfter);
    if (p.debug == 5) { if (scoreBefore + scoreAfter == 12345.0f) *dst = Curve(); return; }
    if (changed) {
        bool keepByBoundary = false;
        if (p0ok) {
            const float x = prevTarget / hpfRmsNextMod;
            const float desired = fminf(fmaxf(x, gain_level_of(15)), gain_level_of(0));
            const float scaleBefore = gain_level_of(before.n == 0 ? 4 : before.level[0]);
            const float scaleAfter = gain_level_of(pts.n == 0 ? 4 : pts.level[0]);
            const float eps = 1e-9f;
            const float errBefore = fabsf(at3_log2f(L2, fmaxf(scaleBefore, eps) / fmaxf(desired, eps)));
            const float errAfter = fabsf(at3_log2f(L2, fmaxf(scaleAfter, eps) / fmaxf(desired, eps)));
            keepByBoundary = (errAfter + 0.20f < errBefore);
        }
        if (!keepByBoundary && scoreAfter > scoreBefore * (1.0f + 0.02f)) pts = before;
    }
    if (pts.n >= 2 && pts.loc[0] == 0 && pts.level[0] == pts.level[1]) {
#pragma unroll
        for (int i = 1; i < 7; ++i) {
            pts.level[i - 1] = pts.level[i];
            pts.loc[i - 1] = pts.loc[i];
        }
        pts.n--;
    }
    if (valid && j == 0) {
        Curve out;
        out.pad = 0;
        out.n = (uint8_t)(have ? pts.n : 0);
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const bool on = have && i < pts.n;
            out.level[i] = (uint8_t)(on ? pts.level[i] : 0);
            out.loc[i] = (uint8_t)(on ? pts.loc[i] : 0);
        }
        *dst = out;
    }
}

}  // namespace at3
