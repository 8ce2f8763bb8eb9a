// Back-end kernels: loudness / flatness / tonal extraction / scale factors, loudness tracking, and the helpers of the
// allocation + quantisation + packing kernel (at3_k_alloc.hpp).
//
// Reference path replaced (paths relative to the reference's src/):
//   atrac3denc.cpp:811-830                  loudness sum, flatness, ExtractTonalComponents, MapTonalComponents, ScaleFrame
//   atrac/atrac_psy_common.cpp:158-199      CalcSpectralFlatnessPerBfu
//   atrac/atrac_scale.cpp:40-188            QuantMantisas, TScaler::Scale / ScaleFrame
//   atrac/atrac_psy_common.h:46-54          TrackLoudness
//   atrac/at3/atrac3_bitstream.cpp:92-847   CLC/VLC cost + emission, CalcBitsAllocation, ConsiderEnergyErr,
//                                           tonal component grouping/coding, TConfigure/TAlloc, WriteSoundUnit
//   lib/bs_encode/encode.cpp:57-129         bisection driver (Start / Continue / Submit / Repeat)
//   lib/bitstream/bitstream.cpp:40-63       MSB-first bit writer
#pragma once
#include "at3_common.hpp"

namespace at3 {

struct BackParams {
    float* specs;            // [S][n_out][2][1024]; tonal lines are zeroed in place
    const float* ges;        // [S][n_blocks][2][4] by frame index, or null (all 1.0)
    const Curve* curves;     // [S][n_blocks][2][4] by frame index (zeros when gain control is off)
    PsyRec* psy;             // [S][n_out][2]
    float* loud;             // [S][n_out] tracked loudness per frame
    float* loud_state;       // [S]
    uint8_t* out;            // [S][n_out][frame_sz]
    int n_blocks;
    int f0;
    int n_streams;
    int no_tonal;
    int js;
    int frame_sz;
    int bfu_idx_const;
    int mono_js;             // one input channel in a joint-stereo container: empty second sound unit (atrac3denc.cpp:843-849)
    struct QuantRec* quant;  // [S][n_out][2] the unit cache's final content, for the QUANT tap; null unless AT3HIP_OPT_QUANT_TAP. cost is zero for units never asked
                             // for; err is zero for those AND for units of BFUs >= 10 whose cost the rate loop only ever needed as unit_bounds gives it (exact below
                             // BFU 19, where there is no energy-adaptive pass): compute_units, which forms e1 / e2, never ran for them (include/at3hip.h)
    int flat_literal;        // AT3HIP_OPT_LITERAL_FORMS: every flatness measure by the literal per-line form (test aid; same results)
    int debug_stop;          // profiling aid (env AT3HIP_DEBUG_STOP, -DAT3HIP_DEBUG_KNOBS builds): stage exits of k_alloc_pack
    unsigned long long* counters;   // [2] at3hip_get_counters: blocks TScaler::Scale would report as "Scale error", values it would report as
                                    // "clipping" (atrac_scale.cpp:150-153, 163-167); added to by k_psy, null = not counted
    int one_channel;         // one input channel (the pipeline runs on (x, x) pairs): only channel 0 is what the reference scales
    unsigned long long* clk; // [2] AT3HIP_TAP_CLOCK: shader cycles (s_memtime) and 100 MHz reference ticks (s_memrealtime) that
                             // workgroup 0 of k_alloc_pack lived - their ratio is the shader clock under the rate loop's load
};

struct QuantRec {            // per (stream, frame, channel)
    float err[7][32];        // e1 / e2 per (wordlen - 1, bfu)
    uint32_t cost[7][32];    // CLC bits | VLC bits << 13
};

// Smallest table index whose scale factor is >= maxAbs, 63 if there is none: std::map::lower_bound on the increasing
// ScaleTable (atrac_scale.cpp:150-160) as a six-step binary search; `scale_tab` should sit in LDS.
__device__ __forceinline__ int scale_index(const float* scale_tab, float maxAbs)
{
    int lo = 0, hi = 63;
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const int mid = (lo + hi) >> 1;
        if (scale_tab[mid] < maxAbs) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// atrac_scale.cpp:141-172. Returns sfi; values/energy optional.
__device__ inline int scale_block(const float* scale_tab, const float* in, int len, float* values, float* energy)
{
    float maxAbs = 0.0f;
    for (int i = 0; i < len; ++i) {
        const float a = fabsf(in[i]);
        if (a > maxAbs) maxAbs = a;
    }
    if (maxAbs > 1.0f) maxAbs = 1.0f;
    const int sfi = scale_index(scale_tab, maxAbs);
    const float sf = scale_tab[sfi];
    float e = 0.0f;
    for (int i = 0; i < len; ++i) {
        const float x = in[i];
        e += x * x;
        if (values) {
            float v = x / sf;
            if (fabsf(v) >= 1.0f) v = (v > 0) ? 0.99999f : -0.99999f;
            values[i] = v;
        }
    }
    if (energy) *energy = e;
    return sfi;
}

// ---- loudness of a channel-frame (atrac3denc.cpp:811-818) --------------------------------------------------------------
// l = sum over the 1024 lines, IN ORDER, of e * GainEnergyScale[band].Frame * LoudnessCurve[i]. The sum is a 1024-step
// chain whatever the hardware, so one lane runs one channel-frame's chain and a wavefront kLoudCf of them side by side
// (k_psy used to spend half its instructions on one lane per workgroup). A lone wavefront issues one instruction every
// ~5 cycles, so the chain wavefront does nothing but read terms (16 bytes at a time) and add them: three producer
// wavefronts of the workgroup fetch the spectra (coalesced 16-byte loads, two chunks in flight), form the terms and lay
// them out per channel-frame in LDS, one 64-line chunk ahead of the consumer.
// A channel-frame's 64 terms of a chunk are a row of kLoudRow floats: 16-byte aligned, and rows 4 banks apart, so that
// the sixteen lanes a 16-byte LDS read serves together hit sixteen different bank quads.
constexpr int kLoudLines = 128;                    // lines per chunk (64 at first: half as many workgroup rendezvous now, -3 us)
constexpr int kLoudRowWords = kLoudLines / 4;      // 16-byte words of a channel-frame's chunk
constexpr int kLoudRow = kLoudLines + 4;
constexpr int kLoudChunks = 1024 / kLoudLines;
constexpr int kLoudCf = 32;                        // channel-frames per workgroup
constexpr int kLoudWords = kLoudCf * kLoudRowWords;   // 16-byte words of a chunk
constexpr int kLoudPer = (kLoudWords + 191) / 192; // words per producer thread
static_assert(192 % kLoudRowWords == 0, "a producer thread keeps its place in the row from word to word");

// (free functions: array arguments of lambdas end up in scratch memory)
__device__ __forceinline__ void loud_request(const float* specs, int c0, int n_cf, int u, int k, float4 (&xr)[kLoudPer])
{
#pragma unroll
    for (int i = 0; i < kLoudPer; ++i) {
        const int w = u + 192 * i, c = c0 + w / kLoudRowWords;
        xr[i] = (w < kLoudWords && c < n_cf) ? *reinterpret_cast<const float4*>(specs + (size_t)c * 1024 + kLoudLines * k + 4 * (w % kLoudRowWords))
                                             : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
}
__device__ __forceinline__ void loud_produce(float* tile, const float* s_curve, const float* s_ges, int u, int k, const float4 (&xr)[kLoudPer])
{
    const float4 cv = *reinterpret_cast<const float4*>(s_curve + kLoudLines * k + 4 * (u % kLoudRowWords));   // (u + 192 i) % kLoudRowWords == u % kLoudRowWords
#pragma unroll
    for (int i = 0; i < kLoudPer; ++i) {
        const int w = u + 192 * i;
        if (w < kLoudWords) {
            const float gg = s_ges[4 * (w / kLoudRowWords) + (kLoudLines * k) / 256];
            const float4 x = xr[i];
            *reinterpret_cast<float4*>(tile + (w / kLoudRowWords) * kLoudRow + 4 * (w % kLoudRowWords)) =
                make_float4(x.x * x.x * gg * cv.x, x.y * x.y * gg * cv.y, x.z * x.z * gg * cv.z, x.w * x.w * gg * cv.w);
        }
    }
}
__device__ __forceinline__ void loud_consume(const float* row, float& l)
{
    const float4* r4 = reinterpret_cast<const float4*>(row);
#pragma unroll
    for (int q = 0; q < kLoudRowWords; q += 4) {
        const float4 a = r4[q], b = r4[q + 1], c = r4[q + 2], d = r4[q + 3];
        l += a.x; l += a.y; l += a.z; l += a.w;
        l += b.x; l += b.y; l += b.z; l += b.w;
        l += c.x; l += c.y; l += c.z; l += c.w;
        l += d.x; l += d.y; l += d.z; l += d.w;
    }
}

// Runs before k_psy, which removes the tonal lines from the spectra.
__global__ __launch_bounds__(256) void k_loud_sum(BackParams p, const Tables* T, int n_cf)
{
    __shared__ __attribute__((aligned(16))) float s_t[2][kLoudCf * kLoudRow];   // [buffer][channel-frame][line in chunk]
    __shared__ __attribute__((aligned(16))) float s_curve[1024];   // LoudnessCurve: a global load per chunk would stall the producers
    __shared__ __attribute__((aligned(16))) float s_ges[kLoudCf * 4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // (the curve, the band scales and the first two chunks of spectra are ONE round trip: every work-item asks for its curve words and,
    // at a clamped index, for a channel-frame's scales before anything is stored - written as load-store pairs under conditions they
    // were three round trips one after the other at the head of the workgroup's life)
    const float4 curve_v = *reinterpret_cast<const float4*>(T->loud_curve + 4 * tid);
    const int c0 = blockIdx.x * kLoudCf;
    const int n_out = p.n_blocks - p.f0;
    float4 ges_v;
    bool ges_ok;
    {
        const int cq = c0 + (tid < kLoudCf ? tid : 0);
        const int c = cq < n_cf ? cq : n_cf - 1;
        const int ch = c & 1, fo = (c >> 1) % n_out, s = (c >> 1) / n_out;
        const float* src = p.ges ? p.ges + ((size_t)s * p.n_blocks + fo + p.f0) * 8 + ch * 4 : T->loud_curve;
        ges_v = *reinterpret_cast<const float4*>(src);
        ges_ok = p.ges && cq < n_cf;
    }
    // producer role: thread u = tid - 64 of 192 forms the terms of the 16-byte words w = u + 192 i of a chunk (word w =
    // lines 4 (w % words per row) .. + 3 of channel-frame w / words per row). The spectra of chunk k + 2 are requested while
    // chunk k + 1 is turned into terms: a producer never waits for a load it has just issued.
    const int u = tid - 64;
    const bool chain = wave == 0 && lane < kLoudCf;
    float4 xa[kLoudPer], xb[kLoudPer];
    if (wave > 0) {
        loud_request(p.specs, c0, n_cf, u, 0, xa);
        loud_request(p.specs, c0, n_cf, u, 1, xb);
    }
    *reinterpret_cast<float4*>(s_curve + 4 * tid) = curve_v;
    if (tid < kLoudCf) *reinterpret_cast<float4*>(s_ges + 4 * tid) = ges_ok ? ges_v : make_float4(1.0f, 1.0f, 1.0f, 1.0f);   // the channel-frames' four band scales
    __syncthreads();   // the curve is in LDS
    if (wave > 0) {
        loud_produce(s_t[0], s_curve, s_ges, u, 0, xa);
        loud_request(p.specs, c0, n_cf, u, 2, xa);
    }
    __syncthreads();
    float l = 0.0f;
    for (int k = 0; k < kLoudChunks; k += 2) {   // chunk k + 1 comes from xb, chunk k + 2 from xa
        if (wave > 0) {
            loud_produce(s_t[(k + 1) & 1], s_curve, s_ges, u, k + 1, xb);
            if (k + 3 < kLoudChunks) loud_request(p.specs, c0, n_cf, u, k + 3, xb);
        } else if (chain) {
            loud_consume(s_t[k & 1] + lane * kLoudRow, l);
        }
        __syncthreads();
        if (wave > 0) {
            if (k + 2 < kLoudChunks) {
                loud_produce(s_t[k & 1], s_curve, s_ges, u, k + 2, xa);
                if (k + 4 < kLoudChunks) loud_request(p.specs, c0, n_cf, u, k + 4, xa);
            }
        } else if (chain) {
            loud_consume(s_t[(k + 1) & 1] + lane * kLoudRow, l);
        }
        __syncthreads();
    }
    if (chain && c0 + lane < n_cf) p.psy[c0 + lane].loud_ch = l;
}

// Flatness, tonal extraction and TScaler::Scale for kPsyCf channel-frames per workgroup. The per-BFU jobs (21 flatness
// measures, 32 scale/energy chains per channel-frame) are one lane each and cost as many steps as the BFU has lines, so
// the jobs of the four channel-frames are dealt to the wavefronts BY LENGTH: a wavefront's pass is as long as its
// longest job, and a pass over 64-line BFUs next to 16-line ones would idle most lanes most of the time.
constexpr int kPsyCf = 4;
__device__ __forceinline__ bool psy_flat_job(int wave, int lane, int& k, int& b)
{
    if (wave == 0) { k = lane / 3; b = 26 + lane % 3; return lane < 12; }      // 64 lines
    if (wave == 1) { k = lane / 10; b = 16 + lane % 10; return lane < 40; }    // 32 lines
    if (wave == 2) { k = lane / 8; b = 8 + lane % 8; return lane < 32; }       // 16 lines
    k = 0; b = 8;
    return false;
}
__device__ __forceinline__ bool psy_scale_job(int wave, int lane, int& k, int& b)
{
    if (wave == 0) { k = lane / 6; b = 26 + lane % 6; return lane < 24; }      // 64 and 128 lines
    if (wave == 1) { k = lane / 10; b = 16 + lane % 10; return lane < 40; }    // 32 lines
    if (wave == 2) { k = lane / 16; b = lane % 16; return true; }              // 8 and 16 lines
    k = 0; b = 0;
    return false;
}

// The literal form of CalcSpectralFlatnessPerBfu's geometric mean (atrac_psy_common.cpp:180-198) for ONE BFU, run by a
// whole wavefront: lane i takes line i's restated glibc log (at3_libm64.hpp; the table look-ups of all lines are in flight
// together), the reference's ordered sum walks the lanes' values as scalars, the restated exp closes it. Rarely needed
// (see k_psy): one lane doing it alone - up to 64 dependent table look-ups in global memory - held its workgroup for
// several times the kernel's normal duration. Wave-uniform call; `sp`, `len`, `arith` are uniform. Returns the flatness.
__device__ __attribute__((noinline)) float flatness_literal_wave(const Libm64* M, const float* sp, int len, double arith, int lane)
{
    const double floor_ = (double)1e-12f;
    double lg = 0.0;
    if (lane < len) {
        const float x = sp[lane];
        const double e = (double)fmaxf(0.0f, x * x);
        lg = at3_log(M, e > floor_ ? e : floor_);
    }
    const uint64_t lb = (uint64_t)__double_as_longlong(lg);
    const int lo = (int)(uint32_t)lb, hi = (int)(uint32_t)(lb >> 32);
    double ml = 0.0;
    for (int i = 0; i < len; ++i) {
        const uint64_t v = (uint64_t)(uint32_t)__builtin_amdgcn_readlane(lo, i) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(hi, i) << 32);
        ml += __longlong_as_double((long long)v);
    }
    ml /= (double)len;
    return (float)fmin(1.0, fmax(0.0, at3_exp(M, ml) / arith));
}

__global__ __launch_bounds__(256) AT3_WAVES_PER_EU(8) void k_psy(BackParams p, const Tables* T, int n_cf)
{
    __shared__ __attribute__((aligned(16))) float s_spec[kPsyCf][1024];
    __shared__ uint16_t s_run_start[kPsyCf][32];
    __shared__ uint8_t s_run_len[kPsyCf][32];
    __shared__ __attribute__((aligned(4))) uint16_t s_tv_pos[kPsyCf][112];   // after the tonal mapping: s_maxbits (20 KB in all: eight workgroups per CU, the batch's 2048 in one round)
    __shared__ float s_tv_val[kPsyCf][112];
    __shared__ uint8_t s_tv_bfu[kPsyCf][112];
    __shared__ float s_scale[64];                  // ScaleTable
    uint32_t (*s_maxbits)[32] = reinterpret_cast<uint32_t (*)[32]>(&s_tv_pos[0][0]);   // per BFU: max |x| as its bit pattern (ordered like the value for x >= 0); the tonal position list's storage, dead by then
    __shared__ int s_any[kPsyCf];                  // some BFU of the channel-frame has a tonal run
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c0 = blockIdx.x * kPsyCf;
    const int ncf = (n_cf - c0 < kPsyCf) ? n_cf - c0 : kPsyCf;
    float* specs0 = p.specs + (size_t)c0 * 1024;
    PsyRec* rec0 = p.psy + c0;

    // (every work-item asks for everything at once, at clamped indices: a load under a condition with a default value is waited for
    // where the paths join, and the ScaleTable request behind it was a second round trip before the first rendezvous)
    const float scale_v = T->scale[tid & 63];
    {
        float4 x4[kPsyCf];
#pragma unroll
        for (int k = 0; k < kPsyCf; ++k) x4[k] = *reinterpret_cast<const float4*>(specs0 + (size_t)(k < ncf ? k : 0) * 1024 + 4 * tid);
#pragma unroll
        for (int k = 0; k < kPsyCf; ++k) *reinterpret_cast<float4*>(s_spec[k] + 4 * tid) = (k < ncf) ? x4[k] : float4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    if (tid < 32 * kPsyCf) {
        (&s_run_len[0][0])[tid] = 0;
        const int k = tid >> 5, b = tid & 31;   // the PSY tap's flatness entries nobody measures
        if (k < ncf && (p.no_tonal || b < 8 || b > 28)) rec0[k].flat[b] = 0.0f;
    }
    if (tid < kPsyCf) s_any[tid] = 0;
    if (tid >= 128 && tid < 192) s_scale[tid - 128] = scale_v;
    __syncthreads();

    int fk, fb;
    const bool flat_job = !p.no_tonal && psy_flat_job(wave, lane, fk, fb) && fk < ncf;
    float flat = 1.0f;
    double arith = 0.0;
    bool want_literal = false;
    if (flat_job) {
        const int b = fb;
        const float* sp = s_spec[fk];
        const int start = bfu_start(b), end = bfu_start(b + 1), len = end - start;
        // CalcSpectralFlatnessPerBfu (atrac_psy_common.cpp:158-199): flat = exp(mean(log(max(e, floor)))) / mean(e) in f64
        // with glibc's log and exp, narrowed to f32. Two forms, same bits:
        //  * literal: one restated glibc log per line (at3_libm64.hpp), the reference's ordered sums, the restated exp;
        //  * short:   the logarithm of a product is the sum of the logarithms - the lines' f64 mantissas are multiplied
        //    (8 .. 64 factors in [0.5, 1), no underflow), their exponents added, ONE log per BFU closes the sum (a log per
        //    line was a third of this kernel). Its ratio differs from the reference's by rounding only: the reference's mean
        //    of <= 64 rounded logs (|log| <= 27.7, partial sums below 2048) is within 1.3e-13 of the exact mean, this form's
        //    (a product of <= 64 factors, one log, exponent sum x ln 2, all in f64) within 4e-14, the two exp / divisions add
        //    1e-15: relative distance below 2e-13. When BOTH ends of [ratio (1 - 1e-12), ratio (1 + 1e-12)] narrow to the
        //    same f32 after the clamp, that f32 is the reference's value; otherwise (about 3 in 100 000 BFUs) the literal
        //    form runs. Any device log / exp within a few hundred ulp serves the short form.
        double prod = 1.0;
        int esum = 0;
        const double floor_ = (double)1e-12f;
        for (int i0 = start; i0 < end; i0 += 8) {
            const float4 xa = *reinterpret_cast<const float4*>(sp + i0), xb = *reinterpret_cast<const float4*>(sp + i0 + 4);
            const float ev[8] = {xa.x * xa.x, xa.y * xa.y, xa.z * xa.z, xa.w * xa.w, xb.x * xb.x, xb.y * xb.y, xb.z * xb.z, xb.w * xb.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const double e = (double)fmaxf(0.0f, ev[k]);
                arith += e;
                const double d = e > floor_ ? e : floor_;
                const uint64_t bits = (uint64_t)__double_as_longlong(d);
                esum += (int)((bits >> 52) & 0x7ffu) - 1022;
                prod *= __longlong_as_double((long long)((bits & 0x800fffffffffffffull) | 0x3fe0000000000000ull));
            }
        }
        arith /= (double)len;
        if (!(arith <= floor_)) {
            const double meanLog = (log(prod) + (double)esum * 0.69314718055994530942) / (double)len;
            const double ratio = exp(meanLog) / arith;
            const float f_lo = (float)fmin(1.0, fmax(0.0, ratio * (1.0 - 1e-12)));
            const float f_hi = (float)fmin(1.0, fmax(0.0, ratio * (1.0 + 1e-12)));
            flat = f_lo;
            want_literal = f_lo != f_hi || p.flat_literal;
        }
    }
    // the literal form for the lanes that asked for it, one after the other, by the whole wavefront (uniform)
    for (unsigned long long rem = __ballot(want_literal); rem; rem &= rem - 1ull) {
        const int src = __builtin_ctzll(rem);
        const int jk = __builtin_amdgcn_readlane(fk, src), jb = __builtin_amdgcn_readlane(fb, src);
        const uint64_t ab = (uint64_t)__double_as_longlong(arith);
        const uint64_t as = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ab, src) |
                            ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(ab >> 32), src) << 32);
        const int jstart = bfu_start(jb);
        const float r = flatness_literal_wave(&T->libm, s_spec[jk] + jstart, bfu_start(jb + 1) - jstart, __longlong_as_double((long long)as), lane);
        if (lane == src) flat = r;
    }
    if (flat_job) {
        const int b = fb;
        rec0[fk].flat[b] = flat;
        if (flat < 0.01f) s_run_len[fk][b] = 1;   // a candidate of the ExtractTonalComponents search below
    }
    __syncthreads();

    // ExtractTonalComponents' search (atrac3denc.cpp:606-625): every window of one to five lines, first maximum wins (ascending start, then
    // ascending length). One lane per SIXTEEN STARTS of a candidate BFU - 160 jobs for the workgroup's 4 x 21 BFUs, sixteen steps each -, and a
    // lane per BFU that takes the first of its chunks' maxima (a later chunk wins only with a strictly larger score, as a later start does in
    // the reference's loop). One lane per BFU it was 64 steps of a 12-lane wavefront for the 64-line BFUs (`tones`: k_psy 35.6 -> 28.4 us;
    // rotating the long jobs over the workgroup's wavefronts instead changed nothing). The chunks' results use the tonal value lists' storage,
    // dead until the mapping.
    float* s_best_score = &s_tv_val[0][0];
    uint16_t* s_best_start = &s_tv_pos[0][0];
    uint8_t* s_best_len = &s_tv_bfu[0][0];
    if (tid < 160) {
        int k, b, chunk;
        if (tid < 48) { k = tid / 12; b = 26 + (tid % 12) / 4; chunk = tid % 4; }
        else if (tid < 128) { k = (tid - 48) / 20; b = 16 + ((tid - 48) % 20) / 2; chunk = (tid - 48) % 2; }
        else { k = (tid - 128) / 8; b = 8 + (tid - 128) % 8; chunk = 0; }
        if (k < ncf && s_run_len[k][b]) {
            const float* sp = s_spec[k];
            const int end = bfu_start(b + 1), cs = bfu_start(b) + 16 * chunk;
            // the five magnitudes a start needs are a shift register fed by one LDS read per start (cs + 4 < end: BFUs are whole chunks)
            float bestScore = -1.0f;
            int bestStart = cs, bestLen = 1;
            float a0 = fabsf(sp[cs]), a1 = fabsf(sp[cs + 1]), a2 = fabsf(sp[cs + 2]), a3 = fabsf(sp[cs + 3]), a4 = fabsf(sp[cs + 4]);
            for (int st = cs; st < cs + 16; ++st) {
                const int ml = 5 < end - st ? 5 : end - st;
                const float nxt = (st + 5 < end) ? fabsf(sp[st + 5]) : 0.0f;
                float score = 0.0f;
                score += a0;
                if (score > bestScore) { bestScore = score; bestStart = st; bestLen = 1; }
                score += a1;
                if (ml >= 2 && score > bestScore) { bestScore = score; bestStart = st; bestLen = 2; }
                score += a2;
                if (ml >= 3 && score > bestScore) { bestScore = score; bestStart = st; bestLen = 3; }
                score += a3;
                if (ml >= 4 && score > bestScore) { bestScore = score; bestStart = st; bestLen = 4; }
                score += a4;
                if (ml >= 5 && score > bestScore) { bestScore = score; bestStart = st; bestLen = 5; }
                a0 = a1; a1 = a2; a2 = a3; a3 = a4; a4 = nxt;
            }
            s_best_score[tid] = bestScore;
            s_best_start[tid] = (uint16_t)bestStart;
            s_best_len[tid] = (uint8_t)bestLen;
        }
    }
    __syncthreads();
    if (tid < 21 * kPsyCf) {
        const int k = tid / 21, b = 8 + tid % 21;
        if (k < ncf && s_run_len[k][b]) {
            const int job0 = b >= 26 ? k * 12 + (b - 26) * 4 : (b >= 16 ? 48 + k * 20 + (b - 16) * 2 : 128 + k * 8 + (b - 8));
            const int nchunks = b >= 26 ? 4 : (b >= 16 ? 2 : 1);
            float bestScore = s_best_score[job0];
            int best = job0;
            for (int c = 1; c < nchunks; ++c) {
                const float sc = s_best_score[job0 + c];
                if (sc > bestScore) { bestScore = sc; best = job0 + c; }
            }
            if (bestScore > 0.0f) {
                s_run_start[k][b] = s_best_start[best];
                s_run_len[k][b] = s_best_len[best];
                s_any[k] = 1;
            } else {
                s_run_len[k][b] = 0;
            }
        }
    }
    __syncthreads();

    static_assert(kPsyCf == 4, "one wavefront of the workgroup per channel-frame");
    if (wave < ncf) {   // one WAVEFRONT per channel-frame (wave-uniform). The reference's extraction and mapping are serial loops
        // (atrac3denc.cpp:627-662); as ONE lane per channel-frame they were two thirds of this kernel on tonal material (46 of 63 us on `tones`).
        // Their result has a closed form: BFU b's run goes to the list at the sum of the earlier runs' lengths; a component starts where a
        // position does not continue the one before it, and every seven positions after such a start.
        const int k0 = wave;
        PsyRec* rec = rec0 + k0;
        int n_comp = 0;
        if (s_any[k0]) {
            float* sp = s_spec[k0];
            float* specs = specs0 + (size_t)k0 * 1024;
            // lane l < 21 owns BFU 8 + l: its run's place in the list by an inclusive scan of the run lengths
            const int b = 8 + (lane < 21 ? lane : 20);
            const int rl = lane < 21 ? (int)s_run_len[k0][b] : 0;
            const int rs = (int)s_run_start[k0][b];
            int incl = rl;
#pragma unroll
            for (int d = 1; d < 32; d <<= 1) {
                const int t = __builtin_amdgcn_ds_bpermute(4 * (lane >= d ? lane - d : lane), incl);
                incl += (lane >= d) ? t : 0;
            }
            const int off = incl - rl;
            const int nv = __builtin_amdgcn_readlane(incl, 20);
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                if (k < rl) {
                    const int pos = rs + k;
                    s_tv_pos[k0][off + k] = (uint16_t)pos;
                    s_tv_val[k0][off + k] = sp[pos];
                    s_tv_bfu[k0][off + k] = (uint8_t)b;
                    sp[pos] = 0.0f;      // (runs of different BFUs never share a line)
                    specs[pos] = 0.0f;
                }
            }
            wave_sync();
            // MapTonalComponents (atrac3denc.cpp:646-662): runs of consecutive positions, at most 7 long. The list has at most 105 entries:
            // index i = lane (first half) and 64 + lane (second half)
            const int i0 = lane, i1 = 64 + lane;
            const int p0 = i0 < nv ? (int)s_tv_pos[k0][i0] : -1, p0m = (i0 > 0 && i0 - 1 < nv) ? (int)s_tv_pos[k0][i0 - 1] : -3;
            const int p1 = i1 < nv ? (int)s_tv_pos[k0][i1] : -1, p1m = (i1 - 1 < nv) ? (int)s_tv_pos[k0][i1 - 1] : -3;
            const unsigned long long B0 = __ballot(i0 < nv && (i0 == 0 || p0 != p0m + 1));   // a position that starts a stretch
            const unsigned long long B1 = __ballot(i1 < nv && p1 != p1m + 1);
            const unsigned long long upto = ~0ull >> (63 - lane);        // bits 0 .. lane
            const unsigned long long above = (lane < 63) ? (~0ull << (lane + 1)) : 0ull;   // bits lane + 1 .. 63
            const unsigned long long m0 = B0 & upto, m1 = B1 & upto;
            const int seg0 = 63 - __builtin_clzll(m0 | 1ull);                                       // (bit 0 of B0 is set whenever nv > 0)
            const int seg1 = m1 ? 64 + 63 - __builtin_clzll(m1) : 63 - __builtin_clzll(B0 | 1ull);
            const bool st0 = i0 < nv && (i0 - seg0) % 7 == 0, st1 = i1 < nv && (i1 - seg1) % 7 == 0;
            const unsigned long long S0 = __ballot(st0), S1 = __ballot(st1);                         // the components' first positions
            n_comp = __builtin_popcountll(S0) + __builtin_popcountll(S1);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (half && S1 == 0ull) break;   // (wave-uniform: most lists have fewer than 64 entries)
                if (half ? st1 : st0) {
                    const int startPos = half ? i1 : i0;
                    const int nb = half ? __builtin_popcountll(S0) + __builtin_popcountll(S1 & upto) - 1 : __builtin_popcountll(S0 & upto) - 1;
                    const unsigned long long nx = (half ? S1 : S0) & above;
                    const int next = nx ? (half ? 64 : 0) + __builtin_ctzll(nx) : ((!half && S1) ? 64 + __builtin_ctzll(S1) : nv);
                    const int len = next - startPos;
                    // TScaler::Scale on the component (atrac_scale.cpp:141-172; scale_block above) as straight-line code over seven slots - the
                    // slots behind `len` hold zeros, which change neither the maximum nor the counts -, the block assembled in registers and
                    // stored as five 8-byte words (it sits at 168 + 40 nb in its PsyRec)
                    float x[7];
#pragma unroll
                    for (int j = 0; j < 7; ++j) {
                        const float t = s_tv_val[k0][startPos + j];   // (startPos + 6 <= 110: inside the list's 112 slots)
                        x[j] = j < len ? t : 0.0f;
                    }
                    float maxAbs = 0.0f;
#pragma unroll
                    for (int j = 0; j < 7; ++j) maxAbs = fabsf(x[j]) > maxAbs ? fabsf(x[j]) : maxAbs;
                    int over = 0;
#pragma unroll
                    for (int j = 0; j < 7; ++j) over += fabsf(x[j]) > 1.0f;
                    if (nb < kMaxTonal) {
                        if (maxAbs > 1.0f) maxAbs = 1.0f;
                        const int sfi = scale_index(s_scale, maxAbs);
                        const float sf = s_scale[sfi];
                        uint32_t w[7];
#pragma unroll
                        for (int j = 0; j < 7; ++j) {
                            float v = x[j] / sf;
                            if (fabsf(v) >= 1.0f) v = (v > 0) ? 0.99999f : -0.99999f;
                            w[j] = j < len ? __float_as_uint(v) : 0u;
                        }
                        static_assert(offsetof(PsyRec, tonal) % 8 == 0 && sizeof(TonalBlock) % 8 == 0, "8-byte stores");
                        uint2* tb = reinterpret_cast<uint2*>(&rec->tonal[nb]);
                        tb[0] = uint2{(uint32_t)s_tv_pos[k0][startPos] | ((uint32_t)s_tv_bfu[k0][startPos] << 16) | ((uint32_t)len << 24), (uint32_t)sfi};
                        tb[1] = uint2{w[0], w[1]};
                        tb[2] = uint2{w[2], w[3]};
                        tb[3] = uint2{w[4], w[5]};
                        tb[4] = uint2{w[6], 0u};
                    }
                    // TScaler::Scale's diagnostics for this component (atrac_scale.cpp:150-153, 163-167): a block whose largest magnitude
                    // exceeds MAX_SCALE = 1.0 is scaled by ScaleTable[63] = 1.0, so its clipped values are those above 1.0. (The reference
                    // scales every component it maps, also those beyond the 24 a sound unit keeps.)
                    if (over && p.counters && !(p.one_channel && ((c0 + k0) & 1))) {
                        atomicAdd(p.counters, 1ull);
                        atomicAdd(p.counters + 1, (unsigned long long)over);
                    }
                }
            }
        }
        if (lane == 0) rec->n_tonal = n_comp < kMaxTonal ? n_comp : kMaxTonal;
    }
    __syncthreads();
    if (tid < 32 * kPsyCf) (&s_maxbits[0][0])[tid] = 0u;   // (the position list is dead: its storage holds the maxima now)
    __syncthreads();

    // TScaler::Scale per BFU (atrac_scale.cpp:141-172) on the residual spectrum: the maximum is order-free (all
    // work-items, LDS atomic max), the scale factor a binary search, the energy an ordered sum (one lane per BFU)
    {
        const int b = bfu_of_line(4 * tid);
#pragma unroll
        for (int k = 0; k < kPsyCf; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(s_spec[k] + 4 * tid);
            const float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
            atomicMax(&s_maxbits[k][b], __float_as_uint(m));
            // "clipping, scaled value": a value above MAX_SCALE = 1.0 sits in a block scaled by ScaleTable[63] = 1.0 (below)
            if (m > 1.0f && p.counters && k < ncf && !(p.one_channel && ((c0 + k) & 1)))
                atomicAdd(p.counters + 1, (unsigned long long)((fabsf(v.x) > 1.0f) + (fabsf(v.y) > 1.0f) + (fabsf(v.z) > 1.0f) + (fabsf(v.w) > 1.0f)));
        }
    }
    __syncthreads();
    int sk, sb;
    if (psy_scale_job(wave, lane, sk, sb) && sk < ncf) {
        const int start = bfu_start(sb), len = bfu_start(sb + 1) - start;
        float maxAbs = __uint_as_float(s_maxbits[sk][sb]);
        if (maxAbs > 1.0f) {   // "Scale error: absSpec > MAX_SCALE" (atrac_scale.cpp:150-153)
            if (p.counters && !(p.one_channel && ((c0 + sk) & 1))) atomicAdd(p.counters, 1ull);
            maxAbs = 1.0f;
        }
        const int sfi = scale_index(s_scale, maxAbs);
        const float4* x4 = reinterpret_cast<const float4*>(s_spec[sk] + start);
        float e = 0.0f;
        for (int i = 0; i < len / 8; ++i) {
            const float4 a = x4[2 * i], b = x4[2 * i + 1];
            e += a.x * a.x;
            e += a.y * a.y;
            e += a.z * a.z;
            e += a.w * a.w;
            e += b.x * b.x;
            e += b.y * b.y;
            e += b.z * b.z;
            e += b.w * b.w;
        }
        rec0[sk].sfi[sb] = (uint8_t)sfi;
        rec0[sk].energy[sb] = e;
    }
}

// TrackLoudness chain (atrac3denc.cpp:833-841, atrac_psy_common.h:46-54): one wavefront per stream. The 64 lanes
// fetch the per-frame channel loudness values in parallel (the chain itself would otherwise wait for one
// dependent global load per frame), lane 0 runs the f64 recurrence out of LDS, all lanes store the results.
constexpr int kLoudChunk = 1024;
__global__ __launch_bounds__(64) void k_loudness(BackParams p)
{
    __shared__ float s_t[kLoudChunk];
    const int s = blockIdx.x;
    const int lane = threadIdx.x;
    const int n_out = p.n_blocks - p.f0;
    float L = p.loud_state[s];
    for (int base = 0; base < n_out; base += kLoudChunk) {
        const int cnt = (n_out - base < kLoudChunk) ? n_out - base : kLoudChunk;
        for (int i = lane; i < cnt; i += 64) {
            const PsyRec* r = p.psy + ((size_t)s * n_out + base + i) * 2;
            const float l0 = r[0].loud_ch, l1 = r[1].loud_ch;
            s_t[i] = p.js ? l0 : (l0 + l1);
        }
        __syncthreads();
        if (lane == 0) {
            const double k = p.js ? 0.02 : 0.01;
            for (int i = 0; i < cnt; ++i) {
                L = (float)(0.98 * (double)L + k * (double)s_t[i]);
                s_t[i] = L;
            }
        }
        __syncthreads();
        for (int i = lane; i < cnt; i += 64) p.loud[(size_t)s * n_out + base + i] = s_t[i];
        __syncthreads();
    }
    if (lane == 0) p.loud_state[s] = L;
}

// ---- bit writer on an LDS word buffer (MSB first) -------------------------------------------------
constexpr int kBitWords = 320;  // 10240 bits: upper bound of one channel's sound unit before truncation

__device__ inline void put_bits(uint32_t* words, int pos, uint32_t val, int n)
{
    // n in 1..23. Bits beyond the buffer are dropped (they are truncated away by the frame size anyway).
    if (n <= 0 || pos + n > kBitWords * 32) return;
    val &= (n >= 32) ? 0xffffffffu : ((1u << n) - 1u);
    const int w = pos >> 5, off = pos & 31;
    const int room = 32 - off;
    if (n <= room) {
        atomicOr(&words[w], val << (room - n));
    } else {
        atomicOr(&words[w], val >> (n - room));
        atomicOr(&words[w + 1], val << (32 - (n - room)));
    }
}

__device__ inline uint32_t huff_entry(int sel, uint32_t idx) { return c_huff[huff_off(sel) + idx]; }

__device__ inline uint32_t vlc_index(int m)
{
    uint32_t h = (m < 0) ? (((uint32_t)(-m)) << 1) | 1u : ((uint32_t)m) << 1;
    if (h) h -= 1;
    return h;
}

// libstdc++ std::sort order for (key, idx) pairs compared by |key| (see oracle/at3_oracle.c std_sort_abs
// for the derivation; QuantMantisas at atrac_scale.cpp:79-83 depends on this order for equal keys).
struct SortItem {
    float key;
    int idx;
};
__device__ inline bool sless(const SortItem& a, const SortItem& b) { return fabsf(a.key) < fabsf(b.key); }

__device__ inline void s_push_heap(SortItem* first, int hole, int top, SortItem value)
{
    int parent = (hole - 1) / 2;
    while (hole > top && sless(first[parent], value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

__device__ inline void s_adjust_heap(SortItem* first, int hole, int len, SortItem value)
{
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (sless(first[child], first[child - 1])) child--;
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

__device__ inline void s_heap_sort(SortItem* first, int len)
{
    if (len >= 2) {
        int parent = (len - 2) / 2;
        for (;;) {
            const SortItem v = first[parent];
            s_adjust_heap(first, parent, len, v);
            if (parent == 0) break;
            parent--;
        }
    }
    while (len > 1) {
        --len;
        const SortItem v = first[len];
        first[len] = first[0];
        s_adjust_heap(first, 0, len, v);
    }
}

__device__ inline void s_unguarded_linear_insert(SortItem* a, int last)
{
    const SortItem v = a[last];
    int next = last - 1;
    while (sless(v, a[next])) {
        a[last] = a[next];
        last = next;
        --next;
    }
    a[last] = v;
}

__device__ inline void s_insertion_sort(SortItem* a, int first, int last)
{
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        if (sless(a[i], a[first])) {
            const SortItem v = a[i];
            for (int j = i; j > first; --j) a[j] = a[j - 1];
            a[first] = v;
        } else {
            s_unguarded_linear_insert(a, i);
        }
    }
}

// `stk`: 72 ints of scratch (LDS), the explicit stack of the introsort loop.
__device__ __attribute__((noinline)) void std_sort_abs(SortItem* a, int n, int* stk)
{
    if (n <= 0) return;
    int lg = 0;
    for (int t = n; t > 1; t >>= 1) ++lg;
    // introsort loop with an explicit stack instead of recursion on the right partition
    int *stk_first = stk, *stk_last = stk + 24, *stk_depth = stk + 48;
    int sp = 0;
    stk_first[0] = 0; stk_last[0] = n; stk_depth[0] = 2 * lg; sp = 1;
    while (sp > 0) {
        --sp;
        int first = stk_first[sp], last = stk_last[sp], depth = stk_depth[sp];
        // The reference recursion handles [cut,last) first (recursively, completely) and then loops on
        // [first,cut). Partitions are disjoint, so the processing order does not affect the result.
        while (last - first > 16) {
            if (depth == 0) {
                s_heap_sort(a + first, last - first);
                break;
            }
            --depth;
            const int mid = first + (last - first) / 2;
            const int ia = first + 1, ib = mid, ic = last - 1;
            int pick;
            if (sless(a[ia], a[ib])) {
                if (sless(a[ib], a[ic])) pick = ib;
                else if (sless(a[ia], a[ic])) pick = ic;
                else pick = ia;
            } else if (sless(a[ia], a[ic])) pick = ia;
            else if (sless(a[ib], a[ic])) pick = ic;
            else pick = ib;
            {
                const SortItem t = a[first];
                a[first] = a[pick];
                a[pick] = t;
            }
            int lo = first + 1, hi = last;
            for (;;) {
                while (sless(a[lo], a[first])) ++lo;
                --hi;
                while (sless(a[first], a[hi])) --hi;
                if (!(lo < hi)) break;
                const SortItem t = a[lo];
                a[lo] = a[hi];
                a[hi] = t;
                ++lo;
            }
            if (sp < 24) {
                stk_first[sp] = lo; stk_last[sp] = last; stk_depth[sp] = depth;
                ++sp;
            }
            last = lo;
        }
    }
    if (n > 16) {
        s_insertion_sort(a, 0, 16);
        for (int i = 16; i < n; ++i) s_unguarded_linear_insert(a, i);
    } else {
        s_insertion_sort(a, 0, n);
    }
}

// Tonal component side information: grouping (GroupTonalComponents, atrac3_bitstream.cpp:338-380) and
// cost / emission (EncodeTonalComponents :382-524). Serial (one lane). With EMIT the bits go to `words`
// starting at bit `pos`; returns the number of bits.
// `scr`: 5 * kMaxTonal + 24 bytes of scratch (LDS): the serial walk's small arrays (private arrays would live in scratch memory).
template <bool EMIT>
__device__ __attribute__((noinline)) int tonal_encode(const PsyRec* rec, const uint8_t* tbits /* [kMaxTonal][6]: quantisers 2..7 */, const uint8_t* alloc,
                                   int n_alloc, uint32_t* words, int pos, uint8_t* scr)
{
    const int nt = rec->n_tonal;
    uint8_t* grp_of = scr;                    // [kMaxTonal]
    uint8_t* mem = scr + kMaxTonal;           // [kMaxTonal]
    uint8_t* sg_start = scr + 2 * kMaxTonal;  // [kMaxTonal]
    uint8_t* cnt = scr + 3 * kMaxTonal;       // [16]
    int tcsgn = 0;
    for (int t = 0; t < nt; ++t) {
        const int bfu = rec->tonal[t].bfu;
        if (bfu >= n_alloc) {
            grp_of[t] = 0xff;
            continue;
        }
        int quant = (int)alloc[bfu] + 4;
        if (quant > 7) quant = 7;
        if (quant < 2) quant = 2;
        grp_of[t] = (uint8_t)(quant * 8 + rec->tonal[t].len);
    }
    // first pass: count sub-groups (needed up front for the 5-bit header)
    for (int g = 16; g < 64; ++g) {
        int nm = 0;
        for (int t = 0; t < nt; ++t)
            if (grp_of[t] == g) mem[nm++] = (uint8_t)t;
        int cur = 0;
        while (cur < nm) {
            int start = cur;
            ++tcsgn;
            int limiter = 0;
            do {
                ++cur;
                if (cur == nm) break;
                if ((int)rec->tonal[mem[cur]].pos - (int)(rec->tonal[mem[start]].pos & ~63) < 64) {
                    ++limiter;
                } else {
                    limiter = 0;
                    start = cur;
                }
            } while (limiter < 7);
        }
    }
    int used = 5;
    if (EMIT) put_bits(words, pos, (uint32_t)tcsgn, 5);
    if (tcsgn == 0) return used;
    if (EMIT) put_bits(words, pos + used, 0, 2);
    used += 2;
    for (int g = 16; g < 64; ++g) {
        int nm = 0;
        for (int t = 0; t < nt; ++t)
            if (grp_of[t] == g) mem[nm++] = (uint8_t)t;
        if (nm == 0) continue;
        // sub-group boundaries
        int nsg = 0;
        {
            int cur = 0;
            while (cur < nm) {
                int start = cur;
                sg_start[nsg++] = (uint8_t)cur;
                int limiter = 0;
                do {
                    ++cur;
                    if (cur == nm) break;
                    if ((int)rec->tonal[mem[cur]].pos - (int)(rec->tonal[mem[start]].pos & ~63) < 64) {
                        ++limiter;
                    } else {
                        limiter = 0;
                        start = cur;
                    }
                } while (limiter < 7);
            }
        }
        const int q = g >> 3;
        const int codedValues = rec->tonal[mem[0]].len;
        for (int sg = 0; sg < nsg; ++sg) {
            const int sgStart = sg_start[sg];
            const int sgEnd = (sg < nsg - 1) ? sg_start[sg + 1] : nm;
            for (int j = 0; j < 16; ++j) cnt[j] = 0;
            for (int j = sgStart; j < sgEnd; ++j) cnt[rec->tonal[mem[j]].pos >> 6]++;
            if (EMIT)
                for (int b = 0; b < 4; ++b) put_bits(words, pos + used + b, (cnt[4 * b] | cnt[4 * b + 1] | cnt[4 * b + 2] | cnt[4 * b + 3]) != 0, 1);
            used += 4;
            if (EMIT) put_bits(words, pos + used, (uint32_t)codedValues - 1, 3);
            used += 3;
            if (EMIT) put_bits(words, pos + used, (uint32_t)q, 3);
            used += 3;
            int lastPos = sgStart;
            for (int j = 0; j < 16; ++j) {
                const int b4 = j & ~3;
                if (!(cnt[b4] | cnt[b4 + 1] | cnt[b4 + 2] | cnt[b4 + 3])) continue;
                const int coded = cnt[j];
                if (EMIT) put_bits(words, pos + used, (uint32_t)coded, 3);
                used += 3;
                int k = lastPos;
                for (; k < lastPos + coded; ++k) {
                    const TonalBlock& tb = rec->tonal[mem[k]];
                    if (EMIT) {
                        put_bits(words, pos + used, tb.sfi, 6);
                        put_bits(words, pos + used + 6, (uint32_t)tb.pos - (uint32_t)j * 64, 6);
                        int bp = pos + used + 12;
                        const float mul = max_quant(q < 7 ? q : 7);
                        for (int z = 0; z < tb.len; ++z) {
                            const int m = __float2int_rn(tb.values[z] * mul);
                            const uint32_t e = huff_entry(q, vlc_index(m));
                            put_bits(words, bp, e & 0xffu, (int)(e >> 8));
                            bp += (int)(e >> 8);
                        }
                    }
                    used += 12 + tbits[mem[k] * 6 + q - 2];
                }
                lastPos = k;
            }
        }
    }
    return used;
}

constexpr int kEaLine0 = 288;              // first spectral line of BFU 19: the energy-adaptive pass of QuantMantisas starts here
constexpr int kEaLines = 1024 - kEaLine0;  // 736

__device__ __forceinline__ uint32_t lds_huff(const uint16_t* s_huff, int sel, uint32_t idx)
{
    return s_huff[huff_off(sel) + idx];
}

// End-of-call state hand-over: PCM history and the last frame's gain curves.
struct StateParams {
    const float* pcm;       // [S][n_blocks][1024][2]
    const float* hist_in;   // [S][kHist][2]
    float* hist_out;        // [S][kHist][2]
    const Curve* curves;    // [S][n_blocks][2][4]
    BandState* state;
    const float* sub;       // [S][8][(n_blocks+2)*256] subbands of this call, or null (no gain control, discrete stereo)
    float* sub_tail;        // [S][8][512]: the last two blocks' subbands, the next call's blocks -2 and -1
    int n_blocks;
    int n_streams;
    int parts;              // 1 = PCM history and subband tail (known once the QMF has run), 2 = last curves, 3 = both
};

// 16-bit PCM (at3hip_encode_s16): x = s / 32768.0f, what libsndfile's sf_readf_float hands the reference for a 16-bit WAV
// (pcm_io_sndfile.cpp:111-113 -> TPCMEngine -> the lambda). 2^-15 is a power of two: the product IS the quotient. Eight samples
// (16 bytes in, 32 out) per work-item; n8 = samples / 8 (a block is 1024 samples per channel, so the count divides).
__global__ __launch_bounds__(256) void k_s16_to_f32(const int16_t* __restrict__ in, float* __restrict__ out, size_t n8)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n8) {
        const uint4 v = reinterpret_cast<const uint4*>(in)[i];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        float f[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f[2 * k] = (float)(int16_t)(w[k] & 0xffffu) * 0x1p-15f;
            f[2 * k + 1] = (float)(int16_t)(w[k] >> 16) * 0x1p-15f;
        }
        reinterpret_cast<float4*>(out)[2 * i] = make_float4(f[0], f[1], f[2], f[3]);
        reinterpret_cast<float4*>(out)[2 * i + 1] = make_float4(f[4], f[5], f[6], f[7]);
    }
}

// One-channel input: every sample becomes an (L, R) = (x, x) pair in the layout the stereo pipeline reads.
__global__ __launch_bounds__(256) void k_mono_to_pairs(const float* __restrict__ mono, float* __restrict__ pairs, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const float x = mono[i];
        float2 v;
        v.x = x;
        v.y = x;
        reinterpret_cast<float2*>(pairs)[i] = v;
    }
}

__global__ void k_state_update(StateParams p)
{
    constexpr int kChunks = (kHist + 255) / 256;
    const int s = blockIdx.x / kChunks;
    const int k = (blockIdx.x % kChunks) * blockDim.x + threadIdx.x;
    if ((p.parts & 1) && k < kHist) {
        const float2* pcm2 = reinterpret_cast<const float2*>(p.pcm) + (size_t)s * p.n_blocks * 1024;
        const float2* hin = reinterpret_cast<const float2*>(p.hist_in) + (size_t)s * kHist;
        float2* hout = reinterpret_cast<float2*>(p.hist_out) + (size_t)s * kHist;
        const int g = p.n_blocks * 1024 - kHist + k;
        hout[k] = (g >= 0) ? pcm2[g] : hin[kHist + g];
    }
    if ((p.parts & 2) && k < 8) p.state[(size_t)s * 8 + k].prev_curve = p.curves[((size_t)s * p.n_blocks + (p.n_blocks - 1)) * 8 + k];
    if ((p.parts & 1) && p.sub && k < 8 * 128) {   // 8 rows x 512 floats as 16-byte words: the tail of every row (its front is the carried part)
        const int row = k >> 7, i = k & 127;
        const float4* src = reinterpret_cast<const float4*>(p.sub + ((size_t)s * 8 + row) * ((size_t)(p.n_blocks + 2) * 256) + (size_t)p.n_blocks * 256);
        reinterpret_cast<float4*>(p.sub_tail + ((size_t)s * 8 + row) * 512)[i] = src[i];
    }
}

}  // namespace at3
