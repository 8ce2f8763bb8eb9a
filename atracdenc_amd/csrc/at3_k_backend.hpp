// Back-end kernels: loudness / flatness / tonal extraction / scale factors, loudness tracking,
// bit allocation + mantissa quantisation + sound-unit packing.
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
    struct QuantRec* quant;  // [S][n_out][2] per-unit tables written by k_quant, read by k_rate_pack
    int8_t* mant;            // [S][n_out][2][7][1024] mantissas for every wordlen
    int debug_stop;          // profiling aid (env AT3HIP_DEBUG_STOP): leave k_quant after phase N; 0 = run everything
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

// One workgroup per (stream, output frame, channel).
__global__ __launch_bounds__(256) void k_psy(BackParams p, const Tables* T)
{
    __shared__ __attribute__((aligned(16))) float s_spec[1024];
    __shared__ __attribute__((aligned(16))) float s_e[1024];
    __shared__ __attribute__((aligned(16))) float s_term[1024];
    __shared__ int s_run_start[32];
    __shared__ int s_run_len[32];
    __shared__ uint16_t s_tv_pos[112];
    __shared__ float s_tv_val[112];
    __shared__ uint8_t s_tv_bfu[112];
    __shared__ float s_scale[64];          // ScaleTable
    __shared__ uint32_t s_maxbits[32];     // per BFU: max |x| as its bit pattern (ordered like the value for x >= 0)
    const int tid = threadIdx.x;
    const int n_out = p.n_blocks - p.f0;
    const int ch = blockIdx.x & 1;
    const int fo = (blockIdx.x >> 1) % n_out;
    const int s = (blockIdx.x >> 1) / n_out;
    const int f = fo + p.f0;
    float* specs = p.specs + (((size_t)s * n_out + fo) * 2 + ch) * 1024;
    PsyRec* rec = p.psy + ((size_t)s * n_out + fo) * 2 + ch;

    // energies, loudness terms (e * GainEnergyScale.Frame * LoudnessCurve, atrac3denc.cpp:811-818) and - for the
    // flatness measure - log(max(e, floor)) are produced by all work-items; only the additions are ordered.
    {
        float g[4] = {1.0f, 1.0f, 1.0f, 1.0f};
        if (p.ges)
            for (int b = 0; b < 4; ++b) g[b] = p.ges[((size_t)s * p.n_blocks + f) * 8 + ch * 4 + b];
        const float4 x4 = *reinterpret_cast<const float4*>(specs + 4 * tid);
        const float4 c4 = *reinterpret_cast<const float4*>(T->loud_curve + 4 * tid);
        const float gg = g[tid >> 6];
        float4 e4, t4;
        e4.x = x4.x * x4.x; e4.y = x4.y * x4.y; e4.z = x4.z * x4.z; e4.w = x4.w * x4.w;
        t4.x = e4.x * gg * c4.x; t4.y = e4.y * gg * c4.y; t4.z = e4.z * gg * c4.z; t4.w = e4.w * gg * c4.w;
        *reinterpret_cast<float4*>(s_spec + 4 * tid) = x4;
        *reinterpret_cast<float4*>(s_e + 4 * tid) = e4;
        *reinterpret_cast<float4*>(s_term + 4 * tid) = t4;
    }
    if (tid < 32) {
        s_run_len[tid] = 0;
        s_maxbits[tid] = 0u;
    }
    if (tid >= 64 && tid < 128) s_scale[tid - 64] = T->scale[tid - 64];
    __syncthreads();

    if (tid == 192) {
        // loudness: strictly sequential 1024-term sum, on its own wavefront, at raised issue priority: the workgroup's
        // lifetime is this chain, and a dependent add that has to queue behind seven other wavefronts costs 8x
        __builtin_amdgcn_s_setprio(3);
        const float4* t4 = reinterpret_cast<const float4*>(s_term);
        float l = 0.0f;
        float4 cur = t4[0];
        for (int i = 0; i < 256; ++i) {
            float4 nxt = cur;
            if (i + 1 < 256) nxt = t4[i + 1];
            l += cur.x;
            l += cur.y;
            l += cur.z;
            l += cur.w;
            cur = nxt;
        }
        rec->loud_ch = l;
        __builtin_amdgcn_s_setprio(0);
    }

    if (!p.no_tonal && tid >= 8 && tid < 29) {
        const int b = tid;
        const int start = bfu_start(b), end = bfu_start(b + 1), len = end - start;
        // CalcSpectralFlatnessPerBfu (atrac_psy_common.cpp:158-199) needs mean(log(max(e, floor))). The logarithm of a
        // product is the sum of the logarithms: the lines' f64 mantissas are multiplied (8 .. 64 factors in [0.5, 1),
        // no underflow), their exponents added, and ONE log per BFU closes the sum - instead of one f64 log per
        // spectral line, which was a third of this kernel. The result differs from the reference's sum of rounded
        // logs by a few 1e-16 relative; it is narrowed to f32 and only compared with 0.01.
        double arith = 0.0, prod = 1.0;
        int esum = 0;
        const double floor_ = (double)1e-12f;
        for (int i0 = start; i0 < end; i0 += 8) {
            const float4 ea = *reinterpret_cast<const float4*>(s_e + i0), eb = *reinterpret_cast<const float4*>(s_e + i0 + 4);
            const float ev[8] = {ea.x, ea.y, ea.z, ea.w, eb.x, eb.y, eb.z, eb.w};
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
        const double meanLog = (log(prod) + (double)esum * 0.69314718055994530942) / (double)len;
        float flat = 1.0f;
        if (!(arith <= (double)1e-12f)) {
            const double ratio = exp(meanLog) / arith;
            flat = (float)fmin(1.0, fmax(0.0, ratio));
        }
        if (flat < 0.01f) {  // ExtractTonalComponents search, atrac3denc.cpp:606-625
            const int maxLen = 5 < len ? 5 : len;
            float bestScore = -1.0f;
            int bestStart = start, bestLen = 1;
            for (int st = start; st < end; ++st) {
                const int ml = maxLen < end - st ? maxLen : end - st;
                float score = 0.0f;
                for (int l = 1; l <= ml; ++l) {
                    score += fabsf(s_spec[st + l - 1]);
                    if (score > bestScore) {
                        bestScore = score;
                        bestStart = st;
                        bestLen = l;
                    }
                }
            }
            if (bestScore > 0.0f) {
                s_run_start[b] = bestStart;
                s_run_len[b] = bestLen;
            }
        }
    }
    __syncthreads();

    if (tid == 0) {
        int nv = 0;
        for (int b = 8; b < 29; ++b) {
            for (int k = 0; k < s_run_len[b]; ++k) {
                const int pos = s_run_start[b] + k;
                s_tv_pos[nv] = (uint16_t)pos;
                s_tv_val[nv] = s_spec[pos];
                s_tv_bfu[nv] = (uint8_t)b;
                ++nv;
                s_spec[pos] = 0.0f;
                specs[pos] = 0.0f;
            }
        }
        // MapTonalComponents (atrac3denc.cpp:646-662): runs of consecutive positions, at most 7 long
        int nb = 0;
        for (int i = 0; i < nv;) {
            const int startPos = i;
            int curPos;
            do {
                curPos = s_tv_pos[i];
                ++i;
            } while (i < nv && s_tv_pos[i] == curPos + 1 && i - startPos < 7);
            const int len = i - startPos;
            TonalBlock tb;
            tb.pos = s_tv_pos[startPos];
            tb.bfu = s_tv_bfu[startPos];
            tb.len = (uint8_t)len;
            for (int j = 0; j < 7; ++j) tb.values[j] = 0.0f;
            for (int j = 0; j < 3; ++j) tb.pad[j] = 0;
            for (int j = 0; j < 4; ++j) tb.pad2[j] = 0;
            tb.sfi = (uint8_t)scale_block(s_scale, s_tv_val + startPos, len, tb.values, nullptr);
            if (nb < kMaxTonal) rec->tonal[nb] = tb;
            ++nb;
        }
        rec->n_tonal = nb < kMaxTonal ? nb : kMaxTonal;
    }
    __syncthreads();

    // TScaler::Scale per BFU (atrac_scale.cpp:141-172) on the residual spectrum: the maximum is order-free (all
    // work-items, LDS atomic max), the scale factor a binary search, the energy an ordered sum (one lane per BFU)
    {
        const float4 v = *reinterpret_cast<const float4*>(s_spec + 4 * tid);
        const float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
        atomicMax(&s_maxbits[bfu_of_line(4 * tid)], __float_as_uint(m));
    }
    __syncthreads();
    if (tid < 32) {
        const int start = bfu_start(tid), len = bfu_start(tid + 1) - start;
        float maxAbs = __uint_as_float(s_maxbits[tid]);
        if (maxAbs > 1.0f) maxAbs = 1.0f;
        const int sfi = scale_index(s_scale, maxAbs);
        const float4* x4 = reinterpret_cast<const float4*>(s_spec + start);
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
        rec->sfi[tid] = (uint8_t)sfi;
        rec->energy[tid] = e;
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

// (code | len << 16) for element i of a BFU quantised with selector wl. Pair-coded selectors (wl == 1)
// put the pair on the even element.
__device__ inline uint32_t spec_code(int wl, bool clc, const int8_t* m, int i)
{
    if (wl > 1) {
        if (clc) {
            const int nb = clc_len(wl);
            return ((uint32_t)m[i] & ((1u << nb) - 1u)) | ((uint32_t)nb << 16);
        }
        const uint32_t e = huff_entry(wl, vlc_index(m[i]));
        return (e & 0xffu) | ((e >> 8) << 16);
    }
    if (i & 1) return 0;
    if (clc) {
        const uint32_t rt[4] = {2, 3, 0, 1};
        const uint32_t code = (rt[m[i] + 2] << 2) | rt[m[i + 1] + 2];
        return code | (4u << 16);
    }
    const uint32_t rt9[9] = {8, 4, 7, 2, 0, 1, 6, 3, 5};
    const uint32_t e = huff_entry(1, rt9[3 * (m[i] + 1) + (m[i + 1] + 1)]);
    return (e & 0xffu) | ((e >> 8) << 16);
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

__device__ __attribute__((noinline)) void std_sort_abs(SortItem* a, int n)
{
    if (n <= 0) return;
    int lg = 0;
    for (int t = n; t > 1; t >>= 1) ++lg;
    // introsort loop with an explicit stack instead of recursion on the right partition
    int stk_first[24], stk_last[24], stk_depth[24];
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

// Energy-adaptive re-rounding pass of QuantMantisas (atrac_scale.cpp:86-128) over the candidates already
// ordered by |delta| (sidx[c] = line inside the BFU; the list is padded to a multiple of four). Only candidates
// that pass the side test of the running pass are listed, and each line occurs once, so the current mantissa of
// a candidate is still lrint(t). Values of four candidates are fetched together; the decisions stay sequential.
__device__ inline float ea_greedy(const float* in, const uint8_t* sidx, int nc, float mul, float inv2, float e1, float e2,
                                  int8_t* mant)
{
    // Only |mantissa| enters the energy bookkeeping: the re-rounded code is |m0| + 1 (e2 < e1; a zero becomes +-1) or
    // |m0| - 1 (e2 > e1), atrac_scale.cpp:86-118. The ordered part per candidate is ex = (e2 - d0) + d1 and the test.
    const bool grow = e2 < e1;
    float dist = fabsf(e2 - e1);
    // the list and the mantissas of the next four candidates are fetched while the current four run through the
    // ordered test (their lines are distinct, so the stores of accepted candidates cannot touch them)
    uint32_t i4 = *reinterpret_cast<const uint32_t*>(sidx);
    int m0s[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) m0s[k] = mant[(i4 >> (8 * k)) & 0xff];
    for (int c0 = 0; c0 < nc; c0 += 4) {
        uint32_t i4n = 0;
        int m0n[4] = {0, 0, 0, 0};
        if (c0 + 4 < nc) {
            i4n = *reinterpret_cast<const uint32_t*>(sidx + c0 + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) m0n[k] = mant[(i4n >> (8 * k)) & 0xff];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (c0 + k < nc) {
                const int m0 = m0s[k];
                const int a0 = m0 < 0 ? -m0 : m0;
                const int a1 = grow ? a0 + 1 : (a0 > 0 ? a0 - 1 : 0);
                float ex = e2;
                ex -= (float)(a0 * a0) * inv2;
                ex += (float)(a1 * a1) * inv2;
                const float nd = fabsf(ex - e1);
                if (nd < dist) {
                    const int idx = (i4 >> (8 * k)) & 0xff;
                    int m;
                    if (m0 > 0) m = a1;
                    else if (m0 < 0) m = -a1;
                    else m = (in[idx] * mul > 0) ? 1 : -1;   // only reached when growing
                    mant[idx] = (int8_t)m;
                    e2 = ex;
                    dist = nd;
                }
            }
        }
        i4 = i4n;
#pragma unroll
        for (int k = 0; k < 4; ++k) m0s[k] = m0n[k];
    }
    return e2;
}

// VLC bit cost of one quantised unit (VLCEnc with a null stream, atrac3_bitstream.cpp:115-149).
__device__ inline uint32_t unit_vlc_bits(int wl, const int8_t* mant, int n)
{
    uint32_t vlc = 0;
    if (wl > 1) {
        for (int j = 0; j < n; ++j) vlc += huff_entry(wl, vlc_index(mant[j])) >> 8;
    } else {
        const uint32_t rt9[9] = {8, 4, 7, 2, 0, 1, 6, 3, 5};
        for (int j = 0; j < n / 2; ++j) vlc += huff_entry(1, rt9[3 * (mant[2 * j] + 1) + (mant[2 * j + 1] + 1)]) >> 8;
    }
    return vlc;
}

// Tonal component side information: grouping (GroupTonalComponents, atrac3_bitstream.cpp:338-380) and
// cost / emission (EncodeTonalComponents :382-524). Serial (one lane). With EMIT the bits go to `words`
// starting at bit `pos`; returns the number of bits.
template <bool EMIT>
__device__ __attribute__((noinline)) int tonal_encode(const PsyRec* rec, const uint8_t* tbits /* [kMaxTonal][8] */, const int* alloc,
                                   int n_alloc, uint32_t* words, int pos)
{
    const int nt = rec->n_tonal;
    uint8_t grp_of[kMaxTonal];
    int tcsgn = 0;
    for (int t = 0; t < nt; ++t) {
        const int bfu = rec->tonal[t].bfu;
        if (bfu >= n_alloc) {
            grp_of[t] = 0xff;
            continue;
        }
        int quant = alloc[bfu] + 4;
        if (quant > 7) quant = 7;
        if (quant < 2) quant = 2;
        grp_of[t] = (uint8_t)(quant * 8 + rec->tonal[t].len);
    }
    // first pass: count sub-groups (needed up front for the 5-bit header)
    for (int g = 16; g < 64; ++g) {
        int mem[kMaxTonal];
        int nm = 0;
        for (int t = 0; t < nt; ++t)
            if (grp_of[t] == g) mem[nm++] = t;
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
        int mem[kMaxTonal];
        int nm = 0;
        for (int t = 0; t < nt; ++t)
            if (grp_of[t] == g) mem[nm++] = t;
        if (nm == 0) continue;
        // sub-group boundaries
        int sg_start[kMaxTonal];
        int nsg = 0;
        {
            int cur = 0;
            while (cur < nm) {
                int start = cur;
                sg_start[nsg++] = cur;
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
            uint8_t cnt[16];
            for (int j = 0; j < 16; ++j) cnt[j] = 0;
            for (int j = sgStart; j < sgEnd; ++j) cnt[rec->tonal[mem[j]].pos >> 6]++;
            int bandFlag[4];
            for (int b = 0; b < 4; ++b) bandFlag[b] = cnt[4 * b] | cnt[4 * b + 1] | cnt[4 * b + 2] | cnt[4 * b + 3];
            if (EMIT)
                for (int b = 0; b < 4; ++b) put_bits(words, pos + used + b, bandFlag[b] != 0, 1);
            used += 4;
            if (EMIT) put_bits(words, pos + used, (uint32_t)codedValues - 1, 3);
            used += 3;
            if (EMIT) put_bits(words, pos + used, (uint32_t)q, 3);
            used += 3;
            int lastPos = sgStart;
            for (int j = 0; j < 16; ++j) {
                if (!bandFlag[j >> 2]) continue;
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
                    used += 12 + tbits[mem[k] * 8 + q];
                }
                lastPos = k;
            }
        }
    }
    return used;
}

// ---- quantisation kernel: one 256-thread workgroup per (stream, output frame, channel) -----------------
//
// Computes what TEncCache would compute lazily - every (bfu, wordlen) unit - so the rate loop that follows is
// pure table look-up. The work is laid out so that wave instructions carry full lanes and the order-dependent
// parts stay short:
//  (A) 1024 scaled values and all 7 x 1024 roundings, four lines per work-item;
//  (B) the 256 strictly ordered energy sums (32 x e1, 224 x e2) on ONE wavefront: chains are packed so that every
//      lane adds exactly 128 terms (1 x 128, 2 x 64, 4 x 32, 8 x 16 or 16 x 8 lines), i.e. 128 lock-step steps;
//  (C) energy-adaptive re-rounding of BFUs 19..31, the 91 units dealt round-robin to the four wavefronts, each
//      working wave-locally (no workgroup barrier): ballot/popcount compaction of the candidates that can be
//      re-rounded into a private scratch, rank sort by |delta| with one lane per candidate (falls back to the
//      libstdc++-order sort when two listed candidates tie); then one lane per unit runs the sequential pass;
//  (D) CLC / VLC bit costs of the final mantissas, four lines x seven wordlens per work-item.
constexpr int kEaLine0 = 288;              // first spectral line of BFU 19
constexpr int kEaLines = 1024 - kEaLine0;  // 736
constexpr int kQuantThreads = 256;

__device__ __forceinline__ uint32_t lds_huff(const uint16_t* s_huff, int sel, uint32_t idx)
{
    return s_huff[huff_off(sel) + idx];
}

__global__ __launch_bounds__(kQuantThreads) void k_quant(BackParams p, const Tables* T)
{
    __shared__ __attribute__((aligned(16))) float s_val[1024];
    __shared__ __attribute__((aligned(16))) int8_t s_mant[7 * 1024];
    __shared__ __attribute__((aligned(16))) uint8_t s_si[7 * kEaLines];   // per unit: candidates ordered by |delta|
    __shared__ __attribute__((aligned(16))) float s_uk[kEaLines + 4];     // per wordlen plane: sort keys listed per unit (+inf padded)
    __shared__ uint32_t s_pu[kEaLines];                                   // candidate lists per size class: unit << 14 | slot in the unit's key list << 7 | line inside the BFU
    __shared__ int s_cnt[2][16];                                          // [0..12] candidates per unit, [13] per plane (double buffered)
    __shared__ uint8_t s_code[7 * (kEaLines / 4)];   // 2 bits per (wordlen, line): 1 = re-roundable when e2 < e1, 2 = when e2 > e1
    __shared__ uint8_t s_nc[91];
    __shared__ uint8_t s_tie[91];
    __shared__ float s_e1[32];
    __shared__ float s_err[8 * 32];            // e2 during phases B/C, then e1 / e2
    __shared__ uint32_t s_vlc[8 * 32];
    __shared__ uint16_t s_huff[130];
    __shared__ int s_anytie;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const size_t cf = blockIdx.x;  // (s * n_out + fo) * 2 + ch
    const float* specs = p.specs + cf * 1024;
    const PsyRec* rec = p.psy + cf;

    if (tid < 130) s_huff[tid] = c_huff[tid];
    s_vlc[tid] = 0;
    if (tid == 0) s_anytie = 0;
    // ---- (A) scaled values (TScaler::Scale) and mantissa = lrint(value * MaxQuant[wl]) ----
    {
        const int i0 = tid * 4;
        const float sf = T->scale[rec->sfi[bfu_of_line(i0)]];
        const float4 x = *reinterpret_cast<const float4*>(specs + i0);
        float v[4] = {x.x / sf, x.y / sf, x.z / sf, x.w / sf};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (fabsf(v[k]) >= 1.0f) v[k] = (v[k] > 0) ? 0.99999f : -0.99999f;
        float4 o;
        o.x = v[0]; o.y = v[1]; o.z = v[2]; o.w = v[3];
        *reinterpret_cast<float4*>(s_val + i0) = o;
        // The energy-adaptive pass (atrac_scale.cpp:66-126) may re-round a line of BFU > 18 only when it is close to a
        // rounding boundary (|delta| < 0.25) AND lies on the side the pass moves: rounded towards zero and below the
        // top code (pass taken when e2 < e1) or rounded away from zero (e2 > e1). Which pass runs is known only after
        // the ordered sums of phase B, so both possibilities are recorded here, where value * mul is in a register anyway.
        const bool ea = i0 >= kEaLine0;
#pragma unroll
        for (int wl = 1; wl <= 7; ++wl) {
            const float mul = max_quant(wl);
            uint32_t pk = 0, code = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float t = v[k] * mul;
                const int m = __float2int_rn(t);
                pk |= (uint32_t)(uint8_t)m << (8 * k);
                if (ea) {
                    const float am = fabsf((float)m), at = fabsf(t);
                    const float delta = t - (truncf(t) + 0.5f);
                    const uint32_t c = (am < at && am < (mul - 1)) ? 1u : (am > at) ? 2u : 0u;
                    code |= (fabsf(delta) < 0.25f ? c : 0u) << (2 * k);
                }
            }
            *reinterpret_cast<uint32_t*>(s_mant + (wl - 1) * 1024 + i0) = pk;
            if (ea) s_code[(wl - 1) * (kEaLines / 4) + ((i0 - kEaLine0) >> 2)] = (uint8_t)code;
        }
    }
    __syncthreads();
    if (p.debug_stop == 1) return;

    // ---- (B) ordered sums on wavefront 0: lane -> `per` chains of `len` lines, chain c = (bfu, kind),
    //      kind 0 = e1 (sum of value^2), kind 1..7 = e2 of that wordlen (sum of mantissa^2 / mul^2) ----
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);   // the other three wavefronts wait for these chains
        int len, first_chain, bfu_top;
        if (lane < 16) { len = 128; first_chain = lane; bfu_top = 31; }
        else if (lane < 32) { len = 64; first_chain = (lane - 16) * 2; bfu_top = 29; }
        else if (lane < 52) { len = 32; first_chain = (lane - 32) * 4; bfu_top = 25; }
        else if (lane < 60) { len = 16; first_chain = (lane - 52) * 8; bfu_top = 15; }
        else { len = 8; first_chain = (lane - 60) * 16; bfu_top = 7; }
        const int lsh = 31 - __builtin_clz(len);   // log2(len)
        float acc = 0.0f;
        for (int pos = 0; pos < 128; pos += 8) {
            const int c = first_chain + (pos >> lsh), off = pos & (len - 1);
            const int bfu = bfu_top - (c >> 3), kind = c & 7;
            const int start = bfu_start(bfu);
            if (off == 0) acc = 0.0f;
            // both element kinds are fetched as 8 raw bytes / 8 floats; terms first, then the 8 dependent adds
            float term[8];
            if (kind == 0) {
                const float4 a = *reinterpret_cast<const float4*>(s_val + start + off);
                const float4 b = *reinterpret_cast<const float4*>(s_val + start + off + 4);
                term[0] = a.x * a.x; term[1] = a.y * a.y; term[2] = a.z * a.z; term[3] = a.w * a.w;
                term[4] = b.x * b.x; term[5] = b.y * b.y; term[6] = b.z * b.z; term[7] = b.w * b.w;
            } else {
                const float mul = max_quant(kind);
                const float inv2 = (float)(1.0 / (double)(mul * mul));
                const uint2 pk = *reinterpret_cast<const uint2*>(s_mant + (kind - 1) * 1024 + start + off);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int m = (int)(int8_t)(((k < 4 ? pk.x : pk.y) >> (8 * (k & 3))) & 0xff);
                    term[k] = (float)(m * m) * inv2;
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += term[k];
            if (off + 8 == len) {
                if (kind == 0) s_e1[bfu] = acc;
                else s_err[kind * 32 + bfu] = acc;
            }
        }
        __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();
    if (p.debug_stop == 2) return;

    // ---- (C) energy-adaptive units (bfu > 18), one wordlen plane (13 units, 736 lines) at a time ----
    // A candidate (|delta| < 0.25) can only ever be re-rounded when it passes the side test of the pass that will
    // run (e2 < e1: rounded down and below the top code; e2 > e1: rounded up; equal: nothing runs), and a skipped
    // candidate changes no state - so only those are listed (atrac_scale.cpp:66-126). The pass visits them by
    // ascending |delta|: the position of a candidate is the number of keys of its unit below its own, which needs
    // the unit's keys as a set only - the lists are filled in arrival order through LDS atomics.
    //   step 1: 64-line chunks -> flag, compact into the unit's key list and into the candidate list of the unit's
    //           size class (128-, 64- and 32-line BFUs: lists at 0, 256, 512 of s_pu)
    //   step 2: wavefront k owns size class k: one lane per listed candidate counts the smaller keys of its unit and
    //           stores its line at that rank. Lists of one class have similar lengths, so the lanes of a wavefront run
    //           the same number of steps; a unit is handled entirely inside one wavefront.
    //   step 3: equal keys collide on a rank; the loser notices on read-back and the unit goes to the exact path (C3)
    for (int i = tid; i < kEaLines + 4; i += kQuantThreads) s_uk[i] = __builtin_huge_valf();
    if (tid < 32) (&s_cnt[0][0])[tid] = 0;
    if (tid < 91) s_tie[tid] = 0;
    __syncthreads();
    for (int wl = 1; wl <= 7; ++wl) {
        const float mul = max_quant(wl);
        int* cnt = s_cnt[wl & 1];
        const uint8_t* codes = s_code + (wl - 1) * (kEaLines / 4);
        uint8_t* plane_sorted = s_si + (wl - 1) * kEaLines;
        for (int c = wave; c < 12; c += 4) {
            const int hi = lane >> 5;
            int bfu, start;   // of this lane's half of the chunk
            if (c < 4) { bfu = 18 + 2 * c + hi; start = 256 + 64 * c + 32 * hi; }
            else if (c < 8) { bfu = 22 + c; start = 256 + 64 * c; }
            else { bfu = 30 + ((c - 8) >> 1); start = 768 + 128 * ((c - 8) >> 1); }
            const int cls = c < 4 ? 2 : c < 8 ? 1 : 0;
            const int line = 256 + 64 * c + lane;
            bool flag = false;
            if (bfu > 18) {
                const float e1 = s_e1[bfu], e2 = s_err[wl * 32 + bfu];
                const uint32_t want = (e2 < e1) ? 1u : (e2 > e1) ? 2u : 3u;   // 3: equal energies, no pass
                flag = ((codes[(line - kEaLine0) >> 2] >> (2 * (line & 3))) & 3u) == want;
            }
            const unsigned long long mask = __ballot(flag);
            const uint32_t half = hi ? (uint32_t)(mask >> 32) : (uint32_t)mask;
            const int below = __popc(half & ((1u << (lane & 31)) - 1u));
            int base_u = 0, base_p = 0;
            if ((lane & 31) == 0) {
                const int n_half = __popc(half);
                if (n_half) {
                    base_u = atomicAdd(&cnt[bfu - 19], n_half);
                    base_p = atomicAdd(&cnt[13 + cls], n_half);
                }
            }
            {
                const int u0 = __builtin_amdgcn_readlane(base_u, 0), u1 = __builtin_amdgcn_readlane(base_u, 32);
                const int p0 = __builtin_amdgcn_readlane(base_p, 0), p1 = __builtin_amdgcn_readlane(base_p, 32);
                base_u = hi ? u1 : u0;
                base_p = (hi ? p1 : p0) + 256 * cls;
            }
            if (flag) {
                const float t = s_val[line] * mul;
                const float key = fabsf(t - (truncf(t) + 0.5f));   // sort key |delta|
                s_uk[start - kEaLine0 + base_u + below] = key;
                s_pu[base_p + below] = ((uint32_t)(bfu - 19) << 14) | ((uint32_t)(base_u + below) << 7) | (uint32_t)(line - start);
            }
        }
        __syncthreads();
        if (wave < 3) {
            const int cls = wave;
            const int total = cnt[13 + cls];
            int rank[4] = {0, 0, 0, 0};
            int where[4] = {0, 0, 0, 0};   // the list entry (unit, slot, line)
#pragma unroll
            for (int rd = 0; rd < 4; ++rd) {
                const int t = rd * 64 + lane;
                if (t < total) {
                    const int pu = (int)s_pu[256 * cls + t];
                    const int ub = pu >> 14;
                    const int ustart = ub < 7 ? 32 * ub : ub < 11 ? 64 * ub - 224 : 128 * ub - 928;
                    const float key = s_uk[ustart + ((pu >> 7) & 127)];
                    const int nc = cnt[ub];
                    const float4* t4 = reinterpret_cast<const float4*>(s_uk + ustart);
                    int r = 0;
                    for (int q = 0; q < nc; q += 4) {
                        const float4 cur = t4[q >> 2];
                        r += (cur.x < key);
                        r += (cur.y < key);
                        r += (cur.z < key);
                        r += (cur.w < key);
                    }
                    plane_sorted[ustart + r] = (uint8_t)(pu & 127);
                    rank[rd] = ustart + r;
                    where[rd] = pu;
                }
            }
            wave_sync();
#pragma unroll
            for (int rd = 0; rd < 4; ++rd) {
                if (rd * 64 + lane < total && plane_sorted[rank[rd]] != (uint8_t)(where[rd] & 127)) {
                    s_tie[(wl - 1) * 13 + (where[rd] >> 14)] = 1;
                    s_anytie = 1;
                }
            }
            const int ub0 = cls == 0 ? 11 : cls == 1 ? 7 : 0, ub1 = cls == 0 ? 13 : cls == 1 ? 11 : 7;
            if (lane < ub1 - ub0) {
                const int ub = ub0 + lane;
                const int nc = cnt[ub];
                const int ustart = ub < 7 ? 32 * ub : ub < 11 ? 64 * ub - 224 : 128 * ub - 928;
                s_nc[(wl - 1) * 13 + ub] = (uint8_t)nc;
                for (int k = nc; k < ((nc + 3) & ~3); ++k) plane_sorted[ustart + k] = 0;   // pad to a multiple of four
            }
            wave_sync();
            // next plane: fresh key lists of this class
            const int k0 = cls == 0 ? 480 : cls == 1 ? 224 : 0, k1 = cls == 0 ? 736 : cls == 1 ? 480 : 224;
            for (int i = k0 + lane; i < k1; i += 64) s_uk[i] = __builtin_huge_valf();
        } else if (lane < 16) {
            s_cnt[(wl + 1) & 1][lane] = 0;   // the other counter buffer: last read one plane ago, next used one plane ahead
        }
        __syncthreads();
    }
    __syncthreads();
    if (p.debug_stop == 3) return;
    // (C3) equal keys among listed candidates: libstdc++'s std::sort order decides (rare). The order of equal
    //      elements depends on the whole array the reference sorts, so the full |delta| < 0.25 list is rebuilt,
    //      sorted with the restated algorithm and then filtered.
    if (s_anytie) {
        SortItem* s_items = reinterpret_cast<SortItem*>(s_uk);   // scratch of the rare tie-order sort: the key lists are dead now
        static_assert(sizeof(SortItem) * 128 <= sizeof(float) * (kEaLines + 4), "tie-sort scratch must fit in the key lists");
        if (tid == 0) {
            for (int u = 0; u < 91; ++u) {
                if (!s_tie[u]) continue;
                const int wl = 1 + u / 13, bfu = 19 + u % 13;
                const int start = bfu_start(bfu), n = bfu_start(bfu + 1) - start;
                const float mul = max_quant(wl);
                int nall = 0;
                for (int j = 0; j < n; ++j) {
                    const float t = s_val[start + j] * mul;
                    const float delta = t - (truncf(t) + 0.5f);
                    if (fabsf(delta) < 0.25f) {
                        s_items[nall].key = delta;
                        s_items[nall].idx = j;
                        ++nall;
                    }
                }
                std_sort_abs(s_items, nall);
                const float e1 = s_e1[bfu], e2 = s_err[wl * 32 + bfu];
                const int dir = (e2 < e1) ? 1 : (e2 > e1) ? -1 : 0;
                uint8_t* sorted = s_si + (wl - 1) * kEaLines + (start - kEaLine0);
                int nc = 0;
                for (int q = 0; q < nall; ++q) {
                    const int j = s_items[q].idx;
                    const float t = s_val[start + j] * mul;
                    const int m0 = __float2int_rn(t);
                    const float am = (float)(m0 < 0 ? -m0 : m0);
                    const bool side = (dir > 0) ? (am < fabsf(t) && am < (mul - 1)) : (dir < 0) ? (am > fabsf(t)) : false;
                    if (side) sorted[nc++] = (uint8_t)j;
                }
            }
        }
        __syncthreads();
    }
    // (C4) sequential re-rounding pass, one lane per unit
    //      Wide BFUs (long candidate lists) share wavefront 0, the 32-line BFUs wavefront 1: a wavefront runs as long
    //      as its longest list.
    if (tid < 128) __builtin_amdgcn_s_setprio(3);   // two wavefronts run the ordered passes, two wait
    if (tid < 42 || (tid >= 64 && tid < 113)) {
        const int wl = (tid < 42) ? 1 + tid / 6 : 1 + (tid - 64) / 7;
        const int bfu = (tid < 42) ? 26 + tid % 6 : 19 + (tid - 64) % 7;
        const int u = (wl - 1) * 13 + (bfu - 19);
        const int start = bfu_start(bfu);
        const int nc = s_nc[u];
        if (nc > 0) {
            const float mul = max_quant(wl);
            const float inv2 = (float)(1.0 / (double)(mul * mul));
            s_err[wl * 32 + bfu] = ea_greedy(s_val + start, s_si + (wl - 1) * kEaLines + (start - kEaLine0), nc, mul, inv2, s_e1[bfu],
                                             s_err[wl * 32 + bfu], s_mant + (wl - 1) * 1024 + start);
        }
    }
    if (tid < 128) __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    if (p.debug_stop == 4) return;

    // ---- (D) e1 / e2; VLC cost of the final mantissas: four lines (one BFU) x seven wordlens per work-item ----
    if (tid < 224) {
        const int wl = 1 + tid / 32, bfu = tid % 32;
        s_err[wl * 32 + bfu] = s_e1[bfu] / s_err[wl * 32 + bfu];
    }
    {
        const int i0 = tid * 4;
        const int bfu = bfu_of_line(i0);
#pragma unroll
        for (int wl = 1; wl <= 7; ++wl) {
            const uint32_t pk = *reinterpret_cast<const uint32_t*>(s_mant + (wl - 1) * 1024 + i0);
            const int m0 = (int)(int8_t)(pk & 0xff), m1 = (int)(int8_t)((pk >> 8) & 0xff);
            const int m2 = (int)(int8_t)((pk >> 16) & 0xff), m3 = (int)(int8_t)(pk >> 24);
            uint32_t bits;
            if (wl > 1) {
                bits = (lds_huff(s_huff, wl, vlc_index(m0)) >> 8) + (lds_huff(s_huff, wl, vlc_index(m1)) >> 8) +
                       (lds_huff(s_huff, wl, vlc_index(m2)) >> 8) + (lds_huff(s_huff, wl, vlc_index(m3)) >> 8);
            } else {
                const uint32_t rt9[9] = {8, 4, 7, 2, 0, 1, 6, 3, 5};
                bits = (lds_huff(s_huff, 1, rt9[3 * (m0 + 1) + (m1 + 1)]) >> 8) + (lds_huff(s_huff, 1, rt9[3 * (m2 + 1) + (m3 + 1)]) >> 8);
            }
            atomicAdd(&s_vlc[wl * 32 + bfu], bits);
        }
    }
    __syncthreads();
    // ---- results to HBM ----
    QuantRec* q = p.quant + cf;
    if (tid < 224) {
        const int wl = 1 + tid / 32, bfu = tid % 32;
        const int n = bfu_start(bfu + 1) - bfu_start(bfu);
        const uint32_t clc = (wl > 1) ? (uint32_t)clc_len(wl) * n : 2u * n;
        q->err[wl - 1][bfu] = s_err[wl * 32 + bfu];
        q->cost[wl - 1][bfu] = clc | (s_vlc[wl * 32 + bfu] << 13);
    }
    {
        uint4* dst = reinterpret_cast<uint4*>(p.mant + cf * 7168);
        const uint4* src = reinterpret_cast<const uint4*>(s_mant);
        for (int i = tid; i < 7168 / 16; i += kQuantThreads) dst[i] = src[i];
    }
}

// ---- rate loop + packing: one wavefront (64 lanes) per (stream, output frame, channel) ------------------
//
// Lane i < 32 owns BFU i and keeps its seven (error, cost) pairs and the ConsiderEnergyErr fixed-point map in
// registers, so one evaluation of CalcBitsAllocation + CalcSpecsBitsConsumption is a handful of VALU ops, a
// DPP row reduction and two readlanes; no LDS round trip and no barrier sits inside the bisection.
__global__ __launch_bounds__(64) void k_rate_pack(BackParams p, const Tables* T)
{
    __shared__ uint32_t s_words[kBitWords];
    __shared__ int s_alloc[32];
    __shared__ uint8_t s_tbits[kMaxTonal * 8];
    __shared__ uint16_t s_huff[130];
    __shared__ int s_misc[4];
    __shared__ uint32_t s_cost[8 * 32];
    __shared__ unsigned long long s_tmask[4];   // per QMF band: set of (quantiser, length) groups among the live tonal blocks

    const int lane = threadIdx.x;
    const int n_out = p.n_blocks - p.f0;
    const size_t cf = blockIdx.x;
    const int ch = (int)(cf & 1);
    const int fo = (int)((cf >> 1) % n_out);
    const int s = (int)((cf >> 1) / n_out);
    const int f = fo + p.f0;
    const PsyRec* recs = p.psy + (cf & ~(size_t)1);
    const PsyRec* rec = recs + ch;
    const Curve* curves = p.curves + ((size_t)s * p.n_blocks + f) * 8;
    const QuantRec* q = p.quant + cf;
    const int8_t* gmant = p.mant + cf * 7168;
    const int half = p.frame_sz >> 1;
    const int n_tonal = rec->n_tonal;

    for (int i = lane; i < kBitWords; i += 64) s_words[i] = 0;
    for (int i = lane; i < 130; i += 64) s_huff[i] = c_huff[i];

    // ---- header + gain info bits, joint-stereo byte shift, target bits (WriteSoundUnit :759-810) ----
    int hdr[2];
    for (int c2 = 0; c2 < 2; ++c2) {
        int bits = (p.js && c2 == 1) ? 14 : 6;
        bits += 2;
        for (int b = 0; b < 4; ++b) bits += 3 + 9 * curves[c2 * 4 + b].n;
        // one input channel, joint stereo: the second element has ONE subband and no gain points (atrac3denc.cpp:843-849)
        if (p.mono_js && c2 == 1) bits = 14 + 2 + 3;
        hdr[c2] = bits;
    }
    int shift = 0;
    if (p.mono_js) {   // CalcMSBytesShift with an empty second element: the maximum (atrac3_bitstream.cpp:745-747)
        const int totalUsed = 12 + hdr[0] + hdr[1];
        shift = (int)((uint32_t)p.frame_sz / 2 - (1 + ((uint32_t)totalUsed - 1) / 8));
    } else if (p.js) {
        const int b0 = -6 - hdr[0], b1 = -6 - hdr[1];
        const int totalUsed = 0 - b0 - b1;
        const int maxShift = (int)((uint32_t)p.frame_sz / 2 - (1 + ((uint32_t)totalUsed - 1) / 8));
        const float m = recs[0].loud_ch, sd = recs[1].loud_ch;
        const float total = sd + m;
        float ratio = 0.0f;
        if (total > 0) ratio = (float)((double)(m / total) - 0.5);
        int v = __float2int_rn((float)p.frame_sz * ratio);
        if (v > maxShift) v = maxShift;
        if (v < -maxShift) v = -maxShift;
        shift = v;
    }
    const int nbytes = (ch == 0) ? half + shift : half - shift;
    int target = -6 - hdr[ch] + 8 * nbytes;
    if (target < 1) target = 1;
    target &= 0xffff;
    const float loudness = p.loud[(size_t)s * n_out + fo] / 0.006f;

    if (p.mono_js && ch == 1) {
        // TConfigure / TAlloc with empty ScaledBlocks (atrac3_bitstream.cpp:590-597, 623-626): JS parameters, one subband
        // without gain points, no tonal components, one BFU of precision 0 in coding mode 1 - 33 bits, then zeros
        __syncthreads();
        if (lane == 0) {
            put_bits(s_words, 0, 0, 1);
            put_bits(s_words, 1, 7, 3);
            for (int k = 0; k < 4; ++k) put_bits(s_words, 4 + 2 * k, 3, 2);
            put_bits(s_words, 12, 3, 2);
            put_bits(s_words, 14, 0, 2);       // numQmfBand - 1
            put_bits(s_words, 16, 0, 3);       // gain points of band 0
            put_bits(s_words, 19, 0, 5);       // tonal sub-groups
            put_bits(s_words, 24, 0, 5);       // numBlocks - 1
            put_bits(s_words, 29, 1, 1);       // coding mode
            put_bits(s_words, 30, 0, 3);       // precision of the one block
        }
        __syncthreads();
        uint8_t* frame1 = p.out + ((size_t)s * n_out + fo) * p.frame_sz;
        for (int j = lane; j < nbytes; j += 64) {
            const int src = nbytes - 1 - j;
            frame1[half + shift + j] = (src < kBitWords * 4) ? (uint8_t)(s_words[src >> 2] >> (24 - 8 * (src & 3))) : 0;
        }
        return;
    }

    // ---- TConfigure: spread (sequential float sums, every lane computes the same value) ----
    const int i = lane & 31;   // BFU owned by this lane (lanes 32..63 mirror 0..31 but never contribute)
    float spread;
    {
        const int my_sfi = rec->sfi[i];   // one load per lane; the ordered sums walk the lanes
        float sum = 0.0f;
        for (int k = 0; k < 32; ++k) sum += (float)__builtin_amdgcn_readlane(my_sfi, k);
        sum /= 32;
        float sigma = 0.0f;
        for (int k = 0; k < 32; ++k) {
            float t = ((float)__builtin_amdgcn_readlane(my_sfi, k) - sum);
            t *= t;
            sigma += t;
        }
        sigma /= 32;
        sigma = sqrtf(sigma);
        if (sigma > 14.0f) sigma = 14.0f;
        spread = sigma / 14.0f;
    }
    // tonal blocks: VLC bit cost for every quantiser 2..7
    for (int idx = lane; idx < n_tonal * 6; idx += 64) {
        const int t = idx / 6, qq = 2 + idx % 6;
        const TonalBlock& tb = rec->tonal[t];
        const float mul = max_quant(qq);
        int bits = 0;
        for (int z = 0; z < tb.len; ++z) bits += (int)(huff_entry(qq, vlc_index(__float2int_rn(tb.values[z] * mul))) >> 8);
        s_tbits[t * 8 + qq] = (uint8_t)bits;
    }
    // Lane t < n_tonal also owns tonal block t (its BFU, length and 64-line block): the cost of the tonal side
    // information is evaluated by these lanes in parallel inside the rate loop (inside the bisection below).
    int tb_bfu = 255, tb_len = 0, tb_blk = 0;
    if (lane < n_tonal) {
        const TonalBlock& tb = rec->tonal[lane];
        tb_bfu = tb.bfu;
        tb_len = tb.len;
        tb_blk = tb.pos >> 6;
    }
    // GroupTonalComponents (atrac3_bitstream.cpp:338-380) closes a sub-group only after EIGHT members of one group inside one
    // 64-line block. A 64-line block holds at most four tonal BFUs (BFUs 8..28 are 16 lines or wider, one run each), so
    // that never happens and every (quantiser, length) group is exactly one sub-group; the check below proves it for this
    // frame (blocks are ordered by position) and sends anything else down the literal, serial path.
    const bool tonal_serial = __ballot(lane + 7 < n_tonal &&
                                       __builtin_amdgcn_ds_bpermute(4 * ((lane + 7) & 63), tb_blk) == tb_blk) != 0ull;
    // per-BFU constants of CalcBitsAllocation (atrac3_bitstream.cpp:272-336)
    float A;
    bool gate;
    int tcount = 0;
    for (int t = 0; t < n_tonal; ++t) tcount += (__builtin_amdgcn_readlane(tb_bfu, t) == (lane & 31));
    float err[8];
    uint32_t cost[8];
    {
        int band = 0;
        if (i >= 18) band = 1;
        if (i >= 26) band = 2;
        if (i >= 30) band = 3;
        float g = 1.0f;
        if (p.ges) g = p.ges[((size_t)s * p.n_blocks + f) * 8 + ch * 4 + band];
        if (!(isfinite(g) && g > 0.0f)) g = 1.0f;
        const float corrected = rec->energy[i] * g;
        const float ath = T->ath_bfu[i] * loudness;
        gate = corrected < ath;
        const float csfi = fmaxf(0.0f, fminf(63.0f, (float)rec->sfi[i] + 1.5f * at3_log2f(T, g)));
        float x = 6.0f;
        if (i < 3) x = 2.8f;
        else if (i < 10) x = 2.6f;
        else if (i < 15) x = 3.3f;
        else if (i <= 20) x = 3.6f;
        else if (i <= 28) x = 4.2f;
        A = spread * (csfi / x) + (1.0f - spread) * (float)c_fixed_alloc[i];
        // (tonal blocks per BFU: counted below from the per-lane copies of the blocks' BFU indices)
        err[0] = 0.0f;
        cost[0] = 0;
#pragma unroll
        for (int wl = 1; wl <= 7; ++wl) {
            err[wl] = q->err[wl - 1][i];
            cost[wl] = q->cost[wl - 1][i];
        }
    }
    // ConsiderEnergyErr as a per-BFU map wl -> wl' (first 10 BFUs, atrac3_bitstream.cpp:241-257, :638-641):
    // BFUs are independent, so iterating the reference's do/while to its fixed point is a closure per BFU.
    // A wordlen keeps climbing while its energy error is out of range, so the map sends wl to the first k >= wl that
    // is acceptable (k = 0 and k = 7 always are): a backward scan over the eight entries.
    uint32_t gmap = 0;
    {
        int g = 7;
        gmap = 7u << 21;
#pragma unroll
        for (int k = 6; k >= 0; --k) {
            const float e = err[k];
            const bool climbs = i < 10 && k > 0 && ((e > 0 && e < 0.7f) || e > 1.2f);
            g = climbs ? g : k;
            gmap |= (uint32_t)g << (3 * k);
        }
    }
    // cost table of this lane's BFU in LDS: the rate loop indexes it with a run-time wordlen
#pragma unroll
    for (int wl = 0; wl <= 7; ++wl)
        if (lane < 32) s_cost[wl * 32 + lane] = cost[wl];
    __syncthreads();

    if (p.debug_stop == 11) return;
    // ---- rate loop: TConfigure / TAlloc under the bisection driver (uniform control flow) ----
    int num_bfu = p.bfu_idx_const ? p.bfu_idx_const : 32;
    if (target < 101) {
        int lim = 1;
        if (target > 5) lim = (target - 5) / 3;
        if (lim < 1) lim = 1;
        if (num_bfu > lim) num_bfu = lim;
    }
    if (num_bfu < 1) num_bfu = 1;
    int mode = 1;
    int bits = 0;
    for (;;) {
        float minL = -8.0f, maxL = 20.0f, curL = 0.0f, lastL = 20.0f;
        bool restart = false;
        for (;;) {
            const bool exhausted = (maxL <= minL);
            float lam;
            if (exhausted) {
                lam = lastL;
            } else {
                curL = (maxL + minL) * 0.5f;
                lam = curL;
            }
            bits = 0;
            if (lane < num_bfu) {
                if (!gate) {
                    const int tmp = (int)(A - lam);
                    if (tmp > 7) bits = 7;
                    else if (tmp < 0) bits = 0;
                    else if (tmp == 0) bits = 1;
                    else bits = tmp;
                }
                // one decrement per tonal block in this BFU while the wordlen is above 2 (:325-333)
                if (bits > 2 && tcount) bits = (bits - tcount > 2) ? bits - tcount : 2;
                bits = (int)((gmap >> (3 * bits)) & 7u);
            }
            const uint32_t mine = (lane < num_bfu) ? s_cost[bits * 32 + i] : 0u;
            const uint32_t rsum = row_allreduce_add(mine);
            const uint32_t acc = (uint32_t)__builtin_amdgcn_readlane((int)rsum, 0) + (uint32_t)__builtin_amdgcn_readlane((int)rsum, 16);
            // the count of non-zero BFUs comes from a ballot: summed as a third field it would need six bits at
            // 32 of 32 (and silently wrapped to zero when every BFU of the frame was coded)
            const uint32_t clc = acc & 0x1fffu, vlc = (acc >> 13) & 0x3fffu;
            const uint32_t nz = (uint32_t)__popcll(__ballot(lane < num_bfu && bits != 0));
            mode = clc <= vlc ? 1 : 0;
            const uint32_t spec_bits = (uint32_t)num_bfu * 3 + 6 * nz + (mode ? clc : vlc);
            uint32_t tonal_bits = 5;
            if (n_tonal > 0 && tonal_serial) {
                if (lane < 32) s_alloc[lane] = bits;
                __syncthreads();
                if (lane == 0) s_misc[0] = tonal_encode<false>(rec, s_tbits, s_alloc, num_bfu, nullptr, 0);
                __syncthreads();
                tonal_bits = (uint32_t)(s_misc[0] & 0xffff);
            } else if (n_tonal > 0) {
                // EncodeTonalComponents with a null stream (atrac3_bitstream.cpp:382-524), one lane per tonal block:
                //   5 (+2 when anything is coded) + per group 4 + 3 + 3 + 12 per QMF band the group touches
                //   + per coded block 6 + 6 + VLC bits of its values at the group's quantiser.
                const bool live = lane < n_tonal && tb_bfu < num_bfu;
                const int wl_t = __builtin_amdgcn_ds_bpermute(4 * (tb_bfu & 31), bits);   // this evaluation's wordlen of the block's BFU
                int qn = wl_t + 4;
                qn = qn > 7 ? 7 : qn;   // >= 4 always, so the lower clamp at 2 is never active here
                if (lane < 4) s_tmask[lane] = 0ull;
                wave_sync();
                uint32_t member = 0;
                if (live) {
                    atomicOr(&s_tmask[tb_blk >> 2], 1ull << ((qn - 2) * 7 + (tb_len - 1)));
                    member = 12u + s_tbits[lane * 8 + qn];
                }
                const uint32_t msum = row_allreduce_add(member);
                const uint32_t members = (uint32_t)__builtin_amdgcn_readlane((int)msum, 0) + (uint32_t)__builtin_amdgcn_readlane((int)msum, 16);
                wave_sync();
                const unsigned long long m0 = s_tmask[0], m1 = s_tmask[1], m2 = s_tmask[2], m3 = s_tmask[3];
                const uint32_t groups = (uint32_t)__popcll(m0 | m1 | m2 | m3);
                const uint32_t group_bands = (uint32_t)(__popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3));
                if (groups) tonal_bits = 5u + 2u + 10u * groups + 12u * group_bands + members;
            }
            const uint32_t total = spec_bits + tonal_bits;
            const int last_alloc = __builtin_amdgcn_readlane(bits, (num_bfu - 1) & 31);
            bool done;
            if (exhausted) {
                done = true;
            } else if (total < (uint32_t)target) {
                lastL = curL;
                maxL = curL - 0.01f;
                done = false;
            } else if (total > (uint32_t)target) {
                minL = curL + 0.01f;
                done = false;
            } else {
                done = true;
            }
            if (!done) continue;
            if (!p.bfu_idx_const && num_bfu > 1 && last_alloc == 0) {
                num_bfu--;
                restart = true;
            }
            break;
        }
        if (!restart) break;
    }
    if (lane < 32) s_alloc[lane] = bits;
    __syncthreads();

    if (p.debug_stop == 12) return;
    // ---- emission (WriteSoundUnit header, EncodeSpecs) ----
    int pos = 0;
    if (lane == 0) {
        if (p.js && ch == 1) {
            put_bits(s_words, 0, 0, 1);
            put_bits(s_words, 1, 7, 3);
            for (int k = 0; k < 4; ++k) put_bits(s_words, 4 + 2 * k, 3, 2);
            put_bits(s_words, 12, 3, 2);
            pos = 14;
        } else {
            put_bits(s_words, 0, 0x28, 6);
            pos = 6;
        }
        put_bits(s_words, pos, 3, 2);
        pos += 2;
        for (int b = 0; b < 4; ++b) {
            const Curve& c = curves[ch * 4 + b];
            put_bits(s_words, pos, c.n, 3);
            pos += 3;
            for (int k = 0; k < c.n; ++k) {
                put_bits(s_words, pos, c.level[k], 4);
                put_bits(s_words, pos + 4, c.loc[k], 5);
                pos += 9;
            }
        }
        pos += tonal_encode<true>(rec, s_tbits, s_alloc, num_bfu, s_words, pos);
        put_bits(s_words, pos, (uint32_t)num_bfu - 1, 5);
        put_bits(s_words, pos + 5, (uint32_t)mode, 1);
        pos += 6;
        s_misc[1] = pos;
    }
    __syncthreads();
    pos = s_misc[1];
    if (p.debug_stop == 13) return;
    const unsigned long long nzmask = __ballot(lane < num_bfu && bits != 0);
    if (lane < num_bfu) put_bits(s_words, pos + 3 * lane, (uint32_t)bits, 3);
    pos += 3 * num_bfu;
    if (lane < num_bfu && bits)
        put_bits(s_words, pos + 6 * __popcll(nzmask & ((1ull << lane) - 1ull)), rec->sfi[lane], 6);
    pos += 6 * __popcll(nzmask);
    // mantissas: 16 spectral lines per lane (BFU sizes are multiples of 8, so at most two BFUs per lane)
    {
        const int base = lane * 16;
        uint32_t code[16];
        int sum = 0;
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
            const int i0 = base + 8 * hlf;
            const int b = bfu_of_line(i0);
            const int wl = (b < num_bfu) ? s_alloc[b] : 0;
            int8_t m8[8];
            if (wl) {
                const uint2 pk = *reinterpret_cast<const uint2*>(gmant + (wl - 1) * 1024 + i0);
#pragma unroll
                for (int k = 0; k < 8; ++k) m8[k] = (int8_t)(((k < 4 ? pk.x : pk.y) >> (8 * (k & 3))) & 0xff);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                uint32_t cl = 0;
                if (wl > 1) {
                    if (mode == 1) {
                        const int nb = clc_len(wl);
                        cl = ((uint32_t)m8[k] & ((1u << nb) - 1u)) | ((uint32_t)nb << 16);
                    } else {
                        const uint32_t e = lds_huff(s_huff, wl, vlc_index(m8[k]));
                        cl = (e & 0xffu) | ((e >> 8) << 16);
                    }
                } else if (wl == 1 && (k & 1) == 0) {
                    if (mode == 1) {
                        const uint32_t rt[4] = {2, 3, 0, 1};
                        cl = ((rt[m8[k] + 2] << 2) | rt[m8[k + 1] + 2]) | (4u << 16);
                    } else {
                        const uint32_t rt9[9] = {8, 4, 7, 2, 0, 1, 6, 3, 5};
                        const uint32_t e = lds_huff(s_huff, 1, rt9[3 * (m8[k] + 1) + (m8[k + 1] + 1)]);
                        cl = (e & 0xffu) | ((e >> 8) << 16);
                    }
                }
                code[8 * hlf + k] = cl;
                sum += (int)(cl >> 16);
            }
        }
        int off = pos + wave_inclusive_scan(sum, lane) - sum;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int n = (int)(code[k] >> 16);
            if (n) {
                put_bits(s_words, off, code[k] & 0xffffu, n);
                off += n;
            }
        }
    }
    __syncthreads();

    // ---- frame assembly (atrac3_bitstream.cpp:826-834): ch0 bytes, then ch1 (byte-reversed when JS) ----
    uint8_t* frame = p.out + ((size_t)s * n_out + fo) * p.frame_sz;
    const int dst0 = (ch == 0) ? 0 : half + shift;
    for (int j = lane; j < nbytes; j += 64) {
        const int src = (p.js && ch == 1) ? (nbytes - 1 - j) : j;
        const uint8_t byte = (src < kBitWords * 4) ? (uint8_t)(s_words[src >> 2] >> (24 - 8 * (src & 3))) : 0;
        frame[dst0 + j] = byte;
    }
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
};

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
    if (k < kHist) {
        const float2* pcm2 = reinterpret_cast<const float2*>(p.pcm) + (size_t)s * p.n_blocks * 1024;
        const float2* hin = reinterpret_cast<const float2*>(p.hist_in) + (size_t)s * kHist;
        float2* hout = reinterpret_cast<float2*>(p.hist_out) + (size_t)s * kHist;
        const int g = p.n_blocks * 1024 - kHist + k;
        hout[k] = (g >= 0) ? pcm2[g] : hin[kHist + g];
    }
    if (k < 8) p.state[(size_t)s * 8 + k].prev_curve = p.curves[((size_t)s * p.n_blocks + (p.n_blocks - 1)) * 8 + k];
    if (p.sub && k < 8 * 128) {   // 8 rows x 512 floats as 16-byte words: the tail of every row (its front is the carried part)
        const int row = k >> 7, i = k & 127;
        const float4* src = reinterpret_cast<const float4*>(p.sub + ((size_t)s * 8 + row) * ((size_t)(p.n_blocks + 2) * 256) + (size_t)p.n_blocks * 256);
        reinterpret_cast<float4*>(p.sub_tail + ((size_t)s * 8 + row) * 512)[i] = src[i];
    }
}

}  // namespace at3
