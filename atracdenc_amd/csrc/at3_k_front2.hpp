// Front-end kernels, second generation: register-blocked QMF (eight outputs per work-item) and a register-resident
// MDCT-512 whose 128-point FFT lives in the sixteen lanes of one DPP row.
//
// Reference path replaced (paths relative to the reference's src/):
//   atrac3denc.cpp:701-713   PCM de-interleave, /4.0, Atrac3AnalysisFilterBank::Analysis
//   qmf/qmf.h:47-64          TQmf<nIn>::Analysis (48-tap two-band QMF), atrac/at3/atrac3_qmf.h:37-41 (tree)
//   atrac3denc.cpp:665-677   Matrixing (LP4 joint stereo)
//   gain_processor.h:87-121  TGainProcessor::Modulate
//   atrac3denc.cpp:33-58     TAtrac3MDCT::Mdct;  lib/mdct/mdct.h:51-104 TMDCT<512>;  kiss_fft.c (128-pt)
//
// QMF. One work-item produces EIGHT consecutive (lower, upper) output pairs of one two-band filter from 31 sample pairs
// (sixteen 16-byte LDS reads that stay in registers for all 384 packed multiply-adds); taps are wave-uniform scalars.
// Accumulation order per output is tap 0..23, multiply then add (no contraction), exactly as qmf.h:54-63.
//
// MDCT. A (channel, band) transform belongs to the 16 lanes of one DPP row: lane (q1, q2) = 4 q1 + q2 holds 8 complex
// points. The decimation-in-time order of kissfft (radix 4, 4, 4, 2) puts a complete 8-point sub-transform (radix-2
// leaves + the m = 2 pass) into one lane; the m = 8 pass combines the four lanes of a quad, the m = 32 pass four lanes
// 4 apart - two 16-byte exchanges through a small conflict-free LDS scratch, no workgroup barrier. The fold needs, per
// lane, sample indices {e, 128+e, 127-e, 255-e}; the odd-index two come from the mirror lane (DPP row_mirror), and the
// windowed overlap of the next frame is a product of the same 16 samples, so it never leaves the registers. The
// post-rotation's two outputs per bin interleave with the mirror lane's into runs of four consecutive spectral lines:
// spectra go to HBM as 16-byte stores straight from registers.
#pragma once
#include "at3_common.hpp"
#include "at3_k_frontend.hpp"

namespace at3 {

// ---- LDS rings of the QMF stages -----------------------------------------------------------------------------
// Logical ring element e = 48 history samples + new samples. Groups of four floats (two sample pairs) are the unit of
// the 16-byte reads; work-item g of a stage reads groups 4g .. 4g+15. Group G is stored in plane G & 3 at slot G >> 2,
// so one read instruction (fixed group offset) touches consecutive slots in consecutive work-items: conflict free.
// Inside a group the two floats of a pair are swapped, (x[2k+1], x[2k]): the operand order of the packed tap product.
constexpr int kHist8 = 48;
#ifndef K1_PCM_H
#define K1_PCM_H 74
#endif
#ifndef K1_S1_H
#define K1_S1_H 42
#endif
constexpr int kPcmH8 = K1_PCM_H;    // slots per plane, >= 67 and = 2 mod 8 (planes 32 bytes out of phase for the 16-byte stores)
constexpr int kS1H8 = K1_S1_H;     // >= 35
constexpr int kPcmRing8 = 16 * kPcmH8;   // floats per channel
constexpr int kS1Ring8 = 16 * kS1H8;     // floats per (channel, half)

template <int H>
__device__ __forceinline__ int ring8_slot(int G)
{
    return (G & 3) * H + (G >> 2);
}
template <int H>
__device__ __forceinline__ int ring8_at(int e)   // physical float index of logical element e
{
    return (ring8_slot<H>(e >> 2) << 2) | ((e & 3) ^ 1);
}

// Eight outputs m = 8g .. 8g+7 of one two-band filter. Output r, tap i uses ring pair k = r + 24 - i (local to the
// work-item's 32 pairs; pair 0 is never used): walking k downwards extends every sum in tap order.
template <int H>
__device__ __forceinline__ void qmf8(const float4* __restrict__ ring, int g, const f2 (&Wp)[24], f2 (&acc)[8])
{
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = mk2(0.0f, 0.0f);
#pragma unroll
    for (int q = 15; q >= 0; --q) {
        const float4 v = ring[(q & 3) * H + g + (q >> 2)];
        const f2 hi = mk2(v.z, v.w), lo = mk2(v.x, v.y);   // pairs 2q+1, 2q
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int i = r + 24 - (2 * q + 1);
            if (i >= 0 && i < 24) acc[r] = acc[r] + Wp[i] * hi;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int i = r + 24 - 2 * q;
            if (i >= 0 && i < 24) acc[r] = acc[r] + Wp[i] * lo;
        }
    }
}

// (lower, upper) outputs of eight accumulators as the two 16-byte groups a ring stores them in (pairs swapped).
__device__ __forceinline__ void qmf8_groups(const f2 (&acc)[8], float4 (&lo)[2], float4 (&up)[2])
{
    float l[8], u[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        l[r] = acc[r].x + acc[r].y;
        u[r] = acc[r].x - acc[r].y;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        lo[h] = make_float4(l[4 * h + 1], l[4 * h], l[4 * h + 3], l[4 * h + 2]);
        up[h] = make_float4(u[4 * h + 1], u[4 * h], u[4 * h + 3], u[4 * h + 2]);
    }
}

__device__ __forceinline__ void load_taps(const Tables* T, f2 (&Wp)[24])
{
#pragma unroll
    for (int i = 0; i < 24; ++i)
        Wp[i] = mk2(__uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(T->qmf_win[2 * i]))),
                    __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(T->qmf_win[2 * i + 1]))));
}

// ---- the QMF tree of ONE CHANNEL of one stream, run by ONE wavefront ------------------------------------------------
// 64 lanes = the 64 eight-output tasks of stage 1, then the 2 x 32 tasks of stage 2 (lanes 0..31: Qmf2 on the lower
// half -> bands 0, 1; lanes 32..63: Qmf3 on the upper half -> bands 3, 2). The wavefront owns its rings, so stage
// boundaries are wave-level rendezvous (LDS executes a wavefront's instructions in order): no workgroup barrier exists
// in these kernels, and wavefronts drift apart freely - one's FIR arithmetic covers another's LDS and HBM latency.
struct QmfLdsW {
    float pcm[kPcmRing8];        // this channel's PCM ring
    float s1[2 * kS1Ring8];      // stage-1 rings: lower half, upper half
};

struct QmfRunW {
    const float2* pcm2;     // [n_blocks][1024] interleaved (L, R) of this stream
    const float2* hist2;    // [kHist] samples before pcm (zeros at stream start)
    int ch;
    float4 nxt[8];          // interleaved PCM of the block after the one in the ring: lane t holds groups t + 64 w
    float4 hist_keep;       // lanes 52..63: the last 48 samples of the tile stored last (the next block's FIR history)
    float4 s1_keep;         // lanes 0..23: one group of stage-1 history
};

__device__ __forceinline__ float pcm_at(const QmfRunW& q, int g)   // sample g (this channel) relative to the call's first sample
{
    const float2 v = (g >= 0) ? q.pcm2[g] : q.hist2[kHist + g];
    return q.ch ? v.y : v.x;
}

// Fetch the tile of block b into registers: group u = samples 4u .. 4u+3 arrive as two 16-byte loads of (L, R) pairs.
__device__ __forceinline__ void tile_fetch(QmfRunW& q, int b, int lane)
{
    // (a block never straddles the history boundary - block starts are multiples of 1024 samples -, so the block's base is
    // wave-uniform: one scalar base and one 32-bit lane offset serve the eight loads)
    const float4* base = (b >= 0) ? reinterpret_cast<const float4*>(q.pcm2 + (ptrdiff_t)b * 1024) : reinterpret_cast<const float4*>(q.hist2 + (kHist + b * 1024));
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const unsigned u = 2u * (unsigned)(lane + 64 * w);
        q.nxt[2 * w] = base[u];
        q.nxt[2 * w + 1] = base[u + 1];
    }
}

// Store the fetched tile into the ring (data / 4.0, exact) and keep the last 48 samples for the history.
__device__ __forceinline__ void tile_store(QmfLdsW& S, QmfRunW& q, int lane)
{
    // (the channel is wave-uniform: ONE scalar branch picks between two straight sequences of sixteen multiplies and four stores;
    // written as a select per value the compiler copied the chosen half of every register pair first)
    float4* ring = reinterpret_cast<float4*>(S.pcm) + ring8_slot<kPcmH8>(kHist8 / 4 + lane);
    // (sixty-four groups on are sixteen slots on in the same plane)
    if (q.ch) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float4 a = q.nxt[2 * w], b = q.nxt[2 * w + 1];   // (L0, R0, L1, R1), (L2, R2, L3, R3)
            const float4 v = make_float4(a.w * 0.25f, a.y * 0.25f, b.w * 0.25f, b.y * 0.25f);
            ring[16 * w] = v;
            if (w == 3) q.hist_keep = v;
        }
    } else {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float4 a = q.nxt[2 * w], b = q.nxt[2 * w + 1];
            const float4 v = make_float4(a.z * 0.25f, a.x * 0.25f, b.z * 0.25f, b.x * 0.25f);
            ring[16 * w] = v;
            if (w == 3) q.hist_keep = v;
        }
    }
}

__device__ __forceinline__ void pcm_hist_store(QmfLdsW& S, const QmfRunW& q, int lane)
{
    if (lane >= 52) reinterpret_cast<float4*>(S.pcm)[ring8_slot<kPcmH8>(lane - 52)] = q.hist_keep;
}

__device__ __forceinline__ void qmf_stage1(QmfLdsW& S, const f2 (&Wp)[24], int lane)
{
    f2 acc[8];
    qmf8<kPcmH8>(reinterpret_cast<const float4*>(S.pcm), lane, Wp, acc);
    float4 lo[2], up[2];
    qmf8_groups(acc, lo, up);
    float4* rl = reinterpret_cast<float4*>(S.s1);
    float4* rh = reinterpret_cast<float4*>(S.s1 + kS1Ring8);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int slot = ring8_slot<kS1H8>(kHist8 / 4 + 2 * lane + h);
        rl[slot] = lo[h];
        rh[slot] = up[h];
    }
}

// Stage 2. Lane (which = lane >> 5, g = lane & 31) returns samples 8g .. 8g+7 of bands (which ? 3 : 0) in `lo` and
// (which ? 2 : 1) in `up`.
__device__ __forceinline__ void qmf_stage2(const QmfLdsW& S, const f2 (&Wp)[24], int lane, float (&lo)[8], float (&up)[8])
{
    const int which = lane >> 5, g = lane & 31;
    f2 acc[8];
    qmf8<kS1H8>(reinterpret_cast<const float4*>(S.s1 + which * kS1Ring8), g, Wp, acc);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        lo[r] = acc[r].x + acc[r].y;
        up[r] = acc[r].x - acc[r].y;
    }
}

__device__ __forceinline__ void s1_hist_fetch(const QmfLdsW& S, QmfRunW& q, int lane)
{
    if (lane < 24) q.s1_keep = reinterpret_cast<const float4*>(S.s1 + (lane / 12) * kS1Ring8)[ring8_slot<kS1H8>(128 + lane % 12)];
}
__device__ __forceinline__ void s1_hist_store(QmfLdsW& S, const QmfRunW& q, int lane)
{
    if (lane < 24) reinterpret_cast<float4*>(S.s1 + (lane / 12) * kS1Ring8)[ring8_slot<kS1H8>(lane % 12)] = q.s1_keep;
}

// The prologue's 144 floats of scratch sit in the body of the PCM ring's first plane (behind its three history slots, in
// front of the second plane), which the tile store that ends the prologue overwrites afterwards.
constexpr int kPrologueTmp = 16;
static_assert(kPrologueTmp >= 12 && kPrologueTmp + 144 <= 4 * kPcmH8, "prologue scratch must fit between the history and plane 1");

// Prologue of a run that starts with block b0: PCM history and stage-1 history of that block, then its tile.
// `tmp` = 144 floats of scratch. On return the tile of block b0 is in the ring; with PREFETCH block b0 + 1 is being fetched.
template <bool PREFETCH>
__device__ __forceinline__ void qmf_prologue(QmfLdsW& S, QmfRunW& q, float* tmp, const f2 (&Wp)[24], int b0, int b_last, int lane)
{
    // stage-1 outputs m = -48 .. -1 need samples -142 .. -1 (output m reads samples 2m - 46 .. 2m + 1)
    tile_fetch(q, b0, lane);
    float h[3];   // (every lane asks, at a clamped index: a load under a lane condition is issued late and waited for where the paths join)
#pragma unroll
    for (int k = 0; k < 3; ++k) h[k] = pcm_at(q, b0 * 1024 - 144 + (lane + 64 * k < 144 ? lane + 64 * k : 143));
    __builtin_amdgcn_sched_barrier(0);   // (all three requested before the first is waited for)
#pragma unroll
    for (int k = 0; k < 3; ++k) h[k] *= 0.25f;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (lane + 64 * k < 144) tmp[lane + 64 * k] = h[k];
    wave_sync();
    if (lane < 48) {
        const float* x = tmp + 144 + 2 * (lane - 48);   // output m = lane - 48; sample s sits at tmp[144 + s]
        float lo = 0.0f, hi = 0.0f;
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            lo += Wp[i].x * x[1 - 2 * i];
            hi += Wp[i].y * x[-2 * i];
        }
        S.s1[ring8_at<kS1H8>(lane)] = lo + hi;
        S.s1[kS1Ring8 + ring8_at<kS1H8>(lane)] = lo - hi;
        S.pcm[ring8_at<kPcmH8>(lane)] = tmp[96 + lane];   // samples -48 .. -1
    }
    wave_sync();   // (the tile overwrites the scratch)
    tile_store(S, q, lane);
    if (PREFETCH && b0 + 1 <= b_last) tile_fetch(q, b0 + 1, lane);
    wave_sync();
}

// Workgroup b of a launch of 2 n workgroups -> (unit u in [0, n), channel). The two channels of a unit read the SAME interleaved PCM, so
// they should meet in one L2: workgroups go round the eight XCDs in turn, hence b and b + 8 run on the same XCD one dispatch step apart.
// (Units beyond the last whole set of eight take neighbouring workgroups: correct, only without the shared L2.)
__device__ __forceinline__ void xcd_pair(int b, int n, int& u, int& ch)
{
    const int full = (n >> 3) << 4;
    if (b < full) {
        const int k = b >> 3;
        ch = k & 1;
        u = ((k >> 1) << 3) | (b & 7);
    } else {
        const int t = b - full;
        ch = t & 1;
        u = (full >> 1) + (t >> 1);
    }
}

// ---- subband analysis only (feeds the gain-control kernels and the MDCT-from-subbands kernel) ----------------------
// Raw L/R subbands of blocks 0 .. n_blocks-1 to HBM; blocks -2 and -1 (look-back of the gain analysis, overlap of the
// first frame) are the previous call's last two, carried in `sub_tail` and copied in front by the (stream, channel)'s runs
// between them. One wavefront = (stream, channel, one of `sub_runs` runs of blocks); a workgroup is four independent wavefronts.
__global__ __launch_bounds__(256) void k_qmf_sub8(FrontParams p, const Tables* T, int n_waves)
{
    __shared__ __attribute__((aligned(16))) QmfLdsW s_q[4];
    const int lane = threadIdx.x & 63;
    const int wave = (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (uniform: the run's indices and base addresses live in scalar registers)
    const int W = blockIdx.x * 4 + wave;
    if (W >= n_waves) return;
    QmfLdsW& S = s_q[wave];
    const int nb2 = p.n_blocks + 2;
    const int nchunks = p.sub_runs;   // runs per (stream, channel): the blocks are dealt out as evenly as possible
    int chunk, ch, s;
    if (nchunks % 4 == 0) {
        // a workgroup is four consecutive runs of ONE (stream, channel): the two channels of a piece of a stream go to the same XCD (xcd_pair)
        const int per = nchunks / 4;
        int u;
        xcd_pair((int)blockIdx.x, n_waves / (2 * nchunks) * per, u, ch);
        s = u / per;
        chunk = (u % per) * 4 + wave;
    } else {
        chunk = W % nchunks;
        ch = (W / nchunks) & 1;
        s = W / (2 * nchunks);
    }
    const int ba = (chunk * p.n_blocks) / nchunks;
    const int bb = ((chunk + 1) * p.n_blocks) / nchunks;
    const size_t sublen = (size_t)nb2 * 256;
    f2 Wp[24];
    load_taps(T, Wp);
    QmfRunW q;
    q.pcm2 = reinterpret_cast<const float2*>(p.pcm) + (size_t)s * p.n_blocks * 1024;
    q.hist2 = reinterpret_cast<const float2*>(p.hist) + (size_t)s * kHist;
    q.ch = ch;
    q.s1_keep = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    qmf_prologue<true>(S, q, S.pcm + kPrologueTmp, Wp, ba, bb - 1, lane);
    const int which = lane >> 5, g = lane & 31;
    float* out_lo = p.sub + ((size_t)s * 8 + ch * 4 + (which ? 3 : 0)) * sublen + 8 * g;
    float* out_up = p.sub + ((size_t)s * 8 + ch * 4 + (which ? 2 : 1)) * sublen + 8 * g;
    for (int b = ba; b < bb; ++b) {
        if (b > ba) s1_hist_store(S, q, lane);
        qmf_stage1(S, Wp, lane);
        wave_sync();
        pcm_hist_store(S, q, lane);
        if (b + 1 < bb) tile_store(S, q, lane);          // block b + 1 (fetched during the previous block)
        if (b + 2 < bb) tile_fetch(q, b + 2, lane);       // lands during stage 2 and the next stage 1
        s1_hist_fetch(S, q, lane);
        float lo[8], up[8];
        qmf_stage2(S, Wp, lane, lo, up);
        float4* o0 = reinterpret_cast<float4*>(out_lo + (size_t)(b + 2) * 256);
        float4* o1 = reinterpret_cast<float4*>(out_up + (size_t)(b + 2) * 256);
        o0[0] = make_float4(lo[0], lo[1], lo[2], lo[3]);
        o0[1] = make_float4(lo[4], lo[5], lo[6], lo[7]);
        o1[0] = make_float4(up[0], up[1], up[2], up[3]);
        o1[1] = make_float4(up[4], up[5], up[6], up[7]);
        wave_sync();
    }
    // The carried blocks -2 and -1 of this (stream, channel) - 4 bands x 512 floats = 512 sixteen-byte words - move in front of
    // the call's subbands: nobody in this kernel reads them, so the runs SHARE the copy (word chunk * 64 + lane, + 64 runs, ...)
    // and make it their last act. The run of block 0 used to do all of it first, as eight load-store pairs that may alias for all
    // the compiler knows: eight global round trips one after the other before that wavefront began its real work.
    {
        const float4* src = reinterpret_cast<const float4*>(p.sub_tail + ((size_t)s * 8 + ch * 4) * 512);
        for (int i = chunk * 64 + lane; i < 512; i += 64 * nchunks) {
            const int band = i >> 7;
            reinterpret_cast<float4*>(p.sub + ((size_t)s * 8 + ch * 4 + band) * sublen)[i & 127] = src[i];
        }
    }
}

// ---- MDCT-512 of one (channel, band) in a 16-lane row ------------------------------------------------------------
constexpr int kRowScratch4 = 80;   // 16-byte slots of exchange scratch per row (4 x 20: one spare 64-byte chunk per quad)

__device__ __forceinline__ float dpp_row_mirror(float v)
{
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x140, 0xf, 0xf, false));
}
// `v` of the mirror lane in the rows of ROWS (bit r = row r of the wavefront), `old` in the others
template <int ROWS>
__device__ __forceinline__ float dpp_row_mirror_rows(float old, float v)
{
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp((int)__float_as_uint(old), (int)__float_as_uint(v), 0x140, ROWS, 0xf, false));
}

// Per-lane constants of the row transform. They depend on the lane's position in its row only (L = 4 q1 + q2), so one
// table of 18 sixteen-byte entries per L serves every row of a workgroup; entry e of lane L sits at [e][L], which a
// wavefront reads as sixteen consecutive slots (its four rows read the same sixteen).
//   0..3    EncodeWindow at {e, 128+e, 127-e, 255-e}, e = 2 b + 32 q3, b = q1 + 4 q2          (q3 = entry)
//   4..7    (cos, sin) of the FFT inputs n2 = b + 16 q3 and n2 = b + 16 q3 + 64                  (q3 = entry - 4)
//   8..11   (cos, sin) of the output bins n2 = 8 q1 + 2 q2 + kappa + 32 i', kappa = 0, 1        (i' = entry - 8)
//   12..14  pass m = 8 twiddles tw[4 j k], k = 2 q2 + kappa, kappa = 0, 1                         (j = entry - 11)
//   15..17  pass m = 32 twiddles tw[j k], k = 8 q1 + 2 q2 + kappa                                 (j = entry - 14)
struct MdctTab {
    float4 e[18][16];
};

// Copied from Tables::mdct_tab by the 256 work-items of the workgroup (one 16-byte load each, 32 of them a second one); the
// caller synchronises before the first use. Request and store are two calls so that a kernel can put its first HBM requests
// between them: loads return in issue order, and the table - out of L2, asked for first - is then in LDS long before they land.
struct MdctTabRegs {
    float4 a, b;
};
// (NT = work-items of the workgroup, 192 or 256: the table has 288 entries, so the first 288 - NT work-items take a second one)
template <int NT = 256>
__device__ __forceinline__ MdctTabRegs mdct_tab_request(const Tables* T, int tid)
{
    static_assert(NT >= 144 && (NT >= 288 || (288 - NT) % 32 == 0), "one or two entries per work-item");
    const float4* flat = reinterpret_cast<const float4*>(&T->mdct_tab[0][0][0]);
    MdctTabRegs r;
    if (NT >= 288) {
        r.a = flat[tid < 288 ? tid : 0];
        r.b = r.a;
    } else {
        r.a = flat[tid];
        r.b = flat[NT + (tid < 288 - NT ? tid : 0)];
    }
    return r;
}
template <int NT = 256>
__device__ __forceinline__ void mdct_tab_store(MdctTab& tab, const MdctTabRegs& r, int tid)
{
    float4* flat = &tab.e[0][0];
    if (NT >= 288) {
        if (tid < 288) flat[tid] = r.a;
    } else {
        flat[tid] = r.a;
        if (tid < 288 - NT) flat[NT + tid] = r.b;
    }
}

// The lane's sixteen samples from its row's 256. The lane reads the even-index pairs (x[i], x[i+1]) at i = e and
// i = 128 + e (e = 2 b + 32 q3) - row_load - and the odd-index two of every q3 arrive from the mirror lane - row_finish.
// `rd2(i)` returns the two consecutive floats (x[i], x[i+1]), i even. All lanes of the wavefront call row_finish.
struct RowRaw {
    f2 r0[4], r1[4];
};
template <typename Rd2>
__device__ __forceinline__ void row_load(Rd2 rd2, int b, RowRaw& r)
{
#pragma unroll
    for (int q3 = 0; q3 < 4; ++q3) {
        const int e = 2 * b + 32 * q3;
        r.r0[q3] = rd2(e);
        r.r1[q3] = rd2(128 + e);
    }
}
__device__ __forceinline__ void row_finish(const RowRaw& r, float (&X)[4][4])
{
#pragma unroll
    for (int q3 = 0; q3 < 4; ++q3) {
        X[q3][0] = r.r0[q3].x;
        X[q3][1] = r.r1[q3].x;
        // x[e + 1] is the mirror lane's x[127 - e''] at q3'' = 3 - q3, x[129 + e] its x[255 - e'']
        X[q3][2] = dpp_row_mirror(r.r0[3 - q3].y);
        X[q3][3] = dpp_row_mirror(r.r1[3 - q3].y);
    }
}
template <typename Rd2>
__device__ __forceinline__ void row_gather(Rd2 rd2, int b, float (&X)[4][4])
{
    RowRaw r;
    row_load(rd2, b, r);
    row_finish(r, X);
}

// Divisors of TGainProcessor::Modulate (gain_processor.h:93-112) for the eight samples of cell `cell / 8`: level
// boundaries and the 8-sample ramps are aligned to these cells, so a cell is untouched (1.0), divided by one level or by
// one running-product ramp. `cv` should be read in place (LDS).
__device__ __forceinline__ void cell_divisors(const Curve& cv, const float* gain_interp, int cell, float (&d)[8])
{
    // the curve as its two 8-byte halves, walked by selects (cell_divisors_packed, at3_k_frontend.hpp)
    const uint4 w = *reinterpret_cast<const uint4*>(&cv);
    cell_divisors_packed((uint64_t)w.x | ((uint64_t)w.y << 32), (uint64_t)w.z | ((uint64_t)w.w << 32), gain_interp, cell, d);
}

__device__ __forceinline__ f2 f2lo(float4 v) { return mk2(v.x, v.y); }
__device__ __forceinline__ f2 f2hi(float4 v) { return mk2(v.z, v.w); }
// (a.x * c + a.y * s, a.y * c - a.x * s): the pre-rotation of mdct.h:76-86 as three packed operations
__device__ __forceinline__ f2 rot_pre(f2 a, f2 cs)
{
    const f2 t1 = a * mk2(cs.x, cs.x);
    const f2 t2 = mk2(a.y, a.x) * mk2(cs.y, cs.y);
    return mk2(t1.x + t2.x, t1.y - t2.y);
}

// One frame of one row. X = the new half's samples (after M/S matrixing and gain modulation) indexed [q3][{e, 128+e,
// 127-e, 255-e}]; pw = the windowed overlap, same indexing, carried in registers from frame to frame; `inv_scale` =
// 1 / GainLevel[first point] when the frame's curve is non-empty (the overlap half is divided by that level,
// gain_processor.h:87-121), else 1. `scratch` = the row's kRowScratch4 16-byte slots. Returns the lane's sixteen
// spectral lines as the four 16-byte stores of mdct_rows_store (odd bands already reversed: see the end of the function).
// WAVE-UNIFORM: every lane of the wavefront must call (DPP and wave-level rendezvous inside).
// XORS: the exchange scratch without its spare chunks - quad q's sixteen slots at 16 q, slot index ^ 4 in odd quads (the
// same bank phase the padded layout gets from its 20-slot stride) - 64 instead of 80 slots per row.
template <bool XORS>
__device__ __forceinline__ int xslot(int quad, int idx)
{
    return XORS ? 16 * quad + (idx ^ ((quad & 1) << 2)) : 20 * quad + idx;
}
// The fold of mdct.h:64-86 in its two halves. A frame's FFT inputs are sums and differences of products of the NEW half's samples
// (A = (r0a, i0b) per q3) and of the windowed OVERLAP (B = (i0a, r0b) per q3): a run that starts in the middle of a stream can form A
// of its first frame at once and B only when the run before it has finished its last block (k_qmf_mdct8's chained runs).
// fold_new also leaves the next frame's overlap = EncodeWindow[i] * new[i] (atrac3denc.cpp:47) in pw.
__device__ __forceinline__ void mdct_fold_old(const float (&pw)[4][4], float inv_scale, float (&B)[4][2])
{
#pragma unroll
    for (int q3 = 0; q3 < 4; ++q3) {
        const float p0 = pw[q3][0] * inv_scale, p1 = pw[q3][1] * inv_scale, p2 = pw[q3][2] * inv_scale, p3 = pw[q3][3] * inv_scale;
        B[q3][0] = p1 - p2;   // i0a
        B[q3][1] = p3 - p0;   // r0b
    }
}
__device__ __forceinline__ void mdct_fold_new(const MdctTab& tab, float (&pw)[4][4], const float (&X)[4][4], int L, bool want_a, float (&A)[4][2])
{
#pragma unroll
    for (int q3 = 0; q3 < 4; ++q3) {
        const float4 w = tab.e[q3][L];
        pw[q3][0] = w.x * X[q3][0];
        pw[q3][1] = w.y * X[q3][1];
        pw[q3][2] = w.z * X[q3][2];
        pw[q3][3] = w.w * X[q3][3];
        if (want_a) {
            A[q3][0] = w.y * X[q3][2] + w.z * X[q3][1];   // r0a
            A[q3][1] = w.w * X[q3][0] + w.x * X[q3][3];   // i0b
        }
    }
}
template <bool XORS>
__device__ __forceinline__ void mdct_row_transform(const MdctTab& tab, const f2 (&tw2)[3], const float (&A)[4][2], const float (&B)[4][2], float4* scratch, int L, float4 (&slots)[4]);

template <bool XORS = false>
__device__ __forceinline__ void mdct_row_frame(const MdctTab& tab, const f2 (&tw2)[3], float (&pw)[4][4], const float (&X)[4][4], float inv_scale,
                                               float4* scratch, int L, bool emit, float4 (&slots)[4])
{
    float A[4][2], B[4][2];
    if (emit) mdct_fold_old(pw, inv_scale, B);
    mdct_fold_new(tab, pw, X, L, emit, A);
    if (!emit) return;   // priming block: only the overlap is wanted (uniform per wavefront)
    mdct_row_transform<XORS>(tab, tw2, A, B, scratch, L, slots);
}

// The transform proper: pre-rotation (mdct.h:76-86), the 128-point FFT in the row's sixteen lanes, post-rotation, store order.
template <bool XORS>
__device__ __forceinline__ void mdct_row_transform(const MdctTab& tab, const f2 (&tw2)[3], const float (&A)[4][2], const float (&B)[4][2], float4* scratch, int L, float4 (&slots)[4])
{
    const int q1 = L >> 2, q2 = L & 3;
    f2 z[8];   // element j = 2 q3 + q4 of the lane's 8-point sub-transform
#pragma unroll
    for (int q3 = 0; q3 < 4; ++q3) {
        const float4 cs = tab.e[4 + q3][L];
        z[2 * q3] = rot_pre(mk2(A[q3][0], B[q3][0]), f2lo(cs));       // n = e < 128: (r0a, i0a)
        z[2 * q3 + 1] = rot_pre(mk2(B[q3][1], A[q3][1]), f2hi(cs));   // n = 128 + e: (r0b, i0b)
    }
    // radix-2 leaves (m = 1, twiddle tw[0]) and the m = 2 pass (butterfly k on elements k, k+2, k+4, k+6)
    {
        const f2 w0 = mk2(1.0f, 0.0f);   // tw128[0] = (cos 0, sin 0)
#pragma unroll
        for (int q3 = 0; q3 < 4; ++q3) bfly2(z[2 * q3], z[2 * q3 + 1], w0);
        bfly4<false>(z[0], z[2], z[4], z[6], w0, w0, w0);
        bfly4<false>(z[1], z[3], z[5], z[7], tw2[0], tw2[1], tw2[2]);
    }
    // exchange 1: butterfly k = 2 q2 + kappa of the m = 8 pass takes element k of the four lanes of the quad
    {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) scratch[xslot<XORS>(q1, 4 * q2 + (jj ^ q2))] = make_float4(z[2 * jj].x, z[2 * jj].y, z[2 * jj + 1].x, z[2 * jj + 1].y);
    }
    wave_sync();
    f2 y[2][4];   // [kappa][i]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 v = scratch[xslot<XORS>(q1, 4 * i + (q2 ^ i))];
        y[0][i] = f2lo(v);
        y[1][i] = f2hi(v);
    }
    wave_sync();
    {
        const float4 t1 = tab.e[12][L], t2 = tab.e[13][L], t3 = tab.e[14][L];
        bfly4<false>(y[0][0], y[0][1], y[0][2], y[0][3], f2lo(t1), f2lo(t2), f2lo(t3));
        bfly4<false>(y[1][0], y[1][1], y[1][2], y[1][3], f2hi(t1), f2hi(t2), f2hi(t3));
    }
    // exchange 2: butterfly k = 8 a + 2 q2 + kappa of the m = 32 pass takes (kappa, i = a) of the lanes (i', q2)
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) scratch[xslot<XORS>(q2, 4 * q1 + (i ^ q1))] = make_float4(y[0][i].x, y[0][i].y, y[1][i].x, y[1][i].y);
    }
    wave_sync();
    f2 u[2][4];   // [kappa][i']
#pragma unroll
    for (int ip = 0; ip < 4; ++ip) {
        const float4 v = scratch[xslot<XORS>(q2, 4 * ip + (q1 ^ ip))];
        u[0][ip] = f2lo(v);
        u[1][ip] = f2hi(v);
    }
    wave_sync();
    {
        const float4 t1 = tab.e[15][L], t2 = tab.e[16][L], t3 = tab.e[17][L];
        bfly4<false>(u[0][0], u[0][1], u[0][2], u[0][3], f2lo(t1), f2lo(t2), f2lo(t3));
        bfly4<false>(u[1][0], u[1][1], u[1][2], u[1][3], f2hi(t1), f2hi(t2), f2hi(t3));
    }
    // post-rotation (mdct.h:92-101): bin n2 -> line 2 n2 (E) and line 255 - 2 n2 (O)
    float E[2][4], O[2][4];
#pragma unroll
    for (int ip = 0; ip < 4; ++ip) {
        const float4 cs = tab.e[8 + ip][L];
#pragma unroll
        for (int kap = 0; kap < 2; ++kap) {
            const float r0 = u[kap][ip].x, i0 = u[kap][ip].y;
            const float cc = kap ? cs.z : cs.x, ss = kap ? cs.w : cs.y;
            E[kap][ip] = -r0 * cc - i0 * ss;
            O[kap][ip] = -r0 * ss + i0 * cc;
        }
    }
    // Lines 2B + 64 i' .. + 3 (B = 8 q1 + 2 q2) = E[0][i'], mirror O[1][3 - i'], E[1][i'], mirror O[0][3 - i'] - and odd bands are stored
    // reversed (atrac3denc.cpp:53-55): the run of i' = 3 - j goes, back to front, to lines 252 - (2B + 64 i') .. + 3 = (60 - 2B) + 64 j .. + 3.
    // So slot j of the lane's store order is, in even rows, the run i' = j and, in odd rows, the reversed run i' = 3 - j:
    //   even rows  E[0][j],         mirror O[1][3 - j], E[1][j],         mirror O[0][3 - j]
    //   odd rows   mirror O[0][j],  E[1][3 - j],        mirror O[1][j],  E[0][3 - j]
    // - one row_mirror move per value with a ROW MASK (the rows that do not take the mirror lane's value keep `old`): no per-lane selects,
    // no lane conditions round the stores, one store address per lane.
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        slots[j].x = dpp_row_mirror_rows<0xA>(E[0][j], O[0][j]);
        slots[j].y = dpp_row_mirror_rows<0x5>(E[1][3 - j], O[1][3 - j]);
        slots[j].z = dpp_row_mirror_rows<0xA>(E[1][j], O[1][j]);
        slots[j].w = dpp_row_mirror_rows<0x5>(E[0][3 - j], O[0][3 - j]);
    }
}

// Store a wavefront's four spectra (row = band) of one channel-frame: `frame_ch` = the channel-frame's 1024 lines (wave-uniform),
// slots as mdct_row_frame returns them. Slot j of lane (band, L) starts at line 256 band + 64 j + (band odd ? 60 - 2B : 2B).
__device__ __forceinline__ void mdct_rows_store(float* frame_ch, const float4 (&slots)[4], int lane)
{
    const int band = lane >> 4, L = lane & 15;
    const int line0 = 16 * (L >> 2) + 4 * (L & 3);
    const unsigned off = (unsigned)(256 * band + ((band & 1) ? 60 - line0 : line0));
    float4* dst = reinterpret_cast<float4*>(frame_ch + off);
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[16 * j] = slots[j];
}

// ---- MDCT from subbands in HBM (the gain-control path: k_qmf_sub8 wrote them for the gain analysis anyway) -----------
// One wavefront = the four bands of one (stream, channel) over a run of frames; no workgroup barrier after the table is
// built. A workgroup is four independent wavefronts. With joint stereo both channels' subbands are read and matrixed.
struct MdctSubParams {
    const float* sub;        // [S][2][4][(n_blocks+2)*256]
    const Curve* curves;     // [S][n_blocks][2][4] by frame index, or null (no gain control)
    const BandState* state;  // [S][2][4]: prev_curve = curve of frame -1
    float* specs;            // [S][n_out][2][1024]
    int n_blocks, f0, frame_runs, js, n_waves;
};

// Request block f - 1 (frame f's new half) of the lane's row: own channel, or both channels for the M/S matrixing.
__device__ __forceinline__ void rows_request(const float* sb_own, const float* sb_l, const float* sb_r, int js, int f, int b, RowRaw& ra, RowRaw& rb)
{
    const size_t off = (size_t)(f + 1) * 256;   // block f - 1 lives at (f - 1 + 2) * 256
    const float* pa = (js ? sb_l : sb_own) + off;
#pragma unroll
    for (int q3 = 0; q3 < 4; ++q3) {
        const int e = 2 * b + 32 * q3;
        ra.r0[q3] = *reinterpret_cast<const f2*>(pa + e);
        ra.r1[q3] = *reinterpret_cast<const f2*>(pa + 128 + e);
    }
    if (js) {
        const float* pb = sb_r + off;
#pragma unroll
        for (int q3 = 0; q3 < 4; ++q3) {
            const int e = 2 * b + 32 * q3;
            rb.r0[q3] = *reinterpret_cast<const f2*>(pb + e);
            rb.r1[q3] = *reinterpret_cast<const f2*>(pb + 128 + e);
        }
    }
}

// NW wavefronts per workgroup: four (25.5 KB of LDS, one table copy per four runs) or three (20.3 KB: a workgroup then fits the 22.5 KB slot ONE retiring
// k_gain_analysis workgroup frees when that kernel's launch is padded to seven per CU; at 25.5 KB it needed 26 KB slots, six per CU - EXPERIMENTS round 5)
template <bool JS, int NW>
__global__ __launch_bounds__(64 * NW) void k_mdct_sub(MdctSubParams p, const Tables* T)
{
    __shared__ __attribute__((aligned(16))) MdctTab s_tab;
    __shared__ __attribute__((aligned(16))) float4 s_x[NW][4 * kRowScratch4];   // exchange scratch; also the rows' divisor tables
    __shared__ __attribute__((aligned(16))) Curve s_cv[NW][4];
    __shared__ float s_gi[32];
    static_assert(kRowScratch4 * 4 >= 256, "a row's 256 divisors share its exchange scratch");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = (int)__builtin_amdgcn_readfirstlane((unsigned)tid >> 6);   // (uniform: the run's indices and base addresses live in scalar registers)
    // (a wavefront beyond the batch stays for the rendezvous below with the last run's indices and leaves after it)
    const bool live = (int)blockIdx.x * NW + wave < p.n_waves;
    const int W = live ? (int)blockIdx.x * NW + wave : p.n_waves - 1;
    const int n_out = p.n_blocks - p.f0;
    const int nchunks = p.frame_runs;   // runs per (stream, channel): the n_out frames are dealt out as evenly as possible
    const int chunk = W % nchunks;
    const int ch = (W / nchunks) & 1;
    const int s = W / (2 * nchunks);
    const int fa = p.f0 + (chunk * n_out) / nchunks;
    const int fb = p.f0 + ((chunk + 1) * n_out) / nchunks;
    const int band = lane >> 4, L = lane & 15;
    const int b = (L >> 2) + 4 * (L & 3);
    float pw[4][4];
#pragma unroll
    for (int q3 = 0; q3 < 4; ++q3)
#pragma unroll
        for (int k = 0; k < 4; ++k) pw[q3][k] = 0.0f;
    f2 tw2[3];   // pass m = 2, butterfly k = 1: tw128[16 j], wave-uniform
#pragma unroll
    for (int j = 0; j < 3; ++j) tw2[j] = ld2(T->tw128 + 16 * (j + 1));
    const size_t sublen = (size_t)(p.n_blocks + 2) * 256;
    const float* sb_own = p.sub + ((size_t)s * 8 + ch * 4 + band) * sublen;
    const float* sb_l = p.sub + ((size_t)s * 8 + band) * sublen;
    const float* sb_r = p.sub + ((size_t)s * 8 + 4 + band) * sublen;
    float4* scratch = s_x[wave] + band * kRowScratch4;
    float* divs = reinterpret_cast<float*>(scratch);
    Curve& cv = s_cv[wave][band];
    float* spec_base = p.specs + ((size_t)s * n_out * 2 + ch) * 1024;   // (uniform)
    // block f - 1 is frame f's new half; block fa - 2 primes the overlap of the run (modulated by frame fa - 1's curve).
    // The subbands of the next block are requested before the current one is transformed: a wavefront waits for HBM once.
    const MdctTabRegs tab_regs = mdct_tab_request<64 * NW>(T, tid);
    const float gi_v = T->gain_interp[(tid & 31) < 31 ? (tid & 31) : 30];
    __builtin_amdgcn_sched_barrier(0);   // (the workgroup's tables are asked for FIRST: see mdct_tab_request)
    RowRaw raw_a, raw_b;
    rows_request(sb_own, sb_l, sb_r, JS, fa - 1, b, raw_a, raw_b);
    // the frame's curve (16 bytes per row, lane 0 of the row) is requested one frame ahead like the subbands: fetched where it
    // is used, every frame of the run waited a global-memory latency for it
    auto curve_request = [&](int f) {
        return *reinterpret_cast<const uint4*>((f < 0) ? &p.state[(size_t)s * 8 + ch * 4 + band].prev_curve
                                                         : &p.curves[((size_t)s * p.n_blocks + f) * 8 + ch * 4 + band]);
    };
    uint4 c4_next = {0u, 0u, 0u, 0u};
    if (p.curves && L == 0) c4_next = curve_request(fa - 1);
    __builtin_amdgcn_sched_barrier(0);
    // the run's first subbands and curve are on their way while the workgroup stores its tables
    if (tid < 32) s_gi[tid] = gi_v;
    mdct_tab_store<64 * NW>(s_tab, tab_regs, tid);
    __syncthreads();   // the only workgroup-level rendezvous: the shared tables
    if (!live) return;
    for (int f = fa - 1; f < fb; ++f) {
        const bool emit = f >= fa;
        float X[4][4];
        if (JS) {
            float XL[4][4], XR[4][4];
            row_finish(raw_a, XL);
            row_finish(raw_b, XR);
#pragma unroll
            for (int q3 = 0; q3 < 4; ++q3)
#pragma unroll
                for (int k = 0; k < 4; ++k) X[q3][k] = ch ? (XL[q3][k] - XR[q3][k]) * 0.5f : (XL[q3][k] + XR[q3][k]) * 0.5f;
        } else {
            row_finish(raw_a, X);
        }
        if (f + 1 < fb) rows_request(sb_own, sb_l, sb_r, JS, f + 1, b, raw_a, raw_b);
        float inv_scale = 1.0f;
        if (p.curves) {
            // the frame's curve: 16 bytes per row, kept in LDS so that its point list can be walked with run-time indices
            if (L == 0) {
                *reinterpret_cast<uint4*>(&cv) = c4_next;
                if (f + 1 < fb) c4_next = curve_request(f + 1);
            }
            wave_sync();
            const bool has_curve = cv.n > 0;
            if (__ballot(has_curve) != 0ull) {   // rare on stationary material, the rule on transients
                if (has_curve) inv_scale = __uint_as_float((uint32_t)(127 - 4 + cv.level[0]) << 23);   // 1 / GainLevel[level[0]], a power of two
                // (all four rows walk - the walk bounds its trips by a ballot, which wants the whole wavefront; a row without a curve gets ones)
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    float d[8];
                    cell_divisors(cv, s_gi, 8 * (2 * L + c2), d);
                    float4* dst = reinterpret_cast<float4*>(divs + 16 * L + 8 * c2);
                    dst[0] = make_float4(d[0], d[1], d[2], d[3]);
                    dst[1] = make_float4(d[4], d[5], d[6], d[7]);
                }
                wave_sync();
                float D[4][4];
                row_gather([&](int i) { return has_curve ? *reinterpret_cast<const f2*>(divs + i) : mk2(1.0f, 1.0f); }, b, D);
                wave_sync();
                if (has_curve) {
#pragma unroll
                    for (int q3 = 0; q3 < 4; ++q3)
#pragma unroll
                        for (int k = 0; k < 4; ++k) X[q3][k] = X[q3][k] / D[q3][k];
                }
            }
        }
        float4 slots[4];
        mdct_row_frame(s_tab, tw2, pw, X, emit ? inv_scale : 1.0f, scratch, L, emit, slots);
        if (emit) mdct_rows_store(spec_base + (size_t)(f - p.f0) * 2048, slots, lane);
    }
}

// ---- fused QMF + MDCT (no gain control, discrete stereo): ONE wavefront per (stream, channel, run of frames) ---------
// Block b carries frame f = b + 1; the two blocks before the run's first frame prime the FIR histories and the MDCT
// overlap. The wavefront runs stage 1, stage 2 and the four bands' transforms of a block back to back; its subbands
// and exchange scratch reuse the rings that are dead at that point (the PCM ring after stage 1, the stage-1 rings after
// stage 2), 10 KB of LDS per wavefront in all. A workgroup is four independent wavefronts sharing the MDCT table.
#ifndef K1_ATTR
#define K1_ATTR
#endif
#ifndef K1_NW
#define K1_NW 4
#endif
#ifndef K1_EARLY
#define K1_EARLY 0
#endif
#ifndef K1_XOR
#define K1_XOR 0
#endif
#ifndef K1_FETCH_AT
#define K1_FETCH_AT 0
#endif
#ifndef K1_OPAQUE
#define K1_OPAQUE (K1_NW > 4)
#endif
#if K1_OPAQUE
#define K1_LANE(l) opaque_lane_value(l)
#else
#define K1_LANE(l) (l)
#endif
constexpr int kFusedWaves = K1_NW;          // wavefronts per workgroup of the fused kernel (they share the MDCT table)
constexpr bool kFusedEarlyTile = K1_EARLY;  // the next tile moves into the PCM ring as soon as the subbands were gathered
constexpr bool kFusedXor = K1_XOR;          // exchange scratch without spare chunks (64 slots per row)
constexpr int kFusedRowScratch4 = kFusedXor ? 64 : kRowScratch4;
// Where the exchange scratch of the four rows starts (in floats from the PCM ring's start): behind the block's subbands when
// it may use the ring's tail, at the stage-1 rings when the next tile moves into the PCM ring before the transform.
constexpr int kFusedScratchAt = kFusedEarlyTile ? kPcmRing8 : 4 * 264;
static_assert(kPcmRing8 >= 4 * 264, "the block's subbands reuse the PCM ring");
static_assert((kPcmRing8 + 2 * kS1Ring8 - kFusedScratchAt) * sizeof(float) >= sizeof(float4) * 4 * kFusedRowScratch4, "the exchange scratch reuses the rings");
#ifdef K1_STAMPS
#define K1_STAMP(k) do { const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime(); ph[k] += t_ - t_prev; t_prev = t_; } while (0)
#else
#define K1_STAMP(k) do { } while (0)
#endif
__global__ __launch_bounds__(64 * kFusedWaves) K1_ATTR void k_qmf_mdct8(FrontParams p, const Tables* T, int n_waves)
{
    constexpr int NW = kFusedWaves;
#ifdef K1_TAB_GLOBAL
    const MdctTab& s_tab = *reinterpret_cast<const MdctTab*>(&T->mdct_tab[0][0][0]);
#else
    __shared__ __attribute__((aligned(16))) MdctTab s_tab;
#endif
    __shared__ __attribute__((aligned(16))) QmfLdsW s_q[NW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = (int)__builtin_amdgcn_readfirstlane((unsigned)tid >> 6);   // (uniform: the run's indices, bounds and base addresses live in scalar registers)
    // The shared table is asked for first and stored behind the run's prologue - the wavefront's first PCM requests do not queue up
    // behind a workgroup rendezvous. (A wavefront past the end of the launch runs the last run's prologue, which writes nothing but
    // its own LDS, and leaves after the rendezvous.)
#ifndef K1_TAB_GLOBAL
    const MdctTabRegs tab_regs = mdct_tab_request<64 * NW>(T, tid);
    __builtin_amdgcn_sched_barrier(0);
#endif
#ifdef K1_STAMPS
    unsigned ph[7] = {0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime(), r_begin = __builtin_amdgcn_s_memrealtime();
    unsigned t_prev = (unsigned)t_begin;
#endif
    const int W0 = blockIdx.x * NW + wave;
    const bool live = W0 < n_waves;
    QmfLdsW& S = s_q[wave];
    const int n_out = p.n_blocks - p.f0;
    const int nchunks = p.frame_runs;   // runs per (stream, channel): the n_out frames are dealt out as evenly as possible
    int chunk, ch, s;
    if (nchunks % NW == 0) {
        // a workgroup is NW consecutive runs of ONE (stream, channel): the two channels of a piece of a stream go to the same XCD (xcd_pair)
        const int per = nchunks / NW;
        int u;
        xcd_pair((int)blockIdx.x, n_waves / (2 * nchunks) * per, u, ch);
        s = u / per;
        chunk = (u % per) * NW + wave;
    } else {
        const int W = live ? W0 : n_waves - 1;
        chunk = W % nchunks;
        ch = (W / nchunks) & 1;
        s = W / (2 * nchunks);
    }
    // Blocks b0 .. b_last of the run; block b carries frame b + 1, whose overlap is block b - 1. The run's FIRST block only leaves its
    // windowed samples behind as the next frame's overlap. Unchained (p.chain == 0) that block is one the run before this one computes as
    // well: every run pays a whole block's FIR to prime its overlap. CHAINED (p.chain != 0; the host picks it when a run is a few frames
    // long, then frame_runs is a multiple of NW), the NW wavefronts of a workgroup are NW consecutive runs of one (stream, channel) that cut
    // the group's frames + ONE priming block between them: wavefront j > 0 starts with a block whose frame it owns, forms the half of that
    // frame's fold that needs the new samples at once (parked in the frame's own slot of the spectra), and finishes the frame after the
    // rendezvous at the end, when wavefront j - 1 has left the other half - the differences of ITS last windowed samples - in LDS.
    int b0, b_last;
    bool deferred = false;
    if (p.chain) {
        const int groups = nchunks / NW, grp = chunk / NW;
        const int ga = p.f0 + (grp * n_out) / groups, gb = p.f0 + ((grp + 1) * n_out) / groups;   // the group's frames
        const int nblk = gb - ga + 1;                                                              // + the priming block ga - 2
        b0 = ga - 2 + (wave * nblk) / NW;
        b_last = ga - 2 + ((wave + 1) * nblk) / NW - 1;
        deferred = wave > 0;
    } else {
        b0 = p.f0 + (chunk * n_out) / nchunks - 2;
        b_last = p.f0 + ((chunk + 1) * n_out) / nchunks - 2;
    }
    f2 Wp[24];
    load_taps(T, Wp);
    QmfRunW q;
    q.pcm2 = reinterpret_cast<const float2*>(p.pcm) + (size_t)s * p.n_blocks * 1024;
    q.hist2 = reinterpret_cast<const float2*>(p.hist) + (size_t)s * kHist;
    q.ch = ch;
    q.s1_keep = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float pw[4][4];
#pragma unroll
    for (int q3 = 0; q3 < 4; ++q3)
#pragma unroll
        for (int k = 0; k < 4; ++k) pw[q3][k] = 0.0f;
    f2 tw2[3];   // pass m = 2, butterfly k = 1: tw128[16 j], wave-uniform
#pragma unroll
    for (int j = 0; j < 3; ++j) tw2[j] = ld2(T->tw128 + 16 * (j + 1));
    float (*sub)[264] = reinterpret_cast<float (*)[264]>(S.pcm);            // [band][256 + pad], after stage 1
    qmf_prologue<false>(S, q, S.pcm + kPrologueTmp, Wp, b0, b_last, lane);
#ifndef K1_TAB_GLOBAL
    mdct_tab_store<64 * NW>(s_tab, tab_regs, tid);
    __syncthreads();   // the only workgroup-level rendezvous: the shared table
#endif
    if (!live) return;   // (never in a chained launch: its grid is whole groups)
    K1_STAMP(0);
    // Per-lane LDS and HBM offsets are formed again in every block, phase by phase, from a lane index the optimiser cannot see
    // through (opaque_lane_value): hoisted in front of the loop - where the compiler puts anything loop-invariant - they were
    // fifty registers held for the whole run, the difference between three and four wavefronts per SIMD.
    const float* frame_base = p.specs + ((size_t)s * n_out * 2 + ch) * 1024;
    for (int blk = b0; blk <= b_last; ++blk) {
        {
            const int ln = K1_LANE(lane);
            if (!kFusedEarlyTile && blk > b0) {
                // the ring held the previous block's subbands until its transform had gathered them: now the tile
                // (fetched one block ago) and the FIR histories move in
                pcm_hist_store(S, q, ln);   // the tail of the previous tile, before tile_store replaces the copy in registers
                tile_store(S, q, ln);
            }
            if (blk > b0) {
                s1_hist_store(S, q, ln);
                wave_sync();
            }
            if (K1_FETCH_AT == 0 && blk + 1 <= b_last) tile_fetch(q, blk + 1, ln);   // lands during this block's arithmetic
        }
        K1_STAMP(1);
        {
            const int ln = K1_LANE(lane);
            qmf_stage1(S, Wp, ln);
            wave_sync();
            s1_hist_fetch(S, q, ln);
            if (K1_FETCH_AT == 1 && blk + 1 <= b_last) tile_fetch(q, blk + 1, ln);   // lands during stage 2
        }
        K1_STAMP(2);
        {
            const int ln = K1_LANE(lane);
            const int which = ln >> 5, g = ln & 31;
            float lo[8], up[8];
            qmf_stage2(S, Wp, ln, lo, up);
            wave_sync();   // every lane is done with the rings (the exchange scratch and the subbands overwrite them)
            float4* o0 = reinterpret_cast<float4*>(&sub[which ? 3 : 0][8 * g]);
            float4* o1 = reinterpret_cast<float4*>(&sub[which ? 2 : 1][8 * g]);
            o0[0] = make_float4(lo[0], lo[1], lo[2], lo[3]);
            o0[1] = make_float4(lo[4], lo[5], lo[6], lo[7]);
            o1[0] = make_float4(up[0], up[1], up[2], up[3]);
            o1[1] = make_float4(up[4], up[5], up[6], up[7]);
        }
        wave_sync();
        K1_STAMP(3);
        {
            const int f = blk + 1;
            float X[4][4];
            {
                const int ln = K1_LANE(lane);
                const int band = ln >> 4, L = ln & 15;
                const int b = (L >> 2) + 4 * (L & 3);
                row_gather([&](int i) { return *reinterpret_cast<const f2*>(&sub[band][i]); }, b, X);
                wave_sync();
                if (kFusedEarlyTile && blk < b_last) {
                    // the subbands are in registers: the next tile (requested earlier in this block) and the tail of this one move into the ring
                    pcm_hist_store(S, q, ln);
                    tile_store(S, q, ln);
                }
            }
            K1_STAMP(4);
            const bool emit = blk > b0;
            float* frame_ch = const_cast<float*>(frame_base) + (size_t)(f - p.f0) * 2048;
            if (emit) {
                float4 slots[4];
                {
                    const int ln = K1_LANE(lane);
                    const int band = ln >> 4, L = ln & 15;
                    float4* scratch = reinterpret_cast<float4*>(S.pcm + kFusedScratchAt) + band * kFusedRowScratch4;   // after stage 2 (and the gather)
                    mdct_row_frame<kFusedXor>(s_tab, tw2, pw, X, 1.0f, scratch, L, true, slots);
                }
                mdct_rows_store(frame_ch, slots, K1_LANE(lane));
            } else {
                float A[4][2];
                mdct_fold_new(s_tab, pw, X, K1_LANE(lane) & 15, deferred, A);
                if (deferred) {
                    float4* park = reinterpret_cast<float4*>(frame_ch) + 2 * K1_LANE(lane);
                    park[0] = make_float4(A[0][0], A[0][1], A[1][0], A[1][1]);
                    park[1] = make_float4(A[2][0], A[2][1], A[3][0], A[3][1]);
                }
            }
            K1_STAMP(5);
        }
    }
    if (p.chain) {
        // hand the overlap half of the next run's first frame on: (i0a, r0b) of this run's last windowed samples, lane for lane
        // (the rings are dead; every wavefront of the group - this is the same for all - arrives here)
        float4* hand = reinterpret_cast<float4*>(S.pcm) + 2 * lane;
        if (wave + 1 < NW) {
            float B[4][2];
            mdct_fold_old(pw, 1.0f, B);
            hand[0] = make_float4(B[0][0], B[0][1], B[1][0], B[1][1]);
            hand[1] = make_float4(B[2][0], B[2][1], B[3][0], B[3][1]);
        }
        __syncthreads();
        if (deferred) {
            const float4* from = reinterpret_cast<const float4*>(s_q[wave - 1].pcm) + 2 * lane;
            const float4 h0 = from[0], h1 = from[1];
            float* frame_ch = const_cast<float*>(frame_base) + (size_t)(b0 + 1 - p.f0) * 2048;
            const float4* park = reinterpret_cast<const float4*>(frame_ch) + 2 * lane;
            const float4 a0 = park[0], a1 = park[1];
            const float A[4][2] = {{a0.x, a0.y}, {a0.z, a0.w}, {a1.x, a1.y}, {a1.z, a1.w}};
#ifdef K1_SABOTAGE
            const float B[4][2] = {{h0.x, h0.y}, {h0.z, h0.w}, {h1.x, h1.y}, {h1.z, 0.0f}};
#else
            const float B[4][2] = {{h0.x, h0.y}, {h0.z, h0.w}, {h1.x, h1.y}, {h1.z, h1.w}};
#endif
            // (the exchange scratch lies behind the hand-over in this wavefront's own rings; the lane's stores below come after its own loads of
            // the parked half: the compiler keeps that order - they may alias - and a lane only overwrites what it and its mirror lane parked)
            float4 slots[4];
            const int band = lane >> 4, L = lane & 15;
            float4* scratch = reinterpret_cast<float4*>(S.pcm + kFusedScratchAt) + band * kFusedRowScratch4;
            wave_sync();
            mdct_row_transform<kFusedXor>(s_tab, tw2, A, B, scratch, L, slots);
            mdct_rows_store(frame_ch, slots, lane);
        }
    }
    K1_STAMP(6);
#ifdef K1_STAMPS
    if (p.clk && lane == 0) {
        unsigned long long* row = p.clk + 16 + (W0 & 255) * 24;   // (256 rows of 24 words: the space of k_alloc_pack's and k_gain_analysis1's phase rows)
        for (int k = 0; k < 6; ++k) atomicAdd(row + k, (unsigned long long)ph[k]);
        atomicAdd(row + 6, __builtin_amdgcn_s_memtime() - t_begin);
        atomicAdd(row + 7, __builtin_amdgcn_s_memrealtime() - r_begin);
        atomicAdd(row + 8, 1ull);
        atomicAdd(row + 9, (unsigned long long)(b_last - b0 + 1));
        atomicAdd(row + 10, (unsigned long long)ph[6]);
        // when the row's wavefronts of this launch started and ended (100 MHz): words 12 .. 15 keep max(~first start), max(last start), max(~first end), max(last end)
        const unsigned long long r_end = __builtin_amdgcn_s_memrealtime();
        atomicMax(row + 12, ~r_begin);
        atomicMax(row + 13, r_begin);
        atomicMax(row + 14, ~r_end);
        atomicMax(row + 15, r_end);
    }
#endif
}

}  // namespace at3
