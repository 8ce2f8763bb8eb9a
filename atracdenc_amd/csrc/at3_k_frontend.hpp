// Front-end kernels, part 1: CalcGainEnergyScale for the encoder, and the stage-level entry points of the C ABI
// (batched TAtrac3MDCT::Mdct and CalcGainEnergyScale on caller-provided buffers). The QMF tree and the encoder's MDCT
// live in at3_k_front2.hpp.
//
// Reference path replaced (paths relative to the reference's src/):
//   atrac3denc.cpp:175-224   CalcGainEnergyScale
//   gain_processor.h:87-121  TGainProcessor::Modulate
//   atrac3denc.cpp:33-58     TAtrac3MDCT::Mdct;  lib/mdct/mdct.h:51-104 TMDCT<512>;  kiss_fft.c (128-pt)
//
#pragma once
#include "at3_common.hpp"

namespace at3 {

struct FrontParams {
    const float* pcm;        // [S][n_blocks][1024][2]
    const float* hist;       // [S][kHist][2] samples preceding pcm (zeros at stream start)
    const Curve* curves;     // [S][n_blocks][2][4] by frame index f (GAIN only)
    const BandState* state;  // [S][2][4]: prev_curve = curve of frame -1 (GAIN only)
    float* specs;            // [S][n_out][2][1024], n_out = n_blocks - f0
    float* ges;              // [S][n_blocks][2][4] GainEnergyScale.Frame by frame index (GAIN only)
    float* sub;              // raw L/R subbands [S][2][4][(n_blocks+2)*256] (k_qmf_sub8 writes, the gain path and k_mdct_sub read)
    const float* sub_tail;   // [S][8][512] subbands of the two blocks before this call (k_state_update of the previous call)
    int n_blocks;
    int f0;                  // first frame index to emit (1 on the first call of a stream, else 0)
    int frame_runs;          // fused kernel: runs per (stream, channel) the output frames are cut into
    int js;
    int sub_runs;            // k_qmf_sub8: runs per (stream, channel) the n_blocks + 2 blocks are cut into
    int debug;               // profiling aid (env AT3HIP_DEBUG_FRONT, debug builds): 3 = skip the energy-scale chains
    int chain;               // fused kernel: runs of one (stream, channel) chained in workgroups (see k_qmf_mdct8), frame_runs a multiple of the workgroup's wavefronts
    unsigned long long* clk; // profiling builds (-DK1_STAMPS): per-phase cycle sums of the fused kernel's wavefronts (AT3HIP_TAP_CLOCK), else null
};

// The divisors of the eight samples of cell `cell / 8` under a curve given as its two 8-byte halves (n, level[7] |
// loc[7], pad): TGainProcessor::Modulate (gain_processor.h:93-112). Level boundaries and the 8-sample ramps are
// aligned to these cells, so a cell is untouched (1), divided by one level or by one running-product ramp.
// The point list is walked WITHOUT lane conditions: every lane runs the same (wave-uniformly bounded) trips and keeps what
// the first matching point gives it by selects - a lane condition costs a wavefront ~50 cycles (EXPERIMENTS.md, round 5), the
// walk as the reference writes it has three per point plus a divergent loop exit. A cell that no point covers keeps level 1,
// a cell that is not a ramp keeps the increment 1: v * 1.0f is v, so the running product needs no condition either.
__device__ __forceinline__ void cell_divisors_packed(uint64_t lo, uint64_t hi, const float* gain_interp, int cell, float (&d)[8])
{
    const int n = (int)(lo & 0xffu);
    float lvl = 1.0f, inc = 1.0f;
    int pos = 0;
    bool open = true;
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        if (__ballot(q < n) == 0ull) break;   // wave-uniform
        const int level = (int)((lo >> (8 * (q + 1))) & 0xffu);
        const int next = (q + 1 < 7) ? (int)((lo >> (8 * ((q + 1 < 7 ? q + 1 : 0) + 1))) & 0xffu) : 4;
        const int lastPos = (int)((hi >> (8 * q)) & 0xffu) << 3;
        const bool live = open & (q < n);
        const bool flat = live & (cell >= pos) & (cell < lastPos);
        const int pos2 = lastPos > pos ? lastPos : pos;
        const bool has_ramp = pos2 < lastPos + 8;
        const bool ramp = live & has_ramp & (cell >= pos2) & (cell < lastPos + 8);   // (never both: flat needs cell < lastPos <= pos2)
        const float step = gain_interp[(((q + 1) < n ? next : 4) - level + 15) & 31];
        const bool hit = flat | ramp;
        lvl = hit ? gain_level_of(level) : lvl;
        inc = ramp ? step : inc;
        open = open & !hit;
        pos = has_ramp ? lastPos + 8 : pos2;
    }
    float v = lvl;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        d[k] = v;
        v *= inc;
    }
}

// CalcGainEnergyScale (atrac3denc.cpp:175-224), the `Frame` value the psychoacoustics and the allocator use: ratio of
// the frame's energy without and with gain modulation. One wavefront per (stream, frame). The value is 1 unless this
// block's or the previous block's curve is non-empty, which is the case for a few per cent of the bands; those need
// five strictly ordered 256-term sums: three over this block (carried overlap, windowed original, windowed modulated)
// and two over the previous one (its "next overlap" scale, which the reference carries forward as
// PrevOverlapGainScale). The frame's modulated bands are taken two at a time: all 64 lanes produce the 2 x 5 x 256
// terms (four samples each), then ten lanes run the ten ordered sums side by side.
#ifndef GES_SPLIT
#define GES_SPLIT 1
#endif
constexpr int kGesSplit = GES_SPLIT;       // workgroups per frame (1, 2 or 4), each taking 8 / kGesSplit of its bands: the pairs of a workgroup run one after another
constexpr int kGesBands = 8 / kGesSplit;
constexpr int kGesRow = 260;   // row stride of the term lists: the ten chain lanes read ten rows at once, in different banks
__global__ __launch_bounds__(64) void k_gain_energy_scale(FrontParams p, const Tables* T, int n_frames_total)
{
    // FOUR ordered sums per band, not the reference's five: its prevStoredEnergy (sum of prevOverlap[i]^2, prevOverlap[i] = EncodeWindow[i] x the
    // previous block's modulated sample) and the previous block's nextModulatedEnergy (sum of (mod x winNext)^2) add the squares of the SAME
    // products in the same order - one chain serves both (s0 below). With five rows of terms the block was 10 784 bytes: fifteen workgroups
    // per CU, 3840 slots for the 4096 workgroups of configs[1] - a second round for the last 256; four rows are 8704 bytes: eighteen per CU.
    __shared__ __attribute__((aligned(16))) float s_terms[2][4][kGesRow];
    __shared__ __attribute__((aligned(16))) Curve s_cv[8][2];
    __shared__ float s_gi[32];
    const int lane = threadIdx.x;
    const int sf = blockIdx.x / kGesSplit, part = blockIdx.x % kGesSplit;   // `part`: which kGesBands bands of the frame's eight (channel-major)
    if (sf >= n_frames_total) return;
    const int nfr = p.n_blocks - p.f0;
    const int f = p.f0 + sf % nfr;
    const int s = sf / nfr;
    const int b = f - 1;   // the block this frame's new half comes from
    // lanes 0..7: the band's two curves as 16-byte words; their point lists are later walked from LDS
    uint4 w_cur = {0u, 0u, 0u, 0u}, w_prev = {0u, 0u, 0u, 0u};
    const bool mine = lane >= part * kGesBands && lane < (part + 1) * kGesBands;   // lane = band index c for the curve loads
    if (mine) {
        w_cur = *reinterpret_cast<const uint4*>(p.curves + ((size_t)s * p.n_blocks + f) * 8 + lane);
        w_prev = *reinterpret_cast<const uint4*>((f - 1 < 0) ? &p.state[(size_t)s * 8 + lane].prev_curve
                                                             : p.curves + ((size_t)s * p.n_blocks + (f - 1)) * 8 + lane);
        *reinterpret_cast<uint4*>(&s_cv[lane][0]) = w_cur;
        *reinterpret_cast<uint4*>(&s_cv[lane][1]) = w_prev;
    }
    const uint32_t active = (uint32_t)__ballot(mine && (((w_cur.x | w_prev.x) & 0xffu) != 0u));   // Curve::n is the first byte
    float* out8 = p.ges + ((size_t)s * p.n_blocks + f) * 8;
    if (mine && !((active >> lane) & 1u)) out8[lane] = 1.0f;   // no modulation on either side: every ratio is exactly 1
    if (active == 0u || p.debug == 3) return;
    // (most frames leave above: the tables are fetched only by those that modulate something)
    if (lane < 32) s_gi[lane] = T->gain_interp[lane < 31 ? lane : 30];
    const float4 wn4 = *reinterpret_cast<const float4*>(T->enc_win + 4 * lane);         // EncodeWindow[i], i = 4 lane + k
    const float4 wr4 = *reinterpret_cast<const float4*>(T->enc_win + 252 - 4 * lane);   // EncodeWindow[255 - i] = wr4[3 - k]
    const float wn[4] = {wn4.x, wn4.y, wn4.z, wn4.w}, wc[4] = {wr4.w, wr4.z, wr4.y, wr4.x};
    wave_sync();
    const size_t sublen = (size_t)(p.n_blocks + 2) * 256;
    const int cj = lane >> 2, kk = lane & 3;   // chain lanes 0..7: band cj of the pair, sum kk
    uint32_t todo = active;
    while (todo) {   // wave-uniform
        int cpair[2];
        cpair[0] = __builtin_ctz(todo);
        todo &= todo - 1u;
        cpair[1] = todo ? __builtin_ctz(todo) : -1;
        if (todo) todo &= todo - 1u;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = cpair[j];
            if (c < 0) continue;
            const int ch = c >> 2, band = c & 3;
            const float* q0 = p.sub + ((size_t)s * 8 + band) * sublen + (size_t)(b + 2) * 256 + 4 * lane;       // left / own channel, current block
            const float* q1 = p.sub + ((size_t)s * 8 + 4 + band) * sublen + (size_t)(b + 2) * 256 + 4 * lane;   // right
            float4 l = *reinterpret_cast<const float4*>(ch ? q1 : q0), lp = *reinterpret_cast<const float4*>((ch ? q1 : q0) - 256);
            if (p.js) {   // M/S matrixing (atrac3denc.cpp:665-677)
                const float4 a0 = *reinterpret_cast<const float4*>(q0), a1 = *reinterpret_cast<const float4*>(q1);
                const float4 b0 = *reinterpret_cast<const float4*>(q0 - 256), b1 = *reinterpret_cast<const float4*>(q1 - 256);
                if (ch) {
                    l.x = (a0.x - a1.x) * 0.5f; l.y = (a0.y - a1.y) * 0.5f; l.z = (a0.z - a1.z) * 0.5f; l.w = (a0.w - a1.w) * 0.5f;
                    lp.x = (b0.x - b1.x) * 0.5f; lp.y = (b0.y - b1.y) * 0.5f; lp.z = (b0.z - b1.z) * 0.5f; lp.w = (b0.w - b1.w) * 0.5f;
                } else {
                    l.x = (a0.x + a1.x) * 0.5f; l.y = (a0.y + a1.y) * 0.5f; l.z = (a0.z + a1.z) * 0.5f; l.w = (a0.w + a1.w) * 0.5f;
                    lp.x = (b0.x + b1.x) * 0.5f; lp.y = (b0.y + b1.y) * 0.5f; lp.z = (b0.z + b1.z) * 0.5f; lp.w = (b0.w + b1.w) * 0.5f;
                }
            }
            const float xc[4] = {l.x, l.y, l.z, l.w}, xp[4] = {lp.x, lp.y, lp.z, lp.w};
            const bool hi_half = lane & 1;
            // the lane's four samples are one half of cell lane / 2: that half's divisors. A band mostly has ONE of its two curves (a transient's
            // curve is `cur` for one frame and `prev` for the next): the other side's walk and divisions are skipped wave-uniformly (x / 1.0f is x)
            float dc4[4] = {1.0f, 1.0f, 1.0f, 1.0f}, dp4[4] = {1.0f, 1.0f, 1.0f, 1.0f};
            uint4 wc4 = *reinterpret_cast<const uint4*>(&s_cv[c][0]), wp4 = *reinterpret_cast<const uint4*>(&s_cv[c][1]);
            // (one band's curves for the whole wavefront: as scalars the point walk's byte fields and bounds cost no vector instructions)
            wc4.x = __builtin_amdgcn_readfirstlane(wc4.x); wc4.y = __builtin_amdgcn_readfirstlane(wc4.y);
            wc4.z = __builtin_amdgcn_readfirstlane(wc4.z); wc4.w = __builtin_amdgcn_readfirstlane(wc4.w);
            wp4.x = __builtin_amdgcn_readfirstlane(wp4.x); wp4.y = __builtin_amdgcn_readfirstlane(wp4.y);
            wp4.z = __builtin_amdgcn_readfirstlane(wp4.z); wp4.w = __builtin_amdgcn_readfirstlane(wp4.w);
            const bool mod_c = (wc4.x & 0xffu) != 0u, mod_p = (wp4.x & 0xffu) != 0u;   // wave-uniform
            // (selected with static indices: `d[4 hi_half + k]` is a run-time index and sends the arrays to scratch memory)
            if (mod_c) {
                float d8[8];
                cell_divisors_packed((uint64_t)wc4.x | ((uint64_t)wc4.y << 32), (uint64_t)wc4.z | ((uint64_t)wc4.w << 32), s_gi, 8 * (lane >> 1), d8);
                dc4[0] = hi_half ? d8[4] : d8[0]; dc4[1] = hi_half ? d8[5] : d8[1]; dc4[2] = hi_half ? d8[6] : d8[2]; dc4[3] = hi_half ? d8[7] : d8[3];
            }
            if (mod_p) {
                float d8[8];
                cell_divisors_packed((uint64_t)wp4.x | ((uint64_t)wp4.y << 32), (uint64_t)wp4.z | ((uint64_t)wp4.w << 32), s_gi, 8 * (lane >> 1), d8);
                dp4[0] = hi_half ? d8[4] : d8[0]; dp4[1] = hi_half ? d8[5] : d8[1]; dp4[2] = hi_half ? d8[6] : d8[2]; dp4[3] = hi_half ? d8[7] : d8[3];
            }
            float mc4[4] = {xc[0], xc[1], xc[2], xc[3]}, mp4[4] = {xp[0], xp[1], xp[2], xp[3]};
            if (mod_c) {
#pragma unroll
                for (int k = 0; k < 4; ++k) mc4[k] = xc[k] / dc4[k];
            }
            if (mod_p) {
#pragma unroll
                for (int k = 0; k < 4; ++k) mp4[k] = xp[k] / dp4[k];
            }
            float t[4][4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float mc = mc4[k], mp = mp4[k];
                const float pv = wn[k] * mp;                // the overlap this block inherited: EncodeWindow[i] * modulated sample
                const float cw = xc[k] * wc[k], mw = mc * wc[k], nw = xp[k] * wn[k];   // (mod x winNext of the previous block IS pv: multiplication commutes)
                t[0][k] = pv * pv;
                t[1][k] = cw * cw;
                t[2][k] = mw * mw;
                t[3][k] = nw * nw;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) *reinterpret_cast<float4*>(&s_terms[j][r][4 * lane]) = float4{t[r][0], t[r][1], t[r][2], t[r][3]};
        }
        wave_sync();
        const int cc = cj == 0 ? cpair[0] : (cj == 1 ? cpair[1] : -1);
        const bool chain = lane < 8 && cc >= 0;
        float acc = 0.0f;
        if (chain) {
            const float4* t4 = reinterpret_cast<const float4*>(s_terms[cj][kk]);
            for (int q0 = 0; q0 < 64; q0 += 16) {
                float4 v[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = t4[q0 + q];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    acc += v[q].x;
                    acc += v[q].y;
                    acc += v[q].z;
                    acc += v[q].w;
                }
            }
        }
        // the four sums of a band sit in lanes 4 cj .. 4 cj + 3; the first of them closes the formula
        const float s1 = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(4 * (lane + 1), (int)__float_as_uint(acc)));
        const float s2 = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(4 * (lane + 2), (int)__float_as_uint(acc)));
        const float s3 = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(4 * (lane + 3), (int)__float_as_uint(acc)));
        if (chain && kk == 0) {
            const Curve& q_cur = s_cv[cc][0];
            const bool h_cur = q_cur.n > 0, h_prev = s_cv[cc][1].n > 0;
            const float s0 = acc;
            // PrevOverlapGainScale: the previous block's NextOverlapScale, 1 when that block had no curve (equal sums)
            float ps = h_prev ? safe_energy_scale(s3, s0) : 1.0f;   // (nextModulatedEnergy of the previous block == s0, see s_terms)
            float frame_scale = 1.0f;
            if (h_cur || ps != 1.0f) {
                if (!isfinite(ps) || ps <= 0.0f) ps = 1.0f;
                const float prevDiv = h_cur ? gain_level_of(q_cur.level[0]) : 1.0f;
                const float prevOrig = s0 * ps;
                const float prevMod = s0 / (prevDiv * prevDiv);
                frame_scale = safe_energy_scale(prevOrig + s1, prevMod + s2);
            }
            out8[cc] = frame_scale;
        }
        wave_sync();   // the term lists are rewritten by the next pair
    }
}

// Batched TAtrac3MDCT::Mdct on caller-provided band buffers (atrac3denc.h:80-86): one workgroup per item.
struct MdctItemsParams {
    float* bands;          // [n][4][512] in/out
    float* specs;          // [n][1024] out
    const int32_t* n_points;  // [n][4] or null
    const int32_t* level;     // [n][4][8]
    const int32_t* loc;       // [n][4][8]
    float* max_levels;        // [n][4] or null: max |new half| after modulation (the maxLevels overload, atrac3denc.cpp:33-58)
};

__global__ __launch_bounds__(128) void k_mdct_items(MdctItemsParams p, const Tables* T)
{
    __shared__ float s_tmp[4 * 512];
    __shared__ cpx s_fft[4 * 128];
    __shared__ cpx s_tw[128];
    __shared__ Curve s_curve[4];
    const int tid = threadIdx.x;
    const int c = tid >> 5, lane = tid & 31;  // band
    const size_t item = blockIdx.x;
    float* band = p.bands + (item * 4 + c) * 512;
    if (tid < 4) {
        Curve cv;
        cv.n = 0;
        if (p.n_points) {
            cv.n = (uint8_t)p.n_points[item * 4 + tid];
            for (int i = 0; i < cv.n && i < 7; ++i) {
                cv.level[i] = (uint8_t)p.level[(item * 4 + tid) * 8 + i];
                cv.loc[i] = (uint8_t)p.loc[(item * 4 + tid) * 8 + i];
            }
        }
        s_curve[tid] = cv;
    }
    s_tw[tid] = T->tw128[tid];
    __syncthreads();
    const bool has_curve = s_curve[c].n > 0;
    const float scale = has_curve ? gain_level_of(s_curve[c].level[0]) : 1.0f;
    float* tmp = s_tmp + c * 512;
    float mx = 0.0f;
    for (int i = lane; i < 256; i += 32) {
        float ov = band[i];
        float v = band[256 + i];
        if (has_curve) {
            ov = ov / scale;
            const float d = curve_divisor(T->gain_interp, s_curve[c], i);
            v = v / d;
            band[256 + i] = v;
        }
        mx = fmaxf(mx, fabsf(v));
        tmp[i] = ov;
        band[i] = T->enc_win[i] * v;
        tmp[256 + i] = T->enc_win[255 - i] * v;
    }
    if (p.max_levels) {   // order-free maximum over the band's 32 lanes (one half of the wavefront)
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
        if (lane == 0) p.max_levels[item * 4 + c] = mx;
    }
    __syncthreads();
    for (int n2 = lane; n2 < 128; n2 += 32) {
        const int n = 2 * n2;
        float r0, i0;
        if (n < 128) {
            r0 = tmp[383 - n] + tmp[384 + n];
            i0 = tmp[128 + n] - tmp[127 - n];
        } else {
            r0 = tmp[383 - n] - tmp[n - 128];
            i0 = tmp[128 + n] + tmp[639 - n];
        }
        const float cc = T->mdct_sincos[n], ss = T->mdct_sincos[n + 1];
        cpx v;
        v.r = r0 * cc + i0 * ss;
        v.i = i0 * cc - r0 * ss;
        s_fft[c * 128 + fft_leaf_pos<128>(n2)] = v;
    }
    __syncthreads();
    fft_lds<128, false>(s_fft, 128, 4, s_tw, tid, 128);
    float* out = p.specs + item * 1024 + c * 256;
    const bool odd = (c & 1);
    for (int n2 = lane; n2 < 128; n2 += 32) {
        const int n = 2 * n2;
        const float r0 = s_fft[c * 128 + n2].r, i0 = s_fft[c * 128 + n2].i;
        const float cc = T->mdct_sincos[n], ss = T->mdct_sincos[n + 1];
        out[odd ? 255 - n : n] = -r0 * cc - i0 * ss;
        out[odd ? n : 255 - n] = -r0 * ss + i0 * cc;
    }
}

// Batched TAtrac3MDCT::CalcGainEnergyScale (atrac3denc.h:75-79, atrac3denc.cpp:175-224) on caller-provided buffers: one
// wavefront per item. Lane j forms the five terms of samples 4j .. 4j+3; lanes 0..4 then run the five strictly ordered
// 256-term sums (stored-overlap energy, windowed original / modulated energy of this half and of the next overlap).
struct GesItemsParams {
    const float* prev_overlap;   // [n][256]
    const float* cur_input;      // [n][256]
    const int32_t* n_points;     // [n] or null (no gain points anywhere)
    const int32_t* level;        // [n][8]
    const int32_t* loc;          // [n][8]
    const float* prev_scale;     // [n] prevOverlapScale
    float* out;                  // [n][4]: Scale.PrevHalf, Scale.CurHalf, Scale.Frame, NextOverlapScale
};

__global__ __launch_bounds__(64) void k_ges_items(GesItemsParams p, const Tables* T)
{
    __shared__ __attribute__((aligned(16))) float s_terms[5][256];
    __shared__ Curve s_cv;
    const int lane = threadIdx.x;
    const size_t item = blockIdx.x;
    if (lane == 0) {
        Curve cv;
        cv.n = 0;
        if (p.n_points) {
            cv.n = (uint8_t)p.n_points[item];
            for (int i = 0; i < cv.n && i < 7; ++i) {
                cv.level[i] = (uint8_t)p.level[item * 8 + i];
                cv.loc[i] = (uint8_t)p.loc[item * 8 + i];
            }
        }
        s_cv = cv;
    }
    __syncthreads();
    const float4 pv = *reinterpret_cast<const float4*>(p.prev_overlap + item * 256 + 4 * lane);
    const float4 cu = *reinterpret_cast<const float4*>(p.cur_input + item * 256 + 4 * lane);
    const float pvv[4] = {pv.x, pv.y, pv.z, pv.w}, cuv[4] = {cu.x, cu.y, cu.z, cu.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = 4 * lane + k;
        const float cur = cuv[k];
        const float mod = cur / curve_divisor(T->gain_interp, s_cv, i);
        const float winCur = T->enc_win[255 - i], winNext = T->enc_win[i];
        const float curWin = cur * winCur, modCurWin = mod * winCur, nextWin = cur * winNext, modNextWin = mod * winNext;
        s_terms[0][i] = pvv[k] * pvv[k];
        s_terms[1][i] = curWin * curWin;
        s_terms[2][i] = modCurWin * modCurWin;
        s_terms[3][i] = nextWin * nextWin;
        s_terms[4][i] = modNextWin * modNextWin;
    }
    __syncthreads();
    float acc = 0.0f;
    if (lane < 5) {
        const float4* t4 = reinterpret_cast<const float4*>(s_terms[lane]);
        for (int q = 0; q < 64; ++q) {
            const float4 v = t4[q];
            acc += v.x;
            acc += v.y;
            acc += v.z;
            acc += v.w;
        }
    }
    const float prevStored = __shfl(acc, 0, 64), curOrig = __shfl(acc, 1, 64), curMod = __shfl(acc, 2, 64);
    const float nextOrig = __shfl(acc, 3, 64), nextMod = __shfl(acc, 4, 64);
    if (lane == 0) {
        float ps = p.prev_scale[item];
        if (!isfinite(ps) || ps <= 0.0f) ps = 1.0f;
        const float prevDiv = s_cv.n > 0 ? gain_level_of(s_cv.level[0]) : 1.0f;
        const float prevOrig = prevStored * ps;
        const float prevMod = prevStored / (prevDiv * prevDiv);
        float4 o;
        o.x = safe_energy_scale(prevOrig, prevMod);
        o.y = safe_energy_scale(curOrig, curMod);
        o.z = safe_energy_scale(prevOrig + curOrig, prevMod + curMod);
        o.w = safe_energy_scale(nextOrig, nextMod);
        *reinterpret_cast<float4*>(p.out + item * 4) = o;
    }
}

}  // namespace at3
