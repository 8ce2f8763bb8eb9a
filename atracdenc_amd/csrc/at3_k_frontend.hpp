// Front-end kernels: QMF subband analysis and the fused QMF + gain modulation + windowed MDCT-512.
//
// Reference path replaced (paths relative to the reference's src/):
//   atrac3denc.cpp:701-713   PCM de-interleave, /4.0, Atrac3AnalysisFilterBank::Analysis
//   qmf/qmf.h:47-64          TQmf<nIn>::Analysis (48-tap two-band QMF), atrac/at3/atrac3_qmf.h:37-41 (tree)
//   atrac3denc.cpp:665-677   Matrixing (LP4 joint stereo)
//   atrac3denc.cpp:175-224   CalcGainEnergyScale
//   gain_processor.h:87-121  TGainProcessor::Modulate
//   atrac3denc.cpp:33-58     TAtrac3MDCT::Mdct;  lib/mdct/mdct.h:51-104 TMDCT<512>;  kiss_fft.c (128-pt)
//
// Work decomposition of the fused kernel: one 256-thread workgroup owns one stream and a run of consecutive frames,
// both channels. Per block it stores the interleaved PCM tile (prefetched one block ahead into registers with
// coalesced float2 loads) into LDS rings, runs the two QMF stages out of LDS with the taps in scalar registers and
// packed fp32 arithmetic, and - for frames - modulates, windows and transforms the four subbands of both channels as
// eight concurrent 128-point FFTs (32 lanes each). FIR histories and the previous block's windowed overlap stay in
// LDS between frames, so each PCM sample is read from HBM once per workgroup run (plus the run's two priming blocks)
// and spectra are written once: algorithmic traffic is 16 KiB per frame. CalcGainEnergyScale is its own kernel
// (k_gain_energy_scale); k_qmf_sub is the QMF-only variant that feeds the gain-control kernels.
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
    float* sub;              // k_qmf_sub only: [S][2][4][(n_blocks+2)*256]
    int n_blocks;
    int f0;                  // first frame index to emit (1 on the first call of a stream, else 0)
    int frames_per_wg;
    int js;
    int sub_blocks_per_wg;   // k_qmf_sub only
    int debug;               // profiling aid (env AT3HIP_DEBUG_FRONT): 1 = skip the energy-scale chains, 2 = ignore curves
};

// Modulated new half of one band (TGainProcessor::Modulate, gain_processor.h:93-112) for the eight samples of cell
// `cell / 8`. Level boundaries and the 8-sample ramps are aligned to these cells, so a cell is untouched, divided by
// one level (a power of two: multiplying by its reciprocal is the same rounding) or by one running-product ramp.
// `cv` should be read in place (LDS / global): a private copy indexed in a loop would live in scratch.
__device__ __forceinline__ void modulate_cell(const Curve& cv, const float* gain_interp, int cell, float (&v)[8])
{
    int kind = 0;   // 0 untouched, 1 constant level, 2 ramp
    float lvl = 1.0f, inc = 1.0f, inv = 1.0f;
    int pos = 0;
    for (int q = 0; q < cv.n; ++q) {
        const int lastPos = (int)cv.loc[q] << 3;
        if (cell >= pos && cell < lastPos) {
            kind = 1;
            inv = __uint_as_float((uint32_t)(127 - 4 + cv.level[q]) << 23);   // 1 / GainLevel
            break;
        }
        if (lastPos > pos) pos = lastPos;
        if (pos < lastPos + 8) {
            if (cell >= pos && cell < lastPos + 8) {
                kind = 2;
                lvl = gain_level_of(cv.level[q]);
                inc = gain_interp[((q + 1) < cv.n ? (int)cv.level[q + 1] : 4) - (int)cv.level[q] + 15];
                break;
            }
            pos = lastPos + 8;
        }
    }
    if (kind == 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = v[k] * inv;
    } else if (kind == 2) {
        float d = lvl;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[k] = v[k] / d;
            d *= inc;
        }
    }
}

// CalcGainEnergyScale (atrac3denc.cpp:175-224), the `Frame` value the psychoacoustics and the allocator use: ratio of
// the frame's energy without and with gain modulation. One wavefront per (stream, frame), all eight bands at once. The
// value is 1 unless this block's or the previous block's curve is non-empty, which is the case for a few per cent of
// the bands (but then often for several bands of the same frame); those need five strictly ordered 256-term sums: three
// over this block (carried overlap, windowed original, windowed modulated) and two over the previous one (its "next
// overlap" scale, which the reference carries forward as PrevOverlapGainScale).
// Lane (band = lane / 8, slot = lane % 8) produces, in four rounds of 64 samples, the terms of cell 8 round + slot of
// its band for both blocks; lanes 0..39 = (band, sum) then extend the 40 ordered sums by 64 terms each.
__global__ __launch_bounds__(64) void k_gain_energy_scale(FrontParams p, const Tables* T, int n_frames_total)
{
    __shared__ __attribute__((aligned(16))) float s_terms[1][8][5][64];   // one wavefront per workgroup: 10 KB, so that
    __shared__ __attribute__((aligned(16))) Curve s_cv[1][8][2];           // every frame of a 4096-frame batch is resident at once
    __shared__ __attribute__((aligned(16))) float s_win[256];
    const int wave = 0, lane = threadIdx.x;
    *reinterpret_cast<float4*>(s_win + 4 * lane) = *reinterpret_cast<const float4*>(T->enc_win + 4 * lane);
    const int sf = blockIdx.x;
    if (sf >= n_frames_total) return;
    const int nfr = p.n_blocks - p.f0;
    const int f = p.f0 + sf % nfr;
    const int s = sf / nfr;
    const int b = f - 1;   // the block this frame's new half comes from
    // lanes 0..7: the band's two curves as 16-byte words; their point lists are later walked from LDS
    uint4 w_cur = {0u, 0u, 0u, 0u}, w_prev = {0u, 0u, 0u, 0u};
    if (lane < 8) {
        w_cur = *reinterpret_cast<const uint4*>(p.curves + ((size_t)s * p.n_blocks + f) * 8 + lane);
        w_prev = *reinterpret_cast<const uint4*>((f - 1 < 0) ? &p.state[(size_t)s * 8 + lane].prev_curve
                                                             : p.curves + ((size_t)s * p.n_blocks + (f - 1)) * 8 + lane);
        *reinterpret_cast<uint4*>(&s_cv[wave][lane][0]) = w_cur;
        *reinterpret_cast<uint4*>(&s_cv[wave][lane][1]) = w_prev;
    }
    const uint32_t active = (uint32_t)__ballot(lane < 8 && (((w_cur.x | w_prev.x) & 0xffu) != 0u));   // Curve::n is the first byte
    float* out8 = p.ges + ((size_t)s * p.n_blocks + f) * 8;
    if (lane < 8 && !((active >> lane) & 1u)) out8[lane] = 1.0f;   // no modulation on either side: every ratio is exactly 1
    if (active == 0u || p.debug == 3) return;
    wave_sync();
    const int c = lane >> 3, slot = lane & 7;
    const bool on = (active >> c) & 1u;
    const Curve& cv_cur = s_cv[wave][c][0];
    const Curve& cv_prev = s_cv[wave][c][1];
    const bool has_cur = cv_cur.n > 0, has_prev = cv_prev.n > 0;
    const int ch = c >> 2, band = c & 3;
    const size_t sublen = (size_t)(p.n_blocks + 2) * 256;
    const float* sb0 = p.sub + ((size_t)s * 8 + band) * sublen + (size_t)(b + 2) * 256;       // left / own channel, current block
    const float* sb1 = p.sub + ((size_t)s * 8 + 4 + band) * sublen + (size_t)(b + 2) * 256;   // right
    float acc = 0.0f;
    const int cc = lane / 5, kk = lane % 5;   // chain lanes: band cc, sum kk
    const bool chain = lane < 40 && ((active >> cc) & 1u);
    // all samples this lane will need (four cells of the current and of the previous block), fetched up front
    float xc[4][8], xp[4][8];
    if (on) {
#pragma unroll
        for (int rd = 0; rd < 4; ++rd) {
            const int cell = 8 * (8 * rd + slot);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float* q0 = sb0 + cell + 4 * h;
                const float* q1 = sb1 + cell + 4 * h;
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
                xc[rd][4 * h] = l.x; xc[rd][4 * h + 1] = l.y; xc[rd][4 * h + 2] = l.z; xc[rd][4 * h + 3] = l.w;
                xp[rd][4 * h] = lp.x; xp[rd][4 * h + 1] = lp.y; xp[rd][4 * h + 2] = lp.z; xp[rd][4 * h + 3] = lp.w;
            }
        }
    }
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {
        if (on) {
            const int cell = 8 * (8 * rd + slot);
            float mc[8], mp[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                mc[k] = xc[rd][k];
                mp[k] = xp[rd][k];
            }
            if (has_cur) modulate_cell(cv_cur, T->gain_interp, cell, mc);
            if (has_prev) modulate_cell(cv_prev, T->gain_interp, cell, mp);
            float (*terms)[64] = s_terms[wave][c];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = cell + k, o = 8 * slot + k;
                const float wc = s_win[255 - i], wn = s_win[i];
                const float pv = wn * mp[k];                // the overlap this block inherited: EncodeWindow[i] * modulated sample
                const float cw = xc[rd][k] * wc, mw = mc[k] * wc, nw = xp[rd][k] * wn, mnw = mp[k] * wn;
                terms[0][o] = pv * pv;
                terms[1][o] = cw * cw;
                terms[2][o] = mw * mw;
                terms[3][o] = nw * nw;
                terms[4][o] = mnw * mnw;
            }
        }
        wave_sync();
        if (chain) {
            const float4* t4 = reinterpret_cast<const float4*>(s_terms[wave][cc][kk]);
            float4 v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = t4[q];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                acc += v[q].x;
                acc += v[q].y;
                acc += v[q].z;
                acc += v[q].w;
            }
        }
        wave_sync();   // the term buffers are rewritten by the next round
    }
    // the five sums of band cc sit in lanes 5 cc .. 5 cc + 4; the first of them closes the formula
    const float s1 = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(4 * (lane + 1), (int)__float_as_uint(acc)));
    const float s2 = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(4 * (lane + 2), (int)__float_as_uint(acc)));
    const float s3 = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(4 * (lane + 3), (int)__float_as_uint(acc)));
    const float s4 = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(4 * (lane + 4), (int)__float_as_uint(acc)));
    if (chain && kk == 0) {
        const Curve& q_cur = s_cv[wave][cc][0];
        const bool h_cur = q_cur.n > 0, h_prev = s_cv[wave][cc][1].n > 0;
        const float s0 = acc;
        // PrevOverlapGainScale: the previous block's NextOverlapScale, 1 when that block had no curve (equal sums)
        float ps = h_prev ? safe_energy_scale(s3, s4) : 1.0f;
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
}

// ---- fused QMF + gain modulation + windowed MDCT-512 ------------------------------------------------------
//
// Register-blocked FIR: one work-item produces four consecutive (lower, upper) output pairs of one two-band
// filter. Output m uses the sample pairs (x[2p], x[2p+1]) for p = m-23 .. m, so four outputs share 27 pairs
// that are fetched from LDS with seven 16-byte reads and then stay in registers for all 192 multiply-adds;
// the 48 taps are wave-uniform scalars. Accumulation order per output is tap 0..23, multiply then add
// (no contraction), exactly as qmf.h:54-63.
template <int H>
__device__ __forceinline__ void qmf4(const float4* __restrict__ xb /* LDS ring + work-item index g, in 16-byte slots */,
                                     const f2 (&Wp)[24] /* tap pairs (W[2i], W[2i+1]), wave-uniform (scalar registers) */,
                                     float (&lower)[4], float (&upper)[4])
{
    // sample pairs arrive as (x[2k+1], x[2k]) in the register pairs the 16-byte loads deliver (see ring_at); one
    // packed multiply forms (W[2i] x[2k+1], W[2i+1] x[2k]) and one packed add extends the two ordered sums of
    // qmf.h:59-66 together.
    // Pair k feeds tap i = r + 23 - k of output r: walking k downwards extends every sum in tap order while each
    // 16-byte group is needed only around its own two steps - the loads trail the arithmetic instead of filling 56
    // registers up front.
    f2 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = mk2(0.0f, 0.0f);
#pragma unroll
    for (int q = 13; q >= 0; --q) {
        const float4 v = xb[(q >> 1) + (q & 1) * H];
        const f2 hi = mk2(v.z, v.w), lo = mk2(v.x, v.y);   // pairs 2q+1, 2q
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = r + 23 - (2 * q + 1);
            if (i >= 0 && i < 24) acc[r] = acc[r] + Wp[i] * hi;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = r + 23 - 2 * q;
            if (i >= 0 && i < 24) acc[r] = acc[r] + Wp[i] * lo;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        lower[r] = acc[r].x + acc[r].y;
        upper[r] = acc[r].x - acc[r].y;
    }
}

// Ring storage order. Work-item g of a QMF stage reads the 16-byte groups 2g .. 2g+13 of its ring; with the groups in
// natural order neighbouring work-items are 32 bytes apart and every 16-byte LDS read is a two-way bank conflict.
// The rings therefore keep even groups in their first half and odd groups in the second (H groups each, H = 8 mod 16
// so that the halves are 128 bytes out of phase): group j lives in slot (j >> 1) + (j & 1) * H, and a stage reads slot
// g + (q >> 1) + (q & 1) * H for q = 0..13 - consecutive work-items, consecutive slots.
constexpr int kPcmH = 136, kS1H = 72;   // slots per half: logical floats [46 history | 1024 new] resp. [46 | 512]
constexpr int kPcmRing = 8 * kPcmH;     // 1088 floats per channel
constexpr int kS1Ring = 8 * kS1H;       // 576 floats per channel and half
// Inside a group the two floats of a sample pair are stored swapped, (x[2k+1], x[2k]): that is the operand order of the
// packed tap product (W[2i] x[2k+1], W[2i+1] x[2k]), so the FIR consumes the loaded register pairs as they are.
template <int H>
__device__ __forceinline__ int ring_at(int e)   // physical float index of logical ring element e
{
    const int j = e >> 2;
    return (((j >> 1) + (j & 1) * H) << 2) | ((e & 3) ^ 1);
}

// Subband analysis only (feeds the gain-control kernels): raw L/R subbands of blocks -2 .. n_blocks-1, the same ring
// scheme as the fused kernel below. One workgroup walks `sub_blocks_per_wg` consecutive blocks of one stream; the
// four subbands leave stage 2 in registers and go to HBM as 16-byte stores.
__global__ __launch_bounds__(256) void k_qmf_sub(FrontParams p, const Tables* T)
{
    __shared__ __attribute__((aligned(16))) float s_pcm[2 * kPcmRing];
    __shared__ __attribute__((aligned(16))) float s_lo[2 * kS1Ring];
    __shared__ __attribute__((aligned(16))) float s_hi[2 * kS1Ring];
    const int tid = threadIdx.x;
    const int nb2 = p.n_blocks + 2;
    const int nchunks = (nb2 + p.sub_blocks_per_wg - 1) / p.sub_blocks_per_wg;
    const int s = blockIdx.x / nchunks;
    const int chunk = blockIdx.x % nchunks;
    const int ba = -2 + chunk * p.sub_blocks_per_wg;
    int bb = ba + p.sub_blocks_per_wg;
    if (bb > p.n_blocks) bb = p.n_blocks;
    f2 Wp[24];
#pragma unroll
    for (int i = 0; i < 24; ++i)
        Wp[i] = mk2(__uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(T->qmf_win[2 * i]))),
                    __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(T->qmf_win[2 * i + 1]))));
    const float2* pcm2 = reinterpret_cast<const float2*>(p.pcm) + (size_t)s * p.n_blocks * 1024;
    const float2* hist2 = reinterpret_cast<const float2*>(p.hist) + (size_t)s * kHist;

    // prologue: stage-1 outputs m = -46..-1 of the first block from samples -138..-1
    for (int k = tid; k < 138; k += 256) {
        const int g = ba * 1024 - 138 + k;
        const float2 v = (g >= 0) ? pcm2[g] : hist2[kHist + g];
        s_pcm[k] = v.x * 0.25f;
        s_pcm[kPcmRing + k] = v.y * 0.25f;
    }
    __syncthreads();
    float keep = 0.0f;
    if (tid < 92) {
        const int ch = tid / 46, mm = tid % 46;
        const float* x = s_pcm + ch * kPcmRing + 2 * mm;
        float lo = 0.0f, hi = 0.0f;
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            lo += Wp[i].x * x[47 - 2 * i];
            hi += Wp[i].y * x[46 - 2 * i];
        }
        s_lo[ch * kS1Ring + ring_at<kS1H>(mm)] = lo + hi;
        s_hi[ch * kS1Ring + ring_at<kS1H>(mm)] = lo - hi;
        keep = s_pcm[ch * kPcmRing + 92 + mm];
    }
    __syncthreads();
    if (tid < 92) s_pcm[(tid / 46) * kPcmRing + ring_at<kPcmH>(tid % 46)] = keep;

    float2 nxt[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int g = ba * 1024 + tid + 256 * q;
        nxt[q] = (g >= 0) ? pcm2[g] : hist2[kHist + g];
    }
    const size_t sublen = (size_t)nb2 * 256;
    for (int b = ba; b < bb; ++b) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = tid + 256 * q;
            s_pcm[ring_at<kPcmH>(46 + k)] = nxt[q].x * 0.25f;       // data / 4.0 (exact)
            s_pcm[kPcmRing + ring_at<kPcmH>(46 + k)] = nxt[q].y * 0.25f;
        }
        if (b + 1 < bb) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int g = (b + 1) * 1024 + tid + 256 * q;
                nxt[q] = (g >= 0) ? pcm2[g] : hist2[kHist + g];
            }
        }
        __syncthreads();
        {   // stage 1 (Qmf1)
            const int ch = tid >> 7, g = tid & 127;
            float lw[4], up[4];
            qmf4<kPcmH>(reinterpret_cast<const float4*>(s_pcm + ch * kPcmRing) + g, Wp, lw, up);
            // ring element of output m is 46 + m: 8-byte aligned pairs, (46 + 4g, +1) and (48 + 4g, +1) sit in two groups
            float* rl = s_lo + ch * kS1Ring;
            float* rh = s_hi + ch * kS1Ring;
            const int e0 = ring_at<kS1H>(47 + 4 * g), e1 = ring_at<kS1H>(49 + 4 * g);   // pair bases: the odd element comes first
            float2 t0, t1;
            t0.x = lw[1]; t0.y = lw[0]; t1.x = lw[3]; t1.y = lw[2];
            *reinterpret_cast<float2*>(rl + e0) = t0; *reinterpret_cast<float2*>(rl + e1) = t1;
            t0.x = up[1]; t0.y = up[0]; t1.x = up[3]; t1.y = up[2];
            *reinterpret_cast<float2*>(rh + e0) = t0; *reinterpret_cast<float2*>(rh + e1) = t1;
        }
        __syncthreads();
        if (tid < 92) keep = s_pcm[(tid / 46) * kPcmRing + ring_at<kPcmH>(1024 + tid % 46)];
        {   // stage 2: Qmf2 on the lower half -> bands 0, 1; Qmf3 on the upper half -> bands 3, 2
            const int ch = tid >> 7, which = (tid >> 6) & 1, g = tid & 63;
            float lw[4], up[4];
            qmf4<kS1H>(reinterpret_cast<const float4*>((which ? s_hi : s_lo) + ch * kS1Ring) + g, Wp, lw, up);
            float4 a, bq;
            a.x = lw[0]; a.y = lw[1]; a.z = lw[2]; a.w = lw[3];
            bq.x = up[0]; bq.y = up[1]; bq.z = up[2]; bq.w = up[3];
            float* out = p.sub + ((size_t)s * 8 + ch * 4) * sublen + (size_t)(b + 2) * 256 + 4 * g;
            *reinterpret_cast<float4*>(out + (which ? 3 : 0) * sublen) = a;
            *reinterpret_cast<float4*>(out + (which ? 2 : 1) * sublen) = bq;
        }
        if (tid < 92) s_pcm[(tid / 46) * kPcmRing + ring_at<kPcmH>(tid % 46)] = keep;
        __syncthreads();
        if (tid < 184) {   // stage-1 history for the next block
            const int hlf = tid / 92, r = tid % 92, ch = r / 46, k = r % 46;
            float* ring = (hlf ? s_hi : s_lo) + ch * kS1Ring;
            ring[ring_at<kS1H>(k)] = ring[ring_at<kS1H>(512 + k)];
        }
    }
}

// FFT buffer layout of the fused kernel: one spare complex slot after every 8, 144 slots per 128-point transform. The
// radix-4 passes with butterfly distance m = 2 and m = 8 put a half-wave's 8-byte accesses 64 B / 256 B apart; the
// padding spreads them over the banks (an 8-way conflict becomes conflict free, a 4-way one 2-way).
constexpr int kFftSlot = 144;
__device__ __forceinline__ constexpr int fft_pad(int i) { return i + (i >> 3); }

template <bool GAIN>
__global__ __launch_bounds__(256, 4) void k_qmf_mdct(FrontParams p, const Tables* T)
{
    __shared__ __attribute__((aligned(16))) float s_pcm[2 * kPcmRing];
    __shared__ __attribute__((aligned(16))) float s_s1[4 * kS1Ring];      // stage-1 rings: lower halves of both channels, then upper
    __shared__ __attribute__((aligned(16))) float s_sub[2 * 4 * 256];     // current block's subbands [ch][band][256]
    __shared__ __attribute__((aligned(16))) float s_prevw[2 * 4 * 256];   // overlap half carried to the next frame
    float* s_lo = s_s1;
    float* s_hi = s_s1 + 2 * kS1Ring;
    // The stage-1 rings are dead between stage 2 and the next block's stage 1 (their 46-sample histories wait in
    // registers meanwhile), so the eight 128-point FFT buffers of the MDCT phase live in the same storage.
    static_assert(4 * kS1Ring * sizeof(float) >= 8 * kFftSlot * sizeof(cpx), "FFT buffers must fit in the stage-1 rings");
    cpx* s_fft = reinterpret_cast<cpx*>(s_s1);
    __shared__ __attribute__((aligned(16))) float s_win[256];
    __shared__ __attribute__((aligned(8))) float s_cs[256];
    __shared__ cpx s_tw[128];
    __shared__ __attribute__((aligned(16))) Curve s_curve[8];
    __shared__ float s_gi[32];               // GainInterpolation
    // Gain-path scratch aliases buffers that are dead between stage 2 and the MDCT fold of the same block:
    // the modulated samples live in the FFT buffer (written by the fold afterwards), the energy-term staging in
    // each channel's PCM ring behind the 46-sample history (rewritten by the next block's tile load).
    float* s_mod = reinterpret_cast<float*>(s_fft);          // 256 floats at the head of each combo's own FFT slot

    const int tid = threadIdx.x;
    const int nchunks = (p.n_blocks - p.f0 + p.frames_per_wg - 1) / p.frames_per_wg;
    const int s = blockIdx.x / nchunks;
    const int chunk = blockIdx.x % nchunks;
    const int fa = p.f0 + chunk * p.frames_per_wg;
    int fb = fa + p.frames_per_wg;
    if (fb > p.n_blocks) fb = p.n_blocks;
    const int n_out = p.n_blocks - p.f0;
    f2 Wp[24];   // the 48 taps live in scalar register pairs for the whole run
#pragma unroll
    for (int i = 0; i < 24; ++i)
        Wp[i] = mk2(__uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(T->qmf_win[2 * i]))),
                    __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(T->qmf_win[2 * i + 1]))));
    const float2* pcm2 = reinterpret_cast<const float2*>(p.pcm) + (size_t)s * p.n_blocks * 1024;
    const float2* hist2 = reinterpret_cast<const float2*>(p.hist) + (size_t)s * kHist;

    s_win[tid] = T->enc_win[tid];
    s_cs[tid] = T->mdct_sincos[tid];
    if (tid < 128) s_tw[tid] = T->tw128[tid];
    if (GAIN && tid < 32) s_gi[tid] = T->gain_interp[tid];
    for (int i = tid; i < 2048; i += 256) s_prevw[i] = 0.0f;
    // gain curve of the frame about to be processed, fetched one block ahead by work-items 0..7 (frame fa-1 only
    // shapes the carried overlap; frame -1 is the curve carried in the stream state)
    uint4 ncv = {0u, 0u, 0u, 0u};   // one Curve, as the 16 bytes it is
    if (GAIN && tid < 8) {
        const int f = fa - 1;
        ncv = *reinterpret_cast<const uint4*>((f < 0) ? &p.state[(size_t)s * 8 + tid].prev_curve
                                                      : &p.curves[((size_t)s * p.n_blocks + f) * 8 + tid]);
    }

    // ---- prologue: FIR histories of the first block ----
    // stage-1 outputs m = -46..-1 need samples -138..-1; they are computed once per workgroup run. The block before the
    // first frame only primes the MDCT overlap: without gain control it runs through the filter bank like any other
    // (b0 = fa - 2); with gain control its subbands already sit in HBM (k_qmf_sub wrote every block's for the gain
    // analysis, same arithmetic), so the run starts at b0 = fa - 1 and the priming block is read instead of recomputed.
    const int b0 = GAIN ? fa - 1 : fa - 2;
    for (int k = tid; k < 138; k += 256) {
        const int g = b0 * 1024 - 138 + k;
        const float2 v = (g >= 0) ? pcm2[g] : hist2[kHist + g];
        s_pcm[k] = v.x * 0.25f;
        s_pcm[kPcmRing + k] = v.y * 0.25f;
    }
    __syncthreads();
    float keep = 0.0f, keep1 = 0.0f;
    if (tid < 92) {
        const int ch = tid / 46, mm = tid % 46;   // output m = mm - 46, pair base = 2 * mm in the temp layout
        const float* x = s_pcm + ch * kPcmRing + 2 * mm;
        float lo = 0.0f, hi = 0.0f;
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            lo += Wp[i].x * x[47 - 2 * i];
            hi += Wp[i].y * x[46 - 2 * i];
        }
        s_lo[ch * kS1Ring + ring_at<kS1H>(mm)] = lo + hi;
        s_hi[ch * kS1Ring + ring_at<kS1H>(mm)] = lo - hi;
        keep = s_pcm[ch * kPcmRing + 92 + mm];    // samples -46..-1 move to the front of the ring
    }
    __syncthreads();
    if (tid < 92) s_pcm[(tid / 46) * kPcmRing + ring_at<kPcmH>(tid % 46)] = keep;

    const int c = tid >> 5;      // (channel, band) combo owning this thread in the MDCT phase; a wave owns 2
    const int lane = tid & 31;

    float2 nxt[4];   // PCM of the block about to be processed, one block of look-ahead in registers
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int g = b0 * 1024 + tid + 256 * q;
        nxt[q] = (g >= 0) ? pcm2[g] : hist2[kHist + g];
    }
    // The PCM tile of a block is stored while the previous block is between its two QMF stages (the ring's sample
    // area is only read by stage 1), so a block needs three workgroup barriers, not four. Work-items 210..255 hold
    // the last 46 samples of the tile they store: these become the FIR history once stage 1 is done with the ring.
    float2 hv = {0.0f, 0.0f};
    auto store_tile = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = tid + 256 * q;
            s_pcm[ring_at<kPcmH>(46 + k)] = nxt[q].x * 0.25f;                 // data / 4.0 (exact)
            s_pcm[kPcmRing + ring_at<kPcmH>(46 + k)] = nxt[q].y * 0.25f;
        }
        hv.x = nxt[3].x * 0.25f;
        hv.y = nxt[3].y * 0.25f;
    };
    store_tile();
    if (b0 + 1 <= fb - 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int g = (b0 + 1) * 1024 + tid + 256 * q;
            nxt[q] = (g >= 0) ? pcm2[g] : hist2[kHist + g];
        }
    }
    __syncthreads();
    if (GAIN) {
        // priming block fa - 2: subbands from k_qmf_sub, M/S matrixing, modulation by frame fa-1's curve, overlap window
        const size_t sublen = (size_t)(p.n_blocks + 2) * 256;
        for (int i = tid; i < 2048; i += 256)
            s_sub[i] = p.sub[((size_t)s * 8 + (i >> 8)) * sublen + (size_t)fa * 256 + (i & 255)];   // block b lives at (b + 2) * 256
        if (tid < 8) {
            *reinterpret_cast<uint4*>(&s_curve[tid]) = ncv;
            ncv = *reinterpret_cast<const uint4*>(&p.curves[((size_t)s * p.n_blocks + fa) * 8 + tid]);
        }
        __syncthreads();
        if (p.js) {
            for (int idx = tid; idx < 1024; idx += 256) {
                const float l = s_sub[idx], r = s_sub[1024 + idx];
                s_sub[idx] = (l + r) * 0.5f;
                s_sub[1024 + idx] = (l - r) * 0.5f;
            }
            __syncthreads();
        }
        float* xs = s_sub + c * 256;
        if (s_curve[c].n > 0 && p.debug != 2) {
            const Curve& cv = s_curve[c];
            const int cell = 8 * lane;   // the lane's own eight samples: modulated in place
            const float4 xa = *reinterpret_cast<const float4*>(xs + cell), xb = *reinterpret_cast<const float4*>(xs + cell + 4);
            float v[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
            modulate_cell(cv, s_gi, cell, v);
            float4 oa, ob;
            oa.x = v[0]; oa.y = v[1]; oa.z = v[2]; oa.w = v[3];
            ob.x = v[4]; ob.y = v[5]; ob.z = v[6]; ob.w = v[7];
            *reinterpret_cast<float4*>(xs + cell) = oa;
            *reinterpret_cast<float4*>(xs + cell + 4) = ob;
        }
        wave_sync();
        for (int i = lane; i < 256; i += 32) s_prevw[c * 256 + i] = s_win[i] * xs[i];
        __syncthreads();
    }
    // block b carries frame f = b + 1; without gain control the block before the first frame only primes the overlap.
    for (int b = b0; b <= fb - 2; ++b) {
        // Thread-derived addresses and roles are the same for every block of the run; left alone the compiler computes
        // them all in front of the loop and keeps them in registers for the whole kernel. The opaque copy of the thread
        // index makes them per-block work again.
        const int tid_ = opaque_lane_value(tid);
        const int c_ = tid_ >> 5, lane_ = tid_ & 31;
        const int f = b + 1;
        const bool is_frame = (f >= fa);
        if (GAIN && tid_ < 8) {
            *reinterpret_cast<uint4*>(&s_curve[tid_]) = ncv;
            if (b + 1 <= fb - 2) ncv = *reinterpret_cast<const uint4*>(&p.curves[((size_t)s * p.n_blocks + f + 1) * 8 + tid_]);
        }
        if (tid_ < 184 && b > b0) {   // stage-1 histories return to the rings (they shared storage with the FFT buffers)
            const int hlf = tid_ / 92, r = tid_ % 92, ch = r / 46, k = r % 46;
            ((hlf ? s_hi : s_lo) + ch * kS1Ring)[ring_at<kS1H>(k)] = keep1;
        }
        // ---- stage 1 (Qmf1): 2 channels x 128 tasks x 4 outputs ----
        {
            const int ch = tid_ >> 7, g = tid_ & 127;
            float lw[4], up[4];
            qmf4<kPcmH>(reinterpret_cast<const float4*>(s_pcm + ch * kPcmRing) + g, Wp, lw, up);
            float4 a, bq;
            a.x = lw[0]; a.y = lw[1]; a.z = lw[2]; a.w = lw[3];
            bq.x = up[0]; bq.y = up[1]; bq.z = up[2]; bq.w = up[3];
            // ring element of output m is 46 + m: 8-byte aligned pairs, (46 + 4g, +1) and (48 + 4g, +1) sit in two groups
            float* rl = s_lo + ch * kS1Ring;
            float* rh = s_hi + ch * kS1Ring;
            const int e0 = ring_at<kS1H>(47 + 4 * g), e1 = ring_at<kS1H>(49 + 4 * g);   // pair bases: the odd element comes first
            float2 t0, t1;
            t0.x = a.y; t0.y = a.x; t1.x = a.w; t1.y = a.z;
            *reinterpret_cast<float2*>(rl + e0) = t0; *reinterpret_cast<float2*>(rl + e1) = t1;
            t0.x = bq.y; t0.y = bq.x; t1.x = bq.w; t1.y = bq.z;
            *reinterpret_cast<float2*>(rh + e0) = t0; *reinterpret_cast<float2*>(rh + e1) = t1;
        }
        __syncthreads();
        // PCM history for the next block (stage 1 is done with the ring), then the next block's tile and the
        // coalesced float2 loads of the block after it: they land during this block's remaining math
        if (tid_ >= 210) {
            s_pcm[ring_at<kPcmH>(tid_ - 210)] = hv.x;
            s_pcm[kPcmRing + ring_at<kPcmH>(tid_ - 210)] = hv.y;
        }
        if (b + 1 <= fb - 2) {
            store_tile();
            if (b + 2 <= fb - 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int g = (b + 2) * 1024 + tid_ + 256 * q;
                    nxt[q] = (g >= 0) ? pcm2[g] : hist2[kHist + g];
                }
            }
        }
        if (tid_ < 184) {   // stage-1 history for the next block, parked in a register across the MDCT phase
            const int hlf = tid_ / 92, r = tid_ % 92, ch = r / 46, k = r % 46;
            keep1 = ((hlf ? s_hi : s_lo) + ch * kS1Ring)[ring_at<kS1H>(512 + k)];
        }
        // ---- stage 2: Qmf2 on the lower half -> bands 0, 1; Qmf3 on the upper half -> bands 3, 2 ----
        {
            const int ch = tid_ >> 7, which = (tid_ >> 6) & 1, g = tid_ & 63;
            float lw[4], up[4];
            qmf4<kS1H>(reinterpret_cast<const float4*>((which ? s_hi : s_lo) + ch * kS1Ring) + g, Wp, lw, up);
            float4 a, bq;
            a.x = lw[0]; a.y = lw[1]; a.z = lw[2]; a.w = lw[3];
            bq.x = up[0]; bq.y = up[1]; bq.z = up[2]; bq.w = up[3];
            float* out = s_sub + ch * 1024;
            *reinterpret_cast<float4*>(out + (which ? 3 : 0) * 256 + 4 * g) = a;
            *reinterpret_cast<float4*>(out + (which ? 2 : 1) * 256 + 4 * g) = bq;
        }
        __syncthreads();
        if (p.js) {  // M/S matrixing in the subband domain
            for (int idx = tid_; idx < 1024; idx += 256) {
                const float l = s_sub[idx], r = s_sub[1024 + idx];
                s_sub[idx] = (l + r) * 0.5f;         // (l + r) / 2.0, exact halving
                s_sub[1024 + idx] = (l - r) * 0.5f;
            }
            __syncthreads();
        }

        // ======== from here on every wavefront works on its own two (channel, band) combos ========
        float* xs = s_sub + c_ * 256;
        float* pw = s_prevw + c_ * 256;
        bool has_curve = false;
        float scale = 1.0f;
        if (GAIN) {
            has_curve = s_curve[c_].n > 0 && p.debug != 2;
            if (has_curve) {
                // lane_ j owns samples 8j .. 8j+7 of the modulated new half (modulate_cell)
                const Curve& cv = s_curve[c_];
                scale = gain_level_of(cv.level[0]);
                const int cell = 8 * lane_;
                const float4 xa = *reinterpret_cast<const float4*>(xs + cell), xb = *reinterpret_cast<const float4*>(xs + cell + 4);
                float v[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
                modulate_cell(cv, s_gi, cell, v);
                float4 oa, ob;
                oa.x = v[0]; oa.y = v[1]; oa.z = v[2]; oa.w = v[3];
                ob.x = v[4]; ob.y = v[5]; ob.z = v[6]; ob.w = v[7];
                *reinterpret_cast<float4*>(s_mod + c_ * (2 * kFftSlot) + cell) = oa;
                *reinterpret_cast<float4*>(s_mod + c_ * (2 * kFftSlot) + cell + 4) = ob;
            }
            wave_sync();
            // (CalcGainEnergyScale runs in k_gain_energy_scale, from the same subbands and curves)
            if (has_curve) {   // Modulate (gain_processor.h:87-121): new half / ramp, overlap half / first level
                const float inv_scale = 1.0f / scale;   // scale is a power of two
                for (int i = lane_; i < 256; i += 32) {
                    xs[i] = s_mod[c_ * (2 * kFftSlot) + i];
                    pw[i] = pw[i] * inv_scale;
                }
            }
            wave_sync();
        }
        if (is_frame) {
            // MDCT-512 fold + pre-rotation straight from the overlap and the windowed new half
            // (in[k] = overlap[k] for k < 256, EncodeWindow[511 - k] * new[k - 256] otherwise; mdct.h:64-86)
            for (int n2 = lane_; n2 < 128; n2 += 32) {
                const int n = 2 * n2;
                float r0, i0;
                if (n < 128) {
                    r0 = s_win[128 + n] * xs[127 - n] + s_win[127 - n] * xs[128 + n];
                    i0 = pw[128 + n] - pw[127 - n];
                } else {
                    r0 = pw[383 - n] - pw[n - 128];
                    i0 = s_win[383 - n] * xs[n - 128] + s_win[n - 128] * xs[383 - n];
                }
                const f2 csn = *reinterpret_cast<const f2*>(s_cs + n);   // (cos, sin) of this bin as one 8-byte read
                const float cc = csn.x, ss = csn.y;
                cpx v;
                v.r = r0 * cc + i0 * ss;
                v.i = i0 * cc - r0 * ss;
                s_fft[c_ * kFftSlot + fft_pad(fft_leaf_pos<128>(n2))] = v;
            }
        }
        wave_sync();
        // next frame's overlap = EncodeWindow[i] * new[i] (atrac3denc.cpp:47)
        for (int i = lane_; i < 256; i += 32) pw[i] = s_win[i] * xs[i];
        if (is_frame) {
            // 128-point FFT of this combo by its 32 lanes: radix-2 leaves, then three radix-4 passes
            cpx* F = s_fft + c_ * kFftSlot;
            {
                const f2 w = ld2(s_tw);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    cpx* a = F + fft_pad(2 * (lane_ + 32 * q));   // an even index and its successor share a group of 8
                    f2 a0 = ld2(a), a1 = ld2(a + 1);
                    bfly2(a0, a1, w);
                    st2(a, a0);
                    st2(a + 1, a1);
                }
            }
            wave_sync();
#pragma unroll
            for (int m = 2; m < 128; m <<= 2) {
                const int fstride = 128 / (4 * m);
                const int g = lane_ / m, k = lane_ % m;
                const int i0 = g * 4 * m + k;
                cpx *B0 = F + fft_pad(i0), *B1 = F + fft_pad(i0 + m), *B2 = F + fft_pad(i0 + 2 * m), *B3 = F + fft_pad(i0 + 3 * m);
                f2 x0 = ld2(B0), x1 = ld2(B1), x2 = ld2(B2), x3 = ld2(B3);
                bfly4<false>(x0, x1, x2, x3, ld2(s_tw + k * fstride), ld2(s_tw + 2 * k * fstride), ld2(s_tw + 3 * k * fstride));
                st2(B0, x0);
                st2(B1, x1);
                st2(B2, x2);
                st2(B3, x3);
                wave_sync();
            }
            // post-rotation (mdct.h:92-101) in place: read this lane_'s four bins, then scatter the 256 lines
            float oa[4], ob[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n2 = lane_ + 32 * q, n = 2 * n2;
                const float r0 = F[fft_pad(n2)].r, i0 = F[fft_pad(n2)].i;
                const f2 csn = *reinterpret_cast<const f2*>(s_cs + n);   // (cos, sin) of this bin as one 8-byte read
                const float cc = csn.x, ss = csn.y;
                oa[q] = -r0 * cc - i0 * ss;
                ob[q] = -r0 * ss + i0 * cc;
            }
            wave_sync();
            float* out = reinterpret_cast<float*>(F);
            const bool odd = (c_ & 1);   // odd bands are stored reversed (atrac3denc.cpp:53-55)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = 2 * (lane_ + 32 * q);
                out[odd ? 255 - n : n] = oa[q];
                out[odd ? n : 255 - n] = ob[q];
            }
            wave_sync();
            float* dst = p.specs + ((size_t)s * n_out + (f - p.f0)) * 2048 + c_ * 256;
#pragma unroll
            for (int q = 0; q < 2; ++q)
                *reinterpret_cast<float4*>(dst + 4 * (lane_ + 32 * q)) = *reinterpret_cast<const float4*>(out + 4 * (lane_ + 32 * q));
        }
        __syncthreads();   // s_sub / rings are rewritten by the next block
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
