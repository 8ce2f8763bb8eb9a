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
// Work decomposition: one 256-thread workgroup owns one stream and a run of consecutive frames, both
// channels. Per block it stages the interleaved PCM tile (+138-sample FIR reach) in LDS with coalesced
// float2 loads, runs the two QMF stages out of LDS (taps in LDS), and - for frames - modulates, windows
// and transforms the four subbands of both channels as eight concurrent 128-point FFTs (32 lanes each).
// The previous block's windowed overlap stays in LDS between frames, so each PCM sample is read from
// HBM once per workgroup run and spectra are written once: algorithmic traffic is 16 KiB per frame.
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
};

// LDS carve (floats). Region A is the QMF working set; the MDCT staging buffer aliases it.
constexpr int kPcmLen = 1168;   // t = -138 .. 1023 -> idx t + 138 (1162 used)
constexpr int kS1Len = 560;     // m = -46 .. 511  -> idx m + 46  (558 used)
constexpr int kRegionA = 2 * kPcmLen + 4 * kS1Len;  // 4576 floats >= 8*512
static_assert(kRegionA >= 8 * 512, "MDCT staging must fit in the QMF region");

__device__ __forceinline__ void load_pcm_tile(const FrontParams& p, int s, int b, float* s_pcm, int tid)
{
    const float2* pcm2 = reinterpret_cast<const float2*>(p.pcm) + (size_t)s * p.n_blocks * 1024;
    const float2* hist2 = reinterpret_cast<const float2*>(p.hist) + (size_t)s * kHist;
    for (int k = tid; k < 1162; k += 256) {
        const int g = b * 1024 + k - 138;
        const float2 v = (g >= 0) ? pcm2[g] : hist2[kHist + g];
        s_pcm[k] = v.x * 0.25f;            // data / 4.0 (exact)
        s_pcm[kPcmLen + k] = v.y * 0.25f;
    }
}

// Both QMF stages for one block of both channels. In: s_pcm. Out: s_sub[2][4][256].
__device__ __forceinline__ void qmf_block(const float* s_qw, const float* s_pcm, float* s_lo, float* s_hi,
                                          float* s_sub, int tid)
{
    // stage 1 (Qmf1): 558 (lo,hi) pairs per channel, m = -46..511
    for (int idx = tid; idx < 2 * 558; idx += 256) {
        const int ch = idx / 558, mm = idx - ch * 558;
        const float* x = s_pcm + ch * kPcmLen + 2 * mm;
        float lo = 0.0f, hi = 0.0f;
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            lo += s_qw[2 * i] * x[47 - 2 * i];
            hi += s_qw[2 * i + 1] * x[46 - 2 * i];
        }
        s_lo[ch * kS1Len + mm] = lo + hi;   // lower
        s_hi[ch * kS1Len + mm] = lo - hi;   // upper
    }
    __syncthreads();
    // stage 2: Qmf2 on the lower half -> bands 0 (lower), 1 (upper); Qmf3 on the upper half -> bands 3, 2
    for (int idx = tid; idx < 2 * 2 * 256; idx += 256) {
        const int ch = idx >> 9, which = (idx >> 8) & 1, j = idx & 255;
        const float* x = (which ? s_hi : s_lo) + ch * kS1Len + 2 * j;
        float lo = 0.0f, hi = 0.0f;
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            lo += s_qw[2 * i] * x[47 - 2 * i];
            hi += s_qw[2 * i + 1] * x[46 - 2 * i];
        }
        float* out = s_sub + ch * 1024;
        if (which == 0) {
            out[0 * 256 + j] = lo + hi;
            out[1 * 256 + j] = lo - hi;
        } else {
            out[3 * 256 + j] = lo + hi;
            out[2 * 256 + j] = lo - hi;
        }
    }
    __syncthreads();
}

// Subband analysis only (feeds the gain-control kernels): raw L/R subbands of blocks -2 .. n_blocks-1.
__global__ __launch_bounds__(256) void k_qmf_sub(FrontParams p, const Tables* T)
{
    __shared__ float s_a[kRegionA];
    __shared__ float s_sub[2 * 4 * 256];
    __shared__ float s_qw[48];
    const int tid = threadIdx.x;
    const int nb2 = p.n_blocks + 2;
    const int s = blockIdx.x / nb2;
    const int b = (int)(blockIdx.x % nb2) - 2;
    float* s_pcm = s_a;
    float* s_lo = s_a + 2 * kPcmLen;
    float* s_hi = s_lo + 2 * kS1Len;
    if (tid < 48) s_qw[tid] = T->qmf_win[tid];
    load_pcm_tile(p, s, b, s_pcm, tid);
    __syncthreads();
    qmf_block(s_qw, s_pcm, s_lo, s_hi, s_sub, tid);
    const size_t sublen = (size_t)nb2 * 256;
    for (int idx = tid; idx < 2048; idx += 256) {
        const int cb = idx >> 8, j = idx & 255;  // cb = ch*4 + band
        p.sub[((size_t)s * 8 + cb) * sublen + (size_t)(b + 2) * 256 + j] = s_sub[idx];
    }
}

template <bool GAIN>
__global__ __launch_bounds__(256) void k_qmf_mdct(FrontParams p, const Tables* T)
{
    __shared__ float s_a[kRegionA];          // QMF working set | MDCT input / output staging (aliased)
    __shared__ float s_sub[2 * 4 * 256];     // current block's subbands [ch][band][256]
    __shared__ float s_prevw[2 * 4 * 256];   // overlap half carried to the next frame (windowed, modulated)
    __shared__ cpx s_fft[8 * 128];
    __shared__ float s_div[GAIN ? 8 * 256 : 1];
    __shared__ float s_qw[48];
    __shared__ float s_win[256];
    __shared__ float s_cs[256];
    __shared__ cpx s_tw[128];
    __shared__ Curve s_curve[8];
    __shared__ float s_nextscale[8];         // NextOverlapScale of the block just processed
    __shared__ float s_sum[8][5];

    const int tid = threadIdx.x;
    const int nchunks = (p.n_blocks - p.f0 + p.frames_per_wg - 1) / p.frames_per_wg;
    const int s = blockIdx.x / nchunks;
    const int chunk = blockIdx.x % nchunks;
    const int fa = p.f0 + chunk * p.frames_per_wg;
    int fb = fa + p.frames_per_wg;
    if (fb > p.n_blocks) fb = p.n_blocks;
    const int n_out = p.n_blocks - p.f0;

    float* s_pcm = s_a;
    float* s_lo = s_a + 2 * kPcmLen;
    float* s_hi = s_lo + 2 * kS1Len;
    float* s_tmp = s_a;  // [8][512]

    if (tid < 48) s_qw[tid] = T->qmf_win[tid];
    s_win[tid] = T->enc_win[tid];
    s_cs[tid] = T->mdct_sincos[tid];
    if (tid < 128) s_tw[tid] = T->tw128[tid];
    if (tid < 8) s_nextscale[tid] = 1.0f;

    const int c = tid >> 5;      // (channel, band) combo owning this thread in the MDCT phase
    const int lane = tid & 31;

    // block b carries frame f = b + 1; the block before the first frame only primes the overlap.
    for (int b = fa - 2; b <= fb - 2; ++b) {
        const int f = b + 1;
        const bool is_frame = (f >= fa);
        __syncthreads();  // previous iteration finished reading s_tmp / tables are loaded
        load_pcm_tile(p, s, b, s_pcm, tid);
        if (GAIN && tid < 8) {
            Curve cv;
            if (f < 0) cv = p.state[(size_t)s * 8 + tid].prev_curve;
            else cv = p.curves[((size_t)s * p.n_blocks + f) * 8 + tid];
            s_curve[tid] = cv;
        }
        __syncthreads();
        qmf_block(s_qw, s_pcm, s_lo, s_hi, s_sub, tid);
        if (p.js) {  // M/S matrixing in the subband domain
            for (int idx = tid; idx < 1024; idx += 256) {
                const float l = s_sub[idx], r = s_sub[1024 + idx];
                s_sub[idx] = (l + r) * 0.5f;         // (l + r) / 2.0, exact halving
                s_sub[1024 + idx] = (l - r) * 0.5f;
            }
            __syncthreads();
        }

        float prev_scale = 1.0f;  // NextOverlapScale of the previous block == PrevOverlapGainScale
        bool has_curve = false;
        if (GAIN) {
            prev_scale = s_nextscale[c];
            has_curve = s_curve[c].n > 0;
            if (has_curve) {
                for (int i = lane; i < 256; i += 32) s_div[c * 256 + i] = curve_divisor(T, s_curve[c], i);
            }
            __syncthreads();
            // CalcGainEnergyScale: five strictly sequential 256-term sums, one lane each.
            const bool need = has_curve || prev_scale != 1.0f;
            if (need && lane < 5) {
                const float* x = s_sub + c * 256;
                const float* pw = s_prevw + c * 256;
                const float* dv = s_div + c * 256;
                float acc = 0.0f;
                if (lane == 0) {
                    for (int i = 0; i < 256; ++i) acc += pw[i] * pw[i];
                } else {
                    const bool modulated = (lane == 2 || lane == 4);
                    const bool next = (lane >= 3);
                    for (int i = 0; i < 256; ++i) {
                        float v = x[i];
                        if (modulated && has_curve) v = v / dv[i];
                        const float w = next ? s_win[i] : s_win[255 - i];
                        const float vw = v * w;
                        acc += vw * vw;
                    }
                }
                s_sum[c][lane] = acc;
            }
            __syncthreads();
            if (lane == 0) {
                float frame_scale = 1.0f, next_scale = 1.0f;
                if (need) {
                    float ps = prev_scale;
                    if (!isfinite(ps) || ps <= 0.0f) ps = 1.0f;
                    const float prevDiv = has_curve ? T->gain_level[s_curve[c].level[0]] : 1.0f;
                    const float prevStored = s_sum[c][0];
                    const float prevOrig = prevStored * ps;
                    const float prevMod = prevStored / (prevDiv * prevDiv);
                    frame_scale = safe_energy_scale(prevOrig + s_sum[c][1], prevMod + s_sum[c][2]);
                    next_scale = safe_energy_scale(s_sum[c][3], s_sum[c][4]);
                }
                s_nextscale[c] = next_scale;
                if (is_frame) p.ges[((size_t)s * p.n_blocks + f) * 8 + c] = frame_scale;
            }
        }

        // Modulate + window (atrac3denc.cpp:39-49). s_tmp aliases the QMF region: all QMF reads are done.
        {
            float* tmp = s_tmp + c * 512;
            float* pw = s_prevw + c * 256;
            const float* x = s_sub + c * 256;
            const float scale = (GAIN && has_curve) ? T->gain_level[s_curve[c].level[0]] : 1.0f;
            for (int i = lane; i < 256; i += 32) {
                float ov = pw[i];
                float v = x[i];
                if (GAIN && has_curve) {
                    ov = ov / scale;
                    v = v / s_div[c * 256 + i];
                }
                tmp[i] = ov;
                pw[i] = s_win[i] * v;
                tmp[256 + i] = s_win[255 - i] * v;
            }
        }
        __syncthreads();
        if (!is_frame) continue;

        // MDCT-512 = fold + pre-rotation -> 128-pt FFT -> post-rotation (mdct.h:51-104)
        {
            const float* in = s_tmp + c * 512;
            for (int n2 = lane; n2 < 128; n2 += 32) {
                const int n = 2 * n2;
                float r0, i0;
                if (n < 128) {
                    r0 = in[383 - n] + in[384 + n];
                    i0 = in[128 + n] - in[127 - n];
                } else {
                    r0 = in[383 - n] - in[n - 128];
                    i0 = in[128 + n] + in[639 - n];
                }
                const float cc = s_cs[n], ss = s_cs[n + 1];
                cpx v;
                v.r = r0 * cc + i0 * ss;
                v.i = i0 * cc - r0 * ss;
                s_fft[c * 128 + fft_leaf_pos<128>(n2)] = v;
            }
        }
        __syncthreads();
        fft_lds<128, false>(s_fft, 128, 8, s_tw, tid, 256);
        {
            float* out = s_tmp + c * 512;  // reuse as output staging [256]
            const bool odd = (c & 1);
            for (int n2 = lane; n2 < 128; n2 += 32) {
                const int n = 2 * n2;
                const float r0 = s_fft[c * 128 + n2].r, i0 = s_fft[c * 128 + n2].i;
                const float cc = s_cs[n], ss = s_cs[n + 1];
                const float a = -r0 * cc - i0 * ss;
                const float bq = -r0 * ss + i0 * cc;
                out[odd ? 255 - n : n] = a;           // odd bands are stored reversed (atrac3denc.cpp:53-55)
                out[odd ? n : 255 - n] = bq;
            }
        }
        __syncthreads();
        {
            float* dst = p.specs + ((size_t)s * n_out + (f - p.f0)) * 2048;
            for (int idx = tid; idx < 2048; idx += 256) dst[idx] = s_tmp[(idx >> 8) * 512 + (idx & 255)];
        }
    }
}

// Batched TAtrac3MDCT::Mdct on caller-provided band buffers (atrac3denc.h:80-86): one workgroup per item.
struct MdctItemsParams {
    float* bands;          // [n][4][512] in/out
    float* specs;          // [n][1024] out
    const int32_t* n_points;  // [n][4] or null
    const int32_t* level;     // [n][4][8]
    const int32_t* loc;       // [n][4][8]
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
    const float scale = has_curve ? T->gain_level[s_curve[c].level[0]] : 1.0f;
    float* tmp = s_tmp + c * 512;
    for (int i = lane; i < 256; i += 32) {
        float ov = band[i];
        float v = band[256 + i];
        if (has_curve) {
            ov = ov / scale;
            const float d = curve_divisor(T, s_curve[c], i);
            v = v / d;
            band[256 + i] = v;
        }
        tmp[i] = ov;
        band[i] = T->enc_win[i] * v;
        tmp[256 + i] = T->enc_win[255 - i] * v;
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

}  // namespace at3
