// ATRAC3plus front-end kernels (gfx950), SURVEY.md 8(f) row f4:
//   k_at3p_pqf   at3plus_pqf_do_analyse (atrac/atrac3plus_pqf/atrac3plus_pqf.c:62-147): 16-band polyphase analysis of a
//                2048-sample frame, one workgroup per (stream, frame, channel)
//   k_at3p_mdct  TAt3pMDCT::Do (atrac/at3p/at3p_mdct.cpp:52-96): 16 x windowed MDCT-256 with the sine / steep window
//                choice per subband, one workgroup per (stream, frame, channel)
// Both are free of recursion across frames: the filter needs the previous 368 input samples, the transform the previous
// frame's subband samples and window flags, so all frames of a call run side by side; the carried state is written by
// k_at3p_state after them. Float operations are the reference's, in its order, without contraction.
#pragma once
#include "at3_common.hpp"
#include "at3p_tables.hpp"

namespace at3p {

using at3::fft_lds;
using at3::fft_leaf_pos;

constexpr int kFrame = 2048;     // samples per channel and frame
constexpr int kOverlap = 368;    // PROTO_SZ - SUBBANDS_NUM

struct PqfParams {
    const Tables* T;
    const float* pcm;       // [S][F][2048][nch] interleaved (what TAt3PEnc::EncodeFrame receives, at3p.cpp:93-97)
    const float* hist;      // [S][nch][368]: the last 368 samples of the previous call
    float* bands;           // [S][F][nch][16][128]
    int32_t n_frames, nch;
};

__global__ __launch_bounds__(256) void k_at3p_pqf(PqfParams p)
{
    __shared__ __attribute__((aligned(16))) float s_x[kFrame + kOverlap];   // later: the frame's 16 x 128 output
    // 128 steps x 16 matrixed values, then - same 8 KB - the 128 eight-point transforms (18 KB per workgroup: eight per CU,
    // and the 4096 workgroups of the 64 x 32 batch are two whole rounds of the chip instead of 2.7)
    __shared__ __attribute__((aligned(16))) float s_yy_f[128 * 16];
    float (*const s_yy)[16] = reinterpret_cast<float (*)[16]>(s_yy_f);
    at3::cpx* const s_f = reinterpret_cast<at3::cpx*>(s_yy_f);
    __shared__ float s_cs[16];
    __shared__ __attribute__((aligned(8))) at3::cpx s_tw[8];

    const Tables* T = p.T;
    const int f = blockIdx.x, sc = blockIdx.y, nch = p.nch;
    const int s = sc / nch, ch = sc - s * nch, tid = threadIdx.x;
    const size_t item = ((size_t)s * p.n_frames + f) * nch + ch;

    // the thread's two prototype rows: yy[k] = y[a] + y[b] (matrixing, :76-79), k = tid & 15 for every value it produces
    const int k = tid & 15;
    const int ra = k + 8, rb = k < 8 ? 7 - k : 39 - k;
    float fa[12], fb[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        fa[j] = T->fir[ra * 12 + j];
        fb[j] = T->fir[rb * 12 + j];
    }
    // The frame's window of samples: every request first (ten per work-item, one source pointer each, clamped past the end), then
    // the stores. As a loop of load-store pairs the compiler kept it a loop: ten global round trips one after the other.
    {
        constexpr int kTot = kFrame + kOverlap, kIt = (kTot + 255) / 256;
        const float cs_v = T->sc32[tid & 15];
        const at3::cpx tw_v = T->tw8[tid & 7];
        float v[kIt];
#pragma unroll
        for (int i = 0; i < kIt; ++i) {
            const int j0 = tid + 256 * i, j = j0 < kTot ? j0 : kTot - 1;
            const int t = j - kOverlap;   // sample index inside the frame
            const float* src = (t >= 0) ? &p.pcm[(((size_t)s * p.n_frames + f) * kFrame + t) * nch + ch]
                             : (f > 0)  ? &p.pcm[(((size_t)s * p.n_frames + f - 1) * kFrame + kFrame + t) * nch + ch]
                                        : &p.hist[((size_t)s * nch + ch) * kOverlap + j];
            v[i] = *src;
        }
#pragma unroll
        for (int i = 0; i < kIt; ++i)
            if (tid + 256 * i < kTot) s_x[tid + 256 * i] = v[i];
        if (tid < 16) s_cs[tid] = cs_v;
        else if (tid < 24) s_tw[tid - 16] = tw_v;
    }
    __syncthreads();

    // vectoring (:62-70): float products, double running sums over the 12 taps, two rows per value
    for (int r = 0; r < 8; ++r) {
        const int step = (tid >> 4) + 16 * r;
        const float* x = s_x + 16 * step;
        double ya = 0, yb = 0;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            ya += (double)(fa[j] * x[j * 32 + ra]);
            yb += (double)(fb[j] * x[j * 32 + rb]);
        }
        s_yy[step][k] = (float)(ya + yb);
    }
    __syncthreads();

    // atde_do_dct4_16 (lib/mdct/mdct.cpp:73-80) = TMIDCT<32> (lib/mdct/mdct.h:117-180) of the 16 values: pre-rotation into
    // the leaf order of an 8-point FFT, 128 transforms side by side
    {
        at3::cpx v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = tid + 256 * r;
            const int step = q >> 3, pt = q & 7, n = 2 * pt;
            const float r0 = s_yy[step][n], i0 = s_yy[step][15 - n];
            const float c = s_cs[n], sn = s_cs[n + 1];
            v[r].r = (float)(-2.0 * (double)(i0 * sn + r0 * c));
            v[r].i = (float)(-2.0 * (double)(i0 * c - r0 * sn));
        }
        __syncthreads();   // every matrixed value has been read: the transforms move in
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = tid + 256 * r;
            s_f[8 * (q >> 3) + fft_leaf_pos<8>(q & 7)] = v[r];
        }
    }
    __syncthreads();
    fft_lds<8, false>(s_f, 8, 128, s_tw, tid, 256);
    // post-rotation; only Buf[8..23] is consumed: res[i] = -Buf[i + 8], subband sb = 15 - i gets res[i] (:86-88)
    float* s_out = s_x;
    for (int q = tid; q < 1024; q += 256) {
        const int step = q >> 3, pt = q & 7, n = 2 * pt;
        const at3::cpx v = s_f[q];
        const float c = s_cs[n], sn = s_cs[n + 1];
        const float r1 = v.r * c + v.i * sn;
        const float i1 = v.r * sn - v.i * c;
        s_out[(15 - n) * 128 + step] = -i1;   // i = n:      Buf[8 + n] = i1
        s_out[n * 128 + step] = -r1;          // i = 15 - n: Buf[23 - n] = r1
    }
    __syncthreads();
    for (int j = tid; j < kFrame; j += 256) p.bands[item * kFrame + j] = s_out[j];
}

struct MdctParams {
    const Tables* T;
    const float* bands;        // [S][F][nch][16][128]
    const uint16_t* flags;     // [S][F][nch] steep-window bits, or null (all sine)
    const float* hist;         // [S][nch][16][128]: the windowed first halves left by the previous call
    float* specs;              // [S][F][nch][2048]
    int32_t n_frames, nch;
    int32_t residual_scale;    // divide the subband samples by 32768 / 1.122018 first (at3p.cpp:143-147)
};

// first / second half of TAt3pMDCT::Do's work buffer for one subband (at3p_mdct.cpp:60-71, 83-95): v = the subband
// sample with index i in [0, 128)
__device__ __forceinline__ float at3p_first_half(float v, const float* w128, const float* w64, bool steep, int i)
{
    if (!steep) return w128[i] * v;
    return i < 32 ? 0.0f : i < 96 ? w64[i - 32] * v : (float)((double)v * 2.0);
}
__device__ __forceinline__ float at3p_second_half(float v, const float* w128, const float* w64, bool steep, int i)
{
    if (!steep) return w128[127 - i] * v;
    return i < 32 ? (float)((double)v * 2.0) : i < 96 ? w64[95 - i] * v : 0.0f;
}

__global__ __launch_bounds__(256) void k_at3p_mdct(MdctParams p)
{
    // the sixteen 256-sample work buffers; once the pre-rotation has read them the lower half holds the sixteen 64-point
    // transforms and the upper half collects the spectrum (18 KB per workgroup: eight per CU, two whole rounds for 64 x 32)
    __shared__ __attribute__((aligned(16))) float s_tmp[16][256];
    at3::cpx* const s_f = reinterpret_cast<at3::cpx*>(&s_tmp[0][0]);
    __shared__ float s_cs[128], s_w128[128], s_w64[64];
    __shared__ __attribute__((aligned(8))) at3::cpx s_tw[64];

    const Tables* T = p.T;
    const int f = blockIdx.x, sc = blockIdx.y, nch = p.nch;
    const int s = sc / nch, ch = sc - s * nch, tid = threadIdx.x;
    const size_t item = ((size_t)s * p.n_frames + f) * nch + ch;
    const unsigned cur_flags = p.flags ? p.flags[item] : 0u;
    const unsigned prev_flags = (p.flags && f > 0) ? p.flags[item - nch] : 0u;

    if (tid < 128) {
        s_cs[tid] = T->sc256[tid];
        s_w128[tid] = T->sine128[tid];
    } else if (tid < 192) {
        s_w64[tid - 128] = T->sine64[tid - 128];
    } else {
        s_tw[tid - 192] = T->tw64[tid - 192];
    }
    // the sixteen subband samples of the work-item: all requested before the tables' rendezvous (one source pointer each), windowed
    // behind it - as a loop of load, window, store the compiler kept it a loop of sixteen global round trips
    float raw[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int j = tid + 256 * i, b = j >> 8, o = j & 255;
        const float* src = (o >= 128) ? p.bands + item * kFrame + 128 * b + (o - 128)
                         : (f > 0)    ? p.bands + (item - nch) * kFrame + 128 * b + o
                                      : p.hist + (((size_t)s * nch + ch) * 16 + b) * 128 + o;
        raw[i] = *src;
    }
    auto scaled = [&](float v) {
        if (p.residual_scale) v = (float)((double)v / (32768.0 / 1.122018));
        return v;
    };
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int j = tid + 256 * i, b = j >> 8, o = j & 255;
        float v;
        if (o >= 128) v = at3p_second_half(scaled(raw[i]), s_w128, s_w64, (cur_flags >> b) & 1, o - 128);
        else if (f > 0) v = at3p_first_half(scaled(raw[i]), s_w128, s_w64, (prev_flags >> b) & 1, o);
        else v = raw[i];
        s_tmp[b][o] = v;
    }
    __syncthreads();
    // TMDCT<256>::operator() (lib/mdct/mdct.h:51-104): pre-rotation into the 64-point FFT's leaf order, 16 transforms
    {
        at3::cpx v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = tid + 256 * r;
            const int b = q >> 6, pt = q & 63, n = 2 * pt;
            const float* in = s_tmp[b];
            float r0, i0;
            if (n < 64) {
                r0 = in[191 - n] + in[192 + n];
                i0 = in[64 + n] - in[63 - n];
            } else {
                r0 = in[191 - n] - in[n - 64];
                i0 = in[64 + n] + in[319 - n];
            }
            const float c = s_cs[n], sn = s_cs[n + 1];
            v[r].r = r0 * c + i0 * sn;
            v[r].i = i0 * c - r0 * sn;
        }
        __syncthreads();   // every work buffer has been read: the transforms move into the lower half
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = tid + 256 * r;
            s_f[64 * (q >> 6) + fft_leaf_pos<64>(q & 63)] = v[r];
        }
    }
    __syncthreads();
    fft_lds<64, false>(s_f, 64, 16, s_tw, tid, 256);
    float* s_out = &s_tmp[8][0];
    for (int q = tid; q < 1024; q += 256) {
        const int b = q >> 6, pt = q & 63, n = 2 * pt;
        const at3::cpx v = s_f[q];
        const float c = s_cs[n], sn = s_cs[n + 1];
        const float o1 = -v.r * c - v.i * sn;
        const float o2 = -v.r * sn + v.i * c;
        float* dst = s_out + 128 * b;
        if (b & 1) {   // SwapArray for odd subbands (at3p_mdct.cpp:77-79)
            dst[127 - n] = o1;
            dst[n] = o2;
        } else {
            dst[n] = o1;
            dst[127 - n] = o2;
        }
    }
    __syncthreads();
    for (int j = tid; j < kFrame; j += 256) p.specs[item * kFrame + j] = s_out[j];
}

// carried state after a call: the last 368 input samples (PQF) / the windowed first halves of the last frame (MDCT)
__global__ void k_at3p_pqf_state(const float* pcm, float* hist, int n_frames, int nch, int n_streams)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_streams * nch * kOverlap) return;
    const int j = i % kOverlap, sc = i / kOverlap, ch = sc % nch, s = sc / nch;
    hist[i] = pcm[(((size_t)s * n_frames + n_frames - 1) * kFrame + (kFrame - kOverlap) + j) * nch + ch];
}
__global__ void k_at3p_mdct_state(MdctParams p, float* hist, int n_streams)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_streams * p.nch * 2048) return;
    const int o = i & 127, b = (i >> 7) & 15, sc = i >> 11, ch = sc % p.nch, s = sc / p.nch;
    const size_t item = ((size_t)s * p.n_frames + p.n_frames - 1) * p.nch + ch;
    const unsigned flags = p.flags ? p.flags[item] : 0u;
    const float* src = p.bands + item * kFrame + 128 * b;
    float v = src[o];
    if (p.residual_scale) v = (float)((double)v / (32768.0 / 1.122018));
    hist[i] = at3p_first_half(v, p.T->sine128, p.T->sine64, (flags >> b) & 1, o);
}

}  // namespace at3p
