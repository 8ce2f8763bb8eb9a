// ATRAC3plus frame writer without tonal block (gfx950), SURVEY.md 8(f) row f4, second half:
//   k_at3p_write  TScaler<NAt3p::TScaleTable>::ScaleFrame (atrac/atrac_scale.cpp:141-191) and
//                 TAt3PBitStream::WriteFrame(channels, nullptr, sces) (atrac/at3p/at3p_bitstream.cpp:99-470, 630-726)
//                 for one (stream, frame) per workgroup, both channels.
// The reference gives every quant unit a fixed word length (TConfigure's table), so nothing but the NUMBER of quant units
// is searched: it starts at 32 and drops to 28, 27, ... while the frame is over 16381 bits (CheckFrameDone under the
// repeat protocol of lib/bs_encode/encode.cpp:100-130). A unit's mantissas and its cheapest code table do not depend on
// that number, so they are formed once; the frame's bit count is monotone in the number of units, so the first count
// that fits is found from one prefix sum. What does not depend on the spectrum at all - the three leading bits,
// TConfigure and TWordLenEncoder for each (channels, units) - is formed on the host (WriteTables::head).
#pragma once
#include "at3_common.hpp"

namespace at3p {

#include "at3p_vlc.inc"

constexpr int kFrameBytes = 2048;                  // TAt3PBitStream(out, 2048) (at3p.cpp:40)
constexpr int kSizeBits = kFrameBytes * 8 - 3;     // FrameSzToAllocBits (at3p_bitstream.cpp:481)

struct WriteHead {
    uint32_t words[8];     // the frame's first bits, most significant bit first
    uint16_t nbits;        // how many of them
    uint16_t fixed_bits;   // bits of everything but the spectra and the tonal part, without the three leading bits
};

struct WriteTables {
    uint16_t vlc[AT3P_VLC_TOTAL];   // code | length << 12 per symbol, table t at off[t]
    uint16_t off[114];
    uint8_t tab[112][2];            // group_size | num_coeffs << 4, bits | is_signed << 4
    float inv_mant[8];              // 1.0f / atrac3p_mant_tab[wl] (TUnit::TUnit)
    float scale[64];                // NAt3p::TScaleTable::ScaleTable
    uint8_t qu_to_sb[32];
    uint8_t sb_powgrps[16];
    WriteHead head[2][33];          // [channels - 1][quant units]
    // for the tables the writer prices (word length - 1 + 7 i, i < 8: tables 0..55): per table offset | group_size << 16 |
    // num_coeffs << 20 | bits << 24 | is_signed << 28, and the code lengths alone, one nibble per symbol - the part of
    // the table block k_at3p_write keeps in LDS
    uint32_t info56[56];
    uint32_t len4[(7812 + 7) / 8];
};
constexpr int kLenEntries = 7812;   // symbols of tables 0..55

// TConfigure::Encode's allocTable (at3p_bitstream.cpp:107-112): 7 x 17, 6 x 9, 5, 5, 4, 3, 2, 1
__host__ __device__ inline int at3p_wordlen(int qu) { return qu < 17 ? 7 : qu < 26 ? 6 : qu < 28 ? 5 : 32 - qu; }
// NAt3p::TScaleTable::BlockSizeTab (at3p_tables.h:62-68)
__host__ __device__ inline int at3p_qu_start(int qu)
{
    return qu < 8 ? 16 * qu : qu < 16 ? 128 + 32 * (qu - 8) : qu < 22 ? 384 + 64 * (qu - 16) : 768 + 128 * (qu - 22);
}
__host__ __device__ inline int at3p_qu_of_line(int line)
{
    return line < 128 ? line >> 4 : line < 384 ? 8 + ((line - 128) >> 5) : line < 768 ? 16 + ((line - 384) >> 6) : 22 + ((line - 768) >> 7);
}

// ---- host: the tables and the spectrum-independent bits -------------------------------------------------------------
inline void head_put(WriteHead& h, uint32_t val, int n)
{
    for (int k = n - 1; k >= 0; --k) {
        if ((val >> k) & 1u) h.words[h.nbits >> 5] |= 0x80000000u >> (h.nbits & 31);
        h.nbits++;
    }
}

inline void build_write_tables(WriteTables* w)
{
    memset(w, 0, sizeof(*w));
    memcpy(w->vlc, AT3P_VLC, sizeof(AT3P_VLC));
    memcpy(w->off, AT3P_VLC_OFF, sizeof(AT3P_VLC_OFF));
    memcpy(w->tab, AT3P_SPEC_TAB, sizeof(AT3P_SPEC_TAB));
    memcpy(w->inv_mant, AT3P_INV_MANT, sizeof(AT3P_INV_MANT));
    memcpy(w->scale, AT3P_SCALE, sizeof(AT3P_SCALE));
    memcpy(w->qu_to_sb, AT3P_QU_TO_SB, sizeof(AT3P_QU_TO_SB));
    memcpy(w->sb_powgrps, AT3P_SB_POWGRPS, sizeof(AT3P_SB_POWGRPS));
    for (int t = 0; t < 56; ++t)
        w->info56[t] = (uint32_t)AT3P_VLC_OFF[t] | ((uint32_t)(AT3P_SPEC_TAB[t][0] & 15) << 16) | ((uint32_t)(AT3P_SPEC_TAB[t][0] >> 4) << 20) |
                       ((uint32_t)(AT3P_SPEC_TAB[t][1] & 15) << 24) | ((uint32_t)(AT3P_SPEC_TAB[t][1] >> 4) << 28);
    for (int e = 0; e < AT3P_VLC_OFF[56]; ++e) w->len4[e >> 3] |= (uint32_t)(AT3P_VLC[e] >> 12) << (4 * (e & 7));
    for (int nch = 1; nch <= 2; ++nch) {
        for (int N = 1; N <= 32; ++N) {
            WriteHead& h = w->head[nch - 1][N];
            head_put(h, 0, 1);                     // WriteFrame (:700-706)
            head_put(h, (uint32_t)nch - 1, 2);
            head_put(h, (uint32_t)N - 1, 5);       // TConfigure (:129-130)
            head_put(h, 0, 1);
            // TWordLenEncoder (:170-252): channel 0 as deltas to the previous unit, channel 1 as deltas to channel 0 (all zero:
            // both channels get the same word lengths); the code table is picked by the largest delta (FindBestWlDeltaEncode)
            int8_t d0[32], dx[32];
            int max0 = 0;
            d0[0] = (int8_t)at3p_wordlen(0);
            dx[0] = 0;
            for (int i = 1; i < N; ++i) {
                const int d = at3p_wordlen(i) - at3p_wordlen(i - 1);
                max0 |= d < 0 ? -d : d;
                d0[i] = (int8_t)(d & 7);
                dx[i] = 0;
            }
            auto best = [&](const int8_t* delta, int maxDelta) {
                const int t0 = maxDelta >= 3 ? 2 : maxDelta == 2 ? 1 : 0, t1 = maxDelta >= 3 ? 3 : t0;
                int bestIdx = 0;
                long consumed = -1;
                for (int i = t0; i <= t1; ++i) {
                    long t = 0;
                    for (int j = 1; j < N; ++j) t += AT3P_WL_VLC[i][delta[j]] >> 12;
                    if (consumed < 0 || t < consumed) {
                        consumed = t;
                        bestIdx = i;
                    }
                }
                return bestIdx;
            };
            {
                const int idx = best(d0, max0);
                head_put(h, 3, 2);
                head_put(h, 0, 2);
                head_put(h, 0, 2);
                head_put(h, (uint32_t)idx, 2);
                head_put(h, (uint32_t)d0[0], 3);
                for (int i = 1; i < N; ++i) head_put(h, AT3P_WL_VLC[idx][d0[i]] & 0xfffu, AT3P_WL_VLC[idx][d0[i]] >> 12);
            }
            if (nch == 2) {
                const int idx = best(dx, 0);
                head_put(h, 1, 2);
                head_put(h, 0, 2);
                head_put(h, (uint32_t)idx, 2);
                for (int i = 0; i < N; ++i) head_put(h, AT3P_WL_VLC[idx][dx[i]] & 0xfffu, AT3P_WL_VLC[idx][dx[i]] >> 12);
            }
            // + TSfIdxEncoder (:254-276), EncodeCodeTab (:278-306), the power groups (:446-453)
            const int pw = 4 * AT3P_SB_POWGRPS[AT3P_QU_TO_SB[N - 1]];
            h.fixed_bits = (uint16_t)((h.nbits - 3) + nch * (2 + 6 * N) + 1 + nch * (4 + 3 * N) + nch * pw);
        }
    }
}

// ---- device ----------------------------------------------------------------------------------------------------------
struct WriteParams {
    const WriteTables* W;
    const float* specs;        // [items][nch][2048]
    const uint16_t* flags;     // [items][nch] steep-window bits per subband, or nullptr (all sine)
    uint8_t* out;              // [items][2048]
    int32_t nch, n_items;
};

__device__ __forceinline__ void frame_put(uint32_t* words, int pos, uint32_t val, int n)   // n in 1..23
{
    if (pos + n > kFrameBytes * 8) return;
    val &= (1u << n) - 1u;
    const int w = pos >> 5, off = pos & 31, room = 32 - off;
    if (n <= room) {
        atomicOr(&words[w], val << (room - n));
    } else {
        atomicOr(&words[w], val >> (n - room));
        atomicOr(&words[w + 1], val << (32 - (n - room)));
    }
}

// a lane's run of codes strung together in registers, handed to the shared frame one 32-bit word at a time
struct BitRun {
    uint32_t* words;
    uint64_t acc;
    int cur, fill;
    __device__ __forceinline__ void start(uint32_t* w, int pos) { words = w; acc = 0; cur = pos >> 5; fill = pos & 31; }
    __device__ __forceinline__ void add(uint32_t v, int n)   // n <= 12
    {
        if (n == 0) return;
        acc |= (uint64_t)(v & ((1u << n) - 1u)) << (64 - fill - n);
        fill += n;
        if (fill >= 32) {
            if (cur < kFrameBytes / 4) atomicOr(&words[cur], (uint32_t)(acc >> 32));
            acc <<= 32;
            fill -= 32;
            ++cur;
        }
    }
    __device__ __forceinline__ void finish()
    {
        if (fill > 0 && cur < kFrameBytes / 4) atomicOr(&words[cur], (uint32_t)(acc >> 32));
    }
};

// The symbols a chunk's 16 mantissas form under a packing (NC coefficients of `bits` bits per symbol, signed or as
// magnitudes with separate sign bits): statically indexed, so everything stays in registers. The candidate tables of a
// word length share one to three packings, so the symbols are formed once per packing and only looked up per table.
template <int NC>
__device__ __forceinline__ int pack_symbols(const int (&q)[16], int bits, int is_signed, uint32_t (&vals)[16])
{
    int n_signs = 0;
    const int mask = (1 << bits) - 1;
#pragma unroll
    for (int sidx = 0; sidx < 16 / NC; ++sidx) {
        uint32_t val = 0;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            int t = q[sidx * NC + i];
            if (!is_signed && t != 0) {
                ++n_signs;
                t = t < 0 ? -t : t;
            } else {
                t &= mask;
            }
            val |= (uint32_t)t << (bits * i);
        }
        vals[sidx] = val & 0xffu;
    }
    return n_signs;
}

// The same with each symbol's sign bits (0 = positive, 1 = negative, in coefficient order) as bits | count << 4: what
// EncodeQuSpectra appends after the symbol's code (at3p_bitstream.cpp:360-368).
template <int NC>
__device__ __forceinline__ void pack_symbols_signs(const int (&q)[16], int bits, int is_signed, uint32_t (&vals)[16], uint32_t (&sgn)[16])
{
    const int mask = (1 << bits) - 1;
#pragma unroll
    for (int sidx = 0; sidx < 16 / NC; ++sidx) {
        uint32_t val = 0, sb = 0, ns = 0;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            int t = q[sidx * NC + i];
            if (!is_signed && t != 0) {
                sb = (sb << 1) | (t < 0 ? 1u : 0u);
                ++ns;
                t = t < 0 ? -t : t;
            } else {
                t &= mask;
            }
            val |= (uint32_t)t << (bits * i);
        }
        vals[sidx] = val & 0xffu;
        sgn[sidx] = sb | (ns << 4);
    }
}

// EncodeQuSpectra (at3p_bitstream.cpp:310-373) for the chosen table: all code words are requested first (one memory latency for the chunk), then
// strung together with the group flags and the sign bits.
template <int NC>
__device__ __forceinline__ void emit_chunk(const WriteTables* W, const int (&q)[16], uint32_t d, BitRun* run)
{
    const int off = (int)(d & 0xffffu), group_size = (int)((d >> 16) & 15u);
    const int cbits = (int)((d >> 24) & 15u), is_signed = (int)(d >> 28);
    uint32_t vals[16], sgn[16], code[16];
    pack_symbols_signs<NC>(q, cbits, is_signed, vals, sgn);
#pragma unroll
    for (int sidx = 0; sidx < 16 / NC; ++sidx) code[sidx] = W->vlc[off + (int)vals[sidx]];
#pragma unroll
    for (int sidx = 0; sidx < 16 / NC; ++sidx) {
        if (group_size != 1 && (sidx % group_size) == 0) run->add(1u, 1);   // group_size is 1, 2 or 4
        run->add(code[sidx] & 0xfffu, (int)(code[sidx] >> 12));
        run->add(sgn[sidx] & 15u, (int)(sgn[sidx] >> 4));
    }
}

__global__ __launch_bounds__(256) void k_at3p_write(WriteParams p)
{
    __shared__ uint32_t s_out[kFrameBytes / 4];
    __shared__ uint16_t s_cbits[256][8];     // bits of each 16-line chunk under each of the eight candidate tables
    __shared__ uint32_t s_qbits[2][32][8];   // the same per quant unit
    __shared__ uint32_t s_max[2][32];
    __shared__ float s_scale[64];
    __shared__ uint8_t s_sfi[2][32], s_tab[2][32];
    __shared__ uint32_t s_best[2][32];
    __shared__ uint32_t s_wsum[4];
    __shared__ int s_n;
    __shared__ uint16_t s_cb[256];           // the chosen table's bits per chunk, in stream order (channel, chunk)
    __shared__ uint32_t s_off[256];          // and their exclusive prefix sums per channel
    __shared__ uint32_t s_info[56];
    __shared__ uint32_t s_len4[(kLenEntries + 7) / 8];

    const WriteTables* W = p.W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nch = p.nch;
    const size_t item = blockIdx.x;
    // The thread's chunk (lines 16 c .. 16 c + 15 of channel ch). Threads are dealt by WORD LENGTH, both channels of a
    // class side by side (7: 2 x 28 chunks, 6: 2 x 52, 5: 2 x 16, 4..1: 2 x 8 each): a wavefront then walks one or two
    // sets of code tables instead of up to six.
    int ch, c;
    if (tid < 56) { ch = tid / 28; c = tid % 28; }
    else if (tid < 160) { ch = (tid - 56) / 52; c = 28 + (tid - 56) % 52; }
    else if (tid < 192) { ch = (tid - 160) >> 4; c = 80 + ((tid - 160) & 15); }
    else { ch = ((tid - 192) >> 3) & 1; c = 96 + 8 * ((tid - 192) >> 4) + ((tid - 192) & 7); }
    const bool active = ch < nch;
    const int qu = at3p_qu_of_line(16 * c);
    const int wl = at3p_wordlen(qu);

    s_out[tid] = 0u;
    s_out[256 + tid] = 0u;
    (&s_qbits[0][0][0])[tid] = 0u;
    (&s_qbits[0][0][0])[256 + tid] = 0u;
    // Every request of the prologue first - the chunk's sixteen lines, the thread's table words at clamped indices, the word
    // length's multiplier for the quantiser further down - and the stores behind them: as load-store pairs under thread conditions
    // and a copy loop they were five global round trips one after the other.
    constexpr int kLenWords = (kLenEntries + 7) / 8, kLenIt = (kLenWords + 255) / 256;
    float x[16];
    const float mul_wl = W->inv_mant[wl];
    {
        const float4* src = reinterpret_cast<const float4*>(p.specs + (item * nch + (active ? ch : 0)) * 2048 + 16 * c);
        float4 xv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) xv[k] = src[k];
        const float scale_v = W->scale[tid & 63];
        const uint32_t info_v = W->info56[(tid >= 64 && tid < 120) ? tid - 64 : 0];
        uint32_t len_v[kLenIt];
#pragma unroll
        for (int i = 0; i < kLenIt; ++i) len_v[i] = W->len4[tid + 256 * i < kLenWords ? tid + 256 * i : kLenWords - 1];
        if (tid < 64) {
            (&s_max[0][0])[tid] = 0u;
            s_scale[tid] = scale_v;
        }
        if (tid >= 64 && tid < 120) s_info[tid - 64] = info_v;
#pragma unroll
        for (int i = 0; i < kLenIt; ++i)
            if (tid + 256 * i < kLenWords) s_len4[tid + 256 * i] = len_v[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 v = active ? xv[k] : float4{0.0f, 0.0f, 0.0f, 0.0f};
            x[4 * k] = v.x; x[4 * k + 1] = v.y; x[4 * k + 2] = v.z; x[4 * k + 3] = v.w;
        }
    }
    __syncthreads();
    // ---- TScaler::Scale: the unit's largest magnitude (order-free), its scale factor, the scaled values ----
    {
        float m = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) m = fmaxf(m, fabsf(x[k]));
        if (active) atomicMax(&s_max[ch][qu], __float_as_uint(m));
    }
    __syncthreads();
    if (tid < 64) {
        float maxAbs = __uint_as_float((&s_max[0][0])[tid]);
        if (maxAbs > 1.0f) maxAbs = 1.0f;
        int lo = 0, hi = 63;   // map::lower_bound: the first entry >= maxAbs
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_scale[mid] < maxAbs) lo = mid + 1;
            else hi = mid;
        }
        (&s_sfi[0][0])[tid] = (uint8_t)lo;
    }
    __syncthreads();
    // ---- QuantMantisas without the energy pass (atrac_scale.cpp:46-54) at the unit's fixed word length ----
    int qv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) qv[k] = 0;
    if (active) {
        const float sf = s_scale[s_sfi[ch][qu]];
        const float mul = mul_wl;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float v = x[k] / sf;
            if (fabsf(v) >= 1.0f) v = (v > 0) ? 0.99999f : -0.99999f;
            qv[k] = __float2int_rn(v * mul);
        }
    }
    __syncthreads();
    // ---- TUnit::GetOrCompute (:387-417): the unit's bits under each of its eight code tables ----
    if (active) {
        uint32_t vals[16], prev_packing = 0xffffffffu;
        int n_signs = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) vals[k] = 0u;
        for (int i = 0; i < 8; ++i) {
            const uint32_t d = s_info[wl - 1 + 7 * i];
            const int off = (int)(d & 0xffffu), group_size = (int)((d >> 16) & 15u), num_coeffs = (int)((d >> 20) & 15u);
            if ((d >> 20) != prev_packing) {   // num_coeffs, bits, is_signed
                prev_packing = d >> 20;
                const int cbits = (int)((d >> 24) & 15u), is_signed = (int)(d >> 28);
                if (num_coeffs == 1) n_signs = pack_symbols<1>(qv, cbits, is_signed, vals);
                else if (num_coeffs == 2) n_signs = pack_symbols<2>(qv, cbits, is_signed, vals);
                else n_signs = pack_symbols<4>(qv, cbits, is_signed, vals);
            }
            const int n_sym = 16 / num_coeffs;
            int bits = n_signs + (group_size != 1 ? n_sym / group_size : 0);   // one flag bit per group
            auto len_of = [&](int sidx) {
                const int e = off + (int)vals[sidx];
                return (int)((s_len4[e >> 3] >> (4 * (e & 7))) & 15u);
            };
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) bits += len_of(sidx);
            if (num_coeffs <= 2) {
#pragma unroll
                for (int sidx = 4; sidx < 8; ++sidx) bits += len_of(sidx);
                if (num_coeffs == 1) {
#pragma unroll
                    for (int sidx = 8; sidx < 16; ++sidx) bits += len_of(sidx);
                }
            }
            s_cbits[tid][i] = (uint16_t)bits;
            atomicAdd(&s_qbits[ch][qu][i], (uint32_t)bits);
        }
    }
    __syncthreads();
    if (tid < 64) {   // the first table with the fewest bits
        const uint32_t* b = &s_qbits[0][0][0] + 8 * tid;
        uint32_t best = b[0];
        int bi = 0;
        for (int i = 1; i < 8; ++i)
            if (b[i] < best) {
                best = b[i];
                bi = i;
            }
        (&s_tab[0][0])[tid] = (uint8_t)bi;
        (&s_best[0][0])[tid] = best;
    }
    __syncthreads();
    // ---- the tonal part's bits (TTonalComponentEncoder::Encode :673-718, formed once with 16 subbands) and the number
    //      of quant units: 32, else the largest count <= 28 the frame has room for ----
    uint32_t fl[2] = {0u, 0u};
    int win_bits[2] = {0, 0};
    for (int k = 0; k < nch; ++k) {
        fl[k] = p.flags ? p.flags[item * nch + k] : 0u;
        win_bits[k] = fl[k] == 0u ? 1 : ((fl[k] & 0xffu) == 0xffu ? 2 : 18);   // IsAllSteep keeps its mask in a uint8_t
    }
    const int tonal_bits = (nch == 2 ? 2 : 0) + win_bits[0] + win_bits[1] + nch + 1 + 1 + 2;
    if (wave == 0) {
        const int q = lane & 31;
        int v = (lane < 32) ? (int)(s_best[0][q] + (nch == 2 ? s_best[1][q] : 0u)) : 0;
        v = at3::wave_inclusive_scan(v, lane);   // lane n - 1: the spectra of the first n units
        const uint32_t fixed = W->head[nch - 1][q + 1].fixed_bits;
        const bool fits = lane < 32 && (uint32_t)v + fixed + (uint32_t)tonal_bits <= (uint32_t)kSizeBits;
        const uint32_t mask = (uint32_t)__ballot(fits);
        int n = 1;
        if (mask >> 31) n = 32;
        else if (mask & 0x0fffffffu) n = 32 - __builtin_clz(mask & 0x0fffffffu);
        if (lane == 0) s_n = n;
    }
    __syncthreads();
    const int N = s_n;
    const WriteHead& H = W->head[nch - 1][N];
    // ---- the frame: head, scale factor indices, code table indices, spectra and power groups per channel, tonal part ----
    if (tid < 8 && H.words[tid]) atomicOr(&s_out[tid], H.words[tid]);
    const int pos_sf = H.nbits;
    const int pos_ct = pos_sf + nch * (2 + 6 * N);
    const int pos_data = pos_ct + 1 + nch * (4 + 3 * N);
    if (tid < 64) {
        const int k = tid >> 5, q = tid & 31;
        if (k < nch && q < N) {
            frame_put(s_out, pos_sf + k * (2 + 6 * N) + 2 + 6 * q, s_sfi[k][q], 6);
            frame_put(s_out, pos_ct + 1 + k * (4 + 3 * N) + 4 + 3 * q, s_tab[k][q], 3);
        }
    }
    if (tid == 64) frame_put(s_out, pos_ct, 1u, 1);   // "use full table"
    const bool coded = active && qu < N;
    const int my_tab = s_tab[active ? ch : 0][qu];
    s_cb[ch * 128 + c] = (uint16_t)(coded ? s_cbits[tid][my_tab] : 0);
    __syncthreads();
    {   // prefix sums in stream order: entry u = channel u / 128, chunk u % 128
        const int v = (int)s_cb[tid];
        const int incl = at3::wave_inclusive_scan(v, lane);
        if (lane == 63) s_wsum[wave] = (uint32_t)incl;
        __syncthreads();
        s_off[tid] = (uint32_t)(incl - v) + ((wave & 1) ? s_wsum[wave - 1] : 0u);
    }
    __syncthreads();
    const int ch_total[2] = {(int)(s_wsum[0] + s_wsum[1]), (int)(s_wsum[2] + s_wsum[3])};
    const int pw = 4 * W->sb_powgrps[W->qu_to_sb[N - 1]];
    const int ch_base[2] = {pos_data, pos_data + ch_total[0] + pw};
    if (coded) {
        BitRun run;
        run.start(s_out, ch_base[ch] + (int)s_off[ch * 128 + c]);
        const uint32_t d = s_info[wl - 1 + 7 * my_tab];
        const int num_coeffs = (int)((d >> 20) & 15u);
        if (num_coeffs == 1) emit_chunk<1>(W, qv, d, &run);
        else if (num_coeffs == 2) emit_chunk<2>(W, qv, d, &run);
        else emit_chunk<4>(W, qv, d, &run);
        run.finish();
    }
    if (c == 0 && active) frame_put(s_out, ch_base[ch] + ch_total[ch], (1u << pw) - 1u, pw);   // (15, 4) per power group
    if (tid == 65) {
        int pos = ch_base[nch - 1] + ch_total[nch - 1] + pw;
        if (nch == 2) pos += 2;   // swap_channels, negate_coeffs
        for (int k = 0; k < nch; ++k) {
            if (win_bits[k] == 2) {
                frame_put(s_out, pos, 2u, 2);
            } else if (win_bits[k] == 18) {
                frame_put(s_out, pos, 3u, 2);
                for (int i = 0; i < 16; ++i) frame_put(s_out, pos + 2 + i, (fl[k] >> i) & 1u, 1);
            }
            pos += win_bits[k];
        }
        pos += nch + 1 + 1;   // gain compensation per channel, no tonal block, no noise info
        frame_put(s_out, pos, 3u, 2);
    }
    __syncthreads();
    uint32_t* dst = reinterpret_cast<uint32_t*>(p.out + item * kFrameBytes);
    dst[tid] = __builtin_bswap32(s_out[tid]);
    dst[256 + tid] = __builtin_bswap32(s_out[256 + tid]);
}

}  // namespace at3p
