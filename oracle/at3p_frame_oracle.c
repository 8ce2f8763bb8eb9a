/* TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's ATRAC3plus frame writer without tonal block
 * (second half): TScaler<NAt3p::TScaleTable>::ScaleFrame (atrac/atrac_scale.cpp:141-191 with the
 * tables of atrac/at3p/at3p_tables.h:44-75) and TAt3PBitStream::WriteFrame(channels, nullptr, sces)
 * (atrac/at3p/at3p_bitstream.cpp:99-135 TConfigure, :137-252 TWordLenEncoder, :254-276 TSfIdxEncoder, :278-470
 * TQuantUnitsEncoder, :630-720 TTonalComponentEncoder / WriteFrame) under the repeat protocol of lib/bs_encode/encode.cpp:100-130.
 * Plain C, scalar; pinned against those functions compiled into oracle/_ref by tests/test_at3p_frame.py. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "at3p_vlc.inc"

#define FRAME_SZ 2048
#define SIZE_BITS (FRAME_SZ * 8 - 3) /* FrameSzToAllocBits (at3p_bitstream.cpp:481) */

static const uint8_t kAllocTable[32] = {7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7,
                                        7, 6, 6, 6, 6, 6, 6, 6, 6, 6, 5, 5, 4, 3, 2, 1}; /* TConfigure::Encode */
static const uint16_t kSpecsPerBlock[32] = {16, 16, 16, 16, 16, 16, 16, 16, 32, 32, 32, 32, 32, 32, 32, 32,
                                            64, 64, 64, 64, 64, 64, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128};
static const uint16_t kBlockStart[33] = {0, 16, 32, 48, 64, 80, 96, 112, 128, 160, 192, 224, 256, 288, 320, 352, 384,
                                         448, 512, 576, 640, 704, 768, 896, 1024, 1152, 1280, 1408, 1536, 1664, 1792, 1920, 2048};

typedef struct {
    uint8_t* buf;
    int pos;
} bits_t;

static void put(bits_t* b, uint32_t val, int n) /* NBitStream::TBitStream::Write: MSB first, val's low n bits */
{
    for (int k = n - 1; k >= 0; --k) {
        if (b->pos < FRAME_SZ * 8 && ((val >> k) & 1u)) b->buf[b->pos >> 3] |= (uint8_t)(0x80u >> (b->pos & 7));
        b->pos++;
    }
}
static void put_vlc(bits_t* b, uint16_t e) { put(b, e & 0xfffu, e >> 12); }

/* TScaler::Scale (atrac_scale.cpp:141-172): returns the scale factor index, writes the scaled values */
static int scale_block(const float* in, int len, float* values)
{
    float maxAbs = 0;
    for (int i = 0; i < len; ++i) {
        const float a = fabsf(in[i]);
        if (a > maxAbs) maxAbs = a;
    }
    if (maxAbs > 1.0f) maxAbs = 1.0f;
    int idx = 0;
    while (idx < 63 && AT3P_SCALE[idx] < maxAbs) ++idx; /* map::lower_bound: first key >= maxAbs */
    const float sf = AT3P_SCALE[idx];
    for (int i = 0; i < len; ++i) {
        float v = in[i] / sf;
        if (fabsf(v) >= 1.0f) v = (v > 0) ? (float)0.99999 : (float)-0.99999;
        values[i] = v;
    }
    return idx;
}

/* EncodeQuSpectra (at3p_bitstream.cpp:310-373): bits of `n` mantissas under table `idx`; written out when b != NULL */
static int qu_spectra(const int* q, int n, int idx, bits_t* b)
{
    const int group_size = AT3P_SPEC_TAB[idx][0] & 15, num_coeffs = AT3P_SPEC_TAB[idx][0] >> 4;
    const int bits = AT3P_SPEC_TAB[idx][1] & 15, is_signed = AT3P_SPEC_TAB[idx][1] >> 4;
    const uint16_t* vlc = AT3P_VLC + AT3P_VLC_OFF[idx];
    int total = 0;
    for (int pos = 0; pos < n;) {
        if (group_size != 1) {
            if (b) put(b, 1, 1);
            total += 1;
        }
        for (int j = 0; j < group_size; ++j) {
            uint32_t val = 0;
            int signs[4] = {0, 0, 0, 0};
            for (int i = 0; i < num_coeffs; ++i) {
                int16_t t = (int16_t)q[pos++];
                if (!is_signed && t != 0) {
                    signs[i] = t > 0 ? 1 : -1;
                    if (t < 0) t = (int16_t)-t;
                } else {
                    t = (int16_t)(t & ((1u << bits) - 1));
                }
                t = (int16_t)(t << (bits * i));
                val |= (uint32_t)(uint16_t)t;
            }
            const uint16_t e = vlc[val & 0xffu];
            if (b) put_vlc(b, e);
            total += e >> 12;
            for (int i = 0; i < 4; ++i) {
                if (signs[i] != 0) {
                    if (b) put(b, signs[i] > 0 ? 0 : 1, 1);
                    total += 1;
                }
            }
        }
    }
    return total;
}

/* FindBestWlDeltaEncode (at3p_bitstream.cpp:137-153) */
static int best_wl_table(const int8_t* delta, int sz, int t0, int t1)
{
    int best = 0;
    long consumed = -1;
    for (int i = t0; i <= t1; ++i) {
        long t = 0;
        for (int j = 1; j < sz; ++j) t += AT3P_WL_VLC[i][delta[j]] >> 12;
        if (consumed < 0 || t < consumed) {
            consumed = t;
            best = i;
        }
    }
    return best;
}
static void wl_range(int maxDelta, int* t0, int* t1)
{
    if (maxDelta >= 3) { *t0 = 2; *t1 = 3; }
    else if (maxDelta == 2) { *t0 = 1; *t1 = 1; }
    else { *t0 = 0; *t1 = 0; }
}

/* TWordLenEncoder::Encode (at3p_bitstream.cpp:170-252): channel 0 as deltas to the previous unit, channel 1 as deltas to
 * channel 0; each list under the code table FindBestWlDeltaEncode picks from the range its largest delta allows */
static void wordlen_section(bits_t* b, const uint8_t* wl0, const uint8_t* wl1, int N, int channels)
{
    int8_t d0[32], dx[32];
    int max0 = 0, maxx;
    {
        const int8_t t = (int8_t)(wl1[0] - wl0[0]);
        maxx = abs(t);
        dx[0] = t & 7;
    }
    d0[0] = (int8_t)wl0[0];
    for (int i = 1; i < N; ++i) {
        const int8_t d = (int8_t)(wl0[i] - wl0[i - 1]);
        const int8_t t = (int8_t)(wl1[i] - wl0[i]);
        max0 |= abs(d);
        d0[i] = d & 7;
        maxx |= abs(t);
        dx[i] = t & 7;
    }
    {
        int t0, t1;
        wl_range(max0, &t0, &t1);
        const int idx = best_wl_table(d0, N, t0, t1);
        put(b, 3, 2);
        put(b, 0, 2);
        put(b, 0, 2);
        put(b, (uint32_t)idx, 2);
        put(b, (uint32_t)d0[0], 3);
        for (int i = 1; i < N; ++i) put_vlc(b, AT3P_WL_VLC[idx][d0[i]]);
    }
    if (channels == 2) {
        int t0, t1;
        wl_range(maxx, &t0, &t1);
        const int idx = best_wl_table(dx, N, t0, t1);
        put(b, 1, 2);
        put(b, 0, 2);
        put(b, (uint32_t)idx, 2);
        for (int i = 0; i < N; ++i) put_vlc(b, AT3P_WL_VLC[idx][dx[i]]);
    }
}

/* The section alone, for the reference's own known-answer test (at3p_bitstream_ut.cpp:112-138): returns its bits */
int at3po_wordlen_bits(const uint8_t* wl0, const uint8_t* wl1, int n, int channels, uint8_t* out /* >= 64 bytes, may be NULL */)
{
    uint8_t tmp[FRAME_SZ];
    memset(tmp, 0, sizeof(tmp));
    bits_t b = {tmp, 0};
    wordlen_section(&b, wl0, wl1, n, channels);
    if (out) memcpy(out, tmp, 64);
    return b.pos;
}

/* The tonal-part bits are formed once, with the subband count of the FIRST pass (32 quant units), and kept across the
 * repeats (TTonalComponentEncoder::Encode returns early when BitsUsed != 0, at3p_bitstream.cpp:663-671). */
static void tonal_part(bits_t* b, int channels, const uint16_t* win)
{
    const int sbNum = AT3P_QU_TO_SB[31] + 1;
    if (channels == 2) put(b, 0, 2);
    for (int ch = 0; ch < channels; ++ch) {
        const uint16_t flags = win ? win[ch] : 0;
        const uint8_t mask = (uint8_t)((1 << sbNum) - 1); /* TAt3pMDCTWin::IsAllSteep keeps the mask in a uint8_t */
        if (flags == 0) {
            put(b, 0, 1);
        } else if ((flags & mask) == mask) {
            put(b, 1, 1);
            put(b, 0, 1);
        } else {
            put(b, 1, 1);
            put(b, 1, 1);
            for (int i = 0; i < sbNum; ++i) put(b, (flags >> i) & 1u, 1);
        }
    }
    for (int ch = 0; ch < channels; ++ch) put(b, 0, 1);
    put(b, 0, 1); /* no tonal block */
    put(b, 0, 1);
    put(b, 3, 2);
}

typedef struct {
    int32_t num_quant_units;
    int32_t bits_used;
    uint8_t sfi[2][32];
    uint8_t tab[2][32];
    uint16_t qu_bits[2][32];
} at3po_frame_info;

/* One frame: specs [channels][2048], win [channels] (NULL = all sine) -> 2048 bytes */
static void write_frame(const float* specs, const uint16_t* win, int channels, uint8_t* out, at3po_frame_info* info)
{
    float values[2][2048];
    int mant[2][2048];
    uint8_t sfi[2][32], tab[2][32];
    int qbits[2][32];
    memset(sfi, 0, sizeof(sfi));
    memset(tab, 0, sizeof(tab));
    memset(qbits, 0, sizeof(qbits));
    for (int ch = 0; ch < channels; ++ch) {
        for (int qu = 0; qu < 32; ++qu) {
            const int start = kBlockStart[qu], n = kSpecsPerBlock[qu];
            sfi[ch][qu] = (uint8_t)scale_block(specs + ch * 2048 + start, n, values[ch] + start);
            /* TUnit::GetOrCompute: QuantMantisas without the energy pass, then the cheapest of the eight tables */
            const int wl = kAllocTable[qu];
            const float mul = AT3P_INV_MANT[wl];
            for (int i = 0; i < n; ++i) mant[ch][start + i] = (int)lrintf(values[ch][start + i] * mul);
            long consumed = -1;
            for (int i = 0, ti = wl - 1; i < 8; ++i, ti += 7) {
                const int t = qu_spectra(mant[ch] + start, n, ti, NULL);
                if (consumed < 0 || t < consumed) {
                    consumed = t;
                    tab[ch][qu] = (uint8_t)i;
                }
            }
            qbits[ch][qu] = (int)consumed;
        }
    }
    uint8_t tonal_buf[FRAME_SZ];
    memset(tonal_buf, 0, sizeof(tonal_buf));
    bits_t tb = {tonal_buf, 0};
    tonal_part(&tb, channels, win);
    const int tonal_bits = tb.pos;

    int N = 32;
    for (;;) {
        memset(out, 0, FRAME_SZ);
        bits_t b = {out, 0};
        put(&b, 0, 1);
        put(&b, (uint32_t)channels - 1, 2);
        /* TConfigure */
        put(&b, (uint32_t)N - 1, 5);
        put(&b, 0, 1);
        wordlen_section(&b, kAllocTable, kAllocTable, N, channels);
        /* TSfIdxEncoder */
        for (int ch = 0; ch < channels; ++ch) {
            put(&b, 0, 2);
            for (int i = 0; i < N; ++i) put(&b, sfi[ch][i], 6);
        }
        /* TQuantUnitsEncoder: code table indices, then per channel the spectra and the power compensation groups */
        put(&b, 1, 1);
        for (int ch = 0; ch < channels; ++ch) {
            put(&b, 0, 1);
            put(&b, 0, 2);
            put(&b, 0, 1);
            for (int i = 0; i < N; ++i) put(&b, tab[ch][i], 3);
        }
        for (int ch = 0; ch < channels; ++ch) {
            for (int qu = 0; qu < N; ++qu) qu_spectra(mant[ch] + kBlockStart[qu], kSpecsPerBlock[qu], kAllocTable[qu] - 1 + 7 * tab[ch][qu], &b);
            const int numPwrSpec = AT3P_SB_POWGRPS[AT3P_QU_TO_SB[N - 1]];
            for (int i = 0; i < numPwrSpec; ++i) put(&b, 15, 4);
        }
        /* TTonalComponentEncoder::CheckFrameDone: the 3 leading bits are outside SizeBits */
        const int consumption = b.pos - 3 + tonal_bits;
        if (consumption > SIZE_BITS) {
            N = (N == 32) ? 28 : N - 1;
            continue;
        }
        /* Dump: the tonal part follows */
        for (int i = 0; i < tonal_bits; ++i) put(&b, (tonal_buf[i >> 3] >> (7 - (i & 7))) & 1u, 1);
        if (info) {
            info->num_quant_units = N;
            info->bits_used = b.pos;
            memcpy(info->sfi, sfi, sizeof(sfi));
            memcpy(info->tab, tab, sizeof(tab));
            for (int ch = 0; ch < 2; ++ch)
                for (int qu = 0; qu < 32; ++qu) info->qu_bits[ch][qu] = (uint16_t)qbits[ch][qu];
        }
        return;
    }
}

/* specs [n_frames][channels][2048], win_flags [n_frames][channels] or NULL -> out [n_frames][2048]; info optional [n_frames] */
int at3po_write_frames(const float* specs, const uint16_t* win_flags, int channels, int n_frames, uint8_t* out, void* info)
{
    if (channels < 1 || channels > 2) return -1;
    for (int f = 0; f < n_frames; ++f)
        write_frame(specs + (size_t)f * channels * 2048, win_flags ? win_flags + (size_t)f * channels : NULL, channels,
                    out + (size_t)f * FRAME_SZ, info ? (at3po_frame_info*)info + f : NULL);
    return n_frames;
}
int at3po_frame_info_size(void) { return (int)sizeof(at3po_frame_info); }
