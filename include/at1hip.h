/* at1hip.h - C ABI of the MI355X-native ATRAC1 encode path.
 *
 * Drop-in boundary: what a host shim inside the reference binds in place of the body of the lambda returned by
 * TAtrac1Encoder::GetLambda (atrac1denc.cpp:180-255) - per 512-sample block and channel: analysis filter bank,
 * transient detection, block-switched MDCT, loudness tracking, scale factors, bit allocation and the 212-byte sound
 * unit - for a batch of independent streams. The container (AEA header, frame writing: aea.cpp) stays on the host.
 * Plain pointers and sizes only; same library (libat3hip.so) and error codes as at3hip.h.
 */
#ifndef AT1HIP_H
#define AT1HIP_H

#include <stddef.h>
#include <stdint.h>

#include "at3hip.h"

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: these declarations are its whole exported surface */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define AT1HIP_FRAME_SIZE 212   /* TAtrac1Data::SoundUnitSize, atrac/at1/atrac1.h:110 */
#define AT1HIP_BLOCK 512        /* TAtrac1Data::NumSamples, atrac/at1/atrac1.h:121 */

typedef struct at1hip_ctx at1hip_ctx;

/* Mirrors NAtrac1::TAtrac1EncodeSettings (atrac/at1/atrac1.h:33-54) plus batch geometry. */
typedef struct at1hip_config {
    int32_t channels;       /* 1 or 2 (Aea->GetChannelNum()) */
    int32_t window_auto;    /* 1 = EWM_AUTO (transient detection), 0 = EWM_NOTRANSIENT with window_mask */
    int32_t window_mask;    /* bit 0 low, bit 1 mid, bit 2 high band use short windows (atrac1denc.cpp:227) */
    int32_t bfu_idx_const;  /* BfuIdxConst: 0 = automatic, else 1..8 */
    int32_t n_streams;      /* independent audio streams encoded side by side */
    int32_t max_blocks;     /* upper bound of 512-sample blocks per stream per at1hip_encode call */
    int32_t device_id;
} at1hip_config;

typedef struct at1hip_timings {
    float total_ms;
    float front_ms;   /* QMF tree + transient detection + MDCT + scale factors */
    float scan_ms;    /* loudness tracking */
    float pack_ms;    /* bit allocation + sound-unit packing */
} at1hip_timings;

/* Replaces: TAtrac1Encoder::TAtrac1Encoder(TCompressedOutputPtr&&, TAtrac1EncodeSettings&&) (atrac1denc.cpp:36-44)
 * for n_streams encoders at once. */
int at1hip_create(const at1hip_config* cfg, at1hip_ctx** out);
void at1hip_destroy(at1hip_ctx* ctx);
const char* at1hip_last_error(const at1hip_ctx* ctx);

/* Replaces: n_blocks invocations of the lambda of TAtrac1Encoder::GetLambda per stream.
 *   pcm        [n_streams][n_blocks][512][channels] float32, interleaved, +-1.0 (pcmengin.h:173-184)
 *   out_frames [n_streams][n_blocks][channels][212] bytes: the buffers handed to ICompressedOutput::WriteFrame, in the
 *              reference's order (channel 0 then channel 1 of each block, atrac1denc.cpp:249-251)
 * flags: AT3HIP_PCM_ON_DEVICE / AT3HIP_OUT_ON_DEVICE as for at3hip_encode; AT3HIP_ASYNC only queues the call (buffers must stay
 * valid until at1hip_sync; host buffers then have to be page-locked for the copies to be asynchronous): consecutive calls follow
 * each other on the device without the host in between. Stream state (filter histories, the high band's delay line, MDCT
 * overlap, detector energies, loudness) is carried between calls. */
int at1hip_encode(at1hip_ctx* ctx, const float* pcm, int32_t n_blocks, uint8_t* out_frames, uint32_t flags);

/* Waits for everything queued on the ctx. (A queued call records no stage-timing events - they are not free between the kernels -: the
 * timings then read zero; a synchronous call is timed.) */
int at1hip_sync(at1hip_ctx* ctx);

/* Back to start-of-stream state for every stream (a fresh TAtrac1Encoder). */
int at1hip_reset(at1hip_ctx* ctx);

int at1hip_get_timings(const at1hip_ctx* ctx, at1hip_timings* out);

/* Intermediate results of the last at1hip_encode call, copied to host memory `dst` (test / debugging interface):
 *   AT1HIP_TAP_SPECTRA  float32 [n_streams][n_blocks][channels][512]  TAtrac1MDCT::Mdct output
 *   AT1HIP_TAP_MASKS    int32   [n_streams][n_blocks][channels]       window masks
 *   AT1HIP_TAP_LOUDNESS float32 [n_streams][n_blocks]                 Loudness after each block's TrackLoudness
 *   AT1HIP_TAP_TABLES   the constant tables as uploaded (at1_tables.hpp layout) */
#define AT1HIP_TAP_SPECTRA 1
#define AT1HIP_TAP_MASKS 2
#define AT1HIP_TAP_LOUDNESS 3
#define AT1HIP_TAP_TABLES 4
int at1hip_read_tap(at1hip_ctx* ctx, int32_t kind, void* dst, size_t bytes);

/* The constant tables as at1hip_create builds them, on the host (no GPU needed): `bytes` must be the size of the
 * at1_tables.hpp layout (AT1HIP_TABLES_BYTES). */
#define AT1HIP_TABLES_BYTES 6904
int at1hip_host_tables(void* dst, size_t bytes);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
