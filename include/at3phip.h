/* at3phip.h - C ABI of the MI355X-native ATRAC3plus path: the 16-band polyphase analysis
 * filter and the windowed MDCT-256 x 16 that turn PCM into the 2048-line spectrum (at3p.cpp:93-99, 139-159), and the
 * frame writer that scales and packs it when there is no tonal block (at3p.cpp:159-163: ScaleFrame, then
 * TAt3PBitStream::WriteFrame(channels, nullptr, sces)). The tonal (GHA) analysis between them needs libgha, an
 * un-vendored submodule of the reference, and is not part of this row: at3phip_encode_frames is the encoder with that
 * analysis finding nothing. Same library (libat3hip.so) and error codes as at3hip.h.
 */
#ifndef AT3PHIP_H
#define AT3PHIP_H

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

#define AT3PHIP_FRAME 2048            /* TAt3PEnc::NumSamples, samples per channel and frame */
#define AT3PHIP_RESIDUAL_SCALE 16u    /* at3phip_mdct / at3phip_pqf_mdct: divide the subband samples by 32768 / 1.122018 first,
                                       * as EncodeFrame does for the residual spectrum (at3p.cpp:143-147) */

typedef struct at3phip_ctx at3phip_ctx;

typedef struct at3phip_config {
    int32_t channels;    /* 1 or 2 */
    int32_t n_streams;   /* independent streams side by side */
    int32_t max_frames;  /* upper bound of 2048-sample frames per stream per call */
    int32_t device_id;
} at3phip_config;

int at3phip_create(const at3phip_config* cfg, at3phip_ctx** out);
void at3phip_destroy(at3phip_ctx* ctx);
const char* at3phip_last_error(const at3phip_ctx* ctx);
/* Start-of-stream state: zeroed filter history (at3plus_pqf_create_a_ctx) and MDCT work buffers (TChannelCtx::MdctBuf). */
int at3phip_reset(at3phip_ctx* ctx);

/* Replaces: at3plus_pqf_do_analyse(ctx, in, out) (atrac/atrac3plus_pqf/atrac3plus_pqf.c:130-147) per channel and frame.
 *   pcm   [n_streams][n_frames][2048][channels] float32 interleaved (the `data` of EncodeFrame, at3p.cpp:93-97)
 *   bands [n_streams][n_frames][channels][16][128] float32: 16 subbands x 128 samples
 * flags: AT3HIP_PCM_ON_DEVICE / AT3HIP_OUT_ON_DEVICE. The 368-sample filter history is carried between calls. */
int at3phip_pqf_analyse(at3phip_ctx* ctx, const float* pcm, int32_t n_frames, float* bands, uint32_t flags);

/* Replaces: TAt3pMDCT::Do(specs, bands, work, winType) (atrac/at3p/at3p_mdct.cpp:52-96) per channel and frame.
 *   bands     as above
 *   win_flags [n_streams][n_frames][channels] uint16, bit b = TAt3pMDCTWin::STEEP for subband b; NULL = all sine
 *   specs     [n_streams][n_frames][channels][2048] float32
 * The work buffer's first halves (THistBuf) are carried between calls. win_flags is always host memory. */
int at3phip_mdct(at3phip_ctx* ctx, const float* bands, int32_t n_frames, const uint16_t* win_flags, float* specs, uint32_t flags);

/* Both steps back to back, the subband samples staying in HBM (optionally returned through `bands`, may be NULL). */
int at3phip_pqf_mdct(at3phip_ctx* ctx, const float* pcm, int32_t n_frames, const uint16_t* win_flags, float* bands, float* specs,
                     uint32_t flags);

/* Replaces: per frame, sces[ch].ScaledBlocks = TScaler<NAt3p::TScaleTable>::ScaleFrame(specs) for each channel
 * (at3p.cpp:159, atrac/atrac_scale.cpp:141-191) and TAt3PBitStream::WriteFrame(channels, nullptr, sces)
 * (atrac/at3p/at3p_bitstream.cpp:694-726): fixed word lengths per quant unit, the cheapest of eight code tables per unit,
 * the number of quant units lowered from 32 to 28, 27, ... until the frame fits.
 *   specs     [n_streams][n_frames][channels][2048] float32 (flags & AT3HIP_PCM_ON_DEVICE: device memory)
 *   win_flags [n_streams][n_frames][channels] uint16 as in at3phip_mdct (TSubbandInfos::Win); NULL = all sine. Host memory.
 *   frames    [n_streams][n_frames][2048] bytes, each what ICompressedOutput::WriteFrame receives
 *             (flags & AT3HIP_OUT_ON_DEVICE: device memory)
 * Frames do not depend on one another. */
#define AT3PHIP_FRAME_BYTES 2048
int at3phip_write_frames(at3phip_ctx* ctx, const float* specs, int32_t n_frames, const uint16_t* win_flags, uint8_t* frames,
                         uint32_t flags);

/* PCM to frames: at3phip_pqf_mdct with AT3PHIP_RESIDUAL_SCALE and sine windows, then at3phip_write_frames, everything in
 * between staying in HBM. This is TAt3PEnc::EncodeFrame (at3p.cpp:89-170) with GHA_PASS_INPUT | GHA_WRITE_RESIUDAL and a
 * tonal analysis that finds nothing; frame f holds input frame f (the reference's two-frame look-ahead delay is the
 * host's to add, see atracdenc_amd/host/at3hip_host.hpp). pcm as in at3phip_pqf_analyse, frames as above. */
int at3phip_encode_frames(at3phip_ctx* ctx, const float* pcm, int32_t n_frames, uint8_t* frames, uint32_t flags);
/* With AT3HIP_ASYNC in `flags` at3phip_encode_frames only queues the call (pcm must stay valid, frames must not be read)
 * and the frame writer of one call runs beside the filter bank and transform of the next (its own stream, spectra
 * double-buffered); at3phip_sync waits for everything queued. Without the flag the call waits itself. A queued call records no
 * stage-timing events (they sit between the kernels and cost the chain): the timing getters then read zero; a synchronous call
 * is timed as before. */
int at3phip_sync(at3phip_ctx* ctx);

/* Device milliseconds the frame writer took in the last at3phip_write_frames / at3phip_encode_frames call. */
int at3phip_get_write_timing(const at3phip_ctx* ctx, float* write_ms);

/* Device milliseconds of the last call: {pqf, mdct}. */
int at3phip_get_timings(const at3phip_ctx* ctx, float* pqf_ms, float* mdct_ms);

/* The constant tables as at3phip_create builds them, on the host (no GPU needed); bytes = AT3PHIP_TABLES_BYTES. */
#define AT3PHIP_TABLES_BYTES 3456
int at3phip_host_tables(void* dst, size_t bytes);
/* The frame writer's tables (code tables, scale table, the spectrum-independent leading bits per channel count and
 * number of quant units) as at3phip_create builds them; bytes = AT3PHIP_WRITE_TABLES_BYTES. */
#define AT3PHIP_WRITE_TABLES_BYTES 41384
int at3phip_host_write_tables(void* dst, size_t bytes);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
