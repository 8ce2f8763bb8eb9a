/*
 * at3hip.h - C ABI of the MI355X-native ATRAC3 encode hot path.
 *
 * Drop-in boundary. The reference (dcherednik/atracdenc) has no FFI: its surface for this path
 * is the C++ class pair TAtrac3Encoder / TAtrac3MDCT. Each entry point below names the
 * reference interface it replaces (paths relative to the reference's src/); the host-side C++
 * mirror of those classes lives in atracdenc_amd/host/at3hip_host.hpp and the binding a
 * maintainer of the reference would add is shown in INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a
 * negative AT3HIP_E* code (never throws, never prints); the caller owns all buffers; one ctx
 * per device, used from one host thread at a time. The library has NO CPU fallback: when no
 * gfx950 device / kernel image is usable, at3hip_create fails with AT3HIP_EDEVICE.
 */
#ifndef AT3HIP_H
#define AT3HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: these declarations are its whole exported surface */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define AT3HIP_OK 0
#define AT3HIP_EINVAL (-1)   /* bad argument / unsupported configuration */
#define AT3HIP_EDEVICE (-2)  /* HIP runtime / device error (see at3hip_last_error) */
#define AT3HIP_ENOMEM (-3)

/* at3hip_encode / at3hip_mdct / at3hip_qmf_mdct flags */
#define AT3HIP_PCM_ON_DEVICE 1u  /* input pointer is device memory (already resident in HBM) */
#define AT3HIP_OUT_ON_DEVICE 2u  /* output pointer is device memory */
#define AT3HIP_ASYNC 4u          /* at3hip_encode only queues the work and returns; pcm must stay valid and out_frames
                                  * must not be read until at3hip_sync(). Consecutive asynchronous calls overlap: the
                                  * front half (QMF, gain control, MDCT) of call N+1 runs beside the back half
                                  * (psychoacoustics, quantisation, rate loop, packing) of call N. */

typedef struct at3hip_ctx at3hip_ctx;

/* Mirrors NAtrac3::TAtrac3EncoderSettings (atrac/at3/atrac3.h:260-277) plus batch geometry. */
typedef struct at3hip_config {
    int32_t bitrate;          /* bit/s as TAtrac3EncoderSettings takes it; 0 = LP2 (132300). The container
                                 row {Bitrate, FrameSz, Js} is chosen like GetContainerParamsForBitrate
                                 (atrac3.cpp:47-53): 66150 -> LP4 192 B joint stereo, 132300 -> LP2 384 B. */
    int32_t channels;         /* SourceChannels: 2, or 1. One channel with a discrete-stereo bitrate: the frame holds the one
                               * sound unit twice (atrac3_bitstream.cpp:836-843); with a joint-stereo bitrate: the mono unit
                               * plus the empty second element of atrac3denc.cpp:843-849. */
    int32_t no_gain_control;  /* NoGainControll */
    int32_t no_tonal;         /* NoTonalComponents */
    int32_t bfu_idx_const;    /* BfuIdxConst (0 = automatic) */
    int32_t n_streams;        /* independent audio streams encoded side by side (batch dimension) */
    int32_t max_blocks;       /* upper bound of PCM blocks per stream per at3hip_encode call */
    int32_t device_id;        /* HIP device ordinal */
} at3hip_config;

/* Per-call device timings in milliseconds (HIP events on the ctx's streams), filled by the last
 * at3hip_encode / at3hip_qmf_mdct call. total_ms spans the first front-half kernel to the last back-half kernel of ONE
 * call (its latency); with AT3HIP_ASYNC consecutive calls overlap, so throughput is not 1 / total_ms.
 * The events sit between the call's kernels and are not free (AT3HIP_OPT_TIMING_EVERY): a call that was not timed reports zeros and
 * qmf_mdct_launches == 0. */
typedef struct at3hip_timings {
    float total_ms;
    float qmf_ms;        /* QMF tree as its own kernel (subbands to HBM: gain control and joint stereo); ~0 when fused */
    float gain_ms;       /* spectral upsampler + AnalyzeGain */
    float curve_ms;      /* CalcCurve / point-0 logic / context scan */
    float qmf_mdct_ms;   /* MDCT-512 from the subbands, or the fused QMF + MDCT kernel (no gain control, discrete stereo).
                          * The QMF + MDCT work of a call (the roofline kernel pair) takes qmf_ms + qmf_mdct_ms. */
    float psy_ms;        /* loudness, flatness, tonal extraction, scale factors */
    float alloc_ms;      /* loudness scan + bit allocation + quantisation + sound-unit packing */
    int32_t qmf_mdct_launches; /* kernels the QMF + MDCT work was spread over: 1 (fused) or 2 */
} at3hip_timings;

/* DEVICE BUFFERS AND STREAMS. With AT3HIP_PCM_ON_DEVICE / AT3HIP_OUT_ON_DEVICE the library reads and writes the caller's buffers on the
 * context's own streams, which are non-blocking: they wait for no other stream, the null stream included. Whatever produces the PCM (or
 * clears the output) on a stream of the caller's must be complete before the call - or the caller hands that stream over with
 * at3hip_set_stream, and the first stage of every call is ordered behind it. (Found the hard way: a benchmark that queued its PCM
 * synthesis with another library and called at3hip_encode at once encoded, on a crowded device, samples that were not there yet.) */

/* Replaces: TAtrac3Encoder::TAtrac3Encoder(TCompressedOutputPtr&&, TAtrac3EncoderSettings&&)
 * (atrac3denc.cpp:93-103) for n_streams encoders at once.
 * A context creates three prioritised HIP streams and a copy stream; the runtime maps a process's streams onto FOUR hardware queues per device
 * by default (GPU_MAX_HW_QUEUES), and streams that share one serialise. Create the contexts at start-up, before the process creates other
 * prioritised streams, and keep them (at3hip_reset starts new audio streams on a context): measured, a first context created behind eight other
 * streams runs at half speed. */
int at3hip_create(const at3hip_config* cfg, at3hip_ctx** out);
void at3hip_destroy(at3hip_ctx* ctx);

/* FrameSz of the selected container row (atrac3.h:211-220). */
int at3hip_frame_size(const at3hip_ctx* ctx);
/* 1 when the container row is joint stereo (LP4). */
int at3hip_joint_stereo(const at3hip_ctx* ctx);
/* Human-readable description of the last error on this ctx (never NULL). */
const char* at3hip_last_error(const at3hip_ctx* ctx);

/* Replaces: n_blocks consecutive calls of the TAtrac3Encoder::GetLambda() functor
 * (atrac3denc.cpp:694-866) on every stream, plus the ICompressedOutput::WriteFrame calls they make
 * (compressed_io.h:56-59, atrac3_bitstream.cpp:845).
 *   pcm        [n_streams][n_blocks][1024][channels] float32, interleaved, +-1.0 (what TPCMEngine hands
 *              to the lambda, pcmengin.h:173-184)
 *   out_frames [n_streams][n_frames][frame_size] bytes, n_frames = *n_frames_out
 * Stream state (QMF history, look-ahead, MDCT overlap, gain-curve context, loudness) is carried
 * between calls, so a stream may be fed in pieces. As in the reference the very first block of a
 * stream only primes the look-ahead (LOOK_AHEAD, atrac3denc.cpp:715-718): the first call returns
 * n_blocks-1 frames per stream, later calls n_blocks. */
int at3hip_encode(at3hip_ctx* ctx, const float* pcm, int32_t n_blocks, uint8_t* out_frames,
                  int32_t* n_frames_out, uint32_t flags);

/* The same with 16-bit PCM: pcm [n_streams][n_blocks][1024][channels] int16, interleaved; every sample becomes
 * s / 32768.0f on the device - what libsndfile's sf_readf_float hands the reference's TPCMEngine for a 16-bit WAV
 * (pcm_io_sndfile.cpp:111-113, pcmengin.h:173-184) - so the frames equal at3hip_encode's on those floats byte for byte.
 * Half the bytes per frame cross the bus: a host-fed context is bound by them ("Host buffers"). Same
 * flags, same stream state (calls of both kinds may alternate on one context). With AT3HIP_PCM_ON_DEVICE the pointer must be
 * 16-byte aligned (the conversion kernel reads eight samples per load; hipMalloc'ed memory is): AT3HIP_EINVAL otherwise. */
int at3hip_encode_s16(at3hip_ctx* ctx, const int16_t* pcm, int32_t n_blocks, uint8_t* out_frames, int32_t* n_frames_out,
                      uint32_t flags);

/* Back to start-of-stream state for every stream (a fresh TAtrac3Encoder). */
int at3hip_reset(at3hip_ctx* ctx);

/* Replaces: TAtrac3MDCT::Mdct(float specs[1024], float* bands[4], TGainModulatorArray)
 * (atrac3denc.h:80-86, atrac3denc.cpp:33-58) with modulators made by
 * TGainProcessor::Modulate(points) (gain_processor.h:87-121), batched over n_items.
 *   bands    [n_items][4][512] float32, each [overlap 256 | new 256]; MUTATED like the reference:
 *            the overlap slot receives EncodeWindow*new and, for bands with gain points, the new
 *            half is divided by the gain ramp.
 *   specs    [n_items][1024] float32 out
 *   n_points [n_items][4], level/loc [n_items][4][8] (int32) or all NULL for no gain modulation. */
int at3hip_mdct(at3hip_ctx* ctx, float* bands, float* specs, const int32_t* n_points,
                const int32_t* level, const int32_t* loc, int32_t n_items, uint32_t flags);

/* Replaces: TAtrac3MDCT::Mdct(float specs[1024], float* bands[4], float maxLevels[4], TGainModulatorArray)
 * (atrac3denc.h:80-83, atrac3denc.cpp:33-58): as at3hip_mdct, plus
 *   max_levels [n_items][4] float32 out: max |new half| per band after gain modulation. */
int at3hip_mdct_levels(at3hip_ctx* ctx, float* bands, float* specs, float* max_levels, const int32_t* n_points,
                       const int32_t* level, const int32_t* loc, int32_t n_items, uint32_t flags);

/* Replaces: static TAtrac3MDCT::CalcGainEnergyScale(prevOverlap[256], curInput[256], gainPoints, prevOverlapScale)
 * (atrac3denc.h:75-79, atrac3denc.cpp:175-224), batched over n_items (one band each).
 *   prev_overlap [n_items][256], cur_input [n_items][256] float32; prev_overlap_scale [n_items]
 *   n_points [n_items], level/loc [n_items][8] (int32) or all NULL for no gain points
 *   out [n_items][4] float32: Scale.PrevHalf, Scale.CurHalf, Scale.Frame, NextOverlapScale
 * Buffers are all host (flags 0) or all device (AT3HIP_PCM_ON_DEVICE|AT3HIP_OUT_ON_DEVICE). */
int at3hip_gain_energy_scale(at3hip_ctx* ctx, const float* prev_overlap, const float* cur_input, const int32_t* n_points,
                             const int32_t* level, const int32_t* loc, const float* prev_overlap_scale, float* out,
                             int32_t n_items, uint32_t flags);

/* The fused batched QMF + windowed MDCT-512 kernel on its own (start-of-stream state, no gain
 * control): PCM [n_streams][n_blocks][1024][channels] -> spectra [n_streams][n_blocks-1][channels][1024].
 * Replaces, per frame and channel: the /4.0 de-interleave + Atrac3AnalysisFilterBank::Analysis
 * (atrac3denc.cpp:701-713, atrac/at3/atrac3_qmf.h:37-41) + TAtrac3MDCT::Mdct (atrac3denc.cpp:802-809).
 * Both pointers must be device memory (flags must contain AT3HIP_PCM_ON_DEVICE|AT3HIP_OUT_ON_DEVICE). */
int at3hip_qmf_mdct(at3hip_ctx* ctx, const float* pcm, int32_t n_blocks, float* specs, uint32_t flags);

/* Timings of the last at3hip_encode / at3hip_qmf_mdct call (after at3hip_sync() for asynchronous calls). */
int at3hip_get_timings(const at3hip_ctx* ctx, at3hip_timings* out);

/* Waits for all queued work of this ctx (the completion point of AT3HIP_ASYNC calls; what a host shim calls before it
 * hands frames to ICompressedOutput::WriteFrame). */
int at3hip_sync(at3hip_ctx* ctx);

/* Stage taps of the most recent at3hip_encode call (the tap points SURVEY.md 8(c) lists; test and diagnosis aid):
 * copies the first `bytes` bytes of an intermediate buffer to host memory. Layouts, with n = blocks of that call,
 * F = frames it produced (frame index f0 .. n-1, f0 = 1 on the first call of a stream else 0):
 *   SPECTRA       float [n_streams][F][2][1024]   spectra after Mdct, tonal lines zeroed        (T3)
 *   CURVES        16-byte records [n_streams][n][2][4]: n, level[7], loc[7], pad - by frame index (T2)
 *   ENERGY_SCALE  float [n_streams][n][2][4] GainEnergyScale.Frame by frame index (gain control only)
 *   PSY           1256-byte records [n_streams][F][2]: float loud_ch, int32 n_tonal, u8 sfi[32], float energy[32],
 *                 tonal blocks 24 x {u16 pos, u8 bfu, len, sfi, pad[3], float values[7], pad[4]}, float flat[32] =
 *                 CalcSpectralFlatnessPerBfu of BFUs 8..28 (0 elsewhere and with NoTonalComponents)  (T4, T5, T6)
 *   LOUDNESS      float [n_streams][F] tracked loudness                                             (T6)
 *   QUANT         1792-byte records [n_streams][F][2]: float err[7][32] (e1/e2), u32 cost[7][32] (CLC | VLC << 13) of the
 *                 units the rate loop quantised (zero = never, or only bounded: the loop brings a unit's bits in when a bound of
 *                 them does not decide its comparison; err is kept for BFUs 0..9 - ConsiderEnergyErr's - and for the units that
 *                 went through the energy-adaptive pass); only with AT3HIP_OPT_QUANT_TAP */
#define AT3HIP_TAP_SPECTRA 1
#define AT3HIP_TAP_CURVES 2
#define AT3HIP_TAP_ENERGY_SCALE 3
#define AT3HIP_TAP_PSY 4
#define AT3HIP_TAP_LOUDNESS 5
#define AT3HIP_TAP_QUANT 6
/* AT3HIP_TAP_CLOCK (diagnostic, 2 x uint64): shader cycles and 100 MHz reference ticks that workgroup 0 of the last call's
 * allocation kernel lived; cycles / ticks x 100 = the shader clock in MHz under the rate loop's load (bench.py reports it
 * as roofline.sclk_mhz_observed). */
#define AT3HIP_TAP_CLOCK 7
/* AT3HIP_TAP_GAIN_ANALYSIS (diagnostic, gain control on): 416-byte records [n_streams][n_blocks of the last call][2][3] of (block,
 * channel, band < 3); the blocks that produced no frame (block 0 of a stream's first call) hold nothing. float highFreqRatio
 * (transient_spectral_upsampler.cpp:99-118), target, mean gain, three context floats, 2 padding floats, float gain[32] (AnalyzeGain's
 * sub-frame RMS), lo[32], hi[32] (quartiles). Records of items below the 5 % gate hold the ratio only. */
#define AT3HIP_TAP_GAIN_ANALYSIS 8
int at3hip_read_tap(at3hip_ctx* ctx, int32_t kind, void* dst, size_t bytes);

/* Timings of the at3hip_encode call `ago` calls back (0 = the most recent one, at most 31); waits for queued work.
 * Zeroed for calls that produced no frames (the LOOK_AHEAD call) and for calls AT3HIP_OPT_TIMING_EVERY left untimed. */
int at3hip_get_timings_ago(at3hip_ctx* ctx, int32_t ago, at3hip_timings* out);

/* Order this ctx's work after a caller-provided hipStream_t (NULL = the ctx's own stream): the first stage of every call
 * is queued on that stream - behind whatever the caller queued there, e.g. the kernel that produces `pcm` -, the later
 * stages run on the ctx's own two streams behind HIP events (the rate loop's at the higher priority). Completion is as
 * before: the call's return, or at3hip_sync() for AT3HIP_ASYNC calls. The stream must outlive its last call's completion. */
int at3hip_set_stream(at3hip_ctx* ctx, void* hip_stream);

/* Options that never change a result: how the work is cut up, or which of two equivalent forms computes it.
 *   AT3HIP_OPT_RUNS              wavefronts ("runs" of consecutive blocks) per (stream, channel) of the QMF / MDCT kernels;
 *                                0 = chosen per call from the batch geometry (default), otherwise >= 1. A run re-derives its FIR
 *                                history and overlap from the samples before its first block, so any cut gives the same bytes.
 *   AT3HIP_OPT_LITERAL_FORMS     0 (default) / 1. The path has two guarded SHORT forms of reference arithmetic; 1 makes both run in
 *                                their LITERAL form for every item (a test and diagnosis aid: same values either way):
 *                                (a) the spectral-flatness measure (CalcSpectralFlatnessPerBfu, atrac_psy_common.cpp:158-199): one
 *                                    log per BFU over a product of mantissas instead of a log per line, literal per-line form as
 *                                    fall-back where rounding could matter;
 *                                (b) highFreqRatio (transient_spectral_upsampler.cpp:99-118): its two 257-term f64 energy sums added
 *                                    in lane order, the f32 kept only when an error bound (4e-13 against a provable 1.15e-13) says
 *                                    the reference's chains round to the same f32, the chains otherwise.
 *                                AT3HIP_OPT_FLATNESS_LITERAL is the former name of this option (same number, kept for source
 *                                compatibility).
 *                                SUPPORTED REFERENCE PLATFORM: both forms restate glibc 2.35's f64 log / exp (and log2f) in the
 *                                variants its ifunc picks on an x86-64 host WITH FMA; glibc selects per CPU, so on a host without
 *                                FMA the reference itself rounds differently in rare last-bit cases and "bit-identical" then means
 *                                identical to the reference run on an FMA host (every box in play). tests/test_libm64.py and the
 *                                gpu-marked pin check the host's libm against the restatement.
 *   AT3HIP_OPT_QUANT_TAP         0 (default) / 1 = keep the AT3HIP_TAP_QUANT records (3.5 KB written per frame).
 *   AT3HIP_OPT_GAIN_FORM         which form of the spectral upsampler / AnalyzeGain kernel runs (same results):
 *                                AT3HIP_GAIN_FORM_TWO_WAVES (0, default) = a two-wavefront workgroup per item,
 *                                AT3HIP_GAIN_FORM_ONE_WAVE (1) = one wavefront per item (faster alone, not in the pipelined step).
 *   AT3HIP_OPT_GAIN_WGS_PER_CU   tuning aid: workgroups per CU of that kernel by LDS padding: 0 = chosen per launch (default),
 *                                1 .. 16 = that many, 256 .. 65536 = the pad itself in bytes.
 *   AT3HIP_OPT_CHAIN             tuning aid for the fused QMF + MDCT kernel (no gain control, discrete stereo): whether the runs of a
 *                                workgroup hand their MDCT overlap on to each other instead of each priming its own from a block of
 *                                PCM: 0 = chosen per call (default), 1 = never, 2 = whenever the cut allows it. Same bytes either way.
 *   AT3HIP_OPT_TIMING_EVERY      which calls record the at3hip_timings stage events: 1 (default) = every call with frames, N > 1 = every
 *                                Nth, 0 = none. The events sit between the kernels of the ctx's streams; recording all of them costs a
 *                                pipelined step of 4096 frames 3 %. A call that was not timed reports all-zero timings with
 *                                qmf_mdct_launches == 0. Same bytes either way.
 * Values outside the ranges above are rejected with AT3HIP_EINVAL (nothing is stored). */
#define AT3HIP_OPT_RUNS 1
#define AT3HIP_OPT_LITERAL_FORMS 2
#define AT3HIP_OPT_FLATNESS_LITERAL AT3HIP_OPT_LITERAL_FORMS
#define AT3HIP_OPT_QUANT_TAP 3
#define AT3HIP_OPT_GAIN_FORM 4
#define AT3HIP_OPT_GAIN_TWO_WAVES AT3HIP_OPT_GAIN_FORM   /* the option's former name (same number; its former value 2 = AT3HIP_GAIN_FORM_ONE_WAVE is still accepted) */
#define AT3HIP_OPT_GAIN_WGS_PER_CU 5
#define AT3HIP_OPT_CHAIN 6
#define AT3HIP_OPT_TIMING_EVERY 7
#define AT3HIP_GAIN_FORM_TWO_WAVES 0
#define AT3HIP_GAIN_FORM_ONE_WAVE 1
int at3hip_set_option(at3hip_ctx* ctx, int32_t option, int32_t value);

/* The reference's overflow diagnostics as counters ("these conditions become counters, not prints").
 * TScaler::Scale (atrac_scale.cpp:141-172) prints "Scale error: absSpec > MAX_SCALE" once per block (BFU or tonal component) whose
 * largest magnitude exceeds 1.0 and "clipping, scaled value: ..." once per value whose scaled magnitude exceeds 1.0; the results
 * are clamped either way (and are bit-identical here). The counters accumulate over every frame this context has encoded since
 * at3hip_create / at3hip_reset / the last at3hip_get_counters(..., reset = 1), summed over streams, and count what ONE
 * TAtrac3Encoder per stream would have printed (a one-channel stream is scaled once, like the reference's single channel).
 * Waits for queued work. */
typedef struct at3hip_counters {
    uint64_t scale_overflow;   /* "Scale error" lines: blocks with max |spectrum| > MAX_SCALE (1.0) */
    uint64_t clipped_values;   /* "clipping" lines: values with |value / scale factor| > 1.0 */
} at3hip_counters;
int at3hip_get_counters(at3hip_ctx* ctx, at3hip_counters* out, int32_t reset);

/* Host-buffer pipeline. The reference's caller hands host floats (TPCMEngine::ApplyProcess, pcmengin.h:152-192): with
 * page-locked buffers from at3hip_host_alloc and AT3HIP_ASYNC calls that alternate between two input and two output
 * buffers, the H2D copy of call N+1 (on a copy stream of the ctx, into device staging double-buffered by call parity), the
 * kernels of call N and the D2H copy of call N-1's frames overlap. `ago` = 0 for the most recent at3hip_encode call, 1
 * for the one before it, up to 3: three calls in flight (three input and three output buffers) keep the device's own
 * three-stage overlap of consecutive calls busy - with 16-bit samples the bus no longer hides a two-deep pipeline's bubbles.
 *   at3hip_wait_input   returns once that call's PCM has left the host buffer (it may be refilled)
 *   at3hip_wait_frames  returns once that call's frames are in out_frames (the completion point of ONE call; at3hip_sync
 *                       waits for all of them)
 * Pageable host memory works too (the copies then block the calling thread, as hipMemcpyAsync does for such memory). */
int at3hip_host_alloc(at3hip_ctx* ctx, size_t bytes, void** out);
/* The host NUMA node the device's PCIe link hangs off (hipDeviceGetPCIBusId -> /sys/bus/pci/devices/<id>/numa_node), or -1 when the
 * platform does not say (single-node hosts, containers without sysfs). A node-level driver pins the host thread that feeds a device -
 * and with it the first-touch placement of the page-locked buffers it allocates - to that node: eight host-fed devices pull ~50 GB/s
 * each, which a single node's memory controllers do not deliver (atracdenc_amd/host/at3hip_host.hpp: TAtrac3EncoderNode). Needs no ctx. */
int at3hip_device_numa_node(int32_t device_id);
int at3hip_host_free(at3hip_ctx* ctx, void* p);
int at3hip_wait_input(at3hip_ctx* ctx, int32_t ago);
int at3hip_wait_frames(at3hip_ctx* ctx, int32_t ago);

/* The constant tables exactly as at3hip_create builds them on this host (libm expressions of the reference's static
 * initialisers, atrac3.h:178-198, qmf.cpp:36-45, atrac_psy_common.cpp:126-156, ...) - no GPU involved. `bytes` must be the
 * size of the table block (see atracdenc_amd/csrc/at3_tables.hpp; the ctypes stub mirrors the layout). Lets a parity
 * suite prove, on the machine that runs the encoder, that the tables equal the reference's. */
int at3hip_host_tables(void* dst, size_t bytes);

/* Library/ABI version: (major << 16) | minor. The minor number grows with every addition to this header:
 *   1.1  rounds 1 - 3 (two calls in flight: at3hip_wait_* accept ago 0 .. 1)
 *   1.2  at3hip_encode_s16, at3hip_wait_* with ago 0 .. 3 (three calls in flight), AT3HIP_TAP_CLOCK / AT3HIP_TAP_GAIN_ANALYSIS,
 *        AT3HIP_OPT_GAIN_FORM / AT3HIP_OPT_GAIN_WGS_PER_CU / AT3HIP_OPT_LITERAL_FORMS with validated values
 *   1.3  at3hip_get_counters; AT3HIP_OPT_GAIN_FORM's values renumbered (the former AT3HIP_OPT_GAIN_TWO_WAVES: 0 = one workgroup of two
 *        wavefronts per item, 1 = the one-wavefront form, formerly 2 - the legacy value 2 is still accepted and means ONE_WAVE)
 *   1.4  AT3HIP_OPT_CHAIN, at3hip_device_numa_node
 *   1.5  AT3HIP_OPT_TIMING_EVERY
 * A host layer compiled against this header checks at3hip_version() >= AT3HIP_VERSION before it relies on them
 * (atracdenc_amd/host/at3hip_host.hpp and the ctypes stub do). */
#define AT3HIP_VERSION_MAJOR 1
#define AT3HIP_VERSION_MINOR 5
#define AT3HIP_VERSION ((AT3HIP_VERSION_MAJOR << 16) | AT3HIP_VERSION_MINOR)
uint32_t at3hip_version(void);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* AT3HIP_H */
