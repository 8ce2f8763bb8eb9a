// libat3hip: C-ABI host layer (include/at3hip.h) over the gfx950 kernels. No CPU fallback: every
// compute entry point launches HIP kernels; failures are reported as error codes.
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>

#include "../../include/at3hip.h"
#include "at3_common.hpp"
#include "at3_host_util.hpp"
#include "at3_k_backend.hpp"
#include "at3_k_alloc.hpp"
#include "at3_k_frontend.hpp"
#include "at3_k_front2.hpp"
#include "at3_k_gain.hpp"

using namespace at3;

namespace {

const struct {
    uint32_t bitrate;
    uint16_t frame_sz;
    uint8_t js;
} kContainer[8] = {{66150, 192, 1},  {93713, 272, 1},  {104738, 304, 0}, {132300, 384, 0},
                   {146081, 424, 0}, {176400, 512, 0}, {264600, 768, 0}, {352800, 1024, 0}};  // atrac3.h:211-220

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

}  // namespace

// AT3HIP_TAP_CLOCK: 16 header words, 2 x 256 rows of 12 per-phase cycle sums (k_alloc_pack, k_gain_analysis1), then, in profiling
// builds, the entry and exit times (100 MHz) of k_alloc_pack's first 16384 workgroups
#ifdef AT3HIP_DEBUG_KNOBS
constexpr size_t kClkWords = 16 + 2 * 256 * 12 + 2 * 16384 + 12 * 16384;   // (... and their own per-phase cycles)
#else
constexpr size_t kClkWords = 16;   // release builds write words 0 and 1 only (the phase rows exist in profiling builds)
#endif
struct at3hip_ctx {
    at3hip_config cfg;
    int frame_sz = 0;
    int js = 0;
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;        // front half: QMF, gain control, fused QMF+MDCT, carried state
    hipStream_t back_stream = nullptr;   // back half: psychoacoustics, quantisation, rate loop, packing
    // With gain control the front half is two stages: the heavy one (QMF, spectra of the gain analysis, upsampled envelopes)
    // depends only on the PCM, the light one (curve context scan, curves, energy scales, MDCT) on it and on the previous
    // call's light stage. The light stage is a chain of short latency-bound kernels; on a stream of its own it runs under
    // the next call's heavy stage and the current call's back half instead of holding the GPU nearly idle between them.
    hipStream_t mid_stream = nullptr;
    float* d_sub_b[2] = {nullptr, nullptr};      // subbands, by call parity (the heavy stage runs one call ahead)
    GainRec* d_rec_b[2] = {nullptr, nullptr};
    hipEvent_t ev_mid_done[2] = {};              // light stage finished with the parity's subbands and gain records
    bool mid_done_valid[2] = {false, false};
    hipEvent_t mid_done_of[2] = {};              // the event that says so for the parity's last call: its ev_mdct_done, or ev_mid_done (not owned)
    // The back half of call N only consumes what the front half of call N produced (spectra, curves, energy scales),
    // and the front half of call N+1 only depends on the front half of call N (carried state): the two halves run on
    // HIP streams of their own (three with gain control, see mid_stream below), buffers that cross between them are double-buffered by call parity, and consecutive calls overlap.
    static constexpr int kSlots = 32;    // timing history (events per call)
    hipEvent_t ev[kSlots][8] = {};
    hipEvent_t ev_front_done = nullptr;  // everything the most recent call queued on `stream` (which may be the caller's)
    bool front_done_valid = false;
    // Host PCM: staged through a copy stream into device staging that is double-buffered by call parity, so that the
    // H2D copy of call N+1 runs beside the kernels of call N (pinned host memory: at3hip_host_alloc).
    hipStream_t h2d_stream = nullptr;
    float* d_pcm_in_b[2] = {nullptr, nullptr};      // [S][max_blocks][1024][channels]
    int16_t* d_s16_b[2] = {nullptr, nullptr};       // the same as 16-bit samples (at3hip_encode_s16 with host memory), by call parity
    hipEvent_t ev_h2d[2] = {};                      // the parity's PCM has arrived
    hipEvent_t ev_pcm_free[2] = {};                 // the parity's staging has been consumed (front half done with it)
    bool pcm_free_valid[2] = {false, false};
    bool h2d_valid[2] = {false, false};
    hipEvent_t ev_back_done[2] = {};     // back half finished with the parity's cross buffers (the ev_host_out of that call: not owned)
    // what the HOST waits for (at3hip_wait_input / at3hip_wait_frames), per call in a ring of four: the parity events above are
    // re-recorded by the call after next, so a caller with three calls in flight would wait for the newest of them
    static constexpr int kHostRing = 4;
    hipEvent_t ev_host_in[kHostRing] = {}, ev_host_out[kHostRing] = {};
    bool host_in_valid[kHostRing] = {}, host_out_valid[kHostRing] = {};
    bool back_done_valid[2] = {false, false};
    long long enc_calls = 0;             // at3hip_encode calls so far
    int last_slot = -1;                  // slot of the most recent call that produced frames
    bool slot_has_frames[kSlots] = {};
    int slot_k1_launches[kSlots] = {};   // kernels the QMF + MDCT work of the slot's call was spread over (1 = fused, 2)
    char err[256] = {0};
    long long blocks_fed = 0;   // per stream
    int chain_mode = 0;      // AT3HIP_OPT_CHAIN: 0 = chosen per call, 1 = never, 2 = whenever the geometry allows
    int timing_every = 1;    // AT3HIP_OPT_TIMING_EVERY: stage timings on every Nth call with frames (0 = never)
    unsigned timing_tick = 0;
    bool slot_timed[kSlots] = {};
    hipEvent_t ev_heavy_done[kSlots] = {}, ev_mdct_done[kSlots] = {};   // a call's own hand-overs between its streams (no timestamps)
    int runs_override = 0;   // AT3HIP_OPT_RUNS: runs per (stream, channel) of the front-end kernels (tuning aid; output is invariant)
    int flat_literal = 0;    // AT3HIP_OPT_LITERAL_FORMS
    int gain_form = AT3HIP_GAIN_FORM_TWO_WAVES;   // AT3HIP_OPT_GAIN_FORM: the two-wavefront workgroups (default) or one wavefront per item (k_gain_analysis1)
    int gain_wgs_per_cu = 0; // AT3HIP_OPT_GAIN_WGS_PER_CU: k_gain_analysis1 workgroups per CU (dynamic LDS padding; 0 = chosen per launch)
    int n_cus = 256;
    size_t lds_per_cu = 0;     // hipDeviceProp_t::maxSharedMemoryPerMultiProcessor; the whole-round LDS padding below is tuned for 160 KB
    int wgs_per_cu = 3;        // resident workgroups per CU of the QMF kernel this context uses (k_qmf_sub8 or the fused one)
    int wgs_per_cu_mdct = 3;   // the same of k_mdct_sub
    int alloc_lds_pad = 0;     // dynamic LDS added to k_alloc_pack's launch: sets how many of its workgroups share a CU
    int dbg_pad[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // AT3HIP_DEBUG_KNOBS builds: dynamic LDS added to other launches (AT3HIP_PAD_*), co-residency experiments
    int dbg_front = 0, dbg_gain = 0, dbg_stop = 0;   // AT3HIP_DEBUG_* (profiling aids), honoured by -DAT3HIP_DEBUG_KNOBS builds only

    Tables* d_tables = nullptr;
    float* d_pcm_in = nullptr;       // one-channel contexts: the samples as (x, x) pairs [S][max_blocks][1024][2]
    float* d_hist[2] = {nullptr, nullptr};
    int hist_cur = 0;
    float* d_sub = nullptr;
    float* d_sub_tail = nullptr;     // [S][8][512] subbands of the last two blocks of the previous call
    GainRec* d_rec = nullptr;
    cpx* d_bins = nullptr;           // [S][B][6][kGainBins] high-passed rfft bins, k_gain_spec -> k_gain_analysis
    BandState* d_state = nullptr;
    Curve* d_curves[2] = {nullptr, nullptr};   // by call parity
    float* d_specs[2] = {nullptr, nullptr};
    float* d_ges[2] = {nullptr, nullptr};
    PsyRec* d_psy = nullptr;
    float* d_loud = nullptr;
    float* d_loud_state = nullptr;
    uint8_t* d_out = nullptr;
    QuantRec* d_quant = nullptr;     // allocated by AT3HIP_OPT_QUANT_TAP
    unsigned long long* d_clk = nullptr;   // AT3HIP_TAP_CLOCK (kClkWords)
    unsigned long long* d_counters = nullptr;   // at3hip_get_counters: [0] "Scale error" blocks, [1] "clipping" values (k_psy adds)
    at3hip_timings tm = {};
    // grow-only device staging of the stage-level entry points (at3hip_mdct, at3hip_gain_energy_scale) for host buffers
    void* d_stage = nullptr;
    size_t stage_bytes = 0;
};

namespace {

int fail(at3hip_ctx* c, int code, const char* what, hipError_t e = hipSuccess)
{
    if (c) {
        if (e != hipSuccess) snprintf(c->err, sizeof(c->err), "%s: %s", what, hipGetErrorString(e));
        else snprintf(c->err, sizeof(c->err), "%s", what);
    }
    return code;
}

#define HIPCHK(c, call)                                                  \
    do {                                                                 \
        hipError_t e_ = (call);                                          \
        if (e_ != hipSuccess) return fail((c), AT3HIP_EDEVICE, #call, e_); \
    } while (0)

template <typename Tp>
int dev_alloc(at3hip_ctx* c, Tp** p, size_t count)
{
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, count * sizeof(Tp) + 256);
    if (e != hipSuccess) return fail(c, AT3HIP_ENOMEM, "hipMalloc", e);
    *p = (Tp*)q;
    return AT3HIP_OK;
}

// Device staging area of at least `bytes` bytes (reallocated only when a call needs more than any call before).
int stage_reserve(at3hip_ctx* c, size_t bytes)
{
    if (bytes <= c->stage_bytes) return AT3HIP_OK;
    if (c->d_stage) (void)hipFree(c->d_stage);
    c->d_stage = nullptr;
    c->stage_bytes = 0;
    const size_t want = bytes + bytes / 2 + 4096;
    hipError_t e = hipMalloc(&c->d_stage, want);
    if (e != hipSuccess) return fail(c, AT3HIP_ENOMEM, "hipMalloc (staging)", e);
    c->stage_bytes = want;
    return AT3HIP_OK;
}

// Waits for everything this context has queued on its streams.
int drain(at3hip_ctx* c)
{
    if (c->h2d_stream) HIPCHK(c, hipStreamSynchronize(c->h2d_stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->mid_stream) HIPCHK(c, hipStreamSynchronize(c->mid_stream));
    HIPCHK(c, hipStreamSynchronize(c->back_stream));
    return AT3HIP_OK;
}

// Stage timings of the call recorded in `slot` (its events must have completed).
void read_timings(const at3hip_ctx* c, int slot, at3hip_timings* tm)
{
    memset(tm, 0, sizeof(*tm));
    if (slot < 0 || !c->slot_has_frames[slot] || !c->slot_timed[slot]) return;   // (all zero, qmf_mdct_launches 0: the call was not timed)
    hipEvent_t const* ev = c->ev[slot];
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, ev[0], ev[1]); tm->qmf_ms = ms;
    (void)hipEventElapsedTime(&ms, ev[1], ev[2]); tm->gain_ms = ms;
    (void)hipEventElapsedTime(&ms, ev[2], ev[3]); tm->curve_ms = ms;
    (void)hipEventElapsedTime(&ms, ev[3], ev[4]); tm->qmf_mdct_ms = ms;
    (void)hipEventElapsedTime(&ms, ev[5], ev[6]); tm->psy_ms = ms;
    (void)hipEventElapsedTime(&ms, ev[6], ev[7]); tm->alloc_ms = ms;
    (void)hipEventElapsedTime(&ms, ev[0], ev[7]); tm->total_ms = ms;   // first front-half kernel to last back-half kernel
    tm->qmf_mdct_launches = c->slot_k1_launches[slot];
}

// Runs per (stream, channel) for the wavefront-per-run kernels (QMF, MDCT, fused): `items` blocks or frames are cut
// into runs of at most 32, one wavefront each. A run costs its items plus a prologue of `prologue` items (FIR histories,
// overlap priming); a SIMD with w resident wavefronts issues at roughly eff(w) of its peak. The run count with the
// smallest estimated time wins: small batches get as many equal runs as one round of the grid holds, large batches long
// runs.
int pick_runs(const at3hip_ctx* c, int items, int wgs_per_cu, double prologue, int step = 1)
{
    if (c->runs_override > 0) return c->runs_override < items ? c->runs_override : items;
    if (items < step) step = 1;   // (`step`: only multiples of it are considered - whole workgroups per (stream, channel), see xcd_pair)
    const long long pairs = 2LL * c->cfg.n_streams;   // (stream, channel)
    const long long simds = (long long)c->n_cus * 4;
    const int cap = wgs_per_cu;                        // resident wavefronts per SIMD (a workgroup is one per SIMD)
    static const double eff[5] = {0.0, 0.55, 0.85, 0.95, 1.0};
    int best = 1;
    double best_t = 1e300;
    const int r_min = (items + 31) / 32;
    const int r_max = r_min > 128 ? r_min + 16 : 128;   // (a run holds at most 32 items: long calls need more than 128 runs)
    for (int r = (r_min + step - 1) / step * step; r <= items && r <= r_max; r += step) {
        const long long waves = pairs * r;
        const long long per_simd = (waves + simds - 1) / simds;   // the fullest SIMD
        const double work = (double)((items + r - 1) / r) + prologue;   // the longest run
        const long long resident = per_simd < cap ? per_simd : cap;
        const double t = (double)per_simd * work / eff[resident > 4 ? 4 : resident];
        if (t < best_t * 0.999) {
            best_t = t;
            best = r;
        }
    }
    return best;
}

// The fused kernel's cut: runs per (stream, channel) and whether the runs of a workgroup are CHAINED (k_qmf_mdct8: kFusedWaves consecutive
// runs share one priming block and hand their overlap on through LDS instead of each re-deriving it from a whole block of PCM). A chained
// wavefront costs its share of (the group's frames + 1) blocks, its FIR prologue and the hand-over with the deferred transform (~0.65 block);
// an unchained one its frames + 1.35. Same time model as pick_runs; chaining wins whenever runs are short (small batches).
struct FusedCut {
    int runs, chain;
};
FusedCut pick_fused(const at3hip_ctx* c, int items)
{
    const int NW = kFusedWaves;
    FusedCut cut = {pick_runs(c, items, c->wgs_per_cu, 1.35), 0};
    if (items < NW) return cut;
    if (c->runs_override > 0) {
        // a forced run count is chained when it can be: a multiple of NW with at least NW frames per group
        const int r = cut.runs;
        if (c->chain_mode != 1 && r % NW == 0 && items / (r / NW) >= NW) cut.chain = 1;
        return cut;
    }
    const long long pairs = 2LL * c->cfg.n_streams;
    const long long simds = (long long)c->n_cus * 4;
    const int cap = c->wgs_per_cu;
    static const double eff[5] = {0.0, 0.55, 0.85, 0.95, 1.0};
    auto cost = [&](int r, double work) {
        const long long waves = pairs * r;
        const long long per_simd = (waves + simds - 1) / simds;
        const long long resident = per_simd < cap ? per_simd : cap;
        return (double)per_simd * work / eff[resident > 4 ? 4 : resident];
    };
    // Whole workgroups of one (stream, channel) - run counts that are multiples of NW - let the kernel put a stream's two channels on
    // the same XCD (xcd_pair): its interleaved PCM then crosses the fabric once instead of twice. Only such cuts are considered.
    const int g_min = (items + 32 * NW - 1) / (32 * NW);   // (a run holds at most ~32 blocks)
    double best_t = 1e300;
    for (int g = g_min; g * NW <= items && g <= 64; ++g) {
        const int r = g * NW;
        if (c->chain_mode != 2) {
            const double t = cost(r, (double)((items + r - 1) / r) + 1.35);
            if (t < best_t * 0.999) {
                best_t = t;
                cut = {r, 0};
            }
        }
        if (c->chain_mode != 1 && items / g >= NW) {
            const int per_group = (items + g - 1) / g + 1;      // the longest group's frames + its priming block
            const double t = cost(r, (double)((per_group + NW - 1) / NW) + 0.65);
            if (t < best_t * 0.999) {
                best_t = t;
                cut = {r, 1};
            }
        }
    }
    return cut;
}

// Dynamic LDS added to k_gain_analysis' launch: it decides how many of its 17 KB workgroups share a CU (nine without). A
// launch of few rounds wants WHOLE rounds - 24 576 workgroups (4096 frames) are 10.67 rounds of 256 x 9 but exactly 16 of
// 256 x 6 - and fewer, fatter slots leave the light stage's and the back half's kernels room beside it: measured on the
// step at 4096 frames 0.333 ms (nine per CU), 0.329 (eight), 0.333 (seven), 0.326 (six), 0.357 (five); at the 1024 x 128
// shard (342 rounds, nothing to round) any padding costs 3 %.
struct LdsChoice {
    int per_cu;
    size_t pad;
};
size_t whole_rounds_pad(const at3hip_ctx* c, long long n_wgs, const LdsChoice* choice, int n_choices, bool ties_to_fewer)
{
    // the per-CU counts of the choices hold for a 160 KB LDS (gfx950): on any other geometry the launch is left alone
    if (c->lds_per_cu != 160u * 1024u) return 0;
    const long long cus = c->n_cus > 0 ? c->n_cus : 256;
    if (n_wgs >= 32 * choice[0].per_cu * cus) return 0;
    double best_waste = 1e30;
    size_t best_pad = 0;
    for (int i = 0; i < n_choices; ++i) {
        const long long slots = cus * choice[i].per_cu;
        const double waste = (double)(((n_wgs + slots - 1) / slots) * slots) / (double)n_wgs;
        if (ties_to_fewer ? waste <= best_waste + 0.005 : waste < best_waste - 0.005) {
            best_waste = waste < best_waste ? waste : best_waste;
            best_pad = choice[i].pad;
        }
    }
    return best_pad;
}
size_t analysis_lds_pad(const at3hip_ctx* c, long long n_wgs)
{
    if (c->lds_per_cu == 160u * 1024u && c->gain_wgs_per_cu > 0 && c->gain_wgs_per_cu < 9) {   // AT3HIP_OPT_GAIN_WGS_PER_CU (tuning aid)
        const size_t each = ((160u * 1024u) / (size_t)c->gain_wgs_per_cu) & ~(size_t)255;
        return each > sizeof(GainLds) + 256 ? each - sizeof(GainLds) - 256 : 0;
    }
    if (c->lds_per_cu == 160u * 1024u && c->gain_wgs_per_cu >= 256) return (size_t)c->gain_wgs_per_cu;   // (values from 256: the pad itself)
    if (c->gain_wgs_per_cu >= 9) return 0;
    // (six per CU at 26 KB each leave 4 KB of the CU's LDS: with the 8704-byte pad of round 3 they left exactly the 10 KB of one
    // k_alloc_pack wavefront, and the step was 0.8 % slower for every CU carrying that lodger - tools/ab_step.sh, --gain-wgs)
    // Since k_mdct_sub's workgroups are three wavefronts (20.3 KB) the fat slot is 22.5 KB, SEVEN per CU: what counts is that the slot ONE retiring
    // analysis workgroup frees takes a workgroup of the light stage - with mdct at 25.5 KB that needed the 26 KB slots of six per CU (a 2 KB larger
    // mdct block cost the step 5 %: AT3HIP_PAD_MDCT, EXPERIMENTS round 5), now seven fit the same launches (+1.2 % on the step, `tones` +2 %;
    // 23.3 KB slots, which leave under 1 KB of the CU, lose 7 %). The choice is still scored as the six-per-CU one it replaces.
    static const LdsChoice kChoice[3] = {{9, 0}, {8, 3328}, {6, 5632}};
    return whole_rounds_pad(c, n_wgs, kChoice, 3, true);   // (equally whole rounds: the fewer, fatter slots measured better)
}
// The one-wavefront form (k_gain_analysis1, 9.5 KB per workgroup): sixteen per CU without padding.
size_t analysis1_lds_pad(const at3hip_ctx* c, long long n_wgs)
{
    if (c->lds_per_cu != 160u * 1024u) return 0;
    if (c->gain_wgs_per_cu > 0 && c->gain_wgs_per_cu < 16) {
        const size_t each = ((160u * 1024u) / (size_t)c->gain_wgs_per_cu) & ~(size_t)255;
        return each > 9728 ? each - 9728 : 0;
    }
    static const LdsChoice kChoice[3] = {{16, 0}, {12, 3584}, {8, 10240}};
    return whole_rounds_pad(c, n_wgs, kChoice, 3, true);
}
// The same for k_gain_spec (9.5 KB per one-wavefront workgroup, sixteen per CU without padding): the 6144 workgroups of
// 4096 frames are one and a half rounds of 256 x 16 and exactly two of 256 x 12 (+0.8 % on the step).
size_t spec_lds_pad(const at3hip_ctx* c, long long n_wgs)
{
    static const LdsChoice kChoice[2] = {{16, 0}, {12, 3584}};
    return whole_rounds_pad(c, n_wgs, kChoice, 2, false);   // (only when it removes a partial round: 16384 frames are six whole rounds as they are)
}

int reset_state(at3hip_ctx* c)
{
    const size_t S = c->cfg.n_streams;
    HIPCHK(c, hipMemsetAsync(c->d_hist[0], 0, S * kHist * 2 * sizeof(float), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_hist[1], 0, S * kHist * 2 * sizeof(float), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_state, 0, S * 8 * sizeof(BandState), c->stream));
    if (c->d_sub_tail) HIPCHK(c, hipMemsetAsync(c->d_sub_tail, 0, S * 8 * 512 * sizeof(float), c->stream));   // silence before the stream
    HIPCHK(c, hipMemsetAsync(c->d_counters, 0, 2 * sizeof(unsigned long long), c->stream));
    float* init = (float*)malloc(S * sizeof(float));
    if (!init) return fail(c, AT3HIP_ENOMEM, "malloc");
    for (size_t i = 0; i < S; ++i) init[i] = 0.006f;  // LoudFactor, atrac3denc.h:115-116
    hipError_t e = hipMemcpyAsync(c->d_loud_state, init, S * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    free(init);
    if (e != hipSuccess) return fail(c, AT3HIP_EDEVICE, "state upload", e);
    c->blocks_fed = 0;
    c->hist_cur = 0;
    return AT3HIP_OK;
}

}  // namespace

extern "C" {

uint32_t at3hip_version(void) { return (uint32_t)AT3HIP_VERSION; }

int at3hip_create(const at3hip_config* cfg, at3hip_ctx** out)
{
    if (!cfg || !out) return AT3HIP_EINVAL;
    *out = nullptr;
    if ((cfg->channels != 1 && cfg->channels != 2) || cfg->n_streams < 1 || cfg->max_blocks < 1 || cfg->bfu_idx_const < 0 ||
        cfg->bfu_idx_const > 32)
        return AT3HIP_EINVAL;
    // the largest grid is one workgroup per (stream, frame, channel, band < 3); gridDim.x is a 31-bit quantity
    if ((long long)cfg->n_streams * cfg->max_blocks * 6 > 0x7fffffffLL) return AT3HIP_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return AT3HIP_EDEVICE;
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return AT3HIP_EINVAL;
    at3hip_ctx* c = new (std::nothrow) at3hip_ctx();
    if (!c) return AT3HIP_ENOMEM;
    c->cfg = *cfg;
    c->device = cfg->device_id;
    const uint32_t br = cfg->bitrate == 0 ? 132300u : (uint32_t)cfg->bitrate;
    int idx = 0;
    while (idx < 7 && kContainer[idx].bitrate < br) ++idx;  // lower_bound, atrac3.cpp:47-53
    c->frame_sz = kContainer[idx].frame_sz;
    c->js = kContainer[idx].js;

    int rc = AT3HIP_OK;
    auto bail = [&](int code) {
        at3hip_destroy(c);
        return code;
    };
    at3host::DeviceGuard guard(c->device);
    if (guard.error() != hipSuccess) return bail(AT3HIP_EDEVICE);
#ifdef AT3HIP_DEBUG_KNOBS
    if (const char* e = getenv("AT3HIP_DEBUG_FRONT")) c->dbg_front = atoi(e);
    {
        static const char* const kPadEnv[8] = {"AT3HIP_PAD_QMF", "AT3HIP_PAD_MDCT", "AT3HIP_PAD_CURVE", "AT3HIP_PAD_GES", "AT3HIP_PAD_TAIL", "AT3HIP_PAD_LOUD", "AT3HIP_PAD_PSY", "AT3HIP_PAD_SCAN"};
        for (int i = 0; i < 8; ++i)
            if (const char* e = getenv(kPadEnv[i])) c->dbg_pad[i] = atoi(e);
    }
    if (const char* e = getenv("AT3HIP_DEBUG_GAIN")) c->dbg_gain = atoi(e);
    if (const char* e = getenv("AT3HIP_DEBUG_STOP")) c->dbg_stop = atoi(e);
    if (const char* e = getenv("AT3HIP_ALLOC_PAD")) c->alloc_lds_pad = atoi(e);
#endif
    {
        // The back half's stream gets the higher priority: its rate loop is the longest kernel of a call and fills every CU's
        // LDS, so a front-half workgroup placed between two of its rounds only delays it, while the front-half kernels of the
        // NEXT call have a whole rate loop's time to spare (measured with the three streams: back high / front low beats
        // the opposite by 0.7 % on white noise, 3-4 % on `tones` and LP4, 6.5 % on `burst`; equal priorities sit between).
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);   // lo = numerically greatest = lowest priority
        if (hipStreamCreateWithPriority(&c->own_stream, hipStreamNonBlocking, prio_lo) != hipSuccess) return bail(AT3HIP_EDEVICE);
        if (hipStreamCreateWithPriority(&c->back_stream, hipStreamNonBlocking, prio_hi) != hipSuccess) return bail(AT3HIP_EDEVICE);
        // The light front stage (curve context, curves, energy scales, MDCT) sits on the step's critical path - the next rate
        // loop waits for its spectra - while the heavy stage it shares the chip with (the NEXT step's spectra and envelopes,
        // whose workgroups fill the LDS) has a whole rate loop of slack: high priority lets its short kernels take the
        // slots the heavy stage's workgroups free (+3 % on the step; a middle priority does nothing).
        if (!cfg->no_gain_control && hipStreamCreateWithPriority(&c->mid_stream, hipStreamNonBlocking, prio_hi) != hipSuccess) return bail(AT3HIP_EDEVICE);
    }
    // (Round 6 tried creating the copy stream only at the first call that copies from host memory - one claim less on the runtime's few hardware queues for
    // contexts fed from device memory. Created that late it shares a hardware queue with one of the kernel streams and the host-fed pipeline halves:
    // 12.4 -> 6.3 M frames/s with 16-bit samples. It is created here, right behind the kernel streams.)
    if (hipStreamCreateWithFlags(&c->h2d_stream, hipStreamNonBlocking) != hipSuccess) return bail(AT3HIP_EDEVICE);
    for (int q = 0; q < 2; ++q)
        if (hipEventCreateWithFlags(&c->ev_h2d[q], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_pcm_free[q], hipEventDisableTiming) != hipSuccess)
            return bail(AT3HIP_EDEVICE);
    c->stream = c->own_stream;
    for (auto& row : c->ev)
        for (auto& e : row)
            if (hipEventCreate(&e) != hipSuccess) return bail(AT3HIP_EDEVICE);
    if (hipEventCreateWithFlags(&c->ev_front_done, hipEventDisableTiming) != hipSuccess) return bail(AT3HIP_EDEVICE);
    for (int q = 0; q < at3hip_ctx::kSlots; ++q)
        if (hipEventCreateWithFlags(&c->ev_heavy_done[q], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_mdct_done[q], hipEventDisableTiming) != hipSuccess)
            return bail(AT3HIP_EDEVICE);
    for (int q = 0; q < at3hip_ctx::kHostRing; ++q)
        if (hipEventCreateWithFlags(&c->ev_host_in[q], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_host_out[q], hipEventDisableTiming) != hipSuccess)
            return bail(AT3HIP_EDEVICE);
    for (auto& e : c->ev_mid_done)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return bail(AT3HIP_EDEVICE);

    const size_t S = cfg->n_streams, B = cfg->max_blocks;
    Tables* host_tables = new (std::nothrow) Tables();
    if (!host_tables) return bail(AT3HIP_ENOMEM);
    build_tables(host_tables);
    rc = dev_alloc(c, &c->d_tables, 1);
    if (rc == AT3HIP_OK) {
        // From pageable memory a blocking copy may return once the data is STAGED: the transfer itself then still runs on the null stream, which the
        // context's non-blocking streams do not wait for. Seen with eight processes on one device: the first call of a fresh process summed its
        // loudness with a curve that had not arrived yet (TrackLoudness state in the millions, silent frames until the stream was reset) - the
        // device is drained here, once, before anything can read the tables.
        hipError_t e = hipMemcpy(c->d_tables, host_tables, sizeof(Tables), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) rc = AT3HIP_EDEVICE;
    }
    delete host_tables;
    if (rc != AT3HIP_OK) return bail(rc);

    if (cfg->channels == 1 && (rc = dev_alloc(c, &c->d_pcm_in, S * B * 2048)) != AT3HIP_OK) return bail(rc);
    // (the host-PCM staging pair is allocated by the first call that hands over host memory)
    if ((rc = dev_alloc(c, &c->d_hist[0], S * kHist * 2)) != AT3HIP_OK) return bail(rc);
    if ((rc = dev_alloc(c, &c->d_hist[1], S * kHist * 2)) != AT3HIP_OK) return bail(rc);
    // subbands go through HBM when gain control analyses them, and for joint stereo (the M/S matrixing needs both channels'
    // subbands, which the one-channel wavefronts of the fused kernel do not have)
    if (!cfg->no_gain_control || c->js) {
        if ((rc = dev_alloc(c, &c->d_sub, S * 8 * (B + 2) * 256)) != AT3HIP_OK) return bail(rc);
        if ((rc = dev_alloc(c, &c->d_sub_tail, S * 8 * 512)) != AT3HIP_OK) return bail(rc);
    }
    if (!cfg->no_gain_control) {
        if ((rc = dev_alloc(c, &c->d_rec, S * B * 6)) != AT3HIP_OK) return bail(rc);
        c->d_rec_b[0] = c->d_rec;
        c->d_sub_b[0] = c->d_sub;
        if ((rc = dev_alloc(c, &c->d_rec_b[1], S * B * 6)) != AT3HIP_OK) return bail(rc);
        if ((rc = dev_alloc(c, &c->d_sub_b[1], S * 8 * (B + 2) * 256)) != AT3HIP_OK) return bail(rc);
        if ((rc = dev_alloc(c, &c->d_bins, S * B * 6 * kGainBins)) != AT3HIP_OK) return bail(rc);
        for (int q = 0; q < 2; ++q)
            if ((rc = dev_alloc(c, &c->d_ges[q], S * B * 8)) != AT3HIP_OK) return bail(rc);
    }
    if ((rc = dev_alloc(c, &c->d_state, S * 8)) != AT3HIP_OK) return bail(rc);
    for (int q = 0; q < 2; ++q) {
        if ((rc = dev_alloc(c, &c->d_curves[q], S * B * 8)) != AT3HIP_OK) return bail(rc);
        if ((rc = dev_alloc(c, &c->d_specs[q], S * B * 2048)) != AT3HIP_OK) return bail(rc);
    }
    if ((rc = dev_alloc(c, &c->d_psy, S * B * 2)) != AT3HIP_OK) return bail(rc);
    if ((rc = dev_alloc(c, &c->d_loud, S * B)) != AT3HIP_OK) return bail(rc);
    if ((rc = dev_alloc(c, &c->d_loud_state, S)) != AT3HIP_OK) return bail(rc);
    if ((rc = dev_alloc(c, &c->d_out, S * B * (size_t)c->frame_sz)) != AT3HIP_OK) return bail(rc);
    if ((rc = dev_alloc(c, &c->d_clk, kClkWords)) != AT3HIP_OK) return bail(rc);
    if (hipMemsetAsync(c->d_clk, 0, (kClkWords) * sizeof(unsigned long long), c->stream) != hipSuccess) return bail(AT3HIP_EDEVICE);   // (reset_state below waits for the stream)
    if ((rc = dev_alloc(c, &c->d_counters, 2)) != AT3HIP_OK) return bail(rc);   // (zeroed by reset_state)
    if ((rc = reset_state(c)) != AT3HIP_OK) return bail(rc);
    hipDeviceProp_t prop;
    const bool have_prop = hipGetDeviceProperties(&prop, c->device) == hipSuccess;
    c->n_cus = (have_prop && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    c->lds_per_cu = have_prop ? (size_t)prop.maxSharedMemoryPerMultiProcessor : 0;
    {
        int nb = 0;
        const bool gain = !c->cfg.no_gain_control;
        const hipError_t e = (gain || c->js) ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_qmf_sub8, 256, 0)
                                             : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_qmf_mdct8, 64 * kFusedWaves, 0);
        // (pick_runs' unit is resident wavefronts per SIMD: a four-wavefront workgroup is one per SIMD)
        c->wgs_per_cu = (e == hipSuccess && nb > 0) ? ((gain || c->js) ? nb : nb * kFusedWaves / 4) : 3;
        nb = 0;
        c->wgs_per_cu_mdct = ((c->js ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (k_mdct_sub<true, 4>), 256, 0)
                                     : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (k_mdct_sub<false, 4>), 256, 0)) == hipSuccess && nb > 0) ? nb : 3;   // (wavefronts per SIMD, pick_runs' unit: the same three for the three-wavefront form, four workgroups per CU)
    }
    *out = c;
    return AT3HIP_OK;
}

void at3hip_destroy(at3hip_ctx* c)
{
    if (!c) return;
    at3host::DeviceGuard guard(c->device);
    // only the context's own streams are waited for: a caller-owned stream (at3hip_set_stream) may already be gone. What
    // the last call queued there (the carried-state update writes buffers freed below) is covered by an event, which
    // outlives the stream
    if (c->front_done_valid) (void)hipEventSynchronize(c->ev_front_done);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    if (c->mid_stream) (void)hipStreamSynchronize(c->mid_stream);
    if (c->back_stream) (void)hipStreamSynchronize(c->back_stream);
    if (c->d_rec_b[1]) (void)hipFree(c->d_rec_b[1]);
    if (c->d_sub_b[1]) (void)hipFree(c->d_sub_b[1]);
    void* bufs[] = {c->d_tables,    c->d_pcm_in,    c->d_hist[0],  c->d_hist[1],  c->d_sub,    c->d_rec,    c->d_state, c->d_curves[0],
                    c->d_curves[1], c->d_specs[0],  c->d_specs[1], c->d_ges[0],   c->d_ges[1], c->d_psy,    c->d_loud,  c->d_loud_state,
                    c->d_out,       c->d_quant,     c->d_stage,    c->d_sub_tail, c->d_bins,     c->d_clk,      c->d_counters};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    for (auto& row : c->ev)
        for (auto& e : row)
            if (e) (void)hipEventDestroy(e);
    if (c->h2d_stream) (void)hipStreamSynchronize(c->h2d_stream);
    for (int q = 0; q < 2; ++q) {
        if (c->d_pcm_in_b[q]) (void)hipFree(c->d_pcm_in_b[q]);
        if (c->d_s16_b[q]) (void)hipFree(c->d_s16_b[q]);
        if (c->ev_h2d[q]) (void)hipEventDestroy(c->ev_h2d[q]);
        if (c->ev_pcm_free[q]) (void)hipEventDestroy(c->ev_pcm_free[q]);
    }
    if (c->h2d_stream) (void)hipStreamDestroy(c->h2d_stream);
    if (c->ev_front_done) (void)hipEventDestroy(c->ev_front_done);
    for (auto& e : c->ev_heavy_done)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ev_mdct_done)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ev_host_in)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ev_host_out)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ev_mid_done)
        if (e) (void)hipEventDestroy(e);
    if (c->mid_stream) (void)hipStreamDestroy(c->mid_stream);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->back_stream) (void)hipStreamDestroy(c->back_stream);
    delete c;
}

int at3hip_frame_size(const at3hip_ctx* c) { return c ? c->frame_sz : AT3HIP_EINVAL; }
int at3hip_joint_stereo(const at3hip_ctx* c) { return c ? c->js : AT3HIP_EINVAL; }
const char* at3hip_last_error(const at3hip_ctx* c) { return c ? c->err : "null context"; }

int at3hip_set_stream(at3hip_ctx* c, void* hip_stream)
{
    if (!c) return AT3HIP_EINVAL;
    at3host::DeviceGuard guard(c->device);
    HIPCHK(c, guard.error());
    const int rc = drain(c);
    if (rc != AT3HIP_OK) return rc;
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return AT3HIP_OK;
}

int at3hip_host_alloc(at3hip_ctx* c, size_t bytes, void** out)
{
    if (!c || !out || bytes == 0) return AT3HIP_EINVAL;
    *out = nullptr;
    at3host::DeviceGuard guard(c->device);
    HIPCHK(c, guard.error());
    void* p = nullptr;
    const hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e != hipSuccess) return fail(c, AT3HIP_ENOMEM, "hipHostMalloc", e);
    *out = p;
    return AT3HIP_OK;
}

int at3hip_device_numa_node(int32_t device_id)
{
    char bus[32] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device_id) != hipSuccess || !bus[0]) return -1;
    for (char* q = bus; *q; ++q)
        if (*q >= 'A' && *q <= 'F') *q = (char)(*q - 'A' + 'a');   // sysfs spells the address in lower case
    char path[96];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE* f = fopen(path, "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}

int at3hip_host_free(at3hip_ctx* c, void* p)
{
    if (!c) return AT3HIP_EINVAL;
    if (!p) return AT3HIP_OK;
    at3host::DeviceGuard guard(c->device);
    HIPCHK(c, guard.error());
    HIPCHK(c, hipHostFree(p));
    return AT3HIP_OK;
}

int at3hip_wait_input(at3hip_ctx* c, int32_t ago)
{
    if (!c || ago < 0 || ago >= at3hip_ctx::kHostRing || ago >= c->enc_calls) return AT3HIP_EINVAL;
    at3host::DeviceGuard guard(c->device);
    HIPCHK(c, guard.error());
    const int q = (int)((c->enc_calls - 1 - ago) % at3hip_ctx::kHostRing);
    if (c->host_in_valid[q]) HIPCHK(c, hipEventSynchronize(c->ev_host_in[q]));
    return AT3HIP_OK;
}

int at3hip_wait_frames(at3hip_ctx* c, int32_t ago)
{
    if (!c || ago < 0 || ago >= at3hip_ctx::kHostRing || ago >= c->enc_calls) return AT3HIP_EINVAL;
    at3host::DeviceGuard guard(c->device);
    HIPCHK(c, guard.error());
    const int q = (int)((c->enc_calls - 1 - ago) % at3hip_ctx::kHostRing);
    if (c->host_out_valid[q]) HIPCHK(c, hipEventSynchronize(c->ev_host_out[q]));
    return AT3HIP_OK;
}

int at3hip_host_tables(void* dst, size_t bytes)
{
    if (!dst || bytes != sizeof(Tables)) return AT3HIP_EINVAL;
    build_tables((Tables*)dst);
    return AT3HIP_OK;
}

int at3hip_set_option(at3hip_ctx* c, int32_t option, int32_t value)
{
    if (!c) return AT3HIP_EINVAL;
    switch (option) {
        case AT3HIP_OPT_RUNS:
            if (value < 0) return fail(c, AT3HIP_EINVAL, "runs must be >= 0");
            c->runs_override = value;
            return AT3HIP_OK;
        case AT3HIP_OPT_CHAIN:
            if (value < 0 || value > 2) return fail(c, AT3HIP_EINVAL, "chain: 0 (per call), 1 (never) or 2 (whenever possible)");
            c->chain_mode = value;
            return AT3HIP_OK;
        case AT3HIP_OPT_TIMING_EVERY:
            if (value < 0) return fail(c, AT3HIP_EINVAL, "timing every: 0 (never) or N >= 1 (every Nth call with frames)");
            c->timing_every = value;
            c->timing_tick = 0;
            return AT3HIP_OK;
        case AT3HIP_OPT_LITERAL_FORMS:
            if (value != 0 && value != 1) return fail(c, AT3HIP_EINVAL, "literal forms: 0 or 1");
            c->flat_literal = value;
            return AT3HIP_OK;
        case AT3HIP_OPT_GAIN_FORM:
            if (value == 2) value = AT3HIP_GAIN_FORM_ONE_WAVE;   // (ABI 1.2's number for the one-wavefront form)
            if (value != AT3HIP_GAIN_FORM_TWO_WAVES && value != AT3HIP_GAIN_FORM_ONE_WAVE)
                return fail(c, AT3HIP_EINVAL, "gain form: AT3HIP_GAIN_FORM_TWO_WAVES or AT3HIP_GAIN_FORM_ONE_WAVE");
            c->gain_form = value;
            return AT3HIP_OK;
        case AT3HIP_OPT_GAIN_WGS_PER_CU:
            if (value < 0 || (value > 16 && value < 256) || value > 64 * 1024) return fail(c, AT3HIP_EINVAL, "workgroups per CU must be 0 .. 16 (or a pad of 256 .. 65536 bytes)");
            c->gain_wgs_per_cu = value;
            return AT3HIP_OK;
        case AT3HIP_OPT_QUANT_TAP: {
            if (value != 0 && value != 1) return fail(c, AT3HIP_EINVAL, "quant tap: 0 or 1");
            at3host::DeviceGuard guard(c->device);
            HIPCHK(c, guard.error());
            const int rc = drain(c);
            if (rc != AT3HIP_OK) return rc;
            if (value && !c->d_quant) return dev_alloc(c, &c->d_quant, (size_t)c->cfg.n_streams * c->cfg.max_blocks * 2);
            if (!value && c->d_quant) {
                (void)hipFree(c->d_quant);
                c->d_quant = nullptr;
            }
            return AT3HIP_OK;
        }
        default: return fail(c, AT3HIP_EINVAL, "unknown option");
    }
}

int at3hip_reset(at3hip_ctx* c)
{
    if (!c) return AT3HIP_EINVAL;
    at3host::DeviceGuard guard(c->device);
    HIPCHK(c, guard.error());
    const int rc = drain(c);
    if (rc != AT3HIP_OK) return rc;
    return reset_state(c);
}

int at3hip_get_timings(const at3hip_ctx* c, at3hip_timings* out)
{
    if (!c || !out) return AT3HIP_EINVAL;
    *out = c->tm;
    return AT3HIP_OK;
}

namespace {
int encode_impl(at3hip_ctx* c, const void* pcm_any, bool s16, int32_t n_blocks, uint8_t* out_frames, int32_t* n_frames_out, uint32_t flags);
}

int at3hip_encode(at3hip_ctx* c, const float* pcm, int32_t n_blocks, uint8_t* out_frames, int32_t* n_frames_out,
                  uint32_t flags)
{
    return encode_impl(c, pcm, false, n_blocks, out_frames, n_frames_out, flags);
}

int at3hip_encode_s16(at3hip_ctx* c, const int16_t* pcm, int32_t n_blocks, uint8_t* out_frames, int32_t* n_frames_out,
                      uint32_t flags)
{
    return encode_impl(c, pcm, true, n_blocks, out_frames, n_frames_out, flags);
}

}  // extern "C"

namespace {

int encode_impl(at3hip_ctx* c, const void* pcm_any, bool s16, int32_t n_blocks, uint8_t* out_frames, int32_t* n_frames_out, uint32_t flags)
{
    const float* pcm = (const float*)pcm_any;   // (float samples unless s16)
    if (!c || !pcm_any || n_blocks < 1 || n_blocks > c->cfg.max_blocks) return c ? fail(c, AT3HIP_EINVAL, "bad argument") : AT3HIP_EINVAL;
    at3host::DeviceGuard guard(c->device);
    HIPCHK(c, guard.error());
    const int S = c->cfg.n_streams;
    const int f0 = (c->blocks_fed == 0) ? 1 : 0;
    const int n_out = n_blocks - f0;
    if (n_out > 0 && !out_frames) return fail(c, AT3HIP_EINVAL, "out_frames is null");
    // k_s16_to_f32 reads the caller's device pointer sixteen bytes at a time
    if (s16 && (flags & AT3HIP_PCM_ON_DEVICE) && ((uintptr_t)pcm_any & 15u) != 0) return fail(c, AT3HIP_EINVAL, "16-bit device PCM must be 16-byte aligned");
    hipStream_t st = c->stream;
    const bool gain = !c->cfg.no_gain_control;

    const int par = (int)(c->enc_calls & 1);
    const float* d_pcm = pcm;
    const size_t n_in = (size_t)S * n_blocks * 1024 * c->cfg.channels;
    const float* d_staged = pcm;   // the call's PCM in device memory, [S][n_blocks][1024][channels]
    const int hq = (int)(c->enc_calls % at3hip_ctx::kHostRing);   // this call's slot in the host-visible event ring
    bool host_in_recorded = false, host_out_recorded = false;
    const bool staged = s16 || !(flags & AT3HIP_PCM_ON_DEVICE);   // the call's float samples live in this parity's staging buffer
    if (staged && !c->d_pcm_in_b[par]) {
        const int rc = dev_alloc(c, &c->d_pcm_in_b[par], (size_t)S * c->cfg.max_blocks * 1024 * c->cfg.channels);
        if (rc != AT3HIP_OK) return rc;
    }
    if (!(flags & AT3HIP_PCM_ON_DEVICE)) {
        // host memory: the copy runs on its own stream into this parity's staging buffer - beside the previous call's
        // kernels when the memory is pinned - and the front half waits for its arrival. 16-bit samples cross the bus as they
        // are (half the bytes: the host-fed rate is bound by them) and become floats on the device, on the same stream.
        if (s16 && !c->d_s16_b[par]) {
            const int rc = dev_alloc(c, &c->d_s16_b[par], (size_t)S * c->cfg.max_blocks * 1024 * c->cfg.channels);
            if (rc != AT3HIP_OK) return rc;
        }
        if (c->pcm_free_valid[par]) HIPCHK(c, hipStreamWaitEvent(c->h2d_stream, c->ev_pcm_free[par], 0));
        if (s16) HIPCHK(c, hipMemcpyAsync(c->d_s16_b[par], pcm_any, n_in * sizeof(int16_t), hipMemcpyHostToDevice, c->h2d_stream));
        else HIPCHK(c, hipMemcpyAsync(c->d_pcm_in_b[par], pcm, n_in * sizeof(float), hipMemcpyHostToDevice, c->h2d_stream));
        HIPCHK(c, hipEventRecord(c->ev_h2d[par], c->h2d_stream));
        c->h2d_valid[par] = true;
        HIPCHK(c, hipEventRecord(c->ev_host_in[hq], c->h2d_stream));
        host_in_recorded = true;
        HIPCHK(c, hipStreamWaitEvent(st, c->ev_h2d[par], 0));
        // (the conversion runs on the front stream, not behind the copy: the copy stream carries nothing but copies, back to back)
        if (s16) hipLaunchKernelGGL(k_s16_to_f32, dim3((unsigned)((n_in / 8 + 255) / 256)), dim3(256), 0, st, c->d_s16_b[par], c->d_pcm_in_b[par], n_in / 8);
        d_staged = c->d_pcm_in_b[par];
    } else if (s16) {
        // 16-bit samples already in HBM: converted on the front stream (which also carries everything that reads the staging)
        hipLaunchKernelGGL(k_s16_to_f32, dim3((unsigned)((n_in / 8 + 255) / 256)), dim3(256), 0, st, (const int16_t*)pcm_any, c->d_pcm_in_b[par], n_in / 8);
        d_staged = c->d_pcm_in_b[par];
    }
    d_pcm = d_staged;
    if (c->cfg.channels == 1) {
        // "No mono mode for atrac3, just make duplicate of first channel" (atrac3_bitstream.cpp:836-843): the one-channel
        // frame is two identical sound units, i.e. the discrete-stereo frame of L = R (TrackLoudness' 0.02 l equals
        // 0.01 (l + l) exactly). The samples are duplicated in HBM and the stereo pipeline runs unchanged.
        // Joint-stereo containers: M = (x + x) / 2 = x exactly, so the M unit is the mono unit (gain analysis, loudness
        // with 0.02 l and all); the rate/pack kernel replaces the S unit by the empty element of atrac3denc.cpp:843-849.
        const size_t n = (size_t)S * n_blocks * 1024;
        hipLaunchKernelGGL(k_mono_to_pairs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_staged, c->d_pcm_in, n);
        d_pcm = c->d_pcm_in;
    }
    uint8_t* d_out = (flags & AT3HIP_OUT_ON_DEVICE) ? out_frames : c->d_out;
    const float* hist = c->d_hist[c->hist_cur];
    float* hist_next = c->d_hist[c->hist_cur ^ 1];
    hipStream_t bk = c->back_stream;
    const int slot = (int)(c->enc_calls % at3hip_ctx::kSlots);
    hipEvent_t* ev = c->ev[slot];
    Curve* d_curves = c->d_curves[par];
    float* d_specs = c->d_specs[par];
    float* d_ges = c->d_ges[par];
    c->slot_has_frames[slot] = false;
    // The timing events sit between the kernels of the three streams; at 4096 frames a step recording all eight takes 3.3 % longer than one recording none
    // (EXPERIMENTS.md, round 6), so a caller that only samples the stage timings asks for every Nth call.
    const bool timed = n_out > 0 && c->timing_every > 0 && (c->timing_tick++ % (unsigned)c->timing_every) == 0;
    c->slot_timed[slot] = timed;
    auto launch_state = [&](hipStream_t on, int parts) {
        StateParams sp;
        sp.pcm = d_pcm;
        sp.hist_in = hist;
        sp.hist_out = hist_next;
        sp.curves = c->d_curves[par];
        sp.state = c->d_state;
        sp.sub = (gain || c->js) ? (gain ? c->d_sub_b[par] : c->d_sub) : nullptr;
        sp.sub_tail = c->d_sub_tail;
        sp.n_blocks = n_blocks;
        sp.n_streams = S;
        sp.parts = parts;
        hipLaunchKernelGGL(k_state_update, dim3((unsigned)(((kHist + 255) / 256) * S)), dim3(256), 0, on, sp);
    };

    // With gain control: `st` carries the heavy stage, `md` the light one (see at3hip_ctx::mid_stream); otherwise md == st.
    hipStream_t md = gain ? c->mid_stream : st;
    float* d_sub = gain ? c->d_sub_b[par] : c->d_sub;
    GainRec* d_rec = gain ? c->d_rec_b[par] : c->d_rec;
    // the light stage of the call before the previous one must be done with this parity's subbands and gain records
    if (gain && c->mid_done_valid[par]) HIPCHK(c, hipStreamWaitEvent(st, c->mid_done_of[par], 0));
    if (timed) HIPCHK(c, hipEventRecord(ev[0], st));
    // the back half of the call before the previous one must be done with this parity's spectra / curves / scales
    if (c->back_done_valid[par]) HIPCHK(c, hipStreamWaitEvent(md, c->ev_back_done[par], 0));
    HIPCHK(c, hipMemsetAsync(d_curves, 0, (size_t)S * n_blocks * 8 * sizeof(Curve), md));
    if (n_out == 0 && (gain || c->js)) {
        // a call that only primes the look-ahead still has to leave its subbands behind for the next call's look-back
        FrontParams fp = {};
        fp.pcm = d_pcm;
        fp.hist = hist;
        fp.sub = d_sub;
        fp.sub_tail = c->d_sub_tail;
        fp.n_blocks = n_blocks;
        fp.f0 = f0;
        fp.js = c->js;
        fp.sub_runs = pick_runs(c, n_blocks, c->wgs_per_cu, 0.35, 4);
        const int n_waves = S * 2 * fp.sub_runs;
        hipLaunchKernelGGL(k_qmf_sub8, dim3((unsigned)((n_waves + 3) / 4)), dim3(256), (size_t)c->dbg_pad[0], st, fp, c->d_tables, n_waves);
    }
    if (n_out > 0) {
        FrontParams fp;
        fp.pcm = d_pcm;
        fp.hist = hist;
        fp.curves = d_curves;
        fp.state = c->d_state;
        fp.specs = d_specs;
        fp.ges = d_ges;
        fp.sub = d_sub;
        fp.sub_tail = c->d_sub_tail;
        fp.n_blocks = n_blocks;
        fp.f0 = f0;
        const bool split = gain || c->js;   // QMF and MDCT as two kernels with the subbands in HBM between them
        {
            const FusedCut cut = split ? FusedCut{0, 0} : pick_fused(c, n_out);
            fp.frame_runs = cut.runs;
            fp.chain = cut.chain;
        }
        fp.sub_runs = 0;
        fp.debug = c->dbg_front;
        fp.clk = nullptr;
        fp.js = c->js;
        auto launch_qmf_sub = [&] {
            fp.sub_runs = pick_runs(c, n_blocks, c->wgs_per_cu, 0.35, 4);
            const int n_waves = S * 2 * fp.sub_runs;   // one wavefront per (stream, channel, run)
            hipLaunchKernelGGL(k_qmf_sub8, dim3((unsigned)((n_waves + 3) / 4)), dim3(256), (size_t)c->dbg_pad[0], st, fp, c->d_tables, n_waves);
        };
        if (gain) {
            GainParams gp;
            gp.sub = d_sub;
            gp.rec = d_rec;
            gp.bins = c->d_bins;
            gp.state = c->d_state;
            gp.curves = d_curves;
            gp.n_blocks = n_blocks;
            gp.f0 = f0;
            gp.js = c->js;
            gp.n_streams = S;
            gp.debug = c->dbg_gain;
            gp.literal = c->flat_literal;
            gp.clk = c->d_clk + 16 + 256 * 12;
            launch_qmf_sub();
            if (timed) HIPCHK(c, hipEventRecord(ev[1], st));
            // PCM history and subband tail: the next call's heavy stage needs nothing else from this one. (Round 6: moved behind the gain analysis - off the
            // chain QMF -> spectra -> analysis - it cost the step 8 %: the next call's QMF kernel follows it on this stream and then misses the window
            // between the analysis and the next rate loop in which it gets its only undisturbed microseconds. EXPERIMENTS.md.)
            launch_state(st, 1);
            {
                const long long spec_wgs = (long long)S * ((n_out + 3) / 4) * 6;   // (four consecutive frames of one (stream, channel, band) per wavefront)
                hipLaunchKernelGGL(k_gain_spec, dim3((unsigned)spec_wgs), dim3(64), spec_lds_pad(c, spec_wgs), st, gp, c->d_tables, S * n_out * 6);
            }
            // default: the two-wavefront form. The one-wavefront form is 11 % faster alone (89 against 101 us at 4096 frames) and
            // leaves the pipelined step 2 % slower at that size, equal at the 1024 x 128 shard (profiles/EXPERIMENTS.md)
            if (c->gain_form != AT3HIP_GAIN_FORM_ONE_WAVE) hipLaunchKernelGGL(k_gain_analysis, dim3(S * n_out * 6), dim3(128), analysis_lds_pad(c, (long long)S * n_out * 6), st, gp, c->d_tables);
            else hipLaunchKernelGGL(k_gain_analysis1, dim3(S * n_out * 6), dim3(64), analysis1_lds_pad(c, (long long)S * n_out * 6), st, gp, c->d_tables);   // one wavefront per item
            if (timed) HIPCHK(c, hipEventRecord(ev[2], st));
            HIPCHK(c, hipEventRecord(c->ev_heavy_done[slot], st));
            HIPCHK(c, hipStreamWaitEvent(md, c->ev_heavy_done[slot], 0));   // the light stage starts when this call's heavy stage is done
            hipLaunchKernelGGL(k_gain_tail, dim3((unsigned)((S * n_out * 6 + 7) / 8)), dim3(256), (size_t)c->dbg_pad[4], md, gp, S * n_out * 6);
            hipLaunchKernelGGL(k_gain_scan, dim3(S * 6), dim3(64), (size_t)c->dbg_pad[7], md, gp, S);
            hipLaunchKernelGGL(k_gain_curve, dim3((S * n_out * 6 + 7) / 8), dim3(256), (size_t)c->dbg_pad[2], md, gp, c->d_tables, S);
            hipLaunchKernelGGL(k_gain_energy_scale, dim3(S * n_out * kGesSplit), dim3(64), (size_t)c->dbg_pad[3], md, fp, c->d_tables, S * n_out);
        } else {
            if (split) launch_qmf_sub();   // joint stereo without gain control: the QMF kernel, timed as qmf_ms
            if (timed) HIPCHK(c, hipEventRecord(ev[1], st));
            if (timed) HIPCHK(c, hipEventRecord(ev[2], st));
        }
        if (timed) HIPCHK(c, hipEventRecord(ev[3], md));
        if (split) {
            // the subbands are in HBM (k_qmf_sub8 wrote them: for the gain analysis, or for the M/S matrixing)
            MdctSubParams mp;
            mp.sub = d_sub;
            mp.curves = gain ? d_curves : nullptr;
            mp.state = c->d_state;
            mp.specs = d_specs;
            mp.n_blocks = n_blocks;
            mp.f0 = f0;
            mp.js = c->js;
            mp.frame_runs = pick_runs(c, n_out, c->wgs_per_cu_mdct, 0.3);
            mp.n_waves = S * 2 * mp.frame_runs;
            // three wavefronts per workgroup (20.3 KB) where k_gain_analysis' launch is padded to fat slots - a light-stage workgroup must fit the slot
            // one retiring analysis workgroup frees (analysis_lds_pad) -, four (25.5 KB; one table copy per four runs) where it fills the chip unpadded
            const bool fat_slots = c->gain_form != AT3HIP_GAIN_FORM_ONE_WAVE && analysis_lds_pad(c, (long long)S * n_out * 6) != 0;   // (the one-wavefront form's launch has no fat slots)
            if (fat_slots) {
                if (c->js) hipLaunchKernelGGL((k_mdct_sub<true, 3>), dim3((unsigned)((mp.n_waves + 2) / 3)), dim3(192), 0, md, mp, c->d_tables);
                else hipLaunchKernelGGL((k_mdct_sub<false, 3>), dim3((unsigned)((mp.n_waves + 2) / 3)), dim3(192), (size_t)c->dbg_pad[1], md, mp, c->d_tables);
            } else {
                if (c->js) hipLaunchKernelGGL((k_mdct_sub<true, 4>), dim3((unsigned)((mp.n_waves + 3) / 4)), dim3(256), 0, md, mp, c->d_tables);
                else hipLaunchKernelGGL((k_mdct_sub<false, 4>), dim3((unsigned)((mp.n_waves + 3) / 4)), dim3(256), (size_t)c->dbg_pad[1], md, mp, c->d_tables);
            }
        } else {
            const int n_waves = S * 2 * fp.frame_runs;
            hipLaunchKernelGGL(k_qmf_mdct8, dim3((unsigned)((n_waves + kFusedWaves - 1) / kFusedWaves)), dim3(64 * kFusedWaves), 0, st, fp, c->d_tables, n_waves);
        }
        if (timed) HIPCHK(c, hipEventRecord(ev[4], md));
        HIPCHK(c, hipEventRecord(c->ev_mdct_done[slot], md));
        c->slot_k1_launches[slot] = split ? 2 : 1;
    }
    if (gain) {
        if (n_out == 0) launch_state(st, 1);   // (with frames it followed the QMF kernel)
        launch_state(md, 2);                     // last curves: the light stage's own hand-over
        // with frames the MDCT kernel was the last reader of the parity's subbands and gain records (the update above touches the curves and
        // the band states, which only this stream reads and writes): its event serves, one record less between the kernels
        if (n_out > 0) c->mid_done_of[par] = c->ev_mdct_done[slot];
        else {
            HIPCHK(c, hipEventRecord(c->ev_mid_done[par], md));
            c->mid_done_of[par] = c->ev_mid_done[par];
        }
        c->mid_done_valid[par] = true;
    } else {
        launch_state(st, 3);
    }
    if (n_out > 0) {
        // ---- back half, on its own stream, after the fused kernel of THIS call ----
        // (Round 6, each an 8 - 11 % LOSS on the step although it shortens a chain: the loudness sums and k_psy queued behind the MDCT on the light stage's stream
        // (no event hop between the streams in front of them); k_gain_energy_scale on this stream beside the MDCT; TrackLoudness inside k_psy (no k_loudness
        // launch); k_loud_sum in 22 KB blocks that fit a slot one retiring analysis workgroup frees. EXPERIMENTS.md: the step's schedule is one of several
        // stable ones, tools/event_timeline.py shows which.)
        HIPCHK(c, hipStreamWaitEvent(bk, c->ev_mdct_done[slot], 0));
        if (timed) HIPCHK(c, hipEventRecord(ev[5], bk));
        BackParams bp;
        bp.specs = d_specs;
        bp.ges = gain ? d_ges : nullptr;
        bp.curves = d_curves;
        bp.psy = c->d_psy;
        bp.loud = c->d_loud;
        bp.loud_state = c->d_loud_state;
        bp.out = d_out;
        bp.n_blocks = n_blocks;
        bp.f0 = f0;
        bp.n_streams = S;
        bp.no_tonal = c->cfg.no_tonal;
        bp.js = c->js;
        bp.frame_sz = c->frame_sz;
        bp.bfu_idx_const = c->cfg.bfu_idx_const;
        bp.mono_js = (c->cfg.channels == 1 && c->js) ? 1 : 0;
        bp.flat_literal = c->flat_literal;
        bp.debug_stop = c->dbg_stop;
        bp.quant = c->d_quant;
        bp.clk = c->d_clk;
        bp.counters = c->d_counters;
        bp.one_channel = c->cfg.channels == 1 ? 1 : 0;
        hipLaunchKernelGGL(k_loud_sum, dim3((unsigned)((S * n_out * 2 + kLoudCf - 1) / kLoudCf)), dim3(256), (size_t)c->dbg_pad[5], bk, bp, c->d_tables, S * n_out * 2);
        hipLaunchKernelGGL(k_psy, dim3((S * n_out * 2 + kPsyCf - 1) / kPsyCf), dim3(256), (size_t)c->dbg_pad[6], bk, bp, c->d_tables, S * n_out * 2);
        if (timed) HIPCHK(c, hipEventRecord(ev[6], bk));
        hipLaunchKernelGGL(k_loudness, dim3(S), dim3(64), 0, bk, bp);
        hipLaunchKernelGGL(k_alloc_pack, dim3(S * n_out * 2), dim3(64), (size_t)c->alloc_lds_pad, bk, bp, c->d_tables);
        if (timed) HIPCHK(c, hipEventRecord(ev[7], bk));
        if (!(flags & AT3HIP_OUT_ON_DEVICE))
            HIPCHK(c, hipMemcpyAsync(out_frames, c->d_out, (size_t)S * n_out * c->frame_sz, hipMemcpyDeviceToHost, bk));
        // one event per call on this stream: the frames are out (at3hip_wait_frames) and the back half is done with the parity's buffers
        HIPCHK(c, hipEventRecord(c->ev_host_out[hq], bk));
        c->ev_back_done[par] = c->ev_host_out[hq];
        c->back_done_valid[par] = true;
        host_out_recorded = true;
        c->slot_has_frames[slot] = true;
        c->last_slot = slot;
    }
    if (c->stream != c->own_stream) {   // (at3hip_destroy waits for the context's own streams themselves)
        HIPCHK(c, hipEventRecord(c->ev_front_done, st));
        c->front_done_valid = true;
    }
    if (staged) {   // (the PCM is read by the first stage and by the carried-state update, both on `st`)
        HIPCHK(c, hipEventRecord(c->ev_pcm_free[par], st));
        c->pcm_free_valid[par] = true;
    }
    HIPCHK(c, hipGetLastError());
    c->host_in_valid[hq] = host_in_recorded;     // (a call without a host copy / without frames has nothing to wait for)
    c->host_out_valid[hq] = host_out_recorded;
    c->hist_cur ^= 1;
    c->blocks_fed += n_blocks;
    c->enc_calls++;
    if (n_frames_out) *n_frames_out = n_out;
    if (!(flags & AT3HIP_ASYNC)) {
        const int rc = drain(c);
        if (rc != AT3HIP_OK) return rc;
        if (n_out > 0) read_timings(c, slot, &c->tm);
    }
    return AT3HIP_OK;
}

}  // namespace

extern "C" {

int at3hip_sync(at3hip_ctx* c)
{
    if (!c) return AT3HIP_EINVAL;
    at3host::DeviceGuard guard(c->device);
    HIPCHK(c, guard.error());
    const int rc = drain(c);
    if (rc != AT3HIP_OK) return rc;
    read_timings(c, c->last_slot, &c->tm);
    return AT3HIP_OK;
}

int at3hip_read_tap(at3hip_ctx* c, int32_t kind, void* dst, size_t bytes)
{
    if (!c || !dst) return AT3HIP_EINVAL;
    at3host::DeviceGuard guard(c->device);
    HIPCHK(c, guard.error());
    const int rc = drain(c);
    if (rc != AT3HIP_OK) return rc;
    if (c->enc_calls == 0 && kind != AT3HIP_TAP_CLOCK) return fail(c, AT3HIP_EINVAL, "no encode call yet");
    const int par = (int)((c->enc_calls - 1) & 1);
    const size_t S = c->cfg.n_streams, B = c->cfg.max_blocks;
    const void* src = nullptr;
    size_t cap = 0;
    switch (kind) {
        case AT3HIP_TAP_SPECTRA: src = c->d_specs[par]; cap = S * B * 2048 * sizeof(float); break;
        case AT3HIP_TAP_CURVES: src = c->d_curves[par]; cap = S * B * 8 * sizeof(Curve); break;
        case AT3HIP_TAP_ENERGY_SCALE: src = c->d_ges[par]; cap = c->d_ges[par] ? S * B * 8 * sizeof(float) : 0; break;
        case AT3HIP_TAP_PSY: src = c->d_psy; cap = S * B * 2 * sizeof(PsyRec); break;
        case AT3HIP_TAP_LOUDNESS: src = c->d_loud; cap = S * B * sizeof(float); break;
        case AT3HIP_TAP_QUANT: src = c->d_quant; cap = c->d_quant ? S * B * 2 * sizeof(QuantRec) : 0; break;
        case AT3HIP_TAP_GAIN_ANALYSIS: src = c->d_rec_b[par]; cap = c->d_rec_b[par] ? S * B * 6 * sizeof(GainRec) : 0; break;
        case AT3HIP_TAP_CLOCK: src = c->d_clk; cap = (kClkWords) * sizeof(unsigned long long); break;   // (from slot 16 on: 256 rows of per-phase cycles of k_alloc_pack, then of k_gain_analysis1; profiling builds)
        default: return fail(c, AT3HIP_EINVAL, "unknown tap");
    }
    if (!src || bytes > cap) return fail(c, AT3HIP_EINVAL, "tap not available or request too large");
    HIPCHK(c, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return AT3HIP_OK;
}

int at3hip_get_counters(at3hip_ctx* c, at3hip_counters* out, int32_t reset)
{
    if (!c || !out) return AT3HIP_EINVAL;
    at3host::DeviceGuard guard(c->device);
    HIPCHK(c, guard.error());
    const int rc = drain(c);
    if (rc != AT3HIP_OK) return rc;
    unsigned long long v[2] = {0ull, 0ull};
    HIPCHK(c, hipMemcpy(v, c->d_counters, sizeof(v), hipMemcpyDeviceToHost));
    out->scale_overflow = v[0];
    out->clipped_values = v[1];
    if (reset) {
        HIPCHK(c, hipMemsetAsync(c->d_counters, 0, sizeof(v), c->back_stream));   // (the stream whose kernels add to them)
        HIPCHK(c, hipStreamSynchronize(c->back_stream));
    }
    return AT3HIP_OK;
}

#if defined(AT3HIP_DEBUG_KNOBS) || defined(AT3HIP_DEBUG_EVENTS)
// PROFILING BUILDS ONLY (-DAT3HIP_DEBUG_EVENTS: the release kernels + this entry point; not in include/at3hip.h): milliseconds from event `ia` of the call
// `ago_a` calls back to event `ib` of the call `ago_b` calls back (events 0 .. 7 of a call: before QMF, after QMF, after the gain analysis, after the curves /
// energy scales, after the MDCT, back half's start, after k_psy, after the rate loop) - the pipelined step's timeline WITHOUT a tracer (tools/event_timeline.py):
// rocprofv3 serialises what it traces, and this pipeline's schedule does not survive that.
__attribute__((visibility("default"))) int at3hip_debug_event_ms(at3hip_ctx* c, int32_t ago_a, int32_t ia, int32_t ago_b, int32_t ib, float* ms)
{
    if (!c || !ms || ago_a < 0 || ago_b < 0 || ago_a >= at3hip_ctx::kSlots || ago_b >= at3hip_ctx::kSlots || ago_a >= c->enc_calls || ago_b >= c->enc_calls ||
        ia < 0 || ia > 7 || ib < 0 || ib > 7)
        return AT3HIP_EINVAL;
    at3host::DeviceGuard guard(c->device);
    const int rc = drain(c);
    if (rc != AT3HIP_OK) return rc;
    const int sa = (int)((c->enc_calls - 1 - ago_a) % at3hip_ctx::kSlots), sb = (int)((c->enc_calls - 1 - ago_b) % at3hip_ctx::kSlots);
    return hipEventElapsedTime(ms, c->ev[sa][ia], c->ev[sb][ib]) == hipSuccess ? AT3HIP_OK : AT3HIP_EDEVICE;
}
#endif

int at3hip_get_timings_ago(at3hip_ctx* c, int32_t ago, at3hip_timings* out)
{
    if (!c || !out || ago < 0 || ago >= at3hip_ctx::kSlots || ago >= c->enc_calls) return AT3HIP_EINVAL;
    at3host::DeviceGuard guard(c->device);
    HIPCHK(c, guard.error());
    const int rc = drain(c);
    if (rc != AT3HIP_OK) return rc;
    read_timings(c, (int)((c->enc_calls - 1 - ago) % at3hip_ctx::kSlots), out);
    return AT3HIP_OK;
}

namespace {

int mdct_impl(at3hip_ctx* c, float* bands, float* specs, float* max_levels, const int32_t* n_points, const int32_t* level,
              const int32_t* loc, int32_t n_items, uint32_t flags)
{
    if (!c || !bands || !specs || n_items < 1) return c ? fail(c, AT3HIP_EINVAL, "bad argument") : AT3HIP_EINVAL;
    if ((n_points != nullptr) != (level != nullptr) || (n_points != nullptr) != (loc != nullptr))
        return fail(c, AT3HIP_EINVAL, "n_points/level/loc must be all null or all set");
    at3host::DeviceGuard guard(c->device);
    HIPCHK(c, guard.error());
    hipStream_t st = c->stream;
    const bool dev = (flags & AT3HIP_PCM_ON_DEVICE) && (flags & AT3HIP_OUT_ON_DEVICE);
    if (!dev && (flags & (AT3HIP_PCM_ON_DEVICE | AT3HIP_OUT_ON_DEVICE)))
        return fail(c, AT3HIP_EINVAL, "mdct buffers must be all host or all device");
    const size_t n = n_items;
    if (!dev && n_points) {
        for (size_t i = 0; i < n * 4; ++i) {
            if (n_points[i] < 0 || n_points[i] > 7) return fail(c, AT3HIP_EINVAL, "n_points out of range");
            for (int k = 0; k < n_points[i]; ++k)
                if (level[i * 8 + k] < 0 || level[i * 8 + k] > 15 || loc[i * 8 + k] < 0 || loc[i * 8 + k] > 31)
                    return fail(c, AT3HIP_EINVAL, "gain point out of range");
        }
    }
    MdctItemsParams mp;
    mp.bands = bands;
    mp.specs = specs;
    mp.n_points = n_points;
    mp.level = level;
    mp.loc = loc;
    mp.max_levels = max_levels;
    if (!dev) {
        // host buffers: one staging block per context, laid out [bands | specs | max_levels | n_points | level | loc]
        const size_t o_specs = n * 2048 * sizeof(float), o_max = o_specs + n * 1024 * sizeof(float);
        const size_t o_np = o_max + n * 4 * sizeof(float), o_lv = o_np + n * 4 * sizeof(int32_t), o_lc = o_lv + n * 32 * sizeof(int32_t);
        const size_t total = o_lc + n * 32 * sizeof(int32_t);
        const int rc = stage_reserve(c, total);
        if (rc != AT3HIP_OK) return rc;
        char* base = (char*)c->d_stage;
        mp.bands = (float*)base;
        mp.specs = (float*)(base + o_specs);
        mp.max_levels = max_levels ? (float*)(base + o_max) : nullptr;
        HIPCHK(c, hipMemcpyAsync(mp.bands, bands, n * 2048 * sizeof(float), hipMemcpyHostToDevice, st));
        if (n_points) {
            mp.n_points = (const int32_t*)(base + o_np);
            mp.level = (const int32_t*)(base + o_lv);
            mp.loc = (const int32_t*)(base + o_lc);
            HIPCHK(c, hipMemcpyAsync(base + o_np, n_points, n * 4 * sizeof(int32_t), hipMemcpyHostToDevice, st));
            HIPCHK(c, hipMemcpyAsync(base + o_lv, level, n * 32 * sizeof(int32_t), hipMemcpyHostToDevice, st));
            HIPCHK(c, hipMemcpyAsync(base + o_lc, loc, n * 32 * sizeof(int32_t), hipMemcpyHostToDevice, st));
        }
    }
    hipLaunchKernelGGL(k_mdct_items, dim3((unsigned)n), dim3(128), 0, st, mp, c->d_tables);
    HIPCHK(c, hipGetLastError());
    if (!dev) {
        HIPCHK(c, hipMemcpyAsync(bands, mp.bands, n * 2048 * sizeof(float), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipMemcpyAsync(specs, mp.specs, n * 1024 * sizeof(float), hipMemcpyDeviceToHost, st));
        if (max_levels) HIPCHK(c, hipMemcpyAsync(max_levels, mp.max_levels, n * 4 * sizeof(float), hipMemcpyDeviceToHost, st));
    }
    HIPCHK(c, hipStreamSynchronize(st));
    return AT3HIP_OK;
}

}  // namespace

int at3hip_mdct(at3hip_ctx* c, float* bands, float* specs, const int32_t* n_points, const int32_t* level,
                const int32_t* loc, int32_t n_items, uint32_t flags)
{
    return mdct_impl(c, bands, specs, nullptr, n_points, level, loc, n_items, flags);
}

int at3hip_mdct_levels(at3hip_ctx* c, float* bands, float* specs, float* max_levels, const int32_t* n_points,
                       const int32_t* level, const int32_t* loc, int32_t n_items, uint32_t flags)
{
    if (!max_levels) return c ? fail(c, AT3HIP_EINVAL, "max_levels is null") : AT3HIP_EINVAL;
    return mdct_impl(c, bands, specs, max_levels, n_points, level, loc, n_items, flags);
}

int at3hip_gain_energy_scale(at3hip_ctx* c, const float* prev_overlap, const float* cur_input, const int32_t* n_points,
                             const int32_t* level, const int32_t* loc, const float* prev_overlap_scale, float* out,
                             int32_t n_items, uint32_t flags)
{
    if (!c || !prev_overlap || !cur_input || !prev_overlap_scale || !out || n_items < 1)
        return c ? fail(c, AT3HIP_EINVAL, "bad argument") : AT3HIP_EINVAL;
    if ((n_points != nullptr) != (level != nullptr) || (n_points != nullptr) != (loc != nullptr))
        return fail(c, AT3HIP_EINVAL, "n_points/level/loc must be all null or all set");
    at3host::DeviceGuard guard(c->device);
    HIPCHK(c, guard.error());
    hipStream_t st = c->stream;
    const bool dev = (flags & AT3HIP_PCM_ON_DEVICE) && (flags & AT3HIP_OUT_ON_DEVICE);
    if (!dev && (flags & (AT3HIP_PCM_ON_DEVICE | AT3HIP_OUT_ON_DEVICE)))
        return fail(c, AT3HIP_EINVAL, "buffers must be all host or all device");
    const size_t n = n_items;
    if (!dev && n_points) {
        for (size_t i = 0; i < n; ++i) {
            if (n_points[i] < 0 || n_points[i] > 7) return fail(c, AT3HIP_EINVAL, "n_points out of range");
            for (int k = 0; k < n_points[i]; ++k)
                if (level[i * 8 + k] < 0 || level[i * 8 + k] > 15 || loc[i * 8 + k] < 0 || loc[i * 8 + k] > 31)
                    return fail(c, AT3HIP_EINVAL, "gain point out of range");
        }
    }
    GesItemsParams gp;
    gp.prev_overlap = prev_overlap;
    gp.cur_input = cur_input;
    gp.n_points = n_points;
    gp.level = level;
    gp.loc = loc;
    gp.prev_scale = prev_overlap_scale;
    gp.out = out;
    if (!dev) {
        const size_t o_cur = n * 256 * sizeof(float), o_ps = 2 * o_cur, o_out = o_ps + n * sizeof(float);
        const size_t o_np = o_out + n * 4 * sizeof(float), o_lv = o_np + n * sizeof(int32_t), o_lc = o_lv + n * 8 * sizeof(int32_t);
        const size_t total = o_lc + n * 8 * sizeof(int32_t);
        const int rc = stage_reserve(c, total);
        if (rc != AT3HIP_OK) return rc;
        char* base = (char*)c->d_stage;
        HIPCHK(c, hipMemcpyAsync(base, prev_overlap, n * 256 * sizeof(float), hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(base + o_cur, cur_input, n * 256 * sizeof(float), hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(base + o_ps, prev_overlap_scale, n * sizeof(float), hipMemcpyHostToDevice, st));
        gp.prev_overlap = (const float*)base;
        gp.cur_input = (const float*)(base + o_cur);
        gp.prev_scale = (const float*)(base + o_ps);
        gp.out = (float*)(base + o_out);
        if (n_points) {
            HIPCHK(c, hipMemcpyAsync(base + o_np, n_points, n * sizeof(int32_t), hipMemcpyHostToDevice, st));
            HIPCHK(c, hipMemcpyAsync(base + o_lv, level, n * 8 * sizeof(int32_t), hipMemcpyHostToDevice, st));
            HIPCHK(c, hipMemcpyAsync(base + o_lc, loc, n * 8 * sizeof(int32_t), hipMemcpyHostToDevice, st));
            gp.n_points = (const int32_t*)(base + o_np);
            gp.level = (const int32_t*)(base + o_lv);
            gp.loc = (const int32_t*)(base + o_lc);
        }
    }
    hipLaunchKernelGGL(k_ges_items, dim3((unsigned)n), dim3(64), 0, st, gp, c->d_tables);
    HIPCHK(c, hipGetLastError());
    if (!dev) HIPCHK(c, hipMemcpyAsync(out, gp.out, n * 4 * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    return AT3HIP_OK;
}

int at3hip_qmf_mdct(at3hip_ctx* c, const float* pcm, int32_t n_blocks, float* specs, uint32_t flags)
{
    if (!c || !pcm || !specs || n_blocks < 2) return c ? fail(c, AT3HIP_EINVAL, "bad argument") : AT3HIP_EINVAL;
    if ((flags & (AT3HIP_PCM_ON_DEVICE | AT3HIP_OUT_ON_DEVICE)) != (AT3HIP_PCM_ON_DEVICE | AT3HIP_OUT_ON_DEVICE))
        return fail(c, AT3HIP_EINVAL, "qmf_mdct needs device pointers");
    at3host::DeviceGuard guard(c->device);
    HIPCHK(c, guard.error());
    {
        const int rc = drain(c);
        if (rc != AT3HIP_OK) return rc;
    }
    hipStream_t st = c->stream;
    const int S = c->cfg.n_streams;
    // start-of-stream history: an all-zero buffer (the spare history buffer is kept zeroed for this)
    float* zero_hist = c->d_hist[c->hist_cur ^ 1];
    HIPCHK(c, hipMemsetAsync(zero_hist, 0, (size_t)S * kHist * 2 * sizeof(float), st));
    FrontParams fp;
    fp.pcm = pcm;
    fp.hist = zero_hist;
    fp.curves = nullptr;
    fp.state = nullptr;
    fp.specs = specs;
    fp.ges = nullptr;
    fp.sub = nullptr;
    fp.sub_runs = 0;
    fp.debug = 0;
#ifdef K1_STAMPS
    fp.clk = c->d_clk;
    for (int r = 0; r < 256; r += 64)   // (the launch's own first / last starts and ends: words 12 .. 15 of every row)
        HIPCHK(c, hipMemset2DAsync(c->d_clk + 16 + 12 + r * 24, 24 * sizeof(unsigned long long), 0, 4 * sizeof(unsigned long long), 64, st));
#else
    fp.clk = nullptr;
#endif
    fp.n_blocks = n_blocks;
    fp.f0 = 1;
    const int n_out = n_blocks - 1;
    fp.js = c->js;
    {
        const FusedCut cut = c->js ? FusedCut{0, 0} : pick_fused(c, n_out);
        fp.frame_runs = cut.runs;
        fp.chain = cut.chain;
    }
    HIPCHK(c, hipEventRecord(c->ev[0][0], st));
    if (c->js) {
        // joint stereo: the M/S matrixing needs both channels' subbands, which go through HBM (as in at3hip_encode)
        if (n_blocks > c->cfg.max_blocks) return fail(c, AT3HIP_EINVAL, "n_blocks exceeds max_blocks");
        // the carried subbands must be the start-of-stream zeros this entry point is defined on
        if (c->blocks_fed != 0) return fail(c, AT3HIP_EINVAL, "qmf_mdct on a joint-stereo context needs a fresh or reset context");
        fp.sub = c->d_sub;
        fp.sub_tail = c->d_sub_tail;
        fp.sub_runs = pick_runs(c, n_blocks, c->wgs_per_cu, 0.35, 4);
        const int n_waves = S * 2 * fp.sub_runs;
        hipLaunchKernelGGL(k_qmf_sub8, dim3((unsigned)((n_waves + 3) / 4)), dim3(256), 0, st, fp, c->d_tables, n_waves);
        MdctSubParams mp;
        mp.sub = c->d_sub;
        mp.curves = nullptr;
        mp.state = nullptr;
        mp.specs = specs;
        mp.n_blocks = n_blocks;
        mp.f0 = 1;
        mp.js = 1;
        mp.frame_runs = pick_runs(c, n_out, c->wgs_per_cu_mdct, 0.3);
        mp.n_waves = S * 2 * mp.frame_runs;
        hipLaunchKernelGGL((k_mdct_sub<true, 4>), dim3((unsigned)((mp.n_waves + 3) / 4)), dim3(256), 0, st, mp, c->d_tables);
    } else {
        const int n_waves = S * 2 * fp.frame_runs;
        hipLaunchKernelGGL(k_qmf_mdct8, dim3((unsigned)((n_waves + kFusedWaves - 1) / kFusedWaves)), dim3(64 * kFusedWaves), 0, st, fp, c->d_tables, n_waves);
    }
    HIPCHK(c, hipEventRecord(c->ev[0][1], st));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(st));
    memset(&c->tm, 0, sizeof(c->tm));
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, c->ev[0][0], c->ev[0][1]);
    c->tm.qmf_mdct_ms = ms;
    c->tm.total_ms = ms;
    c->tm.qmf_mdct_launches = 1;
    return AT3HIP_OK;
}

}  // extern "C"

#if defined(AT3_EMU_HOST) || defined(AT3HIP_DEBUG_KNOBS)
// TEST AND PROFILING BUILDS ONLY (the SIMT harness, -DAT3HIP_DEBUG_KNOBS; not in include/at3hip.h): the divisors of all 256 samples under n arbitrary
// curves (16 bytes each: n, level[7], loc[7], pad - any byte values) from the select walk the pipeline's kernels use (cell_divisors_packed) and from the
// sample-by-sample restatement of TGainProcessor::Modulate (curve_divisor), side by side: tests/test_kernels_simt_harness.py compares the bit patterns.
namespace {
__global__ __launch_bounds__(64) void k_debug_cell_divisors(const Curve* curves, const Tables* T, float* out_packed, float* out_ref, int n)
{
    __shared__ float s_gi[32];
    const int lane = threadIdx.x;
    if (lane < 32) s_gi[lane] = T->gain_interp[lane < 31 ? lane : 30];
    __syncthreads();
    const int item = blockIdx.x * 2 + (lane >> 5), cell = lane & 31;   // two curves per wavefront, a lane per cell of eight samples
    const int it = item < n ? item : n - 1;
    const uint4 w = *reinterpret_cast<const uint4*>(curves + it);
    float d[8];
    cell_divisors_packed((uint64_t)w.x | ((uint64_t)w.y << 32), (uint64_t)w.z | ((uint64_t)w.w << 32), s_gi, 8 * cell, d);
    if (item < n) {
        for (int k = 0; k < 8; ++k) {
            out_packed[(size_t)item * 256 + 8 * cell + k] = d[k];
            out_ref[(size_t)item * 256 + 8 * cell + k] = curve_divisor(s_gi, curves[it], 8 * cell + k);
        }
    }
}
}  // namespace

extern "C" __attribute__((visibility("default"))) int at3hip_debug_cell_divisors(at3hip_ctx* c, const void* curves, int32_t n, float* out_packed, float* out_ref)
{
    if (!c || !curves || n < 1 || !out_packed || !out_ref) return AT3HIP_EINVAL;
    at3host::DeviceGuard guard(c->device);
    Curve* d_cv = nullptr;
    float* d_out = nullptr;
    const size_t nb = (size_t)n * 256 * sizeof(float);
    if (hipMalloc((void**)&d_cv, (size_t)n * sizeof(Curve)) != hipSuccess || hipMalloc((void**)&d_out, 2 * nb) != hipSuccess) return AT3HIP_EDEVICE;
    int rc = AT3HIP_OK;
    if (hipMemcpy(d_cv, curves, (size_t)n * sizeof(Curve), hipMemcpyHostToDevice) != hipSuccess) rc = AT3HIP_EDEVICE;
    if (rc == AT3HIP_OK) {
        hipLaunchKernelGGL(k_debug_cell_divisors, dim3((unsigned)((n + 1) / 2)), dim3(64), 0, c->stream, d_cv, c->d_tables, d_out, d_out + (size_t)n * 256, n);
        if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(out_packed, d_out, nb, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(out_ref, d_out + (size_t)n * 256, nb, hipMemcpyDeviceToHost) != hipSuccess)
            rc = AT3HIP_EDEVICE;
    }
    (void)hipFree(d_cv);
    (void)hipFree(d_out);
    return rc;
}
#endif
