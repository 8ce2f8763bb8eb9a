// C ABI of the ATRAC1 encode path (include/at1hip.h): context, device buffers, kernel launches.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include <hip/hip_runtime.h>

#include "../../include/at1hip.h"
#include "at1_kernels.hpp"
#include "at3_host_util.hpp"

using namespace at1;

static_assert(sizeof(Tables) == AT1HIP_TABLES_BYTES, "at1hip.h documents the table block's size");

struct at1hip_ctx {
    at1hip_config cfg;
    int device = 0;
    int debug_stop = 0;           // AT1HIP_DEBUG_STOP, honoured by -DAT3HIP_DEBUG_KNOBS builds only
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {};
    bool tm_pending = false;      // a call's stage events have not been read yet (AT3HIP_ASYNC)
    Tables* d_tables = nullptr;
    float* d_pcm_in = nullptr;    // staging for host PCM [S][max_blocks][512][nch]
    float* d_hist = nullptr;      // [S][512][nch] last PCM block of the previous call
    float* d_specs = nullptr;     // [S][B][nch][512]
    float* d_values = nullptr;    // [S][B][nch][512]
    float* d_energy = nullptr;    // [S][B][nch][52]
    uint8_t* d_sfi = nullptr;     // [S][B][nch][64]
    int32_t* d_mask = nullptr;    // [S][B][nch]
    float* d_loud_ch = nullptr;   // [S][B][nch]
    float* d_loud_state = nullptr;  // [S]
    float* d_loud_track = nullptr;  // [S][B]
    uint8_t* d_out = nullptr;     // staging for host output [S][B][nch][212]
    long long blocks_fed = 0;
    int last_blocks = 0;
    at1hip_timings tm = {};
    char err[256] = {0};
};

namespace {

int fail(at1hip_ctx* c, int code, const char* what, hipError_t e = hipSuccess)
{
    if (c) {
        if (e != hipSuccess) snprintf(c->err, sizeof(c->err), "%s: %s", what, hipGetErrorString(e));
        else snprintf(c->err, sizeof(c->err), "%s", what);
    }
    return code;
}

#define HIPCHK(c, call)                                                    \
    do {                                                                   \
        hipError_t e_ = (call);                                            \
        if (e_ != hipSuccess) return fail((c), AT3HIP_EDEVICE, #call, e_); \
    } while (0)

template <typename Tp>
int dev_alloc(at1hip_ctx* c, Tp** p, size_t count)
{
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, count * sizeof(Tp) + 256);
    if (e != hipSuccess) return fail(c, AT3HIP_ENOMEM, "hipMalloc", e);
    *p = (Tp*)q;
    return AT3HIP_OK;
}

int reset_state(at1hip_ctx* c)
{
    const size_t S = c->cfg.n_streams;
    HIPCHK(c, hipMemsetAsync(c->d_hist, 0, S * 512 * c->cfg.channels * sizeof(float), c->stream));
    float* init = (float*)malloc(S * sizeof(float));
    if (!init) return fail(c, AT3HIP_ENOMEM, "malloc");
    for (size_t i = 0; i < S; ++i) init[i] = 0.006f;  // LoudFactor, atrac1denc.h:101-102
    hipError_t e = hipMemcpyAsync(c->d_loud_state, init, S * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    free(init);
    if (e != hipSuccess) return fail(c, AT3HIP_EDEVICE, "state upload", e);
    c->blocks_fed = 0;
    return AT3HIP_OK;
}

}  // namespace

extern "C" {

int at1hip_create(const at1hip_config* cfg, at1hip_ctx** out)
{
    if (!cfg || !out) return AT3HIP_EINVAL;
    *out = nullptr;
    if ((cfg->channels != 1 && cfg->channels != 2) || cfg->n_streams < 1 || cfg->max_blocks < 1 || cfg->bfu_idx_const < 0 ||
        cfg->bfu_idx_const > 8 || cfg->window_mask < 0 || cfg->window_mask > 7)
        return AT3HIP_EINVAL;
    if ((long long)cfg->n_streams * cfg->channels > at3host::kMaxGridY) return AT3HIP_EINVAL;   // (stream, channel) is gridDim.y
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return AT3HIP_EDEVICE;
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return AT3HIP_EINVAL;
    at1hip_ctx* c = new (std::nothrow) at1hip_ctx();
    if (!c) return AT3HIP_ENOMEM;
    c->cfg = *cfg;
    c->device = cfg->device_id;
#ifdef AT3HIP_DEBUG_KNOBS
    if (const char* dbg = getenv("AT1HIP_DEBUG_STOP")) c->debug_stop = atoi(dbg);
#endif
    int rc = AT3HIP_OK;
    auto bail = [&](int code) {
        at1hip_destroy(c);
        return code;
    };
    at3host::DeviceGuard guard(c->device);
    if (guard.error() != hipSuccess) return bail(AT3HIP_EDEVICE);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail(AT3HIP_EDEVICE);
    for (auto& e : c->ev)
        if (hipEventCreate(&e) != hipSuccess) return bail(AT3HIP_EDEVICE);

    Tables* host_tables = new (std::nothrow) Tables();
    if (!host_tables) return bail(AT3HIP_ENOMEM);
    build_tables(host_tables);
    rc = dev_alloc(c, &c->d_tables, 1);
    if (rc == AT3HIP_OK && (hipMemcpy(c->d_tables, host_tables, sizeof(Tables), hipMemcpyHostToDevice) != hipSuccess ||
                            hipDeviceSynchronize() != hipSuccess))   // (pageable source: the transfer may still be running when the copy returns, at3hip_create)
        rc = AT3HIP_EDEVICE;
    delete host_tables;
    if (rc != AT3HIP_OK) return bail(rc);

    const size_t S = cfg->n_streams, B = cfg->max_blocks, C = cfg->channels;
    if ((rc = dev_alloc(c, &c->d_pcm_in, S * B * 512 * C)) != AT3HIP_OK) return bail(rc);
    if ((rc = dev_alloc(c, &c->d_hist, S * 512 * C)) != AT3HIP_OK) return bail(rc);
    if ((rc = dev_alloc(c, &c->d_specs, S * B * C * 512)) != AT3HIP_OK) return bail(rc);
    if ((rc = dev_alloc(c, &c->d_values, S * B * C * 512)) != AT3HIP_OK) return bail(rc);
    if ((rc = dev_alloc(c, &c->d_energy, S * B * C * kMaxBfus)) != AT3HIP_OK) return bail(rc);
    if ((rc = dev_alloc(c, &c->d_sfi, S * B * C * 64)) != AT3HIP_OK) return bail(rc);
    if ((rc = dev_alloc(c, &c->d_mask, S * B * C)) != AT3HIP_OK) return bail(rc);
    if ((rc = dev_alloc(c, &c->d_loud_ch, S * B * C)) != AT3HIP_OK) return bail(rc);
    if ((rc = dev_alloc(c, &c->d_loud_state, S)) != AT3HIP_OK) return bail(rc);
    if ((rc = dev_alloc(c, &c->d_loud_track, S * B)) != AT3HIP_OK) return bail(rc);
    if ((rc = dev_alloc(c, &c->d_out, S * B * C * kFrame)) != AT3HIP_OK) return bail(rc);
    if ((rc = reset_state(c)) != AT3HIP_OK) return bail(rc);
    *out = c;
    return AT3HIP_OK;
}

void at1hip_destroy(at1hip_ctx* c)
{
    if (!c) return;
    at3host::DeviceGuard guard(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    void* bufs[] = {c->d_tables, c->d_pcm_in, c->d_hist,       c->d_specs,      c->d_values, c->d_energy,
                    c->d_sfi,    c->d_mask,   c->d_loud_ch,    c->d_loud_state, c->d_loud_track, c->d_out};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    for (auto& e : c->ev)
        if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* at1hip_last_error(const at1hip_ctx* c) { return c ? c->err : "null context"; }

int at1hip_reset(at1hip_ctx* c)
{
    if (!c) return AT3HIP_EINVAL;
    at3host::DeviceGuard guard(c->device);
    HIPCHK(c, guard.error());
    return reset_state(c);
}

int at1hip_encode(at1hip_ctx* c, const float* pcm, int32_t n_blocks, uint8_t* out_frames, uint32_t flags)
{
    if (!c || !pcm || !out_frames || n_blocks < 1 || n_blocks > c->cfg.max_blocks)
        return c ? fail(c, AT3HIP_EINVAL, "bad argument") : AT3HIP_EINVAL;
    at3host::DeviceGuard guard(c->device);
    HIPCHK(c, guard.error());
    const size_t S = c->cfg.n_streams, C = c->cfg.channels, F = (size_t)n_blocks;
    hipStream_t st = c->stream;
    const bool timed = !(flags & AT3HIP_ASYNC);   // a queued call carries no stage-timing events (not free between the kernels): its timings read zero
    const float* d_pcm = pcm;
    if (!(flags & AT3HIP_PCM_ON_DEVICE)) {
        HIPCHK(c, hipMemcpyAsync(c->d_pcm_in, pcm, S * F * 512 * C * sizeof(float), hipMemcpyHostToDevice, st));
        d_pcm = c->d_pcm_in;
    }
    uint8_t* d_out = (flags & AT3HIP_OUT_ON_DEVICE) ? out_frames : c->d_out;

    if (timed) HIPCHK(c, hipEventRecord(c->ev[0], st));
    FrontParams fp;
    fp.T = c->d_tables;
    fp.pcm = d_pcm;
    fp.hist = c->d_hist;
    fp.n_frames = n_blocks;
    fp.nch = (int)C;
    fp.first = c->blocks_fed == 0;
    fp.window_auto = c->cfg.window_auto ? 1 : 0;
    fp.window_mask = c->cfg.window_mask;
    fp.debug = c->debug_stop;
    fp.specs = c->d_specs;
    fp.values = c->d_values;
    fp.energy = c->d_energy;
    fp.sfi = c->d_sfi;
    fp.mask = c->d_mask;
    fp.loud_ch = c->d_loud_ch;
    hipLaunchKernelGGL(k_at1_front, dim3((unsigned)F, (unsigned)(S * C)), dim3(64), 0, st, fp);
    HIPCHK(c, hipGetLastError());
    LoudParams lp;
    lp.T = c->d_tables;
    lp.specs = c->d_specs;
    lp.loud_ch = c->d_loud_ch;
    lp.n_units = (int32_t)(S * F * C);
    hipLaunchKernelGGL(k_at1_loud, dim3((unsigned)((S * F * C + kAt1LoudUnits - 1) / kAt1LoudUnits)), dim3(256), 0, st, lp);
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(k_at1_state, dim3((unsigned)((S * 512 * C + 255) / 256)), dim3(256), 0, st, d_pcm, c->d_hist, n_blocks, (int)C,
                       (int)S);
    if (timed) HIPCHK(c, hipEventRecord(c->ev[1], st));

    ScanParams sp;
    sp.mask = c->d_mask;
    sp.loud_ch = c->d_loud_ch;
    sp.loud_state = c->d_loud_state;
    sp.loud_track = c->d_loud_track;
    sp.n_streams = (int)S;
    sp.n_frames = n_blocks;
    sp.nch = (int)C;
    hipLaunchKernelGGL(k_at1_loud_scan, dim3((unsigned)S), dim3(64), 0, st, sp);
    if (timed) HIPCHK(c, hipEventRecord(c->ev[2], st));

    PackParams pp;
    pp.T = c->d_tables;
    pp.values = c->d_values;
    pp.energy = c->d_energy;
    pp.sfi = c->d_sfi;
    pp.mask = c->d_mask;
    pp.loud_track = c->d_loud_track;
    pp.out = d_out;
    pp.n_items = (int)(S * F * C);
    pp.nch = (int)C;
    pp.bfu_idx_const = c->cfg.bfu_idx_const;
    hipLaunchKernelGGL(k_at1_alloc_pack, dim3((unsigned)((S * F * C + 3) / 4)), dim3(256), 0, st, pp);
    HIPCHK(c, hipGetLastError());
    if (timed) HIPCHK(c, hipEventRecord(c->ev[3], st));
    if (!(flags & AT3HIP_OUT_ON_DEVICE))
        HIPCHK(c, hipMemcpyAsync(out_frames, c->d_out, S * F * C * kFrame, hipMemcpyDeviceToHost, st));
    c->blocks_fed += n_blocks;
    c->last_blocks = n_blocks;
    c->tm_pending = timed;
    if (!timed) memset(&c->tm, 0, sizeof(c->tm));
    // AT3HIP_ASYNC: the call is queued (one stream: consecutive calls follow each other on the device without the host in between);
    // at1hip_sync waits and reads the last call's timings
    return (flags & AT3HIP_ASYNC) ? AT3HIP_OK : at1hip_sync(c);
}

int at1hip_sync(at1hip_ctx* c)
{
    if (!c) return AT3HIP_EINVAL;
    at3host::DeviceGuard guard(c->device);
    HIPCHK(c, guard.error());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->tm_pending) {
        c->tm_pending = false;
        (void)hipEventElapsedTime(&c->tm.front_ms, c->ev[0], c->ev[1]);
        (void)hipEventElapsedTime(&c->tm.scan_ms, c->ev[1], c->ev[2]);
        (void)hipEventElapsedTime(&c->tm.pack_ms, c->ev[2], c->ev[3]);
        (void)hipEventElapsedTime(&c->tm.total_ms, c->ev[0], c->ev[3]);
    }
    return AT3HIP_OK;
}

int at1hip_host_tables(void* dst, size_t bytes)
{
    if (!dst || bytes != sizeof(Tables)) return AT3HIP_EINVAL;
    build_tables((Tables*)dst);
    return AT3HIP_OK;
}

int at1hip_get_timings(const at1hip_ctx* c, at1hip_timings* out)
{
    if (!c || !out) return AT3HIP_EINVAL;
    *out = c->tm;
    return AT3HIP_OK;
}

int at1hip_read_tap(at1hip_ctx* c, int32_t kind, void* dst, size_t bytes)
{
    if (!c || !dst) return AT3HIP_EINVAL;
    const size_t S = c->cfg.n_streams, C = c->cfg.channels, F = (size_t)c->last_blocks;
    const void* src = nullptr;
    size_t need = 0;
    switch (kind) {
        case AT1HIP_TAP_SPECTRA: src = c->d_specs; need = S * F * C * 512 * sizeof(float); break;
        case AT1HIP_TAP_MASKS: src = c->d_mask; need = S * F * C * sizeof(int32_t); break;
        case AT1HIP_TAP_LOUDNESS: src = c->d_loud_track; need = S * F * sizeof(float); break;
        case AT1HIP_TAP_TABLES: src = c->d_tables; need = sizeof(Tables); break;
        default: return fail(c, AT3HIP_EINVAL, "unknown tap");
    }
    if (bytes != need || (kind != AT1HIP_TAP_TABLES && F == 0)) return fail(c, AT3HIP_EINVAL, "tap size");
    at3host::DeviceGuard guard(c->device);
    HIPCHK(c, guard.error());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(dst, src, need, hipMemcpyDeviceToHost));
    return AT3HIP_OK;
}

}  // extern "C"
