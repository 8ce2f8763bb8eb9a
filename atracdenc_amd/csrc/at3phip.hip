// C ABI of the ATRAC3plus front end (include/at3phip.h): context, device buffers, kernel launches.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include <hip/hip_runtime.h>

#include "../../include/at3phip.h"
#include "at3p_kernels.hpp"
#include "at3p_write.hpp"
#include "at3_host_util.hpp"

using namespace at3p;

static_assert(sizeof(Tables) == AT3PHIP_TABLES_BYTES, "at3phip.h documents the table block's size");
static_assert(sizeof(WriteTables) == AT3PHIP_WRITE_TABLES_BYTES, "at3phip.h documents the frame writer's table block size");

struct at3phip_ctx {
    at3phip_config cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[5] = {};
    Tables* d_tables = nullptr;
    float* d_pcm_in = nullptr;     // staging for host PCM   [S][F][2048][nch]
    float* d_bands = nullptr;      // subband samples        [S][F][nch][16][128]
    float* d_specs = nullptr;      // staging for host specs [S][F][nch][2048]
    uint16_t* d_flags = nullptr;   // [S][F][nch]
    float* d_pqf_hist = nullptr;   // [S][nch][368]
    float* d_mdct_hist = nullptr;  // [S][nch][16][128]
    WriteTables* d_wtables = nullptr;
    uint8_t* d_frames = nullptr;   // staging for host frames [S][F][2048]
    // at3phip_encode_frames: the frame writer only needs the spectra of ITS call, so it runs on a stream of its own behind
    // an event and the next call's filter bank and transform overlap it; the spectra in between are double-buffered
    hipStream_t write_stream = nullptr;
    float* d_specs_b[2] = {nullptr, nullptr};
    hipEvent_t ev_specs[2] = {}, ev_write_done[2] = {};
    bool write_done_valid[2] = {false, false};
    long long enc_calls = 0;
    bool ev_from_encode = false;   // the timing events were last recorded by at3phip_encode_frames (at3phip_sync may read all of them)
    float pqf_ms = 0.0f, mdct_ms = 0.0f, write_ms = 0.0f;
    char err[256] = {0};
};

namespace {

int fail(at3phip_ctx* c, int code, const char* what, hipError_t e = hipSuccess)
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
int dev_alloc(at3phip_ctx* c, Tp** p, size_t count)
{
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, count * sizeof(Tp) + 256);
    if (e != hipSuccess) return fail(c, AT3HIP_ENOMEM, "hipMalloc", e);
    *p = (Tp*)q;
    return AT3HIP_OK;
}

int reset_state(at3phip_ctx* c)
{
    const size_t S = c->cfg.n_streams, C = c->cfg.channels;
    HIPCHK(c, hipMemsetAsync(c->d_pqf_hist, 0, S * C * kOverlap * sizeof(float), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_mdct_hist, 0, S * C * 2048 * sizeof(float), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AT3HIP_OK;
}

// Entry points other than at3phip_encode_frames work on the context's main stream only: whatever asynchronous calls left
// on the writer's stream is waited for first.
int quiesce(at3phip_ctx* c)
{
    if (c->write_stream) HIPCHK(c, hipStreamSynchronize(c->write_stream));
    return AT3HIP_OK;
}

int launch_pqf(at3phip_ctx* c, const float* d_pcm, int n_frames, float* d_bands)
{
    const size_t S = c->cfg.n_streams, C = c->cfg.channels;
    PqfParams pp;
    pp.T = c->d_tables;
    pp.pcm = d_pcm;
    pp.hist = c->d_pqf_hist;
    pp.bands = d_bands;
    pp.n_frames = n_frames;
    pp.nch = (int)C;
    hipLaunchKernelGGL(k_at3p_pqf, dim3((unsigned)n_frames, (unsigned)(S * C)), dim3(256), 0, c->stream, pp);
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(k_at3p_pqf_state, dim3((unsigned)((S * C * kOverlap + 255) / 256)), dim3(256), 0, c->stream, d_pcm, c->d_pqf_hist,
                       n_frames, (int)C, (int)S);
    HIPCHK(c, hipGetLastError());
    return AT3HIP_OK;
}

int launch_mdct(at3phip_ctx* c, const float* d_bands, int n_frames, const uint16_t* win_flags, float* d_specs, uint32_t flags)
{
    const size_t S = c->cfg.n_streams, C = c->cfg.channels;
    if (win_flags) HIPCHK(c, hipMemcpyAsync(c->d_flags, win_flags, S * n_frames * C * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream));
    MdctParams mp;
    mp.T = c->d_tables;
    mp.bands = d_bands;
    mp.flags = win_flags ? c->d_flags : nullptr;
    mp.hist = c->d_mdct_hist;
    mp.specs = d_specs;
    mp.n_frames = n_frames;
    mp.nch = (int)C;
    mp.residual_scale = (flags & AT3PHIP_RESIDUAL_SCALE) ? 1 : 0;
    hipLaunchKernelGGL(k_at3p_mdct, dim3((unsigned)n_frames, (unsigned)(S * C)), dim3(256), 0, c->stream, mp);
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(k_at3p_mdct_state, dim3((unsigned)((S * C * 2048 + 255) / 256)), dim3(256), 0, c->stream, mp, c->d_mdct_hist, (int)S);
    HIPCHK(c, hipGetLastError());
    return AT3HIP_OK;
}

int launch_write(at3phip_ctx* c, const float* d_specs, int n_frames, const uint16_t* win_flags, uint8_t* d_frames, hipStream_t on = nullptr)
{
    const size_t S = c->cfg.n_streams, C = c->cfg.channels;
    if (!on) on = c->stream;
    if (win_flags) HIPCHK(c, hipMemcpyAsync(c->d_flags, win_flags, S * n_frames * C * sizeof(uint16_t), hipMemcpyHostToDevice, on));
    WriteParams wp;
    wp.W = c->d_wtables;
    wp.specs = d_specs;
    wp.flags = win_flags ? c->d_flags : nullptr;
    wp.out = d_frames;
    wp.nch = (int)C;
    wp.n_items = (int)(S * n_frames);
    hipLaunchKernelGGL(k_at3p_write, dim3((unsigned)(S * n_frames)), dim3(256), 0, on, wp);
    HIPCHK(c, hipGetLastError());
    return AT3HIP_OK;
}

}  // namespace

extern "C" {

int at3phip_create(const at3phip_config* cfg, at3phip_ctx** out)
{
    if (!cfg || !out) return AT3HIP_EINVAL;
    *out = nullptr;
    if ((cfg->channels != 1 && cfg->channels != 2) || cfg->n_streams < 1 || cfg->max_frames < 1) return AT3HIP_EINVAL;
    if ((long long)cfg->n_streams * cfg->channels > at3host::kMaxGridY) return AT3HIP_EINVAL;   // (stream, channel) is gridDim.y
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return AT3HIP_EDEVICE;
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return AT3HIP_EINVAL;
    at3phip_ctx* c = new (std::nothrow) at3phip_ctx();
    if (!c) return AT3HIP_ENOMEM;
    c->cfg = *cfg;
    c->device = cfg->device_id;
    int rc = AT3HIP_OK;
    auto bail = [&](int code) {
        at3phip_destroy(c);
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
    const size_t S = cfg->n_streams, F = cfg->max_frames, C = cfg->channels;
    if ((rc = dev_alloc(c, &c->d_pcm_in, S * F * C * 2048)) != AT3HIP_OK) return bail(rc);
    if ((rc = dev_alloc(c, &c->d_bands, S * F * C * 2048)) != AT3HIP_OK) return bail(rc);
    if ((rc = dev_alloc(c, &c->d_specs, S * F * C * 2048)) != AT3HIP_OK) return bail(rc);
    if ((rc = dev_alloc(c, &c->d_flags, S * F * C)) != AT3HIP_OK) return bail(rc);
    if ((rc = dev_alloc(c, &c->d_pqf_hist, S * C * kOverlap)) != AT3HIP_OK) return bail(rc);
    if ((rc = dev_alloc(c, &c->d_mdct_hist, S * C * 2048)) != AT3HIP_OK) return bail(rc);
    if ((rc = dev_alloc(c, &c->d_frames, S * F * kFrameBytes)) != AT3HIP_OK) return bail(rc);
    c->d_specs_b[0] = c->d_specs;
    if ((rc = dev_alloc(c, &c->d_specs_b[1], S * F * C * 2048)) != AT3HIP_OK) return bail(rc);
    if (hipStreamCreateWithFlags(&c->write_stream, hipStreamNonBlocking) != hipSuccess) return bail(AT3HIP_EDEVICE);
    for (int q = 0; q < 2; ++q)
        if (hipEventCreateWithFlags(&c->ev_specs[q], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_write_done[q], hipEventDisableTiming) != hipSuccess)
            return bail(AT3HIP_EDEVICE);
    {
        WriteTables* wt = new (std::nothrow) WriteTables();
        if (!wt) return bail(AT3HIP_ENOMEM);
        build_write_tables(wt);
        rc = dev_alloc(c, &c->d_wtables, 1);
        if (rc == AT3HIP_OK && (hipMemcpy(c->d_wtables, wt, sizeof(WriteTables), hipMemcpyHostToDevice) != hipSuccess || hipDeviceSynchronize() != hipSuccess)) rc = AT3HIP_EDEVICE;
        delete wt;
        if (rc != AT3HIP_OK) return bail(rc);
    }
    if ((rc = reset_state(c)) != AT3HIP_OK) return bail(rc);
    *out = c;
    return AT3HIP_OK;
}

void at3phip_destroy(at3phip_ctx* c)
{
    if (!c) return;
    at3host::DeviceGuard guard(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->write_stream) (void)hipStreamSynchronize(c->write_stream);
    if (c->d_specs_b[1]) (void)hipFree(c->d_specs_b[1]);
    for (int q = 0; q < 2; ++q) {
        if (c->ev_specs[q]) (void)hipEventDestroy(c->ev_specs[q]);
        if (c->ev_write_done[q]) (void)hipEventDestroy(c->ev_write_done[q]);
    }
    if (c->write_stream) (void)hipStreamDestroy(c->write_stream);
    void* bufs[] = {c->d_tables, c->d_pcm_in, c->d_bands, c->d_specs, c->d_flags, c->d_pqf_hist, c->d_mdct_hist, c->d_wtables, c->d_frames};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    for (auto& e : c->ev)
        if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* at3phip_last_error(const at3phip_ctx* c) { return c ? c->err : "null context"; }

int at3phip_reset(at3phip_ctx* c)
{
    if (!c) return AT3HIP_EINVAL;
    at3host::DeviceGuard guard(c->device);
    HIPCHK(c, guard.error());
    if (int qrc = quiesce(c)) return qrc;
    return reset_state(c);
}

int at3phip_pqf_analyse(at3phip_ctx* c, const float* pcm, int32_t n_frames, float* bands, uint32_t flags)
{
    if (!c || !pcm || !bands || n_frames < 1 || n_frames > c->cfg.max_frames) return c ? fail(c, AT3HIP_EINVAL, "bad argument") : AT3HIP_EINVAL;
    at3host::DeviceGuard guard(c->device);
    HIPCHK(c, guard.error());
    if (int qrc = quiesce(c)) return qrc;
    const size_t n = (size_t)c->cfg.n_streams * n_frames * c->cfg.channels * 2048;
    const float* d_pcm = pcm;
    if (!(flags & AT3HIP_PCM_ON_DEVICE)) {
        HIPCHK(c, hipMemcpyAsync(c->d_pcm_in, pcm, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
        d_pcm = c->d_pcm_in;
    }
    float* d_bands = (flags & AT3HIP_OUT_ON_DEVICE) ? bands : c->d_bands;
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    c->ev_from_encode = false;
    int rc = launch_pqf(c, d_pcm, n_frames, d_bands);
    if (rc != AT3HIP_OK) return rc;
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    if (!(flags & AT3HIP_OUT_ON_DEVICE)) HIPCHK(c, hipMemcpyAsync(bands, c->d_bands, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipEventElapsedTime(&c->pqf_ms, c->ev[0], c->ev[1]);
    c->mdct_ms = 0.0f;
    return AT3HIP_OK;
}

int at3phip_mdct(at3phip_ctx* c, const float* bands, int32_t n_frames, const uint16_t* win_flags, float* specs, uint32_t flags)
{
    if (!c || !bands || !specs || n_frames < 1 || n_frames > c->cfg.max_frames) return c ? fail(c, AT3HIP_EINVAL, "bad argument") : AT3HIP_EINVAL;
    at3host::DeviceGuard guard(c->device);
    HIPCHK(c, guard.error());
    if (int qrc = quiesce(c)) return qrc;
    const size_t n = (size_t)c->cfg.n_streams * n_frames * c->cfg.channels * 2048;
    const float* d_bands = bands;
    if (!(flags & AT3HIP_PCM_ON_DEVICE)) {
        HIPCHK(c, hipMemcpyAsync(c->d_bands, bands, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
        d_bands = c->d_bands;
    }
    float* d_specs = (flags & AT3HIP_OUT_ON_DEVICE) ? specs : c->d_specs;
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    c->ev_from_encode = false;
    int rc = launch_mdct(c, d_bands, n_frames, win_flags, d_specs, flags);
    if (rc != AT3HIP_OK) return rc;
    HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
    if (!(flags & AT3HIP_OUT_ON_DEVICE)) HIPCHK(c, hipMemcpyAsync(specs, c->d_specs, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipEventElapsedTime(&c->mdct_ms, c->ev[1], c->ev[2]);
    c->pqf_ms = 0.0f;
    return AT3HIP_OK;
}

int at3phip_pqf_mdct(at3phip_ctx* c, const float* pcm, int32_t n_frames, const uint16_t* win_flags, float* bands, float* specs, uint32_t flags)
{
    if (!c || !pcm || !specs || n_frames < 1 || n_frames > c->cfg.max_frames) return c ? fail(c, AT3HIP_EINVAL, "bad argument") : AT3HIP_EINVAL;
    at3host::DeviceGuard guard(c->device);
    HIPCHK(c, guard.error());
    if (int qrc = quiesce(c)) return qrc;
    const size_t n = (size_t)c->cfg.n_streams * n_frames * c->cfg.channels * 2048;
    const float* d_pcm = pcm;
    if (!(flags & AT3HIP_PCM_ON_DEVICE)) {
        HIPCHK(c, hipMemcpyAsync(c->d_pcm_in, pcm, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
        d_pcm = c->d_pcm_in;
    }
    const bool out_dev = (flags & AT3HIP_OUT_ON_DEVICE) != 0;
    float* d_bands = (out_dev && bands) ? bands : c->d_bands;
    float* d_specs = out_dev ? specs : c->d_specs;
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    c->ev_from_encode = false;
    int rc = launch_pqf(c, d_pcm, n_frames, d_bands);
    if (rc != AT3HIP_OK) return rc;
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    rc = launch_mdct(c, d_bands, n_frames, win_flags, d_specs, flags);
    if (rc != AT3HIP_OK) return rc;
    HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
    if (!out_dev) {
        if (bands) HIPCHK(c, hipMemcpyAsync(bands, c->d_bands, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(specs, c->d_specs, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipEventElapsedTime(&c->pqf_ms, c->ev[0], c->ev[1]);
    (void)hipEventElapsedTime(&c->mdct_ms, c->ev[1], c->ev[2]);
    return AT3HIP_OK;
}

int at3phip_write_frames(at3phip_ctx* c, const float* specs, int32_t n_frames, const uint16_t* win_flags, uint8_t* frames, uint32_t flags)
{
    if (!c || !specs || !frames || n_frames < 1 || n_frames > c->cfg.max_frames) return c ? fail(c, AT3HIP_EINVAL, "bad argument") : AT3HIP_EINVAL;
    at3host::DeviceGuard guard(c->device);
    HIPCHK(c, guard.error());
    if (int qrc = quiesce(c)) return qrc;
    const size_t items = (size_t)c->cfg.n_streams * n_frames;
    const size_t n = items * c->cfg.channels * 2048;
    const float* d_specs = specs;
    if (!(flags & AT3HIP_PCM_ON_DEVICE)) {
        HIPCHK(c, hipMemcpyAsync(c->d_specs, specs, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
        d_specs = c->d_specs;
    }
    uint8_t* d_frames = (flags & AT3HIP_OUT_ON_DEVICE) ? frames : c->d_frames;
    HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
    int rc = launch_write(c, d_specs, n_frames, win_flags, d_frames);
    if (rc != AT3HIP_OK) return rc;
    HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
    if (!(flags & AT3HIP_OUT_ON_DEVICE)) HIPCHK(c, hipMemcpyAsync(frames, c->d_frames, items * kFrameBytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipEventElapsedTime(&c->write_ms, c->ev[2], c->ev[3]);
    c->pqf_ms = c->mdct_ms = 0.0f;
    c->ev_from_encode = false;   // ev[2], ev[3] now belong to this call: a later at3phip_sync must not mix them with an older encode's
    return AT3HIP_OK;
}

int at3phip_encode_frames(at3phip_ctx* c, const float* pcm, int32_t n_frames, uint8_t* frames, uint32_t flags)
{
    if (!c || !pcm || !frames || n_frames < 1 || n_frames > c->cfg.max_frames) return c ? fail(c, AT3HIP_EINVAL, "bad argument") : AT3HIP_EINVAL;
    at3host::DeviceGuard guard(c->device);
    HIPCHK(c, guard.error());
    const size_t items = (size_t)c->cfg.n_streams * n_frames;
    const size_t n = items * c->cfg.channels * 2048;
    const float* d_pcm = pcm;
    if (!(flags & AT3HIP_PCM_ON_DEVICE)) {
        HIPCHK(c, hipMemcpyAsync(c->d_pcm_in, pcm, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
        d_pcm = c->d_pcm_in;
    }
    uint8_t* d_frames = (flags & AT3HIP_OUT_ON_DEVICE) ? frames : c->d_frames;
    const int par = (int)(c->enc_calls & 1);
    float* d_specs = c->d_specs_b[par];
    hipStream_t ws = c->write_stream;
    // the writer of the call before the previous one must be done with this parity's spectra
    if (c->write_done_valid[par]) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_write_done[par], 0));
    // the stage timings are events between the kernels and not free (the ATRAC3 path measured ~1.5 us of the dependent chain per record):
    // a call that is only queued carries none - its timings read zero -, a synchronous one carries all five
    const bool timed = !(flags & AT3HIP_ASYNC);
    if (timed) HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    int rc = launch_pqf(c, d_pcm, n_frames, c->d_bands);
    if (rc != AT3HIP_OK) return rc;
    if (timed) HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    rc = launch_mdct(c, c->d_bands, n_frames, nullptr, d_specs, AT3PHIP_RESIDUAL_SCALE);   // sine windows: EncodeFrame's default Win
    if (rc != AT3HIP_OK) return rc;
    if (timed) HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
    HIPCHK(c, hipEventRecord(c->ev_specs[par], c->stream));
    HIPCHK(c, hipStreamWaitEvent(ws, c->ev_specs[par], 0));
    if (timed) HIPCHK(c, hipEventRecord(c->ev[4], ws));
    rc = launch_write(c, d_specs, n_frames, nullptr, d_frames, ws);
    if (rc != AT3HIP_OK) return rc;
    if (timed) HIPCHK(c, hipEventRecord(c->ev[3], ws));
    if (!(flags & AT3HIP_OUT_ON_DEVICE)) HIPCHK(c, hipMemcpyAsync(frames, c->d_frames, items * kFrameBytes, hipMemcpyDeviceToHost, ws));
    HIPCHK(c, hipEventRecord(c->ev_write_done[par], ws));
    c->write_done_valid[par] = true;
    c->enc_calls++;
    c->ev_from_encode = timed;
    if (!timed) c->pqf_ms = c->mdct_ms = c->write_ms = 0.0f;
    if (flags & AT3HIP_ASYNC) return AT3HIP_OK;   // at3phip_sync is the completion point
    return at3phip_sync(c);
}

int at3phip_sync(at3phip_ctx* c)
{
    if (!c) return AT3HIP_EINVAL;
    at3host::DeviceGuard guard(c->device);
    HIPCHK(c, guard.error());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->write_stream));
    if (c->ev_from_encode) {
        float a = 0.0f, b = 0.0f, w = 0.0f;
        if (hipEventElapsedTime(&a, c->ev[0], c->ev[1]) == hipSuccess && hipEventElapsedTime(&b, c->ev[1], c->ev[2]) == hipSuccess &&
            hipEventElapsedTime(&w, c->ev[4], c->ev[3]) == hipSuccess) {
            c->pqf_ms = a;
            c->mdct_ms = b;
            c->write_ms = w;
        }
    }
    return AT3HIP_OK;
}

int at3phip_get_write_timing(const at3phip_ctx* c, float* write_ms)
{
    if (!c) return AT3HIP_EINVAL;
    if (write_ms) *write_ms = c->write_ms;
    return AT3HIP_OK;
}

int at3phip_host_write_tables(void* dst, size_t bytes)
{
    if (!dst || bytes != sizeof(WriteTables)) return AT3HIP_EINVAL;
    build_write_tables((WriteTables*)dst);
    return AT3HIP_OK;
}

int at3phip_get_timings(const at3phip_ctx* c, float* pqf_ms, float* mdct_ms)
{
    if (!c) return AT3HIP_EINVAL;
    if (pqf_ms) *pqf_ms = c->pqf_ms;
    if (mdct_ms) *mdct_ms = c->mdct_ms;
    return AT3HIP_OK;
}

int at3phip_host_tables(void* dst, size_t bytes)
{
    if (!dst || bytes != sizeof(Tables)) return AT3HIP_EINVAL;
    build_tables((Tables*)dst);
    return AT3HIP_OK;
}

}  // extern "C"
