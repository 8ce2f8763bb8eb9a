// Host-side C++ mirror of the reference's surface for the ATRAC3 encode hot path, over the C ABI
// (include/at3hip.h). Same names, argument meaning and error behaviour as the reference classes:
//
//   TAtrac3MDCT      atrac3denc.h:56-92   (Mdct with in-place band mutation, both overloads; CalcGainEnergyScale)
//   TAtrac3Encoder   atrac3denc.h:94-134  (IProcessor::GetLambda() -> functor called once per 1024-sample
//                                          block; ICompressedOutput::WriteFrame once per encoded frame)
//
// The reference encoder is one stream, one frame per lambda call. The GPU path wants thousands of frames
// per launch, so TAtrac3Encoder here buffers `BatchBlocks` lambda calls, returns PROCESSED immediately
// (LOOK_AHEAD for the very first call, as the reference does) and flushes WriteFrame calls in order when
// the batch is full or on Flush()/destruction: observable behaviour equals the reference except latency.
// TAtrac3EncoderBatch is the natural multi-stream form (n independent streams side by side on one GPU) and
// TAtrac3EncoderNode the multi-GPU form: streams are independent, so a node shards them contiguously over its
// devices - one TAtrac3EncoderBatch and one host thread per device, no exchange between devices, no RCCL.
//
// Header-only; link with -lat3hip. Exceptions: std::runtime_error on any at3hip error (the reference
// throws from its sinks and aborts on impossible states; it never returns error codes).
#pragma once

#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>
#if defined(__linux__)
#include <sched.h>
#include <cstdio>
#endif

#include "../../include/at1hip.h"
#include "../../include/at3hip.h"
#include "../../include/at3phip.h"

namespace NAtracDEncHip {

// ---- the two reference interfaces on either side of the hot path (pcmengin.h:111-199, compressed_io.h:56-59)
struct ProcessMeta {
    const uint16_t Channels;
};
enum class EProcessResult { LOOK_AHEAD, PROCESSED };
using TProcessLambda = std::function<EProcessResult(float* data, const ProcessMeta& meta)>;

class ICompressedOutput {
public:
    virtual ~ICompressedOutput() = default;
    virtual void WriteFrame(std::vector<char> data) = 0;
    virtual std::string GetName() const = 0;
    virtual size_t GetChannelNum() const = 0;
};
using TCompressedOutputPtr = std::unique_ptr<ICompressedOutput>;

// NAtrac3::TAtrac3EncoderSettings (atrac/at3/atrac3.h:260-277)
struct TAtrac3EncoderSettings {
    uint32_t Bitrate = 0;            // bit/s; 0 = LP2 132300 (atrac3.cpp:47-53)
    bool NoGainControll = false;
    bool NoTonalComponents = false;
    uint8_t SourceChannels = 2;
    uint32_t BfuIdxConst = 0;
};

struct TGainPoint {                  // TAtrac3Data::SubbandInfo::TGainPoint (atrac3.h:224-227)
    uint32_t Level;
    uint32_t Location;
};

// The library this header runs against must implement at least the ABI minor number the header was written for (at3hip.h lists
// what each one added: three calls in flight need the four-deep wait ring of 1.2, Counters() needs 1.3). Called by every class
// below before its first at3hip_create.
inline void CheckLibraryVersion()
{
    const uint32_t have = at3hip_version();
    if ((have >> 16) != (uint32_t)AT3HIP_VERSION_MAJOR || have < (uint32_t)AT3HIP_VERSION)
        throw std::runtime_error("libat3hip.so implements at3hip ABI " + std::to_string(have >> 16) + "." + std::to_string(have & 0xffffu) +
                                 ", this host layer needs " + std::to_string(AT3HIP_VERSION_MAJOR) + "." + std::to_string(AT3HIP_VERSION_MINOR));
}

inline void Check(int rc, at3hip_ctx* ctx, const char* what)
{
    if (rc != AT3HIP_OK)
        throw std::runtime_error(std::string(what) + ": " + (ctx ? at3hip_last_error(ctx) : "at3hip error " + std::to_string(rc)));
}

// ---- TAtrac3MDCT (atrac3denc.h:56-92) ------------------------------------------------------------------
class TAtrac3MDCT {
public:
    using TGainCurves = std::array<std::vector<TGainPoint>, 4>;  // what MakeGainModulatorArray receives

    explicit TAtrac3MDCT(int deviceId = 0)
    {
        at3hip_config cfg{};
        cfg.channels = 2;
        cfg.n_streams = 1;
        cfg.max_blocks = 2;
        cfg.device_id = deviceId;
        cfg.no_gain_control = 1;
        CheckLibraryVersion();
        Check(at3hip_create(&cfg, &Ctx), nullptr, "at3hip_create");
    }
    ~TAtrac3MDCT() { at3hip_destroy(Ctx); }
    TAtrac3MDCT(const TAtrac3MDCT&) = delete;
    TAtrac3MDCT& operator=(const TAtrac3MDCT&) = delete;

    // Same contract as the reference: bands[b] -> 512 floats [overlap | new], mutated in place.
    void Mdct(float specs[1024], float* bands[4], const TGainCurves& curves = TGainCurves())
    {
        float packed[4 * 512];
        int32_t n[4], level[32] = {0}, loc[32] = {0};
        bool any = false;
        for (int b = 0; b < 4; ++b) {
            memcpy(packed + 512 * b, bands[b], 512 * sizeof(float));
            n[b] = (int32_t)curves[b].size();
            any = any || n[b] > 0;
            for (int i = 0; i < n[b] && i < 8; ++i) {
                level[8 * b + i] = (int32_t)curves[b][i].Level;
                loc[8 * b + i] = (int32_t)curves[b][i].Location;
            }
        }
        Check(at3hip_mdct(Ctx, packed, specs, any ? n : nullptr, any ? level : nullptr, any ? loc : nullptr, 1, 0), Ctx,
              "at3hip_mdct");
        for (int b = 0; b < 4; ++b) memcpy(bands[b], packed + 512 * b, 512 * sizeof(float));
    }

    // The maxLevels overload (atrac3denc.h:80-83): additionally max |new half| per band after gain modulation.
    void Mdct(float specs[1024], float* bands[4], float maxLevels[4], const TGainCurves& curves = TGainCurves())
    {
        float packed[4 * 512];
        int32_t n[4], level[32] = {0}, loc[32] = {0};
        bool any = false;
        for (int b = 0; b < 4; ++b) {
            memcpy(packed + 512 * b, bands[b], 512 * sizeof(float));
            n[b] = (int32_t)curves[b].size();
            any = any || n[b] > 0;
            for (int i = 0; i < n[b] && i < 8; ++i) {
                level[8 * b + i] = (int32_t)curves[b][i].Level;
                loc[8 * b + i] = (int32_t)curves[b][i].Location;
            }
        }
        Check(at3hip_mdct_levels(Ctx, packed, specs, maxLevels, any ? n : nullptr, any ? level : nullptr, any ? loc : nullptr, 1, 0),
              Ctx, "at3hip_mdct_levels");
        for (int b = 0; b < 4; ++b) memcpy(bands[b], packed + 512 * b, 512 * sizeof(float));
    }

    // TAtrac3MDCT::CalcGainEnergyScale (atrac3denc.h:69-79; static in the reference, a member here because the work
    // runs on this object's device context).
    struct TGainEnergyScale {
        float PrevHalf = 1.0f, CurHalf = 1.0f, Frame = 1.0f;
    };
    struct TGainEnergyAnalysis {
        TGainEnergyScale Scale;
        float NextOverlapScale = 1.0f;
    };
    TGainEnergyAnalysis CalcGainEnergyScale(const float prevOverlap[256], const float curInput[256],
                                            const std::vector<TGainPoint>& gainPoints, float prevOverlapScale)
    {
        int32_t n = (int32_t)gainPoints.size(), level[8] = {0}, loc[8] = {0};
        for (int i = 0; i < n && i < 8; ++i) {
            level[i] = (int32_t)gainPoints[i].Level;
            loc[i] = (int32_t)gainPoints[i].Location;
        }
        float out[4];
        Check(at3hip_gain_energy_scale(Ctx, prevOverlap, curInput, n ? &n : nullptr, n ? level : nullptr, n ? loc : nullptr,
                                       &prevOverlapScale, out, 1, 0), Ctx, "at3hip_gain_energy_scale");
        TGainEnergyAnalysis res;
        res.Scale.PrevHalf = out[0];
        res.Scale.CurHalf = out[1];
        res.Scale.Frame = out[2];
        res.NextOverlapScale = out[3];
        return res;
    }

private:
    at3hip_ctx* Ctx = nullptr;
};

// ---- n streams side by side: the batch form of TAtrac3Encoder --------------------------------------------
class TAtrac3EncoderBatch {
public:
    TAtrac3EncoderBatch(const TAtrac3EncoderSettings& s, int nStreams, int maxBlocks, int deviceId = 0)
        : NStreams(nStreams)
    {
        at3hip_config cfg{};
        cfg.bitrate = (int32_t)s.Bitrate;
        cfg.channels = s.SourceChannels;
        cfg.no_gain_control = s.NoGainControll;
        cfg.no_tonal = s.NoTonalComponents;
        cfg.bfu_idx_const = (int32_t)s.BfuIdxConst;
        cfg.n_streams = nStreams;
        cfg.max_blocks = maxBlocks;
        cfg.device_id = deviceId;
        CheckLibraryVersion();
        Check(at3hip_create(&cfg, &Ctx), nullptr, "at3hip_create");
        FrameSz = at3hip_frame_size(Ctx);
        // nothing in this layer reads the stage timings: their HIP events between the kernels cost a pipelined step ~3 % (AT3HIP_OPT_TIMING_EVERY)
        Check(at3hip_set_option(Ctx, AT3HIP_OPT_TIMING_EVERY, 0), Ctx, "at3hip_set_option");
    }
    // Stage timings (at3hip_get_timings / _ago on Handle()) on every Nth call with frames; 0 (this layer's default) = never.
    void SetTimingEvery(int n) { Check(at3hip_set_option(Ctx, AT3HIP_OPT_TIMING_EVERY, n), Ctx, "at3hip_set_option"); }
    ~TAtrac3EncoderBatch() { at3hip_destroy(Ctx); }
    TAtrac3EncoderBatch(const TAtrac3EncoderBatch&) = delete;
    TAtrac3EncoderBatch& operator=(const TAtrac3EncoderBatch&) = delete;

    int FrameSize() const { return FrameSz; }
    // pcm [nStreams][nBlocks][1024][SourceChannels] -> frames [nStreams][nFrames][FrameSize()]; returns nFrames per stream.
    int Encode(const float* pcm, int nBlocks, std::vector<uint8_t>& frames)
    {
        frames.resize((size_t)NStreams * nBlocks * FrameSz);
        int32_t nf = 0;
        Check(at3hip_encode(Ctx, pcm, nBlocks, frames.data(), &nf, 0), Ctx, "at3hip_encode");
        frames.resize((size_t)NStreams * nf * FrameSz);
        return nf;
    }
    // the same with 16-bit samples (converted s / 32768.0f on the device: a 16-bit WAV as sf_readf_float reads it)
    int EncodeS16(const int16_t* pcm, int nBlocks, std::vector<uint8_t>& frames)
    {
        frames.resize((size_t)NStreams * nBlocks * FrameSz);
        int32_t nf = 0;
        Check(at3hip_encode_s16(Ctx, pcm, nBlocks, frames.data(), &nf, 0), Ctx, "at3hip_encode_s16");
        frames.resize((size_t)NStreams * nf * FrameSz);
        return nf;
    }
    void Reset() { Check(at3hip_reset(Ctx), Ctx, "at3hip_reset"); }
    at3hip_ctx* Handle() { return Ctx; }
    // What the reference's TScaler::Scale would have written to stderr for these streams since construction / Reset()
    // ("Scale error: absSpec > MAX_SCALE" per block, "clipping, scaled value" per value; atrac_scale.cpp:150-153, 163-167)
    at3hip_counters Counters(bool reset = false)
    {
        at3hip_counters c{};
        Check(at3hip_get_counters(Ctx, &c, reset ? 1 : 0), Ctx, "at3hip_get_counters");
        return c;
    }

    // A long input fed call by call with the copies hidden: two page-locked PCM buffers and two frame buffers alternate, the
    // calls are asynchronous, so while the GPU encodes call k the host thread fills call k + 1's buffer (`fill`), the copy
    // engine moves it, and call k - 1's frames come back and are handed to `drain` - what TPCMEngine::ApplyProcess
    // (pcmengin.h:152-192) and ICompressedOutput::WriteFrame do around the reference's lambda, batched.
    //   fill(TSample* dst, int maxBlocks) -> blocks written, [nStreams][blocks][1024][channels]; 0 ends the input
    //   drain(const uint8_t* frames, int nFrames): [nStreams][nFrames][FrameSize()], in call order
    // TSample = float, or int16_t (EncodePipelinedS16): 16-bit samples cross the bus at half the bytes - the host-fed rate is
    // bound by exactly those - and become floats on the device.
    // Returns the number of frames per stream.
    template <class TFill, class TDrain>
    long long EncodePipelined(int blocksPerCall, int channels, TFill fill, TDrain drain)
    {
        return EncodePipelinedT<float>(blocksPerCall, channels, fill, drain);
    }
    template <class TFill, class TDrain>
    long long EncodePipelinedS16(int blocksPerCall, int channels, TFill fill, TDrain drain)
    {
        return EncodePipelinedT<int16_t>(blocksPerCall, channels, fill, drain);
    }

private:
    static int EncodeAny(at3hip_ctx* c, const float* pcm, int32_t nb, uint8_t* out, int32_t* nf, uint32_t flags) { return at3hip_encode(c, pcm, nb, out, nf, flags); }
    static int EncodeAny(at3hip_ctx* c, const int16_t* pcm, int32_t nb, uint8_t* out, int32_t* nf, uint32_t flags) { return at3hip_encode_s16(c, pcm, nb, out, nf, flags); }
    template <class TSample, class TFill, class TDrain>
    long long EncodePipelinedT(int blocksPerCall, int channels, TFill fill, TDrain drain)
    {
        struct TPinned {
            at3hip_ctx* Ctx;
            void* P = nullptr;
            TPinned(at3hip_ctx* c, size_t bytes) : Ctx(c) { Check(at3hip_host_alloc(c, bytes, &P), c, "at3hip_host_alloc"); }
            ~TPinned() { at3hip_host_free(Ctx, P); }
        };
        const size_t inFloats = (size_t)NStreams * blocksPerCall * 1024 * channels, outBytes = (size_t)NStreams * blocksPerCall * FrameSz;
        // kDepth calls in flight (at3hip_wait_* reach three calls back, so at most four). Measured on configs[1] (64 x 64 frames per
        // call, profiles/EXPERIMENTS.md round 5): 16-bit samples - the copy of a call is as short as its kernels, and the device
        // overlaps three stages of consecutive calls - 9.6 / 11.5 / 12.4 M frames/s at two / three / four calls in flight (92 % of
        // the bus with four); float samples are bound by the bus at any depth (6.6 / 6.5 / 6.5 M = 95 % of it), so they keep two
        // calls and 64 MB less page-locked memory.
        constexpr int kDepth = sizeof(TSample) == 2 ? 4 : 2;
        std::unique_ptr<TPinned> inBuf[kDepth], outBuf[kDepth];
        TSample* in[kDepth];
        uint8_t* out[kDepth];
        for (int q = 0; q < kDepth; ++q) {
            inBuf[q].reset(new TPinned(Ctx, inFloats * sizeof(TSample)));
            outBuf[q].reset(new TPinned(Ctx, outBytes));
            in[q] = (TSample*)inBuf[q]->P;
            out[q] = (uint8_t*)outBuf[q]->P;
        }
        int32_t nf[kDepth] = {};
        long long total = 0;
        int call = 0, drained = 0;   // calls queued / calls whose frames were handed over
        auto drain_next = [&] {
            const int q = drained % kDepth;
            if (nf[q] > 0) drain(out[q], (int)nf[q]);
            total += nf[q];
            ++drained;
        };
        try {
        for (;; ++call) {
            const int q = call % kDepth;
            if (call >= kDepth) Check(at3hip_wait_input(Ctx, kDepth - 1), Ctx, "at3hip_wait_input");   // call - kDepth read in[q]: gone to the device by now?
            const int nb = fill(in[q], blocksPerCall);
            if (nb <= 0) break;
            // (out[q] is free: the frames of call - kDepth were drained below, after call - 1 was queued)
            Check(EncodeAny(Ctx, in[q], nb, out[q], &nf[q], AT3HIP_ASYNC), Ctx, "at3hip_encode");
            if (call >= kDepth - 1) {   // while the newer calls run: the oldest call's frames
                Check(at3hip_wait_frames(Ctx, kDepth - 1), Ctx, "at3hip_wait_frames");
                drain_next();
            }
        }
        if (call >= 1) {
            Check(at3hip_sync(Ctx), Ctx, "at3hip_sync");
            while (drained < call) drain_next();
        }
        } catch (...) {
            at3hip_sync(Ctx);   // the page-locked buffers are released on the way out: no copy may still be in flight
            throw;
        }
        return total;
    }

    at3hip_ctx* Ctx = nullptr;
    int NStreams;
    int FrameSz = 0;
};

// ---- all GPUs of a node: the stream-sharded form --------------------------------------------------------------
// Contiguous, balanced partition of `total` streams over `parts` devices: (first, count) of part `idx`.
inline std::pair<int, int> ShardStreams(int total, int parts, int idx)
{
    if (parts < 1 || idx < 0 || idx >= parts) throw std::runtime_error("ShardStreams: bad partition");
    const int base = total / parts, rem = total % parts;
    return {idx * base + (idx < rem ? idx : rem), base + (idx < rem ? 1 : 0)};
}

// Pin the calling thread to the host NUMA node of a device (at3hip_device_numa_node; the node's CPUs from
// /sys/devices/system/node/node<N>/cpulist). Memory the thread touches first afterwards - the page-locked staging buffers of
// TAtrac3EncoderBatch::EncodePipelined among it - then lies on that node. Returns the node, or -1 when nothing was changed (unknown
// node, no sysfs, not Linux): a missing pin costs bandwidth on a multi-socket host, never correctness.
inline int PinThreadToDeviceNode(int deviceId)
{
#if defined(__linux__)
    const int node = at3hip_device_numa_node(deviceId);
    if (node < 0) return -1;
    char path[96];
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r");
    if (!f) return -1;
    cpu_set_t set;
    CPU_ZERO(&set);
    int n = 0, a = 0, b = 0;
    for (;;) {   // "0-15,128-143"
        if (fscanf(f, "%d", &a) != 1) break;
        b = a;
        int c = fgetc(f);
        if (c == '-') {
            if (fscanf(f, "%d", &b) != 1) break;
            c = fgetc(f);
        }
        for (int cpu = a; cpu <= b && cpu < CPU_SETSIZE; ++cpu) {
            CPU_SET(cpu, &set);
            ++n;
        }
        if (c != ',') break;
    }
    fclose(f);
    if (n == 0 || sched_setaffinity(0, sizeof(set), &set) != 0) return -1;
    return node;
#else
    (void)deviceId;
    return -1;
#endif
}

class TAtrac3EncoderNode {
public:
    // deviceIds: the HIP ordinals to use (e.g. {0,1,...,7}); streams [first, first + count) of ShardStreams go to deviceIds[i]
    TAtrac3EncoderNode(const TAtrac3EncoderSettings& s, int nStreams, int maxBlocks, const std::vector<int>& deviceIds)
        : NStreams(nStreams), Channels(s.SourceChannels)
    {
        if (deviceIds.empty() || nStreams < (int)deviceIds.size()) throw std::runtime_error("TAtrac3EncoderNode: need >= 1 stream per device");
        for (size_t i = 0; i < deviceIds.size(); ++i) {
            const auto part = ShardStreams(nStreams, (int)deviceIds.size(), (int)i);
            First.push_back(part.first);
            Count.push_back(part.second);
            Parts.emplace_back(new TAtrac3EncoderBatch(s, part.second, maxBlocks, deviceIds[i]));
            DeviceIds.push_back(deviceIds[i]);
        }
        FrameSz = Parts[0]->FrameSize();
    }
    int FrameSize() const { return FrameSz; }
    int Devices() const { return (int)Parts.size(); }
    // pcm [nStreams][nBlocks][1024][SourceChannels] -> frames [nStreams][nFrames][FrameSize()]; returns nFrames per stream.
    // Every device encodes its slice on its own host thread; the slices are disjoint in both buffers.
    int Encode(const float* pcm, int nBlocks, std::vector<uint8_t>& frames)
    {
        const size_t blockFloats = (size_t)1024 * Channels;
        std::vector<std::vector<uint8_t>> out(Parts.size());
        std::vector<int> nf(Parts.size(), 0);
        std::vector<std::string> err(Parts.size());
        std::vector<std::thread> th;
        for (size_t i = 0; i < Parts.size(); ++i)
            th.emplace_back([&, i] {
                try {
                    PinThreadToDeviceNode(DeviceIds[i]);   // (the device's feeder thread and what it allocates: on the device's NUMA node)
                    nf[i] = Parts[i]->Encode(pcm + (size_t)First[i] * nBlocks * blockFloats, nBlocks, out[i]);
                } catch (const std::exception& e) {
                    err[i] = e.what();
                }
            });
        for (auto& t : th) t.join();
        for (size_t i = 0; i < Parts.size(); ++i)
            if (!err[i].empty()) throw std::runtime_error("TAtrac3EncoderNode device " + std::to_string(i) + ": " + err[i]);
        frames.resize((size_t)NStreams * nf[0] * FrameSz);
        for (size_t i = 0; i < Parts.size(); ++i)
            memcpy(frames.data() + (size_t)First[i] * nf[0] * FrameSz, out[i].data(), out[i].size());
        return nf[0];
    }
    void Reset()
    {
        for (auto& p : Parts) p->Reset();
    }

    // The same for a long input: pcm [nStreams][nBlocksTotal][1024][SourceChannels] is cut into calls of `blocksPerCall`
    // blocks; every device's host thread runs TAtrac3EncoderBatch::EncodePipelined on its slice (page-locked staging, copies
    // and kernels of consecutive calls overlapping). frames [nStreams][nFrames][FrameSize()]; returns nFrames per stream.
    // STARTS A NEW STREAM: the node is Reset() first - unlike TAtrac3EncoderBatch::EncodePipelined, which continues whatever its
    // context has been fed - because `frames` is laid out for exactly nBlocksTotal - 1 frames per stream, which only holds from
    // start of stream. Blocks fed through Encode() before this call are therefore NOT continued; callers that feed one stream in
    // pieces use Encode() throughout, or TAtrac3EncoderBatch::EncodePipelined on a part. `frames` is resized before any frame
    // arrives, so when a device's thread throws (e.g. the frame-count guard below) its content is unspecified.
    int EncodePipelined(const float* pcm, int nBlocksTotal, int blocksPerCall, std::vector<uint8_t>& frames)
    {
        const size_t blockFloats = (size_t)1024 * Channels;
        // start of stream: the first block is the look-ahead (atrac3denc.cpp:715-718), so nBlocksTotal blocks give
        // nBlocksTotal - 1 frames per stream - the node is reset here so that this holds whatever was encoded before
        Reset();
        const int nfTotal = nBlocksTotal - 1;
        frames.assign((size_t)NStreams * (nfTotal > 0 ? nfTotal : 0) * FrameSz, 0);
        std::vector<std::string> err(Parts.size());
        std::vector<long long> got(Parts.size(), 0);
        std::vector<std::thread> th;
        for (size_t i = 0; i < Parts.size(); ++i)
            th.emplace_back([&, i] {
                try {
                    PinThreadToDeviceNode(DeviceIds[i]);   // before EncodePipelined allocates its page-locked staging: first touch on the device's node
                    int fed = 0, written = 0;
                    got[i] = Parts[i]->EncodePipelined(
                        blocksPerCall, Channels,
                        [&](float* dst, int maxBlocks) {
                            const int nb = std::min(maxBlocks, nBlocksTotal - fed);
                            for (int s = 0; s < Count[i] && nb > 0; ++s)
                                memcpy(dst + (size_t)s * nb * blockFloats, pcm + ((size_t)(First[i] + s) * nBlocksTotal + fed) * blockFloats,
                                       (size_t)nb * blockFloats * sizeof(float));
                            fed += nb > 0 ? nb : 0;
                            return nb;
                        },
                        [&](const uint8_t* fr, int nf) {
                            if (written + nf > nfTotal) throw std::runtime_error("EncodePipelined: more frames than the output holds");
                            for (int s = 0; s < Count[i]; ++s)
                                memcpy(frames.data() + ((size_t)(First[i] + s) * nfTotal + written) * FrameSz, fr + (size_t)s * nf * FrameSz,
                                       (size_t)nf * FrameSz);
                            written += nf;
                        });
                } catch (const std::exception& e) {
                    err[i] = e.what();
                }
            });
        for (auto& t : th) t.join();
        for (size_t i = 0; i < Parts.size(); ++i)
            if (!err[i].empty()) throw std::runtime_error("TAtrac3EncoderNode device " + std::to_string(i) + ": " + err[i]);
        return (int)got[0];
    }

private:
    std::vector<std::unique_ptr<TAtrac3EncoderBatch>> Parts;
    std::vector<int> First, Count, DeviceIds;
    int NStreams;
    int Channels;
    int FrameSz = 0;
};

// ---- TAtrac3Encoder (atrac3denc.h:94-134): lambda in, WriteFrame out -------------------------------------
class TAtrac3Encoder {
public:
    TAtrac3Encoder(TCompressedOutputPtr&& oma, TAtrac3EncoderSettings&& settings, int batchBlocks = 64, int deviceId = 0)
        : Oma(std::move(oma)), Params(settings), BatchBlocks(batchBlocks), Batch(settings, 1, batchBlocks, deviceId),
          BlockFloats(1024u * settings.SourceChannels)
    {
        // one input channel is accepted for the discrete-stereo bitrates (the frame holds the unit twice,
        // atrac3_bitstream.cpp:836-843); at3hip_create refuses the other combinations
        Pending.reserve((size_t)BatchBlocks * BlockFloats);
    }
    ~TAtrac3Encoder()
    {
        try {
            Flush();
        } catch (...) {
        }
    }

    TProcessLambda GetLambda()
    {
        return [this](float* data, const ProcessMeta& meta) {
            if (meta.Channels != Params.SourceChannels) throw std::runtime_error("TAtrac3Encoder(hip): channel count changed");
            Pending.insert(Pending.end(), data, data + BlockFloats);
            const bool first = (Calls++ == 0);
            if ((int)(Pending.size() / BlockFloats) == BatchBlocks) Flush();
            return first ? EProcessResult::LOOK_AHEAD : EProcessResult::PROCESSED;   // atrac3denc.cpp:715-718, 865
        };
    }

    // Encode what is buffered and hand the frames to the sink in order.
    void Flush()
    {
        const int nb = (int)(Pending.size() / BlockFloats);
        if (nb == 0) return;
        std::vector<uint8_t> frames;
        const int nf = Batch.Encode(Pending.data(), nb, frames);
        Pending.clear();
        const int fsz = Batch.FrameSize();
        for (int i = 0; i < nf; ++i)
            Oma->WriteFrame(std::vector<char>(frames.begin() + (size_t)i * fsz, frames.begin() + (size_t)(i + 1) * fsz));
    }

private:
    TCompressedOutputPtr Oma;
    const TAtrac3EncoderSettings Params;
    const int BatchBlocks;
    TAtrac3EncoderBatch Batch;
    const size_t BlockFloats;
    std::vector<float> Pending;
    uint64_t Calls = 0;
};

// ---- ATRAC1 -----------------------------------------------------------------------
// NAtrac1::TAtrac1EncodeSettings (atrac/at1/atrac1.h:33-54)
struct TAtrac1EncodeSettings {
    enum class EWindowMode { EWM_NOTRANSIENT, EWM_AUTO };
    uint32_t BfuIdxConst = 0;
    EWindowMode WindowMode = EWindowMode::EWM_AUTO;
    uint32_t WindowMask = 0;
    TAtrac1EncodeSettings() = default;
    TAtrac1EncodeSettings(uint32_t bfuIdxConst, EWindowMode windowMode, uint32_t windowMask)
        : BfuIdxConst(bfuIdxConst), WindowMode(windowMode), WindowMask(windowMask)
    {
    }
};

inline void Check1(int rc, at1hip_ctx* ctx, const char* what)
{
    if (rc != AT3HIP_OK) throw std::runtime_error(std::string(what) + ": " + (ctx ? at1hip_last_error(ctx) : "error " + std::to_string(rc)));
}

// TAtrac1Encoder (atrac1denc.h:56-110): lambda in - one call per 512-sample block - and one WriteFrame per channel
// and block out, channel 0 first (atrac1denc.cpp:249-251). The channel count comes from the sink, as in the reference.
// Calls are buffered `batchBlocks` at a time like TAtrac3Encoder above; there is no look-ahead call in this codec.
class TAtrac1Encoder {
public:
    TAtrac1Encoder(TCompressedOutputPtr&& aea, TAtrac1EncodeSettings&& settings, int batchBlocks = 512, int deviceId = 0)
        : Aea(std::move(aea)), Settings(settings), BatchBlocks(batchBlocks), Channels(Aea->GetChannelNum()), BlockFloats(512u * Channels)
    {
        at1hip_config cfg{};
        cfg.channels = (int32_t)Channels;
        cfg.window_auto = Settings.WindowMode == TAtrac1EncodeSettings::EWindowMode::EWM_AUTO;
        cfg.window_mask = (int32_t)Settings.WindowMask;
        cfg.bfu_idx_const = (int32_t)Settings.BfuIdxConst;
        cfg.n_streams = 1;
        cfg.max_blocks = batchBlocks;
        cfg.device_id = deviceId;
        Check1(at1hip_create(&cfg, &Ctx), nullptr, "at1hip_create");
        Pending.reserve((size_t)BatchBlocks * BlockFloats);
    }
    ~TAtrac1Encoder()
    {
        try {
            Flush();
        } catch (...) {
        }
        at1hip_destroy(Ctx);
    }
    TAtrac1Encoder(const TAtrac1Encoder&) = delete;
    TAtrac1Encoder& operator=(const TAtrac1Encoder&) = delete;

    TProcessLambda GetLambda()
    {
        return [this](float* data, const ProcessMeta&) {
            Pending.insert(Pending.end(), data, data + BlockFloats);
            if ((int)(Pending.size() / BlockFloats) == BatchBlocks) Flush();
            return EProcessResult::PROCESSED;
        };
    }

    void Flush()
    {
        const int nb = (int)(Pending.size() / BlockFloats);
        if (nb == 0) return;
        std::vector<uint8_t> units((size_t)nb * Channels * AT1HIP_FRAME_SIZE);
        Check1(at1hip_encode(Ctx, Pending.data(), nb, units.data(), 0), Ctx, "at1hip_encode");
        Pending.clear();
        for (size_t i = 0; i < (size_t)nb * Channels; ++i)
            Aea->WriteFrame(std::vector<char>(units.begin() + i * AT1HIP_FRAME_SIZE, units.begin() + (i + 1) * AT1HIP_FRAME_SIZE));
    }

private:
    TCompressedOutputPtr Aea;
    const TAtrac1EncodeSettings Settings;
    const int BatchBlocks;
    const size_t Channels;
    const size_t BlockFloats;
    at1hip_ctx* Ctx = nullptr;
    std::vector<float> Pending;
};

// ---- ATRAC3plus ---------------------------------------------------------------------------------------------------
// TAt3PEnc (at3p.h, at3p.cpp:37-191) with GHA_PASS_INPUT | GHA_WRITE_RESIUDAL and a tonal analysis that finds nothing
// (the analysis itself needs libgha, which the reference does not vendor): lambda in - one call per 2048-sample frame of
// interleaved PCM - and one 2048-byte WriteFrame out per call after the first. The reference looks one frame ahead and
// encodes the frame BEFORE the current one (PrevBuf, at3p.cpp:115-160), so the first call returns LOOK_AHEAD, the second
// writes a silent frame and call k >= 2 writes input frame k - 2; this class keeps that schedule. Frames are encoded
// `batchFrames` at a time like the encoders above and flushed in order.
class TAt3PEncoder {
public:
    TAt3PEncoder(TCompressedOutputPtr&& out, int channels, int batchFrames = 64, int deviceId = 0)
        : Out(std::move(out)), Channels((size_t)channels), BatchFrames(batchFrames), FrameFloats((size_t)AT3PHIP_FRAME * (size_t)channels)
    {
        at3phip_config cfg{};
        cfg.channels = channels;
        cfg.n_streams = 1;
        cfg.max_frames = batchFrames;
        cfg.device_id = deviceId;
        const int rc = at3phip_create(&cfg, &Ctx);
        if (rc != AT3HIP_OK) throw std::runtime_error("at3phip_create failed: " + std::to_string(rc));
        Pending.reserve((size_t)BatchFrames * FrameFloats);
    }
    ~TAt3PEncoder()
    {
        try {
            Flush();
        } catch (...) {
        }
        at3phip_destroy(Ctx);
    }
    TAt3PEncoder(const TAt3PEncoder&) = delete;
    TAt3PEncoder& operator=(const TAt3PEncoder&) = delete;

    TProcessLambda GetLambda()
    {
        return [this](float* data, const ProcessMeta&) {
            const bool first = Calls == 0;
            ++Calls;
            Pending.insert(Pending.end(), data, data + FrameFloats);
            if ((int)(Pending.size() / FrameFloats) == BatchFrames) Flush();
            return first ? EProcessResult::LOOK_AHEAD : EProcessResult::PROCESSED;
        };
    }

    // Encodes what is buffered and writes the frames that are due: after n calls, n - 1 frames have been written.
    void Flush()
    {
        const int nf = (int)(Pending.size() / FrameFloats);
        if (nf > 0) {
            std::vector<uint8_t> frames((size_t)nf * AT3PHIP_FRAME_BYTES);
            Chk(at3phip_encode_frames(Ctx, Pending.data(), nf, frames.data(), 0), "at3phip_encode_frames");
            Pending.clear();
            for (int i = 0; i < nf; ++i) Ready.emplace_back(frames.begin() + (size_t)i * AT3PHIP_FRAME_BYTES, frames.begin() + (size_t)(i + 1) * AT3PHIP_FRAME_BYTES);
        }
        // call k (0-based) is answered with: nothing (k = 0), silence (k = 1), input frame k - 2
        while (Written + 1 < Calls) {
            if (Written == 0) {
                Out->WriteFrame(SilentFrame());
            } else {
                Out->WriteFrame(std::move(Ready.front()));
                Ready.erase(Ready.begin());
            }
            ++Written;
        }
    }

private:
    void Chk(int rc, const char* what)
    {
        if (rc != AT3HIP_OK) throw std::runtime_error(std::string(what) + ": " + at3phip_last_error(Ctx));
    }
    std::vector<char> SilentFrame()   // an all-zero spectrum through the frame writer (the encoder's PrevBuf starts zeroed)
    {
        std::vector<float> specs(FrameFloats, 0.0f);
        std::vector<uint8_t> frame(AT3PHIP_FRAME_BYTES);
        Chk(at3phip_write_frames(Ctx, specs.data(), 1, nullptr, frame.data(), 0), "at3phip_write_frames");
        return std::vector<char>(frame.begin(), frame.end());
    }
    TCompressedOutputPtr Out;
    const size_t Channels;
    const int BatchFrames;
    const size_t FrameFloats;
    at3phip_ctx* Ctx = nullptr;
    std::vector<float> Pending;
    std::vector<std::vector<char>> Ready;
    size_t Calls = 0, Written = 0;
};

}  // namespace NAtracDEncHip
