// TEST INFRASTRUCTURE: C entry points around atracdenc_amd/host/at3hip_io.hpp so the Python tests can drive the host
// IO layer (containers, frame schedule, WAV reader) next to the reference's own classes (oracle/_ref). No GPU needed.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../atracdenc_amd/host/at3hip_io.hpp"

using namespace NAtracDEncHip;

extern "C" {

int at3host_write_container(int kind, const char* path, const uint8_t* frames, int n_frames, int frame_sz, int js,
                            int num_frames_hint, int nch)
{
    try {
        TCompressedOutputPtr out = kind >= 5 ? CreateAtrac3PlusOutput(kind == 6 ? EContainer::RIFF : kind == 7 ? EContainer::RAW : EContainer::OMA, path,
                                                                      (size_t)nch, (uint32_t)num_frames_hint, (uint32_t)frame_sz)
                                 : kind >= 3 ? CreateAtrac1Output(kind == 3 ? EContainer::AEA : EContainer::RAW, path, (size_t)nch, (uint32_t)num_frames_hint)
                                             : CreateAtrac3Output(kind == 1 ? EContainer::RIFF : kind == 2 ? EContainer::RAW : EContainer::OMA, path,
                                                      (size_t)nch, (uint32_t)num_frames_hint, (uint32_t)frame_sz, js != 0);
        for (int i = 0; i < n_frames; ++i)
            out->WriteFrame(std::vector<char>(frames + (size_t)i * frame_sz, frames + (size_t)(i + 1) * frame_sz));
    } catch (const std::exception&) {
        return -1;
    }
    return 0;
}

int at3host_select_container(const char* out_file)
{
    try {
        return (int)SelectAtrac3Container(out_file) == (int)EContainer::RIFF ? 1 : (int)SelectAtrac3Container(out_file) == (int)EContainer::RAW ? 2 : 0;
    } catch (const std::exception&) {
        return -1;
    }
}

// Same contract as at3ref_engine_trace_step in oracle/ref/ref_harness.cpp.
int at3host_engine_trace_step(uint64_t total_samples, int nch, int step, int look_ahead, float* first_vals, float* last_vals, int max_calls,
                              uint64_t* processed_out, float* tail)
{
    uint64_t pos = 0;
    TPCMEngine engine(4096, (size_t)nch, [&](float* dst, size_t frames) -> size_t {
        uint64_t n = total_samples - pos;
        if (n > frames) n = frames;
        for (uint64_t i = 0; i < n; ++i)
            for (int c = 0; c < nch; ++c) dst[i * nch + c] = (float)(pos + i + 1);
        pos += n;
        return (size_t)n;
    });
    int calls = 0;
    TProcessLambda lambda = [&](float* data, const ProcessMeta& meta) {
        if (calls < max_calls) {
            first_vals[calls] = data[0];
            last_vals[calls] = data[(size_t)(step - 1) * meta.Channels];
        }
        if (tail) memcpy(tail, data, sizeof(float) * step * meta.Channels);
        return (calls++ == 0 && look_ahead) ? EProcessResult::LOOK_AHEAD : EProcessResult::PROCESSED;
    };
    uint64_t processed = 0;
    try {
        while (total_samples > (processed = engine.ApplyProcess((size_t)step, lambda))) {
        }
    } catch (const TNoDataToRead&) {
        *processed_out = processed;
        return -1;
    }
    *processed_out = processed;
    return calls;
}

int at3host_engine_trace(uint64_t total_samples, int nch, float* first_vals, float* last_vals, int max_calls, uint64_t* processed_out,
                         float* tail)
{
    return at3host_engine_trace_step(total_samples, nch, 1024, 1, first_vals, last_vals, max_calls, processed_out, tail);
}

// Reads a WAV file through TWavSource + TPCMEngine exactly as at3hipenc does and returns every block handed to the
// encoder lambda: blocks [max_blocks][step][channels]. Returns the number of blocks, info = {channels, rate, total}.
int at3host_wav_blocks_step(const char* path, float* blocks, int max_blocks, uint64_t* info, int step, int look_ahead)
{
    try {
        TWavSource wav(path);
        const size_t nch = wav.GetChannelNum();
        info[0] = nch;
        info[1] = wav.GetSampleRate();
        info[2] = wav.GetTotalSamples();
        TPCMEngine engine(4096, nch, [&wav](float* dst, size_t frames) { return wav.Read(dst, frames); });
        int calls = 0;
        TProcessLambda lambda = [&](float* data, const ProcessMeta& meta) {
            if (calls < max_blocks) memcpy(blocks + (size_t)calls * step * meta.Channels, data, sizeof(float) * step * meta.Channels);
            return (calls++ == 0 && look_ahead) ? EProcessResult::LOOK_AHEAD : EProcessResult::PROCESSED;
        };
        const uint64_t total = wav.GetTotalSamples();
        try {
            while (total > engine.ApplyProcess((size_t)step, lambda)) {
            }
        } catch (const TNoDataToRead&) {
        }
        return calls;
    } catch (const std::exception&) {
        return -1;
    }
}

int at3host_wav_blocks(const char* path, float* blocks, int max_blocks, uint64_t* info)
{
    return at3host_wav_blocks_step(path, blocks, max_blocks, info, 1024, 1);
}

}  // extern "C"
