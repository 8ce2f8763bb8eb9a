// Callers and data formats either side of the ATRAC3 encode hot path, written from
// scratch for the host side of libat3hip:
//
//   TWavSource          RIFF/WAVE PCM reader, samples normalised the way the reference's reader hands them to
//                       the encoder (pcm_io_sndfile.cpp reads with sf_readf_float: integer PCM / 2^(bits-1))
//   TPCMEngine          the frame schedule of TPCMEngine::ApplyProcess with a reader (pcmengin.h:152-192):
//                       4096-sample reads, the (partially) cleared tail, the LOOK_AHEAD first call, and the drain call that
//                       re-presents the stale head of the last buffer - the file-level behaviour that decides
//                       how many frames an encode produces and what the last look-ahead block contains
//   TOmaOutput          OMA container (oma.cpp:26-52, lib/liboma/src/liboma.c:155-236: 96-byte EA3 header)
//   TAt3RiffOutput      ATRAC3-in-WAV container (at3.cpp:38-262: 76-byte header, lengths back-filled on close)
//   TRawOutput          bare frames (raw.cpp:27-57)
//   SelectAtrac3Container  extension rule of main.cpp:207-220
//   TAeaOutput          ATRAC1's AEA container (aea.cpp:120-189); SelectAtrac1Container main.cpp:196-205
//
// RealMedia output (rm.cpp) is not built. Everything here is plain host C++ over at3hip_host.hpp.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "at3hip_host.hpp"

namespace NAtracDEncHip {

// ---- WAV input ----------------------------------------------------------------------------------------------
class TWavSource {
public:
    explicit TWavSource(const std::string& path) : Fp(fopen(path.c_str(), "rb"))
    {
        if (!Fp) throw std::runtime_error("unable to open input file '" + path + "'");
        uint8_t hdr[12];
        if (fread(hdr, 1, 12, Fp) != 12 || memcmp(hdr, "RIFF", 4) || memcmp(hdr + 8, "WAVE", 4)) Fail("not a RIFF/WAVE file");
        bool haveFmt = false;
        for (;;) {
            uint8_t ck[8];
            if (fread(ck, 1, 8, Fp) != 8) Fail("no data chunk");
            const uint32_t sz = Le32(ck + 4);
            if (!memcmp(ck, "fmt ", 4)) {
                uint8_t f[40] = {0};
                const uint32_t take = sz < sizeof(f) ? sz : (uint32_t)sizeof(f);
                if (sz < 16 || fread(f, 1, take, Fp) != take) Fail("bad fmt chunk");
                Skip(sz - take + (sz & 1));
                Format = Le16(f);
                NumChannels = Le16(f + 2);
                Rate = Le32(f + 4);
                Bits = Le16(f + 14);
                if (Format == 0xFFFE && sz >= 26) Format = Le16(f + 24);   // WAVE_FORMAT_EXTENSIBLE: sub-format GUID head
                haveFmt = true;
            } else if (!memcmp(ck, "data", 4)) {
                if (!haveFmt) Fail("data chunk before fmt chunk");
                DataBytes = sz;
                break;
            } else {
                Skip(sz + (sz & 1));
            }
        }
        if (NumChannels < 1 || NumChannels > 2) Fail("1 or 2 channels expected");
        if (!((Format == 1 && (Bits == 8 || Bits == 16 || Bits == 24 || Bits == 32)) || (Format == 3 && Bits == 32)))
            Fail("unsupported sample format");
        BytesPerFrame = (size_t)NumChannels * (Bits / 8);
        Frames = DataBytes / BytesPerFrame;
    }
    ~TWavSource()
    {
        if (Fp) fclose(Fp);
    }
    TWavSource(const TWavSource&) = delete;
    TWavSource& operator=(const TWavSource&) = delete;

    size_t GetChannelNum() const { return NumChannels; }
    size_t GetSampleRate() const { return Rate; }
    uint64_t GetTotalSamples() const { return Frames; }

    // Up to `frames` sample frames as interleaved floats; returns the number delivered (0 at end of data).
    size_t Read(float* dst, size_t frames)
    {
        const uint64_t left = Frames - Pos;
        if (frames > left) frames = (size_t)left;
        if (!frames) return 0;
        Raw.resize(frames * BytesPerFrame);
        const size_t got = fread(Raw.data(), BytesPerFrame, frames, Fp);
        const size_t n = got * NumChannels;
        const uint8_t* r = Raw.data();
        if (Format == 3) {
            memcpy(dst, r, n * sizeof(float));
        } else if (Bits == 16) {
            for (size_t i = 0; i < n; ++i) dst[i] = (float)(int16_t)Le16(r + 2 * i) / 32768.0f;
        } else if (Bits == 24) {
            for (size_t i = 0; i < n; ++i) {
                const int32_t v = (int32_t)((uint32_t)r[3 * i] << 8 | (uint32_t)r[3 * i + 1] << 16 | (uint32_t)r[3 * i + 2] << 24);
                dst[i] = (float)v / 2147483648.0f;   // the 24-bit value sits in the top of a 32-bit word
            }
        } else if (Bits == 32) {
            for (size_t i = 0; i < n; ++i) dst[i] = (float)(int32_t)Le32(r + 4 * i) / 2147483648.0f;
        } else {
            for (size_t i = 0; i < n; ++i) dst[i] = (float)((int)r[i] - 128) / 128.0f;
        }
        Pos += got;
        return got;
    }

private:
    static uint16_t Le16(const uint8_t* p) { return (uint16_t)(p[0] | p[1] << 8); }
    static uint32_t Le32(const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
    void Skip(uint32_t n)
    {
        if (n && fseek(Fp, (long)n, SEEK_CUR) != 0) Fail("truncated file");
    }
    [[noreturn]] void Fail(const char* what)
    {
        fclose(Fp);
        Fp = nullptr;
        throw std::runtime_error(std::string("WAV: ") + what);
    }

    FILE* Fp;
    uint16_t Format = 0, NumChannels = 0, Bits = 0;
    uint32_t Rate = 0, DataBytes = 0;
    size_t BytesPerFrame = 0;
    uint64_t Frames = 0, Pos = 0;
    std::vector<uint8_t> Raw;
};

// ---- the reader-driven frame schedule -------------------------------------------------------------------------
struct TNoDataToRead : std::exception {};

class TPCMEngine {
public:
    // reader(dst, frames) -> frames delivered (interleaved floats); 0 = end of data
    using TReader = std::function<size_t(float*, size_t)>;

    TPCMEngine(uint16_t bufSize, size_t numChannels, TReader reader)
        : Buf((size_t)bufSize * numChannels, 0.0f), BufFrames(bufSize), Channels((uint16_t)numChannels), Reader(std::move(reader))
    {
    }

    // One read of the whole buffer, then one lambda call per `step` frames (pcmengin.h:152-192). The tail after a
    // short read is cleared the way the reference clears it; an empty read is an error unless a look-ahead call is still owed, in which case exactly one
    // more call is made on the buffer as the previous read left it.
    uint64_t ApplyProcess(size_t step, const TProcessLambda& lambda)
    {
        if (step > BufFrames) throw std::runtime_error("PCM buffer too small");
        bool drain = false;
        const size_t got = Reader(Buf.data(), BufFrames);
        if (got == 0) {
            if (!ToDrain) throw TNoDataToRead();
            drain = true;
        } else if (got < BufFrames) {
            // The reference clears the unread tail with a BYTE count equal to the number of missing floats
            // (TPCMBuffer::Zero, pcmengin.h:93-96: memset(..., len * NumChannels)), i.e. only the first quarter of the
            // tail becomes zero and the rest keeps the samples of the previous read. Files only match if this does too.
            memset(reinterpret_cast<char*>(Buf.data() + got * Channels), 0, (BufFrames - got) * Channels);
        }
        size_t lastPos = 0;
        const ProcessMeta meta = {Channels};
        for (size_t i = 0; i + step <= BufFrames; i += step) {
            if (lambda(Buf.data() + i * Channels, meta) == EProcessResult::PROCESSED) {
                lastPos += step;
                if (drain && ToDrain--) break;
            } else {
                ++ToDrain;
            }
        }
        Processed += lastPos;
        return Processed;
    }

private:
    std::vector<float> Buf;
    size_t BufFrames;
    uint16_t Channels;
    TReader Reader;
    uint64_t Processed = 0, ToDrain = 0;
};

// ---- containers -------------------------------------------------------------------------------------------------
enum class EContainer { OMA, RIFF, RAW, AEA };

inline EContainer SelectAtrac3Container(const std::string& outFile)   // main.cpp:207-220 (AUTO)
{
    std::string ext;
    const size_t dot = outFile.find_last_of('.');
    if (dot != std::string::npos) ext = outFile.substr(dot + 1);
    for (char& ch : ext)
        if (ch >= 'A' && ch <= 'Z') ch = (char)(ch - 'A' + 'a');
    if (ext == "wav" || ext == "at3") return EContainer::RIFF;
    if (ext == "raw" || ext == "dat") return EContainer::RAW;
    if (ext == "rm") throw std::runtime_error("RealMedia output is not built");
    return EContainer::OMA;
}

class TFileOutput : public ICompressedOutput {
public:
    TFileOutput(const std::string& filename, size_t channels) : Fp(fopen(filename.c_str(), "wb")), Channels(channels)
    {
        if (!Fp) throw std::runtime_error("unable to open output file '" + filename + "'");
    }
    ~TFileOutput() override
    {
        if (Fp) fclose(Fp);
    }
    std::string GetName() const override { return {}; }
    size_t GetChannelNum() const override { return Channels; }

protected:
    void Put(const void* p, size_t n, const char* what)
    {
        if (fwrite(p, 1, n, Fp) != n) throw std::runtime_error(what);
    }
    static void Le16(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
    static void Le32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
    FILE* Fp;
    size_t Channels;
};

// 96-byte header: "EA3", 1, 0, 96, 0xFF, 0xFF; big-endian codec word at 32 = id << 24 | js << 17 | rate index << 13 |
// FrameSz / 8 (44100 Hz = index 1); frames follow verbatim.
class TOmaOutput : public TFileOutput {
public:
    TOmaOutput(const std::string& filename, uint32_t frameSize, bool jointStereo) : TFileOutput(filename, 2), FrameSize(frameSize)
    {
        uint8_t h[96] = {0};
        h[0] = 'E'; h[1] = 'A'; h[2] = '3'; h[3] = 1; h[5] = 96; h[6] = 0xFF; h[7] = 0xFF;
        const uint32_t word = (0u << 24) | ((jointStereo ? 1u : 0u) << 17) | (1u << 13) | (frameSize / 8);
        h[32] = (uint8_t)(word >> 24); h[33] = (uint8_t)(word >> 16); h[34] = (uint8_t)(word >> 8); h[35] = (uint8_t)word;
        Put(h, sizeof(h), "can't write header");
    }
    // ATRAC3plus (oma.cpp:28-41, liboma.c:190-206): codec id 1, channel id = channel count, (FrameSz - 8) / 8
    struct TAtrac3Plus {};
    TOmaOutput(const std::string& filename, TAtrac3Plus, size_t numChannels, uint32_t frameSize) : TFileOutput(filename, numChannels), FrameSize(frameSize)
    {
        uint8_t h[96] = {0};
        h[0] = 'E'; h[1] = 'A'; h[2] = '3'; h[3] = 1; h[5] = 96; h[6] = 0xFF; h[7] = 0xFF;
        const uint32_t word = (1u << 24) | (1u << 13) | ((numChannels == 1 ? 1u : 2u) << 10) | ((frameSize - 8) / 8);
        h[32] = (uint8_t)(word >> 24); h[33] = (uint8_t)(word >> 16); h[34] = (uint8_t)(word >> 8); h[35] = (uint8_t)word;
        Put(h, sizeof(h), "can't write header");
    }
    void WriteFrame(std::vector<char> data) override
    {
        if (data.size() < FrameSize) throw std::runtime_error("short frame");
        Put(data.data(), FrameSize, "write error");   // one block of the container's frame size (liboma.c:329-334)
    }

private:
    uint32_t FrameSize;
};

// RIFF/WAVE, format tag 0x270, 14 bytes of ATRAC3 extradata, a "fact" chunk, then "data"; the three length fields are
// written from the frame estimate first and corrected to the frames actually written when the file is closed.
class TAt3RiffOutput : public TFileOutput {
public:
    TAt3RiffOutput(const std::string& filename, size_t numChannels, uint32_t numFrames, uint32_t frameSize, bool jointStereo)
        : TFileOutput(filename, 2), FrameSize(frameSize)
    {
        const uint64_t fileSize = kHeader + (uint64_t)numFrames * frameSize;
        if (fileSize >= UINT32_MAX) throw std::runtime_error("File size is too big for this file format");
        uint8_t h[kHeader] = {0};
        memcpy(h, "RIFF", 4);
        Le32(h + 4, (uint32_t)(fileSize - 8));
        memcpy(h + 8, "WAVE", 4);
        memcpy(h + 12, "fmt ", 4);
        Le32(h + 16, 18 + 14);                       // WAVEFORMATEX + extradata
        Le16(h + 20, 0x270);
        Le16(h + 22, (uint32_t)numChannels);
        Le32(h + 24, 44100);
        Le32(h + 28, frameSize * 44100u / 1024u);
        Le16(h + 32, frameSize);
        Le16(h + 34, 0);
        Le16(h + 36, 14);
        Le16(h + 38, 1);
        Le32(h + 40, 0x1000);                        // PCM bytes per frame: 1024 samples x 2 channels x 2 bytes
        Le16(h + 44, jointStereo ? 1 : 0);
        Le16(h + 46, jointStereo ? 1 : 0);
        Le16(h + 48, 1);
        Le16(h + 50, 0);
        memcpy(h + 52, "fact", 4);
        Le32(h + 56, 8);
        Le32(h + 60, numFrames * 1024u);
        Le32(h + 64, 1024);
        memcpy(h + 68, "data", 4);
        Le32(h + 72, numFrames * frameSize);
        Put(h, sizeof(h), "Cannot write WAV header to file");
    }
    ~TAt3RiffOutput() override
    {
        const uint64_t fileSize = kHeader + FramesWritten * (uint64_t)FrameSize;
        if (FramesWritten > 0 && fileSize < UINT32_MAX) {
            uint8_t v[4];
            Le32(v, (uint32_t)(fileSize - 8));
            fseek(Fp, 4, SEEK_SET);
            fwrite(v, 1, 4, Fp);
            Le32(v, (uint32_t)FramesWritten * 1024u);
            fseek(Fp, 60, SEEK_SET);
            fwrite(v, 1, 4, Fp);
            Le32(v, (uint32_t)FramesWritten * FrameSize);
            fseek(Fp, 72, SEEK_SET);
            fwrite(v, 1, 4, Fp);
        }
    }
    void WriteFrame(std::vector<char> data) override
    {
        Put(data.data(), data.size(), "Cannot write AT3 data to file");
        ++FramesWritten;
    }

private:
    static constexpr size_t kHeader = 76;
    uint32_t FrameSize;
    uint64_t FramesWritten = 0;
};

// ATRAC3plus in RIFF/WAVE (at3.cpp:273-362): WAVE_FORMAT_EXTENSIBLE with the ATRAC3plus subformat GUID, a 4-byte "fact"
// chunk, then "data"; 80 header bytes, the length fields corrected on close like TAt3RiffOutput's.
class TAt3pRiffOutput : public TFileOutput {
public:
    TAt3pRiffOutput(const std::string& filename, size_t numChannels, uint32_t numFrames, uint32_t frameSize)
        : TFileOutput(filename, numChannels), FrameSize(frameSize)
    {
        if (frameSize > UINT16_MAX) throw std::runtime_error("ATRAC3plus frame size is too large for WAV block_align");
        const uint64_t fileSize = kHeader + (uint64_t)numFrames * frameSize;
        if (fileSize >= UINT32_MAX) throw std::runtime_error("File size is too big for this file format");
        static const uint8_t guid[16] = {0xBF, 0xAA, 0x23, 0xE9, 0x58, 0xCB, 0x71, 0x44, 0xA1, 0x19, 0xFF, 0xFA, 0x01, 0xE4, 0xCE, 0x62};
        uint8_t h[kHeader] = {0};
        memcpy(h, "RIFF", 4);
        Le32(h + 4, (uint32_t)(fileSize - 8));
        memcpy(h + 8, "WAVE", 4);
        memcpy(h + 12, "fmt ", 4);
        Le32(h + 16, 18 + 22);                       // WAVEFORMATEX + WAVEFORMATEXTENSIBLE tail
        Le16(h + 20, 0xFFFE);
        Le16(h + 22, (uint32_t)numChannels);
        Le32(h + 24, 44100);
        Le32(h + 28, frameSize * 44100u / 2048u);
        Le16(h + 32, frameSize);
        Le16(h + 34, 16);
        Le16(h + 36, 22);
        Le16(h + 38, 16);                            // valid bits per sample
        Le32(h + 40, numChannels == 1 ? 0x4u : numChannels == 2 ? 0x3u : 0u);   // front centre | front left + right
        memcpy(h + 44, guid, 16);
        memcpy(h + 60, "fact", 4);
        Le32(h + 64, 4);
        Le32(h + 68, numFrames * 2048u);
        memcpy(h + 72, "data", 4);
        Le32(h + 76, numFrames * frameSize);
        Put(h, sizeof(h), "Cannot write WAV header to file");
    }
    ~TAt3pRiffOutput() override
    {
        const uint64_t fileSize = kHeader + FramesWritten * (uint64_t)FrameSize;
        if (FramesWritten > 0 && fileSize < UINT32_MAX) {
            uint8_t v[4];
            Le32(v, (uint32_t)(fileSize - 8));
            fseek(Fp, 4, SEEK_SET);
            fwrite(v, 1, 4, Fp);
            Le32(v, (uint32_t)FramesWritten * 2048u);
            fseek(Fp, 68, SEEK_SET);
            fwrite(v, 1, 4, Fp);
            Le32(v, (uint32_t)FramesWritten * FrameSize);
            fseek(Fp, 76, SEEK_SET);
            fwrite(v, 1, 4, Fp);
        }
    }
    void WriteFrame(std::vector<char> data) override
    {
        if (data.size() != FrameSize) throw std::runtime_error("Unexpected ATRAC3plus frame size");
        Put(data.data(), data.size(), "Cannot write AT3 data to file");
        ++FramesWritten;
    }

private:
    static constexpr size_t kHeader = 80;
    uint32_t FrameSize;
    uint64_t FramesWritten = 0;
};

class TRawOutput : public TFileOutput {
public:
    TRawOutput(const std::string& filename, size_t numChannels, uint32_t frameSize = 0) : TFileOutput(filename, numChannels), FrameSize(frameSize) {}
    void WriteFrame(std::vector<char> data) override
    {
        if (FrameSize) data.resize(FrameSize);
        Put(data.data(), data.size(), "Cannot write raw ATRAC data to file");
    }

private:
    uint32_t FrameSize;
};

inline TCompressedOutputPtr CreateAtrac3Output(EContainer c, const std::string& outFile, size_t numChannels, uint32_t numFrames,
                                               uint32_t frameSize, bool jointStereo)   // main.cpp:391-408
{
    switch (c) {
        case EContainer::RIFF: return TCompressedOutputPtr(new TAt3RiffOutput(outFile, 2, numFrames, frameSize, jointStereo));
        case EContainer::RAW: return TCompressedOutputPtr(new TRawOutput(outFile, numChannels));
        default: return TCompressedOutputPtr(new TOmaOutput(outFile, frameSize, jointStereo));
    }
}

// ATRAC3plus: the extension rule is the ATRAC3 one (main.cpp:222-235), the writers are main.cpp:451-463's
inline EContainer SelectAtrac3PlusContainer(const std::string& outFile) { return SelectAtrac3Container(outFile); }
inline TCompressedOutputPtr CreateAtrac3PlusOutput(EContainer c, const std::string& outFile, size_t numChannels, uint32_t numFrames,
                                                   uint32_t frameSize = 2048)
{
    switch (c) {
        case EContainer::RIFF: return TCompressedOutputPtr(new TAt3pRiffOutput(outFile, numChannels, numFrames, frameSize));
        case EContainer::RAW: return TCompressedOutputPtr(new TRawOutput(outFile, numChannels));
        default: return TCompressedOutputPtr(new TOmaOutput(outFile, TOmaOutput::TAtrac3Plus{}, numChannels, frameSize));
    }
}

// ---- ATRAC1 -----------------------------------------------------------------------------------------------------
inline EContainer SelectAtrac1Container(const std::string& outFile)   // main.cpp:196-205 (AUTO)
{
    std::string ext;
    const size_t dot = outFile.find_last_of('.');
    if (dot != std::string::npos) ext = outFile.substr(dot + 1);
    for (char& ch : ext)
        if (ch >= 'A' && ch <= 'Z') ch = (char)(ch - 'A' + 'a');
    if (ext == "raw" || ext == "dat") return EContainer::RAW;
    return EContainer::AEA;
}

// 2048-byte header: 00 08 00 00, title (at most 15 characters kept) at 4, little-endian frame count at 260, channel
// count at 264; then one all-zero 212-byte unit. The FIRST WriteFrame call is swallowed (aea.cpp:176-181): the file
// carries the dummy unit in its place, every later unit follows resized to 212 bytes.
class TAeaOutput : public TFileOutput {
public:
    TAeaOutput(const std::string& filename, const std::string& title, size_t numChannels, uint32_t numFrames)
        : TFileOutput(filename, numChannels), Title(title)
    {
        uint8_t h[2048];
        memset(h, 0, sizeof(h));
        h[1] = 0x08;
        strncpy(reinterpret_cast<char*>(h) + 4, title.c_str(), 16);
        h[19] = 0;
        Le32(h + 260, numFrames);
        h[264] = (uint8_t)numChannels;
        Put(h, sizeof(h), "Can't write AEA header");
        static const char dummy[212] = {0};
        Put(dummy, sizeof(dummy), "Can't write dummy frame");
    }
    void WriteFrame(std::vector<char> data) override
    {
        if (FirstWrite) {
            FirstWrite = false;
            return;
        }
        data.resize(212);
        Put(data.data(), data.size(), "Can't write AEA frame");
    }
    std::string GetName() const override { return Title.substr(0, 15); }

private:
    std::string Title;
    bool FirstWrite = true;
};

inline TCompressedOutputPtr CreateAtrac1Output(EContainer c, const std::string& outFile, size_t numChannels, uint32_t numFrames)   // main.cpp:320-326
{
    if (c == EContainer::RAW) return TCompressedOutputPtr(new TRawOutput(outFile, numChannels, 212));
    return TCompressedOutputPtr(new TAeaOutput(outFile, "test", numChannels, numFrames));
}

}  // namespace NAtracDEncHip
