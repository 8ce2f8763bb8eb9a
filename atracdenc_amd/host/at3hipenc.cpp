// at3hipenc - command-line ATRAC3 / ATRAC1 / ATRAC3plus encoder on libat3hip: the
// reference tool's `-e atrac3` path (main.cpp:367-425, 659-705), `-e atrac1` path (main.cpp:292-345, 630-648) and
// `-e atrac3plus` path (main.cpp:427-483, 679-686; without the tonal analysis, which needs libgha) with the GPU encoders
// behind the same IProcessor-shaped objects.
//
//   at3hipenc -e atrac3 -i in.wav -o out.{oma|at3|wav|raw|dat} [--bitrate kbit] [--bfuidxconst n] [--notonal]
//             [--nogaincontrol] [--container oma|riff|raw] [--nostdout] [--batch blocks] [--device n]
//   at3hipenc -e atrac1 -i in.wav -o out.{aea|raw|dat} [--bfuidxconst 1..8] [--notransient[=mask]]
//             [--container aea|raw] [--nostdout] [--batch blocks] [--device n]
//   at3hipenc -e atrac3plus -i in.wav -o out.{oma|at3|wav|raw|dat} [--container oma|riff|raw] [--nostdout]
//             [--batch frames] [--device n]
//
// File-level behaviour follows the reference: 44.1 kHz input only, numFrames estimate = samples / 1024 in the
// container header, the look-ahead first call, the drain call at end of input.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "at3hip_io.hpp"

using namespace NAtracDEncHip;

static int usage()
{
    std::cerr << "usage: at3hipenc -e atrac3 -i in.wav -o out.oma [--bitrate kbit] [--bfuidxconst n] [--notonal] [--nogaincontrol]\n"
                 "                 [--container oma|riff|raw] [--nostdout] [--batch blocks] [--device n]\n"
                 "       at3hipenc -e atrac1 -i in.wav -o out.aea [--bfuidxconst 1..8] [--notransient[=mask]]\n"
                 "                 [--container aea|raw] [--nostdout] [--batch blocks] [--device n]\n"
                 "       at3hipenc -e atrac3plus -i in.wav -o out.oma [--container oma|riff|raw] [--nostdout] [--batch frames] [--device n]\n";
    return 1;
}

int main(int argc, char** argv)
{
    std::string inFile, outFile, codec, container;
    uint32_t bitrate = 0, bfuIdxConst = 0;
    bool noTonal = false, noGain = false, noStdOut = false, noTransient = false;
    uint32_t winMask = 0;
    int batch = 256, device = 0;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto need = [&](const char* what) -> const char* {
            if (i + 1 >= argc) {
                std::cerr << "missing value for " << what << "\n";
                exit(usage());
            }
            return argv[++i];
        };
        if (a == "-e" || a == "--encode") codec = need("-e");
        else if (a == "-i") inFile = need("-i");
        else if (a == "-o") outFile = need("-o");
        else if (a == "--bitrate") bitrate = (uint32_t)atoi(need("--bitrate"));
        else if (a == "--bfuidxconst") bfuIdxConst = (uint32_t)atoi(need("--bfuidxconst"));
        else if (a == "--notonal") noTonal = true;
        else if (a == "--nogaincontrol") noGain = true;
        else if (a == "--nostdout") noStdOut = true;
        else if (a.rfind("--notransient", 0) == 0 && (a.size() == 13 || a[13] == '=')) {   // optional_argument, main.cpp:568-577
            noTransient = true;
            if (a.size() > 14) winMask = (uint32_t)atoi(a.c_str() + 14);
        }
        else if (a == "--container") container = need("--container");
        else if (a == "--batch") batch = atoi(need("--batch"));
        else if (a == "--device") device = atoi(need("--device"));
        else return usage();
    }
    if ((codec != "atrac3" && codec != "atrac1" && codec != "atrac3plus") || inFile.empty() || outFile.empty()) return usage();
    if (codec == "atrac3plus") {
        try {
            TWavSource wav(inFile);
            if (wav.GetSampleRate() != 44100) throw std::runtime_error("unsupported sample rate");
            const size_t numChannels = wav.GetChannelNum();
            const uint64_t totalSamples = wav.GetTotalSamples();
            const uint64_t numFrames = totalSamples / 2048;   // main.cpp:440
            EContainer cont;
            if (container.empty()) cont = SelectAtrac3PlusContainer(outFile);
            else if (container == "oma") cont = EContainer::OMA;
            else if (container == "riff") cont = EContainer::RIFF;
            else if (container == "raw") cont = EContainer::RAW;
            else throw std::runtime_error("unrecognized container: " + container);
            TCompressedOutputPtr out = CreateAtrac3PlusOutput(cont, outFile, numChannels, (uint32_t)numFrames, 2048);
            if (!noStdOut)
                std::cout << "Input:\n Filename: " << inFile << "\n Channels: " << numChannels << "\n SampleRate: " << wav.GetSampleRate()
                          << "\n Duration (sec): " << totalSamples / wav.GetSampleRate() << "\nOutput:\n Filename: " << outFile
                          << "\n Codec: ATRAC3Plus" << std::endl;
            TPCMEngine engine(4096, numChannels, [&wav](float* dst, size_t frames) { return wav.Read(dst, frames); });
            TAt3PEncoder encoder(std::move(out), (int)numChannels, batch > 64 ? 64 : batch, device);
            auto lambda = encoder.GetLambda();
            uint64_t processed = 0;
            try {
                while (totalSamples > (processed = engine.ApplyProcess(2048, lambda))) {
                }
            } catch (const TNoDataToRead&) {
                std::cerr << "No more data to read from input" << std::endl;
            }
            encoder.Flush();
            if (!noStdOut) std::cout << "\nDone" << std::endl;
        } catch (const std::exception& ex) {
            std::cerr << "Fatal error: " << ex.what() << std::endl;
            return 1;
        }
        return 0;
    }
    if (codec == "atrac1") {
        if (bfuIdxConst > 8) {
            std::cerr << "ATRAC1 mode, --bfuidxconst is a index of max used BFU. Values [1;8] is allowed\n";
            return 1;
        }
        try {
            TWavSource wav(inFile);
            if (wav.GetSampleRate() != 44100) throw std::runtime_error("unsupported sample rate");
            const size_t numChannels = wav.GetChannelNum();
            const uint64_t totalSamples = wav.GetTotalSamples();
            const uint64_t numFrames = numChannels * totalSamples / 512;   // main.cpp:312
            EContainer cont;
            if (container.empty()) cont = SelectAtrac1Container(outFile);
            else if (container == "aea") cont = EContainer::AEA;
            else if (container == "raw") cont = EContainer::RAW;
            else throw std::runtime_error("unrecognized container: " + container);
            TCompressedOutputPtr out = CreateAtrac1Output(cont, outFile, numChannels, (uint32_t)numFrames);
            if (!noStdOut)
                std::cout << "Input\n Filename: " << inFile << "\n Channels: " << numChannels << "\n SampleRate: " << wav.GetSampleRate()
                          << "\n Duration (sec): " << totalSamples / wav.GetSampleRate() << "\nOutput:\n Filename: " << outFile
                          << "\n Codec: ATRAC1" << std::endl;
            TPCMEngine engine(4096, numChannels, [&wav](float* dst, size_t frames) { return wav.Read(dst, frames); });
            TAtrac1Encoder encoder(std::move(out),
                                   TAtrac1EncodeSettings(bfuIdxConst,
                                                         noTransient ? TAtrac1EncodeSettings::EWindowMode::EWM_NOTRANSIENT
                                                                     : TAtrac1EncodeSettings::EWindowMode::EWM_AUTO,
                                                         winMask),
                                   batch, device);
            auto lambda = encoder.GetLambda();
            uint64_t processed = 0;
            try {
                while (totalSamples > (processed = engine.ApplyProcess(512, lambda))) {
                }
            } catch (const TNoDataToRead&) {
                std::cerr << "No more data to read from input" << std::endl;
            }
            encoder.Flush();
            if (!noStdOut) std::cout << "\nDone" << std::endl;
        } catch (const std::exception& ex) {
            std::cerr << "Fatal error: " << ex.what() << std::endl;
            return 1;
        }
        return 0;
    }
    if (bitrate && (bitrate < 32 || bitrate > 384)) {
        std::cerr << "bitrate must be in [32;384]\n";
        return 1;
    }
    if (bfuIdxConst > 32) {
        std::cerr << "bfuidxconst must be in [1;32]\n";
        return 1;
    }
    try {
        TWavSource wav(inFile);
        if (wav.GetSampleRate() != 44100) throw std::runtime_error("unsupported sample rate");
        const size_t numChannels = wav.GetChannelNum();
        const uint64_t totalSamples = wav.GetTotalSamples();
        const uint64_t numFrames = totalSamples / 1024;

        TAtrac3EncoderSettings settings;
        settings.Bitrate = bitrate * 1024;   // the tool's kbit value reaches the settings as value * 1024 (main.cpp:676)
        settings.NoGainControll = noGain;
        settings.NoTonalComponents = noTonal;
        settings.SourceChannels = (uint8_t)numChannels;
        settings.BfuIdxConst = bfuIdxConst;

        // container parameters come from the encoder context (GetContainerParamsForBitrate)
        at3hip_config probe{};
        probe.bitrate = (int32_t)settings.Bitrate;
        probe.channels = (int32_t)numChannels;
        probe.n_streams = 1;
        probe.max_blocks = 1;
        probe.device_id = device;
        at3hip_ctx* pc = nullptr;
        Check(at3hip_create(&probe, &pc), nullptr, "at3hip_create");
        const uint32_t frameSize = (uint32_t)at3hip_frame_size(pc);
        const bool js = at3hip_joint_stereo(pc) != 0;
        at3hip_destroy(pc);

        EContainer cont;
        if (container.empty()) cont = SelectAtrac3Container(outFile);
        else if (container == "oma") cont = EContainer::OMA;
        else if (container == "riff") cont = EContainer::RIFF;
        else if (container == "raw") cont = EContainer::RAW;
        else throw std::runtime_error("unrecognized container: " + container);

        TCompressedOutputPtr out = CreateAtrac3Output(cont, outFile, numChannels, (uint32_t)numFrames, frameSize, js);
        if (!noStdOut)
            std::cout << "Input:\n Filename: " << inFile << "\n Channels: " << numChannels << "\n SampleRate: " << wav.GetSampleRate()
                      << "\n Duration (sec): " << totalSamples / wav.GetSampleRate() << "\nOutput:\n Filename: " << outFile
                      << "\n Codec: ATRAC3\n Bitrate: " << (frameSize == 384 ? 132300 : frameSize * 44100u * 8u / 1024u) << std::endl;

        TPCMEngine engine(4096, numChannels, [&wav](float* dst, size_t frames) { return wav.Read(dst, frames); });
        TAtrac3Encoder encoder(std::move(out), std::move(settings), batch, device);
        auto lambda = encoder.GetLambda();
        uint64_t processed = 0;
        try {
            while (totalSamples > (processed = engine.ApplyProcess(1024, lambda))) {
            }
        } catch (const TNoDataToRead&) {
            std::cerr << "No more data to read from input" << std::endl;
        }
        encoder.Flush();
        if (!noStdOut) std::cout << "\nDone" << std::endl;
    } catch (const std::exception& ex) {
        std::cerr << "Fatal error: " << ex.what() << std::endl;
        return 1;
    }
    return 0;
}
