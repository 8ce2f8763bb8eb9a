// TEST INFRASTRUCTURE ONLY - never linked into the product library.
//
// Thin C-ABI harness around the *unmodified* reference sources, which are compiled in place
// from /root/reference by oracle/Makefile (target `ref`) into oracle/_ref/libat3ref.so.
// Nothing from the reference is copied into this repository: this file only calls the
// reference's public (and, through ATRAC_UT_PUBLIC / a test-only access hack, internal)
// classes so that
//   * the C restatement in oracle/at3_oracle.c can be validated bit-for-bit, and
//   * golden vectors under tests/golden/ can be generated (tools/gen_golden.py).
// It exists only in this container; the GPU box uses the prebuilt .so (cpu_baseline kind
// "reference") and the committed golden vectors.

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <optional>
#include <sstream>
#include <string>
#include <vector>

// compiled with -fno-access-control (see oracle/Makefile) to reach encoder internals
#define ATRAC_UT_PUBLIC
#include "atrac3denc.h"
#include "atrac/atrac_psy_common.h"
#include "atrac/atrac_scale.h"
#include "transient_detector.h"
#include "transient_spectral_upsampler.h"
#include "qmf/qmf.h"
#include "pcmengin.h"
#include "atrac/at1/atrac1_bitalloc.h"
#include "atrac1denc.h"
#include "oma.h"
#include "at3.h"
#include "raw.h"
#include "aea.h"
#include "atrac/at3p/at3p_mdct.h"
#include "atrac/at3p/at3p_bitstream.h"
#include "atrac/at3p/at3p_tables.h"
extern "C" {
#include "atrac/atrac3plus_pqf/atrac3plus_pqf.h"
#include "atrac/atrac3plus_pqf/ut/atrac3plusdsp.h"
}

using namespace NAtracDEnc;
using namespace NAtrac3;

namespace {

struct TMemOut : public ICompressedOutput {
    std::vector<std::vector<char>>* Frames;
    explicit TMemOut(std::vector<std::vector<char>>* f) : Frames(f) {}
    void WriteFrame(std::vector<char> data) override { Frames->push_back(std::move(data)); }
    std::string GetName() const override { return "mem"; }
    size_t GetChannelNum() const override { return 2; }
};

} // namespace

extern "C" {

// Per channel-frame tap record. Layout mirrored by tools/ and tests/ (ctypes).
struct ref_tap {
    int32_t n_points[4];
    int32_t level[4][8];
    int32_t loc[4][8];
    float ges_frame[4];      // GainEnergyScale[band].Frame
    float loudness_ch;       // sce->Loudness (per channel sum)
    float loudness_track;    // encoder Loudness after TrackLoudness
    int32_t sfi[32];
    float energy[32];
    float values[1024];      // ScaledBlocks[i].Values, concatenated in spectral order
    int32_t n_tonal;         // number of tonal blocks
    int32_t tonal_pos[64];
    int32_t tonal_len[64];
    int32_t tonal_sfi[64];
    float tonal_values[64][8];
};

// Encode nblocks PCM blocks (1024 samples x nch, interleaved, +-1.0) -> nblocks-1 frames.
// Returns number of frames; *frame_sz receives bytes per frame. taps may be NULL, else
// [nframes][nch] records.
int ref_encode(int bitrate, int nch, int no_gain, int no_tonal, int bfu_idx_const,
               const float* pcm, int nblocks, unsigned char* out, int* frame_sz, ref_tap* taps)
{
    std::vector<std::vector<char>> frames;
    TAtrac3EncoderSettings settings((uint32_t)bitrate, no_gain != 0, no_tonal != 0, (uint8_t)nch,
                                    (uint32_t)bfu_idx_const);
    const int fsz = settings.ConteinerParams->FrameSz;
    TAtrac3Encoder enc(TCompressedOutputPtr(new TMemOut(&frames)), std::move(settings));
    auto lambda = enc.GetLambda();
    std::vector<float> blk(1024 * nch);
    int nf = 0;
    for (int b = 0; b < nblocks; ++b) {
        memcpy(blk.data(), pcm + (size_t)b * 1024 * nch, sizeof(float) * 1024 * nch);
        TPCMEngine::ProcessMeta meta{(uint16_t)nch};
        auto r = lambda(blk.data(), meta);
        if (r == TPCMEngine::EProcessResult::PROCESSED) {
            if (taps) {
                for (int ch = 0; ch < nch; ++ch) {
                    ref_tap& t = taps[(size_t)nf * nch + ch];
                    memset(&t, 0, sizeof(t));
                    const auto& sce = enc.SingleChannelElements[ch];
                    for (int band = 0; band < 4 && band < (int)sce.SubbandInfo.Info.size(); ++band) {
                        const auto& pts = sce.SubbandInfo.GetGainPoints(band);
                        t.n_points[band] = (int)pts.size();
                        for (size_t i = 0; i < pts.size() && i < 8; ++i) {
                            t.level[band][i] = pts[i].Level;
                            t.loc[band][i] = pts[i].Location;
                        }
                        t.ges_frame[band] = sce.GainEnergyScale[band].Frame;
                    }
                    t.loudness_ch = sce.Loudness;
                    t.loudness_track = enc.Loudness;
                    size_t pos = 0;
                    for (size_t i = 0; i < sce.ScaledBlocks.size() && i < 32; ++i) {
                        t.sfi[i] = sce.ScaledBlocks[i].ScaleFactorIndex;
                        t.energy[i] = sce.ScaledBlocks[i].Energy;
                        for (float v : sce.ScaledBlocks[i].Values)
                            t.values[pos++] = v;
                    }
                    t.n_tonal = (int)sce.TonalBlocks.size();
                    for (size_t i = 0; i < sce.TonalBlocks.size() && i < 64; ++i) {
                        const auto& tb = sce.TonalBlocks[i];
                        t.tonal_pos[i] = tb.ValPtr->Pos;
                        t.tonal_len[i] = (int)tb.ScaledBlock.Values.size();
                        t.tonal_sfi[i] = tb.ScaledBlock.ScaleFactorIndex;
                        for (size_t j = 0; j < tb.ScaledBlock.Values.size() && j < 8; ++j)
                            t.tonal_values[i][j] = tb.ScaledBlock.Values[j];
                    }
                }
            }
            ++nf;
        }
    }
    for (size_t i = 0; i < frames.size(); ++i) {
        const size_t n = std::min<size_t>(frames[i].size(), fsz);
        memset(out + i * fsz, 0, fsz);
        memcpy(out + i * fsz, frames[i].data(), n);
    }
    if (frame_sz) *frame_sz = fsz;
    return (int)frames.size();
}

// QMF tree on one channel: pcm (already divided by 4 or not - caller decides) -> 4 subbands.
void ref_qmf(const float* pcm, int nblocks, float* sub /* [4][nblocks*256] */)
{
    Atrac3AnalysisFilterBank fb;
    for (int b = 0; b < nblocks; ++b) {
        float* p[4];
        for (int k = 0; k < 4; ++k) p[k] = sub + (size_t)k * nblocks * 256 + (size_t)b * 256;
        fb.Analysis(pcm + (size_t)b * 1024, p);
    }
}

// TAtrac3MDCT::Mdct on caller-owned [4][512] band buffers ([overlap|new]) with optional
// gain points per band; buffers are mutated like the reference does.
void ref_mdct(float* specs, float* bands /* [4][512] */, const int32_t* n_points,
              const int32_t* level /* [4][8] */, const int32_t* loc /* [4][8] */)
{
    static TAtrac3MDCT mdct;
    TAtrac3MDCT::TGainModulatorArray mods;
    for (int b = 0; b < 4; ++b) {
        std::vector<TAtrac3Data::SubbandInfo::TGainPoint> pts;
        for (int i = 0; n_points && i < n_points[b]; ++i)
            pts.push_back({(uint32_t)level[b * 8 + i], (uint32_t)loc[b * 8 + i]});
        mods[b] = mdct.GainProcessor.Modulate(pts);
    }
    float* p[4] = {bands, bands + 512, bands + 1024, bands + 1536};
    mdct.Mdct(specs, p, mods);
}

// CalcGainEnergyScale -> out[4] = {PrevHalf, CurHalf, Frame, NextOverlapScale}
void ref_gain_energy_scale(const float* prevOverlap, const float* cur, int n_points, const int32_t* level,
                           const int32_t* loc, float prevScale, float* out)
{
    std::vector<TAtrac3Data::SubbandInfo::TGainPoint> pts;
    for (int i = 0; i < n_points; ++i) pts.push_back({(uint32_t)level[i], (uint32_t)loc[i]});
    auto r = TAtrac3MDCT::CalcGainEnergyScale(prevOverlap, cur, pts, prevScale);
    out[0] = r.Scale.PrevHalf; out[1] = r.Scale.CurHalf; out[2] = r.Scale.Frame; out[3] = r.NextOverlapScale;
}

void ref_upsample(const float* in512, float* out4096, float* hfr)
{
    static TSpectralUpsampler up(11025.0f, 800.0f);
    auto r = up.Process(in512);
    memcpy(out4096, r.signal.data(), sizeof(float) * 4096);
    *hfr = r.highFreqRatio;
}

void ref_analyze_gain(const float* in, int len, int maxPoints, float* gain, float* lo, float* hi)
{
    std::vector<float> l, h;
    auto g = AnalyzeGain(in, len, maxPoints, true, &l, &h);
    for (size_t i = 0; i < g.size(); ++i) { gain[i] = g[i]; lo[i] = l[i]; hi[i] = h[i]; }
}

// ctx = {LastLevel, LastHpfEnergy, LastTarget}; returns number of points
int ref_calc_curve(const float* gain32, float* ctx, float minScore, const float* lo, const float* hi,
                   int32_t* level, int32_t* loc)
{
    std::vector<float> g(gain32, gain32 + 32), l(lo, lo + 32), h(hi, hi + 32);
    TCurveBuilderCtx c; c.LastLevel = ctx[0]; c.LastHpfEnergy = ctx[1]; c.LastTarget = ctx[2];
    auto pts = CalcCurve(g, c, {}, minScore, nullptr, &l, &h);
    ctx[0] = c.LastLevel; ctx[1] = c.LastHpfEnergy; ctx[2] = c.LastTarget;
    for (size_t i = 0; i < pts.size(); ++i) { level[i] = pts[i].Level; loc[i] = pts[i].Location; }
    return (int)pts.size();
}

int ref_relation_to_idx_hdr(float x) { return NAtracDEnc::RelationToIdx(x); }

float ref_quant_mantisas(const float* in, int n, float mul, int ea, int32_t* mant)
{
    return QuantMantisas(in, 0, (uint32_t)n, mul, ea != 0, mant);
}

void ref_scale_frame(const float* specs, int32_t* sfi, float* energy, float* values)
{
    static TScaler<TAtrac3Data> scaler;
    std::vector<float> s(specs, specs + 1024);
    auto blocks = scaler.ScaleFrame(s, TAtrac3Data::TBlockSizeMod());
    size_t pos = 0;
    for (size_t i = 0; i < blocks.size(); ++i) {
        sfi[i] = blocks[i].ScaleFactorIndex;
        energy[i] = blocks[i].Energy;
        for (float v : blocks[i].Values) values[pos++] = v;
    }
}

void ref_flatness(const float* energy1024, float* flat32)
{
    std::vector<float> e(energy1024, energy1024 + 1024);
    auto f = CalcSpectralFlatnessPerBfu<TAtrac3Data>(e);
    for (size_t i = 0; i < 32; ++i) flat32[i] = f[i];
}

float ref_log2f(float x) { return std::log2(x); }

// Constant tables. Any pointer may be NULL.
void ref_tables(float* scale64, float* encwin256, float* gainlevel16, float* gaininterp31,
                float* qmfwin48, float* loud1024, float* ath1024)
{
    static TAtrac3Data data;
    if (scale64) memcpy(scale64, TAtrac3Data::ScaleTable, 64 * 4);
    if (encwin256) memcpy(encwin256, TAtrac3Data::EncodeWindow, 256 * 4);
    if (gainlevel16) memcpy(gainlevel16, TAtrac3Data::GainLevel, 16 * 4);
    if (gaininterp31) memcpy(gaininterp31, TAtrac3Data::GainInterpolation, 31 * 4);
    if (qmfwin48) { TQmf<512> q; (void)q; memcpy(qmfwin48, TQmfCommon::QmfWindow, 48 * 4); }
    if (loud1024) { auto l = CreateLoudnessCurve(1024); memcpy(loud1024, l.data(), 1024 * 4); }
    if (ath1024) { auto a = CalcATH(1024, 44100); memcpy(ath1024, a.data(), 1024 * 4); }
}

// TMDCT<512> alone (scale 1), 512 -> 256
void ref_mdct512(const float* in512, float* out256)
{
    static NMDCT::TMDCT<512> m(1);
    const auto& r = m(in512);
    memcpy(out256, r.data(), 256 * 4);
}


// ---- callers and data formats either side of the path -------------------------------------------

// The reference's container writers fed with `n_frames` frames of `frame_sz` bytes. kind: 0 OMA, 1 RIFF, 2 raw.
int at3ref_write_container(int kind, const char* path, const uint8_t* frames, int n_frames, int frame_sz, int js,
                           int num_frames_hint, int nch)
{
    try {
        TCompressedOutputPtr out;
        if (kind == 1) out = CreateAt3Output(path, 2, (uint32_t)num_frames_hint, (uint32_t)frame_sz, js != 0);
        else if (kind == 2) out = CreateRawOutput(path, (size_t)nch);
        else if (kind == 3) out = CreateAeaOutput(path, "test", (size_t)nch, (uint32_t)num_frames_hint);
        else if (kind == 4) out = CreateRawOutput(path, (size_t)nch, (uint32_t)frame_sz);   // ATRAC1 raw, main.cpp:323
        else if (kind == 5) out.reset(new TOma(path, "test", (size_t)nch, (uint32_t)num_frames_hint, OMAC_ID_ATRAC3PLUS, (uint32_t)frame_sz, false));   // main.cpp:456-461
        else if (kind == 6) out = CreateAt3POutput(path, (size_t)nch, (uint32_t)num_frames_hint, (uint32_t)frame_sz);
        else if (kind == 7) out = CreateRawOutput(path, (size_t)nch);
        else out.reset(new TOma(path, "test", (size_t)nch, (uint32_t)num_frames_hint, OMAC_ID_ATRAC3, (uint32_t)frame_sz, js != 0));
        for (int i = 0; i < n_frames; ++i)
            out->WriteFrame(std::vector<char>(frames + (size_t)i * frame_sz, frames + (size_t)(i + 1) * frame_sz));
    } catch (const std::exception&) {
        return -1;
    }
    return 0;
}

// TPCMEngine::ApplyProcess driven like main.cpp:697-705 by a reader that delivers `total_samples` frames of a ramp
// (sample value = 1 + frame index, every channel) with the short-read / end-of-data behaviour of TWav::GetPCMReader
// (wav.cpp:46-61). The lambda answers LOOK_AHEAD once, then PROCESSED, and logs the first and last frame value of every
// call, plus - when `tail` is given - the whole 1024 x nch block of the LAST call. Returns the number of lambda calls (or -1 if the engine threw TNoDataToRead), *processed_out = final count.
int at3ref_engine_trace_step(uint64_t total_samples, int nch, int step, int look_ahead, float* first_vals, float* last_vals, int max_calls,
                             uint64_t* processed_out, float* tail)
{
    struct TRampReader : public IPCMReader {
        mutable uint64_t Pos = 0;
        uint64_t Total;
        explicit TRampReader(uint64_t total) : Total(total) {}
        bool Read(TPCMBuffer& data, const uint32_t size) const override
        {
            uint64_t n = Total - Pos;
            if (n > size) n = size;
            if (!n) return false;
            for (uint64_t i = 0; i < n; ++i)
                for (int c = 0; c < data.Channels(); ++c) data[i][c] = (float)(Pos + i + 1);
            if (n != size) data.Zero(n, size - n);   // exactly what TWav::GetPCMReader does with a short read
            Pos += n;
            return true;
        }
    };
    TPCMEngine engine(4096, (size_t)nch, TPCMEngine::TReaderPtr(new TRampReader(total_samples)));
    int calls = 0;
    auto lambda = [&](float* data, const TPCMEngine::ProcessMeta& meta) {
        if (calls < max_calls) {
            first_vals[calls] = data[0];
            last_vals[calls] = data[(size_t)(step - 1) * meta.Channels];
        }
        if (tail) memcpy(tail, data, sizeof(float) * step * meta.Channels);
        return (calls++ == 0 && look_ahead) ? TPCMEngine::EProcessResult::LOOK_AHEAD : TPCMEngine::EProcessResult::PROCESSED;
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


int at3ref_engine_trace(uint64_t total_samples, int nch, float* first_vals, float* last_vals, int max_calls, uint64_t* processed_out,
                        float* tail)
{
    return at3ref_engine_trace_step(total_samples, nch, 1024, 1, first_vals, last_vals, max_calls, processed_out, tail);
}

// ---- ATRAC1 encoder: the reference's TAtrac1Encoder driven one 512-sample block at a time --------
namespace {
struct TCaptureOut : public ICompressedOutput {
    std::vector<uint8_t>* Dst;
    size_t Channels;
    TCaptureOut(std::vector<uint8_t>* dst, size_t ch) : Dst(dst), Channels(ch) {}
    void WriteFrame(std::vector<char> data) override
    {
        data.resize(212);   // the sound unit; the bit writer appends rounding bytes that the AEA container drops (aea.cpp:182)
        Dst->insert(Dst->end(), data.begin(), data.end());
    }
    std::string GetName() const override { return {}; }
    size_t GetChannelNum() const override { return Channels; }
};
}

// CalcATH(512, 44100) and CreateLoudnessCurve(512) as the ATRAC1 encoder uses them (atrac_psy_common.cpp:126-156)
void at1ref_psy_tables(float* ath512, float* loud512)
{
    const std::vector<float> a = CalcATH(512, 44100);
    const std::vector<float> l = CreateLoudnessCurve(512);
    memcpy(ath512, a.data(), 512 * sizeof(float));
    memcpy(loud512, l.data(), 512 * sizeof(float));
}

// pcm [n_blocks][512][nch] -> out [n_blocks][nch][212]; returns bytes written. window_auto = 1: transient detection,
// else `window_mask` (bit 0 low, 1 mid, 2 high band short) for every frame. bfu_idx_const as TAtrac1EncodeSettings.
int at1ref_encode(const float* pcm, int nch, int n_blocks, int window_auto, int window_mask, int bfu_idx_const, uint8_t* out)
{
    std::vector<uint8_t> bytes;
    {
        NAtrac1::TAtrac1EncodeSettings settings((uint32_t)bfu_idx_const,
            window_auto ? NAtrac1::TAtrac1EncodeSettings::EWindowMode::EWM_AUTO : NAtrac1::TAtrac1EncodeSettings::EWindowMode::EWM_NOTRANSIENT,
            (uint32_t)window_mask);
        TAtrac1Encoder enc(TCompressedOutputPtr(new TCaptureOut(&bytes, (size_t)nch)), std::move(settings));
        auto lambda = enc.GetLambda();
        std::vector<float> block(512 * (size_t)nch);
        const TPCMEngine::ProcessMeta meta = {(uint16_t)nch};
        for (int b = 0; b < n_blocks; ++b) {
            memcpy(block.data(), pcm + (size_t)b * 512 * nch, sizeof(float) * 512 * nch);
            lambda(block.data(), meta);
        }
    }
    memcpy(out, bytes.data(), bytes.size());
    return (int)bytes.size();
}

// ---- ATRAC3plus front end: PQF analysis and TAt3pMDCT, one channel ---------------------------------
// in [n_frames][2048] -> subbands [n_frames][16][128], filter state carried over the frames (start-of-stream state first)
void at3pref_pqf_analyse(const float* in, int n_frames, float* out)
{
    at3plus_pqf_a_ctx_t ctx = at3plus_pqf_create_a_ctx();
    for (int f = 0; f < n_frames; ++f) at3plus_pqf_do_analyse(ctx, in + (size_t)f * 2048, out + (size_t)f * 2048);
    at3plus_pqf_free_a_ctx(ctx);
}

// The decoder-side synthesis filter of the reference's unit test (atrac3plus_pqf/ut/atrac3plusdsp.c), for round trips.
void at3pref_ipqf(const float* in, int n_frames, float* out)
{
    Atrac3pIPQFChannelCtx ctx;
    memset(&ctx, 0, sizeof(ctx));
    for (int f = 0; f < n_frames; ++f) ff_atrac3p_ipqf(&ctx, in + (size_t)f * 2048, out + (size_t)f * 2048);
}

// bands [n_frames][16][128], steep-window flag word per frame (bit b = subband b) -> specs [n_frames][2048]; the history
// buffer starts zeroed like TChannelCtx::MdctBuf (at3p.cpp:76)
void at3pref_mdct(const float* bands, const uint16_t* win_flags, int n_frames, float* specs)
{
    TAt3pMDCT mdct;
    TAt3pMDCT::THistBuf hist = {{{0}}};
    for (int f = 0; f < n_frames; ++f) {
        TAt3pMDCT::TPcmBandsData p;
        for (size_t b = 0; b < 16; ++b) p[b] = bands + (size_t)f * 2048 + b * 128;
        TAt3pMDCTWin win;
        for (size_t b = 0; b < 16; ++b)
            if (win_flags && ((win_flags[f] >> b) & 1)) win.SetSteepWin(b);
        mdct.Do(specs + (size_t)f * 2048, p, hist, win);
    }
}

// ATRAC3plus frame writer without tonal block: TScaler<NAt3p::TScaleTable>::ScaleFrame per channel, then
// TAt3PBitStream::WriteFrame(channels, nullptr, sces) as TAt3PEnc::EncodeFrame calls it (at3p.cpp:139-163).
// specs [n_frames][channels][2048], win_flags [n_frames][channels] or NULL -> out [n_frames][2048]
int at3pref_write_frames(const float* specs, const uint16_t* win_flags, int channels, int n_frames, uint8_t* out)
{
    std::vector<std::vector<char>> frames;
    TMemOut mem(&frames);
    TAt3PBitStream bs(&mem, 2048);
    TScaler<NAt3p::TScaleTable> scaler;
    for (int f = 0; f < n_frames; ++f) {
        std::vector<TAt3PBitStream::TSingleChannelElement> sces(channels);
        for (int ch = 0; ch < channels; ++ch) {
            const float* x = specs + ((size_t)f * channels + ch) * 2048;
            std::vector<float> v(x, x + 2048);
            sces[ch].ScaledBlocks = scaler.ScaleFrame(v, NAt3p::TScaleTable::TBlockSizeMod());
            for (size_t b = 0; b < 16; ++b)
                if (win_flags && ((win_flags[(size_t)f * channels + ch] >> b) & 1)) sces[ch].SubbandInfo.Win.SetSteepWin(b);
        }
        bs.WriteFrame(channels, nullptr, sces);
        if (frames.empty() || frames.back().size() != 2048) return -1;
        memcpy(out + (size_t)f * 2048, frames.back().data(), 2048);
        frames.clear();
    }
    return n_frames;
}

} // extern "C"
