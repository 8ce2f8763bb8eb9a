// GPU test of the host-side C++ mirror (atracdenc_amd/host/at3hip_host.hpp): reads like the reference's own
// tests (atrac3denc_ut.cpp) - drive the encoder through GetLambda()/WriteFrame and TAtrac3MDCT::Mdct - and
// checks the results bit-for-bit against the CPU oracle (test infrastructure, linked only into this test).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../atracdenc_amd/host/at3hip_host.hpp"
#include "../../oracle/at3_oracle.h"

using namespace NAtracDEncHip;

extern "C" {   // oracle/at3p_oracle.c, oracle/at3p_frame_oracle.c
void at3po_pqf_analyse(const float* in, int n_frames, float* out);
void at3po_mdct(const float* bands, const uint16_t* win_flags, int n_frames, float* specs);
int at3po_write_frames(const float* specs, const uint16_t* win_flags, int channels, int n_frames, uint8_t* out, void* info);
}

struct TMemOut : ICompressedOutput {
    std::vector<std::vector<char>>* Frames;
    explicit TMemOut(std::vector<std::vector<char>>* f) : Frames(f) {}
    void WriteFrame(std::vector<char> data) override { Frames->push_back(std::move(data)); }
    std::string GetName() const override { return "mem"; }
    size_t GetChannelNum() const override { return 2; }
};

static int fails = 0;
#define EXPECT(cond)                                                   \
    do {                                                               \
        if (!(cond)) {                                                 \
            printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond);      \
            ++fails;                                                   \
        }                                                              \
    } while (0)

int main()
{
    // ---- encoder: 3 kHz bursts (drives gain control), 23 blocks through a batch of 8 ----
    const int nb = 23;
    std::vector<float> pcm((size_t)nb * 2048);
    for (int i = 0; i < nb * 1024; ++i) {
        const double amp = ((i / 3000) % 2 == 0) ? 0.02 : 0.6;
        const double v = amp * sin(2 * M_PI * 3000.0 * i / 44100.0);
        pcm[2 * i] = (float)(lrint(v * 32768.0) / 32768.0);
        pcm[2 * i + 1] = (float)(lrint(0.5 * v * 32768.0) / 32768.0);
    }
    for (uint32_t bitrate : {132300u, 66150u}) {
        std::vector<std::vector<char>> frames;
        {
            TAtrac3EncoderSettings st;
            st.Bitrate = bitrate;
            TAtrac3Encoder enc(TCompressedOutputPtr(new TMemOut(&frames)), std::move(st), 8);
            auto lambda = enc.GetLambda();
            for (int b = 0; b < nb; ++b) {
                const auto r = lambda(pcm.data() + (size_t)b * 2048, ProcessMeta{2});
                EXPECT((b == 0) == (r == EProcessResult::LOOK_AHEAD));
            }
        }   // destructor flushes
        std::vector<unsigned char> exp((size_t)nb * 1024);
        int fsz = 0;
        const int nf = at3o_encode((int)bitrate, 2, 0, 0, 0, pcm.data(), nb, exp.data(), &fsz, nullptr);
        EXPECT(nf == nb - 1);
        EXPECT((int)frames.size() == nf);
        for (int i = 0; i < nf && i < (int)frames.size(); ++i) {
            EXPECT((int)frames[i].size() == fsz);
            EXPECT(memcmp(frames[i].data(), exp.data() + (size_t)i * fsz, fsz) == 0);
        }
        printf("encoder bitrate %u: %d frames of %d bytes compared\n", bitrate, nf, fsz);
    }
    // ---- TAtrac3MDCT::Mdct with a gain curve on band 0 (gain_processor_ut.cpp style) ----
    {
        TAtrac3MDCT mdct;
        float bands[4][512], ref[4][512];
        for (int b = 0; b < 4; ++b)
            for (int i = 0; i < 512; ++i) bands[b][i] = ref[b][i] = 0.25f * (float)sin(0.01 * (i + 1) * (b + 2));
        float* p[4] = {bands[0], bands[1], bands[2], bands[3]};
        TAtrac3MDCT::TGainCurves curves;
        curves[0] = {{6, 4}, {3, 20}};
        float specs[1024], especs[1024];
        mdct.Mdct(specs, p, curves);
        const int32_t n[4] = {2, 0, 0, 0};
        int32_t level[32] = {6, 3}, loc[32] = {4, 20};
        at3o_mdct(especs, &ref[0][0], n, level, loc);
        EXPECT(memcmp(specs, especs, sizeof(specs)) == 0);
        EXPECT(memcmp(bands, ref, sizeof(ref)) == 0);
        printf("TAtrac3MDCT::Mdct compared\n");
    }
    // ---- the maxLevels overload and CalcGainEnergyScale (atrac3denc.h:69-83) ----
    {
        TAtrac3MDCT mdct;
        float bands[4][512];
        for (int b = 0; b < 4; ++b)
            for (int i = 0; i < 512; ++i) bands[b][i] = 0.3f * (float)sin(0.013 * (i + 3) * (b + 1));
        float* p[4] = {bands[0], bands[1], bands[2], bands[3]};
        TAtrac3MDCT::TGainCurves curves;
        curves[1] = {{2, 7}};
        float specs[1024], maxLevels[4];
        mdct.Mdct(specs, p, maxLevels, curves);
        for (int b = 0; b < 4; ++b) {
            float m = 0.0f;
            for (int i = 0; i < 256; ++i) m = std::max(m, std::fabs(bands[b][256 + i]));   // new half as the call left it
            EXPECT(m == maxLevels[b]);
        }
        float prev[256], cur[256], out[4];
        for (int i = 0; i < 256; ++i) {
            prev[i] = 0.2f * (float)cos(0.05 * i);
            cur[i] = 0.4f * (float)sin(0.02 * i + 0.3);
        }
        const std::vector<TGainPoint> pts = {{5, 3}, {3, 17}, {6, 30}};
        const auto res = mdct.CalcGainEnergyScale(prev, cur, pts, 0.75f);
        const int32_t level[8] = {5, 3, 6}, loc[8] = {3, 17, 30};
        at3o_gain_energy_scale(prev, cur, 3, level, loc, 0.75f, out);
        EXPECT(memcmp(&res.Scale.PrevHalf, &out[0], 4) == 0 && memcmp(&res.Scale.CurHalf, &out[1], 4) == 0);
        EXPECT(memcmp(&res.Scale.Frame, &out[2], 4) == 0 && memcmp(&res.NextOverlapScale, &out[3], 4) == 0);
        printf("TAtrac3MDCT maxLevels / CalcGainEnergyScale compared\n");
    }
    // ---- TAtrac3EncoderNode: streams sharded over devices (this box has one: two contexts on device 0) ----
    {
        EXPECT(ShardStreams(7, 3, 0) == std::make_pair(0, 3) && ShardStreams(7, 3, 1) == std::make_pair(3, 2) && ShardStreams(7, 3, 2) == std::make_pair(5, 2));
        const int S = 5, n = 9;
        std::vector<float> batch((size_t)S * n * 2048);
        for (int s = 0; s < S; ++s)
            for (int i = 0; i < n * 2048; ++i) batch[(size_t)s * n * 2048 + i] = pcm[(size_t)((i + 4096 * s) % (nb * 2048))];
        TAtrac3EncoderSettings st;
        TAtrac3EncoderNode node(st, S, n, {0, 0});
        std::vector<uint8_t> frames;
        const int nf = node.Encode(batch.data(), n, frames);
        EXPECT(nf == n - 1 && node.Devices() == 2);
        for (int s = 0; s < S; ++s) {
            std::vector<unsigned char> exp((size_t)n * 1024);
            int fsz = 0;
            const int enf = at3o_encode(132300, 2, 0, 0, 0, batch.data() + (size_t)s * n * 2048, n, exp.data(), &fsz, nullptr);
            EXPECT(enf == nf && fsz == node.FrameSize());
            EXPECT(memcmp(frames.data() + (size_t)s * nf * fsz, exp.data(), (size_t)nf * fsz) == 0);
        }
        printf("TAtrac3EncoderNode (2 contexts) compared; device 0 sits on host NUMA node %d (at3hip_device_numa_node; -1 = the platform does not say)\n", at3hip_device_numa_node(0));
        // the same input through the page-locked, double-buffered pipeline: calls of 2 blocks, copies and kernels overlapping
        node.Reset();
        std::vector<uint8_t> piped;
        const int nfp = node.EncodePipelined(batch.data(), n, 2, piped);
        EXPECT(nfp == nf && piped.size() == frames.size());
        EXPECT(memcmp(piped.data(), frames.data(), frames.size()) == 0);
        printf("TAtrac3EncoderNode::EncodePipelined (pinned staging, 5 calls of <= 2 blocks) compared\n");
        // 16-bit samples: TAtrac3EncoderBatch::EncodeS16 and the 16-bit pipeline against the float path on s / 32768.0f
        std::vector<int16_t> b16((size_t)S * n * 2048);
        std::vector<float> bf((size_t)S * n * 2048);
        for (size_t i = 0; i < b16.size(); ++i) {
            b16[i] = (int16_t)lrintf(std::max(-32768.0f, std::min(32767.0f, batch[i] * 32768.0f)));
            bf[i] = (float)b16[i] / 32768.0f;
        }
        TAtrac3EncoderBatch fb(st, S, n), sb(st, S, n), pb(st, S, n);
        std::vector<uint8_t> ff, fs, fp;
        const int nff = fb.Encode(bf.data(), n, ff), nfs = sb.EncodeS16(b16.data(), n, fs);
        EXPECT(nff == nfs && ff == fs);
        int fed = 0;
        const long long tot = pb.EncodePipelinedS16(
            2, 2,
            [&](int16_t* dst, int maxBlocks) {
                const int nbk = std::min(maxBlocks, n - fed);
                for (int s = 0; s < S && nbk > 0; ++s) memcpy(dst + (size_t)s * nbk * 2048, b16.data() + ((size_t)s * n + fed) * 2048, (size_t)nbk * 2048 * sizeof(int16_t));
                fed += nbk > 0 ? nbk : 0;
                return nbk;
            },
            [&](const uint8_t* fr, int nfr) {
                const size_t at = fp.size() / ((size_t)S * sb.FrameSize());   // frames per stream so far
                fp.resize(fp.size() + (size_t)S * nfr * sb.FrameSize());
                // (kept call-major here; compared call by call below)
                memcpy(fp.data() + at * S * sb.FrameSize(), fr, (size_t)S * nfr * sb.FrameSize());
            });
        EXPECT(tot == nfs);
        {   // the calls returned 1, 2, 2, 2, 1 frames per stream: stream-major inside each call
            const int fsz = sb.FrameSize();
            size_t off = 0;
            int done = 0;
            const int per_call[5] = {1, 2, 2, 2, 1};
            for (int cidx = 0; cidx < 5; ++cidx) {
                for (int s = 0; s < S; ++s)
                    EXPECT(memcmp(fp.data() + off + (size_t)s * per_call[cidx] * fsz, fs.data() + ((size_t)s * nfs + done) * fsz, (size_t)per_call[cidx] * fsz) == 0);
                off += (size_t)S * per_call[cidx] * fsz;
                done += per_call[cidx];
            }
        }
        printf("TAtrac3EncoderBatch::EncodeS16 / EncodePipelinedS16 compared with the float path\n");
    }
    // ---- TAtrac3EncoderNode with EIGHT parts (what a full node runs; this box has one device, so all eight on device 0): the
    // shards' bytes equal ONE context's on the same streams, through Encode and through the pinned pipeline; the overflow
    // counters of a part report what the oracle counts for that part's streams ----
    {
        const int S = 19, n = 7;   // 19 streams over 8 parts: shards of 3, 3, 3, 2, 2, 2, 2, 2
        std::vector<float> batch((size_t)S * n * 2048);
        for (int s = 0; s < S; ++s)
            for (int i = 0; i < n * 2048; ++i)
                batch[(size_t)s * n * 2048 + i] = pcm[(size_t)((i + 2048 * s) % (nb * 2048))] * (s == 4 ? 30.0f : 1.0f);   // stream 4: above full scale
        TAtrac3EncoderSettings st;
        TAtrac3EncoderBatch one(st, S, n);
        std::vector<uint8_t> ref1, got8, piped8;
        const int nf1 = one.Encode(batch.data(), n, ref1);
        TAtrac3EncoderNode node(st, S, n, {0, 0, 0, 0, 0, 0, 0, 0});
        EXPECT(node.Devices() == 8);
        EXPECT(node.Encode(batch.data(), n, got8) == nf1 && got8 == ref1);
        EXPECT(node.EncodePipelined(batch.data(), n, 3, piped8) == nf1 && piped8 == ref1);
        unsigned long long want[2] = {0, 0};
        at3o_diag_counts(nullptr, 1);
        for (int s = 0; s < S; ++s) {
            std::vector<unsigned char> exp((size_t)n * 1024);
            int fsz = 0;
            EXPECT(at3o_encode(132300, 2, 0, 0, 0, batch.data() + (size_t)s * n * 2048, n, exp.data(), &fsz, nullptr) == nf1);
            EXPECT(memcmp(ref1.data() + (size_t)s * nf1 * fsz, exp.data(), (size_t)nf1 * fsz) == 0);
        }
        at3o_diag_counts(want, 1);
        const at3hip_counters c = one.Counters();
        EXPECT(want[0] > 0 && c.scale_overflow == want[0] && c.clipped_values == want[1]);
        EXPECT(one.Counters(true).scale_overflow == want[0] && one.Counters().scale_overflow == 0);
        printf("TAtrac3EncoderNode (8 contexts on one device) equals one context; overflow counters %llu / %llu\n", (unsigned long long)c.scale_overflow,
               (unsigned long long)c.clipped_values);
    }
    // ---- TAt3PEncoder: 5 stereo frames of 2048 samples through a batch of 2; look-ahead call, silent first frame ----
    {
        const int nfr = 5;
        std::vector<std::vector<char>> frames;
        {
            TAt3PEncoder enc(TCompressedOutputPtr(new TMemOut(&frames)), 2, 2);
            auto lambda = enc.GetLambda();
            for (int f = 0; f < nfr; ++f) {
                const auto r = lambda(pcm.data() + (size_t)f * 4096, ProcessMeta{2});
                EXPECT((f == 0) == (r == EProcessResult::LOOK_AHEAD));
            }
        }   // destructor flushes
        EXPECT((int)frames.size() == nfr - 1);
        std::vector<float> specs((size_t)(nfr + 1) * 2 * 2048, 0.0f);   // frame 0: silence, frame k + 1: input frame k
        for (int ch = 0; ch < 2; ++ch) {
            std::vector<float> mono((size_t)nfr * 2048), bands((size_t)nfr * 2048), sp((size_t)nfr * 2048);
            for (size_t i = 0; i < mono.size(); ++i) mono[i] = pcm[2 * i + ch];
            at3po_pqf_analyse(mono.data(), nfr, bands.data());
            for (auto& v : bands) v = (float)(v / (32768.0 / 1.122018));
            at3po_mdct(bands.data(), nullptr, nfr, sp.data());
            for (int f = 0; f < nfr; ++f) memcpy(&specs[((size_t)(f + 1) * 2 + ch) * 2048], &sp[(size_t)f * 2048], 2048 * sizeof(float));
        }
        std::vector<uint8_t> exp((size_t)(nfr + 1) * 2048);
        EXPECT(at3po_write_frames(specs.data(), nullptr, 2, nfr + 1, exp.data(), nullptr) == nfr + 1);
        for (int i = 0; i < nfr - 1 && i < (int)frames.size(); ++i) {
            EXPECT(frames[i].size() == 2048);
            EXPECT(memcmp(frames[i].data(), exp.data() + (size_t)i * 2048, 2048) == 0);
        }
        printf("TAt3PEncoder: %d frames compared\n", (int)frames.size());
    }
    printf(fails ? "HOST SHIM TEST FAILED\n" : "HOST SHIM TEST OK\n");
    return fails ? 1 : 0;
}
