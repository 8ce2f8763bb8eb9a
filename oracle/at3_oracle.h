/*
 * TEST INFRASTRUCTURE ONLY.
 *
 * at3_oracle: a from-scratch scalar C restatement of the ATRAC3 encode hot path of
 * dcherednik/atracdenc (QMF -> gain control -> windowed MDCT-512 -> tonal extraction ->
 * scale factors -> bit allocation -> mantissa quantisation -> sound-unit packing).
 *
 * It is the CPU parity anchor for the HIP path on the GPU box (where /root/reference does
 * not exist). Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * it; the product library (atracdenc_amd/csrc) never links or calls it.
 *
 * Parity pinning: validated bit-for-bit against the real reference (oracle/_ref, built from
 * the unmodified reference sources by oracle/Makefile) by tests/test_oracle_vs_ref.py and
 * against the committed golden vectors under tests/golden/ (generated from oracle/_ref by
 * tools/gen_golden.py).
 */
#ifndef AT3_ORACLE_H
#define AT3_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Per channel-frame tap record; identical layout to ref_tap in oracle/ref/ref_harness.cpp. */
typedef struct at3o_tap {
    int32_t n_points[4];
    int32_t level[4][8];
    int32_t loc[4][8];
    float ges_frame[4];
    float loudness_ch;
    float loudness_track;
    int32_t sfi[32];
    float energy[32];
    float values[1024];
    int32_t n_tonal;
    int32_t tonal_pos[64];
    int32_t tonal_len[64];
    int32_t tonal_sfi[64];
    float tonal_values[64][8];
} at3o_tap;

typedef struct at3o_encoder at3o_encoder;

/* bitrate in bit/s as the reference's TAtrac3EncoderSettings takes it (0 -> LP2 132300). */
at3o_encoder* at3o_create(int bitrate, int nch, int no_gain, int no_tonal, int bfu_idx_const);
void at3o_destroy(at3o_encoder* e);
int at3o_frame_size(const at3o_encoder* e);
int at3o_joint_stereo(const at3o_encoder* e);
/* One lambda call of the reference: 1024*nch interleaved samples in. Returns 0 for the
 * LOOK_AHEAD call (nothing written) and 1 when a frame of frame_size bytes was written. */
int at3o_process(at3o_encoder* e, const float* pcm, unsigned char* out, at3o_tap* taps);
/* Whole-stream convenience, same signature as ref_encode. */
int at3o_encode(int bitrate, int nch, int no_gain, int no_tonal, int bfu_idx_const,
                const float* pcm, int nblocks, unsigned char* out, int* frame_sz, at3o_tap* taps);

/* Stage-level entry points (same signatures as the ref_* taps). */
void at3o_qmf(const float* pcm, int nblocks, float* sub);
void at3o_mdct(float* specs, float* bands, const int32_t* n_points, const int32_t* level, const int32_t* loc);
void at3o_gain_energy_scale(const float* prevOverlap, const float* cur, int n_points, const int32_t* level,
                            const int32_t* loc, float prevScale, float* out);
void at3o_upsample(const float* in512, float* out4096, float* hfr);
void at3o_analyze_gain(const float* in, int len, int maxPoints, float* gain, float* lo, float* hi);
int at3o_calc_curve(const float* gain32, float* ctx, float minScore, const float* lo, const float* hi,
                    int32_t* level, int32_t* loc);
int at3o_relation_to_idx_hdr(float x);
float at3o_quant_mantisas(const float* in, int n, float mul, int ea, int32_t* mant);
void at3o_scale_frame(const float* specs, int32_t* sfi, float* energy, float* values);
/* TScaler::Scale's stderr diagnostics as process-wide counts since the last reset: out2[0] = "Scale error" lines
 * (atrac_scale.cpp:150-153), out2[1] = "clipping" lines (:163-167). */
void at3o_diag_counts(unsigned long long* out2, int reset);
void at3o_flatness(const float* energy1024, float* flat32);
float at3o_log2f(float x);
void at3o_tables(float* scale64, float* encwin256, float* gainlevel16, float* gaininterp31,
                 float* qmfwin48, float* loud1024, float* ath1024);
void at3o_mdct512(const float* in512, float* out256);
/* libstdc++-order-compatible sort of (key,payload) pairs by |key| (see QuantMantisas). */
void at3o_sort_abs(float* key, int32_t* payload, int n);

#ifdef __cplusplus
}
#endif
#endif
