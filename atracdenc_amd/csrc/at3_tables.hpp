// Host-built constant tables of the ATRAC3 encode path, uploaded once per context.
//
// The reference builds these with libm at start-up (atrac/at3/atrac3.h:178-198, qmf/qmf.cpp:36-45,
// lib/mdct/mdct.cpp:25-36, kiss_fft.c:357-363, tools/kiss_fftr.c:51-57,
// transient_spectral_upsampler.cpp:52-68, atrac_psy_common.cpp:126-156, atrac3_bitstream.cpp:694-718).
// Bit-exact parity needs the same table values, so they are generated on the host with the same
// libm expressions and never recomputed with device transcendentals.
#pragma once
#include <stdint.h>

#include "at3_libm64.hpp"

namespace at3 {

struct cpx {
    float r, i;
};

struct Tables {
    float qmf_win[48];
    float scale[64];        // ScaleTable
    float enc_win[256];     // EncodeWindow
    float gain_level[16];
    float gain_interp[32];  // 31 used
    float mdct_sincos[256];
    float planck[512];
    float hpf_w[4];         // raised-cosine transition weights, index i = 0..2
    float loud_curve[1024];
    float ath_bfu[32];
    cpx tw128[128];         // forward, MDCT core
    cpx tw256[256];         // forward, rfft-512 core
    cpx stw256[128];        // rfft-512 super twiddles
    cpx tw2048[2048];       // inverse, irfft-4096 core
    cpx stw2048[1024];      // irfft-4096 super twiddles
    double log2f_tab[16][2];
    double log2f_poly[4];
    // k_gain_analysis: the 27 twiddles work-item t of a 128-thread workgroup needs for the last three passes of the
    // irfft-4096 core, as [slot][t] so that a wavefront fetches each slot as one contiguous 512-byte run (picking them out
    // of tw2048 touched up to 32 cache lines per load). Slots: 0..2 pass m = 32, 3..14 pass m = 128, 15..26 pass m = 512.
    cpx gain_tw[27][128];
    // k_gain_analysis1 (one wavefront per item), lane = 16 R + j:
    //   ga1_twb[u][s][j]   the fifteen twiddles of unit k = 2 j + u of passes m = 32 / 128 (s as in gain_tw: 0..2 tw[16 (s + 1) k],
    //                      3 + 3 jj + q: tw[4 (q + 1) (k + 32 jj)]); the four rows of lanes read the same 128 bytes
    //   ga1_twc[4 u + t][q][lane]  pass m = 512: tw[(q + 1) kk] of butterfly kk = 2 j + u + 32 (4 R + t): 512 contiguous bytes per fetch
    cpx ga1_twb[2][15][16];
    cpx ga1_twc[8][3][64];
    // k_gain_spec: a 256-point transform lives in the 16 lanes of a DPP row, position L = lane & 15 (see at3_k_gain.hpp):
    //   spec16_win[t][L]  Planck window pair {w[2 i], w[2 i + 1]} of complex input i = L + 16 t
    //   spec16_tw[s][L]   s = 0..2: tw256[4 (s + 1) L] (pass m = 16, butterfly k = L);
    //                     s = 3 + 3 j + q - 1: tw256[q (L + 16 j)], j = 0..3, q = 1..3 (pass m = 64, butterfly k = L + 16 j)
    //   spec16_stw[jj][L] super twiddle stw256[k - 1] of bin k = L + 16 jj, jj = 0..7 (k = 0: unused); [8][0] = stw256[127] (bin 128)
    // k_mdct_sub / k_qmf_mdct8: the row transform's per-lane constants, entry e of row position L at [e][L] (see MdctTab in
    // at3_k_front2.hpp for what the 18 entries are) - copied into LDS by every workgroup, one 16-byte load per work-item
    float mdct_tab[18][16][4];
    cpx spec16_win[16][16];
    cpx spec16_tw[15][16];
    cpx spec16_stw[9][16];
    // glibc 2.35's f64 log / exp data (at3_libm64.hpp): the literal form of CalcSpectralFlatnessPerBfu in k_psy
    Libm64 libm;
};

// Fills *t on the host. Pure function of libm.
void build_tables(Tables* t);

}  // namespace at3
