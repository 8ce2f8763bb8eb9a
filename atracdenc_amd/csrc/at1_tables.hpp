// Constant tables of the ATRAC1 encode path, built once on the host with the container's
// libm - the same expressions, evaluated in the same types, as the reference's static initialisers - and uploaded
// to HBM at at1hip_create. Reference: atrac/at1/atrac1.h:83-133, atrac/at1/atrac1_bitalloc.cpp:38-92,130-149,
// qmf/qmf.cpp:25-44, lib/mdct/mdct.cpp:25-36, atrac/atrac_psy_common.cpp:126-156, transient_detector.cpp:48-53.
#pragma once
#include <cstdint>

#include "at3_tables.hpp"

namespace at1 {

using at3::cpx;

constexpr int kMaxBfus = 52;
constexpr int kFrame = 212;    // TAtrac1Data::SoundUnitSize
constexpr int kSamples = 512;  // TAtrac1Data::NumSamples

struct Tables {
    float qmf_win[48];
    float scale[64];      // ScaleTable
    float sine[32];       // SineWindow
    float sc512[256];     // TMDCT<512>(1) SinCos
    float sc256[128];     // TMDCT<256>(0.5)
    float sc64[32];       // TMDCT<64>(0.5)
    cpx tw128[128];       // kissfft forward twiddles of the N/4-point cores
    cpx tw64[64];
    cpx tw16[16];
    float loud[512];      // CreateLoudnessCurve(512)
    float ath_bfu[52];    // At1ATHLong
    float fir[10];        // TTransientDetector::HPFilter taps
    float fix_long[52];   // FixedBitAllocTableLong
    float fix_short[52];  // FixedBitAllocTableShort
    // logf of glibc 2.35 (FMA build): {1/c, ln c} x 16, ln 2, degree-3 polynomial
    double logf_tab[16][2];
    double logf_ln2;
    double logf_poly[3];
};

void build_tables(Tables* t);

}  // namespace at1
