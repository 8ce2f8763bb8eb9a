// Constant tables of the ATRAC3plus front end, built on the host with the container's libm like
// the reference's constructors do. Reference: atrac/atrac3plus_pqf/atrac3plus_pqf.c:50-80 (prototype arrangement),
// lib/mdct/mdct.cpp:25-36,55-66 (rotation tables of TMIDCT<32> / TMDCT<256>), atrac/at3p/at3p_mdct.cpp:33-46 (windows).
#pragma once
#include <cstdint>

#include "at3_tables.hpp"

namespace at3p {

using at3::cpx;

struct Tables {
    float fir[384];      // analysis prototype, 32 rows x 12 taps
    float sc32[16];      // TMIDCT<32>(32 * 128 * 512): the 16-point DCT-IV of the PQF matrixing
    float sc256[128];    // TMDCT<256>(1)
    cpx tw8[8];          // kissfft forward twiddles
    cpx tw64[64];
    float sine128[128];  // SineWin128
    float sine64[64];    // SineWin64
};

void build_tables(Tables* t);

}  // namespace at3p
