// Bit allocation + quantisation + sound-unit packing in ONE kernel, one wavefront per (stream, output frame, channel).
//
// Reference path replaced (paths relative to the reference's src/):
//   atrac/at3/atrac3_bitstream.cpp:92-847   CLC/VLC cost + emission, CalcBitsAllocation, ConsiderEnergyErr, tonal
//                                           component grouping/coding, TConfigure/TAlloc, WriteSoundUnit
//   atrac/atrac_enc_cache.cpp, atrac3_bitstream.cpp:154-173   TEncCache: quantised units computed ON DEMAND
//   atrac/atrac_scale.cpp:40-130            QuantMantisas (rounding, energy sums, energy-adaptive re-rounding)
//   lib/bs_encode/encode.cpp:57-129         bisection driver (Start / Continue / Submit / Repeat)
//
// The rate loop asks for quantised (BFU, wordlen) units as its bisection walks: on typical material about 100 of the
// 224 possible units and a third of their spectral lines (a quarter of the lines that need the energy-adaptive pass).
// The first version quantised all 224 up front in a separate kernel; then a unit was computed the first time an
// allocation asked for it, as the reference's cache does; now it is brought in only as far as the bisection's comparison
// needs it: as a LOWER BOUND of its VLC bits first (unit_bounds: rounding + one length look-up per line), and through
// QuantMantisas' energy-adaptive pass only when no bound decides the comparison or the evaluation ends the bisection
// (compute_units; on white noise a third of the units the reference quantises). By the wavefront that runs the loop:
//   rounding and code lengths of the batch's lines       16 lines per lane
//   the ordered energy sum of every new unit              one lane per unit
//   energy-adaptive candidates (BFU > 18): compaction by ballot, rank by counting smaller keys, exact std::sort order
//   on key ties, then the sequential re-rounding pass     one lane per unit
// Lane i < 32 owns BFU i in the loop itself (allocation, cost look-up, DPP reductions).
#pragma once
#include "at3_k_backend.hpp"

namespace at3 {

// Profiling builds (-DAT3HIP_DEBUG_KNOBS): every wavefront of k_alloc_pack stamps its phase boundaries with the shader-cycle
// counter and adds the cycles it spent per phase to AT3HIP_TAP_CLOCK's slots 2.. (tools/alloc_phase_cycles.sh) - where a
// wavefront's LIFE goes, measured on the real path (the stage exits of debug_stop leave the unit cache empty, so what follows
// them is not the real bisection).
#ifdef AT3HIP_DEBUG_KNOBS
struct PhaseClock {
    unsigned long long last;
    unsigned long long* slots;   // this wavefront's row of 12 counters (256 rows, by workgroup index: no contention to speak of)
    unsigned long long* item;    // the first 16384 workgroups also keep their own 12 (tools/alloc_item_times.sh)
};
#define AT3_PH_END(pc, k)                                                                    \
    do {                                                                                     \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                          \
        if ((pc).slots && threadIdx.x == 0) atomicAdd((pc).slots + (k), t_ - (pc).last);     \
        if ((pc).item && threadIdx.x == 0) (pc).item[k] += t_ - (pc).last;                   \
        (pc).last = t_;                                                                      \
    } while (0)
#else
struct PhaseClock {};
#define AT3_PH_END(pc, k) ((void)0)
#endif
// -DAT3_LOOP_PHASES (with AT3HIP_DEBUG_KNOBS; tools/alloc_loop_phases.sh): the rate loop's own parts instead of the units' - slots 5..9 =
// trip head + memo look-up, allocation + tonal side information, sums + decision, record + comparison + interval, BFU drops; slot 10 = units + emission
#ifdef AT3_LOOP_PHASES
#define AT3_UPH(pc, k, kl) do { if ((kl) >= 0) AT3_PH_END(pc, kl); } while (0)
#define AT3_LPH(pc, k) AT3_PH_END(pc, k)
#else
#define AT3_UPH(pc, k, kl) AT3_PH_END(pc, k)
#define AT3_LPH(pc, k) ((void)0)
#endif

// SIMT harness only (tools/emu, AT3_EMU_HOST): the rate loop counts what it does - and checks every lower bound of unit_bounds
// against the bits compute_units finds later - in g_alloc_stats (tools/emu/emu_runtime.cpp; printed and asserted by run_emu.py):
// 0 trips, 1 hits of exact records, 2 evaluations in place of bound records, 3 evaluations, 4 unit_bounds calls, 5 their units,
// 6 compute_units calls, 7 their units, 8 / 9 comparisons decided by the upper / lower bound, 10 channel-frames,
// 11 bounds compared with the bits, 12 bounds ABOVE the bits (must stay zero)
#ifdef AT3_EMU_HOST
extern "C" unsigned long long g_alloc_stats[16];
#define AT3_STAT(k, n) do { if (lane == 0) g_alloc_stats[k] += (n); } while (0)
#else
#define AT3_STAT(k, n) ((void)0)
#endif
constexpr int kTermLine0 = 96;   // BFUs 0..9 (lines 0..95) are quantised by small_units: no batch ever lists their lines
constexpr int kChgWords = kEaLines / 32;   // 23: one bit per line from BFU 19 on (every such BFU covers whole words)
struct AllocLds {
    float val[1024];                 // scaled spectrum (TScaler::Scale)
    union {
        float term[1024 - kTermLine0];   // a batch's energy terms (mantissa / mul)^2 of line i at i - kTermLine0, summed in line order by one lane per unit
        struct {
            float uk[256];               // then the energy-adaptive pass: ONE unit's (or pair's) sort keys (+inf padded); tie-sort scratch
            uint16_t rec[kEaLines];      // and every unit's candidates ordered by |delta|: line | |m| << 7 | negative << 12
        };
        struct {
            uint32_t words[kBitWords];   // after the rate loop: the sound unit being assembled
            uint16_t huff[130];          // and the code tables
        };
    };
    unsigned long long tmask[4];
    union {
        float err[7 * 10];               // e1 / e2 of (wordlen, BFU < 10) at (wordlen - 1) * 10 + BFU: read once, for ConsiderEnergyErr's map
        // from then on: which lines the energy-adaptive pass re-rounded, per wordlen - with plain rounding everything the
        // emission needs to form the chosen units' mantissas again (they never go to memory): bit (line - 288) of row wordlen - 1
        uint32_t chg[7 * kChgWords];
    };
    uint16_t cost[7 * 32];           // cache: VLC bits of (wordlen, BFU) at (wordlen - 1) * 32 + BFU (the CLC bits are wordlen x lines)
    int misc[4];
    int8_t bm[1024 - kTermLine0];    // mantissas of the units of the current batch (one wordlen per BFU), line i at i - kTermLine0
    uint8_t code[kEaLines / 4];      // 2 bits per line of the batch from BFU 19 on: 1 = re-roundable when e2 < e1, 2 = when e2 > e1
    uint8_t alloc[32];
    uint8_t tbits[kMaxTonal * 6];    // VLC bits of tonal block t at quantiser q = 2..7: [t * 6 + q - 2]
};
static_assert(sizeof(AllocLds) <= 10240, "sixteen workgroups per CU");
static_assert(sizeof(float) * 256 + sizeof(uint16_t) * kEaLines + 72 * sizeof(int) <= sizeof(float) * (1024 - kTermLine0), "tie-sort stack behind the candidate records");
static_assert(sizeof(uint32_t) * kBitWords + sizeof(uint16_t) * 132 + 5 * kMaxTonal + 24 <= sizeof(float) * (1024 - kTermLine0), "tonal walk scratch behind the code tables");
static_assert(sizeof(SortItem) * 128 <= sizeof(float) * 256, "tie-sort scratch must fit in the key list");

// 1 / MaxQuant[wl]^2 as QuantMantisas forms it (atrac_scale.cpp:61: float(1.0 / double(mul * mul))), folded per wordlen
__device__ __forceinline__ float inv_mul2(int wl)
{
    switch (wl) {
        case 1: return (float)(1.0 / (double)(1.5f * 1.5f));
        case 2: return (float)(1.0 / (double)(2.5f * 2.5f));
        case 3: return (float)(1.0 / (double)(3.5f * 3.5f));
        case 4: return (float)(1.0 / (double)(4.5f * 4.5f));
        case 5: return (float)(1.0 / (double)(7.5f * 7.5f));
        case 6: return (float)(1.0 / (double)(15.5f * 15.5f));
        default: return (float)(1.0 / (double)(31.5f * 31.5f));
    }
}

// CLC bits of a unit: clc_len(wordlen) per line, two per line at wordlen 1 (pairs of 4 bits, atrac3_bitstream.cpp:77-90)
__device__ __forceinline__ uint32_t clc_bits(int wl, int n_lines) { return (wl > 1) ? (uint32_t)clc_len(wl) * (uint32_t)n_lines : 2u * (uint32_t)n_lines; }

__device__ __forceinline__ float readlane_f(float v, int l) { return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), l)); }
// MantissasToVlcIndex's 3x3 table {8, 4, 7, 2, 0, 1, 6, 3, 5} (atrac3_bitstream.cpp:92-113) as nibbles of one constant
__device__ __forceinline__ uint32_t vlc_pair_index(int m0, int m1) { return (uint32_t)((0x536102748ull >> (4 * (3 * (m0 + 1) + (m1 + 1)))) & 15ull); }

// Huffman code LENGTH of mantissa m under selector wl >= 2, and of a mantissa pair under selector 1 (VLCEnc with a null
// stream, atrac3_bitstream.cpp:115-149). A code's length depends on |m| only (the sign is the code's last bit), so each
// table is a row of 4-bit lengths indexed by |m|, held in one or two 64-bit constants instead of a memory look-up;
// tests/test_abi.py::test_vlc_length_constants re-derives the constants from the code table in at3_common.hpp.
struct VlcRow {
    unsigned long long lo, hi;   // lengths of |m| = 0..15 and 16..31 (the same row twice below wordlen 7)
};
// The rows, ONE definition for the code lengths (vlc_row) and for unit_bounds' lower-bound rows derived from them (lb_row_of):
constexpr unsigned long long kVlcLo[8] = {0ull, 0ull, 0x331ull, 0x4431ull, 0x55431ull, 0x46654432ull, 0x4777766665554443ull, 0x7766666666555553ull};   // selector wl, |m| = 0..15
constexpr unsigned long long kVlcHi7 = 0x4888888888877777ull;   // selector 7, |m| = 16..31
constexpr int kVlcTop[8] = {0, 0, 1, 3, 4, 7, 15, 31};          // largest |m| rounding produces at wordlen wl (MaxQuant: 1.5, 3.5, 4.5, 7.5, 15.5, 31.5 -> wordlen 2 stops at 1)
__device__ __forceinline__ VlcRow vlc_row(int wl)
{
    unsigned long long k;
    switch (wl) {   // (a switch over compile-time constants: the rows stay immediates, indexing the array would be a memory look-up)
        case 2: k = kVlcLo[2]; break;
        case 3: k = kVlcLo[3]; break;
        case 4: k = kVlcLo[4]; break;
        case 5: k = kVlcLo[5]; break;
        case 6: k = kVlcLo[6]; break;
        default: k = kVlcLo[7]; break;   // wl 7, |m| <= 15
    }
    VlcRow r;
    r.lo = k;
    r.hi = wl == 7 ? kVlcHi7 : k;   // wl 7, 16 <= |m| <= 31
    return r;
}
__device__ __forceinline__ uint32_t vlc_len(const VlcRow& r, int m)
{
    const uint32_t a = (uint32_t)(m < 0 ? -m : m);
    return (uint32_t)((a < 16 ? r.lo : r.hi) >> (4 * (a & 15u))) & 15u;
}
__device__ __forceinline__ uint32_t vlc_len(int wl, int m) { return vlc_len(vlc_row(wl), m); }

// unit_bounds' rows (described there): nibble x = min(len(x), len(x + 1))
constexpr unsigned long long lb_row(unsigned long long row, int top, unsigned next)   // nibble x: min(len(x), len(x + 1)), x + 1 <= top
{
    unsigned long long r = 0;
    for (int x = 0; x < 16; ++x) {
        const unsigned a = (unsigned)((row >> (4 * x)) & 15u);
        const unsigned b = x == 15 ? next : (unsigned)((row >> (4 * (x + 1))) & 15u);
        r |= (unsigned long long)((x + 1 <= top && b < a) ? b : a) << (4 * x);
    }
    return r;
}
constexpr unsigned long long kLbHi7 = lb_row(kVlcHi7, 15, 0u);                    // (x = 31 is the top code: no x + 1)
__device__ __forceinline__ unsigned long long lb_row_of(int wl)
{
    switch (wl) {
        case 2: return lb_row(kVlcLo[2], kVlcTop[2], 0u);
        case 3: return lb_row(kVlcLo[3], kVlcTop[3], 0u);
        case 4: return lb_row(kVlcLo[4], kVlcTop[4], 0u);
        case 5: return lb_row(kVlcLo[5], kVlcTop[5], 0u);
        case 6: return lb_row(kVlcLo[6], kVlcTop[6], 0u);
        default: return lb_row(kVlcLo[7], kVlcTop[7], (unsigned)(kVlcHi7 & 15ull));   // wl 7, |m| <= 15: x + 1 = 16 is the first code of the upper row
    }
}
// Per-wordlen constants as a table ACROSS the lanes (lane k & 7 holds the entries of wordlen k): the wordlen differs from
// lane to lane wherever these are needed, where a `switch` is a chain of compares and selects per constant; a cross-lane
// read (ds_bpermute, no LDS storage, not a vector-ALU instruction) fetches all of them with one address. Every lane must
// be active at a lookup (an inactive lane's entry reads as zero): lookups stand outside divergent code.
struct LaneTab {
    float mq, inv;       // max_quant(k), inv_mul2(k)
    uint32_t lo0, lo1;   // vlc_row(k).lo
    uint32_t bfus;       // NOT a table: the BFUs of this lane's four line quads, bfu_of_line(256 h + 4 lane) in byte h
    uint32_t misc;       // CLC bits per line {0, 2, 3, 3, 4, 4, 5, 6}[k] (a pair's four bits at wordlen 1 are two per line) | huff_off(k) << 8
};
__device__ __forceinline__ LaneTab lane_tab(int lane)
{
    const int k = lane & 7;
    const VlcRow r = vlc_row(k);
    const unsigned long long lo = (lane & 8) ? lb_row_of(k) : r.lo;   // lanes 8..15: the rows of unit_bounds' lower bound
    LaneTab t;
    t.mq = max_quant(k);
    t.inv = inv_mul2(k);
    t.lo0 = (uint32_t)lo;
    t.lo1 = (uint32_t)(lo >> 32);
    t.bfus = (uint32_t)opaque_lane_value(bfu_of_line(4 * lane) | (bfu_of_line(256 + 4 * lane) << 8) | (bfu_of_line(512 + 4 * lane) << 16) |
                                         (bfu_of_line(768 + 4 * lane) << 24));
    t.misc = (uint32_t)(k == 1 ? 2 : clc_len(k)) | ((uint32_t)huff_off(k) << 8);
    return t;
}
__device__ __forceinline__ float tab_f(float v, int wl) { return __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(4 * wl, (int)__float_as_uint(v))); }
__device__ __forceinline__ uint32_t tab_u(uint32_t v, int wl) { return (uint32_t)__builtin_amdgcn_ds_bpermute(4 * wl, (int)v); }
__device__ __forceinline__ VlcRow tab_row(const LaneTab& t, int wl)
{
    const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * wl, (int)t.lo0), b = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * wl, (int)t.lo1);
    VlcRow r;
    r.lo = (unsigned long long)a | ((unsigned long long)b << 32);
    r.hi = wl == 7 ? kVlcHi7 : r.lo;
    return r;
}
__device__ __forceinline__ uint32_t vlc_pair_len(int m0, int m1) { return (uint32_t)((0x545313545ull >> (4 * (3 * (m0 + 1) + (m1 + 1)))) & 15ull); }

__device__ __forceinline__ uint32_t vlc_bits8(int wl, const VlcRow& row, const int (&m)[8])
{
    uint32_t vb = 0;
    if (wl > 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) vb += vlc_len(row, m[k]);
    } else {
#pragma unroll
        for (int k = 0; k < 8; k += 2) vb += vlc_pair_len(m[k], m[k + 1]);
    }
    return vb;
}

// The VLC bit count of the units of round h (lines 256 h + 4 lane .. + 3 per lane) from every lane's four-line count `vb`: a unit's
// lines sit in 4, 8, 16 or 32 NEIGHBOURING lanes of one round (16-, 32-, 64-, 128-line BFUs from line 96 on, all aligned to their
// own size), so its bit count is a sum over a quad, a half row, a row or two rows: DPP adds, no LDS traffic (the first version added
// every lane's count to a per-BFU LDS counter: up to 32 lanes on one address). The unit's first lane stores it in the cache.
__device__ __forceinline__ void unit_cost_store(AllocLds& L, const LaneTab& tab, int h, int wl, uint32_t vb, int lane)
{
    vb += (uint32_t)AT3_DPP(vb, 0xB1, false);   // quad_perm [1, 0, 3, 2]
    vb += (uint32_t)AT3_DPP(vb, 0x4E, false);   // quad_perm [2, 3, 0, 1]: every lane of a quad holds the quad's 16 lines
    bool lead = (lane & 3) == 0;
    if (h == 0) {   // lines 96..191: 16-line BFUs, lines 192..255: two 32-line BFUs
        const uint32_t v8 = vb + (uint32_t)AT3_DPP(vb, 0x141, false);   // row_half_mirror
        if (lane >= 48) {
            vb = v8;
            lead = (lane & 7) == 0;
        }
    } else {
        vb += (uint32_t)AT3_DPP(vb, 0x141, false);
        lead = (lane & 7) == 0;
        if (h >= 2) {   // 64-line BFUs: a row each
            vb += (uint32_t)AT3_DPP(vb, 0x140, false);   // row_mirror
            lead = (lane & 15) == 0;
        }
        if (h == 3) {   // 128-line BFUs: two rows each
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)vb, 0) + (uint32_t)__builtin_amdgcn_readlane((int)vb, 16);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)vb, 32) + (uint32_t)__builtin_amdgcn_readlane((int)vb, 48);
            vb = lane < 32 ? lo : hi;
            lead = (lane & 31) == 0;
        }
    }
    if (wl && lead) L.cost[(wl - 1) * 32 + (int)((tab.bfus >> (8 * h)) & 0xffu)] = (uint16_t)vb;
}

// sum of v[i]^2, i = 0 .. n - 1 in that order (QuantMantisas' e1, atrac_scale.cpp:42-58), n a multiple of 8: eight values per step,
// the next eight requested before this step's chain of additions runs (every step used to wait for its own LDS reads)
__device__ __forceinline__ float ordered_square_sum(const float* v, int n)
{
    const float4* v4 = reinterpret_cast<const float4*>(v);
    float acc = 0.0f;
    float4 a = v4[0], b = v4[1];
    for (int off = 0; off < n; off += 8) {
        float4 na = a, nb = b;
        if (off + 8 < n) {
            na = v4[(off >> 2) + 2];
            nb = v4[(off >> 2) + 3];
        }
        const float term[8] = {a.x * a.x, a.y * a.y, a.z * a.z, a.w * a.w, b.x * b.x, b.y * b.y, b.z * b.z, b.w * b.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += term[k];
        a = na;
        b = nb;
    }
    return acc;
}

// A LOWER BOUND of the VLC bits of the units {(b, wl_b) : bit b of `need`} without the energy-adaptive pass - and below BFU 19,
// where there is no such pass, the bits themselves. The bisection compares a total with the target (lib/bs_encode/encode.cpp:
// 57-129): when the comparison is already decided by a bound of the total, the exact bits are never needed, and the pass (two fifths
// of what a unit costs: candidate lists, ranks, the sequential walk, plus the ordered energy sum it starts from) is only run for the
// units of the few allocations whose totals come too close to the target to call - in practice the final one's.
//   * The pass moves a line by ONE code, only a line close to a rounding boundary (|delta| < 0.25: as the reference forms delta,
//     positive lines only) and only in the direction AWAY from where rounding took it (atrac_scale.cpp:66-126): a line rounded
//     away from zero (|m| > |t|) may end at |m| - 1, a line rounded towards zero at |m| + 1, whatever else the pass asks of it
//     (which of the two passes runs, the top code, the energy test): the set allowed for here contains the set it moves.
//   * A code's length depends on |m| only. With lb(x) = min(len(x), len(x + 1)) a line rounded away from zero costs at least
//     lb(|m| - 1) if it is such a candidate, any other line at least lb(|m|): ONE look-up per line in the row of lb. The tables grow with |m| except for the
//     top code of wordlens 5..7, so lb(x) = len(x) but for the code below the top one: the bound is the plain-rounding bits less a
//     bit or two per line that was rounded up across a length step - within a per cent of the final bits on white noise.
//     (Pairs at wordlen 1: the pair table grows with either |m|, so the pair of the smaller magnitudes bounds it.)
// Rounding and lengths in one go, sixteen lines per lane in registers: nothing is stored but the counts; no lane conditions.
// The rows of lb sit in lanes 8..15 of the lane table's length words (tab_row reads lanes 0..7). Wave-uniform call.
__device__ __forceinline__ void unit_bounds(AllocLds& L, const LaneTab& tab, uint32_t need, int bits, int lane_, PhaseClock& pc)
{
    AT3_UPH(pc, 4, 7);
    // (the lane conditions below are formed here, where they are used: hoisted in front of the rate loop they would sit in scalar
    // registers the loop does not have - it already parks some of its masks in vector lanes)
    const int lane = opaque_lane_value(lane_);
    int wl_h[4];
    float mul_h[4];
    unsigned long long row_h[4];
    {
        int wl_r[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) wl_r[h] = __builtin_amdgcn_ds_bpermute(4 * (int)((tab.bfus >> (8 * h)) & 0xffu), bits);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            mul_h[h] = tab_f(tab.mq, wl_r[h]);
            // lines below BFU 19 (round 0, lanes 0..7 of round 1) are final as rounded: their lengths come from the table itself
            const int src = (h == 0 || (h == 1 && lane < 8)) ? wl_r[h] : wl_r[h] + 8;
            const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * src, (int)tab.lo0), b = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * src, (int)tab.lo1);
            row_h[h] = (unsigned long long)a | ((unsigned long long)b << 32);
            wl_h[h] = ((need >> ((tab.bfus >> (8 * h)) & 0xffu)) & 1u) ? wl_r[h] : 0;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int line0 = 256 * h + 4 * lane;
        const int wl = wl_h[h];
        const float mul = mul_h[h];
        const unsigned long long row = row_h[h];
        const bool ea = h > 1 || (h == 1 && lane >= 8);
        const unsigned long long hi = ea ? kLbHi7 : kVlcHi7;   // (only wordlen 7 has codes from 16 on)
        uint32_t vb = 0;
        const bool any_pairs = __ballot(wl == 1) != 0ull;   // (most rounds have no unit at wordlen 1: its pair look-ups are branched over)
        if (wl) {   // (one condition per round - a call for a few units skips whole rounds -, none per line)
            const float4 va = *reinterpret_cast<const float4*>(L.val + line0);
            const float v[4] = {va.x, va.y, va.z, va.w};
            int ap[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float t = v[k] * mul;
                const float am = fabsf(rintf(t));
                const int a = (int)am;
                // A candidate that was rounded away from zero: the reference's delta = t - (trunc(t) + 0.5) is formed with the SIGNED t, so
                // |delta| >= 0.5 for every negative t (no negative line is ever a candidate); for t > 0 rounded up to m, f = t - m is in
                // [-0.5, 0) (exact: both are multiples of t's last place) and |delta| = f + 0.5 < 0.25 means f < -0.25.
                // (bitwise: `&&` would be compiled into a branch around the second test - a lane condition per line)
                ap[k] = a - (int)((int)ea & (int)(t - am < -0.25f) & (int)(t > 0.0f));
                vb += (uint32_t)((ap[k] < 16 ? row : hi) >> (4 * (ap[k] & 15))) & 15u;
            }
            if (any_pairs) {   // (wave-uniform; at wordlen 1 |m| <= 1)
                const uint32_t vp = vlc_pair_len(ap[0] & 1, ap[1] & 1) + vlc_pair_len(ap[2] & 1, ap[3] & 1);
                vb = wl == 1 ? vp : vb;
            }
        }
        unit_cost_store(L, tab, h, wl, vb, lane);
    }
    wave_sync();
    AT3_UPH(pc, 5, 10);
}

// Quantise the units {(b, wl_b) : bit b of `need`}, wl_b = lane b's `bits` (QuantMantisas + CLC/VLC cost,
// atrac3_bitstream.cpp:154-173, atrac_scale.cpp:40-130). Lane b keeps BFU b's e1 in `my_e1` (from BFU 19 on formed here, on first use). Wave-uniform call.
__device__ __forceinline__ void compute_units(AllocLds& L, const LaneTab& tab, uint32_t need, int bits, float& my_e1, int lane_, float* qerr, PhaseClock& pc, int dbg = 0)
{
    // (lane constants - addresses, masks - are formed where they are used: since the bounds the call is rare (once or twice per channel-frame), and
    // hoisted in front of the rate loop they would hold registers across it that the loop's own code is short of)
    const int lane = opaque_lane_value(lane_);
    AT3_UPH(pc, 4, 7);
    // ---- (1) mantissa = lrint(value * MaxQuant[wl]) for the lines of the needed BFUs; energy-adaptive candidate codes ----
    // Four rounds of four lines per lane, line0 = 256 round + 4 lane: a wavefront's 16-byte LDS accesses are one contiguous
    // kilobyte (sixteen lines per lane, the first layout, put every fourth lane on the same banks).
    int wl_h[4];
    // the four rounds' cross-lane reads first, all in flight together (two dependent round trips for the call instead of two per
    // round): the wordlen of every round's BFU, then MaxQuant and 1 / MaxQuant^2 of those wordlens
    float mul_h[4], inv2_h[4];
    {
        int wl_r[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) wl_r[h] = __builtin_amdgcn_ds_bpermute(4 * (int)((tab.bfus >> (8 * h)) & 0xffu), bits);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            mul_h[h] = tab_f(tab.mq, wl_r[h]);
            inv2_h[h] = tab_f(tab.inv, wl_r[h]);
            wl_h[h] = ((need >> ((tab.bfus >> (8 * h)) & 0xffu)) & 1u) ? wl_r[h] : 0;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int line0 = 256 * h + 4 * lane;
        const float mul = mul_h[h], inv2 = inv2_h[h];
        if (wl_h[h]) {
            const float4 va = *reinterpret_cast<const float4*>(L.val + line0);
            const float v[4] = {va.x, va.y, va.z, va.w};
            uint32_t pk = 0u, code = 0;
            float tm[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float t = v[k] * mul;
                // (the square of the mantissa as a product of floats: |m| <= 31, so m and m * m are exact in fp32 and (float)(m * m) is
                // the same number - an integer multiply is a slow 64-bit encoded instruction here, a float one is not)
                const float r = rintf(t);
                const int m = (int)r;
                pk |= (uint32_t)(uint8_t)m << (8 * k);
                tm[k] = (r * r) * inv2;
                // the pass may re-round a line only when it is close to a rounding boundary (|delta| < 0.25) AND lies on the
                // side the pass moves: rounded towards zero and below the top code (pass taken when e2 < e1) or rounded
                // away from zero (e2 > e1), atrac_scale.cpp:66-126; which pass runs is known after the energy sums
                if (h > 0) {   // (BFU 19, the first one with the pass, starts at line 288: round 1's lanes 0..7 form codes nobody stores)
                    const float am = fabsf((float)m), at = fabsf(t);
                    const float delta = t - (truncf(t) + 0.5f);
                    const uint32_t c = (am < at && am < (mul - 1)) ? 1u : (am > at) ? 2u : 0u;
                    code |= (fabsf(delta) < 0.25f ? c : 0u) << (2 * k);
                }
            }
            *reinterpret_cast<uint32_t*>(L.bm + (line0 - kTermLine0)) = pk;
            if (h > 0 && line0 >= kEaLine0) L.code[(line0 - kEaLine0) >> 2] = (uint8_t)code;
            *reinterpret_cast<float4*>(L.term + (line0 - kTermLine0)) = make_float4(tm[0], tm[1], tm[2], tm[3]);
        }
    }
    wave_sync();
    AT3_UPH(pc, 5, -1);
#ifdef AT3HIP_DEBUG_KNOBS
    if (dbg == 9) return;
#endif
    // ---- (2) e2 = sum of (mantissa / mul)^2, strictly in line order: one lane per unit ----
    const bool mine = lane < 32 && ((need >> lane) & 1u);
    const int my_start = bfu_start(lane & 31), my_n = bfu_start((lane & 31) + 1) - my_start;
    float my_inv2 = tab_f(tab.inv, bits), my_e2 = 0.0f;
    if (mine && lane > 18 && my_e1 < 0.0f) my_e1 = ordered_square_sum(L.val + my_start, my_n);   // (the first unit of this BFU to need the pass)
    if (mine && (lane > 18 || qerr)) {   // below BFU 19 nothing but the QUANT tap reads a unit's quantised energy
        const float4* t4 = reinterpret_cast<const float4*>(L.term + (my_start - kTermLine0));
        float acc = 0.0f;
        // sixteen terms per step (the units here are 16 to 128 lines long), the next sixteen in flight in a second set of
        // registers: the two sets swap roles from one step to the next instead of being copied
        float4 a0 = t4[0], a1 = t4[1], a2 = t4[2], a3 = t4[3];
        float4 b0 = a0, b1 = a1, b2 = a2, b3 = a3;
        for (int off = 0;;) {
            const bool more_b = off + 16 < my_n;
            if (more_b) {
                b0 = t4[(off >> 2) + 4];
                b1 = t4[(off >> 2) + 5];
                b2 = t4[(off >> 2) + 6];
                b3 = t4[(off >> 2) + 7];
            }
            acc += a0.x; acc += a0.y; acc += a0.z; acc += a0.w;
            acc += a1.x; acc += a1.y; acc += a1.z; acc += a1.w;
            acc += a2.x; acc += a2.y; acc += a2.z; acc += a2.w;
            acc += a3.x; acc += a3.y; acc += a3.z; acc += a3.w;
            if (!more_b) break;
            off += 16;
            const bool more_a = off + 16 < my_n;
            if (more_a) {
                a0 = t4[(off >> 2) + 4];
                a1 = t4[(off >> 2) + 5];
                a2 = t4[(off >> 2) + 6];
                a3 = t4[(off >> 2) + 7];
            }
            acc += b0.x; acc += b0.y; acc += b0.z; acc += b0.w;
            acc += b1.x; acc += b1.y; acc += b1.z; acc += b1.w;
            acc += b2.x; acc += b2.y; acc += b2.z; acc += b2.w;
            acc += b3.x; acc += b3.y; acc += b3.z; acc += b3.w;
            if (!more_a) break;
            off += 16;
        }
        my_e2 = acc;
    }
    wave_sync();   // the terms' storage becomes the key list and the candidate records
    AT3_UPH(pc, 6, -1);
#ifdef AT3HIP_DEBUG_KNOBS
    if (dbg == 10) return;
#endif
    // ---- (3) energy-adaptive re-rounding of the new units above BFU 18 (atrac_scale.cpp:66-128) ----
    // A line is a candidate when it passes the side test of the pass that will run (skipped candidates change no state),
    // and the pass visits the candidates by ascending |delta|: the position of a candidate is the number of keys of its unit
    // below its own. Units are walked one after the other (uniform), a unit's lines by the lanes; the candidate's rank, its
    // current |mantissa| and the sign its new mantissa would get go into a 16-bit record at the rank's position.
#ifdef AT3HIP_DEBUG_KNOBS
    const uint32_t ea_need = dbg == 5 ? 0u : (need & 0xfff80000u);
#else
    const uint32_t ea_need = need & 0xfff80000u;
#endif

    int my_nc = 0;   // lane 19 + ub: candidates of its unit
    if (ea_need) {
        uint32_t tie_units = 0;
        // per unit, formed once per batch by the lane that owns it: which way the pass moves its lines (3: equal energies, no
        // pass) and MaxQuant of its wordlen
        const uint32_t my_want = (my_e2 < my_e1) ? 1u : (my_e2 > my_e1) ? 2u : 3u;
        const float my_mul = tab_f(tab.mq, bits);
        // BFUs 19..25 are 32 lines wide: two of them share a pass, one per half of the wavefront (same lists, same ranks,
        // half as many trips through the write / rendezvous / read-back sequence)
        for (uint32_t rem32 = (ea_need >> 19) & 0x7fu; rem32;) {
            const int ubA = __builtin_ctz(rem32);
            rem32 &= rem32 - 1u;
            int ubB = -1;
            if (rem32) {
                ubB = __builtin_ctz(rem32);
                rem32 &= rem32 - 1u;
            }
            const int half = lane >> 5, l = lane & 31;
            const bool has = half == 0 || ubB >= 0;
            const int bfu = 19 + ((half && ubB >= 0) ? ubB : ubA);
            const int start = kEaLine0 + 32 * (bfu - 19), ustart = start - kEaLine0, line = start + l;   // == bfu_start(bfu) for BFUs 19..25
            const uint32_t want = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * bfu, (int)my_want);
            const float mul = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(4 * bfu, (int)__float_as_uint(my_mul)));   // == max_quant(wordlen of the unit)
            // The pass is written WITHOUT lane conditions around its stores (a condition costs the wavefront a compare, two exec-mask
            // instructions and a branch - ~50 cycles, tools/ubench/valu_issue - and this pass had five): every lane of a half stores ONE
            // key and ONE record. Candidates store their |delta| at their slot and their record at their rank; the other lanes store +inf
            // behind the candidates (the key list comes out padded to 32: no second store) and a zero record behind the ranks. A half
            // without a unit (the odd unit out) walks unit A's lines again and stores into spare bytes behind the key lists.
            const float val_l = L.val[line];
            const int m0_l = (int)L.bm[line - kTermLine0];
            const uint32_t cbyte = L.code[(line - kEaLine0) >> 2];
            const bool flag = has & (((cbyte >> (2 * (line & 3))) & 3u) == want);
            const unsigned long long mask = __ballot(flag);
            const uint32_t hm = half ? (uint32_t)(mask >> 32) : (uint32_t)mask;
            const int cnt = __popc(hm), below = __popc(hm & ((1u << l) - 1u));
            const int pos = flag ? below : cnt + (l - below);   // a permutation of 0..31 per half
            float* uk = L.uk + 64 * half;
            const float t = val_l * mul;
            const float key = fabsf(t - (truncf(t) + 0.5f));
            uk[pos] = flag ? key : __builtin_huge_valf();
            const int m0 = m0_l;
            const bool neg = m0 < 0 || (m0 == 0 && !(t > 0));
            const uint32_t recv = flag ? ((uint32_t)l | ((uint32_t)(m0 < 0 ? -m0 : m0) << 7) | ((uint32_t)neg << 12)) : 0u;
            wave_sync();
            // rank = keys of the own list below the own key, eight per step (most units list eight candidates or fewer: 5.5 on average
            // on white noise); the trip count is the longer list's (uniform), the shorter one reads its padding
            const int cnt_lo = __popc((uint32_t)mask), cnt_hi = __popc((uint32_t)(mask >> 32));
            const int cmax = cnt_lo > cnt_hi ? cnt_lo : cnt_hi;
            int rr = 0;
            {
                const float4* t4 = reinterpret_cast<const float4*>(uk);
                for (int q = 0; q < cmax; q += 8) rr = count_below8(t4[(q >> 2)], t4[(q >> 2) + 1], key, rr);
            }
            uint16_t* recw = has ? L.rec + ustart : reinterpret_cast<uint16_t*>(L.uk + 128);
            const int wpos = flag ? rr : pos;
            recw[wpos] = (uint16_t)recv;
            wave_sync();
            // equal keys collide on a rank; the loser notices on read-back and the unit goes to the exact path (a lane that is no
            // candidate reads its own zero back)
            const unsigned long long lost = __ballot(recw[wpos] != (uint16_t)recv);
            if ((uint32_t)lost) tie_units |= 1u << ubA;
            if ((uint32_t)(lost >> 32)) tie_units |= 1u << (ubB >= 0 ? ubB : ubA);
            if (lane == 19 + ubA) my_nc = __popc((uint32_t)mask);
            if (ubB >= 0 && lane == 19 + ubB) my_nc = __popc((uint32_t)(mask >> 32));
            wave_sync();   // the key lists are reused by the next pair
        }
        for (uint32_t rem = (ea_need >> 19) & ~0x7fu; rem; rem &= rem - 1u) {
            const int ub = __builtin_ctz(rem), bfu = 19 + ub;
            const bool two = ub >= 11;   // (uniform) BFUs 30 and 31 are 128 lines long: a second round of 64 lines
            const int start = two ? 768 + 128 * (ub - 11) : 512 + 64 * (ub - 7), ustart = start - kEaLine0;   // == bfu_start(bfu) for BFUs 26..31
            const uint32_t want = (uint32_t)__builtin_amdgcn_readlane((int)my_want, bfu);
            const float mul = readlane_f(my_mul, bfu);
            // as above: every lane stores one key and one record per round, without lane conditions - candidates at their slots and
            // ranks, the others +inf / zero behind them (the list comes out padded to the unit's length)
            bool flag[2];
            float key[2];
            uint32_t recv[2];
            int below[2];
            unsigned long long mask[2] = {0ull, 0ull};
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                flag[r] = false;
                key[r] = 0.0f;
                recv[r] = 0u;
                below[r] = 0;
                if (r == 1 && !two) continue;
                const int j = 64 * r + lane, line = start + j;
                const float val_l = L.val[line];
                const int m0 = (int)L.bm[line - kTermLine0];
                flag[r] = ((L.code[(line - kEaLine0) >> 2] >> (2 * (line & 3))) & 3u) == want;   // (64 r + lane < n in the rounds that run)
                mask[r] = __ballot(flag[r]);
                below[r] = __popcll(mask[r] & ((1ull << lane) - 1ull));
                const float t = val_l * mul;
                key[r] = fabsf(t - (truncf(t) + 0.5f));   // sort key |delta|
                const bool neg = m0 < 0 || (m0 == 0 && !(t > 0));
                recv[r] = flag[r] ? ((uint32_t)j | ((uint32_t)(m0 < 0 ? -m0 : m0) << 7) | ((uint32_t)neg << 12)) : 0u;
            }
            const int c0 = __popcll(mask[0]), cnt_u = c0 + __popcll(mask[1]);
            // positions: round 0's candidates, round 1's, then the lanes that list nothing (round 0's, round 1's) - a permutation of 0..n-1
            int pos[2];
            pos[0] = flag[0] ? below[0] : cnt_u + (lane - below[0]);
            pos[1] = flag[1] ? c0 + below[1] : cnt_u + (64 - c0) + (lane - below[1]);
            L.uk[pos[0]] = flag[0] ? key[0] : __builtin_huge_valf();
            if (two) L.uk[pos[1]] = flag[1] ? key[1] : __builtin_huge_valf();
            wave_sync();
            int wpos[2] = {pos[0], pos[1]};
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (r == 1 && !two) continue;
                const float4* t4 = reinterpret_cast<const float4*>(L.uk);
                int rr = 0;
                for (int q = 0; q < cnt_u; q += 16) {
                    rr = count_below8(t4[(q >> 2)], t4[(q >> 2) + 1], key[r], rr);
                    rr = count_below8(t4[(q >> 2) + 2], t4[(q >> 2) + 3], key[r], rr);
                }
                wpos[r] = flag[r] ? rr : pos[r];
                L.rec[ustart + wpos[r]] = (uint16_t)recv[r];
            }
            wave_sync();
            // equal keys collide on a rank; the loser notices on read-back and the unit goes to the exact path
            bool lost = L.rec[ustart + wpos[0]] != (uint16_t)recv[0];
            if (two) lost = lost || L.rec[ustart + wpos[1]] != (uint16_t)recv[1];
            if (__ballot(lost) != 0ull) tie_units |= 1u << ub;
            if (lane == bfu) my_nc = cnt_u;
            wave_sync();   // the key list is reused by the next unit
        }
        AT3_UPH(pc, 7, -1);
        // equal keys among listed candidates: libstdc++'s std::sort order decides (rare). The order of equal elements
        // depends on the whole array the reference sorts, so the full |delta| < 0.25 list is rebuilt, sorted with the
        // restated algorithm and then filtered.
        for (uint32_t rem = tie_units; rem; rem &= rem - 1u) {
            const int ub = __builtin_ctz(rem), bfu = 19 + ub;
            const float e1 = readlane_f(my_e1, bfu), e2 = readlane_f(my_e2, bfu);
            const float mul = max_quant(__builtin_amdgcn_readlane(bits, bfu));
            if (lane == 0) {
                SortItem* s_items = reinterpret_cast<SortItem*>(L.uk);
                const int start = bfu_start(bfu), n = bfu_start(bfu + 1) - start;
                int nall = 0;
                for (int j = 0; j < n; ++j) {
                    const float t = L.val[start + j] * mul;
                    const float delta = t - (truncf(t) + 0.5f);
                    if (fabsf(delta) < 0.25f) {
                        s_items[nall].key = delta;
                        s_items[nall].idx = j;
                        ++nall;
                    }
                }
                std_sort_abs(s_items, nall, reinterpret_cast<int*>(reinterpret_cast<char*>(L.term) + sizeof(float) * 256 + sizeof(uint16_t) * kEaLines));   // the union's bytes behind the key list and the records
                const int dir = (e2 < e1) ? 1 : (e2 > e1) ? -1 : 0;
                uint16_t* sorted = L.rec + (start - kEaLine0);
                int nc = 0;
                for (int q = 0; q < nall; ++q) {
                    const int j = s_items[q].idx;
                    const float t = L.val[start + j] * mul;
                    const int m0 = __float2int_rn(t);
                    const int a0 = m0 < 0 ? -m0 : m0;
                    const bool side = (dir > 0) ? ((float)a0 < fabsf(t) && (float)a0 < (mul - 1)) : (dir < 0) ? ((float)a0 > fabsf(t)) : false;
                    const bool neg = m0 < 0 || (m0 == 0 && !(t > 0));
                    if (side) sorted[nc++] = (uint16_t)((uint32_t)j | ((uint32_t)a0 << 7) | ((uint32_t)neg << 12));
                }
                // (the same candidates the ballot pass listed, so the unit's count stands)
            }
            wave_sync();
        }
        // the sequential pass, one lane per unit: only |mantissa| enters the energy bookkeeping - the re-rounded code is
        // |m0| + 1 (e2 < e1; a zero becomes +-1) or |m0| - 1 (e2 > e1), atrac_scale.cpp:86-118; the ordered part per
        // candidate is ex = (e2 - d0) + d1 and the test, everything else is ready before the chain reaches it
#ifdef AT3HIP_DEBUG_KNOBS
        if (dbg == 6) my_nc = 0;
#endif
        if (mine && lane > 18 && my_nc > 0) {
            const float e1 = my_e1;
            float e2 = my_e2;
            const bool grow = e2 < e1;
            float dist = fabsf(e2 - e1);
            const uint16_t* rp = L.rec + (my_start - kEaLine0);
            int8_t* mant = L.bm + (my_start - kTermLine0);
            // what the emission needs to form this unit's mantissas again: plain rounding plus the lines moved by one below - a bit per
            // line in the row of this wordlen (a unit is quantised once per wordlen and the rows were cleared before the rate loop, so
            // OR-ing a bit in is storing it: untouched units stay zero)
            uint32_t* cw = L.chg + (bits - 1) * kChgWords + ((my_start - kEaLine0) >> 5);
            uint2 r4 = *reinterpret_cast<const uint2*>(rp);
            for (int c0 = 0; c0 < my_nc; c0 += 4) {
                const uint2 n4 = *reinterpret_cast<const uint2*>(rp + (c0 + 4 < my_nc ? c0 + 4 : c0));   // (the next four, requested now)
                float d0[4], d1[4];
                int idx[4], mnew[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t rc = ((k < 2 ? r4.x : r4.y) >> (16 * (k & 1))) & 0xffffu;
                    const int a0 = (int)((rc >> 7) & 31u);
                    const int a1 = grow ? a0 + 1 : (a0 > 0 ? a0 - 1 : 0);
                    idx[k] = (int)(rc & 127u);
                    mnew[k] = ((rc >> 12) & 1u) ? -a1 : a1;
                    const float fa0 = (float)a0, fa1 = (float)a1;   // (exact, as their squares are: values up to 32)
                    d0[k] = (fa0 * fa0) * my_inv2;
                    d1[k] = (fa1 * fa1) * my_inv2;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    // (one lane condition per candidate, around the two stores only: the chain itself runs on selects)
                    float ex = e2;
                    ex -= d0[k];
                    ex += d1[k];
                    const float nd = fabsf(ex - e1);
                    const bool take = (c0 + k < my_nc) & (nd < dist);
                    if (take) {
                        mant[idx[k]] = (int8_t)mnew[k];
                        atomicOr(cw + (idx[k] >> 5), 1u << (idx[k] & 31));
                    }
                    e2 = take ? ex : e2;
                    dist = take ? nd : dist;
                }
                r4 = n4;
            }
            my_e2 = e2;
        }
    }
    wave_sync();
    AT3_UPH(pc, 8, -1);
    // ---- (4) VLC cost of the final mantissas; (5) cache entries ----
    VlcRow row_h[4];   // (the four rounds' length rows requested together)
#pragma unroll
    for (int h = 0; h < 4; ++h) row_h[h] = tab_row(tab, wl_h[h]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int wl = wl_h[h];
        const VlcRow row = row_h[h];
        uint32_t vb = 0;
        if (wl) {
            const int line0 = 256 * h + 4 * lane;
            const uint32_t pk = *reinterpret_cast<const uint32_t*>(L.bm + (line0 - kTermLine0));
            int m[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) m[k] = (int)(int8_t)((pk >> (8 * k)) & 0xff);
            if (wl > 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k) vb += vlc_len(row, m[k]);
            } else {
                vb = vlc_pair_len(m[0], m[1]) + vlc_pair_len(m[2], m[3]);
            }
        }
        unit_cost_store(L, tab, h, wl, vb, lane);
    }
    if (mine && qerr) qerr[(bits - 1) * 32 + lane] = my_e1 / my_e2;   // BFUs >= 10: nothing but the QUANT tap looks at their energy error
    wave_sync();
    AT3_UPH(pc, 9, 10);
}

// The 70 units of BFUs 0..9 (8 or 16 lines each, no energy-adaptive pass below BFU 19), one lane per unit: ConsiderEnergyErr
// (atrac3_bitstream.cpp:241-257) looks at the first ten BFUs' energy errors at whatever wordlen the allocation gives them,
// and the whole set costs less than one large unit.
__device__ __forceinline__ void small_units(AllocLds& L, const LaneTab& tab, int lane, float my_e1)
{
    // Unit u = 0..13: the 16-line BFUs 8 and 9 at wordlen 1 + u / 2; u = 14..69: BFU (u - 14) % 8 at wordlen 1 + (u - 14) / 8.
    // Pass one: lane u takes unit u's first eight lines. Pass two: lanes 0..13 take their unit's second eight lines (the
    // energy sum goes on in line order) while lanes 14..19 take the six units 64..69 - two passes of eight lines, not three.
    const int uA = lane, uB = 50 + lane;   // (uB is a unit for lanes 14..19 only)
    const int bfuA = uA < 14 ? 8 + (uA & 1) : (uA - 14) & 7, wlA = uA < 14 ? 1 + (uA >> 1) : 1 + ((uA - 14) >> 3);
    const int bfuB = (uB - 14) & 7, wlB = 1 + ((uB - 14) >> 3);
    const bool second = lane < 14, extra = lane >= 14 && lane < 20;
    // e1 of the units' BFUs, from the lanes that own the BFUs (all lanes take part in the exchanges)
    const float e1A = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(4 * bfuA, (int)__float_as_uint(my_e1)));
    const float e1B = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(4 * bfuB, (int)__float_as_uint(my_e1)));
    float e2A = 0.0f, e2B = 0.0f;
    uint32_t vbA = 0, vbB = 0;
    // the units' constants (lanes without a second unit look up wordlen 0)
    const float mulA = tab_f(tab.mq, wlA), invA = tab_f(tab.inv, wlA);
    const VlcRow rowA = tab_row(tab, wlA);
    const int wlB7 = extra ? wlB : 0;
    const float mulB = tab_f(tab.mq, wlB7), invB = tab_f(tab.inv, wlB7);
    const VlcRow rowB = tab_row(tab, wlB7);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const bool own = pass == 0 || second;   // the lane works on unit A (else, in pass two, on unit B if it has one)
        if (pass == 0 || second || extra) {
            const int bfu = own ? bfuA : bfuB, wl = own ? wlA : wlB;
            const int line0 = bfu_start(bfu) + (pass == 1 && second ? 8 : 0);
            const float mul = own ? mulA : mulB;
            const float inv2 = own ? invA : invB;
            const VlcRow row = own ? rowA : rowB;
            const float4 va = *reinterpret_cast<const float4*>(L.val + line0), vb4 = *reinterpret_cast<const float4*>(L.val + line0 + 4);
            const float v[8] = {va.x, va.y, va.z, va.w, vb4.x, vb4.y, vb4.z, vb4.w};
            int m[8];
            float r[8];   // (the rounded products as floats: their squares are exact in fp32, see compute_units)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                r[k] = rintf(v[k] * mul);
                m[k] = (int)r[k];
            }
            float e2 = own ? e2A : e2B;
#pragma unroll
            for (int k = 0; k < 8; ++k) e2 += (r[k] * r[k]) * inv2;
            const uint32_t vb = vlc_bits8(wl, row, m);
            if (own) {
                e2A = e2;
                vbA += vb;
            } else {
                e2B = e2;
                vbB += vb;
            }
        }
    }
    L.err[(wlA - 1) * 10 + bfuA] = e1A / e2A;
    L.cost[(wlA - 1) * 32 + bfuA] = (uint16_t)vbA;
    if (extra) {
        L.err[(wlB - 1) * 10 + bfuB] = e1B / e2B;
        L.cost[(wlB - 1) * 32 + bfuB] = (uint16_t)vbB;
    }
    wave_sync();
}

// EncodeTonalComponents (atrac3_bitstream.cpp:382-524) for a frame whose (quantiser, length) groups are one sub-group each
// (the common case, proven per frame by the caller's `tonal_serial` check), written by one lane per tonal block instead of
// one lane for everything. Stream layout: 5 bits number of groups, 2 zero bits; per group in ascending (quantiser,
// length): 4 band flags, length - 1 (3), quantiser (3); then for each 64-line block of a flagged band its member count
// (3) followed by its members, each scale factor (6), position in the block (6) and the VLC codes of its values.
// Offsets come from one pass over the (at most 24) blocks with cross-lane reads. Returns the bits used (uniform).
__device__ __forceinline__ int tonal_emit_parallel(const PsyRec* rec, const uint8_t* s_tbits, const uint16_t* s_huff, uint32_t* words, int pos,
                                                   int lane, int n_tonal, int num_bfu, int bits, int tb_bfu, int tb_len, int tb_blk)
{
    const bool live = lane < n_tonal && tb_bfu < num_bfu;
    const int wl_t = __builtin_amdgcn_ds_bpermute(4 * (tb_bfu & 31), bits);
    int qn = wl_t + 4;
    qn = qn > 7 ? 7 : qn;
    const int g = live ? qn * 8 + tb_len : 0xff;               // group id, ascending = stream order
    const int sz = live ? 12 + (int)s_tbits[(lane < kMaxTonal ? lane : 0) * 6 + qn - 2] : 0;
    // my block's payload, requested now
    TonalBlock tb = {};
    if (live) tb = rec->tonal[lane];
    unsigned long long gm = 0ull, gb0 = 0ull, gb1 = 0ull, gb2 = 0ull, gb3 = 0ull;   // groups present; per QMF band
    int before_members = 0, in_before = 0, same_blk = 0, fb = 0, total_members = 0;
    bool first_in_group = true, first_in_blk = true;
    for (int t = 0; t < n_tonal; ++t) {   // uniform
        const int g2 = __builtin_amdgcn_readlane(g, t);
        if (g2 == 0xff) continue;
        const int blk2 = __builtin_amdgcn_readlane(tb_blk, t), sz2 = __builtin_amdgcn_readlane(sz, t);
        const unsigned long long bit = 1ull << g2;
        gm |= bit;
        const int band2 = blk2 >> 2;
        if (band2 == 0) gb0 |= bit;
        else if (band2 == 1) gb1 |= bit;
        else if (band2 == 2) gb2 |= bit;
        else gb3 |= bit;
        total_members += sz2;
        if (g2 < g) before_members += sz2;
        if (g2 == g) {
            fb |= 1 << band2;
            if (t < lane) {
                first_in_group = false;
                in_before += sz2;
                if (blk2 == tb_blk) first_in_blk = false;
            }
            if (blk2 == tb_blk) ++same_blk;
        }
    }
    const int groups = __popcll(gm);
    const int group_bands = __popcll(gb0) + __popcll(gb1) + __popcll(gb2) + __popcll(gb3);
    if (lane == 0) put_bits(words, pos, (uint32_t)groups, 5);
    if (groups == 0) return 5;
    if (live) {
        const unsigned long long below = (1ull << g) - 1ull;
        const int groups_before = __popcll(gm & below);
        const int pairs_before = __popcll(gb0 & below) + __popcll(gb1 & below) + __popcll(gb2 & below) + __popcll(gb3 & below);
        const int base = pos + 7 + 10 * groups_before + 12 * pairs_before + before_members;
        const int my_band = tb_blk >> 2;
        const int counts_upto = 4 * __popc((uint32_t)fb & ((1u << my_band) - 1u)) + (tb_blk & 3) + 1;   // count fields up to my block's
        const int at = base + 10 + 3 * counts_upto + in_before;
        if (first_in_group) {
            const uint32_t flags = ((fb & 1) << 3) | (((fb >> 1) & 1) << 2) | (((fb >> 2) & 1) << 1) | ((fb >> 3) & 1);
            put_bits(words, base, (flags << 6) | ((uint32_t)(tb_len - 1) << 3) | (uint32_t)qn, 10);
        }
        if (first_in_blk) put_bits(words, at - 3, (uint32_t)same_blk, 3);
        put_bits(words, at, tb.sfi, 6);
        put_bits(words, at + 6, (uint32_t)tb.pos - (uint32_t)tb_blk * 64u, 6);
        int bp = at + 12;
        const float mul = max_quant(qn);
        const float vals[7] = {tb.values[0], tb.values[1], tb.values[2], tb.values[3], tb.values[4], tb.values[5], tb.values[6]};
#pragma unroll
        for (int z = 0; z < 7; ++z) {
            if (z < tb_len) {
                const int m = __float2int_rn(vals[z] * mul);
                const uint32_t e = lds_huff(s_huff, qn, vlc_index(m));
                put_bits(words, bp, e & 0xffu, (int)(e >> 8));
                bp += (int)(e >> 8);
            }
        }
    }
    return 7 + 10 * groups + 12 * group_bands + total_members;
}

// CalcBitsAllocation for one BFU (atrac3_bitstream.cpp:272-336) followed by ConsiderEnergyErr's closure `gmap`
__device__ __forceinline__ int alloc_bits(float A, bool gate, int tcount, uint32_t gmap, float lam)
{
    // (selects, no lane conditions: the rate loop forms an allocation per trip, and a condition around five instructions costs
    // the wavefront more than the five)
    const int tmp = (int)(A - lam);
    int bits = tmp > 7 ? 7 : tmp < 1 ? 1 : tmp;   // 7 above seven, 1 for zero, tmp between
    bits = (gate || tmp < 0) ? 0 : bits;            // nothing below the threshold in quiet, nothing for a negative difference
    // one decrement per tonal block in this BFU while the wordlen is above 2 (:325-333)
    const int dec = bits - tcount > 2 ? bits - tcount : 2;
    bits = (bits > 2 && tcount) ? dec : bits;
    return (int)((gmap >> (3 * bits)) & 7u);
}

__global__ __launch_bounds__(64) AT3_WAVES_PER_EU(4) void k_alloc_pack(BackParams p, const Tables* T)
{
    __shared__ __attribute__((aligned(16))) AllocLds L;
    uint32_t* s_words = L.words;
    uint8_t* s_alloc = L.alloc;
    uint8_t* s_tbits = L.tbits;
    uint16_t* s_huff = L.huff;
    int* s_misc = L.misc;
    uint16_t* s_cost = L.cost;
    float* s_err = L.err;
    unsigned long long* s_tmask = L.tmask;
    // scratch of the serial tonal walk: the union's bytes behind the bit buffer and the code tables (free whenever it runs)
    uint8_t* tonal_scr = reinterpret_cast<uint8_t*>(L.term) + sizeof(uint32_t) * kBitWords + sizeof(uint16_t) * 132;

    const int lane = threadIdx.x;
    const int n_out = p.n_blocks - p.f0;
    const size_t cf = blockIdx.x;
    // AT3HIP_TAP_CLOCK: workgroup 0 reads the shader-cycle counter and the 100 MHz reference on entry and on exit
    const bool clk_probe = p.clk != nullptr && blockIdx.x == 0;
    unsigned long long clk_t0 = 0ull, clk_r0 = 0ull;
    if (clk_probe) {
        clk_t0 = __builtin_amdgcn_s_memtime();
        clk_r0 = __builtin_amdgcn_s_memrealtime();
    }
    PhaseClock pc;
#ifdef AT3HIP_DEBUG_KNOBS
    pc.last = __builtin_amdgcn_s_memtime();
    pc.slots = p.clk ? p.clk + 16 + (blockIdx.x & 255u) * 12u : nullptr;
    const unsigned long long item_r0 = __builtin_amdgcn_s_memrealtime();   // per-item life (100 MHz): tools/alloc_item_times.sh
    pc.item = (p.clk && blockIdx.x < 16384u) ? p.clk + 16 + 2 * 256 * 12 + 2 * 16384 + 12 * blockIdx.x : nullptr;
    if (pc.item && lane == 0)
        for (int k = 0; k < 12; ++k) pc.item[k] = 0ull;
#endif
    const int ch = (int)(cf & 1);
    const int fo = (int)((cf >> 1) % n_out);
    const int s = (int)((cf >> 1) / n_out);
    const int f = fo + p.f0;
    const PsyRec* recs = p.psy + (cf & ~(size_t)1);
    const PsyRec* rec = recs + ch;
    const Curve* curves = p.curves + ((size_t)s * p.n_blocks + f) * 8;
    const int half = p.frame_sz >> 1;
    // Everything the wavefront wants from global memory before the rate loop is requested HERE, in one batch, and nothing below
    // waits for more than its own value (loads return in issue order). Requested where they were used, the tonal count, the
    // curves' point counts, the channel loudnesses (joint stereo) and the spectrum were four round trips one after the other
    // at the head of every wavefront's life; a load inside an `if` with a default value costs a wait where the two paths join.
    const int n_tonal = rec->n_tonal;
    const uint32_t curve_w0 = *reinterpret_cast<const uint32_t*>(curves + (lane & 7));
    const float loud_m = recs[0].loud_ch, loud_s = recs[1].loud_ch;
    const float loud_raw = p.loud[(size_t)s * n_out + fo];
    const int my_sfi = rec->sfi[lane & 31];   // one load per lane (lanes 32..63 mirror 0..31)
    // wanted by CalcBitsAllocation's per-BFU constants after the small units
    const int pre_band = (lane & 31) >= 30 ? 3 : (lane & 31) >= 26 ? 2 : (lane & 31) >= 18 ? 1 : 0;
    const float* ges_src = p.ges ? p.ges + ((size_t)s * p.n_blocks + f) * 8 + ch * 4 + pre_band : &T->scale[63];   // (ScaleTable[63] = 2^0)
    const float pre_g = *ges_src;
    const float pre_energy = rec->energy[lane & 31];
    const float pre_ath = T->ath_bfu[lane & 31];
    const float my_scale = T->scale[lane];   // ScaleTable has 64 entries: looked up across lanes, not through memory
    const int pre_fixed = (int)c_fixed_alloc[lane & 31];
    float4 x4[4];
    {
        const float* specs = p.specs + cf * 1024;
#pragma unroll
        for (int k = 0; k < 4; ++k) x4[k] = *reinterpret_cast<const float4*>(specs + 4 * (lane + 64 * k));
    }
    __builtin_amdgcn_sched_barrier(0);

    for (int i = lane; i < 7 * 32; i += 64) s_cost[i] = 0u;
    float* qerr = p.quant ? &p.quant[cf].err[0][0] : nullptr;
    if (qerr)
        for (int i = lane; i < 7 * 32; i += 64) qerr[i] = 0.0f;   // (zero = never computed)

    // ---- header + gain info bits, joint-stereo byte shift, target bits (WriteSoundUnit :759-810) ----
    // lanes 0..7 hold the frame's eight gain curves (16 bytes each) from here to the emission
    // (only the point counts are needed before the emission: the curves themselves are fetched again there rather than
    // held in four registers across the whole rate loop)
    const int curve_n = lane < 8 ? (int)(curve_w0 & 0xffu) : 0;
    int hdr[2];
    for (int c2 = 0; c2 < 2; ++c2) {
        int bits = (p.js && c2 == 1) ? 14 : 6;
        bits += 2;
        for (int b = 0; b < 4; ++b) bits += 3 + 9 * __builtin_amdgcn_readlane(curve_n, c2 * 4 + b);
        // one input channel, joint stereo: the second element has ONE subband and no gain points (atrac3denc.cpp:843-849)
        if (p.mono_js && c2 == 1) bits = 14 + 2 + 3;
        hdr[c2] = bits;
    }
    int shift = 0;
    if (p.mono_js) {   // CalcMSBytesShift with an empty second element: the maximum (atrac3_bitstream.cpp:745-747)
        const int totalUsed = 12 + hdr[0] + hdr[1];
        shift = (int)((uint32_t)p.frame_sz / 2 - (1 + ((uint32_t)totalUsed - 1) / 8));
    } else if (p.js) {
        const int b0 = -6 - hdr[0], b1 = -6 - hdr[1];
        const int totalUsed = 0 - b0 - b1;
        const int maxShift = (int)((uint32_t)p.frame_sz / 2 - (1 + ((uint32_t)totalUsed - 1) / 8));
        const float m = loud_m, sd = loud_s;
        const float total = sd + m;
        float ratio = 0.0f;
        if (total > 0) ratio = (float)((double)(m / total) - 0.5);
        int v = __float2int_rn((float)p.frame_sz * ratio);
        if (v > maxShift) v = maxShift;
        if (v < -maxShift) v = -maxShift;
        shift = v;
    }
    const int nbytes = (ch == 0) ? half + shift : half - shift;
    int target = -6 - hdr[ch] + 8 * nbytes;
    if (target < 1) target = 1;
    target &= 0xffff;
    const float loudness = loud_raw / 0.006f;

    if (p.mono_js && ch == 1) {
        for (int i = lane; i < kBitWords; i += 64) s_words[i] = 0;
        // TConfigure / TAlloc with empty ScaledBlocks (atrac3_bitstream.cpp:590-597, 623-626): JS parameters, one subband
        // without gain points, no tonal components, one BFU of precision 0 in coding mode 1 - 33 bits, then zeros
        __syncthreads();
        if (lane == 0) {
            put_bits(s_words, 0, 0, 1);
            put_bits(s_words, 1, 7, 3);
            for (int k = 0; k < 4; ++k) put_bits(s_words, 4 + 2 * k, 3, 2);
            put_bits(s_words, 12, 3, 2);
            put_bits(s_words, 14, 0, 2);       // numQmfBand - 1
            put_bits(s_words, 16, 0, 3);       // gain points of band 0
            put_bits(s_words, 19, 0, 5);       // tonal sub-groups
            put_bits(s_words, 24, 0, 5);       // numBlocks - 1
            put_bits(s_words, 29, 1, 1);       // coding mode
            put_bits(s_words, 30, 0, 3);       // precision of the one block
        }
        __syncthreads();
        uint8_t* frame1 = p.out + ((size_t)s * n_out + fo) * p.frame_sz;
        for (int j = lane; j < nbytes; j += 64) {
            const int src = nbytes - 1 - j;
            frame1[half + shift + j] = (src < kBitWords * 4) ? (uint8_t)(s_words[src >> 2] >> (24 - 8 * (src & 3))) : 0;
        }
        return;
    }


    // ---- scaled values (TScaler::Scale, atrac_scale.cpp:141-172) and e1 = sum of value^2 per BFU, in line order ----
    {
        // (the four rounds' scale factors first: two cross-lane round trips in all, not two per round)
        float sf4[4];
        {
            int sfi4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) sfi4[k] = __builtin_amdgcn_ds_bpermute(4 * bfu_of_line(4 * (lane + 64 * k)), my_sfi);
#pragma unroll
            for (int k = 0; k < 4; ++k) sf4[k] = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(4 * sfi4[k], (int)__float_as_uint(my_scale)));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i0 = 4 * (lane + 64 * k);
            const float sf = sf4[k];
            const float4 x = x4[k];
            float v[4] = {x.x / sf, x.y / sf, x.z / sf, x.w / sf};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (fabsf(v[j]) >= 1.0f) v[j] = (v[j] > 0) ? 0.99999f : -0.99999f;
            *reinterpret_cast<float4*>(L.val + i0) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    __syncthreads();
    AT3_PH_END(pc, 0);
    // lane b < 19: e1 of BFU b. From BFU 19 on e1 is only read by the energy-adaptive pass of one of the BFU's units (its lane
    // forms it then, compute_units; -1 = not yet) - the pass runs for the final allocation's units and few others, the two 128-line
    // BFUs are mostly dropped before that, and their chain of additions is four times the longest one here
    float my_e1 = -1.0f;
    if (lane < 19) my_e1 = ordered_square_sum(L.val + bfu_start(lane), bfu_start(lane + 1) - bfu_start(lane));
    __syncthreads();
    AT3_PH_END(pc, 1);
    // units of the first ten BFUs at every wordlen: ConsiderEnergyErr (atrac3_bitstream.cpp:241-257) looks at their energy
    // errors whatever the allocation, and they are 96 lines in all
#ifdef AT3HIP_DEBUG_KNOBS
    if (p.debug_stop == 1) return;
#endif
    const LaneTab tab = lane_tab(lane);
    small_units(L, tab, lane, my_e1);
    AT3_PH_END(pc, 2);
#ifdef AT3HIP_DEBUG_KNOBS
    if (p.debug_stop == 2) return;
#endif
    // lane i: bit wl set = the cache holds the VLC bits of unit (BFU i, wl); bit 8 + wl set = it holds them or a lower bound (unit_bounds)
    uint32_t valid = lane < 10 ? 0xfefeu : 0u;
    // ---- TConfigure: spread (sequential float sums, every lane computes the same value) ----
    const int i = lane & 31;   // BFU owned by this lane (lanes 32..63 mirror 0..31 but never contribute)
    const int n_i = opaque_lane_value(bfu_start(i + 1) - bfu_start(i));   // (computed once: not re-derived inside the rate loop)
    float spread;
    {
        // The reference's two sequential float sums are sums of exactly representable terms: the 32 indices (integers
        // below 64) and, with m their total, the squares ((32 sfi - m) / 32)^2 = integer / 1024. While the integer total S2
        // of the second sum stays below 2^24 no partial sum rounds, so both come out of integer reductions across the
        // lanes; a frame with a wilder spread (sigma above 22, clamped to 14 anyway) takes the literal loops.
        const uint32_t r1 = row_allreduce_add(lane < 32 ? (uint32_t)my_sfi : 0u);
        const int m = (int)((uint32_t)__builtin_amdgcn_readlane((int)r1, 0) + (uint32_t)__builtin_amdgcn_readlane((int)r1, 16));
        const int d = 32 * my_sfi - m;
        const uint32_t r2 = row_allreduce_add(lane < 32 ? (uint32_t)(d * d) : 0u);
        const uint32_t S2 = (uint32_t)__builtin_amdgcn_readlane((int)r2, 0) + (uint32_t)__builtin_amdgcn_readlane((int)r2, 16);
        float sigma;
        if (S2 < (1u << 24)) {
            sigma = (float)S2 / 1024.0f;
        } else {
            float sum = 0.0f;
            for (int k = 0; k < 32; ++k) sum += (float)__builtin_amdgcn_readlane(my_sfi, k);
            sum /= 32;
            sigma = 0.0f;
            for (int k = 0; k < 32; ++k) {
                float t = ((float)__builtin_amdgcn_readlane(my_sfi, k) - sum);
                t *= t;
                sigma += t;
            }
        }
        sigma /= 32;
        sigma = sqrtf(sigma);
        if (sigma > 14.0f) sigma = 14.0f;
        spread = sigma / 14.0f;
    }
    // tonal blocks: VLC bit cost for every quantiser 2..7
    for (int idx = lane; idx < n_tonal * 6; idx += 64) {
        const int t = idx / 6, qq = 2 + idx % 6;
        const TonalBlock& tb = rec->tonal[t];
        const float mul = max_quant(qq);
        int bits = 0;
        for (int z = 0; z < tb.len; ++z) bits += (int)(huff_entry(qq, vlc_index(__float2int_rn(tb.values[z] * mul))) >> 8);
        s_tbits[t * 6 + qq - 2] = (uint8_t)bits;
    }
    // Lane t < n_tonal also owns tonal block t (its BFU, length and 64-line block): the cost of the tonal side
    // information is evaluated by these lanes in parallel inside the rate loop (inside the bisection below).
    int tb_bfu = 255, tb_len = 0, tb_blk = 0;
    if (lane < n_tonal) {
        const TonalBlock& tb = rec->tonal[lane];
        tb_bfu = tb.bfu;
        tb_len = tb.len;
        tb_blk = tb.pos >> 6;
    }
    // GroupTonalComponents (atrac3_bitstream.cpp:338-380) closes a sub-group only after EIGHT members of one group inside one
    // 64-line block. A 64-line block holds at most four tonal BFUs (BFUs 8..28 are 16 lines or wider, one run each), so
    // that never happens and every (quantiser, length) group is exactly one sub-group; the check below proves it for this
    // frame (blocks are ordered by position) and sends anything else down the literal, serial path.
    // (the cross-lane read stands in front of the condition: inside a short-circuit it would run with some lanes switched off)
    const int blk_7_on = __builtin_amdgcn_ds_bpermute(4 * ((lane + 7) & 63), tb_blk);
    const bool tonal_serial = __ballot(lane + 7 < n_tonal && blk_7_on == tb_blk) != 0ull;
    // per-BFU constants of CalcBitsAllocation (atrac3_bitstream.cpp:272-336)
    float A;
    bool gate;
    int tcount = 0;
    for (int t = 0; t < n_tonal; ++t) tcount += (__builtin_amdgcn_readlane(tb_bfu, t) == (lane & 31));
    {
        int band = 0;
        if (i >= 18) band = 1;
        if (i >= 26) band = 2;
        if (i >= 30) band = 3;
        float g = pre_g;   // (band == pre_band: the same BFU ranges)
        if (!(isfinite(g) && g > 0.0f)) g = 1.0f;
        const float corrected = pre_energy * g;
        const float ath = pre_ath * loudness;
        gate = corrected < ath;
        const float csfi = fmaxf(0.0f, fminf(63.0f, (float)my_sfi + 1.5f * at3_log2f(T, g)));
        float x = 6.0f;
        if (i < 3) x = 2.8f;
        else if (i < 10) x = 2.6f;
        else if (i < 15) x = 3.3f;
        else if (i <= 20) x = 3.6f;
        else if (i <= 28) x = 4.2f;
        A = spread * (csfi / x) + (1.0f - spread) * (float)pre_fixed;
        // (tonal blocks per BFU: counted below from the per-lane copies of the blocks' BFU indices)
    }
    // ConsiderEnergyErr as a per-BFU map wl -> wl' (first 10 BFUs, atrac3_bitstream.cpp:241-257, :638-641): BFUs are
    // independent, so iterating the reference's do/while to its fixed point is a closure per BFU. A wordlen keeps
    // climbing while its energy error is out of range, so the map sends wl to the first k >= wl that is acceptable
    // (k = 0 and k = 7 always are): a backward scan over the eight entries.
    uint32_t gmap = 0;
    {
        int g = 7;
        gmap = 7u << 21;
#pragma unroll
        for (int k = 6; k >= 0; --k) {
            const float e = (k > 0 && i < 10) ? s_err[(k - 1) * 10 + i] : 0.0f;
            const bool climbs = i < 10 && k > 0 && ((e > 0 && e < 0.7f) || e > 1.2f);
            g = climbs ? g : k;
            gmap |= (uint32_t)g << (3 * k);
        }
    }
    if (p.quant) {   // the QUANT tap's energy errors (BFUs 0..9, every wordlen): their storage is about to be reused
        QuantRec* qr = p.quant + cf;
        for (int k = lane; k < 70; k += 64) qr->err[k / 10][k % 10] = s_err[k];
    }
    wave_sync();
    for (int k = lane; k < 7 * kChgWords; k += 64) L.chg[k] = 0u;   // (same bytes as the energy errors, dead from here on)
    wave_sync();
    AT3_PH_END(pc, 3);
    // ---- rate loop: TConfigure / TAlloc under the bisection driver (uniform control flow) ----
    int num_bfu = p.bfu_idx_const ? p.bfu_idx_const : 32;
    if (target < 101) {
        int lim = 1;
        if (target > 5) lim = (target - 5) / 3;
        if (lim < 1) lim = 1;
        if (num_bfu > lim) num_bfu = lim;
    }
    if (num_bfu < 1) num_bfu = 1;
    if (!p.bfu_idx_const) {
        // The reference drops the last BFU and repeats the whole bisection whenever that BFU ends with no bits
        // (atrac3_bitstream.cpp:646-655). A BFU below the threshold in quiet gets none at any lambda, so every bisection
        // with such a BFU on top ends that way whatever it converges to: those passes are skipped, not run (a 3 kHz burst
        // or a handful of sines leaves twenty of them on top, each worth a dozen evaluations).
        const uint32_t quiet = (uint32_t)__ballot(lane < 32 && gate);
        while (num_bfu > 1 && ((quiet >> (num_bfu - 1)) & 1u)) --num_bfu;
    }
    int mode = 1;
    int bits = 0;
    AT3_STAT(10, 1);
    // Evaluations already made, one per lane: lambda -> the CLC | VLC << 13 sums, the count of coded BFUs and the tonal
    // bits for the CURRENT number of BFUs. The reference repeats the whole bisection with one BFU less whenever the last
    // BFU ended without bits (three times per channel-frame on white noise, eleven on the burst input), and each repeat
    // walks the same lambdas until the three header bits it saved tip a comparison: a repeat looks its totals up here -
    // when a BFU is dropped, its own contribution is taken out of every recorded sum - and only a lambda never seen before
    // costs a reduction over the BFUs (and possibly new units).
    // A repeat does not even walk the recorded lambdas one by one: each record also keeps the bisection's state before its
    // evaluation and the way the comparison went, `path` is the set of records the latest pass went through (in lane
    // order: a pass follows the one before it up to the first comparison that goes another way and appends new records
    // from there on), so after a drop all records redo their comparison at once and the repeat resumes at the first one
    // that changed - or at the last one when none did. Anything else (a hit outside that order, a full memo) turns the
    // shortcut off for the frame and the repeat is walked as before.
    float m_lam = 0.0f, m_min = 0.0f, m_max = 0.0f, m_last = 0.0f;
    // m_nz: the count of coded BFUs | kind of the record << 8 (0: the bits, 1 / 2: a bound that decided the comparison, set where the
    // record is made) | the way the comparison went << 10 (0: below the target, 1: above, 2: on it - the pass ended there);
    uint32_t m_acc = 0u, m_nz = 0u, m_ton = 0u;
    int memo_n = 0;
    unsigned long long path = 0ull;
    bool skip_ok = true;
    int resume = -1;
    bool bits_current = false;
    float final_lam = 0.0f;
    for (;;) {
        float minL = -8.0f, maxL = 20.0f, curL = 0.0f, lastL = 20.0f;
        if (resume >= 0) {
            minL = readlane_f(m_min, resume);
            maxL = readlane_f(m_max, resume);
            lastL = readlane_f(m_last, resume);
        }
        bool restart = false;
        for (;;) {
            const bool exhausted = (maxL <= minL);
            const float pre_min = minL, pre_max = maxL, pre_last = lastL;
            float lam;
            if (exhausted) {
                lam = lastL;
            } else {
                curL = (maxL + minL) * 0.5f;
                lam = curL;
            }
            uint32_t acc, nz, tonal_bits = 5;
            int rec_lane = -1;   // the record of this evaluation
            const unsigned long long hit = __ballot(lane < memo_n && m_lam == lam);
            const int hit_k = hit ? __builtin_ctzll(hit) : 0;
            // (a record whose comparison was decided by a bound of its total holds that bound: good for the comparison it
            // decided - and for the same one after BFUs were dropped, as long as it still decides it - but not for an
            // evaluation that ends the bisection, which needs the bits themselves: those are evaluated again, in place)
            const uint32_t hit_nz = hit ? (uint32_t)__builtin_amdgcn_readlane((int)m_nz, hit_k) : 0u;
            AT3_LPH(pc, 5);
            AT3_STAT(0, 1);
            if (hit && (hit_nz >> 8) == 0u) {
                acc = (uint32_t)__builtin_amdgcn_readlane((int)m_acc, hit_k);
                nz = hit_nz;
                tonal_bits = (uint32_t)__builtin_amdgcn_readlane((int)m_ton, hit_k);
                rec_lane = hit_k;
                bits_current = false;
                AT3_STAT(1, 1);
            } else {
            AT3_STAT(3, 1); if (hit) AT3_STAT(2, 1);
            bits = (lane < num_bfu) ? alloc_bits(A, gate, tcount, gmap, lam) : 0;
            const uint32_t clc_i = __umul24(tab_u(tab.misc, bits) & 7u, (uint32_t)n_i);   // == clc_bits(bits, n_i)
            // the count of non-zero BFUs comes from a ballot: summed as a third field it would need six bits at
            // 32 of 32 (and silently wrapped to zero when every BFU of the frame was coded)
            nz = (uint32_t)__popcll(__ballot(bits != 0));
            if (n_tonal > 0 && tonal_serial) {
                if (lane < 32) s_alloc[lane] = (uint8_t)bits;
                __syncthreads();
                if (lane == 0) s_misc[0] = tonal_encode<false>(rec, s_tbits, s_alloc, num_bfu, nullptr, 0, tonal_scr);
                __syncthreads();
                tonal_bits = (uint32_t)(s_misc[0] & 0xffff);
            } else if (n_tonal > 0) {
                // EncodeTonalComponents with a null stream (atrac3_bitstream.cpp:382-524), one lane per tonal block:
                //   5 (+2 when anything is coded) + per group 4 + 3 + 3 + 12 per QMF band the group touches
                //   + per coded block 6 + 6 + VLC bits of its values at the group's quantiser.
                const bool live = lane < n_tonal && tb_bfu < num_bfu;
                const int wl_t = __builtin_amdgcn_ds_bpermute(4 * (tb_bfu & 31), bits);   // this evaluation's wordlen of the block's BFU
                int qn = wl_t + 4;
                qn = qn > 7 ? 7 : qn;   // >= 4 always, so the lower clamp at 2 is never active here
                if (lane < 4) s_tmask[lane] = 0ull;
                wave_sync();
                uint32_t member = 0;
                if (live) {
                    atomicOr(&s_tmask[tb_blk >> 2], 1ull << ((qn - 2) * 7 + (tb_len - 1)));
                    member = 12u + s_tbits[lane * 6 + qn - 2];
                }
                const uint32_t msum = row_allreduce_add(member);
                const uint32_t members = (uint32_t)__builtin_amdgcn_readlane((int)msum, 0) + (uint32_t)__builtin_amdgcn_readlane((int)msum, 16);
                wave_sync();
                const unsigned long long m0 = s_tmask[0], m1 = s_tmask[1], m2 = s_tmask[2], m3 = s_tmask[3];
                const uint32_t groups = (uint32_t)__popcll(m0 | m1 | m2 | m3);
                const uint32_t group_bands = (uint32_t)(__popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3));
                if (groups) tonal_bits = 5u + 2u + 10u * groups + 12u * group_bands + members;
            }
            AT3_LPH(pc, 6);
            // The units this allocation asks for (TEncCache, atrac_enc_cache.cpp) are brought in only as far as the comparison with
            // the target needs them: spec bits = 3 per BFU + 6 per coded BFU + min(CLC, VLC) are at most the CLC bits (known from the
            // wordlens alone) and at least min(CLC, a lower bound of the VLC bits). unit_bounds gives a close lower bound of a new
            // unit's VLC bits - below BFU 19 the bits themselves -, compute_units, with the energy-adaptive pass, the bits. An
            // evaluation that ends the bisection (`exhausted`, or a total on the target - which only bits can show) always gets the
            // bits: the frame is coded from it.
            {
                const uint32_t need = (uint32_t)__ballot(lane < 32 && bits != 0 && !((valid >> (8 + bits)) & 1u));
#ifdef AT3HIP_DEBUG_KNOBS
                if (need && p.debug_stop == 4) {   // the loop without the units (their cache entries stay zero)
                    if (lane < 32 && ((need >> lane) & 1u)) valid |= 0x101u << bits;
                } else
#endif
                if (need) {
                    unit_bounds(L, tab, need, bits, lane, pc);
                    AT3_STAT(4, 1); AT3_STAT(5, __popc(need));
                    if (lane < 32 && ((need >> lane) & 1u)) valid |= (lane < 19 ? 0x101u : 0x100u) << bits;
                }
            }
            // (bits is zero from num_bfu on; the cost is read at a clamped index and dropped rather than read under a lane condition)
            const int cost_at = (bits ? bits - 1 : 0) * 32 + i;
            uint32_t rsum = row_allreduce_add(clc_i | ((bits ? (uint32_t)s_cost[cost_at] : 0u) << 13));
            acc = (uint32_t)__builtin_amdgcn_readlane((int)rsum, 0) + (uint32_t)__builtin_amdgcn_readlane((int)rsum, 16);
            uint32_t kind = 0u;   // 0: acc holds the bits, 1: the total is below the target whatever the units cost, 2: above it
            const uint32_t inexact = (uint32_t)__ballot(lane < 32 && bits != 0 && !((valid >> bits) & 1u));
            if (inexact) {
                if (!exhausted) {
                    const uint32_t base = (uint32_t)num_bfu * 3 + 6 * nz + tonal_bits;
                    const uint32_t clc_b = acc & 0x1fffu, vlb = (acc >> 13) & 0x3fffu;
                    if (base + clc_b < (uint32_t)target) {
                        kind = 1u; AT3_STAT(8, 1);
                        acc = clc_b | (0x3fffu << 13);   // (read as min(CLC, VLC) = CLC below)
                    } else if (base + (clc_b <= vlb ? clc_b : vlb) > (uint32_t)target) {
                        kind = 2u; AT3_STAT(9, 1);
                    }
                }
                if (kind == 0u) {
#if defined(AT3_EMU_HOST) || defined(AT3HIP_DEBUG_KNOBS)
                    const uint32_t bound_i = s_cost[cost_at];
#endif
                    compute_units(L, tab, inexact, bits, my_e1, lane, qerr, pc, p.debug_stop);
                    AT3_STAT(6, 1); AT3_STAT(7, __popc(inexact));
#if defined(AT3_EMU_HOST) || defined(AT3HIP_DEBUG_KNOBS)
                    // every lower bound the loop decided with against the bits that now replace it: none may exceed them (the SIMT harness
                    // asserts `bad 0`; profiling builds count the same on the hardware - AT3HIP_TAP_CLOCK words 13 / 14, tools/fuzz_gpu.py --bounds)
                    const unsigned long long above = __ballot(lane < 32 && ((inexact >> lane) & 1u) && bound_i > (uint32_t)s_cost[cost_at]);
                    AT3_STAT(11, __popc(inexact));
                    AT3_STAT(12, __popcll(above));
#ifndef AT3_EMU_HOST
                    if (p.clk && lane == 0) {
                        atomicAdd(p.clk + 13, (unsigned long long)__popc(inexact));
                        if (above) atomicAdd(p.clk + 14, (unsigned long long)__popcll(above));
                    }
#endif
#endif
                    if (lane < 32 && ((inexact >> lane) & 1u)) valid |= 1u << bits;
                    rsum = row_allreduce_add(clc_i | ((bits ? (uint32_t)s_cost[cost_at] : 0u) << 13));
                    acc = (uint32_t)__builtin_amdgcn_readlane((int)rsum, 0) + (uint32_t)__builtin_amdgcn_readlane((int)rsum, 16);
                }
            }
            AT3_LPH(pc, 7);
            if (hit || memo_n < 64) {
                rec_lane = hit ? hit_k : memo_n;
                if (lane == rec_lane) {
                    m_lam = lam;
                    m_acc = acc;
                    m_nz = (m_nz & 0xc00u) | nz | (kind << 8);
                    m_ton = tonal_bits;
                }
                if (!hit) ++memo_n;
            }
            bits_current = true;
            }
            const uint32_t clc = acc & 0x1fffu, vlc = (acc >> 13) & 0x3fffu;
            mode = clc <= vlc ? 1 : 0;
            const uint32_t spec_bits = (uint32_t)num_bfu * 3 + 6 * nz + (mode ? clc : vlc);
            const uint32_t total = spec_bits + tonal_bits;
            bool done;
            int dir = 2;
            if (exhausted) {
                done = true;
            } else if (total < (uint32_t)target) {
                lastL = curL;
                maxL = curL - 0.01f;
                done = false;
                dir = 0;
            } else if (total > (uint32_t)target) {
                minL = curL + 0.01f;
                done = false;
                dir = 1;
            } else {
                done = true;
            }
            if (skip_ok && !exhausted) {
                if (rec_lane < 0 || (path >> rec_lane) > 1ull) {   // not recorded, or a record the pass already left behind
                    skip_ok = false;
                    path = 0ull;
                } else {
                    path |= 1ull << rec_lane;
                    if (lane == rec_lane) {
                        m_min = pre_min;
                        m_max = pre_max;
                        m_last = pre_last;
                        m_nz = (m_nz & ~0xc00u) | ((uint32_t)dir << 10);
                    }
                }
            }
            AT3_LPH(pc, 8);
            if (!done) continue;
            final_lam = lam;
            const int last_alloc = (p.bfu_idx_const || num_bfu <= 1) ? 1
                                   : __builtin_amdgcn_readlane(bits_current ? bits : alloc_bits(A, gate, tcount, gmap, lam), (num_bfu - 1) & 31);
            if (!p.bfu_idx_const && num_bfu > 1 && last_alloc == 0) {
                for (;;) {
                    // the dropped BFU leaves every recorded evaluation: its bits at that lambda, their cost, its place in the count
                    const int t = num_bfu - 1;
                    const int tc_t = __builtin_amdgcn_readlane(tcount, t);
                    if (tc_t > 0) {
                        memo_n = 0;   // its tonal blocks leave the tonal side information too: nothing recorded stays valid
                    } else {
                        const int b = alloc_bits(readlane_f(A, t), __builtin_amdgcn_readlane((int)gate, t) != 0, 0,
                                                 (uint32_t)__builtin_amdgcn_readlane((int)gmap, t), m_lam);
                        // (the BFU's line count from the lane that owns it: bfu_start() of a wavefront-uniform index is a chain of scalar
                        // compares and branches, some forty instructions per dropped BFU)
                        const int n_t = __builtin_amdgcn_readlane(n_i, t);
                        if (lane < memo_n && b) {
                            // (a bound record was summed with the unit's bound where the cache may hold its bits by now: taking out no less
                            // than went in, never below zero, leaves a bound; exact records take out what went in)
                            const uint32_t went = (uint32_t)s_cost[(b - 1) * 32 + t], have = (m_acc >> 13) & 0x3fffu;
                            m_acc -= clc_bits(b, n_t) | ((went < have ? went : have) << 13);
                            m_nz -= 1u;
                        }
                    }
                    num_bfu--;
                    restart = true;
                    resume = -1;
                    bool unchanged = false;
                    if (skip_ok && memo_n > 0 && path) {
                        const uint32_t c1 = m_acc & 0x1fffu, v1 = (m_acc >> 13) & 0x3fffu;
                        // (for a bound record `tot` is that bound of its total: unchanged while it still decides the comparison the way it did)
                        const uint32_t tot = (uint32_t)num_bfu * 3 + 6 * (m_nz & 0xffu) + (c1 <= v1 ? c1 : v1) + m_ton;
                        const int nd = tot < (uint32_t)target ? 0 : tot > (uint32_t)target ? 1 : 2;
                        const unsigned long long changed = __ballot((uint32_t)nd != ((m_nz >> 10) & 3u)) & path;
                        unchanged = changed == 0ull;
                        resume = changed ? __builtin_ctzll(changed) : 63 - __builtin_clzll(path);
                        path &= (2ull << resume) - 1ull;
                    } else {
                        path = 0ull;
                    }
                    // No recorded comparison goes another way with one BFU less: the repeat would walk the same records, leave the
                    // last of them the same way, run out of interval at the same lambda and stop there (that evaluation's total is not
                    // compared) - with this allocation. So the only thing it would find out is whether the NEW last BFU has bits at
                    // that lambda: looked at here, without the two trips through the loop (on LP4 a channel-frame of white noise
                    // sheds eleven BFUs one after the other this way).
                    if (!unchanged || num_bfu <= 1) break;
                    const int la = __builtin_amdgcn_readlane(alloc_bits(A, gate, tcount, gmap, lam), (num_bfu - 1) & 31);
                    if (la != 0) {
                        restart = false;        // the bisection is over: lam, mode and the records stand
                        bits_current = false;   // (the allocation is formed again for num_bfu BFUs below)
                        break;
                    }
                }
            }
            break;
        }
        AT3_LPH(pc, 9);
        if (!restart) break;
    }
    if (!bits_current) bits = (lane < num_bfu) ? alloc_bits(A, gate, tcount, gmap, final_lam) : 0;   // (mode is the last evaluation's)
    AT3_PH_END(pc, 4);
#ifdef AT3HIP_DEBUG_KNOBS
    if (p.debug_stop == 3) return;
#endif
    if (p.quant) {   // the QUANT tap: the units whose bits the cache holds at the end (err e1 / e2, cost CLC | VLC << 13; zero = not computed, or a bound only)
        QuantRec* qr = p.quant + cf;
        if (lane < 32) s_alloc[lane] = (uint8_t)(valid & 0xfeu);
        wave_sync();
        for (int k = lane; k < 7 * 32; k += 64) {
            const uint32_t vb = ((s_alloc[k & 31] >> (1 + (k >> 5))) & 1u) ? s_cost[k] : 0u;
            qr->cost[k >> 5][k & 31] = vb ? (clc_bits(1 + (k >> 5), bfu_start((k & 31) + 1) - bfu_start(k & 31)) | (vb << 13)) : 0u;
        }
    }
    // (the emission's lane constants are formed here, behind the rate loop, not held across it)
    const int elane = opaque_lane_value(lane);
    // requested now, wanted after the mantissas below have been formed: the frame's curves (lanes 0..7, for the header) and
    // the code tables (their LDS storage was the key lists' until here)
    const uint4 cw = *reinterpret_cast<const uint4*>(curves + (elane & 7));   // (the address is formed here: as the head's, it would be held across the rate loop)   // (every elane asks: a load inside `if (elane < 8)` is waited for where the paths join)
    const uint32_t huff_a = c_huff[elane], huff_b = c_huff[64 + elane], huff_c = elane < 2 ? c_huff[128 + elane] : 0u;
    if (elane < 32) s_alloc[elane] = (uint8_t)bits;
    for (int k = elane; k < kBitWords; k += 64) s_words[k] = 0;   // the key lists are dead: their storage becomes the bit buffer

    // ---- emission (WriteSoundUnit header, EncodeSpecs) ----
    // mantissas: 16 spectral lines per elane (BFU sizes are multiples of 8, so at most two BFUs per elane), formed AGAIN from
    // the scaled values instead of being kept for every unit the rate loop ever asked for: plain rounding, and from BFU 19
    // on the lines the energy-adaptive pass moved by one (its record, one bit per line and wordlen). Which way a recorded
    // line went can be read off the line itself: the pass that adds lists only lines rounded towards zero, the pass that
    // takes away only lines rounded away from it (atrac_scale.cpp:86-118; the side tests of compute_units' codes).
    int wl_h[2];
    int m_h[2][8];
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
        const int i0 = elane * 16 + 8 * hlf;
        const int b = bfu_of_line(i0);
        const int wl = __builtin_amdgcn_ds_bpermute(4 * b, bits);   // (zero from num_bfu on)
        wl_h[hlf] = wl;
        const float mul = tab_f(tab.mq, wl);
        const float4 va = *reinterpret_cast<const float4*>(L.val + i0), vb = *reinterpret_cast<const float4*>(L.val + i0 + 4);
        const float v[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
        uint32_t moved = 0u;
        if (i0 >= kEaLine0 && wl) moved = (L.chg[(wl - 1) * kChgWords + ((i0 - kEaLine0) >> 5)] >> ((i0 - kEaLine0) & 31)) & 0xffu;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float t = v[k] * mul;
            int m = __float2int_rn(t);
            if ((moved >> k) & 1u) {
                const int a0 = m < 0 ? -m : m;
                const int a1 = ((float)a0 < fabsf(t)) ? a0 + 1 : a0 - 1;
                const bool neg = m < 0 || (m == 0 && !(t > 0));
                m = neg ? -a1 : a1;
            }
            m_h[hlf][k] = wl ? m : 0;
        }
    }
    s_huff[elane] = (uint16_t)huff_a;
    s_huff[64 + elane] = (uint16_t)huff_b;
    if (elane < 2) s_huff[128 + elane] = (uint16_t)huff_c;
    __syncthreads();
    int pos = (p.js && ch == 1) ? 14 : 6;
    if (elane == 0) {
        if (p.js && ch == 1) {
            put_bits(s_words, 0, 0, 1);
            put_bits(s_words, 1, 7, 3);
            for (int k = 0; k < 4; ++k) put_bits(s_words, 4 + 2 * k, 3, 2);
            put_bits(s_words, 12, 3, 2);
        } else {
            put_bits(s_words, 0, 0x28, 6);
        }
        put_bits(s_words, pos, 3, 2);
    }
    pos += 2;
    {   // gain points: elane ch * 4 + b writes band b's count and (level, location) pairs out of its registers
        const int n0 = __builtin_amdgcn_readlane(curve_n, ch * 4), n1 = __builtin_amdgcn_readlane(curve_n, ch * 4 + 1);
        const int n2 = __builtin_amdgcn_readlane(curve_n, ch * 4 + 2), n3 = __builtin_amdgcn_readlane(curve_n, ch * 4 + 3);
        const int b = elane - ch * 4;
        if (b >= 0 && b < 4) {
            int at = pos + 3 * b + 9 * ((b > 0 ? n0 : 0) + (b > 1 ? n1 : 0) + (b > 2 ? n2 : 0));
            put_bits(s_words, at, (uint32_t)curve_n, 3);
            at += 3;
            const uint64_t lo = (uint64_t)cw.x | ((uint64_t)cw.y << 32), hi = (uint64_t)cw.z | ((uint64_t)cw.w << 32);
            for (int k = 0; k < curve_n; ++k) {   // Curve: n, level[7] | loc[7], pad
                const uint32_t level = (uint32_t)(lo >> (8 * (k + 1))) & 0xffu, loc = (uint32_t)(hi >> (8 * k)) & 0xffu;
                put_bits(s_words, at, level, 4);
                put_bits(s_words, at + 4, loc, 5);
                at += 9;
            }
        }
        pos += 12 + 9 * (n0 + n1 + n2 + n3);
    }
    if (n_tonal == 0) {   // EncodeTonalComponents without components: five zero bits (atrac3_bitstream.cpp:382-400)
        pos += 5;
    } else if (!tonal_serial) {
        pos += tonal_emit_parallel(rec, s_tbits, s_huff, s_words, pos, elane, n_tonal, num_bfu, bits, tb_bfu, tb_len, tb_blk);
    } else {
        if (elane == 0) s_misc[1] = tonal_encode<true>(rec, s_tbits, s_alloc, num_bfu, s_words, pos, tonal_scr);
        __syncthreads();
        pos += s_misc[1];
    }
    if (elane == 0) {
        put_bits(s_words, pos, (uint32_t)num_bfu - 1, 5);
        put_bits(s_words, pos + 5, (uint32_t)mode, 1);
    }
    pos += 6;
#ifdef AT3HIP_DEBUG_KNOBS
    if (p.debug_stop == 7) return;
#endif
    const unsigned long long nzmask = __ballot(elane < num_bfu && bits != 0);
    if (elane < num_bfu) put_bits(s_words, pos + 3 * elane, (uint32_t)bits, 3);
    pos += 3 * num_bfu;
    if (elane < num_bfu && bits)
        put_bits(s_words, pos + 6 * __popcll(nzmask & ((1ull << elane) - 1ull)), (uint32_t)my_sfi, 6);
    pos += 6 * __popcll(nzmask);
    {
        // The elane's codes are strung together in registers, most significant bit first: first into groups of at most 30
        // bits (four fixed-length codes, or three / three / two Huffman codes per eight lines), then group by group into
        // a 64-bit window that reaches the shared bit buffer one 32-bit word at a time.
        uint32_t grp[6], glen[6];
        int n_grp;
        if (mode == 1) {
            // fixed-length codes: clc_len(wl) bits of the mantissa; at wordlen 1 the pair code (m0 & 3) << 2 | (m1 & 3)
            // (MantissaToCLcIdx {2, 3, 0, 1}[m + 2] == m & 3, atrac3_bitstream.cpp:77-90) is two such codes of two bits
            n_grp = 4;
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                const int wl = wl_h[hlf];
                const int nb = (int)(tab_u(tab.misc, wl) & 7u);
                const uint32_t mask = (1u << nb) - 1u;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    uint32_t g = (uint32_t)m_h[hlf][4 * q] & mask;
                    g = (g << nb) | ((uint32_t)m_h[hlf][4 * q + 1] & mask);
                    g = (g << nb) | ((uint32_t)m_h[hlf][4 * q + 2] & mask);
                    g = (g << nb) | ((uint32_t)m_h[hlf][4 * q + 3] & mask);
                    grp[2 * hlf + q] = g;
                    glen[2 * hlf + q] = 4u * (uint32_t)nb;
                }
            }
            grp[4] = grp[5] = glen[4] = glen[5] = 0u;
        } else {
            n_grp = 6;
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                const int wl = wl_h[hlf];
                const int base = (int)(tab_u(tab.misc, wl) >> 8);
                const int (&m8)[8] = m_h[hlf];
                uint32_t g = 0u, gl = 0u;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    uint32_t e = 0u;   // code | length << 8
                    if (wl > 1) e = s_huff[base + vlc_index(m8[k])];
                    else if (wl == 1 && (k & 1) == 0) e = s_huff[vlc_pair_index(m8[k], m8[k + 1])];
                    const uint32_t n = e >> 8;   // <= 10
                    g = (g << n) | (e & 0xffu);
                    gl += n;
                    if (k == 2 || k == 5 || k == 7) {
                        const int gi = 3 * hlf + (k == 2 ? 0 : k == 5 ? 1 : 2);
                        grp[gi] = g;
                        glen[gi] = gl;
                        g = 0u;
                        gl = 0u;
                    }
                }
            }
        }
        int sum = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) sum += (int)glen[k];
        const int off = pos + wave_inclusive_scan(sum, elane) - sum;
        int cur = off >> 5, fill = off & 31;
        uint64_t acc = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (k < n_grp) {
                const int n = (int)glen[k];   // <= 30
                acc |= (uint64_t)grp[k] << ((64 - fill - n) & 63);
                fill += n;
                if (fill >= 32) {
                    if (cur < kBitWords) atomicOr(&s_words[cur], (uint32_t)(acc >> 32));
                    acc <<= 32;
                    fill -= 32;
                    ++cur;
                }
            }
        }
        if (fill > 0 && cur < kBitWords) atomicOr(&s_words[cur], (uint32_t)(acc >> 32));
    }
    __syncthreads();
#ifdef AT3HIP_DEBUG_KNOBS
    if (p.debug_stop == 8) return;
#endif

    // ---- frame assembly (atrac3_bitstream.cpp:826-834): ch0 bytes, then ch1 (byte-reversed when JS) ----
    uint8_t* frame = p.out + ((size_t)s * n_out + fo) * p.frame_sz;
    const int dst0 = (ch == 0) ? 0 : half + shift;
    if (!p.js && ((p.frame_sz | half) & 3) == 0) {   // whole big-endian words
        for (int j = elane; j < (nbytes >> 2); j += 64)
            *reinterpret_cast<uint32_t*>(frame + dst0 + 4 * j) = (j < kBitWords) ? __builtin_bswap32(s_words[j]) : 0u;
    } else {
        for (int j = elane; j < nbytes; j += 64) {
            const int src = (p.js && ch == 1) ? (nbytes - 1 - j) : j;
            const uint8_t byte = (src < kBitWords * 4) ? (uint8_t)(s_words[src >> 2] >> (24 - 8 * (src & 3))) : 0;
            frame[dst0 + j] = byte;
        }
    }
    if (clk_probe && elane == 0) {
        p.clk[0] = __builtin_amdgcn_s_memtime() - clk_t0;
        p.clk[1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
    }
#ifdef AT3HIP_DEBUG_KNOBS
    AT3_PH_END(pc, 10);
    if (pc.slots && lane == 0) atomicAdd(pc.slots + 11, 1ull);   // wavefronts counted
    if (p.clk && lane == 0 && blockIdx.x < 16384u) {
        unsigned long long* it = p.clk + 16 + 2 * 256 * 12 + 2 * blockIdx.x;
        it[0] = item_r0;
        it[1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

}  // namespace at3
