#!/usr/bin/env python3
"""EXPERIMENT AID (CPU only, test infrastructure): how many quantised units does the bisection need as BITS, how many only as a BOUND?

Patches a temporary copy of oracle/at3_oracle.c (nothing under oracle/ is changed): next to the real evaluation of every trip of
TAlloc::Encode it replays k_alloc_pack's scheme - a lower bound of a new unit's VLC bits (plain rounding; a positive line rounded
up with t - m < -0.25 may end one code lower; lb(x) = min(len(x), len(x + 1))), CLC as the upper bound, the energy-adaptive pass only
when neither decides the comparison or the evaluation ends the bisection - checks every decision taken from a bound against the
real one (BAD must be 0) and counts units and lines per channel-frame. Usage: tools/oracle_bounds_count.py [blocks]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "oracle", "at3_oracle.c")).read()

helper = r'''
unsigned long long G_chan, G_trips, G_real_units, G_real_lines, G_real_ea_lines, G_l1_units, G_l1_lines, G_l2_units, G_l2_ea_lines, G_dec, G_undec, G_bad;
static uint8_t L1v[32][8], L2v[32][8];
static uint32_t L1lb[32][8];
static int vlen1(const huff_t* tab, int a) { return a == 0 ? tab[0].bits : tab[2 * a - 1].bits; }
static uint32_t l1_bound(const float* in, int n, uint32_t wl, int ea)
{
    const float mul = kMaxQuant[wl < 7 ? wl : 7];
    const huff_t* tab = kHuffTab[wl - 1];
    const int top = (int)(mul - 0.5f);
    uint32_t lb = 0;
    int ap[128];
    for (int j = 0; j < n; ++j) {
        const float t = in[j] * mul, r = rintf(t);
        const int a = (int)fabsf(r);
        ap[j] = a - (ea && t > 0.0f && t - fabsf(r) < -0.25f);
    }
    if (wl > 1) {
        for (int j = 0; j < n; ++j) { const int x = ap[j]; int l = vlen1(tab, x); if (x + 1 <= top) { const int l2 = vlen1(tab, x + 1); if (l2 < l) l = l2; } lb += l; }
    } else {
        static const uint32_t rtab[9] = {8, 4, 7, 2, 0, 1, 6, 3, 5};
        for (int j = 0; j < n; j += 2) lb += tab[rtab[3 * (ap[j] + 1) + (ap[j + 1] + 1)]].bits;
    }
    return lb;
}
'''
old = "static void encode_channel(enc_ctx* c, bitw* w)\n{\n"
assert old in src
src = src.replace(old, helper + old + "    memset(L1v, 0, sizeof(L1v)); memset(L2v, 0, sizeof(L2v)); G_chan++;\n")
old = """            uint32_t bits;
            do {
                bits = specs_bits_consumption(c, alloc, n, &mode);
            } while (consider_energy_err(c->energy_err, alloc, n));
            const uint32_t total = bits + encode_tonal(c->sce, alloc, n, NULL);
"""
new = """            uint32_t bits;
            uint8_t was_valid[32][8];
            for (int i_ = 0; i_ < 32; ++i_) for (int w_ = 0; w_ < 8; ++w_) was_valid[i_][w_] = c->cache[i_][w_].valid;
            do {
                bits = specs_bits_consumption(c, alloc, n, &mode);
            } while (consider_energy_err(c->energy_err, alloc, n));
            const uint32_t tonal_b = encode_tonal(c->sce, alloc, n, NULL);
            const uint32_t total = bits + tonal_b;
            G_trips++;
            {
                for (int i_ = 10; i_ < n; ++i_) if (alloc[i_] && !was_valid[i_][alloc[i_]]) { G_real_units++; G_real_lines += kBfuStart[i_+1]-kBfuStart[i_]; if (i_ > 18) G_real_ea_lines += kBfuStart[i_+1]-kBfuStart[i_]; }
                for (int i_ = 10; i_ < n; ++i_) { const uint32_t w_ = alloc[i_]; if (w_ && !L1v[i_][w_]) { const int nl = kBfuStart[i_+1]-kBfuStart[i_]; L1lb[i_][w_] = l1_bound(c->sce->values + kBfuStart[i_], nl, w_, i_ > 18); L1v[i_][w_] = 1; G_l1_units++; G_l1_lines += nl; if (i_ <= 18) L2v[i_][w_] = 1; } }
                uint32_t clc = 0, vlb = 0, nz = 0; int missing = 0;
                for (int i_ = 0; i_ < n; ++i_) {
                    const uint32_t w_ = alloc[i_]; if (!w_) continue; nz++;
                    clc += c->cache[i_][w_].clc_bits;
                    if (i_ < 10 || L2v[i_][w_]) vlb += c->cache[i_][w_].vlc_bits; else { vlb += L1lb[i_][w_]; missing++; if (L1lb[i_][w_] > c->cache[i_][w_].vlc_bits) G_bad++; }
                }
                const uint32_t base = (uint32_t)n * 3 + 6 * nz + tonal_b;
                int decided = 0;
                if (missing && !exhausted) {
                    if (base + clc < c->target_bits) { decided = 1; if (!(total < c->target_bits)) G_bad++; }
                    else if (base + (clc <= vlb ? clc : vlb) > c->target_bits) { decided = 1; if (!(total > c->target_bits)) G_bad++; }
                }
                if (decided) G_dec++;
                else if (missing) { G_undec++; for (int i_ = 19; i_ < n; ++i_) { const uint32_t w_ = alloc[i_]; if (w_ && !L2v[i_][w_]) { L2v[i_][w_] = 1; G_l2_units++; G_l2_ea_lines += kBfuStart[i_+1]-kBfuStart[i_]; } } }
            }
"""
assert old in src
src = src.replace(old, new)
drv = r'''
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "at3_oracle.h"
extern unsigned long long G_chan, G_trips, G_real_units, G_real_lines, G_real_ea_lines, G_l1_units, G_l1_lines, G_l2_units, G_l2_ea_lines, G_dec, G_undec, G_bad;
static unsigned long long s = 88172645463325252ull;
static unsigned rnd(void){ s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (unsigned)(s >> 11); }
int main(int argc, char** argv)
{
    const char* kind = argv[1]; const int bitrate = atoi(argv[2]), nb = atoi(argv[3]);
    float* pcm = malloc(sizeof(float) * nb * 2048);
    for (int i = 0; i < nb * 1024; ++i) {
        if (!strcmp(kind, "noise")) { pcm[2*i] = (float)((int)(rnd() % 16384) - 8192) / 32768.0f; pcm[2*i+1] = (float)((int)(rnd() % 16384) - 8192) / 32768.0f; }
        else if (!strcmp(kind, "burst")) { double amp = ((i / 3000) % 2 == 0) ? 0.02 : 0.6; double l = amp * sin(2 * M_PI * 3000.0 * i / 44100.0); pcm[2*i] = roundf(l * 32768.0) / 32768.0f; pcm[2*i+1] = roundf(0.5 * l * 32768.0) / 32768.0f; }
        else { double l = 0, r = 0; const double f[5] = {440, 1000, 3000, 7000, 11000}; for (int k = 0; k < 5; ++k) { l += 0.1 * sin(2 * M_PI * f[k] * i / 44100.0); r += 0.1 * sin(2 * M_PI * f[k] * i / 44100.0 + 0.3 * k); } pcm[2*i] = roundf(l * 32768.0) / 32768.0f; pcm[2*i+1] = roundf(r * 32768.0) / 32768.0f; }
    }
    unsigned char* out = malloc(1024 * nb);
    int fs = 0;
    at3o_encode(bitrate, 2, 0, 0, 0, pcm, nb, out, &fs, NULL);
    const double c = (double)G_chan;
    printf("%-5s %6d: %llu channel-frames, %.1f evaluations each (%.1f decided by a bound, %.1f not), BAD %llu | reference: %.1f units, %.0f lines, %.0f through the pass | "
           "with bounds: %.1f units bounded (%.0f lines), %.1f quantised (%.0f lines through the pass)\n", kind, bitrate, G_chan, G_trips / c, G_dec / c, G_undec / c, G_bad,
           G_real_units / c, G_real_lines / c, G_real_ea_lines / c, G_l1_units / c, G_l1_lines / c, G_l2_units / c, G_l2_ea_lines / c);
    return G_bad != 0;
}
'''
nb = sys.argv[1] if len(sys.argv) > 1 else "65"
with tempfile.TemporaryDirectory() as d:
    open(os.path.join(d, "o.c"), "w").write(src)
    open(os.path.join(d, "d.c"), "w").write(drv)
    subprocess.check_call(["gcc", "-std=gnu11", "-O2", "-ffp-contract=off", "-w", "-I", os.path.join(ROOT, "oracle"), "-o", os.path.join(d, "run"), os.path.join(d, "d.c"), os.path.join(d, "o.c"), "-lm"])
    rc = 0
    for kind, br in (("noise", 132300), ("burst", 132300), ("tones", 132300), ("noise", 66150)):
        rc |= subprocess.call([os.path.join(d, "run"), kind, str(br), nb])
sys.exit(rc)
