#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (oracle/_ref/libat3ref.so, built from the
unmodified sources under /root/reference by `make -C oracle ref`).

Run in the build container only:  python tools/gen_golden.py
The fixtures are data (inputs + the reference's outputs); no reference source is stored.
Recorded alongside: glibc version and CPU flags relevant to libm's ifunc selection.
"""
import os
import platform
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from at3_testlib import LP2, LP4, SIGNALS, ROOT, have_ref, ref  # noqa: E402

NBLOCKS = 16
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    if not have_ref():
        raise SystemExit("oracle/_ref/libat3ref.so missing: run `make -C oracle ref` where /root/reference exists")
    r = ref()
    os.makedirs(OUT, exist_ok=True)
    meta = dict(glibc=platform.libc_ver()[1], machine=platform.machine(),
                fma="fma" in open("/proc/cpuinfo").read())

    # 1. end-to-end frames: PCM (s16) -> frame bytes for LP2/LP4 x {full, no gain, no tonal}
    enc = {}
    for name, gen in SIGNALS.items():
        pcm = gen(NBLOCKS)
        s16 = np.round(pcm * 32768.0).astype(np.int16)
        assert np.array_equal((s16.astype(np.float32) / np.float32(32768.0)), pcm)
        enc[f"{name}_pcm_s16"] = s16
        for br, brn in ((LP2, "lp2"), (LP4, "lp4")):
            for ng, nt, tag in ((0, 0, "full"), (1, 0, "nogain"), (0, 1, "notonal")):
                frames, taps = r.encode(pcm, br, ng, nt, taps=True)
                enc[f"{name}_{brn}_{tag}_frames"] = frames
                if tag == "full":
                    enc[f"{name}_{brn}_npoints"] = taps["n_points"].astype(np.int8)
                    enc[f"{name}_{brn}_level"] = taps["level"].astype(np.int8)
                    enc[f"{name}_{brn}_loc"] = taps["loc"].astype(np.int8)
                    enc[f"{name}_{brn}_sfi"] = taps["sfi"].astype(np.int8)
                    enc[f"{name}_{brn}_loudness"] = taps["loudness_track"]
    np.savez_compressed(os.path.join(OUT, "encode.npz"), meta=np.array(str(meta)), **enc)

    # 2. stage vectors
    rng = np.random.RandomState(1234)
    st = {}
    for k, v in r.tables().items():
        st[f"table_{k}"] = v
    pcm1 = (rng.randint(-8192, 8192, size=4 * 1024).astype(np.float32) / np.float32(32768.0 * 4.0))
    st["qmf_in"] = pcm1
    st["qmf_out"] = r.qmf(pcm1)
    x = rng.uniform(-1, 1, size=512).astype(np.float32)
    st["mdct512_in"] = x
    st["mdct512_out"] = r.mdct512(x)
    bands = rng.uniform(-0.2, 0.2, size=(4, 512)).astype(np.float32)
    npts = np.array([2, 1, 0, 3], dtype=np.int32)
    level = np.zeros((4, 8), dtype=np.int32)
    loc = np.zeros((4, 8), dtype=np.int32)
    level[0, :2] = (6, 3); loc[0, :2] = (4, 20)
    level[1, :1] = (2,); loc[1, :1] = (0,)
    level[3, :3] = (5, 7, 1); loc[3, :3] = (1, 2, 31)
    specs, mutated = r.mdct(bands, npts, level, loc)
    st.update(mdct_bands_in=bands, mdct_npoints=npts, mdct_level=level, mdct_loc=loc, mdct_specs=specs,
              mdct_bands_out=mutated)
    st["ges_out"] = r.gain_energy_scale(bands[0, :256], bands[0, 256:], level[0, :2], loc[0, :2], 1.25)
    up_in = (rng.randn(512) * np.linspace(0.01, 1.0, 512)).astype(np.float32)
    sig, hfr = r.upsample(up_in)
    g, lo, hi = r.analyze_gain(sig[1024:3072])
    st.update(up_in=up_in, up_out=sig, up_hfr=np.float32(hfr), ag_gain=g, ag_lo=lo, ag_hi=hi)
    # CalcCurve on a step envelope with valid context
    env = np.concatenate([np.full(20, 0.01), np.full(12, 0.5)]).astype(np.float32)
    lv, lc, ctx = r.calc_curve(env, np.array([0.01, 0.01, 0.01], np.float32), 1.9, env * 0.9, env * 1.1)
    st.update(cc_env=env, cc_level=lv, cc_loc=lc, cc_ctx=ctx)
    # RelationToIdx known answers (the reference's unit test pins 27 of these: atrac3denc_ut.cpp:1109-1139)
    xs = np.array([0.0001, 0.0004, 0.001, 0.03, 0.06, 0.12, 0.24, 0.26, 0.49, 0.5, 0.51, 0.99, 1.0, 1.9, 2.0, 3.9,
                   4.0, 7.9, 8.0, 15.9, 16.0, 100.0], dtype=np.float32)
    st["rti_x"] = xs
    st["rti_y"] = np.array([r.relation_to_idx_hdr(v) for v in xs], dtype=np.int32)
    # QuantMantisas incl. energy-adaptive rounding with ties
    q_in = (np.round(rng.uniform(-0.99, 0.99, size=(6, 128)) * 64) / 64).astype(np.float32)
    st["quant_in"] = q_in
    qm, qe = [], []
    for i, mul in enumerate((1.5, 2.5, 4.5, 7.5, 15.5, 31.5)):
        m, e = r.quant_mantisas(q_in[i], mul, 1)
        qm.append(m); qe.append(e)
    st["quant_mant"] = np.array(qm, dtype=np.int32)
    st["quant_err"] = np.array(qe, dtype=np.float32)
    sp = (rng.randn(1024) * 0.05).astype(np.float32)
    sfi, en, vals = r.scale_frame(sp)
    st.update(scale_in=sp, scale_sfi=sfi, scale_energy=en, scale_values=vals)
    st["flat_out"] = r.flatness(sp * sp)
    lx = np.exp(rng.uniform(-30, 30, size=4096)).astype(np.float32)
    st["log2f_x"] = lx
    st["log2f_y"] = np.array([r.log2f(v) for v in lx], dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, "stages.npz"), meta=np.array(str(meta)), **st)
    for f in os.listdir(OUT):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
