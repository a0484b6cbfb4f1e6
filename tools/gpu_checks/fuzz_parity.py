"""Randomised parity sweep: the whole chain on the GPU against the numpy oracle, bit for bit, over random frame shapes, MA
tables, Legendre orders, dtypes, jump parameters and cosmic-ray / bad-pixel densities.
    python tools/gpu_checks/fuzz_parity.py [ncases] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import oracle
from oracle import saturation
from romanimpreprocess_amd import _native, pipeline, synth

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = _native.default_context(0)
cb = pipeline.Calibrator(ctx=ctx)


def random_pattern():
    kind = rng.integers(0, 5)
    if kind == 0:
        return synth.READ_PATTERN_8
    if kind == 1:
        return synth.READ_PATTERN_6
    if kind == 2:
        return synth.READ_PATTERN_16
    g = int(rng.integers(3, 13))
    rp, t = [], 0
    for _ in range(g):
        n = int(rng.integers(1, 5))
        rp.append(list(range(t, t + n)))
        t += n
    return rp


def same(a, b, zero_sign_ok=False):
    if a.dtype.kind != "f":
        return np.array_equal(a, b)
    bad = (a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))
    if zero_sign_ok:
        bad &= ~((a == 0) & (b == 0))
    return not bad.any()


fails, forms = 0, {0: 0, 1: 0, 2: 0, 3: 0}
t0 = time.time()
for case in range(ncases):
    rp = random_pattern()
    ny, nx = int(rng.integers(4, 13)) * 8, int(rng.choice([128, 256, 384]))
    if rng.random() < 0.1:
        # a tenth of the cases: frames of several strips and row ranges (up to 9 strips of the narrow forms, ranges that do
        # not divide the rows), with a read pattern the fused kernel is instantiated for
        rp = [synth.READ_PATTERN_8, synth.READ_PATTERN_6, synth.READ_PATTERN_16][int(rng.integers(0, 3))]
        ny, nx = int(rng.integers(13, 76)) * 4, int(rng.choice([512, 640, 1024]))
    p = int(rng.choice([3, 8, 10]))
    gdt = np.float64 if rng.random() < 0.2 else np.float32
    kdt = np.float64 if rng.random() < 0.4 else np.float32
    excl = bool(rng.random() < 0.7) or len(rp) < 4
    if len(rp) == 3:
        excl = False
    jump = None
    if rng.random() < 0.5:
        jump = {"SthreshA": float(rng.uniform(3, 8)), "SthreshB": float(rng.uniform(3, 6)), "IthreshA": float(rng.uniform(0.1, 2)),
                "IthreshB": float(rng.uniform(200, 2000))}
    seed = int(rng.integers(1, 10**6))
    cal = synth.make_caldir(ny, nx, read_pattern=rp, p_order=p, seed=seed, gain_dtype=gdt, ipc_dtype=kdt,
                            bias_amplitude=float(rng.uniform(0, 3)), bad_lin_frac=float(rng.uniform(0, 0.02)))
    ramp = synth.make_ramp(cal, read_pattern=rp, seed=seed + 1, cr_frac=float(rng.uniform(0, 0.08)))
    # one case in three: dq-init + saturation flagging on the device (with or without the read-pattern rule) against the numpy
    # restatement, thresholds lowered so that a few per cent of the pixels saturate at various groups
    sat = rng.random() < 0.33
    sat_kw = {}
    if sat:
        thr = np.quantile(ramp["data"].astype(np.float32), float(rng.uniform(0.6, 0.95)), axis=0).astype(np.float32)
        thr[rng.random((ny, nx)) < 0.95] = 65535.0
        sdq = np.where(rng.random((ny, nx)) < 0.01, np.uint32(1 << 21), np.uint32(0)).astype(np.uint32)   # NO_SAT_CHECK
        cal = dict(cal)
        cal["saturation"] = {"data": thr, "dq": sdq}
        backup, use_rp = int(rng.integers(0, 3)), bool(rng.random() < 0.5)
        h = {"data": ramp["data"], "groupdq": np.zeros(ramp["data"].shape, np.uint8), "pixeldq": ramp["pixeldq"].copy()}
        saturation.flag_saturation(h, thr, backup=backup, skip_firstn=1, n_pix_grow_sat=1, sat_dq=sdq, read_pattern=rp if use_rp else None)
        if excl:
            h["groupdq"][0] |= np.uint8(1)
        mask_dq = ramp["pixeldq"].copy()
        ramp = dict(ramp, groupdq=h["groupdq"], pixeldq=h["pixeldq"])
        sat_kw = dict(flag_saturation=True, saturation_backup=backup, saturation_read_pattern=use_rp)
    with np.errstate(all="ignore"):
        ref = oracle.calibrate_arrays(ramp, cal, exclude_first=excl, jump_pars=jump)
    lines = np.zeros((len(rp), nx // 128, 2))
    for g in range(len(rp)):
        lines[g] = ref["refpix_diag"][g]["channels"][:nx // 128, 2:4]
    ctx.set_option("fused", int(rng.random() < 0.8))
    ctx.set_option("chain2", int(rng.random() < 0.8))
    cb.load_caldir(1, cal)
    dev_ramp = dict(ramp, groupdq=None, pixeldq=mask_dq) if sat else ramp
    got = cb.calibrate(1, dev_ramp, exclude_first=excl, jump_pars=jump, want_cube=True, channel_lines=lines, **sat_kw)
    forms[ctx.last_chain_form()] += 1
    ok = (same(got["cube"], ref["data"], True) and same(got["groupdq"], ref["groupdq"]) and same(got["pixeldq"], ref["pixeldq"])
          and all(same(got[k], ref[k], True) for k in ("slope", "err_read", "err_poisson")))
    if not ok:
        fails += 1
        print(f"MISMATCH case {case}: G={len(rp)} rp={rp} shape=({ny},{nx}) p={p} gain={gdt.__name__} ipc={kdt.__name__} excl={excl} "
              f"jump={jump} seed={seed} form={ctx.last_chain_form()} sat={sat_kw}", flush=True)
    if (case + 1) % 25 == 0:
        print(f"{case + 1} cases, {fails} mismatches, forms {forms}, {time.time() - t0:.0f} s", flush=True)
ctx.set_option("fused", 1)
ctx.set_option("chain2", 1)
print(f"done: {ncases} cases, {fails} mismatches; kernel forms used (0 stage kernels, 2 fused kernel): {forms}")
sys.exit(1 if fails else 0)
