#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own numerics.

Runs only in the build container, where the reference is mounted read-only at /root/reference
(it never travels to the GPU machine; the fixtures do).  The reference's L3 modules
(``utils/fitting.py``, ``utils/ipc_linearity.py``, ``utils/flatutils.py``,
``utils/reference_subtraction.py``) are imported UNMODIFIED; only their two I/O-only imports are
replaced by in-memory stand-ins before import:
  * ``asdf``                       -> ``asdf.open(path)`` is a context manager returning a nested
                                       dict of numpy arrays registered under that path;
  * ``roman_datamodels.dqflags``   -> ``pixel`` / ``group`` Enum classes with np.uint32 members
                                       (bit values: SURVEY.md Appendix C).
Inputs come from ``tests/golden_cases.py``; fixtures store inputs AND outputs.

Usage:  python tools/make_goldens.py [--check] [--out DIR] [case ...]
  no case            every case, in one process (each case leaves sys.modules as it found it)
  --out DIR          write the fixtures there instead of tests/golden/
  --check            regenerate into a scratch directory and compare with tests/golden/ array by array (names, dtypes,
                     shapes, bytes); exit status 1 on any difference.  tests/test_oracle_golden.py runs it for the fast cases.
"""

import enum
import hashlib
import json
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/src"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.dont_write_bytecode = True

# ---------------------------------------------------------------- stand-ins for the I/O imports
_STORE = {}


class _Tree:
    def __init__(self, tree):
        self.tree = tree

    def __enter__(self):
        return self.tree

    def __exit__(self, *exc):
        return False


def _install_stubs():
    asdf = types.ModuleType("asdf")
    asdf.open = lambda path, *a, **k: _Tree(_STORE[path])
    sys.modules["asdf"] = asdf

    class pixel(np.uint32, enum.Enum):  # bit values: SURVEY.md Appendix C (the full table; maskhandling.py names most of it)
        DO_NOT_USE = 2**0
        SATURATED = 2**1
        JUMP_DET = 2**2
        DROPOUT = 2**3
        GW_AFFECTED_DATA = 2**4
        PERSISTENCE = 2**5
        AD_FLOOR = 2**6
        OUTLIER = 2**7
        UNRELIABLE_ERROR = 2**8
        NON_SCIENCE = 2**9
        DEAD = 2**10
        HOT = 2**11
        WARM = 2**12
        LOW_QE = 2**13
        TELEGRAPH = 2**15
        NONLINEAR = 2**16
        BAD_REF_PIXEL = 2**17
        NO_FLAT_FIELD = 2**18
        NO_GAIN_VALUE = 2**19
        NO_LIN_CORR = 2**20
        NO_SAT_CHECK = 2**21
        UNRELIABLE_BIAS = 2**22
        UNRELIABLE_DARK = 2**23
        UNRELIABLE_SLOPE = 2**24
        UNRELIABLE_FLAT = 2**25
        UNRELIABLE_RESET = 2**28
        OTHER_BAD_PIXEL = 2**30
        REFERENCE_PIXEL = 2**31

    class group(np.uint32, enum.Enum):
        DO_NOT_USE = 2**0
        SATURATED = 2**1
        JUMP_DET = 2**2

    # maskhandling.py imports astropy.io.fits for its file writer only (convert_file); build() never touches it
    astropy = types.ModuleType("astropy")
    astropy_io = types.ModuleType("astropy.io")
    astropy_fits = types.ModuleType("astropy.io.fits")
    astropy.io, astropy_io.fits = astropy_io, astropy_fits
    sys.modules.setdefault("astropy", astropy)
    sys.modules.setdefault("astropy.io", astropy_io)
    sys.modules.setdefault("astropy.io.fits", astropy_fits)
    rdm = types.ModuleType("roman_datamodels")
    dq = types.ModuleType("roman_datamodels.dqflags")
    dq.pixel, dq.group = pixel, group
    rdm.dqflags = dq
    sys.modules["roman_datamodels"] = rdm
    sys.modules["roman_datamodels.dqflags"] = dq


def register(path, roman):
    _STORE[path] = {"roman": roman}
    return path


class _Log:
    def __init__(self):
        self.output = ""

    def append(self, s):
        self.output += s


_install_stubs()
sys.path.insert(0, REF_SRC)
from romanimpreprocess.utils import fitting as ref_fit  # noqa: E402
from romanimpreprocess.utils import flatutils as ref_flat  # noqa: E402
from romanimpreprocess.utils import ipc_linearity as ref_il  # noqa: E402
from romanimpreprocess.utils import reference_subtraction as ref_rs  # noqa: E402
from romanimpreprocess.utils import maskhandling as ref_mh  # noqa: E402
from romanimpreprocess.utils import sky as ref_sky  # noqa: E402

import golden_cases as gc  # noqa: E402
from romanimpreprocess_amd import synth  # noqa: E402


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}.npz  {os.path.getsize(path)/1024:.0f} KiB")


def ref_meta(read_pattern, frame_time):
    """meta dict exactly as gen_cal_image.py:123-140 builds it (restated; that file cannot be imported)."""
    meta = {"frame_time": frame_time, "read_pattern": read_pattern}
    meta["ngrp"] = len(read_pattern)
    meta["tbar"] = np.zeros(meta["ngrp"], dtype=np.float32)
    meta["tau"] = np.zeros(meta["ngrp"], dtype=np.float32)
    meta["N"] = np.zeros(meta["ngrp"], dtype=np.int16)
    for i in range(meta["ngrp"]):
        meta["N"][i] = len(read_pattern[i])
        t0 = read_pattern[i][0]
        meta["tbar"][i] = (t0 + (meta["N"][i] - 1) / 2.0) * frame_time
        meta["tau"][i] = (t0 + (meta["N"][i] - 1) * (2 * meta["N"][i] - 1) / (6.0 * meta["N"][i])) * frame_time
    return meta


# ---------------------------------------------------------------- cases
def case_lin_known_answer():
    # the reference's own literal vector: tests/romanimpreprocess/test_linutils.py:9-48
    z = np.linspace(-1.5, 1.5, 31).reshape((1, 31))
    coefs = np.zeros((4, 1, 31))
    coefs[3] = 1.0
    phi, ex = ref_il._lin(z, coefs)
    literal = np.array([-4.0, -3.4, -2.8, -2.2, -1.6, -1.0, -0.4725, -0.08, 0.1925, 0.36, 0.4375, 0.44, 0.3825,
                        0.28, 0.1475, 0.0, -0.1475, -0.28, -0.3825, -0.44, -0.4375, -0.36, -0.1925, 0.08, 0.4725,
                        1.0, 1.6, 2.2, 2.8, 3.4, 4.0]).reshape(phi.shape)
    assert np.all(np.abs(phi - literal) < 1e-6)
    save("lin_known_answer", z=z, coefs=coefs, phi=phi, ex=ex, literal=literal)


def case_multilin():
    for name, rp, p, seed, dnff, use_ac in (
        ("multilin_p3_g6", synth.READ_PATTERN_6, 3, 11, True, True),
        ("multilin_p8_g8", synth.READ_PATTERN_8, 8, 12, True, True),
        ("multilin_p8_g8_flagfirst", synth.READ_PATTERN_8, 8, 13, False, False),
        ("multilin_p10_g16", synth.READ_PATTERN_16, 10, 14, True, True),
    ):
        c = gc.lin_case(rp, p, seed)
        f = register(f"/mem/{name}_lin.asdf", {"data": c["coefs"], "Smin": c["Smin"], "Smax": c["Smax"],
                                               "Sref": c["Sref"], "dq": c["lin_dq"]})
        ac = (~c["groupdq"] & ref_il.pixel.SATURATED) if use_ac else None
        phi, dq = ref_il.multilin(c["S"], f, do_not_flag_first=dnff, attempt_corr=ac)
        save(name, **c, do_not_flag_first=dnff, use_attempt_corr=use_ac, phi=phi, dq=dq)


def case_ipc():
    for name, seed, gdt, kdt in (("ipc_f32", 21, np.float32, np.float32), ("ipc_k64", 22, np.float32, np.float64),
                                 ("ipc_g64", 23, np.float64, np.float32), ("ipc_g64_k64", 24, np.float64, np.float64)):
        c = gc.ipc_case(seed, gdt, kdt)
        fk = register(f"/mem/{name}_ipc.asdf", {"data": c["K"]})
        fg = register(f"/mem/{name}_gain.asdf", {"data": c["gain"]})
        act = c["cube"][0, 4:-4, 4:-4]
        fwd = ref_il.ipc_fwd(act, c["K"])
        fwd_g = ref_il.ipc_fwd(act, c["K"], gain=c["gain"][4:-4, 4:-4])
        rev = ref_il.ipc_rev(act, c["K"])
        rev_g = ref_il.ipc_rev(act, c["K"], gain=c["gain"][4:-4, 4:-4])
        cube_g = c["cube"].copy()
        log = _Log()
        ref_il.correct_cube(cube_g, fk, log, gain_file=fg)
        cube_n = c["cube"].copy()
        ref_il.correct_cube(cube_n, fk, None, gain_file=None)
        save(name, **c, fwd=fwd, fwd_g=fwd_g, rev=rev, rev_g=rev_g, cube_gain=cube_g, cube_nogain=cube_n)


def case_weights():
    out = {}
    for tag, rp in (("g6", synth.READ_PATTERN_6), ("g8", synth.READ_PATTERN_8), ("g16", synth.READ_PATTERN_16)):
        meta = ref_meta(rp, synth.FRAME_TIME)
        out[f"{tag}_tbar"], out[f"{tag}_tau"], out[f"{tag}_N"] = meta["tbar"], meta["tau"], meta["N"]
        for ef in (True, False):
            for utag, u in (("udef", 0.4 / 1.8 / 6.5**2), ("ubig", 0.05)):
                out[f"{tag}_{'ex' if ef else 'in'}_{utag}"] = ref_fit.construct_weights(u, meta, exclude_first=ef)
    save("weights", **out)


def case_rampfit():
    u_def = 0.4 / 1.8 / 6.5**2
    for name, rp, seed, ef, gdt, jp in (
        ("rampfit_g8", synth.READ_PATTERN_8, 31, True, np.float32, None),
        ("rampfit_g6_custom", synth.READ_PATTERN_6, 32, True, np.float32,
         {"SthreshA": 4.0, "SthreshB": 3.0, "IthreshA": 2.0, "IthreshB": 500.0}),
        ("rampfit_g16", synth.READ_PATTERN_16, 33, True, np.float32, None),
        ("rampfit_g8_include_first", synth.READ_PATTERN_8, 34, False, np.float32, None),
        ("rampfit_g8_gain64", synth.READ_PATTERN_8, 35, True, np.float64, None),
        ("rampfit_g4", [[0], [1, 2], [3, 4, 5], [6]], 36, True, np.float32, None),
    ):
        c = gc.rampfit_case(rp, seed, exclude_first=ef, gain_dtype=gdt)
        meta = ref_meta(rp, synth.FRAME_TIME)
        meta["nborder"] = 4
        meta["K"] = ref_fit.construct_weights(u_def, meta, exclude_first=ef)
        if jp is not None:
            meta["jump_detect_pars"] = jp
        caldir = {"gain": register(f"/mem/{name}_gain.asdf", {"data": c["gain"]}),
                  "read": register(f"/mem/{name}_read.asdf", {"data": c["read"]})}
        rdq = c["groupdq"].copy()
        pdq = c["pixeldq"].copy()
        log = _Log()
        slope, er, ep = ref_fit.ramp_fit(c["data"], rdq, pdq, meta, caldir, log, exclude_first=ef)
        # one plain jump_detect pass too (full ramp), to pin smap and the untruncated flags
        loc = np.zeros_like(rdq)
        s0, er0, ep0, smap = ref_fit.jump_detect(c["data"], loc, pdq, meta, caldir, _Log(), ef, truncate_ramp=None)
        save(name, **c, read_pattern=json.dumps(rp), exclude_first=ef, K=meta["K"],
             jump_pars=json.dumps(jp), slope=slope, err_read=er, err_poisson=ep, groupdq_out=rdq, pixeldq_out=pdq,
             jd_slope=s0, jd_err_read=er0, jd_err_poisson=ep0, jd_smap=smap, jd_flags=loc)


def case_flat():
    for name, seed, gdt in (("flat_f32", 41, np.float32), ("flat_g64", 42, np.float64)):
        c = gc.flat_case(seed, gdt)
        caldir = {"flat": register(f"/mem/{name}_flat.asdf", {"data": c["flat"]}),
                  "gain": register(f"/mem/{name}_gain.asdf", {"data": c["gain"]}),
                  "ipc4d": register(f"/mem/{name}_ipc.asdf", {"data": c["K"]})}
        pdq = c["pixeldq"].copy()
        out = ref_flat.get_flat(caldir, {"nborder": 4}, pdq)
        out_nopdq = ref_flat.get_flat(caldir, {"nborder": 4}, None)
        out_noipc = ref_flat.get_flat(caldir, {"nborder": 4}, c["pixeldq"].copy(), ipc_deconvolve=False)
        save(name, **c, flat_out=out, pixeldq_out=pdq, flat_out_nopdq=out_nopdq, flat_out_noipc=out_noipc)


def case_refpix():
    """Full 4096 x 4224 frames (the reference hard-codes the geometry); inputs are regenerated from
    the seed by golden_cases.refpix_fullframe_inputs, outputs are reduced to tables + hashes."""
    for name, seed in (("refpix_full_a", 51), ("refpix_full_b", 52)):
        c = gc.refpix_fullframe_inputs(seed)
        n = 4096
        image = np.zeros((n, n + 128), dtype=np.float32)
        image[:, :n] = c["data"] - c["dark"]
        image[:, -128:] = c["amp33"] - c["med"]
        gmed = np.median(image[:, -128:])
        image[:, -128:] -= gmed
        cvar = c["C_PINK"] ** 2
        slope = (c["M_PINK"] * cvar / (c["M_PINK"] ** 2 * cvar + c["RU_PINK"] ** 2
                                       + np.median(c["std"]) ** 2 / 128 / np.log(4096)))
        ref_med = np.array([np.median(image[r, n:]) for r in range(n)])
        ctr = np.median(ref_med)
        image = ref_rs.ref_subtraction_row(image, use_ref_channel=True, slope=slope)
        after_row_hash = hashlib.sha256(image.tobytes()).hexdigest()
        bt = np.array([[np.median(image[0:4, ch * 128:(ch + 1) * 128]), np.median(image[4092:4096, ch * 128:(ch + 1) * 128])]
                       for ch in range(33)])
        image = ref_rs.ref_subtraction_channel(image, use_ref_channel=True)
        out = image[:, :n] + c["dark"]
        save(name, seed=seed, slope=np.float64(slope), amp33_median=gmed, ref_med=ref_med.astype(np.float32),
             ctr=np.float32(ctr), bottom_top=bt.astype(np.float32),
             after_row_sha256=after_row_hash, image_sha256=hashlib.sha256(image.tobytes()).hexdigest(),
             data_sha256=hashlib.sha256(out.tobytes()).hexdigest(),
             sample_rows=out[::257].copy(), sample_cols=out[:, ::331].copy())
    # the polyfit / border-pixel variant of the row step (reference unit test test_ref.py)
    c = gc.refpix_fullframe_inputs(53)
    image = np.zeros((4096, 4224), dtype=np.float32)
    image[:, :4096] = c["data"] - c["dark"]
    image = ref_rs.ref_subtraction_row(image, use_ref_channel=False)
    save("refpix_row_polyfit", seed=53, image_sha256=hashlib.sha256(image.tobytes()).hexdigest(),
         sample_rows=image[::257].copy())


def ref_test_row_image():
    """the artificial image of the reference's own unit test (tests/romanimpreprocess/test_ref.py:10-15), rebuilt here"""
    im = np.zeros((4096, 4224), dtype=np.float32)
    im[:, :] = np.cos(np.linspace(0, 2000, 4096))[:, None]
    im[:, -128:] *= 2.0
    for x in range(4224):
        im[:, x] += np.sin(0.1 * x) * np.sin(np.linspace(0, 2000, 4096)) ** 3
    im[:, :-128] += 1.0
    return im


def case_refpix_variants():
    """ref_subtraction_row / ref_subtraction_channel with the arguments calibrateimage does not use: border-pixel medians,
    fitted slope, a Python-float slope (float32 update under numpy's promotion), 32 channels, shifted / narrowed windows.
    Full 4096 x 4224 frames regenerated from the seed; outputs reduced to hashes + sampled rows."""
    c = gc.refpix_fullframe_inputs(54)
    n = 4096
    base = np.zeros((n, n + 128), dtype=np.float32)
    base[:, :n] = c["data"] - c["dark"]
    base[:, -128:] = c["amp33"] - c["med"]
    base[:, -128:] -= np.median(base[:, -128:])
    out = {"seed": 54}

    def keep(tag, image):
        out[f"{tag}_sha256"] = hashlib.sha256(image.tobytes()).hexdigest()
        out[f"{tag}_rows"] = image[::257].copy()

    keep("row_refout_fit", ref_rs.ref_subtraction_row(base.copy(), use_ref_channel=True))               # slope=None
    keep("row_border_pyfloat", ref_rs.ref_subtraction_row(base.copy(), use_ref_channel=False, slope=0.7))
    keep("row_refout_f32", ref_rs.ref_subtraction_row(base.copy(), use_ref_channel=True, slope=np.float32(0.83)))
    keep("chan_32", ref_rs.ref_subtraction_channel(base.copy()))                                        # defaults
    keep("chan_window", ref_rs.ref_subtraction_channel(base.copy(), channel_start=4, channel_end=124, use_ref_channel=True))
    keep("chan_overlap", ref_rs.ref_subtraction_channel(base.copy(), channel_start=0, channel_end=192, use_ref_channel=False))
    # the reference's own unit test image: its assertions are re-run in the tests, the output is pinned here
    im = ref_test_row_image()
    old_std = float(np.std(im))
    ref_rs.ref_subtraction_row(im, use_ref_channel=False)
    assert np.std(im) < 0.75 * old_std and 0.4 < np.std(im[:, :-128]) < 0.5 and 0.99 < np.mean(im[:, :-128]) < 1.01
    keep("test_row", im)
    save("refpix_variants", **out)


def case_jump_detect_trunc():
    """fitting.jump_detect with truncate_ramp (fitting.py:162-167: two-point weights) on the inputs of rampfit_g8 /
    rampfit_g8_include_first / rampfit_g16; outputs only (the inputs are those fixtures')."""
    u_def = 0.4 / 1.8 / 6.5**2
    out = {}
    for name, rp, seed, ef, truncs in (("rampfit_g8", synth.READ_PATTERN_8, 31, True, (4, 6, 8)),
                                       ("rampfit_g8_include_first", synth.READ_PATTERN_8, 34, False, (3, 7)),
                                       ("rampfit_g16", synth.READ_PATTERN_16, 33, True, (5, 12))):
        c = gc.rampfit_case(rp, seed, exclude_first=ef, gain_dtype=np.float32)
        meta = ref_meta(rp, synth.FRAME_TIME)
        meta["nborder"] = 4
        meta["K"] = ref_fit.construct_weights(u_def, meta, exclude_first=ef)
        caldir = {"gain": register(f"/mem/jdt_{name}_gain.asdf", {"data": c["gain"]}),
                  "read": register(f"/mem/jdt_{name}_read.asdf", {"data": c["read"]})}
        for t in truncs:
            loc = np.zeros_like(c["groupdq"])
            s, er, ep, smap = ref_fit.jump_detect(c["data"], loc, c["pixeldq"].copy(), meta, caldir, _Log(), ef, truncate_ramp=t)
            out[f"{name}_t{t}_slope"], out[f"{name}_t{t}_err_read"], out[f"{name}_t{t}_err_poisson"] = s, er, ep
            out[f"{name}_t{t}_smap"], out[f"{name}_t{t}_flags"] = smap, loc
    save("jump_detect_trunc", **out)


def case_chain():
    """multilin -> correct_cube -> construct_weights -> ramp_fit -> get_flat on one small frame, composed
    in the order of calibrateimage (gen_cal_image.py:559-629) with the reference's own functions."""
    for name, rp, p, seed, gdt, kdt in (("chain_g8", synth.READ_PATTERN_8, 8, 61, np.float32, np.float32),
                                        ("chain_g6_prod_dtypes", synth.READ_PATTERN_6, 3, 62, np.float32, np.float64)):
        cal = gc.small_cal(rp, p, seed, gain_dtype=gdt, ipc_dtype=kdt)
        ramp = synth.make_ramp(cal, read_pattern=rp, seed=seed + 1, cr_frac=0.02)
        data = ramp["data"].astype(np.float32)
        rdq, pdq = ramp["groupdq"].copy(), ramp["pixeldq"].copy()
        nb = 4
        L = cal["linearitylegendre"]
        files = {
            "linearitylegendre": register(f"/mem/{name}_lin.asdf", {k: L[k] for k in ("data", "Smin", "Smax", "Sref", "dq")}),
            "gain": register(f"/mem/{name}_gain.asdf", {"data": cal["gain"]["data"]}),
            "read": register(f"/mem/{name}_read.asdf", {"data": cal["read"]["data"]}),
            "ipc4d": register(f"/mem/{name}_ipc.asdf", {"data": cal["ipc4d"]["data"]}),
            "flat": register(f"/mem/{name}_flat.asdf", {"data": cal["flat"]["data"]}),
        }
        b = cal["biascorr"]["data"]
        data[:, nb:-nb, nb:-nb] -= b[b.shape[0] - data.shape[0]:]
        data, dq_lin = ref_il.multilin(data, files["linearitylegendre"], do_not_flag_first=rp[0] == [0],
                                       attempt_corr=~rdq & ref_il.pixel.SATURATED)
        pdq |= dq_lin
        ref_il.correct_cube(data, files["ipc4d"], _Log(), gain_file=files["gain"])
        meta = ref_meta(rp, synth.FRAME_TIME)
        meta["nborder"] = nb
        meta["K"] = ref_fit.construct_weights(0.4 / 1.8 / 6.5**2, meta, exclude_first=True)
        slope, er, ep = ref_fit.ramp_fit(data, rdq, pdq, meta, files, _Log(), exclude_first=True)
        dslope = np.array(cal["dark"]["dark_slope"], dtype=np.float32)[None]
        ref_il.correct_cube(dslope, files["ipc4d"], None, gain_file=files["gain"])
        pdq_before_flat = pdq.copy()
        flat = ref_flat.get_flat(files, meta, pdq)
        save(name, data_u16=ramp["data"], amp33=ramp["amp33"], groupdq=ramp["groupdq"], pixeldq=ramp["pixeldq"],
             read_pattern=json.dumps(rp),
             gain=cal["gain"]["data"], read=cal["read"]["data"], K4d=cal["ipc4d"]["data"], flat=cal["flat"]["data"],
             lin_data=L["data"], Smin=L["Smin"], Smax=L["Smax"], Sref=L["Sref"], lin_dq=L["dq"],
             biascorr=b, dark_slope=cal["dark"]["dark_slope"],
             cube_out=data, slope=slope, err_read=er, err_poisson=ep, groupdq_out=rdq,
             pixeldq_before_flat=pdq_before_flat, pixeldq_out=pdq, dark_slope_ipc=dslope[0], flat_out=flat, K=meta["K"])


def case_post():
    """Post-path reductions with the reference's own maskhandling.PixelMask1 and sky functions."""
    rng = np.random.default_rng(71)
    ny, nx = 90, 140
    dq = np.zeros((ny, nx), np.uint32)
    for bit in (0, 2, 3, 6, 10, 11, 12, 13, 18, 20, 24, 30, 1, 7, 31):  # masked and unmasked flags alike
        hits = rng.random((ny, nx)) < 0.004
        dq[hits] |= np.uint32(1 << bit)
    dq[0, 0] |= np.uint32(1 << 3)   # a DROPOUT in the corner (5x5 growth clipped by the frame)
    dq[ny - 1, nx - 2] |= np.uint32(1 << 10)
    mask = ref_mh.PixelMask1.build(dq)
    save("post_mask", dq=dq, mask=mask.astype(np.uint8))

    ny, nx = 268, 300   # neither a multiple of 8 nor of 4
    y, x = np.mgrid[0:ny, 0:nx]
    img = (0.8 + 0.3 * x / nx - 0.2 * (y / ny) ** 2 + 0.05 * rng.standard_normal((ny, nx))).astype(np.float32)
    for _ in range(12):   # a few bright sources
        cy, cx = rng.integers(10, ny - 10), rng.integers(10, nx - 10)
        img[cy - 3:cy + 4, cx - 3:cx + 4] += np.float32(5.0)
    m = rng.random((ny, nx)) < 0.03
    binned = ref_sky.binkxk(np.where(np.logical_not(m), img, np.nan), 4)
    ctr, width = ref_sky.smooth_mode(binned)
    withnan = img.copy()
    withnan[rng.random((ny, nx)) < 0.01] = np.nan
    withnan[0:33, 0:37] = np.nan   # one block of the 8x8 grid is empty
    out = {"img": img, "mask": m.astype(np.uint8), "binned": binned, "mode": np.array([ctr, width], dtype=np.float64),
           "withnan": withnan}
    for order in (1, 2, 3):
        coef, model = ref_sky.medfit(withnan, order=order)
        out[f"coef{order}"] = np.asarray(coef, dtype=np.float64)
        out[f"model{order}"] = model
    c8, m8 = ref_sky.medfit(img, N=4, order=2)
    out["coef_n4"], out["model_n4"] = np.asarray(c8, dtype=np.float64), m8
    save("post_sky", **out)


def case_il():
    """Inverse linearity and IL.apply with the reference's own ipc_linearity.py (ipc_linearity.py:347-513)."""
    for name, (seed, p, kdt, cdt) in gc.IL_CASES.items():
        c = gc.il_case(seed, p, kdt, cdt)
        lin = register(f"/mem/{name}_lin.asdf", {"data": c["coefs"], "Smin": c["Smin"], "Smax": c["Smax"], "Sref": c["Sref"],
                                                 "dq": c["lin_dq"]})
        gain = register(f"/mem/{name}_gain.asdf", {"data": c["gain"]})
        ipc = register(f"/mem/{name}_ipc.asdf", {"data": c["K"]})
        out = {}
        with np.errstate(all="ignore"):
            raw = (c["Smin"] + (c["Smax"] - c["Smin"]) * np.linspace(-0.05, 1.05, c["Smin"].size).reshape(c["Smin"].shape)).astype(np.float32)
            out["raw"] = raw[6:40, 3:60].copy()
            out["lin_phi"], out["lin_dq"] = ref_il.linearity(out["raw"], lin, origin=(3, 6))
            S, ex = ref_il.invlinearity(c["counts"], lin, origin=(4, 4))
            out["inv_S"], out["inv_ex"] = S, ex.astype(np.uint8)
            out["apply_dn"] = ref_il.IL(lin, gain, ipc).apply(c["counts"])
            out["apply_e_in"] = ref_il.IL(lin, gain, ipc).apply(c["counts"], electrons=True)
            out["apply_e_both"] = ref_il.IL(lin, gain, ipc, start_e=c["start_e"]).apply(c["counts"], electrons=True, electrons_out=True)
            out["apply_noipc"] = ref_il.IL(lin, gain, None, start_e=25.0).apply(c["counts"], electrons_out=True)
            il = ref_il.IL(lin, gain, ipc)
            il.set_dq(ngroup=3)
            out["il_dq"] = il.dq
        out["lin_out_dq"] = out.pop("lin_dq")
        save(name, **c, **out)


def case_il_example():
    """The reference's known-answer test of the IL class (tests/romanimpreprocess/test_workflow.py:382-422: two literal
    2 x 3 arrays) on the reference's own synthetic CALDIR.  `gencal` (:117-332, RandomState(1000), numpy only) and
    `il_example` are taken from the test file with `ast` at run time and executed as they stand, with
    asdf.AsdfFile(tree).write_to(path) keeping the trees in memory.  The fixture keeps the blocks of the calibration arrays
    around the pixels the test looks at, the literals, and what the reference's IL class returns there."""
    import ast

    path = "/root/reference/tests/romanimpreprocess/test_workflow.py"
    tree = ast.parse(open(path).read())
    keep = [n for n in tree.body if (isinstance(n, ast.FunctionDef) and n.name in ("_trim", "gencal", "il_example"))
            or isinstance(n, ast.Assign)]
    mod = ast.Module(body=keep, type_ignores=[])

    class _AF:
        def __init__(self, tr):
            self.tr = tr

        def write_to(self, path_):
            _STORE[path_] = self.tr

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

    sys.modules["asdf"].AsdfFile = _AF
    captured = {}

    class _ILSpy(ref_il.IL):   # the reference's class, recording what apply returns
        def apply(self, counts, **kw):
            out = super().apply(counts, **kw)
            captured.setdefault("outs", []).append(np.array(out[250:272, 130:153]))
            return out

    spy = types.SimpleNamespace(**{k: getattr(ref_il, k) for k in dir(ref_il) if not k.startswith("__")})
    spy.IL = _ILSpy
    ns = {"np": np, "asdf": sys.modules["asdf"], "pixel": sys.modules["roman_datamodels.dqflags"].pixel,
          "ipc_linearity": spy, "os": os}
    exec(compile(mod, path, "exec"), ns)
    ns["gencal"]("/mem/kat", np.random.RandomState(1000))
    files = {k: next(p_ for p_ in _STORE if p_.startswith("/mem/kat_" + k + "_")) for k in ("linearitylegendre", "gain", "ipc4d")}
    ns["il_example"](files["linearitylegendre"], files["gain"], files["ipc4d"])   # the reference's own assertions pass
    targets = [np.array(ast.literal_eval(n.value.args[0]))
               for f in keep if isinstance(f, ast.FunctionDef) and f.name == "il_example"
               for n in f.body if isinstance(n, ast.Assign) and ast.unparse(n.targets[0]) in ("target1", "target2")]
    lin = _STORE[files["linearitylegendre"]]["roman"]
    ya, xa = slice(250, 272), slice(130, 153)          # active-region block; the test reads [260:262, 140:143]
    yf, xf = slice(254, 276), slice(134, 157)          # the same block in full-frame coordinates (border 4)
    save("il_example", coefs=np.array(lin["data"][:, yf, xf]), Smin=np.array(lin["Smin"][yf, xf]),
         Smax=np.array(lin["Smax"][yf, xf]), gain=np.array(_STORE[files["gain"]]["roman"]["data"][yf, xf]),
         K=np.array(_STORE[files["ipc4d"]]["roman"]["data"][:, :, ya, xa]), block_origin=np.array([250, 130]),
         target1=targets[0], target2=targets[1], ref_out1=captured["outs"][0], ref_out2=captured["outs"][1])


def case_harness():
    """Statistics over many realisations: the reference's validation_tests/many_realizations.py is EXECUTED as it stands
    (read from /root/reference at run time) with its simulation and calibration calls replaced by stand-ins that put the
    synthetic realisations of golden_cases.harness_realisation where the script looks for its L1 and L2 files; the FITS
    reader / writer are stand-ins too.  Outputs are reduced to hashes and sampled rows."""
    import shutil
    import yaml

    from romanimpreprocess import pars as ref_pars

    script = "/root/reference/validation_tests/many_realizations.py"
    tmp = os.path.join(REPO, "gpurun_out", "_tmp_harness")
    fits = sys.modules["astropy.io.fits"]
    state = {}

    class _Hdu:
        def __init__(self, data, header=None):
            self.data, self.header = data, header or {}

        def writeto(self, path, overwrite=False):
            state["out"] = np.array(self.data)

    class _Open:
        def __init__(self, hdus):
            self.hdus = hdus

        def __enter__(self):
            return self.hdus

        def __exit__(self, *exc):
            return False

    fits.open = lambda path: _Open([_Hdu(state["truth"], {"EXPTIME": gc.HARNESS_EXPTIME})])
    fits.PrimaryHDU = _Hdu

    sim = types.ModuleType("romanimpreprocess.from_sim.sim_to_isim")
    cal = types.ModuleType("romanimpreprocess.L1_to_L2.gen_cal_image")

    def run_config(cfg):   # "simulate": the realisation index follows from the seed the script has just advanced
        j = (cfg["SEED"] - 100) // 10 - 1
        state["r"] = gc.harness_realisation(state["seed"], j, state["ideal_act"])
        r = state["r"]
        cube = np.stack([r["l1_first"] - 7, r["l1_first"], r["l1_first"] + 9, r["l1_last"]])
        register(cfg["OUT"], {"data": cube})

    def calibrateimage(cfg):
        r = state["r"]
        register(cfg["OUT"], {"data": r["data"], "err": r["err"], "dq": r["dq"]})

    sim.run_config, cal.calibrateimage = run_config, calibrateimage
    pk_sim, pk_cal = types.ModuleType("romanimpreprocess.from_sim"), types.ModuleType("romanimpreprocess.L1_to_L2")
    pk_sim.sim_to_isim, pk_cal.gen_cal_image = sim, cal
    pk_val = types.ModuleType("romanimpreprocess.validation_tests")
    pk_val.__path__ = []
    sys.modules.update({"romanimpreprocess.from_sim": pk_sim, "romanimpreprocess.from_sim.sim_to_isim": sim,
                        "romanimpreprocess.L1_to_L2": pk_cal, "romanimpreprocess.L1_to_L2.gen_cal_image": cal,
                        "romanimpreprocess.validation_tests": pk_val})
    source = open(script).read()
    for name, c in gc.HARNESS_CASES.items():
        shutil.rmtree(tmp, ignore_errors=True)
        os.makedirs(tmp)
        state.update(seed=c["seed"], truth=gc.harness_truth(c["seed"]))
        big = np.zeros((4096, 4096), np.float32)
        big[4:-4, 4:-4] = state["truth"] / float(gc.HARNESS_EXPTIME) / ref_pars.g_ideal
        state["ideal_act"] = gc.harness_orient(big, c["scanum"])[4:-4, 4:-4].copy()
        cfg1 = {"IN": os.path.join(tmp, f"truth_{c['scanum']}.fits"), "OUT": os.path.join(tmp, "l1.asdf")}
        cfg2 = {"IN": cfg1["OUT"], "OUT": os.path.join(tmp, "l2.asdf")}
        for k, cfg in (("c1.yaml", cfg1), ("c2.yaml", cfg2)):
            with open(os.path.join(tmp, k), "w") as f:
                yaml.safe_dump(cfg, f)
        argv = sys.argv
        sys.argv = ["many_realizations", os.path.join(tmp, "c1.yaml"), os.path.join(tmp, "c2.yaml"), str(c["nrun"]), tmp]
        try:
            exec(compile(source, script, "exec"), {"__name__": "romanimpreprocess.validation_tests.many_realizations",
                                                   "__package__": "romanimpreprocess.validation_tests"})
        finally:
            sys.argv = argv
        out = state.pop("out")
        assert out.shape == (8, 4096, 4096) and out.dtype == np.float32
        save(name, seed=c["seed"], nrun=c["nrun"], scanum=c["scanum"], g_ideal=np.float64(ref_pars.g_ideal),
             plane_sha256=np.array([hashlib.sha256(np.ascontiguousarray(p)).hexdigest() for p in out]),
             sample=out[:, list(gc.HARNESS_ROWS)][:, :, ::29].copy(),
             block=out[:, 96:112, 196:220].copy(), nanpix=out[:, 504, 604].copy())
        shutil.rmtree(tmp, ignore_errors=True)


def case_pearson():
    """The reference's GalPoisson modules (importable as they stand: numpy + scipy only): get_tilde_nus for several MA tables and
    weight vectors, and -- with the per-type SAMPLERS replaced by recorders, so that only the reference's classification and
    parameter formulas run -- the Pearson type and the distribution parameters of every element of intensity arrays."""
    import importlib

    ft = importlib.import_module("romanimpreprocess.L1_to_L2.GalPoisson.find_tilnus")
    dw = importlib.import_module("romanimpreprocess.L1_to_L2.GalPoisson.draw_with_tilnus")
    out = {}
    # ---- get_tilde_nus
    tables = {
        "doc": ([1, 2, 4, 4, 4, 1], [2, 3, 5, 23, 44, 49], [-0.1, -0.4, -0.2, 0.2, 0.4, 0.1]),
        "pair": ([4, 4], [5, 12], [-0.5, 0.5]),
    }
    from romanimpreprocess_amd import synth
    READ_PATTERNS = {"g8": synth.READ_PATTERN_8, "g6": synth.READ_PATTERN_6, "g16": synth.READ_PATTERN_16}
    fitting = importlib.import_module("romanimpreprocess.utils.fitting")
    for name, rp in READ_PATTERNS.items():
        meta = ref_meta(rp, 3.04)
        K = fitting.construct_weights(0.4 / 1.8 / 6.5**2, meta, exclude_first=True)
        nb_, ab_ = [len(r) for r in rp], [r[0] for r in rp]
        tables["K_" + name] = (nb_, ab_, np.asarray(K, dtype=np.float64))
        Kt = np.zeros(len(rp), np.float32)   # a two-point weight vector of a truncated ramp (gen_noise_image.py:192-197)
        Kt[3] = 1.0 / (meta["tbar"][3] - meta["tbar"][1])
        Kt[1] = -Kt[3]
        tables["trunc_" + name] = (nb_, ab_, Kt)
    for name, (nb_, ab_, w) in tables.items():
        nus = ft.get_tilde_nus(np.array(nb_), np.array(ab_), np.asarray(w))
        out[f"tn_{name}_N"], out[f"tn_{name}_a"], out[f"tn_{name}_W"] = np.array(nb_), np.array(ab_), np.asarray(w)
        out[f"tn_{name}_out"] = np.array(nus, dtype=np.float64)
    # ---- classification + parameters: samplers replaced by recorders
    rec = {}

    def r1(t21, t31, t41, I, rng=None):
        rec["p1"] = np.stack(dw.solve_beta_parameters_vec(t21, t31, t41, I), axis=-1)
        return np.full(np.shape(I), 1.0)

    def r3(t21, t31, I, rng=None):
        I = np.asarray(I, dtype=float)
        scale = abs(t31) / (2.0 * t21)
        shapes = 4.0 * t21**3 * I / t31**2
        rec["p3"] = np.stack([shapes, np.full(I.shape, scale), shapes * scale, np.full(I.shape, 1.0 if t31 > 0 else -1.0)], axis=-1)
        return np.full(np.shape(I), 3.0)

    def r5(t21, t31, I, rng=None):
        a, b, mu = dw.solve_pearson5_parameters_vec(t21, t31, I)
        rec["p5"] = np.stack([a, b, mu, np.full(np.shape(a), 1.0 if t31 >= 0 else -1.0)], axis=-1)
        return np.full(np.shape(I), 5.0)

    def r6(t21, t31, t41, I, rng=None):
        al, be, sc, sh, _sg = dw.solve_pearson6_params(t21, t31, t41, I)
        rec["p6"] = np.stack([al, be, sc, sh], axis=-1)
        return np.full(np.shape(I), 6.0)

    p4 = []

    def dev(m, nu, *, a=1.0, lam=0.0, size=None, rng=None):
        p4.append((m, nu, a, lam))
        return 4.0

    def ar(m, nu, a, lam, rng=None):
        p4.append((m, nu, a, lam))
        return 4.0

    saved = {k: getattr(dw, k) for k in ("random_from_type1", "random_from_type3", "random_from_type5", "random_from_type6",
                                         "pt4_rvs_devroye", "pt4_rvs_ar", "devroye_acc_rate")}
    dw.devroye_acc_rate = lambda nu, a, m: 1.0   # every type-4 element goes to the first recorder, in element order
    dw.random_from_type1, dw.random_from_type3, dw.random_from_type5, dw.random_from_type6 = r1, r3, r5, r6
    dw.pt4_rvs_devroye, dw.pt4_rvs_ar = dev, ar
    try:
        I = np.concatenate([np.geomspace(1e-3, 1e5, 97), [0.0, -5.0, 0.01, 0.02, 3.0, 1e7]])
        cases = {
            "poisson_like": (1.0, 1.0, 1.0),          # kappa = t41 t21 / t31^2 = 1: type 1
            "neg_skew": (0.8, -0.6, 0.5),             # type 1, negative skew
            "beta_prime": (1.0, 1.0, 1.7),            # 1.5 < kappa < 1.875: type 6
            "beta_prime_neg": (0.5, -0.4, 0.55),      # type 6 (kappa = 1.72), negative skew
            "heavy_tail": (1.0, 0.5, 1.0),            # kappa = 4: type 4
            "heavy_tail_neg": (2.0, -1.0, 3.0),       # type 4, negative third moment
            "symmetric": (1.0, 0.0, 2.0),             # beta_1 = 0: type 4 with nu = 0 (Student-like)
            "light_tail": (1.0, 0.1, -0.5),           # negative excess kurtosis: type 1 near-symmetric beta
        }
        tn = tables["K_" + next(iter(READ_PATTERNS))]
        nus = ft.get_tilde_nus(np.array(tn[0]), np.array(tn[1]), np.asarray(tn[2]))
        cases["ramp_fit_weights"] = (nus[0] * 3.04, nus[1] * 3.04**2, nus[2] * 3.04**3)   # as gen_noise_image.py:214-217 scales them
        for name, (t21, t31, t41) in cases.items():
            rec.clear()
            p4.clear()
            types = dw.draw_from_Pearson(t21, t31, t41, I, rng=np.random.default_rng(1)).astype(np.int32)
            par = np.zeros(I.shape + (4,))
            Ic = np.clip(I, 0.01, None)
            for code, key in ((1, "p1"), (3, "p3"), (5, "p5"), (6, "p6")):
                if np.any(types == code):
                    par[types == code] = rec[key]
            if np.any(types == 4):
                par[types == 4] = np.array(p4)
            del Ic
            out[f"cl_{name}_t"] = np.array([t21, t31, t41])
            out[f"cl_{name}_types"], out[f"cl_{name}_params"] = types, par
        out["cl_I"] = I
    finally:
        for k, v in saved.items():
            setattr(dw, k, v)
    save("pearson_params", **out)


def case_noise1f():
    """``noise_1f_frame`` of from_sim/sim_to_isim.py (:265-303): the module cannot be imported (romanisim, galsim, astropy), but
    the function is numpy-only apart from the deviate draw, so it is taken from the file with ``ast`` and EXECUTED as it stands,
    with ``galsim.GaussianDeviate(rng).generate(a)`` standing in as "fill a from the normals handed in" and ``pars`` being the
    reference's own pars module.  The fixture keeps the seed of the normals, samples of the frame and its SHA-256."""
    import ast
    import importlib

    path = os.path.join(REF_SRC, "romanimpreprocess", "from_sim", "sim_to_isim.py")
    tree = ast.parse(open(path).read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "noise_1f_frame"]
    assert len(keep) == 1
    ref_pars = importlib.import_module("romanimpreprocess.pars")

    class _GD:
        def __init__(self, rng):
            self.rng = rng

        def generate(self, a):
            a[...] = self.rng.standard_normal(a.size).reshape(a.shape)

    ns = {"np": np, "pars": ref_pars, "galsim": types.SimpleNamespace(GaussianDeviate=_GD)}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    out = {}
    for seed in (3, 4):
        frame = ns["noise_1f_frame"](np.random.default_rng(seed))
        assert frame.shape == (ref_pars.nside, ref_pars.channelwidth) and frame.dtype == np.float32
        out[f"s{seed}_sha256"] = np.array(hashlib.sha256(np.ascontiguousarray(frame)).hexdigest())
        out[f"s{seed}_rows"] = frame[::257].copy()
        out[f"s{seed}_std"] = np.float64(np.std(frame))
    save("noise_1f_frame", seeds=np.array([3, 4]), **out)


def case_noise_arith():
    """The arithmetic of the noise-layer loop, ``make_noise_cube`` of L1_to_L2/gen_noise_image.py (:60-331): white read-noise
    injection into the Level-1 cube (:120-134) and the resampled-Poisson layers (:242-324).  The module cannot be imported
    (galsim, astropy, romanisim at import), so the function is taken from the file with ``ast`` and EXECUTED as it stands on a
    64 x 64 frame (``pars.nside_active`` = 56, ``pars.nborder`` = 4 in the namespace it runs in) with recording stand-ins:
    ``galsim.GaussianDeviate(rng).generate(a)`` / ``galsim.PoissonDeviate(rng).generate_from_expectation(a)`` fill from a numpy
    generator and record what they drew; ``asdf`` serves and stores in-memory trees; ``fill_in_refdata_and_1f`` does nothing
    (its arithmetic is pinned by the l1sim fixture); ``calibrateimage`` records the cube it is handed -- the INJECTED cube -- and
    answers with an L2 tree whose data is a fixed linear function of that cube.  ``sky`` is the reference's own module."""
    import ast
    import copy
    import importlib

    path = os.path.join(REF_SRC, "romanimpreprocess", "L1_to_L2", "gen_noise_image.py")
    tree = ast.parse(open(path).read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("make_noise_cube", "_get_subscript")]
    assert len(keep) == 2
    ref_sky = importlib.import_module("romanimpreprocess.utils.sky")
    rng = np.random.default_rng(71)
    ny = nx = 64
    nb = 4
    rp = [[0], [1], [2, 3], [4, 5, 6]]
    G = len(rp)
    frame_time = 3.04
    cube = (3000 + 40 * np.arange(G)[:, None, None] + rng.integers(0, 900, size=(G, ny, nx))).astype(np.uint16)
    cube[2, 10, 10] = 65534          # the injection clips at the top of the u16 range ...
    cube[1, 11, 11] = 1              # ... and at zero
    dark = (2900 + rng.integers(0, 300, size=(G + 1, ny, nx)) / 4.0).astype(np.float32)     # one reference read in front (de = 1)
    read = (5.0 + 4.0 * rng.random((ny, nx))).astype(np.float32)
    gain = (1.5 + 0.05 * rng.standard_normal((ny, nx))).astype(np.float32)
    gain[20, 20] = 0.0               # clipped to 1e-4
    withsky = (0.3 + 1.5 * rng.random((ny - 2 * nb, nx - 2 * nb)) ** 3).astype(np.float32)
    withsky[5, 5] = -0.4             # negative sky: no electrons
    endslice = rng.integers(-1, G, size=(ny - 2 * nb, nx - 2 * nb)).astype(np.int8)
    weights = np.array([0.0, -0.11, 0.02, 0.09], dtype=np.float32)
    tbar = np.array([frame_time * np.mean(g) for g in rp], dtype=np.float32)

    drawn = {"normals": [], "poisson": []}

    class _GD:
        def __init__(self, r):
            pass

        def generate(self, a):
            v = rng.standard_normal(a.shape).astype(a.dtype)
            drawn["normals"].append(v.copy())
            a[...] = v

    class _PD:
        def __init__(self, r):
            pass

        def generate_from_expectation(self, a):
            v = rng.poisson(a).astype(a.dtype)
            drawn["poisson"].append(v.copy())
            a[...] = v

    class _TreeDict(dict):
        @property
        def tree(self):
            return self

    class _Open:
        def __init__(self, key):
            self.t = _TreeDict(_STORE[key])

        def __enter__(self):
            return self.t

        def __exit__(self, *exc):
            return False

    class _AF:
        def __init__(self, t):
            self.t = t

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

        def write_to(self, f):
            _STORE[f.name] = copy.deepcopy(dict(self.t))

    captured = []

    def calibrateimage(cfg):
        t = _STORE[cfg["IN"]]
        c = np.array(t["roman"]["data"])
        captured.append(c.copy())
        act = c[:, nb:-nb, nb:-nb].astype(np.float32)
        _STORE[cfg["OUT"]] = {"roman": {"data": (act[-1] - act[1]) / np.float32(7.0)}}

    tmp = "/tmp/make_goldens_noise_arith"
    os.makedirs(tmp, exist_ok=True)
    base = {"roman": {"data": cube.copy(), "amp33": np.zeros((G, ny, 128), np.uint16),
                      "meta": {"exposure": {"read_pattern": rp, "frame_time": frame_time}}}}
    act0 = cube[:, nb:-nb, nb:-nb].astype(np.float32)
    l2 = {"roman": {"data": (act0[-1] - act0[1]) / np.float32(7.0), "data_withsky": withsky,
                    "meta": {"exposure": {"read_pattern": rp, "frame_time": frame_time}}},
          "processinfo": {"meta": {"tbar": tbar, "read_pattern": rp}, "weights": weights, "exclude_first": True, "endslice": endslice}}
    _STORE["/mem/na_l1.asdf"], _STORE["/mem/na_l2.asdf"] = base, l2
    register("/mem/na_dark.asdf", {"data": dark})
    register("/mem/na_read.asdf", {"data": read})
    register("/mem/na_gain.asdf", {"data": gain})
    layers = ["Ra", "R", "Pr", "Pb1r", "RaPr"]
    config = {"IN": "/mem/na_l1.asdf", "OUT": "/mem/na_l2.asdf",
              "CALDIR": {"dark": "/mem/na_dark.asdf", "read": "/mem/na_read.asdf", "gain": "/mem/na_gain.asdf"},
              "NOISE": {"LAYER": layers, "TEMP": os.path.join(tmp, "temp.asdf")}}
    asdf_ns = types.SimpleNamespace(open=lambda key, *a, **k: _Open(key), AsdfFile=_AF)
    ns = {"np": np, "asdf": asdf_ns, "galsim": types.SimpleNamespace(GaussianDeviate=_GD, PoissonDeviate=_PD), "sys": sys,
          "re": __import__("re"), "deepcopy": copy.deepcopy, "pars": types.SimpleNamespace(nside_active=ny - 2 * nb, nborder=nb),
          "fill_in_refdata_and_1f": lambda *a, **k: None, "calibrateimage": calibrateimage, "sky": ref_sky,
          "get_tilde_nus": None, "draw_from_Pearson": None}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    import contextlib
    import io

    with contextlib.redirect_stdout(io.StringIO()):
        noise = ns["make_noise_cube"](config, object())
    import shutil

    shutil.rmtree(tmp, ignore_errors=True)
    assert noise.shape == (len(layers), ny - 2 * nb, nx - 2 * nb) and noise.dtype == np.float32
    # captured cubes: "Ra" -> 1 (injected data); "R" -> 2 (the dark cube itself, then the injected dark); "RaPr" -> 1
    assert len(captured) == 4 and len(drawn["normals"]) == 3 * G and len(drawn["poisson"]) == 3 * (rp[-1][-1] + 1)
    sky_b1 = ref_sky.medfit(withsky, order=1)[1]
    save("noise_arith", cube=cube, dark=dark, read=read, gain=gain, withsky=withsky, endslice=endslice, weights=weights, tbar=tbar,
         read_pattern=json.dumps(rp), frame_time=np.float64(frame_time), layers=json.dumps(layers),
         normals=np.stack(drawn["normals"]).reshape(3, G, ny - 2 * nb, nx - 2 * nb),
         poisson=np.stack(drawn["poisson"]).reshape(3, rp[-1][-1] + 1, ny - 2 * nb, nx - 2 * nb),
         injected_Ra=captured[0], dark_as_data=captured[1], injected_R=captured[2], injected_RaPr=captured[3],
         sky_b1=sky_b1.astype(np.float32), noise=noise)


def case_l1sim():
    """``make_l1_fullcal`` (:163-262) and ``fill_in_refdata_and_1f`` (:306-403) of from_sim/sim_to_isim.py, taken from the file
    with ``ast`` and EXECUTED as they stand on a 32 x 512 frame (the smallest one that satisfies the border rules the two
    functions and ``IL.apply`` hard-code: (8192 - 504 // 2) % 256 == 4, (8192 - 24 // 2) % 16 == 4; ``pars.nside`` = 32 rows and
    ``pars.channelwidth`` = 16 columns in the namespace they run in).  ``IL`` is the reference's own class
    (utils/ipc_linearity.py, imported unmodified).  Stand-ins: ``galsim.GaussianDeviate(rng).generate(a)`` fills ``a`` from a
    numpy generator and records what it drew; ``asdf.open`` serves in-memory trees; ``rstl1`` (romanisim.l1, absent from the
    reference tree and from this image) is oracle/l1sim.py's restatement of its published algorithm, cosmic rays and
    persistence off.  The fixture stores the calibration arrays, every deviate handed out (for the 1/f part the frames ``noise_1f_frame`` returned), and the outputs."""
    import ast
    import copy
    import warnings

    from oracle import l1sim

    ny, nx, nb, cw = 32, 512, 4, 16
    rp = [[0], [1, 2], [3, 4, 5, 6], [7]]
    read_time = 3.04
    cal = synth.make_caldir(ny, nx, read_pattern=rp, p_order=3, seed=901, bias_amplitude=2.0)
    # make_caldir sizes the reference-output statistics for 128-column channels: cut them to this frame's 16
    a33 = cal["read"]["amp33"]
    a33["med"], a33["std"] = np.ascontiguousarray(a33["med"][:, :cw]), np.ascontiguousarray(a33["std"][:, :cw])
    files = {k: register(f"/mem/l1sim_{k}.asdf", cal[k]) for k in ("read", "gain", "dark", "biascorr", "linearitylegendre", "ipc4d")}

    path = os.path.join(REF_SRC, "romanimpreprocess", "from_sim", "sim_to_isim.py")
    tree = ast.parse(open(path).read())
    names = ("make_l1_fullcal", "noise_1f_frame", "fill_in_refdata_and_1f")
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(keep) == 3
    drawn, captured = [], {}

    class _Rng:   # what the extracted code passes around as `rng`
        def __init__(self, seed):
            self.normal = np.random.default_rng(seed)
            self.binom = np.random.default_rng(seed + 1)

    class _GD:
        def __init__(self, rng):
            self.rng = rng

        def generate(self, a):
            a[...] = self.rng.normal.standard_normal(a.size).reshape(a.shape)
            drawn.append(np.array(a, copy=True))

    def _tij(read_pattern):
        return l1sim.read_pattern_to_tij(read_pattern, read_time)

    def _apportion(counts, tij, inv_linearity=None, crparam=None, persistence=None, tstart=None, rng=None, seed=None):
        reads_e = l1sim.binomial_shares(counts, tij, rng.binom)
        captured["reads_e"] = reads_e
        res = l1sim.apportion_counts_to_resultants(reads_e, tij, lambda e: inv_linearity.apply(e, electrons=True))
        dq = np.zeros(res.shape, dtype=np.uint32)
        dq |= inv_linearity.dq
        return res, dq

    def _readnoise(resultants, tij, rng=None, seed=None, read_noise=None):
        nrm = np.zeros(resultants.shape, dtype="f4")
        _GD(rng).generate(nrm)
        return l1sim.add_read_noise_to_resultants(resultants, tij, read_noise, nrm)

    ns = {"np": np, "galsim": types.SimpleNamespace(GaussianDeviate=_GD), "asdf": sys.modules["asdf"], "copy": copy,
          "warnings": warnings, "IL": ref_il.IL, "u": types.SimpleNamespace(DN=1),
          "pars": types.SimpleNamespace(nborder=nb, nside=ny, channelwidth=cw), "parameters": types.SimpleNamespace(nborder=nb),
          "rstl1": types.SimpleNamespace(read_pattern_to_tij=_tij, apportion_counts_to_resultants=_apportion,
                                         add_read_noise_to_resultants=_readnoise)}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    frames_made, make_frame = [], ns["noise_1f_frame"]

    def _recording_frame(rng):   # the reference's generator (pinned by the noise1f case); its frames are kept, not their 2**11 deviates
        n_before = len(drawn)
        f = make_frame(rng)
        del drawn[n_before:]
        frames_made.append(f.copy())
        return f

    ns["noise_1f_frame"] = _recording_frame

    rng0 = np.random.default_rng(902)
    counts = rng0.poisson(np.where(rng0.random((ny - 2 * nb, nx - 2 * nb)) < 0.03, 9.0e4, 2500.0)).astype(np.float32)
    res, dq = ns["make_l1_fullcal"](types.SimpleNamespace(array=counts), rp, files, rng=_Rng(903))
    normals_reset, normals_read = drawn[0], drawn[1]
    del drawn[:]
    im = l1sim.embed(np.asarray(res), ny, nx, nb)
    im_before = im.copy()
    amp33 = np.zeros((len(rp), ny, cw), dtype=np.uint16)
    ns["fill_in_refdata_and_1f"](im, files, _Rng(904), _tij(rp), fill_in_banding=True, amp33=amp33)
    # draw order: the (ngrp+1, ny, nx) block, then per group: common frame, 32 channel frames, white (ny, cw), amp33's frame
    G = len(rp)
    normals_fill, white33 = drawn[0], np.stack(drawn[1:])
    assert white33.shape == (G, ny, cw) and len(frames_made) == 34 * G
    frames = np.stack(frames_made).reshape(G, 34, ny, cw)
    ref_first, rest = l1sim.extract_ref(im, 1000)   # EXTRACT_REF (:711-730) restated; the block is inline code of a method
    flat = {}
    for k in ("read", "gain", "dark", "biascorr", "linearitylegendre", "ipc4d"):
        for kk, v in cal[k].items():
            if isinstance(v, dict):
                for k3, v3 in v.items():
                    flat[f"cal_{k}_{kk}_{k3}"] = np.asarray(v3)
            else:
                flat[f"cal_{k}_{kk}"] = np.asarray(v)
    save("l1sim", read_pattern_flat=np.array([r for g in rp for r in g]), read_pattern_counts=np.array([len(g) for g in rp]),
         read_time=np.float64(read_time), counts=counts, normals_reset=normals_reset, normals_read=normals_read,
         reads_e=captured["reads_e"], resultants=np.asarray(res, dtype=np.float32), dq=dq, im_before=im_before,
         normals_fill=normals_fill.astype(np.float32), frames=frames, white33=white33.astype(np.float32),
         im_after=im, amp33_after=amp33, **flat)


CASES = {
    "lin_known_answer": case_lin_known_answer, "multilin": case_multilin, "ipc": case_ipc,
    "weights": case_weights, "rampfit": case_rampfit, "flat": case_flat, "refpix": case_refpix,
    "chain": case_chain, "post": case_post, "harness": case_harness, "il": case_il, "il_example": case_il_example,
    "pearson": case_pearson, "noise1f": case_noise1f, "l1sim": case_l1sim,
    "refpix_variants": case_refpix_variants, "jump_detect_trunc": case_jump_detect_trunc, "noise_arith": case_noise_arith,
}


def run_case(k):
    """One case with sys.modules restored afterwards: the cases that EXECUTE reference scripts put stand-in packages
    (romanimpreprocess.L1_to_L2, ...) there, which a later import of the real sub-packages would trip over."""
    before = dict(sys.modules)
    try:
        CASES[k]()
    finally:
        for name in list(sys.modules):
            if name not in before:
                del sys.modules[name]
            elif sys.modules[name] is not before[name]:
                sys.modules[name] = before[name]
        for name, mod in before.items():
            sys.modules.setdefault(name, mod)


def compare_dirs(new, old, names):
    """array-by-array comparison of the fixtures `names` (file stems); returns the list of differences"""
    diffs = []
    for stem in names:
        a_path, b_path = os.path.join(new, stem + ".npz"), os.path.join(old, stem + ".npz")
        if not os.path.exists(b_path):
            diffs.append(f"{stem}: not in {old}")
            continue
        with np.load(a_path, allow_pickle=False) as a, np.load(b_path, allow_pickle=False) as b:
            if sorted(a.files) != sorted(b.files):
                diffs.append(f"{stem}: arrays {sorted(set(a.files) ^ set(b.files))} on one side only")
            for key in sorted(set(a.files) & set(b.files)):
                x, y = a[key], b[key]
                if x.dtype != y.dtype or x.shape != y.shape or x.tobytes() != y.tobytes():
                    diffs.append(f"{stem}[{key}]: {x.dtype}{x.shape} vs {y.dtype}{y.shape}, contents differ")
    return diffs


def main(argv):
    global OUT
    args = list(argv)
    check = "--check" in args
    if check:
        args.remove("--check")
    if "--out" in args:
        i = args.index("--out")
        OUT = os.path.abspath(args[i + 1])
        del args[i:i + 2]
    unknown = [k for k in args if k not in CASES]
    if unknown:
        raise SystemExit(f"unknown case(s) {unknown}; known: {' '.join(CASES)}")
    want = args or list(CASES)
    golden = os.path.join(REPO, "tests", "golden")
    if check:
        import tempfile
        OUT = tempfile.mkdtemp(prefix="goldens_check_")
    for k in want:
        print(k, flush=True)
        run_case(k)
    if check:
        import shutil
        made = sorted(f[:-4] for f in os.listdir(OUT) if f.endswith(".npz"))
        diffs = compare_dirs(OUT, golden, made)
        shutil.rmtree(OUT, ignore_errors=True)
        for d in diffs:
            print("DIFFERENT", d)
        print(f"checked {len(made)} fixture(s) of {len(want)} case(s): {'%d difference(s)' % len(diffs) if diffs else 'identical'}")
        return 1 if diffs else 0
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
