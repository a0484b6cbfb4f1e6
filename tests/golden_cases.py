"""Seeded input builders for the golden fixtures under ``tests/golden/``.

Used by ``tools/make_goldens.py`` (which runs the REFERENCE on these inputs, in the build
container only) and re-used by tests that want extra seeded inputs.  The fixtures store both the
inputs and the reference's outputs, so the tests never depend on regenerating floats
bit-identically on another machine; the one exception is the full-frame reference-pixel case,
whose inputs are built from integer draws only (PCG64 integers are platform independent) and are
regenerated instead of stored (69 MB per group otherwise).
"""

import numpy as np

from romanimpreprocess_amd import synth
from romanimpreprocess_amd.dqflags import pixel

SMALL = (40, 72)  # (ny, nx) of the per-pixel stage fixtures; deliberately not a multiple of 64


def small_cal(read_pattern, p_order, seed, gain_dtype=np.float32, ipc_dtype=np.float32, shape=SMALL):
    """Synthetic CALDIR arrays on a small frame (nx need not be a multiple of 128 here: no refpix)."""
    ny, nx = shape
    # make_caldir only needs nx % 128 == 0 for the amp33 block, which per-pixel stages never touch
    cal = synth.make_caldir(ny, nx, read_pattern=read_pattern, p_order=p_order, seed=seed,
                            gain_dtype=gain_dtype, ipc_dtype=ipc_dtype, bias_amplitude=2.0,
                            bad_lin_frac=0.01)
    return cal


def lin_case(read_pattern, p_order, seed):
    """Raw cube (as f32, i.e. after dq-init) with out-of-range and saturated samples + cal arrays."""
    cal = small_cal(read_pattern, p_order, seed)
    ramp = synth.make_ramp(cal, read_pattern=read_pattern, seed=seed + 1, cr_frac=0.02)
    S = ramp["data"].astype(np.float32)
    G, ny, nx = S.shape
    rng = np.random.default_rng(seed + 2)
    lin = cal["linearitylegendre"]
    # force some samples outside [Smin, Smax] in every group, including group 0
    for g in range(G):
        lo = rng.uniform(size=(ny, nx)) < 0.02
        hi = rng.uniform(size=(ny, nx)) < 0.02
        S[g] = np.where(lo, lin["Smin"] - rng.integers(1, 400, size=(ny, nx)), S[g])
        S[g] = np.where(hi, lin["Smax"] + rng.integers(1, 400, size=(ny, nx)), S[g])
    dq = lin["dq"].copy()
    dq |= cal["mask"]["dq"] & np.uint32(pixel.REFERENCE_PIXEL)
    return {
        "S": S, "coefs": lin["data"], "Smin": lin["Smin"], "Smax": lin["Smax"], "Sref": lin["Sref"],
        "lin_dq": dq, "groupdq": ramp["groupdq"],
    }


def ipc_case(seed, gain_dtype, ipc_dtype, G=3):
    cal = small_cal(synth.READ_PATTERN_6, 3, seed, gain_dtype=gain_dtype, ipc_dtype=ipc_dtype)
    ny, nx = SMALL
    rng = np.random.default_rng(seed + 5)
    cube = (rng.uniform(-50, 3000, size=(G, ny, nx)) + 20000 * (rng.uniform(size=(G, ny, nx)) < 0.01)).astype(np.float32)
    return {"cube": cube, "K": cal["ipc4d"]["data"], "gain": cal["gain"]["data"]}


def rampfit_case(read_pattern, seed, exclude_first=True, gain_dtype=np.float32, cr_frac=0.03):
    """A linearised, IPC-corrected-looking f32 cube + DQ arrays exercising every branch of ramp_fit."""
    cal = small_cal(read_pattern, 3, seed, gain_dtype=gain_dtype)
    ramp = synth.make_ramp(cal, read_pattern=read_pattern, seed=seed + 1, cr_frac=cr_frac,
                           exclude_first=exclude_first)
    G, ny, nx = ramp["data"].shape
    rng = np.random.default_rng(seed + 3)
    t = synth.group_times(read_pattern)
    # linear-ish cube in DN_lin: rate * t + noise + CR steps, built directly (no need to invert anything)
    rate = ramp["rate"].astype(np.float64) + cal["dark"]["dark_slope"]
    rate[:, : nx // 4] = 10.0 ** rng.uniform(-1, 3.3, size=(ny, nx // 4))  # wide range of fluxes
    cube = rate[None] * t[:, None, None] + 300.0
    cube += cal["read"]["data"][None] * rng.normal(size=(G, ny, nx)) / np.sqrt(
        np.array([len(r) for r in read_pattern]))[:, None, None]
    cr = rng.uniform(size=(ny, nx)) < cr_frac
    cr_g = rng.integers(1, G, size=(ny, nx))
    cr_amp = np.where(cr, 10.0 ** rng.uniform(0.5, 3.5, size=(ny, nx)), 0.0)
    for g in range(G):
        cube[g] += np.where(cr_g <= g, cr_amp, 0.0)
    cube = cube.astype(np.float32)
    gdq = np.zeros((G, ny, nx), dtype=np.uint8)
    if exclude_first:
        gdq[0] |= 1
    # rows 4.. : pixel (r, c) with c < G+1 first saturates at group c (c == G: never) -- every index covered
    for c in range(G + 1):
        for g in range(c, G):
            gdq[g, 6::3, 8 + c] |= 2
    # a few fully DO_NOT_USE pixels, a pixel with a pre-existing JUMP_DET, random saturation elsewhere
    gdq[:, 10, 30] |= 1
    gdq[:, 11, 31] |= 1
    gdq[3, 12, 32] |= 4
    sat_from = rng.integers(1, 3 * G, size=(ny, nx))
    for g in range(G):
        gdq[g, :, nx // 2 :] |= np.where(sat_from[:, nx // 2 :] <= g, np.uint8(2), np.uint8(0))
    pdq = cal["mask"]["dq"].copy()
    pdq[20, 20] |= np.uint32(pixel.HOT)
    return {
        "data": cube, "groupdq": gdq, "pixeldq": pdq, "gain": cal["gain"]["data"],
        "read": cal["read"]["data"],
    }


def flat_case(seed, gain_dtype=np.float32):
    cal = small_cal(synth.READ_PATTERN_6, 3, seed, gain_dtype=gain_dtype)
    flat = cal["flat"]["data"].copy()
    rng = np.random.default_rng(seed + 9)
    ny, nx = flat.shape
    flat[10, 10] = 0.05
    flat[11, 12] = 12.0
    flat[12, 14] = -1.0
    flat *= (1 + 0.02 * rng.normal(size=(ny, nx))).astype(np.float32)
    gain = cal["gain"]["data"].copy()
    gain[15, 15] = 0.05
    gain[16, 17] = 0.0
    pdq = cal["mask"]["dq"].copy()
    return {"flat": flat, "gain": gain, "K": cal["ipc4d"]["data"], "pixeldq": pdq}


# ---- full-frame reference-pixel inputs, integer draws only (regenerated, not stored) ----

def refpix_fullframe_inputs(seed, nside=4096):
    rng = np.random.default_rng(seed)
    cw = 128
    rows = np.arange(nside)
    # exactly representable f32 values: multiples of 1/8 well below 2^24/8
    dark = (13000 + rng.integers(0, 1600, size=(nside, nside)) / 8.0).astype(np.float32)
    rown = rng.integers(-24, 25, size=(nside, 1))  # common-mode row offset (DN)
    data = (dark + rown + rng.integers(-40, 41, size=(nside, nside))
            + ((rows[:, None] * 3) // 512) + 5 * (np.arange(nside)[None, :] // cw % 3)).astype(np.float32)
    data[4:-4, 4:-4] += rng.integers(0, 2000, size=(nside - 8, nside - 8)).astype(np.float32)
    med = (29000 + rng.integers(0, 64, size=(nside, cw)) / 4.0).astype(np.float32)
    amp33 = (29000 + rown + rng.integers(-16, 17, size=(nside, cw)) + rng.integers(0, 8, size=(nside, 1))).astype(np.uint16)
    std = (4 + rng.integers(0, 3, size=(nside, cw)) / 2.0).astype(np.float32)
    return {"data": data, "dark": dark, "amp33": amp33, "med": med, "std": std,
            "M_PINK": 0.8, "RU_PINK": 1.0, "C_PINK": 0.8}


# ---- many-realisations statistics (validation_tests/many_realizations.py): full 4096 x 4096 frames, regenerated ----

HARNESS_ROWS = (0, 3, 4, 5, 2047, 4091, 4092, 4095)   # sampled rows stored in the fixture (every 29th column)
HARNESS_EXPTIME = 139.8
HARNESS_CASES = {"harness_a": dict(seed=61, nrun=5, scanum=3), "harness_b": dict(seed=62, nrun=4, scanum=4)}


def harness_truth(seed, nside=4096, nb=4):
    """The 'truth' image of the scene in electrons (what the reference reads from its input FITS file)."""
    rng = np.random.default_rng(seed)
    na = nside - 2 * nb
    y, x = np.mgrid[0:na, 0:na]
    return ((0.6 + 0.3 * x / na + 0.2 * (y / na) ** 2) * HARNESS_EXPTIME * 1.458
            + 40.0 * (rng.random((na, na)) < 1e-4)).astype(np.float32)


def harness_realisation(seed, j, ideal_act, nside=4096, nb=4):
    """Realisation j: the two L1 groups the statistics use (u16) and the L2 planes data / err / dq (active region).
    ``ideal_act``: the oriented ideal slope in the active region.  Values are quantised so that realisations tie."""
    rng = np.random.default_rng([seed, j])
    na = nside - 2 * nb
    first = rng.integers(5000, 5200, size=(nside, nside), dtype=np.uint16)
    last = (first + rng.integers(0, 12, size=(nside, nside), dtype=np.uint16) * 16).astype(np.uint16)
    last[7, 9] = first[7, 9] - 3   # a negative difference
    data = (ideal_act + np.round(rng.standard_normal((na, na)) * 0.05 * 64) / 64).astype(np.float32)
    err = (0.04 + rng.integers(0, 8, size=(na, na)) / 256.0).astype(np.float32)
    dq = np.zeros((na, na), np.uint32)
    for bit in (0, 2, 3, 10, 12, 20, 1, 7):     # copied, plus-grown, 5x5, 3x3 and unmasked flags
        dq[rng.random((na, na)) < 0.002] |= np.uint32(1 << bit)
    dq[100:108, 200:216] |= np.uint32(1)        # never usable: N = 0 -> sentinel
    dq[0, 0] |= np.uint32(1 << 3)               # growth clipped by the edge of the active region
    if j % 2 == 1:
        dq[300:340, 10:50] |= np.uint32(1 << 11)  # usable in some realisations only
    data[104, 208] = np.nan                     # masked everywhere: must not reach the sums
    if j == 1:
        err[500, 600] = np.nan                  # np.median of a column holding a NaN is NaN
        data[2000:2004, 2000:2004] = -0.0
    return {"l1_first": first, "l1_last": last, "data": data, "err": err, "dq": dq}


def harness_orient(ideal_big, scanum):
    return ideal_big[:, ::-1] if scanum % 3 == 0 else ideal_big[::-1, :]


# ---- inverse linearity / IL.apply (simulation side): frames whose active height obeys the reference's border rule ----

IL_SHAPE = (64, 80)   # active (56, 72): (8192 - 56 // 2) % 16 == 4 == the border width


def il_case(seed, p_order, ipc_dtype, counts_dtype):
    cal = small_cal(synth.READ_PATTERN_8, p_order, seed, ipc_dtype=ipc_dtype, shape=IL_SHAPE)
    rng = np.random.default_rng(seed + 5)
    na = (IL_SHAPE[0] - 8, IL_SHAPE[1] - 8)
    counts = rng.uniform(-500.0, 6.0e4, size=na)           # DN_lin scale; electrons when divided by the gain
    counts[rng.random(na) < 0.02] = 3.0e5                  # far beyond the well: z runs into +1
    counts[rng.random(na) < 0.02] = -4.0e4                 # far below: z runs into -1
    counts[3, 5] = np.nan
    lin = cal["linearitylegendre"]
    return {"counts": counts.astype(counts_dtype), "coefs": lin["data"], "Smin": lin["Smin"], "Smax": lin["Smax"],
            "Sref": lin["Sref"], "lin_dq": lin["dq"], "gain": cal["gain"]["data"], "K": cal["ipc4d"]["data"],
            "start_e": rng.uniform(0, 50, size=na).astype(np.float32)}


IL_CASES = {"il_p8_f64": (81, 8, np.float32, np.float64), "il_p3_f32_k64": (82, 3, np.float64, np.float32),
            "il_p10_f64_k64": (83, 10, np.float64, np.float64)}
