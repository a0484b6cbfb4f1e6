"""Synthetic CALDIR arrays and Level-1 ramps (host side, numpy).

There is no network on the build or GPU machines, so benchmarks and tests run on synthetic
inputs.  The calibration arrays mirror the reference's own synthetic CALDIR writer
(``tests/romanimpreprocess/test_workflow.py:117-332``, ``gencal``) in content and dtype; the
ramp follows SURVEY.md section 8(d): sky + Gaussian sources through IPC and an (approximately)
inverted linearity curve, read noise, row-correlated noise seen by the reference pixels and the
reference output (amp33), cosmic-ray steps and saturation.

Everything is keyed on seeds through ``numpy.random.default_rng`` and works for any frame
(ny, nx) with nx a multiple of 128 (the channel width) so that small frames can be used in tests.
"""

import numpy as np

from . import pars
from .dqflags import pixel

# production 8-group MA table (README.rst:61 of the reference) and the survey's 16-group table
READ_PATTERN_8 = [[0], [1], [2, 3], list(range(4, 10)), list(range(10, 26)), list(range(26, 32)), [32, 33], [34]]
_B16 = [0, 1, 2, 3, 4, 6, 8, 10, 13, 16, 19, 22, 25, 28, 31, 34, 35]
READ_PATTERN_16 = [list(range(_B16[i], _B16[i + 1])) for i in range(16)]
# the 6-group pattern of the reference's workflow test (test_workflow.py:29)
READ_PATTERN_6 = [[0], [1, 2], [3, 4, 5], [6, 7, 8, 9, 10], [11, 12], [13]]
FRAME_TIME = 3.04


def group_times(read_pattern, frame_time=FRAME_TIME):
    return np.array([frame_time * np.mean(np.array(r)) for r in read_pattern])


def make_caldir(ny=pars.nside, nx=pars.nside, read_pattern=None, frame_time=FRAME_TIME, p_order=8, seed=1000,
                gain_dtype=np.float32, ipc_dtype=np.float32, nb=pars.nborder, with_biascorr=True,
                bias_amplitude=0.0, bad_lin_frac=0.0, high_order_scale=0.05):
    """Dict of dicts of arrays with the layout of the ``roman`` branch of every CALDIR file."""
    rp = READ_PATTERN_8 if read_pattern is None else read_pattern
    rng = np.random.default_rng(seed)
    G = len(rp)
    t = group_times(rp, frame_time)
    y, x = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    nya, nxa = ny - 2 * nb, nx - 2 * nb
    cal = {}

    # dark: per-group frames = bias pattern + dark_slope * t ; dark_slope log-normal, zero on the border
    dark_slope = 0.005 * 10.0 ** rng.normal(size=(ny, nx))
    for sl in ((slice(None), slice(0, nb)), (slice(None), slice(nx - nb, nx)), (slice(0, nb), slice(None)),
               (slice(ny - nb, ny), slice(None))):
        dark_slope[sl] = 0.0
    bias = 13000 + 200 * np.cos(2.0 * np.pi * x / 256.0) + 100 * np.sin(2.0 * np.pi * y / 256.0) ** 3
    cal["dark"] = {
        "data": np.clip((bias[None] + dark_slope[None] * t[:, None, None]).astype(np.float32), 0.0, 65535.0),
        "dq": np.zeros((ny, nx), dtype=np.uint32),
        "dark_slope": dark_slope.astype(np.float32),
        "dark_slope_err": np.zeros((ny, nx), dtype=np.float32),
    }

    cal["gain"] = {
        "data": np.clip(1.5 + 0.03 * rng.normal(size=(ny, nx)), 1.4, 1.6).astype(gain_dtype),
        "dq": np.zeros((ny, nx), dtype=np.uint32),
    }

    # ipc4d: nearest neighbours 1.5% / 1.3%, diagonals 0.2%, no coupling across the frame edge
    K = np.zeros((3, 3, nya, nxa), dtype=ipc_dtype)
    K[0, 1] = K[2, 1] = 0.015
    K[1, 0] = K[1, 2] = 0.013
    K[0, 0] = K[2, 2] = K[0, 2] = K[2, 0] = 0.002
    K *= (1.0 + 0.05 * rng.normal(size=(1, 1, nya, nxa))).astype(ipc_dtype)  # per-pixel variation
    K[0, :, 0, :] = 0.0
    K[:, 0, :, 0] = 0.0
    K[-1, :, -1, :] = 0.0
    K[:, -1, :, -1] = 0.0
    K[1, 1] = 0.0
    K[1, 1] = 1.0 - np.sum(K, axis=(0, 1))
    cal["ipc4d"] = {"data": K, "dq": np.zeros((ny, nx), dtype=np.uint32)}

    # linearity: Phi(Sref) = 0, Phi'(Sref) = 1, quadratic Legendre term 20..200 DN, small higher orders
    Smin = np.clip(5000 + 500 * np.cos((x + 3 * y) / 100.0), 0.5, 65534.5).astype(np.float32)
    Smax = np.clip(56000 + 10000 * rng.uniform(size=(ny, nx)), 0.5, 65534.5).astype(np.float32)
    Sref = (Smin + 300 + 100 * (x % 2)).astype(np.float32)
    coefs = np.zeros((p_order + 1, ny, nx), dtype=np.float32)
    coefs[2] = 20 + 180 * rng.uniform(size=(ny, nx))
    for L in range(3, p_order + 1):
        coefs[L] = (high_order_scale * 40.0 / L**2) * rng.normal(size=(ny, nx))
    z = 2 * (Sref.astype(np.float64) - Smin) / (Smax.astype(np.float64) - Smin) - 1
    # value and derivative of sum_{L>=2} c_L P_L at z (f64), then c1, c0 to pin Phi(Sref)=0, dPhi/dS=1
    val, der = _legendre_tail(z, coefs[2:].astype(np.float64), 2)
    c1 = (Smax.astype(np.float64) - Smin) / 2.0 - der
    coefs[1] = c1
    coefs[0] = -(c1 * z) - val
    lin_dq = np.zeros((ny, nx), dtype=np.uint32)
    if bad_lin_frac > 0:
        lin_dq |= np.where(rng.uniform(size=(ny, nx)) < bad_lin_frac, pixel.NO_LIN_CORR, 0).astype(np.uint32)
    cal["linearitylegendre"] = {"data": coefs, "dq": lin_dq, "Smin": Smin, "Smax": Smax, "Sref": Sref}

    # mask: reference border + hot/warm from the dark
    mask = np.zeros((ny, nx), dtype=np.uint32)
    mask[:nb, :] |= pixel.REFERENCE_PIXEL
    mask[-nb:, :] |= pixel.REFERENCE_PIXEL
    mask[:, :nb] |= pixel.REFERENCE_PIXEL
    mask[:, -nb:] |= pixel.REFERENCE_PIXEL
    mask |= np.where(dark_slope > 0.25, np.where(dark_slope > 12.5, pixel.HOT, pixel.WARM), 0).astype(np.uint32)
    cal["mask"] = {"dq": mask}

    pflat = (0.95 + 0.1 * (x / nx - 1) - 0.2 * (y / ny * (1 - y / ny))).astype(np.float32)
    pflat[:nb, :] = 0.0
    pflat[-nb:, :] = 0.0
    pflat[:, :nb] = 0.0
    pflat[:, -nb:] = 0.0
    cal["flat"] = {"data": pflat, "dq": np.zeros((ny, nx), dtype=np.uint32)}

    # read noise + reference-output ("amp33") statistics
    med = np.full((ny, pars.channelwidth), 29000.0, dtype=np.float32)
    std = np.full((ny, pars.channelwidth), 4.0, dtype=np.float32)
    for r in range(0, ny, 256):
        std[r] = 5
        med[r] += 30
        if r + 1 < ny:
            med[r + 1] += 15
    cal["read"] = {
        "anc": {"U_PINK": 0.4, "C_PINK": 0.8},
        "data": (6.0 + 5.0 * rng.uniform(size=(ny, nx))).astype(np.float32),
        "resetnoise": (25.0 + 5.0 * rng.uniform(size=(ny, nx))).astype(np.float32),
        "amp33": {"valid": True, "med": med, "std": std, "M_PINK": 0.8, "RU_PINK": 1.0},
    }

    cal["saturation"] = {
        "data": np.clip(Smax - 50, 1.5, None).astype(np.float32),
        "dq": np.zeros((ny, nx), dtype=np.uint32),
    }

    if with_biascorr:
        bc = np.zeros((G, nya, nxa), dtype=np.float32)
        if bias_amplitude:
            bc += (bias_amplitude * rng.normal(size=(G, nya, nxa))).astype(np.float32)
        cal["biascorr"] = {"data": bc, "t0": float(t[1])}
    return cal


def _legendre_tail(z, coefs, l0):
    """sum_{L>=l0} coefs[L-l0] P_L(z) and its z-derivative, f64."""
    pm, p = np.ones_like(z), z.copy()
    dpm, dp = np.zeros_like(z), np.ones_like(z)
    val = np.zeros_like(z)
    der = np.zeros_like(z)
    L = 1
    top = l0 + coefs.shape[0] - 1
    while L <= top:
        if L >= l0:
            val += coefs[L - l0] * p
            der += coefs[L - l0] * dp
        pn = ((2 * L + 1) * z * p - L * pm) / (L + 1)
        dpn = ((2 * L + 1) * (p + z * dp) - L * dpm) / (L + 1)
        pm, p, dpm, dp = p, pn, dp, dpn
        L += 1
    return val, der


def _phi_f64(S, lin):
    z = 2 * (S - lin["Smin"]) / (lin["Smax"].astype(np.float64) - lin["Smin"]) - 1
    c = lin["data"].astype(np.float64)
    val, der = _legendre_tail(z, c[1:], 1)
    return c[0] + val, der * 2.0 / (lin["Smax"].astype(np.float64) - lin["Smin"])


def _ipc_forward(img_act, K):
    """Charge coupling: out[y,x] = sum in[y-dy,x-dx] K[1+dy,1+dx,y-dy,x-dx] (f64, synthetic-data quality)."""
    K = K.astype(np.float64)
    ny, nx = img_act.shape
    out = img_act * K[1, 1]
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dy == 0 and dx == 0:
                continue
            ys = slice(max(0, -dy), ny - max(0, dy))
            xs = slice(max(0, -dx), nx - max(0, dx))
            yd = slice(max(0, dy), ny - max(0, -dy))
            xd = slice(max(0, dx), nx - max(0, -dx))
            out[yd, xd] += img_act[ys, xs] * K[1 + dy, 1 + dx, ys, xs]
    return out


def make_rate_image(ny, nx, seed, sky=None, nsrc=25, peak=4000.0, nb=pars.nborder):
    """Ideal count-rate image in linearised DN/s: flat sky + ``nsrc`` Gaussians (sigma 2 px)."""
    rng = np.random.default_rng(seed + 7919)
    sky = rng.uniform(0.3, 1.5) if sky is None else sky
    rate = np.full((ny, nx), sky, dtype=np.float64)
    yy, xx = np.arange(ny)[:, None], np.arange(nx)[None, :]
    for j in range(nsrc):
        cx = 10 + (nx - 20) * j / float(nsrc)
        cy = 10 + (ny - 20) * ((13 * j) % nsrc) / float(nsrc)
        amp = peak * j / max(nsrc - 1, 1)
        y0, y1 = max(0, int(cy) - 12), min(ny, int(cy) + 13)
        x0, x1 = max(0, int(cx) - 12), min(nx, int(cx) + 13)
        rate[y0:y1, x0:x1] += amp * np.exp(-0.5 * ((xx[:, x0:x1] - cx) ** 2 + (yy[y0:y1] - cy) ** 2) / 4.0)
    rate[:nb, :] = 0
    rate[-nb:, :] = 0
    rate[:, :nb] = 0
    rate[:, -nb:] = 0
    return rate


def make_ramp(cal, read_pattern=None, frame_time=FRAME_TIME, seed=1, cr_frac=1e-3, rate=None,
              exclude_first=True, nb=pars.nborder, saturation_backup=1):
    """One Level-1 ramp + the DQ arrays as they stand after dq-init and saturation flagging.

    Returns dict(data u16 (G,ny,nx), amp33 u16 (G,ny,128), groupdq u8, pixeldq u32, read_pattern,
    frame_time, rate (the ideal DN_lin/s image)).
    """
    rp = READ_PATTERN_8 if read_pattern is None else read_pattern
    G = len(rp)
    lin = cal["linearitylegendre"]
    ny, nx = lin["Smin"].shape
    rng = np.random.default_rng(seed)
    t = group_times(rp, frame_time)
    nread = np.array([len(r) for r in rp], dtype=np.float64)
    if rate is None:
        rate = make_rate_image(ny, nx, seed, nb=nb)
    act = (slice(nb, ny - nb), slice(nb, nx - nb))
    gain = cal["gain"]["data"].astype(np.float64)
    total_rate = rate + cal["dark"]["dark_slope"].astype(np.float64)

    # cosmic rays: a step of 50..5000 DN at a random group on cr_frac of the pixels
    cr_mask = rng.uniform(size=(ny, nx)) < cr_frac
    cr_grp = rng.integers(2, max(G, 3), size=(ny, nx))
    cr_amp = np.where(cr_mask, 50.0 * 100.0 ** rng.uniform(size=(ny, nx)), 0.0)

    sigma_read = cal["read"]["data"].astype(np.float64)
    sat_level = cal["saturation"]["data"].astype(np.float64)
    data = np.zeros((G, ny, nx), dtype=np.uint16)
    amp33 = np.zeros((G, ny, pars.channelwidth), dtype=np.uint16)
    groupdq = np.zeros((G, ny, nx), dtype=np.uint8)
    a33 = cal["read"]["amp33"]
    S_guess = lin["Sref"].astype(np.float64)
    for g in range(G):
        sig = total_rate * t[g] + np.where(cr_grp <= g, cr_amp, 0.0)  # linearised DN
        conv = sig.copy()
        conv[act] = _ipc_forward(sig[act] * gain[act], cal["ipc4d"]["data"]) / gain[act]
        # invert Phi(S) = conv by Newton from the previous group's solution
        S = S_guess.copy()
        for _ in range(4):
            val, der = _phi_f64(S, lin)
            S = S - (val - conv) / np.where(np.abs(der) > 0.2, der, 1.0)
        S_guess = S
        row_noise = 3.0 * rng.normal(size=(ny, 1))  # common-mode row noise seen by everything
        raw = S + row_noise + sigma_read / np.sqrt(nread[g]) * rng.normal(size=(ny, nx))
        if "biascorr" in cal:
            b = cal["biascorr"]["data"]
            raw[act] += b[b.shape[0] - G + g]
        # reference border: dark frame + noise (no light, no non-linearity)
        ref = cal["dark"]["data"][g].astype(np.float64) + row_noise + sigma_read * rng.normal(size=(ny, nx))
        border = np.ones((ny, nx), dtype=bool)
        border[act] = False
        raw = np.where(border, ref, raw)
        sat_now = (raw >= sat_level) & ~border
        raw = np.where(sat_now, np.minimum(raw, sat_level + 200.0), raw)
        data[g] = np.clip(np.rint(raw), 0, 65535).astype(np.uint16)
        groupdq[g] |= np.where(sat_now, np.uint8(pixel.SATURATED), np.uint8(0))
        amp33[g] = np.clip(
            np.rint(a33["med"] + a33["M_PINK"] * row_noise + a33["std"] * rng.normal(size=(ny, pars.channelwidth))),
            0, 65535).astype(np.uint16)
    # saturation is sticky forward in time and flagged `backup` groups early; group 0 is not checked
    for g in range(1, G):
        groupdq[g] |= groupdq[g - 1] & np.uint8(pixel.SATURATED)
    for _ in range(saturation_backup):
        for g in range(1, G - 1):
            groupdq[g] |= groupdq[g + 1] & np.uint8(pixel.SATURATED)
    groupdq[0] &= ~np.uint8(pixel.SATURATED)
    if exclude_first:
        groupdq[0] |= np.uint8(pixel.DO_NOT_USE)
    pixeldq = np.array(cal["mask"]["dq"], dtype=np.uint32, copy=True)
    return {
        "data": data, "amp33": amp33, "groupdq": groupdq, "pixeldq": pixeldq,
        "read_pattern": rp, "frame_time": frame_time, "rate": rate.astype(np.float32),
    }


# ------------------------------------------------------------------ fast full-frame inputs for benchmarks
def _tile_rows(strip, ny, nb, axis):
    """Full-height array from a strip: strip's first/last nb rows stay the frame border, the strip's
    interior rows are repeated to fill the interior."""
    s = np.moveaxis(strip, axis, 0)
    inner = s[nb:-nb]
    reps = -(-(ny - 2 * nb) // inner.shape[0])
    body = np.concatenate([inner] * reps, axis=0)[: ny - 2 * nb]
    out = np.concatenate([s[:nb], body, s[-nb:]], axis=0)
    return np.ascontiguousarray(np.moveaxis(out, 0, axis))


def _tile_active_rows(strip, nya, axis):
    s = np.moveaxis(strip, axis, 0)
    reps = -(-nya // s.shape[0])
    out = np.concatenate([s] * reps, axis=0)[:nya]
    return np.ascontiguousarray(np.moveaxis(out, 0, axis))


def make_tiled_inputs(ny=pars.nside, nx=pars.nside, read_pattern=None, p_order=8, seed=1, strip_rows=256,
                      gain_dtype=np.float32, ipc_dtype=np.float32, cr_frac=1e-3, nb=pars.nborder, ramp_seed=None):
    """(cal, ramp) at full frame size, synthesised on a (strip_rows + 2*nb)-row strip and repeated down the
    frame (seconds instead of minutes on the host; same per-pixel statistics, same dtypes and shapes).
    ``ramp_seed``: seed of the ramp alone (several ramps on one calibration set)."""
    rp = READ_PATTERN_8 if read_pattern is None else read_pattern
    sy = min(strip_rows + 2 * nb, ny)
    cal_s = make_caldir(sy, nx, read_pattern=rp, p_order=p_order, seed=1000 + seed, gain_dtype=gain_dtype,
                        ipc_dtype=ipc_dtype, nb=nb)
    ramp_s = make_ramp(cal_s, read_pattern=rp, seed=seed if ramp_seed is None else ramp_seed, cr_frac=cr_frac, nb=nb)
    if sy == ny:
        return cal_s, ramp_s
    nya = ny - 2 * nb

    def T(a, axis=0):
        return _tile_rows(a, ny, nb, axis)

    cal = {
        "dark": {"data": T(cal_s["dark"]["data"], 1), "dq": T(cal_s["dark"]["dq"]),
                 "dark_slope": T(cal_s["dark"]["dark_slope"]), "dark_slope_err": T(cal_s["dark"]["dark_slope_err"])},
        "gain": {"data": T(cal_s["gain"]["data"]), "dq": T(cal_s["gain"]["dq"])},
        "ipc4d": {"data": _tile_active_rows(cal_s["ipc4d"]["data"], nya, 2), "dq": T(cal_s["ipc4d"]["dq"])},
        "linearitylegendre": {"data": T(cal_s["linearitylegendre"]["data"], 1),
                              **{k: T(cal_s["linearitylegendre"][k]) for k in ("dq", "Smin", "Smax", "Sref")}},
        "mask": {"dq": T(cal_s["mask"]["dq"])},
        "flat": {"data": T(cal_s["flat"]["data"]), "dq": T(cal_s["flat"]["dq"])},
        "read": {"anc": dict(cal_s["read"]["anc"]), "data": T(cal_s["read"]["data"]),
                 "resetnoise": T(cal_s["read"]["resetnoise"]),
                 "amp33": {**cal_s["read"]["amp33"], "med": T(cal_s["read"]["amp33"]["med"]),
                           "std": T(cal_s["read"]["amp33"]["std"])}},
        "saturation": {"data": T(cal_s["saturation"]["data"]), "dq": T(cal_s["saturation"]["dq"])},
    }
    if "biascorr" in cal_s:
        cal["biascorr"] = {"data": _tile_active_rows(cal_s["biascorr"]["data"], nya, 1), "t0": cal_s["biascorr"]["t0"]}
    ramp = {
        "data": T(ramp_s["data"], 1), "amp33": T(ramp_s["amp33"], 1), "groupdq": T(ramp_s["groupdq"], 1),
        "pixeldq": T(ramp_s["pixeldq"]), "read_pattern": rp, "frame_time": ramp_s["frame_time"],
        "rate": T(ramp_s["rate"]),
    }
    return cal, ramp
