"""Host-side scalars of the ramp fit: MA-table meta, weights, per-variant constants, plan descriptor.

These are the microsecond-scale host computations of the reference (SURVEY.md 8a rows A1, A8 and
the scalar part of A9); they stay in Python/numpy so that every scalar is produced by the same
numpy expression the reference evaluates (dtype promotion included), then cross the C-ABI in a
``rip_plan_desc``.

Reference: ``L1_to_L2/gen_cal_image.py:123-145`` (meta), ``utils/fitting.py:20-86``
(``construct_weights``), ``:165-169`` (two-point weights), ``:196-200`` (Poisson coefficient),
``:209`` (read-noise factor), ``:172-184`` (thresholds).
"""

import numpy as np

from . import _native

DEFAULT_RAMP_OPT_PARS = {"slope": 0.4, "gain": 1.8, "sigma_read": 6.5}  # gen_cal_image.py:435
DEFAULT_JUMP_PARS = {"SthreshA": 5.5, "SthreshB": 4.5, "IthreshA": 1.0, "IthreshB": 1000.0}  # fitting.py:172-175


def exposure_meta(read_pattern, frame_time):
    """``meta`` of ``initializationstep``: ngrp, N (int16), tbar, tau (f32; Casertano+22)."""
    ngrp = len(read_pattern)
    N = np.zeros(ngrp, dtype=np.int16)
    tbar = np.zeros(ngrp, dtype=np.float32)
    tau = np.zeros(ngrp, dtype=np.float32)
    for i, grp in enumerate(read_pattern):
        n, first = len(grp), grp[0]
        N[i] = n
        tbar[i] = (first + (n - 1) / 2.0) * frame_time  # f64 -> f32 on store
        tau[i] = (first + (n - 1) * (2 * n - 1) / (6.0 * n)) * frame_time
    return {"frame_time": frame_time, "read_pattern": read_pattern, "ngrp": ngrp, "N": N, "tbar": tbar, "tau": tau}


def construct_weights(u, meta, exclude_first=True):
    """Weight vector K (f32, length ngrp) for slope = sum_i K_i R_i, optimal for the Poisson/read ratio ``u``.

    Same surface as ``fitting.construct_weights``: generalized least squares with the covariance of
    Casertano+22 at fixed u; weights sum to zero.  f64 linear algebra, cast to f32.
    """
    s = 1 if exclude_first else 0
    n = meta["ngrp"] - s
    tb = np.asarray(meta["tbar"][s:], dtype=np.float64)
    ta = np.asarray(meta["tau"][s:], dtype=np.float64)
    cov = np.zeros((n, n))
    for i in range(n):
        cov[i, i] = 1.0 / meta["N"][s + i] + u * ta[i]
        for j in range(i):
            cov[i, j] = cov[j, i] = u * tb[j]
    W = np.linalg.inv(cov)
    colsum = np.sum(W, axis=0)
    Wt = W @ tb
    F0, F1, F2 = np.sum(W), np.sum(Wt), np.dot(tb, Wt)
    out = np.zeros(meta["ngrp"])
    out[s:] = (F0 * Wt - F1 * colsum) / (F0 * F2 - F1**2)
    return out.astype(np.float32)


def ramp_opt_u(ramp_opt_pars=None):
    p = ramp_opt_pars or DEFAULT_RAMP_OPT_PARS
    return float(p["slope"]) / float(p["gain"]) / float(p["sigma_read"]) ** 2  # gen_cal_image.py:438


def _variant_weights(meta, K, g, start, full):
    if full:
        return np.asarray(K, dtype=np.float32)[:g]
    k = np.zeros(g, dtype=np.float32)
    k[-1] = 1.0 / (meta["tbar"][g - 1] - meta["tbar"][start])  # f32 division
    k[start] = -k[-1]
    return k


def _poisson_coef(k, meta, g, start):
    # accumulates in f32: a Python float += np.float32 yields np.float32 under NEP 50
    c = 0.0
    for i in range(start, g):
        c += k[i] ** 2 * meta["tau"][i]
        for j in range(start, i):
            c += 2.0 * k[i] * k[j] * meta["tbar"][j]
    return c


def _read_factor(k, meta, g):
    return np.sqrt(np.sum(k**2 / np.array(meta["N"][:g])))


def plan_desc(meta, K, exclude_first=True, do_not_flag_first=True, jump_pars=None):
    """Fill a ``rip_plan_desc`` from meta + weights (+ optional ``jump_detect_pars``)."""
    G = int(meta["ngrp"])
    if G > _native.RIP_MAX_GROUPS:
        raise ValueError("too many groups")
    start = 1 if exclude_first else 0
    d = _native.PlanDesc()
    d.ngrp, d.exclude_first, d.do_not_flag_first = G, int(bool(exclude_first)), int(bool(do_not_flag_first))
    for i in range(G):
        d.tbar[i] = float(meta["tbar"][i])
        d.tau[i] = float(meta["tau"][i])
        d.nreads[i] = int(meta["N"][i])
        d.K[i] = float(K[i])
    ends = [G] + list(range(G - 1, 2 + start, -1))  # full ramp, then truncations (fitting.py:326)
    d.nvariants = len(ends)
    for v, g in enumerate(ends):
        k = _variant_weights(meta, K, g, start, v == 0)
        d.variant_g[v] = g
        d.variant_coef[v] = float(np.float32(_poisson_coef(k, meta, g, start)))
        d.variant_rfac[v] = float(np.float32(_read_factor(k, meta, g)))
    jp = dict(DEFAULT_JUMP_PARS)
    if jump_pars:
        for key in jp:
            if key in jump_pars:
                jp[key] = float(jump_pars[key])
    d.sthresh_a, d.sthresh_b, d.ithresh_a, d.ithresh_b = jp["SthreshA"], jp["SthreshB"], jp["IthreshA"], jp["IthreshB"]
    return d


def refout_slope(read_tree):
    """Scalar weight of the reference-output row correction (``gen_cal_image.py:542-553``).

    ``read_tree`` is the ``roman`` branch of the read-noise file.  Returns None without ``amp33``.
    """
    a = read_tree.get("amp33")
    if a is None:
        return None
    cvar = read_tree["anc"]["C_PINK"] ** 2
    return a["M_PINK"] * cvar / (a["M_PINK"] ** 2 * cvar + a["RU_PINK"] ** 2 + np.median(a["std"]) ** 2 / 128 / np.log(4096))
