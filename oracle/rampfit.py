"""Oracle: MA-table meta, ramp-fit weights, slope + jump detection, saturation-truncated
refits and flag propagation.  Test infrastructure only.

Follows ``gen_cal_image.py:123-145`` (meta), ``utils/fitting.py:20-86``
(``construct_weights``), ``:89-255`` (``jump_detect``) and ``:258-355`` (``ramp_fit``).

dtype recipe of one fit over groups [0, g) with weights K (f32), ``start`` = 1 if exclude_first:
  slope   = sum_t K[t] * (d[t] - d[1])            f32, sequential t from 0, product rounded then added
  coef    = sum_i K_i^2 tau_i + 2 sum_{j<i} K_i K_j tbar_j     f32 scalar (Python float += f32 -> f32)
  dv      = max(slope / clip(gain, 1e-4, 1e4), 0)   dtype = promote(f32, gain.dtype)
  ep      = f32(sqrt(max(coef * dv, 0)))
  er      = read * f32(sqrt(sum(K^2 / N)))         f32
  sth     = SA + (SB-SA) * ( f64(log_f32(clip(slope, IA, IB) / IA)) / log(IB/IA) )    f64
  for i in start..g-2, di in {1,2} (di=1 only for the last i, or when g-1-start == 2):
     delta = (d[i+di] - d[i]) / (tbar[i+di]-tbar[i]) - slope                    f32
     w     = e_{i+di}/dt - e_i/dt - K                                             f64 vector
     var   = sum_a [ w_a^2 * (dv*tau_a + read^2/N_a)  +  sum_{b<a} ((2 w_a) w_b) * dv * tbar_b ]   f64,
             accumulated in exactly this order; the inner bracket (dv*tau_a + read^2/N_a) is
             evaluated in promote(dv, f32) BEFORE the f64 multiply
     flag JUMP_DET on group i (active region only) where  delta / f32(sqrt(var)) > sth
"""

import numpy as np

DO_NOT_USE = np.uint32(1)
SATURATED = np.uint32(2)
JUMP_DET = np.uint32(4)
REFERENCE_PIXEL = np.uint32(2**31)

DEFAULT_JUMP_PARS = {"SthreshA": 5.5, "SthreshB": 4.5, "IthreshA": 1.0, "IthreshB": 1000.0}


def ma_table_meta(read_pattern, frame_time):
    """N_i, tbar_i, tau_i of Casertano+22 from a read pattern (``gen_cal_image.py:129-140``).

    Computed in f64, stored f32 (tbar, tau) / int16 (N).
    """
    ngrp = len(read_pattern)
    meta = {
        "frame_time": frame_time,
        "read_pattern": read_pattern,
        "ngrp": ngrp,
        "tbar": np.zeros(ngrp, dtype=np.float32),
        "tau": np.zeros(ngrp, dtype=np.float32),
        "N": np.zeros(ngrp, dtype=np.int16),
    }
    for i, reads in enumerate(read_pattern):
        n = len(reads)
        t0 = reads[0]
        meta["N"][i] = n
        meta["tbar"][i] = (t0 + (n - 1) / 2.0) * frame_time
        meta["tau"][i] = (t0 + (n - 1) * (2 * n - 1) / (6.0 * n)) * frame_time
    return meta


def construct_weights(u, meta, exclude_first=True):
    """Fixed-u optimal weights (Casertano+22 without the adaptive step); f64 LAPACK inverse -> f32."""
    start = 1 if exclude_first else 0
    n = meta["ngrp"] - start
    tbar = meta["tbar"][start:].astype(np.float64)
    tau = meta["tau"][start:].astype(np.float64)
    cov = np.zeros((n, n))
    for i in range(n):
        cov[i, i] = 1.0 / meta["N"][start + i] + u * tau[i]
        for j in range(i):
            cov[i, j] = cov[j, i] = u * tbar[j]
    W = np.linalg.inv(cov)
    w_col = np.sum(W, axis=0)
    w_t = W @ tbar
    f0 = np.sum(W)
    f1 = np.sum(w_t)
    f2 = np.dot(tbar, w_t)
    K = np.zeros(meta["ngrp"])
    K[start:] = (f0 * w_t - f1 * w_col) / (f0 * f2 - f1**2)
    return K.astype(np.float32)


def two_point_weights(meta, g, start):
    """Weights of a ramp truncated to groups [0, g): last minus first usable (``fitting.py:165-169``)."""
    K = np.zeros(g, dtype=np.float32)
    K[-1] = 1.0 / (meta["tbar"][g - 1] - meta["tbar"][start])
    K[start] = -K[-1]
    return K


def poisson_coef(K, meta, g, start):
    """f32 scalar multiplying dvardt in the Poisson variance of the slope (``fitting.py:196-200``)."""
    coef = 0.0
    for i in range(start, g):
        coef += K[i] ** 2 * meta["tau"][i]
        for j in range(start, i):
            coef += 2.0 * K[i] * K[j] * meta["tbar"][j]
    return coef


def read_factor(K, meta, g):
    """f32 scalar multiplying the single-read noise (``fitting.py:209``)."""
    return np.sqrt(np.sum(K**2 / np.array(meta["N"][:g])))


def difference_list(g, start):
    """(i, di) pairs tested by the jump detector, in order (``fitting.py:225-229``)."""
    out = []
    for i in range(start, g - 1):
        dimax = 1 if (i == g - 2 or g - 1 - start == 2) else 2
        for di in range(1, dimax + 1):
            out.append((i, di))
    return out


def fit_and_flag(data, flags, gain, read, meta, nborder, exclude_first=True, truncate=None, jump_pars=None):
    """One pass of slope fit + jump flagging (``jump_detect``).  ``flags`` (G,ny,nx) is OR-ed in place.

    Returns slope, err_read, err_poisson (f32 planes) and the significance cube (f32).
    """
    pars = dict(DEFAULT_JUMP_PARS)
    if jump_pars:
        pars.update({k: float(v) for k, v in jump_pars.items() if k in pars})
    SA, SB, IA, IB = pars["SthreshA"], pars["SthreshB"], pars["IthreshA"], pars["IthreshB"]

    start = 1 if exclude_first else 0
    if truncate is None:
        g = meta["ngrp"]
        K = meta["K"]
    else:
        g = truncate
        K = two_point_weights(meta, g, start)
    ny, nx = data.shape[1:]
    tbar, tau, N = meta["tbar"], meta["tau"], meta["N"]

    slope = np.einsum("t,tij->ij", K, data[:g] - data[1][None]).astype(np.float32)

    coef = poisson_coef(K, meta, g, start)
    dvardt = np.clip(slope / np.clip(gain, 1e-4, 1e4), 0.0, None)
    err_poisson = np.sqrt(np.clip(coef * dvardt, 0, None)).astype(np.float32)
    sig2read = read**2
    err_read = (read * read_factor(K, meta, g)).astype(np.float32)

    x = np.clip(slope, IA, IB)
    x = np.log(x / IA) / np.log(IB / IA)
    sthresh = SA + (SB - SA) * x

    pairs = difference_list(g, start)
    smap = np.zeros((2 * (g - start) - 3, ny, nx), dtype=np.float32)
    act = (slice(nborder, ny - nborder), slice(nborder, nx - nborder))
    for sl, (i, di) in enumerate(pairs):
        dt = tbar[i + di] - tbar[i]  # f32 scalar
        delta = (data[i + di] - data[i]) / dt - slope
        w = np.zeros(g)
        w[i + di] = 1.0 / dt
        w[i] = -1.0 / dt
        w -= K
        var = np.zeros((ny, nx))
        for a in range(g):
            var += w[a] ** 2 * (dvardt * tau[a] + sig2read / np.array(N[a]))
            for b in range(a):
                var += 2 * w[a] * w[b] * dvardt * tbar[b]
        smap[sl] = delta / np.sqrt(var).astype(np.float32)
        hit = smap[sl][act] > sthresh[act]
        flags[i][act] |= np.where(hit, JUMP_DET, np.uint32(0)).astype(flags.dtype)
    return slope, err_read, err_poisson, smap


def ramp_fit(data, rdq, pdq, gain, read, meta, exclude_first=True, jump_pars=None):
    """Full fit + refits truncated at the first saturated group + flag propagation (``ramp_fit``).

    ``rdq`` (u8 G,ny,nx) and ``pdq`` (u32 ny,nx) are updated in place.
    """
    G = meta["ngrp"]
    nb = meta["nborder"]
    start = 1 if exclude_first else 0
    sat8 = rdq.dtype.type(SATURATED)

    scratch = np.zeros_like(rdq)
    slope, err_read, err_poisson, _ = fit_and_flag(
        data, scratch, gain, read, meta, nb, exclude_first, None, jump_pars
    )
    never_sat = (rdq[-1] & sat8) == 0
    rdq |= np.where(never_sat[None], scratch, 0).astype(rdq.dtype)

    for iend in range(G - 1, 2 + start, -1):
        first_sat_here = ((rdq[iend] & ~rdq[iend - 1]) & sat8) != 0
        scratch[...] = 0
        s_, er_, ep_, _ = fit_and_flag(data, scratch, gain, read, meta, nb, exclude_first, iend, jump_pars)
        slope = np.where(first_sat_here, s_, slope)
        err_read = np.where(first_sat_here, er_, err_read)
        err_poisson = np.where(first_sat_here, ep_, err_poisson)
        rdq |= np.where(first_sat_here[None], scratch, 0).astype(rdq.dtype)

    rdq32 = rdq.astype(np.uint32)
    pdq2 = np.zeros_like(pdq)
    # flags of groups that are not saturated, minus DO_NOT_USE
    pdq2 |= np.bitwise_or.reduce(np.where((rdq32 & SATURATED) == 0, rdq32, 0), axis=0) & ~DO_NOT_USE
    # DO_NOT_USE only if every group carries it
    pdq2 |= np.where(np.bitwise_and.reduce((rdq32 & DO_NOT_USE) != 0, axis=0), DO_NOT_USE, 0).astype(np.uint32)
    # saturated too early to fit
    pdq2 |= np.where((rdq32[1 + start] & SATURATED) != 0, DO_NOT_USE, 0).astype(np.uint32)
    # any saturation at all
    pdq2 |= np.bitwise_or.reduce(rdq32 & SATURATED, axis=0)
    pdq |= np.where((pdq & REFERENCE_PIXEL) == 0, pdq2, 0).astype(np.uint32)
    return slope, err_read, err_poisson
