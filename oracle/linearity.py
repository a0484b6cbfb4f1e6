"""Oracle: Legendre linearity correction.  Test infrastructure only.

Follows ``utils/ipc_linearity.py:192-231`` (``_lin``) and ``:276-344`` (``multilin``).

All arithmetic is in the dtype of the inputs (f32 on the path), one rounding per
operation, in this order (Appendix A.3 of SURVEY.md):
  t = S - Smin ; t = 2*t ; d = Smax - Smin ; t = t/d ; z = -1 + t
  (group 0 with do_not_flag_first: z clipped to [-1, 1], NaN stays NaN)
  ex = |z| > 1
  phi = c0 ; pp = 1 ; p = z
  for L = 1..P:
     e   = sign(z)**L * (1 + f32(L(L+1)/2) * (|z| - 1))
     phi = phi + cL * (ex ? e : p)
     pn  = (f32((2L+1)/(L+1)) * z) * p - f32(L/(L+1)) * pp ; pp = p ; p = pn
  phi = ((lin.dq & (NO_LIN_CORR|REFERENCE_PIXEL)) == 0) ? phi : S - Sref     [running dq]
  dq |= NO_LIN_CORR where ex and attempt_corr   (not for group 0 when do_not_flag_first)
"""

import numpy as np

NO_LIN_CORR = np.uint32(2**20)
REFERENCE_PIXEL = np.uint32(2**31)


def legendre_series(z, coefs, linextrap=True):
    """phi(z) = sum_L coefs[L] P_L(z), linearly extrapolated outside |z|<=1.  (``_lin``)."""
    absz = np.abs(z)
    ex = absz > 1
    sgn = np.sign(z)
    excess = absz - 1
    phi = np.array(coefs[0], copy=True)  # accumulator keeps the coefficient dtype (in-place adds)
    p_prev = np.ones_like(phi)
    p_cur = np.array(z, copy=True)
    sgn_pow = np.ones_like(z)
    for L in range(1, coefs.shape[0]):
        sgn_pow = sgn_pow * sgn  # = sign(z)**L exactly (values in {-1, 0, 1, nan})
        if linextrap:
            lin = sgn_pow * (1 + (L * (L + 1) / 2.0) * excess)
            phi += coefs[L] * np.where(ex, lin, p_cur)
        else:
            phi += coefs[L] * p_cur
        p_next = (2 * L + 1) / (L + 1) * z * p_cur - L / (L + 1) * p_prev
        p_prev = p_cur
        p_cur = p_next
    return phi, ex


def multilin(S, coefs, Smin, Smax, Sref, lin_dq, do_not_flag_first=True, attempt_corr=None):
    """Linearise a cube S (G,ny,nx).  Returns (phi f32 (G,ny,nx), dq u32 (ny,nx))."""
    ngrp = S.shape[0]
    phi = np.zeros(S.shape, dtype=np.float32)
    dq = np.array(lin_dq, dtype=np.uint32, copy=True)
    bad_bits = np.uint32(NO_LIN_CORR | REFERENCE_PIXEL)
    span = Smax - Smin
    for j in range(ngrp):
        z = -1 + 2 * (S[j] - Smin) / span
        first = j == 0 and do_not_flag_first
        if first:
            z = np.clip(z, -1, 1)
        val, ex = legendre_series(z, coefs)
        phi[j] = np.where((dq & bad_bits) == 0, val, S[j] - Sref)
        if not first:
            hit = ex if attempt_corr is None else np.logical_and(ex, attempt_corr[j] != 0)
            dq |= np.where(hit, NO_LIN_CORR, np.uint32(0)).astype(np.uint32)
    return phi, dq


def attempt_corr_from_groupdq(groupdq):
    """``attempt_corr = ~rdq & pixel.SATURATED`` (``gen_cal_image.py:585-586``): nonzero where the
    group is NOT saturated."""
    return (~groupdq) & np.uint8(2)


def invlinearity(Slin, coefs, Smin, Smax):
    """``ipc_linearity.invlinearity`` (:347-394) on arrays already cut to the block: 24 bisection steps on z with the series
    evaluated without linear extrapolation; returns (S, exflag of the last evaluation)."""
    z = np.zeros_like(Slin)
    for j in range(1, 25):
        phi, exflag = legendre_series(z, coefs, linextrap=False)
        z += np.where(phi < Slin, 1 / 2**j, -1 / 2**j)
    return Smin + (Smax - Smin) / 2.0 * (1 + z), exflag


def il_apply(counts, K, gain, coefs, Smin, Smax, Sref, start_e=0.0, electrons=False, electrons_out=False):
    """``ipc_linearity.IL.apply`` (:459-513): counts (active region) -> DN_raw (or electrons).  ``K`` None skips the IPC;
    full-frame linearity arrays, cut by the reference's border rule ``nb = (8192 - ny//2) % 16``."""
    from . import ipc
    conv = ipc.ipc_fwd(counts + start_e, K) if K is not None else counts + start_e
    nyc, nxc = counts.shape
    g = gain
    if g.shape[0] > nyc:
        b = (g.shape[0] - nyc) // 2
        g = g[b:-b, b:-b]
    nb = (8192 - nyc // 2) % 16
    sl = (slice(nb, nb + nyc), slice(nb, nb + nxc))
    S, _ = invlinearity(conv / (g if electrons else 1.0), coefs[(slice(None),) + sl], Smin[sl], Smax[sl])
    return g * (S - Sref[sl]) if electrons_out else S
