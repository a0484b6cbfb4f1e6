"""Oracle: what happens to the fitted planes after the ramp fit.
Test infrastructure only.

Follows ``gen_cal_image.py:458-475`` (err packaging, trim/embed), ``:188-233`` (dark rate),
``:607-629`` (unpack, flat divide) and ``utils/flatutils.py:20-76`` (``get_flat``).

Steps that live in romancal and are therefore restated from their call sites (PARITY UNPINNED):
  * ``ramp_fit_step._create_image_model`` (``:464``): takes slope/err/var_poisson/dq, trims the
    4-pixel border;  ``_embed_active`` (``:354-375``) puts the border back as zeros.
  * ``dark_current_step.subtract_dark_current`` (``:227``): ``rate[act] -= dark_slope[act]``,
    ``dq[act] |= dark.dq[act]`` (every dark file the reference writes has dq == 0).
"""

import numpy as np

from . import ipc

NO_FLAT_FIELD = np.uint32(2**18)
NO_GAIN_VALUE = np.uint32(2**19)


def embed_active(full_plane, nb):
    """trim to the active region, re-embed with a zero border (f32)."""
    out = np.zeros(full_plane.shape, dtype=np.float32)
    out[nb:-nb, nb:-nb] = full_plane[nb:-nb, nb:-nb]
    return out


def dark_rate_deconvolved(dark_slope, kernel, gain):
    """IPC-deconvolved dark rate (``gen_cal_image.py:217-221``); kernel None -> plain f32 copy."""
    cube = np.array(dark_slope, dtype=np.float32)[None, :, :]
    if kernel is not None:
        ipc.correct_cube(cube, kernel, gain)
    return cube[0]


def get_flat(flat, gain, kernel, nborder, pdq=None, ipc_deconvolve=True):
    """Flat in DN-based units, border = 1, flagged/clipped, IPC-deconvolved (``flatutils.get_flat``).

    ``pdq`` (u32) is updated in place when given.
    """
    ny, nx = flat.shape
    nb = nborder
    act = (slice(nb, ny - nb), slice(nb, nx - nb))
    out = np.ones((ny, nx), dtype=np.float32)
    out[act] = flat[act]
    if pdq is not None:
        pdq |= np.where(np.logical_or(out < 0.1, out > 10), NO_FLAT_FIELD, np.uint32(0)).astype(np.uint32)
    out = np.clip(out, 0.1, 10)
    if ipc_deconvolve:
        g = gain[act]
        if pdq is not None:
            pdq[act] |= np.where(g <= 0.1, NO_GAIN_VALUE, np.uint32(0)).astype(np.uint32)
            g = np.clip(g, 0.1, None)
        out[act] = ipc.ipc_rev(out[act], kernel, gain=g)
    return out


def finish(slope, err_read, err_poisson, pdq, nb, dark_rate, dark_dq, flat_dn, area_factor):
    """From the ramp-fit planes to the flat-fielded L2 planes.

    slope/err_read/err_poisson: f32 (ny,nx) from ``rampfit.ramp_fit``; pdq u32 updated in place by
    the caller before (flat flags) -- here only the dark dq is OR-ed.  ``dark_rate`` is the
    (IPC-deconvolved) dark slope or None; ``flat_dn`` the output of ``get_flat``; ``area_factor``
    f64 plane or None (then 1).  Returns slope, err_read, err_poisson (f32, full frame).
    """
    err = np.hypot(err_read, err_poisson)
    var_poisson = err_poisson**2
    slope = embed_active(slope, nb)
    err = embed_active(err, nb)
    var_poisson = embed_active(var_poisson, nb)
    if dark_rate is not None:
        slope[nb:-nb, nb:-nb] -= dark_rate[nb:-nb, nb:-nb]
        if dark_dq is not None:
            pdq[nb:-nb, nb:-nb] |= dark_dq[nb:-nb, nb:-nb]
    err_poisson = np.sqrt(var_poisson)
    err_read = np.sqrt(np.clip(err**2 - err_poisson**2, 0.0, None))
    if flat_dn is not None:
        af = 1.0 if area_factor is None else area_factor
        flat = (flat_dn / af).astype(np.float32)
        slope /= flat
        err_read /= flat
        err_poisson /= flat
    return slope, err_read, err_poisson
