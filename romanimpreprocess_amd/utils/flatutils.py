"""Flat field in DN units on the GPU -- same call surface as the reference's ``utils/flatutils.py:20``."""

import numpy as np

from .. import _native, calio


def _float_array(a):
    a = np.ascontiguousarray(a)
    if a.dtype not in (np.float32, np.float64):
        a = a.astype(np.float64)
    return a


def get_flat(caldir, meta, pdq, ipc_deconvolve=True, ctx=None):
    """Flat padded with 1 on the reference border, flagged + clipped to [0.1, 10], IPC-deconvolved in DN.

    ``pdq`` (uint32, may be None) receives NO_FLAT_FIELD / NO_GAIN_VALUE in place.
    """
    ctx = ctx or _native.default_context()
    nb = int(meta["nborder"])
    with calio.open_tree(caldir["flat"]) as f:
        flat = np.ascontiguousarray(f["roman"]["data"], dtype=np.float32)
    ny, nx = flat.shape
    g = k = None
    if ipc_deconvolve:
        with calio.open_tree(caldir["gain"]) as f:
            g = _float_array(f["roman"]["data"])
        with calio.open_tree(caldir["ipc4d"]) as f:
            k = _float_array(f["roman"]["data"])
    if pdq is not None and (pdq.dtype != np.uint32 or not pdq.flags.c_contiguous):
        raise TypeError("pdq must be a C-contiguous uint32 plane (updated in place)")
    out = np.empty((ny, nx), np.float32)
    ctx.check(ctx.lib.rip_stage_get_flat(
        ctx.h, flat.ctypes.data, ny, nx, nb, None if g is None else g.ctypes.data,
        0 if g is None else _native.dtype_code(g), None if k is None else k.ctypes.data,
        0 if k is None else _native.dtype_code(k), int(bool(ipc_deconvolve)), None if pdq is None else pdq.ctypes.data,
        out.ctypes.data))
    return out
