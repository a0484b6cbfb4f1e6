"""Ramp fitting on the GPU -- same call surface as the reference's ``utils/fitting.py``.

``construct_weights`` is host numpy (a <= 64x64 f64 inverse, ``fitting.py:20-86``); ``ramp_fit``
(``fitting.py:258-355``: slope + read/Poisson errors, jump detection, saturation-truncated refits,
flag propagation) runs in ``rampfit.hip`` through ``rip_stage_ramp_fit``.
"""

import numpy as np

from .. import _native, calio, plan as planmod
from ..plan import construct_weights  # noqa: F401  (re-exported: same name and meaning as the reference)


def ramp_fit(data, rdq, pdq, meta, caldir, mylog, exclude_first=True, ctx=None):
    """Fit slopes to ``data`` (ngrp,ny,nx) f32; ``rdq`` (u8) and ``pdq`` (u32) are updated in place.

    ``meta`` needs ``ngrp, N, tbar, tau, K, nborder`` (optionally ``jump_detect_pars``); ``caldir`` needs
    ``gain`` and ``read`` (paths or in-memory trees).  Returns (slope, slope_err_read, slope_err_poisson).
    """
    ctx = ctx or _native.default_context()
    data = np.ascontiguousarray(data, dtype=np.float32)
    G, ny, nx = data.shape
    if G != meta["ngrp"]:
        raise ValueError(f"data has {G} groups, meta {meta['ngrp']}")
    if rdq.dtype != np.uint8 or pdq.dtype != np.uint32 or not rdq.flags.c_contiguous or not pdq.flags.c_contiguous:
        raise TypeError("rdq must be a C-contiguous uint8 cube and pdq a C-contiguous uint32 plane (updated in place)")
    with calio.open_tree(caldir["gain"]) as f:
        gain = np.ascontiguousarray(f["roman"]["data"])
    with calio.open_tree(caldir["read"]) as f:
        read = np.ascontiguousarray(f["roman"]["data"], dtype=np.float32)
    if gain.dtype not in (np.float32, np.float64):
        gain = gain.astype(np.float64)
    desc = planmod.plan_desc(meta, meta["K"], exclude_first, True, meta.get("jump_detect_pars"))
    pid = ctx.create_plan(desc)
    try:
        slope = np.empty((ny, nx), np.float32)
        er = np.empty((ny, nx), np.float32)
        ep = np.empty((ny, nx), np.float32)
        ctx.check(ctx.lib.rip_stage_ramp_fit(
            ctx.h, pid, data.ctypes.data, rdq.ctypes.data, pdq.ctypes.data, ny, nx, int(meta["nborder"]),
            gain.ctypes.data, _native.dtype_code(gain), read.ctypes.data, slope.ctypes.data, er.ctypes.data,
            ep.ctypes.data))
    finally:
        ctx.destroy_plan(pid)
    if mylog is not None:
        mylog.append(f"ramp fit on device {ctx.device}: {G} groups, {desc.nvariants} fit variants, K = {meta['K']}\n")
    return slope, er, ep
