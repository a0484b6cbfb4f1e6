"""Ramp fitting on the GPU -- same call surface as the reference's ``utils/fitting.py``.

``construct_weights`` is host numpy (a <= 64x64 f64 inverse, ``fitting.py:20-86``); ``ramp_fit``
(``fitting.py:258-355``: slope + read/Poisson errors, jump detection, saturation-truncated refits,
flag propagation) runs in ``rampfit.hip`` through ``rip_stage_ramp_fit``; ``jump_detect`` (``fitting.py:89-255``: one
pass, significance cube returned) through ``rip_stage_jump_detect``.
"""

import numpy as np

from .. import _native, calio, plan as planmod
from ..plan import construct_weights  # noqa: F401  (re-exported: same name and meaning as the reference)


_JUMP_DET = 4   # dqflags.group.JUMP_DET


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


def jump_detect(data, rdq, pdq, meta, caldir, mylog, exclude_first=True, truncate_ramp=None, ctx=None):
    """One pass of slope fit + jump flagging (``fitting.py:89-255``).  ``data`` (ngrp,ny,nx) f32; ``rdq`` (uint8 -- or, as the
    reference's own cube, uint32 -- of at least the fitted groups) gets JUMP_DET OR-ed in place on the active region; ``pdq`` only gives the frame shape, as
    in the reference.  ``truncate_ramp=t``: groups [0, t) with the two-point weights of ``fitting.py:162-167``.
    Returns (slope, slope_err_read, slope_err_poisson, smap); ``smap`` is (2*(g-start)-3, ny, nx) float32."""
    ctx = ctx or _native.default_context()
    data = np.ascontiguousarray(data, dtype=np.float32)
    ny, nx = np.shape(pdq)
    start = 1 if exclude_first else 0
    g = int(meta["ngrp"]) if truncate_ramp is None else int(truncate_ramp)
    if data.shape[0] < g or data.shape[1:] != (ny, nx):
        raise ValueError(f"data {data.shape} does not hold {g} groups of {(ny, nx)}")
    if 2 * (g - start) - 3 < 0 or g > int(meta["ngrp"]):
        raise ValueError(f"cannot fit {g} groups (exclude_first={exclude_first}, ngrp={meta['ngrp']})")
    if not isinstance(rdq, np.ndarray) or rdq.dtype not in (np.uint8, np.uint32) or rdq.shape[0] < g or rdq.shape[1:] != (ny, nx):
        raise TypeError("rdq must be a uint8 or uint32 cube of at least the fitted groups (updated in place)")
    if truncate_ramp is None:
        K = np.asarray(meta["K"], dtype=np.float32)
    else:   # fitting.py:162-167, float32 as there
        K = np.zeros(g, dtype=np.float32)
        K[-1] = 1.0 / (meta["tbar"][g - 1] - meta["tbar"][start])
        K[start] = -K[-1]
    sub = {"ngrp": g, "tbar": np.asarray(meta["tbar"])[:g], "tau": np.asarray(meta["tau"])[:g], "N": np.asarray(meta["N"])[:g]}
    with calio.open_tree(caldir["gain"]) as f:
        gain = np.ascontiguousarray(f["roman"]["data"])
    with calio.open_tree(caldir["read"]) as f:
        read = np.ascontiguousarray(f["roman"]["data"], dtype=np.float32)
    if gain.dtype not in (np.float32, np.float64):
        gain = gain.astype(np.float64)
    desc = planmod.plan_desc(sub, K, exclude_first, True, meta.get("jump_detect_pars"))
    pid = ctx.create_plan(desc)
    try:
        cube = np.ascontiguousarray(data[:g])
        # the group flags the device reads are the low byte (DO_NOT_USE, SATURATED, JUMP_DET); a uint32 cube keeps its other bits
        flags = np.ascontiguousarray(rdq[:g], dtype=np.uint8) if rdq.dtype == np.uint8 else (rdq[:g] & np.uint32(0xFF)).astype(np.uint8)
        slope, er, ep = (np.empty((ny, nx), np.float32) for _ in range(3))
        smap = np.zeros((2 * (g - start) - 3, ny, nx), np.float32)
        ctx.check(ctx.lib.rip_stage_jump_detect(
            ctx.h, pid, cube.ctypes.data, flags.ctypes.data, ny, nx, int(meta["nborder"]), gain.ctypes.data,
            _native.dtype_code(gain), read.ctypes.data, slope.ctypes.data, er.ctypes.data, ep.ctypes.data, smap.ctypes.data))
        if rdq.dtype == np.uint8:
            rdq[:g] = flags
        else:
            rdq[:g] |= flags.astype(np.uint32) & np.uint32(_JUMP_DET)   # the pass only ever adds JUMP_DET (fitting.py:249)
    finally:
        ctx.destroy_plan(pid)
    if mylog is not None:
        mylog.append(f"truncate at {truncate_ramp}, K = {K}\n")
    return slope, er, ep, smap
