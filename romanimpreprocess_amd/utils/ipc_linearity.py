"""IPC and linearity on the GPU -- same call surface as the reference's ``utils/ipc_linearity.py``
(``ipc_fwd`` :37, ``ipc_rev`` :102, ``correct_cube`` :145, ``multilin`` :276).  The simulation-side
inverse (``invlinearity``, ``IL``) is outside the L1->L2 path.
"""

import numpy as np

from .. import _native, calio


def _float_array(a):
    a = np.ascontiguousarray(a)
    if a.dtype not in (np.float32, np.float64):
        a = a.astype(np.float64)
    return a


def _ipc_image(reverse, order, image, kernel, gain, ctx):
    ctx = ctx or _native.default_context()
    image = _float_array(image)
    kernel = _float_array(kernel)
    ny, nx = image.shape
    if kernel.shape != (3, 3, ny, nx):
        raise ValueError(f"kernel shape {kernel.shape} does not match image {image.shape}")
    g = None if gain is None else _float_array(np.broadcast_to(gain, image.shape))
    t64 = any(a is not None and a.dtype == np.float64 for a in (image, kernel, g))
    out = np.empty((ny, nx), np.float64 if t64 else np.float32)
    ctx.check(ctx.lib.rip_stage_ipc_image(
        ctx.h, int(reverse), int(order), image.ctypes.data, _native.dtype_code(image), ny, nx, kernel.ctypes.data,
        _native.dtype_code(kernel), None if g is None else g.ctypes.data, 0 if g is None else _native.dtype_code(g),
        out.ctypes.data))
    return out


def ipc_fwd(image, kernel, gain=None, ctx=None):
    """out[y,x] = sum_{dy,dx} in[y-dy,x-dx] kernel[1+dy,1+dx,y-dy,x-dx]; with ``gain``: g^-1 K g."""
    return _ipc_image(0, 0, image, kernel, gain, ctx)


def ipc_rev(image, kernel, order=2, gain=None, ctx=None):
    """Neumann-series inverse of ``ipc_fwd`` to the given order (footprint 2*order+1)."""
    return _ipc_image(1, order, image, kernel, gain, ctx)


def correct_cube(data, ipc_file, mylog, gain_file=None, ctx=None):
    """IPC-correct every group of ``data`` (ngrp,ny,nx) f32 IN PLACE (active region; border untouched)."""
    if ipc_file is None:
        if mylog is not None:
            mylog.append("No IPC file specified, skipping ...\n")
        return
    ctx = ctx or _native.default_context()
    if data.dtype != np.float32 or not data.flags.c_contiguous:
        raise TypeError("data must be a C-contiguous float32 cube (corrected in place)")
    with calio.open_tree(ipc_file) as F:
        kernel = _float_array(F["roman"]["data"])
    ngrp, ny, nx = data.shape
    nb = (8192 + (nx - kernel.shape[-1]) // 2) % 16
    if mylog is not None:
        mylog.append(f"IPC kernel center range --> {np.amin(kernel[1, 1]):f},{np.amax(kernel[1, 1]):f}\n")
        mylog.append(f" ..., {ngrp:d} groups, excluding {nb:d} border pixels\n")
    if kernel.shape != (3, 3, ny - 2 * nb, nx - 2 * nb):
        raise ValueError(f"ipc4d shape {kernel.shape} does not match a {ny}x{nx} frame with border {nb}")
    g = None
    if gain_file is not None:
        with calio.open_tree(gain_file) as G:
            g = _float_array(G["roman"]["data"])
    ctx.check(ctx.lib.rip_stage_correct_cube(
        ctx.h, data.ctypes.data, ngrp, ny, nx, nb, kernel.ctypes.data, _native.dtype_code(kernel),
        None if g is None else g.ctypes.data, 0 if g is None else _native.dtype_code(g)))


def multilin(S, linearity_file, origin=(0, 0), do_not_flag_first=True, attempt_corr=None, ctx=None):
    """Linearise the cube ``S`` (ngrp,ny,nx).  Returns (Slin f32 (ngrp,ny,nx), dq u32 (ny,nx))."""
    ctx = ctx or _native.default_context()
    S = np.ascontiguousarray(S, dtype=np.float32)
    ngrp, dy, dx = S.shape
    y0, x0 = origin[1], origin[0]
    sl = (slice(y0, y0 + dy), slice(x0, x0 + dx))
    with calio.open_tree(linearity_file) as F:
        r = F["roman"]
        smin = np.ascontiguousarray(r["Smin"][sl], dtype=np.float32)
        smax = np.ascontiguousarray(r["Smax"][sl], dtype=np.float32)
        sref = np.ascontiguousarray(r["Sref"][sl], dtype=np.float32)
        dq0 = np.ascontiguousarray(r["dq"][sl], dtype=np.uint32)
        coefs = np.ascontiguousarray(r["data"][(slice(None),) + sl], dtype=np.float32)
    ac = None
    if attempt_corr is not None:
        ac = np.ascontiguousarray(np.asarray(attempt_corr) != 0, dtype=np.uint8)
    phi = np.empty(S.shape, np.float32)
    dq = np.empty((dy, dx), np.uint32)
    ctx.check(ctx.lib.rip_stage_multilin(
        ctx.h, S.ctypes.data, ngrp, dy, dx, coefs.shape[0], coefs.ctypes.data, smin.ctypes.data, smax.ctypes.data,
        sref.ctypes.data, dq0.ctypes.data, int(bool(do_not_flag_first)), None if ac is None else ac.ctypes.data,
        phi.ctypes.data, dq.ctypes.data))
    return phi, dq
