"""IPC and linearity on the GPU -- same call surface as the reference's ``utils/ipc_linearity.py``
(``ipc_fwd`` :37, ``ipc_rev`` :102, ``correct_cube`` :145, ``linearity`` :234, ``multilin`` :276) and, from the simulation side
(SURVEY.md 8f row 4), the inverse linearity ``invlinearity`` :347 and the ``IL`` class :397 that romanisim calls.
"""

import sys

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


def linearity(S, linearity_file, origin=(0, 0), ctx=None):
    """Linearity correction of one 2-D image (``ipc_linearity.py:234-273``): the Legendre series with linear extrapolation
    at every pixel (no substitution at flagged pixels, no clipping); dq = file dq | NO_LIN_CORR where extrapolated."""
    ctx = ctx or _native.default_context()
    S = np.ascontiguousarray(S, dtype=np.float32)
    dy, dx = S.shape
    y0, x0 = origin[1], origin[0]
    sl = (slice(y0, y0 + dy), slice(x0, x0 + dx))
    with calio.open_tree(linearity_file) as F:
        r = F["roman"]
        smin = np.ascontiguousarray(r["Smin"][sl], dtype=np.float32)
        smax = np.ascontiguousarray(r["Smax"][sl], dtype=np.float32)
        dq0 = np.array(r["dq"][sl], dtype=np.uint32)
        coefs = np.ascontiguousarray(r["data"][(slice(None),) + sl], dtype=np.float32)
    nodq = np.zeros((dy, dx), np.uint32)   # the cube entry substitutes S - Sref where the file's dq says so: not here
    phi = np.empty((1, dy, dx), np.float32)
    dq = np.empty((dy, dx), np.uint32)
    ctx.check(ctx.lib.rip_stage_multilin(
        ctx.h, S.ctypes.data, 1, dy, dx, coefs.shape[0], coefs.ctypes.data, smin.ctypes.data, smax.ctypes.data,
        smin.ctypes.data, nodq.ctypes.data, 0, None, phi.ctypes.data, dq.ctypes.data))
    return phi[0], dq0 | dq


def invlinearity(Slin, linearity_file, origin=(0, 0), ctx=None):
    """Inverse linearity by 24 bisection steps (``ipc_linearity.py:347-394``; the reference's slowest simulation step).
    ``Slin`` (ny,nx) in DN_lin -> ``(S, exflag)``: S in DN_raw with Slin's float dtype, exflag bool."""
    ctx = ctx or _native.default_context()
    Slin = _float_array(Slin)
    dy, dx = Slin.shape
    y0, x0 = origin[1], origin[0]
    sl = (slice(y0, y0 + dy), slice(x0, x0 + dx))
    with calio.open_tree(linearity_file) as F:
        r = F["roman"]
        smin = np.ascontiguousarray(r["Smin"][sl], dtype=np.float32)
        smax = np.ascontiguousarray(r["Smax"][sl], dtype=np.float32)
        coefs = np.ascontiguousarray(r["data"][(slice(None),) + sl], dtype=np.float32)
    if smin.shape != (dy, dx):
        raise ValueError(f"block {Slin.shape} at origin {origin} leaves the linearity file's frame")
    S = np.empty_like(Slin)
    ex = np.empty((dy, dx), np.uint8)
    ctx.check(ctx.lib.rip_stage_invlinearity(ctx.h, Slin.ctypes.data, _native.dtype_code(Slin), dy, dx, coefs.shape[0],
                                             coefs.ctypes.data, smin.ctypes.data, smax.ctypes.data, S.ctypes.data,
                                             ex.ctypes.data))
    return S, ex.astype(bool)


class IL:
    """IPC + inverse linearity as romanisim calls it (``ipc_linearity.py:397-513``): ``apply`` turns a linearised signal
    into the non-linear, IPC-convolved one.  Same constructor, attributes and methods as the reference's class; the two
    array operations run on the GPU (``rip_stage_ipc_image``, ``rip_stage_invlinearity``)."""

    def __init__(self, linearity_file, gain_file, ipc_file, start_e=0.0, ctx=None):
        self.linearity_file = linearity_file
        self.gain_file = gain_file
        self.ipc_file = ipc_file
        self.start_e = start_e
        self.ctx = ctx
        with calio.open_tree(self.linearity_file) as f:
            self._dq = np.copy(f["roman"]["dq"])

    def set_dq(self, ngroup=1, nborder=4):
        ny, nx = np.shape(self._dq)
        self.dq = np.zeros((ngroup, ny - 2 * nborder, nx - 2 * nborder), dtype=np.uint32)
        self.dq[:, :, :] = self._dq[None, nborder:ny - nborder, nborder:nx - nborder]

    def apply(self, counts, electrons=False, electrons_out=False):
        print("apply", electrons, electrons_out, np.shape(counts))
        sys.stdout.flush()
        if self.ipc_file is not None:
            with calio.open_tree(self.ipc_file) as f:
                counts_conv = ipc_fwd(counts + self.start_e, f["roman"]["data"], ctx=self.ctx)
        else:
            counts_conv = counts + self.start_e
        nyc, nxc = np.shape(counts)
        g_in = 1.0
        g_out = 1.0
        if electrons or electrons_out:
            with calio.open_tree(self.gain_file) as f:
                g = np.asarray(f["roman"]["data"])
                nyg = np.shape(g)[0]
                if nyg > nyc:
                    nb = (nyg - nyc) // 2
                    g = g[nb:-nb, nb:-nb]
            if electrons:
                g_in = g
            if electrons_out:
                g_out = g
        nb = (8192 - nyc // 2) % 16
        S, _ = invlinearity(counts_conv / g_in, self.linearity_file, origin=(nb, nb), ctx=self.ctx)
        if not electrons_out:
            return S
        with calio.open_tree(self.linearity_file) as F:
            return g_out * (S - F["roman"]["Sref"][nb:nb + nyc, nb:nb + nxc])
