"""Oracle: inter-pixel-capacitance forward operator and order-2 deconvolution
(SURVEY.md 8a rows A5-A7).  Test infrastructure only.

Follows ``utils/ipc_linearity.py:37-99`` (``ipc_fwd``), ``:102-142`` (``ipc_rev``) and
``:145-186`` (``correct_cube``).

out[y,x] = sum_{dy,dx} in[y-dy, x-dx] * K[1+dy, 1+dx, y-dy, x-dx], accumulated in the fixed
order  centre, (1,0), (-1,0), (0,1), (0,-1), (1,1), (1,-1), (-1,1), (-1,-1); every product is
rounded, then added (no FMA); a source pixel outside the frame contributes NO term (it is
not added as zero).  Working dtype = promote(image, kernel, gain).
"""

import numpy as np

# (dy, dx) in the accumulation order of the reference, after the centre term
NEIGHBOUR_ORDER = ((1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (1, -1), (-1, 1), (-1, -1))


def _span(d, n):
    """Destination and source slices along one axis for a shift by d (dest = src + d)."""
    if d > 0:
        return slice(d, n), slice(0, n - d)
    if d < 0:
        return slice(0, n + d), slice(-d, n)
    return slice(0, n), slice(0, n)


def ipc_fwd(image, kernel, gain=None):
    im = image if gain is None else gain * image
    ny, nx = im.shape
    out = im * kernel[1, 1]
    for dy, dx in NEIGHBOUR_ORDER:
        yd, ys = _span(dy, ny)
        xd, xs = _span(dx, nx)
        out[yd, xd] += im[ys, xs] * kernel[1 + dy, 1 + dx, ys, xs]
    if gain is not None:
        out /= gain
    return out


def ipc_rev(image, kernel, order=2, gain=None):
    x = image if gain is None else gain * image
    out = np.copy(x)
    for _ in range(order):
        out = out + x - ipc_fwd(out, kernel)
    if gain is not None:
        out /= gain
    return out


def border_of(nx, kernel):
    """``nb = (8192 + (nx - K.shape[-1]) // 2) % 16`` (``ipc_linearity.py:177``)."""
    return (8192 + (nx - kernel.shape[-1]) // 2) % 16


def correct_cube(data, kernel, gain=None):
    """In-place IPC deconvolution of the active region of every group of ``data`` (f32 cube).

    ``data[i, act] = ipc_rev(data[i, act] * g, K) / g`` with g = gain[act] (or 1.0).
    """
    if kernel is None:
        return data
    ngrp, ny, nx = data.shape
    nb = border_of(nx, kernel)
    act = (slice(nb, ny - nb), slice(nb, nx - nb))
    g = 1.0 if gain is None else np.copy(gain[act])
    for i in range(ngrp):
        data[i][act] = ipc_rev(data[i][act] * g, kernel) / g
    return data
