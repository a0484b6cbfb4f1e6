"""Simple sky estimation on the GPU -- same call surface as the reference's ``utils/sky.py`` (``binkxk``,
``smooth_mode``, ``medfit``).  The reductions over the image run in ``libromanhip`` (``post.hip``); the handful of
scalar steps around them (percentile interpolation, peak fit, the 6x6 normal equations, the Legendre tables) follow the
reference on the host."""

import numpy as np
import scipy.stats
from scipy.special import legendre_p

from .. import _native
from ..devarray import DevArray, is_dev


def _f32(a):
    if is_dev(a):   # a float32 plane resident in HBM (devarray.DevArray): handed over as it is
        if a.dtype != np.float32:
            raise TypeError("device arrays must be float32 here")
        return a
    return np.ascontiguousarray(a, dtype=np.float32)


def binkxk(arr, k, mask=None, ctx=None):
    """k x k block means of a 2-D array (``sky.py:20-41``); remainder pixels are ignored.  ``mask`` (True = use NaN
    there) fuses the reference's ``np.where(np.logical_not(m), slope, np.nan)`` (``gen_cal_image.py:642``)."""
    ctx = ctx or _native.default_context()
    a = _f32(arr)
    ny, nx = a.shape
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    out = np.empty((ny // k, nx // k), np.float32)
    ctx.check(ctx.lib.rip_stage_bin_mean(ctx.h, a.ctypes.data, None if m is None else m.ctypes.data, ny, nx, int(k),
                                         out.ctypes.data))
    return out


def _select(ctx, a, y0, x0, ky, kx, nby, nbx, ranks):
    """counts (nblk,), values (nblk, nranks) of the given 0-based ranks among the non-NaN elements of each block."""
    ny, nx = a.shape
    nblk = nby * nbx
    counts = np.zeros(nblk, np.int64)
    if ranks is None:
        ctx.check(ctx.lib.rip_stage_select_ranks(ctx.h, a.ctypes.data, ny, nx, y0, x0, ky, kx, nby, nbx, 0, None,
                                                 counts.ctypes.data, None))
        return counts, None
    r = np.ascontiguousarray(ranks, dtype=np.int64).reshape(nblk, -1)
    vals = np.empty(r.shape, np.float32)
    ctx.check(ctx.lib.rip_stage_select_ranks(ctx.h, a.ctypes.data, ny, nx, y0, x0, ky, kx, nby, nbx, r.shape[1],
                                             r.ctypes.data, counts.ctypes.data, vals.ctypes.data))
    return counts, vals


def _linear_index(n, q):
    """(previous rank, next rank, weight) of numpy's ``linear`` percentile on a float32 array of ``n`` valid values:
    q/100, the virtual index (n-1)*q and the weight are float32 there (numpy/lib/_function_base_impl.py: _quantile)."""
    q32 = np.asanyarray(np.true_divide(q, np.float32(100)))
    vi = np.asanyarray((n - 1) * q32)
    prev = int(np.floor(vi))
    nxt = prev + 1
    if vi >= n - 1:  # numpy takes the last element, indexed -1 (the weight is then vi + 1; both neighbours are equal)
        prev = nxt = -1
    if vi < 0:
        prev = nxt = 0
    gamma = np.asanyarray(np.asanyarray(vi - np.intp(prev)), dtype=vi.dtype)
    return prev % n, nxt % n, gamma


def _lerp32(a, b, t):
    """numpy's _lerp on two float32 order statistics with a float32 weight."""
    a, b = np.asanyarray(a, dtype=np.float32), np.asanyarray(b, dtype=np.float32)
    d = np.subtract(b, a)
    r = np.asanyarray(np.add(a, d * t))
    np.subtract(b, d * (1 - t), out=r, where=t >= 0.5, casting="unsafe", dtype=type(r.dtype))
    return r[()]


def nanpercentiles(arr, qs, ctx=None):
    """``np.nanpercentile(arr, q)`` (linear method, float32 input) for each q: the two neighbouring order statistics are
    found exactly on the GPU, the interpolation repeats numpy's float32 arithmetic, so the results are bit-identical."""
    ctx = ctx or _native.default_context()
    a = _f32(arr)
    if a.ndim != 2:
        a = a.reshape(1, -1)
    ny, nx = a.shape
    n = int(_select(ctx, a, 0, 0, ny, nx, 1, 1, None)[0][0])
    if n == 0:
        return [np.float32(np.nan) for _ in qs]
    idx = [_linear_index(n, q) for q in qs]
    ranks = np.array([r for p, nx_, _ in idx for r in (p, nx_)], dtype=np.int64)
    vals = _select(ctx, a, 0, 0, ny, nx, 1, 1, ranks)[1][0]
    return [_lerp32(vals[2 * i], vals[2 * i + 1], g) for i, (_, _, g) in enumerate(idx)]


def smooth_mode(arr, pc=25.0, pksmooth=0.5, niter=3, ctx=None):
    """Mode of the smoothed histogram, ignoring NaNs (``sky.py:44-97``).  Returns (mode, width of the weighting).

    Start: centre = median, sigma from the inter-percentile range of a Gaussian.  Each iteration evaluates the
    Gaussian-smoothed density on the 19 interior nodes of a 21-node grid over centre +- sigma (GPU reduction over the
    image) and moves the centre to the vertex of the parabola through the highest node and its two neighbours."""
    ctx = ctx or _native.default_context()
    img = _f32(arr)
    p_lo, p_mid, p_hi = nanpercentiles(img, (pc, 50.0, 100.0 - pc), ctx=ctx)
    sigma = (p_hi - p_lo) / (scipy.stats.norm.ppf((100.0 - pc) / 100.0) * 2)
    centre = p_mid
    nodes = 21
    for _ in range(niter):
        grid = centre + np.linspace(-1, 1, nodes) * sigma
        inner = np.ascontiguousarray(grid[1:nodes - 1], dtype=np.float64)
        dens = np.zeros(nodes)
        got = np.zeros(nodes - 2, np.float64)
        ctx.check(ctx.lib.rip_stage_gauss_hist(ctx.h, img.ctypes.data, img.size, inner.ctypes.data, nodes - 2,
                                               float(pksmooth * sigma), got.ctypes.data))
        dens[1:nodes - 1] = got
        top = np.argmax(dens)
        slope_ = (dens[top + 1] - dens[top - 1]) / 2.0
        curv = (dens[top + 1] + dens[top - 1]) / 2.0 - dens[top]
        centre = grid[top] + (grid[1] - grid[0]) * (-slope_ / 2.0 / curv)
    return (centre, sigma * pksmooth)


def block_nanmedians(arr, N, ctx=None):
    """(N, N) nan-medians of the N x N grid of equal blocks centred in the image (``sky.py:139-152``)."""
    ctx = ctx or _native.default_context()
    a = _f32(arr)
    ny, nx = a.shape
    kx, ky = nx // N, ny // N
    px, py = (nx % N) // 2, (ny % N) // 2
    counts, _ = _select(ctx, a, py, px, ky, kx, N, N, None)
    ranks = np.stack([np.maximum(counts - 1, 0) // 2, counts // 2], axis=1)
    _, vals = _select(ctx, a, py, px, ky, kx, N, N, ranks)
    med = np.where(counts > 0, np.mean(vals, axis=1, dtype=np.float32), np.float32(np.nan)).astype(np.float32)
    # an odd count asks twice for the same rank: mean(v, v) = v exactly
    return med.reshape(N, N)


def medfit(arr, N=8, order=2, subtract=False, ctx=None, want_model=True):
    """Low-order 2-D Legendre fit to the block medians (``sky.py:100-191``).  Returns (coef, model f32); with
    ``subtract`` the model is also subtracted from ``arr`` in place (``gen_cal_image.py:646-647``).  ``arr`` may be a
    ``DevArray`` (a plane in HBM); ``want_model=False`` (with ``subtract``) returns None for the model instead of
    bringing 67 MB back to the host."""
    ctx = ctx or _native.default_context()
    if subtract and not ((isinstance(arr, np.ndarray) or is_dev(arr)) and arr.dtype == np.float32 and arr.flags.c_contiguous):
        raise TypeError("subtract=True needs a C-contiguous float32 array (updated in place)")
    if not want_model and not subtract:
        raise ValueError("want_model=False only makes sense with subtract=True")
    a = arr if subtract else _f32(arr)
    ny, nx = a.shape
    kx, ky = nx // N, ny // N
    px, py = (nx % N) // 2, (ny % N) // 2
    # block centres mapped to [-1, 1) as the reference maps them, medians on the GPU
    uc = 2 * (px - 0.5 + kx * np.linspace(0.5, N - 0.5, N)) / nx - 1
    vc = 2 * (py - 0.5 + ky * np.linspace(0.5, N - 0.5, N)) / ny - 1
    ug, vg = np.meshgrid(uc, vc)
    meds = block_nanmedians(a, N, ctx=ctx)

    # basis functions P_i(u) P_j(v), i + j <= order, in the reference's coefficient order
    pairs = [(i, j) for i in range(order + 1) for j in range(order + 1 - i)]
    nc = len(pairs)
    basis = np.stack([np.reshape(legendre_p(i, ug), ug.shape) * np.reshape(legendre_p(j, vg), vg.shape) for i, j in pairs])
    # normal equations accumulated block by block (x index outer, y index inner: the reference's summation order)
    A = np.zeros((nc, nc))
    rhs = np.zeros(nc)
    for bx in range(N):
        for by in range(N):
            m = meds[by, bx]
            if m == m:
                col = basis[:, by, bx]
                A += np.multiply.outer(col, col)
                rhs += m * col
    x = np.linalg.solve(A, rhs)

    # Legendre tables on the pixel grid for the model evaluation on the GPU
    LPX = np.stack([np.reshape(legendre_p(i, np.linspace(-1, 1 - 2 / nx, nx)), nx) for i in range(order + 1)]).astype(np.float64)
    LPY = np.stack([np.reshape(legendre_p(j, np.linspace(-1, 1 - 2 / ny, ny)), ny) for j in range(order + 1)]).astype(np.float64)
    LPX, LPY = np.ascontiguousarray(LPX), np.ascontiguousarray(LPY)
    model = np.empty((ny, nx), np.float32) if want_model else None
    coef = np.ascontiguousarray(x, dtype=np.float64)
    ctx.check(ctx.lib.rip_stage_legendre2d(ctx.h, a.ctypes.data if subtract else None, ny, nx, int(order), LPX.ctypes.data,
                                           LPY.ctypes.data, coef.ctypes.data, int(bool(subtract)),
                                           None if model is None else model.ctypes.data))
    return x, model


def endslice(rdq, nborder, ctx=None):
    """SLICEOUT plane (``gen_cal_image.py:697-712``): int8 (ny-2nb, nx-2nb), index of the last unsaturated group - 1."""
    ctx = ctx or _native.default_context()
    r = np.ascontiguousarray(rdq, dtype=np.uint8)
    G, ny, nx = r.shape
    if G >= 128:
        raise ValueError("too many groups")
    out = np.empty((ny - 2 * nborder, nx - 2 * nborder), np.int8)
    ctx.check(ctx.lib.rip_stage_endslice(ctx.h, r.ctypes.data, G, ny, nx, int(nborder), out.ctypes.data))
    return out
