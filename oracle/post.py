"""Oracle: post-path 2-D reductions.  Test infrastructure only.

numpy restatement of ``utils/maskhandling.py:82-117`` (CombinedMask.build), ``utils/sky.py:20-41`` (binkxk),
``:44-97`` (smooth_mode), ``:100-191`` (medfit) and the SLICEOUT loop of ``gen_cal_image.py:697-712``; pinned by the
goldens ``post_mask`` / ``post_sky`` that ``tools/make_goldens.py post`` makes with the reference's own modules.
"""

import numpy as np
import scipy.stats
from scipy.special import legendre_p

# growth of each pixel-dq bit in the reference's standard mask (maskhandling.py:152-180): bit -> 1 | 5 | 9 | 25
PIXELMASK1 = {0: 1, 2: 5, 3: 25, 4: 1, 5: 1, 6: 5, 8: 1, 9: 1, 10: 9, 11: 9, 12: 1, 13: 9, 15: 1, 18: 9, 19: 9, 20: 9,
              21: 9, 22: 1, 23: 9, 24: 9, 25: 9, 28: 9, 30: 9}


def _shifted(layer, dy, dx):
    """layer moved by (dy, dx) with zero fill (what a zero-padded 'same' convolution sees)."""
    out = np.zeros_like(layer)
    ny, nx = layer.shape
    ys, yd = (slice(0, ny - dy), slice(dy, ny)) if dy >= 0 else (slice(-dy, ny), slice(0, ny + dy))
    xs, xd = (slice(0, nx - dx), slice(dx, nx)) if dx >= 0 else (slice(-dx, nx), slice(0, nx + dx))
    out[yd, xd] = layer[ys, xs]
    return out


def build_mask(dq, growth=PIXELMASK1):
    """True where a pixel is masked: each flagged bit's layer, grown by its footprint, OR-ed together."""
    dq = np.asarray(dq, dtype=np.uint32)
    mask = np.zeros(dq.shape, dtype=bool)
    for bit, grow in growth.items():
        layer = (dq & np.uint32(1 << bit)) != 0
        if grow == 1:
            mask |= layer
            continue
        reach = 2 if grow == 25 else 1
        for dy in range(-reach, reach + 1):
            for dx in range(-reach, reach + 1):
                if grow == 5 and abs(dy) + abs(dx) > 1:
                    continue
                mask |= _shifted(layer, dy, dx)
    return mask


def binkxk(arr, k):
    ny, nx = arr.shape
    return np.mean(arr[: k * (ny // k), : k * (nx // k)].reshape(ny // k, k, nx // k, k), axis=(1, 3))


def smooth_mode(arr, pc=25.0, pksmooth=0.5, niter=3):
    lo, mid, hi = (np.nanpercentile(arr, q) for q in (pc, 50.0, 100.0 - pc))
    sigma = (hi - lo) / (scipy.stats.norm.ppf((100.0 - pc) / 100.0) * 2)
    centre = mid
    nodes = 21
    for _ in range(niter):
        grid = centre + np.linspace(-1, 1, nodes) * sigma
        dens = np.zeros(nodes)
        for i in range(1, nodes - 1):
            w = np.exp(-0.5 * ((grid[i] - arr) / (pksmooth * sigma)) ** 2)
            dens[i] = np.sum(np.where(np.isnan(w), 0.0, w))
        top = np.argmax(dens)
        s1 = (dens[top + 1] - dens[top - 1]) / 2.0
        s2 = (dens[top + 1] + dens[top - 1]) / 2.0 - dens[top]
        centre = grid[top] + (grid[1] - grid[0]) * (-s1 / 2.0 / s2)
    return centre, sigma * pksmooth


def medfit(arr, N=8, order=2):
    ny, nx = arr.shape
    kx, ky = nx // N, ny // N
    px, py = (nx % N) // 2, (ny % N) // 2
    uc = 2 * (px - 0.5 + kx * np.linspace(0.5, N - 0.5, N)) / nx - 1
    vc = 2 * (py - 0.5 + ky * np.linspace(0.5, N - 0.5, N)) / ny - 1
    ug, vg = np.meshgrid(uc, vc)
    with np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            meds = np.nanmedian(arr[py:py + N * ky, px:px + N * kx].reshape(N, ky, N, kx), axis=(1, 3))
    pairs = [(i, j) for i in range(order + 1) for j in range(order + 1 - i)]
    basis = np.stack([np.reshape(legendre_p(i, ug), ug.shape) * np.reshape(legendre_p(j, vg), vg.shape) for i, j in pairs])
    A = np.zeros((len(pairs),) * 2)
    rhs = np.zeros(len(pairs))
    for bx in range(N):
        for by in range(N):
            if meds[by, bx] == meds[by, bx]:
                col = basis[:, by, bx]
                A += np.multiply.outer(col, col)
                rhs += meds[by, bx] * col
    coef = np.linalg.solve(A, rhs)
    LPX = [np.reshape(legendre_p(i, np.linspace(-1, 1 - 2 / nx, nx)), nx) for i in range(order + 1)]
    LPY = [np.reshape(legendre_p(j, np.linspace(-1, 1 - 2 / ny, ny)), ny) for j in range(order + 1)]
    model = np.zeros((ny, nx))
    for k, (i, j) in enumerate(pairs):
        model += coef[k] * np.outer(LPY[j], LPX[i])
    return coef, model.astype(arr.dtype)


def endslice(rdq, nb):
    act = (slice(nb, -nb), slice(nb, -nb))
    out = np.zeros(rdq[0][act].shape, dtype=np.int8) - 1
    for iend in range(1, rdq.shape[0]):
        first = ((rdq[iend][act] & ~rdq[iend - 1][act]) & np.uint8(2)) != 0
        out = np.where(first, np.int8(iend - 1), out)
    return out
