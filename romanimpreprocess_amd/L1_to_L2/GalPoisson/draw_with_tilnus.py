"""Pearson-family deviates with given moment ratios, on the GPU: drop-in for the reference's
``L1_to_L2/GalPoisson/draw_with_tilnus.py:draw_from_Pearson`` (:12-135).  The parameters of every pixel's distribution are
the reference's formulas (pinned by goldens); the deviates come from the device's counter-based generator, keyed by a seed
taken from ``rng`` -- the reference's scipy / numpy streams are not reproduced (see ``csrc/pearson.hip``)."""

import numpy as np

from ... import _native

_calls = [0]


def _seed_from(rng):
    if rng is None:
        rng = np.random.default_rng()
    if hasattr(rng, "integers"):
        return int(rng.integers(0, 2**63 - 1))
    return int(np.random.default_rng(rng).integers(0, 2**63 - 1))


def classify(tilnu_21, tilnu_31, tilnu_41, I_arr, ctx=None):
    """(types int32, params (..., 4) f64) of every element: see ``rip_stage_pearson``."""
    ctx = ctx or _native.default_context()
    I = np.ascontiguousarray(I_arr, dtype=np.float64)
    types = np.empty(I.shape, np.int32)
    params = np.empty(I.shape + (4,), np.float64)
    ctx.check(ctx.lib.rip_stage_pearson(ctx.h, I.size, I.ctypes.data, float(tilnu_21), float(tilnu_31), float(tilnu_41), 0, 0,
                                        None, types.ctypes.data, params.ctypes.data))
    return types, params


def draw_from_Pearson(tilnu_21, tilnu_31, tilnu_41, I_arr, *, atol=0.0, rng=None, ctx=None, stream=None):
    """One deviate per element of ``I_arr`` (f64 array of the same shape; 0 outside the admissible region)."""
    if atol != 0.0:
        raise NotImplementedError("equality bands (atol > 0) are not built: the reference's caller uses the default 0")
    ctx = ctx or _native.default_context()
    I = np.ascontiguousarray(I_arr, dtype=np.float64)
    out = np.empty(I.shape, np.float64)
    if I.size == 0:
        return out
    _calls[0] += 1
    ctx.check(ctx.lib.rip_stage_pearson(ctx.h, I.size, I.ctypes.data, float(tilnu_21), float(tilnu_31), float(tilnu_41),
                                        _seed_from(rng), int(_calls[0] if stream is None else stream) & 0xFFFFFFFF,
                                        out.ctypes.data, None, None))
    return out
