"""Post-path 2-D reductions on the GPU against goldens made by the reference's own
``utils/maskhandling.py`` and ``utils/sky.py`` (tools/make_goldens.py post)."""

import numpy as np
import pytest
from conftest import assert_same_bits, gpu_context, load_golden

from romanimpreprocess_amd.utils import maskhandling, sky

pytestmark = pytest.mark.gpu


def test_pixelmask1_build_is_exact():
    g = load_golden("post_mask")
    got = maskhandling.PixelMask1.build(g["dq"], ctx=gpu_context())
    assert got.dtype == bool
    assert_same_bits(got.astype(np.uint8), g["mask"], "PixelMask1.build")
    assert 0.02 < got.mean() < 0.6


def test_binkxk_and_smooth_mode():
    g = load_golden("post_sky")
    ctx = gpu_context()
    binned = sky.binkxk(g["img"], 4, mask=g["mask"].astype(bool), ctx=ctx)
    ref = g["binned"]
    assert binned.shape == ref.shape
    assert np.array_equal(np.isnan(binned), np.isnan(ref))
    # f32 sums of 16 values in another order than numpy's: a few ulp
    np.testing.assert_allclose(binned, ref, rtol=5e-7, equal_nan=True)
    # mode of the smoothed histogram: order statistics exact, f64 Gaussian sums in another order
    ctr, width = sky.smooth_mode(ref, ctx=ctx)
    np.testing.assert_allclose([ctr, width], g["mode"], rtol=1e-9)
    ctr2, width2 = sky.smooth_mode(binned, ctx=ctx)
    np.testing.assert_allclose([ctr2, width2], g["mode"], rtol=1e-5)


def test_nanpercentiles_match_numpy():
    rng = np.random.default_rng(3)
    a = rng.standard_normal((123, 77)).astype(np.float32)
    a[rng.random(a.shape) < 0.1] = np.nan
    a[5, 5] = np.inf
    a[6, 6] = -np.inf
    qs = (0.0, 25.0, 50.0, 75.0, 99.9, 100.0)
    got = sky.nanpercentiles(a, qs, ctx=gpu_context())
    for q, v in zip(qs, got):
        assert_same_bits(np.float32(v), np.float32(np.nanpercentile(a, q)), f"nanpercentile {q}")


@pytest.mark.parametrize("order", [1, 2, 3])
def test_medfit_matches_reference(order):
    g = load_golden("post_sky")
    ctx = gpu_context()
    arr = g["withnan"].copy()
    coef, model = sky.medfit(arr, order=order, ctx=ctx)
    # block nan-medians are exact, the 6x6 solve is numpy's, the model is accumulated in the reference's order
    assert_same_bits(np.asarray(coef, np.float64), g[f"coef{order}"], "medfit coefficients")
    assert_same_bits(model, g[f"model{order}"], "medfit model")
    work = arr.copy()
    sky.medfit(work, order=order, subtract=True, ctx=ctx)
    assert_same_bits(work, arr - g[f"model{order}"], "arr - model", zero_sign_ok=True)


def test_medfit_other_grid():
    g = load_golden("post_sky")
    coef, model = sky.medfit(g["img"], N=4, order=2, ctx=gpu_context())
    assert_same_bits(np.asarray(coef, np.float64), g["coef_n4"], "medfit coefficients (N=4)")
    assert_same_bits(model, g["model_n4"], "medfit model (N=4)")


def test_block_nanmedians_even_counts_and_empty_blocks():
    rng = np.random.default_rng(11)
    a = rng.standard_normal((64, 96)).astype(np.float32)
    a[rng.random(a.shape) < 0.2] = np.nan
    a[0:8, 0:12] = np.nan
    got = sky.block_nanmedians(a, 8, ctx=gpu_context())
    with np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = np.nanmedian(a.reshape(8, 8, 8, 12), axis=(1, 3))
    assert_same_bits(got, ref.astype(np.float32), "block nan-medians")
    assert np.isnan(got[0, 0])


def test_endslice_matches_the_numpy_loop():
    rng = np.random.default_rng(5)
    G, ny, nx, nb = 8, 40, 72, 4
    rdq = np.zeros((G, ny, nx), np.uint8)
    first = rng.integers(1, 2 * G, size=(ny, nx))  # group where saturation starts (>= G: never)
    for g_ in range(G):
        rdq[g_][first <= g_] |= np.uint8(2)
    rdq[3][rng.random((ny, nx)) < 0.1] |= np.uint8(4)
    act = (slice(nb, -nb), slice(nb, -nb))
    ref = np.zeros((ny - 2 * nb, nx - 2 * nb), np.int8) - 1
    for iend in range(1, G):
        hit = ((rdq[iend][act] & ~rdq[iend - 1][act]) & np.uint8(2)) != 0
        ref = np.where(hit, np.int8(iend - 1), ref)
    got = sky.endslice(rdq, nb, ctx=gpu_context())
    assert_same_bits(got, ref, "endslice")
