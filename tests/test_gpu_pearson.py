"""Pseudo-Poisson noise layers (the ``O`` directives): the Pearson-family sampler on the GPU.

The distribution PARAMETERS of every pixel are pinned by goldens made with the reference's own module
(``tools/make_goldens.py pearson``: ``L1_to_L2/GalPoisson/draw_with_tilnus.py`` with its samplers replaced by recorders); the
DEVIATES come from the device's generator (the reference's scipy / numpy streams are not reproducible): their first four
moments are tested against the targets the sampler is built to match."""

import numpy as np
import pytest
from conftest import gpu_context, load_golden

from romanimpreprocess_amd.L1_to_L2.GalPoisson import draw_with_tilnus as dw

pytestmark = pytest.mark.gpu

CASES = ["poisson_like", "neg_skew", "beta_prime", "beta_prime_neg", "heavy_tail", "heavy_tail_neg", "symmetric", "light_tail",
         "ramp_fit_weights"]


@pytest.mark.parametrize("name", CASES)
def test_types_and_parameters_match_the_reference(name):
    g = load_golden("pearson_params")
    t21, t31, t41 = g[f"cl_{name}_t"]
    types, params = dw.classify(t21, t31, t41, g["cl_I"], ctx=gpu_context())
    ref_t, ref_p = g[f"cl_{name}_types"], g[f"cl_{name}_params"]
    np.testing.assert_array_equal(types, ref_t)
    # f64 formulas in the reference's order; powers (** 3, ** 1.5) may round differently in the last place
    np.testing.assert_allclose(params, ref_p, rtol=1e-11, atol=1e-300)


@pytest.mark.parametrize("name,I", [("poisson_like", 3.0), ("poisson_like", 400.0), ("neg_skew", 20.0), ("beta_prime", 30.0),
                                    ("beta_prime_neg", 60.0), ("heavy_tail", 40.0), ("heavy_tail_neg", 80.0), ("symmetric", 25.0),
                                    ("light_tail", 50.0), ("ramp_fit_weights", 2000.0)])
def test_moments_of_the_deviates(name, I):
    g = load_golden("pearson_params")
    t21, t31, t41 = (float(v) for v in g[f"cl_{name}_t"])
    n = 2_000_000
    x = dw.draw_from_Pearson(t21, t31, t41, np.full(n, I), rng=np.random.default_rng(5), ctx=gpu_context())
    assert np.all(np.isfinite(x))
    m2, m3, m4 = t21 * I, t31 * I, 3.0 * t21**2 * I**2 + t41 * I
    assert m4 > 0 and m2 > 0
    sd = np.sqrt(m2)
    # standard errors of the sample moments from the target moments themselves (sixth / eighth moments bounded generously)
    assert abs(np.mean(x)) < 6.0 * sd / np.sqrt(n)
    assert abs(np.var(x) / m2 - 1.0) < 6.0 * np.sqrt((m4 / m2**2 - 1.0) / n) + 1e-3
    c3 = np.mean((x - x.mean()) ** 3)
    assert abs(c3 - m3) < 0.03 * sd**3 * max(1.0, m4 / m2**2 / 3.0)
    c4 = np.mean((x - x.mean()) ** 4)
    assert abs(c4 / m4 - 1.0) < 0.06 * max(1.0, m4 / m2**2 / 3.0)
    # a different stream gives different deviates, the same arguments the same ones
    y = dw.draw_from_Pearson(t21, t31, t41, np.full(1000, I), rng=np.random.default_rng(5), ctx=gpu_context(), stream=7)
    z = dw.draw_from_Pearson(t21, t31, t41, np.full(1000, I), rng=np.random.default_rng(5), ctx=gpu_context(), stream=7)
    w = dw.draw_from_Pearson(t21, t31, t41, np.full(1000, I), rng=np.random.default_rng(5), ctx=gpu_context(), stream=8)
    assert np.array_equal(y, z) and not np.array_equal(y, w)


def test_inadmissible_and_edge_inputs():
    x = dw.draw_from_Pearson(1.0, 1.0, 1.7, np.array([1e-3, 0.0, -4.0, np.nan, 1e5]), rng=1, ctx=gpu_context())
    t, _p = dw.classify(1.0, 1.0, 1.7, np.array([1e-3, 0.0, -4.0, np.nan, 1e5]), ctx=gpu_context())
    assert t.tolist()[:4] == [0, 0, 0, 0] and t[4] == 6
    assert np.all(x[:4] == 0.0) and np.isfinite(x[4])
    assert dw.draw_from_Pearson(1.0, 1.0, 1.0, np.zeros((0,)), ctx=gpu_context()).shape == (0,)
