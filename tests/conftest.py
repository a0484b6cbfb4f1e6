import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU machine)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


@pytest.fixture
def golden():
    return load_golden


def assert_same_bits(a, b, what="", zero_sign_ok=False):
    """bit-for-bit equality (NaN == NaN; -0 != +0 unless zero_sign_ok), dtype and shape included."""
    a, b = np.asarray(a), np.asarray(b)
    assert a.dtype == b.dtype, f"{what}: dtype {a.dtype} != {b.dtype}"
    assert a.shape == b.shape, f"{what}: shape {a.shape} != {b.shape}"
    if a.dtype.kind == "f":
        ai, bi = a.view(f"u{a.dtype.itemsize}"), b.view(f"u{b.dtype.itemsize}")
        # NaNs of any payload compare equal
        both_nan = np.isnan(a) & np.isnan(b)
        bad = (ai != bi) & ~both_nan
        if zero_sign_ok:
            bad &= ~((a == 0) & (b == 0))
    else:
        bad = a != b
    n = int(np.count_nonzero(bad))
    if n:
        idx = tuple(np.argwhere(bad)[0])
        raise AssertionError(f"{what}: {n} of {a.size} elements differ; first at {idx}: {a[idx]!r} vs {b[idx]!r}")


def gpu_context():
    """The process-wide libromanhip context; the test FAILS (not skips) if the library or GPU is missing."""
    from romanimpreprocess_amd import _native
    return _native.default_context(0)


def l1sim_golden_cal(g):
    """The calibration dict of the ``l1sim`` fixture (keys ``cal_<file>_<entry>[_<sub>]``), and its read pattern."""
    cal = {}
    for key in list(g.keys()):
        if not key.startswith("cal_"):
            continue
        _, name, rest = key.split("_", 2)
        node = cal.setdefault(name, {})
        if name == "read" and rest.split("_", 1)[0] in ("anc", "amp33"):
            sub, leaf = rest.split("_", 1)
            v = g[key]
            node.setdefault(sub, {})[leaf] = v if v.ndim else v.item()
        else:
            v = g[key]
            node[rest] = v if v.ndim else v.item()
    counts, flat = g["read_pattern_counts"], list(g["read_pattern_flat"])
    rp, at = [], 0
    for c in counts:
        rp.append([int(r) for r in flat[at:at + int(c)]])
        at += int(c)
    return cal, rp
