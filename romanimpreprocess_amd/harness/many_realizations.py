"""Variance validation over many noise realisations of one ramp -- this package's counterpart of the
reference's ``validation_tests/many_realizations.py:47-106`` (BASELINE config 5).

For seeds SEED+10, SEED+20, ... one noisy L1 ramp of a fixed ideal-slope scene is generated (own generator:
``synth.make_ramp`` with a fixed ``rate``; the reference uses romanisim+galsim, not available offline), run
through the GPU chain, and the per-pixel moments of the unmasked slopes are accumulated:
    N = sum 1, S1 = sum x, S2 = sum x^2      (over realisations whose pixel dq passes ``good_mask``)
    mean = S1/N, std = sqrt(S2/N - mean^2)   (many_realizations.py:80-89; sentinel -1000 where N == 0, :87)
Seeds are shared round-robin between ranks; the three moment planes are summed with one all-reduce.
Output planes (f32): ideal, N, mean, std, mean - ideal, mean err.  (The reference additionally stores medians
over seeds of L1 differences, L2 and err, which need every seed's plane on one rank; not produced here.)
"""

import numpy as np

from .. import pars, sharding, synth
from ..dqflags import pixel

# bits that make a pixel unusable for the statistics (subset of maskhandling.PixelMask1 of the reference)
BAD_BITS = np.uint32(pixel.DO_NOT_USE | pixel.SATURATED | pixel.JUMP_DET | pixel.NO_LIN_CORR | pixel.NO_FLAT_FIELD
                     | pixel.NO_GAIN_VALUE | pixel.HOT | pixel.DEAD | pixel.REFERENCE_PIXEL)


def good_mask(dq):
    return (dq & BAD_BITS) == 0


def accumulate(moments, slope, err, dq):
    """moments: dict of f64 planes N, S1, S2, E1 updated in place with one realisation."""
    ok = good_mask(dq)
    x = np.where(ok, slope, 0.0).astype(np.float64)
    moments["N"] += ok
    moments["S1"] += x
    moments["S2"] += x * x
    moments["E1"] += np.where(ok, err, 0.0)
    return moments


def finalize(moments, ideal):
    N = moments["N"]
    with np.errstate(divide="ignore", invalid="ignore"):
        mean = moments["S1"] / N
        std = np.sqrt(np.clip(moments["S2"] / N - mean**2, 0.0, None))
        merr = moments["E1"] / N
    empty = N == 0
    out = np.stack([ideal, N, mean, std, mean - ideal, merr]).astype(np.float32)
    out[2:, empty] = -1000.0
    return out


def run(calibrator, slot, cal, nseeds=256, seed0=100, read_pattern=None, device="cpu"):
    """Generate + calibrate ``nseeds`` realisations (this rank's share) and return the 6 output planes."""
    import torch

    rp = synth.READ_PATTERN_8 if read_pattern is None else read_pattern
    ny, nx = cal["gain"]["data"].shape
    rate = synth.make_rate_image(ny, nx, seed0)
    seeds = sharding.scatter_items([seed0 + 10 * (j + 1) for j in range(nseeds)], device=device)
    m = {k: np.zeros((ny, nx), dtype=np.float64) for k in ("N", "S1", "S2", "E1")}
    for sd in seeds:
        ramp = synth.make_ramp(cal, read_pattern=rp, seed=sd, rate=rate)
        res = calibrator.calibrate(slot, ramp, want_groupdq=False)
        accumulate(m, res["slope"], np.hypot(res["err_read"], res["err_poisson"]), res["pixeldq"])
    planes = [torch.from_numpy(m[k]).to(device) for k in ("N", "S1", "S2", "E1")]
    sharding.allreduce_sum_(planes)
    for k, t in zip(("N", "S1", "S2", "E1"), planes):
        m[k] = t.cpu().numpy()
    flat = cal["flat"]["data"].astype(np.float64)
    nb = pars.nborder
    ideal = np.zeros((ny, nx))
    with np.errstate(divide="ignore", invalid="ignore"):
        ideal[nb:-nb, nb:-nb] = (rate / np.clip(flat, 0.1, 10))[nb:-nb, nb:-nb]
    return finalize(m, ideal)
