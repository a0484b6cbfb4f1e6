"""Moment ratios of the ramp-fit slope under Poisson noise (host, numpy).  Same call surface and results as the reference's
``L1_to_L2/GalPoisson/find_tilnus.py`` (``raw_weights`` :12-41, ``get_tilde_nus`` :44-77).

Model: every raw read j >= 1 adds an independent Poisson increment (rate I per frame) to all later reads; group k is the mean
of reads a_k .. a_k + N_k - 1 and the slope is sum_k W_k (group k).  The increment of read j therefore enters the slope with
the coefficient c_j = sum_k W_k (number of reads of group k at or after j) / N_k, and the slope's cumulants are I sum_j c_j^p.
"""

import numpy as np


def raw_weights(N_beta, a_beta):
    """(groups x reads) averaging matrix: entry (k, j) = 1 / N_k for the reads j of group k, else 0."""
    counts, first = np.asarray(N_beta, dtype=int), np.asarray(a_beta, dtype=int)
    if counts.shape != first.shape:
        raise AssertionError("N_beta and a_beta must have the same length")
    reads = np.arange(int(np.max(first + counts)))
    member = (reads[None, :] >= first[:, None]) & (reads[None, :] < (first + counts)[:, None])
    return member / counts[:, None].astype(float)


def get_tilde_nus(N_beta, a_beta, W):
    """(tilnu_21, tilnu_31, tilnu_41, tilnu_42) for the weight vector ``W`` (one weight per group), in units of frames."""
    avg = raw_weights(N_beta, a_beta)
    # reads of group k at or after read j, over N_k: the reversed running sum of the averaging matrix along the read axis
    tail = np.flip(np.cumsum(np.flip(avg, axis=1), axis=1), axis=1)
    coeff = np.asarray(W, dtype=float) @ tail[:, 1:]          # read 0 carries no increment
    k2, k3, k4 = (np.sum(coeff**p) for p in (2, 3, 4))
    gauss4 = 3.0 * k2 * k2
    return k2, k3 - 3.0 * k2 * k2, k4 - 10.0 * k2 * k3 - k2 * gauss4 + 18.0 * k2**3, gauss4
