"""Moment ratios of the ramp-fit slope under Poisson noise (host, numpy): drop-in for the reference's
``L1_to_L2/GalPoisson/find_tilnus.py`` (``raw_weights`` :12-41, ``get_tilde_nus`` :44-77)."""

import numpy as np


def raw_weights(N_beta, a_beta):
    """Matrix L (M groups x N reads) averaging raw reads into groups: group k = mean of reads a_beta[k] .. a_beta[k]+N_beta[k]-1."""
    N_beta, a_beta = np.asarray(N_beta), np.asarray(a_beta)
    assert len(N_beta) == len(a_beta)
    nreads = np.max(a_beta + N_beta)
    L = np.zeros((len(N_beta), nreads))
    for k in range(len(N_beta)):
        L[k, a_beta[k]:a_beta[k] + N_beta[k]] = 1.0 / N_beta[k]
    return L


def get_tilde_nus(N_beta, a_beta, W):
    """(tilnu_21, tilnu_31, tilnu_41, tilnu_42) of the slope sum_k W_k (group k) when every read adds an independent Poisson
    increment: the weight of the increment of read j is the tail sum of W L from read j on."""
    L = raw_weights(N_beta, a_beta)
    T = np.cumsum(L[:, ::-1], axis=1)[:, ::-1]
    WT = np.dot(W, T[:, 1:])
    nu_21 = np.sum(WT**2)
    nu_31 = np.sum(WT**3)
    nu_41 = np.sum(WT**4)
    nu_42 = 3 * nu_21**2
    return nu_21, nu_31 - 3 * nu_21**2, nu_41 - 10 * nu_21 * nu_31 - nu_21 * nu_42 + 18 * nu_21**3, nu_42
