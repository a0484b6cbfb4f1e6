"""Oracle: statistics over many noise realisations.  Test infrastructure only.

numpy restatement of the reference's ``validation_tests/many_realizations.py:47-106``: ideal slope and its flip
(``:47-54``), the per-realisation stores (``:69-80``), the moment formulas and the -1000 sentinel (``:83-89``) and
the eight output planes (``:92-101``).  Pinned by the goldens ``harness_a`` / ``harness_b``, which
``tools/make_goldens.py harness`` makes by executing the reference's script itself on the same synthetic
realisations.
"""

import numpy as np

from . import post


def ideal_slope(truth, exptime, g_ideal, scanum, nside=4096, nb=4):
    big = np.zeros((nside, nside), dtype=np.float32)
    big[nb:-nb, nb:-nb] = truth / float(exptime) / g_ideal
    return big[:, ::-1] if scanum % 3 == 0 else big[::-1, :]


def statistics(realisations, ideal, nside=4096, nb=4, alias=True):
    """``realisations``: iterable of dicts with l1_first, l1_last (u16, full frame), data, err (f32), dq (u32) (active
    region).  ``alias``: the reference's ``images`` and ``err`` stacks are memory maps of the same file (``:58-59``), so
    every store into ``err`` also lands in ``images``.  Returns the (8, nside, nside) f32 stack."""
    act = (slice(nb, -nb), slice(nb, -nb))
    diffs, images, errs = [], [], []
    mom = np.zeros((3, nside - 2 * nb, nside - 2 * nb), dtype=np.float32)
    for r in realisations:
        diffs.append(r["l1_last"].astype(np.float32) - r["l1_first"].astype(np.float32))
        im, er = np.zeros((nside, nside), np.float32), np.zeros((nside, nside), np.float32)
        im[act], er[act] = r["data"], r["err"]
        errs.append(er)
        images.append(er if alias else im)
        w = np.logical_not(post.build_mask(r["dq"]))
        mom[0] += np.where(w, 1, 0.0)
        mom[1] += np.where(w, r["data"], 0.0)
        mom[2] += np.where(w, r["data"] ** 2, 0.0)
    with np.errstate(invalid="ignore"):
        mom[1:] /= mom[0] + 1e-25
        mom[2] = np.sqrt(np.clip(mom[2] - mom[1] ** 2, 0, None))
    mom[1:] = np.where(mom[0][None] > 0.1, mom[1:], -1000.0)
    big = np.zeros((3, nside, nside), dtype=np.float32)
    big[(slice(None),) + act] = mom
    return np.stack([ideal, np.median(np.stack(diffs), axis=0), np.median(np.stack(images), axis=0), big[0], big[1],
                     big[2], big[1] - ideal, np.median(np.stack(errs), axis=0)]).astype(np.float32)
