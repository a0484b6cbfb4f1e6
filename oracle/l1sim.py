"""Oracle: Level-1 synthesis.  Test infrastructure only.

numpy restatements, with every random deviate as an argument, of the simulation-side path that turns an electron-count image
into a raw Level-1 exposure:
  * ``sim_to_isim.make_l1_fullcal``         (``from_sim/sim_to_isim.py:163-262``): reset noise, bias offset, the per-read
    IPC + inverse-linearity model, read noise, bias correction, rounding;
  * ``sim_to_isim.fill_in_refdata_and_1f``  (``:306-403``): reference pixels from the dark file plus white / reset / 1/f noise,
    the correlated noise of the 32 channels, the reference output (amp33), clipping to u16;
  * the ``EXTRACT_REF`` block of ``Image2D.simulate`` (``:711-730``).
Pinned: tools/make_goldens.py takes the two functions from the reference's file with ``ast`` and executes them as they stand
(numpy + the reference's own ``ipc_linearity.IL`` class; galsim's deviate generator replaced by "fill from the normals handed
in", asdf by in-memory trees); tests/test_oracle_golden.py holds the restatements below to those outputs bit for bit.
NOT pinned (``romanisim`` is a dependency absent from the reference tree and from this image, ``romanisim>=0.8`` in its
pyproject): ``apportion_counts_to_resultants``, ``add_read_noise_to_resultants`` and ``read_pattern_to_tij`` restate the
published algorithm of ``romanisim/l1.py`` -- binomial apportioning of the integer counts over the read times, the model's
``apply(..., electrons=True)`` per read, resultant = mean of its reads; cosmic rays and persistence are not restated -- and
the golden run uses these same restatements in romanisim's place.
"""

import numpy as np

from . import ipc, linearity, noise


def read_pattern_to_tij(read_pattern, read_time):
    """Time stamps of the reads of every resultant, seconds (romanisim.l1.read_pattern_to_tij)."""
    return [read_time * np.array(reads) for reads in read_pattern]


def binomial_shares(counts, tij, rng):
    """Cumulative electrons at every read, (nreads, ny, nx) i4: each read takes Binomial(remaining, dt / time left) of the
    counts not yet collected, so the last read holds all of them."""
    total = np.clip(counts, 0, 2 * 10**9).astype("i4")
    t_end = max(float(np.max(t)) for t in tij)
    so_far = np.zeros(total.shape, dtype="i4")
    t_prev, out = 0.0, []
    for times in tij:
        for t in times:
            left = t_end - t_prev
            share = (float(t) - t_prev) / left if left > 0 else 1.0
            so_far = so_far + rng.binomial(total - so_far, min(max(share, 0.0), 1.0)).astype("i4")
            t_prev = float(t)
            out.append(so_far.copy())
    return np.stack(out)


def apportion_counts_to_resultants(reads_e, tij, model):
    """Resultants (ngrp, ny, nx) f4 from the cumulative electrons of every read: model(reads_e[r]) is the raw DN of read r,
    accumulated in f4 and divided by the number of reads."""
    out = np.zeros((len(tij),) + reads_e.shape[1:], dtype="f4")
    r = 0
    for i, times in enumerate(tij):
        acc = np.zeros(reads_e.shape[1:], dtype="f4")
        for _ in times:
            acc += model(reads_e[r])
            r += 1
        out[i] = acc / len(times)
    return out


def add_read_noise_to_resultants(resultants, tij, read_noise, normals):
    """resultants += normal * read_noise / sqrt(reads in the resultant): the f4 product is divided by an f8 array, and the sum
    rounded back into the f4 resultants."""
    kick = np.array(normals, dtype="f4") * read_noise
    kick = kick / np.array([len(t) ** 0.5 for t in tij]).reshape(-1, 1, 1)
    resultants += kick
    return resultants


def make_l1_fullcal(counts, read_pattern, cal, read_time, normals_reset, reads_e, normals_read, nb=4):
    """``make_l1_fullcal``: (rounded resultants f4 (ngrp, na, na), reset-noise image in electrons).  ``counts`` only gives
    shape and dtype here: its apportioned form ``reads_e`` comes from ``binomial_shares``."""
    inner = (slice(nb, -nb), slice(nb, -nb))
    start = np.array(normals_reset, dtype=counts.dtype)
    start *= cal["read"]["resetnoise"][inner]
    start *= cal["gain"]["data"][inner]
    if "biascorr" in cal:
        start -= float(cal["biascorr"]["t0"]) * cal["dark"]["dark_slope"][inner] / cal["gain"]["data"][inner]
    tij = read_pattern_to_tij(read_pattern, read_time)
    lin = cal["linearitylegendre"]
    kern = cal["ipc4d"]["data"] if "ipc4d" in cal else None

    def model(electrons):
        return linearity.il_apply(electrons, kern, cal["gain"]["data"], lin["data"], lin["Smin"], lin["Smax"], lin["Sref"],
                                  start_e=start, electrons=True)

    res = apportion_counts_to_resultants(reads_e, tij, model)
    res = add_read_noise_to_resultants(res, tij, cal["read"]["data"][inner], normals_read)
    if "biascorr" in cal:
        res += cal["biascorr"]["data"]
    res[:, :, :] = np.round(res)
    return res, start


def embed(resultants, ny, nx, nb=4):
    """The u16 cube ``romanisim.l1.make_asdf`` builds around the resultants (zero border; values clipped to the u16 range
    here, where romanisim casts)."""
    cube = np.zeros((resultants.shape[0], ny, nx), dtype=np.uint16)
    cube[:, nb:-nb, nb:-nb] = np.clip(resultants, 0, 65535).astype(np.uint16)
    return cube


def fill_in_refdata_and_1f(im, cal, tij, normals, frames=None, white33=None, amp33=None, nb=4, channelwidth=128):
    """``fill_in_refdata_and_1f`` in place on ``im`` (ngrp, ny, nx) u16 and ``amp33`` (ngrp, ny, channelwidth) u16.
    ``normals`` (ngrp+1, ny, nx) f4; ``frames`` (ngrp, 34, ny, channelwidth) f4 1/f frames in the order the reference draws
    them per group (common, channels 0..31, reference output) or None (no banding); ``white33`` (ngrp, ny, channelwidth) f4."""
    ngrp, ny, nx = im.shape
    work = np.array(normals, dtype=np.float32)
    rd = cal["read"]
    work[:-1] *= rd["data"][None]
    work[-1] *= rd["resetnoise"]
    for j in range(len(tij)):
        work[j] /= len(tij[j]) ** 0.5
    work[:-1] += work[-1][None]
    dark = cal["dark"]["data"]
    work[:-1] += dark[dark.shape[0] - ngrp:]
    work[:-1, nb:ny - nb, nb:nx - nb] = im[:, nb:ny - nb, nb:nx - nb].astype(np.float32)
    info = rd.get("amp33") if amp33 is not None else None
    if frames is not None:
        u_pink, c_pink = float(rd["anc"]["U_PINK"]), float(rd["anc"]["C_PINK"])
        for j in range(len(tij)):
            common = frames[j, 0] * c_pink
            for ch in range(32):
                stripe = frames[j, 1 + ch] * u_pink + common
                if ch % 2 == 1:
                    stripe = stripe[:, ::-1]
                work[j, :, channelwidth * ch:channelwidth * (ch + 1)] += (stripe / len(tij[j]) ** 0.5).astype(np.float32)
            if info is not None and info["valid"]:
                white = np.array(white33[j], dtype=np.float32) * info["std"]
                pink = info["RU_PINK"] * frames[j, 33] + info["M_PINK"] * common
                level = info["med"] + (white + pink) / len(tij[j]) ** 0.5
                amp33[j] = level.astype(np.int64).astype(amp33.dtype)    # a C cast: towards zero, modulo 2**16
    im[:, :, :] = np.clip(np.round(work[:-1]), 0, 2**16 - 1).astype(im.dtype)


def extract_ref(data, offset):
    """``EXTRACT_REF``: (reference read, the later resultants minus it plus ``offset``, clipped to u16)."""
    ref = data[0].copy()
    shift = data[0].astype(np.int32) - offset
    rest = np.clip(data[1:].astype(np.int32) - shift[None], 0, 65535).astype(np.uint16)
    return ref, rest


def noise_frames(normals, rows, width):
    """1/f frames from (nframes, 4*rows*width) deviates."""
    return np.stack([noise.noise_1f_frame(d, rows, width) for d in normals])
