"""Oracle: noise layers.  Test infrastructure only.

numpy restatements, with the random deviates as arguments, of
  * the white read-noise injection of ``gen_noise_image.make_noise_cube`` (``gen_noise_image.py:120-134``),
  * its resampled-Poisson branch (``:285-331``),
  * ``sim_to_isim.noise_1f_frame`` (``sim_to_isim.py:265-303``).
The reference draws its deviates from galsim generators inside these code paths (not importable offline, streams not
reproducible): PARITY UNPINNED for the random streams only.  The ARITHMETIC is pinned (round 3 for the first two): the
reference's functions are numpy-only apart from the draws, so tools/make_goldens.py takes them from their files with ``ast``
and EXECUTES them with deviate stand-ins that hand out recorded values -- ``noise_1f_frame`` (fixture noise_1f_frame.npz) and
the whole layer loop ``make_noise_cube`` (fixture noise_arith.npz: the injected cubes of 'Ra' / 'R' layers and the 'Pr',
'Pb1r', 'RaPr' layers) -- and tests/test_oracle_golden.py holds these restatements to the same arrays bit for bit.
"""

import numpy as np


def inject_read_noise(cube, read_noise, read_pattern, normals, nb=4):
    """u16 cube + sigma_read / sqrt(N_k) * normal per group k on the active region, clipped to the u16 range and rounded
    half to even; ``normals`` (ngrp, ny-2nb, nx-2nb) f32."""
    out = cube.copy()
    inner = (slice(nb, -nb), slice(nb, -nb))
    for k, reads in enumerate(read_pattern):
        level = out[k][inner].astype(np.float32)
        kick = np.array(normals[k], dtype=np.float32)
        kick *= read_noise[inner] / np.sqrt(len(reads))      # f32 array / np.float64 scalar -> f64, stored back as f32
        level += kick
        out[k][inner] = np.round(np.clip(level, 0, 2**16 - 1)).astype(out.dtype)
    return out


def poisson_resample(diff, skylevel, gain, frame_time, read_pattern, weight_rows, has_row, endslice, deviates):
    """``diff`` plus one resampled-Poisson realisation.  ``deviates`` (nreads, ny, nx) f64 Poisson draws of mean
    clip(skylevel * gain * frame_time, 0); ``weight_rows[es]`` = ramp-fit weights of a ramp ending at group es where
    ``has_row[es]``."""
    ngrp = len(read_pattern)
    electrons = np.clip(skylevel * gain * frame_time, 0.0, None)
    running = np.zeros(electrons.shape, dtype=np.float32)
    change = np.zeros((ngrp,) + electrons.shape, dtype=np.float32)
    for read in range(read_pattern[-1][-1] + 1):
        draw = np.array(deviates[read], dtype=np.float64)
        draw -= electrons
        draw /= gain
        running += draw
        for j, reads in enumerate(read_pattern):
            if read in reads:
                change[j] += running / len(reads)
    out = diff
    for es in range(ngrp):
        if not has_row[es]:
            continue
        for j in range(ngrp):
            out += np.where(endslice == es, weight_rows[es][j] * change[j], 0.0)
    return out


def noise_1f_frame(deviates, rows, width):
    """One (rows, width) f32 frame of 1/f noise from 4*rows*width standard normal deviates."""
    n = 2 * rows * width
    f = np.linspace(0, 1 - 1.0 / n, n)
    f[n // 2:] -= 1.0
    amplitude = (1.0e-99 + np.abs(f * n)) ** (-0.5)
    amplitude[0] = 0.0
    spectrum = np.zeros(n, dtype=np.complex128)
    spectrum[:] = deviates[:n]
    spectrum[:] += 1j * deviates[n:]
    spectrum *= amplitude
    series = np.fft.fft(spectrum).real[: n // 2] / np.sqrt(2.0)
    series -= np.mean(series)
    return series.reshape(rows, width).astype(np.float32)
