"""Full-frame timing of the noise-layer kernels (host arrays in and out): python tools/gpu_checks/noise_timing.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from romanimpreprocess_amd import _native, synth
from romanimpreprocess_amd.L1_to_L2 import gen_noise_image as gni

ctx = _native.default_context(0)
rp = synth.READ_PATTERN_8
N, na = 4096, 4088
rng = np.random.default_rng(1)
data = rng.integers(5000, 40000, size=(8, N, N), dtype=np.uint16)
read = np.full((N, N), 7.0, np.float32)
for _ in range(2):
    t0 = time.perf_counter()
    out = gni.inject_read_noise(data, read, rp, seed=1, layer=0, ctx=ctx)
    dt = time.perf_counter() - t0
print(f"read-noise injection, 4096x4096x8 u16 cube, device deviates: {dt*1e3:.1f} ms (host arrays in and out)")
sky_ = (0.3 + rng.random((na, na))).astype(np.float32)
gain = np.full((na, na), 1.5, np.float32)
w = np.zeros((8, 8), np.float32)
w[-1] = np.linspace(-1, 1, 8)
has = np.zeros(8, np.uint8)
has[-1] = 1
es = np.full((na, na), 7, np.int8)
for _ in range(2):
    diff = np.zeros((na, na), np.float32)
    t0 = time.perf_counter()
    gni.poisson_resample(diff, sky_, gain, 3.04, rp, w, has, es, seed=2, layer=0, ctx=ctx)
    dt = time.perf_counter() - t0
print(f"resampled Poisson layer, 4088x4088, 35 reads, device deviates: {dt*1e3:.1f} ms (host arrays in and out); std {diff.std():.4f}")
