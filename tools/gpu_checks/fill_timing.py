"""Host time and device time of L1Synth.fill (fresh reference pixels + correlated noise) on a full frame."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from romanimpreprocess_amd import _native, synth, synth_gpu
from romanimpreprocess_amd.from_sim.sim_to_isim import L1Synth

rp = synth.READ_PATTERN_8
N = 4096
cal = synth_gpu.make_caldir(N, N, read_pattern=rp, p_order=8, seed=1001)
ctx = _native.default_context(0)
s = L1Synth({k: cal[k] for k in ("read", "gain", "dark")}, rp, 1.0, ctx=ctx, nb=4)
dev = s.dev
cube = torch.zeros((8, N, N), dtype=torch.int16, device=dev)
a33 = torch.zeros((8, N, 128), dtype=torch.int16, device=dev)
torch.cuda.synchronize()
for i in range(6):
    t0 = time.perf_counter()
    s.fill(cube, a33, 100 + i, banding=True)
    t1 = time.perf_counter()
    ctx.synchronize()
    t2 = time.perf_counter()
    print(f"fill {i}: call {1e3 * (t1 - t0):8.2f} ms, until done {1e3 * (t2 - t0):8.2f} ms", flush=True)
for i in range(3):
    t0 = time.perf_counter()
    s.fill(cube, a33, 200 + i, banding=False)
    t1 = time.perf_counter()
    ctx.synchronize()
    t2 = time.perf_counter()
    print(f"fill without banding {i}: call {1e3 * (t1 - t0):8.2f} ms, until done {1e3 * (t2 - t0):8.2f} ms", flush=True)
