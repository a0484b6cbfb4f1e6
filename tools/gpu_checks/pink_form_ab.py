"""1/f frames of a full exposure (272 frames of 4096 x 128): the library's transform (option pink_form = 0) against the hand-written
two-pass one (pink_fft.h) -- time per set and agreement of the frames.   python tools/gpu_checks/pink_form_ab.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from romanimpreprocess_amd import _native   # noqa: E402

ctx = _native.default_context(0)
dev = torch.device("cuda", ctx.device)
rows, width, n = 4096, 128, 272
outs = {}
for form in (0, -1, 0, -1):
    ctx.set_option("pink_form", form)
    out = torch.empty((n, rows, width), dtype=torch.float32, device=dev)
    for rep in range(3):
        ctx.synchronize()
        t0 = time.perf_counter()
        ctx.check(ctx.lib.rip_synth_noise_1f(ctx.h, rows, width, n, 31, 7, out.data_ptr()))
        ctx.synchronize()
        dt = time.perf_counter() - t0
    print(f"pink_form {form:2d}: {1e3 * dt:.2f} ms for {n} frames", flush=True)
    outs[form] = out
a, b = outs[0], outs[-1]
d = (a.double() - b.double()).abs()
print(f"frames: std {float(a.double().std()):.4f}; max |difference| {float(d.max()):.3e}; fraction of samples that differ {float((a != b).double().mean()):.2e}")
