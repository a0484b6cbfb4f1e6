"""apportion_kernel alone: time per 4096 x 4096 frame of 35 reads for several mean totals (device Poisson increments)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from romanimpreprocess_amd import _native, synth
from romanimpreprocess_amd.from_sim import sim_to_isim

rp = synth.READ_PATTERN_8
ny = nx = 4096
cal = synth.make_caldir(264, 512, read_pattern=rp, p_order=3, seed=9)
ctx = _native.default_context(0)
s = sim_to_isim.L1Synth(cal, rp, synth.FRAME_TIME, ctx=ctx)
dev = s.dev
na = (ny - 8, nx - 8)
from romanimpreprocess_amd import synth as _s
rate = _s.make_rate_image(ny, nx, 100)
print("rate image: median %.3f, 99th percentile %.1f, max %.1f DN/s" % (np.median(rate), np.percentile(rate, 99), rate.max()))
for total in (1.0, 35.0, 160.0, 340.0, 360.0, 3500.0, "rate"):
    if total == "rate":
        c = torch.from_numpy(np.ascontiguousarray((rate[4:-4, 4:-4] * 1.5 * 106.4).astype(np.float32))).to(dev)
    else:
        c = torch.full(na, float(total), dtype=torch.float32, device=dev)
    out = torch.empty((35,) + na, dtype=torch.int32, device=dev)
    tr = np.ascontiguousarray(s.t_reads, dtype=np.float64)
    def call():
        ctx.check(ctx.lib.rip_synth_apportion(ctx.h, c.data_ptr(), na[0], na[1], 1, 35, tr.ctypes.data, 7, out.data_ptr()))
    torch.cuda.synchronize()
    for _ in range(3):
        call()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        call()
    ctx.synchronize()
    print(f"mean total {total}: {(time.perf_counter() - t0) * 100:.2f} ms per frame", flush=True)

# the many-realisations harness' own mean counts (BASELINE config 5: scene + dark current of the bench's calibration set)
from romanimpreprocess_amd import synth_gpu
cal5 = synth_gpu.make_caldir(ny, nx, read_pattern=rp, p_order=8, seed=1001, ipc_dtype=np.float32, device=0)
s5 = sim_to_isim.L1Synth(cal5, rp, synth.FRAME_TIME, ctx=ctx)
per_s = (np.asarray(rate, dtype=np.float64)[4:-4, 4:-4] + cal5["dark"]["dark_slope"][4:-4, 4:-4]) * cal5["gain"]["data"][4:-4, 4:-4]
cm = np.clip(per_s * s5.t_reads[-1], 0.0, None).astype(np.float32)
print("harness counts: percentiles 1, 50, 99, 99.9:", np.percentile(cm, [1, 50, 99, 99.9]), "per read / 34")
c = torch.from_numpy(cm).to(dev)
torch.cuda.synchronize()
for _ in range(3):
    call()
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    call()
ctx.synchronize()
print(f"harness counts: {(time.perf_counter() - t0) * 100:.2f} ms per frame", flush=True)

# resultants_kernel on the same two scenes (does a frame's hot pixels cost its waves evaluations?)
for label, cm_ in (("smooth scene (no dark current, flat gain)", (rate[4:-4, 4:-4] * 1.5 * 106.4).astype(np.float32)), ("harness counts", cm),
                   ("flat 72 electrons", np.full(na, 72.0, dtype=np.float32))):
    c = torch.from_numpy(np.ascontiguousarray(cm_)).to(dev)
    torch.cuda.synchronize()
    reads_e = s5.apportion(c, 7, poisson=True)
    ctx.synchronize()
    for _ in range(2):
        out_c = s5.resultants(reads_e, 7)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out_c = s5.resultants(reads_e, 7)
    ctx.synchronize()
    print(f"resultants, {label}: {(time.perf_counter() - t0) * 200:.2f} ms per frame", flush=True)
