"""Wall time of the host-array boundary (numpy in, numpy out over PCIe): python tools/gpu_checks/host_path_timing.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from romanimpreprocess_amd import pipeline, synth

rp = synth.READ_PATTERN_8
N = 4096
cal, ramp = synth.make_tiled_inputs(N, N, read_pattern=rp, p_order=8, seed=1, strip_rows=128)
cb = pipeline.Calibrator(device=0)
t0 = time.perf_counter()
cb.load_caldir(0, cal)
cb.synchronize()
print(f"CALDIR upload {time.perf_counter() - t0:.2f} s")
for want_gdq in (True, False):
    for flag_sat in (False,):
        ts = []
        for i in range(6):
            t0 = time.perf_counter()
            res = cb.calibrate(0, ramp, want_groupdq=want_gdq)
            ts.append(time.perf_counter() - t0)
        print(f"calibrate(host arrays, want_groupdq={want_gdq}): best {min(ts)*1e3:.1f} ms, median {sorted(ts)[len(ts)//2]*1e3:.1f} ms  -> {1/min(ts):.1f} ramps/s")

# the same with page-locked arrays on both sides
pin = {k: cb.pinned_empty(v.shape, v.dtype) for k, v in ramp.items() if isinstance(v, np.ndarray)}
for k in pin:
    pin[k][...] = ramp[k]
pin["groupdq"][0] |= 1
ramp_p = dict(ramp, **pin)
out = {"slope": cb.pinned_empty((N, N), np.float32), "err_read": cb.pinned_empty((N, N), np.float32),
       "err_poisson": cb.pinned_empty((N, N), np.float32), "pixeldq": cb.pinned_empty((N, N), np.uint32),
       "groupdq": cb.pinned_empty((8, N, N), np.uint8)}
for want_gdq in (True, False):
    ts = []
    for i in range(6):
        t0 = time.perf_counter()
        res2 = cb.calibrate(0, ramp_p, want_groupdq=want_gdq, out=out)
        ts.append(time.perf_counter() - t0)
    print(f"calibrate(page-locked arrays, want_groupdq={want_gdq}): best {min(ts)*1e3:.1f} ms, median {sorted(ts)[len(ts)//2]*1e3:.1f} ms  -> {1/min(ts):.1f} ramps/s")
assert np.array_equal(res2["slope"], res["slope"], equal_nan=True) and np.array_equal(res2["pixeldq"], res["pixeldq"])

# a batch of page-locked ramps through rip_calibrate_batch (upload / chain / download overlapped)
nb_ = 8
outs = [out] + [{k: cb.pinned_empty(v.shape, v.dtype) for k, v in out.items()} for _ in range(nb_ - 1)]
for want_gdq in (True, False):
    ts = []
    for i in range(3):
        t0 = time.perf_counter()
        many = cb.calibrate_many(0, [ramp_p] * nb_, want_groupdq=want_gdq, out=outs)
        ts.append((time.perf_counter() - t0) / nb_)
    print(f"calibrate_many({nb_} page-locked ramps, want_groupdq={want_gdq}): best {min(ts)*1e3:.1f} ms per ramp -> {1/min(ts):.1f} ramps/s")
assert np.array_equal(many[-1]["slope"], res["slope"], equal_nan=True) and np.array_equal(many[3]["pixeldq"], res["pixeldq"])
