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
