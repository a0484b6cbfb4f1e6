"""BASELINE config 5 end to end on one GPU: N noise realisations of one 4096 x 4096 x 8-group scene generated on the device,
calibrated, stacked in HBM and reduced to the eight statistics planes.\nusage: python tools/gpu_checks/many_realizations_fullsize.py [N=256] [generator=hip|device]   (hip = the reference's synthesis steps as HIP\nkernels, device = the torch recipe of synth_gpu)
Progress goes to gpurun_out/many_realizations_fullsize.log."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401

from romanimpreprocess_amd import pipeline, synth, synth_gpu
from romanimpreprocess_amd.harness import many_realizations as mr

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
gen = sys.argv[2] if len(sys.argv) > 2 else "hip"
os.makedirs("gpurun_out", exist_ok=True)
log = open(os.path.join("gpurun_out", "many_realizations_fullsize.log"), "w")


def say(*a):
    print(*a, file=log, flush=True)
    print(*a, flush=True)


t0 = time.time()
rp = synth.READ_PATTERN_8
cal = synth_gpu.make_caldir(4096, 4096, read_pattern=rp, p_order=8, seed=1001)
cb = pipeline.Calibrator(device=0)
cb.load_caldir(0, cal)
say("CALDIR set generated and resident after", round(time.time() - t0, 1), "s")
tm = {}
t1 = time.time()
planes = mr.run(cb, 0, cal, nseeds=n, seed0=100, read_pattern=rp, generator=gen, timings=tm)
wall = time.time() - t1
good = planes[3][4:-4, 4:-4]
bias = planes[6][4:-4, 4:-4][good > n // 2]
ratio = (planes[5] / np.maximum(planes[7], 1e-9))[4:-4, 4:-4][good > n // 2]
out = {"realisations": n, "generator": gen, "wall_s": wall, "s_per_realisation": wall / n, **tm,
       "median_unmasked_count": float(np.median(good)), "median_bias_DN_per_s": float(np.median(bias)),
       "median_std_over_median_err": float(np.median(ratio))}
say(json.dumps(out))
