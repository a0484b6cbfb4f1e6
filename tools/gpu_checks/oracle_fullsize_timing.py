"""How long does the numpy oracle take on a full 4096 x 4096 frame (f64 ipc4d; 16 groups)?  Progress goes to gpurun_out/."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401  (before libromanhip)

import oracle
from romanimpreprocess_amd import synth, synth_gpu

log = open(os.path.join("gpurun_out", "oracle_fullsize.log"), "w")


def say(*a):
    print(*a, file=log, flush=True)
    print(*a, flush=True)


for name, rp, kdt in (("k64", synth.READ_PATTERN_8, np.float64), ("g16", synth.READ_PATTERN_16, np.float32)):
    t = time.time()
    cal = synth_gpu.make_caldir(4096, 4096, read_pattern=rp, p_order=8, seed=1001, ipc_dtype=kdt)
    ramp = synth_gpu.make_ramp(cal, read_pattern=rp, seed=1)
    say(name, "generated in", round(time.time() - t, 1))
    t = time.time()
    with np.errstate(all="ignore"):
        ref = oracle.calibrate_arrays(ramp, cal, progress=say) if "progress" in oracle.calibrate_arrays.__code__.co_varnames else oracle.calibrate_arrays(ramp, cal)
    say(name, "oracle", round(time.time() - t, 1))
