"""Timing experiment: the fused kernel with phases switched off (results invalid; timing only)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from romanimpreprocess_amd import _native
if os.environ.get("ALTLIB"):
    _native.LIB_PATH = os.path.join(os.path.dirname(_native.__file__), os.environ["ALTLIB"])
from romanimpreprocess_amd import pipeline, synth

NG = int(os.environ.get("NGROUPS", "8"))   # NGROUPS=16: BASELINE config 3
rp = synth.READ_PATTERN_8 if NG == 8 else synth.READ_PATTERN_16
N = 4096
KDT = np.float64 if os.environ.get("IPC64") == "1" else np.float32  # IPC64=1: f64 ipc4d coefficients
cal, ramp = synth.make_tiled_inputs(N, N, read_pattern=rp, p_order=8, seed=1, strip_rows=64, ipc_dtype=KDT)
cb = pipeline.Calibrator(device=0)
cb.ctx.set_option("chain2", int(os.environ.get("CHAIN2", "1")))
cb.load_caldir(0, cal)
pid, meta = cb.plan_for(rp, ramp["frame_time"])
dev = torch.device("cuda", 0)
g = ramp["groupdq"].copy()
g[0] |= 1
t = [torch.from_numpy(ramp["data"].view(np.int16)).to(dev), torch.from_numpy(ramp["amp33"].view(np.int16)).to(dev),
     torch.from_numpy(g).to(dev), torch.from_numpy(ramp["pixeldq"].view(np.int32)).to(dev)]
o = [torch.empty((N, N), dtype=torch.float32, device=dev) for _ in range(3)] + [
    torch.empty((N, N), dtype=torch.int32, device=dev), torch.empty((NG, N, N), dtype=torch.uint8, device=dev)]
torch.cuda.synchronize()


import inspect
KW = {"inputs_complete": True} if "inputs_complete" in inspect.signature(cb.calibrate_device).parameters else {}


def call():
    cb.calibrate_device(0, pid, NG, t[0].data_ptr(), True, t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(),
                        o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), o[4].data_ptr(), **KW)


def run(mask, n=10, batches=5):
    """best batch mean of the fused kernel's duration (ms)"""
    cb.ctx.set_option("chain_dbg", mask)
    for _ in range(3):
        call()
    cb.synchronize()
    best = 1e9
    for _ in range(batches):
        cb.ctx.profile(True)
        cb.ctx.profile_read()
        for _ in range(n):
            call()
        ms, nc = cb.ctx.profile_read()
        cb.ctx.profile(False)
        best = min(best, ms[1] / nc)
    return best


if os.environ.get("CHAIN3", "0") == "1":
    names3 = {0: "full", 1: "no fit", 2: "no groupdq stores", 4: "no plane stores", 6: "no stores", 7: "no fit, no stores", 8: "no F/T"}
    for m in [int(x) for x in sys.argv[1:]] or [0, 1, 2, 4, 6, 7, 8]:
        print(f"dbg={m:3d} {names3.get(m, ''):28s} {run(m):8.3f} ms", flush=True)
    sys.exit(0)
names = {1024: "fit role's second reads hit one cached row (narrow forms)", 0: "full", 1: "no C (O1)", 2: "no E-ipc (O2)", 4: "no fit", 8: "no saturated path", 16: "no legendre",
         32: "no lin slow branch", 3: "no ipc", 7: "no ipc, no fit", 23: "no ipc/fit/legendre", 12: "no fit/no sat"}
for m in [int(x) for x in sys.argv[1:]] or [0, 1, 2, 3, 4, 8, 12, 16, 32, 7, 23]:
    print(f"dbg={m:3d} {names.get(m, ''):28s} {run(m):8.3f} ms", flush=True)
