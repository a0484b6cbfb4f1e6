"""Wall time per ramp of the chain with and without the device dq-init + saturation flagging (4096x4096x8)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from romanimpreprocess_amd import pipeline, synth

rp = synth.READ_PATTERN_8
N = 4096
cal, ramp = synth.make_tiled_inputs(N, N, read_pattern=rp, p_order=8, seed=1, strip_rows=64)
cb = pipeline.Calibrator(device=0)
cb.load_caldir(0, cal)
pid, meta = cb.plan_for(rp, ramp["frame_time"])
dev = torch.device("cuda", 0)
g = ramp["groupdq"].copy()
g[0] |= 1
t = [torch.from_numpy(ramp["data"].view(np.int16)).to(dev), torch.from_numpy(ramp["amp33"].view(np.int16)).to(dev),
     torch.from_numpy(g).to(dev), torch.from_numpy(ramp["pixeldq"].view(np.int32)).to(dev)]
o = [torch.empty((N, N), dtype=torch.float32, device=dev) for _ in range(3)] + [
    torch.empty((N, N), dtype=torch.int32, device=dev), torch.empty((8, N, N), dtype=torch.uint8, device=dev)]
torch.cuda.synchronize()


def run(sat, n=20):
    def call():
        cb.calibrate_device(0, pid, 8, t[0].data_ptr(), True, t[1].data_ptr(), None if sat else t[2].data_ptr(), t[3].data_ptr(),
                            o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), o[4].data_ptr(),
                            flag_saturation=sat)
    for _ in range(3):
        call()
    cb.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        call()
    cb.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


print(f"flags given: {run(False):.3f} ms per ramp;  flags made on the device: {run(True):.3f} ms per ramp")
