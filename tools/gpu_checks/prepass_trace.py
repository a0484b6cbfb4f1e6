"""Per-kernel durations of the reference-pixel pre-pass when it runs ALONE (overlap off): run under
   rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o run -- python3 tools/gpu_checks/prepass_trace.py
and read <dir>/**/run_kernel_stats.csv (tools/gpu_checks/prepass_trace.sh does that and prints the pre-pass rows)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from romanimpreprocess_amd import pipeline, synth

rp = synth.READ_PATTERN_8
N = 4096
cal, ramp = synth.make_tiled_inputs(N, N, read_pattern=rp, p_order=8, seed=1, strip_rows=64)
cb = pipeline.Calibrator(device=0)
cb.ctx.set_option("overlap", 0)
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
for _ in range(20):
    cb.calibrate_device(0, pid, 8, t[0].data_ptr(), True, t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(),
                        o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), o[4].data_ptr())
cb.synchronize()
print("done")
