"""Diagnostic: cycles per row step of every role of the wave-specialised fused kernel (needs libromanhip_stamp.so built with
-DCH_STAMP: bash tools/gpu_checks/build_flags_variant.sh stamp -DRIP_TIMING_BUILD -DCH_STAMP; ROMANHIP_ALLOW_TIMING_BUILD=1).  NGROUPS=16 / IPC64=1 select the variant."""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from romanimpreprocess_amd import _native

_native.LIB_PATH = os.path.join(REPO, "romanimpreprocess_amd", "libromanhip_stamp.so")
import numpy as np
import torch

from romanimpreprocess_amd import pipeline, synth

NG = int(os.environ.get("NGROUPS", "8"))
rp = synth.READ_PATTERN_8 if NG == 8 else synth.READ_PATTERN_16
KDT = np.float64 if os.environ.get("IPC64") == "1" else np.float32
SPLIT = 0   # (the split form -- four roles on half the groups each, profiles/r03_summary.md -- was not kept)
N = 4096
if os.environ.get("TILED") == "1":   # the homogeneous tiled frame of round 1 (many more saturated / jump pixels per wave)
    cal, ramp = synth.make_tiled_inputs(N, N, read_pattern=rp, p_order=8, seed=1, strip_rows=64, ipc_dtype=KDT)
else:                                # the bench's non-periodic frame
    from romanimpreprocess_amd import synth_gpu
    cal = synth_gpu.make_caldir(N, N, read_pattern=rp, p_order=8, seed=1000, ipc_dtype=KDT, device=0)
    ramp = synth_gpu.make_ramp(cal, read_pattern=rp, seed=1, device=0)
cb = pipeline.Calibrator(device=0)
lib = cb.ctx.lib
lib.rip_chain_stamps.restype = C.c_int
lib.rip_chain_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
lib.rip_chain_stamps_n.restype = C.c_int
lib.rip_chain_stamps_n.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
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
out9 = (C.c_double * 9)()
lib.rip_chain_stamps(cb.ctx.h, out9)  # allocates the buffer


def call():
    cb.calibrate_device(0, pid, NG, t[0].data_ptr(), True, t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(),
                        o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), o[4].data_ptr())


split = SPLIT and (NG == 16 or KDT is np.float64)
two_per_cu = (NG == 8 and KDT is np.float32)
# waves per workgroup: 256-column form 8; the wide ring-dropping forms 12 (384 columns), f64 ipc4d x 16 groups 8 (256 columns)
nw = 16 if split else (8 if two_per_cu or (NG == 16 and KDT is np.float64) else 12)
nstrips = 17 if nw == 8 else 11
nwg = 504 if two_per_cu else (256 // nstrips) * nstrips
steps = (4096 + (nwg // nstrips) - 1) // (nwg // nstrips) + 6
roles = ["ingest-lo", "ingest-hi", "fit-lo", "fit-hi"] if split else ["ingest", "fit"]
lab_i = ["-", "-", "A (S1)", "barrier 1", "-", "C + loads (S2)", "-", "barrier 2", "-"]
lab_f = ["issue loads", "-", "O2 + F first half (S1)", "barrier 1", "-", "F second half + T (S2)", "ring reads", "barrier 2", "-"]
out = (C.c_double * (nw * 9))()
for mask in [int(x) for x in sys.argv[1:]] or [0]:
    cb.ctx.set_option("chain_dbg", mask)
    call()
    cb.synchronize()
    lib.rip_chain_stamps_n(cb.ctx.h, nw, out)
    n = 3
    for _ in range(n):
        call()
    lib.rip_chain_stamps_n(cb.ctx.h, nw, out)
    print(f"form {cb.ctx.lib.rip_last_chain_form(cb.ctx.h)} dbg={mask} groups={NG} split={split}: cycles per row step and wave ({steps} steps, {nwg} workgroups)")
    for ri, name in enumerate(roles):
        acc = [0.0] * 9
        wr = nw // len(roles)
        for w in range(wr * ri, wr * ri + wr):
            for i in range(9):
                acc[i] += out[w * 9 + i]
        per = [a / n / nwg / wr / steps for a in acc]
        lab = lab_i if name.startswith("ingest") else lab_f
        print(f"  {name:10s} total {sum(per):7.0f} | " + " | ".join(f"{lab[i]} {per[i]:.0f}" for i in range(9) if lab[i] != "-"))
