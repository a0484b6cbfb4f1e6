"""Diagnostic: per-phase cycle shares of the fused kernel (needs libromanhip_stamp.so built with -DCH_STAMP)."""
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

rp = synth.READ_PATTERN_8
N = 4096
cal, ramp = synth.make_tiled_inputs(N, N, read_pattern=rp, p_order=8, seed=1, strip_rows=64)
cb = pipeline.Calibrator(device=0)
lib = cb.ctx.lib
lib.rip_chain_stamps.restype = C.c_int
lib.rip_chain_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
lib.rip_chain_stamps2.restype = C.c_int
lib.rip_chain_stamps2.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
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
out = (C.c_double * 9)()
lib.rip_chain_stamps(cb.ctx.h, out)  # allocates


NOGDQ = bool(int(os.environ.get("NOGDQ", "0")))


def call():
    cb.calibrate_device(0, pid, 8, t[0].data_ptr(), True, t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(),
                        o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), None if NOGDQ else o[4].data_ptr())


CHAIN2 = bool(int(os.environ.get("CHAIN2", "1")))
CHAIN3 = bool(int(os.environ.get("CHAIN3", "0")))   # (the wave-private kernel: f64 ipc4d x 16 groups only since round 3)
cb.ctx.set_option("chain2", int(CHAIN2))
cb.ctx.set_option("chain3", int(CHAIN3))
if CHAIN3:
    names3 = ["top: late group loads (+drain with 2048)", "A refpix/bias/linearity", "issue next row's loads", "C first iterate",
              "O2 second iterate / gain", "issue coefficient set + tail loads", "barrier", "F fit", "T flags, finish, stores"]
    for mask in [int(x) for x in sys.argv[1:]] or [0]:
        cb.ctx.set_option("chain_dbg", mask)
        call()
        cb.synchronize()
        lib.rip_chain_stamps(cb.ctx.h, out)
        n = 3
        for _ in range(n):
            call()
        lib.rip_chain_stamps(cb.ctx.h, out)
        tot = sum(out)
        print(f"dbg={mask}: ticks per launch summed over waves {tot/n:.4g}")
        for i in range(9):
            print(f"   {names3[i]:44s} {100*out[i]/tot:5.1f}%")
    sys.exit(0)
names2 = ["ingest A lin", "ingest barrier 1", "ingest C + loads", "ingest barrier 2", "fit O2", "fit barrier 1",
          "fit F", "fit barrier 2", "-"]
names18 = ["ingest S1 issue loads", "ingest drain (dbg 2048)", "ingest A lin", "ingest barrier 1", "ingest fetch row r+4",
           "ingest C", "ingest T tail", "ingest barrier 2", "-",
           "fit S1 issue loads", "fit drain (dbg 2048)", "fit O2", "fit barrier 1", "fit S2 issue loads",
           "fit drain (dbg 2048)", "fit F", "fit barrier 2", "-"]
if CHAIN2:
    out18 = (C.c_double * 18)()
    for mask in [int(x) for x in sys.argv[1:]] or [0]:
        cb.ctx.set_option("chain_dbg", mask)
        call()
        cb.synchronize()
        lib.rip_chain_stamps2(cb.ctx.h, out18)
        n = 3
        for _ in range(n):
            call()
        lib.rip_chain_stamps2(cb.ctx.h, out18)
        for role in (0, 1):
            tot = sum(out18[9 * role:9 * role + 9])
            print(f"dbg={mask} role {role}: total ticks per wave per launch {tot/n/2048:.0f}")
            for i in range(9 * role, 9 * role + 8):
                print(f"   {names18[i]:26s} {out18[i]/n/2048/141:8.0f} per step  {100*out18[i]/tot:5.1f}%")
    sys.exit(0)
names = ["P issue loads", "C O1", "barrier 1", "E rest (finish+stores)", "A lin", "barrier 2", "E: O2", "E: fit+flags", "wait all loads (dbg 2048)"]
for mask in [int(x) for x in sys.argv[1:]] or [0]:
    cb.ctx.set_option("chain_dbg", mask)
    call()
    cb.synchronize()
    lib.rip_chain_stamps(cb.ctx.h, out)
    n = 3
    for _ in range(n):
        call()
    lib.rip_chain_stamps(cb.ctx.h, out)
    tot = sum(out)
    nw = 2048
    if CHAIN2:
        names = names2
    print(f"dbg={mask}: total cycles per wave per launch {tot/n/nw:.0f}")
    for i in range(9):
        print(f"   {names[i]:16s} {out[i]/n/nw:10.0f} cycles/wave  {100*out[i]/tot:5.1f}%  ({out[i]/n/nw/141:.0f} per step)")
