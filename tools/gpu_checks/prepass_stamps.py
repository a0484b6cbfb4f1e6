"""Where the single-launch reference-pixel pre-pass (refpix_one.hip) spends its time: clock stamps per workgroup.
    python tools/gpu_checks/prepass_stamps.py [G] [ny]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401

from romanimpreprocess_amd import _native
from romanimpreprocess_amd.utils import reference_subtraction

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ny = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
nx = 4096
rng = np.random.default_rng(1)
dark = (13000 + rng.integers(0, 1600, size=(G, ny, nx)) / 8.0).astype(np.float32)
med = (29000 + rng.integers(0, 64, size=(ny, 128)) / 4.0).astype(np.float32)
amp33 = (29000 + rng.normal(0, 4, size=(G, ny, 128))).astype(np.uint16)
data = (dark + rng.normal(0, 30, size=(G, ny, nx))).astype(np.uint16)
ctx = _native.default_context(0)
ctx.lib.rip_prepass_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
ctx.check(ctx.lib.rip_prepass_stamps(ctx.h, 0, None))
B = (ny + 127) // 128
for rep in range(3):
    reference_subtraction.refpix_tables(data, dark, amp33, med, 0.33, form=1, ctx=ctx)
st = np.zeros((G * B, 16), np.uint64)
ctx.check(ctx.lib.rip_prepass_stamps(ctx.h, G * B, st.ctypes.data))
st = st.astype(np.int64)
names = {0: "start", 1: "phase A", 2: "lv0 hist", 3: "lv0 flush+barrier", 4: "lv0 scan", 5: "lv1 hist", 6: "lv1 flush+barrier",
         7: "lv1 scan", 8: "lv2 hist", 9: "lv2 flush+barrier", 10: "lv2 scan", 11: "exit protocol", 12: "S2 select", 13: "S2 tables",
         14: "S3 lines"}
print(f"G={G} ny={ny} B={B}: clock ticks each phase took, per workgroup (the counters of different XCDs are not comparable): "
      "min / median / max")
for k in range(1, 15):
    ok = (st[:, k] > 0) & (st[:, k - 1] > 0)
    d = st[ok, k] - st[ok, k - 1]
    if d.size:
        print(f"  {k:2d} {names[k]:20s} {d.min():9d} {int(np.median(d)):9d} {d.max():9d}   ({d.size} workgroups)")
ok = st[:, 11] > 0
d = st[ok, 11] - st[ok, 0]
print(f"  start -> exit protocol  {d.min():9d} {int(np.median(d)):9d} {d.max():9d}")
ok = st[:, 14] > 0
d = st[ok, 14] - st[ok, 0]
print(f"  start -> S3 done        {d.min():9d} {int(np.median(d)):9d} {d.max():9d}   (workgroup 0 of each group)")
