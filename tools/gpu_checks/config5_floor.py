"""How much of a realisation (BASELINE config 5) the correlated-noise frames cost: the harness with and without them.
   python tools/gpu_checks/config5_floor.py [nseeds]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from romanimpreprocess_amd import pipeline, synth, synth_gpu   # noqa: E402
from romanimpreprocess_amd.from_sim import sim_to_isim   # noqa: E402
from romanimpreprocess_amd.harness import many_realizations as mr   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
N = 4096
rp = synth.READ_PATTERN_8
cal = synth_gpu.make_caldir(N, N, read_pattern=rp, p_order=8, seed=1001, ipc_dtype=np.float32, device=0)
cb = pipeline.Calibrator(device=0)
cb.load_caldir(0, cal)
if os.environ.get("PINK_FORM"):   # 0: the library's transform for the 1/f frames
    cb.ctx.set_option("pink_form", int(os.environ["PINK_FORM"]))
from romanimpreprocess_amd import _native   # noqa: E402

l1s = sim_to_isim.L1Synth(cal, rp, synth.FRAME_TIME, ctx=_native.Context(0) if os.environ.get("OWN_CTX", "1") == "1" else cb.ctx)
if os.environ.get("PINK_FORM"):
    l1s.ctx.set_option("pink_form", int(os.environ["PINK_FORM"]))
scene = synth.make_rate_image(N, N, 100)
orig = sim_to_isim.L1Synth.make
for label, banding in (("with 1/f frames", True), ("without", False), ("with 1/f frames", True), ("without", False)):
    sim_to_isim.L1Synth.make = (lambda self, counts, seed, poisson=False, banding=True, _b=banding: orig(self, counts, seed, poisson, _b))
    mr.run(cb, 0, cal, nseeds=2, seed0=900, read_pattern=rp, generator="hip", l1synth=l1s)
    tm = {}
    t0 = time.perf_counter()
    mr.run(cb, 0, cal, nseeds=n, seed0=100, read_pattern=rp, generator="hip", timings=tm, rate=scene, l1synth=l1s)
    el = time.perf_counter() - t0
    print(f"{label}: {1e3 * tm['generate_s'] / n:.2f} ms generate + {1e3 * tm['calibrate_and_stack_s'] / n:.2f} ms calibrate and stack per realisation"
          f" ({el:.2f} s in all for {n})", flush=True)
