"""Full-size functional run of the file-level drivers: calibrateimage, then two noise layers, on a synthetic 4096 x 4096 x 8
exposure written to a scratch directory.  python tools/gpu_checks/noise_layers_fullsize.py [scratch_dir]"""
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from romanimpreprocess_amd import calio, synth
from romanimpreprocess_amd.L1_to_L2 import gen_cal_image, gen_noise_image

scratch = sys.argv[1] if len(sys.argv) > 1 else tempfile.mkdtemp(prefix="rip_noise_")
os.makedirs(scratch, exist_ok=True)
rp = synth.READ_PATTERN_8
N = 4096
t0 = time.perf_counter()
cal, ramp = synth.make_tiled_inputs(N, N, read_pattern=rp, p_order=8, seed=1, strip_rows=128)
caldir = {}
for key, fname in {"dark": "dark", "read": "read", "gain": "gain", "linearitylegendre": "linearitylegendre", "ipc4d": "ipc4d",
                   "flat": "pflat", "biascorr": "biascorr", "mask": "mask", "saturation": "saturation"}.items():
    path = os.path.join(scratch, f"roman_wfi_{fname}_TEST_SCA04.asdf")
    calio.write_asdf(path, {"roman": cal[key]})
    caldir[key] = path
calio.write_asdf(os.path.join(scratch, "l1.asdf"),
                 {"roman": {"data": ramp["data"], "amp33": ramp["amp33"],
                            "meta": {"exposure": {"frame_time": synth.FRAME_TIME, "read_pattern": rp},
                                     "instrument": {"detector": "WFI04"}}}})
print(f"synthetic exposure + CALDIR written in {time.perf_counter() - t0:.1f} s")
config = {"IN": os.path.join(scratch, "l1.asdf"), "OUT": os.path.join(scratch, "l2.asdf"), "CALDIR": caldir, "SLICEOUT": True,
          "NOISE": {"LAYER": ["RaS2", "Pr"], "TEMP": os.path.join(scratch, "tmp.asdf"), "SEED": 5,
                    "OUT": os.path.join(scratch, "noise.asdf")}, "NOISE_PRECISION": 32}
t0 = time.perf_counter()
gen_cal_image.calibrateimage(config, verbose=False)
print(f"calibrateimage (files in, files out, CALDIR upload included): {time.perf_counter() - t0:.1f} s")
t0 = time.perf_counter()
gen_cal_image.calibrateimage(config, verbose=False)
print(f"calibrateimage again (CALDIR resident): {time.perf_counter() - t0:.1f} s")
t0 = time.perf_counter()
gen_noise_image.generate_all_noise(config)
print(f"two noise layers: {time.perf_counter() - t0:.1f} s")
out = calio.read_asdf(config["NOISE"]["OUT"])
l2 = calio.read_asdf(config["OUT"])
noise = np.asarray(out["noise"])
good = np.asarray(l2["roman"]["dq"]) == 0
print("layer shapes", noise.shape, "good fraction", good.mean())
print("R layer scatter / read-noise error:", np.std(noise[0][good]) / np.sqrt(np.mean(np.asarray(l2["roman"]["var_rnoise"])[good])))
print("P layer scatter / Poisson error:", np.std(noise[1][good]) / np.sqrt(np.mean(np.asarray(l2["roman"]["var_poisson"])[good])))
if len(sys.argv) <= 1:
    shutil.rmtree(scratch, ignore_errors=True)
