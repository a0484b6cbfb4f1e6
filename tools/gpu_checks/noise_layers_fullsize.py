"""Full-size functional run of the file-level drivers: calibrateimage, then the PRODUCTION list of eight noise layers
(runs/summer2025run/OpenUniverse_to_L1L2.py:124-133: four Rz4PbrS2C and four Rz4OS2C) on a synthetic non-periodic 4096 x 4096 x 8
exposure written to a scratch directory.  python tools/gpu_checks/noise_layers_fullsize.py [scratch_dir]
Progress also goes to gpurun_out/noise_layers_fullsize.log."""
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401  (before libromanhip)

from romanimpreprocess_amd import calio, synth, synth_gpu
from romanimpreprocess_amd.L1_to_L2 import gen_cal_image, gen_noise_image

os.makedirs("gpurun_out", exist_ok=True)
_log = open(os.path.join("gpurun_out", "noise_layers_fullsize.log"), "w")
_print = print


def print(*a, **k):  # noqa: A001
    _print(*a, **k, flush=True)
    _print(*a, file=_log, flush=True)


scratch = sys.argv[1] if len(sys.argv) > 1 else tempfile.mkdtemp(prefix="rip_noise_")
os.makedirs(scratch, exist_ok=True)
rp = synth.READ_PATTERN_8
N = 4096
t0 = time.perf_counter()
cal = synth_gpu.make_caldir(N, N, read_pattern=rp, p_order=8, seed=1001)
ramp = synth_gpu.make_ramp(cal, read_pattern=rp, seed=1)
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
          "RAMP_OPT_PARS": {"slope": 0.4, "gain": 1.8, "sigma_read": 7.0},
          "JUMP_DETECT_PARS": {"SthreshA": 5.5, "SthreshB": 4.5, "IthreshA": 0.6, "IthreshB": 600.0}, "SKYORDER": 2,
          "NOISE": {"LAYER": [f"Rz4PbrS2C{i}" for i in (1, 2, 3, 4)] + [f"Rz4OS2C{i}" for i in (5, 6, 7, 8)],
                    "TEMP": os.path.join(scratch, "tmp.asdf"), "SEED": 5,
                    "OUT": os.path.join(scratch, "noise.asdf")}, "NOISE_PRECISION": 32}
t0 = time.perf_counter()
gen_cal_image.calibrateimage(config, verbose=False)
print(f"calibrateimage (files in, files out, CALDIR upload included): {time.perf_counter() - t0:.1f} s")
t0 = time.perf_counter()
gen_cal_image.calibrateimage(config, verbose=False)
print(f"calibrateimage again (CALDIR resident): {time.perf_counter() - t0:.1f} s")
t0 = time.perf_counter()
gen_noise_image.generate_all_noise(config)
t_first = time.perf_counter() - t0
print(f"eight production noise layers, first call of the process (hipFFT builds its transform plan: ~1.3 s once): {t_first:.2f} s")
t0 = time.perf_counter()
gen_noise_image.generate_all_noise(config)
t_layers = time.perf_counter() - t0
print(f"eight production noise layers: {t_layers:.2f} s ({t_layers / 8:.2f} s per layer)")
out = calio.read_asdf(config["NOISE"]["OUT"])
l2 = calio.read_asdf(config["OUT"])
noise = np.asarray(out["noise"])
good = np.asarray(l2["roman"]["dq"]) == 0
print("layer shapes", noise.shape, "good fraction", good.mean())
vr, vp = np.asarray(l2["roman"]["var_rnoise"])[good], np.asarray(l2["roman"]["var_poisson"])[good]
for i in range(noise.shape[0]):
    # every production layer = a read-noise realisation (clipped at 4 sigma-equivalents) + a Poisson-type one, sky model removed
    print(f"layer {i} ({config['NOISE']['LAYER'][i]}): scatter / sqrt(read^2 + poisson^2) =",
          np.std(noise[i][good]) / np.sqrt(np.mean(vr + vp)))
if len(sys.argv) <= 1:
    shutil.rmtree(scratch, ignore_errors=True)
