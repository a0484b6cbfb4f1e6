"""Turns gpurun_out/prof_<tag>/ (tools/profile_round.sh) into profiles/<tag>_bench_kernel_stats.csv and
profiles/<tag>_hbm_traffic.json.  usage: python tools/profile_summarise.py r01"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import kernel_source_hash  # noqa: E402  (hash of the fused kernel's sources: bench.py only quotes a matching profile)

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(REPO, "gpurun_out", f"prof_{tag}")
dst = os.path.join(REPO, "profiles")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))


def counter(sub, name):
    per = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                per[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return per


fetch, write = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
dominant = max((k for k in fetch if "chain" in k), key=lambda k: sum(fetch[k]) / len(fetch[k]), default=None)
out = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, in a separate pass, --pmc WRITE_SIZE) -- python3 bench.py "
                  "--steps 20 --warmup 2 --clock-ramp-s 0 --no-cpu-baseline --no-extras",
       "kernel": dominant, "per_kernel": {}}
for k in sorted(set(fetch) | set(write)):
    if "at::native" in k or "rocprim" in k or "rocclr" in k or k.strip() == "void":   # torch's kernels of the input synthesis: not the library's
        continue
    out["per_kernel"][k] = {"FETCH_SIZE_KB_mean": sum(fetch[k]) / max(len(fetch[k]), 1), "launches_FETCH_SIZE": len(fetch[k]),
                            "WRITE_SIZE_KB_mean": sum(write[k]) / max(len(write[k]), 1), "launches_WRITE_SIZE": len(write[k])}
if dominant:
    f_b = 1024.0 * sum(fetch[dominant]) / len(fetch[dominant])
    w_b = 1024.0 * sum(write[dominant]) / len(write[dominant])
    out["FETCH_SIZE_bytes"], out["WRITE_SIZE_bytes"] = f_b, w_b
    out["correction"] = ("gfx950 tallies 128-B read requests at 64 B (MI355X_MICROARCH.md, HBM section): FETCH_SIZE x2; calibrated on "
                         "amp33_rows_kernel (10.5 MB read once expected, FETCH_SIZE reports half).  64-B requests (the u8 groupdq "
                         "loads) may be tallied in full, so the true read traffic lies slightly below 2 x FETCH_SIZE.")
    out["traffic_bytes_per_launch"] = 2.0 * f_b + w_b
    out["kernel_source_sha16"] = kernel_source_hash()
json.dump(out, open(os.path.join(dst, f"{tag}_hbm_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "per_kernel"}, indent=1))
