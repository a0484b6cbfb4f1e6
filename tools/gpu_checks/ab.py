"""Same-box A/B timing of library variants: python tools/gpu_checks/ab.py [rounds] tag1 tag2 ...  ('' or 'cur' = working build;
'cur:chain' = working build with the unspecialised kernel).  IPC64=1 / NGROUPS=16 in the environment select the variant
(tools/gpu_checks/phase_timing.py).  Variants are timed alternately in separate processes."""
import os
import statistics
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
args = sys.argv[1:]
rounds = int(args.pop(0)) if args and args[0].isdigit() else 3
tags = args or ["cur"]
res = {t: [] for t in tags}
for _ in range(rounds):
    for t in tags:
        env = dict(os.environ)
        name, _, mode = t.partition(":")
        env["CHAIN2"] = "0" if mode == "chain" else "1"
        if name not in ("", "cur"):
            env["ALTLIB"] = f"libromanhip_{name}.so"
        out = subprocess.run([sys.executable, os.path.join(here, "phase_timing.py"), "0"], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("dbg=")]
        if not line:
            print(t, "FAILED", out.stderr[-400:])
            continue
        res[t].append(float(line[-1].split()[-2]))
for t in tags:
    v = res[t]
    if v:
        print(f"{t:12s} median {statistics.median(v):.3f} ms   min {min(v):.3f}   all {' '.join(f'{x:.3f}' for x in v)}", flush=True)
