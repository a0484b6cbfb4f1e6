#!/bin/bash
# kernel durations of the reference-pixel pre-pass run alone (overlap off): bash tools/gpu_checks/prepass_trace.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for f in 0; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pre_trace_$f -o run -- python3 $R/tools/gpu_checks/prepass_trace.py $f > $R/gpurun_out/pre_trace_$f.log 2>&1 || echo "trace $f failed"
  python3 - $R/gpurun_out/pre_trace_$f <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/run_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if any(k in n for k in ("pre_", "sel_", "amp33_rows", "rowcorr", "chan_kernel", "chain2")):
            print(f"{n[:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
done
