#!/bin/bash
# per-kernel times of the Level-1 synthesis + chain + statistics: rocprofv3 kernel trace of N realisations (default 8)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
n=${1:-8}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/synth_trace -o run -- python3 $R/tools/gpu_checks/many_realizations_fullsize.py $n hip > $R/gpurun_out/synth_trace.log 2>&1 || echo "trace failed"
python3 - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/synth_trace/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e6:8.3f} ms total {float(r['TotalDurationNs'])/1e6:9.1f} ms")
PY
tail -1 $R/gpurun_out/synth_trace.log
