#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the streaming micro-benchmark (its byte count is known exactly): calibrates what the counters
# report for the fused kernel's mix of access widths.  bash tools/gpu_checks/pmc_mix.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcmix_$c -o run -- $R/tools/gpu_checks/stream_mix > $R/gpurun_out/pmcmix_$c.log 2>&1 || echo "pass $c failed"
done
python3 - <<PY
import csv,glob,collections
for c in ("FETCH_SIZE","WRITE_SIZE"):
    d=collections.defaultdict(list)
    for f in glob.glob("$R/gpurun_out/pmcmix_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name']==c: d[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    for k,v in sorted(d.items()):
        print(c, k, "n=%d mean=%.6g KB" % (len(v), sum(v)/len(v)))
PY
