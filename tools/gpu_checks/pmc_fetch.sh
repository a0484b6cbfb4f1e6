#!/bin/bash
# HBM-side read / write volume of the fused kernel for library variants: bash tools/gpu_checks/pmc_fetch.sh cur tag1 ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for tag in "$@"; do
  if [ "$tag" != "cur" ]; then export ALTLIB=libromanhip_$tag.so; else unset ALTLIB; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcf_${tag}_$c -o run -- python3 $R/tools/gpu_checks/phase_timing.py 0 > $R/gpurun_out/pmcf_${tag}_$c.log 2>&1 || echo "pass $tag $c failed"
  done
done
python3 - "$@" <<PY
import csv,glob,sys
for tag in sys.argv[1:]:
    for c in ("FETCH_SIZE","WRITE_SIZE"):
        v=[]
        for f in glob.glob(f"$R/gpurun_out/pmcf_{tag}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if 'chain' in r['Kernel_Name'] and r['Counter_Name']==c: v.append(float(r['Counter_Value']))
        print(tag, c, "n=%d mean=%.4g KB" % (len(v), sum(v)/max(len(v),1)))
PY
