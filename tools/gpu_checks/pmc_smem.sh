#!/bin/bash
# Scalar-memory behaviour of the fused kernel: bash tools/gpu_checks/pmc_smem.sh <tag>
tag=${1:-s}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_SMEM SQ_INSTS_SMEM_NORM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_BUSY_CYCLES" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_DCACHE_BUSY_CYCLES SQC_TC_DATA_READ_REQ SQC_TC_STALL SQC_DCACHE_INPUT_VALID_READYB" \
           "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_$i -o run -- python3 $R/tools/gpu_checks/phase_timing.py 0 > $R/gpurun_out/pmc_${tag}_$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections
d=collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/pmc_${tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'chain' in r['Kernel_Name']:
            d[r['Counter_Name']].append(float(r['Counter_Value']))
with open("$R/gpurun_out/pmc_${tag}_summary.txt","w") as o:
    for k,v in sorted(d.items()):
        line=f"{k:34s} n={len(v):3d} mean={sum(v)/len(v):.4g}"
        print(line); o.write(line+"\n")
PY
