#!/bin/bash
# Collects SQ counters of the fused kernel in separate passes (one rocprofv3 --pmc run per group).
# usage (on the GPU box, from the repo root): bash tools/gpu_checks/pmc_run.sh <tag>
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32"; do
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
