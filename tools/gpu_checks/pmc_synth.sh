#!/bin/bash
# instruction counters of the Level-1 synthesis kernels (resultants_kernel: f64 bisection; apportion_kernel: binomial shares):
# separate rocprofv3 --pmc passes over 4 realisations; summary to gpurun_out/pmc_synth_summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_synth_$i -o run -- python3 $R/tools/gpu_checks/many_realizations_fullsize.py 4 hip > $R/gpurun_out/pmc_synth_$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections
d=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc_synth_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        for key in ('resultants_kernel','apportion_kernel'):
            if key in k: d[key][r['Counter_Name']].append(float(r['Counter_Value']))
with open("$R/gpurun_out/pmc_synth_summary.txt","w") as o:
    for key in d:
        for c,v in sorted(d[key].items()):
            line=f"{key:20s} {c:30s} n={len(v):3d} mean={sum(v)/len(v):.5g}"
            print(line); o.write(line+"\n")
PY
