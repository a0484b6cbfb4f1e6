#!/bin/bash
# Instruction counts of the fused kernel's phases: PMC counters of the debug build (-DC2_DBG, alt library) with phases
# switched off one at a time.  usage (GPU box, repo root): bash tools/gpu_checks/pmc_ablate.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MASKS="0 1 2 4 8 16 32 64 128"
ALTLIB=libromanhip_stamp.so CHAIN2=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_abl -o run -- python3 $R/tools/gpu_checks/phase_timing.py $MASKS > $R/gpurun_out/pmc_abl.log 2>&1 || echo failed
python3 - <<PY
import csv,glob,collections
rows=[]
for f in glob.glob("$R/gpurun_out/pmc_abl/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'chain2' in r['Kernel_Name']:
            rows.append((int(r['Dispatch_Id']), r['Counter_Name'], float(r['Counter_Value'])))
ids=sorted(set(d for d,_,_ in rows))
masks="$MASKS".split()
per=len(ids)//len(masks)
res=collections.defaultdict(dict)
for d,c,v in rows:
    m=masks[min(ids.index(d)//per, len(masks)-1)]
    res[m].setdefault(c,[]).append(v)
ws=266.0e3  # 64-pixel wave-steps per launch (one ingest + one fit wave each)
base={c:sum(v)/len(v) for c,v in res['0'].items()}
with open("$R/gpurun_out/pmc_abl_summary.txt","w") as o:
    for m in masks:
        line=f"mask {m:>3s}: "+"  ".join(f"{c[9:]}={(sum(v)/len(v))/ws:8.1f} ({(sum(v)/len(v)-base[c])/ws:+7.1f})" for c,v in sorted(res[m].items()) if 'INSTS' in c)
        print(line); o.write(line+"\n")
PY
