#!/bin/bash
# fabric traffic (2 x FETCH_SIZE + WRITE_SIZE) of the fused kernel for the bench's variant configurations: bash tools/gpu_checks/pmc_fetch_variants.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "0 8" "1 8" "0 16" "1 16"; do
  set -- $v
  for c in FETCH_SIZE WRITE_SIZE; do
    IPC64=$1 NGROUPS=$2 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcv_$1_$2_$c -o run -- python3 $R/tools/gpu_checks/phase_timing.py 0 > $R/gpurun_out/pmcv_$1_$2_$c.log 2>&1 || echo "pass $v $c failed"
  done
done
python3 - <<PY
import csv,glob,sys
sys.path.insert(0,"$R")
import bench   # the bench's own algorithmic bytes (an earlier version of this script had its own table: 320 / 392 B per pixel for 16 groups, the bench counts 316 / 352)
for k in (("0","8"),("1","8"),("0","16"),("1","16")):
    tot={}
    for c in ("FETCH_SIZE","WRITE_SIZE"):
        v=[]
        for f in glob.glob(f"$R/gpurun_out/pmcv_{k[0]}_{k[1]}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if 'chain2' in r['Kernel_Name'] and r['Counter_Name']==c: v.append(float(r['Counter_Value']))
        tot[c]=1024.0*sum(v)/max(len(v),1)
    t=2*tot["FETCH_SIZE"]+tot["WRITE_SIZE"]; a=bench.alg_bytes(int(k[1]),4096,4096,4,9,ipc_size=8 if k[0]=="1" else 4)[1]["chain_fused"]
    print(f"ipc4d {'f64' if k[0]=='1' else 'f32'} x {k[1]:>2} groups: FETCH_SIZE {tot['FETCH_SIZE']/1e9:.3f} GB (x2), WRITE_SIZE {tot['WRITE_SIZE']/1e9:.3f} GB -> traffic {t/1e9:.2f} GB = {t/a:.3f} x the algorithmic {a/1e9:.2f} GB")
PY
