#!/bin/bash
# kernel time of the production-representative variants: bash tools/gpu_checks/variants.sh [lib tag]
if [ -n "$1" ] && [ "$1" != "cur" ]; then export ROMANHIP_LIB=$GRAFT_REPO_ROOT/romanimpreprocess_amd/libromanhip_$1.so; fi
for cfg in "" "--ipc-dtype f64" "--groups 16" "--groups 16 --ipc-dtype f64" "--p-order 10"; do
  python3 bench.py $cfg --no-cpu-baseline --no-extras --steps 200 --warmup 20 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('${cfg:-(headline)} |', round(d['value'],1), 'ramps/s', round(d['roofline']['kernel_ms'],4), 'ms frac', round(d['roofline']['frac'],3), d['roofline']['kernel_form'])"
done
