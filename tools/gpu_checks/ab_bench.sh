#!/bin/bash
# same-box A/B of library variants on the bench command: bash tools/gpu_checks/ab_bench.sh "<bench flags>" cur tag1 tag2 ...
flags=$1; shift
for i in 1 2; do
for lib in "$@"; do
  if [ "$lib" != "cur" ]; then export ROMANHIP_LIB=$GRAFT_REPO_ROOT/romanimpreprocess_amd/libromanhip_$lib.so; else unset ROMANHIP_LIB; fi
  python3 bench.py $flags --no-cpu-baseline --no-extras --steps 15 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$flags lib=$lib', round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), d['roofline']['kernel_form'])"
done; done
