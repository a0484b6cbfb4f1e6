#!/bin/bash
# same-box A/B of library builds on the default bench command: bash tools/gpu_checks/ab_libs.sh cur tag1 tag2 ...   (two rounds, alternating)
for i in 1 2; do
for lib in "$@"; do
  if [ "$lib" != "cur" ]; then export ROMANHIP_LIB=$GRAFT_REPO_ROOT/romanimpreprocess_amd/libromanhip_$lib.so; else unset ROMANHIP_LIB; fi
  python3 bench.py --no-cpu-baseline --no-extras --steps 600 --warmup 50 $BENCH_FLAGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('lib=$lib', 'wall %.4f ms  fused %.4f ms' % (d['chain']['wall_ms_per_ramp'], d['chain']['kernel_ms'].get('chain_fused', 0)))"
done; done
