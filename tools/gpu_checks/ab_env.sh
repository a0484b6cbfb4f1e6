#!/bin/bash
# same-box A/B of environment settings on the default bench command: bash tools/gpu_checks/ab_env.sh "-" "NAME=VALUE" ...  (two rounds)
for i in 1 2; do
for e in "$@"; do
  if [ "$e" = "-" ]; then envs=""; else envs="$e"; fi
  env $envs python3 bench.py --no-cpu-baseline --no-extras --steps 600 --warmup 50 $BENCH_FLAGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('env $e', 'wall %.4f ms  fused %.4f ms' % (d['chain']['wall_ms_per_ramp'], d['chain']['kernel_ms'].get('chain_fused', 0)))"
done; done
