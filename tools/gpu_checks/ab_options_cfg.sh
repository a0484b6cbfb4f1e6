#!/bin/bash
# same-box A/B of library options on one of the bench's variant configurations:
#   tools/gpu_checks/ab_options_cfg.sh "<bench flags, e.g. --ipc-dtype f64 --groups 8>" "overlap=0" "prepass_form=1" "-" ...
# (arguments after the first as in ab_options.sh); two rounds, alternating
base=$1; shift
for round in 1 2; do
  for cfg in "$@"; do
    flags=""
    if [ "$cfg" != "-" ]; then for kv in ${cfg//,/ }; do flags="$flags --set $kv"; done; fi
    BENCH_PROFILE_EVERY=7 python3 bench.py $base --no-cpu-baseline --no-extras --steps 300 --warmup 30 $flags 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1])
print('$base | $cfg', 'wall %.4f ms  fused %.4f ms' % (j['ms_per_step'], j['roofline']['kernel_ms']))"
  done
done
