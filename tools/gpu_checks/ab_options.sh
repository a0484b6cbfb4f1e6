#!/bin/bash
# same-box A/B of library options on the bench workload: tools/gpu_checks/ab_options.sh "overlap=0" "prepass_form=0" ...
# (each argument is one configuration: comma-separated OPTION=VALUE pairs, "-" = defaults; ENV:NAME=VALUE sets an environment
# variable of the bench process instead); two rounds, alternating
for round in 1 2; do
  for cfg in "$@"; do
    flags=""; envs=""
    if [ "$cfg" != "-" ]; then for kv in ${cfg//,/ }; do
      case "$kv" in ENV:*) envs="$envs ${kv#ENV:}";; *) flags="$flags --set $kv";; esac; done; fi
    env $envs python bench.py --no-cpu-baseline --no-extras --steps 600 --warmup 50 $flags 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1])
print('$cfg', 'wall %.4f ms  fused %.4f ms  prepass(event-to-event) %.4f ms' % (j['chain']['wall_ms_per_ramp'], j['chain']['kernel_ms'].get('chain_fused',0), j['chain']['kernel_ms']['refpix_prepass']))"
  done
done
