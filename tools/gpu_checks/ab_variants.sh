#!/bin/bash
# same-box A/B of library builds on the bench's variant configurations:
#   bash tools/gpu_checks/ab_variants.sh "<lib tags: cur tag1 ...>" "<configs: 'f32 8' 'f64 8' 'f32 16' 'f64 16'>"
libs=$1; cfgs=$2
for i in 1 2; do
for cfg in $cfgs; do
  k=${cfg%%:*}; g=${cfg##*:}
  for lib in $libs; do
    if [ "$lib" != "cur" ]; then export ROMANHIP_LIB=$GRAFT_REPO_ROOT/romanimpreprocess_amd/libromanhip_$lib.so; else unset ROMANHIP_LIB; fi
    BENCH_PROFILE_EVERY=7 python3 bench.py --ipc-dtype $k --groups $g --no-cpu-baseline --no-extras --steps 200 --warmup 20 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ipc4d $k groups $g lib=$lib', 'wall', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4))"
  done
done; done
