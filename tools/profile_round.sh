#!/bin/bash
# Round profile of the bench command on the GPU box (run from the repo root through gpurun):
#   bash tools/profile_round.sh r01
# writes gpurun_out/prof_<tag>/ : kernel stats (rocprofv3 --kernel-trace --stats) and the FETCH_SIZE / WRITE_SIZE PMC
# passes (separate runs, kernel trace only), then tools/profile_summarise.py turns them into profiles/<tag>_*.
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --no-cpu-baseline --no-extras > $O/stats.log 2>&1 || echo "stats pass failed"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o run -- python3 $R/bench.py --steps 20 --warmup 2 --clock-ramp-s 0 --no-cpu-baseline --no-extras > $O/fetch.log 2>&1 || echo "fetch pass failed"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o run -- python3 $R/bench.py --steps 20 --warmup 2 --clock-ramp-s 0 --no-cpu-baseline --no-extras > $O/write.log 2>&1 || echo "write pass failed"
find $O -name "*.csv" | head -20
