#!/bin/bash
# the fused kernel on the bench's non-periodic frame against the homogeneous tiled frame of round 1, same box, alternately
for i in 1 2; do
for f in "" "--tiled"; do
  python3 bench.py $f --no-cpu-baseline --no-extras --steps 15 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('frame=${f:-nonperiodic}', round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['chain']['good_pixel_fraction'],4))"
done; done
