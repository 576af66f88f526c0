#!/bin/bash
# A/B harness: bench.py against the in-tree libuavx.so and every alternative build in tools/ab/*.so (UAVX_LIB), 2 rounds.
# usage: tools/ab_run.sh [bench.py args]
cd "$(dirname "$0")/.."
for rep in 1 2; do
for so in gym_uav_collision_avoidance_amd/csrc/libuavx.so tools/ab/*.so; do
  echo -n "$so rep $rep: "; UAVX_LIB=$PWD/$so python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-large "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f us  frac %.3f' % (d['roofline']['kernel_us'], d['roofline']['frac']))"
done
done
