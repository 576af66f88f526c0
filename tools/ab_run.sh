#!/bin/bash
# A/B harness: bench.py against the in-tree libuavx.so and every alternative build in tools/ab/*.so (UAVX_LIB), 2 rounds.
# usage: tools/ab_run.sh [bench.py args]
cd "$(dirname "$0")/.."
shopt -s nullglob
for rep in 1 2; do
for so in gym_uav_collision_avoidance_amd/csrc/libuavx.so tools/ab/*.so; do
  UAVX_LIB=$PWD/$so python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-large "$@" 2>/dev/null | python tools/benchline.py "$so rep $rep"
done
done
