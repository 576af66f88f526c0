#!/bin/bash
# A/B harness: runs bench.py against alternative builds of libuavx.so (tools/ab/*.so), restoring the real one.
cd "$(dirname "$0")/.."
L=gym_uav_collision_avoidance_amd/csrc/libuavx.so
cp $L /tmp/libuavx_orig.so
for rep in 1 2; do
for so in tools/ab/*.so; do
  cp $so $L
  echo -n "$so rep $rep: "; python bench.py --steps 3000 --warmup 300 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f us' % (d['ms_per_step']*1e3))"
done
done
cp /tmp/libuavx_orig.so $L
