#!/bin/bash
# runs tools/exp_fused_cost.py once per experimental build in tools/ab/*.so (the library is swapped in place)
cd "$(dirname "$0")/.."
L=gym_uav_collision_avoidance_amd/csrc/libuavx.so
cp $L /tmp/libuavx_orig.so
for so in tools/ab/*.so; do
  cp $so $L
  echo "== $so"; python tools/exp_fused_cost.py 2>/dev/null | grep -E "^step  |defaults|polar \+ track  "
done
cp /tmp/libuavx_orig.so $L
