#!/bin/bash
cd "$(dirname "$0")/.."
L=gym_uav_collision_avoidance_amd/csrc/libuavx.so
cp $L /tmp/libuavx_orig.so
for rep in 1 2; do
for so in tools/ab/*.so; do
  cp $so $L
  echo "== $so"; python tools/exp_fused2.py 2>/dev/null | grep -E "ex_agent0_random|ex_agent0_zero|step_fresh"
done
done
cp /tmp/libuavx_orig.so $L
