#!/bin/bash
# runs tools/exp_fused_ext.py against every experimental build in tools/ab/*.so (selected with UAVX_LIB)
cd "$(dirname "$0")/.."
for so in gym_uav_collision_avoidance_amd/csrc/libuavx.so tools/ab/*.so; do
  echo "== $so"; UAVX_LIB=$PWD/$so python tools/exp_fused_ext.py "$@" 2>/dev/null | grep -E "^(step |ex_plain|ex_agent0|ex_all)" | tail -5
done
