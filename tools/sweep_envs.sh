#!/bin/bash
# E-sweep at N=4: launch-bound -> bandwidth-bound transition of the step kernel.
cd "$(dirname "$0")/.."
for E in 4096 16384 32768 65536 131072 262144 1048576 4194304; do
  K=2000; W=200; R=50
  if [ $E -ge 262144 ]; then K=500; W=50; R=16; fi
  if [ $E -ge 4194304 ]; then K=100; W=10; R=4; fi
  python bench.py --envs $E --steps $K --warmup $W --ring $R --no-cpu-baseline --no-large 2>/dev/null | tail -1
done
