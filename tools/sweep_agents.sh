#!/bin/bash
# Agent-count sweep of the step kernel at ~1.57 M agent slots per launch (65 536 x 24 worth): how the compile-time (1, 2, 4, 8) and
# runtime-N paths, single- and multi-wavefront workgroups, hold up against the roofline.  One JSON line per N.
cd "$(dirname "$0")/.."
for N in 1 2 3 4 5 6 7 8 9 10 11 12 13 16 20 24 32 40 48 64; do
  E=$(( 1572864 / N ))
  python bench.py --envs $E --agents $N --ring 8 --steps 500 --warmup 50 --no-cpu-baseline --no-large 2>/dev/null | tail -1
done
