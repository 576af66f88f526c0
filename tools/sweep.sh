#!/bin/bash
# Secondary measurements (not the headline line): other agent counts, the launch-bound 4096-env cases,
# UAVWorld2D, and the fused step_ex path.  Output: one JSON line per case.
cd "$(dirname "$0")/.."
run() { python bench.py --no-cpu-baseline --no-large --steps 2000 --warmup 200 "$@" 2>/dev/null | tail -1; }
run --envs 65536 --agents 4
run --envs 65536 --agents 4 --fused
run --envs 65536 --agents 1
run --envs 65536 --agents 2
run --envs 65536 --agents 8
run --envs 65536 --agents 5
run --envs 32768 --agents 10
run --envs 16384 --agents 24
run --envs 65536 --agents 24 --steps 500 --warmup 50 --ring 8   # all-learner N = 24 (runtime-N path, 3-wavefront workgroups)
run --envs 65536 --agents 8 --bodies 16 --ring 16               # BASELINE configs[4]'s world: 8 learners + 16 scripted bodies, bare step
run --envs 65536 --agents 8 --bodies 16 --ring 16 --fused
run --envs 65536 --cfg5 --ring 16                               # ... + 4-level randomized-reset curriculum + outputs into the replay ring
run --envs 65536 --agents 8 --fused
run --envs 4096 --agents 1
run --envs 4096 --world uw
run --envs 65536 --world uw
run --envs 1048576 --world uw --steps 500 --warmup 50
run --envs 1048576 --agents 4 --steps 300 --warmup 30 --ring 8
