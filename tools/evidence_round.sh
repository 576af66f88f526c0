#!/bin/bash
# Regenerates the bench evidence of a round under gpurun_out/ev/ (copy into profiles/rNN_* afterwards); run on the GPU box.
cd "$(dirname "$0")/.."
O=gpurun_out/ev; mkdir -p $O
python bench.py > $O/bench_default.log 2>$O/bench_default.err && echo default done
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2>/dev/null && echo driver done
python bench.py --fused --no-cpu-baseline > $O/bench_cfg3_fused.log 2>/dev/null && echo cfg3f done
python bench.py --agents 8 --bodies 16 > $O/bench_cfg5.log 2>/dev/null && echo cfg5 done
python bench.py --agents 8 --bodies 16 --fused --no-cpu-baseline > $O/bench_cfg5_fused.log 2>/dev/null && echo cfg5f done
python bench.py --cfg5 > $O/bench_cfg5_designed.log 2>/dev/null && echo cfg5d done
python bench.py --cfg5 --packed-flags --no-cpu-baseline > $O/bench_cfg5_designed_packed.log 2>/dev/null && echo cfg5dp done
python bench.py --fused --packed-flags --no-cpu-baseline > $O/bench_cfg3_fused_packed.log 2>/dev/null && echo cfg3fp done
python bench.py --agents 24 --ring 8 --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_n24.log 2>/dev/null && echo n24 done
python bench.py --agents 8 --fused --no-cpu-baseline > $O/bench_n8_fused.log 2>/dev/null && echo n8f done
python bench.py --envs 4096 --agents 1 --no-large > $O/bench_cfg2_multi.log 2>/dev/null && echo cfg2 done          # BASELINE configs[1], MultiUAVWorld2D
python bench.py --envs 4096 --world uw --no-large > $O/bench_cfg2_uw.log 2>/dev/null && echo cfg2uw done           # BASELINE configs[1], UAVWorld2D
tools/sweep.sh > $O/sweep.jsonl && echo sweep done
tools/sweep_envs.sh > $O/env_sweep.jsonl && echo envsweep done
tools/sweep_agents.sh > $O/agent_sweep.jsonl 2>/dev/null && echo agentsweep done
(python tools/exp_stepk.py 4096 1; python tools/exp_stepk.py 65536 4) > $O/stepk.log 2>/dev/null && echo stepk done
