#!/bin/bash
# Regenerates the rocprofv3 evidence under gpurun_out/prof_* (run on the GPU box through gpurun), then
#   python tools/summarize_profiles.py rNN   condenses it into profiles/.
# Kernel trace and each counter group are separate runs (counter collection serialises and slows kernels).
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf gpurun_out/prof_kt gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_sq
rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/prof_kt -o runc -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_kt.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_fetch -o runc -- python3 bench.py --steps 300 --warmup 30 --mode launch --no-cpu-baseline > gpurun_out/prof_fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_write -o runc -- python3 bench.py --steps 300 --warmup 30 --mode launch --no-cpu-baseline > gpurun_out/prof_write.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d gpurun_out/prof_sq -o runc -- python3 bench.py --steps 300 --warmup 30 --mode launch --no-cpu-baseline > gpurun_out/prof_sq.log 2>&1
find gpurun_out/prof_* -name "*.csv" | head -20
