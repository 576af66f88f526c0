#!/bin/bash
# Regenerates the rocprofv3 evidence under gpurun_out/prof/<shape>/<pass>/ (run on the GPU box through gpurun), then
#   python tools/summarize_profiles.py rNN   condenses it into profiles/.
# usage: tools/profile_round.sh [calib] [ExN[+B][f][cL][r][p] | uwE ...]     e.g.  tools/profile_round.sh calib 65536x4 65536x8+16 65536x4f 65536x8+16fc4r uw1048576
#        (+B: scripted bodies; f: the fused uavx_step_ex path with polar actions, auto-reset and statistics; cL: L-level
#         curriculum; r: outputs into the on-device replay ring -- 65536x8+16fc4r is bench.py --cfg5)
# Kernel trace and each counter group are separate runs (counter collection serialises and slows kernels); the
# program itself follows `--` (no env / bash -c hop).  TCC has 4 counter slots per pass, SQ 8.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof
mkdir -p $OUT
# what the passes were taken on: the hash bench.py compares with the loaded library's, and when (the tool writes both; nothing is keyed by hand)
python3 -c "from gym_uav_collision_avoidance_amd import _lib; import json, datetime; json.dump({'csrc_sha': _lib.source_hash(), 'taken_utc': datetime.datetime.now(datetime.timezone.utc).strftime('%Y-%m-%dT%H:%M:%SZ')}, open('$OUT/meta.json','w'))"
declare -A PASS
PASS[fetch]="FETCH_SIZE"
PASS[write]="WRITE_SIZE"
PASS[tcc_hit]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
PASS[tcc_ea]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_EA0_WRREQ_sum"
PASS[sq]="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
PASS[sq2]="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"
for arg in "$@"; do
  if [ "$arg" = calib ]; then
    [ -x tools/micro/fetch_calib ] || hipcc -O3 --offload-arch=gfx950 -o tools/micro/fetch_calib tools/micro/fetch_calib.hip
    d=$OUT/calib; rm -rf $d; mkdir -p $d
    for p in fetch write tcc_ea tcc_hit; do
      rocprofv3 --output-format csv --kernel-trace --pmc ${PASS[$p]} -d $d/$p -o run -- tools/micro/fetch_calib > $d/$p.log 2>&1
      find $d/$p -name "*kernel_trace.csv" -delete; find $d/$p -name "*agent_info.csv" -delete   # (gpurun copies back at most 64 MiB)
      echo "calib $p done"
    done
    continue
  fi
  d=$OUT/$arg; rm -rf $d; mkdir -p $d
  if [[ "$arg" == uw* ]]; then   # uwE: the UAVWorld2D kernel on E envs
    BARGS="--world uw --envs ${arg#uw} --ring 8 --no-cpu-baseline --no-large"
  else
    fused=""; a=$arg; extra=""
    [[ "$a" == *p ]] && { extra="$extra --packed-flags"; a=${a%p}; }                             # ...p: flags packed into the done bytes
    [[ "$a" == *r ]] && { extra="$extra --replay"; a=${a%r}; }                                  # ...r: outputs into the replay ring
    [[ "$a" =~ c([0-9]+)$ ]] && { extra="$extra --curriculum ${BASH_REMATCH[1]}"; a=${a%c*}; }   # ...cL: L-level curriculum
    [[ "$a" == *f ]] && { fused="--fused"; a=${a%f}; }
    shape=${a%%+*}; bodies=0; [[ "$a" == *+* ]] && bodies=${a##*+}
    E=${shape%%x*}; N=${shape##*x}
    BARGS="--envs $E --agents $N --bodies $bodies --no-cpu-baseline --no-large $fused $extra"
  fi
  rocprofv3 --output-format csv --kernel-trace --stats -d $d/kt -o run -- python3 bench.py $BARGS --steps 1000 --warmup 100 > $d/kt.log 2>&1
  find $d/kt -name "*kernel_trace.csv" -delete; find $d/kt -name "*agent_info.csv" -delete   # keep the stats summary only (64 MiB merge limit)
  echo "$arg kt done"
  for p in fetch write tcc_hit tcc_ea sq sq2; do
    rocprofv3 --output-format csv --kernel-trace --pmc ${PASS[$p]} -d $d/$p -o run -- python3 bench.py $BARGS --steps 100 --warmup 20 --repeats 2 --mode launch > $d/$p.log 2>&1
    find $d/$p -name "*kernel_trace.csv" -delete; find $d/$p -name "*agent_info.csv" -delete
    echo "$arg $p done"
  done
done
find $OUT -name "*.csv" | wc -l
