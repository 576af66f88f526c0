#!/bin/bash
# Counters of the bare step kernel at HBM-sized batches (review item: why does 4 Mi envs stream slower than 1 Mi?): address
# translation (UTCL1), memory-side stalls and request latencies of the L2 (TCC), and the traffic passes -- separate --pmc runs,
# never combined with a trace domain.   usage: tools/profile_large.sh [E ...]    (GPU box; then tools/summarize_counters.py)
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_large
mkdir -p $OUT
declare -A PASS
PASS[utcl1]="TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum"
PASS[ea_stall]="TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum"
PASS[tcc_lat]="TCC_READ_REQ_LATENCY_sum TCC_WRITE_REQ_LATENCY_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
PASS[tcp_lat]="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
PASS[fetch]="FETCH_SIZE"
PASS[write]="WRITE_SIZE"
PASS[tcc_hit]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_sum"
for E in "$@"; do
  d=$OUT/$E; rm -rf $d; mkdir -p $d
  BARGS="--envs $E --agents 4 --ring 6 --no-cpu-baseline --no-large --steps 30 --warmup 6 --repeats 1 --mode launch"
  for p in utcl1 ea_stall tcc_lat tcp_lat fetch write tcc_hit; do
    rocprofv3 --output-format csv --kernel-trace --pmc ${PASS[$p]} -d $d/$p -o run -- python3 bench.py $BARGS > $d/$p.log 2>&1 || echo "$E $p FAILED"
    find $d/$p -name "*kernel_trace.csv" -delete; find $d/$p -name "*agent_info.csv" -delete
    echo "$E $p done"
  done
done
python3 tools/summarize_counters.py $OUT > $OUT/summary.json
cat $OUT/summary.json
