#!/bin/bash
# Vector-memory path counters (TA / TCP / TCC / UTCL1) of the kernels named in $3... for one flag set (run through gpurun):
#   [WL=sbig1] tools/cache_probe.sh <tag> <pipeline> [kernel substrings]
# At most two counters of a block per pass (more: "Request exceeds the capabilities of the hardware to collect", after which
# rocprofv3 aborts and then sits until it is killed -- hence the timeouts).
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; P=$2; shift 2
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
WL=${WL:-span}
COMMON="--workload $WL --only --cpu-sample 0 --no-pcie --e2e 0 --sbig1 0 --steps 1 --warmup 0"
pass () {
  d=$1; shift
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$d -- python3 $R/bench.py --pipeline $P $COMMON > $OUT/$d.log 2>&1 || echo "$d: rc=$? $(grep -m1 -i 'exceeds\|error' $OUT/$d.log | cut -c1-160)"
}
pass c1 TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ
pass c2 TA_DATA_STALLED_BY_TC_CYCLES TCP_TCC_WRITE_REQ TCP_TOTAL_CACHE_ACCESSES TCC_EA0_WRREQ_STALL
pass c3 TCC_REQ TCC_HIT TCC_MISS TCC_TAG_STALL
pass c4 TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_TCC_READ_REQ_LATENCY TCP_TCR_TCP_STALL_CYCLES
for d in c1 c2 c3 c4; do python3 $R/tools/pmc_table.py $OUT/$d "$@" 2>/dev/null; done | tee $OUT/${TAG}_cache.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +20M -delete
