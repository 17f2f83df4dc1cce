#!/bin/bash
# SQ counters of the kernels named in $3... for one flag set on S-pan, or on S-big1 with WL=sbig1 (run through gpurun):
#   [WL=sbig1] tools/sq_probe.sh <tag> <pipeline> [kernel substrings]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; P=$2; shift 2
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
WL=${WL:-span}
COMMON="--workload $WL --only --cpu-sample 0 --no-pcie --e2e 0 --sbig1 0 --steps 1 --warmup 0"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq1 -- python3 $R/bench.py --pipeline $P $COMMON > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SMEM SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq2 -- python3 $R/bench.py --pipeline $P $COMMON > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH --output-format csv -d $OUT/sq3 -- python3 $R/bench.py --pipeline $P $COMMON > $OUT/sq3.log 2>&1
for d in sq1 sq2 sq3; do python3 $R/tools/pmc_table.py $OUT/$d "$@"; done | tee $OUT/${TAG}_sq.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +20M -delete
