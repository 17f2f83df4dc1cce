#!/bin/bash
# Regenerates the rocprof evidence under gpurun_out/ (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>        e.g. r01_v15
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-rXX}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for p in sweep full default; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$p -- python3 $R/bench.py --pipeline $p --steps 3 --warmup 1 --cpu-sample 0 --others 0 > $OUT/stats_$p.log 2>&1
  f=$(find $OUT/stats_$p -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${TAG}_${p}_100m_kernel_stats.csv
  find $OUT/stats_$p -name "*kernel_trace.csv" -delete
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --others 0 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --others 0 > $OUT/pmc_write.log 2>&1
python3 $R/tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write 100000000 > $OUT/${TAG}_hbm_traffic_sweep_100m.json
find $OUT -name "*kernel_trace.csv" -delete
cd $R
python3 bench.py --steps 5 --warmup 1 --others 3 --pcie --e2e 10000000 2> $OUT/bench.err | tail -1 > $OUT/${TAG}_bench_100m.json
ls -la $OUT
