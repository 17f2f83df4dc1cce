#!/bin/bash
# Quick look at one change (run through gpurun from the repo root): per-step times of the three flag sets on S-pan and
# S-big1 plus the kernel table of the flag sets named in $2 (default "sweep").   tools/quick_perf.sh <tag> ["sweep default"]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-quick}
SETS=${2:-sweep}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMON="--only --cpu-sample 0 --no-pcie --e2e 0 --sbig1 0"
for wl in span sbig1; do
  for p in sweep default full; do
    python3 $R/bench.py --workload $wl --pipeline $p --steps 5 --warmup 2 $COMMON --detail $OUT/${wl}_$p.json > $OUT/${wl}_$p.log 2>&1
    python3 - <<P
import json
d = json.load(open("$OUT/${wl}_$p.json"))
print("$wl $p", d.get("ms_per_step"), d.get("ms_per_step_unprofiled"), d.get("parity_ok"))
P
  done
done
for p in $SETS; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$p -- python3 $R/bench.py --workload span --pipeline $p --steps 3 --warmup 1 $COMMON > $OUT/stats_$p.log 2>&1
  f=$(find $OUT/stats_$p -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${TAG}_${p}_kernel_stats.csv && head -25 $f | cut -c1-160
  find $OUT/stats_$p -name "*kernel_trace.csv" -delete
done
