#!/bin/bash
# Quick look at one change (run through gpurun from the repo root): per-step times and the kernel table of the flag sets named
# in $2 on S-pan (10^8) and of those in $3 on S-big1 (10^7).   tools/quick_perf.sh <tag> ["default c5"] ["default"]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-quick}
SPAN=${2:-default sweep full c5}
SBIG=${3:-}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMON="--only --cpu-sample 0 --no-pcie --e2e 0 --sbig1 0"
one () {
  wl=$1; p=$2
  python3 $R/bench.py --workload $wl --pipeline $p --steps 8 --warmup 2 $COMMON --detail $OUT/${wl}_$p.json > $OUT/${wl}_$p.log 2>&1
  python3 - <<P
import json
d = json.load(open("$OUT/${wl}_$p.json"))
ks = d["kernels_ms_per_step"]
print("$wl $p", round(d["ms_per_step"], 2), "all-events", round(d["ms_per_step_all_events"], 2), "no-events", round(d["ms_per_step_unprofiled"], 2), "kernels", round(sum(ks.values()), 2))
print("   ", ", ".join(f"{k} {v:.2f}" for k, v in list(ks.items())[:24]))
P
}
for p in $SPAN; do one span $p; done
for p in $SBIG; do one sbig1 $p; done
