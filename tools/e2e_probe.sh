#!/bin/bash
# Where a 10^8-line file-to-file run spends its time (run through gpurun): phase report of sweepga-gpu with the ingest's own
# lap timers (SWG_PAF_DEBUG) and the streamed call's (SWG_DEBUG).   tools/e2e_probe.sh <tag> [lines]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-probe}
N=${2:-100000000}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
{ nproc; free -g; df -hT /tmp; cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag; uname -r; } > $OUT/host.txt 2>&1
{ time $R/sweepga_amd/bin/paf-synth $N 100 2025 150000000 query > /tmp/in.paf ; } 2> $OUT/synth.log
ls -la /tmp/in.paf >> $OUT/host.txt
run () {  # name, env..., -- args
  name=$1; shift
  for rep in 1 2 3; do
    env SWG_PAF_DEBUG=1 SWG_DEBUG=1 "$@" $R/sweepga_amd/bin/sweepga-gpu /tmp/in.paf --output-file /tmp/out.paf $ARGS > $OUT/run_${name}_$rep.log 2>&1
  done
  echo "== $name"; grep -h "sweepga-gpu\]\|pass 2\|streamed call\|device 0" $OUT/run_${name}_3.log | cut -c1-260
}
ARGS="" run default X=1
ARGS="" run one_piece SWG_STREAM=0
ARGS="--num-mappings 1:1 --scaffold-jump 0" run sweep X=1
sha256sum /tmp/out.paf | cut -c1-16
