#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for args in "100000000 42 1 1" "8193 33 1 1" "34600000 43 2 0" "8193 64 1 0" "8192 17 1 0" "5000 9 1 0" "1000003 56 1 0"; do
$R/sweepga_amd/bin/sort_bench $args | tail -2
done
SWG_SORT_WIDE=1 $R/sweepga_amd/bin/sort_bench 3000000 40 1 0 | tail -1
cd $R
python -m pytest tests/test_gpu_sweep.py tests/test_gpu_wide.py tests/test_gpu_sbig1_full.py tests/test_gpu_cli.py tests/test_gpu_scaffold.py -m gpu -x -q 2>&1 | tail -4
tools/quick_perf.sh q2 "default" 2>&1 | tail -40
