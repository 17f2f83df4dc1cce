#!/bin/bash
# Differential fuzz of the current build against the oracle (run through gpurun from the repo root):
#   tools/fuzz_campaign.sh <tag> [minutes per leg = 3]
# Legs: fuzz_gpu.py plain, with SWG_CHAIN_DEEP=1 (and with SWG_CAND_GENERIC=1 on top; and with --wide-gaps: gap limits at the
# borders of the deep candidate kernel's loops and beyond 2^32 -- added at the very end of round 4, NOT yet run on a GPU), with SWG_SORT_PAIRS=1, with SWG_SORT_BITS8=1,
# with SWG_CHAIN_OLD=1, with SWG_SORT_DROP=0 (no sorts on truncated keys), with SWG_SLOTS=1 (record slots for the scaffold
# stage's gather whatever the shape), and on
# records grouped by query genome with SWG_STREAM_CHUNK=700 (the streamed host path, one context and several; every third
# case without an identity column); fuzz_seams.py; fuzz_cli.py;
# fuzz_large.py (million-record shapes).  Logs under gpurun_out/<tag>_*.log, one summary line per leg on stdout.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-fuzz}
MIN=${2:-3}
cd $R
run () {  # name, env assignments..., -- command
  name=$1; shift
  ( env "$@" ) > gpurun_out/${TAG}_$name.log 2>&1
  echo "$name rc=$? $(tail -1 gpurun_out/${TAG}_$name.log | cut -c1-200)"
}
mkdir -p gpurun_out
export SWG_POISON=1   # value columns the host paths do not send are filled with 0xff bytes: a reader that should not be there shows
# Round 5: the scaffold stage of inputs of up to 65,536 records (every case of fuzz_gpu.py) runs pair-resident (swg_pair.hip)
# unless SWG_GROUP_FUSED=0: `gpu` / `gpu_wide` are that path, the legs with knobs of the global-sort stage switch it off.
run gpu        X=1                python3 tests/fuzz/fuzz_gpu.py --minutes $MIN --seed 11
run gpu_wide   X=1                python3 tests/fuzz/fuzz_gpu.py --minutes $MIN --seed 26 --wide-gaps
run gpu_global SWG_GROUP_FUSED=0  python3 tests/fuzz/fuzz_gpu.py --minutes $MIN --seed 27
run gpu_deep   SWG_GROUP_FUSED=0 SWG_CHAIN_DEEP=1   python3 tests/fuzz/fuzz_gpu.py --minutes $MIN --seed 12
run gpu_pairs  SWG_GROUP_FUSED=0 SWG_SORT_PAIRS=1   python3 tests/fuzz/fuzz_gpu.py --minutes $MIN --seed 13
run gpu_bits8  SWG_GROUP_FUSED=0 SWG_SORT_BITS8=1   python3 tests/fuzz/fuzz_gpu.py --minutes $MIN --seed 14
run gpu_old    SWG_GROUP_FUSED=0 SWG_CHAIN_OLD=1    python3 tests/fuzz/fuzz_gpu.py --minutes $MIN --seed 18
run gpu_nodrop SWG_GROUP_FUSED=0 SWG_SORT_DROP=0    python3 tests/fuzz/fuzz_gpu.py --minutes $MIN --seed 20
run gpu_slots  SWG_GROUP_FUSED=0 SWG_SLOTS=1        python3 tests/fuzz/fuzz_gpu.py --minutes $MIN --seed 21
run gpu_deepg  SWG_GROUP_FUSED=0 SWG_CHAIN_DEEP=1 SWG_CAND_GENERIC=1 python3 tests/fuzz/fuzz_gpu.py --minutes $MIN --seed 22
run gpu_deepw  SWG_GROUP_FUSED=0 SWG_CHAIN_DEEP=1    python3 tests/fuzz/fuzz_gpu.py --minutes $MIN --seed 23 --wide-gaps
run gpu_plain  SWG_WALK_PLAIN=1   python3 tests/fuzz/fuzz_gpu.py --minutes $MIN --seed 32
run gpu_stream SWG_STREAM_CHUNK=700 python3 tests/fuzz/fuzz_gpu.py --minutes $MIN --seed 19 --grouped
run seams      X=1                python3 tests/fuzz/fuzz_seams.py --minutes $MIN --seed 15
run cli        X=1                python3 tests/fuzz/fuzz_cli.py --minutes $MIN --seed 16
run large      X=1                python3 tests/fuzz/fuzz_large.py --seed 17
run large_pm   X=1                python3 tests/fuzz/fuzz_large.py --seed 24 --pair-major
# Round 6: the three opt-in ways of the k = 1 sweep over a pair-major input (fused with the segment sort, streamed behind it, lone
# intervals settled inside it), and the pinned ring of swg_filter_multi wrapping around (every gpu leg runs swg_filter_multi on
# ungrouped records: here with 512 records per slot)
run large_fused  SWG_SEG_SWEEP=1  python3 tests/fuzz/fuzz_large.py --seed 28 --pair-major
run large_stream SWG_SEG_STREAM=1 python3 tests/fuzz/fuzz_large.py --seed 29 --pair-major
run large_lone   SWG_SEG_LONE=1   python3 tests/fuzz/fuzz_large.py --seed 30 --pair-major
run large_plain  SWG_WALK_PLAIN=1  python3 tests/fuzz/fuzz_large.py --seed 33 --pair-major   # (the fused walk's per-lane lists on chunks of thousands of members)
run gpu_ring   SWG_RING_CHUNK=512 python3 tests/fuzz/fuzz_gpu.py --minutes $MIN --seed 31
