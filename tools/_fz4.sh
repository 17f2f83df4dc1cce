cd $GRAFT_REPO_ROOT
export SWG_POISON=1
run () { name=$1; shift; ( env "$@" ) > gpurun_out/r6_fz4_$name.log 2>&1; echo "$name rc=$? $(tail -1 gpurun_out/r6_fz4_$name.log | cut -c1-160)"; }
run gpu        X=1                python3 tests/fuzz/fuzz_gpu.py --minutes 1.5 --seed 41
run gpu_wide   X=1                python3 tests/fuzz/fuzz_gpu.py --minutes 1.5 --seed 42 --wide-gaps
run gpu_global SWG_GROUP_FUSED=0  python3 tests/fuzz/fuzz_gpu.py --minutes 1 --seed 43
run gpu_plain  SWG_WALK_PLAIN=1   python3 tests/fuzz/fuzz_gpu.py --minutes 1 --seed 44
run gpu_stream SWG_STREAM_CHUNK=700 python3 tests/fuzz/fuzz_gpu.py --minutes 1 --seed 45 --grouped
run gpu_ring   SWG_RING_CHUNK=512 python3 tests/fuzz/fuzz_gpu.py --minutes 1 --seed 46
run cli        X=1                python3 tests/fuzz/fuzz_cli.py --minutes 1 --seed 47
run large      X=1                python3 tests/fuzz/fuzz_large.py --seed 48
run large_pm   X=1                python3 tests/fuzz/fuzz_large.py --seed 49 --pair-major
run large_plain SWG_WALK_PLAIN=1  python3 tests/fuzz/fuzz_large.py --seed 50 --pair-major
