set -u
mkdir -p gpurun_out
python3 tests/fuzz/fuzz_gpu.py --minutes 5 --seed 300000 > gpurun_out/r03_m_fuzz_gpu.log 2>&1 &
SWG_CHAIN_DEEP=1 python3 tests/fuzz/fuzz_gpu.py --minutes 5 --seed 400000 > gpurun_out/r03_m_fuzz_gpu_deep.log 2>&1 &
SWG_SORT_PAIRS=1 python3 tests/fuzz/fuzz_gpu.py --minutes 4 --seed 500000 > gpurun_out/r03_m_fuzz_gpu_pairs.log 2>&1 &
python3 tests/fuzz/fuzz_cli.py --minutes 4 --seed 7000 > gpurun_out/r03_m_fuzz_cli.log 2>&1 &
python3 tests/fuzz/fuzz_seams.py --minutes 4 --seed 9000 > gpurun_out/r03_m_fuzz_seams.log 2>&1 &
python3 tests/fuzz/fuzz_large.py --seed 11 > gpurun_out/r03_m_fuzz_large.log 2>&1 &
wait
for f in gpurun_out/r03_m_fuzz_*.log; do echo "== $f"; tail -3 $f | cut -c1-400; done
