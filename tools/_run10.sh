set -u
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests/test_gpu_sweep.py tests/test_gpu_scaffold.py tests/test_gpu_cli.py tests/test_gpu_sbig1.py tests/test_gpu_wide.py -m gpu -x -q > gpurun_out/r03_q_pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r03_q_pytest.log
C="--only --cpu-sample 0 --no-pcie --e2e 0 --sbig1 0 --steps 10 --warmup 2"
for p in sweep full; do
  python3 bench.py --pipeline $p $C --detail gpurun_out/r03_q_span_$p.json > gpurun_out/r03_q_span_$p.line 2>&1
done
python3 bench.py --workload sbig1 --pipeline sweep $C --detail gpurun_out/r03_q_sbig1_sweep.json > gpurun_out/r03_q_sbig1_sweep.line 2>&1
python3 - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_q_*.json')):
    j=json.load(open(f))
    for p,e in j['pipelines'].items():
        k=e['kernels_ms_per_step']; top=sorted(k.items(), key=lambda x:-x[1])[:12]
        print(f, p, round(e['ms_per_step'],2), round(e['ms_per_step_unprofiled'],2), top)
P
