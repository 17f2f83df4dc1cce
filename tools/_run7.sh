set -u
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_scaffold.py tests/test_gpu_sbig1.py tests/test_gpu_cli.py -m gpu -x -q > gpurun_out/r03_k_pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r03_k_pytest.log
C="--only --cpu-sample 0 --no-pcie --e2e 0 --sbig1 0 --steps 10 --warmup 2"
python3 bench.py --workload sbig1 --pipeline default $C --detail gpurun_out/r03_k_sbig1_default.json > gpurun_out/r03_k_sbig1_default.line 2>&1
python3 bench.py --pipeline default $C --detail gpurun_out/r03_k_span_default.json > gpurun_out/r03_k_span_default.line 2>&1
python3 - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_k_*.json')):
    j=json.load(open(f))
    for p,e in j['pipelines'].items():
        k=e['kernels_ms_per_step']; top=sorted(k.items(), key=lambda x:-x[1])[:12]
        print(f, p, round(e['ms_per_step'],2), round(e['ms_per_step_unprofiled'],2), top)
P
