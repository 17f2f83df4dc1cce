set -u
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -m gpu -x -q --deselect tests/test_gpu_sbig1_full.py > gpurun_out/r03_j_pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r03_j_pytest.log
C="--only --cpu-sample 0 --no-pcie --e2e 0 --sbig1 0 --steps 10 --warmup 2"
for p in default sweep full; do
  python3 bench.py --pipeline $p $C --detail gpurun_out/r03_j_span_$p.json > gpurun_out/r03_j_span_$p.line 2>&1
  python3 bench.py --workload sbig1 --pipeline $p $C --detail gpurun_out/r03_j_sbig1_$p.json > gpurun_out/r03_j_sbig1_$p.line 2>&1
done
g++ -O2 -std=c++17 -pthread -o /tmp/shb tests/native/shard_host_bench.cpp && for t in 16 64 128; do /tmp/shb 100000000 100 8 $t 0; done > gpurun_out/r03_j_shard_host.jsonl 2>&1; cat gpurun_out/r03_j_shard_host.jsonl
python3 - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_j_*.json')):
    try: j=json.load(open(f))
    except Exception as e: print(f, 'ERR', e); continue
    if 'pipelines' not in j: continue
    for p,e in j['pipelines'].items():
        k=e['kernels_ms_per_step']; top=sorted(k.items(), key=lambda x:-x[1])[:12]
        print(f, p, round(e['ms_per_step'],2), round(e['ms_per_step_unprofiled'],2), top)
P
