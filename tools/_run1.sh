set -u
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03_a_pytest.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r03_a_pytest.log
C="--only --cpu-sample 0 --no-pcie --e2e 0 --sbig1 0 --steps 10 --warmup 2"
for p in default full; do
  python3 bench.py --pipeline $p $C --detail gpurun_out/r03_a_span_$p.json > gpurun_out/r03_a_span_$p.line 2>&1
  python3 bench.py --workload sbig1 --pipeline $p $C --detail gpurun_out/r03_a_sbig1_$p.json > gpurun_out/r03_a_sbig1_$p.line 2>&1
done
SWG_CHAIN_OLD=1 python3 bench.py --pipeline default $C --detail gpurun_out/r03_a_span_default_old.json > gpurun_out/r03_a_span_default_old.line 2>&1
g++ -O2 -std=c++17 -pthread -o /tmp/shb tests/native/shard_host_bench.cpp && /tmp/shb 100000000 100 8 64 0 > gpurun_out/r03_a_shard_host.json 2>&1; cat gpurun_out/r03_a_shard_host.json
/tmp/shb 100000000 100 8 32 0 >> gpurun_out/r03_a_shard_host.json 2>&1
python3 - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_a_*.json')):
    try: j=json.load(open(f))
    except Exception as e: print(f, 'ERR', e); continue
    if 'pipelines' not in j: continue
    for p,e in j['pipelines'].items():
        k=e['kernels_ms_per_step']; top=sorted(k.items(), key=lambda x:-x[1])[:14]
        print(f, p, round(e['ms_per_step'],2), round(e['ms_per_step_unprofiled'],2), top)
P
