set -u
mkdir -p gpurun_out
C="--only --cpu-sample 0 --no-pcie --e2e 0 --sbig1 0 --steps 5 --warmup 1"
for ring in 256 1024 4096; do
  SWG_SPEC_RING=$ring python3 bench.py --workload sbig1 --pipeline default $C --detail gpurun_out/r03_d_ring$ring.json > /dev/null 2>&1
  python3 - <<P
import json
j=json.load(open('gpurun_out/r03_d_ring$ring.json'))
e=j['pipelines']['default']; k=e['kernels_ms_per_step']
print('ring $ring', round(e['ms_per_step'],2), {x:k[x] for x in ('chain_walk_spec','chain_candidates_wave','spec_init','spec_check') if x in k})
P
done
