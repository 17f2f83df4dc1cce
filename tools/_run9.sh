set -u
mkdir -p gpurun_out
python3 -c "from sweepga_amd import build as b; b.build_cli(); b.build_synth()"
./sweepga_amd/bin/paf-synth 10000000 > /tmp/in10m.paf
for i in 1 2 3; do SWG_DEBUG=1 ./sweepga_amd/bin/sweepga-gpu /tmp/in10m.paf --output-file /tmp/out.paf 2>&1 | grep -v "chaining:\|long units" | tail -7; echo; done
