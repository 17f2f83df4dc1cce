set -u
mkdir -p gpurun_out
C="--only --cpu-sample 0 --no-pcie --e2e 0 --sbig1 0 --steps 1 --warmup 0"
SWG_WALK_STATS=1 python3 bench.py --workload sbig1 --pipeline default $C --detail gpurun_out/r03_c_x.json 2>&1 | grep "swg\]" | tail -4
SWG_WALK_STATS=1 python3 bench.py --pipeline default $C --detail gpurun_out/r03_c_y.json 2>&1 | grep "swg\]" | tail -4
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/r03_c_sq -- python3 $R/bench.py --workload sbig1 --pipeline default $C --detail $R/gpurun_out/r03_c_z.json > $R/gpurun_out/r03_c_sq.log 2>&1
cd $R
python3 tools/pmc_table.py gpurun_out/r03_c_sq | head -40
find gpurun_out/r03_c_sq -name "*.csv" -size +5M -delete
