#!/bin/bash
# Regenerates the rocprof evidence of one version under gpurun_out/<tag>/ (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>        e.g. r02_v3
# For the four flag sets (default, sweep, full, c5) on the S-pan workload (10^8 mappings) and three on S-big1 (10^7, one pair):
#   * rocprofv3 --kernel-trace --stats               -> <tag>_<flags>_<workload>_kernel_stats.csv
#   * two SEPARATE --pmc passes (FETCH_SIZE, WRITE_SIZE, as MI355X_MICROARCH.md prescribes; never combined with other
#     trace domains) reduced by tools/pmc_traffic.py  -> <tag>_hbm_traffic_<flags>_<workload>.json
#   * SQ counters of every flag set (VALU / LDS activity, waits, LDS bank conflicts, occupancy)  -> <tag>_sq_<flags>_<workload>.txt
# The traffic JSONs and SQ tables carry the sha256 of the libsweepga_gpu.so they were taken from (bench.py checks it).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-rXX}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SWG_CALL_MARKER=1   # an empty launch opens every filter call: tools/pmc_traffic.py cuts the traces there
COMMON="--only --cpu-sample 0 --no-pcie --e2e 0 --sbig1 0"
run_set () {  # $1 = workload flag value, $2 = file suffix, $3 = mappings, $4 = flag sets, $5 = flag sets with SQ tables
  WL=$1; SUF=$2; NM=$3; PIPES=$4; SQPIPES=$5
  for p in $PIPES; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_${p}_$SUF -- python3 $R/bench.py --workload $WL --pipeline $p --steps 3 --warmup 1 $COMMON > $OUT/stats_${p}_$SUF.log 2>&1
    f=$(find $OUT/stats_${p}_$SUF -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && { echo "# library $(sha256sum $R/sweepga_amd/libsweepga_gpu.so | cut -c1-64)" > $OUT/${TAG}_${p}_${SUF}_kernel_stats.csv; cat $f >> $OUT/${TAG}_${p}_${SUF}_kernel_stats.csv; }
    find $OUT/stats_${p}_$SUF -name "*kernel_trace.csv" -delete
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_${p}_$SUF -- python3 $R/bench.py --workload $WL --pipeline $p --steps 1 --warmup 0 $COMMON > $OUT/pmc_fetch_${p}_$SUF.log 2>&1
    sha256sum $R/sweepga_amd/libsweepga_gpu.so > $OUT/pmc_fetch_${p}_$SUF/lib_sha256.txt   # (the library the counters were taken from: pmc_traffic.py stamps its JSON with it)
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_${p}_$SUF -- python3 $R/bench.py --workload $WL --pipeline $p --steps 1 --warmup 0 $COMMON > $OUT/pmc_write_${p}_$SUF.log 2>&1
    sha256sum $R/sweepga_amd/libsweepga_gpu.so > $OUT/pmc_write_${p}_$SUF/lib_sha256.txt
    python3 $R/tools/pmc_traffic.py $OUT/pmc_fetch_${p}_$SUF $OUT/pmc_write_${p}_$SUF $NM 4 > $OUT/${TAG}_hbm_traffic_${p}_$SUF.json
    find $OUT -name "*kernel_trace.csv" -delete
  done
  for p in $SQPIPES; do   # SQ counters (VALU / LDS activity, waits, LDS bank conflicts, waves) of every flag set named
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq1_${p}_$SUF -- python3 $R/bench.py --workload $WL --pipeline $p --steps 1 --warmup 0 $COMMON > $OUT/sq1_${p}_$SUF.log 2>&1
    rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d $OUT/sq2_${p}_$SUF -- python3 $R/bench.py --workload $WL --pipeline $p --steps 1 --warmup 0 $COMMON > $OUT/sq2_${p}_$SUF.log 2>&1
    { echo "# rocprofv3 --pmc SQ counters, per-launch averages (tools/pmc_table.py), $p flags, workload $SUF, library $(sha256sum $R/sweepga_amd/libsweepga_gpu.so | cut -c1-12)"; python3 $R/tools/pmc_table.py $OUT/sq1_${p}_$SUF; python3 $R/tools/pmc_table.py $OUT/sq2_${p}_$SUF; } > $OUT/${TAG}_sq_${p}_$SUF.txt
    find $OUT -name "*kernel_trace.csv" -delete
  done
  find $OUT -name "*kernel_trace.csv" -delete
}
run_set span 100m 100000000 "default sweep full c5" "default sweep full c5"   # c5 = BASELINE.json configs[4] as written (many:many + scaffold 1:1 + rescue)
run_set sbig1 sbig1_10m 10000000 "default sweep full" "default"
cd $R
ls -la $OUT | head -60
