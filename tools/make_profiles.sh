#!/bin/bash
# Regenerates the profile artefacts of one round on the GPU box (run through gpurun from the repo root):
#   rocprofv3 --kernel-trace --stats of bench.py  -> gpurun_out/<tag>_bench_kernel_stats.csv
#   rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel trace only) of tools/bench_unit.py -> gpurun_out/<tag>_pmc_hbm_traffic_unit_L1.json
#   bench.py (full line incl. cpu_baseline)      -> gpurun_out/<tag>_bench.json
set -u
TAG=${1:-r01_v3}
R=$(pwd)
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_under_rocprof.log 2>&1
DB=$(find $R/gpurun_out/prof_$TAG -name "*.db" | head -1)
python $R/tools/rocprof_summary.py "$DB" $R/gpurun_out/${TAG}_bench_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$c -- python $R/tools/bench_unit.py --iters 3 > /dev/null 2>&1
done
python $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_pmc_hbm_traffic_unit_L1.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of tools/bench_unit.py: GSTS unit pair at Shift-Net-s level-1 size (T=20, 360x640, C=64; one C-channel tensor = 576000 KB). Raw counter values in KB per launch; MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950, reads served by the 256 MB Infinity Cache may not be counted." $R/gpurun_out/pmc_${TAG}_FETCH_SIZE $R/gpurun_out/pmc_${TAG}_WRITE_SIZE
cd $R
rm -rf gpurun_out/prof_$TAG gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE
timeout 400 python bench.py 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json
