#!/bin/bash
# Round-4 profile artefacts, run through gpurun from the repo root; everything lands in gpurun_out/ (copy what is to be judged into profiles/).
# Counters are collected in their OWN passes with --kernel-trace only.  EVERY traced run is `bench.py --no-parity --no-cpu-baseline`: only
# full windows in the trace; every summary is normalised per window by the windows of its own run (warmup + steps + 1 profiling step) and
# records the hash of the kernel sources (bench.py: csrc_hash) so that bench.py can refuse it after a kernel change.
#   make_profiles_r04.sh TAG [cfg ...]      cfg in: 2 3 4 4fp32 (default: all four)
set -u
TAG=${1:-r04}; shift
CFGS=${@:-2 3 4 4fp32}
R=$(pwd)
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
for c in $CFGS; do
  case $c in
    2) ARGS="--config 2"; NAME=cfg2; ST=2; WU=1;;
    3) ARGS="--config 3"; NAME=cfg3; ST=1; WU=1;;
    4) ARGS="--config 4"; NAME=cfg4_bf16; ST=1; WU=1;;
    4fp32) ARGS="--config 4 --dtype fp32"; NAME=cfg4_fp32; ST=1; WU=1;;
  esac
  B="python $R/bench.py --no-cpu-baseline --no-parity $ARGS"
  WIN=$((ST + WU + 1))
  # kernel trace + stats of the same command
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$NAME -- $B --steps $ST --warmup $WU > $R/gpurun_out/${TAG}_${NAME}_bench_under_rocprof.log 2>&1
  DB=$(find $R/gpurun_out/prof_$NAME -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py "$DB" $R/gpurun_out/${TAG}_${NAME}_kernel_stats.csv
  python $R/tools/rocprof_by_grid.py "$DB" $R/gpurun_out/${TAG}_${NAME}_by_grid.csv
  for cn in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $cn --kernel-trace --output-format csv -d $R/gpurun_out/pmcb_$cn -- $B --steps $ST --warmup $WU > /dev/null 2>&1
  done
  python $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_pmc_hbm_traffic_$NAME.json $WIN "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py --no-parity --no-cpu-baseline $ARGS --steps $ST --warmup $WU ($WIN full windows per trace). Raw KB; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md): HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE." $R/gpurun_out/pmcb_FETCH_SIZE $R/gpurun_out/pmcb_WRITE_SIZE
  rm -rf $R/gpurun_out/prof_$NAME $R/gpurun_out/pmcb_FETCH_SIZE $R/gpurun_out/pmcb_WRITE_SIZE
done
