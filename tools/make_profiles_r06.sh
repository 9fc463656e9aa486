#!/bin/bash
# Round-6 profile artefacts (same procedure as round 5), run through gpurun from the repo root; everything lands in gpurun_out/ (copy what is to be judged into profiles/).
# Counters are collected in their OWN passes with --kernel-trace only (never with --stats / other trace domains).  EVERY traced run is
# `bench.py --no-parity --no-cpu-baseline`: only full windows in the trace; every summary is normalised per window by the windows of its own
# run (warmup + steps + 1 profiling step) and records the sources its kernels were compiled from (bench.py: csrc_files).
#   make_profiles_r06.sh TAG [cfg ...]      cfg in: 2 3 4 4fp32 (default: all four).  4 traces ONE of the four quadrants (a quarter of the
#   launches; PMC_WINDOW_FRACTION=0.25 scales the per-window totals); 4fp32: kernel stats only.
#   Per cfg: kernel stats (+ by grid), HBM traffic (FETCH_SIZE, WRITE_SIZE: two passes), one SQ pass (VALU / MFMA instructions, MFMA busy,
#   wave wait / issue cycles, LDS bank conflicts) -> tools/derive_r05.py -> <TAG>_mfma_util_and_traffic_per_kernel_<cfg>.json
set -u
TAG=${1:-r06}; shift
CFGS=${@:-2 3 4 4fp32}
R=$(pwd)
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
SQ="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for c in $CFGS; do
  FRAC=1; PMC=1
  case $c in
    2) ARGS="--config 2"; NAME=cfg2; ST=2; WU=1;;
    3) ARGS="--config 3"; NAME=cfg3; ST=1; WU=1;;
    4) ARGS="--config 4 --one-quadrant"; NAME=cfg4_bf16; ST=1; WU=1; FRAC=0.25;;
    4fp32) ARGS="--config 4 --dtype fp32"; NAME=cfg4_fp32; ST=1; WU=1; PMC=0;;
  esac
  export PMC_WINDOW_FRACTION=$FRAC
  B="python $R/bench.py --no-cpu-baseline --no-parity $ARGS"
  WIN=$((ST + WU + 1))
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$NAME -- $B --steps $ST --warmup $WU > $R/gpurun_out/${TAG}_${NAME}_bench_under_rocprof.log 2>&1
  DB=$(find $R/gpurun_out/prof_$NAME -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py "$DB" $R/gpurun_out/${TAG}_${NAME}_kernel_stats.csv
  python $R/tools/rocprof_by_grid.py "$DB" $R/gpurun_out/${TAG}_${NAME}_by_grid.csv
  rm -rf $R/gpurun_out/prof_$NAME
  [ $PMC = 1 ] || continue
  for cn in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $cn --kernel-trace --output-format csv -d $R/gpurun_out/pmcb_$cn -- $B --steps $ST --warmup $WU > /dev/null 2>&1
  done
  python $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_pmc_hbm_traffic_$NAME.json $WIN "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py --no-parity --no-cpu-baseline $ARGS --steps $ST --warmup $WU ($WIN traced steps, each $FRAC of a window). Raw KB; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md): HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE." $R/gpurun_out/pmcb_FETCH_SIZE $R/gpurun_out/pmcb_WRITE_SIZE
  rm -rf $R/gpurun_out/pmcb_FETCH_SIZE $R/gpurun_out/pmcb_WRITE_SIZE
  timeout 600 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $R/gpurun_out/pmcs_$NAME -- $B --steps $ST --warmup $WU > /dev/null 2>&1
  python $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_pmc_sq_$NAME.json $WIN "rocprofv3 --pmc $SQ (one pass) of bench.py --no-parity --no-cpu-baseline $ARGS --steps $ST --warmup $WU ($WIN traced steps, each $FRAC of a window): totals per window." $R/gpurun_out/pmcs_$NAME
  rm -rf $R/gpurun_out/pmcs_$NAME
  python $R/tools/derive_r05.py $R/gpurun_out $TAG $NAME $WIN $FRAC
done
