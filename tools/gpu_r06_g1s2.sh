#!/bin/bash
# kernel times of config 4 with and without the g1 store
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  rm -rf $R/gpurun_out/prof_g1s$v
  SN_G1_STORE=$v timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_g1s$v -- python $R/bench.py --no-parity --no-cpu-baseline --config 4 --steps 3 --warmup 1 > $R/gpurun_out/r6g1s_prof$v.log 2>&1
  DB=$(find $R/gpurun_out/prof_g1s$v -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py "$DB" $R/gpurun_out/r6g1s_cfg4_kernel_stats_store$v.csv > /dev/null 2>&1
  echo "== SN_G1_STORE=$v"; head -9 $R/gpurun_out/r6g1s_cfg4_kernel_stats_store$v.csv | cut -c1-170
  rm -rf $R/gpurun_out/prof_g1s$v
done
