#!/bin/bash
# Where do the waves of each kernel spend their cycles?  One rocprofv3 --pmc pass (SQ block, 8 slots, kernel trace only) of
# one bench window:  tools/pmc_sq_probe.sh <tag> [bench.py flags]  ->  gpurun_out/<tag>_pmc_sq_wait.json
TAG=${1:-probe}; shift
R=$(pwd); export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmcw_$TAG -- python $R/bench.py --no-cpu-baseline --no-parity --steps 1 --warmup 1 "$@" > $R/gpurun_out/${TAG}_pmc_sq_wait.log 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_pmc_sq_wait.json 3 "rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU (one pass) of bench.py $*: totals per window (3 full windows in the trace) and per-launch averages by grid. WAIT_ANY = wave parked at s_waitcnt / barrier, WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing; quad-cycles (MI355X_MICROARCH.md)." $R/gpurun_out/pmcw_$TAG
tail -2 $R/gpurun_out/${TAG}_pmc_sq_wait.log
cd $R; rm -rf gpurun_out/pmcw_$TAG
