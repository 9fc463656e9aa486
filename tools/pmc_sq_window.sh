#!/bin/bash
# SQ instruction-mix counters of one bench window (one rocprofv3 --pmc pass, kernel trace only) -> gpurun_out/<tag>_pmc_sq_bench_window.json
TAG=${1:-r01_v3}
R=$(pwd); export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $R/gpurun_out/pmcsq -- python $R/bench.py --no-cpu-baseline --steps 1 --warmup 1 > $R/gpurun_out/pmcsq.log 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_pmc_sq_bench_window.json "rocprofv3 --pmc SQ_* (one pass) of bench.py (Shift-Net-s, 1280x720, one_len 16): per-launch averages over all launches of a kernel in the window. SQ_WAVE_CYCLES / SQ_BUSY_CYCLES count quad-cycles (MI355X_MICROARCH.md)." $R/gpurun_out/pmcsq
tail -3 $R/gpurun_out/pmcsq.log
cd $R; rm -rf gpurun_out/pmcsq
