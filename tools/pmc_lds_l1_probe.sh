#!/bin/bash
# Which per-CU resource do the barrier-phased kernels (K12, K3m, K3g) saturate?    Two rocprofv3 --pmc
# passes (kernel trace only, one counter block each) of one bench window:
#   pass 1, SQ / LDS : instruction-level LDS pressure and where issue stalls come from
#   pass 2, TCP (L1) : request counts, hit rate and stall cycles of the vector-memory path
#   tools/pmc_lds_l1_probe.sh <tag> [bench.py flags]  ->  gpurun_out/<tag>_pmc_lds.json, gpurun_out/<tag>_pmc_l1.json
# Counter names are the ones `rocprofv3 -L` lists on this image (profiles/ keeps the list of the round it was written in).
TAG=${1:-probe}; shift
R=$(pwd); export TMPDIR=/tmp; cd /tmp
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM"
L1="TCP_TOTAL_ACCESSES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TA_TA_BUSY"
timeout 600 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $R/gpurun_out/pmcl_$TAG -- python $R/bench.py --no-cpu-baseline --no-parity --steps 1 --warmup 1 "$@" > $R/gpurun_out/${TAG}_pmc_lds.log 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_pmc_lds.json 3 "rocprofv3 --pmc $SQ (one pass) of bench.py $*: totals per window (3 full windows in the trace) and per-launch averages by grid; SQ cycle counters are quad-cycles (MI355X_MICROARCH.md)." $R/gpurun_out/pmcl_$TAG
timeout 600 rocprofv3 --pmc $L1 --kernel-trace --output-format csv -d $R/gpurun_out/pmct_$TAG -- python $R/bench.py --no-cpu-baseline --no-parity --steps 1 --warmup 1 "$@" > $R/gpurun_out/${TAG}_pmc_l1.log 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_pmc_l1.json 3 "rocprofv3 --pmc $L1 (one pass) of bench.py $*: totals per window (3 full windows in the trace) and per-launch averages by grid." $R/gpurun_out/pmct_$TAG
tail -2 $R/gpurun_out/${TAG}_pmc_lds.log $R/gpurun_out/${TAG}_pmc_l1.log
cd $R; rm -rf gpurun_out/pmcl_$TAG gpurun_out/pmct_$TAG
