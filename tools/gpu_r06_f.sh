#!/bin/bash
# Round 6, GPU call F: counters of the streaming conv kernel (HBM traffic, SQ wait / issue / LDS) on three CAB sizes, tile kernel beside it.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
R=$PWD
cd /tmp; export TMPDIR=/tmp
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS"
SQ2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"
CMD="python $R/tools/cab_ab.py --rounds 1 --reps 2 --variants 0,t --cases 14x20x720x1280,24x52x720x1280,36x52x360x640"
for cn in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $cn --kernel-trace --output-format csv -d $R/gpurun_out/pmcc_$cn -- $CMD > /dev/null 2>&1
done
timeout 600 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $R/gpurun_out/pmcc_sq -- $CMD > /dev/null 2>&1
timeout 600 rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $R/gpurun_out/pmcc_sq2 -- $CMD > /dev/null 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/r06_pmc_conv_kernels_cab_ab.json 1 "rocprofv3 --pmc (FETCH_SIZE | WRITE_SIZE | $SQ | $SQ2: four passes) of $CMD; by_grid_per_launch is the table to read (a grid = one case)" $R/gpurun_out/pmcc_FETCH_SIZE $R/gpurun_out/pmcc_WRITE_SIZE $R/gpurun_out/pmcc_sq $R/gpurun_out/pmcc_sq2
rm -rf $R/gpurun_out/pmcc_*
python - <<PY
import json
d=json.load(open("$R/gpurun_out/r06_pmc_conv_kernels_cab_ab.json"))
for k,v in sorted(d["by_grid_per_launch"].items()):
    if "conv3" in k:
        print(k[:110]); print("   ", {a.replace("_per_launch",""): b for a,b in v.items()})
PY
