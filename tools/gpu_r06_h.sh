#!/bin/bash
# Round 6, GPU call H: what bounds the streaming conv?  Ablations (wrong results): x1 no DMA, x2 no B reads / MFMAs, x4 no stores, combinations; 16 channels.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
( timeout 900 python tools/cab_ab.py --variants r,r/x1,r/x2,r/x4,r/x3,r/x5,r/x6,r/x7 --cases 14x20x720x1280,24x52x720x1280 ) > gpurun_out/r6h_ablation.txt 2>&1; grep "^AB\|^==\|Error\|error" gpurun_out/r6h_ablation.txt
