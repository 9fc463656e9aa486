#!/bin/bash
# Round 6, GPU call I: the streaming fused CAB (cabp_kernel): bit-identity tests, then A/B against the two-launch form and the tile-form fused CAB.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streaming_fused_cab" ) > gpurun_out/r6i_tests.txt 2>&1; tail -n 6 gpurun_out/r6i_tests.txt
( timeout 900 python tools/cab_ab.py --variants 0,s8,p,p/d3,p/w1,p/w3,p/d3/w2,p/d3/w4 --cases 14x20x720x1280,18x20x360x640,24x52x720x1280,22x20x180x320 ) > gpurun_out/r6i_cab_ab.txt 2>&1; grep "^AB\|^==\|Error\|error" gpurun_out/r6i_cab_ab.txt
