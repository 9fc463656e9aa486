#!/bin/bash
# Round 6, GPU call K: compute waves of the streaming conv / fused CAB no longer carry a pending global load into the tile loop (pin()):
# tests, then A/B of the CAB forms at every width.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streaming_fused_cab or streaming_conv or test_cab" ) > gpurun_out/r6k_tests.txt 2>&1; tail -n 3 gpurun_out/r6k_tests.txt
( timeout 900 python tools/cab_ab.py --variants 0,d,r,t,p,p/d3/w3 --cases 14x20x720x1280,18x20x360x640,24x52x720x1280,36x52x360x640,48x52x180x320,64x20x360x640 ) > gpurun_out/r6k_cab_ab.txt 2>&1; grep "^AB\|^==\|Error\|error" gpurun_out/r6k_cab_ab.txt
