#!/bin/bash
# Round 6, GPU call J: streaming fused CAB with the 16-channel weights in registers: tests, A/B at 16 channels (three sizes).
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streaming_fused_cab" ) > gpurun_out/r6j_tests.txt 2>&1; tail -n 3 gpurun_out/r6j_tests.txt
( timeout 900 python tools/cab_ab.py --variants 0,p,p/d3,p/w3,p/d3/w3 --cases 14x20x720x1280,14x12x1080x1920,14x8x240x480 ) > gpurun_out/r6j_cab_ab.txt 2>&1; grep "^AB\|^==\|Error\|error" gpurun_out/r6j_cab_ab.txt
