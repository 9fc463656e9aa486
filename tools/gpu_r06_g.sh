#!/bin/bash
# Round 6, GPU call G: residual operand of the streaming conv through LDS (loader DMA) vs registers; tests + A/B.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streaming_conv or test_cab or full_size_properties" ) > gpurun_out/r6g_tests.txt 2>&1; tail -n 4 gpurun_out/r6g_tests.txt
( timeout 900 python tools/cab_ab.py --variants 0,w1,w2,w3,r,t --cases 14x20x720x1280,18x20x360x640,24x52x720x1280,36x52x360x640,48x52x180x320 ) > gpurun_out/r6g_cab_ab.txt 2>&1; grep "^AB\|^==\|Error\|error" gpurun_out/r6g_cab_ab.txt
