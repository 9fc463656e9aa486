#!/bin/bash
# Round 6, GPU call C: the streaming 3x3 conv (csrc/sn_conv3p.hip) -- bit-identity against the tile kernel, CAB tests, interleaved A/B.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streaming_conv or fused_cab or test_cab or test_conv" ) > gpurun_out/r6c_tests.txt 2>&1; tail -n 12 gpurun_out/r6c_tests.txt
( timeout 600 python tools/cab_ab.py --variants 0,t,w1,w2,w3 --cases 14x20x720x1280,18x20x360x640,24x52x720x1280 ) > gpurun_out/r6c_cab_ab.txt 2>&1; grep "^AB\|^==\|Error\|error" gpurun_out/r6c_cab_ab.txt
