#!/bin/bash
# Round 6, GPU call B: the fused dense CAB (csrc/sn_cabf.hip) -- bit-identity / oracle tests, interleaved A/B against the two-launch form.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_cab or test_cab" ) > gpurun_out/r6b_tests.txt 2>&1; tail -n 15 gpurun_out/r6b_tests.txt
( timeout 600 python tools/cab_ab.py ) > gpurun_out/r6b_cab_ab.txt 2>&1; grep "^AB\|^==\|Error\|error" gpurun_out/r6b_cab_ab.txt
