#!/bin/bash
# Round 6, GPU call W: K0 on the matrix cores (banded GEMM over a channel-planar LDS window): parity, then timing against the VALU kernel.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "shiftconv_on_the_matrix" ) > gpurun_out/r6w_tests.txt 2>&1; grep -n "passed\|failed" gpurun_out/r6w_tests.txt; grep -n "^E " gpurun_out/r6w_tests.txt | head -8
( timeout 600 python tools/k0_ab.py ) > gpurun_out/r6w_k0_ab.txt 2>&1; grep -v amdgpu gpurun_out/r6w_k0_ab.txt | tail -12
