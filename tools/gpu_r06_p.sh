#!/bin/bash
# Round 6, GPU call P: segment length of the streaming conv as a function of the image only (temporal split bit-identity): the tests that failed,
# the conv tests, and the CAB timings again.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
( timeout 1800 python -m pytest tests/test_temporal_split.py tests/test_gpu_multirank.py tests/test_gpu_parity.py -x -q -m gpu -k "temporal_split or multirank or streaming or test_cab or fused_cab or geometry or wavefront or streams" ) > gpurun_out/r6p_tests.txt 2>&1; grep -n "passed\|failed" gpurun_out/r6p_tests.txt
( timeout 900 python tools/cab_ab.py --variants d,p16 --cases 14x20x720x1280,18x20x360x640,22x20x180x320,24x52x720x1280,36x52x360x640,48x52x180x320,24x36x272x448 ) > gpurun_out/r6p_cab_ab.txt 2>&1; grep "^AB\|^==\|Error\|error" gpurun_out/r6p_cab_ab.txt
