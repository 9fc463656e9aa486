#!/bin/bash
# Round 6, GPU call S: pixel-shuffle epilogue storing whole pixels: tests (bf16 + fp32 engines), per-label timings of the non-plain convs.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
( timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_io.py tests/test_gpu_fp32.py -x -q -m gpu -k "skip_upsample or conv_epilogues or test_conv or unet or whole_net or full_size or hipgraph or cli or fp32" ) > gpurun_out/r6s_tests.txt 2>&1; grep -n "passed\|failed" gpurun_out/r6s_tests.txt; grep -n "^E " gpurun_out/r6s_tests.txt | head -5
( timeout 600 python tools/conv_labels.py --config 2 ) > gpurun_out/r6s_conv_labels_cfg2.txt 2>&1; grep "not a plain" gpurun_out/r6s_conv_labels_cfg2.txt | head -4
( timeout 600 python tools/conv_labels.py --config 3 ) > gpurun_out/r6s_conv_labels_cfg3.txt 2>&1; grep "not a plain" gpurun_out/r6s_conv_labels_cfg3.txt | head -4
