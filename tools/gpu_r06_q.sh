#!/bin/bash
# Round 6, GPU call Q: every sn_conv2d launch of a config-2 and a config-3 window by label (which convs are not on the streaming kernel, what they cost)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
( timeout 600 python tools/conv_labels.py --config 2 ) > gpurun_out/r6q_conv_labels_cfg2.txt 2>&1; grep -v "amdgpu.ids" gpurun_out/r6q_conv_labels_cfg2.txt | head -60
( timeout 600 python tools/conv_labels.py --config 3 ) > gpurun_out/r6q_conv_labels_cfg3.txt 2>&1; grep -v "amdgpu.ids" gpurun_out/r6q_conv_labels_cfg3.txt | head -60
