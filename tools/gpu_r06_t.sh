#!/bin/bash
# Round 6, GPU call T: stride-2 convs of <= 24 input channels on 8 x 32 tiles: tests, per-label timings, config 2 / 3 windows against SN_CONV_S2_SMALL=1.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
( timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "test_conv or conv_epilogues or unet or whole_net" ) > gpurun_out/r6t_tests.txt 2>&1; grep -n "passed\|failed" gpurun_out/r6t_tests.txt; grep -n "^E " gpurun_out/r6t_tests.txt | head -5
( timeout 600 python tools/conv_labels.py --config 2 ) > gpurun_out/r6t_conv_labels_cfg2.txt 2>&1; grep "down" gpurun_out/r6t_conv_labels_cfg2.txt | head -8
( timeout 600 python tools/conv_labels.py --config 3 ) > gpurun_out/r6t_conv_labels_cfg3.txt 2>&1; grep "down" gpurun_out/r6t_conv_labels_cfg3.txt | head -8
B="python bench.py --no-cpu-baseline --no-parity"
for r in 1 2; do for v in 1 0; do
  ( SN_CONV_S2_SMALL=$v timeout 300 $B --steps 8 --warmup 3 ) > gpurun_out/r6t_bench_cfg2_small${v}_$r.json 2>> gpurun_out/r6t_bench.err
  ( SN_CONV_S2_SMALL=$v timeout 300 $B --config 3 --steps 3 --warmup 1 ) > gpurun_out/r6t_bench_cfg3_small${v}_$r.json 2>> gpurun_out/r6t_bench.err
  for c in 2 3; do python -c "
import json; d=json.load(open('gpurun_out/r6t_bench_cfg${c}_small${v}_$r.json')); print('cfg$c SN_CONV_S2_SMALL=$v:', d['value'], 'fps', d['ms_per_step'], 'ms', {k[:14]: v['ms_per_window'] for k, v in d['dominant_kernel']['by_template'].items() if 'conv' in k})"; done
done; done
