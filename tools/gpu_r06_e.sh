#!/bin/bash
# Round 6, GPU call E: streaming conv with weights in LDS (40 / 48 / 64 channels), residual register sets; tests + A/B + bench.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streaming_conv or test_cab or test_conv or full_size_properties" ) > gpurun_out/r6e_tests.txt 2>&1; tail -n 4 gpurun_out/r6e_tests.txt
( timeout 600 python tools/cab_ab.py --variants 0,t --cases 14x20x720x1280,18x20x360x640,24x52x720x1280,36x52x360x640,48x52x180x320,64x20x360x640 ) > gpurun_out/r6e_cab_ab.txt 2>&1; grep "^AB\|^==\|Error\|error" gpurun_out/r6e_cab_ab.txt
B="python bench.py --no-cpu-baseline --no-parity"
for v in 0 1; do
  ( SN_CONV_TILES=$v timeout 300 $B --steps 8 --warmup 3 ) > gpurun_out/r6e_bench_cfg2_tiles${v}.json 2>> gpurun_out/r6e_bench.err
  ( SN_CONV_TILES=$v timeout 300 $B --config 3 --steps 3 --warmup 1 ) > gpurun_out/r6e_bench_cfg3_tiles${v}.json 2>> gpurun_out/r6e_bench.err
  for c in 2 3; do python -c "
import json; d=json.load(open('gpurun_out/r6e_bench_cfg${c}_tiles${v}.json')); print('cfg$c SN_CONV_TILES=$v:', d['value'], 'fps', d['ms_per_step'], 'ms', {k[:12]: v['ms_per_window'] for k, v in d['dominant_kernel']['by_template'].items()})"; done
done
