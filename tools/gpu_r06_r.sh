#!/bin/bash
# Round 6, GPU call R: pixel-shuffle conv with permuted rows (8-byte stores), SkipUpSample as low-resolution 1x1 + sn_upsample2_add: tests, per-label
# conv timings, config 2 / 3 windows against SN_SKIPUP_LOWRES=0.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
( timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_io.py -x -q -m gpu -k "skip_upsample or conv_epilogues or test_conv or unet or whole_net or full_size or hipgraph or cli" ) > gpurun_out/r6r_tests.txt 2>&1; grep -n "passed\|failed" gpurun_out/r6r_tests.txt; grep -n "^E " gpurun_out/r6r_tests.txt | head -5
( timeout 600 python tools/conv_labels.py --config 2 ) > gpurun_out/r6r_conv_labels_cfg2.txt 2>&1; grep "not a plain" gpurun_out/r6r_conv_labels_cfg2.txt | head -12
( timeout 600 python tools/conv_labels.py --config 3 ) > gpurun_out/r6r_conv_labels_cfg3.txt 2>&1; grep "not a plain" gpurun_out/r6r_conv_labels_cfg3.txt | head -8
B="python bench.py --no-cpu-baseline --no-parity"
for r in 1 2; do for v in 0 1; do
  ( SN_SKIPUP_LOWRES=$v timeout 300 $B --steps 8 --warmup 3 ) > gpurun_out/r6r_bench_cfg2_lowres${v}_$r.json 2>> gpurun_out/r6r_bench.err
  ( SN_SKIPUP_LOWRES=$v timeout 300 $B --config 3 --steps 3 --warmup 1 ) > gpurun_out/r6r_bench_cfg3_lowres${v}_$r.json 2>> gpurun_out/r6r_bench.err
  for c in 2 3; do python -c "
import json; d=json.load(open('gpurun_out/r6r_bench_cfg${c}_lowres${v}_$r.json')); print('cfg$c SN_SKIPUP_LOWRES=$v:', d['value'], 'fps', d['ms_per_step'], 'ms', {k[:14]: v['ms_per_window'] for k, v in d['dominant_kernel']['by_template'].items()})"; done
done; done
grep -i "error\|Traceback" gpurun_out/r6r_bench.err | head -5
