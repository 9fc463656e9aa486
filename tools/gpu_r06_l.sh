#!/bin/bash
# Round 6, GPU call L: streaming fused CAB, weights in LDS vs conv2's in registers; whole window with SN_CAB_FUSED=p16 vs 0 (interleaved).
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streaming_fused_cab" ) > gpurun_out/r6l_tests.txt 2>&1; tail -n 3 gpurun_out/r6l_tests.txt
( timeout 900 python tools/cab_ab.py --variants d,p,p/x2,p/x1 --cases 14x20x720x1280,14x12x1080x1920,14x20x360x640 ) > gpurun_out/r6l_cab_ab.txt 2>&1; grep "^AB\|^==\|Error\|error" gpurun_out/r6l_cab_ab.txt
B="python bench.py --no-cpu-baseline --no-parity"
for r in 1 2; do for v in 0 p16; do
  ( SN_CAB_FUSED=$v timeout 300 $B --steps 8 --warmup 3 ) > gpurun_out/r6l_bench_cfg2_cab${v}_$r.json 2>> gpurun_out/r6l_bench.err
  python -c "
import json; d=json.load(open('gpurun_out/r6l_bench_cfg2_cab${v}_$r.json')); print('cfg2 SN_CAB_FUSED=$v:', d['value'], 'fps', d['ms_per_step'], 'ms median', d.get('ms_per_step_median'), {k[:12]: v['ms_per_window'] for k, v in d['dominant_kernel']['by_template'].items()})"
done; done
