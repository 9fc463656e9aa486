#!/bin/bash
# Round 6, GPU call V: CAB's closed-form CALayer in one launch: tests, config 2 / 3 / 4 windows against SN_CABCA_TWO=1.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
( timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_temporal_split.py -x -q -m gpu -k "test_cab or fused_cab or unet or whole_net or temporal_split" ) > gpurun_out/r6v_tests.txt 2>&1; grep -n "passed\|failed" gpurun_out/r6v_tests.txt; grep -n "^E " gpurun_out/r6v_tests.txt | head -5
B="python bench.py --no-cpu-baseline --no-parity"
for r in 1 2; do for v in 1 0; do
  ( SN_CABCA_TWO=$v timeout 300 $B --steps 8 --warmup 3 ) > gpurun_out/r6v_bench_cfg2_two${v}_$r.json 2>> gpurun_out/r6v_bench.err
  ( SN_CABCA_TWO=$v timeout 300 $B --config 3 --steps 3 --warmup 1 ) > gpurun_out/r6v_bench_cfg3_two${v}_$r.json 2>> gpurun_out/r6v_bench.err
  ( SN_CABCA_TWO=$v timeout 300 $B --config 4 --steps 3 --warmup 1 ) > gpurun_out/r6v_bench_cfg4_two${v}_$r.json 2>> gpurun_out/r6v_bench.err
  for c in 2 3 4; do python -c "
import json; d=json.load(open('gpurun_out/r6v_bench_cfg${c}_two${v}_$r.json')); print('cfg$c SN_CABCA_TWO=$v:', d['value'], 'fps', d['ms_per_step'], 'ms', {k[:14]: v['ms_per_window'] for k, v in d['dominant_kernel']['by_template'].items() if 'cab_ca' in k}, d['kernels'].get('sn_cab_ca'))"; done
done; done
