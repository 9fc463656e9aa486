#!/bin/bash
# Round 6, GPU call X: K0 on the matrix cores as the default: GSTS / whole-net / temporal-split tests, config 2 and 3 windows against SN_K0_MFMA=0.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
( timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_temporal_split.py tests/test_gpu_io.py -x -q -m gpu -k "shiftconv or gsts or unit or whole_net or full_size or temporal_split or wavefront or streams or geometry or range_guard or cli" ) > gpurun_out/r6x_tests.txt 2>&1; grep -n "passed\|failed" gpurun_out/r6x_tests.txt; grep -n "^E " gpurun_out/r6x_tests.txt | head -5
B="python bench.py --no-cpu-baseline --no-parity"
for r in 1 2; do for v in 0 1; do
  ( SN_K0_MFMA=$v timeout 300 $B --steps 8 --warmup 3 ) > gpurun_out/r6x_bench_cfg2_k0m${v}_$r.json 2>> gpurun_out/r6x_bench.err
  ( SN_K0_MFMA=$v timeout 300 $B --config 3 --steps 3 --warmup 1 ) > gpurun_out/r6x_bench_cfg3_k0m${v}_$r.json 2>> gpurun_out/r6x_bench.err
  for c in 2 3; do python -c "
import json; d=json.load(open('gpurun_out/r6x_bench_cfg${c}_k0m${v}_$r.json')); print('cfg$c SN_K0_MFMA=$v:', d['value'], 'fps', d['ms_per_step'], 'ms', {k[:14]: v['ms_per_window'] for k, v in d['dominant_kernel']['by_template'].items() if 'shift' in k}, 'unit frac', d['roofline']['frac'])"; done
done; done
