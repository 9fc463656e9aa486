#!/bin/bash
# Round 6, GPU call Z: the walking form of K0 on the matrix cores (window as a ring of rows) as the default: parity, GSTS / whole-net / temporal-split tests,
# A/B against the tile form, config 2 / 3 windows.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
( timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_temporal_split.py -x -q -m gpu -k "shiftconv or gsts or unit or whole_net or full_size or temporal_split or wavefront or streams or geometry" ) > gpurun_out/r6z_tests.txt 2>&1; grep -n "passed\|failed" gpurun_out/r6z_tests.txt; grep -n "^E " gpurun_out/r6z_tests.txt | head -5
( timeout 600 python tools/k0_ab.py ) > gpurun_out/r6z_k0_ab.txt 2>&1; grep "^K0" gpurun_out/r6z_k0_ab.txt | cut -c1-105
B="python bench.py --no-cpu-baseline --no-parity"
( timeout 300 $B --steps 8 --warmup 3 ) > gpurun_out/r6z_bench_cfg2.json 2>> gpurun_out/r6z_bench.err
( timeout 300 $B --config 3 --steps 3 --warmup 1 ) > gpurun_out/r6z_bench_cfg3.json 2>> gpurun_out/r6z_bench.err
for c in 2 3; do python -c "
import json; d=json.load(open('gpurun_out/r6z_bench_cfg${c}.json')); print('cfg$c:', d['value'], 'fps', d['ms_per_step'], 'ms', {k[:14]: v['ms_per_window'] for k, v in d['dominant_kernel']['by_template'].items() if 'shift' in k}, 'unit frac', d['roofline']['frac'])"; done
