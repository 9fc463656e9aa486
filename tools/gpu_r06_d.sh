#!/bin/bash
# Round 6, GPU call D: whole GPU suite on the streaming conv kernel, then bench A/B (streaming vs tile kernel) on configs 2 and 3.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
( timeout 600 python tools/cab_ab.py --variants 0,t --cases 14x20x720x1280,18x20x360x640,22x20x180x320,24x52x720x1280 ) > gpurun_out/r6d_cab_ab.txt 2>&1; grep "^AB\|^==\|Error\|error" gpurun_out/r6d_cab_ab.txt
B="python bench.py --no-cpu-baseline --no-parity"
for rep in 1 2; do
  for v in 0 1; do
    ( SN_CONV_TILES=$v timeout 300 $B --steps 8 --warmup 3 ) > gpurun_out/r6d_bench_cfg2_tiles${v}_$rep.json 2>> gpurun_out/r6d_bench.err
    python -c "
import json; d=json.load(open('gpurun_out/r6d_bench_cfg2_tiles${v}_$rep.json')); print('cfg2 SN_CONV_TILES=$v rep $rep:', d['value'], 'fps', d['ms_per_step'], 'ms', {k: v['ms_per_window'] for k, v in d['dominant_kernel']['by_template'].items()})"
  done
done
for v in 0 1; do
  ( SN_CONV_TILES=$v timeout 300 $B --config 3 --steps 3 --warmup 1 ) > gpurun_out/r6d_bench_cfg3_tiles${v}.json 2>> gpurun_out/r6d_bench.err
  python -c "
import json; d=json.load(open('gpurun_out/r6d_bench_cfg3_tiles${v}.json')); print('cfg3 SN_CONV_TILES=$v:', d['value'], 'fps', d['ms_per_step'], 'ms', {k: v['ms_per_window'] for k, v in d['dominant_kernel']['by_template'].items()})"
done
( timeout 2400 python -m pytest tests -x -q -m gpu ) > gpurun_out/r6d_tests.txt 2>&1; tail -n 5 gpurun_out/r6d_tests.txt
