#!/bin/bash
# Round 6, GPU call N: guard_scope (one range-guard check per window of four quadrants): tests, then config 4 (bf16) HEAD vs the round-4 tree again.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
R=$PWD
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_io.py -x -q -m gpu -k "range_guard or quadrant or cli or device_io" ) > gpurun_out/r6n_tests.txt 2>&1; tail -n 3 gpurun_out/r6n_tests.txt
unset PYTHONPATH
for r in 1 2 3; do for t in r04 head; do
  if [ $t = head ]; then d=$R; else d=$R/ab_trees/$t; fi
  ( cd $d && timeout 300 python bench.py --config 4 --no-cpu-baseline --no-parity --steps 3 --warmup 1 ) > gpurun_out/r6n_cfg4_${t}_$r.json 2>> gpurun_out/r6n_bench.err
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r6n_cfg4_${t}_$r.json'))
    print('cfg4 $t round $r:', d['value'], 'fps', d['ms_per_step'], 'ms; kernel sum', round(sum(v['ms_total'] for v in d.get('kernels',{}).values()),1))
except Exception as e:
    print('cfg4 $t round $r: FAILED', e)
PY
done; done
( cd $R && timeout 300 python bench.py --no-cpu-baseline --no-parity --steps 8 --warmup 3 ) > gpurun_out/r6n_cfg2_head.json 2>> gpurun_out/r6n_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r6n_cfg2_head.json')); print('cfg2 head:', d['value'], 'fps', d['ms_per_step'], 'ms median', d.get('ms_per_step_median'))"
grep -i "error\|Traceback" gpurun_out/r6n_bench.err | head
