#!/bin/bash
# Round 6, GPU call M: (1) tests of the round's host-side changes; (2) VERDICT r05 item 3: config 4 (bf16) whole-window A/B of the round-4 tree,
# the round-5 tree and HEAD, interleaved (ab_trees/: git archive of dcb17ab and 92062c0, libraries built here; not committed).
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
R=$PWD
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streaming_fused_cab or range_guard" ) > gpurun_out/r6m_tests.txt 2>&1; tail -n 3 gpurun_out/r6m_tests.txt
unset PYTHONPATH
for r in 1 2 3; do for t in r04 r05 head; do
  if [ $t = head ]; then d=$R; else d=$R/ab_trees/$t; fi
  ( cd $d && timeout 300 python bench.py --config 4 --no-cpu-baseline --no-parity --steps 3 --warmup 1 ) > gpurun_out/r6m_cfg4_${t}_$r.json 2>> gpurun_out/r6m_bench.err
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r6m_cfg4_${t}_$r.json'))
    ks={k: round(v['ms_total'],1) for k,v in d.get('kernels',{}).items() if v['ms_total']>=4}
    print('cfg4 $t round $r:', d['value'], 'fps', d['ms_per_step'], 'ms', ks)
except Exception as e:
    print('cfg4 $t round $r: FAILED', e)
PY
done; done
tail -n 5 gpurun_out/r6m_bench.err
