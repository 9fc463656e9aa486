#!/bin/bash
# Round 6, GPU call O: the whole GPU suite, the smoke entry point and the default bench line on the round's library (SN_CAB_FUSED default p16).
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
( timeout 3000 python -m pytest tests -x -q -m gpu ) > gpurun_out/r6o_tests.txt 2>&1; tail -n 5 gpurun_out/r6o_tests.txt
( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > gpurun_out/r6o_smoke.txt 2>&1; tail -n 2 gpurun_out/r6o_smoke.txt
( timeout 900 python bench.py ) > gpurun_out/r6o_bench.json 2> gpurun_out/r6o_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r6o_bench.json')); print({k: d[k] for k in ('value','ms_per_step','ms_per_step_median','host_transfers')}); print(d['roofline']['frac'], d['cpu_baseline'], d.get('parity'))"
