#!/bin/bash
# Round 5, GPU call B: interleaved A/B of the phase-1 builds (round-4 library, current kernel, team sizes, 4 stagers, role / memory skip variants),
# the phase-1 / GSTS / guard / temporal-split tests on the row-block pool, one bench line.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
( timeout 600 python tools/p1_ab.py --rounds 6 --reps 4 ) > gpurun_out/r5b_p1_ab.txt 2>&1
( timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "phase1 or gsts or squeeze or unit_parity or range_guard or tickets or hipgraph or denoise_unit" ) > gpurun_out/r5b_tests.txt 2>&1
( timeout 900 python -m pytest tests/test_temporal_split.py -x -q -m gpu ) > gpurun_out/r5b_tests_split.txt 2>&1
( timeout 600 python bench.py --steps 10 --warmup 3 ) > gpurun_out/r5b_bench.json 2> gpurun_out/r5b_bench.err
grep "^AB\|^==" gpurun_out/r5b_p1_ab.txt
tail -n 5 gpurun_out/r5b_tests.txt gpurun_out/r5b_tests_split.txt
head -c 400 gpurun_out/r5b_bench.json
