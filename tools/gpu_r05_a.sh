#!/bin/bash
# Round 5, GPU call A: the re-partitioned phase-1 kernel (row-list chunks, teams, border strips, 4 stagers at C = 64) -- correctness against the
# oracle, time per launch for every team size and against the 2-stager build, then the targeted parity tests and one bench line.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
( timeout 600 python tools/check_phase1.py --name gshift_deblur2 --teams 0,1,2,4,8 ) > gpurun_out/r5a_check_p1_deblur2.txt 2>&1
( timeout 600 python tools/check_phase1.py --name gshift_deblur1 --teams 0,1,2,4,8 ) > gpurun_out/r5a_check_p1_deblur1.txt 2>&1
( timeout 300 python tools/p1r_variants.py time base nsw2 ) > gpurun_out/r5a_variants.txt 2>&1
( timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "phase1 or gsts or squeeze or unit_parity or range_guard or tickets or gather" ) > gpurun_out/r5a_tests.txt 2>&1
( timeout 600 python bench.py --steps 10 --warmup 3 ) > gpurun_out/r5a_bench.json 2> gpurun_out/r5a_bench.err
tail -5 gpurun_out/r5a_check_p1_deblur2.txt gpurun_out/r5a_variants.txt gpurun_out/r5a_tests.txt
head -c 600 gpurun_out/r5a_bench.json
