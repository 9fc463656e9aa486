#!/bin/bash
# Round 5, GPU call D: K0 with paired LDS reads and K4 with the LDS-DMA shortcut against the round-4 library (interleaved), their parity tests, bench.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
( timeout 600 python tools/p1_ab.py --rounds 6 --reps 4 --k0-only ) > gpurun_out/r5d_k0_k4_ab.txt 2>&1
( timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gsts or unit_parity or full_size or whole_net" ) > gpurun_out/r5d_tests.txt 2>&1
( timeout 900 python -m pytest tests/test_temporal_split.py -x -q -m gpu ) > gpurun_out/r5d_tests_split.txt 2>&1
( timeout 600 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r5d_bench.json 2> gpurun_out/r5d_bench.err
grep "^AB\|^==" gpurun_out/r5d_k0_k4_ab.txt
tail -n 4 gpurun_out/r5d_tests.txt gpurun_out/r5d_tests_split.txt
head -c 300 gpurun_out/r5d_bench.json
