#!/bin/bash
# Round 5, GPU call F: K0 with its channel chunks split over two neighbouring workgroups per tile, against the round-4 library; parity; bench.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
( timeout 600 python tools/p1_ab.py --rounds 6 --reps 4 --k0-only ) > gpurun_out/r5f_k0_split_ab.txt 2>&1
( timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gsts or unit_parity or launch_geometry or wavefront" ) > gpurun_out/r5f_tests.txt 2>&1
( timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > gpurun_out/r5f_bench.json 2> gpurun_out/r5f_bench.err
( timeout 600 python bench.py --config 3 --steps 4 --warmup 2 --no-cpu-baseline --no-parity ) > gpurun_out/r5f_bench_cfg3.json 2>> gpurun_out/r5f_bench.err
grep "^AB.*K0\|^==" gpurun_out/r5f_k0_split_ab.txt
tail -n 4 gpurun_out/r5f_tests.txt
head -c 260 gpurun_out/r5f_bench.json; echo; head -c 260 gpurun_out/r5f_bench_cfg3.json
