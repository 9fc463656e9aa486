#!/bin/bash
# Round 5, GPU call C: the whole -m gpu suite on the final phase-1 kernel, then one bench line per BASELINE config (+ the frame-wavefront
# schedule on config 2 and both product arithmetics of the fp32 engine on config 4).
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
( timeout 1500 python -m pytest tests -q -m gpu -x ) > gpurun_out/r5c_tests.txt 2>&1
tail -n 6 gpurun_out/r5c_tests.txt
B="python bench.py --no-cpu-baseline --no-parity"
( timeout 300 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r5c_bench_cfg2.json 2> gpurun_out/r5c_bench_cfg2.err
( timeout 300 $B --steps 10 --warmup 3 --schedule frame --frame-group 4 ) > gpurun_out/r5c_bench_cfg2_frame4.json 2>> gpurun_out/r5c_bench.err
( timeout 300 $B --steps 10 --warmup 3 --schedule frame --frame-group 2 ) > gpurun_out/r5c_bench_cfg2_frame2.json 2>> gpurun_out/r5c_bench.err
( timeout 300 $B --config 3 --steps 4 --warmup 2 ) > gpurun_out/r5c_bench_cfg3.json 2>> gpurun_out/r5c_bench.err
( timeout 300 $B --config 4 --steps 4 --warmup 2 ) > gpurun_out/r5c_bench_cfg4_bf16.json 2>> gpurun_out/r5c_bench.err
( timeout 300 $B --config 4 --dtype fp32 --steps 2 --warmup 1 ) > gpurun_out/r5c_bench_cfg4_fp32_split.json 2>> gpurun_out/r5c_bench.err
( timeout 300 $B --config 5 --steps 4 --warmup 2 ) > gpurun_out/r5c_bench_cfg5.json 2>> gpurun_out/r5c_bench.err
( timeout 300 $B --config 6 --steps 4 --warmup 2 ) > gpurun_out/r5c_bench_cfg6.json 2>> gpurun_out/r5c_bench.err
for f in gpurun_out/r5c_bench_*.json; do echo "$f: $(head -c 230 $f)"; done
