#!/bin/bash
# Round 5, final GPU call: counters re-taken on the final sources (PMC files are bound to the hashes of the translation units), the whole -m gpu
# suite, smoke, and the bench lines of every configuration with their PMC traffic.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
bash tools/make_profiles_r05.sh r05 2 3 4 4fp32 > gpurun_out/r05_make_profiles_final.log 2>&1
tail -n 3 gpurun_out/r05_make_profiles_final.log
for f in r05_pmc_hbm_traffic_cfg2.json r05_pmc_hbm_traffic_cfg3.json r05_pmc_hbm_traffic_cfg4_bf16.json; do cp gpurun_out/$f profiles/$f; done     # this call's bench lines read them
( timeout 1500 python -m pytest tests -q -m gpu -x ) > gpurun_out/r05_final_tests.txt 2>&1
tail -n 4 gpurun_out/r05_final_tests.txt
( timeout 300 python __graft_entry__.py smoke ) > gpurun_out/r05_final_smoke.txt 2>&1; tail -n 2 gpurun_out/r05_final_smoke.txt
B="python bench.py --no-cpu-baseline --no-parity"
( timeout 300 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
( timeout 300 $B --config 3 --steps 4 --warmup 2 ) > gpurun_out/r05_bench_cfg3.json 2>> gpurun_out/r05_bench.err
( timeout 300 $B --config 4 --steps 4 --warmup 2 ) > gpurun_out/r05_bench_cfg4_bf16.json 2>> gpurun_out/r05_bench.err
for f in gpurun_out/r05_bench.json gpurun_out/r05_bench_cfg3.json gpurun_out/r05_bench_cfg4_bf16.json; do echo "$f: $(head -c 1000 $f | python -c 'import sys,json,re; s=sys.stdin.read(); m=re.search(r"\"value\": ([0-9.]+)", s); t=re.search(r"\"traffic\": ([0-9.a-z]+)", s); f=re.search(r"\"frac\": ([0-9.]+)", s); print(m.group(1), f.group(1), t.group(1))')"; done
