#!/bin/bash
# Round 5, GPU call G: the phase-1 step without its transcendentals / with two stagers / roles alone at four stagers (evidence for DESIGN 3.1c), then the
# counter passes re-taken on the final sources and the bench lines that read them.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
( timeout 600 python tools/p1_ab.py --rounds 6 --reps 4 ) > gpurun_out/r5g_p1_ab.txt 2>&1
grep "^AB\|^==" gpurun_out/r5g_p1_ab.txt | grep -v "K0_\|K4_"
bash tools/make_profiles_r05.sh r05 2 3 4 > gpurun_out/r05_make_profiles_final2.log 2>&1
tail -n 2 gpurun_out/r05_make_profiles_final2.log
for f in r05_pmc_hbm_traffic_cfg2.json r05_pmc_hbm_traffic_cfg3.json r05_pmc_hbm_traffic_cfg4_bf16.json; do cp gpurun_out/$f profiles/$f; done
B="python bench.py --no-cpu-baseline --no-parity"
( timeout 300 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
( timeout 300 $B --config 3 --steps 4 --warmup 2 ) > gpurun_out/r05_bench_cfg3.json 2>> gpurun_out/r05_bench.err
( timeout 300 $B --config 4 --steps 4 --warmup 2 ) > gpurun_out/r05_bench_cfg4_bf16.json 2>> gpurun_out/r05_bench.err
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "phase1 or gsts_pieces" ) > gpurun_out/r5g_tests.txt 2>&1; tail -n 2 gpurun_out/r5g_tests.txt
for f in gpurun_out/r05_bench.json gpurun_out/r05_bench_cfg3.json gpurun_out/r05_bench_cfg4_bf16.json; do echo "$f: $(head -c 1000 $f | python -c 'import sys,re; s=sys.stdin.read(); m=re.search(r"\"value\": ([0-9.]+)", s); t=re.search(r"\"traffic\": ([0-9.a-z]+)", s); f=re.search(r"\"frac\": ([0-9.]+)", s); print(m.group(1), f.group(1), t.group(1))')"; done
