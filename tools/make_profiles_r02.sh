#!/bin/bash
# Round-2 profile artefacts, run through gpurun from the repo root: everything lands in gpurun_out/ (copy what is to be judged
# into profiles/).  Counters are collected in their OWN passes with --kernel-trace only (never with sys/hip traces).
#   1  rocprofv3 --kernel-trace --stats of bench.py                 -> r02_bench_kernel_stats.csv
#   2  rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes)     -> r02_pmc_hbm_traffic_bench_window.json
#   3  rocprofv3 --pmc SQ_* incl. the MFMA counters this build has   -> r02_pmc_sq_mfma_bench_window.json
#   4  bench lines of BASELINE configs 3, 4 (bf16 and fp32), 5       -> r02_cfg*.json
#   5  the full default bench line (config 2, with cpu_baseline)     -> r02_bench.json
set -u
TAG=${1:-r02}
R=$(pwd)
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -- python $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $R/gpurun_out/${TAG}_bench_under_rocprof.log 2>&1
DB=$(find $R/gpurun_out/prof_$TAG -name "*.db" | head -1)
python $R/tools/rocprof_summary.py "$DB" $R/gpurun_out/${TAG}_bench_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcb_$c -- python $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1
done
python $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_pmc_hbm_traffic_bench_window.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py (Shift-Net-s, 1280x720, one_len 16): averages per launch over ALL launches of a kernel in the window (all levels). Raw KB; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md): HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE." $R/gpurun_out/pmcb_FETCH_SIZE $R/gpurun_out/pmcb_WRITE_SIZE
# MFMA evidence: take the counters this rocprofv3 really lists (8 SQ slots per pass)
rocprofv3 -L > $R/gpurun_out/${TAG}_counters_available.txt 2>&1
SQ=$(python - <<PY
import re
txt = open("$R/gpurun_out/${TAG}_counters_available.txt").read()
want = ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_BF16",
        "SQ_INSTS_VALU_MFMA_BF16", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"]
have = [w for w in want if re.search(r"\\b" + w + r"\\b", txt)]
print(" ".join(have[:8]))
PY
)
echo "SQ counters used: $SQ" > $R/gpurun_out/${TAG}_pmc_sq.log
timeout 300 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $R/gpurun_out/pmcsq -- python $R/bench.py --no-cpu-baseline --steps 1 --warmup 1 >> $R/gpurun_out/${TAG}_pmc_sq.log 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_pmc_sq_mfma_bench_window.json "rocprofv3 --pmc $SQ (one pass) of bench.py (Shift-Net-s, 1280x720, one_len 16): per-launch averages over all launches of a kernel in the window. SQ_WAVE_CYCLES / SQ_BUSY_CYCLES count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES counts cycles (MI355X_MICROARCH.md)." $R/gpurun_out/pmcsq
cd $R
rm -rf gpurun_out/prof_$TAG gpurun_out/pmcb_FETCH_SIZE gpurun_out/pmcb_WRITE_SIZE gpurun_out/pmcsq
timeout 300 python bench.py --no-cpu-baseline --config 3 --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/${TAG}_cfg3_deblur1_720p_T48.json
timeout 300 python bench.py --no-cpu-baseline --config 4 --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/${TAG}_cfg4_denoise1_480p_T32_quadrants_bf16.json
timeout 600 python bench.py --no-cpu-baseline --config 4 --dtype fp32 --steps 1 --warmup 1 2>/dev/null | tail -1 > gpurun_out/${TAG}_cfg4_denoise1_480p_T32_quadrants_fp32.json
timeout 300 python bench.py --no-cpu-baseline --config 5 --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/${TAG}_cfg5_deblur1_1080p_T12.json
timeout 600 python bench.py 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json
tail -c 600 gpurun_out/${TAG}_bench.json
