#!/bin/bash
# End-of-round measurement set (run through gpurun from the repo root): PMC + kernel stats of configs 2 and 3 on the final sources, the bench
# lines of configs 2-6 (4 in bf16, float32 split and float32 exact), kernel stats of config 4 in float32.  Everything lands in gpurun_out/.
set -u
R=$(pwd)
TAG=${1:-r04c}
bash tools/make_profiles_r04.sh $TAG 2 3
# the PMC summaries must sit under profiles/ for bench.py to pick them up (csrc_hash is checked there)
cp gpurun_out/${TAG}_pmc_hbm_traffic_cfg2.json profiles/r04_pmc_hbm_traffic_cfg2.json
cp gpurun_out/${TAG}_pmc_hbm_traffic_cfg3.json profiles/r04_pmc_hbm_traffic_cfg3.json
timeout 300 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --config 3 > gpurun_out/${TAG}_cfg3.json 2> /dev/null
timeout 200 python bench.py --config 4 --no-cpu-baseline > gpurun_out/${TAG}_cfg4_bf16.json 2> /dev/null
timeout 200 python bench.py --config 4 --dtype fp32 --fp32_exact --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_cfg4_fp32_exact.json 2> /dev/null
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cfg4f -- python $R/bench.py --no-cpu-baseline --no-parity --config 4 --dtype fp32 --steps 1 --warmup 1 > /dev/null 2>&1
DB=$(find $R/gpurun_out/prof_cfg4f -name "*.db" | head -1)
python $R/tools/rocprof_summary.py "$DB" $R/gpurun_out/${TAG}_cfg4_fp32_kernel_stats.csv
rm -rf $R/gpurun_out/prof_cfg4f
cd $R
python - <<PY
import json
for n in ("bench", "cfg3", "cfg4_bf16", "cfg4_fp32_exact"):
    try:
        d = json.load(open("gpurun_out/${TAG}_%s.json" % n))
        r = d.get("roofline", {})
        print(n, d["value"], d["ms_per_step"], r.get("frac"), r.get("traffic"))
    except Exception as e:
        print(n, "failed", e)
PY
