#!/bin/bash
# End-of-round measurement set of round 6 (run through gpurun from the repo root): kernel stats + PMC traffic + SQ pass of configs 2, 3 and 4
# (bf16; fp32 kernel stats) on the final sources, then the bench lines of every configuration.  Everything lands in gpurun_out/; the PMC
# summaries are copied to profiles/ first so that the bench lines carry `traffic` (bench.py checks the source hashes recorded in them).
set -u
R=$(pwd)
TAG=${1:-r06}
bash tools/make_profiles_r06.sh $TAG 2 3 4 4fp32 > gpurun_out/${TAG}_make_profiles.log 2>&1
for c in cfg2 cfg3 cfg4_bf16; do cp gpurun_out/${TAG}_pmc_hbm_traffic_$c.json profiles/r06_pmc_hbm_traffic_$c.json; done
timeout 400 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --config 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_cfg3.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --config 4 --no-cpu-baseline > gpurun_out/${TAG}_bench_cfg4_bf16.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --config 4 --dtype fp32 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_cfg4_fp32_split.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --config 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_cfg5.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --config 6 --no-cpu-baseline > gpurun_out/${TAG}_bench_cfg6.json 2>> gpurun_out/${TAG}_bench.err
python - <<PY
import json
for n in ("bench", "bench_cfg3", "bench_cfg4_bf16", "bench_cfg4_fp32_split", "bench_cfg5", "bench_cfg6"):
    try:
        d = json.load(open("gpurun_out/${TAG}_%s.json" % n))
        r = d.get("roofline", {})
        w = d.get("whole_net_roofline", {})
        print(n, d["value"], d["ms_per_step"], d.get("ms_per_step_median"), "unit frac", r.get("frac"), "traffic", r.get("traffic"), "whole", w.get("frac"), w.get("traffic"), "peak GB", d.get("peak_device_memory_gb"))
    except Exception as e:
        print(n, "failed", e)
PY
grep -i "Traceback\|Error" gpurun_out/${TAG}_bench.err | head -5
