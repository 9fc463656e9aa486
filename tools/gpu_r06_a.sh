#!/bin/bash
# Round 6, GPU call A: the streams schedule (frame groups on concurrent HIP streams) -- bit-identity tests, interleaved bench A/B against unit-major,
# a kernel trace of the overlap, and the missing evidence for f2: one FETCH_SIZE / WRITE_SIZE pass of --schedule frame.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
R=$PWD
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streams or wavefront" ) > gpurun_out/r6a_tests.txt 2>&1; tail -n 3 gpurun_out/r6a_tests.txt
B="python bench.py --no-cpu-baseline --no-parity --steps 8 --warmup 3"
for rep in 1 2; do
  for s in "unit" "streams --stream-groups 2" "streams --stream-groups 3" "streams --stream-groups 4"; do
    tag=$(echo $s | tr -d ' -'); 
    ( timeout 300 $B --schedule $s ) > gpurun_out/r6a_bench_${tag}_$rep.json 2>> gpurun_out/r6a_bench.err
    python - <<PY
import json
d=json.load(open("gpurun_out/r6a_bench_${tag}_$rep.json"))
r=d["roofline"]
print("$s rep $rep:", d["value"], "fps", d["ms_per_step"], "ms; unit frac", r["frac"], "chain wall", r.get("gsts_chain_wall_ms"), "kernel sum", r.get("gsts_kernel_sum_ms"))
PY
  done
done
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_streams -- python $R/bench.py --no-cpu-baseline --no-parity --steps 1 --warmup 1 --schedule streams --stream-groups 2 > $R/gpurun_out/r6a_trace_streams.log 2>&1
DB=$(find $R/gpurun_out/prof_streams -name "*.db" | head -1)
python $R/tools/stream_overlap_timeline.py "$DB" $R/gpurun_out/r06_streams_overlap_timeline_cfg2.txt
python $R/tools/rocprof_summary.py "$DB" $R/gpurun_out/r06_cfg2_streams_kernel_stats.csv
rm -rf $R/gpurun_out/prof_streams
# f2 evidence: HBM traffic of the frame wavefront (groups of 4) against unit-major, same passes as profiles/r05_pmc_hbm_traffic_cfg2.json
for sch in frame unit; do
  for cn in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $cn --kernel-trace --output-format csv -d $R/gpurun_out/pmcb_$cn -- python $R/bench.py --no-cpu-baseline --no-parity --config 2 --schedule $sch --steps 2 --warmup 1 > /dev/null 2>&1
  done
  python $R/tools/pmc_summary.py $R/gpurun_out/r06_pmc_hbm_traffic_cfg2_schedule_$sch.json 4 "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py --no-parity --no-cpu-baseline --config 2 --schedule $sch --steps 2 --warmup 1 (4 traced windows). Raw KB; HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE." $R/gpurun_out/pmcb_FETCH_SIZE $R/gpurun_out/pmcb_WRITE_SIZE
  rm -rf $R/gpurun_out/pmcb_FETCH_SIZE $R/gpurun_out/pmcb_WRITE_SIZE
done
