#!/bin/bash
# Round 5, GPU call E: (a) does Infinity-Cache residency speed up a dense CAB?  (b) hipGraph replay on the launch-dense configurations.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
( timeout 300 python tools/cab_mall_probe.py ) > gpurun_out/r5e_cab_mall_probe.txt 2>&1
B="python bench.py --no-cpu-baseline --no-parity"
for g in 0 1; do
  ( SN_GRAPH=$g timeout 300 $B --config 4 --steps 4 --warmup 3 ) > gpurun_out/r5e_cfg4_bf16_graph$g.json 2>> gpurun_out/r5e.err
  ( SN_GRAPH=$g timeout 300 $B --config 2 --steps 10 --warmup 3 ) > gpurun_out/r5e_cfg2_graph$g.json 2>> gpurun_out/r5e.err
  ( SN_GRAPH=$g timeout 300 $B --config 3 --steps 4 --warmup 3 ) > gpurun_out/r5e_cfg3_graph$g.json 2>> gpurun_out/r5e.err
done
grep CABPROBE gpurun_out/r5e_cab_mall_probe.txt
for f in gpurun_out/r5e_cfg*.json; do echo "$f: $(head -c 200 $f | cut -c60-200)"; done
