#!/bin/bash
# the denoisers' second pass from stored g1 rows (sn_phase1_opts.g1_store): parity, then config 4 with and without it, alternating
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu -x -k "denoisers_two_passes or (unit_parity and denoise) or (gsts_pieces and denoise) or denoise_unit or (temporal and denoise)" > gpurun_out/r6g1s_tests.txt 2>&1
tail -4 gpurun_out/r6g1s_tests.txt
for i in 1 2 3; do
  for v in 1 0; do
    SN_G1_STORE=$v python bench.py --no-parity --no-cpu-baseline --config 4 --steps 6 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('G1_STORE=$v cfg4 bf16', d['value'], 'frames/s', d['ms_per_step'], 'ms')"
  done
done | tee gpurun_out/r6g1s_cfg4_ab.txt
