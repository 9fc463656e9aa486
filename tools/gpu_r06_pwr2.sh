#!/bin/bash
# clock / package power of the whole hot path: rocm-smi polled while bench.py runs configs 2 and 3
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in 2 3; do
  python bench.py --no-parity --no-cpu-baseline --config $cfg --steps 40 --warmup 3 > gpurun_out/r6pwr_bench_cfg$cfg.json 2> gpurun_out/r6pwr_bench_cfg$cfg.err &
  BP=$!
  : > gpurun_out/r6pwr_bench_cfg${cfg}_smi.txt
  while kill -0 $BP 2>/dev/null; do
    /opt/rocm/bin/rocm-smi --showclocks --showpower --json 2>/dev/null | python -c "
import sys, json, time
try:
    d = json.load(sys.stdin); c = d[sorted(d)[0]]
    print(round(time.time(), 2), c.get('sclk clock speed:'), c.get('Current Socket Graphics Package Power (W)'))
except Exception as e: print('err', e)" >> gpurun_out/r6pwr_bench_cfg${cfg}_smi.txt
    sleep 0.25
  done
  wait $BP
  python - <<PY
import json, re
rows = [l.split() for l in open('gpurun_out/r6pwr_bench_cfg${cfg}_smi.txt') if not l.startswith('err')]
pts = [(float(r[0]), int(re.sub(r'\D', '', r[1])), float(r[2])) for r in rows if len(r) == 3]
busy = [p for p in pts if p[2] > 900]
line = json.loads(open('gpurun_out/r6pwr_bench_cfg${cfg}.json').read().strip().splitlines()[-1])
if busy:
    s = sorted(p[1] for p in busy); w = sorted(p[2] for p in busy)
    print('cfg${cfg}: %.1f frames/s; %d samples above 900 W of %d: sclk MHz min / median / max %d / %d / %d, package power W median / max %.0f / %.0f' % (line['value'], len(busy), len(pts), s[0], s[len(s)//2], s[-1], w[len(w)//2], w[-1]))
else:
    print('cfg${cfg}: no busy samples', len(pts))
PY
done 2>&1 | tee gpurun_out/r6pwr_bench_summary.txt
