#!/bin/bash
# which role of the fused phase 1 draws the power: the P1R_SKIP measurement builds under tools/power_probe.py
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/power_probe.py 3 --variants > gpurun_out/r6pwr_roles.txt 2>&1
grep -v amdgpu gpurun_out/r6pwr_roles.txt | python -c "
import sys, re
cur = None; acc = []
def flush():
    if cur and acc:
        s = sorted(a[0] for a in acc[1:]); w = sorted(a[1] for a in acc[1:])
        print(cur, '| sclk median %d MHz, power median %.0f W' % (s[len(s)//2], w[len(w)//2]))
for l in sys.stdin:
    if l.startswith('PWR'):
        flush(); cur = l.split(';')[0].strip(); acc = []
    elif 'sclk' in l:
        m = re.search(r\"sclk clock speed:': '\((\d+)Mhz\)'.*Power \(W\)': '([\d.]+)'\", l)
        if m: acc.append((int(m.group(1)), float(m.group(2))))
flush()" | tee gpurun_out/r6pwr_roles_summary.txt
