#!/bin/bash
# clock / power of the fused phase 1 on random and all-zero activations
mkdir -p gpurun_out
export TMPDIR=/tmp
/opt/rocm/bin/rocm-smi --showclocks --showpower --json 2>&1 | head -c 1500 > gpurun_out/r6pwr_smi_raw.txt
timeout 600 python tools/power_probe.py 4 > gpurun_out/r6pwr_probe.txt 2>&1
grep -v amdgpu gpurun_out/r6pwr_probe.txt | cut -c1-330 | tail -60
