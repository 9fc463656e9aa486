#!/bin/bash
# paired 16-byte LDS stores in the fused phase 1: lane semantics, parity of everything that runs the kernel, interleaved A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/permlane_test tools/ubench/permlane_test.hip 2>/dev/null && /tmp/permlane_test | head -2 > gpurun_out/r6nc2_permlane.txt
cut -c1-400 gpurun_out/r6nc2_permlane.txt
timeout 1500 python -m pytest tests -q -m gpu -x -k "phase1 or unit_parity or gsts_pieces or geometry or denoise_unit or temporal_split" > gpurun_out/r6nc2_tests.txt 2>&1
tail -5 gpurun_out/r6nc2_tests.txt
( timeout 900 python tools/p1_ab.py ) > gpurun_out/r6nc2_p1_ab.txt 2>&1
grep -v amdgpu gpurun_out/r6nc2_p1_ab.txt | grep -v "K0_\|K4_" | cut -c1-150 | tail -40
