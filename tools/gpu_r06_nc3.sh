#!/bin/bash
# which of the paired 16-byte LDS stores of the fused phase 1 pay: interleaved A/B of the variants (tools/p1r_variants.py pair*, opad0) against the round's kernel ("old")
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python tools/p1_ab.py ) > gpurun_out/r6nc3_p1_ab.txt 2>&1
grep -v amdgpu gpurun_out/r6nc3_p1_ab.txt | grep -v "K0_\|K4_\|_team" | cut -c1-150 | tail -48
