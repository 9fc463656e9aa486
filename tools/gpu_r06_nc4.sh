#!/bin/bash
# paired 16-byte LDS stores + region-aligned g2 items in the fused phase 1: parity of everything that runs the kernel, interleaved A/B (10 rounds)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x -k "phase1 or unit_parity or gsts_pieces or geometry or denoise_unit or temporal_split" > gpurun_out/r6nc4_tests.txt 2>&1
tail -3 gpurun_out/r6nc4_tests.txt
( timeout 1500 python tools/p1_ab.py --rounds 10 ) > gpurun_out/r6nc4_p1_ab.txt 2>&1
grep "^AB\|^==" gpurun_out/r6nc4_p1_ab.txt | grep -v "K0_\|K4_\|_team" | cut -c1-150
