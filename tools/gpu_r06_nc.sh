#!/bin/bash
# LDS bank conflicts of the fused phase 1: conflict-free (wrong-result) addresses per ring, the upper bound of a re-pitch
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/p1r_variants.py time base nc7 nc23 nc31 base nc7 nc23 nc31 base nc7 nc23 nc31 base > gpurun_out/r6nc_variants2.txt 2>&1
tail -40 gpurun_out/r6nc_variants2.txt
