#!/bin/bash
# does the launch time of the fused phase 1 depend on the operand data?  the shipped kernel on random and on all-zero activations, alternating
mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2 3; do
  P1R_DATA=randn timeout 300 python tools/p1r_variants.py time base base 2>&1 | grep VAR | sed 's/^/randn /'
  P1R_DATA=zero  timeout 300 python tools/p1r_variants.py time base base 2>&1 | grep VAR | sed 's/^/zero  /'
done > gpurun_out/r6nc5_data_dependence.txt
cat gpurun_out/r6nc5_data_dependence.txt
