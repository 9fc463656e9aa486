#!/bin/bash
# after the g1 store of the denoisers: the full GPU suite, the smoke, then the whole end-of-round measurement set on the final sources
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r6f2_tests.txt 2>&1
tail -3 gpurun_out/r6f2_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r6f2_smoke.txt 2>&1; tail -2 gpurun_out/r6f2_smoke.txt
bash tools/final_r06.sh r06 2>&1 | tail -12
