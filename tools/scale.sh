#!/bin/bash
# The north-star scaling series on one node: bench.py --config 6 (Shift-Net+, 1920x1080, one_len 16: one window per GPU, weak scaling) at 1, 2, 4 and
# 8 GPUs.  One JSON line per N in scale_out/ with whole-job frames/s, per-rank ms per step and the halo-exchange time of every rank
# (clip_parallel.assemble_window: two raw frames to / from each neighbour, on a side stream).  usage: tools/scale.sh [steps] [warmup]
cd "$(dirname "$0")/.." || exit 1
ST=${1:-6}; WU=${2:-2}
mkdir -p scale_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for N in 1 2 4 8; do
  if [ "$(python -c 'import torch; print(torch.cuda.device_count())')" -lt "$N" ]; then echo "only $(python -c 'import torch; print(torch.cuda.device_count())') device(s): stopping before N=$N"; break; fi
  python bench.py --config 6 --gpus $N --steps $ST --warmup $WU --no-cpu-baseline > scale_out/cfg6_n$N.json 2> scale_out/cfg6_n$N.err
  python - "$N" <<'PY'
import json, sys
n = sys.argv[1]
d = json.loads([l for l in open(f"scale_out/cfg6_n{n}.json") if l.startswith("{")][-1])
print(f"N={n}: {d['value']:.1f} frames/s, {d['ms_per_step']:.1f} ms/step, unit frac {d['roofline']['frac']}, per rank {d.get('per_rank')}")
PY
done
