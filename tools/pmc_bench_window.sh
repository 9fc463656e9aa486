R=$(pwd); export TMPDIR=/tmp
timeout 300 python bench.py --no-cpu-baseline --variant gshift_deblur1 --height 1080 --width 1920 --one-len 12 --steps 2 --warmup 1 2>/dev/null | tail -1 > gpurun_out/cfg5_deblur1_1080p_T12.json
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcb_$c -- python $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1
done
python $R/tools/pmc_summary.py $R/gpurun_out/r01_v3_pmc_hbm_traffic_bench_window.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py (Shift-Net-s, 1280x720, one_len 16): averages per launch over ALL launches of a kernel in the window (all levels). Raw KB; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md)." $R/gpurun_out/pmcb_FETCH_SIZE $R/gpurun_out/pmcb_WRITE_SIZE
cd $R; rm -rf gpurun_out/pmcb_FETCH_SIZE gpurun_out/pmcb_WRITE_SIZE
