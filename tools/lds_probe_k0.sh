R=$(pwd); export TMPDIR=/tmp; cd /tmp
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU"
for tag in rows tile; do
  LIB=$R/shift-net_amd/lib/libshiftnet_hip.so; [ $tag = tile ] && LIB=$R/shift-net_amd/lib/dev/libshiftnet_hip_k0tile.so
  timeout 300 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $R/gpurun_out/pmcl_$tag -- python $R/bench.py --no-cpu-baseline --no-parity --steps 1 --warmup 1 --lib $LIB > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$R/gpurun_out/pmcl_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        key = "K0" if "shiftconv" in k else "P1c2" if "cab_phase1_kernel<3" in k else "P1c1" if "cab_phase1_kernel<2" in k else "K4" if "scale_gemm" in k else "conv16" if "conv3_fast_kernel<1" in k else None
        if key and int(r["Grid_Size"]) > 60000:
            acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVES": n[key] += 1
for k, c in acc.items():
    w = c["SQ_WAVE_CYCLES"]
    print("$tag", k, "launches", n[k], "LDS insts/wave %.0f" % (c["SQ_INSTS_LDS"] / c["SQ_WAVES"]), "VALU/wave %.0f" % (c["SQ_INSTS_VALU"] / c["SQ_WAVES"]),
          "active_lds/wavecyc %.3f" % (c["SQ_ACTIVE_INST_LDS"] / w), "wait_lds/wavecyc %.3f" % (c["SQ_WAIT_INST_LDS"] / w),
          "bank_conflict/idx_active %.3f" % (c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1)), "idx_active per launch %.3g" % (c["SQ_LDS_IDX_ACTIVE"] / n[k]))
PY
  rm -rf $R/gpurun_out/pmcl_$tag
done
