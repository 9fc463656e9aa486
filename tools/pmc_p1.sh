R=$(pwd); export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_p1a -- python $R/tools/p1_variants.py run 0 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_p1b -- python $R/tools/p1_variants.py run 0 > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/pmc_p1a", "gpurun_out/pmc_p1b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "cab_phase1" in r["Kernel_Name"]:
                k = "KS3" if "<3" in r["Kernel_Name"] else "KS2"
                e = acc[k][r["Counter_Name"]]; e[0] += float(r["Counter_Value"]); e[1] += 1
    for k, cs in acc.items():
        print(d, k, {c: round(v[0] / v[1]) for c, v in cs.items()})
PY
rm -rf gpurun_out/pmc_p1a gpurun_out/pmc_p1b
