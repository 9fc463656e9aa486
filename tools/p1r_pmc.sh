#!/bin/bash
# PMC passes (kernel trace only, one SQ counter set per pass) over the role-split fused phase-1 kernel alone (tools/p1r_variants.py time <variant>):
# where the wave cycles go (parked at s_waitcnt / barrier, issue stalls, issuing), VALU / MFMA / LDS pressure, LDS bank conflicts.
#   tools/p1r_pmc.sh [variant ...]   ->  gpurun_out/p1r_pmc_<variant>.txt
R=$(pwd); export TMPDIR=/tmp; cd /tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA"
P3="SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"
for v in "${@:-base}"; do
  out=$R/gpurun_out/p1r_pmc_$v.txt; : > $out
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    rm -rf /tmp/p1rpmc
    timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/p1rpmc -- python $R/tools/p1r_variants.py time $v > /tmp/p1rpmc.log 2>&1 || { echo "pass $i failed" >> $out; tail -5 /tmp/p1rpmc.log >> $out; continue; }
    python - >> $out <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(float)
for f in glob.glob("/tmp/p1rpmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "cab_phase1r_kernel" not in k: continue
        key = k[k.index("cab_phase1r_kernel"):][:34]
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": n[key] += 1
for k, c in sorted(acc.items()):
    print("pass $i", "$v", k, "launches", n[k], " ".join(f"{cn}={cv / n[k]:.4g}" for cn, cv in sorted(c.items())))
PY
  done
  cat $out
done
rm -rf /tmp/p1rpmc
