#!/bin/bash
# Round 6, GPU call Y: what the two K0 kernels really fetch: L2 hits / misses and the read requests of the L2 to the fabric (EA), per kernel.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
R=$PWD
export PYTHONPATH="$PWD:$PWD/shift-net_amd:$PYTHONPATH"
cd /tmp; export TMPDIR=/tmp
CMD="python $R/tools/k0_ab.py --rounds 1 --reps 1"
for cn in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum"; do
  tag=$(echo $cn | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $cn --kernel-trace --output-format csv -d $R/gpurun_out/pmck_$tag -- $CMD > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$R/gpurun_out/pmck_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "shiftconv" not in k: continue
        key = ("mfma" if "mfma" in k else "valu") + " grid " + r.get("Grid_Size", "?")
        tot[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[key][r["Counter_Name"]] += 1
with open("$R/gpurun_out/r6y_k0_l2_counters.txt", "w") as out:
    for key in sorted(tot):
        line = key + ": " + ", ".join(f"{c} {tot[key][c] / max(cnt[key][c], 1):.4g}" for c in sorted(tot[key])) + f"  (launches {max(cnt[key].values())})"
        print(line); out.write(line + "\n")
PY
rm -rf $R/gpurun_out/pmck_*
