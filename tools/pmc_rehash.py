#!/usr/bin/env python3
"""Add the per-translation-unit source record (csrc_files) to PMC summaries written before bench.py had it.

Only legitimate while the tree still IS the one the summary was measured on: the script refuses a file whose whole-tree csrc_hash differs
from the current sources, so the per-unit hashes it writes are those of the measured sources.
    python tools/pmc_rehash.py profiles/r04_pmc_hbm_traffic_cfg2.json ...
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_files, csrc_hash  # noqa: E402

for path in sys.argv[1:]:
    doc = json.load(open(path))
    if doc.get("csrc_hash") != csrc_hash():
        sys.exit(f"{path}: measured on csrc_hash {doc.get('csrc_hash')}, the tree is {csrc_hash()}: re-measure instead")
    doc["csrc_files"] = csrc_files(doc["kernels_per_window"].keys())
    out = {}
    for k, v in doc.items():                       # keep csrc_files next to csrc_hash
        out[k] = v
        if k == "csrc_hash":
            out["csrc_files"] = doc["csrc_files"]
    json.dump(out, open(path, "w"), indent=1)
    print(path, "->", doc["csrc_files"])
