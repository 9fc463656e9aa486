#!/usr/bin/env python3
"""Instruction mix of the largest loop of every kernel whose mangled name contains PATTERN (CPU only: compiles FILE to gfx950 ISA).
usage: isa_loop_stats.py csrc/file.hip PATTERN"""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

src, pat = sys.argv[1], sys.argv[2]
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-S", "--cuda-device-only",
                    "-o", out, src], check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
for st in [i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and pat in l]:
    en = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i])
    body = lines[st:en]
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    best = None
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i and (best is None or i - labels[m.group(1)] > best[1] - best[0]):
            best = (labels[m.group(1)], i)
    ins = [l.strip().split()[0] for l in body[best[0]:best[1]] if l.startswith("\t") and not l.strip().startswith((";", "."))]
    kind = Counter("mfma" if i.startswith("v_mfma") else "valu" if i.startswith("v_") else "lds" if i.startswith("ds_") else
                   "vmem" if i.startswith(("global_", "buffer_")) else "salu" if i.startswith("s_") else "other" for i in ins)
    print(lines[st].split(":")[0][:70], dict(kind))
    print("   ", Counter(i for i in ins if i.startswith("v_") and not i.startswith("v_mfma")).most_common(16))
