#!/usr/bin/env python3
"""Instruction mix of EVERY loop of a kernel in a gfx950 .s file (hipcc --save-temps), innermost loops with their VALU opcode histogram: the step
loops of the three roles of cab_phase1r_kernel are the innermost loops with 32 MFMAs (A), 48 / 56 MFMAs (B) and none but global loads (S, two
steps per trip).  usage: role_loops.py file.s KERNEL-NAME-SUBSTRING"""
import re,sys
from collections import Counter
lines=open(sys.argv[1]).read().split("\n"); pat=sys.argv[2]
for st in [i for i,l in enumerate(lines) if re.match(r"^_Z\S*:",l) and pat in l]:
    en=next(i for i in range(st,len(lines)) if "s_endpgm" in lines[i]); body=lines[st:en]
    labels={l.split(":")[0]:i for i,l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:",l)}
    loops=[]
    for i,l in enumerate(body):
        m=re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)",l)
        if m and m.group(1) in labels and labels[m.group(1)]<i: loops.append((labels[m.group(1)],i))
    print(lines[st].split(":")[0][:80])
    for a,b in sorted(loops):
        ins=[l.strip().split()[0] for l in body[a:b] if l.startswith("\t") and not l.strip().startswith((";","."))]
        if len(ins)<150: continue
        inner=[x for x in loops if x!=(a,b) and a<=x[0] and x[1]<=b and (x[1]-x[0])>100]
        k=Counter("mfma" if i.startswith("v_mfma") else "valu" if i.startswith("v_") else "lds" if i.startswith("ds_") else "vmem" if i.startswith(("global_","buffer_","scratch_")) else "salu" if i.startswith("s_") else "other" for i in ins)
        print(f"  loop lines {a}-{b} ({len(ins)} instrs, {len(inner)} big inner loops): {dict(k)}")
        if not inner:
            print("     ",Counter(i for i in ins if i.startswith("v_") and not i.startswith("v_mfma")).most_common(14))
