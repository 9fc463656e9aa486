#!/usr/bin/env python3
"""Rough VGPR liveness profile of a gfx9 assembly listing (hipcc -S): for the instructions between two line numbers (a loop body,
treated as straight-line code with a back edge) print how many VGPRs are live after each instruction and where the maximum sits.
usage: vgpr_liveness.py file.s first_line last_line [every]"""
import re
import sys

RX = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(tok):
    out = set()
    for m in RX.finditer(tok):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def main():
    f, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    every = int(sys.argv[4]) if len(sys.argv) > 4 else 40
    lines = open(f).read().split("\n")[a - 1:b]
    ins = []
    for i, ln in enumerate(lines):
        s = ln.split(";")[0].strip()
        if not s or s.endswith(":") or s.startswith("."):
            continue
        op, _, rest = s.partition(" ")
        ops = [o.strip() for o in rest.split(",")] if rest else []
        if not ops:
            continue
        store_like = op.startswith(("ds_write", "global_store", "scratch_store", "buffer_store", "s_", "v_cmp", "ds_add")) or op in ("s_waitcnt",)
        if store_like:
            d, u = set(), set().union(*[regs(o) for o in ops]) if ops else set()
        else:
            d = regs(ops[0])
            u = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
            if op.startswith(("v_mfma", "v_fmac", "v_pk_fmac", "v_dot2c", "v_mac")) or "dpp" in op or "dpp" in s:
                u |= d                       # accumulate / tied-old forms read their destination
        ins.append((a + i, op, d, u, s))
    live = set()
    prof = [None] * len(ins)
    for _ in range(2):                       # second pass: values live around the back edge
        for k in range(len(ins) - 1, -1, -1):
            _, _, d, u, _ = ins[k]
            prof[k] = len(live)
            live = (live - d) | u
    mx = max(range(len(ins)), key=lambda k: prof[k])
    print(f"{len(ins)} instructions, max live VGPRs {prof[mx]} after line {ins[mx][0]}: {ins[mx][4][:80]}")
    for k in range(0, len(ins), every):
        print(f"  line {ins[k][0]:5d} live {prof[k]:4d}  {ins[k][4][:70]}")


if __name__ == "__main__":
    main()
