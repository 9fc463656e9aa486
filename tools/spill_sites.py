#!/usr/bin/env python3
"""Where the scratch spills / reloads of each kernel of a gfx950 .s file sit: per kernel, every scratch_ instruction with the innermost
backward-branch loop that contains it (label, length in lines, MFMA count of the loop).  A spill inside a step loop costs every step; one in a
prologue / epilogue does not.  usage: spill_sites.py file.s [kernel-name-substring]"""
import re
import sys


def main():
    txt = open(sys.argv[1]).read().split("\n")
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    starts = [(i, l.split(":")[0]) for i, l in enumerate(txt) if re.match(r"^_Z\w+:", l)]
    for k, (a, name) in enumerate(starts):
        b = starts[k + 1][0] if k + 1 < len(starts) else len(txt)
        if pat not in name:
            continue
        body = txt[a:b]
        end = next((i for i, l in enumerate(body) if l.strip().startswith("s_endpgm")), len(body))
        body = body[:end + 1]
        labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        loops = []
        for i, l in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i, m.group(1)))
        sc = [(i, l.strip()) for i, l in enumerate(body) if "scratch_" in l]
        if not sc:
            print(f"{name}: no scratch traffic ({len(body)} lines)")
            continue
        print(f"{name}: {len(sc)} scratch instructions")
        summary = {}
        for i, l in sc:
            inner = [lp for lp in loops if lp[0] <= i <= lp[1]]
            if inner:
                lp = min(inner, key=lambda t: t[1] - t[0])
                nm = sum(1 for x in body[lp[0]:lp[1]] if "v_mfma" in x)
                key = f"loop {lp[2]} ({lp[1] - lp[0]} lines, {nm} mfma, depth {len(inner)})"
            else:
                key = "outside loops"
            summary.setdefault(key, []).append(l.split()[0])
        for k2, v in summary.items():
            st = sum(1 for x in v if "store" in x)
            print(f"    {k2}: {st} stores, {len(v) - st} loads")


if __name__ == "__main__":
    main()
