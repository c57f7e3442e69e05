#!/usr/bin/env python
"""Where the spill code of a kernel sits relative to its MFMA stream (reads the -save-temps .s of `make asm`)."""
import re, sys
from collections import Counter
S = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "k_render"
lines = open(S).read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_ZN3nsr\w+:", l)]
for i, name in starts:
    if flt not in name: continue
    j = i
    while ".end_amdhsa_kernel" not in lines[j]: j += 1
    body = lines[i:j]
    mf = 0; ev = []
    for k, l in enumerate(body):
        if "v_mfma" in l: mf += 1
        if "scratch_" in l: ev.append((mf, k, l.strip()))
    print(name, "lines", len(body), "mfma", mf, "scratch ops", len(ev))
    c = Counter((e[0] // 64) * 64 for e in ev)
    print(sorted(c.items()))
