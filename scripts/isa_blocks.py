#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel (from the --dump of scripts/isa_segments.py on stdin or a file)."""
import re
import sys

L = open(sys.argv[1]).read().split("\n")
blk, cur = [], ["entry", 0, 0, 0, 0, 0, 0]


def cls(op):
    if op.startswith("v_mfma"): return 2
    if op.startswith("ds_"): return 3
    if op.startswith(("buffer_", "global_", "scratch_")): return 4
    if op.startswith("v_"): return 1
    if op.startswith("s_nop"): return 6
    return 5


for l in L:
    t = l.strip()
    if not t: continue
    if re.match(r"^\.LBB\d+_\d+:", t):
        blk.append(cur)
        cur = [t[:-1], 0, 0, 0, 0, 0, 0]
        continue
    if t.startswith("---"):
        cur[0] += " |BAR"
        continue
    op = t.split()[0]
    cur[cls(op)] += 1
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        cur[0] += " ->" + t.split()[-1].replace(".LBB", "")
blk.append(cur)
print("label".ljust(60), "valu mfma lds vmem salu nop")
for b in blk:
    print(b[0][:58].ljust(60), *[str(x).rjust(4) for x in b[1:]])
