#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing, split at s_barrier (the workgroup barriers separate the schedule's segments).

    python scripts/isa_segments.py attn.s attn_img_kernelILi4ELb1ELb1ELi2ELb0ELb0ELb0E [--dump]
"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")): return "vmem"
    if op.startswith("v_pk_"): return "valu_pk"
    if op.startswith(("v_exp", "v_rcp", "v_rsq", "v_sqrt", "v_log", "v_sin", "v_cos")): return "valu_trans"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    dump = "--dump" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.split(";")[0].rstrip().endswith(":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    seg, segs, labels = Counter(), [], []
    cur_label = "entry"
    for l in lines[start + 1:end + 1]:
        t = l.strip()
        if re.match(r"^[.\w$]+:", t):
            if dump: print(t.split(";")[0].strip())
            continue
        if not t or t.startswith((";", ".", "//")):
            continue
        op = t.split()[0]
        c = classify(op)
        seg[c] += 1
        if dump: print("   ", t.split(";")[0].strip())
        if c == "barrier":
            segs.append(seg)
            seg = Counter()
            if dump: print("  ---------------- barrier", len(segs))
    segs.append(seg)
    keys = ["mfma", "valu", "valu_trans", "valu_pk", "lds", "vmem", "salu", "waitcnt", "nop", "branch"]
    print("seg  " + " ".join(f"{k:>10}" for k in keys))
    tot = Counter()
    for i, s in enumerate(segs):
        print(f"{i:3d}  " + " ".join(f"{s[k]:10d}" for k in keys))
        tot.update(s)
    print("tot  " + " ".join(f"{tot[k]:10d}" for k in keys))


if __name__ == "__main__":
    main()
