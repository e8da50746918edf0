#!/usr/bin/env python3
"""Per-basic-block instruction census of one kernel in a hipcc -S listing: MFMA / ds_read / VALU / scratch counts, to see
whether spills or stray VALU sit inside a K loop.  usage: tools/asm_blocks.py file.s mangled-name-substring"""
import re
import sys

s = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(s) if l.startswith("_Z") and key in l and ":" in l and not l.startswith("\t"))
end = next(i for i in range(start + 1, len(s)) if s[i].startswith(".Lfunc_end"))
blk = "entry"
order = []
stat = {}
for l in s[start + 1:end]:
    m = re.match(r"^(\.LBB\S+):", l)
    if m:
        blk = m.group(1)
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    op = t.split()[0]
    d = stat.setdefault(blk, {})
    if blk not in order:
        order.append(blk)
    for k, pat in (("mfma", "v_mfma"), ("ds_read", "ds_read"), ("ds_write", "ds_write"), ("glds", "global_load_lds"), ("gload", "global_load"),
                   ("gstore", "global_store"), ("scr_st", "scratch_store"), ("scr_ld", "scratch_load"), ("barrier", "s_barrier"),
                   ("waitcnt", "s_waitcnt"), ("branch", "s_cbranch")):
        if op.startswith(pat):
            d[k] = d.get(k, 0) + 1
    if op.startswith("v_") and not op.startswith("v_mfma"):
        d["valu"] = d.get("valu", 0) + 1
    if op.startswith("s_") and not op.startswith(("s_waitcnt", "s_barrier", "s_cbranch", "s_nop")):
        d["salu"] = d.get("salu", 0) + 1
    d["n"] = d.get("n", 0) + 1
for b in order:
    d = stat[b]
    if d.get("n", 0) >= 8:
        print("%-14s" % b, " ".join(f"{k}={v}" for k, v in d.items()))
