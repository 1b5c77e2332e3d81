#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a hipcc -save-temps .s file: asm_blocks.py file.s <mangled-name-substring>"""
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2]
i = s.index(key)
i = s.index("\n", s.index(":", i))
j = s.index(".Lfunc_end", i)
blk = "entry"
stats = {blk: dict(mfma=0, scratch=0, valu=0, ds=0, glob=0, salu=0, wait=0, n=0)}
order = [blk]
for ln in s[i:j].split("\n"):
    t = ln.strip()
    if t.startswith(".LBB") and ":" in t:
        blk = t.split(":")[0]
        stats[blk] = dict(mfma=0, scratch=0, valu=0, ds=0, glob=0, salu=0, wait=0, n=0)
        order.append(blk)
        continue
    if not t or t.startswith(";") or t.startswith("."):
        continue
    st = stats[blk]
    st["n"] += 1
    op = t.split()[0]
    if "mfma" in op: st["mfma"] += 1
    elif op.startswith("scratch"): st["scratch"] += 1
    elif op.startswith("ds_"): st["ds"] += 1
    elif op.startswith("global") or op.startswith("buffer"): st["glob"] += 1
    elif op.startswith("s_waitcnt"): st["wait"] += 1
    elif op.startswith("v_"): st["valu"] += 1
    elif op.startswith("s_"): st["salu"] += 1
for b in order:
    if stats[b]["n"] > 8:
        print(b, stats[b])
