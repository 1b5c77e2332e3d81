#!/usr/bin/env python3
"""Run-length view of a kernel's instruction stream in a hipcc -save-temps .s file: M = MFMA, v = other vector ALU,
d = LDS, g = global memory, w = s_waitcnt, s = scalar, b = branch/label.  usage: isa_runs.py file.s mangled-name-prefix"""
import sys

s = open(sys.argv[1]).read().split("\n")
name = sys.argv[2]
start = next(i for i, l in enumerate(s) if l.startswith(name) and l.rstrip().split(";")[0].rstrip().endswith(":"))
end = next(i for i in range(start, len(s)) if s[i].startswith(".Lfunc_end"))
seq = []
for l in s[start + 1:end]:
    t = l.strip()
    if not t or t.startswith((".", ";", "//")):
        continue
    if t.split(";")[0].rstrip().endswith(":"):
        seq.append("|")
        continue
    op = t.split()[0]
    seq.append("M" if op.startswith("v_mfma") else "v" if op.startswith("v_") else "d" if op.startswith("ds_") else
               "g" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "w" if op.startswith("s_waitcnt") else
               "b" if op.startswith(("s_cbranch", "s_branch")) else "s" if op.startswith("s_") else "?")
txt = "".join(seq)
out, prev, cnt = [], None, 0
for c in txt:
    if c == prev:
        cnt += 1
    else:
        if prev:
            out.append(f"{prev}{cnt}" if cnt > 1 else prev)
        prev, cnt = c, 1
out.append(f"{prev}{cnt}")
print(f"{len(seq)} instructions: {txt.count('M')} MFMA, {txt.count('v')} VALU, {txt.count('d')} LDS, {txt.count('g')} global")
print(" ".join(out))
