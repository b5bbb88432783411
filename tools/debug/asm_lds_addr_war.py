#!/usr/bin/env python3
"""Count, per kernel, LDS instructions whose ADDRESS VGPR is overwritten by one of the next few instructions
(before the covering s_waitcnt) -- the pattern suspected behind the packed-f32 build's non-repeatable K2.
usage: asm_lds_addr_war.py file.s [kernel-substring] [window]"""
import re
import sys

src = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
win = int(sys.argv[3]) if len(sys.argv) > 3 else 3
s = open(src).read()


def regs(tok):
    m = re.search(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"\|?-?v(\d+)\|?", tok.strip())
    return {int(m.group(1))} if m else set()


for km in re.finditer(r"^(_Z\S+):\s*;", s, re.M):
    name = km.group(1)
    if sub not in name:
        continue
    i = km.end()
    j = s.index(".Lfunc_end", i)
    L = [l.split(";")[0].strip() for l in s[i:j].split("\n")]
    L = [l for l in L if l and not l.startswith(".") and not l.endswith(":")]
    hits = []
    for k, l in enumerate(L):
        if not l.startswith("ds_"):
            continue
        ops = [x.strip() for x in l.split(None, 1)[1].split(",")]
        op = l.split()[0]
        addr = regs(ops[0]) if op.startswith(("ds_write", "ds_bpermute")) else (regs(ops[1]) if len(ops) > 1 else set())
        if op.startswith("ds_bpermute"):
            addr = regs(ops[1])
        for d in range(1, win + 1):
            if k + d >= len(L):
                break
            n = L[k + d]
            if n.startswith("s_waitcnt"):
                break
            if n.startswith("v_") and not n.startswith("v_cmp") and len(n.split(None, 1)) > 1:
                dst = regs(n.split(None, 1)[1].split(",")[0])
                if dst & addr:
                    hits.append((k, l, d, n))
                    break
    print(f"{len(hits):4d} {name}")
    for h in hits[:6]:
        print("       ", h[1], " -> +%d: " % h[2], h[3])
