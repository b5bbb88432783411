#!/usr/bin/env python3
"""Insert instructions behind chosen MFMAs / behind the n-th instruction after a chosen MFMA of one kernel.
usage: asm_edit_at.py in.s out.s <kernel> <spec> [<spec> ...]     spec = "<mfma index>[+<k>]:<text>"
  "42:s_nop 15"      s_nop 15 right behind v_mfma_f32_16x16x32 number 42 (program order, 0-based)
  "43+2:s_nop 7"     s_nop 7 behind the 2nd instruction after MFMA 43"""
import sys

src, dst, kern = sys.argv[1:4]
specs = {}
for sp in sys.argv[4:]:
    where, text = sp.split(":", 1)
    idx, off = (where.split("+") + ["0"])[:2]
    specs.setdefault(int(idx), []).append((int(off), text))
L = open(src).read().split("\n")
start = next(i for i, l in enumerate(L) if l.startswith(kern + ":"))
end = next(i for i in range(start, len(L)) if L[i].startswith(".Lfunc_end"))
ins = {}
n = 0
for i in range(start, end):
    if L[i].strip().startswith("v_mfma_f32_16x16x32"):
        for off, text in specs.get(n, []):
            j, seen = i, 0
            while seen < off:
                j += 1
                t = L[j].split(";")[0].strip()
                if t and not t.startswith(".") and not t.endswith(":"):
                    seen += 1
            ins.setdefault(j, []).append(text)
        n += 1
out = []
for i, l in enumerate(L):
    out.append(l)
    for t in ins.get(i, []):
        out.append("\t" + t)
open(dst, "w").write("\n".join(out))
print("inserted", sum(len(v) for v in ins.values()))
