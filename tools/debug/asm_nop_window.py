#!/usr/bin/env python3
"""Region bisection of the nop-sensitivity of the packed-f32 K2: `s_nop 7` behind EVERY
v_mfma_f32_16x16x32_f16 makes every rollout non-repeatable; here the nops go behind the MFMAs number lo..hi-1 only (in
program order inside ONE kernel), so that successive GPU runs close in on the instruction whose delay matters.
usage: asm_nop_window.py in.s out.s <mangled kernel> lo hi ["nop text"]"""
import sys

src, dst, kern, lo, hi = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
nop = sys.argv[6] if len(sys.argv) > 6 else "s_nop 7"
L = open(src).read().split("\n")
start = next(i for i, l in enumerate(L) if l.startswith(kern + ":"))
end = next(i for i in range(start, len(L)) if L[i].startswith(".Lfunc_end"))
out, n, k = [], 0, 0
for i, l in enumerate(L):
    out.append(l)
    if start < i < end and l.strip().startswith("v_mfma_f32_16x16x32"):
        if lo <= n < hi:
            out.append("\t" + nop)
            k += 1
        n += 1
open(dst, "w").write("\n".join(out))
print(f"{n} MFMAs in the kernel, nops behind {k}")
