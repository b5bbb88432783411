#!/usr/bin/env python3
"""Print the instruction-class shape (run-length encoded) of one kernel from the gfx950 assembly.
usage: tools/asm_shape.py <substring of mangled kernel name> [start_token]
M = MFMA, r = ds_read, w = ds_write/bpermute, W(..) = s_waitcnt, T = v_exp/v_rcp, v = other VALU, s = SALU, G = global/scratch
"""
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
asm = "/tmp/phnn_shape.s"
key = sys.argv[1]
adjoint = "grad" in key or "vjp" in key  # the adjoint kernels live in phnn_grad.hip
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",  # as the Makefile
                "--cuda-device-only", "-S", "-o", asm,
                os.path.join(root, "phnn_mpc_amd/csrc", "phnn_grad.hip" if adjoint else "phnn_mpc.hip")],
               check=True, stderr=subprocess.DEVNULL)
s = open(asm).read()
key = sys.argv[1]
i = s.index(key)
i = s.index("\n", i)
j = s.index(".Lfunc_end", i)


def cat(l):
    l = l.strip()
    if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"):
        return None
    op = l.split()[0]
    if op.startswith("v_mfma"):
        return "M"
    if op.startswith("ds_read"):
        return "r"
    if op.startswith("ds_write") or op.startswith("ds_bperm"):
        return "w"
    if op.startswith("s_waitcnt"):
        return "W(" + l.split(None, 1)[1].split(";")[0].strip().replace("lgkmcnt", "L").replace("vmcnt", "V") + ")"
    if op.startswith("v_exp") or op.startswith("v_rcp"):
        return "T"
    if op.startswith("v_"):
        return "v"
    if op.startswith("s_"):
        return "s"
    if op.startswith(("global_", "scratch_", "buffer_")):
        return "G"
    return "?"


seq = [c for c in map(cat, s[i:j].split("\n")) if c]
out, prev, n = [], None, 0
for c in seq:
    if c == prev:
        n += 1
    else:
        if prev:
            out.append(prev + (str(n) if n > 1 else ""))
        prev, n = c, 1
out.append(prev + str(n))
txt = " ".join(out)
print(len(seq), "instructions;", sum(1 for c in seq if c == "M"), "MFMA;", sum(1 for c in seq if c == "r"), "ds_read;",
      sum(1 for c in seq if c == "G"), "global/scratch")
print(txt if len(sys.argv) < 3 else txt[int(sys.argv[2]):int(sys.argv[2]) + 4000])
