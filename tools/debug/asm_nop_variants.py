#!/usr/bin/env python3
"""Hazard bisection on the packed-f32 build of the adjoint unit: insert s_nop around one class of
instructions in the gfx950 assembly and see (tools/debug/nondet3.py on the GPU box) whether K2's run-to-run differences
disappear.  usage: asm_nop_variants.py grad_pk.s outdir  ->  outdir/<rule>.s for every rule below.
s_nop never changes results; a rule that removes the differences names the instruction pair that needs the wait."""
import re
import sys

src, outdir = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")


def op(l):
    t = l.split(";")[0].strip()
    return t.split()[0] if t and not t.startswith(".") and not t.endswith(":") else ""


def regs(tok):
    """VGPR numbers named by an operand token: v12, v[12:15], |v12|, -v12"""
    m = re.search(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.search(r"\bv(\d+)\b", tok)
    return {int(m.group(1))} if m else set()


def operands(l):
    t = l.split(";")[0].strip()
    parts = t.split(None, 1)
    return [x.strip() for x in parts[1].split(",")] if len(parts) > 1 else []


def is_valu(o):
    return o.startswith("v_") and not o.startswith("v_mfma")


RULES = {}


def rule(f):
    RULES[f.__name__] = f
    return f


@rule
def base(i, o):  # unchanged packed build: expected to show the differences
    return None, None


@rule
def after_pk(i, o):  # anything consuming / following a packed-f32 result too early
    return None, ("s_nop 1" if o.startswith("v_pk_") else None)


@rule
def before_pk(i, o):  # a packed-f32 op reading something written just before
    return ("s_nop 1" if o.startswith("v_pk_") else None), None


@rule
def after_xdl(i, o):  # write-after-read on the operands of a 16-bit MFMA, or too-early reads of its result
    return None, ("s_nop 7" if o.startswith("v_mfma_f32_16x16x32") else None)


@rule
def before_xdl(i, o):  # VALU result consumed by a 16-bit MFMA too early
    return ("s_nop 3" if o.startswith("v_mfma_f32_16x16x32") else None), None


@rule
def around_sgemm(i, o):  # the f32 MFMAs (4x4x1, 16x16x4) run on the vector ALUs, next to the packed ops
    f32 = o.startswith("v_mfma_f32_4x4x1") or o.startswith("v_mfma_f32_16x16x4_f32")
    return ("s_nop 3" if f32 else None), ("s_nop 7" if f32 else None)


@rule
def after_trans(i, o):  # exp / rcp result consumed by a packed op (trans forwarding)
    return None, ("s_nop 1" if o.startswith(("v_exp_", "v_rcp_")) else None)


@rule
def after_cvt_mix(i, o):  # the f16 split: cvt_pk -> fma_mix (inline asm) -> cvt_pk
    return None, ("s_nop 1" if o.startswith(("v_cvt_pk_f16", "v_fma_mix")) else None)


@rule
def after_lds_vmem_wait(i, o):  # a packed op right behind the s_waitcnt that covers its operand
    return None, ("s_nop 3" if o == "s_waitcnt" else None)


@rule
def war_ab_only(i, o):  # s_nop 7 after a 16-bit MFMA only when one of the next 3 VALU ops overwrites its A / B registers
    if not o.startswith("v_mfma_f32_16x16x32"):
        return None, None
    ops_ = operands(lines[i])
    ab = regs(ops_[1]) | regs(ops_[2])
    seen = 0
    for l in lines[i + 1:i + 12]:
        oo = op(l)
        if not oo:
            continue
        if is_valu(oo):
            seen += 1
            if regs(operands(l)[0]) & ab:
                return None, "s_nop 7"
            if seen >= 3:
                break
        if oo.startswith("v_mfma"):
            break
    return None, None


for name, f in RULES.items():
    out, n = [], 0
    for i, l in enumerate(lines):
        o = op(l)
        pre, post = f(i, o) if o else (None, None)
        if pre:
            out.append("\t" + pre)
            n += 1
        out.append(l)
        if post:
            out.append("\t" + post)
            n += 1
    open(f"{outdir}/{name}.s", "w").write("\n".join(out))
    print(f"{name}: {n} s_nop inserted")
