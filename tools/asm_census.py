#!/usr/bin/env python3
"""Opcode histogram of the innermost (largest) loop of one kernel in a gfx950 assembly file.
usage: tools/asm_census.py <file.s> <mangled kernel name> [top]
The time-step loop is found as the backward branch spanning the most instructions."""
import collections
import re
import sys

s = open(sys.argv[1]).read()
name = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
i = s.index("\n" + name + ":")
j = s.index(".Lfunc_end", i)
lines = [l.split(";")[0].strip() for l in s[i:j].split("\n")]
labels = {l[:-1]: k for k, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:$", l)}
best = None
for k, l in enumerate(lines):
    m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"s_branch\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < k:
        if best is None or k - labels[m.group(1)] > best[1] - best[0]:
            best = (labels[m.group(1)], k)
body = [l.split()[0] for l in lines[best[0]:best[1]] if l and not l.startswith((";", ".")) and not l.endswith(":")]
c = collections.Counter(body)
print("loop instructions:", len(body))
groups = collections.Counter()
for op, n in c.items():
    g = ("mfma" if op.startswith("v_mfma") else "trans" if op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq"))
         else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else "salu" if op.startswith("s_") else "vmem")
    groups[g] += n
print(dict(groups))
for op, n in c.most_common(top):
    print(f"{n:6d} {op}")
