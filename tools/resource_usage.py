#!/usr/bin/env python3
"""Compact per-kernel register / scratch report for the gfx950 build (no GPU needed)."""
import os
import re
import subprocess
import sys

csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "phnn_mpc_amd", "csrc")
out = ""
for src, extra in (("phnn_mpc.hip", []), ("phnn_grad.hip", []), ("phnn_wgrad.hip", []), ("phnn_split.hip", [])):  # as the Makefile
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
           "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
           "--cuda-device-only", "-c", "-o", "/dev/null", src, "-Rpass-analysis=kernel-resource-usage"] + extra
    cmd += sys.argv[1:]
    out += subprocess.run(cmd, cwd=csrc, capture_output=True, text=True).stderr
rows, cur = [], {}
for line in out.splitlines():
    m = re.search(r"remark: (.*?) \[-Rpass", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        if cur:
            rows.append(cur)
        cur = {"name": t.split(": ", 1)[1]}
    elif ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
if cur:
    rows.append(cur)
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True,
                       text=True).stdout.splitlines()
for r, dem in zip(rows, names):
    dem = dem.replace("void ", "").replace("(RollParams)", "").replace("(PointParams)", "").replace("(WgradParams)", "")
    print("%-58s vgpr %4s agpr %3s sgpr %4s scratch %5s occ %2s lds %s" % (
        dem, r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("SGPRs", "?"), r.get("ScratchSize [bytes/lane]", "?"),
        r.get("Occupancy [waves/SIMD]", "?"), r.get("LDS Size [bytes/block]", "?")))
