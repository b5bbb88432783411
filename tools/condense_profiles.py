#!/usr/bin/env python3
"""Condense gpurun_out/final/ (written by tools/collect_profiles.sh on the GPU box) into profiles/<prefix>_*:
kernel stats CSV, the rollout kernels' rows of the kernel trace, the two PMC CSVs reduced to the rollout kernels,
and profiles/traffic_measured.json = per-launch HBM bytes (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per kernel."""
import csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "final")
prefix = sys.argv[1] if len(sys.argv) > 1 else "r01_final"
dst = os.path.join(ROOT, "profiles")


def one(pattern):
    m = glob.glob(os.path.join(src, pattern), recursive=True)
    assert m, pattern
    return max(m, key=os.path.getmtime)  # gpurun merges successive collections into the same tree: take the latest


shutil.copy(one("trace/**/*kernel_stats.csv"), os.path.join(dst, prefix + "_kernel_stats.csv"))
with open(one("trace/**/*kernel_trace.csv")) as f, open(os.path.join(dst, prefix + "_kernel_trace_rollout.csv"), "w") as g:
    for n, line in enumerate(f):
        if n == 0 or "k_rollout" in line:
            g.write(line)
vals = {}
for tag, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    rows = [r for r in csv.DictReader(open(one(tag + "/**/*counter_collection.csv"))) if "k_rollout" in r["Kernel_Name"]]
    with open(os.path.join(dst, f"{prefix}_{tag}_size.csv"), "w") as g:
        g.write("Kernel_Name,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp\n")
        for r in rows:
            g.write('"%s",%s,%s,%s,%s\n' % (r["Kernel_Name"], r["Counter_Name"], r["Counter_Value"], r["Start_Timestamp"], r["End_Timestamp"]))
    for k in ("fwd", "grad"):
        v = [float(r["Counter_Value"]) for r in rows if "k_rollout_" + k in r["Kernel_Name"] and r["Counter_Name"] == counter]
        vals[(k, counter)] = sum(v) / len(v)
out = {"phnn_cartpole:euler:B65536:H50:stash": {}}
for name, k in (("K1", "fwd"), ("K2", "grad")):
    fk, wk = vals[(k, "FETCH_SIZE")], vals[(k, "WRITE_SIZE")]
    out["phnn_cartpole:euler:B65536:H50:stash"][name] = {"FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk,
                                                         "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
out["note"] = ("separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of bench.py --steps 3 (f16x2 kernels, stash = a2, q1, dH, R_net outputs); "
               "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE halves wide streaming reads)")
json.dump(out, open(os.path.join(dst, "traffic_measured.json"), "w"), indent=1)
for tag in ("wgrad", "rk4"):  # per-kernel stats of the training pass (f4) and of config 5 on the RK4 stash
    m = glob.glob(os.path.join(src, f"trace_{tag}/**/*kernel_stats.csv"), recursive=True)
    if m:
        shutil.copy(max(m, key=os.path.getmtime), os.path.join(dst, f"{prefix}_{tag}_kernel_stats.csv"))
        shutil.copy(os.path.join(src, f"{tag}_probe.txt"), os.path.join(dst, f"{prefix}_{tag}_probe.txt"))
for mode in ("bf16x3", "f32"):  # kernel stats of the 24-bit-exact product modes (their roofline.frac is checkable from these)
    m = glob.glob(os.path.join(src, f"trace_{mode}/**/*kernel_stats.csv"), recursive=True)
    if m:
        shutil.copy(max(m, key=os.path.getmtime), os.path.join(dst, f"{prefix}_{mode}_kernel_stats.csv"))
        shutil.copy(os.path.join(src, f"bench_under_rocprof_{mode}.json"), os.path.join(dst, f"{prefix}_bench_under_rocprof_{mode}.json"))
for a, b in (("bench.json", "_bench.json"), ("bench_under_rocprof.json", "_bench_under_rocprof.json"), ("other_configs.jsonl", "_other_configs.jsonl"),
             ("bench_config4.json", "_bench_config4.json"), ("latency_probe.txt", "_latency_probe.txt"),
             ("train_step_probe.txt", "_train_step_probe.txt"), ("parity_margin_trained.txt", "_parity_margin_trained.txt")):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, prefix + b))
print(json.dumps(out, indent=1))
