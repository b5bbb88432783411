#!/usr/bin/env python3
"""Average PMC counter values per kernel from rocprofv3 --pmc CSV output: tools/pmc_summary.py <dir> [<dir> ...]"""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "k_rollout" not in k:
                continue
            short = ("K1 " if "k_rollout_fwd" in k else "K2 ")
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(acc):
        print(d, k, "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(acc[k].items())))
