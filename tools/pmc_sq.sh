#!/bin/bash
# SQ counter passes of the bench workload (separate runs, no trace domains): where the waves of K1 / K2 spend their cycles.
# Output: gpurun_out/pmc_sq/<pass>/..._counter_collection.csv ; summarise with tools/pmc_summary.py
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/pmc_sq"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  rocprofv3 --pmc $2 --output-format csv -d "$OUT/$1" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --preheat 0 --no-cpu-baseline --no-other-modes > /dev/null 2> "$OUT/$1.log"
}
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
run b "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
run c "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_IFETCH"
cd "$ROOT"
python3 - <<PY
import csv, glob, collections
for p in "abc":
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True)
    if not f:
        print("pass", p, "no output"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = "K1" if "k_rollout_fwd" in r["Kernel_Name"] else "K2" if "k_rollout_grad" in r["Kernel_Name"] else None
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        print(k, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())))
PY
