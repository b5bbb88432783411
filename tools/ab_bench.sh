#!/bin/bash
# Same-box comparison of several builds of the library: tools/ab_bench.sh libA.so libB.so ...  (alternating passes).
# Candidates are loaded through PHNN_LIB_PATH (phnn_mpc_amd/_capi.py); the product library
# phnn_mpc_amd/csrc/libphnn_mpc.so is never touched.
set -e
cd "$(dirname "$0")/.."
for rep in $(seq 1 ${AB_REPS:-2}); do
  for v in "$@"; do
    PHNN_LIB_PATH="$(realpath "$v")" python bench.py --steps ${AB_STEPS:-20} --warmup 3 --no-cpu-baseline --no-other-modes ${AB_ARGS} 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$v', round(d['value']/1e6,3),'M/s K2',d['roofline']['launch_ms'],'K1',d['roofline']['k1_launch_ms'])"
  done
done
