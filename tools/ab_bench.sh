#!/bin/bash
# same-box comparison of several builds of the library: tools/ab_bench.sh libA.so libB.so ...  (two alternating passes)
set -e
cd "$(dirname "$0")/.."
for rep in $(seq 1 ${AB_REPS:-2}); do
  for v in "$@"; do
    cp "$v" phnn_mpc_amd/csrc/libphnn_mpc.so
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$v', round(d['value']/1e6,3),'M/s K2',d['roofline']['launch_ms'],'K1',d['roofline']['k1_launch_ms'])"
  done
done
