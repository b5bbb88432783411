#!/bin/bash
# GPU box: count K2's non-repeatable rollouts (tools/debug/nondet3.py) for every build/ab/lib_v_*.so, then the product.
mkdir -p gpurun_out/bisect
for lib in build/ab/lib_v_*.so; do
  PHNN_LIB_PATH=$PWD/$lib timeout -k 10 120 python tools/debug/nondet3.py 2>&1 | tail -1 | tee -a gpurun_out/bisect/result.txt
done
timeout -k 10 120 python tools/debug/nondet3.py 2>&1 | tail -1 | tee -a gpurun_out/bisect/result.txt
