#!/bin/bash
# Run on the MI355X box (gpurun): collects the evidence that profiles/ keeps for one round into gpurun_out/final/.
#   kernel trace + stats of the bench workload (default f16x2 mode, and the bf16x3 / all-f32 modes), FETCH_SIZE /
#   WRITE_SIZE PMC passes (separate runs, no trace domains), kernel traces of the training pass (f4) and of config 5
#   (RK4 stash), the plain bench line, the other BASELINE configurations, the single-plant latency probe, the end-to-end
#   training step and the trained-weights parity margins.  Every step appends to a file under gpurun_out/ (progress).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/final"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --steps 50 --warmup 5 --no-cpu-baseline --no-other-modes > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.log"
for mode in bf16x3 f32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$mode" -- python3 "$ROOT/bench.py" --steps 50 --warmup 5 --no-cpu-baseline --no-other-modes --matmul $mode > "$OUT/bench_under_rocprof_$mode.json" 2> "$OUT/trace_$mode.log"
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --preheat 0 --no-cpu-baseline --no-other-modes > /dev/null 2> "$OUT/pmc_fetch.log"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --preheat 0 --no-cpu-baseline --no-other-modes > /dev/null 2> "$OUT/pmc_write.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_wgrad" -- python3 "$ROOT/tools/wgrad_probe.py" > "$OUT/wgrad_probe.txt" 2> "$OUT/trace_wgrad.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_rk4" -- python3 "$ROOT/tools/rk4_probe.py" > "$OUT/rk4_probe.txt" 2> "$OUT/trace_rk4.log"
cd "$ROOT"
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python3 bench.py --config config4 --no-cpu-baseline --no-other-modes > "$OUT/bench_config4.json" 2> "$OUT/bench_config4.err"
python3 tools/bench_configs.py > "$OUT/other_configs.jsonl" 2> "$OUT/other_configs.err"
python3 tools/latency_probe.py > "$OUT/latency_probe.txt" 2> "$OUT/latency_probe.err"
python3 tools/train_step_probe.py 2>&1 | grep "training step" > "$OUT/train_step_probe.txt"
python3 tests/parity_margin_trained.py > "$OUT/parity_margin_trained.txt" 2> "$OUT/parity_margin_trained.err"
tail -c 400 "$OUT/bench.json"
