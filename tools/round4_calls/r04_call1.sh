#!/bin/bash
# Round 4, call 1: the new parity tests (rank-of-8 shapes, w8a8 triangle + 40-layer forward, reference through the plugin), the dist suite after the
# comm-stream change, and a first bench line with the in-run box calibration.
set +e
OUT=gpurun_out/r04_call1
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_rank_shapes.py tests/test_plugin_reference.py tests/test_gpu_dist.py -m gpu -q --timeout 900 --durations=25 > "$OUT/pytest_new.log" 2>&1; echo "pytest_new rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a "$OUT/summary.txt"
tail -40 "$OUT/pytest_new.log" | cut -c1-300 >> "$OUT/summary.txt"
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_full_size.py -m gpu -q --timeout 900 --durations=25 -k "fp8_block or hunyuan13b_block or full_forward" > "$OUT/pytest_full.log" 2>&1; echo "pytest_full rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a "$OUT/summary.txt"
tail -30 "$OUT/pytest_full.log" | cut -c1-300 >> "$OUT/summary.txt"
cp gpurun_out/parity_summary.jsonl "$OUT/" 2>/dev/null
t0=$(date +%s)
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-config1 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a "$OUT/summary.txt"
cat "$OUT/bench.json" >> "$OUT/summary.txt"; tail -5 "$OUT/bench.err" >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
