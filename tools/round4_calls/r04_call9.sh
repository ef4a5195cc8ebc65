#!/bin/bash
# Round 4, call 9: the i2v block at Wan-14B 720p dimensions vs the oracle's rows (first run).
set +e
OUT=gpurun_out/r04_call9
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
timeout 600 python -m pytest tests/test_gpu_full_size.py -m gpu -q --timeout 600 -k "i2v" --durations=3 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/summary.txt"; tail -25 "$OUT/pytest.log" | cut -c1-300 >> "$OUT/summary.txt"
grep i2v gpurun_out/parity_summary.jsonl | tail -2 >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
