#!/bin/bash
# Round 4, call 16 (last GPU seconds): first contact of the continuous single-stream w8a8 GEMM (gemm256c8.hip): bit-equality vs gemm256.hip, a/b timings.
set +e
OUT=gpurun_out/r04_call16
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
timeout 75 python tools/gemm_fp8_continuous_check.py > "$OUT/check.jsonl" 2> "$OUT/check.err"; echo "rc=$?" > "$OUT/summary.txt"
cat "$OUT/check.jsonl" >> "$OUT/summary.txt"; tail -5 "$OUT/check.err" >> "$OUT/summary.txt"
cut -c1-1500 "$OUT/summary.txt"
