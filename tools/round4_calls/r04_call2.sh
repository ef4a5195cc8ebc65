#!/bin/bash
# Round 4, call 2: first contact of the continuous-pipeline GEMM (gemm256c.hip): bit-equality with the one-tile-per-workgroup form, then A/B timings.
set +e
OUT=gpurun_out/r04_call2
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
timeout 240 python tools/gemm_continuous_check.py > "$OUT/gemm_continuous.json" 2> "$OUT/gemm_continuous.err"; echo "gemm_continuous rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a "$OUT/summary.txt"
cat "$OUT/gemm_continuous.json" | cut -c1-3000 >> "$OUT/summary.txt"; tail -12 "$OUT/gemm_continuous.err" | cut -c1-600 >> "$OUT/summary.txt"
for m in gemm misc; do timeout 120 tools/x2v_check $m > "$OUT/check_$m.log" 2>&1; echo "check_$m rc=$?" | tee -a "$OUT/summary.txt"; tail -2 "$OUT/check_$m.log" >> "$OUT/summary.txt"; done
cat "$OUT/summary.txt"
