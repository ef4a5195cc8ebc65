#!/bin/bash
# Round 6, call 16: round-robin items; of the persistent short-walk attention form (attn_fwd_p9_kernel): x2v_check attn, the new GPU test, the A/B timing.
set +e
OUT=gpurun_out/r06_call16
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
timeout 300 tools/x2v_check attn > "$OUT/x2v_check_attn.log" 2>&1; echo "x2v_check attn rc=$?" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/x2v_check_attn.log" >> "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 600 -k "attention" > "$OUT/pytest_attn.log" 2>&1; echo "pytest attention rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a "$OUT/summary.txt"; tail -15 "$OUT/pytest_attn.log" | cut -c1-300 >> "$OUT/summary.txt"
timeout 600 python tools/probes/cross_attn_ab.py > "$OUT/cross_attn_ab.txt" 2>&1; cat "$OUT/cross_attn_ab.txt" | tee -a "$OUT/summary.txt"
echo "total $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
