#!/bin/bash
# Round 6, call 1: first contact of the producer/consumer attention kernel (attn_pc.hip, X2V_ATTN_PC=1): correctness (x2v_check attn incl. the
# spike / anti-aligned cases, the attention legs of the GPU suite), then A-B-A-B launch time against the ping-pong kernel.
set +e
OUT=gpurun_out/r06_call1
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
X2V_ATTN_PC=1 timeout 120 tools/x2v_check attn > "$OUT/x2v_check_attn_pc.log" 2>&1; echo "x2v_check attn (pc) rc=$? $(grep -c PASS "$OUT/x2v_check_attn_pc.log") PASS $(grep -c FAIL "$OUT/x2v_check_attn_pc.log") FAIL" | tee -a "$OUT/summary.txt"
grep FAIL "$OUT/x2v_check_attn_pc.log" | head -20 >> "$OUT/summary.txt"
for rep in 1 2; do
  for pc in 0 1; do
    echo "rep$rep pc=$pc H40: $(X2V_ATTN_PC=$pc timeout 120 tools/x2v_check pattn 12 75600 40 6 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
  done
done
for pc in 0 1; do
  echo "pc=$pc H5: $(X2V_ATTN_PC=$pc timeout 120 tools/x2v_check pattn 12 75600 5 24 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
  echo "pc=$pc 1.3B: $(X2V_ATTN_PC=$pc timeout 120 tools/x2v_check pattn 12 20280 12 60 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
  echo "pc=$pc cross-like Sk=512: $(X2V_ATTN_PC=$pc timeout 120 tools/x2v_check pattn 12 4096 40 50 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
done
X2V_ATTN_PC=1 timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_full_size.py tests/test_gpu_rank_shapes.py tests/test_gpu_boundary.py -m gpu -q --timeout 600 -k "attention or attn" > "$OUT/pytest_attn_pc.log" 2>&1; echo "pytest attention (pc) rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a "$OUT/summary.txt"; tail -8 "$OUT/pytest_attn_pc.log" | cut -c1-240 >> "$OUT/summary.txt"
echo "total $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
