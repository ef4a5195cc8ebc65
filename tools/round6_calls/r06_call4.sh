#!/bin/bash
# Round 6, call 4: producer/consumer attention, second form (S-wave tile rotated by one chunk over a 3-slot K ring, row sums on the matrix pipe,
# packed-max trigger): correctness, clock, knock-outs, trace.
set +e
OUT=gpurun_out/r06_call4
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
X2V_ATTN_PC=1 timeout 120 tools/x2v_check attn > "$OUT/x2v_check_attn_pc.log" 2>&1; echo "x2v_check attn (pc) rc=$? $(grep -c PASS "$OUT/x2v_check_attn_pc.log") PASS $(grep -c FAIL "$OUT/x2v_check_attn_pc.log") FAIL" | tee -a "$OUT/summary.txt"
grep FAIL "$OUT/x2v_check_attn_pc.log" | head -20 >> "$OUT/summary.txt"
for rep in 1 2; do
echo "v9      : $(X2V_ATTN_PC=0 timeout 120 tools/x2v_check pattn 12 75600 40 6 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
echo "pc      : $(X2V_ATTN_PC=1 timeout 120 tools/x2v_check pattn 12 75600 40 6 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
done
for v in knock3 knock4 knock5; do
  echo "pc_$v: $(X2V_ATTN_PC=1 LD_LIBRARY_PATH=tools/probes/ab/pc_$v timeout 120 tools/x2v_check pattn 12 75600 40 6 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
done
for v in; do
  echo "== pc_$v" | tee -a "$OUT/summary.txt"
  X2V_DUMP_TRACE=1 X2V_ATTN_PC=1 LD_LIBRARY_PATH=tools/probes/ab/pc_$v timeout 120 tools/x2v_check pattn 12 75600 40 3 2>&1 | tail -9 | tee -a "$OUT/summary.txt"
done
X2V_ATTN_PC=1 timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_full_size.py tests/test_gpu_rank_shapes.py tests/test_gpu_boundary.py -m gpu -q --timeout 600 -k "attention or attn" > "$OUT/pytest_attn_pc.log" 2>&1; echo "pytest attention (pc) rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a "$OUT/summary.txt"; tail -8 "$OUT/pytest_attn_pc.log" | cut -c1-240 >> "$OUT/summary.txt"
echo "total $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
