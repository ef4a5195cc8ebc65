#!/bin/bash
# Round 6, call 5: the producer / consumer attention probe in its final form (tools/probes/attn_pc.hip, variant 13 of x2v_check against a library
# from tools/probes/build_attn_pc.sh) vs the ping-pong kernel: correctness, A-B-A-B by the clock, and the counter passes the write-up cites.
set +e
OUT=gpurun_out/r06_call5
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=. LD_LIBRARY_PATH=tools/probes/ab/pc
t0=$(date +%s)
timeout 120 tools/x2v_check attn > "$OUT/x2v_check_attn.log" 2>&1; echo "x2v_check attn (variants 0 4 5 6 12 13) rc=$? $(grep -c PASS "$OUT/x2v_check_attn.log") PASS $(grep -c FAIL "$OUT/x2v_check_attn.log") FAIL; variant 13: $(grep -c 'variant=13 .*PASS' "$OUT/x2v_check_attn.log") PASS" | tee -a "$OUT/summary.txt"
for rep in 1 2 3; do
  echo "ping-pong (12): $(timeout 120 tools/x2v_check pattn 12 75600 40 6 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
  echo "prod/cons (13): $(timeout 120 tools/x2v_check pattn 13 75600 40 6 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
done
for sh in "75600 5 24" "20280 12 60" "119056 24 2"; do
  echo "ping-pong (12) $sh: $(timeout 120 tools/x2v_check pattn 12 $sh 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
  echo "prod/cons (13) $sh: $(timeout 120 tools/x2v_check pattn 13 $sh 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
done
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  for var in 12 13; do
    (cd /tmp && LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/probes/ab/pc timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/pmc/v${var}_set$i" -o pmc -- "$GRAFT_REPO_ROOT/tools/x2v_check" pattn $var 75600 40 2 > "$GRAFT_REPO_ROOT/$OUT/pmc_v${var}_set$i.log" 2>&1)
  done
done
python tools/pmc_summary.py "$OUT/pmc" > "$OUT/pmc_summary.txt" 2>&1
find "$OUT/pmc" -name "*kernel_trace.csv" -size +5M -delete
echo "total $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
cat "$OUT/summary.txt"; grep -A20 "attn_fwd" "$OUT/pmc_summary.txt"
