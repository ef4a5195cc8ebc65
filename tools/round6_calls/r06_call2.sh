#!/bin/bash
# Round 6, call 2: producer/consumer attention after the exact-max fix: correctness again, then where its time goes — static priority of the S-waves
# (0..3), fragment depth 6, knock-out probes (no O-wave MFMAs / no softmax arithmetic / no S-wave MFMAs: results invalid, timing only), and one
# counter pass on both kernels.
set +e
OUT=gpurun_out/r06_call2
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
X2V_ATTN_PC=1 timeout 120 tools/x2v_check attn > "$OUT/x2v_check_attn_pc.log" 2>&1; echo "x2v_check attn (pc) rc=$? $(grep -c PASS "$OUT/x2v_check_attn_pc.log") PASS $(grep -c FAIL "$OUT/x2v_check_attn_pc.log") FAIL" | tee -a "$OUT/summary.txt"
grep FAIL "$OUT/x2v_check_attn_pc.log" | head -20 >> "$OUT/summary.txt"
echo "v9      : $(X2V_ATTN_PC=0 timeout 120 tools/x2v_check pattn 12 75600 40 6 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
echo "pc      : $(X2V_ATTN_PC=1 timeout 120 tools/x2v_check pattn 12 75600 40 6 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
for v in prio0 prio1 prio3 depth6 knock1 knock2 knock3; do
  echo "pc_$v: $(X2V_ATTN_PC=1 LD_LIBRARY_PATH=tools/probes/ab/pc_$v timeout 120 tools/x2v_check pattn 12 75600 40 6 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
done
echo "v9      : $(X2V_ATTN_PC=0 timeout 120 tools/x2v_check pattn 12 75600 40 6 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
echo "pc      : $(X2V_ATTN_PC=1 timeout 120 tools/x2v_check pattn 12 75600 40 6 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
X2V_ATTN_PC=1 timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_full_size.py tests/test_gpu_rank_shapes.py tests/test_gpu_boundary.py -m gpu -q --timeout 600 -k "attention or attn" > "$OUT/pytest_attn_pc.log" 2>&1; echo "pytest attention (pc) rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a "$OUT/summary.txt"; tail -8 "$OUT/pytest_attn_pc.log" | cut -c1-240 >> "$OUT/summary.txt"
# counters: separate passes, kernel-trace only
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  for pc in 0 1; do
    (cd /tmp && X2V_ATTN_PC=$pc timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/pmc/pc${pc}_set$i" -o pmc -- "$GRAFT_REPO_ROOT/tools/x2v_check" pattn 12 75600 40 2 > "$GRAFT_REPO_ROOT/$OUT/pmc_pc${pc}_set$i.log" 2>&1)
  done
done
python tools/pmc_summary.py "$OUT/pmc" > "$OUT/pmc_summary.txt" 2>&1
find "$OUT/pmc" -name "*kernel_trace.csv" -size +5M -delete
echo "total $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
cat "$OUT/summary.txt"; cat "$OUT/pmc_summary.txt"
