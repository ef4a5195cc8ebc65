#!/bin/bash
# One gpurun call = everything we want from the GPU this iteration. Every stage has its own timeout and
# never aborts the script; all output lands under gpurun_out/ (merged back by gpurun).
set +e
OUT=gpurun_out/${RUN_TAG:-run}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
STAGES=${STAGES:-"check microbench pytest bench13 bench14 prof"}
echo "stages: $STAGES" | tee "$OUT/summary.txt"
rocm-smi --showproductname 2>/dev/null | head -8 >> "$OUT/summary.txt"
for st in $STAGES; do
  t0=$(date +%s)
  case $st in
    check)
      for m in ${CHECKS:-probe misc norm rope gemm attn fp8 conv}; do
        timeout 180 tools/x2v_check $m > "$OUT/check_$m.log" 2>&1; echo "check $m rc=$?" | tee -a "$OUT/summary.txt"
        tail -1 "$OUT/check_$m.log" >> "$OUT/summary.txt"
      done ;;
    microbench)
      timeout 600 tools/x2v_check ${MICRO:-benchbig} > "$OUT/microbench.log" 2>&1; echo "microbench rc=$?" | tee -a "$OUT/summary.txt" ;;
    pytest)
      timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/summary.txt"
      tail -3 "$OUT/pytest.log" >> "$OUT/summary.txt" ;;
    pytestall)
      timeout 900 python -m pytest tests -m gpu -q --timeout 600 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/summary.txt"
      tail -15 "$OUT/pytest.log" >> "$OUT/summary.txt" ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/summary.txt"
      tail -2 "$OUT/smoke.log" >> "$OUT/summary.txt" ;;
    bench13)
      timeout 600 python bench.py --workload wan1.3b_480px49f --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench13.json" 2> "$OUT/bench13.err"; echo "bench13 rc=$?" | tee -a "$OUT/summary.txt"
      cat "$OUT/bench13.json" >> "$OUT/summary.txt" ;;
    bench14)
      timeout 1200 python bench.py --steps ${B14_STEPS:-1} --warmup ${B14_WARMUP:-1} > "$OUT/bench14.json" 2> "$OUT/bench14.err"; echo "bench14 rc=$?" | tee -a "$OUT/summary.txt"
      cat "$OUT/bench14.json" >> "$OUT/summary.txt" ;;
    prof)
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --workload ${PROF_WL:-wan1.3b_480px49f} --steps 1 --warmup ${PROF_WARMUP:-1} --no-cpu-baseline > "$GRAFT_REPO_ROOT/$OUT/prof_bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof.err"); echo "prof rc=$?" | tee -a "$OUT/summary.txt"
      f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-220 >> "$OUT/summary.txt"
      # keep only the small summaries (the raw trace can be large)
      find "$OUT/prof" -name "*kernel_trace.csv" -size +20M -delete ;;
    pmc)
      # counters in their own passes (no tracing domains besides --kernel-trace); SQ has 8 slots, TCC 4
      i=0
      for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
                 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" \
                 "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
        i=$((i+1))
        IFS=';' read -ra TARGETS <<< "${PMC_TARGETS:-pattn 0 20280 12 2;pgemm 20280 8960 1536 2}"
        for target in "${TARGETS[@]}"; do
          tag=$(echo $target | tr ' ' '_')
          (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/pmc/${tag}_set$i" -o pmc -- "$GRAFT_REPO_ROOT/tools/x2v_check" $target > "$GRAFT_REPO_ROOT/$OUT/pmc_${tag}_set$i.log" 2>&1)
        done
      done
      python tools/pmc_summary.py "$OUT/pmc" > "$OUT/pmc_summary.txt" 2>&1; cat "$OUT/pmc_summary.txt" >> "$OUT/summary.txt" ;;
  esac
  echo "stage $st took $(( $(date +%s) - t0 ))s" | tee -a "$OUT/summary.txt"
done
cat "$OUT/summary.txt"
