#!/bin/bash
# round 3, GPU call 1: new full-size parity tests + XCD-aware attention mapping A/B (timing and L2/fabric traffic) + a short bench
set +e
OUT=gpurun_out/r03_call1
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
rocm-smi --showproductname 2>/dev/null | head -6 > $OUT/summary.txt
t0=$(date +%s)
timeout 1500 python -m pytest tests/test_gpu_full_size.py -q -x --timeout 900 --durations=20 > $OUT/pytest_full_size.log 2>&1; echo "pytest full_size rc=$?" | tee -a $OUT/summary.txt
tail -30 $OUT/pytest_full_size.log >> $OUT/summary.txt
echo "full-size took $(( $(date +%s) - t0 ))s" | tee -a $OUT/summary.txt
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_dist.py tests/test_gpu_ops.py -q --timeout 600 > $OUT/pytest_other.log 2>&1; echo "pytest other rc=$?" | tee -a $OUT/summary.txt
tail -8 $OUT/pytest_other.log >> $OUT/summary.txt
echo "other tests took $(( $(date +%s) - t0 ))s" | tee -a $OUT/summary.txt
for x in 0 1 0 1; do
  X2V_ATTN_XCD=$x timeout 120 tools/x2v_check pattn 12 75600 40 3 2>&1 | tail -1 | sed "s/^/XCD=$x /" | tee -a $OUT/summary.txt
done
for x in 0 1; do
  X2V_ATTN_XCD=$x timeout 120 tools/x2v_check pattn 12 20280 12 10 2>&1 | tail -1 | sed "s/^/XCD=$x /" | tee -a $OUT/summary.txt
  X2V_ATTN_XCD=$x timeout 120 tools/x2v_check pattn 12 75600 5 5 2>&1 | tail -1 | sed "s/^/XCD=$x /" | tee -a $OUT/summary.txt
done
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  for x in 0 1; do
    (cd /tmp && X2V_ATTN_XCD=$x timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/pmc/xcd${x}/set$i" -o pmc -- "$GRAFT_REPO_ROOT/tools/x2v_check" pattn 12 75600 40 1 > "$GRAFT_REPO_ROOT/$OUT/pmc_xcd${x}_set$i.log" 2>&1)
  done
done
for x in 0 1; do echo "--- PMC X2V_ATTN_XCD=$x" >> $OUT/pmc_summary.txt; python tools/pmc_summary.py $OUT/pmc/xcd$x >> $OUT/pmc_summary.txt 2>&1; done; cat $OUT/pmc_summary.txt >> $OUT/summary.txt
find $OUT/pmc -name "*kernel_trace.csv" -size +5M -delete
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench14.json 2> $OUT/bench14.err; echo "bench14 rc=$?" | tee -a $OUT/summary.txt
cat $OUT/bench14.json >> $OUT/summary.txt
cat $OUT/summary.txt
