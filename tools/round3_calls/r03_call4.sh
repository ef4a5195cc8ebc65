#!/bin/bash
# round 3, GPU call 4: GEMM epilogue fix verified; the reference itself on the GPU through the plugin; CFG pair at 1.3B; PMC traffic of the bench command
set +e
OUT=gpurun_out/r03_call4
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
: > $OUT/summary.txt
for m in gemm attn; do timeout 300 tools/x2v_check $m > $OUT/check_$m.log 2>&1; echo "check $m rc=$? $(tail -1 $OUT/check_$m.log)" | tee -a $OUT/summary.txt; done
for a in "151296 5120 5120 3 0 0" "151296 5120 5120 3 0 2" "151296 5120 13824 3 0 2" "20280 1536 1536 5 0 0" "20280 1536 1536 5 0 2" "20280 1536 8960 5 0 2"; do
  set -- $a
  timeout 120 tools/x2v_check pgemm $1 $2 $3 $4 $5 $6 2>&1 | tail -1 | tee -a $OUT/summary.txt
done
t0=$(date +%s)
X2V_REFERENCE_ROOT=$GRAFT_REPO_ROOT/oracle/_ref/reference timeout 900 python -m pytest tests/test_plugin_reference.py -m gpu -q -s --timeout 600 > $OUT/pytest_reference.log 2>&1; echo "pytest reference-on-gpu rc=$?" | tee -a $OUT/summary.txt
grep "REFERENCE_ON_GPU_OK\|passed\|failed\|Error" $OUT/pytest_reference.log | cut -c1-300 | tail -8 >> $OUT/summary.txt
timeout 1500 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_full_size.py tests/test_gpu_dist.py tests/test_gpu_model.py -q --timeout 900 > $OUT/pytest_sel.log 2>&1; echo "pytest selected rc=$?" | tee -a $OUT/summary.txt
tail -12 $OUT/pytest_sel.log | cut -c1-300 >> $OUT/summary.txt
echo "tests took $(( $(date +%s) - t0 ))s" | tee -a $OUT/summary.txt
cp gpurun_out/parity_summary.jsonl $OUT/ 2>/dev/null
for f in "" "--cfg-pair" "" "--cfg-pair"; do
  timeout 300 python bench.py --workload wan1.3b_480px49f --steps 3 --warmup 1 --no-cpu-baseline $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench13 $f', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])" | tee -a $OUT/summary.txt
done
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/pmc_bench/set$i" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$GRAFT_REPO_ROOT/$OUT/pmc_bench_set$i.log" 2>&1); echo "pmc bench set$i rc=$?" | tee -a $OUT/summary.txt
done
python tools/pmc_traffic.py $OUT/pmc_bench "attn_fwd_v9_kernel<8, 8, true, true>" 75600 40 2 > $OUT/r03_pmc_attn_traffic.json 2>> $OUT/summary.txt; cat $OUT/r03_pmc_attn_traffic.json >> $OUT/summary.txt
find $OUT/pmc_bench -name "*.csv" -size +3M -delete
cat $OUT/summary.txt
