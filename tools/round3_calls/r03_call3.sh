#!/bin/bash
# round 3, GPU call 3: v9 attention as the default — whole -m gpu suite; head-blocked-K experiment; residual-epilogue GEMM timing;
# rocprof kernel stats of config #2 (Wan-1.3B 480p) and a short 14B bench
set +e
OUT=gpurun_out/r03_call3
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
: > $OUT/summary.txt
for mr in "0 0" "0 1" "1 0" "1 1"; do
  set -- $mr
  X2V_ATTN_MAP=$1 X2V_ATTN_ROT=$2 timeout 120 tools/x2v_check pattn 12 75600 40 3 2>&1 | tail -1 | sed "s/^/map=$1 rot=$2 /" | tee -a $OUT/summary.txt
  X2V_ATTN_MAP=$1 X2V_ATTN_ROT=$2 timeout 120 tools/x2v_check pattnb 75600 40 3 2>&1 | tail -1 | sed "s/^/map=$1 rot=$2 /" | tee -a $OUT/summary.txt
done
timeout 60 tools/x2v_check pattn 12 75600 5 5 2>&1 | tail -1 | sed "s/^/auto /" | tee -a $OUT/summary.txt
timeout 60 tools/x2v_check pattn 12 20280 12 10 2>&1 | tail -1 | sed "s/^/auto /" | tee -a $OUT/summary.txt
for a in "151296 5120 5120 3 0 0" "151296 5120 5120 3 0 2" "151296 5120 13824 3 0 2" "151296 13824 5120 3 0 1" "75600 5120 5120 3 0 2" "20280 1536 1536 5 0 2" "20280 1536 8960 5 0 2"; do
  set -- $a
  timeout 120 tools/x2v_check pgemm $1 $2 $3 $4 $5 $6 2>&1 | tail -1 | tee -a $OUT/summary.txt
done
t0=$(date +%s)
timeout 1700 python -m pytest tests -m gpu -q --timeout 900 --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?" | tee -a $OUT/summary.txt
tail -30 $OUT/pytest_gpu.log | cut -c1-300 >> $OUT/summary.txt
echo "gpu suite took $(( $(date +%s) - t0 ))s" | tee -a $OUT/summary.txt
cp gpurun_out/parity_summary.jsonl $OUT/ 2>/dev/null
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof13" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --workload wan1.3b_480px49f --steps 2 --warmup 1 --no-cpu-baseline > "$GRAFT_REPO_ROOT/$OUT/prof13_bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof13.err"); echo "prof13 rc=$?" | tee -a $OUT/summary.txt
f=$(find $OUT/prof13 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f" | cut -c1-200 >> $OUT/summary.txt
find $OUT/prof13 -name "*kernel_trace.csv" -size +20M -delete
timeout 300 python bench.py --workload wan1.3b_480px49f --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench13.json 2> $OUT/bench13.err; echo "bench13 rc=$?" | tee -a $OUT/summary.txt; cut -c1-600 $OUT/bench13.json >> $OUT/summary.txt
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench14.json 2> $OUT/bench14.err; echo "bench14 rc=$?" | tee -a $OUT/summary.txt
cut -c1-1700 $OUT/bench14.json >> $OUT/summary.txt
cat $OUT/summary.txt
