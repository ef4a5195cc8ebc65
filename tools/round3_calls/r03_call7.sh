#!/bin/bash
# round 3, GPU call 7: persistent tile loop of gemm256s (X2V_GEMM_PERSIST = workgroup waves over the 256 CUs; 0 = one workgroup per tile)
set +e
OUT=gpurun_out/r03_call7
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
: > $OUT/summary.txt
X2V_GEMM_PERSIST=1 timeout 300 tools/x2v_check gemm > $OUT/check_gemm_p1.log 2>&1; echo "check gemm persist=1 rc=$? $(tail -1 $OUT/check_gemm_p1.log)" | tee -a $OUT/summary.txt
X2V_GEMM_PERSIST=1 timeout 900 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_full_size.py -q --timeout 600 -k "gemm or block or v_projection" > $OUT/pytest_p1.log 2>&1; echo "pytest persist=1 rc=$?" | tee -a $OUT/summary.txt; tail -4 $OUT/pytest_p1.log | cut -c1-200 >> $OUT/summary.txt
timeout 300 python -m pytest tests/test_gpu_vae.py -q --timeout 300 -k split > $OUT/pytest_vae.log 2>&1; echo "pytest vae split rc=$?" | tee -a $OUT/summary.txt; tail -2 $OUT/pytest_vae.log | cut -c1-200 >> $OUT/summary.txt
for a in "151296 5120 5120 3 0 0" "151296 5120 5120 3 0 2" "151296 13824 5120 3 0 1" "151296 5120 13824 3 0 2" "75600 5120 5120 3 0 0" "20280 1536 1536 10 0 0" "20280 1536 1536 10 0 2" "20280 8960 1536 10 0 1" "20280 1536 8960 10 0 2"; do
  for s in 0 1 2 4; do
    set -- $a
    X2V_GEMM_PERSIST=$s timeout 120 tools/x2v_check pgemm $1 $2 $3 $4 $5 $6 2>&1 | tail -1 | sed "s/^/persist=$s /" | tee -a $OUT/summary.txt
  done
done
for s in 0 1 2; do
  X2V_GEMM_PERSIST=$s timeout 300 python bench.py --workload wan1.3b_480px49f --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench13 persist=$s', d['ms_per_step'], d['roofline']['avg_launch_ms'])" | tee -a $OUT/summary.txt
done
for s in 0 1; do
  X2V_GEMM_PERSIST=$s timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench14 persist=$s', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])" | tee -a $OUT/summary.txt
done
cat $OUT/summary.txt
