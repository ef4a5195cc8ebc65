#!/bin/bash
# round 3, GPU call 5: tile-phase stagger of the 256x256 GEMM kernels (epilogue HBM bursts) — standalone shapes, then step level (1.3B, 14B, fp8 distilled)
set +e
OUT=gpurun_out/r03_call5
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
: > $OUT/summary.txt
X2V_GEMM_STAGGER=2 timeout 300 tools/x2v_check gemm > $OUT/check_gemm_s2.log 2>&1; echo "check gemm stagger=2 rc=$? $(tail -1 $OUT/check_gemm_s2.log)" | tee -a $OUT/summary.txt
for a in "151296 5120 5120 3 0 0" "151296 5120 5120 3 0 2" "151296 13824 5120 3 0 1" "151296 5120 13824 3 0 2" "20280 1536 1536 10 0 0" "20280 1536 1536 10 0 2" "20280 8960 1536 10 0 1" "20280 1536 8960 10 0 2"; do
  for s in 0 1 2 4 8; do
    set -- $a
    X2V_GEMM_STAGGER=$s timeout 120 tools/x2v_check pgemm $1 $2 $3 $4 $5 $6 2>&1 | tail -1 | sed "s/^/stagger=$s /" | tee -a $OUT/summary.txt
  done
done
for s in 0 2 4; do
  X2V_GEMM_STAGGER=$s timeout 300 python bench.py --workload wan1.3b_480px49f --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench13 stagger=$s', d['ms_per_step'], d['roofline']['avg_launch_ms'])" | tee -a $OUT/summary.txt
done
for s in 0 2 4; do
  X2V_GEMM_STAGGER=$s timeout 600 python bench.py --fp8 --distill --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench fp8 distill stagger=$s', d['ms_per_step'], d['roofline']['avg_launch_ms'])" | tee -a $OUT/summary.txt
done
for s in 0 2; do
  X2V_GEMM_STAGGER=$s timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench14 stagger=$s', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])" | tee -a $OUT/summary.txt
done
cat $OUT/summary.txt
