#!/bin/bash
# round 3, GPU call 2: v9 (16x16x32) attention correctness + timing, work-mapping x key-walk-rotation matrix, PMC of the candidates,
# full-size parity tests (redesigned data), anchored-tolerance model tests
set +e
OUT=gpurun_out/r03_call2
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
: > $OUT/summary.txt
for cfg in "8 0 0" "9 0 0" "9 1 1" "8 1 2" "9 1 4" "8 2 5"; do
  set -- $cfg
  X2V_ATTN_GEN=$1 X2V_ATTN_MAP=$2 X2V_ATTN_ROT=$3 timeout 300 tools/x2v_check attn > $OUT/check_attn_g$1_m$2_r$3.log 2>&1; echo "check attn gen=$1 map=$2 rot=$3 rc=$? $(tail -1 $OUT/check_attn_g$1_m$2_r$3.log)" | tee -a $OUT/summary.txt
done
X2V_ATTN_GEN=9 timeout 120 tools/x2v_check dattn > $OUT/dattn_g9.log 2>&1; echo "dattn gen=9 rc=$?" | tee -a $OUT/summary.txt; grep -c FAIL $OUT/dattn_g9.log >> $OUT/summary.txt; head -40 $OUT/dattn_g9.log >> $OUT/summary.txt
X2V_ATTN_GEN=8 timeout 120 tools/x2v_check dattn > $OUT/dattn_g8.log 2>&1; echo "dattn gen=8 rc=$? fails $(grep -c FAIL $OUT/dattn_g8.log)" | tee -a $OUT/summary.txt
grep FAIL $OUT/check_attn_g9_m0_r0.log | head -20 >> $OUT/summary.txt
for gen in 8 9; do
  for mr in "0 0" "1 0" "1 1" "1 5" "1 2" "1 3" "1 4" "2 0" "2 1" "0 1" "0 2"; do
    set -- $mr
    X2V_ATTN_GEN=$gen X2V_ATTN_MAP=$1 X2V_ATTN_ROT=$2 timeout 120 tools/x2v_check pattn 12 75600 40 3 2>&1 | tail -1 | sed "s/^/gen=$gen map=$1 rot=$2 /" | tee -a $OUT/summary.txt
  done
done
for gen in 8 9; do
  for mr in "0 0" "1 1" "1 2"; do
    set -- $mr
    X2V_ATTN_GEN=$gen X2V_ATTN_MAP=$1 X2V_ATTN_ROT=$2 timeout 120 tools/x2v_check pattn 12 20280 12 10 2>&1 | tail -1 | sed "s/^/gen=$gen map=$1 rot=$2 /" | tee -a $OUT/summary.txt
    X2V_ATTN_GEN=$gen X2V_ATTN_MAP=$1 X2V_ATTN_ROT=$2 timeout 120 tools/x2v_check pattn 12 75600 5 5 2>&1 | tail -1 | sed "s/^/gen=$gen map=$1 rot=$2 /" | tee -a $OUT/summary.txt
  done
done
i=0
for set in "FETCH_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  i=$((i+1))
  for cfg in "8 1 1" "8 1 2" "9 0 0" "9 1 1"; do
    set -- $cfg
    tag=g$1_m$2_r$3
    (cd /tmp && X2V_ATTN_GEN=$1 X2V_ATTN_MAP=$2 X2V_ATTN_ROT=$3 timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/pmc/$tag/set$i" -o pmc -- "$GRAFT_REPO_ROOT/tools/x2v_check" pattn 12 75600 40 1 > "$GRAFT_REPO_ROOT/$OUT/pmc_${tag}_set$i.log" 2>&1)
  done
done
for tag in g8_m1_r1 g8_m1_r2 g9_m0_r0 g9_m1_r1; do echo "--- PMC $tag" >> $OUT/pmc_summary.txt; python tools/pmc_summary.py $OUT/pmc/$tag >> $OUT/pmc_summary.txt 2>&1; done
cat $OUT/pmc_summary.txt >> $OUT/summary.txt
find $OUT/pmc -name "*kernel_trace.csv" -size +5M -delete
t0=$(date +%s)
timeout 1500 python -m pytest tests/test_gpu_full_size.py -q --timeout 900 --durations=20 > $OUT/pytest_full_size.log 2>&1; echo "pytest full_size rc=$?" | tee -a $OUT/summary.txt
tail -40 $OUT/pytest_full_size.log | cut -c1-400 >> $OUT/summary.txt
echo "full-size took $(( $(date +%s) - t0 ))s" | tee -a $OUT/summary.txt
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_model.py -q --timeout 800 -k "anchored or config1" > $OUT/pytest_truth.log 2>&1; echo "pytest truth rc=$?" | tee -a $OUT/summary.txt
tail -12 $OUT/pytest_truth.log | cut -c1-400 >> $OUT/summary.txt
echo "truth tests took $(( $(date +%s) - t0 ))s" | tee -a $OUT/summary.txt
cp gpurun_out/parity_summary.jsonl $OUT/ 2>/dev/null
cat $OUT/summary.txt
