#!/bin/bash
# Round 4, call 7: kernel stats of configs #4 and #5 on the final tree; PMC traffic passes of the bench command (refresh of profiles/*_pmc_attn_traffic.json).
set +e
OUT=gpurun_out/r04_call7
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof_fp8" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --fp8 --distill --steps 2 --warmup 1 --no-cpu-baseline --no-calibration > "$GRAFT_REPO_ROOT/$OUT/prof_fp8_bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof_fp8.err"); echo "prof_fp8 rc=$?" | tee -a "$OUT/summary.txt"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof_hy" -o bench -- python "$GRAFT_REPO_ROOT/tools/hunyuan_bench.py" --steps 1 --warmup 1 > "$GRAFT_REPO_ROOT/$OUT/prof_hy_bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof_hy.err"); echo "prof_hy rc=$?" | tee -a "$OUT/summary.txt"
for c in FETCH_SIZE WRITE_SIZE "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/pmc/$tag" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-calibration > "$GRAFT_REPO_ROOT/$OUT/pmc_$tag.log" 2>&1); echo "pmc $tag rc=$?" | tee -a "$OUT/summary.txt"
done
python tools/pmc_traffic.py "$OUT/pmc" "attn_fwd_v9_kernel<8, 8, true, true>" 75600 40 2.0 > "$OUT/pmc_attn_traffic.json" 2>> "$OUT/summary.txt"; cat "$OUT/pmc_attn_traffic.json" >> "$OUT/summary.txt"
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT/pmc" -name "*counter_collection.csv" -size +30M -delete
for p in prof_fp8 prof_hy; do f=$(find "$OUT/$p" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-70,180-300 >> "$OUT/summary.txt"; done
cat "$OUT/summary.txt"
