#!/bin/bash
# Round 5, part F: torch-free self-test of the final library (tools/x2v_check, every group) and the kernel stats of the w8a8 distilled step on the final tree (unscaled fp8 MFMA encoding)
set +e
OUT=gpurun_out/r05_finalF
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
for g in probe misc norm rope gemm attn fp8 conv; do timeout 120 tools/x2v_check $g > "$OUT/x2v_check_$g.log" 2>&1; echo "x2v_check $g rc=$? $(grep -c PASS "$OUT/x2v_check_$g.log") PASS $(grep -c FAIL "$OUT/x2v_check_$g.log") FAIL" | tee -a "$OUT/summary.txt"; done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof_fp8" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --fp8 --distill --steps 2 --warmup 1 --no-cpu-baseline --no-calibration > "$GRAFT_REPO_ROOT/$OUT/prof_fp8_bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof_fp8.err"); echo "prof fp8 rc=$? at $(( $(date +%s) - t0 )) s" | tee -a "$OUT/summary.txt"
f=$(find "$OUT/prof_fp8" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_wan14b_fp8_distill.csv" && head -8 "$f" | cut -c1-70,200-330 >> "$OUT/summary.txt"
find "$OUT/prof_fp8" -name "*kernel_trace.csv" -delete
echo "total $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
