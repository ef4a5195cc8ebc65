#!/bin/bash
# Round 6, call 26: kernel stats of the i2v step (both cross-attentions on the persistent launch form) and of the w8a8 distilled step on the final tree, with their bench lines.
set +e
OUT=gpurun_out/r06_call26
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof_i2v" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --i2v --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > "$GRAFT_REPO_ROOT/$OUT/bench_i2v.json" 2> "$GRAFT_REPO_ROOT/$OUT/i2v.err"); echo "i2v rc=$?" >> "$OUT/summary.txt"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof_fp8" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --fp8 --distill --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > "$GRAFT_REPO_ROOT/$OUT/bench_fp8_distill.json" 2> "$GRAFT_REPO_ROOT/$OUT/fp8.err"); echo "fp8 distill rc=$?" >> "$OUT/summary.txt"
find "$OUT" -name "*kernel_trace.csv" -delete
for k in i2v fp8; do echo "== $k" >> "$OUT/summary.txt"; head -12 "$OUT"/prof_$k/*kernel_stats.csv | cut -c1-70,200-330 >> "$OUT/summary.txt"; done
tail -1 "$OUT/bench_i2v.json" | cut -c1-400 >> "$OUT/summary.txt"; tail -1 "$OUT/bench_fp8_distill.json" | cut -c1-400 >> "$OUT/summary.txt"
echo "total $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
