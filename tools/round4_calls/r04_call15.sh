#!/bin/bash
# Round 4, call 15 (last GPU minutes): rocprofv3 --kernel-trace --stats of the default bench command on the final tree (key walk from tile 0).
set +e
OUT=gpurun_out/r04_call15
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
(cd /tmp && timeout 130 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --probe-ms 500 > "$GRAFT_REPO_ROOT/$OUT/prof_bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof_bench.err")
echo "rc=$?" > "$OUT/summary.txt"
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && head -6 "$f" | cut -c1-200 >> "$OUT/summary.txt"
tail -1 "$OUT/prof_bench.json" | cut -c1-600 >> "$OUT/summary.txt"
find "$OUT/prof" -type f ! -name "*kernel_stats.csv" -delete
cat "$OUT/summary.txt"
