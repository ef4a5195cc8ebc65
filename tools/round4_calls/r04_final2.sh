#!/bin/bash
# Round 4, final tree: the whole GPU suite, smoke, the default bench line, config #2 kernel stats with the continuous GEMM, the 50-step end-to-end run.
set +e
OUT=gpurun_out/r04_final2
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
run() { name=$1; shift; t0=$(date +%s); "$@"; echo "$name rc=$? ($(( $(date +%s) - t0 )) s)" >> "$OUT/summary.txt"; }
run pytest timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --durations=10 > "$OUT/pytest.log" 2>&1; tail -16 "$OUT/pytest.log" | cut -c1-200 >> "$OUT/summary.txt"; cp gpurun_out/parity_summary.jsonl "$OUT/" 2>/dev/null
run smoke timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; tail -2 "$OUT/smoke.log" >> "$OUT/summary.txt"
run bench_default timeout 900 python bench.py --no-cpu-config1 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; cat "$OUT/bench_default.json" >> "$OUT/summary.txt"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof13" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --workload wan1.3b_480px49f --steps 2 --warmup 1 --no-cpu-baseline --no-calibration > "$GRAFT_REPO_ROOT/$OUT/prof13_bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof13.err"); echo "prof13 rc=$?" >> "$OUT/summary.txt"
find "$OUT/prof13" -name "*kernel_trace.csv" -delete
run e2e_14b timeout 900 python tools/e2e.py --steps 50 > "$OUT/e2e_wan14b_720p.json" 2> "$OUT/e2e14.err"; cat "$OUT/e2e_wan14b_720p.json" >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
