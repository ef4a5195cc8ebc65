#!/bin/bash
# Round 5, final tree, part A: the whole GPU suite, smoke, the default bench line (with the reference CPU baseline), its rocprofv3 kernel stats, the other configs' step lines.
set +e
OUT=gpurun_out/r05_finalA
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
say() { echo "$@" | tee -a "$OUT/summary.txt"; }
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --durations=10 > "$OUT/pytest.log" 2>&1; say "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -16 "$OUT/pytest.log" | cut -c1-200 >> "$OUT/summary.txt"; cp gpurun_out/parity_summary.jsonl "$OUT/" 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; say "smoke rc=$?: $(tail -1 "$OUT/smoke.log")"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; say "bench_default rc=$? at $(( $(date +%s) - t0 )) s"; tail -1 "$OUT/bench_default.json" >> "$OUT/summary.txt"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof14" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$GRAFT_REPO_ROOT/$OUT/prof14_bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof14.err"); say "prof14 rc=$?"
f=$(find "$OUT/prof14" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_wan14b_720p.csv" && head -10 "$f" | cut -c1-80,200-320 >> "$OUT/summary.txt"
find "$OUT/prof14" -name "*kernel_trace.csv" -delete
timeout 300 python bench.py --workload wan1.3b_480px49f --steps 8 --warmup 2 --no-cpu-baseline > "$OUT/bench13.json" 2> "$OUT/bench13.err"; say "bench13 rc=$?: $(python -c "import json; d=json.loads([l for l in open('$OUT/bench13.json') if l.startswith('{')][-1]); print('ms_per_step %.1f' % d['ms_per_step'])" 2>&1)"
timeout 400 python bench.py --fp8 --distill --steps 4 --warmup 1 --no-cpu-baseline > "$OUT/bench14_fp8_distill.json" 2> "$OUT/bench14_fp8.err"; say "bench fp8 distill rc=$?: $(python -c "import json; d=json.loads([l for l in open('$OUT/bench14_fp8_distill.json') if l.startswith('{')][-1]); print('ms_per_step %.1f' % d['ms_per_step'])" 2>&1)"
timeout 400 python tools/hunyuan_bench.py --steps 1 --warmup 1 > "$OUT/hunyuan13b.json" 2> "$OUT/hunyuan.err"; say "hunyuan rc=$?: $(tail -1 "$OUT/hunyuan13b.json" | cut -c1-300)"
timeout 200 python tools/vae_bench.py --split --reps 2 > "$OUT/vae_wan_720p81f_split.json" 2> "$OUT/vae.err"; say "vae: $(tail -1 "$OUT/vae_wan_720p81f_split.json" | cut -c1-260)"
say "total $(( $(date +%s) - t0 )) s"
cat "$OUT/summary.txt"
