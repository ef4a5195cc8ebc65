#!/bin/bash
# Round 5, final tree, part D: the driver's round-4 command (20 timed steps after 5 warm-up steps) and the kernel stats of the i2v step.
set +e
OUT=gpurun_out/r05_finalD
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof_i2v" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --i2v --steps 1 --warmup 1 --no-cpu-baseline --no-calibration > "$GRAFT_REPO_ROOT/$OUT/prof_i2v_bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof_i2v.err"); echo "prof i2v rc=$? at $(( $(date +%s) - t0 )) s" | tee -a "$OUT/summary.txt"
f=$(find "$OUT/prof_i2v" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_wan14b_i2v_720p.csv" && head -9 "$f" | cut -c1-70,200-320 >> "$OUT/summary.txt"
find "$OUT/prof_i2v" -name "*kernel_trace.csv" -delete
timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_driver_command.json" 2> "$OUT/bench_driver.err"; echo "bench --steps 20 --warmup 5 rc=$? at $(( $(date +%s) - t0 )) s: $(python -c "import json; d=json.loads([l for l in open('$OUT/bench_driver_command.json') if l.startswith('{')][-1]); print('ms_per_step %.1f attn %.2f ms frac %.4f of_probe %.4f probe %.0f / %.0f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['frac_of_probe'], d['box_calibration']['mfma_probe_tflops_before'], d['box_calibration']['mfma_probe_tflops_after']))" 2>&1)" | tee -a "$OUT/summary.txt"
echo "total $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
