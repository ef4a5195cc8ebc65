# Validation of the tree with the one-GPU two-stream CFG form as default where the pair pass is off: whole GPU suite, smoke, default bench,
# config #2 bench (+ its rocprof kernel stats) and end-to-end run.
set +e
OUT=gpurun_out/r03_final3
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
: > $OUT/summary.txt
t0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -q --timeout 1300 --durations=15 > $OUT/pytest.log 2>&1; echo "pytest -m gpu rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a $OUT/summary.txt
tail -24 $OUT/pytest.log | cut -c1-200 >> $OUT/summary.txt
cp gpurun_out/parity_summary.jsonl $OUT/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt; tail -1 $OUT/smoke.log >> $OUT/summary.txt
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
python -c "import json; d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['cpu_baseline']['value'], d['config']['cfg_form'])" | tee -a $OUT/summary.txt
for v in "" "--no-cfg-streams"; do
  timeout 300 python bench.py --workload wan1.3b_480px49f --steps 10 --warmup 2 --no-cpu-baseline $v > "$OUT/bench13${v}.json" 2> "$OUT/bench13${v}.err"; echo "bench13 $v rc=$?" | tee -a $OUT/summary.txt
  python -c "import json; d=json.loads(open('$OUT/bench13${v}.json').read().strip().splitlines()[-1]); print('bench13', d['ms_per_step'], d['config']['step_frac_of_bf16_peak'], d['config']['cfg_form'][:30])" | tee -a $OUT/summary.txt
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof13" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --workload wan1.3b_480px49f --steps 3 --warmup 1 --no-cpu-baseline > "$GRAFT_REPO_ROOT/$OUT/prof13_bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof13.err"); echo "prof13 rc=$?" | tee -a $OUT/summary.txt
find "$OUT/prof13" -name "*kernel_trace.csv" -delete
timeout 300 python tools/e2e.py --workload wan1.3b_480px49f --steps 50 > $OUT/e2e_wan13b_480p.json 2> $OUT/e2e13.err; echo "e2e13 rc=$?" | tee -a $OUT/summary.txt; cat $OUT/e2e_wan13b_480p.json >> $OUT/summary.txt
cat $OUT/summary.txt
