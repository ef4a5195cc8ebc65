set +e
OUT=gpurun_out/r03_final2
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
: > $OUT/summary.txt
t0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -q --timeout 1300 --durations=15 > $OUT/pytest.log 2>&1; echo "pytest -m gpu rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a $OUT/summary.txt
tail -24 $OUT/pytest.log | cut -c1-200 >> $OUT/summary.txt
cp gpurun_out/parity_summary.jsonl $OUT/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt; tail -1 $OUT/smoke.log >> $OUT/summary.txt
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
python -c "import json; d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['cpu_baseline']['value'])" | tee -a $OUT/summary.txt
cat $OUT/summary.txt
