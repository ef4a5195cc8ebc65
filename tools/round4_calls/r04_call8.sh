#!/bin/bash
# Round 4, call 8: sanity of the final bench.py (traffic file lookup changed after the final-tree run) + the N-rank plumbing runs + smoke.
set +e
OUT=gpurun_out/r04_call8
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-config1 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?" | tee -a "$OUT/summary.txt"
python -c "import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['traffic'], d['roofline']['frac_of_probe'], d['cpu_baseline']['value'])" >> "$OUT/summary.txt" 2>&1
timeout 600 python -m pytest tests/test_gpu_dist.py -m gpu -q --timeout 600 -k "bench_n8 or e2e_n8" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/pytest.log" >> "$OUT/summary.txt"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" >> "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
