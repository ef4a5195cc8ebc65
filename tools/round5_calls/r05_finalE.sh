#!/bin/bash
# Round 5, part E: re-validation of the last host-side edits (Hunyuan Ulysses CommTimer brackets, bench.py's probe definition): the dist tests that drive them + a short bench line
set +e
OUT=gpurun_out/r05_finalE
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_hunyuan.py -m gpu -q --timeout 500 -x -k "hunyuan or bench_n8 or rccl" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s): $(tail -1 "$OUT/pytest.log" | cut -c1-120)" | tee -a "$OUT/summary.txt"
timeout 200 python bench.py --workload wan1.3b_480px49f --steps 4 --warmup 2 --no-cpu-baseline --probe-ms 500 > "$OUT/bench13.json" 2> "$OUT/bench13.err"; echo "bench13 rc=$?: $(python -c "import json; d=json.loads([l for l in open('$OUT/bench13.json') if l.startswith('{')][-1]); print('ms_per_step %.1f of_probe %.4f def %s' % (d['ms_per_step'], d['roofline']['frac_of_probe'], d['box_calibration']['mfma_probe_definition']))" 2>&1)" | tee -a "$OUT/summary.txt"
echo "total $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
