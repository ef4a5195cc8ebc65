#!/bin/bash
# Round 4, call 4: the continuous GEMM as the dispatcher's default: GPU tests that touch a GEMM, then step-level A/B (X2V_GEMM_CONTINUOUS=0/1) at configs #2 and #3.
set +e
OUT=gpurun_out/r04_call4
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
timeout 1000 python -m pytest tests/test_gpu_ops.py tests/test_gpu_boundary.py tests/test_gpu_bench_shapes.py tests/test_gpu_model.py tests/test_gpu_hunyuan.py tests/test_gpu_full_size.py tests/test_gpu_rank_shapes.py tests/test_gpu_dist.py tests/test_gpu_edge.py \
  -m gpu -q --timeout 900 --durations=12 -k "not config1_full_run and not full_forward and not fp8_block and not attention" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a "$OUT/summary.txt"
tail -25 "$OUT/pytest.log" | cut -c1-300 >> "$OUT/summary.txt"
for c in 1 0 1 0; do
  X2V_GEMM_CONTINUOUS=$c timeout 300 python bench.py --workload wan1.3b_480px49f --steps 5 --warmup 2 --no-cpu-baseline --no-calibration > "$OUT/b13_c$c.json" 2> "$OUT/b13.err"
  echo "bench13 continuous=$c: $(python -c "import json,sys; d=json.loads(open('$OUT/b13_c$c.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['cfg_form'][:20])")" | tee -a "$OUT/summary.txt"
done
for c in 1 0; do
  X2V_GEMM_CONTINUOUS=$c timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/b14_c$c.json" 2> "$OUT/b14.err"
  echo "bench14 continuous=$c: $(python -c "import json,sys; d=json.loads(open('$OUT/b14_c$c.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['box_calibration']['mfma_probe_tflops'])")" | tee -a "$OUT/summary.txt"
done
cat "$OUT/summary.txt"
