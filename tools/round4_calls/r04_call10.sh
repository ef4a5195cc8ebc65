#!/bin/bash
# Round 4, call 10: the dist GPU workers with the tight per-branch oracle leg.
set +e
OUT=gpurun_out/r04_call10
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
timeout 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_hunyuan.py -m gpu -q --timeout 600 -k "ulysses or teacache" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/summary.txt"; tail -15 "$OUT/pytest.log" | cut -c1-300 >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
