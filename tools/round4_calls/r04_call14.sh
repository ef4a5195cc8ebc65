#!/bin/bash
# Round 4, call 14 (last GPU minutes): HunyuanVideo driver without the staggered key walk — its GPU tests.
set +e
OUT=gpurun_out/r04_call14
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
timeout 150 python -m pytest tests/test_gpu_hunyuan.py tests/test_gpu_full_size.py -m gpu -q --timeout 150 -x -k "(test_gpu_hunyuan or hunyuan13b_block or attention_hunyuan) and not full_forward" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/summary.txt"; tail -4 "$OUT/pytest.log" | cut -c1-200 >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
