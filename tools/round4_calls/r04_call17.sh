#!/bin/bash
# Round 4, call 17 (the last GPU seconds): the w8a8 GEMM tests with the continuous kernel as the dispatcher's default for unblocked operands.
set +e
OUT=gpurun_out/r04_call17
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
timeout 42 python -m pytest tests/test_gpu_bench_shapes.py -m gpu -q -x -k "fp8_natural" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" > "$OUT/summary.txt"; tail -3 "$OUT/pytest.log" | cut -c1-300 >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
