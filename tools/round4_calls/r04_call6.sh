#!/bin/bash
# Round 4, call 6: the i2v GPU test + the fixed rank-shape attention test; continuous-GEMM A/B: scheduling-group size, non-temporal output stores, non-temporal x loads.
set +e
OUT=gpurun_out/r04_call6
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_rank_shapes.py tests/test_gpu_vae.py -m gpu -q --timeout 600 -k "i2v or rank_of_8_both or split_fp16 or cfg_pair" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a "$OUT/summary.txt"
tail -25 "$OUT/pytest.log" | cut -c1-300 >> "$OUT/summary.txt"
run() { tag=$1; shift; echo "-- $tag" >> "$OUT/summary.txt"; for spec in "151200 5120 5120 6 V 0" "151200 13824 5120 6 V 1" "151200 5120 13824 4 V 2" "20280 1536 1536 20 V 0" "20280 8960 1536 20 V 1" "20280 1536 8960 20 V 2"; do
    for v in "$@"; do timeout 100 tools/x2v_check pgemm ${spec/V/$v} 2>&1 | tail -1 >> "$OUT/summary.txt"; done; done; }
run "default build: one-tile (4) vs continuous (5), groups of 4 / 2 / 8 m-tiles" 4 5 517 2053
LD_LIBRARY_PATH=$PWD/tools/probes/ab/ntstore run "non-temporal output stores (continuous)" 5
LD_LIBRARY_PATH=$PWD/tools/probes/ab/ntx run "non-temporal x-operand DMA (continuous)" 5
cat "$OUT/summary.txt"
