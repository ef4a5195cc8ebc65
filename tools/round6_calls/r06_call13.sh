#!/bin/bash
# Round 6, call 13: separate-cache VAE convolution with the corrected test predicate: VAE / dist / ABI GPU tests + decode timing.
set +e
OUT=gpurun_out/r06_call13
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_vae.py tests/test_gpu_dist.py tests/test_abi.py -m gpu -q --timeout 600 > "$OUT/pytest_vae.log" 2>&1; echo "pytest vae rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a "$OUT/summary.txt"; tail -12 "$OUT/pytest_vae.log" | cut -c1-400 >> "$OUT/summary.txt"
for rep in 1 2; do
  echo "chunk 4: $(timeout 300 python tools/vae_bench.py --split --chunk-frames 4 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
done
echo "total $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
