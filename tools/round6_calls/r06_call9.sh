#!/bin/bash
# Round 6, call 9: first contact of the 128-pixel x 96-cout VAE convolution kernel (csrc/vae16g.hip): the VAE GPU tests (incl. the new parity test against conv3d
# and the two older kernels), then the 720p x 81f decode a-b-a-b against the 64-pixel halo kernel (X2V_VAE_CONV16=halo64), then a kernel trace of one decode.
set +e
OUT=gpurun_out/r06_call9
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_vae.py -m gpu -q --timeout 600 -x > "$OUT/pytest_vae.log" 2>&1; echo "pytest vae rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a "$OUT/summary.txt"; tail -12 "$OUT/pytest_vae.log" | cut -c1-400 >> "$OUT/summary.txt"
for rep in 1 2; do
  echo "128x96: $(timeout 300 python tools/vae_bench.py --split 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
  echo "halo64: $(X2V_VAE_CONV16=halo64 timeout 300 python tools/vae_bench.py --split 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof" -o vae -- python "$GRAFT_REPO_ROOT/tools/vae_bench.py" --split > "$GRAFT_REPO_ROOT/$OUT/prof.log" 2>&1)
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-200 >> "$OUT/summary.txt"
find "$OUT/prof" -name "*kernel_trace.csv" -size +5M -delete
echo "total $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
