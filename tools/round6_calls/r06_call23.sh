#!/bin/bash
# Round 6, call 23: the 128-cout form of the 128-pixel convolution kernel (vae16g.hip NCB = 8: 8 x 8 accumulator tiles, all 256 AGPRs) for the HunyuanVideo VAE's 128 / 256 / 512-channel
# convolutions, which stayed on the 64-pixel halo kernel: parity (conv3d + the other kernels, the Hunyuan VAE fixtures), then the 720p x 129f tiled decode a/b/a/b against
# X2V_VAE_CONV16=halo64 and a kernel-stats pass of each.
set +e
OUT=gpurun_out/r06_call23
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_vae.py tests/test_gpu_hunyuan_vae.py -m gpu -q --timeout 600 > "$OUT/pytest.log" 2>&1; echo "pytest vae rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a "$OUT/summary.txt"; tail -8 "$OUT/pytest.log" | cut -c1-300 >> "$OUT/summary.txt"
for rep in 1 2; do
  for k in "" halo64; do
    echo "X2V_VAE_CONV16='$k' tile: $(X2V_VAE_CONV16=$k timeout 300 python tools/hunyuan_vae_bench.py 2>&1 | tail -1 | cut -c1-300)" | tee -a "$OUT/summary.txt"
    echo "X2V_VAE_CONV16='$k' full: $(X2V_VAE_CONV16=$k timeout 400 python tools/hunyuan_vae_bench.py --full 2>&1 | tail -1 | cut -c1-300)" | tee -a "$OUT/summary.txt"
  done
done
for k in g halo64; do
  kk=$k; [ $k = g ] && kk=""
  (cd /tmp && X2V_VAE_CONV16=$kk timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof_$k" -o vae -- python "$GRAFT_REPO_ROOT/tools/hunyuan_vae_bench.py" --full > "$GRAFT_REPO_ROOT/$OUT/prof_$k.log" 2>&1)
  find "$OUT/prof_$k" -name "*kernel_trace.csv" -delete
  echo "== kernel stats, X2V_VAE_CONV16=$kk" >> "$OUT/summary.txt"; head -9 "$OUT"/prof_$k/*kernel_stats.csv | cut -c1-70,120-220 >> "$OUT/summary.txt"
done
echo "total $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
