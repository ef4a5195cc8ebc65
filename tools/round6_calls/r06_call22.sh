#!/bin/bash
# Round 6, call 22: (1) the pixel-wise producer (vae_prep, fp16 outputs) with a third of a pixel's chunks per lane (96 / 192 / 384 channels = 24 / 48 / 96 chunks: 8 / 16 / 32 lanes
# per pixel, every lane busy) against the power-of-two lane groups (X2V_VAE_PREP_POW2=1: a quarter of each wave idle): VAE GPU tests, then the 720p x 81f decode a/b/a/b with a
# kernel-stats pass of each.  (2) the new w8a8 operator test at the rank-of-8 shapes.  (3) end to end on this round's tree: T2V-14B 720p 50 steps + VAE, I2V-14B 720p 40 steps + VAE.
set +e
OUT=gpurun_out/r06_call22
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_vae.py tests/test_gpu_hunyuan_vae.py "tests/test_gpu_rank_shapes.py::test_w8a8_operator_on_the_exchange_buffers_wan14b_rank_of_8" -m gpu -q --timeout 500 > "$OUT/pytest.log" 2>&1; echo "pytest vae + w8a8 rank shapes rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a "$OUT/summary.txt"; tail -5 "$OUT/pytest.log" | cut -c1-300 >> "$OUT/summary.txt"
for rep in 1 2; do
  for p in 0 1; do
    echo "X2V_VAE_PREP_POW2=$p: $(X2V_VAE_PREP_POW2=$p timeout 300 python tools/vae_bench.py --latent 16,21,90,160 --split 2>&1 | tail -1 | cut -c1-260)" | tee -a "$OUT/summary.txt"
  done
done
for p in 0 1; do
  (cd /tmp && X2V_VAE_PREP_POW2=$p timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof_pow2_$p" -o vae -- python "$GRAFT_REPO_ROOT/tools/vae_bench.py" --latent 16,21,90,160 --split > "$GRAFT_REPO_ROOT/$OUT/prof_pow2_$p.log" 2>&1)
  find "$OUT/prof_pow2_$p" -name "*kernel_trace.csv" -delete
  echo "== kernel stats, X2V_VAE_PREP_POW2=$p" >> "$OUT/summary.txt"; grep -h "vae_prep\|conv16g" "$OUT"/prof_pow2_$p/*kernel_stats.csv | cut -c1-60,100-200 | head -6 >> "$OUT/summary.txt"
done
echo "vae done $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
timeout 900 python tools/e2e.py --steps 50 > "$OUT/e2e_wan14b_720p.json" 2> "$OUT/e2e14.err"; echo "e2e 14B rc=$?" >> "$OUT/summary.txt"; cat "$OUT/e2e_wan14b_720p.json" >> "$OUT/summary.txt"
timeout 900 python tools/e2e.py --i2v > "$OUT/e2e_wan14b_i2v_720p.json" 2> "$OUT/e2e_i2v.err"; echo "e2e i2v rc=$?" >> "$OUT/summary.txt"; cat "$OUT/e2e_wan14b_i2v_720p.json" >> "$OUT/summary.txt"
echo "total $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
