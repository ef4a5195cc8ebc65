#!/bin/bash
# Round 5, final tree, part B: end-to-end runs (denoise loop + VAE decode): i2v 480p and 720p (the reference's published configuration: 40 steps, CFG), w8a8 distilled, Wan-1.3B 480p.
set +e
OUT=gpurun_out/r05_finalB
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
say() { echo "$@" | tee -a "$OUT/summary.txt"; }
timeout 400 python tools/e2e.py --fp8 --distill > "$OUT/e2e_wan14b_fp8_distill.json" 2> "$OUT/e2e_fp8.err"; say "e2e fp8 distill rc=$?: $(tail -1 "$OUT/e2e_wan14b_fp8_distill.json" | cut -c1-420)"
timeout 300 python tools/e2e.py --workload wan1.3b_480px49f --steps 50 > "$OUT/e2e_wan13b_480p.json" 2> "$OUT/e2e13.err"; say "e2e 1.3B rc=$?: $(tail -1 "$OUT/e2e_wan13b_480p.json" | cut -c1-420)"
timeout 500 python tools/e2e.py --workload wan14b_i2v_480px81f > "$OUT/e2e_wan14b_i2v_480p.json" 2> "$OUT/e2e_i2v480.err"; say "e2e i2v 480p rc=$? at $(( $(date +%s) - t0 )) s: $(tail -1 "$OUT/e2e_wan14b_i2v_480p.json" | cut -c1-420)"
timeout 900 python tools/e2e.py --i2v > "$OUT/e2e_wan14b_i2v_720p.json" 2> "$OUT/e2e_i2v720.err"; say "e2e i2v 720p rc=$? at $(( $(date +%s) - t0 )) s: $(tail -1 "$OUT/e2e_wan14b_i2v_720p.json" | cut -c1-420)"
say "total $(( $(date +%s) - t0 )) s"
cat "$OUT/summary.txt"
