#!/bin/bash
# Round 5, final tree, part C: the north star's end-to-end number on one GPU — Wan2.1-T2V-14B bf16, 720p x 81f, 50 steps (CFG) + VAE decode.
set +e
OUT=gpurun_out/r05_finalC
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
timeout 900 python tools/e2e.py --steps 50 > "$OUT/e2e_wan14b_720p.json" 2> "$OUT/e2e14.err"; echo "e2e 14B 720p rc=$? ($(( $(date +%s) - t0 )) s): $(tail -1 "$OUT/e2e_wan14b_720p.json" | cut -c1-500)" | tee -a "$OUT/summary.txt"
