#!/bin/bash
# Round 5, call 4:
#   1. VAE: the halo-tiled convolution with bare barriers (the loader's counted wait no longer defeated by __syncthreads' VMEM drain): tests + A-B-A-B vs the fenced build
#   2. MFMA-busy counters for the kernels of this round's step (gemm256c, gemm256c8, attn v9 <8,8,true,false>): one --pmc pass of tools/pmc_kernel_loop.py
#   3. roofline.traffic of the TIMED attention instantiation: FETCH_SIZE / WRITE_SIZE / TCC passes of the bench command itself -> profiles/r05_pmc_attn_traffic.json
#   4. what hipBLASLt launches for the seven shapes (kernel names from a kernel trace of tools/gemm_vs_hipblaslt.py) + the comparison itself
set +e
OUT=gpurun_out/r05_call4
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
say() { echo "$@" | tee -a "$OUT/summary.txt"; }
timeout 600 python -m pytest tests/test_gpu_vae.py tests/test_gpu_hunyuan_vae.py -m gpu -q --timeout 500 -x > "$OUT/pytest_vae.log" 2>&1; say "pytest vae rc=$? ($(( $(date +%s) - t0 )) s): $(tail -1 "$OUT/pytest_vae.log" | cut -c1-120)"
for rep in 1 2; do
  for v in default vhfenced; do
    if [ "$v" = default ]; then unset X2V_LIB_PATH; else export X2V_LIB_PATH=tools/probes/ab/$v/libx2v_hip.so; fi
    timeout 200 python tools/vae_bench.py --split --reps 2 > "$OUT/vae_${v}_$rep.json" 2> "$OUT/vae_${v}_$rep.err"
    say "vae rep$rep $v: $(python -c "import json; d=json.loads([l for l in open('$OUT/vae_${v}_$rep.json') if l.startswith('{')][-1]); print('%.3f s  %.1f TFLOP/s' % (d['seconds'], d['tflops_per_s']))" 2>&1)"
  done
done
unset X2V_LIB_PATH
timeout 200 python tools/hunyuan_vae_bench.py > "$OUT/hunyuan_vae.json" 2> "$OUT/hunyuan_vae.err"; say "hunyuan vae: $(cut -c1-300 "$OUT/hunyuan_vae.json" | tail -1)"
say "--- vae done at $(( $(date +%s) - t0 )) s"
# 2. MFMA-busy counters
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/pmc_mfma" -o pmc -- python "$GRAFT_REPO_ROOT/tools/pmc_kernel_loop.py" 2 > "$GRAFT_REPO_ROOT/$OUT/pmc_mfma_run.json" 2> "$GRAFT_REPO_ROOT/$OUT/pmc_mfma.err"); say "pmc mfma rc=$?"
timeout 120 python tools/pmc_kernel_loop.py 4 > "$OUT/kernel_loop_unprofiled.json" 2>> "$OUT/pmc_mfma.err"; say "unprofiled: $(cat "$OUT/kernel_loop_unprofiled.json" | tail -1 | cut -c1-600)"
python tools/pmc_summary.py "$OUT/pmc_mfma" > "$OUT/pmc_mfma_summary.txt" 2>&1; grep -A5 "gemm256c\|attn_fwd_v9" "$OUT/pmc_mfma_summary.txt" | head -40 >> "$OUT/summary.txt"
find "$OUT/pmc_mfma" -name "*kernel_trace.csv" -delete
say "--- pmc mfma done at $(( $(date +%s) - t0 )) s"
# 3. traffic of the timed attention instantiation, from the bench command itself
for c in FETCH_SIZE WRITE_SIZE "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/pmc/$tag" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-calibration > "$GRAFT_REPO_ROOT/$OUT/pmc_$tag.log" 2>&1); say "pmc $tag rc=$?"
done
python tools/pmc_traffic.py "$OUT/pmc" "attn_fwd_v9_kernel<8, 8, true, false>" 75600 40 2.0 > "$OUT/pmc_attn_traffic.json" 2>> "$OUT/summary.txt"; cat "$OUT/pmc_attn_traffic.json" >> "$OUT/summary.txt"
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT/pmc" -name "*counter_collection.csv" -size +30M -delete
say "--- traffic done at $(( $(date +%s) - t0 )) s"
# 4. hipBLASLt: the comparison, then its kernel names
timeout 300 python tools/gemm_vs_hipblaslt.py > "$OUT/gemm_vs_hipblaslt.json" 2> "$OUT/gemm_vs_hipblaslt.err"; say "gemm_vs_hipblaslt rc=$?"
python - >> "$OUT/summary.txt" 2>&1 <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05_call4/gemm_vs_hipblaslt.json") if l.startswith("{")][-1])
for r in d["gemm"]:
    print(r["M"], r["K"], r["N"], "ours %.0f / %.0f  hipBLASLt %.0f / %.0f" % (r["x2v_TFLOPs_0"], r["x2v_TFLOPs_1"], r["hipblaslt_TFLOPs_0"], r["hipblaslt_TFLOPs_1"]))
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof_blaslt" -o g -- python "$GRAFT_REPO_ROOT/tools/gemm_vs_hipblaslt.py" > /dev/null 2> "$GRAFT_REPO_ROOT/$OUT/prof_blaslt.err"); f=$(find "$OUT/prof_blaslt" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats_gemm_vs_hipblaslt.csv" && grep -i "Cijk\|gemm256" "$f" | cut -c1-420 | head -16 >> "$OUT/summary.txt"
find "$OUT/prof_blaslt" -name "*kernel_trace.csv" -delete
say "total $(( $(date +%s) - t0 )) s"
cat "$OUT/summary.txt"
