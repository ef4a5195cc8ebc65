#!/bin/bash
# Round 5, call 10: VAE halo convolution, slot form with two k-steps of fragment read-ahead (three fragment sets, step loop unrolled by 3) vs one k-step
set +e
OUT=gpurun_out/r05_call10
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
say() { echo "$@" | tee -a "$OUT/summary.txt"; }
timeout 600 python -m pytest tests/test_gpu_vae.py tests/test_gpu_hunyuan_vae.py -m gpu -q --timeout 500 -x > "$OUT/pytest_vae.log" 2>&1; say "pytest vae rc=$? ($(( $(date +%s) - t0 )) s): $(tail -1 "$OUT/pytest_vae.log" | cut -c1-120)"
for rep in 1 2; do
  for v in default vhdist1; do
    if [ "$v" = default ]; then unset X2V_LIB_PATH; else export X2V_LIB_PATH=tools/probes/ab/$v/libx2v_hip.so; fi
    timeout 200 python tools/vae_bench.py --split --reps 2 > "$OUT/vae_${v}_$rep.json" 2> "$OUT/vae_${v}_$rep.err"
    say "vae split rep$rep $v: $(python -c "import json; d=json.loads([l for l in open('$OUT/vae_${v}_$rep.json') if l.startswith('{')][-1]); print('%.3f s  %.1f TFLOP/s' % (d['seconds'], d['tflops_per_s']))" 2>&1)"
  done
done
unset X2V_LIB_PATH
timeout 200 python tools/vae_bench.py --conv16 --reps 2 > "$OUT/vae16.json" 2> "$OUT/vae16.err"; say "vae fp16-operands: $(python -c "import json; d=json.loads([l for l in open('$OUT/vae16.json') if l.startswith('{')][-1]); print('%.3f s  %.1f TFLOP/s' % (d['seconds'], d['tflops_per_s']))" 2>&1)"
timeout 200 python tools/hunyuan_vae_bench.py > "$OUT/hunyuan_vae.json" 2> "$OUT/hunyuan_vae.err"; say "hunyuan vae tile: $(python -c "import json; d=json.loads([l for l in open('$OUT/hunyuan_vae.json') if l.startswith('{')][-1]); print('%.4f s  %.1f TFLOP/s' % (d['seconds'], d['tflops_per_s']))" 2>&1)"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/pmc" -o pmc -- python "$GRAFT_REPO_ROOT/tools/vae_bench.py" --split --reps 1 > "$GRAFT_REPO_ROOT/$OUT/pmc.log" 2>&1); say "pmc rc=$?"
python tools/pmc_summary.py "$OUT/pmc" > "$OUT/pmc_vae_summary.txt" 2>&1; grep -A6 "conv16h_kernelILi3E\|conv16h_kernelILi4E" "$OUT/pmc_vae_summary.txt" | head -16 >> "$OUT/summary.txt"
find "$OUT/pmc" -name "*kernel_trace.csv" -delete; find "$OUT/pmc" -name "*counter_collection.csv" -size +20M -delete
say "total $(( $(date +%s) - t0 )) s"
cat "$OUT/summary.txt"
