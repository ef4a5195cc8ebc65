#!/bin/bash
# Round 5, call 5: the VAE halo convolution with its weight slabs in a 4-slot ring (three steps of flight): tests, then A-B-A-B against the one-step-ahead
# form (ring of 2, same bare barriers) on the 720p x 81f decode, the Hunyuan tile, and the end-to-end config #4 run.
set +e
OUT=gpurun_out/r05_call5
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
say() { echo "$@" | tee -a "$OUT/summary.txt"; }
timeout 600 python -m pytest tests/test_gpu_vae.py tests/test_gpu_hunyuan_vae.py -m gpu -q --timeout 500 -x > "$OUT/pytest_vae.log" 2>&1; say "pytest vae rc=$? ($(( $(date +%s) - t0 )) s): $(tail -1 "$OUT/pytest_vae.log" | cut -c1-120)"
for rep in 1 2; do
  for v in default vhring2; do
    if [ "$v" = default ]; then unset X2V_LIB_PATH; else export X2V_LIB_PATH=tools/probes/ab/$v/libx2v_hip.so; fi
    timeout 200 python tools/vae_bench.py --split --reps 2 > "$OUT/vae_${v}_$rep.json" 2> "$OUT/vae_${v}_$rep.err"
    say "vae split rep$rep $v: $(python -c "import json; d=json.loads([l for l in open('$OUT/vae_${v}_$rep.json') if l.startswith('{')][-1]); print('%.3f s  %.1f TFLOP/s' % (d['seconds'], d['tflops_per_s']))" 2>&1)"
  done
done
for v in default vhring2; do
  if [ "$v" = default ]; then unset X2V_LIB_PATH; else export X2V_LIB_PATH=tools/probes/ab/$v/libx2v_hip.so; fi
  timeout 200 python tools/vae_bench.py --conv16 --reps 2 > "$OUT/vae16_${v}.json" 2> "$OUT/vae16_${v}.err"
  say "vae fp16-operands $v: $(python -c "import json; d=json.loads([l for l in open('$OUT/vae16_${v}.json') if l.startswith('{')][-1]); print('%.3f s  %.1f TFLOP/s' % (d['seconds'], d['tflops_per_s']))" 2>&1)"
  timeout 200 python tools/hunyuan_vae_bench.py > "$OUT/hunyuan_vae_$v.json" 2> "$OUT/hunyuan_vae_$v.err"; say "hunyuan vae tile $v: $(python -c "import json; d=json.loads([l for l in open('$OUT/hunyuan_vae_$v.json') if l.startswith('{')][-1]); print('%.4f s  %.1f TFLOP/s' % (d['seconds'], d['tflops_per_s']))" 2>&1)"
done
unset X2V_LIB_PATH
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof_vae" -o v -- python "$GRAFT_REPO_ROOT/tools/vae_bench.py" --split --reps 1 > /dev/null 2> "$GRAFT_REPO_ROOT/$OUT/prof_vae.err"); f=$(find "$OUT/prof_vae" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats_wan_vae_720p81f_split.csv" && head -12 "$f" | cut -c1-90,250-330 >> "$OUT/summary.txt"
find "$OUT/prof_vae" -name "*kernel_trace.csv" -delete
say "total $(( $(date +%s) - t0 )) s"
cat "$OUT/summary.txt"
