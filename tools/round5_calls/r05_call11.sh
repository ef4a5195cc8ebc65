#!/bin/bash
# Round 5, call 11: VAE halo convolution as it ships (slot form, one k-step of read-ahead, halo pieces spread over taps 0..5): tests, and the halo spreading A/B
set +e
OUT=gpurun_out/r05_call11
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
say() { echo "$@" | tee -a "$OUT/summary.txt"; }
timeout 600 python -m pytest tests/test_gpu_vae.py tests/test_gpu_hunyuan_vae.py -m gpu -q --timeout 500 -x > "$OUT/pytest_vae.log" 2>&1; say "pytest vae rc=$? ($(( $(date +%s) - t0 )) s): $(tail -1 "$OUT/pytest_vae.log" | cut -c1-120)"
for rep in 1 2; do
  for v in default vhnospread vhfenced; do
    if [ "$v" = default ]; then unset X2V_LIB_PATH; else export X2V_LIB_PATH=tools/probes/ab/$v/libx2v_hip.so; fi
    timeout 200 python tools/vae_bench.py --split --reps 2 > "$OUT/vae_${v}_$rep.json" 2> "$OUT/vae_${v}_$rep.err"
    say "vae split rep$rep $v: $(python -c "import json; d=json.loads([l for l in open('$OUT/vae_${v}_$rep.json') if l.startswith('{')][-1]); print('%.3f s  %.1f TFLOP/s' % (d['seconds'], d['tflops_per_s']))" 2>&1)"
  done
done
unset X2V_LIB_PATH
say "total $(( $(date +%s) - t0 )) s"
cat "$OUT/summary.txt"
