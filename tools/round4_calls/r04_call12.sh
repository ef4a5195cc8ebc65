#!/bin/bash
# Round 4, call 12: staggered key walk on / off, A-B-A-B at sustained load (6 timed steps each) — confirmation of call 11.
set +e
OUT=gpurun_out/r04_call12
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
one() { tag=$1; shift; timeout 300 env "$@" python bench.py --steps 6 --warmup 2 --no-cpu-baseline --probe-ms 800 > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"
  echo "$tag: $(python -c "import json; d=json.loads(open('$OUT/bench_$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; print('ms_per_step %.1f  attn_ms %.2f  frac_of_probe %.4f  probe %.0f' % (d['ms_per_step'], r['avg_launch_ms'], r['frac_of_probe'], d['box_calibration']['mfma_probe_tflops']))")" | tee -a "$OUT/summary.txt"; }
one no_stagger_1 X2V_ATTN_ROT=0
one stagger_1 X2V_ATTN_ROT=1
one no_stagger_2 X2V_ATTN_ROT=0
one stagger_2 X2V_ATTN_ROT=1
cat "$OUT/summary.txt"
