#!/bin/bash
# Round 4, call 11: the two +-1 % launch-form decisions of the attention path re-checked at SUSTAINED load on one box (8 timed steps each, probe-normalised):
# staggered key walk on / off (X2V_ATTN_ROT=0), CFG pair pass vs one forward after the other (--no-cfg-pair).
set +e
OUT=gpurun_out/r04_call11
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
one() { tag=$1; shift; timeout 400 env "$@" python bench.py --steps 8 --warmup 2 --no-cpu-baseline ${EXTRA} > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"
  echo "$tag: $(python -c "import json; d=json.loads(open('$OUT/bench_$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; print('ms_per_step %.1f  attn_ms_per_forward %.2f  frac %.4f  frac_of_probe %.4f  probe %.0f  form: %s' % (d['ms_per_step'], r['avg_launch_ms']/r['forwards_per_launch'], r['frac'], r['frac_of_probe'], d['box_calibration']['mfma_probe_tflops'], d['config']['cfg_form'][:40]))")" | tee -a "$OUT/summary.txt"; }
EXTRA="" one default A=1
EXTRA="" one no_stagger X2V_ATTN_ROT=0
EXTRA="--no-cfg-pair --no-cfg-streams" one sequential A=1
EXTRA="" one default_again A=1
cat "$OUT/summary.txt"
