#!/bin/bash
# Round 5, call 9: knock-out timing probes of the VAE halo convolution (results invalid by construction): what does a step wait for?
set +e
OUT=gpurun_out/r05_call9
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=. X2V_PROBE_RUN=1
t0=$(date +%s)
say() { echo "$@" | tee -a "$OUT/summary.txt"; }
for v in vhinter vhprobe1 vhprobe2 vhprobe3 vhprobe4 vhprobe8 vhinter; do
  X2V_LIB_PATH=tools/probes/ab/$v/libx2v_hip.so timeout 150 python tools/vae_bench.py --split --reps 2 > "$OUT/vae_$v.json" 2> "$OUT/vae_$v.err"
  say "$v: $(python -c "import json; d=json.loads([l for l in open('$OUT/vae_$v.json') if l.startswith('{')][-1]); print('%.3f s' % d['seconds'])" 2>&1 | tail -1)"
done
say "total $(( $(date +%s) - t0 )) s"
cat "$OUT/summary.txt"
