#!/bin/bash
# Round 6, call 18: persistent short-walk form with the q / o traffic in the early role (LDS-staged output flush): checks, A/B, knock-outs.
set +e
OUT=gpurun_out/r06_call18
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
timeout 300 tools/x2v_check attn > "$OUT/x2v_check_attn.log" 2>&1; echo "x2v_check attn rc=$?" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/x2v_check_attn.log" >> "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 600 -k "attention" > "$OUT/pytest_attn.log" 2>&1; echo "pytest attention rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a "$OUT/summary.txt"; tail -15 "$OUT/pytest_attn.log" | cut -c1-300 >> "$OUT/summary.txt"
timeout 600 python tools/probes/cross_attn_ab.py 2>&1 | grep "cross attention" | cut -c1-330 | tee -a "$OUT/summary.txt"
for v in p9ko1 p9ko6 p9ko7; do
  export X2V_LIB_PATH=$PWD/tools/probes/ab/$v/libx2v_hip.so
  echo "== $v (timing probe, results invalid)" | tee -a "$OUT/summary.txt"
  X2V_AB_SHAPES="75600,512,40;75600,1024,40" timeout 300 python tools/probes/cross_attn_ab.py 2>&1 | grep "cross attention" | cut -c1-330 | tee -a "$OUT/summary.txt"
done
echo "total $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
