#!/bin/bash
# Round 6, call 17: where the persistent short-walk form spends its block overhead: knock-out builds (timing probes, results invalid).
set +e
OUT=gpurun_out/r06_call17
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
for rep in 1 2; do
for v in main p9ko1 p9ko6 p9ko7 p9ko8; do
  if [ $v = main ]; then unset X2V_LIB_PATH; else export X2V_LIB_PATH=$PWD/tools/probes/ab/$v/libx2v_hip.so; fi
  echo "== $v" | tee -a "$OUT/summary.txt"
  X2V_AB_SHAPES="75600,512,40;75600,1024,40" timeout 300 python tools/probes/cross_attn_ab.py 2>&1 | grep "cross attention" | cut -c1-330 | tee -a "$OUT/summary.txt"
done
done
