#!/bin/bash
# Round 6, call 24: the self-attention kernel with a matrix half-step's closing barrier 1..4 fragment slots (2 MFMAs each) BEFORE its end (attn.hip X2V_A9_TAIL): the
# waves behind the barrier then start their fragment reads under the last MFMAs of the waves in front of it instead of behind them (the matrix pipe's idle share is
# 30 %; a hand-over costs a barrier + an LDS round trip).  Bit-identical arithmetic.  x2v_check attn for every build, then pattn a-b-a-b at the 14B shape, the
# rank-of-8 shape, the 1.3B shape and HunyuanVideo's.
set +e
OUT=gpurun_out/r06_call24
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
VARS="main tail1 tail2 tail3 tail4"
for v in $VARS; do
  if [ $v = main ]; then L=lightx2v_amd; else L=tools/probes/ab/$v; fi
  echo "$v: $(LD_LIBRARY_PATH=$L timeout 200 tools/x2v_check attn 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
done
for shape in "75600 40 12" "75600 5 40" "20280 12 100" "119056 24 6"; do
  for rep in 1 2 3; do
    for v in $VARS; do
      if [ $v = main ]; then L=lightx2v_amd; else L=tools/probes/ab/$v; fi
      echo "$v ($shape): $(LD_LIBRARY_PATH=$L timeout 120 tools/x2v_check pattn 12 $shape 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
    done
  done
done
echo "total $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
