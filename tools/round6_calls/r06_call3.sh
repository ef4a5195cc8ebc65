#!/bin/bash
# Round 6, call 3: where the producer/consumer kernel's interval goes: per-wave cycle sums between / at the barriers (probe build, X2V_PC_TRACE)
# for the full kernel and for the knock-outs "O-waves without MFMAs" and "S-waves without MFMAs"; more knock-outs by the clock.
set +e
OUT=gpurun_out/r06_call3
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
echo "v9      : $(X2V_ATTN_PC=0 timeout 120 tools/x2v_check pattn 12 75600 40 6 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
echo "pc      : $(X2V_ATTN_PC=1 timeout 120 tools/x2v_check pattn 12 75600 40 6 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
for v in knock1 knock4 knock5 knock6; do
  echo "pc_$v: $(X2V_ATTN_PC=1 LD_LIBRARY_PATH=tools/probes/ab/pc_$v timeout 120 tools/x2v_check pattn 12 75600 40 6 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
done
for v in trace trace_k1 trace_k3; do
  echo "== pc_$v" | tee -a "$OUT/summary.txt"
  X2V_DUMP_TRACE=1 X2V_ATTN_PC=1 LD_LIBRARY_PATH=tools/probes/ab/pc_$v timeout 120 tools/x2v_check pattn 12 75600 40 3 2>&1 | tail -9 | tee -a "$OUT/summary.txt"
done
echo "total $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
