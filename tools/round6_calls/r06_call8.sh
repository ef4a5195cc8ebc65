#!/bin/bash
# Round 6, call 8: where gemm256c waits (call 7's counters: 4.5x the vendor kernel's s_waitcnt time at 75600 x 13824 -> 5120).  Knock-out builds (results invalid,
# timing only): ko1 = no "tile t+1 has landed" vmcnt wait, ko2 = no "fragments are in registers" lgkmcnt wait; slot-plan variants: DMA pieces every 5 / 6 slots
# (no late pieces), READY at 106 with a read per slot, k-step-1 reads a slot apart, FREE at 44.  x2v_check pgemm M N K iters, a-b-a-b by the clock.
set +e
OUT=gpurun_out/r06_call8
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
for shape in "75600 5120 13824" "75600 13824 5120" "151200 5120 5120" "20280 1536 8960"; do
  for rep in 1 2; do
    for v in main ko1 ko2 step5 step6 ready106 step5r106 r1s1 free44; do
      if [ $v = main ]; then L=lightx2v_amd; else L=tools/probes/ab/$v; fi
      echo "$v ($shape): $(LD_LIBRARY_PATH=$L timeout 120 tools/x2v_check pgemm $shape 12 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
    done
  done
done
echo "total $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
