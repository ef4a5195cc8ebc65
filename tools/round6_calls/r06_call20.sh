#!/bin/bash
# Round 6, call 20: (1) LDS-DMA pieces of a workgroup's four waves issued in different MFMA slots (gemm256c.hip C_STAGGER: 2 = by wave parity, 4 = every wave its
# own slot; D = slots apart) — the one structural difference to the vendor kernel's loop not yet tested; x2v_check gemm for the bits, pgemm a-b-a-b for the clock.
# (2) rocprofv3 --kernel-trace --stats of the headline command alone (--no-other-configs: the other legs launch the same kernel instantiations and would mix into the averages).
set +e
OUT=gpurun_out/r06_call20
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
VARS="main stag2 stag4 stag4d2 stag2d1"
for v in $VARS; do
  if [ $v = main ]; then L=lightx2v_amd; else L=tools/probes/ab/$v; fi
  echo "$v: $(LD_LIBRARY_PATH=$L timeout 200 tools/x2v_check gemm 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
done
for shape in "75600 5120 13824" "75600 13824 5120" "151200 5120 5120" "75600 5120 5120" "20280 1536 8960" "20280 8960 1536"; do
  for rep in 1 2; do
    for v in $VARS; do
      if [ $v = main ]; then L=lightx2v_amd; else L=tools/probes/ab/$v; fi
      echo "$v ($shape): $(LD_LIBRARY_PATH=$L timeout 120 tools/x2v_check pgemm $shape 12 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
    done
  done
done
echo "gemm a/b done $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-other-configs > "$GRAFT_REPO_ROOT/$OUT/prof_bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof.err"); echo "prof rc=$?" | tee -a "$OUT/summary.txt"
find "$OUT/prof" -name "*kernel_trace.csv" -delete
head -8 "$OUT"/prof/*kernel_stats.csv | cut -c1-60,200-400 >> "$OUT/summary.txt"
echo "total $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
