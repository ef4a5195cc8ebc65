#!/bin/bash
# Round 6, call 21: (1) call 20's stagger variants lost 11 % (2 groups) / 21 % (4 groups) — but they selected the group with a scalar branch around every piece, and the
# loss scales with the taken branches (16 / 48 per K tile), so it measured the branches.  Here every group runs its own copy of the loop (no branch inside a K tile):
# wrap0 = the restructured source with one group (must equal main), stag2 / stag2d1 = two groups 3 / 1 slots apart, stag4 = four groups a slot apart (4 x the loop: I-cache).
# (2) roofline.traffic refreshed on this round's tree: the bench command (headline leg only) under --pmc FETCH_SIZE / WRITE_SIZE / TCC, separate passes.
set +e
OUT=gpurun_out/r06_call21
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
VARS="main wrap0 stag2 stag2d1 stag4"
for v in $VARS; do
  if [ $v = main ]; then L=lightx2v_amd; else L=tools/probes/ab/$v; fi
  echo "$v: $(LD_LIBRARY_PATH=$L timeout 200 tools/x2v_check gemm 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
done
for shape in "75600 5120 13824" "75600 13824 5120" "151200 5120 5120" "20280 1536 8960" "20280 8960 1536"; do
  for rep in 1 2; do
    for v in $VARS; do
      if [ $v = main ]; then L=lightx2v_amd; else L=tools/probes/ab/$v; fi
      echo "$v ($shape): $(LD_LIBRARY_PATH=$L timeout 120 tools/x2v_check pgemm $shape 12 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
    done
  done
done
echo "gemm a/b done $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/pmc/set$i" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > "$GRAFT_REPO_ROOT/$OUT/pmc_set$i.log" 2>&1); echo "pmc set $i rc=$?" >> "$OUT/summary.txt"
done
python tools/pmc_traffic.py "$OUT/pmc" "attn_fwd_v9_kernel<8, 8, true, false>" 75600 40 2 > "$OUT/pmc_attn_traffic.json" 2>> "$OUT/summary.txt"
find "$OUT/pmc" -name "*kernel_trace.csv" -delete; find "$OUT/pmc" -name "*counter_collection.csv" -size +8M -delete
cat "$OUT/pmc_attn_traffic.json" >> "$OUT/summary.txt"
echo "total $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
