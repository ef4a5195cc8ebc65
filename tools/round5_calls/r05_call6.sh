#!/bin/bash
# Round 5, call 6: what the VAE halo convolution waits for — PMC passes of the 720p x 81f decode (tools/vae_bench.py --split), per kernel instantiation.
set +e
OUT=gpurun_out/r05_call6
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/pmc/set$i" -o pmc -- python "$GRAFT_REPO_ROOT/tools/vae_bench.py" --split --reps 1 > "$GRAFT_REPO_ROOT/$OUT/pmc_set$i.log" 2>&1); echo "pmc set$i rc=$? at $(( $(date +%s) - t0 )) s" | tee -a "$OUT/summary.txt"
done
python tools/pmc_summary.py "$OUT/pmc" > "$OUT/pmc_vae_summary.txt" 2>&1
grep -A22 "vae_conv16h_kernelILi3E\|vae_conv16h_kernelILi4E" "$OUT/pmc_vae_summary.txt" | head -60 >> "$OUT/summary.txt"
find "$OUT/pmc" -name "*kernel_trace.csv" -delete; find "$OUT/pmc" -name "*counter_collection.csv" -size +20M -delete
echo "total $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
