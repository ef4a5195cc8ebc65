#!/bin/bash
# Round 6, call 10: Wan VAE decode 720p x 81f with the 128 x 96 kernel: latent frames per decoder pass (2 = default so far, 4, 5: fewer cache moves and tile tails,
# more memory), the step barrier's slot (30 / 40 / 47), and counters of the new kernel (matrix pipe busy, LDS, clock).
set +e
OUT=gpurun_out/r06_call10
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
for rep in 1 2; do
  for cf in 2 4 5; do
    echo "chunk $cf: $(timeout 300 python tools/vae_bench.py --split --chunk-frames $cf 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
  done
  for v in gbar30 gbar47; do
    echo "$v chunk 2: $(X2V_LIB_PATH=tools/probes/ab/$v/libx2v_hip.so timeout 300 python tools/vae_bench.py --split 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
  done
done
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/pmc/set$i" -o pmc -- python "$GRAFT_REPO_ROOT/tools/vae_bench.py" --split > "$GRAFT_REPO_ROOT/$OUT/pmc_set$i.log" 2>&1)
done
python - <<'PY' > "$OUT/pmc_summary.txt" 2>&1
import csv, glob, collections
# the 720p 96 -> 96 launches of the new kernel: grid 256 workgroups, the longest launches; mean counter value per launch over launches longer than 6 ms
dur = {}
for f in glob.glob("gpurun_out/r06_call10/pmc/set*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "conv16g" in r["Kernel_Name"]:
            dur[(f.split("/")[-2], r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/r06_call10/pmc/set*/*counter_collection.csv"):
    s = f.split("/")[-2]
    for r in csv.DictReader(open(f)):
        if "conv16g" in r["Kernel_Name"] and dur.get((s, r["Dispatch_Id"]), 0) > 6.0:
            acc[r["Counter_Name"]][(s, r["Dispatch_Id"])] += float(r["Counter_Value"])
for c, d in sorted(acc.items()):
    print("%-32s mean=%.6g n=%d" % (c, sum(d.values()) / len(d), len(d)))
big = [v for v in dur.values() if v > 6.0]
print("launches > 6 ms: n=%d mean %.3f ms (under the counter passes)" % (len(big), sum(big) / max(1, len(big))))
PY
find "$OUT/pmc" -name "*kernel_trace.csv" -size +5M -delete
echo "total $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
cat "$OUT/pmc_summary.txt"
