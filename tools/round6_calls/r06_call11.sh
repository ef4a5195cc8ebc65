#!/bin/bash
# Round 6, call 11: the 128-pixel kernel with the conflict-free 64-byte-row swizzle and its one-cout-block form for the 3-channel head: VAE GPU tests, decode timings
# (2 and 4 latent frames per pass), kernel stats, and the LDS counters again.
set +e
OUT=gpurun_out/r06_call11
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_vae.py -m gpu -q --timeout 600 -x > "$OUT/pytest_vae.log" 2>&1; echo "pytest vae rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a "$OUT/summary.txt"; tail -12 "$OUT/pytest_vae.log" | cut -c1-400 >> "$OUT/summary.txt"
for rep in 1 2; do
  for cf in 2 4; do
    echo "chunk $cf: $(timeout 300 python tools/vae_bench.py --split --chunk-frames $cf 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
  done
done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof" -o vae -- python "$GRAFT_REPO_ROOT/tools/vae_bench.py" --split --chunk-frames 4 > "$GRAFT_REPO_ROOT/$OUT/prof.log" 2>&1)
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-200 >> "$OUT/summary.txt"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/pmc/set1" -o pmc -- python "$GRAFT_REPO_ROOT/tools/vae_bench.py" --split > "$GRAFT_REPO_ROOT/$OUT/pmc_set1.log" 2>&1)
python - <<'PY' > "$OUT/pmc_summary.txt" 2>&1
import csv, glob, collections
dur = {}
for f in glob.glob("gpurun_out/r06_call11/pmc/set*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "conv16g" in r["Kernel_Name"]:
            dur[(f.split("/")[-2], r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/r06_call11/pmc/set*/*counter_collection.csv"):
    s = f.split("/")[-2]
    for r in csv.DictReader(open(f)):
        if "conv16g" in r["Kernel_Name"] and dur.get((s, r["Dispatch_Id"]), 0) > 6.0:
            acc[r["Counter_Name"]][(s, r["Dispatch_Id"])] += float(r["Counter_Value"])
for c, d in sorted(acc.items()):
    print("%-32s mean=%.6g n=%d" % (c, sum(d.values()) / len(d), len(d)))
big = [v for v in dur.values() if v > 6.0]
print("launches > 6 ms: n=%d mean %.3f ms (under the counter pass)" % (len(big), sum(big) / max(1, len(big))))
PY
find "$OUT" -name "*kernel_trace.csv" -size +5M -delete
echo "total $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
cat "$OUT/pmc_summary.txt"
