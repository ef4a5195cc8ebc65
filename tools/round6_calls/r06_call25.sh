#!/bin/bash
# Round 6, call 25: counter pass on the 128-cout form of the 128-pixel convolution kernel inside the HunyuanVideo VAE's 720p x 129f tiled decode (matrix pipe busy share,
# LDS instructions / conflicts), launches above 1.5 ms; the same for the final Wan decode's vae_conv16g_kernel<6> launches above 6 ms.
set +e
OUT=gpurun_out/r06_call25
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/pmc/hunyuan" -o pmc -- python "$GRAFT_REPO_ROOT/tools/hunyuan_vae_bench.py" --full > "$GRAFT_REPO_ROOT/$OUT/pmc_hunyuan.log" 2>&1)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/pmc/wan" -o pmc -- python "$GRAFT_REPO_ROOT/tools/vae_bench.py" --split > "$GRAFT_REPO_ROOT/$OUT/pmc_wan.log" 2>&1)
python - <<'PY' > "$OUT/summary.txt" 2>&1
import csv, glob, collections
for tag, thr in (("hunyuan", 1.5), ("wan", 6.0)):
    dur = {}
    for f in glob.glob(f"gpurun_out/r06_call25/pmc/{tag}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv16g" in r["Kernel_Name"]:
                dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f"gpurun_out/r06_call25/pmc/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv16g" in r["Kernel_Name"] and dur.get(r["Dispatch_Id"], 0) > thr:
                acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    print(f"== {tag}: vae_conv16g launches above {thr} ms")
    m = {}
    for c, d in sorted(acc.items()):
        m[c] = sum(d.values()) / len(d)
        print("   %-32s mean=%.6g n=%d" % (c, m[c], len(d)))
    big = [v for v in dur.values() if v > thr]
    ms = sum(big) / max(1, len(big))
    print("   launches: n=%d mean %.3f ms (under the counter pass)" % (len(big), ms))
    if "GRBM_GUI_ACTIVE" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        print("   matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) = %.3f; clock = GRBM_GUI_ACTIVE / 8 / time = %.3f GHz"
              % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024), m["GRBM_GUI_ACTIVE"] / 8 / (ms * 1e-3) / 1e9))
    if "SQ_LDS_IDX_ACTIVE" in m and "SQ_LDS_BANK_CONFLICT" in m:
        print("   LDS bank conflicts / LDS cycles = %.4f" % (m["SQ_LDS_BANK_CONFLICT"] / max(1.0, m["SQ_LDS_IDX_ACTIVE"])))
PY
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*counter_collection.csv" -size +4M -delete
echo "total $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
