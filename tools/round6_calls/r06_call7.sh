#!/bin/bash
# Round 6, call 7: the deep-K GEMM gap to hipBLASLt, counter by counter.  (1) tools/gemm_vs_hipblaslt.py on the tree's library and on the A/B build whose
# k-step holds src0 (the W fragment) over 8 consecutive MFMAs as the vendor kernel does (tools/probes/ab/src0); (2) rocprofv3 --pmc passes (kernel-trace only,
# one counter set per pass) of the vendor kernel and of gemm256c at 75600 x 13824 -> 5120 and 75600 x 5120 -> 13824.
set +e
OUT=gpurun_out/r06_call7
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
timeout 300 python -m pytest tests/test_plugin_reference.py -m gpu -q --timeout 280 > "$OUT/pytest_plugin.log" 2>&1; echo "pytest plugin rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/pytest_plugin.log" | cut -c1-300 >> "$OUT/summary.txt"; grep "REFERENCE_" "$OUT/pytest_plugin.log" >> "$OUT/summary.txt"
for rep in 1 2; do
  ITERS=20 timeout 300 python tools/gemm_vs_hipblaslt.py > "$OUT/vs_main_$rep.json" 2> "$OUT/vs_main_$rep.err"; echo "main rep $rep rc=$?" >> "$OUT/summary.txt"
  X2V_LIB_PATH=tools/probes/ab/src0/libx2v_hip.so ITERS=20 timeout 300 python tools/gemm_vs_hipblaslt.py > "$OUT/vs_src0_$rep.json" 2> "$OUT/vs_src0_$rep.err"; echo "src0 rep $rep rc=$?" >> "$OUT/summary.txt"
done
python - <<'PY' >> "$OUT/summary.txt" 2>&1
import json
for tag in ("main_1", "src0_1", "main_2", "src0_2"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r06_call7/vs_{tag}.json") if l.startswith("{")][-1])
    except Exception as e:
        print(tag, "unreadable", e); continue
    for r in d["gemm"]:
        print(tag, r["M"], r["K"], r["N"], "ours %.0f %.0f  hipblaslt %.0f %.0f" % (r["x2v_TFLOPs_0"], r["x2v_TFLOPs_1"], r["hipblaslt_TFLOPs_0"], r["hipblaslt_TFLOPs_1"]))
PY
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
  i=$((i+1))
  for shape in "75600 13824 5120" "75600 5120 13824"; do
    tagshape=$(echo $shape | tr ' ' 'x')
    for which in blaslt ours; do
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/pmc/${which}_${tagshape}_set$i" -o pmc -- python "$GRAFT_REPO_ROOT/tools/gemm_pmc_one.py" $which $shape 3 > "$GRAFT_REPO_ROOT/$OUT/pmc_${which}_${tagshape}_set$i.log" 2>&1)
    done
  done
done
for shape in 75600x13824x5120 75600x5120x13824; do
  echo "##### $shape" >> "$OUT/pmc_summary.txt"
  python tools/pmc_summary.py "$OUT"/pmc/*_${shape}_set* >> "$OUT/pmc_summary.txt" 2>&1
done
find "$OUT/pmc" -name "*kernel_trace.csv" -size +5M -delete
echo "total $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
cat "$OUT/summary.txt"; grep -v "elementwise\|distribution" "$OUT/pmc_summary.txt" | head -150
