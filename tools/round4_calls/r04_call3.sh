#!/bin/bash
# Round 4, call 3: what do the continuous kernel's 8-byte stores cost?  (probe build with the stores predicated off; results invalid)
set +e
OUT=gpurun_out/r04_call3
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
for tag in default nostore; do
  if [ $tag = default ]; then unset X2V_LIB_PATH; else export X2V_LIB_PATH=$PWD/tools/probes/ab/$tag/libx2v_hip.so; fi
  TIME_ANYWAY=1 ITERS=20 timeout 200 python tools/gemm_continuous_check.py > "$OUT/gemm_$tag.json" 2> "$OUT/gemm_$tag.err"; echo "$tag rc=$?" | tee -a "$OUT/summary.txt"
  python - "$OUT/gemm_$tag.json" >> "$OUT/summary.txt" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("mismatches", d["n_mismatches"])
for r in d.get("timing", []):
    print(r["M"], r["K"], r["N"], " ".join(f"{e}: {r[e+'_one_tile_TFLOPs_1']:.0f}/{r[e+'_continuous_TFLOPs_1']:.0f}" for e in ("plain","gelu","resid")))
PY
done
cat "$OUT/summary.txt"
