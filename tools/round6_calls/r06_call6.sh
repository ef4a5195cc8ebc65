#!/bin/bash
# Round 6, call 6: the whole GPU suite on the cleaned-up library (nil-result A/B macros removed from attn / vae / gemm256c) incl. the new
# reference-HunyuanModel-through-the-plugin leg (reference staged by hand for this call), then the driver's bench command with the new
# `other_configs` legs (short: 2 timed steps).
set +e
OUT=gpurun_out/r06_call6
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
timeout 300 python -m pytest tests/test_plugin_reference.py -m gpu -q --timeout 280 > "$OUT/pytest_plugin.log" 2>&1; echo "pytest plugin rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a "$OUT/summary.txt"; tail -6 "$OUT/pytest_plugin.log" | cut -c1-300 >> "$OUT/summary.txt"; grep "REFERENCE_" "$OUT/pytest_plugin.log" >> "$OUT/summary.txt"
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > "$OUT/pytest_all.log" 2>&1; echo "pytest -m gpu rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a "$OUT/summary.txt"; tail -6 "$OUT/pytest_all.log" | cut -c1-300 >> "$OUT/summary.txt"
timeout 900 python bench.py --steps 2 --warmup 1 > "$OUT/bench_other_configs.json" 2> "$OUT/bench_other_configs.err"; echo "bench rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a "$OUT/summary.txt"
python - <<'PY' >> "$OUT/summary.txt" 2>&1
import json
d = json.loads([l for l in open("gpurun_out/r06_call6/bench_other_configs.json") if l.startswith("{")][-1])
print("headline ms_per_step %.1f value %.4f frac %.4f cpu_baseline kind %s" % (d["ms_per_step"], d["value"], d["roofline"]["frac"], d.get("cpu_baseline", {}).get("kind")))
for k, v in d.get("other_configs", {}).items():
    print(" ", k, json.dumps({a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if not isinstance(b, (dict, list))})[:400])
    if isinstance(v, dict) and isinstance(v.get("roofline"), dict):
        print("     roofline frac %.4f achieved %.1f" % (v["roofline"]["frac"], v["roofline"]["achieved"]))
PY
tail -3 "$OUT/bench_other_configs.err" >> "$OUT/summary.txt"
echo "total $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
