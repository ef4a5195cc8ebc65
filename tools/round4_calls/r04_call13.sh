#!/bin/bash
# Round 4, call 13 (last): the Wan drivers without the staggered key walk — model-level tests, the full-size attention launches, smoke, the default bench line;
# HunyuanVideo step with / without the stagger (its driver still sets it).
set +e
OUT=gpurun_out/r04_call13
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
timeout 240 python -m pytest tests/test_gpu_model.py tests/test_gpu_full_size.py -m gpu -q --timeout 240 -x -k "(cfg_pair or i2v_branch or tiny or attention_cfg_pair or attention_wan14b or teacache) and not config1" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a "$OUT/summary.txt"; tail -4 "$OUT/pytest.log" | cut -c1-200 >> "$OUT/summary.txt"
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> "$OUT/summary.txt"
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --probe-ms 800 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?" >> "$OUT/summary.txt"
python -c "import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['frac_of_probe'], r['kernel'][:90], r['traffic'])" >> "$OUT/summary.txt" 2>&1
for rot in 0 1; do X2V_ATTN_ROT=$rot timeout 120 python tools/hunyuan_bench.py --steps 2 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hunyuan X2V_ATTN_ROT=$rot ms_per_step', d['ms_per_step'])" >> "$OUT/summary.txt" 2>&1; done
cat "$OUT/summary.txt"
