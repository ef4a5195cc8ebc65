#!/bin/bash
# Round 5, call 1 — what round 4 built after its GPU minutes ran out, measured in one box:
#   1. w8a8 step (config #4) with the continuous fp8 GEMM (default) vs the ping-pong kernel (X2V_GEMM_FP8_CONTINUOUS=0), a/b/a/b, + rocprofv3 kernel stats
#   2. x2v_gemm_fp8_blocked (never run on a GPU so far): bit-equality with the row-major operator, on the ping-pong kernel and (X2V_GEMM_FP8_CONTINUOUS=2) the continuous one
#   3. bf16 continuous GEMM with the epilogue walk software-pipelined (variant build C_EPI_PIPELINED=1): bit-equality + timings vs today's
#   4. the continuous fp8 GEMM with the unscaled MFMA encoding (variant build C8_NOSCALE=1): bit-equality + timings
# Before the call, HERE (hipcc, no GPU needed):   tools/build_variant.sh epipipe -DC_EPI_PIPELINED=1; tools/build_variant.sh noscale -DC8_NOSCALE=1
set +e
OUT=gpurun_out/r05_call1
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
one() { tag=$1; shift; timeout 120 env "$@" python bench.py --fp8 --distill --steps 4 --warmup 1 --no-cpu-baseline --probe-ms 500 > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"
  echo "$tag: $(python -c "import json; d=json.loads(open('$OUT/bench_$tag.json').read().strip().splitlines()[-1]); print('ms_per_step %.1f' % d['ms_per_step'])" 2>&1)" | tee -a "$OUT/summary.txt"; }
one fp8_continuous_1 X2V_GEMM_FP8_CONTINUOUS=1
one fp8_pingpong_1 X2V_GEMM_FP8_CONTINUOUS=0
one fp8_continuous_2 X2V_GEMM_FP8_CONTINUOUS=1
one fp8_pingpong_2 X2V_GEMM_FP8_CONTINUOUS=0
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof_fp8" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --fp8 --distill --steps 2 --warmup 1 --no-cpu-baseline --no-calibration > "$GRAFT_REPO_ROOT/$OUT/prof_fp8_bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof_fp8.err")
f=$(find "$OUT/prof_fp8" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_fp8_distill.csv" && head -8 "$f" | cut -c1-160 >> "$OUT/summary.txt"
find "$OUT/prof_fp8" -type f ! -name "*kernel_stats.csv" -delete
timeout 120 python tools/gemm_fp8_blocked_check.py > "$OUT/fp8_blocked_pingpong.json" 2> "$OUT/fp8_blocked_pingpong.err"; echo "blocked fp8 (ping-pong) rc=$?: $(cut -c1-400 "$OUT/fp8_blocked_pingpong.json")" | tee -a "$OUT/summary.txt"
X2V_GEMM_FP8_CONTINUOUS=2 timeout 120 python tools/gemm_fp8_blocked_check.py > "$OUT/fp8_blocked_continuous.json" 2> "$OUT/fp8_blocked_continuous.err"; echo "blocked fp8 (continuous) rc=$?: $(cut -c1-400 "$OUT/fp8_blocked_continuous.json")" | tee -a "$OUT/summary.txt"
if [ -f tools/probes/ab/epipipe/libx2v_hip.so ]; then
  timeout 200 python tools/gemm_continuous_check.py > "$OUT/gemm_bf16_today.json" 2> "$OUT/gemm_bf16_today.err"; echo "bf16 today rc=$?" | tee -a "$OUT/summary.txt"
  X2V_LIB_PATH=tools/probes/ab/epipipe/libx2v_hip.so timeout 200 python tools/gemm_continuous_check.py > "$OUT/gemm_bf16_epipipe.json" 2> "$OUT/gemm_bf16_epipipe.err"; echo "bf16 pipelined epilogue rc=$?" | tee -a "$OUT/summary.txt"
  python - >> "$OUT/summary.txt" <<'PY'
import json
for tag in ("today", "epipipe"):
    try:
        d = json.loads(open(f"gpurun_out/r05_call1/gemm_bf16_{tag}.json").read().strip().splitlines()[-1])
        print(tag, "equality_cases", d["equality_cases"], "mismatches", d["n_mismatches"])
        for r in d.get("timing", []):
            print("  ", r["M"], r["K"], r["N"], {k: v for k, v in r.items() if "continuous" in k})
    except Exception as e:
        print(tag, "unreadable:", e)
PY
fi
if [ -f tools/probes/ab/noscale/libx2v_hip.so ]; then  # 4. the unscaled MFMA encoding in the continuous fp8 GEMM (tools/build_variant.sh noscale -DC8_NOSCALE=1): same bits? faster?
  timeout 90 python tools/gemm_fp8_continuous_check.py > "$OUT/fp8_scaled.jsonl" 2> "$OUT/fp8_scaled.err"; echo "fp8 continuous (scaled encoding) rc=$?" | tee -a "$OUT/summary.txt"
  X2V_LIB_PATH=tools/probes/ab/noscale/libx2v_hip.so timeout 90 python tools/gemm_fp8_continuous_check.py > "$OUT/fp8_noscale.jsonl" 2> "$OUT/fp8_noscale.err"; echo "fp8 continuous (unscaled encoding) rc=$?" | tee -a "$OUT/summary.txt"
  head -1 "$OUT/fp8_noscale.jsonl" | cut -c1-300 >> "$OUT/summary.txt"
  python - >> "$OUT/summary.txt" <<'PY'
import json
for tag in ("scaled", "noscale"):
    try:
        for line in open(f"gpurun_out/r05_call1/fp8_{tag}.jsonl").read().strip().splitlines()[1:]:
            r = json.loads(line)
            print(tag, r["M"], r["K"], r["N"], {k: v for k, v in r.items() if "continuous" in k})
    except Exception as e:
        print(tag, "unreadable:", e)
PY
fi
cat "$OUT/summary.txt"
