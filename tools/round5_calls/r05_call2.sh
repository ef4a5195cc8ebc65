#!/bin/bash
# Round 5, call 2: first contact of this round's changes + measurements that need no new kernel.
#   1. targeted GPU tests: GEMM dispatch / blocked fp8 / strided equality, the dist workers (w8a8 Ulysses leg, RCCL world-1 attend_blocked), fp8 model tests
#   2. smoke
#   3. the reference CPU baseline on this box's host cores (thread sweep)
#   4. attention priority / prefetch-depth A-B-A-B (variant libraries under tools/probes/ab: a9prio1/2/3, a9depth6) with x2v_check pattn at sustained length
#   5. config-#2 accounting: rocprofv3 kernel trace of the 1.3B 480p step in the SEQUENTIAL form -> per-kernel fractions + idle share (tools/kernel_gaps.py)
#   6. i2v 720p bench line (the reference's published configuration), 2 timed steps
set +e
OUT=gpurun_out/r05_call2
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
run() { name=$1; shift; t0=$(date +%s); "$@"; echo "$name rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a "$OUT/summary.txt"; }
run pytest timeout 900 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_dist.py tests/test_abi.py tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q --timeout 600 --durations=8 -x > "$OUT/pytest.log" 2>&1; tail -14 "$OUT/pytest.log" | cut -c1-220 >> "$OUT/summary.txt"
run smoke timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log" >> "$OUT/summary.txt"
run ref_cpu timeout 400 env HIP_VISIBLE_DEVICES= python -m oracle.ref_cpu_baseline --threads 8,16,32,64,128 > "$OUT/ref_cpu_baseline.json" 2> "$OUT/ref_cpu.err"; cut -c1-900 "$OUT/ref_cpu_baseline.json" >> "$OUT/summary.txt"
# 4. attention A/B: 12 launches of one 40-head 75600-token forward each (~1 s per run), A-B-A-B over the variants, twice
for rep in 1 2; do
  for v in default a9prio1 a9prio2 a9prio3 a9depth6; do
    if [ "$v" = default ]; then L=lightx2v_amd; else L=tools/probes/ab/$v; fi
    echo "rep$rep $v: $(LD_LIBRARY_PATH=$L timeout 120 tools/x2v_check pattn 12 75600 40 12 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
  done
done
# 5. config #2, sequential form
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof13" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --workload wan1.3b_480px49f --no-cfg-streams --no-cfg-pair --steps 2 --warmup 1 --no-cpu-baseline --no-calibration > "$GRAFT_REPO_ROOT/$OUT/prof13_bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof13.err"); echo "prof13 rc=$?" | tee -a "$OUT/summary.txt"
kt=$(find "$OUT/prof13" -name "*kernel_trace.csv" | head -1); ks=$(find "$OUT/prof13" -name "*kernel_stats.csv" | head -1)
[ -n "$kt" ] && python tools/kernel_gaps.py "$kt" 20280 1536 8960 12 > "$OUT/config2_accounting.json" 2>> "$OUT/summary.txt"; [ -n "$ks" ] && cp "$ks" "$OUT/kernel_stats_wan1.3b_480p_sequential.csv"
find "$OUT/prof13" -name "*kernel_trace.csv" -delete
python -c "
import json; d=json.load(open('$OUT/config2_accounting.json')); print('config2 sequential: wall %.1f ms busy %.1f ms idle %.3f median gap %.1f us' % (d['wall_ms'], d['busy_ms'], d['idle_share'], d['median_gap_us'])); print(json.dumps(d['gemm_by_shape'])); [print(k['kernel'][:50], k['calls'], round(k['mean_us'],1), round(k['share_of_kernel_time'],4), k.get('tflops')) for k in d['kernels'][:14]]" >> "$OUT/summary.txt" 2>&1
for f in two_streams sequential; do
  fl=$([ $f = sequential ] && echo "--no-cfg-streams --no-cfg-pair" || echo "")
  timeout 200 python bench.py --workload wan1.3b_480px49f $fl --steps 6 --warmup 2 --no-cpu-baseline --probe-ms 500 > "$OUT/bench13_$f.json" 2> "$OUT/bench13_$f.err"
  echo "bench13 $f: $(python -c "import json; d=json.loads(open('$OUT/bench13_$f.json').read().strip().splitlines()[-1]); print('ms_per_step %.1f  %s' % (d['ms_per_step'], d['config']['cfg_form'][:40]))" 2>&1)" | tee -a "$OUT/summary.txt"
done
# 6. i2v
run bench_i2v timeout 400 python bench.py --i2v --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_i2v_720p.json" 2> "$OUT/bench_i2v.err"
python -c "import json; d=json.loads(open('$OUT/bench_i2v_720p.json').read().strip().splitlines()[-1]); print('i2v 720p: ms_per_step %.1f frac %.3f kernels %s' % (d['ms_per_step'], d['roofline']['frac'], json.dumps(d['kernels'])[:600]))" >> "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
