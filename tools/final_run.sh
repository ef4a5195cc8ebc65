#!/bin/bash
# Round-end validation batch (one gpurun call): checks, the whole GPU test suite, smoke, the default bench line, the rocprof
# summary of the same command, and the secondary measurements quoted in DESIGN.md.  Every stage has its own timeout.
set +e
OUT=gpurun_out/${RUN_TAG:-final}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
run() { name=$1; shift; t0=$(date +%s); "$@"; echo "$name rc=$? ($(( $(date +%s) - t0 )) s)" >> "$OUT/summary.txt"; }  # (stdout of "$@" may be redirected by the caller)
for m in probe misc norm rope gemm attn fp8 conv; do run "check_$m" timeout 200 tools/x2v_check $m > "$OUT/check_$m.log" 2>&1; tail -1 "$OUT/check_$m.log" >> "$OUT/summary.txt"; done
run pytest timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --durations=12 > "$OUT/pytest.log" 2>&1; tail -18 "$OUT/pytest.log" | cut -c1-200 >> "$OUT/summary.txt"; cp gpurun_out/parity_summary.jsonl "$OUT/" 2>/dev/null
run smoke timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; tail -2 "$OUT/smoke.log" >> "$OUT/summary.txt"
# the DRIVER's exact command (BENCH_rNN.json is produced by it: 25 sustained steps at the 1400 W cap, not the 3-step default)
run bench_driver_command timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_command.json" 2> "$OUT/bench_driver_command.err"; cat "$OUT/bench_driver_command.json" >> "$OUT/summary.txt"
run bench_default timeout 900 python bench.py --no-cpu-baseline > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; cat "$OUT/bench_default.json" >> "$OUT/summary.txt"
# kernel stats of the headline leg alone: the other_configs legs launch the same kernel instantiations and would mix into the per-kernel averages
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-other-configs > "$GRAFT_REPO_ROOT/$OUT/prof_bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof.err"); echo "prof rc=$?" | tee -a "$OUT/summary.txt"
find "$OUT/prof" -name "*kernel_trace.csv" -size +20M -delete
run bench13 timeout 600 python bench.py --workload wan1.3b_480px49f --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench13.json" 2> "$OUT/bench13.err"; cat "$OUT/bench13.json" >> "$OUT/summary.txt"
run bench_fp8_distill timeout 600 python bench.py --fp8 --distill --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench14_fp8_distill.json" 2> "$OUT/bench14_fp8_distill.err"; cat "$OUT/bench14_fp8_distill.json" >> "$OUT/summary.txt"
run gemm_vs_hipblaslt timeout 300 python tools/gemm_vs_hipblaslt.py > "$OUT/gemm_vs_hipblaslt.json" 2> "$OUT/gemm_cmp.err"; cat "$OUT/gemm_vs_hipblaslt.json" >> "$OUT/summary.txt"
run hunyuan timeout 600 python tools/hunyuan_bench.py > "$OUT/hunyuan13b.json" 2> "$OUT/hunyuan13b.err"; cat "$OUT/hunyuan13b.json" >> "$OUT/summary.txt"
run e2e_13b timeout 300 python tools/e2e.py --workload wan1.3b_480px49f --steps 50 > "$OUT/e2e_wan13b_480p.json" 2> "$OUT/e2e13.err"; cat "$OUT/e2e_wan13b_480p.json" >> "$OUT/summary.txt"
run e2e_fp8_distill timeout 400 python tools/e2e.py --fp8 --distill > "$OUT/e2e_wan14b_fp8_distill.json" 2> "$OUT/e2e_fp8.err"; cat "$OUT/e2e_wan14b_fp8_distill.json" >> "$OUT/summary.txt"
run vae_hunyuan timeout 400 python tools/hunyuan_vae_bench.py --full > "$OUT/vae_hunyuan_720p129f.json" 2> "$OUT/vae_hunyuan.err"; cat "$OUT/vae_hunyuan_720p129f.json" >> "$OUT/summary.txt"
run vae_wan_split timeout 300 python tools/vae_bench.py --latent 16,21,90,160 --split > "$OUT/vae_wan_720p81f_split.json" 2> "$OUT/vae_wan.err"; cat "$OUT/vae_wan_720p81f_split.json" >> "$OUT/summary.txt"
# attention kernel generations, same box (tools/build_attn_v8_variant.sh must have been run before the call: the variant library travels with the snapshot)
if [ -f tools/probes/ab/attn_r3/libx2v_hip.so ] && [ "${ATTN_AB:-1}" = "1" ]; then
  for gen in 8 9; do
    X2V_LIB_PATH=$PWD/tools/probes/ab/attn_r3/libx2v_hip.so X2V_ATTN_GEN=$gen timeout 600 python bench.py --steps ${ATTN_AB_STEPS:-10} --warmup 3 --no-cpu-baseline > "$OUT/bench_attn_gen$gen.json" 2> "$OUT/bench_attn_gen$gen.err"
    echo "attention generation $gen (round-3 source, X2V_ATTN_GEN): $(python -c "import json; d=json.loads(open('$OUT/bench_attn_gen$gen.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'attn_ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'], 'frac_of_probe', d['roofline']['frac_of_probe'], 'probe', d['box_calibration']['mfma_probe_tflops'])")" | tee -a "$OUT/summary.txt"
  done
fi
# where are the attention launch's L2 misses served from?  fabric-side read counters of the TCC: requests, requests in flight (their ratio = mean read
# latency in L2 clocks), DRAM-destined requests and DRAM-credit stalls — plain grid (the launcher's choice at 40 heads) and the XCD-aware mapping forced
for map in 0 1; do
  i=0
  for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_32B_sum TCC_BUSY_sum" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    (cd /tmp && X2V_ATTN_MAP=$map timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/pmc_ea/map${map}_set$i" -o pmc -- "$GRAFT_REPO_ROOT/tools/x2v_check" pattn 12 75600 40 2 > "$GRAFT_REPO_ROOT/$OUT/pmc_ea_map${map}_set$i.log" 2>&1)
  done
  echo "== attention S=75600 H=40, X2V_ATTN_MAP=$map" >> "$OUT/pmc_ea_summary.txt"; python tools/pmc_summary.py "$OUT/pmc_ea/map${map}_set1" "$OUT/pmc_ea/map${map}_set2" "$OUT/pmc_ea/map${map}_set3" >> "$OUT/pmc_ea_summary.txt" 2>&1; tail -3 "$OUT/pmc_ea_map${map}_set1.log" >> "$OUT/pmc_ea_summary.txt"
done
find "$OUT/pmc_ea" -name "*kernel_trace.csv" -delete; cat "$OUT/pmc_ea_summary.txt" >> "$OUT/summary.txt"
if [ "${E2E14:-1}" = "1" ]; then
  run e2e_14b timeout 900 python tools/e2e.py --steps 50 > "$OUT/e2e_wan14b_720p.json" 2> "$OUT/e2e14.err"; cat "$OUT/e2e_wan14b_720p.json" >> "$OUT/summary.txt"
fi
cat "$OUT/summary.txt"
