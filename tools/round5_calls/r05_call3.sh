#!/bin/bash
# Round 5, call 3: the attention kernel without the compiler's VMEM drain in front of its barriers (bare s_barrier instead of __syncthreads) and with
# the late waves' first fragment reads in front of the even barrier — correctness first, then A-B-A-B against the old form and the two half-changes.
set +e
OUT=gpurun_out/r05_call3
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
timeout 120 tools/x2v_check attn > "$OUT/x2v_check_attn.log" 2>&1; echo "x2v_check attn rc=$? $(grep -c PASS "$OUT/x2v_check_attn.log") PASS $(grep -c FAIL "$OUT/x2v_check_attn.log") FAIL" | tee -a "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_full_size.py tests/test_gpu_rank_shapes.py tests/test_gpu_bench_shapes.py -m gpu -q --timeout 600 -x -k "attention or attn" > "$OUT/pytest_attn.log" 2>&1; echo "pytest attention rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a "$OUT/summary.txt"; tail -5 "$OUT/pytest_attn.log" | cut -c1-200 >> "$OUT/summary.txt"
for rep in 1 2 3; do
  for v in default a9old a9nofence_nopf a9fence_pf; do
    if [ "$v" = default ]; then L=lightx2v_amd; else L=tools/probes/ab/$v; fi
    echo "rep$rep $v: $(LD_LIBRARY_PATH=$L timeout 120 tools/x2v_check pattn 12 75600 40 12 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
  done
done
for v in default a9old; do   # the 8-GPU rank shape (5 heads, XCD-aware mapping) and the 1.3B shape
  if [ "$v" = default ]; then L=lightx2v_amd; else L=tools/probes/ab/$v; fi
  echo "H5 $v: $(LD_LIBRARY_PATH=$L timeout 120 tools/x2v_check pattn 12 75600 5 24 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
  echo "1.3B $v: $(LD_LIBRARY_PATH=$L timeout 120 tools/x2v_check pattn 12 20280 12 60 2>&1 | tail -1)" | tee -a "$OUT/summary.txt"
done
# whole step at sustained load, new vs old (6 timed steps each, a-b-a-b)
for rep in 1 2; do
  for v in default a9old; do
    if [ "$v" = default ]; then unset X2V_LIB_PATH; else export X2V_LIB_PATH=tools/probes/ab/$v/libx2v_hip.so; fi
    timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --probe-ms 500 > "$OUT/bench_${v}_$rep.json" 2> "$OUT/bench_${v}_$rep.err"
    echo "bench rep$rep $v: $(python -c "import json; d=json.loads([l for l in open('$OUT/bench_${v}_$rep.json') if l.startswith('{')][-1]); print('ms_per_step %.1f attn %.2f ms frac %.4f of_probe %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['frac_of_probe']))" 2>&1)" | tee -a "$OUT/summary.txt"
  done
done
unset X2V_LIB_PATH
echo "total $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
