#!/bin/bash
# Round 4, call 5: the HunyuanVideo-13B full-size forward test (first run: timings + measured errors), and the list of memory-side counters rocprofv3 offers here.
set +e
OUT=gpurun_out/r04_call5
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_full_size.py -m gpu -q --timeout 900 --durations=5 -k "hunyuan13b_720p_129f_full_forward" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a "$OUT/summary.txt"
tail -30 "$OUT/pytest.log" | cut -c1-400 >> "$OUT/summary.txt"
grep -i hunyuan gpurun_out/parity_summary.jsonl | tail -2 >> "$OUT/summary.txt"
(cd /tmp && timeout 120 rocprofv3 -L > "$GRAFT_REPO_ROOT/$OUT/counters.txt" 2>&1)
grep -i -o "name: *[A-Za-z0-9_]*\(HBM\|MALL\|DRAM\|UMC\|EA_RD\|EA_WR\|EA0_RD\|EA0_WR\|TCC_REQ\|TCC_READ\|TCC_WRITE\|TCP_TCC\)[A-Za-z0-9_]*" "$OUT/counters.txt" | sort -u | head -80 >> "$OUT/summary.txt"
wc -l "$OUT/counters.txt" >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
