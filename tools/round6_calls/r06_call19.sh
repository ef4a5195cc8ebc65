#!/bin/bash
# Round 6, call 19: persistent short-walk attention form as shipped (q pieces issued by the early role, register stores at the block hand-over): checks + A/B.
set +e
OUT=gpurun_out/r06_call19
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=.
t0=$(date +%s)
timeout 300 tools/x2v_check attn > "$OUT/x2v_check_attn.log" 2>&1; echo "x2v_check attn rc=$?" | tee -a "$OUT/summary.txt"; tail -2 "$OUT/x2v_check_attn.log" >> "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_boundary.py tests/test_gpu_model.py tests/test_abi.py -m gpu -q --timeout 600 > "$OUT/pytest.log" 2>&1; echo "pytest ops/boundary/model/abi rc=$? ($(( $(date +%s) - t0 )) s)" | tee -a "$OUT/summary.txt"; tail -6 "$OUT/pytest.log" | cut -c1-300 >> "$OUT/summary.txt"
timeout 600 python tools/probes/cross_attn_ab.py 2>&1 | grep "cross attention" | cut -c1-330 | tee -a "$OUT/summary.txt"
echo "total $(( $(date +%s) - t0 )) s" >> "$OUT/summary.txt"
