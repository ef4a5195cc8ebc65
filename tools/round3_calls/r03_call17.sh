#!/bin/bash
# CFG branches on two compute streams on ONE GPU: bit-identity test, then A/B at config #2 (1.3B 480p) and at the headline workload.
set +e
OUT=gpurun_out/r03_call17; mkdir -p $OUT
export TMPDIR=/tmp PYTHONPATH=.
timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -k "cfg_pair" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for rep in 1 2; do
  timeout 200 python bench.py --workload wan1.3b_480px49f --steps 6 --warmup 2 --no-cpu-baseline > $OUT/b13_seq_$rep.json 2> $OUT/b13_seq_$rep.err
  timeout 200 python bench.py --workload wan1.3b_480px49f --steps 6 --warmup 2 --no-cpu-baseline --cfg-streams > $OUT/b13_streams_$rep.json 2> $OUT/b13_streams_$rep.err
  timeout 200 python bench.py --workload wan1.3b_480px49f --steps 6 --warmup 2 --no-cpu-baseline --cfg-pair > $OUT/b13_pair_$rep.json 2> $OUT/b13_pair_$rep.err
done
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/b14_pair.json 2> $OUT/b14_pair.err
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --cfg-streams > $OUT/b14_streams.json 2> $OUT/b14_streams.err
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_call17/b1*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "%.2f ms/step"%d["ms_per_step"], d["config"]["cfg_form"][:40], "attn avg %.3f ms"%d["roofline"]["avg_launch_ms"])
    except Exception as e: print(f, "ERR", e)
P
