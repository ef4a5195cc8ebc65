#!/bin/bash
# Calibration of the by-size CFG rules on two sizes between the BASELINE configs: pair pass vs two compute streams vs one forward after the other.
set +e
OUT=gpurun_out/r03_call18; mkdir -p $OUT
export TMPDIR=/tmp PYTHONPATH=.
for wl in wan14b_480px81f wan1.3b_720px81f; do
  for v in "--cfg-pair" "--cfg-streams" "--no-cfg-pair --no-cfg-streams"; do
    tag=$(echo "$v" | tr -d ' -' ); 
    timeout 300 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline $v > $OUT/${wl}_${tag}.json 2> $OUT/${wl}_${tag}.err
  done
done
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_call18/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print("%-52s %9.2f ms/step  %s"%(f.split("/")[-1], d["ms_per_step"], d["config"]["cfg_form"][:28]))
    except Exception as e: print(f, "ERR", e)
P
