#!/bin/bash
# Samples rocm-smi power / clocks while a kernel loop runs: evidence for "the matrix kernels run at the board power limit".
#   tools/power_probe.sh <tag> <x2v_check args...>
tag=$1; shift
out=gpurun_out/power; mkdir -p $out
( for i in $(seq 1 40); do rocm-smi --showpower --showclocks --showtemp --json 2>/dev/null | head -c 4000; echo; sleep 0.25; done ) > $out/$tag.smi.jsonl &
smi=$!
sleep 1
timeout 60 tools/x2v_check "$@" > $out/$tag.run.log 2>&1
wait $smi
python - "$out/$tag.smi.jsonl" "$tag" <<'PY'
import json, sys
pw, sclk = [], []
for line in open(sys.argv[1]):
    line = line.strip()
    if not line.startswith("{"):
        continue
    try:
        d = json.loads(line)
    except Exception:
        continue
    for card, v in d.items():
        for k, val in v.items():
            kl = k.lower()
            try:
                if "power" in kl and "(w)" in kl:
                    pw.append(float(val))
                if "sclk" in kl and "clock" in kl:
                    sclk.append(float(str(val).strip("()Mhz").lower().replace("mhz", "")))
            except Exception:
                pass
print(sys.argv[2], "power W: n=%d max=%.0f p50=%.0f" % (len(pw), max(pw or [0]), sorted(pw)[len(pw) // 2] if pw else 0),
      "| sclk MHz: n=%d min=%.0f p50=%.0f max=%.0f" % (len(sclk), min(sclk or [0]), sorted(sclk)[len(sclk) // 2] if sclk else 0, max(sclk or [0])))
PY
tail -1 $out/$tag.run.log
