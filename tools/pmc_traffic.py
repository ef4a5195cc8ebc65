"""roofline.traffic for bench.py: per-launch L2<->fabric bytes of the dominant kernel from rocprofv3 --pmc passes of the BENCH COMMAND ITSELF.

    cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <out>/fetch -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline
    cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE ... -d <out>/write ...
    cd /tmp && rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum ... -d <out>/tcc ...
    python tools/pmc_traffic.py <out> <kernel-name-substring> <tokens> <heads> <forwards_per_launch> > profiles/r03_pmc_attn_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB; FETCH is doubled (MI355X_MICROARCH.md "HBM": gfx950 tallies 128-byte requests at 64 B; WRITE_SIZE is
uncalibrated and tiny here).  The counter sits at the L2-fabric boundary and includes Infinity-Cache hits."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, pat, tokens, heads, fpl = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
acc = defaultdict(list)
name = None
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    per_dispatch = defaultdict(float)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if pat in row.get("Kernel_Name", ""):
                name = row["Kernel_Name"].split("(")[0]
                per_dispatch[(row["Counter_Name"], row.get("Dispatch_Id"))] += float(row["Counter_Value"])
    for (c, _), v in per_dispatch.items():
        acc[c].append(v)
mean = {c: sum(v) / len(v) for c, v in acc.items()}
fetch, write = mean.get("FETCH_SIZE", 0.0), mean.get("WRITE_SIZE", 0.0)
hit, miss = mean.get("TCC_HIT_sum", 0.0), mean.get("TCC_MISS_sum", 0.0)
out = {
    "kernel": name, "tokens": tokens, "heads": heads, "forwards_per_launch": fpl, "launches_profiled": {c: len(v) for c, v in acc.items()},
    "FETCH_SIZE_kib_per_launch": fetch, "WRITE_SIZE_kib_per_launch": write, "hbm_bytes_per_launch": fetch * 1024 * 2 + write * 1024,
    "l2_hit_rate": hit / (hit + miss) if hit + miss else None,
    "note": "rocprofv3 --pmc passes of `python bench.py --steps 1 --warmup 0 --no-cpu-baseline` itself (FETCH_SIZE, WRITE_SIZE, TCC in separate passes, kernel-trace only), "
            "mean over the launches of this instantiation; FETCH doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B). The counter sits at the "
            "L2-fabric boundary and includes Infinity-Cache hits: the plain grid keeps ~2 heads' K / V^T (77 MB) cache-resident, so most of this traffic never reaches HBM.",
}
print(json.dumps(out, indent=1))
