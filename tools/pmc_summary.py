"""Summarise rocprofv3 --pmc CSV output (counter_collection.csv): per kernel, mean counter value per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
files = [f for root in sys.argv[1:] for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True))]
for f in files:
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "?")
            if "fill" in name or "memset" in name.lower():
                continue
            short = name.replace("(anonymous namespace)::", "").split("(")[0][-56:]
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for kern, ctrs in acc.items():
    print(f"== {kern}")
    for c, vals in sorted(ctrs.items()):
        # one row per (dispatch, counter[, dimension]); sum dimensions per dispatch is not recoverable here -> report mean and n
        print(f"   {c:34s} mean={sum(vals) / len(vals):.6g}  n={len(vals)}")
    if "TCC_EA0_RDREQ_sum" in ctrs and "TCC_EA0_RDREQ_LEVEL_sum" in ctrs:  # mean fabric-side read latency in L2 clocks (TCC_EA0_RDREQ_LEVEL's own description)
        rd, lvl = sum(ctrs["TCC_EA0_RDREQ_sum"]) / len(ctrs["TCC_EA0_RDREQ_sum"]), sum(ctrs["TCC_EA0_RDREQ_LEVEL_sum"]) / len(ctrs["TCC_EA0_RDREQ_LEVEL_sum"])
        print(f"   -> mean EA read latency = LEVEL / RDREQ = {lvl / max(rd, 1):.1f} L2 clocks")
