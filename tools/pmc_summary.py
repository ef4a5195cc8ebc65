"""Summarise rocprofv3 --pmc CSV output (counter_collection.csv): per kernel, mean counter value per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "?")
            if "fill" in name or "memset" in name.lower():
                continue
            short = name.split("(")[0][-48:]
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for kern, ctrs in acc.items():
    print(f"== {kern}")
    for c, vals in sorted(ctrs.items()):
        # one row per (dispatch, counter[, dimension]); sum dimensions per dispatch is not recoverable here -> report mean and n
        print(f"   {c:34s} mean={sum(vals) / len(vals):.6g}  n={len(vals)}")
