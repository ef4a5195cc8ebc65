#!/usr/bin/env python
"""Config-#2 accounting: from a rocprofv3 --kernel-trace CSV of a SEQUENTIAL-form bench run (one compute stream), per kernel the
calls, mean duration, share of the busy time and — for the matrix kernels whose algorithmic FLOPs the caller names — TFLOP/s and fraction of the
2.5 PFLOP/s bf16 peak; plus the idle share: wall time from the first to the last kernel of the timed steps minus the union of the kernel intervals
(what a hipGraph capture of the block could win at most).  usage: kernel_gaps.py <kernel_trace.csv> <tokens S> <dim D> <ffn F> <heads H> <skip_first_n_kernels_frac>"""
import csv
import json
import sys
from collections import defaultdict

path, S, D, F, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
skip_frac = float(sys.argv[6]) if len(sys.argv) > 6 else 0.34  # the warm-up step of a --warmup 1 --steps 2 run
rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[int(len(rows) * skip_frac):]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy, cur_s, cur_e = 0, None, None
for s, e, _ in rows:  # union of intervals
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
gaps = sorted((rows[i + 1][0] - max(r[1] for r in rows[max(0, i - 3):i + 1]) for i in range(len(rows) - 1)), reverse=True)
per = defaultdict(list)
for s, e, n in rows:
    per[n.split("(")[0].replace("void ", "").replace("x2v::", "").replace("(anonymous namespace)::", "")[:60]].append(e - s)
M = S  # rows per launch in the sequential form (one forward per launch)
flops = {  # algorithmic FLOP per launch of the matrix kernels at this model's shapes, by (kernel substring, mean-duration rank is not needed: one shape each)
    "attn_fwd_v9_kernel<8, 8, true": 4.0 * S * S * H * 128,
    "attn_fwd_v9_kernel<8, 8, false": 4.0 * S * 512 * H * 128,
}
out = {"trace": path, "kernels_in_window": len(rows), "wall_ms": (t1 - t0) / 1e6, "busy_ms": busy / 1e6, "idle_share": 1.0 - busy / (t1 - t0),
       "largest_gaps_us": [g / 1e3 for g in gaps[:5]], "median_gap_us": gaps[len(gaps) // 2] / 1e3, "kernels": []}
tot = sum(sum(v) for v in per.values())
for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    rec = {"kernel": n, "calls": len(v), "mean_us": sum(v) / len(v) / 1e3, "share_of_kernel_time": sum(v) / tot}
    for key, fl in flops.items():
        if key in n:
            rec["tflops"] = fl / (sum(v) / len(v)) / 1e3
            rec["frac_of_2.5PF"] = rec["tflops"] / 2500.0
    if "gemm256" in n or "gemm_kernel" in n:
        rec["note"] = "mixed shapes: see gemm_by_shape"
    out["kernels"].append(rec)
# GEMM launches by shape: D->D (q,k,v,o,cross q,o), D->F (+GELU), F->D; classify each launch of a gemm kernel by its duration relative to 2 M N K / 1.2 PF
gem = [(e - s, n) for s, e, n in rows if "gemm256" in n]
shapes = {"D->D": 2.0 * M * D * D, "D->F": 2.0 * M * D * F, "F->D": 2.0 * M * F * D}
byshape = defaultdict(list)
resid = [d for d, n in gem if "<" in n and n.split("<")[1].startswith("2")]
resid_cut = 2.5 * min(resid) if resid else 0  # ffn2 (F->D) runs ~F/D times longer than o / cross-o (D->D)
for d, n in gem:
    epi = n.split("<")[1].split(">")[0] if "<" in n else "?"
    # epilogue 1 = GELU (only ffn0: D->F); 2 = gated residual (o, cross o: D->D; ffn2: F->D); 0 = plain (q, k, cross q: D->D; v->V^T is gemm256s<0, true>)
    if epi.startswith("1"):
        byshape["D->F +GELU"].append(d)
    elif epi.startswith("2"):
        byshape["F->D +residual" if d > resid_cut else "D->D +residual"].append(d)
    else:
        byshape["D->D plain / V^T"].append(d)
fl_of = {"D->F +GELU": shapes["D->F"], "F->D +residual": shapes["F->D"], "D->D +residual": shapes["D->D"], "D->D plain / V^T": shapes["D->D"]}
out["gemm_by_shape"] = {k: {"calls": len(v), "mean_us": sum(v) / len(v) / 1e3, "tflops": fl_of[k] / (sum(v) / len(v)) / 1e3, "frac_of_2.5PF": fl_of[k] / (sum(v) / len(v)) / 1e3 / 2500.0}
                        for k, v in byshape.items() if v}
print(json.dumps(out, indent=1))
