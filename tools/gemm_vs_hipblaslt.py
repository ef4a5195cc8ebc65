#!/usr/bin/env python
"""GEMM-only comparison at the 14B / 1.3B denoise shapes: torch.addmm (hipBLASLt, what the reference's Default MMWeight runs,
common/ops/mm/mm_weight.py:81-88) beside lib.gemm, 20 back-to-back launches each (long enough for the power governor to settle).
Run under `rocprofv3 --kernel-trace --stats` to get the library's kernel names.  One JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightx2v_amd import lib  # noqa: E402


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    lib.init()
    iters = int(os.environ.get("ITERS", "20"))
    rows = []
    shapes = ((75600, 5120, 5120), (75600, 5120, 15360), (75600, 5120, 13824), (75600, 13824, 5120), (20280, 1536, 1536), (20280, 1536, 8960), (20280, 8960, 1536))
    for M, K, N in shapes:
        x = torch.randn(M, K, dtype=torch.bfloat16, device="cuda")
        w = torch.randn(N, K, dtype=torch.bfloat16, device="cuda") / K**0.5
        b = torch.randn(N, dtype=torch.bfloat16, device="cuda")
        y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        wt = w.t()
        fl = 2.0 * M * N * K
        r = {"M": M, "K": K, "N": N, "x2v_kernel": lib.gemm_kernel_choice(M, N, K)}
        for rep in range(2):  # a/b/a/b: order effects of the power governor show as a spread
            ms_t = timed(lambda: torch.addmm(b, x, wt, out=y), iters)
            ms_x = timed(lambda: lib.gemm(x, w, b, out=y), iters)
            r[f"hipblaslt_TFLOPs_{rep}"] = fl / ms_t / 1e9
            r[f"x2v_TFLOPs_{rep}"] = fl / ms_x / 1e9
        rows.append(r)
    print(json.dumps({"iters": iters, "gemm": rows}))


if __name__ == "__main__":
    main()
