#!/usr/bin/env python
"""ONE matrix kernel at one shape, a few launches, for rocprofv3 --pmc passes that compare the vendor library's kernel with ours counter by counter.
    python tools/gemm_pmc_one.py ours|blaslt M K N [launches]
`blaslt` = torch.addmm (hipBLASLt: Custom_Cijk_..._SK3_MT256x256x64_MI16x16x1 at these shapes), `ours` = lib.gemm (gemm256c).  Prints the HIP-event time too."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightx2v_amd import lib  # noqa: E402


def main():
    which, M, K, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    n = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    lib.init(0)
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device="cuda") / K**0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g, device="cuda").to(torch.bfloat16)
    y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    wt = w.t()
    fn = (lambda: torch.addmm(b, x, wt, out=y)) if which == "blaslt" else (lambda: lib.gemm(x, w, b, out=y))
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / n
    print(json.dumps({"which": which, "M": M, "K": K, "N": N, "launches": n, "ms": ms, "tflops": 2.0 * M * N * K / ms / 1e9}))


if __name__ == "__main__":
    main()
