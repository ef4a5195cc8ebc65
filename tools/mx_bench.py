#!/usr/bin/env python
"""Timing of the MXFP8 quantiser (HBM-bound) and block-scaled GEMM (MFMA-bound) at the Wan2.1-14B 720p shapes; one JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightx2v_amd import lib  # noqa: E402


def timed(fn, iters=5):
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
    out = {"quant": [], "gemm": []}
    for M, K in ((75600, 5120), (75600, 13824), (20280, 1536)):
        x = torch.randn(M, K, dtype=torch.bfloat16, device="cuda")
        ms = timed(lambda: lib.quant_mxfp8(x))
        byt = M * K * (2 + 1 + 1 / 32)
        out["quant"].append({"M": M, "K": K, "ms": ms, "GB/s": byt / ms / 1e6, "frac_of_6290": byt / ms / 1e6 / 6290})
    for M, K, N in ((75600, 5120, 5120), (75600, 5120, 13824), (75600, 13824, 5120), (20280, 1536, 8960)):
        a, sa = lib.quant_mxfp8(torch.randn(M, K, dtype=torch.bfloat16, device="cuda"))
        w, sw = lib.quant_mxfp8(torch.randn(N, K, dtype=torch.bfloat16, device="cuda"))
        y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        ms = timed(lambda: lib.gemm_mxfp8(a, sa, w, sw, out=y))
        ms128 = timed(lambda: lib.gemm_mxfp8(a, sa, w, sw, out=y, variant=1))
        tf = 2.0 * M * N * K / ms / 1e9
        xq, sx = lib.quant_fp8_rowwise(torch.randn(M, K, dtype=torch.bfloat16, device="cuda"))
        swc = torch.ones(N, 1, dtype=torch.float32, device="cuda")
        ms8 = timed(lambda: lib.gemm_fp8(xq, sx, w, swc, out=y))
        out["gemm"].append({"M": M, "K": K, "N": N, "ms": ms, "TFLOP/s": tf, "frac_of_5000": tf / 5000, "tile128_ms": ms128, "per_channel_fp8_256x256_ms": ms8,
                            "per_channel_fp8_TFLOP/s": 2.0 * M * N * K / ms8 / 1e9})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
