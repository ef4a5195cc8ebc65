#!/usr/bin/env python
"""The three matrix kernels of the benched step, a few launches each at the step's shapes, for rocprofv3 --pmc passes (MFMA-busy
counters existed only for round 1-3 kernels): gemm256c (bf16 continuous, 151200 x 5120 -> 5120 and -> 13824 + GELU), gemm256c8 (w8a8 continuous,
75600 x 5120 -> 5120), attn_fwd_v9 (the paired CFG launch, 2 x 40 heads x 75600).  Prints each kernel's HIP-event time for the same launches.
    cd /tmp && rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d OUT -o pmc -- python tools/pmc_kernel_loop.py"""
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightx2v_amd import lib  # noqa: E402


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    lib.init(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    g = torch.Generator(device="cuda").manual_seed(0)
    out = {}
    M, D, F = 151200, 5120, 13824
    x = torch.randn(M, D, generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn(D, D, generator=g, device="cuda") / math.sqrt(D)).to(torch.bfloat16)
    w0 = (torch.randn(F, D, generator=g, device="cuda") / math.sqrt(D)).to(torch.bfloat16)
    b, b0 = torch.zeros(D, dtype=torch.bfloat16, device="cuda"), torch.zeros(F, dtype=torch.bfloat16, device="cuda")
    y, h = torch.empty(M, D, dtype=torch.bfloat16, device="cuda"), torch.empty(M, F, dtype=torch.bfloat16, device="cuda")
    ms = timed(lambda: lib.gemm(x, w, b, out=y), n)
    out["gemm256c<0> 151200x5120->5120"] = {"ms": ms, "tflops": 2.0 * M * D * D / ms / 1e9}
    ms = timed(lambda: lib.gemm(x, w0, b0, epilogue=lib.EPI_GELU_TANH, out=h), n)
    out["gemm256c<1> 151200x5120->13824 +GELU"] = {"ms": ms, "tflops": 2.0 * M * D * F / ms / 1e9}
    del h, w0
    M8 = 75600
    xq, sx = lib.quant_fp8_rowwise(x[:M8])
    wq, sw = lib.quant_fp8_rowwise(w)
    ms = timed(lambda: lib.gemm_fp8(xq, sx, wq, sw, b, out=y[:M8]), n)
    out["gemm256c8<0> 75600x5120->5120 (w8a8)"] = {"ms": ms, "tflops": 2.0 * M8 * D * D / ms / 1e9}
    S, H, Sp = 75600, 40, 75648
    q = torch.randn(2 * Sp, H * 128, generator=g, device="cuda").to(torch.bfloat16)
    k = torch.randn(2 * Sp, H * 128, generator=g, device="cuda").to(torch.bfloat16)
    v = torch.randn(2 * Sp, H * 128, generator=g, device="cuda").to(torch.bfloat16)
    vt = lib.transpose_heads(v, H)
    del v
    ms = timed(lambda: lib.attention_batched(q, k, vt, H, 2, Sp, S, prescaled=True, stagger=False), max(1, n // 2))
    out["attn_fwd_v9<8,8,true,false> 2 x 40 x 75600 (paired)"] = {"ms": ms, "tflops": 2 * 4.0 * S * S * H * 128 / ms / 1e9}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
