#!/usr/bin/env python
"""First contact of x2v_gemm_fp8_blocked (the w8a8 GEMM on the Ulysses exchange buffers' layouts): N-blocked y, K-blocked x codes and K-blocked x
+ gate-residual against the row-major operator (variant 2) — BIT-equality, at a small shape (128x128 kernel), a mid shape and the 8-GPU rank shape
M = 9450.  Run once as is (blocked operands on the ping-pong kernel) and once with X2V_GEMM_FP8_CONTINUOUS=2 (on the continuous kernel).  One JSON line."""
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightx2v_amd import lib  # noqa: E402


def main():
    lib.init()
    g = torch.Generator(device="cuda").manual_seed(11)
    bad, n = [], 0
    for M, K, N, nb in ((300, 512, 256, 2), (4100, 2560, 5120, 4), (9450, 5120, 5120, 8), (9450, 5120, 13824, 8), (9450, 13824, 5120, 8)):
        x = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, generator=g, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
        xq, sx = lib.quant_fp8_rowwise(x)
        wq, sw = lib.quant_fp8_rowwise(w)
        b = torch.randn(N, generator=g, device="cuda").to(torch.bfloat16)
        res = torch.randn(M, N, generator=g, device="cuda").to(torch.bfloat16)
        gate = (torch.randn(N, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
        forced = 2 if lib.gemm_kernel_choice(M, N, K, fp8=True) == 2 else 1
        ref = lib.gemm_fp8(xq, sx, wq, sw, b, variant=forced)
        ref_g = lib.gemm_fp8(xq, sx, wq, sw, b, epilogue=lib.EPI_GELU_TANH, variant=forced)
        cases = []
        if N % (nb * 8) == 0:
            out = torch.full((nb, M + 3, N // nb), 7.0, dtype=torch.bfloat16, device="cuda")
            lib.gemm_fp8_blocked(xq, sx, wq, sw, b, out=out[:, 1 : M + 1])
            cases.append(("N-blocked y", out[:, 1 : M + 1].transpose(0, 1).reshape(M, N), ref))
            cases.append(("N-blocked y: rows around the blocks untouched", out[:, 0], torch.full_like(out[:, 0], 7.0)))
            out2 = torch.empty((nb, M, N // nb), dtype=torch.bfloat16, device="cuda")
            lib.gemm_fp8_blocked(xq, sx, wq, sw, b, epilogue=lib.EPI_GELU_TANH, out=out2)
            cases.append(("N-blocked y + gelu", out2.transpose(0, 1).reshape(M, N), ref_g))
        if K % (nb * 128) == 0:
            xb = xq.view(torch.uint8).view(M, nb, K // nb).transpose(0, 1).contiguous().view(torch.float8_e4m3fn)
            cases.append(("K-blocked x", lib.gemm_fp8_blocked(xb, sx, wq, sw, b), ref))
            r1, r2 = res.clone(), res.clone()
            lib.gemm_fp8_blocked(xb, sx, wq, sw, b, epilogue=lib.EPI_RESIDUAL, resid=r1, gate=gate)
            lib.gemm_fp8(xq, sx, wq, sw, b, epilogue=lib.EPI_RESIDUAL, resid=r2, gate=gate, variant=forced)
            cases.append(("K-blocked x + gate-residual", r1, r2))
        for name, a_, b_ in cases:
            n += 1
            if not torch.equal(a_, b_):
                d = (a_.float() - b_.float()).abs()
                bad.append({"M": M, "K": K, "N": N, "case": name, "mismatch_frac": (d > 0).float().mean().item(), "max": d.max().item()})
    print(json.dumps({"mode": os.environ.get("X2V_GEMM_FP8_CONTINUOUS", "default"), "cases": n, "n_mismatches": len(bad), "mismatches": bad[:10]}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
