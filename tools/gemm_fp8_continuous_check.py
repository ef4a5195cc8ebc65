#!/usr/bin/env python
"""First contact of the continuous single-stream w8a8 GEMM (gemm256c8.hip, variant 5 of x2v_gemm_fp8_variant) against the ping-pong kernel
(gemm256.hip, variant 2): BIT-equality over ragged M, minimal and long K, all epilogues and scheduling-group sizes, then the timings of both on
the w8a8 step's projection shapes (a/b/a/b).  One JSON line per stage on stdout (equality first, so a timeout keeps it); run it under `timeout`."""
import json
import math
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


def operands(M, K, N, g):
    x = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
    xq, sx = lib.quant_fp8_rowwise(x)
    wq, sw = lib.quant_fp8_rowwise(w)
    return xq, sx, wq, sw


def equal_checks():
    g = torch.Generator(device="cuda").manual_seed(8)
    bad = []
    n = 0
    for M, K, N in ((300, 512, 256), (256, 768, 512), (4100, 2560, 5120), (9450, 5120, 5120), (1000, 13824, 256), (33000, 512, 1024), (257, 1024, 768)):
        xq, sx, wq, sw = operands(M, K, N, g)
        b = torch.randn(N, generator=g, device="cuda").to(torch.bfloat16)
        res = torch.randn(M, N, generator=g, device="cuda").to(torch.bfloat16)
        gate = (torch.randn(N, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
        for gm in (0, 7):
            for epi, kw in ((lib.EPI_NONE, {}), (lib.EPI_NONE, {"bias": None}), (lib.EPI_GELU_TANH, {}), (lib.EPI_SILU, {}), (lib.EPI_RESIDUAL, {"gate": gate}), (lib.EPI_RESIDUAL, {"gate": None})):
                bias = kw.get("bias", b)
                outs = []
                for form in (2, 5):
                    if epi == lib.EPI_RESIDUAL:
                        r = res.clone()
                        lib.gemm_fp8(xq, sx, wq, sw, bias, epilogue=epi, resid=r, gate=kw["gate"], variant=form | (gm << 8))
                        outs.append(r)
                    else:
                        y = torch.full((M + 2, N), 7.0, dtype=torch.bfloat16, device="cuda")  # rows around the output: nothing may be written past M
                        lib.gemm_fp8(xq, sx, wq, sw, bias, epilogue=epi, out=y[1 : M + 1], variant=form | (gm << 8))
                        outs.append(y)
                n += 1
                if not torch.equal(outs[0], outs[1]):
                    d = (outs[0].float() - outs[1].float()).abs()
                    bad.append({"M": M, "K": K, "N": N, "epi": epi, "gm": gm, "kw": sorted(kw), "mismatch_frac": (d > 0).float().mean().item(), "max": d.max().item(),
                                "nan": bool(torch.isnan(outs[1].float()).any().item())})
    return n, bad


def main():
    lib.init()
    n, bad = equal_checks()
    print(json.dumps({"equality_cases": n, "mismatches": bad[:12], "n_mismatches": len(bad)}), flush=True)
    if bad and os.environ.get("TIME_ANYWAY") != "1":
        return 1
    iters = int(os.environ.get("ITERS", "10"))
    g = torch.Generator(device="cuda").manual_seed(9)
    rows = []
    for M, K, N in ((75600, 5120, 5120), (75600, 5120, 13824), (75600, 13824, 5120), (75600, 5120, 15360), (9450, 5120, 5120)):
        xq, sx, wq, sw = operands(M, K, N, g)
        b = torch.randn(N, dtype=torch.bfloat16, device="cuda")
        y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        gate = torch.randn(N, dtype=torch.bfloat16, device="cuda")
        fl = 2.0 * M * N * K
        r = {"M": M, "K": K, "N": N}
        for epi, name in ((lib.EPI_NONE, "plain"), (lib.EPI_GELU_TANH, "gelu"), (lib.EPI_RESIDUAL, "resid")):
            for rep in range(2):
                for form, tag in ((2, "pingpong"), (5, "continuous")):
                    if epi == lib.EPI_RESIDUAL:
                        fn = lambda: lib.gemm_fp8(xq, sx, wq, sw, b, epilogue=epi, resid=y, gate=gate, variant=form)  # noqa: E731
                    else:
                        fn = lambda: lib.gemm_fp8(xq, sx, wq, sw, b, epilogue=epi, out=y, variant=form)  # noqa: E731
                    r[f"{name}_{tag}_TFLOPs_{rep}"] = round(fl / timed(fn, iters) / 1e9, 1)
        rows.append(r)
        print(json.dumps(r), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
