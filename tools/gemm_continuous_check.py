#!/usr/bin/env python
"""The continuous-pipeline form of the single-stream 256x256 GEMM (gemm256c.hip, variant 5) against its one-output-tile-per-workgroup form
(gemm256s.hip, variant 4): BIT-equality over ragged M, minimal and long K, all epilogues, blocked operands and scheduling-group sizes, then the
timings of both forms on the benchmark's projection shapes (20 back-to-back launches each, a/b/a/b).  One JSON line; run it under `timeout`."""
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


def equal_checks():
    g = torch.Generator(device="cuda").manual_seed(5)
    bad = []
    n = 0
    for M, K, N in ((300, 256, 256), (256, 384, 512), (4100, 2560, 5120), (20280, 1536, 1536), (9450, 5120, 5120), (1000, 13824, 256), (66000, 256, 1024), (257, 512, 768)):
        x = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, generator=g, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
        b = torch.randn(N, generator=g, device="cuda").to(torch.bfloat16)
        res = torch.randn(M, N, generator=g, device="cuda").to(torch.bfloat16)
        gate = (torch.randn(N, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
        for gm in (0, 1, 7):
            for epi, kw in ((lib.EPI_NONE, {}), (lib.EPI_NONE, {"bias": None}), (lib.EPI_GELU_TANH, {}), (lib.EPI_SILU, {}), (lib.EPI_RESIDUAL, {"gate": gate}), (lib.EPI_RESIDUAL, {"gate": None})):
                bias = kw.get("bias", b)
                outs = []
                for form in (4, 5):
                    if epi == lib.EPI_RESIDUAL:
                        r = res.clone()
                        lib.gemm(x, w, bias, epilogue=epi, resid=r, gate=kw["gate"], variant=form | (gm << 8))
                        outs.append(r)
                    else:
                        y = torch.full((M + 2, N), 7.0, dtype=torch.bfloat16, device="cuda")  # rows around the output: nothing may be written past M
                        lib.gemm(x, w, bias, epilogue=epi, out=y[1 : M + 1], variant=form | (gm << 8))
                        outs.append(y)
                n += 1
                if not torch.equal(outs[0], outs[1]):
                    d = (outs[0].float() - outs[1].float()).abs()
                    bad.append({"M": M, "K": K, "N": N, "epi": epi, "gm": gm, "kw": sorted(kw), "mismatch_frac": (d > 0).float().mean().item(), "max": d.max().item()})
        # blocked operands go through the dispatcher's default form (continuous where it applies): against form 4 on the row-major tensors
        nb = 2
        if K % (nb * 64) == 0 and N % (nb * 128) == 0:
            ref = lib.gemm(x, w, b, variant=4)
            out = torch.full((nb, M + 3, N // nb), 7.0, dtype=torch.bfloat16, device="cuda")[:, 1 : M + 1]
            lib.gemm(x, w, b, out=out)
            xb = x.view(M, nb, K // nb).transpose(0, 1).contiguous()
            r1, r2 = res.clone(), res.clone()
            lib.gemm(xb, w, b, epilogue=lib.EPI_RESIDUAL, resid=r1, gate=gate)
            lib.gemm(x, w, b, epilogue=lib.EPI_RESIDUAL, resid=r2, gate=gate, variant=4)
            n += 3
            for name, a_, b_ in (("N-blocked y", out.transpose(0, 1).reshape(M, N), ref), ("K-blocked x", lib.gemm(xb, w, b), ref), ("K-blocked x + residual", r1, r2)):
                if not torch.equal(a_, b_):
                    bad.append({"M": M, "K": K, "N": N, "blocked": name})
    return n, bad


def main():
    lib.init()
    n, bad = equal_checks()
    out = {"equality_cases": n, "mismatches": bad[:12], "n_mismatches": len(bad)}
    if bad and os.environ.get("TIME_ANYWAY") != "1":
        print(json.dumps(out))
        return 1
    iters = int(os.environ.get("ITERS", "20"))
    rows = []
    shapes = ((75600, 5120, 5120), (151200, 5120, 5120), (151200, 5120, 13824), (151200, 13824, 5120), (20280, 1536, 1536), (20280, 1536, 8960), (20280, 8960, 1536), (9450, 5120, 5120))
    for M, K, N in shapes:
        x = torch.randn(M, K, dtype=torch.bfloat16, device="cuda")
        w = torch.randn(N, K, dtype=torch.bfloat16, device="cuda") / K**0.5
        b = torch.randn(N, dtype=torch.bfloat16, device="cuda")
        y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        gate = torch.randn(N, dtype=torch.bfloat16, device="cuda")
        fl = 2.0 * M * N * K
        r = {"M": M, "K": K, "N": N}
        for epi, name in ((lib.EPI_NONE, "plain"), (lib.EPI_GELU_TANH, "gelu"), (lib.EPI_RESIDUAL, "resid")):
            for rep in range(2):
                for form, tag in ((4, "one_tile"), (5, "continuous")):
                    if epi == lib.EPI_RESIDUAL:
                        fn = lambda: lib.gemm(x, w, b, epilogue=epi, resid=y, gate=gate, variant=form)  # noqa: E731
                    else:
                        fn = lambda: lib.gemm(x, w, b, epilogue=epi, out=y, variant=form)  # noqa: E731
                    r[f"{name}_{tag}_TFLOPs_{rep}"] = round(fl / timed(fn, iters) / 1e9, 1)
        rows.append(r)
        print(json.dumps(r), file=sys.stderr, flush=True)
    out["timing"] = rows
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
