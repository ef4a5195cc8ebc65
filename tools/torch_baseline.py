#!/usr/bin/env python
"""Second baseline: what the reference's 'Default' ops reach on the same GPU through ROCm PyTorch — torch.addmm
(hipBLASLt) and F.scaled_dot_product_attention — at the Wan shapes, beside this repository's kernels.  One JSON line."""
import json
import os
import sys

import torch
import torch.nn.functional as F

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
    out = {"gemm": [], "attention": []}
    for M, K, N in ((75600, 5120, 5120), (75600, 5120, 13824), (75600, 13824, 5120), (20280, 1536, 8960), (20280, 1536, 1536)):
        x = torch.randn(M, K, dtype=torch.bfloat16, device="cuda")
        w = torch.randn(N, K, dtype=torch.bfloat16, device="cuda") / K**0.5
        b = torch.randn(N, dtype=torch.bfloat16, device="cuda")
        y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        wt = w.t()
        ms_t = timed(lambda: torch.addmm(b, x, wt, out=y))  # mm_weight.py:81-88
        ms_x = timed(lambda: lib.gemm(x, w, b, out=y))
        fl = 2.0 * M * N * K
        out["gemm"].append({"M": M, "K": K, "N": N, "torch_addmm_ms": ms_t, "torch_TFLOP/s": fl / ms_t / 1e9, "x2v_ms": ms_x, "x2v_TFLOP/s": fl / ms_x / 1e9})
    for S, H in ((20280, 12), (75600, 5)):
        q, k, v = (torch.randn(S, H, 128, dtype=torch.bfloat16, device="cuda") for _ in range(3))
        fl = 4.0 * S * S * H * 128
        ms_x = timed(lambda: lib.attention(q, k, v, H, variant=lib.ATTN_FAST), iters=3)
        rec = {"S": S, "H": H, "x2v_ms": ms_x, "x2v_TFLOP/s": fl / ms_x / 1e9}
        try:
            qq, kk, vv = (t.transpose(0, 1).unsqueeze(0) for t in (q, k, v))  # attn_weight.py:209-239 (torch_sdpa)
            ms_t = timed(lambda: F.scaled_dot_product_attention(qq, kk, vv), iters=2)
            rec.update({"torch_sdpa_ms": ms_t, "torch_TFLOP/s": fl / ms_t / 1e9})
        except Exception as e:  # noqa: BLE001
            rec["torch_sdpa_error"] = str(e)[:200]
        out["attention"].append(rec)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
