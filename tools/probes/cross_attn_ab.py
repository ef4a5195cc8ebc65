"""Cross-attention launch forms timed on one box, interleaved: the row-major-V pipeline kernel (x2v_attn_fwd_bf16), the ping-pong kernel as one-walk
workgroups (X2V_ATTN_VT_ONE_WALK) and as the persistent short-walk launch (the default for 3..32 whole key tiles).  Shapes: Wan-14B 720p text context
(75 600 x 512 x 40), the stacked CFG pair's slot (75 648 rows), Wan-1.3B 480p (20 280 x 512 x 12), an 8-GPU Ulysses rank (9450 x 512 x 40)."""
import sys

import torch

sys.path.insert(0, "/root/repo")
from lightx2v_amd import lib  # noqa: E402

lib.init()


def t(fn, n=40):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


import os  # noqa: E402

SHAPES = ((75600, 512, 40), (75648, 512, 40), (20280, 512, 12), (9450, 512, 40), (75600, 256, 40), (75600, 1024, 40))
if os.environ.get("X2V_AB_SHAPES"):
    SHAPES = tuple(tuple(int(x) for x in sh.split(",")) for sh in os.environ["X2V_AB_SHAPES"].split(";"))
for S, Sk, H in SHAPES:
    q = torch.randn(S, H * 128, dtype=torch.bfloat16, device="cuda")
    k = torch.randn(Sk, H * 128, dtype=torch.bfloat16, device="cuda")
    v = torch.randn(Sk, H * 128, dtype=torch.bfloat16, device="cuda")
    o = torch.empty_like(q)
    vt = lib.transpose_heads(v, H)
    fl = 4.0 * S * Sk * H * 128
    for rep in range(1 if os.environ.get("X2V_AB_SHAPES") else 2):
        m0 = t(lambda: lib.attention(q, k, v, H, out=o, variant=0))
        m1 = t(lambda: lib.attention(q, k, None, H, out=o, variant=lib.ATTN_FAST | lib.ATTN_ONE_WALK, vt=vt))
        o1 = o.clone()
        m2 = t(lambda: lib.attention(q, k, None, H, out=o, variant=lib.ATTN_FAST, vt=vt))
        same = torch.equal(o, o1)
        print(f"cross attention S={S} Sk={Sk} H={H} plan={lib.attn_vt_launch_plan(S, Sk, H, with_short=True)}: pipeline {m0:.3f} ms {fl / m0 / 1e9:.0f} TF | one-walk {m1:.3f} ms {fl / m1 / 1e9:.0f} TF | "
              f"persistent {m2:.3f} ms {fl / m2 / 1e9:.0f} TF ({fl / m2 / 1e9 / 2500:.3f} of peak) | {m1 / m2:.3f}x | bit-equal {same}", flush=True)
