"""Reads the softmax weights out of an attention launch: V = indicator rows (d = key // (Sk/128)), so o[q, d] = the probability mass of the keys
mapped to d.  Prints, per key tile and 16-key chunk, the worst relative deviation from torch — locates WHICH keys a broken kernel weighs wrongly.
    X2V_ATTN_PC=1 python tools/probes/pc_key_weights.py [Sk]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lightx2v_amd import lib

lib.init(0)
Sk = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Sq, H = 256, 1
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.randn(Sq, 128, generator=g, device="cuda").to(torch.bfloat16)
k = torch.randn(Sk, 128, generator=g, device="cuda").to(torch.bfloat16)
per = Sk // 128
v = torch.zeros(Sk, 128, device="cuda")
v[torch.arange(Sk), torch.arange(Sk) // per] = 1.0
v = v.to(torch.bfloat16)
vt = lib.transpose_heads(v, H)
o = lib.attention(q, k, None, H, variant=lib.ATTN_FAST, vt=vt).float()
p = torch.softmax(q.float() @ k.float().t() / 128 ** 0.5, dim=-1)
ref = p.view(Sq, 128, per).sum(-1)
rel = ((o - ref).abs() / ref.clamp_min(1e-6))
print("worst rel deviation per output column block of 8 (= keys", 8 * per, "per block):")
for b in range(16):
    print(f"  cols {8*b:3d}-{8*b+7:3d} keys {8*b*per:4d}-{(8*b+8)*per-1:4d}: max {rel[:, 8*b:8*b+8].max().item():.3e}  mean {rel[:, 8*b:8*b+8].mean().item():.3e}")
bad_rows = (rel.max(dim=1).values > 0.05).nonzero().flatten().tolist()
print("rows with > 5 % deviation:", len(bad_rows), bad_rows[:40])
