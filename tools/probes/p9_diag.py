"""Where the persistent short-walk form differs from the one-walk form (debug aid): mismatch counts by row-in-block, column, block."""
import sys

import torch

sys.path.insert(0, "/root/repo")
from lightx2v_amd import lib  # noqa: E402

lib.init()
Sq, Sk, H = 3365, 512, 40
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randn(Sq, H * 128, generator=g, device="cuda").to(torch.bfloat16)
k = torch.randn(Sk, H * 128, generator=g, device="cuda").to(torch.bfloat16)
v = torch.randn(Sk, H * 128, generator=g, device="cuda").to(torch.bfloat16)
vt = lib.transpose_heads(v, H)
a = lib.attention(q, k, None, H, variant=lib.ATTN_FAST, vt=vt)
b = lib.attention(q, k, None, H, variant=lib.ATTN_FAST | lib.ATTN_ONE_WALK, vt=vt)
bad = (a != b).view(Sq, H, 128)
print("mismatching elements", bad.sum().item(), "of", bad.numel())
rows = bad.any(2)  # [Sq, H]
nqb = (Sq + 255) // 256
by_pass = {}
for h in range(H):
    for blk in range(nqb):
        item = h * nqb + blk
        r = rows[blk * 256 : (blk + 1) * 256, h]
        n = r.numel()
        d = by_pass.setdefault(item // 256, [0, 0, 0, 0])
        d[0] += r.view(-1)[: n].sum().item()
        d[1] += n
        rr = r.nonzero().flatten()
        d[2] += ((rr % 32) < 16).sum().item()
        d[3] += ((rr % 32) >= 16).sum().item()
print("bad rows by pass (bad, rows, bad in g=0, bad in g=1):", by_pass)
for h in (0,):
    for blk in range(nqb):
        r = rows[blk * 256 : (blk + 1) * 256, h]
        item = h * nqb + blk
        idx = r.nonzero().flatten().tolist()
        print(f"head {h} block {blk} item {item} (wg {item % 256}, pass {item // 256}): bad rows {len(idx)} {idx[:6]}..{idx[-3:]}")
cols = bad.any(0).any(0).nonzero().flatten().tolist()
print("bad cols", len(cols), cols[:20])
A, B = a.view(Sq, H, 128).float(), b.view(Sq, H, 128).float()
for r, h in ((16, 0), (17, 0), (48, 0), (144, 0), (16 + 256, 0), (3, 26 // 14)):
    d = (A[r, h] - B[r, h])
    print(f"row {r} head {h}: max |d| {d.abs().max().item():.4g}  rel L2 {(d.norm() / B[r, h].norm()).item():.3g}  first values persistent {A[r, h, :6].tolist()} one-walk {B[r, h, :6].tolist()}")
    blk = r // 256
    cand = B[blk * 256 : (blk + 1) * 256]  # [256, H, 128]
    hit = (cand == A[r, h]).all(2).nonzero().tolist()
    print("   equals one-walk (row, head) of the block:", [(blk * 256 + x, y) for x, y in hit][:8])
    # per dv tile T (16 columns): is the error confined to some tiles?
    print("   max |d| per 16-column tile:", [round(d[16 * T : 16 * T + 16].abs().max().item(), 4) for T in range(8)])
    nb = (A[r, h] != B[r, h]).nonzero().flatten().tolist()
    print("   differing columns:", nb[:40])
# is a wrong g=1 row of a pass-0 block the (correct) row of the workgroup's NEXT block, or computed with the next block's q against this head's keys?
for r, h in ((16, 0), (17, 0), (48, 0), (144, 0)):
    item = h * nqb + r // 256
    nxt = item + 256
    h2, b2 = nxt // nqb, nxt % nqb
    cand = B[b2 * 256 : (b2 + 1) * 256, h2]
    hit = (cand == A[r, h]).all(1).nonzero().flatten().tolist()
    near = (cand - A[r, h]).abs().max(1).values
    print(f"row {r} head {h}: next item {nxt} = head {h2} block {b2}; equals its rows {hit}; closest row {near.argmin().item()} max|d| {near.min().item():.4g}")
    # q of the next block's row against THIS head's keys
    qn = q.view(Sq, H, 128)[b2 * 256 + (r % 256), h2].float()
    kk, vv = k.view(Sk, H, 128)[:, h].float(), v.view(Sk, H, 128)[:, h].float()
    o = torch.softmax((kk @ qn) / 128 ** 0.5, 0) @ vv
    print(f"    attention of the NEXT block's q row against this head's K/V: max|d| vs persistent {(o - A[r, h]).abs().max().item():.4g}")
    kk2, vv2 = k.view(Sk, H, 128)[:, h2].float(), v.view(Sk, H, 128)[:, h2].float()
    qc = q.view(Sq, H, 128)[r, h].float()
    o2 = torch.softmax((kk2 @ qc) / 128 ** 0.5, 0) @ vv2
    print(f"    attention of THIS q row against the NEXT block's head K/V: max|d| vs persistent {(o2 - A[r, h]).abs().max().item():.4g}")
# chunk routing: where (row, 16-byte chunk) of the one-walk result does each chunk of a wrong row come from?
print("---- chunk routing")
for r, h in ((16, 0), (21, 0), (48, 0), (240, 0)):
    blk = r // 256
    cand = B[blk * 256 : (blk + 1) * 256, h].reshape(256, 16, 8)  # [row, chunk, 8]
    out = []
    for c in range(16):
        mine = A[r, h, 8 * c : 8 * c + 8]
        hit = (cand == mine).all(2).nonzero().tolist()
        out.append(hit[:2])
    print(f"row {r} head {h}: chunk c comes from (row, chunk):", out)
