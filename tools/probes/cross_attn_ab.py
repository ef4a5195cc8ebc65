import torch, sys
sys.path.insert(0, "/root/repo")
from lightx2v_amd import lib
lib.init()
S, Sk, H = 75600, 512, 40
q = torch.randn(S, H*128, dtype=torch.bfloat16, device="cuda")
k = torch.randn(Sk, H*128, dtype=torch.bfloat16, device="cuda")
v = torch.randn(Sk, H*128, dtype=torch.bfloat16, device="cuda")
o = torch.empty_like(q)
vt = lib.transpose_heads(v, H)
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for rep in range(2):
    m0 = t(lambda: lib.attention(q, k, v, H, out=o, variant=0))
    o0 = o.clone()
    m1 = t(lambda: lib.attention(q, k, v, H, out=o, variant=lib.ATTN_FAST, vt=vt))
    d = (o.float() - o0.float()).abs().max().item()
    fl = 4.0 * S * Sk * H * 128
    print(f"cross attention S={S} Sk={Sk} H={H}: pipe kernel {m0:.3f} ms {fl/m0/1e9:.0f} TF | ping-pong on V^T {m1:.3f} ms {fl/m1/1e9:.0f} TF | max |d| {d:.3e}")
