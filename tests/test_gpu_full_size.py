"""Parity AT THE SIZES THE HEADLINE BENCHMARK RUNS (VERDICT r2 "what's missing" #1): the attention launch, the projections and one whole
block at S = 75 600 tokens (Wan2.1-14B 720p x 81 frames: 40 heads, D 5120, F 13824), the 151 200-row CFG pair pass, and HunyuanVideo's
119 056-token joint sequence x 24 heads — against the CPU oracle evaluated on SAMPLED ROWS.

Every op of the block is row-wise except self-attention, whose queries are row-wise too: row r of the output needs q[r] and ALL keys /
values.  So a full-size launch is checked by computing, on the CPU, exactly those output rows: `O.attention_rows` (the reference's
`torch_sdpa`, common/ops/attn/attn_weight.py:229-239, on a subset of the query rows), `O.mm` (mm_weight.py:81-88), `O.mm_fp8`
(:236-245,310-318) and `O.wan_block_rows` (= wan/infer/transformer_infer.py:289-508 restricted to a row subset).  Sampled rows always include the
first and last query block (the last one is partial: 75 600 = 295 x 256 + 80), rows whose byte offsets cross 2^31 / 2^32, and for
attention queries aimed at planted keys in the first tile, across the 65 536-key boundary, and in the last (partial: 16 keys) key tile.

Tolerances: attention — on the heads whose scores are N(0,1)-like (the case the reference's own acceptance is quoted for) |d| <= 2^-7 |ref| + 1e-3
(one bf16 ulp of the output + the reference's acceptance atol, attentions/distributed/ring/tests/test.py:97) on all but 1e-4 of the elements; on EVERY head, including
the peaky ones (score spread up to 2, where two bf16 roundings of the same fp32 q legitimately move a near-tie between top keys by more than any
elementwise bound), the fp32 triangle err(HIP vs fp32 truth) <= 1.5 x err(reference operator vs truth) + 1e-4; GEMM <= 1 bf16 ulp + atol on all but 2e-3 of the elements (test_gpu_ops.py); block relative L2 <= 1e-2 and
err(HIP vs fp32 truth) <= 1.5 x err(bf16 oracle vs truth).  Measured numbers are appended to gpurun_out/parity_summary.jsonl.
"""
import math

import pytest
import torch

from tests.util import assert_bf16_close, record, rel_l2

pytestmark = pytest.mark.gpu

S_WAN, H_WAN, D_WAN, F_WAN = 75600, 40, 5120, 13824


@pytest.fixture(scope="module")
def lib():
    from lightx2v_amd import lib as L

    L.init()
    return L


def sample_rows(S, n, block=256, seed=0, must=()):
    """n distinct rows of [0, S): the whole first `block`-row tile's corners, the last (possibly partial) tile, the middle, + seeded random."""
    last0 = (S - 1) // block * block
    fixed = {0, 1, 31, 32, 63, 64, block - 1, block, S // 2, S // 2 + 1, last0 - 1, last0, last0 + 1, S - 2, S - 1, *must}
    fixed = {r for r in fixed if 0 <= r < S}
    gen = torch.Generator().manual_seed(seed)
    extra = [int(r) for r in torch.randperm(S, generator=gen)[: 2 * n].tolist() if int(r) not in fixed][: max(0, n - len(fixed))]
    return torch.tensor(sorted(fixed) + extra, dtype=torch.long)


N_PLAIN = 8  # leading heads with unit score spread (well-conditioned: elementwise comparison with the reference operator)


def _spread(H, n_plain=N_PLAIN):
    return torch.cat([torch.ones(min(n_plain, H)), torch.linspace(1.0, 2.0, max(H - n_plain, 0))]).cuda().repeat_interleave(128)


def _attn_inputs(S_rows, S, H, seed, ld=None, n_plain=N_PLAIN):
    """q32 (fp32; score spread 1 on the first N_PLAIN heads, rising to 2 on the others so that diffuse and peaky softmax rows both occur), k, v
    bf16 on the GPU."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    D = H * 128
    q32 = torch.randn(S_rows, D, generator=g, device="cuda") * _spread(H, n_plain)
    k = torch.randn(S_rows, D, generator=g, device="cuda").to(torch.bfloat16)
    v = torch.randn(S_rows, D, generator=g, device="cuda").to(torch.bfloat16)
    return q32, k, v


def _plant(q32, k, base, S, pairs):
    """Aim query row r at key j IN HEAD 0: k[j, :128] = 3 q[r, :128], so its score 3 |q_0|^2 / sqrt(128) ~ 34 stands ~22 above the log of the sum
    of all other keys' weights (log 75600 + 1/2): o[r, :128] must be v[j, :128] to 2^-6.  Other queries see the planted key with a score of
    spread 3 instead of 1 — one mildly stronger key among 75 600, not an ill-conditioned row."""
    for r, j in pairs:
        k[base + j, :128] = (q32[base + r, :128] * 3.0).to(torch.bfloat16)


def _check_attention_rows(lib, got_rows, q32_rows, k_cpu, v_cpu, H, name, ulps=1.0, atol=1e-3, n_plain=N_PLAIN):
    from oracle import wan_oracle as O

    n = q32_rows.shape[0]
    qb = q32_rows.to(torch.bfloat16).view(n, H, 128)  # what the reference's operator is handed
    ref = O.attention_rows(qb, k_cpu.view(-1, H, 128), v_cpu.view(-1, H, 128))
    # fp32 truth from the UNROUNDED q: the reference rounds q once (bf16(q)), the HIP path rounds it once (bf16(q * scale*log2e) inside the
    # producer's rounding) — both are one rounding away from the same fp32 q
    tru = torch.empty(n, H * 128)
    for h in range(H):
        sc = (q32_rows[:, h * 128 : (h + 1) * 128] @ k_cpu[:, h * 128 : (h + 1) * 128].float().t()) / math.sqrt(128.0)
        tru[:, h * 128 : (h + 1) * 128] = torch.softmax(sc, dim=-1) @ v_cpu[:, h * 128 : (h + 1) * 128].float()
    got = got_rows.float().cpu()
    assert torch.isfinite(got).all(), f"{name}: non-finite values"
    worst, per_head = 0.0, []
    for h in range(H):
        sl = slice(h * 128, (h + 1) * 128)
        nt = tru[:, sl].norm().item()
        e_hip, e_ref = (got[:, sl] - tru[:, sl]).norm().item() / nt, (ref[:, sl].float() - tru[:, sl]).norm().item() / nt
        per_head.append((e_hip, e_ref))
        worst = max(worst, e_hip / max(e_ref, 1e-12))
    e_hip, e_ref = rel_l2(got, tru), rel_l2(ref, tru)
    record(name, rows=n, heads=H, keys=k_cpu.shape[0], err_hip_vs_fp32=e_hip, err_ref_vs_fp32=e_ref, worst_head_ratio=worst, max_abs_vs_ref=(got - ref.float()).abs().max().item(),
           per_head_first_last=[list(per_head[0]), list(per_head[-1])])
    for h, (eh, er) in enumerate(per_head):
        assert eh <= 1.5 * er + 1e-4, f"{name}: head {h}: err vs fp32 truth {eh:.3e} > 1.5 x the reference operator's {er:.3e}"
    plain = min(n_plain, H) * 128
    # one bf16 ulp of the output (two independently rounded results may sit on either side of a rounding boundary: 2^-8 .. 2^-7 relative) + atol
    assert_bf16_close(got[:, :plain], ref[:, :plain], ulps=ulps, atol=atol, bad_frac=1e-4, name=name + " (unit-spread heads)")
    return ref


PLANTS = [(5, 0), (77, 63), (300, 64), (40000, 65535), (40001, 65536), (75599, 75583), (75598, 75584), (1000, 75599)]  # (query row, key)


def test_attention_wan14b_720p_full_size(lib):
    """The launch bench.py times: x2v_attn_fwd_bf16_vt at Sq = Sk = 75 600, H = 40, pre-scaled q, V^T operand (1182 key tiles, the last
    holding 16 keys; 296 query blocks, the last holding 80 rows) — transformer_infer.py:369-379 at q, k, v [75600, 40, 128]."""
    S, H = S_WAN, H_WAN
    q32, k, v = _attn_inputs(S, S, H, seed=1)
    _plant(q32, k, 0, S, PLANTS)
    q_pre = (q32 * lib.ATTN_PRESCALE).to(torch.bfloat16)
    vt = lib.transpose_heads(v, H)
    var = lib.ATTN_FAST | lib.ATTN_Q_PRESCALED | lib.ATTN_STAGGER  # the staggered walk (what the drivers passed through round 3); the walk from tile 0 — what they pass now — is `out0` below
    out = lib.attention(q_pre, k, None, H, variant=var, vt=vt)
    assert torch.isfinite(out.float()).all()
    rows = sample_rows(S, 192, must=[r for r, _ in PLANTS])
    k_cpu, v_cpu = k.cpu(), v.cpu()
    _check_attention_rows(lib, out[rows.cuda()], q32[rows.cuda()].cpu(), k_cpu, v_cpu, H, "attn 14B 720p S=75600 H=40")
    for r, j in PLANTS:  # key-index mapping pinned independently of the oracle
        assert (out[r, :128].float() - v[j, :128].float()).abs().max().item() <= 2 ** -6, (r, j)
    # the walk without the stagger (what the Ulysses driver launches): same values up to the fp32 summation order
    out0 = lib.attention(q_pre, k, None, H, variant=lib.ATTN_FAST | lib.ATTN_Q_PRESCALED, vt=vt)
    assert torch.equal(out0[:256], out[:256]), "query block 0 starts at tile 0 either way"
    assert rel_l2(out0, out) <= 5e-3 and not torch.equal(out0, out)
    del out0
    # the kernel folding the scale itself, and the general entry on row-major V, on a few query blocks of the same problem
    blk = torch.cat([torch.arange(0, 256), torch.arange(S - 80, S)]).cuda()
    qb = q32.to(torch.bfloat16)
    ref_rows = out[blk]
    pl = N_PLAIN * 128
    o2 = lib.attention(qb[blk].contiguous(), k, None, H, variant=lib.ATTN_FAST, vt=vt)
    assert_bf16_close(o2[:, :pl], ref_rows[:, :pl].cpu(), ulps=1, atol=2e-3, bad_frac=1e-4, name="kernel-side scale vs pre-scaled q (unit-spread heads)")
    assert rel_l2(o2, ref_rows) <= 2e-2
    o3 = lib.attention(qb[blk].contiguous(), k, v, H)
    assert_bf16_close(o3[:, :pl], ref_rows[:, :pl].cpu(), ulps=1, atol=2e-3, bad_frac=1e-4, name="row-major-V pipeline vs ping-pong kernel (unit-spread heads)")
    assert rel_l2(o3, ref_rows) <= 2e-2


def test_attention_cfg_pair_launch_full_size(lib):
    """x2v_attn_fwd_bf16_vt_batched as `WanTransformerInfer.infer_self_attn` launches it in pair mode at 14B 720p: two sequences of 75 600
    tokens in slots of 75 648 rows (151 296 stacked rows; grid z = 2; V^T over the stacked rows), every slot row a query."""
    S, H, B = S_WAN, H_WAN, 2
    Sp = (S + 63) // 64 * 64
    q32, k, v = _attn_inputs(B * Sp, S, H, seed=2)
    for b in range(B):
        q32[b * Sp + S : (b + 1) * Sp] = 0
        k[b * Sp + S : (b + 1) * Sp] = 0
        v[b * Sp + S : (b + 1) * Sp] = 0
        _plant(q32, k, b * Sp, S, PLANTS[b::2])
    q_pre = (q32 * lib.ATTN_PRESCALE).to(torch.bfloat16)
    vt = lib.transpose_heads(v, H)
    out = lib.attention_batched(q_pre, k, vt, H, B, Sp, S, prescaled=True, stagger=True)
    assert torch.isfinite(out.float()).all(), "padding rows must be written"
    # the form the fused driver launches since round 4 (wan.SELF_ATTN_STAGGER = False: the walk starts at tile 0 for every block): the same values up to
    # the fp32 summation order, identical where the staggered walk starts at tile 0 too (query block 0 of either sequence)
    out_ns = lib.attention_batched(q_pre, k, vt, H, B, Sp, S, prescaled=True, stagger=False)
    assert rel_l2(out_ns, out) <= 5e-3 and torch.isfinite(out_ns.float()).all()
    for b in range(B):
        assert torch.equal(out_ns[b * Sp : b * Sp + 256], out[b * Sp : b * Sp + 256])
        for r, j in PLANTS[b::2]:
            assert (out_ns[b * Sp + r, :128].float() - v[b * Sp + j, :128].float()).abs().max().item() <= 2 ** -6, (b, r, j)
    del out_ns
    for b in range(B):
        rows = sample_rows(S, 96, seed=10 + b, must=[r for r, _ in PLANTS[b::2]])
        sl = slice(b * Sp, b * Sp + S)
        _check_attention_rows(lib, out[sl][rows.cuda()], q32[sl][rows.cuda()].cpu(), k[sl].cpu(), v[sl].cpu(), H, f"attn pair launch, sequence {b}")
        for r, j in PLANTS[b::2]:
            assert (out[b * Sp + r, :128].float() - v[b * Sp + j, :128].float()).abs().max().item() <= 2 ** -6, (b, r, j)


def test_attention_hunyuan_720p_129f_full_size(lib):
    """HunyuanVideo-13B 720p x 129 frames (config #5): 118 800 image tokens + 256 text tokens (200 valid) x 24 heads; the first varlen segment
    (image + valid text = 119 000 rows) is one dense launch on strided views of the fused QKV buffer [rows, 3 x 3072], exactly as
    `HunyuanTransformerInfer._attention` issues it (hunyuan/infer/transformer_infer.py:119-146 with cu_seqlens of pre_infer.py:50-56)."""
    H, n_img, n_txt, n_valid = 24, 118800, 256, 200
    D = H * 128
    L = n_img + n_txt
    S = n_img + n_valid
    g = torch.Generator(device="cuda").manual_seed(3)
    q32 = torch.randn(L, D, generator=g, device="cuda") * _spread(H)
    qkv = torch.randn(L, 3 * D, generator=g, device="cuda").to(torch.bfloat16)
    qkv[:, :D] = (q32 * lib.ATTN_PRESCALE).to(torch.bfloat16)
    q, k, v = qkv[:, :D], qkv[:, D : 2 * D], qkv[:, 2 * D :]
    plants = [(7, 118799), (118800, 3), (118999, 118999), (60000, 65536)]
    for r, j in plants:
        k[j, :128] = (q32[r, :128] * 3.0).to(torch.bfloat16)
    out = torch.zeros(L, D, dtype=torch.bfloat16, device="cuda")
    var = lib.ATTN_FAST | lib.ATTN_Q_PRESCALED | lib.ATTN_STAGGER
    lib.attention(q[:S], k[:S], v[:S], H, 128, out=out[:S], variant=var)
    lib.attention(q[S:], k[S:], v[S:], H, 128, out=out[S:], variant=var)
    assert torch.isfinite(out.float()).all()
    rows = sample_rows(S, 160, seed=4, must=[r for r, _ in plants] + [n_img - 1, n_img])
    _check_attention_rows(lib, out[rows.cuda()], q32[rows.cuda()].cpu(), k[:S].cpu().contiguous(), v[:S].cpu().contiguous(), H, "attn Hunyuan 720p129f S=119000 H=24")
    for r, j in plants:
        assert (out[r, :128].float() - v[j, :128].float()).abs().max().item() <= 2 ** -6, (r, j)
    # 56 keys: the bf16 rounding of P is no longer averaged over thousands of keys — the small-shape tolerance of test_gpu_ops.py (atol 8e-3);
    # the triangle (HIP 2.8e-3 vs the reference operator's 2.8e-3 from the fp32 truth on MI355X) is the sharp check here
    _check_attention_rows(lib, out[S:], q32[S:].cpu(), k[S:].cpu().contiguous(), v[S:].cpu().contiguous(), H, "attn Hunyuan padded-text segment", ulps=1.0, atol=8e-3)


# ------------------------------------------------------------------------------------------------ projections
@pytest.mark.parametrize("M", [75600, 151200])
@pytest.mark.parametrize("K,N", [(5120, 5120), (5120, 13824), (13824, 5120)])
def test_gemm_bf16_full_size_rows_vs_oracle(lib, M, K, N):
    """The six projection launches of a 14B 720p block at their real row counts (one forward: 75 600; the CFG pair pass: 151 200 — x of
    ffn_2 is 4.18 GB, byte offsets cross 2^31 and approach 2^32), all epilogues, sampled rows vs `O.mm` + the reference's separate ops."""
    from oracle import wan_oracle as O

    assert lib.gemm_kernel_choice(M, N, K) == 3
    g = torch.Generator(device="cuda").manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
    b = (torch.randn(N, generator=g, device="cuda") * 0.1).to(torch.bfloat16)
    cross31 = (1 << 31) // (2 * K)  # first row whose byte offset in x is >= 2^31
    rows = sample_rows(M, 256, seed=M + N, must=[cross31 - 1, cross31, cross31 + 1, (1 << 31) // (2 * N), (1 << 32) // (2 * N) - 1, (1 << 32) // (2 * N)])
    rc = rows.cuda()
    xr, wc, bc = x[rc].cpu(), w.cpu(), b.cpu()
    ref = O.mm(xr, wc, bc)
    got = lib.gemm(x, w, b)
    assert_bf16_close(got[rc], ref, ulps=1, atol=2e-3, bad_frac=1e-3, name=f"gemm {M}x{K}x{N}")
    assert torch.isfinite(got.float()).all()
    record(f"gemm bf16 {M}x{K}x{N}", rel_l2=rel_l2(got[rc], ref))
    del got
    ge = lib.gemm(x, w, b, epilogue=lib.EPI_GELU_TANH)
    assert_bf16_close(ge[rc], torch.nn.functional.gelu(ref, approximate="tanh"), ulps=1, atol=2e-3, bad_frac=2e-3, name="gemm+gelu")
    del ge
    res = torch.randn(M, N, generator=g, device="cuda").to(torch.bfloat16)
    gate = (torch.randn(1, N, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
    ref_r = res[rc].cpu()
    ref_r.add_(ref * gate.cpu().squeeze(0))
    untouched = res.clone()
    lib.gemm(x, w, b, epilogue=lib.EPI_RESIDUAL, resid=res, gate=gate)
    assert_bf16_close(res[rc], ref_r, ulps=1, atol=6e-3, bad_frac=2e-3, name="gemm+gate-residual")
    assert (res != untouched).float().mean().item() > 0.9 and torch.isfinite(res.float()).all()
    if N == D_WAN and K == D_WAN:  # the v projection writing V^T (x2v_gemm_bf16_vt) at the same size
        vt = lib.gemm_vt(x, w, b, H_WAN)
        v = lib.gemm(x, w, b)
        t_last = (M - 1) // 64
        for t in (0, 1, M // 128, t_last - 1, t_last):
            n_valid = min(64, M - t * 64)
            blk = v[t * 64 : t * 64 + n_valid].view(n_valid, H_WAN, 128).permute(1, 2, 0)  # [H, 128, keys]
            assert torch.equal(vt[:, t, :, :n_valid], blk), f"V^T tile {t}"
            assert not vt[:, t, :, n_valid:].any()


@pytest.mark.parametrize("M,K,N", [(75600, 5120, 5120), (151200, 5120, 13824), (151200, 13824, 5120)])
def test_gemm_fp8_full_size_rows_vs_oracle(lib, M, K, N):
    """Config #4's w8a8 projections (per-token x per-channel, mm_weight.py:236-245,310-318) at full row counts: the quantiser's codes and
    scales on sampled rows vs `O.quant_fp8_per_token`, then the GEMM on the ORACLE's codes for those rows (a quantiser tie cannot hide in
    the GEMM tolerance) vs `O.mm_fp8`."""
    from oracle import wan_oracle as O

    assert lib.gemm_kernel_choice(M, N, K, fp8=True) == 2
    g = torch.Generator(device="cuda").manual_seed(M + K + N + 8)
    x = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
    b = (torch.randn(N, generator=g, device="cuda") * 0.1).to(torch.bfloat16)
    rows = sample_rows(M, 192, seed=M + N + 1, must=[(1 << 31) // K, (1 << 32) // K - 1 if (1 << 32) // K - 1 < M else 0])
    rc = rows.cuda()
    xr = x[rc].cpu()
    wq, sw = O.quant_fp8_weight_per_channel(w.cpu())
    xq_ref, sx_ref = O.quant_fp8_per_token(xr)
    xq, sx = lib.quant_fp8_rowwise(x)
    assert torch.allclose(sx[rc].cpu(), sx_ref, rtol=1e-6, atol=0)
    mism = (xq[rc].cpu().view(torch.uint8) != xq_ref.view(torch.uint8)).float().mean().item()
    assert mism <= 1e-3, f"{mism} of e4m3 codes differ"
    xq.view(torch.uint8)[rc] = xq_ref.view(torch.uint8).cuda()
    sx[rc] = sx_ref.cuda()
    wqd, swd = wq.cuda(), sw.cuda()
    ref = O.mm_fp8(xr, wq, sw, b.cpu())
    got = lib.gemm_fp8(xq, sx, wqd, swd, b)
    assert_bf16_close(got[rc], ref, ulps=1, atol=4e-3, bad_frac=2e-3, name=f"fp8 gemm {M}x{K}x{N}")
    assert torch.isfinite(got.float()).all()
    record(f"gemm fp8 {M}x{K}x{N}", rel_l2=rel_l2(got[rc], ref), code_mismatch=mism)
    del got
    ge = lib.gemm_fp8(xq, sx, wqd, swd, b, epilogue=lib.EPI_GELU_TANH)
    assert_bf16_close(ge[rc], torch.nn.functional.gelu(ref, approximate="tanh"), ulps=1, atol=4e-3, bad_frac=2e-3, name="fp8 gemm+gelu")
    del ge
    res = torch.randn(M, N, generator=g, device="cuda").to(torch.bfloat16)
    gate = (torch.randn(1, N, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
    ref_r = res[rc].cpu()
    ref_r.add_(ref * gate.cpu().squeeze(0))
    lib.gemm_fp8(xq, sx, wqd, swd, b, epilogue=lib.EPI_RESIDUAL, resid=res, gate=gate)
    assert_bf16_close(res[rc], ref_r, ulps=1, atol=8e-3, bad_frac=2e-3, name="fp8 gemm+gate-residual")


# ------------------------------------------------------------------------------------------------ one whole block
def test_wan14b_block_720p_pair_pass_vs_oracle_rows():
    """One Wan2.1-14B block at the benchmark's own shape — 75 600 tokens (token grid 21 x 45 x 80), both CFG forwards in one pass over
    151 296 stacked rows, exactly as `WanModel._forward_pair` drives `infer_block` — vs `O.wan_block_rows` on sampled rows of both
    forwards, plus the fp32 evaluation of the same graph as truth: err(HIP) <= 1.5 x err(bf16 oracle)."""
    from lightx2v_amd import lib, scheduler, synth, wan
    from oracle import wan_oracle as O

    dims = dict(synth.WAN_DIMS["wan2.1-14b"], num_layers=1)
    wl = synth.WORKLOADS["wan14b_720px81f"]
    ts = wl["target_shape"]
    S = synth.seq_len_of(ts)
    assert S == S_WAN
    wd = synth.synth_wan_weights(dims, seed=31)
    lat, ctx, ctx_null = synth.synth_inputs(dims, ts)
    t = torch.tensor(777)
    embed_o, grid, x_o, embed0_o, _, context_c = O.wan_pre_infer(wd, dims, lat.to(torch.bfloat16), t, ctx)
    _, _, _, _, _, context_u = O.wan_pre_infer(wd, dims, lat.to(torch.bfloat16), t, ctx_null)
    freqs = O.rope_freqs_table(128)
    rows = sample_rows(S, 128, seed=5)
    ref_c, ref_u = O.wan_block_rows(wd, 0, dims, grid, x_o, embed0_o, freqs, [context_c, context_u], rows)
    with O.truth_precision(torch.float32):
        tru_c, tru_u = O.wan_block_rows(O.upcast(wd), 0, dims, grid, x_o.float(), embed0_o.float(), freqs, [context_c.float(), context_u.float()], rows)

    cfg = wan.default_config(dims, target_shape=ts, target_video_length=wl["frames"], infer_steps=4)
    model = wan.WanModel(cfg, {k: v.cuda() for k, v in wd.items()})
    sch = scheduler.WanScheduler(cfg, device="cuda")
    sch.prepare(latents=lat)
    model.set_scheduler(sch)
    sch.step_pre(1)
    tr = model.transformer_infer
    Sp = (S + 63) // 64 * 64
    X = torch.zeros((2 * Sp, dims["dim"]), dtype=torch.bfloat16, device="cuda")
    X[:S].copy_(x_o)
    X[Sp : Sp + S].copy_(x_o)
    grid_sizes = torch.tensor([list(grid)], dtype=torch.long)
    rope = wan.rope_cos_sin_table(128, "cuda")
    tr._pair = (S, Sp)
    try:
        out = tr.infer_block(model.transformer_weights.blocks[0], grid_sizes, embed_o.cuda(), X, embed0_o.cuda(), torch.tensor([S]), rope, (context_c.cuda(), context_u.cuda()))
    finally:
        tr._pair = None
    assert torch.isfinite(out.float()).all()
    rc = rows.cuda()
    for name, got, ref, tru in (("cond", out[:S][rc], ref_c, tru_c), ("uncond", out[Sp : Sp + S][rc], ref_u, tru_u)):
        e = rel_l2(got, ref)
        e_hip, e_ref = rel_l2(got, tru), rel_l2(ref, tru)
        record(f"Wan-14B block S=75600 pair pass ({name})", rows=len(rows), rel_l2_vs_oracle=e, err_hip_vs_fp32=e_hip, err_oracle_vs_fp32=e_ref)
        assert e <= 1e-2, f"{name}: relative L2 vs oracle {e:.3e}"
        assert e_hip <= 1.5 * e_ref + 1e-4, f"{name}: err vs fp32 truth {e_hip:.3e} > 1.5 x the bf16 oracle's {e_ref:.3e}"


def test_wan14b_i2v_block_720p_vs_oracle_rows():
    """The i2v block at the size the reference publishes its numbers for (Wan2.1-I2V-14B 720p: 75 600 tokens, 40 heads, 257 CLIP + 512 text context rows)
    — `infer_block` with the second cross-attention over the image tokens and the cached image K / V — vs `O.wan_block_rows` on sampled rows, with the fp32
    evaluation as truth (err(HIP) <= 1.5 x err(bf16 oracle)).  Block-boundary inputs come from the oracle's i2v pre-infer (36-channel patch embedding,
    CLIP-feature MLP); the HIP pre-infer at this size is checked against them too."""
    from lightx2v_amd import scheduler, synth, wan
    from oracle import wan_oracle as O

    dims = dict(synth.WAN_DIMS["wan2.1-14b"], num_layers=1, task="i2v", clip_dim=1280)
    wl = synth.WORKLOADS["wan14b_720px81f"]
    ts = wl["target_shape"]
    S = synth.seq_len_of(ts)
    wd = synth.synth_wan_i2v_weights(dims, seed=33)
    lat, ctx, _ = synth.synth_inputs(dims, ts)
    image = synth.synth_i2v_inputs(dims, ts)
    t = torch.tensor(640)
    embed_o, grid, x_o, embed0_o, _, context_o = O.wan_pre_infer(wd, dims, lat.to(torch.bfloat16), t, ctx, image=image)
    assert context_o.shape[0] == 257 + 512 and x_o.shape[0] == S
    freqs = O.rope_freqs_table(128)
    rows = sample_rows(S, 96, seed=15)
    ref = O.wan_block_rows(wd, 0, dims, grid, x_o, embed0_o, freqs, context_o, rows)
    with O.truth_precision(torch.float32):
        tru = O.wan_block_rows(O.upcast(wd), 0, dims, grid, x_o.float(), embed0_o.float(), freqs, context_o.float(), rows)
    cfg = wan.default_config(dims, task="i2v", in_dim=36, cross_attn_2_type="hip_flash", target_shape=ts, target_video_length=wl["frames"], infer_steps=4, enable_cfg=False)
    model = wan.WanModel(cfg, {k: v.cuda() for k, v in wd.items()})
    sch = scheduler.WanScheduler(cfg, device="cuda")
    sch.prepare(latents=lat)
    sch.timesteps[1] = 640
    model.set_scheduler(sch)
    sch.step_pre(1)
    inputs = {"text_encoder_output": {"context": [c.cuda() for c in ctx], "context_null": []}, "image_encoder_output": {k: v.cuda() for k, v in image.items()}}
    embed, grid_sizes, (x, embed0, seq_lens, rope, context) = model.pre_infer.infer(model.pre_weight, inputs, positive=True)
    assert rel_l2(x, x_o) <= 1e-2 and rel_l2(context[:257], context_o[:257]) <= 1e-2 and rel_l2(context[257:], context_o[257:]) <= 1e-2
    tr = model.transformer_infer
    out = tr.infer_block(model.transformer_weights.blocks[0], grid_sizes, embed_o.cuda(), x_o.cuda().clone(), embed0_o.cuda(), torch.tensor([S]), rope, context_o.cuda())
    assert torch.isfinite(out.float()).all()
    got = out[rows.cuda()]
    e, e_hip, e_ref = rel_l2(got, ref), rel_l2(got, tru), rel_l2(ref, tru)
    record("Wan-14B i2v block S=75600", rows=len(rows), rel_l2_vs_oracle=e, err_hip_vs_fp32=e_hip, err_oracle_vs_fp32=e_ref)
    assert e <= 1e-2, f"i2v block: relative L2 vs oracle {e:.3e}"
    assert e_hip <= 1.5 * e_ref + 1e-4, f"i2v block: err vs fp32 truth {e_hip:.3e} > 1.5 x the bf16 oracle's {e_ref:.3e}"


# ------------------------------------------------------------------------------------------------ HunyuanVideo blocks at config #5's size
@pytest.mark.parametrize("kind", ["double", "single"])
def test_hunyuan13b_block_720p_129f_vs_oracle_rows(kind):
    """One HunyuanVideo-13B double block / single block (hidden 3072, 24 heads, mlp 12288) at config #5's full token count — 118 800 image tokens
    (token grid 33 x 45 x 80) + 256 text tokens, 200 of them valid (two attention segments) — through `HunyuanTransformerInfer.infer`, against
    `oracle.hunyuan_oracle.double_block_rows` / `single_block_rows` (hunyuan/infer/transformer_infer.py:81-384 restricted to sampled image rows;
    the k / v projections of all 119 056 rows run on the host).  Inputs are seeded tensors at the block boundary; relative L2 <= 4e-3 (about 3 x the
    measured 1.2e-3 / 3.4e-4; the 60-block forward against the fp32 graph is test_hunyuan13b_720p_129f_full_forward_vs_fp32_truth)."""
    from lightx2v_amd import hunyuan as hy, synth
    from oracle import hunyuan_oracle as H

    dims = dict(synth.HUNYUAN_DIMS["hunyuan-13b"], double_blocks=1 if kind == "double" else 0, single_blocks=0 if kind == "double" else 1)
    pref = "double_blocks." if kind == "double" else "single_blocks."
    wd = {k: v for k, v in synth.synth_hunyuan_weights(dict(dims, double_blocks=1, single_blocks=1), seed=21).items() if k.startswith(pref)}
    grid = (33, 45, 80)
    n_img, n_txt, n_valid = grid[0] * grid[1] * grid[2], dims["text_len"], 200
    assert n_img == 118800 and n_txt == 256
    gen = torch.Generator().manual_seed(6)
    img = torch.randn(n_img, dims["hidden"], generator=gen).to(torch.bfloat16)
    txt = torch.randn(n_txt, dims["hidden"], generator=gen).to(torch.bfloat16)
    vec = torch.randn(1, dims["hidden"], generator=gen).to(torch.bfloat16)
    cos, sin = H.rope_tables(list(grid))
    cu = torch.tensor([0, n_img + n_valid, n_img + n_txt], dtype=torch.int32)
    rows = sample_rows(n_img, 96, seed=8)
    with torch.no_grad():
        if kind == "double":
            ref, _ = H.double_block_rows(wd, 0, img, txt, vec, (cos, sin), dims["heads"], cu, rows)
        else:
            ref = H.single_block_rows(wd, 0, torch.cat((img, txt), 0), vec, n_txt, (cos, sin), dims["heads"], dims["hidden"], cu, rows)[: len(rows)]
    cfg = hy.default_config(dims, infer_steps=4)
    tw = hy.HunyuanTransformerWeights(cfg)
    tw.load({k: v.cuda() for k, v in wd.items()})
    tr = hy.HunyuanTransformerInfer(cfg)
    out, _ = tr.infer(tw, img.cuda(), txt.cuda(), vec.cuda(), cu, n_img + n_txt, (cos.cuda(), sin.cuda()))
    assert out.shape[0] == n_img and torch.isfinite(out.float()).all()
    e = rel_l2(out[rows.cuda()], ref)
    record(f"Hunyuan-13B {kind} block at 118800 + 256 tokens", rows=len(rows), rel_l2_vs_oracle=e)
    assert e <= 4e-3, f"Hunyuan-13B {kind} block at full size: relative L2 vs oracle {e:.3e}"  # measured on MI355X: 1.2e-3 (double) / 3.4e-4 (single)


# ------------------------------------------------------------------------------------------------ config #4: a w8a8 block at full size
@pytest.mark.parametrize("tokens,ref_rounding", [("2560", False), ("2560", True), ("75600", False)])
def test_wan14b_fp8_block_vs_oracle_rows(tokens, ref_rounding):
    """BASELINE config #4's block — Wan2.1-14B with the w8a8 operator class (per-channel e4m3 weights quantised at load, per-token dynamic
    activations; mm_weight.py:236-245,287-319) — through `infer_block` (LayerNorm fused with the activation quantisation, fp8 MFMA GEMMs, bf16
    attention) against the oracle's same block inside `O.fp8_blocks()`, whose composition is pinned bit-exactly to the reference's own model
    (tests/test_oracle_golden.py::test_fp8_block_mode_bit_exact_against_live_reference_model).  At 2 560 tokens on all rows; at the benchmark's
    75 600 tokens (one forward: config #4 has no CFG) on sampled rows via `wan_block_rows`.
    Tolerance.  A w8a8 graph is far more sensitive to upstream rounding than the bf16 one: an activation that differs by one bf16 ulp lands on
    another e4m3 code (3 mantissa bits: a 6 % step) in ~7 % of the cases, and a GEMM passes that on undamped — so two correct implementations of the
    same w8a8 block that round their LayerNorm differently sit ~1e-2 apart (measured on MI355X: 1.10e-2 at both sizes, against 2.9e-3 for the
    bf16 block), while the graph's own quantisation error is 2.6e-2.  That claim is ANCHORED, not asserted: the fp32
    evaluation of the unquantised block is the truth, and the HIP w8a8 block may be at most 1.25 x as far from it as the reference's w8a8 arithmetic
    (the oracle inside `fp8_blocks()`) is — the same triangle every bf16 test carries — plus: at most half of the quantisation error away from the
    w8a8 oracle itself."""
    from lightx2v_amd import scheduler, synth, wan
    from oracle import wan_oracle as O

    dims = dict(synth.WAN_DIMS["wan2.1-14b"], num_layers=1)
    ts = (16, 5, 32, 64) if tokens == "2560" else synth.WORKLOADS["wan14b_720px81f"]["target_shape"]
    S = synth.seq_len_of(ts)
    assert S == int(tokens)
    wd = synth.synth_wan_weights(dims, seed=41)
    lat, ctx, _ = synth.synth_inputs(dims, ts)
    t = torch.tensor(500)
    embed_o, grid, x_o, embed0_o, _, context_o = O.wan_pre_infer(wd, dims, lat.to(torch.bfloat16), t, ctx)
    freqs = O.rope_freqs_table(128)
    rows = torch.arange(S) if S <= 4096 else sample_rows(S, 128, seed=9)
    with O.fp8_blocks():
        ref = O.wan_block_rows(wd, 0, dims, grid, x_o, embed0_o, freqs, context_o, rows)
    ref_bf16 = O.wan_block_rows(wd, 0, dims, grid, x_o, embed0_o, freqs, context_o, rows)
    with O.truth_precision(torch.float32):
        tru = O.wan_block_rows(O.upcast(wd), 0, dims, grid, x_o.float(), embed0_o.float(), freqs, context_o.float(), rows)
    cfg = wan.default_config(dims, target_shape=ts, target_video_length=(ts[1] - 1) * 4 + 1, infer_steps=4, enable_cfg=False, hip_ref_rounding=ref_rounding,
                             mm_config={"mm_type": "W-fp8-channel-sym-A-fp8-channel-sym-dynamic-Hip", "weight_auto_quant": True})
    model = wan.WanModel(cfg, {k: v.cuda() for k, v in wd.items()})
    sch = scheduler.WanScheduler(cfg, device="cuda")
    sch.prepare(latents=lat)
    model.set_scheduler(sch)
    sch.step_pre(1)
    tr = model.transformer_infer
    grid_sizes = torch.tensor([list(grid)], dtype=torch.long)
    rope = wan.rope_cos_sin_table(128, "cuda")
    out = tr.infer_block(model.transformer_weights.blocks[0], grid_sizes, embed_o.cuda(), x_o.cuda().clone(), embed0_o.cuda(), torch.tensor([S]), rope, context_o.cuda())
    assert torch.isfinite(out.float()).all()
    got = out[rows.cuda()]
    e, e_q = rel_l2(got, ref), rel_l2(ref, ref_bf16)
    e_hip, e_ref = rel_l2(got, tru), rel_l2(ref, tru)
    record(f"Wan-14B w8a8 block S={S} (ref_rounding={ref_rounding})", rows=len(rows), rel_l2_vs_fp8_oracle=e, fp8_oracle_vs_bf16_oracle=e_q, err_hip_w8a8_vs_fp32=e_hip,
           err_oracle_w8a8_vs_fp32=e_ref)
    # the triangle: the HIP w8a8 block is no further from the fp32 truth than the reference's w8a8 arithmetic is (x 1.25)
    assert e_hip <= 1.25 * e_ref, f"w8a8 block at S={S}: {e_hip:.3e} from the fp32 truth, the w8a8 oracle {e_ref:.3e}"
    # the w8a8 graph sits e_q (quantisation error) away from the bf16 graph; the HIP block must match the w8a8 ORACLE much closer than that
    assert e <= 0.5 * e_q, f"w8a8 block at S={S}: relative L2 vs the w8a8 oracle {e:.3e} (quantisation error of the graph itself: {e_q:.3e})"


# ------------------------------------------------------------------------------------------------ config #2: the whole 30-layer forward at S = 20 280
def test_wan13b_config2_full_forward_vs_fp32_truth():
    """BASELINE config #2 in full depth and length: the 30-layer Wan2.1-1.3B conditional forward at 480p x 49 frames (20 280 tokens), HIP path vs
    the oracle's statements evaluated in fp32 through plain PyTorch on the GPU (`O.truth_precision(float32, device="cuda")`; the bf16 CPU
    oracle would need ~8e13 FLOP of host attention).  The anchored test at 1 280 tokens measured the HIP forward AND the reference's bf16 CPU path
    1.2e-2 from this fp32 graph (30 layers of bf16 rounding); the same bound, with head room, must hold at the real sequence length."""
    from lightx2v_amd import scheduler, synth, wan
    from oracle import wan_oracle as O

    dims = synth.WAN_DIMS["wan2.1-1.3b"]
    wl = synth.WORKLOADS["wan1.3b_480px49f"]
    ts = wl["target_shape"]
    assert synth.seq_len_of(ts) == 20280
    wd = synth.synth_wan_weights(dims, seed=17)
    lat, ctx, ctx_null = synth.synth_inputs(dims, ts)
    t = torch.tensor(640)
    with O.truth_precision(torch.float32, device="cuda"), torch.no_grad():
        tru = O.wan_forward(O.upcast(wd, device="cuda"), dims, lat.to(torch.bfloat16).float().cuda(), t, O.upcast(ctx, device="cuda")).cpu()
    torch.cuda.empty_cache()
    cfg = wan.default_config(dims, target_shape=ts, target_video_length=wl["frames"], infer_steps=4)
    model = wan.WanModel(cfg, {k: v.cuda() for k, v in wd.items()})
    sch = scheduler.WanScheduler(cfg, device="cuda")
    sch.prepare(latents=lat)
    sch.timesteps[1] = 640
    model.set_scheduler(sch)
    sch.step_pre(1)
    inputs = {"text_encoder_output": {"context": [c.cuda() for c in ctx], "context_null": [c.cuda() for c in ctx_null]}}
    got = model._forward(inputs, True)
    assert got.shape == tru.shape and torch.isfinite(got).all()
    e = rel_l2(got, tru)
    record("Wan-1.3B 30-layer forward at S=20280 (config #2) vs fp32 truth", err_hip_vs_fp32=e)
    assert e <= 2e-2, f"config #2 forward: {e:.3e} from the fp32 graph (1.2e-2 at S = 1280 for the HIP path and for the reference's bf16 path alike)"


# ------------------------------------------------------------------------------------------------ configs #3 / #4: the whole benched forward
@pytest.fixture(scope="module")
def wan14b_720p():
    """Weights (what bench.py builds: seed 0, generated on the device), inputs and — computed once for both tests below, ~100 s of fp32 GEMMs — the
    fp32 TRUTH of the 40-layer conditional forward at 75 600 tokens: the oracle's statements through plain PyTorch on the same GPU
    (`O.truth_precision(float32, device="cuda")`; attention exactly, in query chunks: the full fp32 score tensor would be 914 GB)."""
    from lightx2v_amd import synth
    from oracle import wan_oracle as O

    dims = synth.WAN_DIMS["wan2.1-14b"]
    wl = synth.WORKLOADS["wan14b_720px81f"]
    ts = wl["target_shape"]
    assert synth.seq_len_of(ts) == S_WAN
    wd = synth.synth_wan_weights(dims, seed=0, device="cuda", gen_device="cuda")
    lat, ctx, ctx_null = synth.synth_inputs(dims, ts)
    t = torch.tensor(500)
    with O.truth_precision(torch.float32, device="cuda"), torch.no_grad():
        wd32 = {k: v.float() for k, v in wd.items()}
        tru = O.wan_forward(wd32, dims, lat.to(torch.bfloat16).float().cuda(), t, O.upcast(ctx, device="cuda")).cpu()
        del wd32
    torch.cuda.empty_cache()
    yield dict(dims=dims, wl=wl, ts=ts, wd=wd, lat=lat, ctx=ctx, ctx_null=ctx_null, t=t, tru=tru)
    wd.clear()
    torch.cuda.empty_cache()


def _hip_forward(env, **cfg_extra):
    from lightx2v_amd import scheduler, wan

    cfg = wan.default_config(env["dims"], target_shape=env["ts"], target_video_length=env["wl"]["frames"], infer_steps=4, **cfg_extra)
    model = wan.WanModel(cfg, env["wd"])
    sch = scheduler.WanScheduler(cfg, device="cuda")
    sch.prepare(latents=env["lat"])
    sch.timesteps[1] = int(env["t"])
    model.set_scheduler(sch)
    sch.step_pre(1)
    inputs = {"text_encoder_output": {"context": [c.cuda() for c in env["ctx"]], "context_null": [c.cuda() for c in env["ctx_null"]]}}
    got = model._forward(inputs, True).float().cpu()
    del model
    torch.cuda.empty_cache()
    return got


def test_wan14b_720p_full_forward_vs_fp32_truth(wan14b_720p):
    """THE forward the benchmark times — Wan2.1-14B, all 40 layers, 720p x 81 frames = 75 600 tokens (pre-infer, 40 fused blocks with the staggered
    16x16x32 attention launch, post-infer) — against the fp32 truth of the fixture above.  Bound: 2e-2 relative L2 on the noise prediction; the
    anchored tests show both the HIP path and the reference's bf16 CPU path 1.2e-2 from the fp32 graph after 30 layers at 1 280 tokens (and the
    HIP path 1.28e-2 at 20 280), so 40 layers at 75 600 tokens have head room without hiding a broken layer (one wrong block moves the result
    by O(1))."""
    env = wan14b_720p
    got = _hip_forward(env)
    assert torch.isfinite(got).all()
    e = rel_l2(got, env["tru"])
    record("Wan-14B 40-layer forward at S=75600 (the benched forward) vs fp32 truth", err_hip_vs_fp32=e)
    assert got.shape == env["tru"].shape and e <= 2e-2, f"the benched forward is {e:.3e} from the fp32 graph"


def test_wan14b_720p_w8a8_full_forward_anchored_to_fp32_truth(wan14b_720p):
    """BASELINE config #4's forward in full depth and length: Wan2.1-14B with the w8a8 operator class in all 40 blocks
    (mm_weight.py:236-245,287-319; LayerNorm fused with the per-token quantisation, fp8 MFMA GEMMs, bf16 attention) at 75 600 tokens, one
    forward (config #4 has no CFG), in a triangle with the same fp32 truth:
        err(HIP w8a8 forward vs truth)  <=  1.25 x err(the ORACLE's w8a8 forward vs truth).
    The oracle's leg is `O.wan_forward` inside `O.fp8_blocks()` — the statements pinned bit-exactly to the reference's own w8a8 model
    (tests/test_oracle_golden.py) — evaluated through plain PyTorch on this GPU (bf16 activations; `xq.float() @ wq.float().t()` in fp32 exactly as
    written; torch's SDPA): on the CPU its attention alone would be 5e15 FLOP.  No library GEMM or attention of this repo is in that leg."""
    from oracle import wan_oracle as O

    env = wan14b_720p
    got = _hip_forward(env, enable_cfg=False, mm_config={"mm_type": "W-fp8-channel-sym-A-fp8-channel-sym-dynamic-Hip", "weight_auto_quant": True})
    assert torch.isfinite(got).all()
    with O.fp8_blocks(), torch.no_grad():
        ref = O.wan_forward(env["wd"], env["dims"], env["lat"].to(torch.bfloat16).cuda(), env["t"], [c.cuda() for c in env["ctx"]]).float().cpu()
    torch.cuda.empty_cache()
    e_hip, e_ref, e = rel_l2(got, env["tru"]), rel_l2(ref, env["tru"]), rel_l2(got, ref)
    record("Wan-14B w8a8 40-layer forward at S=75600 (config #4) vs fp32 truth", err_hip_w8a8_vs_fp32=e_hip, err_oracle_w8a8_vs_fp32=e_ref, hip_vs_oracle_w8a8=e)
    assert got.shape == env["tru"].shape
    assert e_hip <= 1.25 * e_ref, f"w8a8 forward: {e_hip:.3e} from the fp32 truth, the oracle's w8a8 forward {e_ref:.3e}"


# ------------------------------------------------------------------------------------------------ config #5: the whole HunyuanVideo-13B forward
def test_hunyuan13b_720p_129f_full_forward_anchored():
    """BASELINE config #5's forward in full length — HunyuanVideo-13B at 720p x 129 frames: 118 800 image + 256 text tokens (200 valid: two attention
    segments), pre-infer with the masked token refiner, double + single block stack, post-infer (hunyuan/infer/{pre,transformer,post}_infer.py) —
    through `HunyuanModel.infer`, two legs:
      A. FULL DEPTH, 20 double + 40 single blocks: against the ORACLE's own bf16 statements (`oracle.hunyuan_oracle.forward`, pinned bit-exactly to the
         reference's classes at a reduced width) evaluated through plain PyTorch on this GPU — `torch.addmm`, torch's SDPA, bf16 elementwise ops: an
         independent implementation of the same graph at a size (1.0e16 FLOP of attention) the host cannot run.  Two bf16 evaluations of a 60-block
         stack that round differently: relative L2 <= 2.5e-2 on the noise prediction (measured on MI355X: 1.58e-2; at a fifth of the depth each sits
         9.4e-3 from the fp32 graph and 8.2e-3 from the other, leg B).
      B. TRUTH-ANCHORED at the same token count and a fifth of the depth (4 double + 8 single blocks, the first ones of the same checkpoint): the same
         statements in fp32 (`O.truth_precision`, exact attention in query chunks) are the truth, and err(HIP vs truth) <= 1.5 x err(bf16 oracle
         through PyTorch vs truth).  (The fp32 evaluation of all 60 blocks at 119 056 tokens is ~4 minutes of fp32 GEMMs per run.)"""
    from lightx2v_amd import hunyuan as hy, synth
    from oracle import hunyuan_oracle as H
    from oracle import wan_oracle as O

    dims = synth.HUNYUAN_DIMS["hunyuan-13b"]
    wl = synth.HUNYUAN_WORKLOADS["hunyuan13b_720px129f"]
    ts = wl["target_shape"]
    wd = synth.synth_hunyuan_weights(dims, seed=0, device="cuda", gen_device="cuda")  # what tools/hunyuan_bench.py builds
    lat, text_states, mask, ts2 = synth.synth_hunyuan_inputs(dims, ts, valid_text=200)
    inputs = {"text_encoder_output": {"text_encoder_1_text_states": text_states.cuda(), "text_encoder_1_attention_mask": mask.cuda(), "text_encoder_2_text_states": ts2.cuda()}}

    def hip_forward(d):
        cfg = hy.default_config(d, infer_steps=50)
        model = hy.HunyuanModel(cfg, wd)
        sch = hy.HunyuanScheduler(cfg)
        sch.prepare(lat)
        model.set_scheduler(sch)
        sch.step_pre(1)
        model.infer(inputs)
        out = sch.noise_pred.float().cpu()
        meta = (sch.timesteps[1].cpu(), sch.guidance.cpu(), tuple(int(x) for x in (ts[2], ts[3] // 2, ts[4] // 2)))
        del model
        torch.cuda.empty_cache()
        return out, meta

    def oracle_forward(d, dtype):
        cos, sin = H.rope_tables(list(grid), dtype=dtype)
        w = wd if dtype == torch.bfloat16 else {k: v.to(dtype) for k, v in wd.items() if not k.startswith(("double_blocks.", "single_blocks.")) or
                                                   int(k.split(".")[1]) < (d["double_blocks"] if k.startswith("double") else d["single_blocks"])}
        with torch.no_grad():
            out = H.forward(w, d, lat.to(torch.bfloat16).to(dtype).cuda(), t, guidance, text_states.to(dtype).cuda(), mask.cuda(), ts2.to(dtype).cuda(), (cos.cuda(), sin.cuda()))
        out = out.float().cpu()
        del w
        torch.cuda.empty_cache()
        return out

    got, (t, guidance, grid) = hip_forward(dims)
    assert grid[0] * grid[1] * grid[2] == 118800 and torch.isfinite(got).all()
    ref = oracle_forward(dims, torch.bfloat16)
    e_full = rel_l2(got, ref)
    d_r = dict(dims, double_blocks=4, single_blocks=8)
    got_r, _ = hip_forward(d_r)
    ref_r = oracle_forward(d_r, torch.bfloat16)
    with O.truth_precision(torch.float32):
        tru_r = oracle_forward(d_r, torch.float32)
    e_hip, e_ref = rel_l2(got_r, tru_r), rel_l2(ref_r, tru_r)
    record("Hunyuan-13B forward at 118800 + 256 tokens (config #5)", full_depth_hip_vs_oracle_bf16_on_gpu=e_full, depth12_err_hip_vs_fp32=e_hip, depth12_err_oracle_vs_fp32=e_ref,
           depth12_hip_vs_oracle=rel_l2(got_r, ref_r))
    assert got.shape == ref.shape and e_full <= 2.5e-2, f"60-block forward: {e_full:.3e} from the oracle's bf16 evaluation"
    assert e_hip <= 1.5 * e_ref + 1e-4, f"12-block forward: {e_hip:.3e} from the fp32 graph, the oracle's bf16 evaluation {e_ref:.3e}"
