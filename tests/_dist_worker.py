"""world_size-2 gloo worker for tests/test_dist_cpu.py (CPU): checks the Ulysses exchange logic of
lightx2v_amd.ulysses against single-process results, with the CPU oracle as the attention function."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    dist.init_process_group("gloo")
    r, n = dist.get_rank(), dist.get_world_size()
    from lightx2v_amd import ulysses
    from oracle import wan_oracle as O

    gen = torch.Generator().manual_seed(0)
    S, H, d = 96, 2 * n, 128  # heads divide by the world size (2 -> 4 heads, 3 -> 6)
    q, k, v = (torch.randn(S, H * d, generator=gen).to(torch.bfloat16) for _ in range(3))
    full = O.sdpa(q.view(S, H, d), k.view(S, H, d), v.view(S, H, d))
    sl = slice(r * S // n, (r + 1) * S // n)

    # exchange primitives are exact permutations
    qh = ulysses.seq2head(q[sl].contiguous())
    assert torch.equal(qh, q[:, r * H * d // n : (r + 1) * H * d // n])
    back = ulysses.head2seq(qh)
    assert torch.equal(back, q[sl])

    attn = ulysses.UlyssesAttention(attn_fn=lambda a, b, c, h, hd: O.sdpa(a.view(-1, h, hd), b.view(-1, h, hd), c.view(-1, h, hd)), overlap=False)
    out = attn(q[sl].contiguous(), k[sl].contiguous(), v[sl].contiguous(), H, d)
    assert torch.equal(out, full[sl]), (out.float() - full[sl].float()).abs().max()

    # the fused driver's copy-free path: blocked exchange buffers in, blocked receive buffer out (no transposing copies inside)
    s_loc, hdn = S // n, H * d // n
    b = attn.buffers(s_loc, H * d, torch.bfloat16, "cpu")
    for name, src in (("sq", q), ("sk", k), ("sv", v)):
        b[name].copy_(src[sl].view(s_loc, n, hdn).transpose(0, 1))  # what the GEMM epilogue / norm+RoPE kernel write on the GPU
    before = attn.copies
    ro = attn.attend_blocked(b, H, d)
    assert attn.copies == before, "the blocked path must not fall back to the transposing entry"
    assert ro.shape == (n, s_loc, hdn)
    assert torch.equal(ro.transpose(0, 1).reshape(s_loc, H * d), full[sl]), "blocked Ulysses attention differs from the single-process result"
    assert torch.equal(ro.transpose(0, 1).reshape(s_loc, H * d), out), "blocked and row-major entries must agree bit for bit"

    # shard / gather around the block stack, including zero padding when S % N != 0
    x = torch.randn(S + 1, 8, generator=gen)
    xs = ulysses.pre_process(x)
    padded = -(-(S + 1) // n) * n
    assert xs.shape[0] == padded // n
    g = ulysses.post_process(xs)
    assert torch.equal(g[: S + 1], x) and torch.equal(g[S + 1 :], torch.zeros(padded - S - 1, 8))

    # against the reference's own functions in this real 2-process run (authoring container only: /root/reference is absent elsewhere):
    # comm/all2all.py:6-89 (seq<->head all-to-all layouts) and utils/wan/processor.py:9-37 (shard / gather)
    from oracle import ref_import

    if ref_import.reference_available():
        ref_import.patch_and_import()
        from lightx2v.attentions.distributed.comm.all2all import all2all_head2seq, all2all_seq2head
        from lightx2v.attentions.distributed.utils.wan import processor as ref_proc

        mine = ulysses.seq2head(q[sl].contiguous())
        theirs = all2all_seq2head(q[sl].contiguous().view(-1, H, d))
        assert torch.equal(mine.view(S, H // n, d), theirs), "seq2head differs from the reference's all2all_seq2head"
        o_h = torch.randn(S, (H // n) * d, generator=torch.Generator().manual_seed(5 + r)).to(torch.bfloat16)
        assert torch.equal(ulysses.head2seq(o_h).view(-1, H, d), all2all_head2seq(o_h.view(S, H // n, d))), "head2seq differs from all2all_head2seq"
        xr = torch.randn(S + 1, 8, generator=torch.Generator().manual_seed(3))
        a, b = ulysses.pre_process(xr), ref_proc.pre_process(xr)
        assert torch.equal(a, b)
        assert torch.equal(ulysses.post_process(a), ref_proc.post_process(b.contiguous()))
        if r == 0:
            print("REFERENCE_EXCHANGE_OK")

    hunyuan_checks(r, n, ulysses, O)
    teacache_decision_is_rank_invariant(r, n, ulysses)

    # distributed oracle forward == single-process oracle forward (pins compute_freqs_dist + sharding; S % N == 0)
    from lightx2v_amd import synth

    dims = synth.WAN_DIMS["wan-tiny"]
    if dims["num_heads"] % n:  # the tiny model has 2 heads: the forward comparison runs at world size 2 only
        dist.barrier()
        if r == 0:
            print("DIST_OK")
        dist.destroy_process_group()
        return
    wd = synth.synth_wan_weights(dims, seed=0)
    lat, ctx, _ = synth.synth_inputs(dims, (16, 3, 8, 8))
    t = torch.tensor(500)
    embed, grid, xfull, embed0, s, context = O.wan_pre_infer(wd, dims, lat.to(torch.bfloat16), t, ctx)
    freqs = O.rope_freqs_table(128)
    ref = xfull.clone()
    for i in range(dims["num_layers"]):
        ref = O.wan_block(wd, i, dims, grid, ref, embed0, freqs, context)
    xs = ulysses.pre_process(xfull)
    sp_attn = lambda a, b, c: attn(a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1), c.reshape(c.shape[0], -1), dims["num_heads"], 128)
    for i in range(dims["num_layers"]):
        xs = O.wan_block(wd, i, dims, grid, xs, embed0, freqs, context, sp=(r, n, sp_attn))
    got = ulysses.post_process(xs)
    # sharded GEMMs run with a different M, so oneDNN may block/sum differently: agreement to bf16 rounding, not bits
    rel = ((got.float() - ref.float()).norm() / ref.float().norm()).item()
    assert rel < 1e-2 and (got.float() - ref.float()).abs().max() <= 2 ** -3, rel
    dist.barrier()
    if r == 0:
        print("DIST_OK")
    dist.destroy_process_group()


def teacache_decision_is_rank_invariant(r, n, ulysses):
    """HunyuanTransformerInferTeaCaching under Ulysses: the relative-L1 change that decides whether the next step skips the block
    stack is reduced over the sequence-parallel group, so every rank computes the SAME number from different shards (a rank-local decision
    could diverge at the threshold: one rank skips the all-to-alls, the others hang in them) — and it equals the unsharded tensors' ratio."""
    from lightx2v_amd import hunyuan as hy

    g = torch.Generator().manual_seed(21)
    full_prev = torch.randn(8 * n, 64, generator=g).to(torch.bfloat16)
    full_now = (full_prev.float() + 0.05 * torch.randn(8 * n, 64, generator=g)).to(torch.bfloat16)
    tc = hy.HunyuanTransformerInferTeaCaching.__new__(hy.HunyuanTransformerInferTeaCaching)
    tc.parallel_attention = None
    assert tc._sharded_rel_l1(full_now, full_prev) is None  # not sharded: the single-GPU arithmetic stays as it is
    tc.parallel_attention = ulysses.UlyssesHunyuanAttention(overlap=False)
    sl = slice(8 * r, 8 * (r + 1))
    rel = tc._sharded_rel_l1(full_now[sl], full_prev[sl])
    want = ((full_now - full_prev).abs().float().sum().double() / full_prev.abs().float().sum().double()).item()
    assert abs(rel - want) <= 1e-6 * want, (rel, want)
    gathered = [None] * n
    dist.all_gather_object(gathered, rel)
    assert all(x == gathered[0] for x in gathered), gathered


def hunyuan_checks(r, n, ulysses, O):
    """HunyuanVideo's joint image+text attention under Ulysses and the latent / RoPE-table sharding around the model:
    against the single-process result, and — where /root/reference exists — against the reference's own ulysses_attn
    (attentions/distributed/ulysses/attn.py:7-91) and hunyuan processor (utils/hunyuan/processor.py:5-77) in this same 2-process run."""
    gen = torch.Generator().manual_seed(11)
    H, d = 2 * n, 128
    t_, hh, ww = (2, 4, 6) if n <= 3 else (2, n, 3)  # token grid (latent 8 x 12): split along h when h % n == 0 (world 2), else along w (world 3); world 8: h = 8
    axis = 1 if hh % n == 0 else 2
    n_img_full, n_txt = t_ * hh * ww, 10
    qf, kf, vf = (torch.randn(n_img_full + n_txt, H * d, generator=gen).to(torch.bfloat16) for _ in range(3))

    # each rank holds its image rows (contiguous slab of the h axis of every frame) and all text rows
    idx = torch.arange(n_img_full).view(t_, hh, ww)
    mine = torch.chunk(idx, n, dim=axis)[r].reshape(-1)
    order = torch.cat([torch.chunk(idx, n, dim=axis)[j].reshape(-1) for j in range(n)])  # rank-major image order seen by the attention
    rows = torch.cat([mine, n_img_full + torch.arange(n_txt)])
    q, k, v = qf[rows].contiguous(), kf[rows].contiguous(), vf[rows].contiguous()
    n_img = mine.numel()

    def sdpa2d(a, b, c, h, oo):
        dd = a.shape[1] // h
        oo.copy_(O.sdpa(a.reshape(-1, h, dd), b.reshape(-1, h, dd), c.reshape(-1, h, dd)))

    ua = ulysses.UlyssesHunyuanAttention(attn_fn=sdpa2d)
    out = torch.empty_like(q)
    ua(q, k, v, n_img, (n_txt, n_txt), H, out)
    # single process: attention over [all image tokens in the rank-major order ; text] with all heads
    allrows = torch.cat([order, n_img_full + torch.arange(n_txt)])
    full = O.sdpa(qf[allrows].view(-1, H, d), kf[allrows].view(-1, H, d), vf[allrows].view(-1, H, d))
    inv = {int(g): i for i, g in enumerate(allrows.tolist())}
    want = full[[inv[int(g)] for g in rows.tolist()]]
    # attention is permutation-equivariant in the keys up to fp summation order: bf16-rounding agreement, not bits
    assert (out.float() - want.float()).abs().max() <= 2 ** -6, (out.float() - want.float()).abs().max()

    # the fused driver's copy-free path over the same collectives (CPU tensors; on the GPU the buffers are written by the GEMM / norm kernels):
    # head-blocked send buffers in, K-blocked projection inputs out — must equal the row-major entry bit for bit, both text-mask cases
    hdn = H * d // n
    for n_valid in (n_txt, n_txt - 3):
        ref_out = torch.empty_like(q)
        ua(q, k, v, n_img, (n_valid, n_txt), H, ref_out)
        bufs = ua.buffers(n_img, n_txt, H * d, 4 * hdn, torch.bfloat16, "cpu")
        for i, src in enumerate((q, k, v)):
            bufs["snd"][i].copy_(src[:n_img].view(n_img, n, hdn).transpose(0, 1))
        before = ua.copies
        a_img, a_txt = ua.attend_blocked(bufs, (q[n_img:], k[n_img:], v[n_img:]), (n_valid, n_txt), H)
        assert ua.copies == before
        assert a_img.shape == (n + 4, n_img, hdn) and a_txt.shape == (n + 4, n_txt, hdn)
        assert torch.equal(a_img[:n].transpose(0, 1).reshape(n_img, H * d), ref_out[:n_img]), "blocked Hunyuan exchange: image rows differ"
        assert torch.equal(a_txt[:n].transpose(0, 1).reshape(n_txt, H * d), ref_out[n_img:]), "blocked Hunyuan exchange: text rows differ"

    # masked text: rows beyond n_valid attend among themselves only (two segments, as the single-GPU path)
    out2 = torch.empty_like(q)
    ua(q, k, v, n_img, (n_txt - 3, n_txt), H, out2)
    assert torch.isfinite(out2.float()).all()
    seg1 = torch.cat([order, n_img_full + torch.arange(n_txt - 3)])
    f1 = O.sdpa(qf[seg1].view(-1, H, d), kf[seg1].view(-1, H, d), vf[seg1].view(-1, H, d))
    inv1 = {int(g): i for i, g in enumerate(seg1.tolist())}
    want1 = f1[[inv1[int(g)] for g in rows[: n_img + n_txt - 3].tolist()]]
    assert (out2[: n_img + n_txt - 3].float() - want1.float()).abs().max() <= 2 ** -6
    pad = n_img_full + torch.arange(n_txt - 3, n_txt)
    f2 = O.sdpa(qf[pad].view(-1, H, d), kf[pad].view(-1, H, d), vf[pad].view(-1, H, d))
    assert (out2[n_img + n_txt - 3 :].float() - f2.float()).abs().max() <= 2 ** -6

    # latent / RoPE-table sharding and the gather of the noise prediction
    lat = torch.randn(1, 16, t_, 2 * hh, 2 * ww, generator=gen)
    cos, sin = torch.randn(n_img_full, d, generator=gen), torch.randn(n_img_full, d, generator=gen)
    l2, c2, s2, split_dim = ulysses.hunyuan_pre_process(lat, cos, sin)
    assert split_dim == axis - 3 and torch.equal(l2, torch.chunk(lat, n, dim=split_dim)[r]) and torch.equal(c2, cos[mine]) and torch.equal(s2, sin[mine])
    assert torch.equal(ulysses.hunyuan_post_process(l2, split_dim), lat)
    w3 = ww if n <= 3 else n  # world 8: a w axis the group divides
    latw = torch.randn(1, 16, t_, 2 * 3, 2 * w3, generator=gen)  # h = 3: the other axis than above at either world size
    l3, c3, s3, sd3 = ulysses.hunyuan_pre_process(latw, torch.randn(t_ * 3 * w3, d, generator=gen), torch.randn(t_ * 3 * w3, d, generator=gen))
    assert sd3 == (-2 if 3 % n == 0 else -1) and torch.equal(ulysses.hunyuan_post_process(l3, sd3), latw)

    from oracle import ref_import

    if ref_import.reference_available():
        ref_import.patch_and_import()
        from lightx2v.attentions.distributed.ulysses import attn as ref_attn
        from lightx2v.attentions.distributed.utils.hunyuan import processor as ref_proc

        # the reference's dispatcher reaches flash-attn (absent here): one unmasked segment [0, s] is plain SDPA, which is what stands in
        ref_attn.attention = lambda attention_type, q, k, v, **kw: O.sdpa(q, k, v)
        got_ref = ref_attn.ulysses_attn(q.view(-1, H, d), k.view(-1, H, d), v.view(-1, H, d), img_qkv_len=n_img,
                                        cu_seqlens_qkv=torch.tensor([0, n_img + n_txt], dtype=torch.int32), attention_type="flash_attn2")
        assert torch.equal(out, got_ref), "UlyssesHunyuanAttention differs from the reference's ulysses_attn"
        rl, rc, rs, rsd = ref_proc.pre_process(lat, cos, sin)
        assert rsd == split_dim and torch.equal(rl, l2) and torch.equal(rc, c2) and torch.equal(rs, s2)
        assert torch.equal(ref_proc.post_process(rl.contiguous(), rsd), ulysses.hunyuan_post_process(l2, split_dim))
        rl3, _, _, rsd3 = ref_proc.pre_process(latw, torch.zeros(t_ * 3 * w3, d), torch.zeros(t_ * 3 * w3, d))
        assert rsd3 == sd3 and torch.equal(rl3, l3)
        if r == 0:
            print("REFERENCE_HUNYUAN_EXCHANGE_OK")


if __name__ == "__main__":
    main()
