"""The operator objects behave like the reference classes they stand in for, at the points VERDICT r1 found loose:
multi-sequence attention behind `hip_flash` (attn_weight.py:76-97 with HunyuanVideo's three-entry cu_seqlens,
hunyuan/infer/pre_infer.py:50-56), the Conv3d layout of the patch embedding (common/ops/conv/conv3d.py:40-50),
`row_slice` on the MM operators (all three weight formats) and cache invalidation when weights change."""
import math

import pytest
import torch

from tests.util import assert_bf16_close

pytestmark = pytest.mark.gpu


def test_hip_flash_multi_segment_matches_varlen_oracle():
    from lightx2v_amd import ops, registry
    from oracle import hunyuan_oracle as H

    gen = torch.Generator().manual_seed(1)
    n_img, n_valid, n_txt, heads = 700, 37, 64, 3
    total = n_img + n_txt
    q, k, v = (torch.randn(total, heads, 128, generator=gen).to(torch.bfloat16) for _ in range(3))
    cu = torch.tensor([0, n_img + n_valid, total], dtype=torch.int32)
    ref = H.varlen_attention(q, k, v, cu)
    attn = registry.ATTN_WEIGHT_REGISTER["hip_flash"]()
    for cu_in in (cu, cu.cuda(), cu.tolist()):
        got = attn.apply(q.cuda(), k.cuda(), v.cuda(), cu_seqlens_q=cu_in, cu_seqlens_kv=cu_in, max_seqlen_q=total, max_seqlen_kv=total)
        assert got.shape == (total, heads * 128)
        assert_bf16_close(got, ref.reshape(total, -1), ulps=0.128, atol=4e-3, name="hip_flash, two segments")
    # different q / kv boundaries (cross-attention style) and a padded tail of query rows that belong to no sequence
    cq, ck = [0, 300, 600], [0, 50, 64]
    kk, vv = k[:64], v[:64]
    got = ops.hip_flash(q.cuda(), kk.cuda(), vv.cuda(), cq, ck, max_seqlen_q=total)
    from oracle import wan_oracle as O

    for (qa, qb), (ka, kb) in (((0, 300), (0, 50)), ((300, 600), (50, 64))):
        assert_bf16_close(got[qa:qb], O.sdpa(q[qa:qb], kk[ka:kb], vv[ka:kb]), ulps=0.128, atol=4e-3, name=f"segment {qa}:{qb}")
    assert not got[600:].any(), "rows outside every sequence must be zero"
    from lightx2v_amd.lib import X2VError

    with pytest.raises(X2VError):
        ops.hip_flash(q.cuda(), k.cuda(), v.cuda(), [0, total + 1], [0, total])
    with pytest.raises(X2VError):
        ops.hip_flash(q.cuda(), k.cuda(), v.cuda(), [0, 10, total], [0, total])


def test_hip_flash_fresh_cu_tensors_recycled_at_one_address():
    """The reference's op-by-op loop builds fresh device cu_seqlens per attention call (wan/infer/transformer_infer.py:73-77
    `_calculate_q_k_len`: cat + cumsum, version 0, two entries) and drops them, so the caching allocator hands the SAME address to the next
    one with other contents: self-attention [0, S], cross-attention keys [0, 512], another request's S.  The boundary cache must never serve
    one tensor's boundaries for another (a (data_ptr, version, numel) key did)."""
    from lightx2v_amd import ops
    from oracle import wan_oracle as O

    gen = torch.Generator().manual_seed(2)
    H = 2
    q, k, v = (torch.randn(640, H, 128, generator=gen).to(torch.bfloat16) for _ in range(3))
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    seen = set()
    for n_q, n_k in ((640, 640), (640, 96), (300, 300), (640, 33), (128, 640)):
        def cu_of(n):  # as the reference: a fresh tensor each time, freed right after the call
            return torch.cat([torch.zeros(1, dtype=torch.int32, device="cuda"), torch.tensor([n], dtype=torch.int32, device="cuda")]).cumsum(0, dtype=torch.int32)

        cq, ck = cu_of(n_q), cu_of(n_k)
        seen.add((cq.data_ptr(), ck.data_ptr()))
        got = ops.hip_flash(qd[:n_q], kd[:n_k], vd[:n_k], cq, ck, max_seqlen_q=n_q, max_seqlen_kv=n_k)
        del cq, ck
        assert_bf16_close(got, O.sdpa(q[:n_q], k[:n_k], v[:n_k]), ulps=0.128, atol=4e-3, name=f"fresh cu ({n_q}, {n_k})")
    assert len(ops._CU_CACHE) <= 8
    # an in-place edit of a cached tensor is seen through its version counter
    cu = torch.tensor([0, 640], dtype=torch.int32, device="cuda")
    ops.hip_flash(qd, kd, vd, cu, cu)
    cu[1] = 100
    got = ops.hip_flash(qd, kd, vd, cu, cu, max_seqlen_q=640)
    assert_bf16_close(got[:100], O.sdpa(q[:100], k[:100], v[:100]), ulps=0.128, atol=4e-3, name="edited cu")
    assert not got[100:].any()


def test_patch_embedding_returns_the_conv3d_layout():
    from lightx2v_amd import registry

    gen = torch.Generator().manual_seed(2)
    D, C, T, Hh, Ww = 256, 16, 3, 8, 12
    w = (torch.randn(D, C, 1, 2, 2, generator=gen) * 0.1).to(torch.bfloat16)
    b = torch.randn(D, generator=gen).to(torch.bfloat16)
    x = torch.randn(1, C, T, Hh, Ww, generator=gen).to(torch.bfloat16)
    ref = torch.nn.functional.conv3d(x, w, b, stride=(1, 2, 2))
    op = registry.CONV3D_WEIGHT_REGISTER["hip_patch"]("w", "b", stride=(1, 2, 2))
    op.load({"w": w.cuda(), "b": b.cuda()})
    y = op.apply(x.cuda())
    assert y.shape == ref.shape == (1, D, T, Hh // 2, Ww // 2)
    assert_bf16_close(y, ref, ulps=1, atol=2e-3, bad_frac=2e-3, name="patch embedding (Conv3d layout)")
    tok = y.flatten(2).transpose(1, 2)  # the caller's next two ops (pre_infer.py:59)
    assert tok.is_contiguous() and tok.data_ptr() == y.data_ptr(), "flatten(2).transpose(1,2) must land on the GEMM output without a copy"


@pytest.mark.parametrize("mm_type", ["Hip-bf16", "W-fp8-channel-sym-A-fp8-channel-sym-dynamic-Hip", "W-mxfp8-A-mxfp8-dynamic-Hip"])
def test_mm_row_slice_equals_sliced_full_apply(mm_type):
    from lightx2v_amd import lib, registry

    gen = torch.Generator().manual_seed(3)
    M, K, N = 300, 512, 1024
    wd = {"w.weight": (torch.randn(N, K, generator=gen) / math.sqrt(K)).to(torch.bfloat16).cuda(), "w.bias": torch.randn(N, generator=gen).to(torch.bfloat16).cuda()}
    x = torch.randn(M, K, generator=gen).to(torch.bfloat16).cuda()
    op = registry.MM_WEIGHT_REGISTER[mm_type]("w.weight", "w.bias")
    op.set_config({"weight_auto_quant": True})
    op.load(wd)
    full = op.apply(x)
    for sl in (slice(0, 384), slice(384, None)):
        part = op.apply(x, row_slice=sl)
        assert torch.equal(part, full[:, sl]), (mm_type, sl)
        if hasattr(op, "quantize_input"):
            assert torch.equal(op.apply(x, row_slice=sl, quantized=op.quantize_input(x)), part)
    g = op.apply(x, epilogue=lib.EPI_GELU_TANH, row_slice=slice(384, None))
    assert torch.equal(g, op.apply(x, epilogue=lib.EPI_GELU_TANH)[:, 384:])


def test_caches_follow_weight_updates():
    """the step-invariant caches are keyed by weight objects; an in-place weight edit and a re-load must both be noticed."""
    from lightx2v_amd import scheduler, synth, wan

    dims, wl = synth.WAN_DIMS["wan-tiny"], synth.WORKLOADS["wan-tiny"]
    wd = {k: v.cuda() for k, v in synth.synth_wan_weights(dims, seed=0).items()}
    lat, ctx, ctx_null = synth.synth_inputs(dims, wl["target_shape"])
    inputs = {"text_encoder_output": {"context": [c.cuda() for c in ctx], "context_null": [c.cuda() for c in ctx_null]}}
    cfg = wan.default_config(dims, target_shape=wl["target_shape"], target_video_length=wl["frames"], infer_steps=3)
    model = wan.WanModel(cfg, wd)
    sch = scheduler.WanScheduler(cfg, device="cuda")
    sch.prepare(latents=lat)
    model.set_scheduler(sch)
    sch.step_pre(0)
    model.infer(inputs)
    first = sch.noise_pred.clone()
    model.infer(inputs)
    assert torch.equal(sch.noise_pred, first)
    wd["blocks.0.cross_attn.k.weight"].mul_(0.5)  # in place: same object ids, same data_ptr, new version
    model.infer(inputs)
    second = sch.noise_pred.clone()
    assert not torch.equal(second, first), "stale cross-attention K served after an in-place weight update"
    wd2 = dict(wd)
    wd2["text_embedding.2.weight"] = wd["text_embedding.2.weight"] * 0.5
    model._init_weights(wd2)  # the reference's LoRA-switch path
    assert not model.transformer_infer._cross_kv_cache and not model.pre_infer._text_cache
    model.infer(inputs)
    assert not torch.equal(sch.noise_pred, second)
    fresh = wan.WanModel(cfg, wd2)
    fresh.set_scheduler(sch)
    third = sch.noise_pred.clone()
    fresh.infer(inputs)
    assert torch.equal(sch.noise_pred, third), "re-loaded model must equal a freshly built one"


def test_blocked_gemm_and_rope_equal_their_row_major_forms():
    """x2v_gemm_bf16_blocked / x2v_rmsnorm_rope_blocked_bf16 (the Ulysses exchange buffers as kernel operands): N-blocked y, K-blocked x
    and the out-of-place blocked norm+RoPE give exactly the bits of the row-major kernels followed by the reference's transposing copies
    (comm/all2all.py:29-33, 70-75); both GEMM tilings."""
    from lightx2v_amd import lib
    from lightx2v_amd.wan import rope_cos_sin_table

    gen = torch.Generator().manual_seed(4)
    nb = 4
    for M, K, N in ((300, 512, 1024), (4100, 2560, 5120)):  # 128x128 kernel / 256x256 kernel (5120 = 4 blocks of 10 K-tiles)
        x = torch.randn(M, K, generator=gen).to(torch.bfloat16).cuda()
        w = (torch.randn(N, K, generator=gen) / math.sqrt(K)).to(torch.bfloat16).cuda()
        b = torch.randn(N, generator=gen).to(torch.bfloat16).cuda()
        ref = lib.gemm(x, w, b)
        out = torch.full((nb, M + 3, N // nb), 7.0, dtype=torch.bfloat16, device="cuda")[:, 1 : M + 1]  # strided rows inside a poisoned buffer
        lib.gemm(x, w, b, out=out)
        assert torch.equal(out.transpose(0, 1).reshape(M, N), ref), (M, K, N, "N-blocked y")
        xb = x.view(M, nb, K // nb).transpose(0, 1).contiguous()  # [nb, M, K/nb]
        assert torch.equal(lib.gemm(xb, w, b), ref), (M, K, N, "K-blocked x")
        res = torch.randn(M, N, generator=gen).to(torch.bfloat16).cuda()
        gate = torch.randn(N, generator=gen).to(torch.bfloat16).cuda()
        r1, r2 = res.clone(), res.clone()
        lib.gemm(x, w, b, epilogue=lib.EPI_RESIDUAL, resid=r1, gate=gate)
        lib.gemm(xb, w, b, epilogue=lib.EPI_RESIDUAL, resid=r2, gate=gate)
        assert torch.equal(r1, r2), (M, K, N, "K-blocked x, gate-residual epilogue")
        assert torch.equal(lib.gemm(xb, w, b, epilogue=lib.EPI_GELU_TANH), lib.gemm(x, w, b, epilogue=lib.EPI_GELU_TANH))
    # norm + RoPE into blocked outputs: both kernel forms (few rows -> per-row kernel, many rows -> streaming kernel)
    H, grid = 4, (5, 30, 40)
    tab = rope_cos_sin_table(128, "cuda")
    wq, wk = (1 + 0.1 * torch.randn(H * 128, generator=gen)).to(torch.bfloat16).cuda(), (1 + 0.1 * torch.randn(H * 128, generator=gen)).to(torch.bfloat16).cuda()
    for S in (77, 6000):
        q = torch.randn(S, H * 128, generator=gen).to(torch.bfloat16).cuda()
        k = torch.randn(S, H * 128, generator=gen).to(torch.bfloat16).cuda()
        q1, k1 = q.clone(), k.clone()
        lib.rmsnorm_rope_(q1, k1, wq, wk, tab, grid, H, s0=11, q_out_scale=lib.ATTN_PRESCALE)
        qo = torch.full((2, S, H * 64), 7.0, dtype=torch.bfloat16, device="cuda")
        ko = torch.full((2, S, H * 64), 7.0, dtype=torch.bfloat16, device="cuda")
        lib.rmsnorm_rope_blocked(q, k, wq, wk, tab, grid, H, qo, ko, s0=11, q_out_scale=lib.ATTN_PRESCALE)
        assert torch.equal(qo.transpose(0, 1).reshape(S, H * 128), q1) and torch.equal(ko.transpose(0, 1).reshape(S, H * 128), k1), S


def test_layernorm_quant_fp8_is_the_two_kernels_fused():
    """x2v_layernorm_quant_fp8 == x2v_quant_fp8_rowwise(x2v_layernorm_bf16(...)) bit for bit — plain, affine and modulated LayerNorm, ragged
    row widths, both LayerNorm kernel forms on the unfused side (M small / large) — and the fp8 operator class consumes the shared pair."""
    from lightx2v_amd import lib, registry

    gen = torch.Generator().manual_seed(8)
    for M, D in ((37, 1536), (3000, 5120), (64, 3072), (5, 256)):
        x = (torch.randn(M, D, generator=gen) * 1.7 + 0.3).to(torch.bfloat16).cuda()
        x[M // 2] = 0  # an all-zero token: the scale floor 1 / (448 * 512)
        w, b = (1 + 0.1 * torch.randn(D, generator=gen)).to(torch.bfloat16).cuda(), (0.1 * torch.randn(D, generator=gen)).to(torch.bfloat16).cuda()
        sc, sh = (0.2 * torch.randn(1, D, generator=gen)).to(torch.bfloat16).cuda(), (0.2 * torch.randn(1, D, generator=gen)).to(torch.bfloat16).cuda()
        for kw in ({}, {"weight": w, "bias": b}, {"scale": sc, "shift": sh}):
            xq, sx = lib.layernorm_quant_fp8(x, **kw)
            rq, rs = lib.quant_fp8_rowwise(lib.layernorm(x, **kw))
            assert torch.equal(sx, rs) and torch.equal(xq.view(torch.uint8), rq.view(torch.uint8)), (M, D, list(kw))
    op = registry.MM_WEIGHT_REGISTER["W-fp8-channel-sym-A-fp8-channel-sym-dynamic-Hip"]("w.weight", "w.bias")
    op.set_config({"weight_auto_quant": True})
    D, N = 1536, 512
    op.load({"w.weight": (torch.randn(N, D, generator=gen) / math.sqrt(D)).to(torch.bfloat16).cuda(), "w.bias": torch.randn(N, generator=gen).to(torch.bfloat16).cuda()})
    x = torch.randn(100, D, generator=gen).to(torch.bfloat16).cuda()
    pair = op.layernorm_quantize(x, scale=sc[:, :D] if sc.shape[1] >= D else None, shift=sh[:, :D] if sh.shape[1] >= D else None)
    assert torch.equal(op.apply(None, quantized=pair), op.apply(lib.layernorm(x, scale=sc[:, :D] if sc.shape[1] >= D else None, shift=sh[:, :D] if sh.shape[1] >= D else None)))
