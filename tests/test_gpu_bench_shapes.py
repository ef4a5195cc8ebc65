"""Parity of the kernels and dimensions that bench.py / the BASELINE configs really execute, against the CPU oracle.

The op-level tests in test_gpu_ops.py use small shapes, which the GEMM dispatcher sends to the 128x128 kernel and where attention
runs a handful of key tiles.  Here the shapes are chosen so that
  * `lib.gemm` / `lib.gemm_fp8` take the 256x256 kernels (bf16: single-stream, fp8: ping-pong) BY THEIR OWN DISPATCH (asserted through
    x2v_gemm_kernel_choice, and by bit-equality with the forced variant) — the Wan-14B projections 5120→5120, 5120→13824, 13824→5120, ragged M included,
    all four epilogues;
  * the ping-pong attention kernel on pre-transposed V with a pre-scaled q (what the fused block drivers launch) sees >= 8192 keys
    through strided fused-QKV views, against torch SDPA and exact fp32 attention;
  * one Wan block runs at 14B dimensions (D 5120, 40 heads, F 13824; configs #3/#4) and one at config #2's sequence length
    (1.3B, S = 20 280), one HunyuanVideo double + single block at D 3072 / 24 heads (config #5) — each vs the oracle.
Reference call sites: common/ops/mm/mm_weight.py:81-88, common/ops/attn/attn_weight.py:229-239,
models/networks/wan/infer/transformer_infer.py:289-508, models/networks/hunyuan/infer/transformer_infer.py:81-384.

Tolerances are those of test_gpu_ops.py / test_gpu_model.py (bf16: 1 ulp = 2^-7 relative):
GEMM <= 1 ulp + atol on all but <= 2e-3 of the elements; attention at >= 8192 keys: the reference's own acceptance
allclose(rtol=1e-3, atol=1e-3) (attentions/distributed/ring/tests/test.py:97) for every kernel variant, plus the triangle bound against
fp32 attention (our error <= 1.5x the reference CPU kernel's own); block outputs relative L2 <= 1e-2.
"""
import math

import pytest
import torch

from tests.util import assert_bf16_close, assert_rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from lightx2v_amd import lib as L

    L.init()
    return L


def dev(t):
    return t.cuda()


# ------------------------------------------------------------------------------------------------ GEMM, 256x256 kernel
WAN14B_GEMMS = [(4096, 5120, 5120), (4096, 5120, 13824), (4100, 13824, 5120), (4100, 5120, 5120)]  # (M, K, N); 4100 = ragged M


@pytest.mark.parametrize("M,K,N", WAN14B_GEMMS)
def test_gemm256_bf16_natural_dispatch_vs_oracle(lib, M, K, N):
    from oracle import wan_oracle as O

    assert lib.gemm_kernel_choice(M, N, K) == 3, "shape must take the 256x256 single-stream kernel by the dispatcher's own rule"
    gen = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=gen).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=gen) / math.sqrt(K)).to(torch.bfloat16)
    b = (torch.randn(N, generator=gen) * 0.1).to(torch.bfloat16)
    xd, wd_, bd = dev(x), dev(w), dev(b)
    ref = O.mm(x, w, b)
    got = lib.gemm(xd, wd_, bd)
    assert_bf16_close(got, ref, ulps=1, atol=2e-3, bad_frac=1e-3, name="gemm256")
    assert torch.equal(got, lib.gemm(xd, wd_, bd, variant=3)), "variant 0 did not run the 256x256 kernel"
    assert torch.equal(got, lib.gemm(xd, wd_, bd, variant=2)), "the two 256x256 kernels must agree bit for bit"
    assert_bf16_close(lib.gemm(xd, wd_), O.mm(x, w), ulps=1, atol=2e-3, bad_frac=1e-3, name="gemm256 no bias")
    # fused epilogues against the reference's separate ops (transformer_infer.py:402,468,488-503; pre_infer.py:74-76)
    ref_g = torch.nn.functional.gelu(ref, approximate="tanh")
    assert_bf16_close(lib.gemm(xd, wd_, bd, epilogue=lib.EPI_GELU_TANH), ref_g, ulps=1, atol=2e-3, bad_frac=2e-3, name="gemm256+gelu")
    assert_bf16_close(lib.gemm(xd, wd_, bd, epilogue=lib.EPI_SILU), torch.nn.functional.silu(ref), ulps=1, atol=2e-3, bad_frac=2e-3, name="gemm256+silu")
    res = torch.randn(M, N, generator=gen).to(torch.bfloat16)
    gate = (torch.randn(1, N, generator=gen) * 0.5).to(torch.bfloat16)
    for g in (gate, None):
        ref_r = res.clone()
        ref_r.add_(ref * g.squeeze(0) if g is not None else ref)
        r = dev(res).clone()
        out = lib.gemm(xd, wd_, bd, epilogue=lib.EPI_RESIDUAL, resid=r, gate=None if g is None else dev(g))
        assert out.data_ptr() == r.data_ptr()
        assert_bf16_close(r, ref_r, ulps=1, atol=6e-3, bad_frac=2e-3, name=f"gemm256+residual(gate={g is not None})")
    # the two tilings agree to summation-order effects (both compared with the oracle above; this catches a tile-local defect)
    assert_bf16_close(lib.gemm(xd, wd_, bd, variant=1), got.cpu(), ulps=1, atol=2e-3, bad_frac=1e-3, name="128^2 vs 256^2")


@pytest.mark.parametrize("M,K,N", [(4096, 5120, 5120), (4100, 5120, 13824), (300, 64, 264), (257, 128, 520), (1000, 192, 256), (515, 320, 8), (2049, 1536, 1544)])
def test_gemm256_single_stream_kernel_is_bit_equal_to_ping_pong(lib, M, K, N):
    """The two 256x256 kernels (variant 2: two waves per SIMD in ping-pong, variant 3: one software-pipelined wave per SIMD) run the same
    16x16x32 MFMA over the same k order, so every output must be EQUAL — all epilogues, ragged M and N, K loops of 1..216 tiles (the
    single-stream kernel's prologue / tail paths), K-blocked x and N-blocked y (the Ulysses exchange buffers)."""
    gen = torch.Generator().manual_seed(M * 7 + K + N)
    x = dev(torch.randn(M, K, generator=gen).to(torch.bfloat16))
    w = dev((torch.randn(N, K, generator=gen) / math.sqrt(K)).to(torch.bfloat16))
    b = dev((torch.randn(N, generator=gen) * 0.1).to(torch.bfloat16))
    for epi in (lib.EPI_NONE, lib.EPI_GELU_TANH, lib.EPI_SILU):
        a2, a3 = lib.gemm(x, w, b, epilogue=epi, variant=2), lib.gemm(x, w, b, epilogue=epi, variant=3)
        assert torch.equal(a2, a3), f"epilogue {epi}: max |d| = {(a2.float() - a3.float()).abs().max().item():.3e}"
    assert torch.equal(lib.gemm(x, w, None, variant=2), lib.gemm(x, w, None, variant=3))
    res = dev(torch.randn(M, N, generator=gen).to(torch.bfloat16))
    gate = dev((torch.randn(1, N, generator=gen) * 0.5).to(torch.bfloat16))
    for g in (gate, None):
        r2, r3 = res.clone(), res.clone()
        lib.gemm(x, w, b, epilogue=lib.EPI_RESIDUAL, resid=r2, gate=g, variant=2)
        lib.gemm(x, w, b, epilogue=lib.EPI_RESIDUAL, resid=r3, gate=g, variant=3)
        assert torch.equal(r2, r3)
    assert torch.isfinite(lib.gemm(x, w, b, variant=3).float()).all()


@pytest.mark.parametrize("M,K,H", [(4096, 5120, 40), (4100, 5120, 40), (3001, 1536, 12), (300, 512, 4)])
def test_v_projection_with_vt_epilogue_equals_gemm_then_transpose(lib, M, K, H):
    """x2v_gemm_bf16_vt (the v projection writing V^T [H, ceil(M/64), 128, 64] from the GEMM epilogue) against x2v_gemm_bf16 followed by
    x2v_transpose_heads_bf16: EQUAL, including the zero fill of the tokens between M and the next multiple of 64 (ragged M) and the small
    shape where the wrapper runs the two-kernel sequence; and the attention kernel fed by it equals the one fed by the transposed copy."""
    N = H * 128
    gen = torch.Generator().manual_seed(M + H)
    x = dev(torch.randn(M, K, generator=gen).to(torch.bfloat16))
    w = dev((torch.randn(N, K, generator=gen) / math.sqrt(K)).to(torch.bfloat16))
    b = dev((torch.randn(N, generator=gen) * 0.1).to(torch.bfloat16))
    v = lib.gemm(x, w, b)
    ref = lib.transpose_heads(v, H)
    got = lib.gemm_vt(x, w, b, H)
    assert got.shape == ref.shape and torch.equal(got, ref), f"max |d| = {(got.float() - ref.float()).abs().max().item():.3e}"
    assert torch.equal(lib.gemm_vt(x, w, None, H), lib.transpose_heads(lib.gemm(x, w), H))
    if M <= 4096:
        q, k = (dev(torch.randn(M, N, generator=gen).to(torch.bfloat16)) for _ in range(2))
        a1 = lib.attention(q, k, v, H, variant=lib.ATTN_FAST, vt=ref)
        a2 = lib.attention(q, k, None, H, variant=lib.ATTN_FAST, vt=got)
        assert torch.equal(a1, a2)


@pytest.mark.parametrize("M,K,N", [(4096, 5120, 5120), (4100, 5120, 13824), (4096, 13824, 5120)])
def test_gemm256_fp8_natural_dispatch_vs_oracle(lib, M, K, N):
    """Per-token x per-channel w8a8 (mm_weight.py:236-245,310-318) on the 256x256 kernel's fp8 mode — the kernel config #4 runs."""
    from oracle import wan_oracle as O

    assert lib.gemm_kernel_choice(M, N, K, fp8=True) == 2
    gen = torch.Generator().manual_seed(M + K + N + 8)
    x = torch.randn(M, K, generator=gen).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=gen) / math.sqrt(K)).to(torch.bfloat16)
    b = (torch.randn(N, generator=gen) * 0.1).to(torch.bfloat16)
    wq, sw = O.quant_fp8_weight_per_channel(w)
    xq_ref, sx_ref = O.quant_fp8_per_token(x)
    xq, sx = lib.quant_fp8_rowwise(dev(x))
    assert torch.allclose(sx.cpu(), sx_ref, rtol=1e-6, atol=0)
    # the GEMM is compared on the ORACLE's codes so that a quantiser tie cannot hide in (or leak into) the GEMM tolerance
    xq_o, sx_o = dev(xq_ref), dev(sx_ref)
    ref = O.mm_fp8(x, wq, sw, b)
    got = lib.gemm_fp8(xq_o, sx_o, dev(wq), dev(sw), dev(b))
    assert_bf16_close(got, ref, ulps=1, atol=4e-3, bad_frac=2e-3, name="fp8 gemm256")
    assert torch.equal(got, lib.gemm_fp8(xq_o, sx_o, dev(wq), dev(sw), dev(b), variant=2))
    ref_g = torch.nn.functional.gelu(ref, approximate="tanh")
    assert_bf16_close(lib.gemm_fp8(xq_o, sx_o, dev(wq), dev(sw), dev(b), epilogue=lib.EPI_GELU_TANH), ref_g, ulps=1, atol=4e-3, bad_frac=2e-3, name="fp8 gemm256+gelu")
    res = torch.randn(M, N, generator=gen).to(torch.bfloat16)
    gate = (torch.randn(1, N, generator=gen) * 0.5).to(torch.bfloat16)
    ref_r = res.clone()
    ref_r.add_(ref * gate.squeeze(0))
    r = dev(res).clone()
    lib.gemm_fp8(xq_o, sx_o, dev(wq), dev(sw), dev(b), epilogue=lib.EPI_RESIDUAL, resid=r, gate=dev(gate))
    assert_bf16_close(r, ref_r, ulps=1, atol=8e-3, bad_frac=2e-3, name="fp8 gemm256+gate-residual")
    mism = (xq.cpu().view(torch.uint8) != xq_ref.view(torch.uint8)).float().mean().item()
    assert mism <= 1e-3, f"{mism} of e4m3 codes differ"


# ------------------------------------------------------------------------------------------------ attention, bench variant
@pytest.mark.parametrize("Sq,Sk,H", [(1024, 8192, 2), (777, 9001, 3), (2048, 12352, 1)])
def test_bench_attention_variant_vs_oracle_long_keys(lib, Sq, Sk, H):
    """ATTN_FAST | ATTN_Q_PRESCALED (ping-pong kernel, V^T operand, q carrying scale*log2e) with >= 128 key tiles, q/k/v as strided views
    of one fused-QKV buffer — vs torch SDPA (attn_weight.py:229-239) and fp32 attention."""
    from oracle import wan_oracle as O

    gen = torch.Generator().manual_seed(Sq + Sk + H)
    S = max(Sq, Sk)
    qkv = torch.randn(S, 3 * H * 128, generator=gen).to(torch.bfloat16)
    D = H * 128
    q, k, v = qkv[:Sq, :D], qkv[:Sk, D : 2 * D], qkv[:Sk, 2 * D :]
    ref = O.sdpa(q.reshape(Sq, H, 128), k.reshape(Sk, H, 128), v.reshape(Sk, H, 128))
    f32 = O.attention_fp32(q.reshape(Sq, H, 128), k.reshape(Sk, H, 128), v.reshape(Sk, H, 128))
    e_ref = (ref.float() - f32).abs().max().item()
    d = dev(qkv)
    qd, kd, vd = d[:Sq, :D], d[:Sk, D : 2 * D], d[:Sk, 2 * D :]
    # what the block driver does: the producer folds scale*log2(e) into q inside q's one rounding (x2v_rmsnorm_rope_scaled_bf16)
    q_pre = (qd.float() * lib.ATTN_PRESCALE).to(torch.bfloat16)
    # With >= 8192 keys |o| <~ 0.1, so the bf16 rounding of the output no longer needs its own allowance: the tolerance here is the
    # reference's own acceptance, allclose(rtol=1e-3, atol=1e-3) (attentions/distributed/ring/tests/test.py:97), for every kernel
    rtol_ulps, atol = 1e-3 / 0.0078125, 1e-3
    got = lib.attention(q_pre, kd, vd, H, variant=lib.ATTN_FAST | lib.ATTN_Q_PRESCALED)
    assert_bf16_close(got, ref, ulps=rtol_ulps, atol=atol, name="ping-pong prescaled vs torch_sdpa")
    e_ours = (got.float().cpu() - f32).abs().max().item()
    assert e_ours <= 1.5 * e_ref + 2e-4, (e_ours, e_ref)
    # the kernel folding the scale itself (one more rounding of q) and the default entry, same inputs
    got2 = lib.attention(qd, kd, vd, H, variant=lib.ATTN_FAST)
    assert_bf16_close(got2, ref, ulps=rtol_ulps, atol=atol, name="ping-pong vs torch_sdpa")
    got0 = lib.attention(qd, kd, vd, H)
    assert_bf16_close(got0, ref, ulps=rtol_ulps, atol=atol, name="default kernel vs torch_sdpa")
    assert (got0.float().cpu() - f32).abs().max().item() <= 1.5 * e_ref + 2e-4


def test_bench_attention_variant_properties_full_size(lib):
    """Config #2's sequence length (S = 20 280) on the kernel the model launches: rows of P sum to 1, identical keys give mean(V), a dominant
    key (the lazy-rescale branch, taken late in the key loop) returns its V row, and a spike in the FIRST tile followed by ordinary keys
    keeps the result finite and correct (first-tile adoption of the running max)."""
    S, H = 20280, 2
    var = lib.ATTN_FAST | lib.ATTN_Q_PRESCALED
    gen = torch.Generator(device="cuda").manual_seed(0)
    q = torch.randn(S, H * 128, generator=gen, device="cuda", dtype=torch.float32).to(torch.bfloat16)
    k = torch.randn(S, H * 128, generator=gen, device="cuda", dtype=torch.float32).to(torch.bfloat16)
    qp = (q.float() * lib.ATTN_PRESCALE).to(torch.bfloat16)
    ones = torch.ones(S, H * 128, device="cuda", dtype=torch.bfloat16)
    o = lib.attention(qp, k, ones, H, variant=var)
    assert (o.float() - 1).abs().max().item() <= 2 ** -7
    v = torch.randn(S, H * 128, generator=gen, device="cuda", dtype=torch.float32).to(torch.bfloat16)
    o = lib.attention(qp, torch.zeros_like(k), v, H, variant=var)
    assert (o.float() - v.float().mean(0, keepdim=True)).abs().max().item() <= 2e-3
    for spike_at in (12345, 3):
        k2 = k.clone()
        k2[spike_at] = (q[7].float() * 3).to(torch.bfloat16)
        o = lib.attention(qp[:64].contiguous(), k2, v, H, variant=var)
        assert torch.isfinite(o.float()).all()
        assert (o[7].float() - v[spike_at].float()).abs().max().item() <= 2 ** -6, spike_at
        other = lib.attention(q[:64].contiguous(), k2, v, H)
        assert_bf16_close(o, other.cpu(), ulps=0.256, atol=8e-3, name=f"spike at key {spike_at}: ping-pong vs default kernel")


# ------------------------------------------------------------------------------------------------ blocks at real widths
def _one_wan_block(dims, target_shape, frames, wd, lat, ctx, ref_rounding):
    from lightx2v_amd import scheduler, wan

    cfg = wan.default_config(dims, target_shape=target_shape, target_video_length=frames, infer_steps=4, hip_ref_rounding=ref_rounding)
    model = wan.WanModel(cfg, {k: v.cuda() for k, v in wd.items()})
    sch = scheduler.WanScheduler(cfg, device="cuda")
    sch.prepare(latents=lat)
    sch.timesteps[1] = 777
    model.set_scheduler(sch)
    sch.step_pre(1)
    inputs = {"text_encoder_output": {"context": [c.cuda() for c in ctx], "context_null": [c.cuda() for c in ctx]}}
    embed, grid_sizes, (x, embed0, seq_lens, freqs, context) = model.pre_infer.infer(model.pre_weight, inputs, positive=True)
    return model, grid_sizes, embed, x, embed0, seq_lens, freqs, context


@pytest.mark.parametrize("ref_rounding", [False, True])
def test_wan14b_block_vs_oracle(ref_rounding):
    """One Wan2.1-14B block (D 5120, 40 heads, F 13824; configs #3/#4) from the ORACLE's pre-infer tensors, so that only the block's own
    error is measured; both rounding modes.  2560 tokens (token grid 5 x 16 x 32): the dispatcher sends a projection to the 256x256
    GEMM kernel from ~2400 rows on, so q / k / v / o / ffn_0 / ffn_2 all run on the kernel the bench runs."""
    from lightx2v_amd import lib, synth
    from oracle import wan_oracle as O

    dims = dict(synth.WAN_DIMS["wan2.1-14b"], num_layers=1)
    ts = (16, 5, 32, 64)  # (C, T, H, W) latent → 5 x 16 x 32 = 2560 tokens
    S = synth.seq_len_of(ts)
    assert lib.gemm_kernel_choice(S, dims["dim"], dims["dim"]) == 3 and lib.gemm_kernel_choice(S, dims["ffn_dim"], dims["dim"]) == 3
    wd = synth.synth_wan_weights(dims, seed=11)
    lat, ctx, _ = synth.synth_inputs(dims, ts)
    t = torch.tensor(777)
    embed_o, grid, x_o, embed0_o, _, context_o = O.wan_pre_infer(wd, dims, lat.to(torch.bfloat16), t, ctx)
    ref = O.wan_block(wd, 0, dims, grid, x_o.clone(), embed0_o, O.rope_freqs_table(128), context_o)
    model, grid_sizes, embed, x, embed0, seq_lens, freqs, context = _one_wan_block(dims, ts, 17, wd, lat, ctx, ref_rounding)
    assert_rel(x, x_o, 1e-2, "14B patch embedding")
    tr = model.transformer_infer
    out = tr.infer_block(model.transformer_weights.blocks[0], grid_sizes, embed, x_o.cuda().clone(), embed0_o.cuda(), seq_lens, freqs, context_o.cuda())
    assert_rel(out, ref, 1e-2, f"Wan-14B block (ref_rounding={ref_rounding})")


def test_wan13b_block_config2_sequence_vs_oracle():
    """One Wan2.1-1.3B block at BASELINE config #2's shape: 480p x 49 frames → S = 20 280 tokens (the CPU oracle's attention over
    20 280^2 x 12 heads takes ~10-30 s on the box's host cores)."""
    from lightx2v_amd import synth
    from oracle import wan_oracle as O

    dims = dict(synth.WAN_DIMS["wan2.1-1.3b"], num_layers=1)
    wl = synth.WORKLOADS["wan1.3b_480px49f"]
    ts = wl["target_shape"]
    assert synth.seq_len_of(ts) == 20280
    wd = synth.synth_wan_weights(dims, seed=12)
    lat, ctx, _ = synth.synth_inputs(dims, ts)
    t = torch.tensor(777)
    embed_o, grid, x_o, embed0_o, _, context_o = O.wan_pre_infer(wd, dims, lat.to(torch.bfloat16), t, ctx)
    ref = O.wan_block(wd, 0, dims, grid, x_o.clone(), embed0_o, O.rope_freqs_table(128), context_o)
    model, grid_sizes, embed, x, embed0, seq_lens, freqs, context = _one_wan_block(dims, ts, wl["frames"], wd, lat, ctx, False)
    tr = model.transformer_infer
    out = tr.infer_block(model.transformer_weights.blocks[0], grid_sizes, embed, x_o.cuda().clone(), embed0_o.cuda(), seq_lens, freqs, context_o.cuda())
    assert_rel(out, ref, 1e-2, "Wan-1.3B block at S = 20280")


@pytest.mark.parametrize("ref_rounding", [False, True])
def test_hunyuan13b_width_blocks_vs_oracle(ref_rounding):
    """HunyuanVideo-13B width (hidden 3072, 24 heads, mlp 12288): one double block + one single block on 2304 image tokens + 256 text
    tokens (56 of them padding → two attention segments), vs oracle/hunyuan_oracle.py; inputs are seeded tensors at the block
    boundary (the pre-infer is covered at tiny width by test_gpu_hunyuan.py)."""
    from lightx2v_amd import hunyuan as hy, synth
    from oracle import hunyuan_oracle as H

    dims = dict(synth.HUNYUAN_DIMS["hunyuan-13b"], double_blocks=1, single_blocks=1)
    wd = {k: v for k, v in synth.synth_hunyuan_weights(dims, seed=21).items() if k.startswith(("double_blocks.", "single_blocks."))}
    grid = (4, 24, 24)  # (T, H/2, W/2) tokens = 2304
    n_img, n_txt, n_valid = grid[0] * grid[1] * grid[2], dims["text_len"], 200
    gen = torch.Generator().manual_seed(5)
    img = torch.randn(n_img, dims["hidden"], generator=gen).to(torch.bfloat16)
    txt = torch.randn(n_txt, dims["hidden"], generator=gen).to(torch.bfloat16)
    vec = torch.randn(1, dims["hidden"], generator=gen).to(torch.bfloat16)
    cos, sin = H.rope_tables(list(grid))
    cu = torch.tensor([0, n_img + n_valid, n_img + n_txt], dtype=torch.int32)
    with torch.no_grad():
        ref, _ = H.transformer_infer(wd, dims, img, txt, vec, cu, (cos, sin))
    cfg = hy.default_config(dims, infer_steps=4, hip_ref_rounding=ref_rounding)
    tw = hy.HunyuanTransformerWeights(cfg)
    tw.load({k: v.cuda() for k, v in wd.items()})
    tr = hy.HunyuanTransformerInfer(cfg)
    out, _ = tr.infer(tw, img.cuda(), txt.cuda(), vec.cuda(), cu, n_img + n_txt, (cos.cuda(), sin.cuda()))
    assert_rel(out, ref, 1e-2, f"Hunyuan-13B-width double+single block (ref_rounding={ref_rounding})")


def test_batched_attention_equals_per_sequence_launches(lib):
    """x2v_attn_fwd_bf16_vt_batched (grid z = sequence; the two CFG forwards of a step in one launch) against one x2v_attn_fwd_bf16_vt launch per
    sequence: EQUAL on the valid rows, with a slot size that pads each sequence (2100 -> 2112 rows), V^T from the stacked v, strided q / k
    views, and the padding rows of the output written (finite)."""
    H, S, Sp, B = 3, 2100, 2112, 2
    gen = torch.Generator().manual_seed(9)
    qkv = dev(torch.randn(B * Sp, 3 * H * 128, generator=gen).to(torch.bfloat16))
    q, k, v = qkv[:, : H * 128], qkv[:, H * 128 : 2 * H * 128], qkv[:, 2 * H * 128 :]
    vt = lib.transpose_heads(v, H)  # over the stacked rows: [H, B*Sp/64, 128, 64]
    out = torch.full((B * Sp, H * 128), float("nan"), dtype=torch.bfloat16, device="cuda")
    lib.attention_batched(q, k, vt, H, B, Sp, S, out=out, one_launch=True)
    assert torch.isfinite(out.float()).all(), "padding rows of the output must be written"
    for mode in (False, None):  # one launch per sequence on the stacked buffers / the size rule: same bits
        assert torch.equal(lib.attention_batched(q, k, vt, H, B, Sp, S, one_launch=mode), out)
    for b in range(B):
        rows = slice(b * Sp, b * Sp + S)
        ref = lib.attention(q[rows], k[rows], v[rows], H, variant=lib.ATTN_FAST)
        assert torch.equal(out[rows], ref), f"sequence {b}: max |d| = {(out[rows].float() - ref.float()).abs().max().item():.3e}"
    o2 = lib.attention_batched(q, k, vt, H, B, Sp, S, all_rows_query=False)
    assert torch.equal(o2[:S], out[:S]) and torch.equal(o2[Sp : Sp + S], out[Sp : Sp + S])


@pytest.mark.parametrize("M,K,N", [(300, 256, 256), (4100, 2560, 5120), (20280, 1536, 1536), (9450, 5120, 5120), (1000, 13824, 256), (66000, 256, 1024)])
def test_gemm_continuous_pipeline_equals_one_tile_per_workgroup(M, K, N):
    """gemm256c.hip (variant 5: persistent workgroups, the K loop running on into the next output tile, wave-private epilogue) against
    gemm256s.hip (variant 4: one output tile per workgroup): same MFMA, same k order, same rounding points -> the SAME BITS, over one-tile and
    multi-tile workgroups (2 / 340 / 480 / 740 tiles), 4 to 216 K tiles, ragged M, every epilogue, bias / gate absent, rows around the output
    untouched, scheduling-group sizes, and blocked operands through the dispatcher's default form.  Also: shapes the continuous form does not
    take are refused when forced and served by the other form otherwise (mm_weight.py:81-88 is the op both implement)."""
    from lightx2v_amd import lib

    lib.init()
    g = torch.Generator(device="cuda").manual_seed(M + K + N)
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
                    y = torch.full((M + 2, N), 7.0, dtype=torch.bfloat16, device="cuda")  # nothing may be written outside rows [1, M + 1)
                    lib.gemm(x, w, bias, epilogue=epi, out=y[1 : M + 1], variant=form | (gm << 8))
                    outs.append(y)
            assert torch.equal(outs[0], outs[1]), (M, K, N, epi, gm, sorted(kw))
            assert (outs[1][0] == 7).all() and (outs[1][-1] == 7).all() if epi != lib.EPI_RESIDUAL else True
    nb = 2
    if K % (nb * 64) == 0 and N % (nb * 128) == 0:
        ref = lib.gemm(x, w, b, variant=4)
        out = torch.full((nb, M + 3, N // nb), 7.0, dtype=torch.bfloat16, device="cuda")[:, 1 : M + 1]
        lib.gemm(x, w, b, out=out)
        assert torch.equal(out.transpose(0, 1).reshape(M, N), ref), "N-blocked y"
        xb = x.view(M, nb, K // nb).transpose(0, 1).contiguous()
        assert torch.equal(lib.gemm(xb, w, b), ref), "K-blocked x"
        r1, r2 = res.clone(), res.clone()
        lib.gemm(xb, w, b, epilogue=lib.EPI_RESIDUAL, resid=r1, gate=gate)
        lib.gemm(x, w, b, epilogue=lib.EPI_RESIDUAL, resid=r2, gate=gate, variant=4)
        assert torch.equal(r1, r2), "K-blocked x + residual"
    # shapes outside the continuous form: an odd number of K tiles, N not a multiple of 256
    xo = x[:, : K - 64] if K >= 256 + 64 else None
    if xo is not None:
        with pytest.raises(lib.X2VError):
            lib.gemm(xo, w[:, : K - 64], b, variant=5)
        assert torch.equal(lib.gemm(xo, w[:, : K - 64], b, variant=3), lib.gemm(xo, w[:, : K - 64], b, variant=4))
    if N > 256:
        with pytest.raises(lib.X2VError):
            lib.gemm(x, w[: N - 128], b[: N - 128], variant=5)


@pytest.mark.parametrize("M,K,N", [(300, 512, 256), (4100, 2560, 5120), (9450, 5120, 5120), (1000, 13824, 256), (33000, 512, 1024)])
def test_gemm_fp8_continuous_pipeline_equals_ping_pong(M, K, N):
    """gemm256c8.hip (variant 5 of the w8a8 operator: gemm256c's continuous single-stream pipeline on v_mfma_scale_f32_32x32x64_f8f6f4, dequantisation in
    the accumulator layout, wave-private transposition strip) against gemm256.hip's fp8 mode (variant 2): same MFMA, same k order, same rounding
    points -> the SAME BITS, over one-tile and multi-tile workgroups, 4 to 108 K tiles, ragged M, every epilogue, bias / gate absent, rows around the
    output untouched, scheduling-group sizes; the dispatcher's natural choice (variant 0) gives those bits too (mm_weight.py:287-319 is the op both
    implement; the statements of tools/gemm_fp8_continuous_check.py, profiles/r04_call16_*)."""
    from lightx2v_amd import lib

    lib.init()
    g = torch.Generator(device="cuda").manual_seed(M + K + N + 1)
    x = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
    xq, sx = lib.quant_fp8_rowwise(x)
    wq, sw = lib.quant_fp8_rowwise(w)
    b = torch.randn(N, generator=g, device="cuda").to(torch.bfloat16)
    res = torch.randn(M, N, generator=g, device="cuda").to(torch.bfloat16)
    gate = (torch.randn(N, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
    natural = lib.gemm_kernel_choice(M, N, K, fp8=True) == 2  # else variant 0 is the 128x128 kernel (another k order inside the MFMA: other bits)
    for gm in (0, 7):
        for epi, kw in ((lib.EPI_NONE, {}), (lib.EPI_NONE, {"bias": None}), (lib.EPI_GELU_TANH, {}), (lib.EPI_SILU, {}), (lib.EPI_RESIDUAL, {"gate": gate}), (lib.EPI_RESIDUAL, {"gate": None})):
            bias = kw.get("bias", b)
            outs = []
            for form in (2, 5, 0) if natural else (2, 5):
                if epi == lib.EPI_RESIDUAL:
                    r = res.clone()
                    lib.gemm_fp8(xq, sx, wq, sw, bias, epilogue=epi, resid=r, gate=kw["gate"], variant=form | (gm << 8))
                    outs.append(r)
                else:
                    y = torch.full((M + 2, N), 7.0, dtype=torch.bfloat16, device="cuda")  # nothing may be written outside rows [1, M + 1)
                    lib.gemm_fp8(xq, sx, wq, sw, bias, epilogue=epi, out=y[1 : M + 1], variant=form | (gm << 8))
                    outs.append(y)
            assert torch.equal(outs[0], outs[1]), (M, K, N, epi, gm, sorted(kw))
            if natural:
                assert torch.equal(outs[0], outs[2]), ("natural dispatch", M, K, N, epi, gm, sorted(kw))
            if epi != lib.EPI_RESIDUAL:
                assert (outs[1][0] == 7).all() and (outs[1][-1] == 7).all()
    # a shape outside the continuous form (an odd number of K tiles) is refused when forced and served by the ping-pong kernel otherwise
    if K >= 512 + 128:
        xo, wo = xq[:, : K - 128], wq[:, : K - 128]
        with pytest.raises(lib.X2VError):
            lib.gemm_fp8(xo, sx, wo, sw, b, variant=5)
        if lib.gemm_kernel_choice(M, N, K - 128, fp8=True) == 2:
            assert torch.equal(lib.gemm_fp8(xo, sx, wo, sw, b), lib.gemm_fp8(xo, sx, wo, sw, b, variant=2))


@pytest.mark.parametrize("M,K,N,nb", [(300, 512, 256, 2), (4100, 2560, 5120, 4), (9450, 5120, 5120, 8), (9450, 5120, 13824, 8), (9450, 13824, 5120, 8)])
def test_gemm_fp8_blocked_equals_row_major_and_the_oracle(M, K, N, nb):
    """x2v_gemm_fp8_blocked (the entry had argument-validation tests only): the w8a8 GEMM on the Ulysses exchange buffers'
    layouts — N-blocked y (what MMWeightFp8Hip.apply(out=3-D) launches for the q / k / v projections), K-blocked x codes, K-blocked x + gated
    residual — at a 128x128-kernel shape, a mid shape and the 8-GPU rank shape M = 9450 of the three Wan-14B projections.  Legs: (1) the SAME BITS as
    the row-major operator on the ping-pong kernel (variant 2), which is what the dispatcher's natural choice — the continuous kernel gemm256c8 since
    round 5 — must reproduce on block-strided operands (profiles/r05_call1_*: 23 / 23 on both kernels at first contact); (2) the oracle's restated
    mm_weight.py:236-245 + :310-318 on a row sample (<= 1 bf16 ulp on all but 0.2 % of the elements, the tolerance of test_gpu_full_size's w8a8 leg);
    (3) rows around the blocks untouched; (4) which kernel ran is asserted through x2v_gemm_kernel_choice's form bit."""
    from lightx2v_amd import lib
    from oracle import wan_oracle as O
    from tests.util import assert_bf16_close

    lib.init()
    g = torch.Generator(device="cuda").manual_seed(11 + M + N)
    x = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
    xq, sx = lib.quant_fp8_rowwise(x)
    wq, sw = lib.quant_fp8_rowwise(w)
    b = torch.randn(N, generator=g, device="cuda").to(torch.bfloat16)
    res = torch.randn(M, N, generator=g, device="cuda").to(torch.bfloat16)
    gate = (torch.randn(N, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
    family, continuous = lib.gemm_kernel_choice(M, N, K, fp8=True, with_form=True)
    assert continuous == (family == 2 and (K // 128) % 2 == 0 and K // 128 >= 4 and N % 256 == 0 and lib.switches()["X2V_GEMM_FP8_CONTINUOUS"] >= 1)
    forced = 2 if family == 2 else 1
    ref = lib.gemm_fp8(xq, sx, wq, sw, b, variant=forced)
    ref_g = lib.gemm_fp8(xq, sx, wq, sw, b, epilogue=lib.EPI_GELU_TANH, variant=forced)
    # leg 2 first: the row-major reference output itself against the oracle (codes are the HIP quantiser's: its own oracle leg is test_gpu_ops.py)
    rows = torch.randperm(M, generator=torch.Generator().manual_seed(1))[:64]
    want = ((xq[rows.cuda()].float().cpu() @ wq.float().cpu().t()) * sx[rows.cuda()].cpu() * sw.cpu().t() + b.float().cpu()).to(torch.bfloat16)
    assert_bf16_close(ref[rows.cuda()], want, ulps=1, atol=2e-3, bad_frac=2e-3, name=f"gemm_fp8 {M}x{K}x{N} vs oracle arithmetic")
    assert_bf16_close(lib.gemm_fp8(*lib.quant_fp8_rowwise(x[rows.cuda()]), wq, sw, b), O.mm_fp8(x[rows.cuda()].cpu(), wq.cpu(), sw.cpu(), b.cpu()), ulps=1, atol=2e-3, bad_frac=2e-3,
                      name=f"quant + gemm_fp8 {K}->{N} vs O.mm_fp8")
    out = torch.full((nb, M + 3, N // nb), 7.0, dtype=torch.bfloat16, device="cuda")
    lib.gemm_fp8_blocked(xq, sx, wq, sw, b, out=out[:, 1 : M + 1])
    assert torch.equal(out[:, 1 : M + 1].transpose(0, 1).reshape(M, N), ref), "N-blocked y"
    assert (out[:, 0] == 7).all() and (out[:, M + 1 :] == 7).all(), "N-blocked y: rows around the blocks written"
    out2 = torch.empty((nb, M, N // nb), dtype=torch.bfloat16, device="cuda")
    lib.gemm_fp8_blocked(xq, sx, wq, sw, b, epilogue=lib.EPI_GELU_TANH, out=out2)
    assert torch.equal(out2.transpose(0, 1).reshape(M, N), ref_g), "N-blocked y + gelu"
    if K % (nb * 128) == 0:
        xb = xq.view(torch.uint8).view(M, nb, K // nb).transpose(0, 1).contiguous().view(torch.float8_e4m3fn)
        assert torch.equal(lib.gemm_fp8_blocked(xb, sx, wq, sw, b), ref), "K-blocked x"
        r1, r2 = res.clone(), res.clone()
        lib.gemm_fp8_blocked(xb, sx, wq, sw, b, epilogue=lib.EPI_RESIDUAL, resid=r1, gate=gate)
        lib.gemm_fp8(xq, sx, wq, sw, b, epilogue=lib.EPI_RESIDUAL, resid=r2, gate=gate, variant=forced)
        assert torch.equal(r1, r2), "K-blocked x + gate-residual"
    # the de-blocking quantisation pass (x2v_quant_fp8_rowwise_blocked): a K-blocked bf16 x gives the codes and scales of the row-major x
    xbb = x.view(M, nb, K // nb).transpose(0, 1).contiguous()
    q2, s2 = lib.quant_fp8_rowwise(xbb)
    assert torch.equal(q2.view(torch.uint8), xq.view(torch.uint8)) and torch.equal(s2, sx)
    pad = torch.zeros(nb, M + 2, K // nb + 8, dtype=torch.bfloat16, device="cuda")  # strided blocks: row stride and block stride both padded
    pad[:, 1 : M + 1, : K // nb] = xbb
    q3, s3 = lib.quant_fp8_rowwise(pad[:, 1 : M + 1, : K // nb])
    assert torch.equal(q3.view(torch.uint8), xq.view(torch.uint8)) and torch.equal(s3, sx)


@pytest.mark.parametrize("fp8", [False, True])
def test_gemm_continuous_forms_on_strided_operands(fp8):
    """the equality tests of the continuous kernels used contiguous operands only.  Here ldx > K (x a column slice of a wider tensor: the
    fused-QKV / cat views of the drivers), ldw > K and ldy > N (y a column slice: what a projection writing into a wider buffer does), for every
    epilogue, bf16 and w8a8: continuous form (variant 5) == one-tile / ping-pong form (variant 4 / 2), columns around y untouched, and the residual
    epilogue with ldr == ldy > N (the only residual layout the continuous form takes) equal as well."""
    from lightx2v_amd import lib

    lib.init()
    M, K, N = 4100, 2560, 5120
    g = torch.Generator(device="cuda").manual_seed(5)
    xw = torch.randn(M, K + 256, generator=g, device="cuda").to(torch.bfloat16)
    ww = (torch.randn(N, K + 128, generator=g, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
    x, w = xw[:, 128 : 128 + K], ww[:, :K]
    b = torch.randn(N, generator=g, device="cuda").to(torch.bfloat16)
    gate = (torch.randn(N, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
    resw = torch.randn(M, N + 512, generator=g, device="cuda").to(torch.bfloat16)
    if fp8:
        xq_c, sx = lib.quant_fp8_rowwise(x.contiguous())
        wq_c, sw = lib.quant_fp8_rowwise(w.contiguous())
        xqw = torch.zeros(M, K + 256, dtype=torch.uint8, device="cuda")
        xqw[:, 128 : 128 + K] = xq_c.view(torch.uint8)
        wqw = torch.zeros(N, K + 128, dtype=torch.uint8, device="cuda")
        wqw[:, :K] = wq_c.view(torch.uint8)
        xo, wo = xqw.view(torch.float8_e4m3fn)[:, 128 : 128 + K], wqw.view(torch.float8_e4m3fn)[:, :K]
        run = lambda form, **kw: lib.gemm_fp8(xo, sx, wo, sw, b, variant=form, **kw)  # noqa: E731
        forms = (2, 5, 0)
    else:
        run = lambda form, **kw: lib.gemm(x, w, b, variant=form, **kw)  # noqa: E731
        forms = (4, 5, 0)
    for epi in (lib.EPI_NONE, lib.EPI_GELU_TANH, lib.EPI_SILU):
        outs = []
        for form in forms:
            yw = torch.full((M + 2, N + 512), 7.0, dtype=torch.bfloat16, device="cuda")
            run(form, epilogue=epi, out=yw[1 : M + 1, 256 : 256 + N])
            outs.append(yw)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (fp8, epi)
        assert (outs[1][:, :256] == 7).all() and (outs[1][:, 256 + N :] == 7).all() and (outs[1][0] == 7).all() and (outs[1][-1] == 7).all()
    outs = []
    for form in forms:
        r = resw.clone()
        run(form, epilogue=lib.EPI_RESIDUAL, resid=r[:, 256 : 256 + N], gate=gate)
        outs.append(r)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (fp8, "residual")
    assert torch.equal(outs[1][:, :256], resw[:, :256]) and torch.equal(outs[1][:, 256 + N :], resw[:, 256 + N :])
