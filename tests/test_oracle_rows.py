"""CPU checks of the oracle's full-size helpers: `wan_block_rows` (a block evaluated on a subset of the token rows, used by the
S = 75 600 parity tests) equals `wan_block` on those rows, `attention_rows` equals `sdpa` / `attention_fp32`, and `truth_precision`
(the same graph without bf16 rounding points) stays close to, but is not, the bf16 graph."""
import torch

from lightx2v_amd import synth
from oracle import wan_oracle as O
from tests.util import rel_l2


def _setup(seed=3):
    dims = dict(synth.WAN_DIMS["wan-tiny"], num_layers=1)
    ts = synth.WORKLOADS["wan-tiny"]["target_shape"]
    wd = synth.synth_wan_weights(dims, seed=seed)
    lat, ctx, _ = synth.synth_inputs(dims, ts)
    embed, grid, x, embed0, s, context = O.wan_pre_infer(wd, dims, lat.to(torch.bfloat16), torch.tensor(600), ctx)
    return dims, wd, grid, x, embed0, context, lat, ctx


def test_block_rows_equals_block():
    dims, wd, grid, x, embed0, context, _, _ = _setup()
    freqs = O.rope_freqs_table(dims["dim"] // dims["num_heads"])
    full = O.wan_block(wd, 0, dims, grid, x.clone(), embed0, freqs, context)
    S = x.shape[0]
    rows = torch.tensor([0, 1, 31, 32, S // 2, S - 2, S - 1, 7])
    got = O.wan_block_rows(wd, 0, dims, grid, x, embed0, freqs, context, rows)
    # Every op is row-independent on the CPU backend EXCEPT torch's bf16 `rsqrt` (rms_norm), whose vectorised body and scalar tail round
    # differently, so which rows are one ulp off depends on a row's position in the tensor (measured: q / k / v / ffn GEMMs, LayerNorm,
    # GELU, RoPE and SDPA give identical bits for a row subset; rms_norm differs on whole rows by one bf16 ulp).  Hence rounding-level
    # agreement, not bit equality.
    assert rel_l2(got, full[rows]) <= 5e-3
    assert (got.float() - full[rows].float()).abs().max().item() <= 2 ** -5 * full.float().abs().max().item()
    ctx2 = (context.float() * 0.5).to(context.dtype)
    pair = O.wan_block_rows(wd, 0, dims, grid, x, embed0, freqs, [context, ctx2], rows)
    assert torch.equal(pair[0], got)
    assert rel_l2(pair[1], O.wan_block(wd, 0, dims, grid, x.clone(), embed0, freqs, ctx2)[rows]) <= 5e-3


def test_block_rows_equals_block_i2v():
    """Same for the i2v block (second cross-attention over the 257 CLIP tokens, transformer_infer.py:405-455)."""
    dims = dict(synth.WAN_DIMS["wan-tiny-i2v"], num_layers=1)
    ts = (16, 3, 8, 8)
    wd = synth.synth_wan_i2v_weights(dims, seed=3)
    lat, ctx, _ = synth.synth_inputs(dims, ts)
    image = synth.synth_i2v_inputs(dims, ts)
    embed, grid, x, embed0, s, context = O.wan_pre_infer(wd, dims, lat.to(torch.bfloat16), torch.tensor(600), ctx, image=image)
    freqs = O.rope_freqs_table(128)
    full = O.wan_block(wd, 0, dims, grid, x.clone(), embed0, freqs, context)
    S = x.shape[0]
    rows = torch.tensor([0, 1, 31, 32, S // 2, S - 2, S - 1, 7])
    got = O.wan_block_rows(wd, 0, dims, grid, x, embed0, freqs, context, rows)
    assert rel_l2(got, full[rows]) <= 5e-3
    t2v = O.wan_block_rows(wd, 0, dict(dims, task="t2v"), grid, x, embed0, freqs, context[O.I2V_CLIP_TOKENS :], rows)
    assert rel_l2(got, t2v) > 1e-3, "the image branch must contribute"


def test_attention_rows():
    gen = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(300, 3, 128, generator=gen).to(torch.bfloat16) for _ in range(3))
    rows = torch.tensor([0, 5, 299, 150])
    assert rel_l2(O.attention_rows(q[rows], k, v), O.sdpa(q, k, v)[rows]) <= 2e-3
    assert torch.allclose(O.attention_rows(q[rows], k, v, fp32=True), O.attention_fp32(q, k, v)[rows], rtol=1e-5, atol=1e-6)


def test_truth_precision_block_and_forward():
    dims, wd, grid, x, embed0, context, lat, ctx = _setup()
    freqs = O.rope_freqs_table(dims["dim"] // dims["num_heads"])
    ref = O.wan_block(wd, 0, dims, grid, x.clone(), embed0, freqs, context)
    with O.truth_precision(torch.float32):
        tr = O.wan_block(O.upcast(wd), 0, dims, grid, x.float(), embed0.float(), freqs, context.float())
    assert tr.dtype == torch.float32
    e = rel_l2(ref, tr)
    assert 1e-4 < e < 2e-2, e
    with O.truth_precision(torch.float64):
        tr64 = O.wan_block(O.upcast(wd, torch.float64), 0, dims, grid, x.double(), embed0.double(), freqs, context.double())
    assert rel_l2(tr, tr64) < 1e-5
    # whole CFG forward
    t = torch.tensor(600)
    ref = O.wan_model_infer(wd, dims, lat.to(torch.bfloat16), t, ctx, ctx, 5.0)
    with O.truth_precision():
        tr = O.wan_model_infer(O.upcast(wd), dims, lat.to(torch.bfloat16), t, O.upcast(ctx), O.upcast(ctx), 5.0)
    assert tr.dtype == torch.float32 and 1e-4 < rel_l2(ref, tr) < 5e-2
    assert O._ACT == [torch.bfloat16]


def test_hunyuan_block_rows_equal_blocks():
    """oracle.hunyuan_oracle.double_block_rows / single_block_rows (a block evaluated on a subset of the image rows + all text rows; used by the
    119 056-token parity test) against the full blocks, with padded text (two attention segments)."""
    from oracle import hunyuan_oracle as H

    dims = synth.HUNYUAN_DIMS["hunyuan-tiny"]
    wd = synth.synth_hunyuan_weights(dims, seed=9)
    gen = torch.Generator().manual_seed(1)
    grid = (2, 6, 8)
    n_img, n_txt, n_valid, D = grid[0] * grid[1] * grid[2], dims["text_len"], 11, dims["hidden"]
    img = torch.randn(n_img, D, generator=gen).to(torch.bfloat16)
    txt = torch.randn(n_txt, D, generator=gen).to(torch.bfloat16)
    vec = torch.randn(1, D, generator=gen).to(torch.bfloat16)
    freqs = H.rope_tables(list(grid))
    cu = torch.tensor([0, n_img + n_valid, n_img + n_txt], dtype=torch.int32)
    rows = torch.tensor([0, 1, 17, n_img // 2, n_img - 2, n_img - 1, 5])
    with torch.no_grad():
        fi, ft = H.double_block(wd, 0, img, txt, vec, freqs, dims["heads"], cu)
        ri, rt = H.double_block_rows(wd, 0, img, txt, vec, freqs, dims["heads"], cu, rows)
        assert rel_l2(ri, fi[rows]) <= 5e-3 and rel_l2(rt, ft) <= 5e-3, (rel_l2(ri, fi[rows]), rel_l2(rt, ft))
        x = torch.cat((img, txt), 0)
        fs = H.single_block(wd, 0, x, vec, n_txt, freqs, dims["heads"], D, cu)
        rs = H.single_block_rows(wd, 0, x, vec, n_txt, freqs, dims["heads"], D, cu, rows)
        sel = torch.cat((rows, n_img + torch.arange(n_txt)))
        assert rel_l2(rs, fs[sel]) <= 5e-3, rel_l2(rs, fs[sel])
