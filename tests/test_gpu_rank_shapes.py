"""Parity for what ONE RANK of the 8-GPU sequence-parallel run launches, at the real dimensions of BASELINE configs #3 and #5.

With 8 ranks (attentions/distributed/ulysses/attn.py:7-91, comm/all2all.py:6-89) a Wan2.1-14B 720p x 81f rank holds 9 450 of the 75 600 tokens
and, inside the attention, 5 of the 40 heads over ALL tokens.  Its launches are therefore not the single-GPU ones at a smaller size but different
kernel branches:
  * self-attention: `x2v_attn_fwd_bf16_vt` at H = 5, Sk = 75 600, on the received exchange buffers (token stride 640), once per head->seq piece
    (query rows [0, 37 800) and [37 800, 75 600)), pre-scaled q, NO stagger, and — by the launcher's rule, 5 heads x 38.7 MB fit the Infinity
    Cache — on the XCD-remapped grid with 148 x 5 = 740 workgroups (740 % 8 = 4: the uneven branch of the remap arithmetic);
  * projections: `x2v_gemm_bf16_blocked` with 8 N-blocks (v -> its send buffer [8, 9450, 640]) and 8 K-blocks (the received head->seq buffer ->
    the output projection with the gate-residual epilogue) at M = 9 450 = 36 x 256 + 234;
  * `x2v_rmsnorm_rope_blocked_bf16` with the rank's token offset s0 = r x 9450 into the (21, 45, 80) grid.
HunyuanVideo-13B 720p x 129f (config #5): 3 of 24 heads over 118 800 image + 200 valid text tokens.
Each is compared with the CPU oracle on sampled rows (O.attention_rows / O.mm / the reference's RoPE statements), with the fp32 triangle for attention.
The last two tests run one whole Wan-14B block / one HunyuanVideo-13B double block AT THESE DIMENSIONS through the 8-ranks-on-one-GPU harness
(the product's Ulysses driver; gloo with host-staged collectives because the box has one GPU) against `O.wan_block_rows` / `H.double_block_rows`."""
import math
import os
import subprocess
import sys

import pytest
import torch

from tests.test_gpu_full_size import PLANTS, S_WAN, _attn_inputs, _check_attention_rows, _plant, sample_rows
from tests.util import assert_bf16_close, record, rel_l2

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD = 8


@pytest.fixture(scope="module")
def lib():
    from lightx2v_amd import lib as L

    L.init()
    return L


def test_attention_wan14b_rank_of_8_both_head2seq_pieces(lib):
    """ulysses.UlyssesAttention.attend_blocked at N = 8 on Wan-14B 720p: q/k/v = the received buffers viewed [75600, 5 x 128], one launch per
    head->seq piece writing o[rows] in place; 3 unit-spread heads + 2 peaky ones; planted keys in head 0."""
    S, H = S_WAN, 40 // WORLD
    half = (WORLD // 2) * (S // WORLD)
    assert half == 37800
    assert lib.attn_vt_launch_plan(half, S, H) == (True, False), "the rank shape must take the XCD-remapped grid without the stagger"
    assert (((half + 255) // 256) * H) % 8 != 0, "this shape exercises the uneven (r8 != 0) branch of the remap"
    q32, k, v = _attn_inputs(S, S, H, seed=11, n_plain=3)
    _plant(q32, k, 0, S, PLANTS)
    q_pre = (q32 * lib.ATTN_PRESCALE).to(torch.bfloat16)
    vt = lib.transpose_heads(v, H)
    var = lib.ATTN_FAST | lib.ATTN_Q_PRESCALED  # what WanTransformerInfer.infer_self_attn passes under Ulysses
    o = torch.full((S, H * 128), float("nan"), dtype=torch.bfloat16, device="cuda")
    for rows in (slice(0, half), slice(half, S)):
        lib.attention(q_pre[rows], k, None, H, variant=var, vt=vt, out=o[rows])
    assert torch.isfinite(o.float()).all()
    rows = sample_rows(S, 192, seed=12, must=[r for r, _ in PLANTS] + [half - 1, half, half + 255, half + 256])
    _check_attention_rows(lib, o[rows.cuda()], q32[rows.cuda()].cpu(), k.cpu(), v.cpu(), H, "attn Wan-14B rank of 8: S=75600 H=5, two pieces", n_plain=3)
    for r, j in PLANTS:
        assert (o[r, :128].float() - v[j, :128].float()).abs().max().item() <= 2 ** -6, (r, j)
    # One launch over all 75 600 rows (also on the remapped grid: 296 x 5 workgroups).  The lazy rescale of the online softmax is a WAVE-uniform
    # decision (one row whose max jumps — a planted key — makes its whole 32-row wave rescale), so a row's bits depend on which rows share its wave:
    # the first piece's rows up to the last whole wave ([0, 37 792): the piece ends 8 rows into wave 1181, whose other rows are past Sq there and
    # real rows here) sit in the same waves in both launches and must carry the same bits (no stagger: the walk starts at tile 0 for every block);
    # the second piece starts at row 37 800 = 147 x 256 + 168, so all its rows have other wave-mates than in the full launch: same values to
    # rounding, not the same bits.
    o_all = lib.attention(q_pre, k, None, H, variant=var, vt=vt)
    whole = half // 32 * 32
    assert torch.equal(o_all[:whole], o[:whole])
    assert rel_l2(o_all[whole:], o[whole:]) <= 2e-3


def test_attention_hunyuan13b_rank_of_8(lib):
    """ulysses.UlyssesHunyuanAttention.attend_blocked at N = 8 on HunyuanVideo-13B 720p x 129f: 3 heads, keys = 118 800 image + 200 valid text rows of
    the joint buffers [119 056, 384]; piece 1 = image rows of ranks 0-3, piece 2 = the rest + the valid text queries; the 56 padded text rows attend
    among themselves (hunyuan/infer/transformer_infer.py:119-146, cu_seqlens of pre_infer.py:50-56)."""
    H, n_img, n_txt, n_valid = 24 // WORLD, 118800, 256, 200
    tot, nq = n_img, n_img + n_valid
    half = (WORLD // 2) * (n_img // WORLD)
    assert lib.attn_vt_launch_plan(half, nq, H) == (True, False)
    q32, k, v = _attn_inputs(tot + n_txt, nq, H, seed=13, n_plain=2)
    plants = [(7, 118799), (118800, 3), (118999, 118999), (60000, 65536), (half - 1, half), (half, half - 1)]
    _plant(q32, k, 0, nq, plants)
    q_pre = (q32 * lib.ATTN_PRESCALE).to(torch.bfloat16)
    var = lib.ATTN_FAST | lib.ATTN_Q_PRESCALED
    vt = lib.transpose_heads(v[:nq], H)
    o = torch.full((tot + n_txt, H * 128), float("nan"), dtype=torch.bfloat16, device="cuda")
    lib.attention(q_pre[:half], k[:nq], v[:nq], H, 128, out=o[:half], variant=var, vt=vt)
    lib.attention(q_pre[half:nq], k[:nq], v[:nq], H, 128, out=o[half:nq], variant=var, vt=vt)
    lib.attention(q_pre[nq:], k[nq:], v[nq:], H, 128, out=o[nq:], variant=var)
    assert torch.isfinite(o.float()).all()
    rows = sample_rows(nq, 160, seed=14, must=[r for r, _ in plants] + [n_img - 1, n_img])
    _check_attention_rows(lib, o[rows.cuda()], q32[rows.cuda()].cpu(), k[:nq].cpu().contiguous(), v[:nq].cpu().contiguous(), H, "attn Hunyuan-13B rank of 8: S=119000 H=3", n_plain=2)
    for r, j in plants:
        assert (o[r, :128].float() - v[j, :128].float()).abs().max().item() <= 2 ** -6, (r, j)
    _check_attention_rows(lib, o[nq:], q32[nq:].cpu(), k[nq:].cpu().contiguous(), v[nq:].cpu().contiguous(), H, "attn Hunyuan-13B rank of 8: padded-text segment", atol=8e-3, n_plain=2)


@pytest.mark.parametrize("K,N,epi", [(5120, 5120, "v: N-blocked y"), (5120, 5120, "o: K-blocked x + gate-residual"), (5120, 13824, "ffn0: gelu"), (13824, 5120, "ffn2: gate-residual")])
def test_gemm_wan14b_rank_of_8_rows_vs_oracle(lib, K, N, epi):
    """The projections of one Wan-14B block as an 8-GPU rank launches them: M = 9 450 rows (37 row tiles, the last holding 234 rows); v writes its
    seq->head send buffer [8, 9450, 640] from the epilogue (comm/all2all.py:29-33 without the transposing copy), the output projection reads the
    received head->seq buffer [8, 9450, 640] as a K-blocked x (all2all.py:70-75) and adds the gated result to x; ffn0 / ffn2 are row-major at this M.
    Sampled rows vs `O.mm` + the reference's separate ops, and bit-equality of the blocked forms with the row-major kernel."""
    from oracle import wan_oracle as O

    M, nb = S_WAN // WORLD, WORLD
    assert M == 9450 and lib.gemm_kernel_choice(M, N, K) == 3
    g = torch.Generator(device="cuda").manual_seed(K + N + len(epi))
    x = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device="cuda") / math.sqrt(K)).to(torch.bfloat16)
    b = (torch.randn(N, generator=g, device="cuda") * 0.1).to(torch.bfloat16)
    rows = sample_rows(M, 160, seed=K + N)
    rc = rows.cuda()
    ref = O.mm(x[rc].cpu(), w.cpu(), b.cpu())
    if epi.startswith("v:"):
        out = torch.full((nb, M + 2, N // nb), 7.0, dtype=torch.bfloat16, device="cuda")[:, 1 : M + 1]  # a strided view inside a poisoned buffer
        lib.gemm(x, w, b, out=out)
        got = out.transpose(0, 1).reshape(M, N)
        assert torch.equal(got, lib.gemm(x, w, b)), "N-blocked y must carry the row-major kernel's bits"
        assert_bf16_close(got[rc], ref, ulps=1, atol=2e-3, bad_frac=1e-3, name=epi)
        record(f"rank-of-8 gemm {M}x{K}x{N} ({epi})", rel_l2=rel_l2(got[rc], ref))
    elif epi.startswith("o:"):
        xb = x.view(M, nb, K // nb).transpose(0, 1).contiguous()  # [8, 9450, 640]: the received head->seq buffer
        res = torch.randn(M, N, generator=g, device="cuda").to(torch.bfloat16)
        gate = (torch.randn(1, N, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
        ref_r = res[rc].cpu()
        ref_r.add_(ref * gate.cpu().squeeze(0))
        r1, r2 = res.clone(), res.clone()
        lib.gemm(xb, w, b, epilogue=lib.EPI_RESIDUAL, resid=r1, gate=gate)
        lib.gemm(x, w, b, epilogue=lib.EPI_RESIDUAL, resid=r2, gate=gate)
        assert torch.equal(r1, r2), "K-blocked x must give the row-major kernel's bits"
        assert_bf16_close(r1[rc], ref_r, ulps=1, atol=6e-3, bad_frac=2e-3, name=epi)
        record(f"rank-of-8 gemm {M}x{K}x{N} ({epi})", rel_l2=rel_l2(r1[rc], ref_r))
    elif epi.startswith("ffn0"):
        got = lib.gemm(x, w, b, epilogue=lib.EPI_GELU_TANH)
        assert_bf16_close(got[rc], torch.nn.functional.gelu(ref, approximate="tanh"), ulps=1, atol=2e-3, bad_frac=2e-3, name=epi)
    else:
        res = torch.randn(M, N, generator=g, device="cuda").to(torch.bfloat16)
        gate = (torch.randn(1, N, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
        ref_r = res[rc].cpu()
        ref_r.add_(ref * gate.cpu().squeeze(0))
        lib.gemm(x, w, b, epilogue=lib.EPI_RESIDUAL, resid=res, gate=gate)
        assert_bf16_close(res[rc], ref_r, ulps=1, atol=6e-3, bad_frac=2e-3, name=epi)


def test_w8a8_operator_on_the_exchange_buffers_wan14b_rank_of_8(lib):
    """The w8a8 operator class (mm_weight.py:236-245, :287-319) on an 8-GPU rank's Ulysses buffers at the Wan-14B dimensions, through the calls the
    copy-free driver makes: `apply(x, out=3-D)` writes the seq->head send buffer [8, 9450, 640] from the epilogue of the continuous fp8 kernel
    (all2all.py:29-33 without the transposing copy), `apply(3-D x, residual)` de-blocks the received head->seq buffer in its quantisation pass
    (all2all.py:70-75) and multiplies row-major.  Both must carry the row-major operator's bits; the row-major result is checked against the oracle's
    restated quantise + scaled-mm on a row sample; the form bit says which kernel ran."""
    from lightx2v_amd.ops import MMWeightFp8Hip
    from oracle import wan_oracle as O

    M, nb, D = S_WAN // WORLD, WORLD, 5120
    family, continuous = lib.gemm_kernel_choice(M, D, D, fp8=True, with_form=True)
    assert (family, continuous) == (2, lib.switches()["X2V_GEMM_FP8_CONTINUOUS"] >= 1)
    g = torch.Generator(device="cuda").manual_seed(20)
    x = torch.randn(M, D, generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn(D, D, generator=g, device="cuda") / math.sqrt(D)).to(torch.bfloat16)
    b = (torch.randn(D, generator=g, device="cuda") * 0.1).to(torch.bfloat16)
    mm = MMWeightFp8Hip("p.weight", "p.bias")
    mm.config = {"weight_auto_quant": True}
    mm.load({"p.weight": w, "p.bias": b})
    row = mm.apply(x)
    rows = sample_rows(M, 96, seed=3)
    rc = rows.cuda()
    ref = O.mm_fp8(x[rc].cpu(), mm.weight.cpu(), mm.weight_scale.cpu(), b.cpu())
    assert_bf16_close(row[rc], ref, ulps=1, atol=2e-3, bad_frac=2e-3, name="w8a8 operator, rank-of-8 rows vs O.mm_fp8")
    record(f"rank-of-8 w8a8 operator {M}x{D}x{D}", rel_l2=rel_l2(row[rc], ref))
    # q / k / v: the send buffer, a strided view inside a poisoned allocation
    buf = torch.full((nb, M + 2, D // nb), 7.0, dtype=torch.bfloat16, device="cuda")
    mm.apply(x, out=buf[:, 1 : M + 1])
    assert torch.equal(buf[:, 1 : M + 1].transpose(0, 1).reshape(M, D), row), "N-blocked y (y_cbw = 640) must carry the row-major operator's bits"
    assert (buf[:, 0] == 7).all() and (buf[:, M + 1] == 7).all(), "rows around the blocks written"
    # output projection: K-blocked x (the receive buffer) + gated residual
    xb = x.view(M, nb, D // nb).transpose(0, 1).contiguous()
    res = torch.randn(M, D, generator=g, device="cuda").to(torch.bfloat16)
    gate = (torch.randn(1, D, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
    r1, r2 = res.clone(), res.clone()
    mm.apply(xb, epilogue=lib.EPI_RESIDUAL, resid=r1, gate=gate)
    mm.apply(x, epilogue=lib.EPI_RESIDUAL, resid=r2, gate=gate)
    assert torch.equal(r1, r2), "K-blocked x through the de-blocking quantisation pass must give the row-major operator's bits"
    ref_r = res[rc].cpu()
    ref_r.add_(ref * gate.cpu().squeeze(0))
    assert_bf16_close(r1[rc], ref_r, ulps=1, atol=6e-3, bad_frac=2e-3, name="w8a8 o-projection + gate-residual")


@pytest.mark.parametrize("rank", [0, 7])
def test_rmsnorm_rope_blocked_wan14b_rank_of_8_vs_oracle(lib, rank):
    """x2v_rmsnorm_rope_blocked_bf16 as rank r of 8 calls it on Wan-14B 720p: 9 450 rows whose grid positions start at s0 = r x 9450 of the
    (21, 45, 80) token grid, 40 heads, written into the [8, 9450, 640] send buffers — vs the reference's statements (RMSNorm of
    rms_norm_weight.py:111-113 in the fp32-statistics model + compute_freqs_dist / apply_rotary_emb, wan/infer/utils.py:86-115) on the CPU."""
    from lightx2v_amd.wan import rope_cos_sin_table
    from oracle import wan_oracle as O

    H, grid, M = 40, (21, 45, 80), S_WAN // WORLD
    g = torch.Generator().manual_seed(20 + rank)
    q = torch.randn(M, H * 128, generator=g).to(torch.bfloat16)
    k = torch.randn(M, H * 128, generator=g).to(torch.bfloat16)
    wq = (1 + 0.1 * torch.randn(H * 128, generator=g)).to(torch.bfloat16)
    wk = (1 + 0.1 * torch.randn(H * 128, generator=g)).to(torch.bfloat16)
    qo = torch.full((WORLD, M, H * 128 // WORLD), 7.0, dtype=torch.bfloat16, device="cuda")
    ko = torch.full_like(qo, 7.0)
    lib.rmsnorm_rope_blocked(q.cuda(), k.cuda(), wq.cuda(), wk.cuda(), rope_cos_sin_table(128, "cuda"), grid, H, qo, ko, s0=rank * M)
    freqs_i = O.compute_freqs_dist(M, 64, grid, O.rope_freqs_table(128), rank, WORLD)
    for name, got_b, src, w in (("q", qo, q, wq), ("k", ko, k, wk)):
        got = got_b.transpose(0, 1).reshape(M, H * 128)
        ref = O.apply_rotary_emb(O.rms_norm_fp32(src, w).view(M, H, 128), freqs_i).reshape(M, H * 128)
        assert_bf16_close(got, ref, ulps=1, atol=2e-3, bad_frac=2e-3, name=f"rmsnorm_rope_blocked {name} rank {rank}")
        record(f"rank-of-8 rmsnorm+rope blocked ({name}, rank {rank}, s0={rank * M})", rel_l2=rel_l2(got, ref))


# ------------------------------------------------------------------------------------------------ one block through the Ulysses driver, real dims
def _run_ranks(worker, port, tmp, extra_env=None):
    env = dict(os.environ, OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0", X2V_RANK_SHAPES_DIR=str(tmp), **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(WORLD), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", worker)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-4000:]
    return p.stdout


def test_wan14b_block_720p_through_ulysses_world8_vs_oracle_rows(tmp_path):
    """One Wan2.1-14B block on the 720p x 81f grid, sharded over 8 ranks exactly as config #3 runs it (9 450 tokens and 5 heads per rank; blocked
    exchange buffers, two-piece head->seq, K-blocked output projection, RoPE offsets r x 9450) — all eight ranks driving the box's one GPU — against
    `O.wan_block_rows` on sampled rows of ranks 0, 3 and 7, with the fp32 evaluation of the same rows as truth.  The parent process builds the
    block-boundary inputs with the oracle and hands them to the ranks through a scratch file; the ranks return their shard's sampled rows."""
    from lightx2v_amd import synth
    from oracle import wan_oracle as O

    dims = dict(synth.WAN_DIMS["wan2.1-14b"], num_layers=1)
    ts = synth.WORKLOADS["wan14b_720px81f"]["target_shape"]
    S = synth.seq_len_of(ts)
    assert S == S_WAN and S % WORLD == 0
    wd = synth.synth_wan_weights(dims, seed=31)
    lat, ctx, _ = synth.synth_inputs(dims, ts)
    embed_o, grid, x_o, embed0_o, _, context_o = O.wan_pre_infer(wd, dims, lat.to(torch.bfloat16), torch.tensor(777), ctx)
    s_local = S // WORLD
    local = sample_rows(s_local, 40, seed=5)
    torch.save({"x": x_o, "embed": embed_o, "embed0": embed0_o, "context": context_o, "grid": list(grid), "local_rows": local}, tmp_path / "inputs.pt")
    out = _run_ranks("_dist_gpu_worker_rank_shapes.py", 29561, tmp_path, {"X2V_RANK_SHAPES_MODEL": "wan"})
    assert "RANK_SHAPES_WAN_OK" in out
    freqs = O.rope_freqs_table(128)
    ranks = (0, 3, 7)
    rows = torch.cat([r * s_local + local for r in ranks])
    ref = O.wan_block_rows(wd, 0, dims, grid, x_o, embed0_o, freqs, context_o, rows)
    with O.truth_precision(torch.float32):
        tru = O.wan_block_rows(O.upcast(wd), 0, dims, grid, x_o.float(), embed0_o.float(), freqs, context_o.float(), rows)
    got = torch.cat([torch.load(tmp_path / f"out_rank{r}.pt") for r in ranks])
    assert torch.isfinite(got.float()).all()
    for i, r in enumerate(ranks):
        sl = slice(i * len(local), (i + 1) * len(local))
        e, e_hip, e_ref = rel_l2(got[sl], ref[sl]), rel_l2(got[sl], tru[sl]), rel_l2(ref[sl], tru[sl])
        record(f"Wan-14B block S=75600 through Ulysses world 8, rank {r}", rows=len(local), rel_l2_vs_oracle=e, err_hip_vs_fp32=e_hip, err_oracle_vs_fp32=e_ref)
        assert e <= 1e-2, f"rank {r}: relative L2 vs oracle {e:.3e}"
        assert e_hip <= 1.5 * e_ref + 1e-4, f"rank {r}: err vs fp32 truth {e_hip:.3e} > 1.5 x the bf16 oracle's {e_ref:.3e}"


def test_hunyuan13b_double_block_720p_129f_through_ulysses_world8_vs_oracle_rows(tmp_path):
    """One HunyuanVideo-13B double block at config #5's size through `UlyssesHunyuanAttention` with 8 ranks on the one GPU: each rank owns a
    10-column slab of the 33 x 45 x 80 token grid (14 850 image tokens; utils/hunyuan/processor.py:5-50 splits W because 45 % 8 != 0), the 256 text
    tokens are replicated, 3 heads per rank — vs `H.double_block_rows` evaluated on the WHOLE sequence in its natural order for sampled tokens of
    ranks 0, 3 and 7 (a token's result does not depend on where the other tokens sit)."""
    from lightx2v_amd import synth
    from oracle import hunyuan_oracle as H

    dims = dict(synth.HUNYUAN_DIMS["hunyuan-13b"], double_blocks=1, single_blocks=0)
    wd = {k: v for k, v in synth.synth_hunyuan_weights(dict(dims, double_blocks=1, single_blocks=1), seed=21).items() if k.startswith("double_blocks.")}
    grid = (33, 45, 80)
    n_img, n_txt, n_valid = grid[0] * grid[1] * grid[2], dims["text_len"], 200
    gen = torch.Generator().manual_seed(6)
    img = torch.randn(n_img, dims["hidden"], generator=gen).to(torch.bfloat16)
    txt = torch.randn(n_txt, dims["hidden"], generator=gen).to(torch.bfloat16)
    vec = torch.randn(1, dims["hidden"], generator=gen).to(torch.bfloat16)
    cos, sin = H.rope_tables(list(grid))
    w_local = grid[2] // WORLD
    n_local = grid[0] * grid[1] * w_local
    local = sample_rows(n_local, 32, seed=8)
    torch.save({"img": img, "txt": txt, "vec": vec, "cos": cos, "sin": sin, "grid": list(grid), "n_valid": n_valid, "local_rows": local}, tmp_path / "inputs.pt")
    out = _run_ranks("_dist_gpu_worker_rank_shapes.py", 29563, tmp_path, {"X2V_RANK_SHAPES_MODEL": "hunyuan"})
    assert "RANK_SHAPES_HUNYUAN_OK" in out
    ranks = (0, 3, 7)
    index = torch.arange(n_img).view(*grid)
    rows = torch.cat([index[:, :, r * w_local : (r + 1) * w_local].reshape(-1)[local] for r in ranks])  # natural-order index of each rank's sampled tokens
    cu = torch.tensor([0, n_img + n_valid, n_img + n_txt], dtype=torch.int32)
    with torch.no_grad():
        ref, _ = H.double_block_rows(wd, 0, img, txt, vec, (cos, sin), dims["heads"], cu, rows)
    got = torch.cat([torch.load(tmp_path / f"out_rank{r}.pt") for r in ranks])
    assert torch.isfinite(got.float()).all()
    for i, r in enumerate(ranks):
        sl = slice(i * len(local), (i + 1) * len(local))
        e = rel_l2(got[sl], ref[sl])
        record(f"Hunyuan-13B double block 118800+256 tokens through Ulysses world 8, rank {r}", rows=len(local), rel_l2_vs_oracle=e)
        assert e <= 4e-3, f"rank {r}: relative L2 vs oracle {e:.3e}"
