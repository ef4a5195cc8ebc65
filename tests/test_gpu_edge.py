"""Edge cases and error behaviour of the C-ABI through the Python wrappers: empty inputs, ragged sizes around the tile
edges, strided views, and the argument errors a caller can make.  The reference's convention is a Python exception
(RuntimeError from TORCH_CHECK in its native ops); here every x2v_* returns a negative code with a message
behind x2v_last_error() and the wrapper raises X2VError (a RuntimeError) — nothing is silently computed on a fallback."""
import pytest
import torch

from tests.util import assert_bf16_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from lightx2v_amd import lib as L

    L.init()
    return L


def bf(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).cuda()


def test_empty_inputs_are_no_ops(lib):
    D, N, H = 256, 384, 2
    x0 = torch.empty(0, D, dtype=torch.bfloat16, device="cuda")
    w, b = bf(N, D, seed=1, scale=0.05), bf(N, seed=2)
    assert lib.gemm(x0, w, b).shape == (0, N)
    assert lib.rmsnorm(x0, bf(D, seed=3)).shape == (0, D)
    assert lib.layernorm(x0, scale=bf(D, seed=4), shift=bf(D, seed=5)).shape == (0, D)
    xq, sx = lib.quant_fp8_rowwise(x0)
    assert xq.shape == (0, D) and sx.shape == (0, 1)
    q0 = torch.empty(0, H * 128, dtype=torch.bfloat16, device="cuda")
    k, v = bf(70, H * 128, seed=6), bf(70, H * 128, seed=7)
    assert lib.attention(q0, k, v, H).shape == (0, H * 128)
    assert lib.attention(q0, k, v, H, variant=lib.ATTN_FAST).shape == (0, H * 128)
    with pytest.raises(lib.X2VError):
        lib.attention(bf(5, H * 128), k[:0], v[:0], H)  # no keys
    torch.cuda.synchronize()


@pytest.mark.parametrize("M", [1, 31, 33, 255, 257])
def test_ragged_rows_do_not_touch_neighbours(lib, M):
    """Outputs are written into the middle of a larger poisoned buffer: rows outside [0, M) and columns outside [0, N) keep the
    poison for every kernel family (both GEMM tilings, norms, attention variants)."""
    D, N, H = 512, 320, 2
    x = bf(M, D, seed=M)
    w, b = bf(N, D, seed=1, scale=0.05), bf(N, seed=2)
    for variant in (1, 2):
        big = torch.full((M + 2, N + 64), 7.0, dtype=torch.bfloat16, device="cuda")
        out = big[1 : M + 1, 32 : 32 + N]
        lib.gemm(x, w, b, out=out, variant=variant)
        ref = lib.gemm(x, w, b, variant=variant)
        assert torch.equal(out, ref)
        big[1 : M + 1, 32 : 32 + N] = 7.0
        assert (big == 7.0).all(), f"gemm variant {variant} wrote outside its [M, N] window"
    big = torch.full((M + 2, D), 7.0, dtype=torch.bfloat16, device="cuda")
    lib.rmsnorm(x, bf(D, seed=3), out=big[1 : M + 1])
    assert (big[0] == 7.0).all() and (big[M + 1] == 7.0).all()
    q, k, v = bf(M, H * 128, seed=10), bf(M + 3, H * 128, seed=11), bf(M + 3, H * 128, seed=12)
    ref = lib.attention(q, k, v, H)
    for variant in (lib.ATTN_FAST,):
        big = torch.full((M + 2, H * 128 + 128), 7.0, dtype=torch.bfloat16, device="cuda")
        out = big[1 : M + 1, : H * 128]
        lib.attention(q, k, v, H, out=out, variant=variant)
        assert_bf16_close(out.float().cpu(), ref.float().cpu(), ulps=2, atol=4e-3, bad_frac=1e-3, name=f"attention variant {variant} M={M}")
        assert (big[0] == 7.0).all() and (big[M + 1] == 7.0).all() and (big[:, H * 128 :] == 7.0).all()


def test_strided_views_match_contiguous(lib):
    """Fused-QKV style views (token stride 3*H*128) and a column-sliced GEMM operand give the results of their contiguous copies."""
    S, H = 130, 2
    qkv = bf(S, 3 * H * 128, seed=20)
    q, k, v = qkv[:, : H * 128], qkv[:, H * 128 : 2 * H * 128], qkv[:, 2 * H * 128 :]
    for variant in (0, lib.ATTN_FAST):
        a = lib.attention(q, k, v, H, variant=variant)
        b = lib.attention(q.contiguous(), k.contiguous(), v.contiguous(), H, variant=variant)
        assert torch.equal(a, b), f"attention variant {variant}: strided != contiguous"
    xw = bf(S, 1024, seed=21)
    w = bf(256, 512, seed=22, scale=0.05)
    assert torch.equal(lib.gemm(xw[:, 512:], w), lib.gemm(xw[:, 512:].contiguous(), w))


def test_argument_errors_raise(lib):
    D, N = 256, 128
    x, w = bf(8, D), bf(N, D, seed=1)
    with pytest.raises(lib.X2VError):
        lib.gemm(x.float(), w)  # dtype
    with pytest.raises(lib.X2VError):
        lib.gemm(x.cpu(), w)  # host tensor: there is no CPU path
    with pytest.raises(lib.X2VError):
        lib.gemm(x, bf(N, D + 8, seed=2))  # K mismatch
    with pytest.raises(lib.X2VError):
        lib.gemm(x, w, bias=torch.zeros(N, device="cuda"))  # fp32 bias where bf16 is read
    with pytest.raises(lib.X2VError):
        lib.gemm(x, w, bias=bf(N + 1))  # bias length
    with pytest.raises(lib.X2VError):
        lib.gemm(x, w, epilogue=lib.EPI_RESIDUAL)  # residual epilogue without a residual
    with pytest.raises(lib.X2VError):
        lib.gemm(x.t(), w)  # inner stride != 1
    with pytest.raises(lib.X2VError):
        lib.gemm(x[:, 4:], w[:, 4:])  # rows not 16-byte aligned
    with pytest.raises(lib.X2VError):
        lib.rmsnorm(x, bf(D + 8))  # weight length
    with pytest.raises(lib.X2VError):
        lib.layernorm(x, scale=bf(D))  # scale without shift
    with pytest.raises(lib.X2VError):
        lib.attention(bf(4, 2 * 64), bf(4, 2 * 64), bf(4, 2 * 64), 2, head_dim=64)  # only head_dim 128 is built
    with pytest.raises(lib.X2VError):
        lib.attention(bf(4, 256), bf(6, 256), bf(5, 256), 2)  # k / v row counts differ
    with pytest.raises(lib.X2VError):
        lib.attention(bf(4, 256), bf(6, 256), bf(6, 256), 2, variant=77)  # unknown variant
    # the failure is reported, not sticky: the next valid call works and the message names the cause
    try:
        lib.attention(bf(4, 256), bf(6, 256), bf(6, 256), 2, variant=77)
    except lib.X2VError as e:
        assert "variant" in str(e)
    assert torch.isfinite(lib.gemm(x, w).float()).all()


@pytest.mark.parametrize("M,D", [(700, 5120), (1500, 1536), (300, 3072), (9, 5120)])
def test_streaming_row_kernels_equal_per_row_kernels(lib, M, D):
    """The persistent ("streaming") forms of LayerNorm(+affine/+modulate) and of the fused q/k RMSNorm + RoPE must give the bits of
    the one-block-per-row forms (same arithmetic, different schedule) — the dispatcher picks by M, so a sequence-parallel shard and
    the unsharded run may take different forms.  Strided rows, partial last chunks (D / 8 not a multiple of 256) and M smaller than
    the resident grid included."""
    H = D // 128
    wide = bf(M, D + 64, seed=M)
    x = wide[:, :D]  # token stride D + 64
    w, b, sc, sh = bf(D, seed=1), bf(D, seed=2, scale=0.1), bf(D, seed=3, scale=0.1), bf(D, seed=4, scale=0.1)
    for kw in (dict(scale=sc, shift=sh), dict(weight=w, bias=b), dict(), dict(weight=w, bias=b, scale=sc, shift=sh), dict(weight=w)):
        a = lib.layernorm(x, variant=1, **kw)
        s = lib.layernorm(x, variant=2, **kw)
        assert torch.equal(a, s), f"layernorm {sorted(kw)}: streaming != per-row"
    g = torch.Generator().manual_seed(5)
    ang = torch.rand(1024, 64, generator=g) * 6.28
    cs = torch.stack([ang.cos(), ang.sin()], dim=-1).float().contiguous().cuda()
    grid = (3, 10, 12)  # 360 grid tokens: rows beyond them (s0 + row >= 360) take the identity rotation
    for round_mode in (lib.ROUND_FP32, lib.ROUND_REF):
        for wq, wk in ((w, b), (None, None)):
            outs = []
            for variant in (1, 2):
                q, k = x.clone(), bf(M, D, seed=6)
                lib.rmsnorm_rope_(q, k, wq, wk, cs, grid, H, s0=5, round_mode=round_mode, q_out_scale=0.1275, variant=variant)
                outs.append((q, k))
            assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), f"rmsnorm_rope mode {round_mode} norm={wq is not None}"
    with pytest.raises(lib.X2VError):
        lib.layernorm(bf(8, 256), variant=2)  # the streaming form covers 512 < D <= 8192
