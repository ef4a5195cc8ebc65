"""MXFP8 quantise + block-scaled GEMM (lightx2v_amd/csrc/mx.hip through the C-ABI) against the CPU oracle (oracle/mx_oracle.py) and
against the reference package's own acceptance test (lightx2v_kernel/test/mxfp8_mxfp8/test_mxfp8_quant.py:20-37:
error(mm_pred, linear(a, w, bias)) < 1e-2 on randn inputs, bias = rand * 10)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from lightx2v_amd import lib as L

    L.init()
    return L


@pytest.mark.parametrize("M,K", [(1, 128), (5, 384), (257, 1536), (1000, 5120), (64, 13824)])
def test_quant_bit_exact(lib, M, K):
    from oracle import mx_oracle as MX

    gen = torch.Generator().manual_seed(M * 7 + K)
    x = (torch.randn(M, K, generator=gen) * torch.logspace(-4, 4, M).unsqueeze(1)).to(torch.bfloat16)
    if M > 4:
        x[3, :32] = 0  # all-zero block → scale byte 0, zero elements
        x[4, 32 : K if K < 64 else 64] = 448.0
    q, sc = lib.quant_mxfp8(x.cuda())
    rq, rs = MX.quant_mxfp8(x)
    assert sc.shape == (K // 128, M, 4)
    sc = lib.mx_scales_rowmajor(sc)
    assert torch.equal(sc.cpu(), rs), "scale bytes differ"
    assert torch.equal(lib.mx_scales_tiled(rs.cuda()), lib.quant_mxfp8(x.cuda())[1])
    assert torch.equal(q.cpu().view(torch.uint8), rq.view(torch.uint8)), "e4m3 elements differ"
    # strided input / output views (token stride larger than K)
    if K >= 64:
        big = torch.zeros(M, K + 64, dtype=torch.bfloat16)
        big[:, :K] = x
        q2, sc2 = lib.quant_mxfp8(big.cuda()[:, :K])
        assert torch.equal(lib.mx_scales_rowmajor(sc2).cpu(), rs) and torch.equal(q2.cpu().view(torch.uint8), rq.view(torch.uint8))


@pytest.mark.parametrize("variant", [1, 2])  # 128x128-tile kernel / 256x256-tile ping-pong kernel (forced; 0 picks by shape)
@pytest.mark.parametrize("M,K,N", [(1, 128, 8), (130, 256, 136), (257, 1536, 1536), (512, 5120, 1280), (333, 1024, 72), (700, 2176, 520)])
def test_gemm_vs_oracle(lib, M, K, N, variant):
    """Same quantised operands on both sides: the only differences are fp32 accumulation order and the final bf16 rounding."""
    from oracle import mx_oracle as MX
    from tests.util import assert_bf16_close

    gen = torch.Generator().manual_seed(M + K + N)
    a = (torch.randn(M, K, generator=gen) * torch.logspace(-2, 2, M).unsqueeze(1)).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=gen) * torch.logspace(-1, 1, N).unsqueeze(1) / K**0.5).to(torch.bfloat16)
    # make the block scales differ strongly along K as well (exercises every scale byte of a K-tile)
    ramp = torch.logspace(-3, 3, K // 32).repeat_interleave(32)
    a = (a.float() * ramp).to(torch.bfloat16)
    w = (w.float() / ramp).to(torch.bfloat16)
    bias = torch.randn(N, generator=gen).to(torch.bfloat16)
    qa, sa = MX.quant_mxfp8(a)
    qw, sw = MX.quant_mxfp8(w)
    alpha = torch.tensor(0.75, dtype=torch.float32)
    ref = MX.gemm_mxfp8(qa, sa, qw, sw, alpha=0.75, bias=bias)
    sa_t, sw_t = lib.mx_scales_tiled(sa.cuda()), lib.mx_scales_tiled(sw.cuda())
    got = lib.gemm_mxfp8(qa.cuda(), sa_t, qw.cuda(), sw_t, alpha=alpha.cuda(), bias=bias.cuda(), variant=variant)
    assert_bf16_close(got, ref, ulps=1, atol=1e-2 * ref.float().abs().mean().item(), bad_frac=1e-3, name=f"mxfp8 gemm {M}x{K}x{N}")
    got1 = lib.gemm_mxfp8(qa.cuda(), sa_t, qw.cuda(), sw_t, variant=variant)
    assert_bf16_close(got1, MX.gemm_mxfp8(qa, sa, qw, sw), ulps=1, atol=1e-2 * ref.float().abs().mean().item(), bad_frac=1e-3, name="mxfp8 gemm (no alpha/bias)")


def test_exact_on_lossless_inputs(lib):
    """Small integers and power-of-two block maxima quantise without loss, and their products sum exactly in fp32: bit-exact GEMM."""
    gen = torch.Generator().manual_seed(9)
    a = torch.randint(-8, 9, (200, 384), generator=gen).to(torch.bfloat16)
    w = torch.randint(-8, 9, (264, 384), generator=gen).to(torch.bfloat16)
    a[:, 128:256] *= 2.0**10  # different scale bytes per K block
    w[:, 128:256] *= 2.0**-10
    qa, sa = lib.quant_mxfp8(a.cuda())
    qw, sw = lib.quant_mxfp8(w.cuda())
    for variant in (1, 2):
        y = lib.gemm_mxfp8(qa, sa, qw, sw, variant=variant)
        assert torch.equal(y.float().cpu(), (a.float() @ w.float().T).to(torch.bfloat16).float()), variant


@pytest.mark.parametrize("m,k,n", [(257, 1536, 1536), (1024, 5120, 5120), (13325, 3072, 1536), (512, 8960, 1536)])
def test_reference_acceptance(m, k, n):
    """The reference's own test (test_mxfp8_quant.py:20-37) through the mirrored Python API."""
    from lightx2v_amd.mx import cutlass_scaled_mxfp8_mm, scaled_fp8_quant
    from oracle.mx_oracle import snr_error

    torch.manual_seed(0)
    activation = torch.randn(m, k, dtype=torch.bfloat16, device="cuda")
    aq, asc = scaled_fp8_quant(activation)
    weight = torch.randn(n, k, dtype=torch.bfloat16, device="cuda")
    wq, wsc = scaled_fp8_quant(weight)
    bias = torch.rand(1, n, dtype=torch.bfloat16, device="cuda") * 10
    alpha = torch.tensor(1.0, device="cuda", dtype=torch.float32)
    pred = cutlass_scaled_mxfp8_mm(aq, wq, asc, wsc, alpha=alpha, bias=bias)
    real = torch.nn.functional.linear(activation, weight, bias=bias).to(torch.bfloat16)
    assert pred.shape == (m, n) and pred.dtype == torch.bfloat16
    assert snr_error(pred, real) < 1e-2


def test_argument_errors(lib):
    x = torch.zeros(4, 96, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(lib.X2VError):
        lib.quant_mxfp8(x)  # K % 128 != 0
    q = torch.zeros(4, 128, dtype=torch.float8_e4m3fn, device="cuda")
    s = torch.zeros(4, 4, dtype=torch.uint8, device="cuda")
    with pytest.raises(lib.X2VError):
        lib.gemm_mxfp8(q, s, q, s)  # scale tensors not in the [K/128, rows, 4] layout


def test_scale_byte_to_k_block_association(lib):
    """The hardware applies scale byte j of a row's K-tile dword to k block j and to nothing else (this pins the fragment k order the
    kernel has to use: with data only in block j, doubling byte i doubles the result iff i == j)."""
    M = N = 32
    K = 128
    for side in ("a", "b"):
        for j in range(4):
            for i in range(4):
                a, b = torch.zeros(M, K), torch.zeros(N, K)
                (a if side == "a" else b)[:, 32 * j : 32 * j + 32] = 1
                (b if side == "a" else a)[:] = 1
                sa = torch.full((M, 4), 127, dtype=torch.uint8)
                sb = torch.full((N, 4), 127, dtype=torch.uint8)
                (sa if side == "a" else sb)[:, i] = 128
                for variant in (1, 2):
                    y = lib.gemm_mxfp8(a.to(torch.float8_e4m3fn).cuda(), lib.mx_scales_tiled(sa.cuda()), b.to(torch.float8_e4m3fn).cuda(), lib.mx_scales_tiled(sb.cuda()),
                                       variant=variant).float().cpu()
                    assert (y == (64.0 if i == j else 32.0)).all(), (side, j, i, variant, y[0, 0].item())


def test_epilogues_and_operator_class(lib):
    """The fused epilogues of the MXFP8 GEMM equal the separate ops applied to the plain MXFP8 product; the operator class runs the
    tiny Wan model within fp8 tolerance of the bf16 one."""
    from tests.util import assert_bf16_close, assert_rel

    gen = torch.Generator().manual_seed(21)
    M, K, N = 300, 512, 264
    x = torch.randn(M, K, generator=gen).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=gen) / K**0.5).to(torch.bfloat16).cuda()
    b = torch.randn(N, generator=gen).to(torch.bfloat16).cuda()
    xq, sx = lib.quant_mxfp8(x)
    wq, sw = lib.quant_mxfp8(w)
    y = lib.gemm_mxfp8(xq, sx, wq, sw, bias=b)
    g = lib.gemm_mxfp8(xq, sx, wq, sw, bias=b, epilogue=lib.EPI_GELU_TANH)
    y2 = lib.gemm_mxfp8(xq, sx, wq, sw, bias=b, variant=2)
    assert_bf16_close(g, torch.nn.functional.gelu(y2.float(), approximate="tanh").to(torch.bfloat16), ulps=1, atol=2e-3, bad_frac=2e-3, name="mx gemm+gelu")
    res = torch.randn(M, N, generator=gen).to(torch.bfloat16).cuda()
    gate = (torch.randn(1, N, generator=gen) * 0.5).to(torch.bfloat16).cuda()
    want = res.clone()
    want.add_(y2 * gate.squeeze(0))
    r = res.clone()
    out = lib.gemm_mxfp8(xq, sx, wq, sw, bias=b, epilogue=lib.EPI_RESIDUAL, resid=r, gate=gate)
    assert out.data_ptr() == r.data_ptr()
    assert_bf16_close(r, want, ulps=1, atol=6e-3, bad_frac=2e-3, name="mx gemm+gate-residual")

    from lightx2v_amd import scheduler, synth, wan

    dims, wl = synth.WAN_DIMS["wan-tiny"], synth.WORKLOADS["wan-tiny"]
    wd = {k: v.cuda() for k, v in synth.synth_wan_weights(dims, seed=0).items()}
    lat, ctx, ctx_null = synth.synth_inputs(dims, wl["target_shape"])
    inputs = {"text_encoder_output": {"context": [c.cuda() for c in ctx], "context_null": [c.cuda() for c in ctx_null]}}
    outs = []
    for mm in ({"mm_type": "Hip-bf16"}, {"mm_type": "W-mxfp8-A-mxfp8-dynamic-Hip", "weight_auto_quant": True}):
        cfg = wan.default_config(dims, target_shape=wl["target_shape"], target_video_length=wl["frames"], infer_steps=4, mm_config=mm)
        model = wan.WanModel(cfg, wd)
        sch = scheduler.WanScheduler(cfg, device="cuda")
        sch.prepare(latents=lat)
        model.set_scheduler(sch)
        sch.step_pre(0)
        model.infer(inputs)
        outs.append(sch.noise_pred.float().cpu())
    assert_rel(outs[1], outs[0], 1e-1, "wan-tiny forward, MXFP8 linears vs bf16")
    from tests.util import rel_l2

    assert rel_l2(outs[1], outs[0]) > 1e-4
