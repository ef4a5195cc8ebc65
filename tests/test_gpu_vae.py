"""Wan VAE decode on the HIP kernels (csrc/vae.hip, lightx2v_amd/vae.py) against
  * the fixture generated from the unmodified reference (tests/golden/wan_vae_tiny.safetensors), and
  * the CPU oracle (oracle/wan_vae_oracle.py, pinned bit-exact to that fixture) at the real channel widths.
fp32 end to end; the only differences are summation order inside the convolutions (MFMA k-blocking vs oneDNN) and
fast-math exp in SiLU/softmax: tolerance |d| <= 2e-3 absolute on outputs in [-1, 1] and relative L2 <= 1e-3."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _check(got, ref, what, atol=2e-3, rel=1e-3):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    d = (got - ref).abs().max().item()
    r = ((got - ref).norm() / ref.norm().clamp_min(1e-30)).item()
    assert d <= atol and r <= rel, f"{what}: max abs {d:.3e}, rel L2 {r:.3e}"


def test_vae_conv_matches_conv3d():
    from lightx2v_amd import lib

    g = torch.Generator().manual_seed(0)
    for (T, H, W, Cin, Cout, k, nc) in [(2, 9, 11, 32, 96, (3, 3, 3), 2), (1, 6, 7, 16, 40, (3, 3, 3), 1), (3, 5, 20, 64, 3, (3, 3, 3), 0), (2, 17, 16, 96, 128, (1, 3, 3), 0),
                                        (2, 4, 9, 32, 64, (3, 1, 1), 2)]:
        kt, kh, kw = k
        x = torch.randn(T, H, W, Cin, generator=g)
        cache = torch.randn(kt - 1, H, W, Cin, generator=g) if kt > 1 else None
        if cache is not None and nc < kt - 1:
            cache[: kt - 1 - nc] = 0
        w = torch.randn(Cout, Cin, kt, kh, kw, generator=g) / (Cin * kt * kh * kw) ** 0.5
        b = torch.randn(Cout, generator=g)
        resid = torch.randn(T, H, W, Cout, generator=g)
        xin = x if cache is None else torch.cat([cache, x], 0)
        xin = F.pad(xin.permute(3, 0, 1, 2), (kw // 2, kw // 2, kh // 2, kh // 2, 0, 0))
        ref = F.conv3d(xin.unsqueeze(0), w, b)[0].permute(1, 2, 3, 0) + resid
        ph, pw = kh // 2, kw // 2
        buf = torch.zeros(kt - 1 + T, H + 2 * ph, W + 2 * pw, Cin, device="cuda")
        buf[:, ph : ph + H, pw : pw + W] = (x if cache is None else torch.cat([cache, x], 0)).cuda()
        out = torch.empty(T, H, W, Cout, device="cuda")
        wcl = w.permute(0, 2, 3, 4, 1).contiguous().cuda()
        lib.vae_conv(buf, ((H + 2 * ph) * (W + 2 * pw) * Cin, (W + 2 * pw) * Cin, Cin), wcl, out, T, H, W, bias=b.cuda(), resid=resid.cuda())
        _check(out, ref, f"conv {k} Cin={Cin} Cout={Cout}", atol=2e-4, rel=1e-5)
    # time-split epilogue + clamp
    T, H, W, C = 2, 5, 8, 32
    x = torch.randn(T, H, W, C, generator=g)
    w = torch.randn(2 * C, C, 3, 1, 1, generator=g) / (3 * C) ** 0.5
    b = torch.randn(2 * C, generator=g)
    y = F.conv3d(F.pad(x.permute(3, 0, 1, 2), (0, 0, 0, 0, 2, 0)).unsqueeze(0), w, b)[0]  # [2C, T, H, W]
    y = y.reshape(2, C, T, H, W)
    ref = torch.stack((y[0], y[1]), dim=2).reshape(C, 2 * T, H, W).permute(1, 2, 3, 0).clamp(-1, 1)
    buf = torch.zeros(2 + T, H, W, C, device="cuda")
    buf[2:] = x.cuda()
    out = torch.empty(2 * T, H, W, C, device="cuda")
    lib.vae_conv(buf, (H * W * C, W * C, C), w.permute(0, 2, 3, 4, 1).contiguous().cuda(), out, T, H, W, bias=b.cuda(), flags=lib.VCONV_TSPLIT | lib.VCONV_CLAMP)
    _check(out, ref, "time-split + clamp", atol=2e-4, rel=1e-5)


def test_vae_prep_and_softmax():
    from lightx2v_amd import lib

    g = torch.Generator().manual_seed(1)
    for C in (16, 96, 192, 384):
        x = torch.randn(2, 5, 6, C, generator=g) * 2
        gamma = 1 + 0.1 * torch.randn(C, generator=g)
        ref = F.silu(F.normalize(x, dim=-1) * C**0.5 * gamma)
        y = torch.zeros(2, 7, 8, C, device="cuda")
        lib.vae_prep(x.cuda(), y[:, 1:, 1:], (7 * 8 * C, 8 * C), gamma=gamma.cuda(), silu=True)
        _check(y[:, 1:6, 1:7], ref, f"norm+silu C={C}", atol=1e-5, rel=1e-6)
        assert y[:, 0].abs().max() == 0 and y[:, :, 0].abs().max() == 0 and y[:, 6].abs().max() == 0 and y[:, :, 7].abs().max() == 0
        up = torch.zeros(2, 12, 14, C, device="cuda")
        lib.vae_prep(x.cuda(), up[:, 1:, 1:], (12 * 14 * C, 14 * C), upsample=True)
        refu = F.interpolate(x.permute(0, 3, 1, 2), scale_factor=(2.0, 2.0), mode="nearest-exact").permute(0, 2, 3, 1)
        assert torch.equal(up[:, 1:11, 1:13].cpu(), refu)
    a, b = torch.rand(16, generator=g) + 0.5, torch.randn(16, generator=g)
    z = torch.randn(3, 4, 4, 16, generator=g)
    y = torch.empty(3, 4, 4, 16, device="cuda")
    lib.vae_prep(z.cuda(), y, (256, 64), a=a.cuda(), b=b.cuda())
    assert torch.equal(y.cpu(), z / a + b)
    s = torch.randn(37, 64, generator=g) * 3
    got = lib.softmax_rows_(s.clone().cuda(), 0.25)
    _check(got, torch.softmax(s * 0.25, dim=-1), "softmax rows", atol=1e-6, rel=1e-5)


def test_vae_decode_matches_reference_fixture():
    from safetensors.torch import load_file

    from lightx2v_amd import synth, vae

    gld = load_file(os.path.join(GOLDEN, "wan_vae_tiny.safetensors"))
    dim, seed = int(gld["dim"]), int(gld["seed"])
    sd = synth.synth_wan_vae_weights(dim=dim, seed=seed)
    m = vae.WanVAE(sd, dim=dim, conv16=False)  # the fp32 matrix instruction
    out = m.decode(gld["z"].cuda())
    assert out.shape == (1, 3, 9, 64, 64)
    _check(out[0], gld["decoded"], "WanVAE.decode vs reference fixture")
    # decoding again must give the same answer (caches are cleared, buffers reused)
    out2 = m.decode(gld["z"].cuda())
    assert torch.equal(out, out2)


def test_vae_decode_frame_batched_is_bit_identical():
    """chunk_frames latent frames per decoder pass (default 4) against the reference's one-frame-at-a-time chunking (vae.py:722-736):
    every kernel reduces an output pixel in an order independent of the number of frames in the launch and the conv caches are the
    leading frames of the input buffers, so the decoded video must be EQUAL bit for bit — fp32 and the opt-in fp16-operand form, with
    chunk sizes that do / do not divide T - 1 (T = 6: chunks 1+5, 1+2+2+1, 1+3+2, 1+4+1)."""
    from safetensors.torch import load_file

    from lightx2v_amd import synth, vae

    gld = load_file(os.path.join(GOLDEN, "wan_vae_tiny.safetensors"))
    dim, seed = int(gld["dim"]), int(gld["seed"])
    sd = synth.synth_wan_vae_weights(dim=dim, seed=seed)
    z = torch.randn(16, 6, 8, 8, generator=torch.Generator().manual_seed(5)).cuda()
    for conv16 in (False, True, "split"):
        ref = vae.WanVAE(sd, dim=dim, conv16=conv16, chunk_frames=1).decode(z)
        assert ref.shape == (1, 3, 21, 64, 64) and torch.isfinite(ref).all()
        for g in (2, 3, 4, 5, 8):
            m = vae.WanVAE(sd, dim=dim, conv16=conv16, chunk_frames=g)
            out = m.decode(z)
            assert torch.equal(out, ref), f"chunk_frames={g} conv16={conv16}: max |d| = {(out - ref).abs().max().item():.3e}"
            assert torch.equal(m.decode(z), ref)  # second decode on the grown buffers
    # the committed fixture through the default construction (hi/lo split operands, two latent frames per pass)
    out = vae.WanVAE(sd, dim=dim).decode(gld["z"].cuda())
    _check(out[0], gld["decoded"], "WanVAE.decode (defaults) vs reference fixture")


def test_vae_decode_fp16_operands_opt_in():
    """Opt-in fast decode (WanVAE(conv16=True)): fp16 operands for the 3x3(x3) convolutions, everything else fp32 — against the fp32
    fixture generated from the reference.  Stated tolerance on outputs in [-1, 1]: |d| <= 2e-2, relative L2 <= 1e-2.  The tiny model's
    32- and 128-channel stages exercise the channel padding (32 -> 64) of the 16-bit kernel's operand buffers; dim = 96 does 96 -> 128."""
    from safetensors.torch import load_file

    from lightx2v_amd import lib, synth, vae
    from oracle import wan_vae_oracle as V

    # the 16-bit convolutions themselves (halo-tiled kernel for 3x3 spatial taps, per-tap kernel otherwise or with flag 4) against conv3d
    # on the same fp16-rounded operands (fp32 accumulate): tight.  Shapes: ragged tiles in H and W, padded channels, Cout 3, 2-D kernels.
    g = torch.Generator().manual_seed(2)
    for (T, H, W, Cin, Cout, kt, kh, kw) in [(2, 9, 11, 96, 160, 3, 3, 3), (2, 9, 40, 96, 160, 3, 3, 3), (1, 17, 33, 64, 3, 3, 3, 3), (2, 8, 32, 128, 96, 1, 3, 3),
                                             (1, 21, 70, 64, 64, 3, 3, 3), (2, 6, 20, 64, 32, 3, 1, 1)]:
        x = torch.randn(T, H, W, Cin, generator=g).half()
        w = (torch.randn(Cout, Cin, kt, kh, kw, generator=g) / (kt * kh * kw * Cin) ** 0.5).half()
        b = torch.randn(Cout, generator=g)
        res = torch.randn(T, H, W, Cout, generator=g)
        ph, pw = kh // 2, kw // 2
        xin = F.pad(x.float().permute(3, 0, 1, 2), (pw, pw, ph, ph, kt - 1, 0))
        ref = F.conv3d(xin.unsqueeze(0), w.float(), b)[0].permute(1, 2, 3, 0) + res
        cp = (Cin + 63) // 64 * 64
        buf = torch.zeros(kt - 1 + T, H + 2 * ph, W + 2 * pw, cp, dtype=torch.float16, device="cuda")
        buf[kt - 1 :, ph : ph + H, pw : pw + W, :Cin] = x.cuda()
        w16 = torch.zeros(Cout, kt, kh, kw, cp, dtype=torch.float16, device="cuda")
        w16[..., :Cin] = w.permute(0, 2, 3, 4, 1).cuda()
        strides = ((H + 2 * ph) * (W + 2 * pw) * cp, (W + 2 * pw) * cp, cp)
        for flags in (0, 4):
            out = torch.full((T, H, W, Cout), float("nan"), device="cuda")
            lib.vae_conv16(buf, strides, w16, out, T, H, W, bias=b.cuda(), resid=res.cuda(), flags=flags)
            _check(out, ref, f"fp16-operand conv {(kt, kh, kw)} {H}x{W} Cin={Cin} Cout={Cout} flags={flags}", atol=3e-4, rel=1e-5)
    # norm + SiLU written as fp16 into a channel-padded buffer
    C = 96
    xx = torch.randn(2, 5, 6, C, generator=g) * 2
    gamma = 1 + 0.1 * torch.randn(C, generator=g)
    y = torch.zeros(2, 7, 8, 128, dtype=torch.float16, device="cuda")
    lib.vae_prep(xx.cuda(), y[:, 1:, 1:], (7 * 8 * 128, 8 * 128), gamma=gamma.cuda(), silu=True)
    refp = F.silu(F.normalize(xx, dim=-1) * C**0.5 * gamma)
    _check(y[:, 1:6, 1:7, :C], refp.half(), "norm+silu -> fp16 padded buffer", atol=2e-3, rel=1e-3)
    assert y[..., C:].abs().max() == 0 and y[:, 0].abs().max() == 0
    # whole decodes
    gld = load_file(os.path.join(GOLDEN, "wan_vae_tiny.safetensors"))
    dim, seed = int(gld["dim"]), int(gld["seed"])
    m = vae.WanVAE(synth.synth_wan_vae_weights(dim=dim, seed=seed), dim=dim, conv16=True)
    out = m.decode(gld["z"].cuda())
    assert m.model.decoder.w16, "no convolution took the 16-bit path"
    _check(out[0], gld["decoded"], "WanVAE.decode (fp16 conv operands) vs reference fixture", atol=2e-2, rel=1e-2)
    sd = synth.synth_wan_vae_weights(dim=96, seed=3)
    z = torch.randn(16, 2, 6, 8, generator=torch.Generator().manual_seed(6))
    mean, inv_std = torch.tensor(synth.WAN_VAE_MEAN), 1.0 / torch.tensor(synth.WAN_VAE_STD)
    with torch.no_grad():
        ref96 = V.wan_vae_decode(sd, z, mean, inv_std, dim=96)
    out96 = vae.WanVAE(sd, dim=96, conv16=True).decode(z.cuda())
    _check(out96[0], ref96, "WanVAE.decode dim 96 (fp16 conv operands) vs oracle", atol=2e-2, rel=1e-2)


def test_vae_decode_split_fp16_is_fp32_grade():
    """WanVAE(conv16="split"): activations and weights of the 3x3(x3) convolutions as hi + lo fp16 pairs (~22 mantissa bits), three 16-bit
    products per fp32 product, fp32 accumulation — held to the SAME tolerance as the all-fp32 decode (|d| <= 2e-3, relative L2 <= 1e-3;
    the fp16-operand form needs 2e-2 / 1e-2), against the reference fixture and against the oracle at the released widths; the split of a
    single convolution against fp32 conv3d of the unrounded operands; bit-identical across frame chunkings like the other forms."""
    from safetensors.torch import load_file

    from lightx2v_amd import lib, synth, vae
    from oracle import wan_vae_oracle as V

    g = torch.Generator().manual_seed(7)
    for (T, H, W, Cin, Cout, kt, kh, kw) in [(2, 9, 11, 96, 160, 3, 3, 3), (1, 17, 33, 64, 3, 3, 3, 3), (2, 8, 32, 128, 96, 1, 3, 3)]:
        x = torch.randn(T, H, W, Cin, generator=g) * 3
        w = torch.randn(Cout, Cin, kt, kh, kw, generator=g) / (kt * kh * kw * Cin) ** 0.5
        b = torch.randn(Cout, generator=g)
        ph, pw = kh // 2, kw // 2
        xin = F.pad(x.permute(3, 0, 1, 2), (pw, pw, ph, ph, kt - 1, 0))
        ref = F.conv3d(xin.unsqueeze(0).double(), w.double(), b.double())[0].permute(1, 2, 3, 0).float()
        cp = (3 * Cin + 63) // 64 * 64
        buf = torch.zeros(kt - 1 + T, H + 2 * ph, W + 2 * pw, cp, dtype=torch.float16, device="cuda")
        strides = ((H + 2 * ph) * (W + 2 * pw) * cp, (W + 2 * pw) * cp, cp)
        lib.vae_prep(x.cuda(), buf[kt - 1 :, ph:, pw:], strides[:2], split=True)
        hi = x.half()
        # planes [hi | hi * 2^-12 | lo] against weights [hi | lo * 2^12 | hi]: the power-of-two pair keeps the weights' lo halves normal fp16 numbers
        assert torch.equal(buf[kt - 1 :, ph : ph + H, pw : pw + W, :Cin].cpu(), hi)
        assert torch.equal(buf[kt - 1 :, ph : ph + H, pw : pw + W, Cin : 2 * Cin].cpu(), (hi.float() / 4096).half())
        assert torch.equal(buf[kt - 1 :, ph : ph + H, pw : pw + W, 2 * Cin : 3 * Cin].cpu(), (x - hi.float()).half())
        wcl = w.permute(0, 2, 3, 4, 1).contiguous()
        whi = wcl.half()
        wlo = ((wcl - whi.float()) * 4096).half()
        assert (wlo.float().abs() >= 2.0 ** -14).float().mean().item() > 0.98 and ((wcl - whi.float()).half().float().abs() < 2.0 ** -14).float().mean().item() > 0.5, \
            "the scaled lo halves must be normal fp16 numbers (most unscaled ones are subnormal)"
        w16 = torch.zeros(Cout, kt, kh, kw, cp, dtype=torch.float16)
        w16[..., :Cin], w16[..., Cin : 2 * Cin], w16[..., 2 * Cin : 3 * Cin] = whi, wlo, whi
        out = torch.full((T, H, W, Cout), float("nan"), device="cuda")
        lib.vae_conv16(buf, strides, w16.cuda(), out, T, H, W, bias=b.cuda())
        _check(out, ref, f"split conv {(kt, kh, kw)} Cin={Cin} Cout={Cout}", atol=2e-5, rel=2e-6)
    # SMALL activations, the common case behind RMS-norm + SiLU: for |x| < 0.25 the middle plane hi * 2^-12 is an fp16 SUBNORMAL.  If the
    # matrix unit flushed subnormal inputs, the xh.wl term would vanish for those pixels and the result would fall back to ~11 bits; it must stay
    # fp32-grade (same bound as above) with activations in [1e-3, 0.25] and weights of the usual 1/sqrt(fan-in) scale
    T, H, W, Cin, Cout, kt, kh, kw = 2, 8, 16, 96, 64, 3, 3, 3
    x = (torch.rand(T, H, W, Cin, generator=g) * (0.25 - 1e-3) + 1e-3) * (torch.randint(0, 2, (T, H, W, Cin), generator=g) * 2 - 1)
    w = torch.randn(Cout, Cin, kt, kh, kw, generator=g) / (kt * kh * kw * Cin) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    xin = F.pad(x.permute(3, 0, 1, 2), (1, 1, 1, 1, kt - 1, 0))
    ref = F.conv3d(xin.unsqueeze(0).double(), w.double(), b.double())[0].permute(1, 2, 3, 0).float()
    cp = (3 * Cin + 63) // 64 * 64
    buf = torch.zeros(kt - 1 + T, H + 2, W + 2, cp, dtype=torch.float16, device="cuda")
    strides = ((H + 2) * (W + 2) * cp, (W + 2) * cp, cp)
    lib.vae_prep(x.cuda(), buf[kt - 1 :, 1:, 1:], strides[:2], split=True)
    mid = buf[kt - 1 :, 1 : 1 + H, 1 : 1 + W, Cin : 2 * Cin].float().abs()
    assert ((mid > 0) & (mid < 2.0 ** -14)).float().mean().item() > 0.95, "this case must put the middle plane into the fp16 subnormal range"
    wcl = w.permute(0, 2, 3, 4, 1).contiguous()
    whi = wcl.half()
    w16 = torch.zeros(Cout, kt, kh, kw, cp, dtype=torch.float16)
    w16[..., :Cin], w16[..., Cin : 2 * Cin], w16[..., 2 * Cin : 3 * Cin] = whi, ((wcl - whi.float()) * 4096).half(), whi
    out = torch.full((T, H, W, Cout), float("nan"), device="cuda")
    lib.vae_conv16(buf, strides, w16.cuda(), out, T, H, W, bias=b.cuda())
    _check(out, ref, "split conv, activations in the subnormal range of the middle plane", atol=2e-5, rel=2e-6)
    # large-magnitude activations (beyond the fp16 range: hi saturates at 65504, lo carries the rest) against tiny weights (1e-3
    # scale: every unscaled lo half would be a subnormal) — still fp32-grade relative to the result's scale
    T, H, W, Cin, Cout, kt, kh, kw = 1, 8, 16, 64, 32, 1, 3, 3
    x = torch.randn(T, H, W, Cin, generator=g) * 300
    x[0, 3, 5, :8] = torch.tensor([7.0e4, -9.0e4, 6.6e4, 1.2e5, -6.55e4, 3.0e4, -1.0e5, 6.5504e4])
    w = torch.randn(Cout, Cin, kt, kh, kw, generator=g) * 1e-3
    xin = F.pad(x.permute(3, 0, 1, 2), (1, 1, 1, 1, 0, 0))
    ref = F.conv3d(xin.unsqueeze(0).double(), w.double())[0].permute(1, 2, 3, 0).float()
    cp = (3 * Cin + 63) // 64 * 64
    buf = torch.zeros(T, H + 2, W + 2, cp, dtype=torch.float16, device="cuda")
    strides = ((H + 2) * (W + 2) * cp, (W + 2) * cp, cp)
    lib.vae_prep(x.cuda(), buf[:, 1:, 1:], strides[:2], split=True)
    assert torch.isfinite(buf.float()).all(), "hi must saturate, not overflow"
    wcl = w.permute(0, 2, 3, 4, 1).contiguous()
    whi = wcl.half()
    w16 = torch.zeros(Cout, kt, kh, kw, cp, dtype=torch.float16)
    w16[..., :Cin], w16[..., Cin : 2 * Cin], w16[..., 2 * Cin : 3 * Cin] = whi, ((wcl - whi.float()) * 4096).half(), whi
    out = torch.full((T, H, W, Cout), float("nan"), device="cuda")
    lib.vae_conv16(buf, strides, w16.cuda(), out, T, H, W, bias=torch.zeros(Cout).cuda())
    err = (out.cpu() - ref).abs()
    near = torch.zeros(T, H, W, dtype=torch.bool)
    near[0, 2:5, 4:7] = True  # outputs whose 3x3 window holds the out-of-range pixel
    scale_far, scale_near = ref[~near].abs().max().item(), ref[near].abs().max().item()
    # in range: fp32-grade.  Out of range: hi saturates, lo = x - 65504 is O(x), so the dropped lo.lo product is 2^-12 of that pixel's term — the
    # guard buys a finite, 11-bit-accurate result where an unguarded split would produce inf / NaN
    assert err[~near].max().item() <= 4e-6 * scale_far, (err[~near].max().item(), scale_far)
    assert err[near].max().item() <= 2.0 ** -10 * scale_near, (err[near].max().item(), scale_near)
    gld = load_file(os.path.join(GOLDEN, "wan_vae_tiny.safetensors"))
    dim, seed = int(gld["dim"]), int(gld["seed"])
    sd_t = synth.synth_wan_vae_weights(dim=dim, seed=seed)
    m = vae.WanVAE(sd_t, dim=dim, conv16="split")
    out = m.decode(gld["z"].cuda())
    assert m.model.decoder.w16, "no convolution took the 16-bit path"
    _check(out[0], gld["decoded"], "WanVAE.decode (hi/lo split) vs reference fixture")
    assert torch.equal(out, vae.WanVAE(sd_t, dim=dim, conv16="split", chunk_frames=1).decode(gld["z"].cuda()))
    sd = synth.synth_wan_vae_weights(dim=96, seed=3)
    z = torch.randn(16, 2, 6, 8, generator=torch.Generator().manual_seed(6))
    mean, inv_std = torch.tensor(synth.WAN_VAE_MEAN), 1.0 / torch.tensor(synth.WAN_VAE_STD)
    with torch.no_grad():
        ref96 = V.wan_vae_decode(sd, z, mean, inv_std, dim=96)
    _check(vae.WanVAE(sd, dim=96, conv16="split").decode(z.cuda())[0], ref96, "WanVAE.decode dim 96 (hi/lo split) vs oracle")


def test_vae_decode_real_widths_vs_oracle():
    """dim = 96 (384/384/384/192/96 channels, the released Wan2.1 VAE widths) on a small latent; checker = CPU oracle."""
    from lightx2v_amd import synth, vae
    from oracle import wan_vae_oracle as V

    sd = synth.synth_wan_vae_weights(dim=96, seed=3)
    g = torch.Generator().manual_seed(11)
    z = torch.randn(16, 2, 4, 8, generator=g)
    mean, inv_std = torch.tensor(synth.WAN_VAE_MEAN), 1.0 / torch.tensor(synth.WAN_VAE_STD)
    with torch.no_grad():
        ref = V.wan_vae_decode(sd, z, mean, inv_std, dim=96)
    out = vae.WanVAE(sd, dim=96, conv16=False).decode(z.cuda())
    assert out.shape == (1, 3, 5, 32, 64)
    _check(out[0], ref, "WanVAE.decode (dim 96) vs oracle")


def test_vae_conv16_128x96_kernel_against_conv3d_and_the_other_kernels():
    """The 128-pixel x 96-cout convolution kernel (csrc/vae16g.hip: what x2v_vae_conv_f16 runs for 3x3 kernels with Cout % 96 == 0, i.e. every 3x3(x3)
    convolution of the Wan decoder but its head) against F.conv3d on the same fp16-rounded operands (fp32 accumulate; tolerance as the other 16-bit
    kernels' test: |d| <= 3e-4, relative L2 <= 1e-5) and against the 64-pixel halo kernel (flag 8) and the per-tap kernel (flag 4), which reduce in another
    order.  Shapes: ragged tiles in H and W (tile = 16 x 32 pixels), one to four cout tiles, kt 1 and 3 with a non-zero 2-frame cache, 2..12 32-channel
    slabs (odd and even counts: both halo buffers end a tile), residual, clamp, the zero-tail flag (288 channels in a 320-channel buffer: bit-identical
    with and without the flag), several tiles per workgroup (more tiles than CUs), and the one-cout-block form the decoder's 3-channel head takes
    (Cout 3 and 16: weight rows beyond Cout masked, element-wise stores); and the 128-cout form of the same kernel (Cout % 128 == 0 and not a multiple of 96)."""
    from lightx2v_amd import lib

    g = torch.Generator().manual_seed(11)
    cached_runs = 0
    for (T, H, W, Cin, Cout, kt, tail) in [(2, 8, 32, 128, 96, 1, 0), (2, 9, 11, 64, 192, 3, 0), (1, 17, 40, 288, 96, 3, 32), (3, 33, 70, 96, 384, 3, 32), (2, 16, 64, 192, 288, 1, 0),
                                           (9, 90, 160, 64, 96, 3, 0), (1, 17, 33, 64, 3, 3, 0), (2, 36, 70, 288, 3, 3, 32), (2, 9, 40, 128, 16, 1, 0),
                                           # the 128-cout form (8 x 8 accumulator tiles; the HunyuanVideo VAE's widths): one, two and four cout tiles
                                           (2, 17, 40, 128, 128, 3, 0), (1, 33, 70, 256, 256, 1, 0), (2, 16, 64, 512, 256, 3, 0), (1, 9, 33, 64, 512, 1, 0)]:
        cp = (Cin + 63) // 64 * 64
        assert cp - Cin == tail
        x = torch.randn(kt - 1 + T, H, W, Cin, generator=g).half()  # the leading kt - 1 frames are the cache
        w = (torch.randn(Cout, Cin, kt, 3, 3, generator=g) / (kt * 9 * Cin) ** 0.5).half()
        b = torch.randn(Cout, generator=g)
        res = torch.randn(T, H, W, Cout, generator=g)
        xin = F.pad(x.float().permute(3, 0, 1, 2), (1, 1, 1, 1, 0, 0))
        ref = (F.conv3d(xin.unsqueeze(0).cuda(), w.float().cuda(), b.cuda())[0].permute(1, 2, 3, 0) + res.cuda()).clamp(-1, 1)
        buf = torch.zeros(kt - 1 + T, H + 2, W + 2, cp, dtype=torch.float16, device="cuda")
        buf[:, 1 : 1 + H, 1 : 1 + W, :Cin] = x.cuda()
        w16 = torch.zeros(Cout, kt, 3, 3, cp, dtype=torch.float16, device="cuda")
        w16[..., :Cin] = w.permute(0, 2, 3, 4, 1).cuda()
        strides = ((H + 2) * (W + 2) * cp, (W + 2) * cp, cp)
        outs = {}
        for flags in (0, lib.VCONV_HALO64, lib.VCONV_PER_TAP) + ((lib.VCONV_ZERO_TAIL32,) if tail else ()):
            out = torch.full((T, H, W, Cout), float("nan"), device="cuda")
            lib.vae_conv16(buf, strides, w16, out, T, H, W, bias=b.cuda(), resid=res.cuda(), flags=flags | lib.VCONV_CLAMP)
            _check(out, ref, f"16-bit conv kt={kt} {T}x{H}x{W} Cin={Cin} Cout={Cout} flags={flags}", atol=3e-4, rel=1e-5)
            outs[flags] = out
        _check(outs[0], outs[lib.VCONV_HALO64], "128 x 96 kernel vs 64-pixel halo kernel", atol=1e-5, rel=1e-6)
        assert lib.vae_conv16_cached_ok(W, w16) == (kt > 1 and W >= 16)  # W < 16 takes the per-tap kernel, which has no separate-cache form
        if lib.vae_conv16_cached_ok(W, w16):  # the feature cache in a tensor of its own (x2v_vae_conv_f16_cached), the buffer's leading frames poisoned: bit-identical
            cached_runs += 1
            cache = buf[: kt - 1].clone()
            poisoned = buf.clone()
            poisoned[: kt - 1] = float("nan")
            out = torch.full((T, H, W, Cout), float("nan"), device="cuda")
            lib.vae_conv16(poisoned, strides, w16, out, T, H, W, bias=b.cuda(), resid=res.cuda(), flags=lib.VCONV_CLAMP, cache=cache)
            assert torch.equal(out, outs[0]), "separate cache tensor"
        if tail:
            assert torch.equal(outs[0], outs[lib.VCONV_ZERO_TAIL32]), "skipping the zero slab must not change a bit"
    assert cached_runs >= 4
    # frame batching: a launch over T frames = T one-frame launches, bit for bit
    T, H, W, Cin, Cout, kt = 4, 20, 48, 128, 96, 3
    x = torch.randn(kt - 1 + T, H + 2, W + 2, Cin, generator=g).half().cuda()
    w16 = (torch.randn(Cout, kt, 3, 3, Cin, generator=g) / (kt * 9 * Cin) ** 0.5).half().cuda()
    strides = ((H + 2) * (W + 2) * Cin, (W + 2) * Cin, Cin)
    whole = torch.empty(T, H, W, Cout, device="cuda")
    lib.vae_conv16(x, strides, w16, whole, T, H, W)
    for t in range(T):
        one = torch.empty(1, H, W, Cout, device="cuda")
        lib.vae_conv16(x[t:], strides, w16, one, 1, H, W)
        assert torch.equal(one[0], whole[t]), t
