"""HunyuanVideo VAE decode on the HIP kernels (lightx2v_amd/hunyuan_vae.py) against the CPU oracle
(oracle/hunyuan_vae_oracle.py — parity UNPINNED: the reference module needs `diffusers`, absent offline; the oracle restates
its sources).  fp32 on both sides: |d| <= 2e-3 on outputs in [0, 1], relative L2 <= 1e-3."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _check(got, ref, what, atol=2e-3, rel=1e-3):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    d = (got - ref).abs().max().item()
    r = ((got - ref).norm() / ref.norm().clamp_min(1e-30)).item()
    assert d <= atol and r <= rel, f"{what}: max abs {d:.3e}, rel L2 {r:.3e}"


def test_hunyuan_vae_ops():
    from lightx2v_amd import lib

    g = torch.Generator().manual_seed(0)
    # GroupNorm as a per-channel affine + SiLU + temporal/spatial upsample, replicate borders
    T, H, W, C, G = 3, 5, 6, 64, 8
    x = torch.randn(T, H, W, C, generator=g) * 2 + 0.3
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    ref = F.silu(F.group_norm(x.permute(3, 0, 1, 2).unsqueeze(0), G, gamma, beta, 1e-6))[0]  # [C,T,H,W]
    mul, add = lib.groupnorm_affine(x.cuda(), G, gamma.cuda(), beta.cuda(), 1e-6)
    to, ho, wo = 2 * T - 1, 2 * H, 2 * W
    buf = torch.full((2 + to, ho + 2, wo + 2, C), 7.0, device="cuda")
    lib.vae_prep_ex(x.cuda(), buf[2:, 1:, 1:], ((ho + 2) * (wo + 2) * C, (wo + 2) * C), mul=mul, add=add, silu=True, up_hw=True, up_t=True)
    lib.vae_replicate_border_(buf, 2, 1)
    first = F.interpolate(ref[:, :1].permute(1, 0, 2, 3), scale_factor=(2.0, 2.0), mode="nearest").permute(1, 0, 2, 3)
    other = F.interpolate(ref[:, 1:].unsqueeze(0), scale_factor=(2.0, 2.0, 2.0), mode="nearest")[0]
    up = torch.cat((first, other), dim=1)
    padded = F.pad(up.unsqueeze(0), (1, 1, 1, 1, 2, 0), mode="replicate")[0].permute(1, 2, 3, 0)
    _check(buf, padded, "groupnorm+silu+upsample+replicate pad", atol=2e-5, rel=1e-5)
    # frame-causal softmax with padded key columns
    hw, nfr = 6, 3
    n, npad = hw * nfr, 32
    s = torch.randn(npad, npad, generator=g) * 3
    got = lib.softmax_rows_causal_(s.clone().cuda(), 0.3, hw, n_keys=n)
    fr = torch.arange(n) // hw
    masked = torch.where(fr[None, :] <= fr[:, None], s[:n, :n] * 0.3, torch.tensor(float("-inf")))
    _check(got[:n, :n], masked.softmax(-1), "causal softmax", atol=1e-6, rel=1e-5)
    assert got[:n, n:].abs().max() == 0
    # blends
    a, b = torch.randn(4, 7, 5, 3, generator=g), torch.randn(4, 6, 5, 3, generator=g)
    refb = b.clone()
    for y in range(3):
        refb[:, y] = a[:, -3 + y] * (1 - y / 3) + refb[:, y] * (y / 3)
    _check(lib.blend_axis_(a.cuda(), b.clone().cuda(), 1, 3), refb, "blend_v", atol=1e-6, rel=1e-6)


def test_hunyuan_vae_decode_tiled_vs_oracle():
    """Reduced widths (32/64/128/128 channels, 8 groups), tile sizes scaled down so that z [16,6,12,10] takes the temporal
    tiling path (2 tiles) with spatial tiling inside (2x2 tiles each) and every blend."""
    from lightx2v_amd import hunyuan_vae, synth
    from oracle import hunyuan_vae_oracle as V

    cfg = synth.HUNYUAN_VAE_TINY_CFG
    sd = synth.synth_hunyuan_vae_weights(cfg, seed=1)
    z = torch.randn(1, 16, 6, 12, 10, generator=torch.Generator().manual_seed(3)) * 0.5
    with torch.no_grad():
        ref = V.vae_decode(sd, z, cfg)
    out = hunyuan_vae.VideoEncoderKLCausal3DModel(sd, cfg).decode(z.cuda())
    assert out.shape == (1, 3, 21, 96, 80)
    _check(out, ref, "Hunyuan VAE tiled decode")


def test_hunyuan_vae_real_widths_single_tile_vs_oracle():
    """Released channel plan (128/256/512/512, 32 groups) on one small tile: z [16, 2, 4, 8] → [3, 5, 32, 64]."""
    from lightx2v_amd import hunyuan_vae, synth
    from oracle import hunyuan_vae_oracle as V

    cfg = synth.HUNYUAN_VAE_CFG
    sd = synth.synth_hunyuan_vae_weights(cfg, seed=2)
    z = torch.randn(1, 16, 2, 4, 8, generator=torch.Generator().manual_seed(4)) * 0.5
    with torch.no_grad():
        ref = V.vae_decode(sd, z, cfg)
    out = hunyuan_vae.VideoEncoderKLCausal3DModel(sd, cfg).decode(z.cuda())
    assert out.shape == (1, 3, 5, 32, 64)
    _check(out, ref, "Hunyuan VAE decode (released widths)")
