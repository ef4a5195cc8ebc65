"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's Wan VAE *decode* path (fp32), used as the checker
for the HIP decoder (`lightx2v_amd/vae.py`).  Never imported by the product path.

reference: /root/reference/lightx2v/models/video_encoders/hf/wan/vae.py
  CausalConv3d :19-44 · RMS_norm :47-59 · Upsample :62-67 · Resample :70-159 · ResidualBlock :185-223 ·
  AttentionBlock :226-262 · Decoder3d :377-489 · WanVAE_.decode :713-738 · WanVAE.decode (clamp) :931-957

Pinned: `tests/golden/wan_vae_tiny.safetensors` is generated from the unmodified reference (oracle/gen_golden.py::gen_vae)
and `tests/test_oracle_golden.py` requires this restatement to reproduce it (same torch ops in the same order → bit-exact
on the same host).

State dict names are the reference module tree's (`decoder.*`, `conv2.*`), so the synthetic weights of
`lightx2v_amd.synth.synth_wan_vae_weights` load into the reference model and into both restatements unchanged.
Tensors are [C, T, H, W] (batch 1 dropped).
"""
import torch
import torch.nn.functional as F

CACHE_T = 2  # vae.py:16


def causal_conv3d(x, w, b, cache=None):
    """vae.py:19-44: temporal padding 2*pt in front (filled by `cache` frames first), 'same' zero padding spatially."""
    kt, kh, kw = w.shape[2:]
    pad_t = kt - 1
    if cache is not None and pad_t > 0:
        x = torch.cat([cache, x], dim=1)
        pad_t -= cache.shape[1]
    x = F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2, pad_t, 0))
    return F.conv3d(x.unsqueeze(0), w, b).squeeze(0)


def rms_norm(x, gamma):
    """vae.py:47-59: F.normalize over channels * sqrt(C) * gamma (bias = 0)."""
    c = x.shape[0]
    return F.normalize(x, dim=0) * (c**0.5) * gamma.reshape(c, 1, 1, 1)


class _Cache:
    """Per-conv feature cache with the reference's update rule (vae.py:199-214, 443-456, 468-483)."""

    def __init__(self):
        self.slots = {}

    def conv(self, key, x, w, b):
        old = self.slots.get(key)
        keep = x[:, -CACHE_T:].clone()
        if keep.shape[1] < 2 and old is not None:
            keep = torch.cat([old[:, -1:], keep], dim=1)
        y = causal_conv3d(x, w, b, old)
        self.slots[key] = keep
        return y


def residual_block(sd, p, x, cache):
    """vae.py:185-223."""
    h = causal_conv3d(x, sd[p + "shortcut.weight"], sd[p + "shortcut.bias"]) if (p + "shortcut.weight") in sd else x
    y = F.silu(rms_norm(x, sd[p + "residual.0.gamma"]))
    y = cache.conv(p + "c1", y, sd[p + "residual.2.weight"], sd[p + "residual.2.bias"])
    y = F.silu(rms_norm(y, sd[p + "residual.3.gamma"]))
    y = cache.conv(p + "c2", y, sd[p + "residual.6.weight"], sd[p + "residual.6.bias"])
    return y + h


def attention_block(sd, p, x):
    """vae.py:226-262: per frame, single head over h*w tokens, dim C."""
    c, t, h, w = x.shape
    xf = x.permute(1, 0, 2, 3)  # (t) c h w
    n = F.normalize(xf, dim=1) * (c**0.5) * sd[p + "norm.gamma"].reshape(1, c, 1, 1)
    qkv = F.conv2d(n, sd[p + "to_qkv.weight"], sd[p + "to_qkv.bias"])
    q, k, v = qkv.reshape(t, 1, c * 3, -1).permute(0, 1, 3, 2).contiguous().chunk(3, dim=-1)
    o = F.scaled_dot_product_attention(q, k, v)
    o = o.squeeze(1).permute(0, 2, 1).reshape(t, c, h, w)
    o = F.conv2d(o, sd[p + "proj.weight"], sd[p + "proj.bias"])
    return o.permute(1, 0, 2, 3) + x


def resample_up(sd, p, x, cache, mode):
    """vae.py:108-143 (upsample2d / upsample3d): optional causal time conv that doubles T, then nearest-exact 2x + conv2d."""
    c, t, h, w = x.shape
    if mode == "upsample3d":
        key = p + "time"
        state = cache.slots.get(key)
        if state is None:
            cache.slots[key] = "Rep"  # first chunk: no temporal upsampling, nothing cached (vae.py:113-115)
        else:
            keep = x[:, -CACHE_T:].clone()
            if keep.shape[1] < 2:
                if isinstance(state, str):
                    keep = torch.cat([torch.zeros_like(keep), keep], dim=1)
                else:
                    keep = torch.cat([state[:, -1:], keep], dim=1)
            y = causal_conv3d(x, sd[p + "time_conv.weight"], sd[p + "time_conv.bias"], None if isinstance(state, str) else state)
            cache.slots[key] = keep
            y = y.reshape(2, c, t, h, w)
            x = torch.stack((y[0], y[1]), dim=2).reshape(c, t * 2, h, w)
            t = t * 2
    xf = x.permute(1, 0, 2, 3)
    xf = F.interpolate(xf.float(), scale_factor=(2.0, 2.0), mode="nearest-exact")
    xf = F.conv2d(xf, sd[p + "resample.1.weight"], sd[p + "resample.1.bias"], padding=1)
    return xf.permute(1, 0, 2, 3)


def decoder_plan(dim=96, dim_mult=(1, 2, 4, 4), num_res_blocks=2, temperal_upsample=(True, True, False)):
    """Decoder3d.__init__ (vae.py:377-434): the `upsamples` Sequential as a list of (index, kind, in_dim, out_dim)."""
    dims = [dim * u for u in [dim_mult[-1]] + list(dim_mult[::-1])]
    plan, idx = [], 0
    for i, (in_dim, out_dim) in enumerate(zip(dims[:-1], dims[1:])):
        if i in (1, 2, 3):
            in_dim = in_dim // 2
        for _ in range(num_res_blocks + 1):
            plan.append((idx, "res", in_dim, out_dim))
            idx += 1
            in_dim = out_dim
        if i != len(dim_mult) - 1:
            plan.append((idx, "upsample3d" if temperal_upsample[i] else "upsample2d", out_dim, out_dim // 2))
            idx += 1
    return dims, plan


def decoder_chunk(sd, x, cache, plan):
    """Decoder3d.forward (vae.py:436-489) on one chunk of latent frames."""
    x = cache.conv("conv1", x, sd["decoder.conv1.weight"], sd["decoder.conv1.bias"])
    x = residual_block(sd, "decoder.middle.0.", x, cache)
    x = attention_block(sd, "decoder.middle.1.", x)
    x = residual_block(sd, "decoder.middle.2.", x, cache)
    for idx, kind, _, _ in plan:
        p = f"decoder.upsamples.{idx}."
        x = residual_block(sd, p, x, cache) if kind == "res" else resample_up(sd, p, x, cache, kind)
    x = F.silu(rms_norm(x, sd["decoder.head.0.gamma"]))
    return cache.conv("head", x, sd["decoder.head.2.weight"], sd["decoder.head.2.bias"])


def wan_vae_decode(sd, z, mean, inv_std, dim=96, clamp=True):
    """WanVAE_.decode (vae.py:713-738) + WanVAE.decode's clamp (vae.py:951-955).  z [16, T, h, w] fp32 →
    [3, 1 + 4 (T-1), 8h, 8w]: one latent frame at a time through the cached decoder."""
    zc = z.shape[0]
    z = z / inv_std.view(zc, 1, 1, 1) + mean.view(zc, 1, 1, 1)
    x = F.conv3d(z.unsqueeze(0), sd["conv2.weight"], sd["conv2.bias"]).squeeze(0)
    _, plan = decoder_plan(dim)
    cache = _Cache()
    outs = [decoder_chunk(sd, x[:, i : i + 1], cache, plan) for i in range(x.shape[1])]
    out = torch.cat(outs, dim=1)
    return out.float().clamp_(-1, 1) if clamp else out


def wan_vae_decode_dist(sd, z, mean, inv_std, world_size, split_dim, dim=96):
    """WanVAE.decode_dist (vae.py:883-929) evaluated for every rank in turn: slab + 1-latent-pixel halo (2 on the outer side of
    the edge ranks), independent decode, crop 8 px per halo pixel, concatenate.  z [16, T, h, w]; split_dim 2 (H) or 3 (W)."""
    total = z.shape[split_dim]
    chunk, pad = total // world_size, 1
    outs = []
    for r in range(world_size):
        if r == 0:
            lo, hi = 0, chunk + 2 * pad
        elif r == world_size - 1:
            lo, hi = total - (chunk + 2 * pad), total
        else:
            lo, hi = r * chunk - pad, (r + 1) * chunk + pad
        img = wan_vae_decode(sd, z.narrow(split_dim, lo, hi - lo).contiguous(), mean, inv_std, dim=dim)  # [3, T', H', W']
        if r == 0:
            img = img.narrow(split_dim, 0, chunk * 8)
        elif r == world_size - 1:
            img = img.narrow(split_dim, img.shape[split_dim] - chunk * 8, chunk * 8)
        else:
            img = img.narrow(split_dim, 8 * pad, img.shape[split_dim] - 16 * pad)
        outs.append(img)
    return torch.cat(outs, dim=split_dim)
