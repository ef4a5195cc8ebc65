"""TEST INFRASTRUCTURE ONLY — CPU restatement (plain PyTorch, fp32) of the reference's HunyuanVideo VAE *decode* path.

reference: /root/reference/lightx2v/models/video_encoders/hf/autoencoder_kl_causal_3d/
  model.py:33-44                       VideoEncoderKLCausal3DModel.decode: z / scaling_factor, enable_tiling, decode, x/2+0.5, clamp(0,1)
  autoencoder_kl_causal_3d.py:296-345  _decode / decode;  :347-364 blend_v/h/t;  :405-451 spatial_tiled_decode;  :487-518 temporal_tiled_decode
  vae.py:133-283                       DecoderCausal3D (conv_in, mid block, 4 up blocks, GroupNorm + SiLU + conv_out)
  unet_causal_3d_blocks.py:48-63       prepare_causal_attention_mask;  :65-91 CausalConv3d (replicate padding);  :94-197 UpsampleCausal3D;
                                       :261-420 ResnetBlockCausal3D;  :526-640 UNetMidBlockCausal3D;  :693-758 UpDecoderBlockCausal3D
  (the mid block's `Attention` is diffusers' class with `_from_deprecated_attn_block`: GroupNorm over channels, single head of
   dim C, to_q/to_k/to_v/to_out.0 Linear with bias, fp32 softmax, residual connection, rescale_output_factor 1)

PARITY UNPINNED: the reference module imports `diffusers` (ConfigMixin / ModelMixin / Attention), which is absent offline, so no
fixture can be generated from it here; this file restates the published algorithm from the reference's own sources above and
the HIP path is checked against it.  The reference runs this VAE in fp16 (hunyuan_runner.py:40); both this restatement and the
HIP path compute in fp32 (more accurate; differences from an fp16 run are of fp16-rounding size).

State-dict names are diffusers' (`decoder.conv_in.conv.weight`, `decoder.mid_block.resnets.0.norm1.weight`, …, `post_quant_conv.weight`).
Tensors are [B=1, C, T, H, W].
"""
import math

import torch
import torch.nn.functional as F

CFG = dict(block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=16, norm_num_groups=32, sample_size=256, sample_tsize=64,
           scaling_factor=0.476986, time_compression_ratio=4, spatial_compression_ratio=8, tile_overlap_factor=0.25)


def causal_conv3d(x, w, b):
    """unet_causal_3d_blocks.py:65-91: replicate padding (k//2 each side in H, W; k-1 frames in front in T)."""
    k = w.shape[-1]
    x = F.pad(x, (k // 2, k // 2, k // 2, k // 2, k - 1, 0), mode="replicate")
    return F.conv3d(x, w, b)


def upsample(x, factor_t, factor_hw):
    """UpsampleCausal3D.forward (:168-187): nearest; the first frame is only upsampled spatially."""
    first, other = x[:, :, :1], x[:, :, 1:]
    if other.shape[2] > 0:
        other = F.interpolate(other, scale_factor=(factor_t, factor_hw, factor_hw), mode="nearest")
    b, c, _, h, w = first.shape
    first = F.interpolate(first.view(b, c, h, w), scale_factor=(factor_hw, factor_hw), mode="nearest").unsqueeze(2)
    return torch.cat((first, other), dim=2) if other.shape[2] > 0 else first


def resnet(sd, p, x, groups):
    """ResnetBlockCausal3D.forward (:377-420), temb = None, output_scale_factor 1."""
    h = F.silu(F.group_norm(x, groups, sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6))
    h = causal_conv3d(h, sd[p + "conv1.conv.weight"], sd[p + "conv1.conv.bias"])
    h = F.silu(F.group_norm(h, groups, sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6))
    h = causal_conv3d(h, sd[p + "conv2.conv.weight"], sd[p + "conv2.conv.bias"])
    if (p + "conv_shortcut.conv.weight") in sd:
        x = causal_conv3d(x, sd[p + "conv_shortcut.conv.weight"], sd[p + "conv_shortcut.conv.bias"])
    return x + h


def mid_attention(sd, p, x, groups):
    """UNetMidBlockCausal3D.forward (:629-634) + the deprecated-attn-block Attention: tokens (f h w), frame-causal mask."""
    b, c, t, h, w = x.shape
    n = t * h * w
    tok = x.permute(0, 2, 3, 4, 1).reshape(b, n, c)
    res = tok
    y = F.group_norm(tok.transpose(1, 2), groups, sd[p + "group_norm.weight"], sd[p + "group_norm.bias"], 1e-6).transpose(1, 2)
    q = F.linear(y, sd[p + "to_q.weight"], sd[p + "to_q.bias"])
    k = F.linear(y, sd[p + "to_k.weight"], sd[p + "to_k.bias"])
    v = F.linear(y, sd[p + "to_v.weight"], sd[p + "to_v.bias"])
    frame = torch.arange(n) // (h * w)
    mask = torch.where(frame[None, :] <= frame[:, None], 0.0, float("-inf"))
    s = torch.baddbmm(mask.unsqueeze(0).expand(b, -1, -1), q, k.transpose(1, 2), beta=1, alpha=1.0 / math.sqrt(c))
    o = torch.bmm(s.softmax(dim=-1), v)
    o = F.linear(o, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"]) + res
    return o.reshape(b, t, h, w, c).permute(0, 4, 1, 2, 3)


def up_plan(cfg=CFG):
    """DecoderCausal3D.__init__ (vae.py:176-211): per up block (in_ch, out_ch, time_factor, hw_factor or None)."""
    boc = list(reversed(cfg["block_out_channels"]))
    n_sp = int(math.log2(cfg["spatial_compression_ratio"]))
    n_t = int(math.log2(cfg["time_compression_ratio"]))
    plan, prev = [], boc[0]
    for i, out in enumerate(boc):
        final = i == len(boc) - 1
        sp = i < n_sp
        tm = i >= len(boc) - 1 - n_t and not final
        plan.append((prev, out, 2 if tm else 1, 2 if sp else 1, sp or tm))
        prev = out
    return plan


def decoder(sd, z, cfg=CFG):
    """DecoderCausal3D.forward (vae.py:222-283)."""
    g = cfg["norm_num_groups"]
    x = causal_conv3d(z, sd["decoder.conv_in.conv.weight"], sd["decoder.conv_in.conv.bias"])
    x = resnet(sd, "decoder.mid_block.resnets.0.", x, g)
    x = mid_attention(sd, "decoder.mid_block.attentions.0.", x, g)
    x = resnet(sd, "decoder.mid_block.resnets.1.", x, g)
    for i, (_, _, ft, fhw, has_up) in enumerate(up_plan(cfg)):
        for j in range(cfg["layers_per_block"] + 1):
            x = resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}.", x, g)
        if has_up:
            x = upsample(x, ft, fhw)
            x = causal_conv3d(x, sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.conv.weight"], sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.conv.bias"])
    x = F.silu(F.group_norm(x, g, sd["decoder.conv_norm_out.weight"], sd["decoder.conv_norm_out.bias"], 1e-6))
    return causal_conv3d(x, sd["decoder.conv_out.conv.weight"], sd["decoder.conv_out.conv.bias"])


def _blend(a, b, extent, dim):
    """blend_v / blend_h / blend_t (autoencoder_kl_causal_3d.py:347-364)."""
    extent = min(a.shape[dim], b.shape[dim], extent)
    for i in range(extent):
        ia = [slice(None)] * 5
        ib = [slice(None)] * 5
        ia[dim], ib[dim] = -extent + i, i
        b[tuple(ib)] = a[tuple(ia)] * (1 - i / extent) + b[tuple(ib)] * (i / extent)
    return b


def _tile_sizes(cfg):
    lat = int(cfg["sample_size"] / (2 ** (len(cfg["block_out_channels"]) - 1)))
    return cfg["sample_size"], lat, cfg["sample_tsize"], cfg["sample_tsize"] // cfg["time_compression_ratio"]


def spatial_tiled_decode(sd, z, cfg=CFG):
    """autoencoder_kl_causal_3d.py:405-451."""
    smp, lat, _, _ = _tile_sizes(cfg)
    overlap = int(lat * (1 - cfg["tile_overlap_factor"]))
    extent = int(smp * cfg["tile_overlap_factor"])
    limit = smp - extent
    rows = []
    for i in range(0, z.shape[-2], overlap):
        row = []
        for j in range(0, z.shape[-1], overlap):
            tile = z[:, :, :, i : i + lat, j : j + lat]
            tile = F.conv3d(tile, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
            row.append(decoder(sd, tile, cfg))
        rows.append(row)
    out_rows = []
    for i, row in enumerate(rows):
        out = []
        for j, tile in enumerate(row):
            if i > 0:
                tile = _blend(rows[i - 1][j], tile, extent, 3)
            if j > 0:
                tile = _blend(row[j - 1], tile, extent, 4)
            out.append(tile[:, :, :, :limit, :limit])
        out_rows.append(torch.cat(out, dim=-1))
    return torch.cat(out_rows, dim=-2)


def temporal_tiled_decode(sd, z, cfg=CFG):
    """autoencoder_kl_causal_3d.py:487-518 (spatial tiling is always on: model.py:38 enable_tiling())."""
    smp, lat, smp_t, lat_t = _tile_sizes(cfg)
    overlap = int(lat_t * (1 - cfg["tile_overlap_factor"]))
    extent = int(smp_t * cfg["tile_overlap_factor"])
    t_limit = smp_t - extent
    row = []
    for i in range(0, z.shape[2], overlap):
        tile = z[:, :, i : i + lat_t + 1]
        if tile.shape[-1] > lat or tile.shape[-2] > lat:
            dec = spatial_tiled_decode(sd, tile, cfg)
        else:
            dec = decoder(sd, F.conv3d(tile, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"]), cfg)
        if i > 0:
            dec = dec[:, :, 1:]
        row.append(dec)
    out = []
    for i, tile in enumerate(row):
        if i > 0:
            tile = _blend(row[i - 1], tile, extent, 2)
            out.append(tile[:, :, :t_limit])
        else:
            out.append(tile[:, :, : t_limit + 1])
    return torch.cat(out, dim=2)


def vae_decode(sd, latents, cfg=CFG):
    """VideoEncoderKLCausal3DModel.decode (model.py:33-44) → AutoencoderKLCausal3D._decode (:296-312) with tiling enabled."""
    z = latents / cfg["scaling_factor"]
    smp, lat, smp_t, lat_t = _tile_sizes(cfg)
    if z.shape[2] > lat_t:
        img = temporal_tiled_decode(sd, z, cfg)
    elif z.shape[-1] > lat or z.shape[-2] > lat:
        img = spatial_tiled_decode(sd, z, cfg)
    else:
        img = decoder(sd, F.conv3d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"]), cfg)
    return (img / 2 + 0.5).clamp(0, 1)
