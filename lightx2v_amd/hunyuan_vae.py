"""HunyuanVideo VAE decode (AutoencoderKLCausal3D, tiled) on the HIP kernels of csrc/vae.hip — host-side mirror.

reference (paths relative to /root/reference/lightx2v/models/video_encoders/hf/autoencoder_kl_causal_3d/):
  model.py:33-44 (VideoEncoderKLCausal3DModel.decode) · autoencoder_kl_causal_3d.py:286-312 (_decode), :347-364 (blend_*),
  :405-451 (spatial_tiled_decode), :487-518 (temporal_tiled_decode) · vae.py:133-283 (DecoderCausal3D) ·
  unet_causal_3d_blocks.py:65-91 (CausalConv3d), :94-197 (UpsampleCausal3D), :261-420 (ResnetBlockCausal3D),
  :526-640 (UNetMidBlockCausal3D), :693-758 (UpDecoderBlockCausal3D)

Same class / method names and diffusers state-dict tensor names.  Computation is fp32 on the fp32-input MFMA (the reference
runs this VAE in fp16; fp32 is the more accurate side of that), channels-last, on the kernels built for the Wan VAE:
  * every 3x3x3 CausalConv3d = x2v_vae_conv_f32 over a buffer [2 + T][H+2][W+2][C] whose borders are filled by
    x2v_vae_replicate_border_f32 (F.pad mode="replicate": nearest pixel spatially, first frame temporally);
  * GroupNorm = fp64-accumulated statistics reduced to a per-channel affine (x2v_groupnorm_affine_f32) that the producer
    kernel x2v_vae_prep_ex_f32 applies together with SiLU and the nearest upsampling (time factor 2 leaves the first frame
    single) while writing the next convolution's buffer — one pass per layer input;
  * the mid block's frame-causal attention over (f h w) tokens = fused q|k|v GEMM, QK^T GEMM, prefix-masked row softmax,
    PV GEMM, output projection with the residual in its epilogue (1x1 taps of the same convolution kernel);
  * tile blending = x2v_blend_axis_f32; the final x/2+0.5, clamp(0,1) = one pass of the producer kernel.
Torch is used for allocation, tile slicing / cropping / concatenation and layout changes — memory plumbing only.
"""
import math

import torch

from . import lib, synth


def _cl(w):
    """[Cout, Cin, kt, kh, kw] → [Cout, kt, kh, kw, Cin] contiguous fp32."""
    return w.float().permute(0, 2, 3, 4, 1).contiguous()


class DecoderCausal3D:
    """reference: autoencoder_kl_causal_3d/vae.py:133-283."""

    def __init__(self, sd, cfg, device, conv16=True):
        self.cfg, self.device = cfg, device
        # 3x3x3 convolutions with Cin % 64 == 0 (all but conv_in) take fp16 operands when conv16 — the reference's precision for this VAE
        # (hunyuan_runner.py:40: dtype fp16); accumulation, bias, residual stream, GroupNorm statistics and the attention stay fp32
        self.conv16 = conv16
        self.groups = cfg["norm_num_groups"]
        self.plan = synth.hunyuan_vae_up_plan(cfg)
        self.w = {}
        for k, v in sd.items():
            if not k.startswith("decoder."):
                continue
            v = v.to(device=device, dtype=torch.float32)
            self.w[k] = _cl(v) if v.dim() == 5 else v.contiguous()
        a = "decoder.mid_block.attentions.0."
        self.w[a + "qkv.weight"] = torch.cat([self.w[a + f"to_{n}.weight"] for n in "qkv"], 0).contiguous()  # load-time fusion of the three projections
        self.w[a + "qkv.bias"] = torch.cat([self.w[a + f"to_{n}.bias"] for n in "qkv"], 0).contiguous()
        self._bufs = {}
        self.w16 = {k: v.to(torch.float16).contiguous() for k, v in self.w.items() if conv16 and v.dim() == 5 and v.shape[1:4] == (3, 3, 3) and v.shape[4] % 64 == 0}

    # ---- building blocks ----------------------------------------------------------------------------------------------
    def _conv3(self, name, x, affine=None, silu=False, up_t=False, up_hw=False, resid=None):
        """[GroupNorm affine → SiLU → upsample →] replicate-padded causal 3x3x3 conv `name` on plain x [T,H,W,C]."""
        t, h, w, c = x.shape
        to = 2 * t - 1 if up_t else t
        ho, wo = (2 * h, 2 * w) if up_hw else (h, w)
        w16 = self.w16.get(name + ".weight")
        key = (to, ho, wo, c, w16 is not None)
        buf = self._bufs.get(key)
        if buf is None:
            buf = self._bufs[key] = torch.empty((2 + to, ho + 2, wo + 2, c), dtype=torch.float16 if w16 is not None else torch.float32, device=x.device)
        strides = ((ho + 2) * (wo + 2) * c, (wo + 2) * c, c)
        mul, add = affine if affine is not None else (None, None)
        lib.vae_prep_ex(x, buf[2:, 1:, 1:, :], strides[:2], mul=mul, add=add, silu=silu, up_hw=up_hw, up_t=up_t)
        lib.vae_replicate_border_(buf, 2, 1)
        wt = self.w[name + ".weight"]
        out = torch.empty((to, ho, wo, wt.shape[0]), dtype=torch.float32, device=x.device)
        if w16 is not None:
            lib.vae_conv16(buf, strides, w16, out, to, ho, wo, bias=self.w[name + ".bias"], resid=resid)
        else:
            lib.vae_conv(buf, strides, wt, out, to, ho, wo, bias=self.w[name + ".bias"], resid=resid)
        return out

    def _conv1(self, name, x, resid=None):
        t, h, w, c = x.shape
        wt = self.w[name + ".weight"]
        out = torch.empty((t, h, w, wt.shape[0]), dtype=torch.float32, device=x.device)
        lib.vae_conv(x, (h * w * c, w * c, c), wt, out, t, h, w, bias=self.w[name + ".bias"], resid=resid)
        return out

    def _gn(self, name, x):
        return lib.groupnorm_affine(x, self.groups, self.w[name + ".weight"], self.w[name + ".bias"], 1e-6)

    def resnet(self, p, x):
        """reference: ResnetBlockCausal3D.forward (unet_causal_3d_blocks.py:377-420), temb None, output_scale_factor 1."""
        h = self._conv3(p + "conv1.conv", x, affine=self._gn(p + "norm1", x), silu=True)
        short = self._conv1(p + "conv_shortcut.conv", x) if (p + "conv_shortcut.conv.weight") in self.w else x
        return self._conv3(p + "conv2.conv", h, affine=self._gn(p + "norm2", h), silu=True, resid=short)

    def mid_attention(self, p, x):
        """reference: UNetMidBlockCausal3D.forward (:629-634) + diffusers' Attention (deprecated-attn-block form)."""
        t, h, w, c = x.shape
        n, hw = t * h * w, h * w
        npad = (n + 15) // 16 * 16  # the PV GEMM reduces over keys in 16-float slabs: pad the token axis with zero rows
        mul, add = self._gn(p + "group_norm", x)
        y = torch.zeros((1, 1, npad, c), dtype=torch.float32, device=x.device)
        lib.vae_prep_ex(x.reshape(1, 1, n, c), y, (npad * c, npad * c), mul=mul, add=add)
        qkv = torch.empty((1, 1, npad, 3 * c), dtype=torch.float32, device=x.device)
        lib.vae_conv(y, (npad * c, npad * c, c), self.w[p + "qkv.weight"], qkv, 1, 1, npad, bias=self.w[p + "qkv.bias"])
        q = qkv.view(npad, 3 * c)
        if npad > n:
            q[n:].zero_()  # padded tokens: zero k and v rows (bias would otherwise leak into them)
        k, vt = q[:, c : 2 * c], q[:, 2 * c :].t().contiguous()
        scores = torch.empty((npad, npad), dtype=torch.float32, device=x.device)
        lib.vae_conv(q, (npad * 3 * c, npad * 3 * c, 3 * c), k, scores, 1, 1, npad, w_row_stride=3 * c, cin=c)
        lib.softmax_rows_causal_(scores, 1.0 / math.sqrt(c), hw, n_keys=n)
        o = torch.empty((1, 1, npad, c), dtype=torch.float32, device=x.device)
        lib.vae_conv(scores, (npad * npad, npad * npad, npad), vt, o, 1, 1, npad)
        out = torch.empty_like(x)
        lib.vae_conv(o, (npad * c, npad * c, c), self.w[p + "to_out.0.weight"], out.view(1, 1, n, c), 1, 1, n, bias=self.w[p + "to_out.0.bias"], resid=x.view(1, 1, n, c))
        return out

    def forward(self, z):
        """z [T, h, w, 16] (after post_quant_conv) → [4(T-1)+1, 8h, 8w, 3]."""
        x = self._conv3("decoder.conv_in.conv", z)
        x = self.resnet("decoder.mid_block.resnets.0.", x)
        x = self.mid_attention("decoder.mid_block.attentions.0.", x)
        x = self.resnet("decoder.mid_block.resnets.1.", x)
        for i, (_, _, ft, fhw, has_up) in enumerate(self.plan):
            for j in range(self.cfg["layers_per_block"] + 1):
                x = self.resnet(f"decoder.up_blocks.{i}.resnets.{j}.", x)
            if has_up:  # UpsampleCausal3D (:168-197): nearest upsample, then conv
                x = self._conv3(f"decoder.up_blocks.{i}.upsamplers.0.conv.conv", x, up_t=ft == 2, up_hw=fhw == 2)
        return self._conv3("decoder.conv_out.conv", x, affine=self._gn("decoder.conv_norm_out", x), silu=True)


class AutoencoderKLCausal3D:
    """reference: autoencoder_kl_causal_3d.py:58-518 (decode side, tiling enabled as model.py:38 always does)."""

    def __init__(self, sd, cfg=None, device="cuda", conv16=True, group=None):
        """group: a torch.distributed process group (e.g. the sequence-parallel group of the denoise loop) — the independent tiles of
        the tiled decode are then shared out over its ranks (not in the reference, whose `parallel_vae` covers the Wan VAE only)."""
        self.cfg = cfg or synth.HUNYUAN_VAE_CFG
        self.device = device
        self.group = group
        self.decoder = DecoderCausal3D(sd, self.cfg, device, conv16=conv16)
        zc = self.cfg["latent_channels"]
        self.pq_w = sd["post_quant_conv.weight"].to(device=device, dtype=torch.float32).reshape(zc, zc).contiguous()
        self.pq_b = sd["post_quant_conv.bias"].to(device=device, dtype=torch.float32).contiguous()
        self.tile_sample_min_size = self.cfg["sample_size"]
        self.tile_latent_min_size = int(self.cfg["sample_size"] / (2 ** (len(self.cfg["block_out_channels"]) - 1)))
        self.tile_sample_min_tsize = self.cfg["sample_tsize"]
        self.tile_latent_min_tsize = self.cfg["sample_tsize"] // self.cfg["time_compression_ratio"]
        self.tile_overlap_factor = self.cfg["tile_overlap_factor"]

    def post_quant_conv(self, z):
        t, h, w, c = z.shape
        out = torch.empty_like(z)
        lib.vae_conv(z, (h * w * c, w * c, c), self.pq_w, out, t, h, w, bias=self.pq_b)
        return out

    # blend_v / blend_h / blend_t (:347-364) on channels-last [T, H, W, C] tiles, in place on b
    def blend_v(self, a, b, extent):
        return lib.blend_axis_(a, b, 1, extent)

    def blend_h(self, a, b, extent):
        return lib.blend_axis_(a, b, 2, extent)

    def blend_t(self, a, b, extent):
        return lib.blend_axis_(a, b, 0, extent)

    # The reference decodes tile after tile inside the tiling loops.  Here the loops run twice over the same tile order: once to list
    # the latent tiles, once to blend the decoded ones — in between, the tiles are decoded either serially or, with a process group,
    # one share per rank (tiles are independent; every rank then holds all of them and blends redundantly: bit-identical output).
    def _spatial_tiles(self, z):
        lat = self.tile_latent_min_size
        overlap = int(lat * (1 - self.tile_overlap_factor))
        return [[z[:, i : i + lat, j : j + lat, :].contiguous() for j in range(0, z.shape[2], overlap)] for i in range(0, z.shape[1], overlap)]

    def _temporal_tiles(self, z):
        lat_t = self.tile_latent_min_tsize
        overlap = int(lat_t * (1 - self.tile_overlap_factor))
        return [z[i : i + lat_t + 1] for i in range(0, z.shape[0], overlap)]

    def _needs_spatial(self, z):
        return z.shape[1] > self.tile_latent_min_size or z.shape[2] > self.tile_latent_min_size

    def _jobs(self, z):
        """Latent tiles in the order the blending loops consume their decodes."""
        if z.shape[0] > self.tile_latent_min_tsize:
            out = []
            for tt in self._temporal_tiles(z):
                out += [t for row in self._spatial_tiles(tt) for t in row] if self._needs_spatial(tt) else [tt.contiguous()]
            return out
        if self._needs_spatial(z):
            return [t for row in self._spatial_tiles(z) for t in row]
        return [z]

    def _decode_jobs(self, jobs):
        one = lambda tile: self.decoder.forward(self.post_quant_conv(tile))  # noqa: E731
        if self.group is None:
            return [one(t) for t in jobs]
        import torch.distributed as dist

        n, r = dist.get_world_size(self.group), dist.get_rank(self.group)
        tc, sc = self.cfg["time_compression_ratio"], self.cfg["spatial_compression_ratio"]
        outs = [one(t) if k % n == r else torch.empty((tc * (t.shape[0] - 1) + 1, sc * t.shape[1], sc * t.shape[2], 3), dtype=torch.float32, device=t.device)
                for k, t in enumerate(jobs)]  # own share first (all ranks compute concurrently) ...
        for k, o in enumerate(outs):  # ... then every tile travels from its owner to everyone
            dist.broadcast(o, src=dist.get_global_rank(self.group, k % n) if self.group is not dist.group.WORLD else k % n, group=self.group)
        return outs

    def spatial_tiled_decode(self, z, decoded=None):
        """reference :405-451.  z [T, H, W, 16]; `decoded`: iterator over the decoded tiles in `_spatial_tiles` order."""
        smp = self.tile_sample_min_size
        extent = int(smp * self.tile_overlap_factor)
        limit = smp - extent
        tiles = self._spatial_tiles(z)
        if decoded is None:
            decoded = iter(self._decode_jobs([t for row in tiles for t in row]))
        rows = [[next(decoded) for _ in row] for row in tiles]
        out_rows = []
        for i, row in enumerate(rows):
            out = []
            for j, tile in enumerate(row):
                if i > 0:
                    self.blend_v(rows[i - 1][j], tile, extent)
                if j > 0:
                    self.blend_h(row[j - 1], tile, extent)
                out.append(tile[:, :limit, :limit, :])
            out_rows.append(torch.cat(out, dim=2))
        return torch.cat(out_rows, dim=1)

    def temporal_tiled_decode(self, z, decoded=None):
        """reference :487-518."""
        smp_t = self.tile_sample_min_tsize
        extent = int(smp_t * self.tile_overlap_factor)
        t_limit = smp_t - extent
        if decoded is None:
            decoded = iter(self._decode_jobs(self._jobs(z)))
        row = []
        for i, tile in enumerate(self._temporal_tiles(z)):
            dec = self.spatial_tiled_decode(tile, decoded) if self._needs_spatial(tile) else next(decoded)
            if i > 0:
                dec = dec[1:]
            row.append(dec.contiguous())
        out = []
        for i, tile in enumerate(row):
            if i > 0:
                self.blend_t(row[i - 1], tile, extent)
                out.append(tile[:t_limit])
            else:
                out.append(tile[: t_limit + 1])
        return torch.cat(out, dim=0)

    def _decode(self, z):
        """reference :286-301 (both tilings enabled)."""
        if z.shape[0] > self.tile_latent_min_tsize:
            return self.temporal_tiled_decode(z)
        if self._needs_spatial(z):
            return self.spatial_tiled_decode(z)
        return self.decoder.forward(self.post_quant_conv(z))

    def decode(self, z):
        return self._decode(z)


class VideoEncoderKLCausal3DModel:
    """reference: autoencoder_kl_causal_3d/model.py:6-44 (decode side)."""

    def __init__(self, sd, cfg=None, device="cuda", conv16=True, group=None):
        """conv16 (default): fp16 operands for the 3x3x3 convolutions — the reference's precision for this VAE; False: everything fp32.
        group: process group whose ranks share the tiles of the tiled decode (see AutoencoderKLCausal3D)."""
        self.model = AutoencoderKLCausal3D(sd, cfg, device, conv16=conv16, group=group)
        self.device = device

    def decode(self, latents, generator=None, config=None):
        """latents [1, 16, T, h, w] → image [1, 3, 4(T-1)+1, 8h, 8w] fp32 in [0, 1] (on the device; the reference moves it to the host)."""
        z = latents[0].to(self.device, torch.float32).permute(1, 2, 3, 0).contiguous()
        zs = torch.empty_like(z)
        c = z.shape[-1]
        inv = torch.full((c,), 1.0 / self.model.cfg["scaling_factor"], dtype=torch.float32, device=z.device)
        lib.vae_prep_ex(z, zs, (z.shape[1] * z.shape[2] * c, z.shape[2] * c), mul=inv)  # latents / scaling_factor
        img = self.model.decode(zs).contiguous()  # [T, H, W, 3]
        flat = img.view(1, 1, img.numel() // 4, 4)
        half = torch.full((4,), 0.5, dtype=torch.float32, device=z.device)
        out = torch.empty_like(flat)
        lib.vae_prep_ex(flat, out, (img.numel(), img.numel()), mul=half, add=half, clamp01=True)  # (x / 2 + 0.5).clamp(0, 1)
        return out.view(img.shape).permute(3, 0, 1, 2).unsqueeze(0).contiguous()
