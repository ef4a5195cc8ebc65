"""Wan VAE decode on the HIP kernels of csrc/vae.hip — host-side mirror of the reference's decode path.

reference: lightx2v/models/video_encoders/hf/wan/vae.py — WanVAE.decode :931-957 → WanVAE_.decode :713-738 →
Decoder3d.forward :436-489 → ResidualBlock :185-223 / AttentionBlock :226-262 / Resample :70-159 / CausalConv3d :19-44.
Same class and method names, same state-dict tensor names (`decoder.*`, `conv2.*`), same cache-carrying decoder (a 2-frame
cache per causal conv; the reference feeds it one latent frame at a time, here `chunk_frames` at a time with bit-identical
output), fp32 like the reference (vae.py:794) — by default with the big convolutions' operands split into hi + lo fp16 halves
(fp32-grade results on the 16-bit matrix instruction, see WanVAE.__init__), or on the fp32 matrix instruction (`conv16=False`).

What is laid out differently for the MI355X (288 GB HBM, fp32-input MFMA):
  * activations are channels-last [T, H, W, C] so both implicit-GEMM operands are K-contiguous;
  * every 3x3(x3) convolution reads a persistent zero-bordered buffer [2 + T][H+2][W+2][C]; its two leading frames
    ARE the reference's feat_cache entry (update rule "last two frames of [cache | x]" = one 2-frame move), so there
    is no torch.cat / F.pad / clone per conv per chunk (vae.py:36-41,199-214);
  * RMS_norm + SiLU (and the nearest-exact 2x upsample, and the latent un-normalisation) are one pixel-wise kernel
    that writes straight into the next convolution's buffer; residual adds, the final clamp and the upsample3d
    channel→frame interleave (vae.py:136-138) are convolution epilogues;
  * the single-head attention block is GEMM (QK^T) → row softmax → GEMM (PV) on the same fp32 MFMA convolution
    kernel (1x1 taps), per frame.
Torch is used for allocation, 2-frame cache moves (copy_), the V transpose and the boundary layout changes
([C,T,H,W] ↔ channels-last) — memory plumbing, no arithmetic.
"""
import torch

from . import lib, synth


class _ConvInput:
    """Zero-bordered, cache-carrying input of one convolution: the kernel reads [lead + t][H + 2p][W + 2p][C] contiguous frames, the first `lead`
    of which are the reference's `feat_cache` entry (CACHE_T = 2, vae.py:16,199-214).

    Memory (round 3): only the `lead` cache frames belong to the convolution; the frame buffer itself is SHARED by every convolution of
    the same geometry (`pool`, owned by the decoder — convolutions run one after the other on one stream).  `acquire()` moves the cache
    into the head of the shared buffer before the producer writes the new frames behind it, `roll()` saves the last `lead` input frames
    back.  One more 2-frame move per convolution than with a private buffer each (+2 % decode time) for a third of the memory: at
    720p the 33 private buffers took 89 GB at 2 latent frames per pass."""

    def __init__(self, t_max, h, w, c, kt, pad, device, dtype=torch.float32, split=False, pool=None):
        self.lead, self.pad, self.h, self.w = kt - 1, pad, h, w
        # fp16 operand buffers (16-bit convolution) pad the channel axis to the kernel's 64-channel K step; the pad channels are never
        # written and stay zero (as do the matching weight columns).  split: three planes [hi | hi * 2^-12 | lo] of the c channels
        self.c = c if dtype == torch.float32 else ((3 * c if split else c) + 63) // 64 * 64
        self.hp, self.wp = h + 2 * pad, w + 2 * pad
        self.t_max = t_max
        key = (t_max + self.lead, self.hp, self.wp, self.c, c, dtype)  # real c in the key: the zero pad channels must be the same set
        pool = pool if pool is not None else {}
        if key not in pool:
            pool[key] = torch.zeros((self.lead + t_max, self.hp, self.wp, self.c), dtype=dtype, device=device)
        self.buf = pool[key]  # borders and pad channels stay zero: every user writes interiors (and whole cache frames, which carry zero borders)
        self.cache = torch.zeros((self.lead, self.hp, self.wp, self.c), dtype=dtype, device=device) if self.lead else None
        self.strides = (self.hp * self.wp * self.c, self.wp * self.c, self.c)  # frame, row, pixel (elements)

    def acquire(self, copy=True):
        """Put this convolution's cache frames in front of the shared frame buffer (call before the producer writes the new frames).
        copy=False: the kernel reads the cache frames from `self.cache` itself (lib.vae_conv16(cache=...)): the buffer's leading frames stay
        stale and `roll` must be told (head_valid=False)."""
        if self.lead and copy:
            self.buf[: self.lead].copy_(self.cache)
        return self

    def interior(self):
        """View whose first element is where pixel (t=0, h=0, w=0) of the new frames goes."""
        return self.buf[self.lead :, self.pad :, self.pad :, :]

    def roll(self, t, head_valid=True):
        """Cache update (vae.py:199-214): keep the last `lead` frames of [cache | x]."""
        if not self.lead:
            return
        if head_valid or t >= self.lead:  # t >= lead: the last `lead` frames all lie among the new ones
            self.cache.copy_(self.buf[t : t + self.lead])
        else:  # fewer new frames than the cache holds and the buffer's head is stale: shift the cache, append the new frames
            keep = self.cache[t:].clone()
            self.cache[: self.lead - t].copy_(keep)
            self.cache[self.lead - t :].copy_(self.buf[self.lead : self.lead + t])

    def reset(self):
        if self.lead:
            self.cache.zero_()


def _cl(weight):
    """[Cout, Cin, (kt,) kh, kw] → [Cout, kt, kh, kw, Cin] contiguous (load-time layout change)."""
    if weight.dim() == 4:
        weight = weight.unsqueeze(2)
    return weight.permute(0, 2, 3, 4, 1).contiguous()


class Decoder3d:
    """reference: vae.py:377-489."""

    def __init__(self, sd, dim, latent_hw, device, conv16=False):
        self.device = device
        # conv16: False = fp32 convolutions on the fp32 matrix instruction (the reference's precision); True = fp16 operands (opt-in fast
        # decode, 11 mantissa bits); "split" = hi/lo fp16 split of activations and weights, three 16-bit products per fp32 product
        # accumulated in fp32 (~22 mantissa bits per operand: fp32-grade results at 16-bit matrix speed)
        self.conv16 = conv16
        self.split = conv16 == "split"
        self.dims, self.plan = synth.wan_vae_decoder_plan(dim)
        self.w = {}
        for k, v in sd.items():
            if not k.startswith("decoder."):
                continue
            v = v.to(device=device, dtype=torch.float32)
            self.w[k] = _cl(v) if (k.endswith(".weight") and v.dim() >= 4) else v.reshape(-1).contiguous()
        self.w16 = {}
        self.w16_tail = {}
        if conv16:
            for k, v in self.w.items():
                if k.endswith(".weight") and v.dim() == 5 and v.shape[1] * v.shape[2] * v.shape[3] > 1 and v.shape[4] >= 32:
                    cin = v.shape[4]
                    cp = ((3 * cin if self.split else cin) + 63) // 64 * 64
                    w16 = torch.zeros((*v.shape[:4], cp), dtype=torch.float16, device=device)
                    hi = v.to(torch.float16)
                    w16[..., :cin] = hi
                    if self.split:  # [hi | lo * 2^12 | hi] against activations [hi | hi * 2^-12 | lo]: the scaled lo halves are normal fp16 numbers
                        w16[..., cin : 2 * cin] = ((v - hi.float()) * 4096.0).to(torch.float16)
                        w16[..., 2 * cin : 3 * cin] = hi
                    self.w16[k] = w16
                    self.w16_tail[k] = cp - (3 * cin if self.split else cin)  # zero channels behind the planes (weights and activations alike)
        self.h0, self.w0 = latent_hw
        self._bufs = {}
        self._pool = {}  # shared frame buffers by geometry (see _ConvInput)
        self._rep = {}  # upsample3d time-conv state: False until the first chunk has passed (the reference's "Rep", vae.py:113-115)

    # ---- buffers ------------------------------------------------------------------------------------------------------
    def _input(self, key, t, h, w, c, kt, pad, dtype=torch.float32, copy_cache=True):
        b = self._bufs.get(key)
        if b is None or b.t_max < t:
            nb = _ConvInput(max(t, 4 if kt == 3 else t), h, w, c, kt, pad, self.device, dtype, split=self.split and dtype == torch.float16, pool=self._pool)
            if b is not None and b.lead:
                nb.cache.copy_(b.cache)
            self._bufs[key] = b = nb
        return b.acquire(copy=copy_cache)

    def clear_cache(self):
        """reference: WanVAE_.clear_cache (vae.py:752-760)."""
        for b in self._bufs.values():
            b.reset()
        self._rep = {}

    # ---- layers -------------------------------------------------------------------------------------------------------
    def _conv_cached(self, key, name, x, gamma=None, silu=False, resid=None, flags=0, upsample=False, kt=3):
        """[RMS_norm → SiLU →] (cached, zero-padded) conv `name` on plain x [T,H,W,C]; returns plain [T',H',W',Cout]."""
        t, h, w, c = x.shape
        wt = self.w[name + ".weight"]
        cout, wkt, kh, kw, _ = wt.shape
        ho, wo = (2 * h, 2 * w) if upsample else (h, w)
        w16 = self.w16.get(name + ".weight")
        if w16 is not None and self.w16_tail[name + ".weight"] == 32:  # e.g. 3 x 96 = 288 planes in a 320-channel buffer: the 32-channel-slab kernel skips the zero slab
            flags |= lib.VCONV_ZERO_TAIL32
        # the 128-pixel kernel reads the 2-frame cache from its own tensor: no copy in front of the shared frame buffer
        sep = w16 is not None and lib.vae_conv16_cached_ok(wo, w16, flags)
        b = self._input(key, t, ho, wo, c, wkt, kh // 2, torch.float16 if w16 is not None else torch.float32, copy_cache=not sep)
        lib.vae_prep(x, b.interior(), b.strides[:2], gamma=gamma, silu=silu, upsample=upsample, split=self.split and w16 is not None)
        if flags & lib.VCONV_TSPLIT:
            out = torch.empty((2 * t, ho, wo, cout // 2), dtype=torch.float32, device=x.device)
        else:
            out = torch.empty((t, ho, wo, cout), dtype=torch.float32, device=x.device)
        if w16 is not None:
            lib.vae_conv16(b.buf, b.strides, w16, out, t, ho, wo, bias=self.w[name + ".bias"], resid=resid, flags=flags, cache=b.cache if sep else None)
        else:
            lib.vae_conv(b.buf, b.strides, wt, out, t, ho, wo, bias=self.w[name + ".bias"], resid=resid, flags=flags)
        b.roll(t, head_valid=not sep)
        return out

    def _conv1x1(self, name, x, resid=None):
        t, h, w, c = x.shape
        wt = self.w[name + ".weight"]
        out = torch.empty((t, h, w, wt.shape[0]), dtype=torch.float32, device=x.device)
        lib.vae_conv(x, (h * w * c, w * c, c), wt, out, t, h, w, bias=self.w[name + ".bias"], resid=resid)
        return out

    def residual_block(self, p, x):
        """reference: ResidualBlock.forward (vae.py:185-223)."""
        h = self._conv1x1(p + "shortcut", x) if (p + "shortcut.weight") in self.w else x
        y = self._conv_cached(p + "c1", p + "residual.2", x, gamma=self.w[p + "residual.0.gamma"], silu=True)
        return self._conv_cached(p + "c2", p + "residual.6", y, gamma=self.w[p + "residual.3.gamma"], silu=True, resid=h)

    def attention_block(self, p, x):
        """reference: AttentionBlock.forward (vae.py:226-262) — per frame, one head of dim C over h*w tokens."""
        t, h, w, c = x.shape
        n = h * w
        npad = (n + 15) // 16 * 16  # the PV GEMM reduces over keys in 16-float slabs: zero-padded token rows beyond n
        xn = torch.empty_like(x)
        lib.vae_prep(x, xn, (n * c, w * c), gamma=self.w[p + "norm.gamma"])
        qkv = self._conv1x1(p + "to_qkv", xn)  # [t, h, w, 3C]
        out = torch.empty_like(x)
        scores = torch.empty((npad, npad), dtype=torch.float32, device=x.device)
        o = torch.empty((1, 1, npad, c), dtype=torch.float32, device=x.device)
        qp = torch.zeros((npad, 3 * c), dtype=torch.float32, device=x.device) if npad > n else None
        for f in range(t):
            q = qkv[f].reshape(n, 3 * c)
            if qp is not None:
                qp[:n].copy_(q)
                q = qp
            k = q[:, c : 2 * c]
            vt = q[:, 2 * c :].t().contiguous()  # [C, npad]: the PV GEMM wants the reduction index contiguous
            # S = Q K^T: "pixels" = query tokens (stride 3C), "weights" = key rows (stride 3C)
            lib.vae_conv(q, (npad * 3 * c, npad * 3 * c, 3 * c), k, scores, 1, 1, npad, w_row_stride=3 * c, cin=c)
            lib.softmax_rows_causal_(scores, 1.0 / (c**0.5), npad, n_keys=n)  # hw = npad: no frame mask, only the padding columns are cut
            lib.vae_conv(scores, (npad * npad, npad * npad, npad), vt, o, 1, 1, npad)
            lib.vae_conv(o, (npad * c, w * c, c), self.w[p + "proj.weight"], out[f : f + 1], 1, h, w, bias=self.w[p + "proj.bias"], resid=x[f : f + 1])
        return out

    def resample(self, p, x, mode):
        """reference: Resample.forward, upsample2d / upsample3d (vae.py:108-143)."""
        if mode == "upsample3d":
            if not self._rep.get(p, False):
                self._rep[p] = True  # first chunk: no temporal upsampling, nothing cached
            else:
                x = self._conv_cached(p + "time", p + "time_conv", x, flags=lib.VCONV_TSPLIT)
        return self._conv_cached(p + "sp", p + "resample.1", x, upsample=True)

    def forward(self, x):
        """One chunk: x [T, h, w, z] (output of conv2) → [T', 8h, 8w, 3], clamped (the reference clamps after the
        concat of all chunks, vae.py:951-955 — elementwise, so per chunk is the same)."""
        x = self._conv_cached("conv1", "decoder.conv1", x)
        x = self.residual_block("decoder.middle.0.", x)
        x = self.attention_block("decoder.middle.1.", x)
        x = self.residual_block("decoder.middle.2.", x)
        for idx, kind, _, _ in self.plan:
            p = f"decoder.upsamples.{idx}."
            x = self.residual_block(p, x) if kind == "res" else self.resample(p, x, kind)
        return self._conv_cached("head", "decoder.head.2", x, gamma=self.w["decoder.head.0.gamma"], silu=True, flags=lib.VCONV_CLAMP)


def chunk_bounds(t, chunk_frames):
    """[a, b) latent-frame ranges of the decoder passes: the first frame alone (the one chunk without temporal upsampling, Resample "Rep",
    vae.py:113-115), then `chunk_frames` at a time."""
    edges = [0, 1] + list(range(1 + chunk_frames, t, chunk_frames)) + ([t] if t > 1 else [])
    return list(zip(edges[:-1], edges[1:]))


class WanVAE_:
    """reference: vae.py:640-760 (decode side)."""

    def __init__(self, sd, dim=96, z_dim=16, device="cuda", conv16="split", chunk_frames=4):
        """chunk_frames: latent frames per pass through the decoder after the first one (default 4: 88 GB of buffers at 720p, 2.8 % faster than 2 at 56 GB —
        fewer cache moves and tile tails; profiles/r06_call10_*).  The reference pushes ONE latent frame at a time
        through its cache-carrying decoder (vae.py:722-736) to bound memory; every kernel here reduces each output pixel in an order that
        does not depend on how many frames share the launch, and the 2-frame caches are the leading frames of the conv input buffers, so
        any chunking gives bit-identical output — larger chunks only fill the GPU better in the low-resolution stages
        (90x160 latents: 57 workgroups per frame and 32 output channels) and launch 4x fewer kernels."""
        self.dim, self.z_dim, self.device, self.conv16 = dim, z_dim, device, conv16
        self.chunk_frames = max(1, int(chunk_frames))
        self.sd = sd
        self.conv2_w = sd["conv2.weight"].to(device=device, dtype=torch.float32).reshape(z_dim, z_dim).contiguous()
        self.conv2_b = sd["conv2.bias"].to(device=device, dtype=torch.float32).contiguous()
        self.decoder = None

    def decode(self, z, scale):
        """z [1, 16, T, h, w] fp32; scale = [mean, inv_std] → [1, 3, 1 + 4 (T-1), 8h, 8w], clamped to [-1, 1]."""
        zc, t, h, w = z.shape[1:]
        if self.decoder is None or (self.decoder.h0, self.decoder.w0) != (h, w):
            self.decoder = Decoder3d(self.sd, self.dim, (h, w), self.device, conv16=self.conv16)
        self.decoder.clear_cache()
        zl = z[0].permute(1, 2, 3, 0).contiguous().float()  # [T, h, w, 16]
        zn = torch.empty_like(zl)
        lib.vae_prep(zl, zn, (h * w * zc, w * zc), a=scale[1].float().contiguous(), b=scale[0].float().contiguous())
        x = torch.empty_like(zn)
        lib.vae_conv(zn, (h * w * zc, w * zc, zc), self.conv2_w, x, t, h, w, bias=self.conv2_b)
        outs = [self.decoder.forward(x[a:b]) for a, b in chunk_bounds(t, self.chunk_frames)]
        self.decoder.clear_cache()
        video = torch.cat(outs, dim=0)  # [T_out, H, W, 3]
        return video.permute(3, 0, 1, 2).unsqueeze(0).contiguous()


class WanVAE:
    """reference: vae.py:789-957 (decode side; `use_tiling` is not built)."""

    def __init__(self, sd, z_dim=16, dim=96, device="cuda", parallel=False, conv16="split", chunk_frames=4):
        """conv16 selects how the 3x3(x3) convolutions multiply (accumulation, residual stream, norms and attention are fp32 in every mode):
          "split" (default)  activations and weights as hi + lo fp16 pairs (~22 mantissa bits), three 16-bit products per fp32 product:
                             fp32-grade — it meets the all-fp32 decode's tolerance against the reference's fp32 decode (vae.py:794) — at
                             2.3x the speed of the fp32 matrix instruction, which cannot do this decode under 4.07 s even at its peak;
          False              operands in fp32 on v_mfma_f32_32x32x2_f32;
          True               operands rounded to fp16 (11 mantissa bits; the precision class of cuDNN's TF32 default): fastest."""
        self.device, self.parallel = device, parallel
        self.mean = torch.tensor(synth.WAN_VAE_MEAN, dtype=torch.float32, device=device)
        self.inv_std = 1.0 / torch.tensor(synth.WAN_VAE_STD, dtype=torch.float32, device=device)
        self.scale = [self.mean, self.inv_std]
        self.model = WanVAE_(sd, dim=dim, z_dim=z_dim, device=device, conv16=conv16, chunk_frames=chunk_frames)

    def decode_dist(self, zs, world_size, cur_rank, split_dim):
        """reference: vae.py:883-929 — each rank decodes its slab of the latent (split along H = dim 2 or W = dim 3) plus a
        one-latent-pixel halo on each side (two on the outer side for the edge ranks), crops the halo's 8 output pixels and
        all-gathers the slabs."""
        import torch.distributed as dist

        total = zs.shape[split_dim]
        chunk, pad = total // world_size, 1
        if cur_rank == 0:
            lo, hi = 0, chunk + 2 * pad
        elif cur_rank == world_size - 1:
            lo, hi = total - (chunk + 2 * pad), total
        else:
            lo, hi = cur_rank * chunk - pad, (cur_rank + 1) * chunk + pad
        zs = zs.narrow(split_dim, lo, hi - lo).contiguous()
        images = self.model.decode(zs.unsqueeze(0).to(self.device), self.scale)
        px = split_dim + 1
        if cur_rank == 0:
            images = images.narrow(px, 0, chunk * 8)
        elif cur_rank == world_size - 1:
            images = images.narrow(px, images.shape[px] - chunk * 8, chunk * 8)
        else:
            images = images.narrow(px, 8 * pad, images.shape[px] - 16 * pad)
        images = images.contiguous()
        gathered = torch.empty((world_size * images.shape[0], *images.shape[1:]), dtype=images.dtype, device=images.device)
        dist.all_gather_into_tensor(gathered, images)
        return torch.cat(list(gathered.chunk(world_size, dim=0)), dim=px)

    def decode(self, zs, generator=None, config=None):
        """zs [16, T, h, w] → images [1, 3, T_out, 8h, 8w] fp32 in [-1, 1] (vae.py:931-957; `parallel` = the decode_dist
        branch when H or W divides by the world size, else the plain decode like the reference's fallback)."""
        if self.parallel:
            import torch.distributed as dist

            n, r = dist.get_world_size(), dist.get_rank()
            if zs.shape[3] % n == 0:
                return self.decode_dist(zs, n, r, 3)
            if zs.shape[2] % n == 0:
                return self.decode_dist(zs, n, r, 2)
        return self.model.decode(zs.unsqueeze(0).to(self.device), self.scale)
