"""Synthetic Wan2.1 DiT checkpoints and inputs (no network: no real checkpoints or prompts exist here).

Tensor names and shapes follow the checkpoint layout the reference binds by name
(reference: lightx2v/models/networks/wan/weights/transformer_weights.py:106,134-180,225-282,334-348,
pre_weights.py:17-40, post_weights.py:17-18).  Recipe: seeded normal, std 0.02 for
linear weights, norm weights 1 (+ small jitter so a swapped weight is detectable), biases small,
`modulation` std 0.1.
"""
import math
import torch

# name -> (dim, ffn_dim, num_heads, num_layers); dims come from the Wan2.1 checkpoints' config.json
WAN_DIMS = {
    "wan2.1-1.3b": dict(dim=1536, ffn_dim=8960, num_heads=12, num_layers=30, text_len=512, text_dim=4096),
    "wan2.1-14b": dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40, text_len=512, text_dim=4096),
    # miniature used by CPU tests / golden vectors (same head_dim=128 so RoPE split [22,21,21] is exercised)
    "wan-tiny": dict(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_len=32, text_dim=64),
    # 8 heads: the smallest model a group of 8 ranks can share (Ulysses needs heads % N == 0); plumbing runs of the N-rank code paths
    "wan-tiny-h8": dict(dim=1024, ffn_dim=2048, num_heads=8, num_layers=2, text_len=32, text_dim=64),
    # Wan2.1-I2V-14B (the configuration the reference publishes its numbers for, docs/EN/source/getting_started/benchmark_source.md:29-57): the 14B
    # DiT with the 36-channel patch embedding, the CLIP ViT-H/14 feature MLP (1280 -> D) and per block k_img / v_img / norm_k_img
    "wan2.1-14b-i2v": dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40, text_len=512, text_dim=4096, clip_dim=1280, task="i2v"),
    "wan-tiny-i2v": dict(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_len=32, text_dim=64, clip_dim=64, task="i2v"),
}

# BASELINE.json workloads: latent target_shape (C, T, H, W) (reference: wan_runner.py:260-280)
WORKLOADS = {
    "wan1.3b_256x256x17f": dict(model="wan2.1-1.3b", target_shape=(16, 5, 32, 32), frames=17),
    "wan1.3b_480px49f": dict(model="wan2.1-1.3b", target_shape=(16, 13, 60, 104), frames=49),
    "wan14b_720px81f": dict(model="wan2.1-14b", target_shape=(16, 21, 90, 160), frames=81),
    # two sizes between the BASELINE configs, used to calibrate the by-size rules (CFG pair pass / two-stream form): 32 760 and 75 600 tokens
    "wan14b_480px81f": dict(model="wan2.1-14b", target_shape=(16, 21, 60, 104), frames=81),
    "wan1.3b_720px81f": dict(model="wan2.1-1.3b", target_shape=(16, 21, 90, 160), frames=81),
    "wan1.3b_480px81f": dict(model="wan2.1-1.3b", target_shape=(16, 21, 60, 104), frames=81),  # 32 760 tokens x 12 heads: 1536 attention workgroups
    # the reference's published benchmark (configs/bench/lightx2v_2.json: I2V-14B, 81 frames, 40 steps, CFG scale 5, shift 5) at its two resolutions
    "wan14b_i2v_720px81f": dict(model="wan2.1-14b-i2v", target_shape=(16, 21, 90, 160), frames=81, infer_steps=40, sample_guide_scale=5.0, sample_shift=5.0),
    "wan14b_i2v_480px81f": dict(model="wan2.1-14b-i2v", target_shape=(16, 21, 60, 104), frames=81, infer_steps=40, sample_guide_scale=5.0, sample_shift=5.0),
    "wan-tiny-i2v": dict(model="wan-tiny-i2v", target_shape=(16, 3, 8, 8), frames=9),
    "wan-tiny": dict(model="wan-tiny", target_shape=(16, 3, 8, 8), frames=9),
    "wan-tiny-h8": dict(model="wan-tiny-h8", target_shape=(16, 3, 16, 12), frames=9),  # 144 tokens: divisible by 8
}


def seq_len_of(target_shape, patch=(1, 2, 2)):
    _, t, h, w = target_shape
    return (t // patch[0]) * (h // patch[1]) * (w // patch[2])


def _randn(shape, std, gen, device, dtype):
    if gen.device.type == "cpu":
        t = torch.randn(shape, generator=gen, dtype=torch.float32) * std
        return t.to(device=device, dtype=dtype)
    return (torch.randn(shape, generator=gen, dtype=torch.float32, device=gen.device) * std).to(dtype)


def synth_wan_weights(dims, seed=0, device="cpu", dtype=torch.bfloat16, gen_device="cpu", in_dim=16, out_dim=16, freq_dim=256):
    """Return {checkpoint tensor name: tensor}.  gen_device="cpu" gives a device-independent stream
    (parity tests); gen_device="cuda" fills large models directly in HBM (bench)."""
    D, F, L = dims["dim"], dims["ffn_dim"], dims["num_layers"]
    text_dim = dims.get("text_dim", 4096)
    gen = torch.Generator(device=gen_device)
    gen.manual_seed(seed)
    wd = {}

    def lin(name, n, k, std=0.02, bias_std=0.02):
        wd[f"{name}.weight"] = _randn((n, k), std, gen, device, dtype)
        wd[f"{name}.bias"] = _randn((n,), bias_std, gen, device, dtype)

    def norm_w(name, n):
        wd[name] = (1.0 + _randn((n,), 0.05, gen, device, torch.float32)).to(dtype)

    wd["patch_embedding.weight"] = _randn((D, in_dim, 1, 2, 2), 0.05, gen, device, dtype)
    wd["patch_embedding.bias"] = _randn((D,), 0.02, gen, device, dtype)
    lin("text_embedding.0", D, text_dim, std=1.0 / math.sqrt(text_dim))
    lin("text_embedding.2", D, D, std=1.0 / math.sqrt(D))
    lin("time_embedding.0", D, freq_dim, std=1.0 / math.sqrt(freq_dim))
    lin("time_embedding.2", D, D, std=1.0 / math.sqrt(D))
    lin("time_projection.1", 6 * D, D, std=0.5 / math.sqrt(D))
    for i in range(L):
        p = f"blocks.{i}"
        wd[f"{p}.modulation"] = _randn((1, 6, D), 0.1, gen, device, dtype)
        for attn in ("self_attn", "cross_attn"):
            for proj in ("q", "k", "v", "o"):
                lin(f"{p}.{attn}.{proj}", D, D, std=1.0 / math.sqrt(D))
            norm_w(f"{p}.{attn}.norm_q.weight", D)
            norm_w(f"{p}.{attn}.norm_k.weight", D)
        norm_w(f"{p}.norm3.weight", D)
        wd[f"{p}.norm3.bias"] = _randn((D,), 0.02, gen, device, dtype)
        lin(f"{p}.ffn.0", F, D, std=1.0 / math.sqrt(D))
        lin(f"{p}.ffn.2", D, F, std=1.0 / math.sqrt(F))
    lin("head.head", out_dim * 4, D, std=1.0 / math.sqrt(D))
    wd["head.modulation"] = _randn((1, 2, D), 0.1, gen, device, dtype)
    return wd


I2V_CLIP_TOKENS = 257  # CLIP ViT-H/14 tokens the i2v cross-attention sees (wan/infer/transformer_infer.py:406-407 splits the context there)


def synth_wan_i2v_weights(dims, seed=0, device="cpu", dtype=torch.bfloat16, gen_device="cpu"):
    """The i2v checkpoint (wan/weights/pre_weights.py:42-58, transformer_weights.py:285-312): the t2v tensors with a 36-channel patch embedding
    (16 noise + 4 mask + 16 conditioning-latent channels), the CLIP-feature MLP `img_emb.proj.{0,1,3,4}` and per block `cross_attn.{k_img,v_img,
    norm_k_img}`.  The extra tensors come from a second generator so that the base stream stays what `synth_wan_weights` draws."""
    wd = synth_wan_weights(dims, seed=seed, device=device, dtype=dtype, gen_device=gen_device, in_dim=36)
    D, L, C = dims["dim"], dims["num_layers"], dims.get("clip_dim", 1280)
    gen = torch.Generator(device=gen_device)
    gen.manual_seed(seed + 7777)

    def lin(name, n, k):
        wd[f"{name}.weight"] = _randn((n, k), 1.0 / math.sqrt(k), gen, device, dtype)
        wd[f"{name}.bias"] = _randn((n,), 0.02, gen, device, dtype)

    def ln(name, n):
        wd[f"{name}.weight"] = (1.0 + _randn((n,), 0.05, gen, device, torch.float32)).to(dtype)
        wd[f"{name}.bias"] = _randn((n,), 0.02, gen, device, dtype)

    ln("img_emb.proj.0", C)
    lin("img_emb.proj.1", C, C)
    lin("img_emb.proj.3", D, C)
    ln("img_emb.proj.4", D)
    for i in range(L):
        p = f"blocks.{i}.cross_attn"
        lin(f"{p}.k_img", D, D)
        lin(f"{p}.v_img", D, D)
        wd[f"{p}.norm_k_img.weight"] = (1.0 + _randn((D,), 0.05, gen, device, torch.float32)).to(dtype)
    return wd


def synth_i2v_inputs(dims, target_shape, seed=42, device="cpu"):
    """The image-encoder outputs the i2v path consumes (wan_runner.py:240-256): `clip_encoder_out` [257, clip_dim] bf16 (CLIP stand-in) and
    `vae_encode_out` [4 + 16, T, H, W] bf16 (first-frame mask + VAE latents of the conditioning video)."""
    g = torch.Generator().manual_seed(seed + 3)
    clip = torch.randn(I2V_CLIP_TOKENS, dims.get("clip_dim", 1280), generator=g).to(torch.bfloat16)
    _, t, h, w = target_shape
    msk = torch.zeros(4, t, h, w)
    msk[:, 0] = 1.0  # the first latent frame is given
    y = torch.cat([msk, torch.randn(16, t, h, w, generator=g)]).to(torch.bfloat16)
    return {"clip_encoder_out": clip.to(device), "vae_encode_out": y.to(device)}


def synth_inputs(dims, target_shape, seed=42, device="cpu"):
    """Latents exactly as the reference seeds them (wan/scheduler.py:25-28,54-63: randn(target_shape),
    fp32, seed 42) but always drawn from the CPU stream so CPU oracle and GPU path see identical noise
    (SURVEY.md appendix A.10); context/context_null stand in for the T5 output (seeds +1/+2)."""
    g = torch.Generator().manual_seed(seed)
    latents = torch.randn(*target_shape, generator=g, dtype=torch.float32)
    text_dim = dims.get("text_dim", 4096)
    text_len = dims.get("text_len", 512)
    n_tok = max(4, (text_len * 3) // 4)  # prompts are shorter than text_len; the rest is zero-padded by pre_infer
    g1 = torch.Generator().manual_seed(seed + 1)
    g2 = torch.Generator().manual_seed(seed + 2)
    ctx = torch.randn(n_tok, text_dim, generator=g1).to(torch.bfloat16)
    ctx_null = torch.randn(max(4, n_tok // 8), text_dim, generator=g2).to(torch.bfloat16)
    return latents.to(device), [ctx.to(device)], [ctx_null.to(device)]


# Wan2.1 VAE latent statistics (model constants of the released VAE: vae.py:804-839)
WAN_VAE_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508, 0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]
WAN_VAE_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743, 3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]


def wan_vae_decoder_plan(dim=96, dim_mult=(1, 2, 4, 4), num_res_blocks=2, temperal_upsample=(True, True, False)):
    """Decoder3d.__init__ (vae.py:377-434): channel plan and the `upsamples` Sequential as (index, kind, in_dim, out_dim)."""
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


def synth_wan_vae_weights(dim=96, z_dim=16, seed=0, device="cpu"):
    """Seeded fp32 weights of the Wan VAE *decoder* under the reference's state-dict names (`decoder.*`, `conv2.*`;
    WanVAE_ / Decoder3d module tree, vae.py:377-434,640-676).  Convs are scaled ~1/sqrt(fan_in) so activations stay O(1)."""
    gen = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(name, cout, cin, *k, gain=1.0):
        fan = cin
        for kk in k:
            fan *= kk
        sd[f"{name}.weight"] = (torch.randn((cout, cin, *k), generator=gen) * (gain / math.sqrt(fan))).to(device)
        sd[f"{name}.bias"] = (torch.randn((cout,), generator=gen) * 0.05).to(device)

    def gamma(name, c, *ones):
        sd[name] = (1.0 + 0.1 * torch.randn((c, *ones), generator=gen)).to(device)

    def res(p, cin, cout):
        gamma(p + "residual.0.gamma", cin, 1, 1, 1)
        conv(p + "residual.2", cout, cin, 3, 3, 3, gain=1.4)
        gamma(p + "residual.3.gamma", cout, 1, 1, 1)
        conv(p + "residual.6", cout, cout, 3, 3, 3, gain=1.4)
        if cin != cout:
            conv(p + "shortcut", cout, cin, 1, 1, 1)

    dims, plan = wan_vae_decoder_plan(dim)
    conv("conv2", z_dim, z_dim, 1, 1, 1)
    conv("decoder.conv1", dims[0], z_dim, 3, 3, 3)
    res("decoder.middle.0.", dims[0], dims[0])
    gamma("decoder.middle.1.norm.gamma", dims[0], 1, 1)
    conv("decoder.middle.1.to_qkv", dims[0] * 3, dims[0], 1, 1)
    conv("decoder.middle.1.proj", dims[0], dims[0], 1, 1)
    res("decoder.middle.2.", dims[0], dims[0])
    for idx, kind, cin, cout in plan:
        p = f"decoder.upsamples.{idx}."
        if kind == "res":
            res(p, cin, cout)
        else:
            conv(p + "resample.1", cin // 2, cin, 3, 3)
            if kind == "upsample3d":
                conv(p + "time_conv", cin * 2, cin, 3, 1, 1)
    gamma("decoder.head.0.gamma", dims[-1], 1, 1, 1)
    conv("decoder.head.2", 3, dims[-1], 3, 3, 3, gain=0.5)
    return sd


# HunyuanVideo DiT (reference: hunyuan/infer/transformer_infer.py:13-17 hard-codes the 13B numbers)
HUNYUAN_DIMS = {
    "hunyuan-13b": dict(hidden=3072, heads=24, mlp=12288, double_blocks=20, single_blocks=40, text_dim=4096, text_dim_2=768, text_len=256, refiner_mlp=12288),
    "hunyuan-tiny": dict(hidden=256, heads=2, mlp=512, double_blocks=2, single_blocks=3, text_dim=64, text_dim_2=64, text_len=16, refiner_mlp=512),
}
HUNYUAN_TEACACHE_TINY_THRESH = 0.33  # TeaCache threshold of the tiny fixture (tests/golden/hunyuan_teacache.safetensors): a mix of computed and skipped steps
HUNYUAN_WORKLOADS = {
    # latent target_shape (1, 16, T, H, W): 720p x 129 frames (BASELINE config #5) and a plumbing size
    "hunyuan13b_720px129f": dict(model="hunyuan-13b", target_shape=(1, 16, 33, 90, 160), frames=129),
    "hunyuan-tiny": dict(model="hunyuan-tiny", target_shape=(1, 16, 3, 8, 12), frames=9),
}


def synth_hunyuan_weights(dims, seed=0, device="cpu", dtype=torch.bfloat16, gen_device="cpu"):
    """{checkpoint tensor name: tensor} for the HunyuanVideo DiT under the reference's names
    (hunyuan/weights/pre_weights.py:9-84, transformer_weights.py:18-71, post_weights.py:10-11)."""
    h, mlp, hd = dims["hidden"], dims["mlp"], dims["hidden"] // dims["heads"]
    gen = torch.Generator(device=gen_device)
    gen.manual_seed(seed)
    wd = {}

    def lin(name, n, k, gain=1.0, bias_std=0.02):
        wd[f"{name}.weight"] = _randn((n, k), gain / math.sqrt(k), gen, device, dtype)
        wd[f"{name}.bias"] = _randn((n,), bias_std, gen, device, dtype)

    def norm_w(name, n):
        wd[name] = (1.0 + _randn((n,), 0.05, gen, device, torch.float32)).to(dtype)

    wd["img_in.proj.weight"] = _randn((h, 16, 1, 2, 2), 0.1, gen, device, dtype)
    wd["img_in.proj.bias"] = _randn((h,), 0.02, gen, device, dtype)
    lin("txt_in.input_embedder", h, dims["text_dim"])
    lin("txt_in.t_embedder.mlp.0", h, 256)
    lin("txt_in.t_embedder.mlp.2", h, h)
    lin("txt_in.c_embedder.linear_1", h, dims["text_dim"])
    lin("txt_in.c_embedder.linear_2", h, h)
    for j in range(2):
        p = f"txt_in.individual_token_refiner.blocks.{j}"
        norm_w(f"{p}.norm1.weight", h)
        wd[f"{p}.norm1.bias"] = _randn((h,), 0.02, gen, device, dtype)
        lin(f"{p}.self_attn_qkv", 3 * h, h)
        lin(f"{p}.self_attn_proj", h, h)
        norm_w(f"{p}.norm2.weight", h)
        wd[f"{p}.norm2.bias"] = _randn((h,), 0.02, gen, device, dtype)
        lin(f"{p}.mlp.fc1", dims["refiner_mlp"], h)
        lin(f"{p}.mlp.fc2", h, dims["refiner_mlp"])
        lin(f"{p}.adaLN_modulation.1", 2 * h, h, gain=0.5)
    lin("time_in.mlp.0", h, 256)
    lin("time_in.mlp.2", h, h)
    lin("vector_in.in_layer", h, dims["text_dim_2"])
    lin("vector_in.out_layer", h, h)
    lin("guidance_in.mlp.0", h, 256)
    lin("guidance_in.mlp.2", h, h)
    for i in range(dims["double_blocks"]):
        for s in ("img", "txt"):
            p = f"double_blocks.{i}.{s}"
            lin(f"{p}_mod.linear", 6 * h, h, gain=0.5)
            lin(f"{p}_attn_qkv", 3 * h, h)
            norm_w(f"{p}_attn_q_norm.weight", hd)
            norm_w(f"{p}_attn_k_norm.weight", hd)
            lin(f"{p}_attn_proj", h, h)
            lin(f"{p}_mlp.fc1", mlp, h)
            lin(f"{p}_mlp.fc2", h, mlp)
    for i in range(dims["single_blocks"]):
        p = f"single_blocks.{i}"
        lin(f"{p}.linear1", 3 * h + mlp, h)
        lin(f"{p}.linear2", h, h + mlp)
        norm_w(f"{p}.q_norm.weight", hd)
        norm_w(f"{p}.k_norm.weight", hd)
        lin(f"{p}.modulation.linear", 3 * h, h, gain=0.5)
    lin("final_layer.linear", 64, h)
    lin("final_layer.adaLN_modulation.1", 2 * h, h, gain=0.5)
    return wd


def synth_hunyuan_inputs(dims, target_shape, seed=42, valid_text=None):
    """latents fp32 [1,16,T,H,W] (cast by the caller), text_states [1,L,text_dim] bf16, text_mask [1,L] int64 (first
    `valid_text` tokens valid; default all), text_states_2 [1,text_dim_2] bf16 (CLIP pooled stand-in)."""
    g = torch.Generator().manual_seed(seed)
    latents = torch.randn(*target_shape, generator=g, dtype=torch.float32)
    L = dims["text_len"]
    text_states = torch.randn(1, L, dims["text_dim"], generator=g).to(torch.bfloat16)
    mask = torch.zeros(1, L, dtype=torch.int64)
    mask[:, : (L if valid_text is None else valid_text)] = 1
    text_states_2 = torch.randn(1, dims["text_dim_2"], generator=g).to(torch.bfloat16)
    return latents, text_states, mask, text_states_2


# HunyuanVideo VAE (AutoencoderKLCausal3D; config of hunyuan-video-t2v-720p/vae)
HUNYUAN_VAE_CFG = dict(block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=16, norm_num_groups=32, sample_size=256, sample_tsize=64,
                       scaling_factor=0.476986, time_compression_ratio=4, spatial_compression_ratio=8, tile_overlap_factor=0.25)
HUNYUAN_VAE_TINY_CFG = dict(block_out_channels=(32, 64, 128, 128), layers_per_block=2, latent_channels=16, norm_num_groups=8, sample_size=64, sample_tsize=16,
                            scaling_factor=0.476986, time_compression_ratio=4, spatial_compression_ratio=8, tile_overlap_factor=0.25)


def hunyuan_vae_up_plan(cfg):
    """DecoderCausal3D.__init__ (autoencoder_kl_causal_3d/vae.py:176-211): per up block (in_ch, out_ch, time factor, hw factor, has_upsampler)."""
    boc = list(reversed(cfg["block_out_channels"]))
    n_sp, n_t = int(math.log2(cfg["spatial_compression_ratio"])), int(math.log2(cfg["time_compression_ratio"]))
    plan, prev = [], boc[0]
    for i, out in enumerate(boc):
        final = i == len(boc) - 1
        sp, tm = i < n_sp, (i >= len(boc) - 1 - n_t and not final)
        plan.append((prev, out, 2 if tm else 1, 2 if sp else 1, sp or tm))
        prev = out
    return plan


def synth_hunyuan_vae_weights(cfg, seed=0, device="cpu"):
    """Seeded fp32 decoder weights of AutoencoderKLCausal3D under diffusers' state-dict names (`decoder.*`, `post_quant_conv.*`)."""
    gen = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(name, cout, cin, k, gain=1.0):
        sd[f"{name}.weight"] = (torch.randn((cout, cin, k, k, k), generator=gen) * (gain / math.sqrt(cin * k**3))).to(device)
        sd[f"{name}.bias"] = (torch.randn((cout,), generator=gen) * 0.05).to(device)

    def lin(name, cout, cin):
        sd[f"{name}.weight"] = (torch.randn((cout, cin), generator=gen) / math.sqrt(cin)).to(device)
        sd[f"{name}.bias"] = (torch.randn((cout,), generator=gen) * 0.05).to(device)

    def norm(name, c):
        sd[f"{name}.weight"] = (1.0 + 0.1 * torch.randn((c,), generator=gen)).to(device)
        sd[f"{name}.bias"] = (0.05 * torch.randn((c,), generator=gen)).to(device)

    def res(p, cin, cout):
        norm(p + "norm1", cin)
        conv(p + "conv1.conv", cout, cin, 3, gain=1.4)
        norm(p + "norm2", cout)
        conv(p + "conv2.conv", cout, cout, 3, gain=1.4)
        if cin != cout:
            conv(p + "conv_shortcut.conv", cout, cin, 1)

    zc, top = cfg["latent_channels"], cfg["block_out_channels"][-1]
    conv("post_quant_conv", zc, zc, 1)
    conv("decoder.conv_in.conv", top, zc, 3)
    res("decoder.mid_block.resnets.0.", top, top)
    norm("decoder.mid_block.attentions.0.group_norm", top)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        lin(f"decoder.mid_block.attentions.0.{n}", top, top)
    res("decoder.mid_block.resnets.1.", top, top)
    for i, (cin, cout, _, _, has_up) in enumerate(hunyuan_vae_up_plan(cfg)):
        for j in range(cfg["layers_per_block"] + 1):
            res(f"decoder.up_blocks.{i}.resnets.{j}.", cin if j == 0 else cout, cout)
        if has_up:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv.conv", cout, cout, 3)
    norm("decoder.conv_norm_out", cfg["block_out_channels"][0])
    conv("decoder.conv_out.conv", 3, cfg["block_out_channels"][0], 3, gain=0.5)
    return sd


def workload_setup(name, seed=0, device="cuda"):
    """Everything bench.py / tools/e2e.py need for a named workload: (dims, config overrides, weight dict, latents, inputs dict).  t2v: seeded
    weights + text context; i2v (dims["task"] == "i2v"): the i2v checkpoint, the config keys of configs/bench/lightx2v_2.json and the seeded
    CLIP / VAE-encode stand-ins of `synth_i2v_inputs` (the encoders are out of scope, SURVEY §2.1: their outputs are inputs here)."""
    wl = WORKLOADS[name]
    dims = WAN_DIMS[wl["model"]]
    i2v = dims.get("task") == "i2v"
    gen_device = device if str(device).startswith("cuda") else "cpu"
    wd = (synth_wan_i2v_weights if i2v else synth_wan_weights)(dims, seed=seed, device=device, gen_device=gen_device)
    lat, ctx, ctx_null = synth_inputs(dims, wl["target_shape"])
    inputs = {"text_encoder_output": {"context": [c.to(device) for c in ctx], "context_null": [c.to(device) for c in ctx_null]}}
    overrides = {k: wl[k] for k in ("sample_guide_scale", "sample_shift") if k in wl}
    if i2v:
        overrides.update(task="i2v", in_dim=36, cross_attn_2_type="hip_flash")
        inputs["image_encoder_output"] = {k: v.to(device) for k, v in synth_i2v_inputs(dims, wl["target_shape"]).items()}
    return dims, overrides, wd, lat, inputs
