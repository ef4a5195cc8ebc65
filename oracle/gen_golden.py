"""TEST INFRASTRUCTURE ONLY — generates `tests/golden/*.safetensors` by running the UNMODIFIED reference
(/root/reference, imported through oracle/ref_import.py) on CPU with fixed seeds.  The reference holds no
golden vectors of its own for this path, so these fixtures are what pins the oracle and
the HIP path.  Run in the authoring container only:

    python -m oracle.gen_golden            # rewrites tests/golden/

The fixtures hold inputs + reference outputs; weights are re-derived from `lightx2v_amd.synth` seeds (a
checksum of every weight tensor is stored so a drifting RNG is detected, not silently accepted).
"""
import os
import sys

import torch
from safetensors.torch import save_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from lightx2v_amd import synth  # noqa: E402
from oracle import ref_import  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
BF16 = torch.bfloat16


def weights_checksum(wd):
    acc = torch.zeros((), dtype=torch.float64)
    for k in sorted(wd):
        acc += wd[k].double().abs().sum()
    return acc.reshape(1)


def gen_ops():
    """Operator-level fixtures: each reference op object called directly on seeded inputs."""
    ref_import.patch_and_import()
    from lightx2v.utils.registry_factory import MM_WEIGHT_REGISTER, RMS_WEIGHT_REGISTER, LN_WEIGHT_REGISTER, ATTN_WEIGHT_REGISTER
    from lightx2v.models.networks.wan.infer.utils import compute_freqs, apply_rotary_emb, sinusoidal_embedding_1d
    from lightx2v.models.networks.wan.infer.pre_infer import WanPreInfer

    g = torch.Generator().manual_seed(1234)
    out = {}

    def rn(*shape, std=1.0):
        return (torch.randn(*shape, generator=g) * std).to(BF16)

    # MM Default  (mm_weight.py:70-96)
    M, K, N = 200, 256, 384
    x, w, b = rn(M, K), rn(N, K, std=0.06), rn(N, std=0.1)
    mmw = MM_WEIGHT_REGISTER["Default"]("w", "b")
    mmw.load({"w": w, "b": b})
    out.update(mm_x=x, mm_w=w, mm_b=b, mm_y=mmw.apply(x))
    mmw2 = MM_WEIGHT_REGISTER["Default"]("w", None)
    mmw2.load({"w": w})
    out.update(mm_y_nobias=mmw2.apply(x))

    # RMSNorm (rms_norm_weight.py:53-118; "sgl-kernel" key falls back to torch here)
    D = 768
    x, w = rn(130, D, std=2.0), (1 + rn(D, std=0.1).float()).to(BF16)
    rms = RMS_WEIGHT_REGISTER["sgl-kernel"]("w")
    rms.load({"w": w})
    out.update(rms_x=x, rms_w=w, rms_y=rms.apply(x))

    # LayerNorm no-affine + modulate, and affine (layer_norm_weight.py:78-111; transformer_infer.py:329-334,404)
    x = rn(130, D, std=3.0) + 0.5
    scale, shift = rn(1, D, std=0.3), rn(1, D, std=0.3)
    ln = LN_WEIGHT_REGISTER["Default"]()
    ln.load({})
    y = ln.apply(x)
    out.update(ln_x=x, ln_y=y.clone())
    y.mul_(1 + scale.squeeze(0)).add_(shift.squeeze(0))
    out.update(ln_scale=scale, ln_shift=shift, ln_mod_y=y)
    w3, b3 = (1 + rn(D, std=0.1).float()).to(BF16), rn(D, std=0.1)
    ln3 = LN_WEIGHT_REGISTER["Default"]("w", "b")
    ln3.load({"w": w3, "b": b3})
    out.update(ln_w=w3, ln_b=b3, ln_affine_y=ln3.apply(x))

    # gate-residual (transformer_infer.py:402,503) and plain residual (:468)
    xr, yr, gate = rn(130, D), rn(130, D), rn(1, D, std=0.5)
    x1 = xr.clone()
    x1.add_(yr * gate.squeeze(0))
    x2 = xr.clone()
    x2.add_(yr)
    out.update(res_x=xr, res_y=yr, res_gate=gate, res_gated=x1, res_plain=x2)

    # GELU-tanh (transformer_infer.py:492)
    h = rn(64, 512, std=2.0)
    out.update(gelu_x=h, gelu_y=torch.nn.functional.gelu(h, approximate="tanh"))

    # 3-axis RoPE (utils.py:7-20,107-115; table pre_infer.py:12-19), grid (3,4,6) → 72 tokens, 2 heads
    cfg = ref_import.make_config(dict(dim=256, ffn_dim=512, num_heads=2, num_layers=1))
    freqs = WanPreInfer(cfg).freqs
    grid = torch.tensor([[3, 4, 6]])
    q = rn(72, 2, 128)
    fi = compute_freqs(64, grid, freqs)
    out.update(rope_x=q, rope_y=apply_rotary_emb(q, fi), rope_grid=grid)

    # attention (attn_weight.py:209-239 torch_sdpa): self S=192, cross Sk=40, H=2, d=128
    attn = ATTN_WEIGHT_REGISTER["torch_sdpa"]()
    q, k, v = rn(192, 2, 128), rn(192, 2, 128), rn(192, 2, 128)
    out.update(attn_q=q, attn_k=k, attn_v=v, attn_o=attn.apply(q, k, v, max_seqlen_q=192, max_seqlen_kv=192))
    kc, vc = rn(40, 2, 128), rn(40, 2, 128)
    out.update(xattn_k=kc, xattn_v=vc, xattn_o=attn.apply(q, kc, vc, max_seqlen_q=192, max_seqlen_kv=40))

    # timestep sinusoid (utils.py:161-172)
    t = torch.tensor([999, 727, 3], dtype=torch.int64)
    out.update(sin_t=t, sin_y=sinusoidal_embedding_1d(256, t))

    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(GOLDEN, "ops.safetensors"))
    print("ops.safetensors:", len(out), "tensors")


def _ref_model_infer(R, sch, cfg, inputs):
    """wan/model.py:197-226 glue around the reference's own pre/transformer/post infer objects."""
    embed, grid_sizes, pre_out = R["pre"].infer(R["pre_w"], inputs, positive=True)
    x = R["tr"].infer(R["tr_w"], grid_sizes, embed, *pre_out)
    cond = R["post"].infer(R["post_w"], x, embed, grid_sizes)[0]
    sch.noise_pred = cond
    if cfg["enable_cfg"]:
        embed, grid_sizes, pre_out = R["pre"].infer(R["pre_w"], inputs, positive=False)
        x = R["tr"].infer(R["tr_w"], grid_sizes, embed, *pre_out)
        uncond = R["post"].infer(R["post_w"], x, embed, grid_sizes)[0]
        sch.noise_pred = uncond + cfg.sample_guide_scale * (sch.noise_pred - uncond)
    return cond


def gen_model(name="wan-tiny", workload="wan-tiny", steps=4):
    """Model-level fixtures: one block with per-phase outputs, one full forward, and the 4-step CFG denoise
    loop driven by the reference's WanScheduler (default_runner.py:97-114)."""
    ref_import.patch_and_import()
    from lightx2v.models.schedulers.wan.scheduler import WanScheduler

    dims = synth.WAN_DIMS[name]
    wl = synth.WORKLOADS[workload]
    wd = synth.synth_wan_weights(dims, seed=0)
    latents, ctx, ctx_null = synth.synth_inputs(dims, wl["target_shape"])
    cfg = ref_import.make_config(dims, target_shape=wl["target_shape"], target_video_length=wl["frames"], infer_steps=steps)
    R = ref_import.build_reference_wan(cfg, wd)
    sch = WanScheduler(cfg)
    sch.device = torch.device("cpu")
    sch.prepare()
    sch.latents = latents.clone()  # inject (appendix A.10)
    for m in ("pre", "post"):
        R[m].set_scheduler(sch)
    inputs = {"text_encoder_output": {"context": ctx, "context_null": ctx_null}}
    out = dict(latents0=latents, weights_checksum=weights_checksum(wd), timesteps=sch.timesteps.clone(), sigmas=sch.sigmas.clone())

    # --- single block, phase by phase (transformer_infer.py:289-306)
    sch.step_pre(0)
    embed, grid_sizes, (x, embed0, seq_lens, freqs, context) = R["pre"].infer(R["pre_w"], inputs, positive=True)
    out.update(pre_x=x.clone(), pre_embed=embed.clone(), pre_embed0=embed0.clone(), pre_context=context.clone())
    tr, blk = R["tr"], R["tr_w"].blocks[0]
    mods = tr.infer_modulation(blk.compute_phases[0], embed0)
    xb = x.clone()
    y_out = tr.infer_self_attn(blk.compute_phases[1], grid_sizes, xb, seq_lens, freqs, mods[0], mods[1])
    out.update(b0_self_attn_y=y_out.clone())
    xb, attn_out = tr.infer_cross_attn(blk.compute_phases[2], xb, context, y_out, mods[2])
    out.update(b0_x_after_self=xb.clone(), b0_cross_attn_out=attn_out.clone())
    y = tr.infer_ffn(blk.compute_phases[3], xb, attn_out, mods[3], mods[4])
    out.update(b0_x_after_cross=xb.clone(), b0_ffn_y=y.clone())
    xb = tr.post_process(xb, y, mods[5])
    out.update(b0_x_out=xb.clone())

    # --- denoise loop
    for i in range(steps):
        sch.step_pre(i)
        cond = _ref_model_infer(R, sch, cfg, inputs)
        if i == 0:
            out.update(step0_cond=cond.clone(), step0_noise_pred=sch.noise_pred.clone())
        sch.step_post()
        out[f"latents_after_step{i}"] = sch.latents.clone()

    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(GOLDEN, f"{name}_model.safetensors"))
    print(f"{name}_model.safetensors:", len(out), "tensors")


def gen_model_i2v(name="wan-tiny-i2v", steps=3):
    """The i2v branch of the same hot path (pre_infer.py:44-55,100-113; transformer_infer.py:405-455 — the only configuration the reference publishes
    numbers for): the reference's own pre / transformer / post infer objects built with task = "i2v" (36-channel patch embedding, img_emb MLP, per-block
    k_img / v_img / norm_k_img and the second cross-attention), pre-infer outputs, block 0 phase by phase, and a 3-step CFG denoise loop."""
    ref_import.patch_and_import()
    from lightx2v.models.schedulers.wan.scheduler import WanScheduler

    dims = synth.WAN_DIMS[name]
    ts, frames = (16, 3, 8, 8), 9
    wd = synth.synth_wan_i2v_weights(dims, seed=0)
    latents, ctx, ctx_null = synth.synth_inputs(dims, ts)
    image = synth.synth_i2v_inputs(dims, ts)
    cfg = ref_import.make_config(dims, task="i2v", in_dim=36, target_shape=ts, target_video_length=frames, infer_steps=steps, lat_h=ts[2], lat_w=ts[3])
    R = ref_import.build_reference_wan(cfg, wd)
    sch = WanScheduler(cfg)
    sch.device = torch.device("cpu")
    sch.prepare(image_encoder_output=image)
    sch.latents = latents.clone()
    for m in ("pre", "post"):
        R[m].set_scheduler(sch)
    inputs = {"text_encoder_output": {"context": ctx, "context_null": ctx_null}, "image_encoder_output": image}
    out = dict(latents0=latents, weights_checksum=weights_checksum(wd), clip_encoder_out=image["clip_encoder_out"], vae_encode_out=image["vae_encode_out"])
    sch.step_pre(0)
    embed, grid_sizes, (x, embed0, seq_lens, freqs, context) = R["pre"].infer(R["pre_w"], inputs, positive=True)
    out.update(pre_x=x.clone(), pre_embed0=embed0.clone(), pre_context=context.clone())
    tr, blk = R["tr"], R["tr_w"].blocks[0]
    mods = tr.infer_modulation(blk.compute_phases[0], embed0)
    xb = x.clone()
    y_out = tr.infer_self_attn(blk.compute_phases[1], grid_sizes, xb, seq_lens, freqs, mods[0], mods[1])
    xb, attn_out = tr.infer_cross_attn(blk.compute_phases[2], xb, context, y_out, mods[2])
    out.update(b0_x_after_self=xb.clone(), b0_cross_attn_out=attn_out.clone())
    y = tr.infer_ffn(blk.compute_phases[3], xb, attn_out, mods[3], mods[4])
    xb = tr.post_process(xb, y, mods[5])
    out.update(b0_x_out=xb.clone())
    for i in range(steps):
        sch.step_pre(i)
        cond = _ref_model_infer(R, sch, cfg, inputs)
        if i == 0:
            out.update(step0_cond=cond.clone(), step0_noise_pred=sch.noise_pred.clone())
        sch.step_post()
        out[f"latents_after_step{i}"] = sch.latents.clone()
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(GOLDEN, f"{name}_model.safetensors"))
    print(f"{name}_model.safetensors:", len(out), "tensors")


def gen_scheduler_only():
    """Scheduler known-answer fixture at the real step counts (50 steps shift 8; 4 steps shift 8) with a
    synthetic, deterministic `noise_pred` (so it pins UniPC without needing the DiT)."""
    ref_import.patch_and_import()
    from lightx2v.models.schedulers.wan.scheduler import WanScheduler

    out = {}
    for steps, shift in ((50, 8.0), (4, 8.0), (10, 3.0)):
        cfg = ref_import.make_config(synth.WAN_DIMS["wan-tiny"], infer_steps=steps, sample_shift=shift, target_shape=(16, 2, 4, 4))
        sch = WanScheduler(cfg)
        sch.device = torch.device("cpu")
        sch.prepare()
        g = torch.Generator().manual_seed(7)
        lat0 = torch.randn(16, 2, 4, 4, generator=g)
        sch.latents = lat0.clone()
        tag = f"s{steps}_sh{int(shift)}"
        out[f"{tag}_lat0"] = lat0
        out[f"{tag}_timesteps"] = sch.timesteps.clone()
        out[f"{tag}_sigmas"] = sch.sigmas.clone()
        for i in range(steps):
            sch.step_pre(i)
            # deterministic pseudo-model: depends on latents and step so errors propagate
            sch.noise_pred = torch.sin(sch.latents.float() * 1.3 + 0.1 * i) + 0.05 * i
            sch.step_post()
        out[f"{tag}_final"] = sch.latents.clone()
    # step-distill scheduler (BASELINE config #4): the reference re-noises with the unseeded global RNG (step_distill/scheduler.py:55);
    # the fixture seeds it (torch.manual_seed(100 + i)) right before each step_post so that the draw is reproducible
    from lightx2v.models.schedulers.wan.step_distill.scheduler import WanStepDistillScheduler

    cfg = ref_import.make_config(synth.WAN_DIMS["wan-tiny"], infer_steps=4, sample_shift=5.0, target_shape=(16, 2, 4, 4))
    cfg["denoising_step_list"] = [1000, 750, 500, 250]
    sch = WanStepDistillScheduler(cfg)
    sch.device = torch.device("cpu")
    sch.prepare(None)
    lat0 = torch.randn(16, 2, 4, 4, generator=torch.Generator().manual_seed(8))
    sch.latents = lat0.clone()
    out["distill_lat0"], out["distill_timesteps"], out["distill_sigmas"] = lat0, sch.timesteps.clone(), sch.sigmas.clone()
    for i in range(4):
        sch.step_pre(i)
        sch.noise_pred = torch.cos(sch.latents.float() * 0.7 + 0.2 * i)
        torch.manual_seed(100 + i)
        sch.step_post()
        out[f"distill_lat{i + 1}"] = sch.latents.clone()
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(GOLDEN, "scheduler.safetensors"))
    print("scheduler.safetensors:", len(out), "tensors")


TEA_COEFFS = [[-5.21862437e04, 9.23041404e03, -5.28275948e02, 1.36987616e01, -4.99875664e-02],
              [2.39676752e03, -1.31110545e03, 2.01331979e02, -8.29855975e00, 1.37887774e-01]]  # configs/caching/teacache/wan_t2v_1_3b_tea_480p.json


TEA_TEST_COEFFS = [[0, 0, 0, 1.0, 0], [0, 0, 0.5, 1.0, 0]]  # with random weights the released polynomials saturate (all-skip / all-calc);
# these make the rescaled distance O(1) so that threshold 2.5 yields a mixed calc/skip pattern on the tiny model


def gen_teacache(name="wan-tiny", workload="wan-tiny", steps=16, thresh=2.5):
    """TeaCache fixture: the reference's WanTransformerInferTeaCaching (wan/infer/feature_caching/transformer_infer.py:9-171)
    driving the 16-step CFG loop of the tiny model; records the per-step calc/skip decisions of both branches and the
    latents after every step."""
    ref_import.patch_and_import()
    from lightx2v.models.networks.wan.infer.feature_caching.transformer_infer import WanTransformerInferTeaCaching
    from lightx2v.models.schedulers.wan.scheduler import WanScheduler

    dims = synth.WAN_DIMS[name]
    wl = synth.WORKLOADS[workload]
    wd = synth.synth_wan_weights(dims, seed=0)
    latents, ctx, ctx_null = synth.synth_inputs(dims, wl["target_shape"])
    for use_ret in (True, False):
        cfg = ref_import.make_config(dims, target_shape=wl["target_shape"], target_video_length=wl["frames"], infer_steps=steps, feature_caching="Tea",
                                     coefficients=TEA_TEST_COEFFS, use_ret_steps=use_ret, teacache_thresh=thresh)
        R = ref_import.build_reference_wan(cfg, wd)
        R["tr"] = WanTransformerInferTeaCaching(cfg)
        sch = WanScheduler(cfg)
        sch.device = torch.device("cpu")
        sch.prepare()
        sch.latents = latents.clone()
        sch.caching_records = [True] * steps
        for m in ("pre", "post", "tr"):
            R[m].set_scheduler(sch)
        inputs = {"text_encoder_output": {"context": ctx, "context_null": ctx_null}}
        tag = "ret" if use_ret else "noret"
        out = {} if use_ret else out  # noqa: F821
        for i in range(steps):
            sch.step_pre(i)
            _ref_model_infer(R, sch, cfg, inputs)
            sch.step_post()
            out[f"{tag}_latents_after_step{i}"] = sch.latents.clone()
        out[f"{tag}_records_cond"] = torch.tensor(sch.caching_records, dtype=torch.int32)
        out[f"{tag}_records_uncond"] = torch.tensor(sch.caching_records_2, dtype=torch.int32)
        print(tag, "cond", "".join("C" if c else "." for c in sch.caching_records), "uncond", "".join("C" if c else "." for c in sch.caching_records_2))
    out.update(latents0=latents, thresh=torch.tensor([thresh]), steps=torch.tensor([steps]))
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(GOLDEN, "wan-tiny_teacache.safetensors"))
    print("wan-tiny_teacache.safetensors:", len(out), "tensors")


def gen_vae(dim=32, seed=1):
    """Wan VAE decode fixture: the reference's WanVAE_ (vae.py:640-738) at a reduced channel width (dim 32 →
    128/128/128/64/32 channels; same depth, same cache logic) decoding z [16,3,8,8] → [3,9,64,64] one latent frame
    at a time, followed by WanVAE.decode's clamp (vae.py:951-955)."""
    ref_import.patch_and_import()
    from lightx2v.models.video_encoders.hf.wan.vae import WanVAE_

    sd = synth.synth_wan_vae_weights(dim=dim, seed=seed)
    m = WanVAE_(dim=dim, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[], temperal_downsample=[False, True, True], dropout=0.0).eval()
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and not [k for k in missing if k.startswith(("decoder", "conv2"))], (missing, unexpected)
    g = torch.Generator().manual_seed(7)
    z = torch.randn(16, 3, 8, 8, generator=g)
    mean, inv_std = torch.tensor(synth.WAN_VAE_MEAN), 1.0 / torch.tensor(synth.WAN_VAE_STD)
    with torch.no_grad():
        raw = m.decode(z.unsqueeze(0), [mean, inv_std])[0]
    out = dict(z=z, mean=mean, inv_std=inv_std, decoded_raw=raw.clone(), decoded=raw.float().clamp_(-1, 1), dim=torch.tensor([dim]), seed=torch.tensor([seed]),
               weights_checksum=weights_checksum(sd))
    # decode_dist (vae.py:883-929), world size 2 and 3, split along W and along H: the reference's own slab / halo / crop code, run once
    # per rank in this process — torch.distributed.all_gather is replaced by a recorder (first pass: every rank's cropped slab is
    # captured; second pass: the gather is served from the captured slabs) and torch.cuda.synchronize by a no-op
    import lightx2v.models.video_encoders.hf.wan.vae as ref_vae

    wv = object.__new__(ref_vae.WanVAE)  # the constructor reads a checkpoint file; everything decode_dist touches is set here
    wv.model, wv.scale, wv.device, wv.dtype, wv.parallel, wv.use_tiling = m, [mean, inv_std], "cpu", torch.float32, True, False
    zd = torch.randn(16, 2, 6, 12, generator=torch.Generator().manual_seed(9))
    out["z_dist"] = zd
    real_gather, real_sync = ref_vae.dist.all_gather, torch.cuda.synchronize
    torch.cuda.synchronize = lambda *a, **k: None
    try:
        for world, split_dim in ((2, 3), (3, 3), (2, 2), (3, 2)):
            slabs = {}

            def record(full, img, _slabs=slabs):
                _slabs[record.rank] = img.clone()
                for i, t in enumerate(full):
                    t.copy_(_slabs.get(i, torch.zeros_like(t)) if _slabs.get(i, t).shape == t.shape else torch.zeros_like(t))

            ref_vae.dist.all_gather = record
            with torch.no_grad():
                for r in range(world):
                    record.rank = r
                    wv.decode_dist(zd, world, r, split_dim)
                record.rank = 0
                full = wv.decode_dist(zd, world, 0, split_dim)
            out[f"decoded_dist_w{world}_d{split_dim}"] = full[0].clone()
    finally:
        ref_vae.dist.all_gather, torch.cuda.synchronize = real_gather, real_sync
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(GOLDEN, "wan_vae_tiny.safetensors"))
    print("wan_vae_tiny.safetensors:", len(out), "tensors")


def gen_hunyuan(name="hunyuan-tiny", seed=4):
    """HunyuanVideo DiT fixtures from the reference's own infer objects at a reduced width (hidden 256, 2 heads of
    128, 2 double + 3 single blocks): scheduler tables, pre-infer outputs, one double block, one single block, the
    full forward.  Text mask all ones (then the `torch_sdpa` op's dense attention equals the flash varlen call)."""
    out = run_reference_hunyuan(synth.HUNYUAN_DIMS[name], synth.HUNYUAN_WORKLOADS[name]["target_shape"], seed)
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(GOLDEN, "hunyuan_tiny.safetensors"))
    print("hunyuan_tiny.safetensors:", len(out), "tensors; noise_pred", tuple(out["noise_pred"].shape), out["noise_pred"].dtype)


def gen_hunyuan_teacache(name="hunyuan-tiny", seed=4, steps=10, thresh=None):
    """HunyuanVideo TeaCache fixture (hunyuan/infer/feature_caching/transformer_infer.py:7-135): the reference's own
    `HunyuanTransformerInferTeaCaching` inside a `steps`-step denoise loop at the tiny width — the reference's pre-infer / post-infer per step, its
    TeaCache class deciding per step whether the block stack runs or the cached residual is re-applied, the Euler update
    (schedulers/hunyuan/scheduler.py:256-260, restated: the scheduler class itself needs `diffusers`).  Stored: the per-step decisions
    (caching_records), the accumulated distances, the transformer output and the latents of every step."""
    ref_import.patch_and_import()
    from easydict import EasyDict
    from lightx2v.common.modules.weight_module import WeightModule, WeightModuleList
    from lightx2v.models.networks.hunyuan.infer.feature_caching.transformer_infer import HunyuanTransformerInferTeaCaching
    from lightx2v.models.networks.hunyuan.infer.post_infer import HunyuanPostInfer
    from lightx2v.models.networks.hunyuan.infer.pre_infer import HunyuanPreInfer
    from lightx2v.models.networks.hunyuan.weights.post_weights import HunyuanPostWeights
    from lightx2v.models.networks.hunyuan.weights.pre_weights import HunyuanPreWeights
    from lightx2v.models.networks.hunyuan.weights.transformer_weights import HunyuanTransformerDoubleBlock, HunyuanTransformerSingleBlock, HunyuanTransformerWeights
    from lightx2v.models.schedulers.hunyuan import scheduler as ref_sched
    from oracle import hunyuan_oracle as HO

    dims = synth.HUNYUAN_DIMS[name]
    ts = synth.HUNYUAN_WORKLOADS[name]["target_shape"]
    wd = synth.synth_hunyuan_weights(dims, seed=seed)
    lat, text_states, text_mask, text_states_2 = synth.synth_hunyuan_inputs(dims, ts)
    thresh = synth.HUNYUAN_TEACACHE_TINY_THRESH if thresh is None else thresh
    cfg = EasyDict(task="t2v", do_mm_calib=False, mm_config={}, attention_type="torch_sdpa", cpu_offload=False, feature_caching="Tea", teacache_thresh=thresh)
    timesteps, sigmas = ref_sched.set_timesteps_sigmas(steps, 7.0, device=torch.device("cpu"))
    fc, fs = ref_sched.get_nd_rotary_pos_embed([16, 56, 56], [ts[2], ts[3] // 2, ts[4] // 2], theta=256, use_real=True, theta_rescale_factor=1)

    class Sched:  # what pre-/post-infer and the TeaCache class read (pre_infer.py:15-19, post_infer.py:19, feature_caching/transformer_infer.py:28,45-56)
        pass

    sch = Sched()
    sch.timesteps, sch.infer_steps, sch.caching_records = timesteps, steps, [True] * steps
    sch.freqs_cos, sch.freqs_sin = fc.to(BF16), fs.to(BF16)
    sch.guidance = torch.tensor([6.0], dtype=BF16) * 1000.0
    pre_w, post_w = HunyuanPreWeights(cfg), HunyuanPostWeights(cfg)
    tr_w = HunyuanTransformerWeights.__new__(HunyuanTransformerWeights)
    WeightModule.__init__(tr_w)
    tr_w.config = cfg
    tr_w.add_module("double_blocks", WeightModuleList([HunyuanTransformerDoubleBlock(i, cfg) for i in range(dims["double_blocks"])]))
    tr_w.add_module("single_blocks", WeightModuleList([HunyuanTransformerSingleBlock(i, cfg) for i in range(dims["single_blocks"])]))
    for w in (pre_w, post_w, tr_w):
        w.load(wd)

    class SDPA4D:
        def apply(self, q, k, v, attn_mask=None):
            x = torch.nn.functional.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), attn_mask=attn_mask)
            x = x.transpose(1, 2)
            return x.reshape(x.shape[0], x.shape[1], -1)

    pre_w.txt_in_attn_1 = SDPA4D()
    pre, tr, post = HunyuanPreInfer(cfg), HunyuanTransformerInferTeaCaching(cfg), HunyuanPostInfer(cfg)
    pre.heads_num = tr.heads_num = dims["heads"]
    tr.hidden_size, tr.mlp_hidden_dim = dims["hidden"], dims["mlp"]
    tr.double_blocks_num, tr.single_blocks_num = dims["double_blocks"], dims["single_blocks"]
    for m in (pre, post, tr):
        m.set_scheduler(sch)
    inputs = {"text_encoder_output": {"text_encoder_1_text_states": text_states, "text_encoder_1_attention_mask": text_mask, "text_encoder_2_text_states": text_states_2}}
    out = dict(latents0=lat, thresh=torch.tensor([thresh], dtype=torch.float64), timesteps=timesteps, sigmas=sigmas)
    latents, acc = lat.clone(), []
    with torch.no_grad():
        for i in range(steps):
            sch.step_index, sch.latents = i, latents.to(BF16)
            img, txt, vec, cu, max_len, freqs = pre.infer(pre_w, inputs)
            img_o, vec_o = tr.infer(tr_w, img, txt, vec, cu, max_len, freqs)
            noise_pred = post.infer(post_w, img_o, vec_o)
            latents = HO.euler_step(latents, noise_pred, sigmas, i)
            acc.append(float(tr.accumulated_rel_l1_distance))
            out[f"tr_img_{i}"] = img_o.clone()
            out[f"latents_{i}"] = latents.clone()
    out["records"] = torch.tensor([int(bool(r)) for r in sch.caching_records])
    out["accumulated"] = torch.tensor(acc, dtype=torch.float64)
    out["weights_checksum"] = weights_checksum(wd)
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(GOLDEN, "hunyuan_teacache.safetensors"))
    print("hunyuan_teacache.safetensors: records", out["records"].tolist(), "accumulated", [round(a, 4) for a in acc])
    return out


def run_reference_hunyuan(dims, ts, seed):
    """The reference run behind gen_hunyuan for any (dims, latent target_shape, seed): returns the tensor dict (also used live by
    tests/test_oracle_golden.py on shapes the committed fixture does not hold)."""
    ref_import.patch_and_import()
    from easydict import EasyDict
    from lightx2v.common.modules.weight_module import WeightModule, WeightModuleList
    from lightx2v.models.networks.hunyuan.infer.post_infer import HunyuanPostInfer
    from lightx2v.models.networks.hunyuan.infer.pre_infer import HunyuanPreInfer
    from lightx2v.models.networks.hunyuan.infer.transformer_infer import HunyuanTransformerInfer
    from lightx2v.models.networks.hunyuan.weights.post_weights import HunyuanPostWeights
    from lightx2v.models.networks.hunyuan.weights.pre_weights import HunyuanPreWeights
    from lightx2v.models.networks.hunyuan.weights.transformer_weights import HunyuanTransformerDoubleBlock, HunyuanTransformerSingleBlock, HunyuanTransformerWeights
    from lightx2v.models.schedulers.hunyuan import scheduler as ref_sched

    wd = synth.synth_hunyuan_weights(dims, seed=seed)
    lat, text_states, text_mask, text_states_2 = synth.synth_hunyuan_inputs(dims, ts)
    cfg = EasyDict(task="t2v", do_mm_calib=False, mm_config={}, attention_type="torch_sdpa", cpu_offload=False, feature_caching="NoCaching")
    out = {}

    # scheduler tables (pure functions of the reference module; diffusers stub only satisfies its import line)
    timesteps, sigmas = ref_sched.set_timesteps_sigmas(4, 7.0, device=torch.device("cpu"))
    rope_sizes = [ts[2], ts[3] // 2, ts[4] // 2]
    fc, fs = ref_sched.get_nd_rotary_pos_embed([16, 56, 56], rope_sizes, theta=256, use_real=True, theta_rescale_factor=1)
    fc, fs = fc.to(BF16), fs.to(BF16)
    out.update(sched_timesteps=timesteps, sched_sigmas=sigmas, freqs_cos=fc, freqs_sin=fs)

    class Sched:  # the attributes HunyuanPreInfer / HunyuanPostInfer read (pre_infer.py:15-19, post_infer.py:19)
        pass

    sch = Sched()
    sch.latents, sch.timesteps, sch.step_index = lat.to(BF16), timesteps, 1
    sch.freqs_cos, sch.freqs_sin = fc, fs
    sch.guidance = torch.tensor([6.0], dtype=BF16) * 1000.0

    pre_w, post_w = HunyuanPreWeights(cfg), HunyuanPostWeights(cfg)
    tr_w = HunyuanTransformerWeights.__new__(HunyuanTransformerWeights)  # same classes, fewer blocks than the hard-coded 20 + 40
    WeightModule.__init__(tr_w)
    tr_w.config = cfg
    tr_w.add_module("double_blocks", WeightModuleList([HunyuanTransformerDoubleBlock(i, cfg) for i in range(dims["double_blocks"])]))
    tr_w.add_module("single_blocks", WeightModuleList([HunyuanTransformerSingleBlock(i, cfg) for i in range(dims["single_blocks"])]))
    for w in (pre_w, post_w, tr_w):
        w.load(wd)

    class SDPA4D:  # 4-D form of TorchSDPAWeight.apply (attn_weight.py:229-239) for pre_infer.py:117-119,140 — see oracle header
        def apply(self, q, k, v, attn_mask=None):
            x = torch.nn.functional.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), attn_mask=attn_mask)
            x = x.transpose(1, 2)
            return x.reshape(x.shape[0], x.shape[1], -1)

    pre_w.txt_in_attn_1 = SDPA4D()
    pre, tr, post = HunyuanPreInfer(cfg), HunyuanTransformerInfer(cfg), HunyuanPostInfer(cfg)
    pre.heads_num = tr.heads_num = dims["heads"]
    tr.hidden_size, tr.mlp_hidden_dim = dims["hidden"], dims["mlp"]
    tr.double_blocks_num, tr.single_blocks_num = dims["double_blocks"], dims["single_blocks"]
    pre.set_scheduler(sch)
    post.set_scheduler(sch)
    inputs = {"text_encoder_output": {"text_encoder_1_text_states": text_states, "text_encoder_1_attention_mask": text_mask, "text_encoder_2_text_states": text_states_2}}
    with torch.no_grad():
        img, txt, vec, cu, max_len, freqs = pre.infer(pre_w, inputs)
        out.update(latents=lat, text_states=text_states, text_mask=text_mask, text_states_2=text_states_2, t=timesteps[1].reshape(1).clone(), guidance=sch.guidance.clone(),
                   pre_img=img.clone(), pre_txt=txt.clone(), pre_vec=vec.clone(), cu_seqlens=cu.clone(), max_seqlen=torch.tensor([max_len]))
        i1, t1 = tr.infer_double_block(tr_w.double_blocks[0], img, txt, vec, cu, max_len, freqs, None, None)
        out.update(d0_img=i1.clone(), d0_txt=t1.clone())
        x = torch.cat((i1, t1), 0)
        x1 = tr.infer_single_block(tr_w.single_blocks[0], x, vec, txt.shape[0], cu, max_len, freqs, None, None)
        out.update(s0_in=x.clone(), s0_out=x1.clone())
        img_o, vec_o = tr.infer(tr_w, img, txt, vec, cu, max_len, freqs)
        out.update(tr_img=img_o.clone())
        out.update(noise_pred=post.infer(post_w, img_o, vec_o).clone())
    out["weights_checksum"] = weights_checksum(wd)
    out["seed"] = torch.tensor([seed])
    return out


def gen_hunyuan_vae(seed=1):
    """HunyuanVideo VAE decode fixture: the reference's own AutoencoderKLCausal3D (autoencoder_kl_causal_3d.py) built at a reduced
    width and tile size (synth.HUNYUAN_VAE_TINY_CFG: 32/64/128/128 channels, 8 groups, 64-px / 16-frame tiles) and driven exactly as
    VideoEncoderKLCausal3DModel.decode does (model.py:33-44: scale, enable_tiling, decode, x/2+0.5, clamp) on z [1,16,6,12,10] — the
    temporal tiling path with 2x2 spatial tiles inside each temporal tile and every blend — plus one untiled decoder call.
    fp32 on CPU (the reference runs fp16 on the GPU).  `diffusers` is absent: oracle/ref_shims/diffusers supplies plumbing and a
    restated `Attention` (the mid block's single attention op) — everything else in the fixture is the reference's code."""
    ref_import.patch_and_import()
    from lightx2v.models.video_encoders.hf.autoencoder_kl_causal_3d.autoencoder_kl_causal_3d import AutoencoderKLCausal3D

    cfg = synth.HUNYUAN_VAE_TINY_CFG
    vae = AutoencoderKLCausal3D(
        in_channels=3, out_channels=3, down_block_types=("DownEncoderBlockCausal3D",) * 4, up_block_types=("UpDecoderBlockCausal3D",) * 4,
        block_out_channels=cfg["block_out_channels"], layers_per_block=cfg["layers_per_block"], latent_channels=cfg["latent_channels"],
        norm_num_groups=cfg["norm_num_groups"], sample_size=cfg["sample_size"], sample_tsize=cfg["sample_tsize"], scaling_factor=cfg["scaling_factor"],
        time_compression_ratio=cfg["time_compression_ratio"], spatial_compression_ratio=cfg["spatial_compression_ratio"], mid_block_add_attention=True,
    )
    sd = synth.synth_hunyuan_vae_weights(cfg, seed=seed)
    missing, unexpected = vae.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith(("encoder.", "quant_conv.")) for k in missing), (missing[:3], unexpected[:3])
    vae.requires_grad_(False)
    vae.eval()
    out = {}
    with torch.no_grad():
        z = torch.randn(1, 16, 6, 12, 10, generator=torch.Generator().manual_seed(3)) * 0.5
        latents = z / vae.config.scaling_factor
        vae.enable_tiling()
        image = vae.decode(latents, return_dict=False, generator=None)[0]
        out["z_tiled"], out["image_tiled"] = z, (image / 2 + 0.5).clamp(0, 1).float()
        z1 = torch.randn(1, 16, 3, 6, 5, generator=torch.Generator().manual_seed(5)) * 0.5  # fits one tile: plain decoder
        out["z_single"] = z1
        out["image_single"] = (vae.decode(z1 / vae.config.scaling_factor, return_dict=False)[0] / 2 + 0.5).clamp(0, 1).float()
    out["weights_checksum"] = weights_checksum(sd).reshape(1)
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(GOLDEN, "hunyuan_vae_tiny.safetensors"))
    print("hunyuan vae fixture:", {k: tuple(v.shape) for k, v in out.items()})


def gen_fp8():
    """w8a8-fp8 fixture (BASELINE config #4's GEMM leg): the reference's own operator class
    MMWeightWfp8channelAfp8channeldynamicVllm (common/ops/mm/mm_weight.py:287-319) — `weight_auto_quant` load from a bf16 weight and
    load of a converter-format pair, then apply() — executed on CPU over the restated vLLM kernels of oracle/ref_shims/vllm."""
    ref_import.patch_and_import()
    from lightx2v.utils.registry_factory import MM_WEIGHT_REGISTER

    gen = torch.Generator().manual_seed(12)
    M, K, N = 96, 256, 160
    x = torch.randn(M, K, generator=gen).to(torch.bfloat16)
    x[3] = 0  # an all-zero token: the scale floor of the dynamic quantiser
    x[5] *= 40
    w = (torch.randn(N, K, generator=gen) / K**0.5).to(torch.bfloat16)
    w[7] = 0  # an all-zero out channel: the 1e-5 clamp of the weight quantiser
    b = (torch.randn(N, generator=gen) * 0.1).to(torch.bfloat16)
    out = {"x": x, "w": w, "b": b}
    cls = MM_WEIGHT_REGISTER["W-fp8-channel-sym-A-fp8-channel-sym-dynamic-Vllm"]
    op = cls("w.weight", "w.bias")
    op.set_config({"weight_auto_quant": True})
    op.load({"w.weight": w.clone(), "w.bias": b.clone()})
    out["auto_wq"] = op.weight.t().contiguous().view(torch.uint8)  # stored [N, K]
    out["auto_wscale"] = op.weight_scale.clone()
    out["auto_y"] = op.apply(x.clone())
    xq, sx = op.act_quant_func(x.clone())
    out["xq"], out["sx"] = xq.view(torch.uint8), sx
    op2 = cls("w.weight", "w.bias")
    op2.set_config({})
    op2.load({"w.weight": op.weight.t().contiguous().clone(), "w.weight_scale": op.weight_scale.to(torch.bfloat16), "w.bias": b.clone()})  # as a loader hands it over
    out["ckpt_y"] = op2.apply(x.clone())
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(GOLDEN, "fp8_mm.safetensors"))
    print("fp8 fixture:", {k: tuple(v.shape) for k, v in out.items()})


CONVERTER_DIMS = dict(dim=64, ffn_dim=128, num_heads=1, num_layers=2, text_len=8, text_dim=64)
CONVERTER_CASES = {
    "fp8_by_block": dict(linear_dtype="torch.float8_e4m3fn", save_by_block=True, chunk_size=100),
    "int8_chunked": dict(linear_dtype="torch.int8", save_by_block=False, chunk_size=20),
}


def gen_converter():
    """Checkpoint-format fixture: the reference's own converter (tools/convert/converter.py::convert_weights, driven through its
    argparse namespace exactly as `main()` builds it, :668-707) quantising a small fp32 Wan checkpoint to e4m3 per-block files and to
    int8 chunk files.  Every tensor of every output file is stored as `<case>/<file>/<key>`, the index json as metadata.
    (qtorch is absent: oracle/ref_shims/qtorch restates float_quantize(4, 3, nearest), see there.)"""
    import argparse
    import importlib.util
    import json
    import tempfile

    from safetensors import safe_open

    ref_import.patch_and_import()
    spec = importlib.util.spec_from_file_location("ref_converter", os.path.join(ref_import.REFERENCE_ROOT, "tools", "convert", "converter.py"))
    conv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(conv)
    src = synth.synth_wan_weights(CONVERTER_DIMS, seed=11, dtype=torch.float32)
    out, meta = {}, {}
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "src"))
        save_file({k: v.contiguous() for k, v in src.items()}, os.path.join(tmp, "src", "model.safetensors"))
        for case, c in CONVERTER_CASES.items():
            args = argparse.Namespace(
                source=os.path.join(tmp, "src"), output=os.path.join(tmp, case), output_ext=".safetensors", output_name="converted", direction=None,
                chunk_size=c["chunk_size"], model_type="wan_dit", save_by_block=c["save_by_block"], quantized=True, bits=8, device="cpu",
                linear_dtype=eval(c["linear_dtype"]), non_linear_dtype=torch.float32, lora_path=None, lora_alpha=[1.0], copy_no_weight_files=False,
                key_idx=2, target_keys=["self_attn", "cross_attn", "ffn"], ignore_key=None,
            )
            conv.convert_weights(args)
            with open(os.path.join(args.output, "diffusion_pytorch_model.safetensors.index.json")) as fh:
                meta[case] = json.dumps(json.load(fh), sort_keys=True)
            for name in sorted(f for f in os.listdir(args.output) if f.endswith(".safetensors")):
                with safe_open(os.path.join(args.output, name), framework="pt") as fh:
                    for k in fh.keys():
                        out[f"{case}/{name}/{k}"] = fh.get_tensor(k).clone()
    save_file(out, os.path.join(GOLDEN, "converter_tiny.safetensors"), metadata=meta)
    print("converter fixture:", len(out), "tensors")


if __name__ == "__main__":
    os.makedirs(GOLDEN, exist_ok=True)
    torch.manual_seed(0)
    which = sys.argv[1:] or ["ops", "model", "sched", "vae", "hunyuan", "teacache", "hunyuan_teacache", "converter", "hunyuan_vae", "fp8"]
    if "ops" in which:
        gen_ops()
    if "model" in which:
        gen_model()
    if "model_i2v" in which or "model" in which:
        gen_model_i2v()
    if "sched" in which:
        gen_scheduler_only()
    if "vae" in which:
        gen_vae()
    if "hunyuan" in which:
        gen_hunyuan()
    if "teacache" in which:
        gen_teacache()
    if "hunyuan_teacache" in which:
        gen_hunyuan_teacache()
    if "converter" in which:
        gen_converter()
    if "hunyuan_vae" in which:
        gen_hunyuan_vae()
    if "fp8" in which:
        gen_fp8()
