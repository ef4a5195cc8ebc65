"""Pins the oracle (oracle/wan_oracle.py) against fixtures generated FROM THE REFERENCE
(oracle/gen_golden.py).  CPU only; bit-exact (same torch CPU kernels in the same order)."""
import pytest
import torch

from lightx2v_amd import synth
from oracle import wan_oracle as O


def eq(a, b):
    assert a.shape == b.shape and a.dtype == b.dtype
    assert torch.equal(a, b), f"max diff {(a.float() - b.float()).abs().max().item()}"


def test_ops_bit_exact(golden_ops):
    g = golden_ops
    eq(O.mm(g["mm_x"], g["mm_w"], g["mm_b"]), g["mm_y"])
    eq(O.mm(g["mm_x"], g["mm_w"]), g["mm_y_nobias"])
    eq(O.rms_norm(g["rms_x"], g["rms_w"]), g["rms_y"])
    eq(O.layer_norm(g["ln_x"]), g["ln_y"])
    y = O.layer_norm(g["ln_x"])
    y.mul_(1 + g["ln_scale"].squeeze(0)).add_(g["ln_shift"].squeeze(0))
    eq(y, g["ln_mod_y"])
    eq(O.layer_norm(g["ln_x"], g["ln_w"], g["ln_b"]), g["ln_affine_y"])
    f, h, w = g["rope_grid"][0].tolist()
    fi = O.compute_freqs(64, (f, h, w), O.rope_freqs_table(128))
    eq(O.apply_rotary_emb(g["rope_x"], fi), g["rope_y"])
    eq(O.sdpa(g["attn_q"], g["attn_k"], g["attn_v"]), g["attn_o"])
    eq(O.sdpa(g["attn_q"], g["xattn_k"], g["xattn_v"]), g["xattn_o"])
    eq(O.sinusoidal_embedding_1d(256, g["sin_t"]), g["sin_y"])


def test_fp32_legs_are_close(golden_ops):
    """The fp32-statistics variants (what the reference's GPU wheels compute) stay within bf16 rounding
    of the bf16-chain CPU path — this is the tolerance the HIP parity tests quote."""
    g = golden_ops
    y = O.rms_norm_fp32(g["rms_x"], g["rms_w"]).float()
    r = g["rms_y"].float()
    assert ((y - r).abs() <= 0.02 * r.abs() + 1e-2).all()
    o = O.attention_fp32(g["attn_q"], g["attn_k"], g["attn_v"])
    assert (o - g["attn_o"].float()).abs().max() < 2e-2


def test_block_and_forward_bit_exact(golden_model):
    g = golden_model
    dims = synth.WAN_DIMS["wan-tiny"]
    wl = synth.WORKLOADS["wan-tiny"]
    wd = synth.synth_wan_weights(dims, seed=0)
    acc = sum(wd[k].double().abs().sum() for k in sorted(wd))
    assert abs(acc.item() - g["weights_checksum"].item()) < 1e-6 * acc.item(), "synthetic weight stream drifted"
    lat, ctx, ctx_null = synth.synth_inputs(dims, wl["target_shape"])
    eq(lat, g["latents0"])
    t0 = g["timesteps"][0]
    embed, grid, x, embed0, s, context = O.wan_pre_infer(wd, dims, lat.to(torch.bfloat16), t0, ctx)
    eq(x, g["pre_x"])
    eq(embed, g["pre_embed"])
    eq(embed0, g["pre_embed0"])
    eq(context, g["pre_context"])
    tr = {}
    xb = O.wan_block(wd, 0, dims, grid, x.clone(), embed0, O.rope_freqs_table(128), context, trace=tr)
    eq(tr["x_after_self"], g["b0_x_after_self"])
    eq(xb, g["b0_x_out"])
    eq(O.wan_forward(wd, dims, lat.to(torch.bfloat16), t0, ctx), g["step0_cond"])


def test_denoise_loop_bit_exact(golden_model):
    g = golden_model
    dims = synth.WAN_DIMS["wan-tiny"]
    wl = synth.WORKLOADS["wan-tiny"]
    wd = synth.synth_wan_weights(dims, seed=0)
    lat, ctx, ctx_null = synth.synth_inputs(dims, wl["target_shape"])
    sch = O.WanSchedulerOracle(4, 8.0, lat)
    assert torch.equal(sch.timesteps, g["timesteps"])
    assert torch.equal(sch.sigmas, g["sigmas"])
    for i in range(4):
        sch.step_pre(i)
        sch.noise_pred = O.wan_model_infer(wd, dims, sch.latents, sch.timesteps[i], ctx, ctx_null, 6.0)
        if i == 0:
            eq(sch.noise_pred, g["step0_noise_pred"])
        sch.step_post()
        eq(sch.latents, g[f"latents_after_step{i}"])


def test_scheduler_known_answers(golden_sched):
    g = golden_sched
    for steps, shift in ((50, 8.0), (4, 8.0), (10, 3.0)):
        tag = f"s{steps}_sh{int(shift)}"
        sch = O.WanSchedulerOracle(steps, shift, g[f"{tag}_lat0"])
        assert torch.equal(sch.timesteps, g[f"{tag}_timesteps"])
        assert torch.equal(sch.sigmas, g[f"{tag}_sigmas"])
        for i in range(steps):
            sch.step_pre(i)
            sch.noise_pred = torch.sin(sch.latents.float() * 1.3 + 0.1 * i) + 0.05 * i
            sch.step_post()
        eq(sch.latents, g[f"{tag}_final"])


def test_vae_decode_oracle_bit_exact():
    """oracle/wan_vae_oracle.py reproduces the reference's WanVAE_.decode fixture (same torch CPU ops, same order)."""
    import os

    from safetensors.torch import load_file

    from lightx2v_amd import synth
    from oracle import wan_vae_oracle as V

    gld = load_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wan_vae_tiny.safetensors"))
    sd = synth.synth_wan_vae_weights(dim=int(gld["dim"]), seed=int(gld["seed"]))
    chk = sum(v.double().abs().sum() for _, v in sorted(sd.items()))
    assert torch.allclose(chk.reshape(1), gld["weights_checksum"], rtol=1e-12), "synthetic VAE weights drifted from the fixture's"
    with torch.no_grad():
        raw = V.wan_vae_decode(sd, gld["z"], gld["mean"], gld["inv_std"], dim=int(gld["dim"]), clamp=False)
        out = V.wan_vae_decode(sd, gld["z"], gld["mean"], gld["inv_std"], dim=int(gld["dim"]))
    assert torch.equal(raw, gld["decoded_raw"])
    assert torch.equal(out, gld["decoded"])
    # decode_dist (vae.py:883-929): the reference's own slab / halo / crop / gather code at world sizes 2 and 3, both split axes
    for world, split_dim in ((2, 3), (3, 3), (2, 2), (3, 2)):
        with torch.no_grad():
            got = V.wan_vae_decode_dist(sd, gld["z_dist"], gld["mean"], gld["inv_std"], world, split_dim, dim=int(gld["dim"]))
        assert torch.equal(got, gld[f"decoded_dist_w{world}_d{split_dim}"]), (world, split_dim)


def test_hunyuan_vae_oracle_bit_exact():
    """oracle/hunyuan_vae_oracle.py reproduces the fixture generated by running the reference's AutoencoderKLCausal3D
    (gen_golden.py::gen_hunyuan_vae): the tiled decode (temporal + spatial tiles, all blends) and the plain decoder."""
    import os

    from safetensors.torch import load_file

    from lightx2v_amd import synth
    from oracle import hunyuan_vae_oracle as V

    gld = load_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hunyuan_vae_tiny.safetensors"))
    cfg = synth.HUNYUAN_VAE_TINY_CFG
    sd = synth.synth_hunyuan_vae_weights(cfg, seed=1)
    chk = sum(v.double().abs().sum() for _, v in sorted(sd.items()))
    assert torch.allclose(chk.reshape(1), gld["weights_checksum"], rtol=1e-12), "synthetic Hunyuan VAE weights drifted from the fixture's"
    with torch.no_grad():
        assert torch.equal(V.vae_decode(sd, gld["z_single"], cfg), gld["image_single"])
        assert torch.equal(V.vae_decode(sd, gld["z_tiled"], cfg), gld["image_tiled"])


def test_hunyuan_oracle_bit_exact():
    """oracle/hunyuan_oracle.py reproduces the fixture generated from the reference's Hunyuan pre/transformer/post infer
    objects and scheduler functions (tests/golden/hunyuan_tiny.safetensors)."""
    import os

    from safetensors.torch import load_file

    from lightx2v_amd import synth
    from oracle import hunyuan_oracle as H

    g = load_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hunyuan_tiny.safetensors"))
    dims = synth.HUNYUAN_DIMS["hunyuan-tiny"]
    wd = synth.synth_hunyuan_weights(dims, seed=int(g["seed"]))
    ts = synth.HUNYUAN_WORKLOADS["hunyuan-tiny"]["target_shape"]
    tt, ss = H.set_timesteps_sigmas(4, 7.0)
    assert torch.equal(tt, g["sched_timesteps"]) and torch.equal(ss, g["sched_sigmas"])
    fc, fs = H.rope_tables([ts[2], ts[3] // 2, ts[4] // 2])
    assert torch.equal(fc, g["freqs_cos"]) and torch.equal(fs, g["freqs_sin"])
    with torch.no_grad():
        img, txt, vec, cu, ml = H.pre_infer(wd, dims, g["latents"].to(torch.bfloat16), g["t"][0], g["guidance"], g["text_states"], g["text_mask"], g["text_states_2"])
        assert torch.equal(img, g["pre_img"]) and torch.equal(txt, g["pre_txt"]) and torch.equal(vec, g["pre_vec"])
        assert cu.tolist() == g["cu_seqlens"].tolist() and ml == int(g["max_seqlen"])
        i1, t1 = H.double_block(wd, 0, img, txt, vec, (fc, fs), dims["heads"], cu)
        assert torch.equal(i1, g["d0_img"]) and torch.equal(t1, g["d0_txt"])
        x1 = H.single_block(wd, 0, g["s0_in"], vec, txt.shape[0], (fc, fs), dims["heads"], dims["hidden"], cu)
        assert torch.equal(x1, g["s0_out"])
        noise = H.forward(wd, dims, g["latents"].to(torch.bfloat16), g["t"][0], g["guidance"], g["text_states"], g["text_mask"], g["text_states_2"], (fc, fs))
        assert torch.equal(noise, g["noise_pred"])
    # the product-side table builder (lightx2v_amd.hunyuan.rope_tables) is the same function of the grid
    from lightx2v_amd import hunyuan as hy

    c2, s2 = hy.rope_tables([ts[2], ts[3] // 2, ts[4] // 2])
    assert torch.equal(c2, fc) and torch.equal(s2, fs)


def _hunyuan_teacache_inputs(g):
    from lightx2v_amd import synth

    dims = synth.HUNYUAN_DIMS["hunyuan-tiny"]
    ts = synth.HUNYUAN_WORKLOADS["hunyuan-tiny"]["target_shape"]
    wd = synth.synth_hunyuan_weights(dims, seed=4)
    lat, text_states, text_mask, text_states_2 = synth.synth_hunyuan_inputs(dims, ts)
    assert torch.equal(lat, g["latents0"])
    return dims, ts, wd, text_states, text_mask, text_states_2


def test_hunyuan_teacache_oracle_bit_exact():
    """oracle.hunyuan_oracle.TeaCacheOracle inside the oracle's 10-step loop reproduces the run of the reference's own
    HunyuanTransformerInferTeaCaching (tests/golden/hunyuan_teacache.safetensors, oracle/gen_golden.py::gen_hunyuan_teacache): same per-step
    decisions (two consecutive skipped steps among them), same accumulated distances, bit-identical transformer outputs and latents."""
    import os

    from safetensors.torch import load_file

    from oracle import hunyuan_oracle as H

    g = load_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hunyuan_teacache.safetensors"))
    dims, ts, wd, text_states, text_mask, text_states_2 = _hunyuan_teacache_inputs(g)
    steps, thresh = g["records"].numel(), float(g["thresh"])
    assert g["records"].tolist() == [1, 1, 0, 0, 1, 1, 1, 1, 1, 1], "the fixture must hold skipped steps"
    timesteps, sigmas = H.set_timesteps_sigmas(steps, 7.0)
    assert torch.equal(timesteps, g["timesteps"]) and torch.equal(sigmas, g["sigmas"])
    freqs = H.rope_tables([ts[2], ts[3] // 2, ts[4] // 2])
    guidance = torch.tensor([6.0], dtype=torch.bfloat16) * 1000.0
    tea = H.TeaCacheOracle(steps, thresh)
    latents = g["latents0"].clone()
    with torch.no_grad():
        for i in range(steps):
            img, txt, vec, cu, _ = H.pre_infer(wd, dims, latents.to(torch.bfloat16), timesteps[i], guidance, text_states, text_mask, text_states_2)
            img, vec = tea.infer(wd, dims, i, img, txt, vec, cu, freqs)
            assert torch.equal(img, g[f"tr_img_{i}"]), f"step {i}: transformer output"
            latents = H.euler_step(latents, H.post_infer(wd, img, vec, latents.shape), sigmas, i)
            assert torch.equal(latents, g[f"latents_{i}"]), f"step {i}: latents"
            assert float(tea.accumulated) == float(g["accumulated"][i]), (i, tea.accumulated, g["accumulated"][i])
    assert [int(r) for r in tea.records] == g["records"].tolist()


def test_teacache_oracle_bit_exact():
    """TeaCacheOracle + the oracle denoise loop reproduce the reference's WanTransformerInferTeaCaching run: same
    calc/skip decisions in both CFG branches and bit-identical latents after every step, both `use_ret_steps` modes."""
    import os

    from safetensors.torch import load_file

    from lightx2v_amd import synth
    from oracle import wan_oracle as O
    from oracle.gen_golden import TEA_TEST_COEFFS

    g = load_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wan-tiny_teacache.safetensors"))
    dims = synth.WAN_DIMS["wan-tiny"]
    wd = synth.synth_wan_weights(dims, seed=0)
    _, ctx, ctx_null = synth.synth_inputs(dims, synth.WORKLOADS["wan-tiny"]["target_shape"])
    steps, thresh = int(g["steps"]), float(g["thresh"])
    for tag, use_ret in (("ret", True), ("noret", False)):
        tea = O.TeaCacheOracle(steps, thresh, TEA_TEST_COEFFS, use_ret)
        bad = []
        O.denoise_loop(wd, dims, g["latents0"], ctx, ctx_null, steps, 8.0, 6.0, True, tea,
                       step_callback=lambda i, lat: bad.append(i) if not torch.equal(lat, g[f"{tag}_latents_after_step{i}"]) else None)
        assert [int(c) for c in tea.records[True]] == g[f"{tag}_records_cond"].tolist()
        assert [int(c) for c in tea.records[False]] == g[f"{tag}_records_uncond"].tolist()
        assert not bad, (tag, bad)
        assert 0 < sum(tea.records[True]) < steps  # the fixture exercises both paths


def test_fp8_oracle_reproduces_reference_class_outputs():
    """oracle/wan_oracle.py's w8a8 functions against the fixture produced by the reference's own fp8 operator class
    (gen_golden.py::gen_fp8; the two vLLM kernels under it are restated stubs, everything else is the reference's code)."""
    import os

    from safetensors.torch import load_file

    g = load_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fp8_mm.safetensors"))
    wq, sw = O.quant_fp8_weight_per_channel(g["w"])
    assert torch.equal(wq.view(torch.uint8), g["auto_wq"]) and torch.equal(sw, g["auto_wscale"])
    assert sw[7].item() == (torch.tensor(1e-5) / 448.0).item()  # all-zero channel: the quantiser's clamp
    xq, sx = O.quant_fp8_per_token(g["x"])
    assert torch.equal(xq.view(torch.uint8), g["xq"]) and torch.equal(sx, g["sx"])
    assert sx[3].item() == torch.tensor(1.0 / (448.0 * 512.0), dtype=torch.float32).item()  # all-zero token: the dynamic quantiser's scale floor
    assert torch.equal(O.mm_fp8(g["x"], wq, sw, g["b"]), g["auto_y"])
    assert torch.equal(O.mm_fp8(g["x"], wq, sw.to(torch.bfloat16).float(), g["b"]), g["ckpt_y"])


@pytest.mark.parametrize("case", [
    dict(dim=384, ffn_dim=640, num_heads=3, num_layers=1, text_len=20, text_dim=48, ts=(16, 2, 6, 10), seed=2),
    dict(dim=128, ffn_dim=256, num_heads=1, num_layers=3, text_len=8, text_dim=32, ts=(16, 5, 4, 6), seed=3),
    dict(dim=512, ffn_dim=1024, num_heads=4, num_layers=1, text_len=64, text_dim=64, ts=(16, 1, 14, 18), seed=4),
])
def test_wan_oracle_bit_exact_against_live_reference_on_other_shapes(case):
    """Where /root/reference exists: the unmodified reference (pre-infer, block stack, post-infer, CFG combine, UniPC step) run side
    by side with the oracle on architectures and latent grids the committed fixture does not hold (odd head counts, one frame,
    non-square grids, short text) — two denoise steps, noise predictions equal bit for bit."""
    from oracle import ref_import

    if not ref_import.reference_available():
        pytest.skip("reference checkout not present (authoring container only)")
    ref_import.patch_and_import()
    from lightx2v.models.schedulers.wan.scheduler import WanScheduler as RefScheduler

    from oracle.gen_golden import _ref_model_infer

    dims = {k: v for k, v in case.items() if k not in ("ts", "seed")}
    ts = case["ts"]
    wd = synth.synth_wan_weights(dims, seed=case["seed"])
    lat, ctx, ctx_null = synth.synth_inputs(dims, ts)
    cfg = ref_import.make_config(dims, target_shape=ts, target_video_length=(ts[1] - 1) * 4 + 1, infer_steps=3)
    R = ref_import.build_reference_wan(cfg, wd)
    sch = RefScheduler(cfg)
    sch.device = torch.device("cpu")
    sch.prepare()
    sch.latents = lat.clone()
    for m in ("pre", "post"):
        R[m].set_scheduler(sch)
    inputs = {"text_encoder_output": {"context": ctx, "context_null": ctx_null}}
    for i in range(2):
        sch.step_pre(i)
        _ref_model_infer(R, sch, cfg, inputs)
        mine = O.wan_model_infer(wd, dims, sch.latents, sch.timesteps[i], ctx, ctx_null, 6.0)
        assert torch.equal(mine, sch.noise_pred), f"step {i}"
        sch.step_post()


@pytest.mark.parametrize("dim,seed,zshape", [(16, 3, (16, 1, 6, 10)), (32, 5, (16, 4, 4, 4)), (48, 6, (16, 2, 10, 6))])
def test_wan_vae_oracle_bit_exact_against_live_reference_on_other_shapes(dim, seed, zshape):
    """Where /root/reference exists: the reference's WanVAE_.decode side by side with the oracle at other channel widths and latent
    shapes (a single latent frame — no temporal cache reuse —, four frames, non-square grids)."""
    from oracle import ref_import

    if not ref_import.reference_available():
        pytest.skip("reference checkout not present (authoring container only)")
    ref_import.patch_and_import()
    from lightx2v.models.video_encoders.hf.wan.vae import WanVAE_

    from oracle import wan_vae_oracle as V

    sd = synth.synth_wan_vae_weights(dim=dim, seed=seed)
    m = WanVAE_(dim=dim, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[], temperal_downsample=[False, True, True], dropout=0.0).eval()
    m.load_state_dict(sd, strict=False)
    z = torch.randn(*zshape, generator=torch.Generator().manual_seed(seed))
    mean, inv_std = torch.tensor(synth.WAN_VAE_MEAN), 1.0 / torch.tensor(synth.WAN_VAE_STD)
    with torch.no_grad():
        raw = m.decode(z.unsqueeze(0), [mean, inv_std])[0]
        mine = V.wan_vae_decode(sd, z, mean, inv_std, dim=dim, clamp=False)
    assert torch.equal(raw.reshape(mine.shape), mine)


@pytest.mark.parametrize("ts,seed,over", [((1, 16, 1, 12, 8), 9, dict(text_len=8)), ((1, 16, 5, 4, 20), 10, dict(double_blocks=1, single_blocks=2, mlp=768))])
def test_hunyuan_oracle_bit_exact_against_live_reference_on_other_shapes(ts, seed, over):
    """Where /root/reference exists: the reference's Hunyuan pre / double / single / post infer objects run live
    (gen_golden.run_reference_hunyuan) on another latent grid, text length and block plan than the committed fixture's."""
    from oracle import ref_import

    if not ref_import.reference_available():
        pytest.skip("reference checkout not present (authoring container only)")
    from oracle import gen_golden as G
    from oracle import hunyuan_oracle as H

    dims = dict(synth.HUNYUAN_DIMS["hunyuan-tiny"], **over)
    g = G.run_reference_hunyuan(dims, ts, seed)
    wd = synth.synth_hunyuan_weights(dims, seed=seed)
    fc, fs = H.rope_tables([ts[2], ts[3] // 2, ts[4] // 2])
    assert torch.equal(fc, g["freqs_cos"]) and torch.equal(fs, g["freqs_sin"])
    with torch.no_grad():
        img, txt, vec, cu, ml = H.pre_infer(wd, dims, g["latents"].to(torch.bfloat16), g["t"][0], g["guidance"], g["text_states"], g["text_mask"], g["text_states_2"])
        assert torch.equal(img, g["pre_img"]) and torch.equal(txt, g["pre_txt"]) and torch.equal(vec, g["pre_vec"])
        i1, t1 = H.double_block(wd, 0, img, txt, vec, (fc, fs), dims["heads"], cu)
        assert torch.equal(i1, g["d0_img"]) and torch.equal(t1, g["d0_txt"])
        noise = H.forward(wd, dims, g["latents"].to(torch.bfloat16), g["t"][0], g["guidance"], g["text_states"], g["text_mask"], g["text_states_2"], (fc, fs))
        assert torch.equal(noise, g["noise_pred"])


@pytest.mark.parametrize("seed,shape", [(2, (1, 16, 2, 11, 9)), (3, (1, 16, 5, 7, 13)), (4, (1, 16, 9, 8, 8))])
def test_hunyuan_vae_oracle_bit_exact_against_live_reference_on_other_shapes(seed, shape):
    """Where /root/reference exists: the reference's AutoencoderKLCausal3D (tiling enabled, as VideoEncoderKLCausal3DModel.decode runs
    it) side by side with the oracle on latent shapes with ragged spatial tiles, spatial-only tiling and temporal tiling of a grid
    that fits one spatial tile."""
    from oracle import ref_import

    if not ref_import.reference_available():
        pytest.skip("reference checkout not present (authoring container only)")
    ref_import.patch_and_import()
    from lightx2v.models.video_encoders.hf.autoencoder_kl_causal_3d.autoencoder_kl_causal_3d import AutoencoderKLCausal3D

    from oracle import hunyuan_vae_oracle as V

    cfg = synth.HUNYUAN_VAE_TINY_CFG
    vae = AutoencoderKLCausal3D(
        in_channels=3, out_channels=3, down_block_types=("DownEncoderBlockCausal3D",) * 4, up_block_types=("UpDecoderBlockCausal3D",) * 4,
        block_out_channels=cfg["block_out_channels"], layers_per_block=cfg["layers_per_block"], latent_channels=cfg["latent_channels"],
        norm_num_groups=cfg["norm_num_groups"], sample_size=cfg["sample_size"], sample_tsize=cfg["sample_tsize"], scaling_factor=cfg["scaling_factor"],
        time_compression_ratio=cfg["time_compression_ratio"], spatial_compression_ratio=cfg["spatial_compression_ratio"], mid_block_add_attention=True,
    )
    sd = synth.synth_hunyuan_vae_weights(cfg, seed=seed)
    vae.load_state_dict(sd, strict=False)
    vae.requires_grad_(False)
    vae.eval()
    vae.enable_tiling()
    z = torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * 0.5
    with torch.no_grad():
        ref = (vae.decode(z / vae.config.scaling_factor, return_dict=False, generator=None)[0] / 2 + 0.5).clamp(0, 1).float()
        assert torch.equal(V.vae_decode(sd, z, cfg), ref)


def test_fp8_oracle_bit_exact_against_live_reference_class_on_random_shapes():
    """Where /root/reference exists: the reference's MMWeightWfp8channelAfp8channeldynamicVllm (auto-quantised load + apply, over the
    restated vLLM stubs) side by side with the oracle's w8a8 functions on random shapes and magnitudes, with and without bias, with
    all-zero tokens and out channels."""
    from oracle import ref_import

    if not ref_import.reference_available():
        pytest.skip("reference checkout not present (authoring container only)")
    ref_import.patch_and_import()
    from lightx2v.utils.registry_factory import MM_WEIGHT_REGISTER

    cls = MM_WEIGHT_REGISTER["W-fp8-channel-sym-A-fp8-channel-sym-dynamic-Vllm"]
    g = torch.Generator().manual_seed(1)
    for trial in range(12):
        M, K, N = (int(torch.randint(1, 9, (1,), generator=g)) * m for m in (7, 32, 16))
        mag = 10.0 ** float(torch.randint(-3, 3, (1,), generator=g))
        x = (torch.randn(M, K, generator=g) * mag).to(torch.bfloat16)
        w = (torch.randn(N, K, generator=g) / K**0.5).to(torch.bfloat16)
        b = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16)
        if trial % 3 == 0:
            x[0] = 0
            w[-1] = 0
        use_bias = trial % 4 != 1
        op = cls("w.weight", "w.bias" if use_bias else None)
        op.set_config({"weight_auto_quant": True})
        op.load({"w.weight": w.clone(), "w.bias": b.clone()})
        wq, sw = O.quant_fp8_weight_per_channel(w)
        assert torch.equal(op.apply(x.clone()), O.mm_fp8(x, wq, sw, b if use_bias else None)), (trial, M, K, N)
        xq, sx = op.act_quant_func(x.clone())
        mq, ms = O.quant_fp8_per_token(x)
        assert torch.equal(xq.view(torch.uint8), mq.view(torch.uint8)) and torch.equal(sx.reshape(-1), ms.reshape(-1))


@pytest.mark.parametrize("steps,thresh,use_ret", [(12, 4.0, True), (12, 8.0, False)])
def test_teacache_oracle_bit_exact_against_live_reference(steps, thresh, use_ret):
    """Where /root/reference exists: the reference's WanTransformerInferTeaCaching driving a CFG loop on another latent grid, weight
    seed, step count and threshold than the committed fixture — same calc/skip decisions for both branches, same latents each step."""
    from oracle import ref_import

    if not ref_import.reference_available():
        pytest.skip("reference checkout not present (authoring container only)")
    ref_import.patch_and_import()
    from lightx2v.models.networks.wan.infer.feature_caching.transformer_infer import WanTransformerInferTeaCaching
    from lightx2v.models.schedulers.wan.scheduler import WanScheduler as RefScheduler

    from oracle import gen_golden as G

    dims, ts = synth.WAN_DIMS["wan-tiny"], (16, 2, 8, 12)
    wd = synth.synth_wan_weights(dims, seed=5)
    lat, ctx, ctx_null = synth.synth_inputs(dims, ts)
    cfg = ref_import.make_config(dims, target_shape=ts, target_video_length=5, infer_steps=steps, feature_caching="Tea", coefficients=G.TEA_TEST_COEFFS,
                                 use_ret_steps=use_ret, teacache_thresh=thresh)
    R = ref_import.build_reference_wan(cfg, wd)
    R["tr"] = WanTransformerInferTeaCaching(cfg)
    sch = RefScheduler(cfg)
    sch.device = torch.device("cpu")
    sch.prepare()
    sch.latents = lat.clone()
    sch.caching_records = [True] * steps
    for m in ("pre", "post", "tr"):
        R[m].set_scheduler(sch)
    inputs = {"text_encoder_output": {"context": ctx, "context_null": ctx_null}}
    ref_lat = []
    for i in range(steps):
        sch.step_pre(i)
        G._ref_model_infer(R, sch, cfg, inputs)
        sch.step_post()
        ref_lat.append(sch.latents.clone())
    tea = O.TeaCacheOracle(steps, thresh, G.TEA_TEST_COEFFS, use_ret)
    mine = []
    O.denoise_loop(wd, dims, lat, ctx, ctx_null, steps, 8.0, 6.0, True, tea, step_callback=lambda i, latents: mine.append(latents.clone()))
    assert [bool(c) for c in tea.records[True]] == [bool(c) for c in sch.caching_records]
    assert [bool(c) for c in tea.records[False]] == [bool(c) for c in sch.caching_records_2]
    assert 0 < sum(bool(c) for c in sch.caching_records) < steps  # both the compute and the skip path ran
    assert len(mine) == steps and all(torch.equal(a, b) for a, b in zip(mine, ref_lat))


def test_fp8_block_mode_bit_exact_against_live_reference_model():
    """Where /root/reference exists: the reference's Wan forward built with `mm_type: W-fp8-channel-sym-A-fp8-channel-sym-dynamic-Vllm`,
    `weight_auto_quant` (BASELINE config #4's operator class; the two vLLM kernels are the restated stubs of oracle/ref_shims/vllm, as for
    tests/golden/fp8_mm.safetensors) against the oracle inside `fp8_blocks()`: the CFG noise prediction of two steps, bit for bit — pins WHICH
    layers are w8a8 (the blocks' ten linears, not pre-/post-infer) and how the quantised operator composes with the rest of the graph."""
    from oracle import ref_import

    if not ref_import.reference_available():
        pytest.skip("reference checkout not present (authoring container only)")
    ref_import.patch_and_import()
    from lightx2v.models.schedulers.wan.scheduler import WanScheduler as RefScheduler

    from lightx2v_amd import synth
    from oracle import wan_oracle as O
    from oracle.gen_golden import _ref_model_infer

    dims = dict(synth.WAN_DIMS["wan-tiny"], num_layers=2)
    ts = (16, 2, 8, 12)
    wd = synth.synth_wan_weights(dims, seed=13)
    lat, ctx, ctx_null = synth.synth_inputs(dims, ts)
    cfg = ref_import.make_config(dims, target_shape=ts, target_video_length=5, infer_steps=3,
                                 mm_config={"mm_type": "W-fp8-channel-sym-A-fp8-channel-sym-dynamic-Vllm", "weight_auto_quant": True})
    R = ref_import.build_reference_wan(cfg, wd)
    sch = RefScheduler(cfg)
    sch.device = torch.device("cpu")
    sch.prepare()
    sch.latents = lat.clone()
    for m in ("pre", "post"):
        R[m].set_scheduler(sch)
    inputs = {"text_encoder_output": {"context": ctx, "context_null": ctx_null}}
    for i in range(2):
        sch.step_pre(i)
        _ref_model_infer(R, sch, cfg, inputs)
        with O.fp8_blocks():
            mine = O.wan_model_infer(wd, dims, sch.latents, sch.timesteps[i], ctx, ctx_null, 6.0)
        assert torch.equal(mine, sch.noise_pred), f"step {i}: max |d| = {(mine - sch.noise_pred).abs().max().item():.3e}"
        assert not torch.equal(mine, O.wan_model_infer(wd, dims, sch.latents, sch.timesteps[i], ctx, ctx_null, 6.0)), "the w8a8 graph must differ from the bf16 one"
        sch.step_post()
