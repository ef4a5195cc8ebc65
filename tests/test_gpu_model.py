"""Model-level parity of the HIP Wan path against fixtures generated from the reference (wan-tiny, full
4-step CFG denoise loop) and against the CPU oracle at Wan2.1-1.3B dimensions (BASELINE config #1 shapes).

End-to-end tolerances are relative-L2 (errors of ~30 chained bf16 ops do not stay within one ulp):
  pre-infer tensors 1e-2; one block 1e-2; wan-tiny forward 2e-2; 4-step loop 3e-2; 1.3B 2-block forward 2e-2.
With config key `hip_ref_rounding` the norm kernels reproduce the reference's bf16 chain (tighter legs).
"""
import pytest
import torch

from tests.util import assert_rel, record, rel_l2

pytestmark = pytest.mark.gpu


def _to_dev(wd):
    return {k: v.cuda() for k, v in wd.items()}


@pytest.fixture(scope="module")
def tiny():
    from lightx2v_amd import synth, wan, scheduler

    dims = synth.WAN_DIMS["wan-tiny"]
    wl = synth.WORKLOADS["wan-tiny"]
    wd = synth.synth_wan_weights(dims, seed=0)
    cfg = wan.default_config(dims, target_shape=wl["target_shape"], target_video_length=wl["frames"], infer_steps=4, hip_ref_rounding=True)
    model = wan.WanModel(cfg, _to_dev(wd))
    lat, ctx, ctx_null = synth.synth_inputs(dims, wl["target_shape"])
    sch = scheduler.WanScheduler(cfg, device="cuda")
    sch.prepare(latents=lat)
    model.set_scheduler(sch)
    inputs = {"text_encoder_output": {"context": [c.cuda() for c in ctx], "context_null": [c.cuda() for c in ctx_null]}}
    return model, sch, inputs, wd


def test_tiny_pre_infer_and_block(tiny, golden_model):
    model, sch, inputs, wd = tiny
    g = golden_model
    sch.prepare(latents=g["latents0"])
    sch.step_pre(0)
    assert torch.equal(sch.timesteps.cpu(), g["timesteps"])
    embed, grid_sizes, (x, embed0, seq_lens, freqs, context) = model.pre_infer.infer(model.pre_weight, inputs, positive=True)
    assert_rel(x, g["pre_x"], 1e-2, "patch embedding")
    assert_rel(embed, g["pre_embed"], 1e-2, "time embedding")
    assert_rel(embed0, g["pre_embed0"], 1e-2, "time projection")
    assert_rel(context, g["pre_context"], 1e-2, "text embedding")
    # one block from the REFERENCE's inputs so only the block's own error is measured
    tr = model.transformer_infer
    blk = model.transformer_weights.blocks[0]
    xb = g["pre_x"].cuda().clone()
    e0, ctx_ref = g["pre_embed0"].cuda(), g["pre_context"].cuda()
    mods = tr.infer_modulation(blk.compute_phases[0], e0)
    xb = tr.infer_self_attn(blk.compute_phases[1], grid_sizes, xb, seq_lens, freqs, mods[0], mods[1], mods[2])
    assert_rel(xb, g["b0_x_after_self"], 1e-2, "x after self-attention")
    xb = tr.infer_cross_attn(blk.compute_phases[2], xb, ctx_ref)
    assert_rel(xb, g["b0_x_after_cross"], 1e-2, "x after cross-attention")
    xb = tr.infer_ffn(blk.compute_phases[3], xb, mods[3], mods[4], mods[5])
    assert_rel(xb, g["b0_x_out"], 1e-2, "block output")


def test_tiny_forward_and_denoise_loop(tiny, golden_model):
    from lightx2v_amd.scheduler import run_denoise_loop

    model, sch, inputs, wd = tiny
    g = golden_model
    sch.prepare(latents=g["latents0"])
    sch.step_pre(0)
    cond = model._forward(inputs, True)
    assert_rel(cond, g["step0_cond"], 2e-2, "conditional forward")
    sch.prepare(latents=g["latents0"])
    errs = []
    run_denoise_loop(model, sch, inputs, step_callback=lambda i: errs.append(rel_l2(sch.latents, g[f"latents_after_step{i}"])))
    assert max(errs) <= 3e-2, errs
    assert sch.latents.dtype == torch.float32 and torch.isfinite(sch.latents).all()


def test_wan13b_two_blocks_vs_oracle():
    """Wan2.1-1.3B dims (D 1536, F 8960, 12 heads), BASELINE config #1 token count (256x256x17f → S = 1280),
    2 of 30 layers so the CPU oracle finishes in seconds; both rounding modes."""
    from lightx2v_amd import synth, wan, scheduler
    from oracle import wan_oracle as O

    dims = dict(synth.WAN_DIMS["wan2.1-1.3b"], num_layers=2)
    ts = synth.WORKLOADS["wan1.3b_256x256x17f"]["target_shape"]
    wd = synth.synth_wan_weights(dims, seed=5)
    lat, ctx, ctx_null = synth.synth_inputs(dims, ts)
    t = torch.tensor(888)
    ref = O.wan_forward(wd, dims, lat.to(torch.bfloat16), t, ctx)
    for ref_rounding in (False, True):
        cfg = wan.default_config(dims, target_shape=ts, target_video_length=17, infer_steps=4, hip_ref_rounding=ref_rounding)
        model = wan.WanModel(cfg, _to_dev(wd))
        sch = scheduler.WanScheduler(cfg, device="cuda")
        sch.prepare(latents=lat)
        sch.timesteps[2] = 888
        model.set_scheduler(sch)
        sch.step_pre(2)
        inputs = {"text_encoder_output": {"context": [c.cuda() for c in ctx], "context_null": [c.cuda() for c in ctx_null]}}
        got = model._forward(inputs, True)
        assert got.shape == ref.shape == (16, 5, 32, 32)
        assert_rel(got, ref, 2e-2, f"1.3B 2-block forward (ref_rounding={ref_rounding})")


def test_operator_api_drop_in():
    """The registered operator objects follow the reference's apply() contracts."""
    from lightx2v_amd import registry, ops  # noqa: F401
    from oracle import wan_oracle as O

    gen = torch.Generator().manual_seed(0)
    wd = {"w": (torch.randn(256, 128, generator=gen) * 0.1).to(torch.bfloat16), "b": torch.randn(256, generator=gen).to(torch.bfloat16), "n": torch.ones(256, dtype=torch.bfloat16)}
    x = torch.randn(100, 128, generator=gen).to(torch.bfloat16)
    mm = registry.MM_WEIGHT_REGISTER["Hip-bf16"]("w", "b")
    mm.load({k: v.cuda() for k, v in wd.items()})
    y = mm.apply(x.cuda())
    assert y.shape == (100, 256) and y.dtype == torch.bfloat16
    assert_rel(y, O.mm(x, wd["w"], wd["b"]), 5e-3, "MMWeightHip.apply")
    rms = registry.RMS_WEIGHT_REGISTER["hip"]("n")
    rms.load({k: v.cuda() for k, v in wd.items()})
    assert_rel(rms.apply(y), O.rms_norm_fp32(y.cpu(), wd["n"]), 5e-3, "RMSWeightHip.apply")
    ln = registry.LN_WEIGHT_REGISTER["hip"]()
    ln.load({})
    assert_rel(ln.apply(y), O.layer_norm(y.cpu()), 5e-3, "LNWeightHip.apply")
    attn = registry.ATTN_WEIGHT_REGISTER["hip_flash"]()
    q = torch.randn(70, 2, 128, generator=gen).to(torch.bfloat16)
    o = attn.apply(q.cuda(), q.cuda(), q.cuda(), cu_seqlens_q=torch.tensor([0, 70]), cu_seqlens_kv=torch.tensor([0, 70]), max_seqlen_q=70, max_seqlen_kv=70)
    assert o.shape == (70, 256)
    assert_rel(o, O.sdpa(q, q, q), 1e-2, "HipFlashAttnWeight.apply")
    mm.to_cpu()
    assert mm.weight.device.type == "cpu"
    mm.to_cuda()
    assert mm.weight.is_cuda


def test_teacache_matches_reference_decisions_and_latents():
    """feature_caching="Tea" on the HIP path vs the fixture generated from the reference's TeaCache class: identical
    calc/skip pattern in both CFG branches (the decision input is the tiny time-embedding tensor) and latents within
    the denoise-loop tolerance after all 16 steps."""
    import os

    from safetensors.torch import load_file

    from lightx2v_amd import scheduler, synth, wan
    from lightx2v_amd.scheduler import run_denoise_loop

    g = load_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wan-tiny_teacache.safetensors"))
    coeffs = [[0, 0, 0, 1.0, 0], [0, 0, 0.5, 1.0, 0]]  # oracle/gen_golden.py::TEA_TEST_COEFFS
    dims = synth.WAN_DIMS["wan-tiny"]
    wl = synth.WORKLOADS["wan-tiny"]
    wd = _to_dev(synth.synth_wan_weights(dims, seed=0))
    _, ctx, ctx_null = synth.synth_inputs(dims, wl["target_shape"])
    inputs = {"text_encoder_output": {"context": [c.cuda() for c in ctx], "context_null": [c.cuda() for c in ctx_null]}}
    steps = int(g["steps"])
    for tag, use_ret in (("ret", True), ("noret", False)):
        cfg = wan.default_config(dims, target_shape=wl["target_shape"], target_video_length=wl["frames"], infer_steps=steps, hip_ref_rounding=True,
                                 feature_caching="Tea", coefficients=coeffs, use_ret_steps=use_ret, teacache_thresh=float(g["thresh"]))
        model = wan.WanModel(cfg, wd)
        sch = scheduler.WanScheduler(cfg, device="cuda")
        sch.prepare(latents=g["latents0"])
        model.set_scheduler(sch)
        errs = []
        run_denoise_loop(model, sch, inputs, step_callback=lambda i: errs.append(rel_l2(sch.latents, g[f"{tag}_latents_after_step{i}"])))
        assert [int(c) for c in sch.caching_records] == g[f"{tag}_records_cond"].tolist(), tag
        assert [int(c) for c in sch.caching_records_2] == g[f"{tag}_records_uncond"].tolist(), tag
        assert max(errs) <= 5e-2, (tag, errs)


def test_quantized_checkpoint_roundtrip_forward(tmp_path):
    """a bf16 checkpoint directory converted to the reference's e4m3 per-block layout, read back through the
    config-driven loader and run through the fp8 operator class gives (a) bit-identical results to the same quantised tensors handed
    over in memory, (b) a forward within fp8 tolerance of the bf16 model's."""
    from safetensors.torch import save_file

    from lightx2v_amd import checkpoint as ck
    from lightx2v_amd import scheduler, synth, wan

    dims = synth.WAN_DIMS["wan-tiny"]
    wl = synth.WORKLOADS["wan-tiny"]
    wd = synth.synth_wan_weights(dims, seed=0)
    src, out = tmp_path / "bf16", tmp_path / "bf16" / "fp8"
    src.mkdir()
    save_file({k: v.contiguous() for k, v in wd.items()}, str(src / "model.safetensors"))
    ck.convert_checkpoint(str(src), str(out), model_type="wan_dit", quantized=True, linear_dtype=torch.float8_e4m3fn, non_linear_dtype=torch.bfloat16,
                          save_by_block=True)
    mm = {"mm_type": "W-fp8-channel-sym-A-fp8-channel-sym-dynamic-Hip"}
    cfg = wan.default_config(dims, target_shape=wl["target_shape"], target_video_length=wl["frames"], infer_steps=4, mm_config=mm)
    loaded = ck.load_for_config(str(src), cfg, device="cuda")  # default dit_quantized_ckpt = <model_path>/fp8 (model.py:38-40)
    assert loaded["blocks.0.ffn.0.weight"].dtype == torch.float8_e4m3fn and loaded["blocks.0.ffn.0.weight_scale"].shape == (dims["ffn_dim"], 1)
    in_mem = ck.quantize_model({k: v.clone() for k, v in wd.items()}, **{k: ck.MODEL_TYPE_KEYS["wan_dit"][k] for k in ("target_keys", "key_idx", "ignore_key")},
                               non_linear_dtype=torch.bfloat16)
    lat, ctx, ctx_null = synth.synth_inputs(dims, wl["target_shape"])
    inputs = {"text_encoder_output": {"context": [c.cuda() for c in ctx], "context_null": [c.cuda() for c in ctx_null]}}

    def forward(cfg_, weights):
        model = wan.WanModel(cfg_, weights)
        sch = scheduler.WanScheduler(cfg_, device="cuda")
        sch.prepare(latents=lat)
        model.set_scheduler(sch)
        sch.step_pre(0)
        model.infer(inputs)
        return sch.noise_pred.float().cpu()

    a = forward(cfg, loaded)
    b = forward(cfg, _to_dev(in_mem))
    assert torch.equal(a, b), "checkpoint round trip changed the quantised tensors"
    ref = forward(wan.default_config(dims, target_shape=wl["target_shape"], target_video_length=wl["frames"], infer_steps=4), _to_dev(wd))
    assert_rel(a, ref, 1e-1, "fp8 checkpoint forward vs bf16 forward")  # w8a8 on a random 2-block model: 6.4e-2 measured
    assert rel_l2(a, ref) > 1e-4  # it really is the quantised path


def test_cross_kv_cache_is_transparent():
    """reusing the step-invariant text-MLP output and per-block cross-attention K/V gives bit-identical latents over a
    denoise loop, really skips the recomputation, and follows a changed prompt."""
    from lightx2v_amd import scheduler, synth, wan

    dims, wl = synth.WAN_DIMS["wan-tiny"], synth.WORKLOADS["wan-tiny"]
    wd = _to_dev(synth.synth_wan_weights(dims, seed=0))
    lat, ctx, ctx_null = synth.synth_inputs(dims, wl["target_shape"])
    inputs = {"text_encoder_output": {"context": [c.cuda() for c in ctx], "context_null": [c.cuda() for c in ctx_null]}}

    def run(cache, inp):
        cfg = wan.default_config(dims, target_shape=wl["target_shape"], target_video_length=wl["frames"], infer_steps=3, cache_cross_kv=cache)
        model = wan.WanModel(cfg, wd)
        sch = scheduler.WanScheduler(cfg, device="cuda")
        sch.prepare(latents=lat)
        model.set_scheduler(sch)
        scheduler.run_denoise_loop(model, sch, inp)
        return model, sch.latents.float().cpu()

    m_on, on = run(True, inputs)
    m_off, off = run(False, inputs)
    assert torch.equal(on, off)
    tr = m_on.transformer_infer
    assert len(tr._cross_kv_cache) == 2 and all(len(e["kv"]) == dims["num_layers"] for e in tr._cross_kv_cache.values())
    assert len(m_on.pre_infer._text_cache) == 2 and not m_off.transformer_infer._cross_kv_cache and not m_off.pre_infer._text_cache
    # the cached K really is what a fresh computation gives
    entry = next(iter(tr._cross_kv_cache.values()))
    blk = m_on.transformer_weights.blocks[0].compute_phases[2] if hasattr(m_on.transformer_weights.blocks[0], "compute_phases") else None
    if blk is not None and id(blk) in entry["kv"]:
        k_cached = entry["kv"][id(blk)][0]
        tr.cache_cross_kv = False
        k_fresh, _, _ = tr._cross_kv(blk, entry["ctx"])
        tr.cache_cross_kv = True
        assert torch.equal(k_cached, k_fresh)
    # a different prompt (new tensors) is not served from the cache; an in-place edit of the same tensor is noticed too
    inputs2 = {"text_encoder_output": {"context": [c.cuda() * 0.5 for c in ctx], "context_null": [c.cuda() for c in ctx_null]}}
    _, other = run(True, inputs2)
    assert not torch.equal(other, on)
    cfg = wan.default_config(dims, target_shape=wl["target_shape"], target_video_length=wl["frames"], infer_steps=3)
    model = wan.WanModel(cfg, wd)
    sch = scheduler.WanScheduler(cfg, device="cuda")
    sch.prepare(latents=lat)
    model.set_scheduler(sch)
    mutable = {"text_encoder_output": {"context": [c.cuda().clone() for c in ctx], "context_null": [c.cuda() for c in ctx_null]}}
    sch.step_pre(0)
    model.infer(mutable)
    first = sch.noise_pred.clone()
    mutable["text_encoder_output"]["context"][0].mul_(0.5)
    model.infer(mutable)
    assert not torch.equal(sch.noise_pred, first)


def test_baseline_config1_full_run_vs_oracle():
    """BASELINE config #1 in full — Wan2.1-T2V-1.3B (30 layers), 256x256x17f (S = 1280), 4 steps, CFG, shift 8, guide 6 — the whole denoise
    loop on the HIP path against the CPU oracle on identical noise / text embeddings / weights.  The oracle's 240 block evaluations take
    tens of seconds on the box's host cores.  Tolerance: relative L2 of the final latents <= 5e-2 (bf16 chains of 30 blocks x 8
    forwards; the per-forward error measured by the 2-block test is ~1e-2), and the first step's noise prediction <= 3e-2."""
    from lightx2v_amd import scheduler, synth, wan
    from oracle import wan_oracle as O

    dims = synth.WAN_DIMS["wan2.1-1.3b"]
    wl = synth.WORKLOADS["wan1.3b_256x256x17f"]
    ts = wl["target_shape"]
    wd = synth.synth_wan_weights(dims, seed=0)
    lat, ctx, ctx_null = synth.synth_inputs(dims, ts)
    steps, shift, guide = 4, 8.0, 6.0
    ref_steps = []
    ref = O.denoise_loop(wd, dims, lat, ctx, ctx_null, steps, shift, guide, step_callback=lambda i, x: ref_steps.append(x.clone()))
    cfg = wan.default_config(dims, target_shape=ts, target_video_length=wl["frames"], infer_steps=steps, sample_shift=shift, sample_guide_scale=guide)
    model = wan.WanModel(cfg, _to_dev(wd))
    sch = scheduler.WanScheduler(cfg, device="cuda")
    sch.prepare(latents=lat)
    model.set_scheduler(sch)
    inputs = {"text_encoder_output": {"context": [c.cuda() for c in ctx], "context_null": [c.cuda() for c in ctx_null]}}
    got_steps = []
    scheduler.run_denoise_loop(model, sch, inputs, step_callback=lambda i: got_steps.append(sch.latents.float().cpu().clone()))
    assert len(got_steps) == len(ref_steps) == steps
    assert_rel(got_steps[0], ref_steps[0], 3e-2, "config #1: latents after step 1")
    assert_rel(got_steps[-1], ref, 5e-2, "config #1: final latents after 4 CFG steps")
    # the tolerances above anchored: the SAME loop evaluated in fp32 with no bf16 rounding points is the truth; the HIP
    # path may be at most 1.5 x as far from it as the reference's own bf16 CPU path is (both are bf16 realisations of that graph)
    tru_steps = []
    with O.truth_precision(torch.float32, device="cuda"):  # the oracle's statements in fp32 through plain PyTorch on the GPU (the scheduler stays on the host)
        O.denoise_loop(O.upcast(wd, device="cuda"), dims, lat, O.upcast(ctx, device="cuda"), O.upcast(ctx_null, device="cuda"), steps, shift, guide,
                       step_callback=lambda i, x: tru_steps.append(x.clone()))
    for i in (0, steps - 1):
        e_hip, e_ref = rel_l2(got_steps[i], tru_steps[i]), rel_l2(ref_steps[i], tru_steps[i])
        record(f"config #1 loop, latents after step {i + 1}", err_hip_vs_fp32=e_hip, err_oracle_vs_fp32=e_ref, hip_vs_oracle=rel_l2(got_steps[i], ref_steps[i]))
        assert e_hip <= 1.5 * e_ref + 1e-3, f"step {i + 1}: HIP is {e_hip:.3e} from the fp32 truth, the bf16 oracle {e_ref:.3e}"


def test_wan13b_block_and_forward_errors_are_anchored_to_fp32_truth():
    """Block and forward tolerances anchored to an fp32 evaluation of the same graph (SURVEY §7 "fp32-reference triangle"): Wan2.1-1.3B width,
    config #1's 1280 tokens; one block from the oracle's pre-infer tensors, and the whole 30-layer conditional forward.  For each:
    err(HIP vs truth) <= 1.5 x err(bf16 oracle vs truth); the measured numbers go to gpurun_out/parity_summary.jsonl."""
    from lightx2v_amd import scheduler, synth, wan
    from oracle import wan_oracle as O

    dims = synth.WAN_DIMS["wan2.1-1.3b"]
    ts = synth.WORKLOADS["wan1.3b_256x256x17f"]["target_shape"]
    wd = synth.synth_wan_weights(dims, seed=7)
    lat, ctx, ctx_null = synth.synth_inputs(dims, ts)
    t = torch.tensor(640)
    latb = lat.to(torch.bfloat16)
    wd32, ctx32 = O.upcast(wd, device="cuda"), O.upcast(ctx, device="cuda")
    embed_o, grid, x_o, embed0_o, _, context_o = O.wan_pre_infer(wd, dims, latb, t, ctx)
    freqs = O.rope_freqs_table(128)
    ref_blk = O.wan_block(wd, 0, dims, grid, x_o.clone(), embed0_o, freqs, context_o)
    ref_fwd = O.wan_forward(wd, dims, latb, t, ctx)
    with O.truth_precision(torch.float32, device="cuda"):  # plain PyTorch fp32 on the GPU
        tru_blk = O.wan_block(wd32, 0, dims, grid, x_o.float().cuda(), embed0_o.float().cuda(), freqs.cuda(), context_o.float().cuda()).cpu()
        tru_fwd = O.wan_forward(wd32, dims, latb.float().cuda(), t, ctx32).cpu()
    del wd32
    cfg = wan.default_config(dims, target_shape=ts, target_video_length=17, infer_steps=4)
    model = wan.WanModel(cfg, _to_dev(wd))
    sch = scheduler.WanScheduler(cfg, device="cuda")
    sch.prepare(latents=lat)
    sch.timesteps[1] = 640
    model.set_scheduler(sch)
    sch.step_pre(1)
    inputs = {"text_encoder_output": {"context": [c.cuda() for c in ctx], "context_null": [c.cuda() for c in ctx_null]}}
    embed, grid_sizes, (x, embed0, seq_lens, rope, context) = model.pre_infer.infer(model.pre_weight, inputs, positive=True)
    got_blk = model.transformer_infer.infer_block(model.transformer_weights.blocks[0], grid_sizes, embed, x_o.cuda().clone(), embed0_o.cuda(), seq_lens, rope, context_o.cuda())
    got_fwd = model._forward(inputs, True)
    for name, got, ref, tru in (("1.3B block S=1280", got_blk, ref_blk, tru_blk), ("1.3B 30-layer forward S=1280", got_fwd, ref_fwd, tru_fwd)):
        e_hip, e_ref = rel_l2(got, tru), rel_l2(ref, tru)
        record(name, err_hip_vs_fp32=e_hip, err_oracle_vs_fp32=e_ref, hip_vs_oracle=rel_l2(got, ref))
        assert e_hip <= 1.5 * e_ref + 2e-4, f"{name}: HIP is {e_hip:.3e} from the fp32 truth, the bf16 oracle {e_ref:.3e}"


@pytest.mark.parametrize("fp8", [False, True])
def test_cfg_pair_pass_is_bit_identical_to_separate_forwards(fp8):
    """WanModel.infer runs the conditional and unconditional forwards of a CFG step as ONE pass over [cond tokens | uncond tokens]
    (config `cfg_pair`, default on; wan/model.py:197-226 runs them one after the other): every output row is computed from the same operands
    in the same order, so noise predictions and latents over a denoise loop must be EQUAL to the separate-forward path — token counts that
    are / are not multiples of 64 (padding rows inside the stacked buffer), bf16 and fp8 operators."""
    from lightx2v_amd import scheduler, synth, wan

    dims = synth.WAN_DIMS["wan-tiny"]
    wd = _to_dev(synth.synth_wan_weights(dims, seed=1))
    extra = {"mm_config": {"mm_type": "W-fp8-channel-sym-A-fp8-channel-sym-dynamic-Hip", "weight_auto_quant": True}} if fp8 else {}
    for ts, frames in (((16, 3, 12, 10), 9), ((16, 2, 16, 16), 5)):  # 90 tokens (pad to 128) / 128 tokens
        lat, ctx, ctx_null = synth.synth_inputs(dims, ts)
        inputs = {"text_encoder_output": {"context": [c.cuda() for c in ctx], "context_null": [c.cuda() for c in ctx_null]}}
        outs = {}
        # True / False: the pair pass forced on / off; "streams": the two forwards block by block on two compute streams (wan.CfgBranchStreams)
        for pair in (True, False, "streams"):
            cfg = wan.default_config(dims, target_shape=ts, target_video_length=frames, infer_steps=3, cfg_pair=pair is True, cfg_branch_streams=pair == "streams", **extra)
            model = wan.WanModel(cfg, wd)
            sch = scheduler.WanScheduler(cfg, device="cuda")
            sch.prepare(latents=lat)
            model.set_scheduler(sch)
            assert model._pair_ok(inputs) == (pair is True)  # forced either way here; the default ("auto") decides by size
            sch.step_pre(0)
            model.infer(inputs)
            pred = sch.noise_pred.float().clone()
            sch.step_post()
            for i in (1, 2):
                sch.step_pre(i)
                model.infer(inputs)
                sch.step_post()
            outs[pair] = (pred, sch.latents.float().clone())
        assert torch.isfinite(outs[True][1]).all()
        assert torch.equal(outs[True][0], outs[False][0]), f"noise prediction differs: max |d| = {(outs[True][0] - outs[False][0]).abs().max().item():.3e}"
        assert torch.equal(outs[True][1], outs[False][1]), "latents after 3 steps differ"
        assert torch.equal(outs["streams"][0], outs[False][0]) and torch.equal(outs["streams"][1], outs[False][1]), "two-stream CFG branches changed the result"
        assert model._cfg_interleave._streams is not None, "the two-stream path was not taken"


@pytest.mark.parametrize("ref_rounding", [True, False])
def test_i2v_branch_vs_reference_fixture(ref_rounding):
    """The i2v branch (pre_infer.py:44-55,100-113; transformer_infer.py:405-455 — the configuration the reference publishes its numbers for) on the HIP
    path against the fixture the reference's own i2v objects produced (tests/golden/wan-tiny-i2v_model.safetensors, oracle/gen_golden.py::gen_model_i2v):
    pre-infer (36-channel patch embedding = a K = 144 GEMM zero-padded to 192, the CLIP-feature MLP with its exact GELU in front of the text context),
    block 0 from the reference's inputs (second cross-attention over the 257 image tokens, added in bf16), the conditional forward and the 3-step CFG
    loop; then the three CFG forms (pair pass / two streams / sequential) must stay bit-identical with the image K / V caches in play."""
    import os

    from safetensors.torch import load_file

    from lightx2v_amd import scheduler, synth, wan
    from lightx2v_amd.scheduler import run_denoise_loop

    g = load_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wan-tiny-i2v_model.safetensors"))
    dims = synth.WAN_DIMS["wan-tiny-i2v"]
    ts, frames = (16, 3, 8, 8), 9
    wd = _to_dev(synth.synth_wan_i2v_weights(dims, seed=0))
    lat, ctx, ctx_null = synth.synth_inputs(dims, ts)
    image = {k: v.cuda() for k, v in synth.synth_i2v_inputs(dims, ts).items()}
    inputs = {"text_encoder_output": {"context": [c.cuda() for c in ctx], "context_null": [c.cuda() for c in ctx_null]}, "image_encoder_output": image}

    def build(**kw):
        cfg = wan.default_config(dims, task="i2v", in_dim=36, cross_attn_2_type="hip_flash", target_shape=ts, target_video_length=frames, infer_steps=3,
                                 hip_ref_rounding=ref_rounding, **kw)
        model = wan.WanModel(cfg, wd)
        sch = scheduler.WanScheduler(cfg, device="cuda")
        sch.prepare(latents=lat)
        model.set_scheduler(sch)
        return model, sch

    model, sch = build()
    sch.step_pre(0)
    embed, grid_sizes, (x, embed0, seq_lens, freqs, context) = model.pre_infer.infer(model.pre_weight, inputs, positive=True)
    assert context.shape[0] == 257 + dims["text_len"]
    assert_rel(x, g["pre_x"], 1e-2, "i2v patch embedding (36 channels)")
    assert_rel(context[:257], g["pre_context"][:257], 1e-2, "CLIP-feature MLP")
    assert_rel(context[257:], g["pre_context"][257:], 1e-2, "text embedding")
    assert model.pre_infer.infer(model.pre_weight, inputs, positive=True)[2][4] is context, "the context object must be reused across steps (cross K/V caches key on it)"
    tr, blk = model.transformer_infer, model.transformer_weights.blocks[0]
    xb, e0, ctx_ref = g["pre_x"].cuda().clone(), g["pre_embed0"].cuda(), g["pre_context"].cuda()
    mods = tr.infer_modulation(blk.compute_phases[0], e0)
    xb = tr.infer_self_attn(blk.compute_phases[1], grid_sizes, xb, seq_lens, freqs, mods[0], mods[1], mods[2])
    assert_rel(xb, g["b0_x_after_self"], 1e-2, "x after self-attention")
    xb = tr.infer_cross_attn(blk.compute_phases[2], xb, ctx_ref)
    xb = tr.infer_ffn(blk.compute_phases[3], xb, mods[3], mods[4], mods[5])
    assert_rel(xb, g["b0_x_out"], 1e-2, "i2v block output")
    assert_rel(model._forward(inputs, True), g["step0_cond"], 2e-2, "i2v conditional forward")
    outs = {}
    for form, kw in (("pair", dict(cfg_pair=True)), ("streams", dict(cfg_pair=False, cfg_branch_streams=True)), ("sequential", dict(cfg_pair=False, cfg_branch_streams=False))):
        if ref_rounding and form == "pair":
            continue  # the pair pass needs the fp32-statistics mode
        m, s = build(**kw)
        errs = []
        run_denoise_loop(m, s, inputs, step_callback=lambda i: errs.append(rel_l2(s.latents, g[f"latents_after_step{i}"])))
        assert max(errs) <= 3e-2, (form, errs)
        outs[form] = s.latents.clone()
        record(f"i2v 3-step CFG loop ({form}, ref_rounding={ref_rounding})", worst_rel_l2_vs_reference_fixture=max(errs))
    ref_form = outs["sequential"]
    for form, o in outs.items():
        assert torch.equal(o, ref_form), f"CFG form {form} changed the i2v result"
