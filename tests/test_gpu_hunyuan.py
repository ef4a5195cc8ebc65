"""HunyuanVideo DiT on the HIP path (lightx2v_amd/hunyuan.py) against the fixture generated from the reference
(tests/golden/hunyuan_tiny.safetensors) and, for padded text (two attention segments), against the CPU oracle.
Tolerances are relative L2 like the Wan model tests: pre-infer 1e-2, one block 1e-2, forward 2e-2 (bf16 chains)."""
import os

import pytest
import torch

from tests.util import assert_rel

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def setup():
    from safetensors.torch import load_file

    from lightx2v_amd import hunyuan as hy, synth

    g = load_file(os.path.join(GOLDEN, "hunyuan_tiny.safetensors"))
    dims = synth.HUNYUAN_DIMS["hunyuan-tiny"]
    wd = synth.synth_hunyuan_weights(dims, seed=int(g["seed"]))
    cfg = hy.default_config(dims, infer_steps=4, hip_ref_rounding=True)
    model = hy.HunyuanModel(cfg, {k: v.cuda() for k, v in wd.items()})
    sch = hy.HunyuanScheduler(cfg)
    sch.prepare(g["latents"])
    model.set_scheduler(sch)
    return model, sch, g, wd, dims


def _inputs(g):
    return {"text_encoder_output": {"text_encoder_1_text_states": g["text_states"].cuda(), "text_encoder_1_attention_mask": g["text_mask"].cuda(),
                                    "text_encoder_2_text_states": g["text_states_2"].cuda()}}


def test_headnorm_rope_matches_reference_chain():
    from lightx2v_amd import lib
    from oracle import hunyuan_oracle as H
    from oracle.wan_oracle import rms_norm

    gen = torch.Generator().manual_seed(3)
    L, Hh, n_rope = 37, 3, 29
    qkv = (torch.randn(L, 3 * Hh * 128, generator=gen) * 1.5).to(torch.bfloat16)
    wq, wk = (1 + 0.1 * torch.randn(128, generator=gen)).to(torch.bfloat16), (1 + 0.1 * torch.randn(128, generator=gen)).to(torch.bfloat16)
    cos, sin = H.rope_tables([1, 1, n_rope])
    D = Hh * 128
    q, k = qkv[:, :D].reshape(L, Hh, 128), qkv[:, D : 2 * D].reshape(L, Hh, 128)
    qr, kr = rms_norm(q, wq), rms_norm(k, wk)
    q2, k2 = H.apply_rotary_emb(qr[:n_rope], kr[:n_rope], cos, sin)
    qr, kr = torch.cat((q2, qr[n_rope:]), 0), torch.cat((k2, kr[n_rope:]), 0)
    d = qkv.cuda()
    lib.headnorm_rope_(d[:, :D], d[:, D : 2 * D], wq.cuda(), wk.cuda(), cos.cuda(), sin.cuda(), Hh, n_rope, 1e-6, lib.ROUND_REF)
    for got, ref, nm in ((d[:, :D], qr, "q"), (d[:, D : 2 * D], kr, "k")):
        diff = (got.float().cpu() - ref.reshape(L, D).float()).abs()
        # bit-exact except rows whose rstd lands on the other side of a bf16 rounding boundary: torch's CPU bf16 rsqrt is
        # not the correctly rounded 1/sqrt (0.03 % - 4 % of inputs differ by one bf16 ulp depending on its code path), so
        # a few whole (token, head) rows may start the rotary step one ulp away (its sum of two rounded products then
        # moves by at most two); nothing may be further than that
        ulp = ref.reshape(L, D).float().abs() * 2.0 ** -7 + 1e-30
        assert (diff > 0).float().mean() < 5e-2 and bool((diff <= 2 * ulp).all()), (nm, diff.max(), (diff > 0).float().mean())
    assert torch.equal(d[:, 2 * D :].cpu(), qkv[:, 2 * D :])  # v untouched


def test_scheduler_and_pre_infer(setup):
    model, sch, g, _, _ = setup
    assert torch.equal(sch.timesteps.cpu(), g["sched_timesteps"]) and torch.equal(sch.sigmas, g["sched_sigmas"])
    assert torch.equal(sch.freqs_cos.cpu(), g["freqs_cos"]) and torch.equal(sch.freqs_sin.cpu(), g["freqs_sin"])
    sch.step_pre(1)
    img, txt, vec, cu, max_len, _ = model.pre_infer.infer(model.pre_weight, _inputs(g))
    assert cu.tolist() == g["cu_seqlens"].tolist() and max_len == int(g["max_seqlen"])
    assert_rel(img, g["pre_img"], 1e-2, "img_in")
    assert_rel(vec, g["pre_vec"], 1e-2, "vec")
    assert_rel(txt, g["pre_txt"], 1e-2, "token refiner output")


def test_blocks_and_forward(setup):
    model, sch, g, _, dims = setup
    tr = model.transformer_infer
    freqs = (sch.freqs_cos, sch.freqs_sin)
    img, txt, vec = g["pre_img"].cuda(), g["pre_txt"].cuda(), g["pre_vec"].cuda()
    n_img = img.shape[0]
    # one double block and one single block from the REFERENCE's inputs
    import copy

    cfg1 = dict(tr.config, double_blocks_num=1, single_blocks_num=0)
    tr1 = type(tr)(cfg1)
    w1 = copy.copy(model.transformer_weights)
    out, _ = tr1.infer(w1, img, txt, vec, g["cu_seqlens"], int(g["max_seqlen"]), freqs)
    assert_rel(out, g["d0_img"], 1e-2, "double block 0 (img)")
    x = g["s0_in"].cuda()
    tr1._segs = [(0, x.shape[0])]
    ws = tr1._workspace(x)
    from lightx2v_amd import lib

    tr1.infer_single_block(model.transformer_weights.single_blocks[0], x, txt.shape[0], lib.activation(vec, 3), freqs, ws)
    assert_rel(x, g["s0_out"], 1e-2, "single block 0")
    out, _ = tr.infer(model.transformer_weights, img, txt, vec, g["cu_seqlens"], int(g["max_seqlen"]), freqs)
    assert_rel(out, g["tr_img"], 2e-2, "block stack")
    sch.step_pre(1)
    model.infer(_inputs(g))
    assert sch.noise_pred.dtype == torch.float32 and sch.noise_pred.shape == g["noise_pred"].shape
    assert_rel(sch.noise_pred, g["noise_pred"], 2e-2, "HunyuanModel.infer")
    lat0 = sch.latents.clone()
    sch.step_post()
    from oracle import hunyuan_oracle as H

    assert_rel(sch.latents, H.euler_step(lat0.cpu(), g["noise_pred"], g["sched_sigmas"], 1), 2e-2, "Euler step")


def test_padded_text_two_segments_vs_oracle(setup):
    """10 of 16 text tokens valid → cu_seqlens [0, S+10, S+16]: two attention segments and the masked token refiner."""
    from lightx2v_amd import synth
    from oracle import hunyuan_oracle as H

    model, sch, g, wd, dims = setup
    ts = synth.HUNYUAN_WORKLOADS["hunyuan-tiny"]["target_shape"]
    lat, text_states, mask, ts2 = synth.synth_hunyuan_inputs(dims, ts, seed=9, valid_text=10)
    sch.prepare(lat)
    sch.step_pre(2)
    inputs = {"text_encoder_output": {"text_encoder_1_text_states": text_states.cuda(), "text_encoder_1_attention_mask": mask.cuda(), "text_encoder_2_text_states": ts2.cuda()}}
    model.infer(inputs)
    with torch.no_grad():
        ref = H.forward(wd, dims, lat.to(torch.bfloat16), sch.timesteps[2].cpu(), sch.guidance.cpu(), text_states, mask, ts2, (sch.freqs_cos.cpu(), sch.freqs_sin.cpu()))
    assert_rel(sch.noise_pred, ref, 2e-2, "forward with padded text")


def test_forward_default_rounding_fast_attention(setup):
    """Default mode: fp32 norm statistics, q pre-scaled by the norm+RoPE kernel, v3 attention kernel (x2v.h variants)."""
    from lightx2v_amd import hunyuan as hy

    _, _, g, wd, dims = setup
    cfg = hy.default_config(dims, infer_steps=4)
    model = hy.HunyuanModel(cfg, {k: v.cuda() for k, v in wd.items()})
    sch = hy.HunyuanScheduler(cfg)
    sch.prepare(g["latents"])
    model.set_scheduler(sch)
    sch.step_pre(1)
    model.infer(_inputs(g))
    assert_rel(sch.noise_pred, g["noise_pred"], 2e-2, "HunyuanModel.infer (default rounding)")


@pytest.mark.parametrize("ref_rounding", [False, True])
def test_hunyuan_teacache_matches_the_reference_run(ref_rounding):
    """HunyuanModel(feature_caching="Tea") over the 10-step loop of tests/golden/hunyuan_teacache.safetensors (generated by running the
    reference's own HunyuanTransformerInferTeaCaching, hunyuan/infer/feature_caching/transformer_infer.py:7-135): the same per-step
    decisions — steps 2 and 3 re-apply the cached residual — and latents / transformer outputs within the model-level tolerances
    (the skipped steps must reproduce `img += previous_residual` to rounding: their error is the previous computed step's)."""
    from safetensors.torch import load_file

    from lightx2v_amd import hunyuan as hy, synth
    from tests.util import rel_l2

    g = load_file(os.path.join(GOLDEN, "hunyuan_teacache.safetensors"))
    dims = synth.HUNYUAN_DIMS["hunyuan-tiny"]
    ts = synth.HUNYUAN_WORKLOADS["hunyuan-tiny"]["target_shape"]
    wd = synth.synth_hunyuan_weights(dims, seed=4)
    lat, text_states, text_mask, text_states_2 = synth.synth_hunyuan_inputs(dims, ts)
    steps = g["records"].numel()
    cfg = hy.default_config(dims, infer_steps=steps, hip_ref_rounding=ref_rounding, feature_caching="Tea", teacache_thresh=float(g["thresh"]))
    model = hy.HunyuanModel(cfg, {k: v.cuda() for k, v in wd.items()})
    assert type(model.transformer_infer) is hy.HunyuanTransformerInferTeaCaching
    sch = hy.HunyuanScheduler(cfg)
    sch.prepare(lat)
    model.set_scheduler(sch)
    inputs = {"text_encoder_output": {"text_encoder_1_text_states": text_states.cuda(), "text_encoder_1_attention_mask": text_mask.cuda(),
                                      "text_encoder_2_text_states": text_states_2.cuda()}}
    outs = []
    orig = model.transformer_infer.infer

    def spy(*a, **kw):
        img, vec = orig(*a, **kw)
        outs.append(img.float().cpu().clone())
        return img, vec

    model.transformer_infer.infer = spy
    for i in range(steps):
        sch.step_pre(i)
        model.infer(inputs)
        sch.step_post()
        assert rel_l2(outs[i], g[f"tr_img_{i}"]) <= 3e-2, (i, rel_l2(outs[i], g[f"tr_img_{i}"]))
        assert rel_l2(sch.latents, g[f"latents_{i}"]) <= 3e-2, (i, rel_l2(sch.latents, g[f"latents_{i}"]))
    assert [int(bool(r)) for r in sch.caching_records] == g["records"].tolist(), sch.caching_records
    assert g["records"].tolist().count(0) == 2
