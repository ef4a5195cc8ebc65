"""Host logic (no GPU): the product scheduler reproduces the reference's UniPC trajectories bit for bit on
CPU (fixtures generated from the reference by oracle/gen_golden.py)."""
import pytest
import torch

from lightx2v_amd.scheduler import WanScheduler, WanStepDistillScheduler
from lightx2v_amd.wan import default_config
from lightx2v_amd import synth


def _cfg(steps, shift, shape):
    return default_config(synth.WAN_DIMS["wan-tiny"], infer_steps=steps, sample_shift=shift, target_shape=shape)


def test_unipc_matches_reference_known_answers(golden_sched):
    g = golden_sched
    for steps, shift in ((50, 8.0), (4, 8.0), (10, 3.0)):
        tag = f"s{steps}_sh{int(shift)}"
        sch = WanScheduler(_cfg(steps, shift, (16, 2, 4, 4)), device="cpu")
        sch.prepare(latents=g[f"{tag}_lat0"])
        assert torch.equal(sch.timesteps, g[f"{tag}_timesteps"])
        assert torch.equal(sch.sigmas, g[f"{tag}_sigmas"])
        for i in range(steps):
            sch.step_pre(i)
            sch.noise_pred = torch.sin(sch.latents.float() * 1.3 + 0.1 * i) + 0.05 * i
            sch.step_post()
        assert torch.equal(sch.latents, g[f"{tag}_final"]), (sch.latents - g[f"{tag}_final"]).abs().max()


def test_seq_len_and_seeded_latents():
    sch = WanScheduler(_cfg(4, 8.0, (16, 21, 90, 160)), device="cpu")
    sch.prepare()
    assert sch.seq_len == 75600  # BASELINE config 3: 720p x 81f
    assert sch.latents.shape == (16, 21, 90, 160) and sch.latents.dtype == torch.float32
    sch2 = WanScheduler(_cfg(4, 8.0, (16, 13, 60, 104)), device="cpu")
    sch2.prepare()
    assert sch2.seq_len == 20280  # config 2: 480p x 49f


def test_step_distill_schedule_and_update():
    cfg = _cfg(4, 5.0, (16, 2, 4, 4))
    cfg["denoising_step_list"] = [1000, 750, 500, 250]
    noise = []

    def noise_fn(x):
        g = torch.Generator().manual_seed(len(noise))
        n = torch.randn(x.shape, generator=g)
        noise.append(n)
        return n

    sch = WanStepDistillScheduler(cfg, device="cpu", noise_fn=noise_fn)
    sch.prepare()
    # reference: step_distill/scheduler.py:32-40 — sigma table from linspace(1,0,1001)[:-1], shift 5
    sig = torch.linspace(1.0, 0.0, 1001)[:-1]
    sig = 5.0 * sig / (1 + 4.0 * sig)
    idx = [0, 250, 500, 750]
    assert torch.equal(sch.sigmas, sig[idx])
    assert torch.equal(sch.timesteps, (sig * 1000)[idx])
    lat = sch.latents.clone()
    for i in range(4):
        sch.step_pre(i)
        lat_b = sch.latents.clone()
        sch.noise_pred = torch.cos(sch.latents.float())
        sch.step_post()
        x0 = lat_b.float() - sch.sigmas[i].item() * torch.cos(lat_b.float())
        if i < 3:
            s1 = sch.sigmas[i + 1].item()
            x0 = (1 - s1) * x0 + s1 * noise[i]
        assert torch.equal(sch.latents, x0.to(torch.bfloat16))


def test_step_distill_matches_reference_fixture():
    """tests/golden/scheduler.safetensors `distill_*`: the reference's own WanStepDistillScheduler (gen_golden.py::gen_scheduler_only) with
    the global RNG seeded before every step_post; same sigma table, timesteps and latents after every step, bit for bit."""
    import os

    from safetensors.torch import load_file

    g = load_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scheduler.safetensors"))
    cfg = _cfg(4, 5.0, (16, 2, 4, 4))
    cfg["denoising_step_list"] = [1000, 750, 500, 250]
    step = [0]

    def noise_fn(x):
        torch.manual_seed(100 + step[0])
        return torch.randn_like(x)

    sch = WanStepDistillScheduler(cfg, device="cpu", noise_fn=noise_fn)
    sch.prepare(latents=g["distill_lat0"])
    assert torch.equal(sch.sigmas, g["distill_sigmas"]) and torch.equal(sch.timesteps, g["distill_timesteps"])
    for i in range(4):
        step[0] = i
        sch.step_pre(i)
        sch.noise_pred = torch.cos(sch.latents.float() * 0.7 + 0.2 * i)
        sch.step_post()
        assert sch.latents.dtype == g[f"distill_lat{i + 1}"].dtype
        assert torch.equal(sch.latents, g[f"distill_lat{i + 1}"]), f"latents after step {i + 1}"


@pytest.mark.parametrize("steps,shift", [(1, 8.0), (2, 8.0), (3, 5.0), (7, 1.0), (20, 17.0)])
def test_unipc_bit_exact_against_live_reference_at_edge_step_counts(steps, shift):
    """Where /root/reference exists: the reference's WanScheduler, run side by side on CPU at step counts the committed fixture does
    not hold — one step (predictor only, order 1), two (warm-up then final order-1 step), odd counts, no shift, a large shift."""
    from oracle import ref_import

    if not ref_import.reference_available():
        pytest.skip("reference checkout not present (authoring container only)")
    ref_import.patch_and_import()
    from lightx2v.models.schedulers.wan.scheduler import WanScheduler as RefScheduler

    from lightx2v_amd import scheduler, synth

    cfg = ref_import.make_config(synth.WAN_DIMS["wan-tiny"], infer_steps=steps, sample_shift=shift, target_shape=(16, 2, 4, 4))
    ref = RefScheduler(cfg)
    ref.device = torch.device("cpu")
    ref.prepare()
    ours = scheduler.WanScheduler(dict(cfg), device="cpu")
    lat0 = torch.randn(16, 2, 4, 4, generator=torch.Generator().manual_seed(steps))
    ours.prepare(latents=lat0)
    ref.latents = lat0.clone()
    assert torch.equal(ours.timesteps, ref.timesteps) and torch.equal(ours.sigmas, ref.sigmas.cpu())
    for i in range(steps):
        for s in (ref, ours):
            s.step_pre(i)
            s.noise_pred = torch.sin(s.latents.float() * 1.3 + 0.1 * i) + 0.05 * i
            s.step_post()
        assert torch.equal(ours.latents, ref.latents), f"step {i}"


@pytest.mark.parametrize("steps,grid", [(4, (3, 8, 12)), (50, (33, 90, 160)), (7, (1, 6, 10))])
def test_hunyuan_scheduler_against_live_reference_functions(steps, grid):
    """Where /root/reference exists: the flow-match tables (set_timesteps_sigmas, shift 7), the (T, H/2, W/2) RoPE tables
    (get_nd_rotary_pos_embed, dims [16, 56, 56], theta 256, bf16) and the t2v Euler update of the reference's HunyuanScheduler.step_post
    (schedulers/hunyuan/scheduler.py:237-260), called unbound on a stand-in object since its constructor needs a CUDA generator."""
    from types import SimpleNamespace

    from oracle import ref_import

    if not ref_import.reference_available():
        pytest.skip("reference checkout not present (authoring container only)")
    ref_import.patch_and_import()
    from lightx2v.models.schedulers.hunyuan import scheduler as ref_sched

    from lightx2v_amd import hunyuan as hy

    t, h, w = grid
    ours = hy.HunyuanScheduler({"infer_steps": steps}, device="cpu")
    tt, ss = ref_sched.set_timesteps_sigmas(steps, 7.0, device=torch.device("cpu"))
    assert torch.equal(ours.timesteps, tt) and torch.equal(ours.sigmas, ss)
    small = min(t, 3), min(h, 12), min(w, 16)  # the table is a pure function of the grid: full size for the rope check only
    lat = torch.randn(1, 16, small[0], small[1], small[2], generator=torch.Generator().manual_seed(steps))
    ours.prepare(lat)
    fc, fs = ref_sched.get_nd_rotary_pos_embed([16, 56, 56], [t, h // 2, w // 2], theta=256, use_real=True, theta_rescale_factor=1)
    c2, s2 = hy.rope_tables([t, h // 2, w // 2])
    assert torch.equal(c2, fc.to(torch.bfloat16)) and torch.equal(s2, fs.to(torch.bfloat16))
    assert torch.equal(ours.guidance, torch.tensor([6.0], dtype=torch.bfloat16) * 1000.0)
    ref = SimpleNamespace(config=SimpleNamespace(task="t2v"), latents=lat.clone(), sigmas=ss, step_index=0, noise_pred=None)
    for i in range(min(steps, 5)):
        pred = torch.cos(ours.latents.float() * 0.9 + i).to(torch.bfloat16)
        ours.step_pre(i)
        ref.step_index = i
        ours.noise_pred = ref.noise_pred = pred
        ours.step_post()
        ref_sched.HunyuanScheduler.step_post(ref)
        assert torch.equal(ours.latents, ref.latents) and ours.latents.dtype == ref.latents.dtype, f"step {i}"
