"""The fused sampler-step kernels (csrc/sched.hip: CFG combine + WanScheduler.step_post, and the step-distill update) against the CPU
oracle `oracle.WanSchedulerOracle` (bit-exact against the reference's scheduler, tests/test_oracle_golden.py) — whole trajectories,
bit for bit: integer-exact is the bar for this fp32 elementwise path because every op is rounded exactly where the reference rounds."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(steps, shift, ts, **kw):
    return dict({"infer_steps": steps, "sample_shift": shift, "target_shape": ts, "patch_size": (1, 2, 2), "seed": 0}, **kw)


def _bits(t):
    return t.detach().float().cpu().contiguous().view(torch.int32)


@pytest.mark.parametrize("steps,shift,cfg_on", [(1, 8.0, True), (2, 8.0, True), (3, 5.0, False), (4, 8.0, True), (10, 3.0, True), (50, 8.0, True)])
def test_fused_unipc_step_bit_exact_vs_oracle(steps, shift, cfg_on):
    from lightx2v_amd import scheduler
    from oracle import wan_oracle as O

    ts = (16, 3, 10, 14)
    lat0 = torch.randn(*ts, generator=torch.Generator().manual_seed(steps))
    lat0[0, 0, 0, :3] = torch.tensor([0.0, -0.0, 1e-30])
    ref = O.WanSchedulerOracle(steps, shift, lat0)
    ours = scheduler.WanScheduler(_cfg(steps, shift, ts), device="cuda")
    ours.prepare(latents=lat0)
    assert ours.fused_step_post
    for i in range(steps):
        ref.step_pre(i)
        ours.step_pre(i)
        assert torch.equal(ours.latents.cpu(), ref.latents), f"step {i}: bf16 latents entering the step"
        cond = torch.sin(ref.latents.float() * 1.3 + 0.1 * i) + 0.05 * i
        uncond = torch.cos(ref.latents.float() * 0.7 - 0.2 * i)
        if i == 1:
            cond[0, 0, 0, 0] = uncond[0, 0, 0, 0] = 0.0
        if cfg_on:
            ref.noise_pred = uncond + 6.0 * (cond - uncond)  # wan/model.py:218 on CPU
            ours.set_cfg_parts(cond.cuda(), uncond.cuda(), 6.0)
        else:
            ref.noise_pred = cond
            ours.noise_pred = cond.cuda()
        ref.step_post()
        ours.step_post()
        assert ours.latents.dtype == torch.float32 and ours.latents.is_cuda
        for a, b, nm in ((ours.latents, ref.latents, "latents"), (ours.last_sample, ref.last_sample, "last_sample"), (ours.model_outputs[-1], ref.model_outputs[-1], "x0")):
            assert torch.equal(_bits(a), _bits(b)), f"step {i}: {nm} differs from the oracle (max |d| {(a.cpu() - b).abs().max().item():.3e})"


def test_fused_step_matches_torch_path_fp32_latents_and_noise_pred_property():
    """The reference's non-BF16 mode keeps fp32 latents through step_pre; the lazily materialised `noise_pred` equals the CFG formula; and the
    torch op sequence (fused_step_post=False) on the same device gives the same trajectory up to the device's own division rounding —
    compared on CPU, where torch divides exactly as the reference does."""
    from lightx2v_amd import lib, scheduler

    ts, steps = (16, 2, 6, 6), 5
    lat0 = torch.randn(*ts, generator=torch.Generator().manual_seed(3))
    cpu = scheduler.WanScheduler(_cfg(steps, 8.0, ts), device="cpu")
    gpu = scheduler.WanScheduler(_cfg(steps, 8.0, ts), device="cuda")
    for s in (cpu, gpu):
        s.bf16_latents = False
        s.prepare(latents=lat0)
    for i in range(steps):
        cpu.step_pre(i)
        gpu.step_pre(i)
        cond = torch.sin(cpu.latents * 1.1 + i)
        uncond = torch.cos(cpu.latents * 0.9 - i)
        cpu.noise_pred = uncond + 5.0 * (cond - uncond)
        gpu.set_cfg_parts(cond.cuda(), uncond.cuda(), 5.0)
        assert torch.equal(_bits(gpu.noise_pred), _bits(cpu.noise_pred))
        gpu.set_cfg_parts(cond.cuda(), uncond.cuda(), 5.0)
        cpu.step_post()
        gpu.step_post()
        assert torch.equal(_bits(gpu.latents), _bits(cpu.latents)), f"step {i}"
    # the kernel can also hand back the combined prediction
    n = lat0.numel()
    np_, x0, sample, new = lib.unipc_step(cond.cuda(), uncond.cuda(), lat0.cuda(), None, None, None, [5.0, 0.5] + [0.0] * 6 + [0.9, 0.1, -0.2, 1.0], 0, 1, want_noise_pred=True)
    assert torch.equal(_bits(np_), _bits(uncond + 5.0 * (cond - uncond))) and x0.numel() == n


def test_fused_distill_step_bit_exact():
    """step_distill/scheduler.py:40-56 with injected re-noise tensors: the fused kernel vs the torch path on CPU (bit-exact against the
    reference fixture, tests/test_host_scheduler.py), bf16 latents, with and without CFG."""
    from lightx2v_amd import scheduler

    ts = (16, 2, 6, 10)
    lat0 = torch.randn(*ts, generator=torch.Generator().manual_seed(9))
    noise = [torch.randn(*ts, generator=torch.Generator().manual_seed(100 + i)) for i in range(4)]
    for cfg_on in (False, True):
        step = [0]
        cfg = _cfg(4, 5.0, ts, denoising_step_list=[1000, 750, 500, 250])
        cpu = scheduler.WanStepDistillScheduler(cfg, device="cpu", noise_fn=lambda x: noise[step[0]])
        gpu = scheduler.WanStepDistillScheduler(cfg, device="cuda", noise_fn=lambda x: noise[step[0]].cuda())
        cpu.prepare(latents=lat0)
        gpu.prepare(latents=lat0)
        for i in range(4):
            step[0] = i
            cpu.step_pre(i)
            gpu.step_pre(i)
            cond = torch.cos(cpu.latents.float() * 0.7 + 0.2 * i)
            uncond = torch.sin(cpu.latents.float() * 0.4 - 0.1 * i)
            if cfg_on:
                cpu.noise_pred = uncond + 4.0 * (cond - uncond)
                gpu.set_cfg_parts(cond.cuda(), uncond.cuda(), 4.0)
            else:
                cpu.noise_pred = cond
                gpu.noise_pred = cond.cuda()
            cpu.step_post()
            gpu.step_post()
            assert gpu.latents.dtype == cpu.latents.dtype == torch.bfloat16
            assert torch.equal(gpu.latents.cpu(), cpu.latents), f"cfg={cfg_on} step {i}"
