"""CPU check of the fused step_post's host half (lightx2v_amd/scheduler.py::_step_post_fused): the coefficients it extracts and the op
order csrc/sched.hip hard-codes, replayed by tests/sched_emul.py in torch fp32, must reproduce the torch-path scheduler — itself
bit-exact against the reference (tests/test_host_scheduler.py) — bit for bit, signs of zero included."""
import pytest
import torch

from lightx2v_amd import lib, scheduler
from tests import sched_emul


def _cfg(steps, shift, ts):
    return {"infer_steps": steps, "sample_shift": shift, "target_shape": ts, "patch_size": (1, 2, 2), "seed": 0}


@pytest.mark.parametrize("steps,shift,cfg_on", [(1, 8.0, True), (2, 8.0, True), (3, 5.0, False), (4, 8.0, True), (10, 3.0, True), (50, 8.0, True), (7, 1.0, False)])
def test_fused_step_host_half_matches_torch_path(monkeypatch, steps, shift, cfg_on):
    ts = (16, 2, 4, 6)
    ref = scheduler.WanScheduler(_cfg(steps, shift, ts), device="cpu")
    fused = scheduler.WanScheduler(_cfg(steps, shift, ts), device="cpu")
    lat0 = torch.randn(*ts, generator=torch.Generator().manual_seed(steps))
    lat0[0, 0, 0, :3] = torch.tensor([0.0, -0.0, 1e-30])  # zeros of both signs and a tiny value go through the chain too
    ref.prepare(latents=lat0)
    fused.prepare(latents=lat0)

    def fake_unipc_step(cond, uncond, latents, last_sample, m0, m1, coef, order_c, order_p, want_noise_pred=False):
        mo, x0, sample, new_lat = sched_emul.unipc_step(cond, uncond, latents, last_sample, m0, m1, coef, order_c, order_p)
        return None, x0, sample, new_lat

    monkeypatch.setattr(lib, "unipc_step", fake_unipc_step)
    for i in range(steps):
        for s in (ref, fused):
            s.step_pre(i)
        cond = torch.sin(ref.latents.float() * 1.3 + 0.1 * i) + 0.05 * i
        uncond = torch.cos(ref.latents.float() * 0.7 - 0.2 * i)
        if i == 1:
            cond[0, 0, 0, 0] = uncond[0, 0, 0, 0] = 0.0
        if cfg_on:
            ref.noise_pred = uncond + 6.0 * (cond - uncond)
            fused.set_cfg_parts(cond, uncond, 6.0)
            assert torch.equal(fused.noise_pred, ref.noise_pred)
            fused.set_cfg_parts(cond, uncond, 6.0)  # reading noise_pred materialised it; hand the branches over again
        else:
            ref.noise_pred = cond
            fused.noise_pred = cond
        ref.step_post()
        fused._step_post_fused()
        for a, b, nm in ((fused.latents, ref.latents, "latents"), (fused.last_sample, ref.last_sample, "last_sample"), (fused.model_outputs[-1], ref.model_outputs[-1], "x0")):
            assert a.dtype == b.dtype == torch.float32
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), f"step {i}: {nm} differs (bitwise)"
        assert fused.this_order == ref.this_order and fused.lower_order_nums == ref.lower_order_nums
