"""Host-side schedulers of the denoise loop (reference: lightx2v/models/schedulers/scheduler.py:5-21,
wan/scheduler.py:9-360, wan/step_distill/scheduler.py:8-56).  Same public surface — `prepare()`,
`step_pre(i)`, `step_post()`, `.latents`, `.timesteps`, `.sigmas`, `.noise_pred`, `clear()` — so the
reference's runner loop (default_runner.py:97-114) drives them unchanged.

Two forms of `step_post`, bit-identical to each other and to the reference's CPU result:
  * latents on a HIP device: ONE launch of x2v_unipc_step_f32 / x2v_distill_step_f32 (csrc/sched.hip) that also does the
    CFG combine when the model handed over the two branches (`set_cfg_parts`) — ~25 torch launches and their temporaries become one
    pass over the 19 MB latent tensor, which matters for the 4-step distilled and the 256x256 configs;
  * anywhere else (CPU tests, the reference-parity fixtures): the torch op sequence below.
Scalar coefficients are fp32 0-dim tensors combined in the reference's order in both forms.
"""
import math

import numpy as np
import torch


class BaseScheduler:
    def __init__(self, config):
        self.config = config
        self.step_index = 0
        self.latents = None
        self.infer_steps = config["infer_steps"]
        self._noise_pred = None
        self._cfg_parts = None  # (cond, uncond, guide): the CFG combine deferred into the fused step_post
        self.bf16_latents = True  # DTYPE=BF16 mode of the reference (utils/envs.py; all shipped scripts)
        self.fused_step_post = bool(config.get("fused_step_post", True)) if hasattr(config, "get") else True

    # `noise_pred` is what the model writes and step_post reads (scheduler.py:5-21).  A model may instead hand over the two CFG
    # branches; the combined tensor then only materialises if somebody reads the attribute (same three fp32 ops as wan/model.py:218).
    @property
    def noise_pred(self):
        if self._noise_pred is None and self._cfg_parts is not None:
            cond, uncond, guide = self._cfg_parts
            self._noise_pred = uncond + guide * (cond - uncond)
        return self._noise_pred

    @noise_pred.setter
    def noise_pred(self, value):
        self._noise_pred, self._cfg_parts = value, None

    def set_cfg_parts(self, cond, uncond, guide):
        self._noise_pred, self._cfg_parts = None, (cond, uncond, float(guide))

    def _branches(self):
        """(cond, uncond | None, guide) as contiguous fp32 tensors for the fused kernels."""
        if self._cfg_parts is not None and self._noise_pred is None:
            cond, uncond, guide = self._cfg_parts
            return cond.to(torch.float32).contiguous(), uncond.to(torch.float32).contiguous(), guide
        return self.noise_pred.to(torch.float32).contiguous(), None, 0.0

    def step_pre(self, step_index):
        self.step_index = step_index
        if self.bf16_latents:
            self.latents = self.latents.to(dtype=torch.bfloat16)

    def clear(self):
        pass


class WanScheduler(BaseScheduler):
    """Flow-matching UniPC (bh2), solver order 2, predictor + corrector in fp32."""

    def __init__(self, config, device="cuda"):
        super().__init__(config)
        self.device = torch.device(device)
        self.sample_shift = config["sample_shift"]
        self.num_train_timesteps = 1000
        self.solver_order = 2
        self.disable_corrector = []
        # per-step calc/skip records of the feature-caching infer classes (wan/scheduler.py:22, feature_caching/scheduler.py)
        self.caching_records = [True] * config["infer_steps"]
        self.caching_records_2 = [True] * config["infer_steps"]

    # ---- setup -----------------------------------------------------------------------------------
    def prepare(self, image_encoder_output=None, latents=None):
        """`latents` injects the initial noise (parity runs: CPU and GPU randn streams differ, SURVEY.md
        appendix A.10); otherwise seeded randn on this scheduler's device as wan/scheduler.py:24-28,54-63."""
        ts = self.config["target_shape"]
        if latents is not None:
            self.latents = latents.to(self.device, torch.float32).clone()
        else:
            gen = torch.Generator(device=self.device)
            gen.manual_seed(self.config["seed"])
            self.latents = torch.randn(ts[0], ts[1], ts[2], ts[3], dtype=torch.float32, device=self.device, generator=gen)
        ps = self.config["patch_size"]
        self.seq_len = math.ceil((ts[2] * ts[3]) / (ps[1] * ps[2]) * ts[1])
        n = self.num_train_timesteps
        train_sigmas = torch.from_numpy(1.0 - np.linspace(1, 1 / n, n)[::-1].copy()).to(torch.float32)
        self.sigma_min, self.sigma_max = train_sigmas[-1].item(), train_sigmas[0].item()
        self.set_timesteps(self.infer_steps, shift=self.sample_shift)

    def set_timesteps(self, infer_steps, shift=1.0):
        sig = np.linspace(self.sigma_max, self.sigma_min, infer_steps + 1).copy()[:-1]
        sig = shift * sig / (1 + (shift - 1) * sig)
        self.timesteps = torch.from_numpy(sig * self.num_train_timesteps).to(device=self.device, dtype=torch.int64)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0]]).astype(np.float32))  # kept on the host
        self.model_outputs = [None] * self.solver_order
        self.lower_order_nums = 0
        self.last_sample = None
        self.this_order = None

    # ---- UniPC pieces ----------------------------------------------------------------------------
    @staticmethod
    def _lambda(sigma):
        return torch.log(1 - sigma) - torch.log(sigma)

    def _coeffs(self, idx_t, idx_s0, order, d1_base):
        """Shared front half of the predictor/corrector: h, r_k, D1 differences, the bh2 (R, b) system."""
        sigma_t, sigma_s0 = self.sigmas[idx_t], self.sigmas[idx_s0]
        lam_s0 = self._lambda(sigma_s0)
        h = self._lambda(sigma_t) - lam_s0
        m0 = self.model_outputs[-1]
        rks, d1s = [], []
        for i in range(1, order):
            mi = self.model_outputs[-(i + 1)]
            rk = (self._lambda(self.sigmas[d1_base - i]) - lam_s0) / h
            rks.append(rk)
            d1s.append((mi - m0) / rk)
        rks.append(1.0)
        rks = torch.tensor(rks)
        hh = -h
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = torch.expm1(hh)
        R, b, fact = [], [], 1
        for i in range(1, order + 1):
            R.append(torch.pow(rks, i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        D1s = torch.stack(d1s, dim=1) if d1s else None
        return sigma_t, sigma_s0, 1 - sigma_t, h_phi_1, B_h, torch.stack(R), torch.tensor(b), D1s, m0

    def multistep_uni_p_bh_update(self, sample, order):
        sigma_t, sigma_s0, alpha_t, h_phi_1, B_h, R, b, D1s, m0 = self._coeffs(self.step_index + 1, self.step_index, order, self.step_index)
        x_t_ = sigma_t / sigma_s0 * sample - alpha_t * h_phi_1 * m0
        if D1s is not None:
            rhos_p = torch.tensor([0.5], dtype=sample.dtype) if order == 2 else torch.linalg.solve(R[:-1, :-1], b[:-1]).to(sample.dtype)
            pred_res = torch.einsum("k,bkc...->bc...", rhos_p.to(sample.device), D1s)
        else:
            pred_res = 0
        return (x_t_ - alpha_t * B_h * pred_res).to(sample.dtype)

    def multistep_uni_c_bh_update(self, this_model_output, last_sample, order):
        sigma_t, sigma_s0, alpha_t, h_phi_1, B_h, R, b, D1s, m0 = self._coeffs(self.step_index, self.step_index - 1, order, self.step_index - 1)
        rhos_c = torch.tensor([0.5], dtype=last_sample.dtype) if order == 1 else torch.linalg.solve(R, b).to(last_sample.dtype)
        x_t_ = sigma_t / sigma_s0 * last_sample - alpha_t * h_phi_1 * m0
        corr_res = torch.einsum("k,bkc...->bc...", rhos_c[:-1].to(last_sample.device), D1s) if D1s is not None else 0
        return (x_t_ - alpha_t * B_h * (corr_res + rhos_c[-1] * (this_model_output - m0))).to(last_sample.dtype)

    def _scalars(self, idx_t, idx_s0, order, d1_base):
        """The scalar half of `_coeffs` as Python floats (each an exactly represented fp32): (a, b, c, r_1, R, b_vec)."""
        sigma_t, sigma_s0 = self.sigmas[idx_t], self.sigmas[idx_s0]
        lam_s0 = self._lambda(sigma_s0)
        h = self._lambda(sigma_t) - lam_s0
        rks = [(self._lambda(self.sigmas[d1_base - i]) - lam_s0) / h for i in range(1, order)]
        rk1 = rks[0].item() if rks else 1.0
        rks = torch.tensor(rks + [1.0])
        hh = -h
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = torch.expm1(hh)
        R, b, fact = [], [], 1
        for i in range(1, order + 1):
            R.append(torch.pow(rks, i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        alpha_t = 1 - sigma_t
        return (sigma_t / sigma_s0).item(), (alpha_t * h_phi_1).item(), (alpha_t * B_h).item(), rk1, torch.stack(R), torch.tensor(b)

    def _step_post_fused(self):
        """CFG combine + x0 prediction + corrector + predictor in one launch (csrc/sched.hip)."""
        from . import lib

        i = self.step_index
        cond, uncond, guide = self._branches()
        use_corrector = i > 0 and (i - 1) not in self.disable_corrector and self.last_sample is not None
        order_c = self.this_order if use_corrector else 0
        coef = [guide, self.sigmas[i].item()] + [0.0] * 10
        if use_corrector:
            a, b, c, rk, R, bvec = self._scalars(i, i - 1, order_c, i - 1)
            rhos_c = torch.tensor([0.5], dtype=torch.float32) if order_c == 1 else torch.linalg.solve(R, bvec).to(torch.float32)
            coef[2:8] = [a, b, c, rk, rhos_c[0].item() if order_c == 2 else 0.0, rhos_c[-1].item()]
        this_order = min(self.solver_order, len(self.timesteps) - i)
        order_p = min(this_order, self.lower_order_nums + 1)  # multistep warm-up
        if order_p > 2 or order_c > 2:
            raise NotImplementedError("fused step_post: solver order <= 2 (the reference's setting)")
        a, b, c, rk, _, _ = self._scalars(i + 1, i, order_p, i)
        coef[8:12] = [a, b, c, rk]
        m0, m1 = self.model_outputs[-1], self.model_outputs[-2]
        lat = self.latents if self.latents.dtype in (torch.bfloat16, torch.float32) else self.latents.to(torch.float32)
        _, x0, sample, new_lat = lib.unipc_step(cond, uncond, lat.contiguous(), self.last_sample if use_corrector else None, m0, m1 if (order_c == 2) else None, coef,
                                                order_c, order_p)
        self.model_outputs = self.model_outputs[1:] + [x0]
        self.this_order = order_p
        self.last_sample = sample
        self.latents = new_lat
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1

    def step_post(self):
        if self.fused_step_post and self.latents.is_cuda:
            return self._step_post_fused()
        model_output = self.noise_pred.to(torch.float32)
        sample = self.latents.to(torch.float32)
        use_corrector = self.step_index > 0 and (self.step_index - 1) not in self.disable_corrector and self.last_sample is not None
        x0_pred = sample - self.sigmas[self.step_index] * model_output  # flow-matching x0 prediction
        if use_corrector:
            sample = self.multistep_uni_c_bh_update(x0_pred, self.last_sample, self.this_order)
        self.model_outputs = self.model_outputs[1:] + [x0_pred]
        this_order = min(self.solver_order, len(self.timesteps) - self.step_index)
        self.this_order = min(this_order, self.lower_order_nums + 1)  # multistep warm-up
        self.last_sample = sample
        self.latents = self.multistep_uni_p_bh_update(sample, self.this_order)
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1

    def reset(self):
        self.model_outputs = [None] * self.solver_order
        self.lower_order_nums = 0
        self.last_sample = None
        self.this_order = None
        self.noise_pred = None  # (the setter drops pending CFG branches as well)


class WanStepDistillScheduler(WanScheduler):
    """4-step distilled sampling (BASELINE config #4): x0 = x - sigma*v, then re-noise to the next sigma.
    `noise_fn` lets tests inject the re-noise tensor (the reference draws it from the unseeded global RNG,
    step_distill/scheduler.py:55, so whole trajectories are not reproducible there either)."""

    def __init__(self, config, device="cuda", noise_fn=None):
        super().__init__(config, device)
        self.denoising_step_list = list(config["denoising_step_list"])
        self.infer_steps = len(self.denoising_step_list)
        self.noise_fn = noise_fn or torch.randn_like

    def set_timesteps(self, infer_steps, shift=1.0):
        n = self.num_train_timesteps
        sig = torch.linspace(1.0, 0.0, n + 1)[:-1]
        sig = shift * sig / (1 + (shift - 1) * sig)
        idx = [n - s for s in self.denoising_step_list]
        self.timesteps = (sig * n)[idx].to(self.device)
        self.sigmas = sig[idx].to("cpu")

    def step_post(self):
        if self.fused_step_post and self.latents.is_cuda and self.latents.dtype in (torch.bfloat16, torch.float32):
            from . import lib

            cond, uncond, guide = self._branches()
            sigma = self.sigmas[self.step_index].item()
            noise, nxt = None, 0.0
            if self.step_index < self.infer_steps - 1:
                nxt = self.sigmas[self.step_index + 1].item()
                noise = self.noise_fn(cond).to(torch.float32).contiguous()
            _, self.latents = lib.distill_step(cond, uncond, guide, self.latents.contiguous(), noise, sigma, 1 - nxt, nxt)
            return
        flow_pred = self.noise_pred.to(torch.float32)
        sigma = self.sigmas[self.step_index].item()
        x0 = self.latents.to(torch.float32) - sigma * flow_pred
        if self.step_index < self.infer_steps - 1:
            nxt = self.sigmas[self.step_index + 1].item()
            noise = self.noise_fn(x0)
            x0 = ((1 - nxt) * x0 + nxt * noise).type_as(noise)
        self.latents = x0.to(self.latents.dtype)


def run_denoise_loop(model, scheduler, inputs, step_callback=None):
    """reference: models/runners/default_runner.py:97-114 — step_pre → model.infer → step_post."""
    for step in range(scheduler.infer_steps):
        scheduler.step_pre(step)
        model.infer(inputs)
        scheduler.step_post()
        if step_callback is not None:
            step_callback(step)
    return scheduler.latents
