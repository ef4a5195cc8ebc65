"""A torch-CPU restatement of csrc/sched.hip's per-element op chain (one separately rounded fp32 op per line, same order), used to check
the fused step's coefficient plumbing and op order on CPU against the torch-path scheduler — the kernel itself is compared with the
oracle on the GPU (tests/test_gpu_sched.py)."""
import torch


def _t(x):
    return torch.tensor(x, dtype=torch.float32)


def unipc_step(cond, uncond, latents, last_sample, m0p, m1p, coef, order_c, order_p):
    guide, sigma_i, c_a, c_b, c_c, c_rk, c_rho0, c_rhol, p_a, p_b, p_c, p_rk = (_t(float(c)) for c in coef)
    mo = cond
    if uncond is not None:
        d = mo - uncond
        g = guide * d
        mo = uncond + g
    sample = latents.to(torch.float32)
    sm = sigma_i * mo
    x0 = sample - sm
    if order_c > 0:
        ta = c_a * last_sample
        tb = c_b * m0p
        xt = ta - tb
        dx = x0 - m0p
        rl = c_rhol * dx
        if order_c >= 2:
            dm = m1p - m0p
            d1 = dm / c_rk
            corr = c_rho0 * d1
            t = corr + rl
        else:
            t = _t(0.0) + rl
        ct = c_c * t
        sample = xt - ct
    ta = p_a * sample
    tb = p_b * x0
    xt = ta - tb
    if order_p >= 2:
        dm = m0p - x0
        d1 = dm / p_rk
        pred = _t(0.5) * d1
        pt = p_c * pred
    else:
        pt = p_c * _t(0.0)
    return mo, x0, sample, xt - pt
