#!/usr/bin/env python
"""Times the HunyuanVideo VAE decode (HIP path) on one full-size tile and, with --full, on the whole 720p x 129-frame latent.
    python tools/hunyuan_vae_bench.py [--full] [--fp32]      (default: fp16 convolution operands, the reference's precision)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightx2v_amd import hunyuan_vae, lib, synth  # noqa: E402


def main():
    full = "--full" in sys.argv
    lib.init(0)
    cfg = synth.HUNYUAN_VAE_CFG
    conv16 = "--fp32" not in sys.argv
    m = hunyuan_vae.VideoEncoderKLCausal3DModel(synth.synth_hunyuan_vae_weights(cfg, seed=0), cfg, conv16=conv16)
    shape = (1, 16, 33, 90, 160) if full else (1, 16, 17, 32, 32)
    z = (torch.randn(*shape, generator=torch.Generator().manual_seed(1)) * 0.5).cuda()
    flops = [0.0]
    orig = lib.vae_conv

    def counted(xp, strides, weight, out, T, H, W, **kw):
        cin = kw.get("cin") or (weight.shape[-1] if weight.dim() == 5 else weight.shape[1])
        taps = weight.shape[1] * weight.shape[2] * weight.shape[3] if weight.dim() == 5 else 1
        flops[0] += 2.0 * T * H * W * weight.shape[0] * cin * taps
        return orig(xp, strides, weight, out, T, H, W, **kw)

    orig16 = lib.vae_conv16

    per_tap = "--per-tap" in sys.argv  # force the per-tap 16-bit kernel (flag 4) instead of the halo-tiled one

    def counted16(xp, strides, weight, out, T, H, W, **kw):
        if per_tap:
            kw["flags"] = kw.get("flags", 0) | 4
        flops[0] += 2.0 * T * H * W * weight.shape[0] * weight.shape[4] * 27
        flops16[0] += 2.0 * T * H * W * weight.shape[0] * weight.shape[4] * 27
        return orig16(xp, strides, weight, out, T, H, W, **kw)

    flops16 = [0.0]
    hunyuan_vae.lib.vae_conv = counted
    hunyuan_vae.lib.vae_conv16 = counted16
    if not full:
        m.decode(z)  # warm-up
        flops[0] = flops16[0] = 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = m.decode(z)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    print(json.dumps({"workload": f"hunyuan_vae_decode z{list(shape)} -> {list(out.shape)}", "conv_operands": "fp16" if conv16 else "fp32", "seconds": dt,
                      "conv_tflop": flops[0] / 1e12, "conv_tflop_on_fp16_kernel": flops16[0] / 1e12, "tflops_per_s": flops[0] / dt / 1e12,
                      "hbm_gb": torch.cuda.max_memory_allocated() / 1e9}))


if __name__ == "__main__":
    main()
