#!/usr/bin/env python
"""Times WanVAE.decode (HIP path) on a synthetic latent of the BASELINE shape (720p x 81 frames: z [16,21,90,160]) and
reports the convolution kernel's algorithmic TFLOP/s (2*T*H*W*Cout*Cin*taps per launch, summed over launches).
    python tools/vae_bench.py [--latent 16,21,90,160] [--reps 1]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightx2v_amd import lib, synth, vae  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--latent", default="16,21,90,160")
    ap.add_argument("--reps", type=int, default=1)
    ap.add_argument("--conv16", action="store_true", help="opt-in fp16 convolution operands")
    ap.add_argument("--split", action="store_true", help="hi/lo fp16 split of the convolution operands (fp32-grade; WanVAE's default)")
    ap.add_argument("--chunk-frames", type=int, default=4, help="latent frames per decoder pass (1 = the reference's chunking)")
    a = ap.parse_args()
    shape = tuple(int(v) for v in a.latent.split(","))
    lib.init(0)
    sd = synth.synth_wan_vae_weights(dim=96, seed=0)
    m = vae.WanVAE(sd, dim=96, conv16=("split" if a.split else a.conv16), chunk_frames=a.chunk_frames)
    z = torch.randn(*shape, generator=torch.Generator().manual_seed(5)).cuda()
    flops = [0.0]
    orig = lib.vae_conv

    def counted(xp, strides, weight, out, T, H, W, **kw):
        cin = kw.get("cin") or (weight.shape[-1] if weight.dim() == 5 else weight.shape[1])
        taps = weight.shape[1] * weight.shape[2] * weight.shape[3] if weight.dim() == 5 else 1
        flops[0] += 2.0 * T * H * W * weight.shape[0] * cin * taps
        return orig(xp, strides, weight, out, T, H, W, **kw)

    orig16 = lib.vae_conv16

    def counted16(xp, strides, weight, out, T, H, W, **kw):
        cin = weight.shape[4] - (32 if kw.get("flags", 0) & lib.VCONV_ZERO_TAIL32 else 0)  # channels multiplied (the 32-channel-slab kernel skips a zero tail)
        flops[0] += 2.0 * T * H * W * weight.shape[0] * cin * weight.shape[1] * weight.shape[2] * weight.shape[3]
        return orig16(xp, strides, weight, out, T, H, W, **kw)

    lib.vae_conv = counted
    vae.lib.vae_conv = counted
    vae.lib.vae_conv16 = counted16
    out = m.decode(z)  # warm-up: allocates every buffer
    torch.cuda.synchronize()
    f1 = flops[0]
    t0 = time.perf_counter()
    for _ in range(a.reps):
        out = m.decode(z)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    assert torch.isfinite(out).all()
    print(json.dumps({"workload": f"wan_vae_decode z{list(shape)} -> {list(out.shape)}", "conv_operands": "fp16 hi/lo split" if a.split else "fp16" if a.conv16 else "fp32", "chunk_frames": a.chunk_frames, "seconds": dt, "conv_tflop": f1 / 1e12, "tflops_per_s": f1 / dt / 1e12,
                      "frac_of_fp32_mfma_peak_157": f1 / dt / 1e12 / 157.3, "hbm_gb_allocated": torch.cuda.max_memory_allocated() / 1e9}))


if __name__ == "__main__":
    main()
