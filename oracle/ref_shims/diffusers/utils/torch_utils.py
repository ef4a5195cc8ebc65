"""`randn_tensor` stand-in: seeded normal noise (the fixtures feed latents from a file, so its stream is never compared)."""
import torch


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    g = generator[0] if isinstance(generator, (list, tuple)) else generator
    draw_on = g.device if g is not None else "cpu"  # a generator only draws on its own device (the reference's Hunyuan scheduler seeds a cuda one)
    return torch.randn(tuple(shape), generator=g, dtype=dtype or torch.float32, device=draw_on).to(device or "cpu")
