"""`randn_tensor` stand-in: seeded normal noise (the fixtures feed latents from a file, so its stream is never compared)."""
import torch


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    g = generator[0] if isinstance(generator, (list, tuple)) else generator
    return torch.randn(tuple(shape), generator=g, dtype=dtype or torch.float32).to(device or "cpu")
