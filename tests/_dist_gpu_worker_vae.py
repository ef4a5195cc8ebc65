"""world_size-2 worker: WanVAE.decode with parallel=True (decode_dist: slab + halo per rank, all_gather) on one GPU through
the gloo + host-staged-collective shim of tests/_dist_gpu_worker.py; checker = the oracle's per-rank restatement."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _dist_gpu_worker import _host_staged  # noqa: E402


def main():
    dist.init_process_group("gloo")
    r = dist.get_rank()
    dist.all_gather_into_tensor = _host_staged(dist.all_gather_into_tensor)
    torch.cuda.set_device(0)
    from lightx2v_amd import lib, synth, vae
    from oracle import wan_vae_oracle as V

    lib.init(0)
    sd = synth.synth_wan_vae_weights(dim=32, seed=6)
    z = torch.randn(16, 2, 6, 10, generator=torch.Generator().manual_seed(8))  # W = 10 -> 5 per rank (+2 halo): 7 x 6 = 42 tokens (not a multiple of 16)
    mean, inv_std = torch.tensor(synth.WAN_VAE_MEAN), 1.0 / torch.tensor(synth.WAN_VAE_STD)
    out = vae.WanVAE(sd, dim=32, parallel=True).decode(z.cuda())
    with torch.no_grad():
        ref = V.wan_vae_decode_dist(sd, z, mean, inv_std, 2, 3, dim=32)
    got = out[0].float().cpu()
    assert got.shape == ref.shape == (3, 5, 48, 80), (got.shape, ref.shape)
    d = (got - ref).abs().max().item()
    assert d <= 2e-3, f"rank {r}: decode_dist max abs diff {d:.3e}"
    dist.barrier()
    if r == 0:
        print(f"DIST_GPU_VAE_OK maxabs={d:.2e}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
