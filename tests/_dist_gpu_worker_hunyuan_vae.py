"""world_size-2 worker: tile-parallel HunyuanVideo VAE decode (AutoencoderKLCausal3D with a process group: each rank decodes its share
of the independent tiles, tiles are broadcast from their owners, blending is replicated) on one GPU through the gloo + host-staged
shim; checker = the same decode on one rank, which it must reproduce bit for bit."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _host_staged_bcast(fn):
    def wrapped(t, src=0, group=None, **kw):
        if t.is_cuda:
            torch.cuda.current_stream().synchronize()
            h = t.detach().cpu()
            r = fn(h, src=src, group=group, **kw)
            t.copy_(h)
            return r
        return fn(t, src=src, group=group, **kw)

    return wrapped


def main():
    dist.init_process_group("gloo")
    r = dist.get_rank()
    dist.broadcast = _host_staged_bcast(dist.broadcast)
    torch.cuda.set_device(0)
    from lightx2v_amd import hunyuan_vae, lib, synth

    lib.init(0)
    cfg = synth.HUNYUAN_VAE_TINY_CFG
    sd = synth.synth_hunyuan_vae_weights(cfg, seed=1)
    z = (torch.randn(1, 16, 6, 12, 10, generator=torch.Generator().manual_seed(3)) * 0.5).cuda()  # 2 temporal x 2x2 spatial tiles = 8 jobs
    single = hunyuan_vae.VideoEncoderKLCausal3DModel(sd, cfg)
    shared = hunyuan_vae.VideoEncoderKLCausal3DModel(sd, cfg, group=dist.group.WORLD)
    ref = single.decode(z)
    calls = [0]
    fwd = shared.model.decoder.forward

    def counting(x):
        calls[0] += 1
        return fwd(x)

    shared.model.decoder.forward = counting
    got = shared.decode(z)
    assert calls[0] == 4, f"rank {r} decoded {calls[0]} of 8 tiles (expected its half)"
    assert got.shape == ref.shape == (1, 3, 21, 96, 80)
    assert torch.equal(got, ref), f"rank {r}: tile-parallel decode differs from the single-rank decode (max abs {(got - ref).abs().max().item():.3e})"
    dist.barrier()
    if r == 0:
        print("DIST_GPU_HUNYUAN_VAE_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
