"""world_size-2 (or -8) worker: Ulysses-sharded Hunyuan forward (all ranks on cuda:0, gloo + host-staged collectives — see
tests/_dist_gpu_worker.py) must match the single-GPU forward."""
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
    dist.all_to_all_single = _host_staged(dist.all_to_all_single)
    dist.all_gather_into_tensor = _host_staged(dist.all_gather_into_tensor)
    torch.cuda.set_device(0)
    from lightx2v_amd import hunyuan as hy, lib, synth, ulysses

    lib.init(0)
    n = dist.get_world_size()
    dims = synth.HUNYUAN_DIMS["hunyuan-tiny"]
    ts = synth.HUNYUAN_WORKLOADS["hunyuan-tiny"]["target_shape"]
    if n > 2:  # the node size on one GPU: one head per rank, a token grid whose h axis the group divides (16 rows of tokens)
        dims = dict(dims, hidden=128 * n, heads=n, mlp=2048, refiner_mlp=2048)
        ts = (1, 16, 3, 4 * n, 12)
    wd = {k: v.cuda() for k, v in synth.synth_hunyuan_weights(dims, seed=2).items()}
    lat, text_states, mask, ts2 = synth.synth_hunyuan_inputs(dims, ts, seed=5, valid_text=11)
    inputs = {"text_encoder_output": {"text_encoder_1_text_states": text_states.cuda(), "text_encoder_1_attention_mask": mask.cuda(), "text_encoder_2_text_states": ts2.cuda()}}
    outs = {}
    for mode in ("single", "ulysses"):
        cfg = hy.default_config(dims, infer_steps=4)
        model = hy.HunyuanModel(cfg, wd)
        sch = hy.HunyuanScheduler(cfg)
        sch.prepare(lat)
        model.set_scheduler(sch)
        if mode == "ulysses":
            ulysses.parallelize_hunyuan(model)
        sch.step_pre(1)
        model.infer(inputs)
        outs[mode] = sch.noise_pred.float().cpu()
        if mode == "ulysses":
            pa = model.transformer_infer.parallel_attention
            assert pa.copies == 0 and pa._buffers, "the fused driver must take the copy-free blocked exchange path (double and single blocks)"
        assert sch.latents.shape == lat.shape
    a, b = outs["single"], outs["ulysses"]
    assert a.shape == b.shape
    rel = ((a - b).norm() / a.norm()).item()
    assert rel < 5e-3, f"rank {r}: ulysses vs single-GPU relative L2 {rel:.3e}"
    dist.barrier()
    if r == 0:
        print(f"DIST_GPU_HUNYUAN_OK rel={rel:.2e}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
