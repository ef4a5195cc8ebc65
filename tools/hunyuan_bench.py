#!/usr/bin/env python
"""Times one HunyuanVideo-13B denoise step (HIP path) on a synthetic 720p x 129-frame latent (BASELINE config #5 on
one GPU: 118 800 image + 256 text tokens, 20 double + 40 single blocks) with seeded random weights.
    python tools/hunyuan_bench.py [--workload hunyuan13b_720px129f] [--steps 1] [--warmup 1]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/hunyuan_bench.py --gpus N
        (config #5 as specified: Ulysses over RCCL, latent grid split along h or w, 24 heads -> N in {1, 2, 3, 4, 6, 8})"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightx2v_amd import hunyuan as hy, lib, synth  # noqa: E402


def step_flops(d, n_img, n_txt):
    L, D, F = n_img + n_txt, d["hidden"], d["mlp"]
    dbl = 2 * L * D * (3 * D + D + 2 * F) + 4 * L * L * D       # qkv, proj, fc1, fc2 (+ joint attention)
    sgl = 2 * L * D * (3 * D + F) + 2 * L * (D + F) * D + 4 * L * L * D
    return d["double_blocks"] * dbl + d["single_blocks"] * sgl


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="hunyuan13b_720px129f")
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gpus", type=int, default=1)
    a = ap.parse_args()
    from lightx2v_amd import launch

    world, rank, local_rank = launch.ranks(__file__, a.gpus)  # bare `--gpus N`: re-runs itself as N ranks under torch.distributed.run
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch.distributed as dist

        if launch.one_gpu_test():  # plumbing mode (lightx2v_amd/launch.py): all ranks on one GPU over gloo with host-staged collectives, timings meaningless
            dist.init_process_group("gloo")
            launch.host_staged_collectives(dist)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    lib.init(local_rank)
    wl = synth.HUNYUAN_WORKLOADS[a.workload]
    dims = synth.HUNYUAN_DIMS[wl["model"]]
    cfg = hy.default_config(dims, infer_steps=50)
    wd = synth.synth_hunyuan_weights(dims, seed=0, device="cuda", gen_device="cuda")
    model = hy.HunyuanModel(cfg, wd)
    del wd
    if world > 1:
        from lightx2v_amd import ulysses

        if dims["heads"] % world:
            raise SystemExit(f"Ulysses needs heads % N == 0 ({dims['heads']} heads, N={world})")
        ulysses.parallelize_hunyuan(model)
    lat, text_states, mask, ts2 = synth.synth_hunyuan_inputs(dims, wl["target_shape"], valid_text=(dims["text_len"] * 3) // 4)
    sch = hy.HunyuanScheduler(cfg)
    sch.prepare(lat)
    model.set_scheduler(sch)
    inputs = {"text_encoder_output": {"text_encoder_1_text_states": text_states.cuda(), "text_encoder_1_attention_mask": mask.cuda(), "text_encoder_2_text_states": ts2.cuda()}}

    def one(i):
        sch.step_pre(i)
        model.infer(inputs)
        sch.step_post()

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(a.warmup):
        one(i)
    fence()
    comm_timer = None
    if world > 1:  # HIP-event accounting of the exchanges, as bench.py does for Wan (ulysses.CommTimer)
        from lightx2v_amd import ulysses

        pa = getattr(model.transformer_infer, "parallel_attention", None)
        if pa is not None and hasattr(pa, "comm_timer"):
            comm_timer = pa.comm_timer = ulysses.CommTimer()
    t0 = time.perf_counter()
    for i in range(a.steps):
        one(a.warmup + i)
    fence()
    dt = (time.perf_counter() - t0) / a.steps
    if dist is not None:  # slowest rank
        tmax = torch.tensor([dt], device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    assert torch.isfinite(sch.latents).all()
    comm = None
    if comm_timer is not None:
        comm_timer.enabled = False
        c_ms, e_ms, n_coll = comm_timer.totals_ms()
        mine = torch.tensor([c_ms / a.steps, e_ms / a.steps, n_coll / a.steps], device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(mine, op=dist.ReduceOp.MAX)
        comm = {"comm_ms_per_step": mine[0].item(), "exposed_comm_ms_per_step": mine[1].item(), "exchanges_per_step": mine[2].item()}
    _, _, t, h, w = wl["target_shape"]
    n_img = t * (h // 2) * (w // 2)
    fl = step_flops(dims, n_img, dims["text_len"])
    if rank == 0:
        print(json.dumps({"workload": a.workload, "n_gpus": world, "parallelism": f"ulysses-sp{world}" if world > 1 else "single", "tokens": n_img + dims["text_len"],
                          "ms_per_step": dt * 1e3, "step_tflop": fl / 1e12, "tflops_per_s": fl / dt / 1e12, "tflops_per_s_per_gpu": fl / dt / 1e12 / world,
                          "frac_of_bf16_peak": fl / dt / 1e12 / world / 2500.0, "frames_per_s_50_steps": wl["frames"] / (50 * dt),
                          "hbm_gb": torch.cuda.max_memory_allocated() / 1e9, "comm": comm}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
