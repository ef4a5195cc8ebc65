#!/usr/bin/env python
"""End-to-end wall clock of the hot path on one GPU: the full denoise loop (all infer_steps, CFG) followed by the VAE decode of the
final latents — the quantity BASELINE.json's north star asks for next to the per-step number ("end-to-end wall-clock and frames/sec").
Synthetic weights and inputs of the named shape (no text encoder: its output is an input here, as in bench.py).  One JSON line.
    python tools/e2e.py [--workload wan14b_720px81f] [--steps 50] [--fp8|--mxfp8] [--distill] [--teacache T]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/e2e.py --gpus N ...
        (N GPUs of one node: Ulysses sequence parallel denoise loop over RCCL + the halo-split `decode_dist` VAE decode)"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightx2v_amd import lib, scheduler, synth, vae, wan  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="wan14b_720px81f")
    ap.add_argument("--steps", type=int, default=0, help="0 = the workload's own schedule length (50; 40 for the i2v benchmark workloads)")
    ap.add_argument("--i2v", action="store_true", help="--workload wan14b_i2v_720px81f: the reference's published benchmark (I2V-14B, 40 steps, CFG 5, shift 5; configs/bench/lightx2v_2.json)")
    ap.add_argument("--fp8", action="store_true")
    ap.add_argument("--mxfp8", action="store_true")
    ap.add_argument("--distill", action="store_true", help="4-step distilled schedule, no CFG (BASELINE config #4)")
    ap.add_argument("--vae16", action="store_true", help="fastest VAE decode: convolution operands rounded to fp16")
    ap.add_argument("--vae32", action="store_true", help="VAE convolutions on the fp32 matrix instruction (default: hi/lo fp16 split, fp32-grade)")
    ap.add_argument("--teacache", type=float, default=0.0, help="TeaCache threshold (0 = off); uses the released 14B 720p coefficients")
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
    if a.i2v:
        a.workload = "wan14b_i2v_720px81f"
    wl = synth.WORKLOADS[a.workload]
    dims = synth.WAN_DIMS[wl["model"]]
    steps = 4 if a.distill else (a.steps or wl.get("infer_steps", 50))
    extra = {}
    if a.fp8:
        extra["mm_config"] = {"mm_type": "W-fp8-channel-sym-A-fp8-channel-sym-dynamic-Hip", "weight_auto_quant": True}
    if a.mxfp8:
        extra["mm_config"] = {"mm_type": "W-mxfp8-A-mxfp8-dynamic-Hip", "weight_auto_quant": True}
    if a.distill:
        extra.update(enable_cfg=False, denoising_step_list=[1000, 750, 500, 250], sample_shift=5.0)
    if a.teacache > 0:
        # configs/caching/teacache/wan_t2v_tea_720p.json of the reference: coefficients for Wan2.1-T2V-14B 720p
        extra.update(feature_caching="Tea", teacache_thresh=a.teacache, use_ret_steps=False,
                     coefficients=[[8.10705460e03, 2.13393892e03, -3.72934672e02, 1.66203073e01, -4.17769401e-02],
                                   [-114.36346466, 65.26524496, -18.82220707, 4.91518089, -0.23412683]])
    if world > 1:
        if dims["num_heads"] % world:
            raise SystemExit(f"Ulysses needs num_heads % N == 0 ({dims['num_heads']} heads, N={world})")
        extra["parallel_attn_type"] = "ulysses"
    _, overrides, wd, lat, inputs = synth.workload_setup(a.workload, seed=0, device="cuda")  # i2v: the i2v checkpoint + seeded CLIP / VAE-encode stand-ins
    extra = {**overrides, **extra}
    cfg = wan.default_config(dims, target_shape=wl["target_shape"], target_video_length=wl["frames"], infer_steps=steps, **extra)
    model = wan.WanModel(cfg, wd)
    del wd
    sch = (scheduler.WanStepDistillScheduler if a.distill else scheduler.WanScheduler)(cfg, device="cuda")
    sch.prepare(latents=lat)
    model.set_scheduler(sch)
    decoder = vae.WanVAE(synth.synth_wan_vae_weights(dim=96, seed=0), dim=96, conv16=(True if a.vae16 else False if a.vae32 else "split"), parallel=world > 1)
    # warm-up outside the clock: one step on a scratch scheduler state (allocator pools, lazy tables) and a short decode
    sch.step_pre(0)
    model.infer(inputs)
    decoder.decode(torch.zeros(16, 2, wl["target_shape"][2], wl["target_shape"][3], device="cuda"))
    sch.reset() if hasattr(sch, "reset") else None
    sch.prepare(latents=lat)
    if a.teacache > 0:
        model.transformer_infer.cnt = 0

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    scheduler.run_denoise_loop(model, sch, inputs)
    fence()
    t1 = time.perf_counter()
    video = decoder.decode(sch.latents.float())
    fence()
    t2 = time.perf_counter()
    assert torch.isfinite(video).all() and torch.isfinite(sch.latents).all()
    frames = wl["frames"]
    rec = {"workload": a.workload, "task": dims.get("task", "t2v"), "guide_scale": cfg["sample_guide_scale"], "sample_shift": cfg["sample_shift"], "n_gpus": world, "parallelism": f"ulysses-sp{world} + decode_dist" if world > 1 else "single", "steps": steps, "cfg": bool(cfg["enable_cfg"]), "gemm_dtype": "mxfp8" if a.mxfp8 else "fp8" if a.fp8 else "bf16",
           "teacache_thresh": a.teacache, "vae_conv_operands": "fp16" if a.vae16 else "fp32" if a.vae32 else "fp16 hi/lo split (fp32-grade)", "denoise_s": t1 - t0, "ms_per_step": (t1 - t0) * 1e3 / steps, "vae_decode_s": t2 - t1, "total_s": t2 - t0,
           "frames": frames, "video_shape": list(video.shape), "fps_denoise_only": frames / (t1 - t0), "fps_with_vae": frames / (t2 - t0),
           "hbm_gb_peak": torch.cuda.max_memory_allocated() / 1e9, "data": "synthetic weights / latents / text embeddings"}
    if a.teacache > 0:
        rec_c, rec_u = list(getattr(sch, "caching_records", [])), list(getattr(sch, "caching_records_2", []))
        rec["teacache_forwards_computed"] = int(sum(bool(v) for v in rec_c + rec_u))
        rec["teacache_forwards_total"] = len(rec_c) + len(rec_u)
    if rank == 0:
        print(json.dumps(rec))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
