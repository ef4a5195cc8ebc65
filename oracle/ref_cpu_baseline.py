"""TEST / MEASUREMENT INFRASTRUCTURE ONLY — times the *unmodified* reference (ModelTC/lightx2v, imported through oracle/ref_import.py) on this
host's CPU cores for BASELINE config #1 (Wan2.1-T2V-1.3B bf16, 256x256x17f = 1280 tokens, 4 CFG steps: the reference's own CPU-runnable case,
SURVEY.md §8d "How the reference CPU path is timed beside it").  `bench.py`'s `cpu_baseline` leg runs this file as a SUBPROCESS with the GPUs
hidden (HIP_VISIBLE_DEVICES=""), so that ref_import's GPU-less patches apply and nothing of the reference touches the device; nothing in the
product path imports it.

What runs is the reference's own code, object for object: `WanModel` (models/networks/wan/model.py:28-226: its `_load_ckpt` reads a safetensors
checkpoint of the seeded synthetic weights from a temp dir, its weight trees, `WanPreInfer` / `WanTransformerInfer` / `WanPostInfer`, its
`infer` with the fp32 CFG combine) and `WanScheduler` (models/schedulers/wan/scheduler.py), driven by the loop of
`DefaultRunner.run` (models/runners/default_runner.py:97-114) with `time.perf_counter()` at the three boundaries its `ProfilingContext4Debug`
puts there (utils/profiler.py:17-33; `torch.cuda.synchronize` is a no-op on CPU).  Config: DTYPE=BF16, `mm_config = {}` ("Default": torch.addmm),
`torch_sdpa` attention — the only reference configuration that needs no third-party kernel package.

The thread count is chosen by the clock, not assumed: one conditional forward per candidate count (the first one after an untimed warm-up
forward), the fastest one runs the 4 steps.  (Round 4's baseline ran at torch's default of one thread per hardware thread — 128 on the GPU box —
and was 3-4x slower per step than 8 threads in the authoring container: oversubscribed small bf16 matmuls.)

Prints ONE JSON line.  Usage:  HIP_VISIBLE_DEVICES= python -m oracle.ref_cpu_baseline [--threads 8,16,32,64] [--steps 4]
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="", help="comma-separated candidate thread counts (default: 8, 16, 32, 64 capped at the host's count)")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--tiny", action="store_true", help="the wan-tiny plumbing model instead of config #1 (tests/test_bench_accounting.py pins the script's latents to tests/golden with it)")
    args = ap.parse_args()

    import torch

    if torch.cuda.is_available():
        raise SystemExit("oracle.ref_cpu_baseline must run with the GPUs hidden (HIP_VISIBLE_DEVICES=\"\"): it is the CPU baseline")
    from safetensors.torch import save_file

    from lightx2v_amd import synth
    from oracle import ref_import

    if not ref_import.reference_available():
        raise SystemExit("no reference checkout (neither /root/reference nor oracle/_ref/reference)")
    ref_import.patch_and_import()
    from lightx2v.models.networks.wan.model import WanModel
    from lightx2v.models.schedulers.wan.scheduler import WanScheduler

    host_threads = os.cpu_count() or 8
    cands = [int(t) for t in args.threads.split(",") if t] or [t for t in (8, 16, 32, 64) if t <= host_threads] or [host_threads]
    dims = synth.WAN_DIMS["wan-tiny" if args.tiny else "wan2.1-1.3b"]
    wl = synth.WORKLOADS["wan-tiny" if args.tiny else "wan1.3b_256x256x17f"]
    ts = wl["target_shape"]
    wd = synth.synth_wan_weights(dims, seed=0)
    lat, ctx, ctx_null = synth.synth_inputs(dims, ts)
    with tempfile.TemporaryDirectory() as ckpt:
        save_file({k: v.contiguous() for k, v in wd.items()}, os.path.join(ckpt, "model.safetensors"))
        del wd
        cfg = ref_import.make_config(dims, target_shape=ts, target_video_length=wl["frames"], infer_steps=args.steps, model_path=ckpt)
        model = WanModel(ckpt, cfg, torch.device("cpu"))
    inputs = {"text_encoder_output": {"context": ctx, "context_null": ctx_null}}

    def fresh_scheduler():
        sch = WanScheduler(cfg)
        sch.device = torch.device("cpu")  # hard-coded "cuda" at wan/scheduler.py:12
        sch.prepare()
        sch.latents = lat.clone()  # the CPU noise stream of synth_inputs, the same the GPU legs use
        model.set_scheduler(sch)
        return sch

    # thread sweep: one conditional+unconditional model.infer per candidate on a throw-away scheduler
    sweep = {}
    sch = fresh_scheduler()
    sch.step_pre(step_index=0)
    torch.set_num_threads(cands[0])
    model.infer(inputs)  # untimed: first touch of the weights, allocator warm-up
    for t in cands:
        torch.set_num_threads(t)
        t0 = time.perf_counter()
        model.infer(inputs)
        sweep[t] = time.perf_counter() - t0
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)

    # the timed run: default_runner.py:97-114
    sch = fresh_scheduler()
    marks = {"step_pre": [], "infer": [], "step_post": []}
    t_run = time.perf_counter()
    for step_index in range(sch.infer_steps):
        a = time.perf_counter()
        sch.step_pre(step_index=step_index)
        b = time.perf_counter()
        model.infer(inputs)
        c = time.perf_counter()
        sch.step_post()
        d = time.perf_counter()
        marks["step_pre"].append(b - a)
        marks["infer"].append(c - b)
        marks["step_post"].append(d - c)
    total = time.perf_counter() - t_run
    assert torch.isfinite(sch.latents).all()
    per_step = [p + i + q for p, i, q in zip(marks["step_pre"], marks["infer"], marks["step_post"])]
    print(json.dumps({
        "kind": "reference",
        "reference_root": ref_import.REFERENCE_ROOT,
        "workload": "wan-tiny (plumbing)" if args.tiny else "BASELINE config #1: Wan2.1-T2V-1.3B bf16, 256x256x17f (1280 tokens), 4 steps, CFG",
        "frames": wl["frames"],
        "total_s": total,
        "frames_per_s": wl["frames"] / total,
        "ms_per_step": [round(x * 1e3, 1) for x in per_step],
        "ms_per_step_median": sorted(per_step)[len(per_step) // 2] * 1e3,
        "phases_ms": {k: [round(x * 1e3, 2) for x in v] for k, v in marks.items()},
        "threads": best,
        "host_threads": host_threads,
        "thread_sweep_s_per_infer": {str(k): round(v, 3) for k, v in sweep.items()},
        "latents_abs_sum": float(sch.latents.double().abs().sum()),
    }), flush=True)


if __name__ == "__main__":
    main()
