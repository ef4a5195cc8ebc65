#!/usr/bin/env python
"""bench.py — denoise-step latency and video frames/sec of the Wan2.1 DiT hot path on MI355X.

  python bench.py --gpus 1 --steps K --warmup W                     (1 GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
                                                                     (Ulysses sequence parallel over RCCL/xGMI)

A "step" = one full denoise step of the reference loop (default_runner.py:97-114): scheduler.step_pre →
WanModel.infer (conditional + unconditional forward, fp32 CFG combine) → scheduler.step_post, on synthetic
latents/context of the BASELINE shape with seeded random weights of the named architecture (no checkpoints or
prompts exist offline).  Everything is resident in HBM before the timed region.

Workload (config.workload): "wan14b_720px81f" = Wan2.1-T2V-14B bf16, 720p x 81 frames — the configuration
BASELINE.json's metric is quoted on; it fits one 288 GB MI355X (28 GB weights + <10 GB activations).
N > 1 shards the SAME video across ranks (strong scaling).

Output: ONE JSON line on rank 0.  value = video frames/sec = frames / (infer_steps x step latency), denoise
loop only (VAE decode excluded).  `roofline` is for the dominant kernel (self-attention forward:
72 % of the step's FLOPs at 720p); `cpu_baseline` times the reference's CPU path on this box's host cores:
BASELINE config #1 end to end by the UNMODIFIED reference (kind "reference"; the oracle port where no reference checkout travels with the tree), plus one
block of the BENCHED workload at its full sequence length on a bounded row sample by the oracle port beside it.
"""
import argparse
import contextlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="wan14b_720px81f")
    ap.add_argument("--infer-steps", type=int, default=0, help="length of the full denoise schedule (frames/sec denominator); 0 = the workload's own (50; 40 for the i2v benchmark workloads)")
    ap.add_argument("--i2v", action="store_true", help="shorthand for --workload wan14b_i2v_720px81f (the reference's published benchmark: I2V-14B, 40 steps, CFG)")
    ap.add_argument("--no-cfg", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-rows", type=int, default=256, help="query rows of the benched block the CPU baseline evaluates (scaled by S / rows)")
    ap.add_argument("--no-cpu-config1", action="store_true", help="skip the end-to-end CPU run of BASELINE config #1 beside the baseline (~80 s on 128 threads)")
    ap.add_argument("--ref-rounding", action="store_true", help="norm kernels reproduce the reference's bf16 rounding chain")
    ap.add_argument("--no-cfg-pair", action="store_true", help="run the conditional and unconditional forwards of a step separately (default: by size)")
    ap.add_argument("--cfg-streams", action="store_true", help="one GPU, CFG: force the two forwards block by block on two compute streams (wan.CfgBranchStreams; default: by size — on for 1.3B 480p)")
    ap.add_argument("--no-cfg-streams", action="store_true", help="never put the two CFG forwards on two compute streams")
    ap.add_argument("--cfg-pair", action="store_true", help="force the one-pass form of the two CFG forwards (default: by size — on for 14B 720p, off for 1.3B 480p)")
    ap.add_argument("--mxfp8", action="store_true", help="MXFP8 GEMMs (e4m3 + e8m0 per 32 K, weights and activations; gfx950 block-scaled MFMA), bf16 attention")
    ap.add_argument("--fp8", action="store_true", help="w8a8 e4m3 GEMMs (BASELINE config #4): weights auto-quantised per channel at load, per-token dynamic activations")
    ap.add_argument("--hang-timeout", type=int, default=0, help="N > 1: seconds without progress after which the watchdog dumps the stage it is stuck in and exits 124 (0 = 240 for N > 1, off for N = 1)")
    ap.add_argument("--no-calibration", action="store_true", help="skip the in-process MFMA probe before / after the timed region and the power / clock samples")
    ap.add_argument("--probe-ms", type=int, default=1500, help="duration of each box-calibration probe (lib.mfma_probe)")
    ap.add_argument("--distill", action="store_true", help="4-step distilled schedule of config #4 (no CFG, denoising_step_list 1000/750/500/250, shift 5)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the untimed-for-`value` legs behind the headline steps (BASELINE configs #2, #4, #5 on one GPU and the Wan VAE decode; `other_configs`)")
    ap.add_argument("--other-configs", action="store_true", help="run the `other_configs` legs even when the headline workload / flags are not the default ones")
    return ap.parse_args()


def step_flops(dims, S, text_len, cfg_forwards, cross_kv_cached=True):
    """Algorithmic FLOPs of one denoise step: per block GEMM = 12 S D^2 + 4 Lc D^2 + 4 S D F,
    ATTN = 4 S^2 D + 4 S Lc D.  The 4 Lc D^2 term (cross-attention K/V projections of the text) is step-invariant: with `cache_cross_kv`
    (the default) it is computed once per prompt and is NOT part of a timed step, so it is not counted."""
    D, F, L = dims["dim"], dims["ffn_dim"], dims["num_layers"]
    gemm = 12 * S * D * D + (0 if cross_kv_cached else 4 * text_len * D * D) + 4 * S * D * F
    attn = 4 * S * S * D + 4 * S * text_len * D
    return L * (gemm + attn) * cfg_forwards, L * attn * cfg_forwards


class AttnTimer:
    """HIP-event timing of every attention-kernel launch (self and cross attention run the SAME kernel, so the
    per-kernel average matches what `rocprofv3 --kernel-trace --stats` reports for it), on the stream the kernel
    is launched on (torch's current stream)."""

    def __init__(self):
        self.pairs = {"self": [], "cross": []}
        self.enabled = False

    @contextlib.contextmanager
    def __call__(self, kind):
        if not self.enabled:
            yield
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        yield
        b.record()
        self.pairs[kind].append((a, b))

    def total_ms(self, kind):
        return sum(a.elapsed_time(b) for a, b in self.pairs[kind])

    def count(self, kind):
        return len(self.pairs[kind])


class Watchdog:
    """First-contact insurance for the multi-GPU run: a collective that RCCL rejects or that hangs must cost minutes, not the
    driver's whole 1800 s.  `tick(stage)` marks progress; a daemon thread exits the process with code 124 — after printing the stage, the rank and
    every Python thread's stack to stderr — when no tick arrived for `limit` seconds.  (The process-group timeout passed to init_process_group makes
    RCCL's own watchdog abort a stuck collective on the same scale; this one also covers a hang outside a collective.)"""

    def __init__(self, limit, rank):
        import threading

        self.limit, self.rank, self.stage, self.t = limit, rank, "start", time.monotonic()
        if limit > 0:
            threading.Thread(target=self._run, daemon=True).start()

    def tick(self, stage):
        self.stage, self.t = stage, time.monotonic()

    def _run(self):
        import faulthandler

        while True:
            time.sleep(2.0)
            idle = time.monotonic() - self.t
            if idle > self.limit:
                print(f"bench watchdog: rank {self.rank} made no progress for {idle:.0f} s in stage '{self.stage}' — giving up (exit 124)", file=sys.stderr, flush=True)
                faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                os._exit(124)


class SmiSampler:
    """Board power / engine clock / temperature of this rank's GPU during the timed region, read by `rocm-smi --json` from a host thread every
    `period` seconds (rank 0, N = 1 only).  Context for a reader comparing two boxes: the matrix kernels run at the 1400 W board limit, and what
    differs between boxes is the clock they are granted there.  Never fails the bench: any error leaves `summary()` empty."""

    def __init__(self, device_index, period=5.0):
        import threading

        self.dev, self.period, self.rows = device_index, period, []
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _read(self):
        import subprocess

        p = subprocess.run(["rocm-smi", "-d", str(self.dev), "--showpower", "--showclocks", "--showtemp", "--json"], capture_output=True, text=True, timeout=10)
        d = json.loads(p.stdout)
        card = d.get(f"card{self.dev}") or next(iter(d.values()))
        row = {}
        for k, v in card.items():
            kl = k.lower()
            name = "power_w" if "power" in kl and "socket" in kl else "sclk_mhz" if kl.startswith("sclk") else "mclk_mhz" if kl.startswith("mclk") else "temp_c" if "temperature" in kl and ("hotspot" in kl or "junction" in kl) else None
            if name is None or name in row:
                continue
            num = "".join(ch for ch in str(v).replace("Mhz", "").replace("MHz", "") if ch.isdigit() or ch == ".")
            if num:
                row[name] = float(num)
        return row

    def _run(self):
        while not self._stop.is_set():
            try:
                row = self._read()
                if row:
                    self.rows.append(row)
            except Exception:  # noqa: BLE001
                pass
            self._stop.wait(self.period)

    def __enter__(self):
        self._thread.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        self._thread.join(timeout=15)

    def summary(self):
        out = {"samples": len(self.rows)}
        for key in ("power_w", "sclk_mhz", "mclk_mhz", "temp_c"):
            vals = [r[key] for r in self.rows if key in r]
            if vals:
                out[key] = {"mean": sum(vals) / len(vals), "min": min(vals), "max": max(vals)}
        return out


def _smi_index(local_rank):
    """rocm-smi numbers the node's physical devices; HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES renumber what this process sees."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        ids = [t.strip() for t in os.environ.get(var, "").split(",") if t.strip()]
        if ids and local_rank < len(ids) and ids[local_rank].isdigit():
            return int(ids[local_rank])
    return local_rank


def reference_cpu_run(timeout_s=600):
    """BASELINE config #1 by the UNMODIFIED reference on this box's host cores: oracle/ref_cpu_baseline.py in a subprocess with the GPUs hidden
    (its docstring has the what and how).  Returns its JSON record, or None plus the reason where no reference checkout travels with the tree."""
    import subprocess

    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        p = subprocess.run([sys.executable, "-m", "oracle.ref_cpu_baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout_s)
    except Exception as e:  # noqa: BLE001
        return None, f"{type(e).__name__}: {str(e)[:200]}"
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        return None, f"rc {p.returncode}: {(p.stderr or p.stdout).strip()[-300:]}"
    return json.loads(lines[-1]), None


def cpu_baseline(dims, S_full, ts, text_len, frames, infer_steps, cfg_forwards, n_rows, with_config1=True):
    """The reference's CPU path timed on this box's host cores (baseline only — never the thing shipped, never the target).

    `value` / `kind: "reference"` = BASELINE config #1 (the reference's own CPU-runnable case: Wan2.1-1.3B, 256x256x17f, 4 CFG steps) run end to end
    by the UNMODIFIED reference (its WanModel + WanScheduler in DefaultRunner.run's loop, perf_counter at default_runner.py:102-109's boundaries;
    SURVEY.md §8d) at the thread count that minimises its step time — `reference_cpu_run` above.  It is a DIFFERENT workload than the GPU line's
    `value` and must never be divided into it; where no reference checkout travels with the tree the same loop runs on the oracle
    (oracle/wan_oracle.py, pinned bit-exactly to the reference) and `kind` says "port".

    `benched_workload_estimate` (always the port: the unmodified reference cannot evaluate a row sample) = the BENCHED workload (same architecture,
    same token count): ONE block at the full sequence length on a bounded row sample — the part of the block that needs all rows (norm1 + modulate,
    k / v projections, norm_k, RoPE(k): `all_rows_s`) runs in full, everything row-wise (q, the attention of the sampled queries against ALL keys,
    cross-attention, projections, FFN) on `n_rows` rows and is scaled by S / n_rows; step = layers x cfg forwards x block.  No FLOP-ratio
    extrapolation across sequence lengths (attention is quadratic in S)."""
    from lightx2v_amd import synth
    from oracle import wan_oracle as O

    host_threads = torch.get_num_threads()
    est_threads = host_threads  # 14B-sized matmuls at 75 600 rows: torch's default (one thread per hardware thread) is the fast setting here
    dd = dict(dims, num_layers=1)
    wd = synth.synth_wan_weights(dd, seed=1)
    g = torch.Generator().manual_seed(0)
    grid = (ts[1], ts[2] // 2, ts[3] // 2)
    x = torch.randn(S_full, dims["dim"], generator=g).to(torch.bfloat16)
    embed0 = (torch.randn(6, dims["dim"], generator=g) * 0.1).to(torch.bfloat16)
    context = torch.randn(text_len, dims["dim"], generator=g).to(torch.bfloat16)
    rows = torch.randperm(S_full, generator=g)[:n_rows]
    freqs = O.rope_freqs_table(128)
    tm = {}
    t0 = time.perf_counter()
    O.wan_block_rows(wd, 0, dd, grid, x, embed0, freqs, context, rows, timing=tm)
    sample_s = time.perf_counter() - t0
    block_s = tm["all_rows_s"] + tm["sampled_rows_s"] * S_full / n_rows
    step_s = block_s * dims["num_layers"] * cfg_forwards
    flop_step, _ = step_flops(dims, S_full, text_len, cfg_forwards, cross_kv_cached=False)
    estimate = {
        "value": frames / (infer_steps * step_s),
        "unit": "frames/s",
        "cores": est_threads,
        "kind": "port",
        "ms_per_step_est": step_s * 1e3,
        "tflops_per_s": flop_step / step_s / 1e12,
        "sample": f"oracle (reference CPU path restated, pinned bit-exactly to the reference): one "
        f"{dims['dim']}d/{dims['num_heads']}h block at the benched S={S_full}: all-rows part (norm1, k/v projections, norm_k, RoPE) in full {tm['all_rows_s']:.1f} s + "
        f"{n_rows} sampled rows of the row-wise part (q, self-attention vs all {S_full} keys, cross-attention, o, FFN) {tm['sampled_rows_s']:.1f} s x {S_full}/{n_rows}; "
        f"step = {dims['num_layers']} layers x {cfg_forwards} forwards x {block_s:.0f} s; {sample_s:.0f} s of CPU work on {est_threads} threads",
    }
    del wd, x
    if not with_config1:
        return estimate
    ref, why = reference_cpu_run()
    if ref is not None:
        return {
            "value": ref["frames_per_s"],
            "unit": "frames/s",
            "cores": ref["threads"],
            "kind": "reference",
            "comparable_to_value": False,  # BASELINE config #1, not the benched workload: never divide the line's `value` by this one
            "same_workload_value": estimate["value"],  # the benched workload on these host cores (port, extrapolated from a row sample: `benched_workload_estimate`)
            "sample": f"the UNMODIFIED reference ({ref['reference_root']}: WanModel + WanScheduler in DefaultRunner.run's loop, DTYPE=BF16, Default mm, torch_sdpa; "
            f"oracle/ref_cpu_baseline.py) on {ref['workload']} — the reference's own CPU-runnable configuration, a DIFFERENT workload than this line's `value`; "
            f"{ref['total_s']:.1f} s on {ref['threads']} of {ref['host_threads']} host threads (fastest of the swept counts)",
            "workload": ref["workload"],
            "total_s": ref["total_s"],
            "ms_per_step": ref["ms_per_step"],
            "ms_per_step_median": ref["ms_per_step_median"],
            "phases_ms": ref["phases_ms"],
            "host_threads": ref["host_threads"],
            "thread_sweep_s_per_infer": ref["thread_sweep_s_per_infer"],
            "benched_workload_estimate": estimate,
        }
    # no reference checkout on this box: the same loop on the oracle (1280-token matmuls: oversubscribed at one thread per hardware thread,
    # round 4 measured 19-28 s/step on 128 threads against ~6 s on 8)
    est_threads = min(host_threads, 32)
    torch.set_num_threads(est_threads)
    d1 = synth.WAN_DIMS["wan2.1-1.3b"]
    wl1 = synth.WORKLOADS["wan1.3b_256x256x17f"]
    wd = synth.synth_wan_weights(d1, seed=0)
    lat, ctx, ctx_null = synth.synth_inputs(d1, wl1["target_shape"])
    step_t = []
    t0 = time.perf_counter()
    O.denoise_loop(wd, d1, lat, ctx, ctx_null, 4, 8.0, 6.0, step_callback=lambda i, x: step_t.append(time.perf_counter()))
    total = time.perf_counter() - t0
    torch.set_num_threads(host_threads)
    per_step = [b - a for a, b in zip([t0] + step_t[:-1], step_t)]
    return {
        "value": wl1["frames"] / total,
        "unit": "frames/s",
        "cores": est_threads,
        "kind": "port",
        "comparable_to_value": False,
        "same_workload_value": estimate["value"],
        "sample": f"oracle (the reference's CPU path restated, pinned bit-exactly to it; no reference checkout on this box: {why}) on BASELINE config #1: Wan2.1-T2V-1.3B bf16, "
        f"256x256x17f (1280 tokens), 4 steps, CFG — a DIFFERENT workload than this line's `value`; {total:.1f} s on {est_threads} threads",
        "workload": "BASELINE config #1: Wan2.1-T2V-1.3B bf16, 256x256x17f (1280 tokens), 4 steps, CFG",
        "total_s": total,
        "ms_per_step": [round(x * 1e3, 1) for x in per_step],
        "benched_workload_estimate": estimate,
    }


def _attn_roofline(timer, steps, S, heads, layers, fwd):
    """Self-attention launches of one sub-run as a roofline object (the dominant kernel of every configuration here): algorithmic FLOPs per launch
    4 S^2 heads 128 x forwards per launch over the HIP-event average on the launch stream."""
    n_self, ms_self = timer.count("self"), timer.total_ms("self")
    if not n_self:
        return None
    forwards_per_launch = layers * fwd / (n_self / max(1, steps))
    avg_ms = ms_self / n_self
    achieved = 4.0 * S * S * heads * 128 * forwards_per_launch / (avg_ms * 1e-3) / 1e12
    return {"kernel": "x2v::attn_fwd self-attention launches (x2v_attn_fwd_bf16_vt[_batched])", "bound": "mfma", "achieved": achieved, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / BF16_MFMA_PEAK_TFLOPS, "avg_launch_ms": avg_ms, "launches_timed": n_self, "forwards_per_launch": forwards_per_launch}


def run_wan_config(workload, steps, warmup, fp8=False, distill=False):
    """One more BASELINE configuration behind the headline steps (`other_configs`): build the named Wan workload, `warmup` untimed + `steps` timed
    denoise steps bracketed by synchronize(), the self-attention launches timed by HIP events as in the headline run."""
    from lightx2v_amd import scheduler, synth, wan

    wl = synth.WORKLOADS[workload]
    dims = synth.WAN_DIMS[wl["model"]]
    ts = wl["target_shape"]
    S = synth.seq_len_of(ts)
    infer_steps = 4 if distill else wl.get("infer_steps", 50)
    extra = {}
    if fp8:
        extra["mm_config"] = {"mm_type": "W-fp8-channel-sym-A-fp8-channel-sym-dynamic-Hip", "weight_auto_quant": True}
    if distill:
        extra.update(denoising_step_list=[1000, 750, 500, 250], sample_shift=5.0)
    extra.update({k: wl[k] for k in ("sample_guide_scale", "sample_shift") if k in wl and k not in extra})
    cfg = wan.default_config(dims, target_shape=ts, target_video_length=wl["frames"], infer_steps=infer_steps, enable_cfg=not distill, cfg_pair="auto", **extra)
    _, _, wd, lat, inputs = synth.workload_setup(workload, seed=0, device="cuda")
    model = wan.WanModel(cfg, wd)
    del wd
    sch = (scheduler.WanStepDistillScheduler if distill else scheduler.WanScheduler)(cfg, device="cuda")
    sch.prepare(latents=lat)
    model.set_scheduler(sch)
    timer = AttnTimer()
    model.transformer_infer.attn_time_hook = timer

    def one(i):
        sch.step_pre(i % sch.infer_steps if distill else i)
        model.infer(inputs)
        sch.step_post()

    for i in range(warmup):
        one(i)
    torch.cuda.synchronize()
    timer.enabled = True
    t0 = time.perf_counter()
    for i in range(steps):
        one(warmup + i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    timer.enabled = False
    assert torch.isfinite(sch.latents).all(), f"non-finite latents ({workload})"
    fwd = 1 if distill else 2
    flop_step, _ = step_flops(dims, S, dims["text_len"], fwd)
    rec = {"workload": workload, "dtype": "fp8-e4m3 w8a8 GEMMs, bf16 attention" if fp8 else "bf16", "schedule": "step-distill 4 steps (no CFG)" if distill else "UniPC, CFG",
           "tokens": S, "steps": steps, "warmup": warmup, "ms_per_step": ms, "infer_steps": infer_steps, "frames": wl["frames"],
           "frames_per_s": wl["frames"] / (infer_steps * ms * 1e-3), "denoise_loop_s": infer_steps * ms * 1e-3,
           "step_tflops_per_s": flop_step / (ms * 1e-3) / 1e12, "roofline": _attn_roofline(timer, steps, S, dims["num_heads"], dims["num_layers"], fwd)}
    del model, sch, inputs, lat
    torch.cuda.empty_cache()
    return rec


def run_hunyuan_config(workload, steps, warmup):
    """BASELINE config #5's denoise step on ONE GPU (HunyuanVideo-13B bf16, 720p x 129 frames: 118 800 image + 256 text tokens; no CFG, embedded guidance)."""
    from lightx2v_amd import hunyuan as hy, synth

    wl = synth.HUNYUAN_WORKLOADS[workload]
    dims = synth.HUNYUAN_DIMS[wl["model"]]
    cfg = hy.default_config(dims, infer_steps=50)
    wd = synth.synth_hunyuan_weights(dims, seed=0, device="cuda", gen_device="cuda")
    model = hy.HunyuanModel(cfg, wd)
    del wd
    lat, text_states, mask, ts2 = synth.synth_hunyuan_inputs(dims, wl["target_shape"], valid_text=(dims["text_len"] * 3) // 4)
    sch = hy.HunyuanScheduler(cfg)
    sch.prepare(lat)
    model.set_scheduler(sch)
    inputs = {"text_encoder_output": {"text_encoder_1_text_states": text_states.cuda(), "text_encoder_1_attention_mask": mask.cuda(), "text_encoder_2_text_states": ts2.cuda()}}

    def one(i):
        sch.step_pre(i)
        model.infer(inputs)
        sch.step_post()

    for i in range(warmup):
        one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        one(warmup + i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    assert torch.isfinite(sch.latents).all(), "non-finite latents (hunyuan)"
    _, _, t, h, w = wl["target_shape"]
    n_img, n_txt = t * (h // 2) * (w // 2), dims["text_len"]
    L, D, F = n_img + n_txt, dims["hidden"], dims["mlp"]
    flop = dims["double_blocks"] * (2 * L * D * (3 * D + D + 2 * F) + 4 * L * L * D) + dims["single_blocks"] * (2 * L * D * (3 * D + F) + 2 * L * (D + F) * D + 4 * L * L * D)
    rec = {"workload": workload, "dtype": "bf16", "schedule": "flow-match Euler, embedded guidance (no CFG)", "tokens": L, "steps": steps, "warmup": warmup, "ms_per_step": ms,
           "infer_steps": 50, "frames": wl["frames"], "frames_per_s": wl["frames"] / (50 * ms * 1e-3), "denoise_loop_s": 50 * ms * 1e-3,
           "step_tflops_per_s": flop / (ms * 1e-3) / 1e12, "step_frac_of_bf16_peak": flop / (ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS}
    del model, sch, inputs, lat
    torch.cuda.empty_cache()
    return rec


def run_wan_vae_decode(latent_shape=(16, 21, 90, 160)):
    """WanVAE.decode of the headline workload's latent (720p x 81 frames), fp32-grade hi/lo split of the 16-bit convolution operands (the default
    form): one warm-up decode (allocates the cache-carrying buffers), one timed.  The convolution launches' algorithmic FLOPs are counted on the way."""
    from lightx2v_amd import lib, synth, vae

    m = vae.WanVAE(synth.synth_wan_vae_weights(dim=96, seed=0), dim=96, conv16="split")
    z = torch.randn(*latent_shape, generator=torch.Generator().manual_seed(5)).cuda()
    flops = [0.0]
    orig16 = lib.vae_conv16

    def counted16(xp, strides, weight, out, T, H, W, **kw):
        cin = weight.shape[4] - (32 if kw.get("flags", 0) & lib.VCONV_ZERO_TAIL32 else 0)  # channels multiplied: the 32-channel-slab kernel skips a zero tail
        flops[0] += 2.0 * T * H * W * weight.shape[0] * cin * weight.shape[1] * weight.shape[2] * weight.shape[3]
        return orig16(xp, strides, weight, out, T, H, W, **kw)

    lib.vae_conv16 = counted16
    try:
        out = m.decode(z)
        torch.cuda.synchronize()
        flops[0] = 0.0
        t0 = time.perf_counter()
        out = m.decode(z)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        lib.vae_conv16 = orig16
    assert torch.isfinite(out).all(), "non-finite VAE output"
    rec = {"workload": f"wan_vae_decode z{list(latent_shape)} -> {list(out.shape)}", "conv_operands": "fp16 hi/lo split (fp32-grade)", "decode_s": dt,
           "conv16_tflop": flops[0] / 1e12, "roofline": {"kernel": "x2v::vae_conv16g_kernel<6> (the 16-bit halo-tiled 3x3x3 convolutions, 128 pixels x 96 couts per wave; three MFMA products per fp32-grade product)",
                                                         "bound": "mfma", "achieved": flops[0] / dt / 1e12, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                                         "frac": flops[0] / dt / 1e12 / BF16_MFMA_PEAK_TFLOPS, "note": "whole-decode average: numerator = the 16-bit convolution launches' FLOPs only"}}
    del m, out, z
    torch.cuda.empty_cache()
    return rec


def run_hunyuan_vae_decode(latent_shape=(1, 16, 33, 90, 160)):
    """The HunyuanVideo VAE's tiled decode of config #5's latent (720p x 129 frames), fp16 convolution operands as the reference runs it
    (autoencoder_kl_causal_3d.py:347-518: spatial + temporal tiles with blends).  One decode, timed with its first-use allocations (the tiles reuse
    their buffers from the second tile on); the 16-bit convolution launches' algorithmic FLOPs are counted on the way."""
    from lightx2v_amd import hunyuan_vae, lib, synth

    cfg = synth.HUNYUAN_VAE_CFG
    m = hunyuan_vae.VideoEncoderKLCausal3DModel(synth.synth_hunyuan_vae_weights(cfg, seed=0), cfg, conv16=True)
    z = (torch.randn(*latent_shape, generator=torch.Generator().manual_seed(1)) * 0.5).cuda()
    flops = [0.0]
    orig16 = lib.vae_conv16

    def counted16(xp, strides, weight, out, T, H, W, **kw):
        flops[0] += 2.0 * T * H * W * weight.shape[0] * weight.shape[4] * weight.shape[1] * weight.shape[2] * weight.shape[3]
        return orig16(xp, strides, weight, out, T, H, W, **kw)

    lib.vae_conv16 = counted16
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = m.decode(z)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        lib.vae_conv16 = orig16
    assert torch.isfinite(out).all(), "non-finite VAE output (hunyuan)"
    rec = {"workload": f"hunyuan_vae_decode z{list(latent_shape)} -> {list(out.shape)}", "conv_operands": "fp16 (the reference's precision)", "decode_s": dt,
           "conv16_tflop": flops[0] / 1e12, "roofline": {"kernel": "x2v::vae_conv16g_kernel<8> (the 16-bit halo-tiled 3x3x3 convolutions, 128 pixels x 128 couts per wave)",
                                                         "bound": "mfma", "achieved": flops[0] / dt / 1e12, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                                         "frac": flops[0] / dt / 1e12 / BF16_MFMA_PEAK_TFLOPS, "note": "whole-decode average: numerator = the 16-bit convolution launches' FLOPs only"}}
    del m, out, z
    torch.cuda.empty_cache()
    return rec


def other_configs(headline_ms_per_step, frames, infer_steps):
    """BASELINE.json's other single-GPU configurations and the VAE decode, run by the driver's command behind the headline steps (frames/s
    "both with and without VAE"; the reference runs the decode after the loop, default_runner.py:202-221).  None of it enters `value`."""
    out = {}
    for key, fn in (("config2_wan1.3b_480px49f", lambda: run_wan_config("wan1.3b_480px49f", 2, 1)),
                    ("config4_wan14b_w8a8_distill_720px81f", lambda: run_wan_config("wan14b_720px81f", 2, 1, fp8=True, distill=True)),
                    ("config5_hunyuan13b_720px129f_1gpu", lambda: run_hunyuan_config("hunyuan13b_720px129f", 1, 1)),
                    ("wan_vae_decode_720px81f", run_wan_vae_decode),
                    ("hunyuan_vae_decode_720px129f", run_hunyuan_vae_decode)):
        t0 = time.perf_counter()
        try:
            out[key] = fn()
        except Exception as e:  # noqa: BLE001 — a leg that fails must not cost the headline line
            out[key] = {"failed": f"{type(e).__name__}: {str(e)[:300]}"}
            torch.cuda.empty_cache()
        out[key]["leg_wall_s"] = time.perf_counter() - t0
    dec = out["wan_vae_decode_720px81f"].get("decode_s")
    if dec is not None:
        loop_s = infer_steps * headline_ms_per_step * 1e-3
        out["value_with_vae"] = {"value": frames / (loop_s + dec), "unit": "frames/s", "definition": f"frames / ({infer_steps} x headline step latency + one VAE decode): {loop_s:.1f} s + {dec:.2f} s"}
        c4 = out["config4_wan14b_w8a8_distill_720px81f"]
        if "denoise_loop_s" in c4:
            c4["end_to_end_with_vae_s"] = c4["denoise_loop_s"] + dec
    c5, dec5 = out["config5_hunyuan13b_720px129f_1gpu"], out["hunyuan_vae_decode_720px129f"].get("decode_s")
    if "denoise_loop_s" in c5 and dec5 is not None:
        c5["end_to_end_with_vae_s"] = c5["denoise_loop_s"] + dec5
        c5["frames_per_s_with_vae"] = c5["frames"] / c5["end_to_end_with_vae_s"]
    return out


def ulysses_self_check(dist, world, rank, one_gpu_plumbing=False):
    """N > 1, before the timed region: a one-layer Wan model with the real run's per-rank head count (5 x 128 per rank: at N = 8 exactly a Wan-14B
    layer; in the one-GPU plumbing mode a small 2-layer one) on a short token grid runs one CFG step sharded over the N ranks (the product Ulysses
    path: RCCL all-to-alls on the blocked exchange buffers) and unsharded on every rank; the two noise predictions must agree to 5e-3
    relative L2 on every rank (same kernels on re-partitioned rows).  The result travels in the JSON line so that a scaling record
    proves N ranks really exchanged data."""
    from lightx2v_amd import scheduler, synth, wan

    # the per-rank shapes of the real run where the node is full: 5 heads of 128 per rank (N = 8: D = 5120, 40 heads, F = 13824 = one Wan-14B layer)
    dims = dict(synth.WAN_DIMS["wan2.1-14b"], dim=640 * world, num_heads=5 * world, num_layers=1)
    if one_gpu_plumbing:
        dims = dict(synth.WAN_DIMS["wan-tiny"], dim=256 * world, num_heads=2 * world, ffn_dim=1024, num_layers=2)  # N ranks share one GPU: keep it small
    ts = (16, 3, 8 * world, 12)  # tokens divisible by N
    wd = synth.synth_wan_weights(dims, seed=3, device="cuda", gen_device="cuda")
    lat, ctx, ctx_null = synth.synth_inputs(dims, ts)
    inputs = {"text_encoder_output": {"context": [c.cuda() for c in ctx], "context_null": [c.cuda() for c in ctx_null]}}
    def one(pat, streams=True, blocked=True, split=True):
        cfg = wan.default_config(dims, target_shape=ts, target_video_length=9, infer_steps=4, parallel_attn_type=pat, cfg_branch_streams=streams)
        model = wan.WanModel(cfg, wd)
        if pat is not None:  # per-INSTANCE settings (the timed model gets the same ones from main)
            model.transformer_infer.blocked_exchange = blocked
            model.transformer_infer.parallel_attention.split_head2seq = split
        sch = scheduler.WanScheduler(cfg, device="cuda")
        sch.prepare(latents=lat)
        model.set_scheduler(sch)
        sch.step_pre(0)
        model.infer(inputs)
        torch.cuda.synchronize()
        return sch.noise_pred.float(), (model.transformer_infer.parallel_attention.split_head2seq if pat is not None else None)

    outs = [one(None)[0]]
    # The designed path first: the two CFG branches on two compute streams, blocked exchange buffers, head->seq in two overlapped pieces
    # (ulysses.py itself probes the split-size collective form once and falls back to a single head->seq exchange).  Should this
    # PyTorch/RCCL build still reject a form (argument errors are raised on every rank alike, before anything is sent), the simpler paths
    # are tried; the one that ran is named in the JSON line and used for the timed model.
    path, errors, settings = None, [], None
    for name, streams, blocked, halves in (("CFG branches on two streams, blocked buffers, head->seq in two overlapped pieces", True, True, True),
                                           ("sequential CFG branches, blocked buffers, head->seq in two overlapped pieces", False, True, True),
                                           ("sequential CFG branches, blocked buffers, one head->seq exchange", False, True, False),
                                           ("sequential CFG branches, row-major exchange (reference form, transposing copies)", False, False, False)):
        try:
            o, split_used = one("ulysses", streams, blocked, halves)
            outs.append(o)
            if blocked and halves and not split_used:
                name += " -> ONE head->seq exchange (split-size collective form rejected by the probe)"
            path, settings = name, {"cfg_branch_streams": streams, "blocked_exchange": blocked, "split_head2seq": bool(split_used) if blocked else False}
            break
        except Exception as e:  # noqa: BLE001
            errors.append(f"{name}: {type(e).__name__}: {str(e)[:200]}")
            if rank == 0:
                print(f"bench: Ulysses path '{name}' failed: {errors[-1]}", file=sys.stderr)
    if path is None:
        raise SystemExit("bench: no Ulysses exchange path works on this node: " + " | ".join(errors))
    rel = ((outs[0] - outs[1]).norm() / outs[0].norm()).reshape(1)
    worst = rel.clone() if dist.get_backend() == "nccl" else rel.cpu().clone()
    dist.all_reduce(worst, op=dist.ReduceOp.MAX)
    ok = bool(worst.item() < 5e-3)
    if not ok:
        raise SystemExit(f"bench: Ulysses self-check failed on rank {rank}: relative L2 {rel.item():.3e} (worst {worst.item():.3e})")
    return {"ranks": world, "worst_rel_l2": worst.item(), "tolerance": 5e-3, "passed": ok, "exchange_path": path, "rejected_paths": errors, "settings": settings}


def main():
    args = parse()
    from lightx2v_amd import launch

    # bare `python bench.py --gpus N` re-runs itself as N ranks under torch.distributed.run (lightx2v_amd/launch.py)
    world, rank, local_rank = launch.ranks(__file__, args.gpus)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch.distributed as dist

        import datetime

        # a stuck / rejected collective is aborted by the process group's own watchdog after this long instead of the 10-minute default
        pg_timeout = datetime.timedelta(seconds=max(60, args.hang_timeout or 240))
        if launch.one_gpu_test():  # plumbing mode: N ranks on ONE GPU over gloo with host-staged collectives (lightx2v_amd/launch.py) — timings meaningless
            dist.init_process_group("gloo", timeout=pg_timeout)
            launch.host_staged_collectives(dist)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=pg_timeout)
    watchdog = Watchdog((args.hang_timeout or 240) if world > 1 else args.hang_timeout, rank)
    watchdog.tick("process group up")

    from lightx2v_amd import lib, scheduler, synth, wan

    lib.init(local_rank)
    if args.i2v:
        args.workload = "wan14b_i2v_720px81f"
    wl = synth.WORKLOADS[args.workload]
    dims = synth.WAN_DIMS[wl["model"]]
    i2v = dims.get("task") == "i2v"
    if not args.infer_steps:
        args.infer_steps = wl.get("infer_steps", 50)
    ts = wl["target_shape"]
    S = synth.seq_len_of(ts)
    enable_cfg = not (args.no_cfg or args.distill)
    if args.distill:
        args.infer_steps = 4
        if args.warmup + args.steps > 4:  # the distilled schedule has 4 entries: wrap around instead of indexing past it
            print(f"bench: --distill has a 4-step schedule; step indices wrap modulo 4 (warmup {args.warmup} + steps {args.steps})", file=sys.stderr)
    extra = {}
    if args.fp8:
        extra["mm_config"] = {"mm_type": "W-fp8-channel-sym-A-fp8-channel-sym-dynamic-Hip", "weight_auto_quant": True}
    if args.mxfp8:
        extra["mm_config"] = {"mm_type": "W-mxfp8-A-mxfp8-dynamic-Hip", "weight_auto_quant": True}
    if args.distill:
        extra.update(denoising_step_list=[1000, 750, 500, 250], sample_shift=5.0)
    if args.no_cfg_streams or args.cfg_streams:
        extra["cfg_branch_streams"] = bool(args.cfg_streams)
    extra.update({k: wl[k] for k in ("sample_guide_scale", "sample_shift") if k in wl and k not in extra})
    if i2v:
        extra.update(task="i2v", in_dim=36, cross_attn_2_type="hip_flash")
    cfg = wan.default_config(
        dims, target_shape=ts, target_video_length=wl["frames"], infer_steps=args.infer_steps, enable_cfg=enable_cfg,
        parallel_attn_type="ulysses" if world > 1 else None, hip_ref_rounding=args.ref_rounding, cfg_pair=(False if (args.no_cfg_pair or args.cfg_streams) else True if args.cfg_pair else "auto"), **extra,
    )
    if world > 1 and dims["num_heads"] % world != 0:
        raise SystemExit(f"Ulysses needs num_heads % N == 0 ({dims['num_heads']} heads, N={world})")

    # identical weights/inputs on every rank: seeded device generator (weights are replicated under Ulysses)
    _, _, wd, lat, inputs = synth.workload_setup(args.workload, seed=0, device="cuda")
    model = wan.WanModel(cfg, wd)
    del wd
    watchdog.tick("model built")  # weight synthesis / load-time quantisation can take a while on a cold box
    sch = (scheduler.WanStepDistillScheduler if args.distill else scheduler.WanScheduler)(cfg, device="cuda")
    sch.prepare(latents=lat)
    model.set_scheduler(sch)
    timer = AttnTimer()
    model.transformer_infer.attn_time_hook = timer

    def one_step(i):
        i = i % sch.infer_steps if args.distill else i
        sch.step_pre(i)
        model.infer(inputs)
        sch.step_post()
        watchdog.tick(f"step {i} enqueued")  # host-side progress; a device-side hang surfaces at the next fence, which then stops ticking

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
        watchdog.tick("fence passed")

    sp_check = None
    if world > 1:
        watchdog.tick("Ulysses self-check")
        sp_check = ulysses_self_check(dist, world, rank, launch.one_gpu_test())
        watchdog.tick("Ulysses self-check passed")
        model.config["cfg_branch_streams"] = sp_check["settings"]["cfg_branch_streams"] and not args.no_cfg_streams
        model.transformer_infer.blocked_exchange = sp_check["settings"]["blocked_exchange"]
        model.transformer_infer.parallel_attention.split_head2seq = sp_check["settings"]["split_head2seq"]
    cfg_form_timing = None
    if world > 1 and enable_cfg and sp_check["settings"]["cfg_branch_streams"] and not args.no_cfg_streams and not args.cfg_streams:
        # N > 1, untimed: the two CFG forms by the clock, one step each after one step of first use (buffers, allocator pools).  On one GPU the
        # two-stream form costs 3.5 % at 14B sizes (two big attention launches evict each other's K / V) and what it gains under Ulysses — one
        # branch's kernels under the other's exchanges — has never been measured on a node, so the bench does not guess: every rank times both,
        # the MAX over ranks decides, the same on every rank.  (`--cfg-streams` / `--no-cfg-streams` pin the form.)
        cfg_form_timing = {}
        def restart():  # back to the initial noise and an empty multistep history (as tools/e2e.py does after its warm-up step)
            sch.reset()
            sch.prepare(latents=lat)

        for form, flag in (("two_streams_ms", True), ("sequential_ms", False)):
            model.config["cfg_branch_streams"] = flag
            restart()
            one_step(0)
            fence()
            tf = time.perf_counter()
            one_step(1)
            fence()
            dt = torch.tensor([time.perf_counter() - tf], device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            cfg_form_timing[form] = dt.item() * 1e3
        model.config["cfg_branch_streams"] = cfg_form_timing["two_streams_ms"] <= cfg_form_timing["sequential_ms"]
        cfg_form_timing["chosen"] = "two streams" if model.config["cfg_branch_streams"] else "sequential"
        restart()
    # Box calibration, UNTIMED, in this process: what the bare bf16 MFMA instruction sustains on this board right before and right after the timed
    # region (lib.mfma_probe -> x2v_mfma_probe_bf16).  Boxes of the pool differ by several percent under the 1400 W limit; with these
    # two numbers in the line, `roofline.frac_of_probe` can be compared between runs where `roofline.frac` (against the nominal 2.5 PFLOP/s) cannot.
    calib = None
    if not args.no_calibration:
        calib = {"kernel": "v_mfma_f32_16x16x32_bf16, operands in registers, 8 waves per CU on every CU (x2v_mfma_probe_bf16)", "probe_ms": args.probe_ms,
                 "mfma_probe_tflops_before": lib.mfma_probe(args.probe_ms)}
        watchdog.tick("calibration probe done")
    # the warm-up steps run BETWEEN the first probe and the timed region (a probe right in front of t0 would make the timed steps
    # start from a board the bare-MFMA loop had just heated past its sustained state); the step's own mix brings it to the state it is timed in
    for i in range(args.warmup):
        one_step(i)
    fence()
    sampler = SmiSampler(_smi_index(local_rank)) if (calib is not None and world == 1) else contextlib.nullcontext()
    comm_timer = None
    if world > 1:
        from lightx2v_amd import ulysses

        comm_timer = ulysses.CommTimer()
        model.transformer_infer.parallel_attention.comm_timer = comm_timer  # the CFG-branch driver hands it to the second branch's exchanges
    timer.enabled = True
    with sampler:
        t0 = time.perf_counter()
        for i in range(args.steps):
            one_step(args.warmup + i)
        fence()
        elapsed = time.perf_counter() - t0
    timer.enabled = False
    comm = None
    if comm_timer is not None:
        comm_timer.enabled = False
        c_ms, e_ms, n_coll = comm_timer.totals_ms()
        mine = {"rank": rank, "comm_ms_per_step": c_ms / args.steps, "exposed_comm_ms_per_step": e_ms / args.steps, "exchanges_per_step": n_coll / args.steps}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        comm = {"comm_ms_per_step": max(r["comm_ms_per_step"] for r in per_rank), "exposed_comm_ms_per_step": max(r["exposed_comm_ms_per_step"] for r in per_rank),
                "per_rank": per_rank,
                "definition": "HIP events on the communication stream around every exchange (comm) and on the compute streams around every join behind it (exposed = "
                              "the compute stream had nothing to run but the wait); max over ranks; what replaced the reference's two torch.cuda.synchronize() per attention (ulysses/attn.py:48,85)"}
    if calib is not None:
        calib["mfma_probe_tflops_after"] = lib.mfma_probe(args.probe_ms)
        # the reference value for `frac_of_probe`: the probe taken right behind the timed steps, i.e. on the board in the state the steps ran in.  With
        # warm-up steps the first probe sits in front of them and reads a cooler board (2065 vs 2024 TFLOP/s in a 25-step run) — it is
        # reported, not averaged in; without warm-up steps both probes bracket the timed region as in round 4 and the mean is used
        calib["mfma_probe_tflops"] = calib["mfma_probe_tflops_after"] if args.warmup > 0 else 0.5 * (calib["mfma_probe_tflops_before"] + calib["mfma_probe_tflops_after"])
        calib["mfma_probe_definition"] = "after the timed steps" if args.warmup > 0 else "mean of before / after the timed steps"
        calib["mfma_probe_tflops_mean"] = 0.5 * (calib["mfma_probe_tflops_before"] + calib["mfma_probe_tflops_after"])  # round 4's definition, for comparisons across rounds
        calib["mfma_probe_frac_of_nominal_peak"] = calib["mfma_probe_tflops"] / BF16_MFMA_PEAK_TFLOPS
        if world == 1:
            calib["smi_during_timed_region"] = sampler.summary()
    if dist is not None:
        tmax = torch.tensor([elapsed], device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()
    assert torch.isfinite(sch.latents).all(), "non-finite latents"

    ms_per_step = elapsed * 1e3 / args.steps
    fps = wl["frames"] / (args.infer_steps * ms_per_step * 1e-3)
    fwd = 2 if enable_cfg else 1
    ctx_len = dims["text_len"] + (synth.I2V_CLIP_TOKENS if i2v else 0)  # i2v: a second cross-attention over the 257 CLIP tokens (transformer_infer.py:437-455)
    flop_step, flop_attn = step_flops(dims, S, ctx_len, fwd)
    # dominant kernel: the attention forward kernel (self-attention launches carry 99 % of its FLOPs).
    # Algorithmic FLOPs per launch = 4 * Sq * Sk * heads * 128: self Sk = S with H/N heads under
    # Ulysses (all S queries), cross Sk = text_len with all H heads on this rank's S/N queries.
    heads_local = dims["num_heads"] // world
    s_local = -(-S // world)
    flop_self = 4.0 * S * S * heads_local * 128
    flop_cross = 4.0 * s_local * (ctx_len / (2 if i2v else 1)) * dims["num_heads"] * 128  # per launch; i2v launches text and CLIP keys separately (mean)
    n_self, n_cross = timer.count("self"), timer.count("cross")
    ms_self, ms_cross = timer.total_ms("self"), timer.total_ms("cross")
    # FLOPs per self-attention LAUNCH = the step's self-attention FLOPs / its launches: one launch covers both CFG forwards in pair mode
    # (wan.WanModel._forward_pair), half a layer's query rows under Ulysses (head->seq overlap, ulysses.py), one forward's layer otherwise
    launches_per_step = n_self / max(1, args.steps)
    forwards_per_launch = dims["num_layers"] * fwd / max(launches_per_step, 1e-9)
    flop_self *= forwards_per_launch
    # roofline object = the self-attention launches of the dominant kernel (x2v::attn_fwd_v9_kernel: 99 % of the attention
    # FLOPs, 72 % of the step's); cross-attention runs a different instantiation and is reported beside it
    attn_ms = ms_self / max(n_self, 1)
    flop_launch = flop_self
    achieved = flop_launch / (attn_ms * 1e-3) / 1e12 if attn_ms > 0 else 0.0
    traffic, traffic_note = None, "no PMC summary for this shape under profiles/"
    # PMC counters need their own rocprofv3 passes (tools/pmc_traffic.py: this very command under --pmc FETCH_SIZE / WRITE_SIZE, the rows of this
    # kernel instantiation averaged per launch); the summary they wrote for THIS launch form is reported, never one of another form
    pmc_path = next((q for q in (os.path.join(ROOT, "profiles", f"r0{r}_pmc_attn_traffic.json") for r in (6, 5, 4, 3)) if os.path.exists(q)), None)  # the newest round's passes
    if world == 1 and pmc_path is not None:
        with open(pmc_path) as fh:
            pmc = json.load(fh)
        if pmc.get("tokens") == S and pmc.get("heads") == heads_local and abs(pmc.get("forwards_per_launch", 0) - forwards_per_launch) < 1e-6:
            traffic = pmc["hbm_bytes_per_launch"]
            traffic_note = pmc["note"]
            if "true, true" in str(pmc.get("kernel", "")) and not wan.SELF_ATTN_STAGGER:
                traffic_note += "  (Counters taken on the staggered-walk instantiation of this launch form; the walk's starting tile changes when a tile is fetched, not what is fetched.)"
    model_label = {"wan2.1-14b": "Wan2.1-14B", "wan2.1-1.3b": "Wan2.1-1.3B", "wan2.1-14b-i2v": "Wan2.1-I2V-14B"}.get(wl["model"], wl["model"])
    res_label = {"wan14b_720px81f": "720p 81f", "wan1.3b_480px49f": "480p 49f", "wan1.3b_256x256x17f": "256x256 17f", "wan14b_i2v_720px81f": "720p 81f",
                 "wan14b_i2v_480px81f": "480p 81f"}.get(args.workload, args.workload)
    fast_attn = not args.ref_rounding
    il = getattr(model, "_cfg_interleave", None)
    cfg_form = ("no CFG" if fwd == 1 else "pair pass: both forwards as one launch sequence over stacked rows" if model._pair_ok(inputs)
                else "two compute streams, block by block (kernel durations are measured while the other branch's kernels share the chip)" if il is not None and il._streams is not None and model.config.get("cfg_branch_streams", "auto") is not False
                else "one forward after the other")
    out = {
        "metric": f"denoise-step latency (ms) + video frames/sec, {model_label} {res_label} @{world}/8 GPU",
        "value": fps,
        "unit": "frames/s",
        "n_gpus": world,
        "rccl_world": (dist.get_world_size() if dist is not None else 1),
        "rccl_backend": (dist.get_backend() if dist is not None else None),
        "sp_self_check": sp_check,
        "comm": comm,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "mxfp8 (e4m3 + e8m0/32) GEMMs, bf16 attention" if args.mxfp8 else "fp8-e4m3 w8a8 GEMMs, bf16 attention" if args.fp8 else "bf16",
        "data": "synthetic" if not launch.one_gpu_test() else "synthetic — PLUMBING RUN: all ranks share one GPU over gloo with host-staged collectives, timings meaningless",
        "config": {
            "workload": args.workload,
            "tokens": S,
            "latent_shape": list(ts),
            "frames": wl["frames"],
            "infer_steps": args.infer_steps,
            "cfg_forwards_per_step": fwd,
            "cfg_form": cfg_form,
            "cfg_form_timing": cfg_form_timing,
            "parallelism": f"ulysses-sp{world}" if world > 1 else "single",
            "schedule": "step-distill 4 steps (no CFG)" if args.distill else "UniPC",
            "fps_definition": "frames / (infer_steps * ms_per_step), denoise loop only (no text encoder / VAE)",
            "step_tflop": flop_step / 1e12,
            "step_tflops_per_s_per_gpu": flop_step / (ms_per_step * 1e-3) / 1e12 / world,
            "step_frac_of_bf16_peak": flop_step / (ms_per_step * 1e-3) / 1e12 / world / BF16_MFMA_PEAK_TFLOPS,
        },
        "roofline": {
            "kernel": ((f"x2v::attn_fwd_v9_kernel<8, 8, true, {'true' if wan.SELF_ATTN_STAGGER else 'false'}> (ping-pong on 16x16x32 MFMA, q prescaled, "
                        f"{'staggered key walk' if wan.SELF_ATTN_STAGGER else 'key walk from tile 0'}; self-attention launches")
                       + ("; one launch = both CFG forwards of a layer)" if abs(forwards_per_launch - 2.0) < 1e-6 else ")") if fast_attn
                       else "x2v::attn_fwd_pipe_kernel<8, 8> (reference-rounding mode; self-attention launches)"),
            "bound": "mfma",
            "achieved": achieved,
            "peak": BF16_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / BF16_MFMA_PEAK_TFLOPS,
            "frac_of_probe": (achieved / calib["mfma_probe_tflops"]) if calib and calib.get("mfma_probe_tflops") else None,
            "frac_of_probe_mean": (achieved / calib["mfma_probe_tflops_mean"]) if calib and calib.get("mfma_probe_tflops_mean") else None,
            "traffic": traffic,
            "algorithmic_bytes_per_launch": 4.0 * S * heads_local * 128 * 2 * forwards_per_launch,
            "forwards_per_launch": forwards_per_launch,
            "launches_timed": n_self,
            "avg_launch_ms": attn_ms,
            "flop_per_launch": flop_launch,
            "self_attention": {"launches": n_self, "avg_ms": ms_self / max(n_self, 1), "tflops": flop_self * n_self / max(ms_self, 1e-9) / 1e9},
            "cross_attention": {"launches": n_cross, "avg_ms": ms_cross / max(n_cross, 1), "tflops": flop_cross * n_cross / max(ms_cross, 1e-9) / 1e9},
            "traffic_note": traffic_note,
        },
    }
    # Which kernels this process really ran: the library's process-wide A/B switches as latched from the environment,
    # and the dispatcher's own answer for the step's shapes (x2v_gemm_kernel_choice: tile family + continuous-form bit; x2v_attn_vt_launch_plan)
    M_rows = (2 if "pair pass" in cfg_form else 1) * s_local
    D_, F_ = dims["dim"], dims["ffn_dim"]
    is_fp8 = bool(args.fp8)

    def _gk(n, k):
        fam, cont = lib.gemm_kernel_choice(M_rows, n, k, fp8=is_fp8, with_form=True)
        names = {1: "gemm128", 2: "gemm256c8 (continuous w8a8)" if cont else "gemm256 (ping-pong w8a8)", 3: "gemm256c (continuous)" if cont else "gemm256s (one tile per workgroup)"}
        return {"M": M_rows, "N": n, "K": k, "family": fam, "continuous": cont, "kernel": names.get(fam, str(fam)) if not args.mxfp8 else "gemm256 (mxfp8 mode)"}

    plan = lib.attn_vt_launch_plan(S, S, heads_local, batch=2 if "pair pass" in cfg_form else 1, stagger=wan.SELF_ATTN_STAGGER)
    out["kernels"] = {
        "switches": lib.switches(),
        "switches_set_in_env": {k: v for k, v in os.environ.items() if k.startswith("X2V_")},
        "gemm": {"qkvo": _gk(D_, D_), "ffn0": _gk(F_, D_), "ffn2": _gk(D_, F_)},
        "self_attention_launch_plan": {"xcd_remap": bool(plan[0]), "staggered_walk": bool(plan[1]), "Sq": S, "Sk": S, "heads": heads_local},
    }
    out["box_calibration"] = calib
    default_headline = args.workload == "wan14b_720px81f" and not (args.fp8 or args.mxfp8 or args.distill or args.no_cfg or args.ref_rounding)
    if world == 1 and not args.no_other_configs and (default_headline or args.other_configs):
        # free the headline model first (28 GB of weights + the step's buffers); the legs build their own
        model.transformer_infer.attn_time_hook = None
        del model, sch, inputs, lat
        torch.cuda.empty_cache()
        watchdog.tick("other configs")
        out["other_configs"] = other_configs(ms_per_step, wl["frames"], args.infer_steps)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(dims, S, ts, dims["text_len"], wl["frames"], args.infer_steps, fwd, args.cpu_baseline_rows, not args.no_cpu_config1)
            except Exception as e:  # noqa: BLE001 — the measured line must not be lost to a host-side failure (e.g. out of host memory)
                out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port", "sample": f"failed: {type(e).__name__}: {str(e)[:200]}"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
