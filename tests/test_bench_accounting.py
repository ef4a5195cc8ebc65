"""The algorithmic work bench.py divides by (roofline.achieved, step_tflops) must be SURVEY.md §8(d)'s per-unit figures:
per Wan block and forward GEMM = 12 S D^2 + 4 Lc D^2 + 4 S D F, ATTN = 4 S^2 D + 4 S Lc D; step = L x (GEMM + ATTN) x forwards."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_step_flops_match_survey_table(bench):
    from lightx2v_amd import synth

    # config 2 (1.3B 480p x 49f, S = 20280) and config 3 (14B 720p x 81f, S = 75600), CFG = 2 forwards
    rows = {"wan1.3b_480px49f": (20280, 1.695e12, 2.591e12, 2.572e14), "wan14b_720px81f": (75600, 4.524e13, 1.178e14, 1.305e16)}
    for name, (S, gemm_blk, attn_blk, step) in rows.items():
        wl = synth.WORKLOADS[name]
        dims = synth.WAN_DIMS[wl["model"]]
        assert synth.seq_len_of(wl["target_shape"]) == S
        total, attn = bench.step_flops(dims, S, dims["text_len"], 2, cross_kv_cached=False)  # the survey's table counts every GEMM of the reference block
        L = dims["num_layers"]
        assert attn / (2 * L) == pytest.approx(attn_blk, rel=2e-3)
        assert (total - attn) / (2 * L) == pytest.approx(gemm_blk, rel=2e-3)
        assert total == pytest.approx(step, rel=2e-3)
    # what a timed step really executes: the text K/V projections (4 Lc D^2 per block) are step-invariant and cached (wan.py cache_cross_kv)
    d14 = synth.WAN_DIMS["wan2.1-14b"]
    full, _ = bench.step_flops(d14, 75600, 512, 2, cross_kv_cached=False)
    timed, _ = bench.step_flops(d14, 75600, 512, 2)
    assert full - timed == 2 * d14["num_layers"] * 4 * 512 * d14["dim"] ** 2 and timed / full > 0.999
    # the distilled config runs one forward per step: half the CFG step
    dims = synth.WAN_DIMS["wan2.1-14b"] if "wan2.1-14b" in synth.WAN_DIMS else synth.WAN_DIMS[synth.WORKLOADS["wan14b_720px81f"]["model"]]
    assert bench.step_flops(dims, 75600, 512, 1)[0] * 2 == bench.step_flops(dims, 75600, 512, 2)[0]


def test_roofline_peak_is_the_dense_bf16_figure(bench):
    # /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters": 2.5 PFLOP/s dense bf16 (the 5 PFLOP/s headline includes 2:1 sparsity)
    assert bench.BF16_MFMA_PEAK_TFLOPS == 2500.0


@pytest.mark.parametrize("script", ["bench.py", "tools/e2e.py", "tools/hunyuan_bench.py"])
def test_multi_gpu_launch_contract(script):
    """`python <script> --gpus N` without a launcher environment starts N ranks of itself under torch.distributed.run (the
    driver's own command form must not need a human for N > 1): on this GPU-less box both ranks come up with their RANK / LOCAL_RANK,
    find no device and stop with a message; a launcher whose --nproc-per-node differs from --gpus is refused."""
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, script), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    text = p.stderr + p.stdout
    assert "no launcher environment, starting -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1" in text
    import torch

    if not torch.cuda.is_available():
        # the launcher tears the other rank down as soon as one exits, so only one of the two messages is guaranteed to make it out
        assert p.returncode != 0 and ("rank 0 needs cuda:0" in text or "rank 1 needs cuda:1" in text)
    p = subprocess.run([sys.executable, os.path.join(ROOT, script), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"), cwd=ROOT)
    assert p.returncode != 0 and "--nproc-per-node must equal --gpus" in (p.stderr + p.stdout)


def test_watchdog_exits_124_when_no_progress_arrives(tmp_path):
    """bench.Watchdog (first-contact insurance of the multi-GPU run): no tick for `limit` seconds -> the stage and every thread's stack on stderr, exit
    code 124; regular ticks keep the process alive.  Run in a child process (the watchdog ends its process with os._exit)."""
    import subprocess
    import sys

    code = (
        "import importlib.util, sys, time\n"
        f"spec = importlib.util.spec_from_file_location('bench_module', {os.path.join(ROOT, 'bench.py')!r})\n"
        "m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)\n"
        "w = m.Watchdog(1, 3)\n"
        "for _ in range(4):\n"
        "    time.sleep(0.5); w.tick('alive')\n"
        "print('TICKED', flush=True)\n"
        "w.tick('stuck in the exchange')\n"
        "time.sleep(30)\n"
        "print('NOT REACHED')\n"
    )
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60, cwd=ROOT)
    assert p.returncode == 124, (p.returncode, p.stderr[-500:])
    assert "TICKED" in p.stdout and "NOT REACHED" not in p.stdout
    assert "rank 3 made no progress" in p.stderr and "stuck in the exchange" in p.stderr


def test_reference_cpu_baseline_script_reproduces_the_golden_latents():
    """bench.py's cpu_baseline leg (kind "reference") runs oracle/ref_cpu_baseline.py in a subprocess with the GPUs hidden: the UNMODIFIED
    reference's WanModel + WanScheduler in DefaultRunner.run's loop.  On the wan-tiny plumbing model its final latents must be the ones
    tests/golden/wan-tiny_model.safetensors holds (written by oracle/gen_golden.py from the same reference) — the baseline script times what the
    fixtures pin, with the thread sweep and the timing marks in between."""
    import json
    import subprocess

    from safetensors.torch import load_file

    from oracle import ref_import

    if not ref_import.reference_available():
        pytest.skip("reference checkout not present")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, "-m", "oracle.ref_cpu_baseline", "--tiny", "--threads", "1,2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["kind"] == "reference" and rec["threads"] in (1, 2) and len(rec["ms_per_step"]) == 4
    gold = load_file(os.path.join(ROOT, "tests", "golden", "wan-tiny_model.safetensors"))["latents_after_step3"]
    assert rec["latents_abs_sum"] == float(gold.double().abs().sum())


def test_kernel_gaps_accounting_on_a_synthetic_trace(tmp_path):
    """tools/kernel_gaps.py (the config-#2 accounting of profiles/r05_call2_*): on a synthetic rocprofv3 kernel trace with known durations and gaps it must
    report the union-of-intervals busy time, the idle share, and classify the GEMM launches by epilogue and duration (ffn2's F -> D residual launches run
    F / D times longer than the D -> D ones)."""
    import csv
    import json
    import subprocess

    rows, t = [], 1000
    names = {"q": "void x2v::gemm256c_kernel<0>(char const*)", "o": "void x2v::gemm256c_kernel<2>(char const*)", "f0": "void x2v::gemm256c_kernel<1>(char const*)",
             "f2": "void x2v::gemm256c_kernel<2>(char const*)", "attn": "void x2v::attn_fwd_v9_kernel<8, 8, true, false>(x)", "ln": "void x2v::layernorm_stream_kernel<1, false, true>(x)"}
    dur = {"q": 80_000, "o": 90_000, "f0": 480_000, "f2": 400_000, "attn": 2_000_000, "ln": 30_000}
    for layer in range(20):
        for k in ("ln", "q", "attn", "o", "ln", "f0", "f2"):
            rows.append((t, t + dur[k], names[k]))
            t += dur[k] + 10_000  # 10 us between kernels
    p = tmp_path / "kernel_trace.csv"
    with open(p, "w") as fh:
        w = csv.writer(fh)
        w.writerow(["Kind", "Start_Timestamp", "End_Timestamp", "Kernel_Name"])
        for s, e, n in rows:
            w.writerow(["KERNEL_DISPATCH", s, e, n])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_gaps.py"), str(p), "20280", "1536", "8960", "12", "0.0"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-1000:]
    d = json.loads(out.stdout)
    busy = sum(dur[k] for k in ("ln", "q", "attn", "o", "ln", "f0", "f2")) * 20
    assert abs(d["busy_ms"] - busy / 1e6) < 1e-6 and abs(d["idle_share"] - (139 * 10_000) / (busy + 139 * 10_000)) < 1e-6 and abs(d["median_gap_us"] - 10.0) < 1e-9
    g = d["gemm_by_shape"]
    assert g["D->D plain / V^T"]["calls"] == 20 and g["D->D +residual"]["calls"] == 20 and g["F->D +residual"]["calls"] == 20 and g["D->F +GELU"]["calls"] == 20
    assert abs(g["F->D +residual"]["tflops"] - 2.0 * 20280 * 8960 * 1536 / 400e-6 / 1e12) < 1e-6
    attn = next(k for k in d["kernels"] if k["kernel"].startswith("attn_fwd_v9_kernel<8, 8, true"))
    assert abs(attn["tflops"] - 4.0 * 20280 * 20280 * 12 * 128 / 2e-3 / 1e12) < 1e-6
