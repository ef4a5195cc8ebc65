import torch


def rel_l2(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    return ((got - ref).norm() / (ref.norm() + 1e-12)).item()


def assert_bf16_close(got, ref, ulps=1.0, atol=0.0, bad_frac=0.0, name=""):
    """|got - ref| <= ulps * 2^-7 * |ref| + atol elementwise (2^-7 = the widest relative bf16 ulp), except for a
    fraction `bad_frac` of elements (rounding-boundary flips from a different fp32 summation order)."""
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(got).all(), f"{name}: non-finite values"
    diff = (got - ref).abs()
    tol = ulps * 0.0078125 * ref.abs() + atol
    bad = (diff > tol).float().mean().item()
    assert bad <= bad_frac, f"{name}: {bad:.2e} of elements outside {ulps} ulp (+{atol}); max diff {diff.max().item():.4g}, ref absmax {ref.abs().max().item():.4g}"


def assert_rel(got, ref, tol, name=""):
    assert torch.isfinite(got.float()).all(), f"{name}: non-finite values"
    e = rel_l2(got, ref)
    assert e <= tol, f"{name}: relative L2 error {e:.3e} > {tol}"


def record(name, **numbers):
    """Append measured parity numbers to gpurun_out/parity_summary.jsonl (merged back from the GPU box; the round's summary under
    profiles/ is assembled from it).  Never fails a test."""
    import json
    import os

    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        path = os.environ.get("X2V_PARITY_LOG") or os.path.join(root, "gpurun_out", "parity_summary.jsonl")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "a") as f:
            f.write(json.dumps({"name": name, **{k: (float(v) if isinstance(v, (int, float)) else v) for k, v in numbers.items()}}) + "\n")
    except OSError:
        pass
