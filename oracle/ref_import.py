"""TEST INFRASTRUCTURE ONLY — imports the *unmodified* reference (ModelTC/lightx2v, read-only at
/root/reference) on CPU so that (1) the restated oracle in `oracle/wan_oracle.py` can be validated against
it and (2) golden vectors can be generated (`oracle/gen_golden.py`).

/root/reference exists only in the authoring container, never on the GPU box: nothing in `-m gpu` tests,
`smoke()` or `bench.py` may import this module.

Shims applied (SURVEY.md §8c recipe):
  * `loguru`, `easydict` stub packages (oracle/ref_shims) ahead of the reference on sys.path;
  * `torch.cuda.get_device_capability` patched (called at import time by
    lightx2v/common/ops/attn/attn_weight.py:26 and lightx2v/attentions/common/sage_attn2.py:3);
  * `torch.empty(..., pin_memory=True)` → unpinned (mm_weight.py:77, rms_norm_weight.py:23, ... allocate
    pinned mirrors at load; no GPU here);
  * `Tensor.cuda()` → identity (pre_infer.py:20,60 call `.cuda()` unconditionally);
  * `torch.zeros(..., device="cuda")` → CPU (hunyuan/infer/pre_infer.py:50);
  * a `diffusers` stub package (oracle/ref_shims/diffusers) providing `randn_tensor`, the only symbol the Hunyuan
    scheduler module needs from it at import time (schedulers/hunyuan/scheduler.py:3);
  * `DTYPE=BF16` (every shipped script exports it).
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_SHIMS = os.path.join(_HERE, "ref_shims")
# Where the unmodified reference is: $X2V_REFERENCE_ROOT, else /root/reference (the authoring container), else the staged copy that
# `__graft_entry__.build()` puts under oracle/_ref/reference/ (oracle/stage_reference.sh: git-ignored — never in history — but it travels to the
# GPU box with the snapshot, so that the reference-through-the-plugin GPU tests run there instead of skipping).
_STAGED = os.path.join(_HERE, "_ref", "reference")
REFERENCE_ROOT = os.environ.get("X2V_REFERENCE_ROOT") or next((p for p in ("/root/reference", _STAGED) if os.path.isdir(os.path.join(p, "lightx2v"))), "/root/reference")

_patched = False


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "lightx2v"))


def patch_and_import():
    """Idempotently patch torch for a GPU-less import and put the reference on sys.path."""
    global _patched
    if not reference_available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")
    import torch

    if not _patched:
        os.environ.setdefault("DTYPE", "BF16")
        for p in (REFERENCE_ROOT, _SHIMS):
            if p not in sys.path:
                sys.path.insert(0, p)
        if not torch.cuda.is_available():
            torch.cuda.get_device_capability = lambda *a, **k: (9, 4)
            _orig_empty = torch.empty

            def _empty(*a, **k):
                k.pop("pin_memory", None)
                return _orig_empty(*a, **k)

            torch.empty = _empty
            _orig_zeros = torch.zeros

            def _zeros(*a, **k):  # hunyuan/infer/pre_infer.py:50 allocates cu_seqlens with device="cuda"
                if str(k.get("device", "")).startswith("cuda"):
                    k.pop("device")
                return _orig_zeros(*a, **k)

            torch.zeros = _zeros
            torch.Tensor.cuda = lambda self, *a, **k: self
            torch.Tensor.pin_memory = lambda self, *a, **k: self
            torch.cuda.synchronize = lambda *a, **k: None
            torch.cuda.empty_cache = lambda *a, **k: None
        _patched = True
        print(f"oracle.ref_import: unmodified reference imported from {REFERENCE_ROOT}" + (" (the staged copy)" if REFERENCE_ROOT == _STAGED else ""), file=sys.stderr)
    import lightx2v  # noqa: F401
    import lightx2v.common.ops  # noqa: F401  (populates the operator registries)

    return lightx2v


def make_config(dims: dict, **overrides):
    """EasyDict config with the keys the Wan hot path reads (WanTransformerInfer.__init__,
    WanPreInfer.__init__, WanTransformerWeights.__init__, WanScheduler.__init__)."""
    patch_and_import()
    from easydict import EasyDict

    cfg = dict(
        task="t2v",
        model_cls="wan2.1",
        dim=dims["dim"],
        ffn_dim=dims["ffn_dim"],
        num_heads=dims["num_heads"],
        num_layers=dims["num_layers"],
        freq_dim=256,
        text_len=dims.get("text_len", 512),
        in_dim=16,
        out_dim=16,
        eps=1e-6,
        patch_size=(1, 2, 2),
        vae_stride=(4, 8, 8),
        cpu_offload=False,
        do_mm_calib=False,
        mm_config={},
        self_attn_1_type="torch_sdpa",
        cross_attn_1_type="torch_sdpa",
        cross_attn_2_type="torch_sdpa",
        attention_type="torch_sdpa",
        feature_caching="NoCaching",
        parallel_attn_type=None,
        enable_cfg=True,
        sample_guide_scale=6.0,
        sample_shift=8.0,
        infer_steps=4,
        seed=42,
        target_video_length=17,
        target_shape=(16, 5, 32, 32),
    )
    cfg.update(overrides)
    return EasyDict(cfg)


def build_reference_wan(cfg, weight_dict):
    """Instantiate the reference's weight trees + infer objects on CPU from a name→tensor dict."""
    patch_and_import()
    from lightx2v.models.networks.wan.weights.pre_weights import WanPreWeights
    from lightx2v.models.networks.wan.weights.post_weights import WanPostWeights
    from lightx2v.models.networks.wan.weights.transformer_weights import WanTransformerWeights
    from lightx2v.models.networks.wan.infer.pre_infer import WanPreInfer
    from lightx2v.models.networks.wan.infer.post_infer import WanPostInfer
    from lightx2v.models.networks.wan.infer.transformer_infer import WanTransformerInfer

    pre_w, post_w, tr_w = WanPreWeights(cfg), WanPostWeights(cfg), WanTransformerWeights(cfg)
    for w in (pre_w, post_w, tr_w):
        w.load(weight_dict)
    return dict(
        pre_w=pre_w,
        post_w=post_w,
        tr_w=tr_w,
        pre=WanPreInfer(cfg),
        post=WanPostInfer(cfg),
        tr=WanTransformerInfer(cfg),
    )
