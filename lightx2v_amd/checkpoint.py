"""On-disk checkpoint formats of the DiT hot path: the quantised / block-split layout the reference's converter writes and its
model loaders read.

reference:
  tools/convert/converter.py:294-339   quantize_tensor  — per-out-channel symmetric scale = max|w| (clamped at 1e-5) / qmax, e4m3 (RNE,
                                       saturating at ±448) or int8 (round-half-even, clamp) weights, fp32 scales [N, 1]
  tools/convert/converter.py:342-408   quantize_model   — which keys are quantised (2-D tensors whose key_idx-th name part is a target
                                       module), `<name>.weight` + `<name>.weight_scale`, everything else cast to `non_linear_dtype`
  tools/convert/converter.py:518-580   file layout      — `block_{i}.safetensors` + `non_block.safetensors` (save_by_block) or
                                       `<name>_part{k}.safetensors` chunks, plus `diffusion_pytorch_model.safetensors.index.json`
  models/networks/wan/model.py:77-144  loaders          — plain `*.safetensors` directory, index-driven quantised directory, and the
                                       split (`non_block.safetensors` now, blocks lazily) form; fp32 tensors become bf16 unless the
                                       key names a norm / embedding / modulation / time layer (model.py:148-156)

This is load-time host code (not on the per-step path): tensors are produced with torch ops on whatever device they live on and
handed to the operator objects of `ops.py`, whose `MMWeightFp8Hip.load` consumes exactly these `weight` / `weight_scale` pairs.
The Diffusers <-> LightX2V key renaming tables of the converter (:16-290) and LoRA merging (:411-466) are not part of the path.
"""
import glob
import json
import os
import re
from collections import defaultdict

import torch
from safetensors import safe_open
from safetensors import torch as st

# model.py:148-156 — layers that stay fp32 when DTYPE != BF16-everything
SKIP_BF16 = ("norm", "embedding", "modulation", "time", "img_emb.proj.0", "img_emb.proj.4")

# converter.py:674-705 — which modules of each model type are quantised, and which dotted name part identifies them
MODEL_TYPE_KEYS = {
    "wan_dit": dict(key_idx=2, target_keys=["self_attn", "cross_attn", "ffn"], ignore_key=None),
    "hunyuan_dit": dict(
        key_idx=2,
        target_keys=["img_mod", "img_attn_qkv", "img_attn_proj", "img_mlp", "txt_mod", "txt_attn_qkv", "txt_attn_proj", "txt_mlp", "linear1", "linear2", "modulation"],
        ignore_key=None,
    ),
    "wan_t5": dict(key_idx=2, target_keys=["attn", "ffn"], ignore_key=None),
    "wan_clip": dict(key_idx=3, target_keys=["attn", "mlp"], ignore_key="textual"),
}

INDEX_NAME = "diffusion_pytorch_model.safetensors.index.json"


# ------------------------------------------------------------------------------------------------ quantisation
def quantize_tensor(w, dtype=torch.float8_e4m3fn):
    """converter.py:294-339.  w [N, K] → (w_q [N, K] `dtype`, scales fp32-or-w.dtype [N, 1])."""
    if w.dim() != 2:
        raise ValueError(f"Only 2D tensors supported. Got {w.dim()}D tensor")
    if torch.isnan(w).any():
        raise ValueError("Tensor contains NaN values")
    max_val = w.abs().amax(dim=1, keepdim=True).clamp(min=1e-5)
    if dtype == torch.float8_e4m3fn:
        finfo = torch.finfo(dtype)
        scales = max_val / finfo.max
        # the reference rounds with qtorch's float_quantize(x, exp=4, man=3, "nearest") after clipping to ±448: round-to-nearest-even
        # onto the e4m3 grid, which is what the conversion below does
        w_q = torch.clip(w / scales, finfo.min, finfo.max).float().to(dtype)
    elif dtype == torch.int8:
        scales = max_val / 127
        w_q = torch.clamp(torch.round(w / scales), -128, 127).to(dtype)
    else:
        raise ValueError(f"unsupported linear dtype {dtype}")
    return w_q.reshape(w.shape), scales.view(w.shape[0], -1)


def quantize_model(weights, target_keys, key_idx=2, ignore_key=None, linear_dtype=torch.float8_e4m3fn, non_linear_dtype=torch.float32):
    """converter.py:342-408 (in place, returns `weights`)."""
    for key in list(weights.keys()):
        if ignore_key is not None and ignore_key in key:
            del weights[key]
            continue
        t = weights[key]
        parts = key.split(".")
        if not isinstance(t, torch.Tensor) or t.dim() != 2 or len(parts) < key_idx + 1 or parts[key_idx] not in target_keys:
            if t.dtype != non_linear_dtype:
                weights[key] = t.to(non_linear_dtype)
            continue
        w_q, scales = quantize_tensor(t, linear_dtype)
        weights[key] = w_q
        weights[key + "_scale"] = scales
    return weights


# ------------------------------------------------------------------------------------------------ writing
def save_checkpoint(weights, out_dir, save_by_block=False, chunk_size=100, output_name="converted"):
    """converter.py:518-580: safetensors files + the index json.  Returns the index dict."""
    os.makedirs(out_dir, exist_ok=True)
    index = {"metadata": {"total_size": 0}, "weight_map": {}}

    def write(name, tensors):
        path = os.path.join(out_dir, name)
        st.save_file({k: v.contiguous() for k, v in tensors.items()}, path)
        for k in tensors:
            index["weight_map"][k] = name
        index["metadata"]["total_size"] += os.path.getsize(path)

    if save_by_block:
        groups, rest = defaultdict(dict), {}
        pat = re.compile(r"blocks\.(\d+)\.")
        for k, v in weights.items():
            m = pat.search(k)
            if m:
                groups[m.group(1)][k] = v
            else:
                rest[k] = v
        for idx, tensors in groups.items():
            write(f"block_{idx}.safetensors", tensors)
        if rest:
            write("non_block.safetensors", rest)
    else:
        chunk, part = {}, 0
        for i, (k, v) in enumerate(weights.items()):
            chunk[k] = v
            if chunk_size > 0 and (i + 1) % chunk_size == 0:
                write(f"{output_name}_part{part}.safetensors", chunk)
                chunk, part = {}, part + 1
        if chunk:
            write(f"{output_name}_part{part}.safetensors", chunk)
    with open(os.path.join(out_dir, INDEX_NAME), "w", encoding="utf-8") as fh:
        json.dump(index, fh, indent=2)
    return index


def convert_checkpoint(source, out_dir, model_type="wan_dit", quantized=False, linear_dtype=torch.float8_e4m3fn, non_linear_dtype=torch.float32,
                       save_by_block=False, chunk_size=100, output_name="converted", device="cpu"):
    """converter.py:447-583 without key renaming / LoRA: read every *.safetensors (or .pt/.pth) under `source`, optionally quantise,
    write the chunked or per-block layout."""
    files = [source] if os.path.isfile(source) else sorted(glob.glob(os.path.join(source, "*.safetensors")) + glob.glob(os.path.join(source, "*.pth")) + glob.glob(os.path.join(source, "*.pt")))
    if not files:
        raise ValueError("No .pth, .pt, or .safetensors files found")
    merged = {}
    for path in files:
        if path.endswith(".safetensors"):
            with safe_open(path, framework="pt", device=device) as fh:
                part = {k: fh.get_tensor(k) for k in fh.keys()}
        else:
            part = torch.load(path, map_location=device, weights_only=True)
            if model_type == "hunyuan_dit" and "module" in part:
                part = part["module"]
        dup = set(part) & set(merged)
        if dup:
            raise ValueError(f"Duplicate keys found: {dup} in file {path}")
        merged.update(part)
    if quantized:
        sel = MODEL_TYPE_KEYS[model_type]
        quantize_model(merged, sel["target_keys"], sel["key_idx"], sel["ignore_key"], linear_dtype, non_linear_dtype)
    return save_checkpoint(merged, out_dir, save_by_block, chunk_size, output_name)


# ------------------------------------------------------------------------------------------------ reading
def _place(key, t, device, use_bf16):
    """model.py:77-79,116-123: fp32 tensors become bf16 unless the layer is one of SKIP_BF16 (and DTYPE is not all-bf16)."""
    if t.dtype == torch.float32 and (use_bf16 or all(s not in key for s in SKIP_BF16)):
        t = t.to(torch.bfloat16)
    return t.to(device)


def load_ckpt(model_path, device="cuda", use_bf16=True):
    """model.py:81-98: every *.safetensors in the directory (or its `original/` subdirectory).  Unlike the quantised loaders this one
    converts every tensor (not only fp32 ones) to bf16 unless skipped (model.py:77-79)."""
    files = glob.glob(os.path.join(model_path, "*.safetensors")) or glob.glob(os.path.join(model_path, "original", "*.safetensors"))
    if not files:
        raise FileNotFoundError(f"No .safetensors files found in directory: {model_path}")
    out = {}
    for path in files:
        with safe_open(path, framework="pt") as fh:
            for k in fh.keys():
                t = fh.get_tensor(k)
                out[k] = (t.to(torch.bfloat16) if use_bf16 or all(s not in k for s in SKIP_BF16) else t).to(device)
    return out


def load_quant_ckpt(ckpt_path, device="cuda", use_bf16=True):
    """model.py:100-126: index-driven load of a quantised directory (chunked or per-block files alike)."""
    idx = [f for f in os.listdir(ckpt_path) if f.endswith(".index.json")]
    if not idx:
        raise FileNotFoundError(f"No *.index.json found in {ckpt_path}")
    with open(os.path.join(ckpt_path, idx[0])) as fh:
        index = json.load(fh)
    out = {}
    for name in sorted(set(index["weight_map"].values())):
        with safe_open(os.path.join(ckpt_path, name), framework="pt") as fh:
            for k in fh.keys():
                out[k] = _place(k, fh.get_tensor(k), device, use_bf16)
    return out


def load_quant_split_ckpt(ckpt_path, device="cuda", use_bf16=True):
    """model.py:128-144: the non-block tensors of a save_by_block directory (blocks are read on demand by `load_block`)."""
    out = {}
    with safe_open(os.path.join(ckpt_path, "non_block.safetensors"), framework="pt", device="cpu") as fh:
        for k in fh.keys():
            out[k] = _place(k, fh.get_tensor(k), device, use_bf16)
    return out


def load_block(ckpt_path, block_index, device="cuda", use_bf16=True):
    """One `block_{i}.safetensors` of a save_by_block directory (the unit the reference's lazy loader streams,
    mm_weight.py:48-67 `load_from_disk`)."""
    out = {}
    with safe_open(os.path.join(ckpt_path, f"block_{block_index}.safetensors"), framework="pt", device="cpu") as fh:
        for k in fh.keys():
            out[k] = _place(k, fh.get_tensor(k), device, use_bf16)
    return out


def load_for_config(model_path, config, device="cuda", use_bf16=True):
    """The branch of WanModel.__init__/_init_weights (model.py:35-46,146-170): a non-default mm_type without `weight_auto_quant` reads
    the quantised directory `dit_quantized_ckpt` (default `<model_path>/<scheme>` with scheme = the mm_type's second dash-separated
    word, e.g. fp8 / int8); otherwise the plain bf16 directory."""
    mm = config.get("mm_config", {}) or {}
    mm_type = mm.get("mm_type", "Default")
    quantized = mm_type not in ("Default", "Hip-bf16")
    if quantized and not mm.get("weight_auto_quant", False):
        ckpt = config.get("dit_quantized_ckpt") or os.path.join(model_path, mm_type.split("-")[1])
        if config.get("lazy_load", False):
            d = load_quant_split_ckpt(ckpt, device, use_bf16)
            n_blocks = len(glob.glob(os.path.join(ckpt, "block_*.safetensors")))
            for i in range(n_blocks):
                d.update(load_block(ckpt, i, device, use_bf16))
            return d
        return load_quant_ckpt(ckpt, device, use_bf16)
    return load_ckpt(model_path, device, use_bf16)
