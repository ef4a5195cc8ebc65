"""Wan2.1 DiT denoise forward on the HIP kernels — host-side mirror of the reference's Wan network
(reference: lightx2v/models/networks/wan/{model.py, infer/{pre_infer,transformer_infer,post_infer}.py,
weights/{pre,post,transformer}_weights.py}).  Same class and method names, same checkpoint tensor names,
same config keys; the block is issued as a fixed sequence of 14 kernel launches:

  LN+modulate → q,k,v GEMMs → fused q/k RMSNorm+RoPE → attention → o GEMM (+gate-residual epilogue)
  → LN affine → q GEMM → RMSNorm ; k,v GEMMs on context → RMSNorm → cross attention → o GEMM (+residual)
  → LN+modulate → ffn_0 GEMM (+GELU) → ffn_2 GEMM (+gate-residual)

Compared with the reference's op-by-op loop this removes per block: the float64 RoPE round trip
(utils.py:107-115, ≈4 passes over 3 GB f64 copies at 720p), the per-block complex128 freqs table
(utils.py:7-20), three standalone residual passes and the standalone GELU pass over [S, F].
"""
import math

import torch

from . import lib
from .registry import ATTN_WEIGHT_REGISTER, CONV3D_WEIGHT_REGISTER, LN_WEIGHT_REGISTER, MM_WEIGHT_REGISTER, RMS_WEIGHT_REGISTER, TENSOR_REGISTER
from . import ops  # noqa: F401  (registers the HIP operator classes)
from .weight_module import WeightModule, WeightModuleList


def _weight_signature(*ops_):
    """(data_ptr, version) of every tensor the given operator objects hold: changes when a checkpoint is re-loaded into the same
    objects or a tensor is updated in place."""
    sig = []
    for op in ops_:
        for name in ("weight", "weight_scale", "bias"):
            t = getattr(op, name, None)
            if torch.is_tensor(t):
                sig.append((t.data_ptr(), t._version))
    return tuple(sig)


def _ln_then_mm_input(op, x, weight=None, bias=None, scale=None, shift=None, eps=1e-6):
    """LayerNorm feeding linear layers of operator class type(op): (bf16 activation | None, kwargs for op.apply).  Quantised classes get
    the activation quantised ONCE for all consumers — fused into the LayerNorm kernel where the class offers it (fp8) — instead of once
    inside every apply() (the reference quantises the same norm output three times for q, k, v: mm_weight.py:236-245)."""
    if hasattr(op, "layernorm_quantize"):
        return None, {"quantized": op.layernorm_quantize(x, weight, bias, scale, shift, eps)}
    n = lib.layernorm(x, weight, bias, scale=scale, shift=shift, eps=eps)
    if hasattr(op, "quantize_input"):
        return n, {"quantized": op.quantize_input(n)}
    return n, {}


def _cfg(config, key, default=None):
    try:
        return config[key]
    except (KeyError, TypeError):
        return default


# ------------------------------------------------------------------------------------------------ weights
class WanModulation(WeightModule):
    def __init__(self, block_index, task, mm_type, config, lazy_load=False, lazy_load_file=None):
        super().__init__()
        self.config = config
        self.add_module("modulation", TENSOR_REGISTER["Default"](f"blocks.{block_index}.modulation", lazy_load, lazy_load_file))


class WanSelfAttention(WeightModule):
    """reference: wan/weights/transformer_weights.py:109-215."""

    def __init__(self, block_index, task, mm_type, config, lazy_load=False, lazy_load_file=None):
        super().__init__()
        self.config = config
        p = f"blocks.{block_index}.self_attn"
        self.add_module("norm1", LN_WEIGHT_REGISTER["Default"]())
        for proj in ("q", "k", "v", "o"):
            self.add_module(f"self_attn_{proj}", MM_WEIGHT_REGISTER[mm_type](f"{p}.{proj}.weight", f"{p}.{proj}.bias", lazy_load, lazy_load_file))
        self.add_module("self_attn_norm_q", RMS_WEIGHT_REGISTER["sgl-kernel"](f"{p}.norm_q.weight", lazy_load, lazy_load_file))
        self.add_module("self_attn_norm_k", RMS_WEIGHT_REGISTER["sgl-kernel"](f"{p}.norm_k.weight", lazy_load, lazy_load_file))
        self.add_module("self_attn_1", ATTN_WEIGHT_REGISTER[config["self_attn_1_type"]]())


class WanCrossAttention(WeightModule):
    """reference: wan/weights/transformer_weights.py:218-313 (t2v and i2v)."""

    def __init__(self, block_index, task, mm_type, config, lazy_load=False, lazy_load_file=None):
        super().__init__()
        self.config = config
        b = f"blocks.{block_index}"
        p = f"{b}.cross_attn"
        self.add_module("norm3", LN_WEIGHT_REGISTER["Default"](f"{b}.norm3.weight", f"{b}.norm3.bias", lazy_load, lazy_load_file))
        for proj in ("q", "k", "v", "o"):
            self.add_module(f"cross_attn_{proj}", MM_WEIGHT_REGISTER[mm_type](f"{p}.{proj}.weight", f"{p}.{proj}.bias", lazy_load, lazy_load_file))
        self.add_module("cross_attn_norm_q", RMS_WEIGHT_REGISTER["sgl-kernel"](f"{p}.norm_q.weight", lazy_load, lazy_load_file))
        self.add_module("cross_attn_norm_k", RMS_WEIGHT_REGISTER["sgl-kernel"](f"{p}.norm_k.weight", lazy_load, lazy_load_file))
        self.add_module("cross_attn_1", ATTN_WEIGHT_REGISTER[config["cross_attn_1_type"]]())
        if task == "i2v":  # transformer_weights.py:285-312: K / V projections and k-norm of the 257 CLIP tokens, a second attention operator
            for proj in ("k_img", "v_img"):
                self.add_module(f"cross_attn_{proj}", MM_WEIGHT_REGISTER[mm_type](f"{p}.{proj}.weight", f"{p}.{proj}.bias", lazy_load, lazy_load_file))
            self.add_module("cross_attn_norm_k_img", RMS_WEIGHT_REGISTER["sgl-kernel"](f"{p}.norm_k_img.weight", lazy_load, lazy_load_file))
            self.add_module("cross_attn_2", ATTN_WEIGHT_REGISTER[_cfg(config, "cross_attn_2_type", config["cross_attn_1_type"])]())


class WanFFN(WeightModule):
    """reference: wan/weights/transformer_weights.py:316-366."""

    def __init__(self, block_index, task, mm_type, config, lazy_load=False, lazy_load_file=None):
        super().__init__()
        self.config = config
        b = f"blocks.{block_index}"
        self.add_module("norm2", LN_WEIGHT_REGISTER["Default"]())
        self.add_module("ffn_0", MM_WEIGHT_REGISTER[mm_type](f"{b}.ffn.0.weight", f"{b}.ffn.0.bias", lazy_load, lazy_load_file))
        self.add_module("ffn_2", MM_WEIGHT_REGISTER[mm_type](f"{b}.ffn.2.weight", f"{b}.ffn.2.bias", lazy_load, lazy_load_file))


class WanTransformerAttentionBlock(WeightModule):
    """reference: wan/weights/transformer_weights.py:33-87 — four compute phases."""

    def __init__(self, block_index, task, mm_type, config):
        super().__init__()
        self.block_index, self.config = block_index, config
        self.compute_phases = WeightModuleList([cls(block_index, task, mm_type, config) for cls in (WanModulation, WanSelfAttention, WanCrossAttention, WanFFN)])
        self.add_module("compute_phases", self.compute_phases)


class WanTransformerWeights(WeightModule):
    """reference: wan/weights/transformer_weights.py:14-30."""

    def __init__(self, config):
        super().__init__()
        self.blocks_num = config["num_layers"]
        self.task = config["task"]
        self.config = config
        mm_config = _cfg(config, "mm_config") or {}
        self.mm_type = mm_config.get("mm_type", "Default")
        self.blocks = WeightModuleList([WanTransformerAttentionBlock(i, self.task, self.mm_type, config) for i in range(self.blocks_num)])
        self.add_module("blocks", self.blocks)


class WanPreWeights(WeightModule):
    """reference: wan/weights/pre_weights.py:9-64 (t2v and i2v)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.add_module("patch_embedding", CONV3D_WEIGHT_REGISTER["hip_patch"]("patch_embedding.weight", "patch_embedding.bias", stride=(1, 2, 2)))
        for name in ("text_embedding.0", "text_embedding.2", "time_embedding.0", "time_embedding.2", "time_projection.1"):
            self.add_module(name.replace(".", "_"), MM_WEIGHT_REGISTER["Default"](f"{name}.weight", f"{name}.bias"))
        if config["task"] == "i2v":  # pre_weights.py:42-58: the CLIP-feature MLP img_emb = LayerNorm, Linear, GELU, Linear, LayerNorm
            self.add_module("proj_0", LN_WEIGHT_REGISTER["Default"]("img_emb.proj.0.weight", "img_emb.proj.0.bias"))
            self.add_module("proj_1", MM_WEIGHT_REGISTER["Default"]("img_emb.proj.1.weight", "img_emb.proj.1.bias"))
            self.add_module("proj_3", MM_WEIGHT_REGISTER["Default"]("img_emb.proj.3.weight", "img_emb.proj.3.bias"))
            self.add_module("proj_4", LN_WEIGHT_REGISTER["Default"]("img_emb.proj.4.weight", "img_emb.proj.4.bias"))


class WanPostWeights(WeightModule):
    """reference: wan/weights/post_weights.py:9-18."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.register_parameter("norm", LN_WEIGHT_REGISTER["Default"]())
        self.add_module("head", MM_WEIGHT_REGISTER["Default"]("head.head.weight", "head.head.bias"))
        self.register_parameter("head_modulation", TENSOR_REGISTER["Default"]("head.modulation"))


# ------------------------------------------------------------------------------------------------ infer
def rope_params(max_seq_len, dim, theta=10000):
    """reference: wan/infer/utils.py:151-158 (float64 angles)."""
    return torch.outer(torch.arange(max_seq_len), 1.0 / torch.pow(theta, torch.arange(0, dim, 2).to(torch.float64).div(dim)))


def rope_cos_sin_table(head_dim, device):
    """The reference's complex128 `freqs` [1024, d/2] (pre_infer.py:12-19) as a float32 (cos, sin) table
    [1024, 64, 2] — built once; the kernel indexes it by each token's (t,h,w) instead of materialising the
    per-block [S,1,64] complex table."""
    d = head_dim
    ang = torch.cat([rope_params(1024, d - 4 * (d // 6)), rope_params(1024, 2 * (d // 6)), rope_params(1024, 2 * (d // 6))], dim=1)
    return torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1).to(torch.float32).contiguous().to(device)


class WanPreInfer:
    """reference: wan/infer/pre_infer.py:6-120 (t2v and i2v)."""

    def __init__(self, config):
        d = config["dim"] // config["num_heads"]
        assert config["dim"] % config["num_heads"] == 0 and d % 2 == 0
        if d != 128:
            raise lib.X2VError(f"head_dim {d}: the HIP attention/RoPE kernels are built for head_dim 128")
        self.task = config["task"]
        self.freq_dim = config["freq_dim"]
        self.dim = config["dim"]
        self.text_len = config["text_len"]
        self.freqs = None  # float32 (cos,sin) table, created on first use on the right device
        self.head_dim = d
        # step-invariant text path: the text MLP output per input context list, see _text_context
        self.cache_text_context = bool(_cfg(config, "cache_cross_kv", True))
        self._text_cache = {}

    def set_scheduler(self, scheduler):
        self.scheduler = scheduler

    def infer(self, weights, inputs, positive, kv_start=0, kv_end=0):
        sch = self.scheduler
        latents = sch.latents
        dev = latents.device
        if self.freqs is None or self.freqs.device != dev:
            self.freqs = rope_cos_sin_table(self.head_dim, dev)
        t = torch.stack([sch.timesteps[sch.step_index]])
        context = inputs["text_encoder_output"]["context" if positive else "context_null"]
        seq_len = sch.seq_len

        if self.task == "i2v":  # pre_infer.py:44-55: noise latents + [first-frame mask | conditioning latents] along the channel axis (36 channels)
            latents = torch.cat([latents, inputs["image_encoder_output"]["vae_encode_out"].to(latents.dtype)], dim=0)
        x = weights.patch_embedding.apply(latents.unsqueeze(0))  # [1, D, T, H/2, W/2] (pre_infer.py:57)
        grid_sizes = torch.tensor([list(x.shape[2:])], dtype=torch.long)
        x = x.flatten(2).transpose(1, 2).squeeze(0)  # :59 — a view chain back onto the GEMM's contiguous [S, D]
        s = x.shape[0]
        assert s <= seq_len
        if s < seq_len:
            x = torch.cat([x, x.new_zeros(seq_len - s, x.shape[1])], dim=0)
        seq_lens = torch.tensor([s], dtype=torch.long)

        embed = lib.sinusoid_embed(t.flatten().to(dev), self.freq_dim)
        embed = weights.time_embedding_0.apply(embed, epilogue=lib.EPI_SILU)  # Linear + SiLU (pre_infer.py:70-74)
        embed = weights.time_embedding_2.apply(embed)
        embed0 = lib.activation(embed, lib.EPI_SILU)
        embed0 = weights.time_projection_1.apply(embed0).unflatten(1, (6, self.dim))

        context = self.full_context(weights, inputs, positive)
        return embed, grid_sizes, (x, embed0.squeeze(0), seq_lens, self.freqs, context)

    I2V_CLIP_TOKENS = 257  # transformer_infer.py:406-407

    def full_context(self, weights, inputs, positive):
        """The cross-attention context of one CFG branch: the text MLP output (t2v), with the CLIP-feature MLP output in front of it for i2v
        (pre_infer.py:100-113).  Both parts are step-invariant; with `cache_cross_kv` the SAME tensor object is returned on every step, which is
        what lets the block stack reuse its cross-attention K / V (text and image) across the denoise loop."""
        text = self._text_context(weights, inputs["text_encoder_output"]["context" if positive else "context_null"])
        if self.task != "i2v":
            return text
        clip = inputs["image_encoder_output"]["clip_encoder_out"]
        key = (id(weights), id(clip), id(text))
        sig = _weight_signature(weights.proj_1, weights.proj_3)
        hit = self._text_cache.get(key) if self.cache_text_context else None
        if hit is not None and hit[3] is weights and hit[4] == sig and hit[0][0] is clip and hit[0][1] is text and hit[1] == clip._version:
            return hit[2]
        c = lib.layernorm(clip.to(text.dtype), weights.proj_0.weight, weights.proj_0.bias, eps=weights.proj_0.eps)
        c = lib.activation(weights.proj_1.apply(c), lib.ACT_GELU_ERF)  # Linear, exact GELU (:104-106)
        c = weights.proj_3.apply(c)
        c = lib.layernorm(c, weights.proj_4.weight, weights.proj_4.bias, eps=weights.proj_4.eps)
        out = torch.cat([c, text], dim=0)  # [257 + text_len, D]: once per prompt, not per step
        if self.cache_text_context:
            if len(self._text_cache) >= 8:
                self._text_cache.pop(next(iter(self._text_cache)))
            self._text_cache[key] = ((clip, text), clip._version, out, weights, sig)
        return out

    def _text_context(self, weights, context):
        """pre_infer.py:86-96: pad the T5 rows to text_len, Linear + GELU-tanh + Linear.  The prompt embeddings do not change during a
        denoise loop, so with `cache_cross_kv` the result is computed once per input tensor list and the SAME output tensor object is
        handed to the block stack on every step — which is what lets the transformer reuse each block's cross-attention K/V.  Cache
        entries pin their input tensors AND the weights object (so an id cannot be recycled) and check the version counters of the
        inputs and the (data_ptr, version) of the two weight tensors, so a re-loaded or in-place updated checkpoint (the reference's
        LoRA-switch path re-runs `_init_weights`) is never served stale; at most 4 contexts are kept."""
        key = None
        if self.cache_text_context:
            key = (id(weights),) + tuple(id(u) for u in context)
            sig = _weight_signature(weights.text_embedding_0, weights.text_embedding_2)
            hit = self._text_cache.get(key)
            if hit is not None and hit[3] is weights and hit[4] == sig and all(u._version == ver for u, ver in zip(hit[0], hit[1])):
                return hit[2]
        stacked = torch.stack([torch.cat([u, u.new_zeros(self.text_len - u.size(0), u.size(1))]) for u in context]).squeeze(0)
        out = weights.text_embedding_0.apply(stacked, epilogue=lib.EPI_GELU_TANH)
        out = weights.text_embedding_2.apply(out)
        if key is not None:
            if len(self._text_cache) >= 4:
                self._text_cache.pop(next(iter(self._text_cache)))
            self._text_cache[key] = (list(context), [u._version for u in context], out, weights, sig)
        return out

    def clear_text_cache(self):
        self._text_cache.clear()


# Staggered key walk of the single-GPU self-attention launches (x2v.h X2V_ATTN_VT_STAGGER: query block b starts its walk (b mod 8) tiles in).  Round 3
# measured it +1.3 % at Wan-14B 720p in 2-step runs and turned it on; at SUSTAINED load (6-8 timed steps, A-B-A-B on one box, round 4:
# profiles/r04_call12_stagger_abab_sustained.txt) it is 0.9 % SLOWER — 9400.6 / 9417.4 ms per step without it, 9488.4 / 9485.7 with it (attention 164.3-164.7
# vs 166.5 ms per paired launch) — so the fused Wan drivers no longer set it.  One switch for every single-GPU launch form (pair pass, two streams,
# sequential), which keeps them bit-identical to each other; the flag stays in the C-ABI.
SELF_ATTN_STAGGER = False


class WanTransformerInfer:
    """reference: wan/infer/transformer_infer.py:12-508 (no-offload path).  `parallel_attention`
    (set by lightx2v_amd.ulysses.parallelize_wan) replaces the local self-attention exactly where the
    reference injects it (:381-388)."""

    def __init__(self, config):
        self.config = config
        self.task = config["task"]
        self.attention_type = _cfg(config, "attention_type", "hip_flash")
        self.blocks_num = config["num_layers"]
        self.phases_num = 4
        self.num_heads = config["num_heads"]
        self.head_dim = config["dim"] // config["num_heads"]
        self.parallel_attention = None
        self.sp_rank, self.sp_world = 0, 1
        self.round_mode = lib.ROUND_REF if _cfg(config, "hip_ref_rounding", False) else lib.ROUND_FP32
        self.infer_conditional = True
        self._rope_cs = None
        self.attn_time_hook = None  # bench.py: callable(kind) -> context manager timing the attention launch
        self.cache_cross_kv = bool(_cfg(config, "cache_cross_kv", True))
        self._cross_kv_cache = {}  # (id(block weights), id(context)) -> (context, version, k, v); see _cross_kv
        # CFG pair mode (WanModel._forward_pair): x holds BOTH forwards of a step, [cond rows | pad | uncond rows | pad], (S, S_pad) here
        self._pair = None

    def set_scheduler(self, scheduler):
        self.scheduler = scheduler

    def switch_status(self):
        self.infer_conditional = not self.infer_conditional

    def infer(self, weights, grid_sizes, embed, x, embed0, seq_lens, freqs, context, audio_dit_blocks=None):
        return self._infer_without_offload(weights, grid_sizes, embed, x, embed0, seq_lens, freqs, context)

    def _infer_without_offload(self, weights, grid_sizes, embed, x, embed0, seq_lens, freqs, context, audio_dit_blocks=None):
        for block_idx in range(self.blocks_num):
            x = self.infer_block(weights.blocks[block_idx], grid_sizes, embed, x, embed0, seq_lens, freqs, context)
        return x

    def infer_block(self, weights, grid_sizes, embed, x, embed0, seq_lens, freqs, context):
        shift_msa, scale_msa, gate_msa, c_shift_msa, c_scale_msa, c_gate_msa = self.infer_modulation(weights.compute_phases[0], embed0)
        x = self.infer_self_attn(weights.compute_phases[1], grid_sizes, x, seq_lens, freqs, shift_msa, scale_msa, gate_msa)
        x = self.infer_cross_attn(weights.compute_phases[2], x, context)
        x = self.infer_ffn(weights.compute_phases[3], x, c_shift_msa, c_scale_msa, c_gate_msa)
        return x

    def infer_modulation(self, weights, embed0):
        # [1,6,D] + [6,D] → six [1,D] rows (transformer_infer.py:308-319); 6*D elements: host-side plumbing
        return (weights.modulation.tensor + embed0).chunk(6, dim=1)

    def infer_self_attn(self, weights, grid_sizes, x, seq_lens, freqs, shift_msa, scale_msa, gate_msa):
        """transformer_infer.py:321-396 + the `x.add_(y * gate_msa)` of :402 folded into the o-projection."""
        # norm1 has no affine (transformer_weights.py:127-129); quantised operator classes get n1 quantised once for q, k and v
        n1, mmkw = _ln_then_mm_input(weights.self_attn_q, x, scale=scale_msa, shift=shift_msa, eps=weights.norm1.eps)
        grid = tuple(int(g) for g in grid_sizes[0].tolist())
        s_local = x.shape[0]
        if freqs.is_complex():  # driven by the reference's WanPreInfer: complex128 [1024, 64] (pre_infer.py:12-19)
            if self._rope_cs is None or self._rope_cs.device != x.device:
                self._rope_cs = torch.stack([freqs.real, freqs.imag], dim=-1).to(torch.float32).contiguous().to(x.device)
            freqs = self._rope_cs
        # default (fp32-statistics) mode: q leaves the norm+RoPE kernel already multiplied by softmax_scale*log2(e) inside
        # its one rounding, and the attention kernel variant that expects that skips the per-score FMA (x2v.h)
        fast = self.round_mode == lib.ROUND_FP32
        variant = (lib.ATTN_FAST | lib.ATTN_Q_PRESCALED) if fast else 0
        rope_args = dict(s0=self.sp_rank * s_local, eps=weights.self_attn_norm_q.eps, round_mode=self.round_mode, q_out_scale=lib.ATTN_PRESCALE if fast else 1.0)
        pa = self.parallel_attention
        if pa is not None and hasattr(pa, "attend_blocked") and x.is_cuda and self.blocked_exchange and self._blocked_ok(weights):
            # Ulysses without layout copies: the exchange buffers [N, S/N, (H/N)d] are kernel operands (ulysses.py)
            b = pa.buffers(s_local, x.shape[1], x.dtype, x.device)
            weights.self_attn_v.apply(n1, out=b["sv"], **mmkw)  # v needs no norm / RoPE: projected first, straight into its send buffer,
            v_pending = pa.begin_exchange_blocked(b["sv"], b["rv"])  # and exchanged under the q / k projections and the norm+RoPE kernel
            q = weights.self_attn_q.apply(n1, **mmkw)
            k = weights.self_attn_k.apply(n1, **mmkw)
            lib.rmsnorm_rope_blocked(q, k, weights.self_attn_norm_q.weight, weights.self_attn_norm_k.weight, freqs, grid, self.num_heads, b["sq"], b["sk"], **rope_args)
            attn = pa.attend_blocked(b, self.num_heads, self.head_dim, timer=self._timed, variant=variant, v_pending=v_pending)
            return weights.self_attn_o.apply(attn, epilogue=lib.EPI_RESIDUAL, resid=x, gate=gate_msa)  # K-blocked x
        # Ulysses, row-major form: v is projected first and its seq→head exchange runs on the communication stream under the q and k
        # projections and the norm+RoPE kernel
        if self._pair is not None:
            # both CFG forwards in one pass: projections and row kernels run on the stacked rows, the two self-attentions are one launch
            S, Sp = self._pair
            if not mmkw and hasattr(weights.self_attn_v, "apply_vt"):
                vt = weights.self_attn_v.apply_vt(n1, self.num_heads)
            else:
                vt = lib.transpose_heads(weights.self_attn_v.apply(n1, **mmkw), self.num_heads)
            q = weights.self_attn_q.apply(n1, **mmkw)
            k = weights.self_attn_k.apply(n1, **mmkw)
            for b in range(2):  # token b*Sp + i of either forward sits at grid position i
                rows = slice(b * Sp, b * Sp + S)
                lib.rmsnorm_rope_(q[rows], k[rows], weights.self_attn_norm_q.weight, weights.self_attn_norm_k.weight, freqs, grid, self.num_heads, **rope_args)
            attn = lib.attention_batched(q, k, vt, self.num_heads, 2, Sp, S, prescaled=True, stagger=SELF_ATTN_STAGGER, timed=lambda fn: self._timed("self", fn))
            return weights.self_attn_o.apply(attn, epilogue=lib.EPI_RESIDUAL, resid=x, gate=gate_msa)
        if pa is None and fast and not mmkw and hasattr(weights.self_attn_v, "apply_vt"):
            v, vt = None, weights.self_attn_v.apply_vt(n1, self.num_heads)  # V^T from the v projection's epilogue (the attention kernel's operand)
        else:
            v, vt = weights.self_attn_v.apply(n1, **mmkw), None
        v_pending = pa.begin_exchange(v) if hasattr(pa, "begin_exchange") else None
        q = weights.self_attn_q.apply(n1, **mmkw)
        k = weights.self_attn_k.apply(n1, **mmkw)
        lib.rmsnorm_rope_(q, k, weights.self_attn_norm_q.weight, weights.self_attn_norm_k.weight, freqs, grid, self.num_heads, **rope_args)
        if pa is None:
            # the ping-pong kernel reads V^T; transposed outside the timed launch so the hook times the attention kernel alone
            if vt is None and fast:
                vt = lib.transpose_heads(v, self.num_heads)
            # single GPU: the same key-walk form as the pair pass (SELF_ATTN_STAGGER above), so the launch forms stay bit-identical
            attn = self._timed("self", lambda: lib.attention(q, k, v, self.num_heads, self.head_dim, variant=variant | (lib.ATTN_STAGGER if (fast and SELF_ATTN_STAGGER) else 0), vt=vt))
        else:
            attn = pa(q=q, k=k, v=v if v_pending is None else v_pending, num_heads=self.num_heads, head_dim=self.head_dim, timer=self._timed, variant=variant)
        return weights.self_attn_o.apply(attn, epilogue=lib.EPI_RESIDUAL, resid=x, gate=gate_msa)

    blocked_exchange = True  # Ulysses: exchange buffers as kernel operands (False: the reference's row-major form with its transposing copies)

    @staticmethod
    def _blocked_ok(weights):
        """The copy-free Ulysses path needs operator objects whose apply() takes block-strided operands (the bf16 and the w8a8 classes)."""
        return all(getattr(getattr(weights, n), "accepts_blocked", False) for n in ("self_attn_v", "self_attn_o"))

    def infer_cross_attn(self, weights, x, context):
        """transformer_infer.py:398-465 (t2v) + the `x.add_(attn_out)` of :468 folded into the o-projection."""
        n3, mmkw = _ln_then_mm_input(weights.cross_attn_q, x, weights.norm3.weight, weights.norm3.bias, eps=weights.norm3.eps)
        q = weights.cross_attn_q.apply(n3, **mmkw)
        lib.rmsnorm(q, weights.cross_attn_norm_q.weight, weights.cross_attn_norm_q.eps, out=q, round_mode=self.round_mode)
        i2v = self.task == "i2v"  # :405-407,437-455: the first 257 context rows are CLIP tokens with their own K / V projections and a second attention

        def attend(qr, ctx, out=None):
            k, v, vt = self._cross_kv(weights, ctx, "text" if i2v else None)
            kw = dict(variant=lib.ATTN_FAST, vt=vt) if vt is not None else {}
            a = self._timed("cross", lambda: lib.attention(qr, k, v, self.num_heads, self.head_dim, out=out, **kw))
            if i2v:
                k, v, vt = self._cross_kv(weights, ctx, "img")
                kw = dict(variant=lib.ATTN_FAST, vt=vt) if vt is not None else {}
                a_img = self._timed("cross", lambda: lib.attention(qr, k, v, self.num_heads, self.head_dim, **kw))
                lib.gate_residual_(a, a_img)  # attn_out.add_(img_attn_out) (:451), in the activation dtype
            return a

        if self._pair is not None:
            S, Sp = self._pair
            attn = torch.empty_like(q)
            for b, ctx in enumerate(context):  # (conditional, unconditional) contexts; every row of a forward's slot is a query
                rows = slice(b * Sp, (b + 1) * Sp)
                attend(q[rows], ctx, out=attn[rows])
            return weights.cross_attn_o.apply(attn, epilogue=lib.EPI_RESIDUAL, resid=x, gate=None)
        return weights.cross_attn_o.apply(attend(q, context), epilogue=lib.EPI_RESIDUAL, resid=x, gate=None)

    def _cross_kv(self, weights, context, part=None):
        """k = RMSNorm(W_k context), v = W_v context (transformer_infer.py:419-424); i2v (`part` "text" / "img"): of the context rows behind / in
        front of the 257 CLIP tokens, the image part through k_img / v_img / norm_k_img (:437-440) — cached under the same context object.  The text context and the weights do not change
        between denoise steps, so with config `cache_cross_kv` (default on) each block's pair is computed once per context
        tensor OBJECT (WanPreInfer hands the same object over on every step) and reused — the same values the reference recomputes
        every step (0.8 GB for Wan-14B with CFG).  An entry pins its context tensor and its weights object (their ids cannot be recycled)
        and checks the context's version counter and the (data_ptr, version) of the k / v / norm_k weight tensors; contexts other than
        the two most recent ones (cond / uncond) are evicted, so a caller that passes fresh tensors every step gets the reference
        behaviour without growth.  `WanModel._init_weights` clears the cache as well."""
        n_clip = WanPreInfer.I2V_CLIP_TOKENS
        if part == "img":
            op_k, op_v, op_n, rows = weights.cross_attn_k_img, weights.cross_attn_v_img, weights.cross_attn_norm_k_img, slice(0, n_clip)
        else:
            op_k, op_v, op_n, rows = weights.cross_attn_k, weights.cross_attn_v, weights.cross_attn_norm_k, (slice(n_clip, None) if part == "text" else slice(None))

        def compute():
            src = context[rows]
            k = op_k.apply(src)
            lib.rmsnorm(k, op_n.weight, op_n.eps, out=k, round_mode=self.round_mode)
            v = op_v.apply(src)
            # V^T for the ping-pong attention kernel (0.96 vs 1.20 ms per launch at 14B 720p on the 512-key context)
            return k, v, (lib.transpose_heads(v, self.num_heads) if (self.round_mode == lib.ROUND_FP32 and v.is_cuda) else None)

        if not self.cache_cross_kv:
            return compute()
        per_ctx = self._cross_kv_cache.get(id(context))
        if per_ctx is None or per_ctx["ctx"] is not context or per_ctx["version"] != context._version:
            while len(self._cross_kv_cache) >= 2:
                self._cross_kv_cache.pop(next(iter(self._cross_kv_cache)))
            per_ctx = self._cross_kv_cache[id(context)] = {"ctx": context, "version": context._version, "kv": {}}
        sig = _weight_signature(op_k, op_v, op_n)
        hit = per_ctx["kv"].get((id(weights), part))
        if hit is None or hit[2] is not weights or hit[3] != sig:  # the entry pins its weights object; tensors re-loaded / edited in place: recompute
            k, v, vt = compute()
            hit = per_ctx["kv"][(id(weights), part)] = (k, v, weights, sig, vt)
        return hit[0], hit[1], hit[4]

    def clear_cross_kv(self):
        self._cross_kv_cache.clear()

    def infer_ffn(self, weights, x, c_shift_msa, c_scale_msa, c_gate_msa):
        """transformer_infer.py:467-508: LN+modulate, ffn_0 (+GELU-tanh), ffn_2 (+`x.add_(y * c_gate)`)."""
        n2, mmkw = _ln_then_mm_input(weights.ffn_0, x, scale=c_scale_msa, shift=c_shift_msa, eps=weights.norm2.eps)
        h = weights.ffn_0.apply(n2, epilogue=lib.EPI_GELU_TANH, **mmkw)
        return weights.ffn_2.apply(h, epilogue=lib.EPI_RESIDUAL, resid=x, gate=c_gate_msa)

    def _timed(self, kind, fn):
        if self.attn_time_hook is None:
            return fn()
        with self.attn_time_hook(kind):
            return fn()



class WanTransformerInferTeaCaching(WanTransformerInfer):
    """reference: wan/infer/feature_caching/transformer_infer.py:9-171 — TeaCache around the fused block stack.
    Per CFG branch the polynomial-rescaled relative L1 change of the modulation input (embed0 with `use_ret_steps`,
    else embed; a [6, D] / [1, D] tensor — the comparison is host-side control flow exactly as in the reference,
    one .item() per forward) is accumulated; below `teacache_thresh` the block stack is skipped and the previous
    residual x_out - x_in of that branch is re-applied.  The two [S, D] elementwise passes (residual capture, re-apply)
    run on the gate-residual kernel (gate = -1 / none)."""

    def __init__(self, config):
        super().__init__(config)
        self.cnt = 0
        self.teacache_thresh = config["teacache_thresh"]
        self.use_ret_steps = config["use_ret_steps"]
        self.coefficients = config["coefficients"][0 if self.use_ret_steps else 1]
        self.ret_steps = 5 * 2 if self.use_ret_steps else 1 * 2
        self.cutoff_steps = config["infer_steps"] * 2 if self.use_ret_steps else config["infer_steps"] * 2 - 2
        self.accumulated_rel_l1_distance_even = self.accumulated_rel_l1_distance_odd = 0
        self.previous_e0_even = self.previous_e0_odd = None
        self.previous_residual_even = self.previous_residual_odd = None
        self._minus_one = None

    def calculate_should_calc(self, embed, embed0):
        import numpy as np

        inp = embed0 if self.use_ret_steps else embed
        tag = "even" if self.infer_conditional else "odd"
        if self.cnt < self.ret_steps or self.cnt >= self.cutoff_steps:
            should_calc = True
            setattr(self, f"accumulated_rel_l1_distance_{tag}", 0)
        else:
            prev = getattr(self, f"previous_e0_{tag}")
            acc = getattr(self, f"accumulated_rel_l1_distance_{tag}") + np.poly1d(self.coefficients)(((inp - prev).abs().mean() / prev.abs().mean()).cpu().item())
            should_calc = not (acc < self.teacache_thresh)
            setattr(self, f"accumulated_rel_l1_distance_{tag}", 0 if should_calc else acc)
        setattr(self, f"previous_e0_{tag}", inp.clone())
        return should_calc

    def infer(self, weights, grid_sizes, embed, x, embed0, seq_lens, freqs, context, audio_dit_blocks=None):
        index = self.scheduler.step_index
        records = self.scheduler.caching_records if self.infer_conditional else self.scheduler.caching_records_2
        if index <= self.scheduler.infer_steps - 1:
            records[index] = self.calculate_should_calc(embed, embed0)
        tag = "even" if self.infer_conditional else "odd"
        if records[index]:
            ori_x = x.clone()
            x = super().infer(weights, grid_sizes, embed, x, embed0, seq_lens, freqs, context)
            if self._minus_one is None or self._minus_one.device != x.device:
                self._minus_one = torch.full((x.shape[1],), -1.0, dtype=x.dtype, device=x.device)
            res = x.clone()
            lib.gate_residual_(res, ori_x, self._minus_one)  # res = x_out - x_in (bf16, as the reference's tensor subtraction)
            setattr(self, f"previous_residual_{tag}", res)
        else:
            lib.gate_residual_(x, getattr(self, f"previous_residual_{tag}"))  # x.add_(previous_residual)
        if _cfg(self.config, "enable_cfg", True):
            self.switch_status()
        self.cnt += 1
        return x

    def clear(self):
        self.previous_residual_even = self.previous_residual_odd = None
        self.previous_e0_even = self.previous_e0_odd = None

class WanPostInfer:
    """reference: wan/infer/post_infer.py:6-50."""

    def __init__(self, config):
        self.out_dim = config["out_dim"]
        self.patch_size = (1, 2, 2)

    def set_scheduler(self, scheduler):
        self.scheduler = scheduler

    def infer_rows(self, weights, x, e):
        """LN + modulate + head Linear on token rows: [rows, D] -> [rows, 64] (post_infer.py:15-40)."""
        e = (weights.head_modulation.tensor + e.unsqueeze(1)).chunk(2, dim=1)  # [1,2,D] → shift, scale
        x = lib.layernorm(x, scale=e[1].squeeze(0), shift=e[0].squeeze(0), eps=weights.norm.eps)
        return weights.head.apply(x)

    def infer(self, weights, x, e, grid_sizes):
        return [u.float() for u in self.unpatchify(self.infer_rows(weights, x, e), grid_sizes)]

    def unpatchify(self, x, grid_sizes):
        c = self.out_dim
        out = []
        for v in grid_sizes.tolist():
            u = x[: math.prod(v)].view(*v, *self.patch_size, c)
            u = torch.einsum("fhwpqrc->cfphqwr", u)
            out.append(u.reshape(c, *[i * j for i, j in zip(v, self.patch_size)]))
        return out


# ------------------------------------------------------------------------------------------------ model
def cfg_form_by_size(seq_len, num_heads):
    """How one GPU runs the two forwards of a CFG step (wan/model.py:197-226 runs one after the other) when the config leaves it to the size
    ("auto"): by the self-attention workgroups of ONE forward (256-row query blocks x heads; the chip holds 512 at a time).  Measured on MI355X,
    step time against one forward after the other (profiles/r03_cfg_two_streams_ab.txt, r02_cfg_pair_ab.log):
        960 (Wan-1.3B 480p)   two streams -3.9 ... -7.8 %   pair pass +1 %
       1536 (Wan-1.3B 480p x 81f) two streams -1.0 %        pair pass -0.8 %      (both inside the +-1 % run-to-run noise of one box)
       3552 (Wan-1.3B 720p)   two streams +0.8 %            pair pass -0.1 %
       5120 (Wan-14B 480p)    two streams +3.7 %            pair pass -0.2 %
      11840 (Wan-14B 720p)    two streams +2.7 ... +3.5 %   pair pass -0.8 %
    TWO forms, one threshold (round 4: the third band — "one forward after the other" between 2048 and 4096 — rested on differences inside the
    noise and was one more code path for the parity suite; it is still what `cfg_pair=False, cfg_branch_streams=False` gives): 'streams'
    (CfgBranchStreams) where a launch is a few part-empty rounds of the chip, 'pair' (WanModel._forward_pair) from 2048 workgroups on, where the pair
    pass never measured worse than -0.1 % and the streams never better than +0.8 %.  Both are bit-identical to the sequential order
    (tests/test_gpu_model.py)."""
    workgroups = ((int(seq_len) + 255) // 256) * int(num_heads)
    return "pair" if workgroups >= 2048 else "streams"


class CfgBranchStreams:
    """The conditional and unconditional forwards of a CFG step (wan/model.py:197-226) enqueued block by block on two compute streams.

    The two forwards share latents, timestep and weights and are independent until the CFG combine.  Run one after the other, every launch
    that does not divide into whole rounds of the 256 CUs leaves its last round part empty (Wan-1.3B 480p: a self-attention is 960 workgroups
    for 512 slots, a D x D projection 480 tiles for 256 CUs) and nothing can use the idle CUs; with the two branches on two streams the
    hardware fills one branch's tail with the other's next kernel.  Every kernel sees the same operands as in the sequential order, so
    the result is bit-identical (tests/test_gpu_model.py).  Large shapes take the pair pass instead (WanModel._pair_ok): there a launch is
    tens of rounds and one stacked launch is the better form.  lightx2v_amd.ulysses derives the sequence-parallel form (whose gain is the
    exchange of one branch under the kernels of the other)."""

    def __init__(self, wan_model):
        self.model = wan_model
        self.enabled = True
        self._streams = None

    def _attention_ok(self, tr):
        return tr.parallel_attention is None

    def usable(self, inputs):
        m = self.model
        tr = m.transformer_infer
        return self.enabled and m.config["enable_cfg"] and m.scheduler.latents.is_cuda and type(tr) is WanTransformerInfer and self._attention_ok(tr)

    def _setup(self):
        if self._streams is None:
            self._streams = (torch.cuda.Stream(), torch.cuda.Stream())
        return self._streams

    # hooks of the sequence-parallel form
    def _shard(self, x):
        return x

    def _gather(self, x):
        return x

    def _enter_branch(self, tr, b):
        pass

    def _leave(self, tr):
        pass

    def forward_pair(self, inputs):
        """Returns (cond, uncond) noise predictions, each a list-less fp32 tensor as WanModel._forward returns."""
        m = self.model
        tr = m.transformer_infer
        sa, sb = self._setup()
        cur = torch.cuda.current_stream()
        embed, grid_sizes, (x, embed0, seq_lens, freqs, ctx_c) = m.pre_infer.infer(m.pre_weight, inputs, positive=True)
        ctx_u = m.pre_infer.full_context(m.pre_weight, inputs, False)
        xa = self._shard(x)
        xb = xa.clone()
        sa.wait_stream(cur)
        sb.wait_stream(cur)
        blocks = m.transformer_weights.blocks
        try:
            for i in range(tr.blocks_num):
                for b, (st, ctx) in enumerate(((sa, ctx_c), (sb, ctx_u))):
                    with torch.cuda.stream(st):
                        self._enter_branch(tr, b)
                        if b == 0:
                            xa = tr.infer_block(blocks[i], grid_sizes, embed, xa, embed0, seq_lens, freqs, ctx)
                        else:
                            xb = tr.infer_block(blocks[i], grid_sizes, embed, xb, embed0, seq_lens, freqs, ctx)
        finally:
            self._leave(tr)
        outs = []
        for st, xs in ((sa, xa), (sb, xb)):
            with torch.cuda.stream(st):
                outs.append(m.post_infer.infer(m.post_weight, self._gather(xs), embed, grid_sizes)[0])
        cur.wait_stream(sa)
        cur.wait_stream(sb)
        # allocator bookkeeping: xa / xb were allocated under `cur` and used under a branch stream, the outputs the other way round (the joins
        # above already order every later use behind both branches)
        xa.record_stream(sa)
        xb.record_stream(sb)
        for t in outs:
            t.record_stream(cur)
        return outs[0], outs[1]


class WanModel:
    """reference: wan/model.py:28-226.  Built from an in-memory checkpoint dict (the reference's
    `_init_weights(weight_dict)` path, :146-170); tensors must already be on the target device."""

    pre_weight_class = WanPreWeights
    post_weight_class = WanPostWeights
    transformer_weight_class = WanTransformerWeights

    def __init__(self, config, weight_dict, device="cuda"):
        self.config = config
        self.device = device
        self._init_infer_class()
        self._init_weights(weight_dict)
        self._init_infer()
        pat = _cfg(config, "parallel_attn_type")
        if pat:
            if pat != "ulysses":
                raise NotImplementedError(f"parallel_attn_type={pat}: only 'ulysses' is built (the north star's SP scheme)")
            from . import ulysses

            ulysses.parallelize_wan(self)

    def _init_infer_class(self):
        fc = _cfg(self.config, "feature_caching", "NoCaching")  # reference: wan/model.py:61-75
        if fc == "NoCaching":
            tr_cls = WanTransformerInfer
        elif fc == "Tea":
            tr_cls = WanTransformerInferTeaCaching
        else:
            raise NotImplementedError(f"feature_caching={fc}: only 'NoCaching' and 'Tea' are built")
        self.pre_infer_class, self.post_infer_class, self.transformer_infer_class = WanPreInfer, WanPostInfer, tr_cls

    def _init_weights(self, weight_dict):
        self.original_weight_dict = weight_dict
        self.pre_weight = self.pre_weight_class(self.config)
        self.post_weight = self.post_weight_class(self.config)
        self.transformer_weights = self.transformer_weight_class(self.config)
        self.pre_weight.load(weight_dict)
        self.post_weight.load(weight_dict)
        self.transformer_weights.load(weight_dict)
        # step-invariant caches are keyed by weight objects: a re-load (the reference's LoRA switch re-runs this method) drops them
        if getattr(self, "transformer_infer", None) is not None:
            self.transformer_infer.clear_cross_kv()
        if getattr(self, "pre_infer", None) is not None:
            self.pre_infer.clear_text_cache()

    def _init_infer(self):
        self.pre_infer = self.pre_infer_class(self.config)
        self.post_infer = self.post_infer_class(self.config)
        self.transformer_infer = self.transformer_infer_class(self.config)

    def set_scheduler(self, scheduler):
        self.scheduler = scheduler
        self.pre_infer.set_scheduler(scheduler)
        self.post_infer.set_scheduler(scheduler)
        self.transformer_infer.set_scheduler(scheduler)

    def to_cpu(self):
        for w in (self.pre_weight, self.post_weight, self.transformer_weights):
            w.to_cpu()

    def to_cuda(self):
        for w in (self.pre_weight, self.post_weight, self.transformer_weights):
            w.to_cuda()

    def _forward(self, inputs, positive):
        embed, grid_sizes, pre_infer_out = self.pre_infer.infer(self.pre_weight, inputs, positive=positive)
        x = self.transformer_infer.infer(self.transformer_weights, grid_sizes, embed, *pre_infer_out)
        return self.post_infer.infer(self.post_weight, x, embed, grid_sizes)[0]

    def _pair_ok(self, inputs):
        """The two forwards of a CFG step as ONE pass over [cond tokens | uncond tokens] (config `cfg_pair`: True / False / "auto" = by size, the default): same latents, same
        timestep, same weights — only the text context of the cross-attention differs — so every projection and row kernel runs on 2 S rows
        (a 14B 720p projection fills 46.2 rounds of 256 CUs instead of 2 x 23.1 -> 2 x 24) and the two self-attentions are one launch
        (92.5 rounds instead of 2 x 47).  Each output row is computed from the same operands in the same order as in the separate
        forwards, so the result is bit-identical (tests/test_gpu_model.py).  Needs the plain single-GPU block driver in its fp32-statistics mode."""
        tr = self.transformer_infer
        want = _cfg(self.config, "cfg_pair", "auto")
        if want == "auto":
            # measured on MI355X: -0.8 % of a Wan-14B 720p step (11 840 attention workgroups per forward), +1 % of a Wan-1.3B 480p step (960):
            # pair only when one forward's self-attention already fills the chip many times over (lib.attention_batched's rule)
            want = cfg_form_by_size(self.scheduler.seq_len, tr.num_heads) == "pair"
        return (bool(want) and type(tr) is WanTransformerInfer and tr.parallel_attention is None and tr.round_mode == lib.ROUND_FP32
                and tr.attention_type == "hip_flash" and self.scheduler.latents.is_cuda)

    def _forward_pair(self, inputs):
        tr = self.transformer_infer
        embed, grid_sizes, (x, embed0, seq_lens, freqs, ctx_c) = self.pre_infer.infer(self.pre_weight, inputs, positive=True)
        S = x.shape[0]
        if int(seq_lens[0]) != S:
            return None  # token buffer padded beyond the grid: the separate forwards handle it
        ctx_u = self.pre_infer.full_context(self.pre_weight, inputs, False)
        Sp = (S + 63) // 64 * 64  # a forward's slot: whole 64-token blocks of V^T
        X = torch.empty((2 * Sp, x.shape[1]), dtype=x.dtype, device=x.device)
        for b in range(2):
            X[b * Sp : b * Sp + S].copy_(x)
            X[b * Sp + S : (b + 1) * Sp].zero_()  # padding rows: zero in, finite throughout (every kernel writes all rows of its output)
        tr._pair = (S, Sp)
        try:
            X = tr.infer(self.transformer_weights, grid_sizes, embed, X, embed0, seq_lens, freqs, (ctx_c, ctx_u))
        finally:
            tr._pair = None
        rows = self.post_infer.infer_rows(self.post_weight, X, embed)
        return tuple(self.post_infer.unpatchify(rows[b * Sp : b * Sp + S], grid_sizes)[0].float() for b in range(2))

    @torch.no_grad()
    def infer(self, inputs):
        """cond forward, uncond forward, fp32 CFG combine (model.py:197-226)."""
        pair = self._forward_pair(inputs) if (self.config["enable_cfg"] and self._pair_ok(inputs)) else None
        if pair is None and self.config["enable_cfg"]:
            # the two CFG branches block by block on two compute streams (config `cfg_branch_streams`: True / False / "auto", the default): under
            # Ulysses ulysses.CfgBranchStreams (set by parallelize_wan; "auto" = off there), on one GPU CfgBranchStreams above, "auto" = by size —
            # the measured table is in cfg_form_by_size's docstring (one place)
            il = getattr(self, "_cfg_interleave", None)
            want = _cfg(self.config, "cfg_branch_streams", "auto")
            if want == "auto":
                # under Ulysses (il installed by parallelize_wan): OFF until a multi-GPU run has shown the two-stream form safe and faster there
                # (bench.py at N > 1 times both forms and sets the key explicitly); on one GPU: by size
                want = il is None and cfg_form_by_size(self.scheduler.seq_len, self.transformer_infer.num_heads) == "streams"
            want = bool(want) and self.scheduler.latents.is_cuda
            if want and il is None:
                il = self._cfg_interleave = CfgBranchStreams(self)
            if want and il.usable(inputs):
                pair = il.forward_pair(inputs)
        if pair is not None:
            cond, uncond = pair
        else:
            cond = self._forward(inputs, True)
            if not self.config["enable_cfg"]:
                self.scheduler.noise_pred = cond
                return
            uncond = self._forward(inputs, False)
        if hasattr(self.scheduler, "set_cfg_parts"):
            # our schedulers fold `uncond + guide * (cond - uncond)` (:218) into the fused step_post launch; reading
            # scheduler.noise_pred still yields the combined tensor
            self.scheduler.set_cfg_parts(cond, uncond, self.config["sample_guide_scale"])
        else:
            self.scheduler.noise_pred = uncond + self.config["sample_guide_scale"] * (cond - uncond)


def default_config(dims, **overrides):
    """Config dict with the keys the hot path reads (same names as the reference's JSON configs)."""
    cfg = dict(
        task="t2v", model_cls="wan2.1", dim=dims["dim"], ffn_dim=dims["ffn_dim"], num_heads=dims["num_heads"], num_layers=dims["num_layers"],
        freq_dim=256, text_len=dims.get("text_len", 512), in_dim=16, out_dim=16, eps=1e-6, patch_size=(1, 2, 2), vae_stride=(4, 8, 8),
        cpu_offload=False, mm_config={"mm_type": "Hip-bf16"}, self_attn_1_type="hip_flash", cross_attn_1_type="hip_flash", attention_type="hip_flash",
        feature_caching="NoCaching", parallel_attn_type=None, enable_cfg=True, sample_guide_scale=6.0, sample_shift=8.0, infer_steps=50, seed=42,
        target_video_length=81, target_shape=(16, 21, 90, 160),
    )
    cfg.update(overrides)
    return cfg
