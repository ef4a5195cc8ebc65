"""HunyuanVideo DiT forward on the HIP kernels — host-side mirror of the reference's Hunyuan network.

reference (paths relative to /root/reference/lightx2v/):
  models/networks/hunyuan/weights/{pre,post,transformer}_weights.py — weight trees / checkpoint tensor names
  models/networks/hunyuan/infer/pre_infer.py:6-154, transformer_infer.py:9-384, post_infer.py:4-33, utils_bf16.py:5-31
  models/networks/hunyuan/model.py:151-158 (HunyuanModel.infer)
  models/schedulers/hunyuan/scheduler.py:175-179,237-260,278-319 (sigmas, Euler step, RoPE tables)

Same class / method / tensor names and call contracts as the reference, so the objects slot into its runner; the
kernels are the ones built for Wan (GEMM with fused epilogues, LayerNorm+adaLN, attention) plus
x2v_headnorm_rope_bf16 (per-head RMSNorm + real RoPE on the q/k column blocks of the fused QKV output).

Layout decisions (MI355X, everything resident):
  * image and text tokens live in ONE [L_img + L_txt, *] buffer from the first block on: the img and txt GEMMs write
    their row ranges of the joint qkv / attention buffers, so the reference's per-block torch.cat of q, k, v and the
    attention output split (transformer_infer.py:119-121,146) do not exist;
  * q, k, v are column blocks of the fused QKV GEMM output [L, 3*H*128] (token stride 3*H*128) — no rearrange copy;
  * single blocks: linear1 is issued as two GEMMs over row blocks of its weight (qkv rows: no activation; mlp rows:
    GELU-tanh epilogue writing straight into the linear2 input buffer next to where attention writes) — the
    torch.split / gelu / torch.cat of transformer_infer.py:334,372-373 cost nothing;
  * every `x + out * gate` is the residual epilogue of the producing GEMM.
flash-attn's varlen call with cu_seqlens = [0, L_img + n_valid_txt, L_img + L_txt] (pre_infer.py:50-56) becomes two
dense launches of the attention kernel (the second — padded text rows among themselves — is tiny).
"""
import math

import torch

from . import lib
from .registry import ATTN_WEIGHT_REGISTER, CONV3D_WEIGHT_REGISTER, LN_WEIGHT_REGISTER, MM_WEIGHT_REGISTER, RMS_WEIGHT_REGISTER
from .weight_module import WeightModule, WeightModuleList
from . import ops  # noqa: F401  (registers the HIP operator classes)

BF16 = torch.bfloat16


# ------------------------------------------------------------------------------------------------ weights
def _mm(config, w, b):
    mm_type = (config.get("mm_config") or {}).get("mm_type", "Hip-bf16")
    return MM_WEIGHT_REGISTER[mm_type](w, b)


class HunyuanPreWeights(WeightModule):
    """reference: hunyuan/weights/pre_weights.py:5-84."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        mm = lambda n: MM_WEIGHT_REGISTER["Hip-bf16"](f"{n}.weight", f"{n}.bias")  # noqa: E731
        self.add_module("img_in_proj", CONV3D_WEIGHT_REGISTER["hip_patch"]("img_in.proj.weight", "img_in.proj.bias", stride=(1, 2, 2)))
        for attr, name in (
            ("txt_in_input_embedder", "txt_in.input_embedder"), ("txt_in_t_embedder_mlp_0", "txt_in.t_embedder.mlp.0"), ("txt_in_t_embedder_mlp_2", "txt_in.t_embedder.mlp.2"),
            ("txt_in_c_embedder_linear_1", "txt_in.c_embedder.linear_1"), ("txt_in_c_embedder_linear_2", "txt_in.c_embedder.linear_2"),
            ("time_in_mlp_0", "time_in.mlp.0"), ("time_in_mlp_2", "time_in.mlp.2"), ("vector_in_in_layer", "vector_in.in_layer"), ("vector_in_out_layer", "vector_in.out_layer"),
            ("guidance_in_mlp_0", "guidance_in.mlp.0"), ("guidance_in_mlp_2", "guidance_in.mlp.2"),
        ):
            self.add_module(attr, mm(name))
        for j in range(2):
            p, a = f"txt_in.individual_token_refiner.blocks.{j}", f"txt_in_individual_token_refiner_blocks_{j}"
            self.add_module(f"{a}_norm1", LN_WEIGHT_REGISTER["hip"](f"{p}.norm1.weight", f"{p}.norm1.bias", eps=1e-6))
            self.add_module(f"{a}_self_attn_qkv", mm(f"{p}.self_attn_qkv"))
            self.add_module(f"{a}_self_attn_proj", mm(f"{p}.self_attn_proj"))
            self.add_module(f"{a}_norm2", LN_WEIGHT_REGISTER["hip"](f"{p}.norm2.weight", f"{p}.norm2.bias", eps=1e-6))
            self.add_module(f"{a}_mlp_fc1", mm(f"{p}.mlp.fc1"))
            self.add_module(f"{a}_mlp_fc2", mm(f"{p}.mlp.fc2"))
            self.add_module(f"{a}_adaLN_modulation_1", mm(f"{p}.adaLN_modulation.1"))
        self.add_module("txt_in_attn_1", ATTN_WEIGHT_REGISTER["hip_flash"]())


class HunyuanPostWeights(WeightModule):
    """reference: hunyuan/weights/post_weights.py:5-11 (final_layer.linear is the fp32 `Default-Force-FP32` op)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.add_module("final_layer_linear", MMWeightForceFP32Hip("final_layer.linear.weight", "final_layer.linear.bias"))
        self.add_module("final_layer_adaLN_modulation_1", MM_WEIGHT_REGISTER["Hip-bf16"]("final_layer.adaLN_modulation.1.weight", "final_layer.adaLN_modulation.1.bias"))


@MM_WEIGHT_REGISTER("Hip-Force-FP32")
class MMWeightForceFP32Hip(ops._Movable):
    """reference: mm_weight.py:99-108 — weight/bias promoted to fp32 at load, fp32 input.  fp32 GEMM on the fp32-input
    MFMA through the implicit-GEMM convolution kernel with a 1x1 tap (csrc/vae.hip)."""

    _tensor_attrs = ("weight", "bias")

    def __init__(self, weight_name, bias_name, lazy_load=False, lazy_load_file=None):
        self.weight_name, self.bias_name = weight_name, bias_name
        self.config, self.weight, self.bias = {}, None, None

    def load(self, weight_dict):
        self.weight = weight_dict[self.weight_name].to(torch.float32).contiguous()
        self.bias = weight_dict[self.bias_name].to(torch.float32).contiguous() if self.bias_name is not None else None

    def apply(self, input_tensor):
        x = input_tensor.to(torch.float32).contiguous()
        m, k = x.shape
        out = torch.empty((m, self.weight.shape[0]), dtype=torch.float32, device=x.device)
        lib.vae_conv(x, (m * k, m * k, k), self.weight, out, 1, 1, m, bias=self.bias)
        return out

    def state_dict(self, destination=None):
        destination = {} if destination is None else destination
        destination[self.weight_name] = self.weight.cpu().detach().clone()
        if self.bias is not None:
            destination[self.bias_name] = self.bias.cpu().detach().clone()
        return destination


class HunyuanTransformerDoubleBlock(WeightModule):
    """reference: hunyuan/weights/transformer_weights.py:18-49."""

    def __init__(self, block_index, config):
        super().__init__()
        self.block_index, self.config = block_index, config
        p = f"double_blocks.{block_index}"
        for s in ("img", "txt"):
            self.add_module(f"{s}_mod", _mm(config, f"{p}.{s}_mod.linear.weight", f"{p}.{s}_mod.linear.bias"))
            self.add_module(f"{s}_attn_qkv", _mm(config, f"{p}.{s}_attn_qkv.weight", f"{p}.{s}_attn_qkv.bias"))
            self.add_module(f"{s}_attn_q_norm", RMS_WEIGHT_REGISTER["hip"](f"{p}.{s}_attn_q_norm.weight", eps=1e-6))
            self.add_module(f"{s}_attn_k_norm", RMS_WEIGHT_REGISTER["hip"](f"{p}.{s}_attn_k_norm.weight", eps=1e-6))
            self.add_module(f"{s}_attn_proj", _mm(config, f"{p}.{s}_attn_proj.weight", f"{p}.{s}_attn_proj.bias"))
            self.add_module(f"{s}_mlp_fc1", _mm(config, f"{p}.{s}_mlp.fc1.weight", f"{p}.{s}_mlp.fc1.bias"))
            self.add_module(f"{s}_mlp_fc2", _mm(config, f"{p}.{s}_mlp.fc2.weight", f"{p}.{s}_mlp.fc2.bias"))
        self.add_module("double_attn", ATTN_WEIGHT_REGISTER["hip_flash"]())


class HunyuanTransformerSingleBlock(WeightModule):
    """reference: hunyuan/weights/transformer_weights.py:52-71."""

    def __init__(self, block_index, config):
        super().__init__()
        self.block_index, self.config = block_index, config
        p = f"single_blocks.{block_index}"
        self.add_module("linear1", _mm(config, f"{p}.linear1.weight", f"{p}.linear1.bias"))
        self.add_module("linear2", _mm(config, f"{p}.linear2.weight", f"{p}.linear2.bias"))
        self.add_module("q_norm", RMS_WEIGHT_REGISTER["hip"](f"{p}.q_norm.weight", eps=1e-6))
        self.add_module("k_norm", RMS_WEIGHT_REGISTER["hip"](f"{p}.k_norm.weight", eps=1e-6))
        self.add_module("modulation", _mm(config, f"{p}.modulation.linear.weight", f"{p}.modulation.linear.bias"))
        self.add_module("single_attn", ATTN_WEIGHT_REGISTER["hip_flash"]())


class HunyuanTransformerWeights(WeightModule):
    """reference: hunyuan/weights/transformer_weights.py:5-15 (20 double + 40 single blocks unless the config says otherwise)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.double_blocks_num = config.get("double_blocks_num", 20)
        self.single_blocks_num = config.get("single_blocks_num", 40)
        self.add_module("double_blocks", WeightModuleList([HunyuanTransformerDoubleBlock(i, config) for i in range(self.double_blocks_num)]))
        self.add_module("single_blocks", WeightModuleList([HunyuanTransformerSingleBlock(i, config) for i in range(self.single_blocks_num)]))


# ------------------------------------------------------------------------------------------------ infer
def _segments(cu_seqlens_qkv):
    cu = cu_seqlens_qkv.tolist() if torch.is_tensor(cu_seqlens_qkv) else list(cu_seqlens_qkv)
    return [(a, b) for a, b in zip(cu[:-1], cu[1:]) if b > a]


class HunyuanTransformerInfer:
    """reference: hunyuan/infer/transformer_infer.py:9-384 (no-offload path, t2v)."""

    def __init__(self, config):
        self.config = config
        self.double_blocks_num = config.get("double_blocks_num", 20)
        self.single_blocks_num = config.get("single_blocks_num", 40)
        self.heads_num = config.get("heads_num", 24)
        self.hidden_size = config.get("hidden_size", 3072)
        self.mlp_hidden_dim = config.get("mlp_hidden_dim", 12288)
        self.round_mode = lib.ROUND_REF if config.get("hip_ref_rounding", False) else lib.ROUND_FP32
        self.parallel_attention = None
        self._segs = None
        # fp32-statistics mode: q leaves the norm+RoPE kernel carrying softmax_scale*log2(e) (one rounding), see x2v.h
        self._qs = lib.ATTN_PRESCALE if self.round_mode == lib.ROUND_FP32 else 1.0

    def set_scheduler(self, scheduler):
        self.scheduler = scheduler

    def infer(self, weights, img, txt, vec, cu_seqlens_qkv, max_seqlen_qkv, freqs_cis, token_replace_vec=None, frist_frame_token_num=None):
        """transformer_infer.py:66-79.  img [L_img, D], txt [L_txt, D], vec [1, D] → (img, vec)."""
        if token_replace_vec is not None:
            raise NotImplementedError("i2v token-replace modulation (transformer_infer.py:95-100) is not built")
        n_img, n_txt = img.shape[0], txt.shape[0]
        self._segs = _segments(cu_seqlens_qkv)  # one host read per forward (the reference's flash call reads them per block)
        cu = cu_seqlens_qkv.tolist() if torch.is_tensor(cu_seqlens_qkv) else list(cu_seqlens_qkv)
        self._sp_lens = (n_img, cu[1] - n_img, n_txt)  # image tokens on this rank, valid text tokens, text tokens
        x = torch.empty((n_img + n_txt, self.hidden_size), dtype=BF16, device=img.device)
        x[:n_img].copy_(img)
        x[n_img:].copy_(txt)
        vec_silu = lib.activation(vec, 3)  # silu(vec) is the same for every block (:82,313)
        ws = self._workspace(x)
        for i in range(self.double_blocks_num):
            self.infer_double_block(weights.double_blocks[i], x, n_img, vec_silu, freqs_cis, ws)
        for i in range(self.single_blocks_num):
            self.infer_single_block(weights.single_blocks[i], x, n_txt, vec_silu, freqs_cis, ws)
        return x[:n_img], vec

    def _workspace(self, x):
        L, D, F = x.shape[0], self.hidden_size, self.mlp_hidden_dim
        dev = x.device
        e = lambda *s: torch.empty(s, dtype=BF16, device=dev)  # noqa: E731
        return dict(mod=e(L, D), qkv=e(L, 3 * D), cat=e(L, D + F), hid=e(L, F))

    def _blocked(self, *ops):
        """The copy-free Ulysses path (ulysses.UlyssesHunyuanAttention.attend_blocked): the exchange buffers are GEMM / norm operands.
        Needs operators that take block-strided operands (the bf16 class) and head blocks that are whole K tiles."""
        pa = self.parallel_attention
        return (pa is not None and hasattr(pa, "attend_blocked") and self._qs != 1.0 and pa.blocked_ok(self.hidden_size, self.mlp_hidden_dim)
                and all(getattr(o, "accepts_blocked", False) and not hasattr(o, "quantize_input") for o in ops))  # bf16 class only: the fused-QKV / cat layouts are not wired for w8a8

    def _attend_blocked(self, bufs, qkv_txt):
        D = self.hidden_size
        n_img, n_valid, n_txt = self._sp_lens
        txt = (qkv_txt[:, :D], qkv_txt[:, D : 2 * D], qkv_txt[:, 2 * D :])
        return self.parallel_attention.attend_blocked(bufs, txt, (n_valid, n_txt), self.heads_num, variant=lib.ATTN_FAST | lib.ATTN_Q_PRESCALED)

    def _attention(self, q, k, v, out):
        variant = (lib.ATTN_FAST | lib.ATTN_Q_PRESCALED) if self._qs != 1.0 else 0
        if self.parallel_attention is not None:  # Ulysses (reference hook: transformer_infer.py:130-144,358-369)
            n_img, n_valid, n_txt = self._sp_lens
            return self.parallel_attention(q, k, v, n_img, (n_valid, n_txt), self.heads_num, out, variant=variant)
        # single GPU: the key walk starts at tile 0 (round 4: the staggered walk, x2v.h X2V_ATTN_VT_STAGGER, measured slower at sustained load — Wan-14B
        # -0.9 % A-B-A-B, this model 8821 vs 8889 ms per step; wan.SELF_ATTN_STAGGER is the one switch for both drivers)
        from .wan import SELF_ATTN_STAGGER

        for a, b in self._segs:
            lib.attention(q[a:b], k[a:b], v[a:b], self.heads_num, 128, out=out[a:b], variant=variant | (lib.ATTN_STAGGER if (variant and SELF_ATTN_STAGGER) else 0))

    def infer_double_block(self, weights, x, n_img, vec_silu, freqs_cis, ws):
        """transformer_infer.py:81-310 on the joint buffer x = [img ; txt]."""
        D, H = self.hidden_size, self.heads_num
        img, txt = x[:n_img], x[n_img:]
        i_sh1, i_sc1, i_g1, i_sh2, i_sc2, i_g2 = weights.img_mod.apply(vec_silu).chunk(6, dim=-1)
        t_sh1, t_sc1, t_g1, t_sh2, t_sc2, t_g2 = weights.txt_mod.apply(vec_silu).chunk(6, dim=-1)
        mod, qkv = ws["mod"], ws["qkv"]
        # LN + modulate → fused QKV, image and text rows of the same buffers
        lib.layernorm(img, scale=i_sc1, shift=i_sh1, eps=1e-6, out=mod[:n_img])
        lib.layernorm(txt, scale=t_sc1, shift=t_sh1, eps=1e-6, out=mod[n_img:])
        cos, sin = freqs_cis
        if self._blocked(weights.img_attn_qkv, weights.img_attn_proj, weights.txt_attn_proj):
            # Ulysses, copy-free: the image QKV GEMM writes the head-blocked send buffers, norm + RoPE run in place on them, the output
            # projections read the received [N, rows, hd/N] buffers as K-blocked x (attentions/distributed/ulysses/attn.py:36-46,75-91 without
            # its transposing copies)
            bufs = self.parallel_attention.buffers(n_img, x.shape[0] - n_img, D, self.mlp_hidden_dim, BF16, x.device)
            snd = bufs["snd"]
            weights.img_attn_qkv.apply(mod[:n_img], out=snd.view(-1, n_img, snd.shape[3]))
            weights.txt_attn_qkv.apply(mod[n_img:], out=qkv[n_img:])
            lib.headnorm_rope_blocked_(snd[0], snd[1], weights.img_attn_q_norm.weight, weights.img_attn_k_norm.weight, cos, sin, H, n_img, 1e-6, self.round_mode, self._qs)
            lib.headnorm_rope_(qkv[n_img:, :D], qkv[n_img:, D : 2 * D], weights.txt_attn_q_norm.weight, weights.txt_attn_k_norm.weight, None, None, H, 0, 1e-6, self.round_mode,
                               self._qs)
            a_img, a_txt = self._attend_blocked(bufs, qkv[n_img:])
            nb = snd.shape[1]
            weights.img_attn_proj.apply(a_img[:nb], epilogue=lib.EPI_RESIDUAL, resid=img, gate=i_g1)
            weights.txt_attn_proj.apply(a_txt[:nb], epilogue=lib.EPI_RESIDUAL, resid=txt, gate=t_g1)
            return self._double_block_mlp(weights, img, txt, n_img, mod, ws, i_sh2, i_sc2, i_g2, t_sh2, t_sc2, t_g2, x)
        weights.img_attn_qkv.apply(mod[:n_img], out=qkv[:n_img])
        weights.txt_attn_qkv.apply(mod[n_img:], out=qkv[n_img:])
        q, k, v = qkv[:, :D], qkv[:, D : 2 * D], qkv[:, 2 * D :]
        lib.headnorm_rope_(q[:n_img], k[:n_img], weights.img_attn_q_norm.weight, weights.img_attn_k_norm.weight, cos, sin, H, n_img, 1e-6, self.round_mode, self._qs)
        lib.headnorm_rope_(q[n_img:], k[n_img:], weights.txt_attn_q_norm.weight, weights.txt_attn_k_norm.weight, None, None, H, 0, 1e-6, self.round_mode, self._qs)
        attn = ws["cat"][:, :D]
        self._attention(q, k, v, attn)
        # x += proj(attn) * gate1 ; x += fc2(gelu(fc1(LN(x)*(1+scale2)+shift2))) * gate2   — per stream
        weights.img_attn_proj.apply(attn[:n_img], epilogue=lib.EPI_RESIDUAL, resid=img, gate=i_g1)
        weights.txt_attn_proj.apply(attn[n_img:], epilogue=lib.EPI_RESIDUAL, resid=txt, gate=t_g1)
        return self._double_block_mlp(weights, img, txt, n_img, mod, ws, i_sh2, i_sc2, i_g2, t_sh2, t_sc2, t_g2, x)

    def _double_block_mlp(self, weights, img, txt, n_img, mod, ws, i_sh2, i_sc2, i_g2, t_sh2, t_sc2, t_g2, x):
        lib.layernorm(img, scale=i_sc2, shift=i_sh2, eps=1e-6, out=mod[:n_img])
        lib.layernorm(txt, scale=t_sc2, shift=t_sh2, eps=1e-6, out=mod[n_img:])
        hid = ws["hid"]
        weights.img_mlp_fc1.apply(mod[:n_img], epilogue=lib.EPI_GELU_TANH, out=hid[:n_img])
        weights.txt_mlp_fc1.apply(mod[n_img:], epilogue=lib.EPI_GELU_TANH, out=hid[n_img:])
        weights.img_mlp_fc2.apply(hid[:n_img], epilogue=lib.EPI_RESIDUAL, resid=img, gate=i_g2)
        weights.txt_mlp_fc2.apply(hid[n_img:], epilogue=lib.EPI_RESIDUAL, resid=txt, gate=t_g2)
        return x

    def infer_single_block(self, weights, x, txt_seq_len, vec_silu, freqs_cis, ws):
        """transformer_infer.py:312-384."""
        D, H = self.hidden_size, self.heads_num
        n_img = x.shape[0] - txt_seq_len
        shift, scale, gate = weights.modulation.apply(vec_silu).chunk(3, dim=-1)
        mod, qkv, cat = ws["mod"], ws["qkv"], ws["cat"]
        lib.layernorm(x, scale=scale, shift=shift, eps=1e-6, out=mod)
        # linear1 = [qkv | mlp] rows of one checkpoint tensor, two epilogues: through the operator (bf16 / fp8 / mxfp8 alike), the
        # activation quantised once when the operator is a quantised one
        l1 = weights.linear1
        cos, sin = freqs_cis
        if self._blocked(l1, weights.linear2):
            # Ulysses, copy-free: linear2's input cat(attn, mlp) (transformer_infer.py:377) exists only as the K-blocked buffers a_img / a_txt —
            # blocks [0, N) are the receive buffers of the head->seq exchange / the text gather, the rest is written N-blocked by the GELU GEMM
            bufs = self.parallel_attention.buffers(n_img, txt_seq_len, D, self.mlp_hidden_dim, BF16, x.device)
            snd, a_img, a_txt = bufs["snd"], bufs["a_img"], bufs["a_txt"]
            nb = snd.shape[1]
            qkv_rows, mlp_rows = slice(0, 3 * D), slice(3 * D, None)
            l1.apply(mod[:n_img], out=snd.view(-1, n_img, snd.shape[3]), row_slice=qkv_rows)
            l1.apply(mod[n_img:], out=qkv[n_img:], row_slice=qkv_rows)
            l1.apply(mod[:n_img], epilogue=lib.EPI_GELU_TANH, out=a_img[nb:], row_slice=mlp_rows)
            l1.apply(mod[n_img:], epilogue=lib.EPI_GELU_TANH, out=a_txt[nb:], row_slice=mlp_rows)
            lib.headnorm_rope_blocked_(snd[0], snd[1], weights.q_norm.weight, weights.k_norm.weight, cos, sin, H, n_img, 1e-6, self.round_mode, self._qs)
            lib.headnorm_rope_(qkv[n_img:, :D], qkv[n_img:, D : 2 * D], weights.q_norm.weight, weights.k_norm.weight, None, None, H, 0, 1e-6, self.round_mode, self._qs)
            self._attend_blocked(bufs, qkv[n_img:])
            weights.linear2.apply(a_img, epilogue=lib.EPI_RESIDUAL, resid=x[:n_img], gate=gate)
            weights.linear2.apply(a_txt, epilogue=lib.EPI_RESIDUAL, resid=x[n_img:], gate=gate)
            return x
        pre = {"quantized": l1.quantize_input(mod)} if hasattr(l1, "quantize_input") else {}
        l1.apply(mod, out=qkv, row_slice=slice(0, 3 * D), **pre)                                          # qkv rows of linear1
        l1.apply(mod, epilogue=lib.EPI_GELU_TANH, out=cat[:, D:], row_slice=slice(3 * D, None), **pre)    # mlp rows, GELU, into linear2's input
        q, k, v = qkv[:, :D], qkv[:, D : 2 * D], qkv[:, 2 * D :]
        lib.headnorm_rope_(q, k, weights.q_norm.weight, weights.k_norm.weight, cos, sin, H, n_img, 1e-6, self.round_mode, self._qs)
        self._attention(q, k, v, cat[:, :D])
        weights.linear2.apply(cat, epilogue=lib.EPI_RESIDUAL, resid=x, gate=gate)
        return x



class HunyuanTransformerInferTeaCaching(HunyuanTransformerInfer):
    """reference: hunyuan/infer/feature_caching/transformer_infer.py:7-135 — TeaCache around the fused block stack.  After every forward the
    first double block's modulated input of the returned image tokens (LayerNorm + modulate with `img_mod(vec)` — applied to vec WITHOUT the
    SiLU the blocks put in front, as the reference does, :21) is compared with the previous step's: the polynomial-rescaled relative L1
    change accumulates in `accumulated_rel_l1_distance`; while it stays below `teacache_thresh` the NEXT step skips the block stack and
    re-applies the cached residual img_out - img_in (`scheduler.caching_records[i + 1]`).  First and last step always compute.  The decision
    is host-side control flow exactly as in the reference (one .item() per step); LayerNorm+modulate is the HIP kernel, the two [L, D]
    elementwise passes (residual capture / re-apply) run on the gate-residual kernel."""

    COEFFICIENTS = [7.33226126e02, -4.01131952e02, 6.75869174e01, -3.14987800e00, 9.61237896e-02]

    def __init__(self, config):
        super().__init__(config)
        self.teacache_thresh = config["teacache_thresh"]
        self.accumulated_rel_l1_distance = 0
        self.previous_modulated_input = None
        self.previous_residual = None
        self.coefficients = list(self.COEFFICIENTS)
        self._minus_one = None

    def calculate_should_calc(self, img, vec, weights):
        import numpy as np

        mod = weights.double_blocks[0].img_mod.apply(vec)  # [1, 6 D]: shift, scale, ... of block 0 (no SiLU: reference :21)
        D = self.hidden_size
        modulated = lib.layernorm(img, scale=mod[:, D : 2 * D], shift=mod[:, :D], eps=1e-6)
        index = self.scheduler.step_index
        if index == 0 or index == self.scheduler.infer_steps - 1:
            should_calc = True
            self.accumulated_rel_l1_distance = 0
        else:
            prev = self.previous_modulated_input
            rel = self._sharded_rel_l1(modulated, prev)
            if rel is None:
                rel = ((modulated - prev).abs().mean() / prev.abs().mean()).cpu().item()
            self.accumulated_rel_l1_distance += np.poly1d(self.coefficients)(rel)
            should_calc = not (self.accumulated_rel_l1_distance < self.teacache_thresh)
            if should_calc:
                self.accumulated_rel_l1_distance = 0
        self.previous_modulated_input = modulated
        return should_calc

    def _sharded_rel_l1(self, modulated, prev):
        """Under Ulysses every rank holds its own image shard, and a decision taken from the local shard can differ between ranks right at the
        threshold — one rank would skip the block stack (no all-to-all) while the others wait in it: a hang (the reference, whose ranks decide
        locally in feature_caching/transformer_infer.py:21-50, has the same hazard).  Here the numerator and denominator of the relative L1 change
        are summed over the sequence-parallel group first, so all ranks see one number: the ratio of the GLOBAL sums, which is the ratio of the global
        means whatever the shard sizes are (both sums run over the same element count).  It is the fp32 form of the single-GPU bf16 mean ratio: a
        decision within bf16 rounding of the threshold may differ from the single-GPU run's (tests/test_dist_cpu.py pins decisions on a fixture).
        None when not sharded (the single-GPU arithmetic, pinned to the reference fixture, stays as it is)."""
        import torch.distributed as dist

        pa = self.parallel_attention
        if pa is None or not dist.is_available() or not dist.is_initialized() or dist.get_world_size(getattr(pa, "group", None)) < 2:
            return None
        group = getattr(pa, "group", None)
        pair = torch.stack([(modulated - prev).abs().float().sum(), prev.abs().float().sum()]).double()
        if dist.get_backend(group) != "nccl":
            pair = pair.cpu()

        def reduce():
            dist.all_reduce(pair, op=dist.ReduceOp.SUM, group=group)
            return pair

        pair = pa.on_comm(reduce, pair) if hasattr(pa, "on_comm") else reduce()  # like every collective of the driver: on the one communication stream
        return (pair[0] / pair[1]).item()

    def infer(self, weights, img, txt, vec, cu_seqlens_qkv, max_seqlen_qkv, freqs_cis, token_replace_vec=None, frist_frame_token_num=None):
        index = self.scheduler.step_index
        records = self.scheduler.caching_records
        if records[index]:
            ori = img.clone()
            img, vec = super().infer(weights, img, txt, vec, cu_seqlens_qkv, max_seqlen_qkv, freqs_cis, token_replace_vec, frist_frame_token_num)
            if self._minus_one is None or self._minus_one.device != img.device:
                self._minus_one = torch.full((img.shape[1],), -1.0, dtype=img.dtype, device=img.device)
            res = img.clone()
            lib.gate_residual_(res, ori, self._minus_one)  # res = img_out - img_in (bf16, as the reference's tensor subtraction :116)
            self.previous_residual = res
        else:
            img = img.contiguous()
            lib.gate_residual_(img, self.previous_residual)  # img += previous_residual (:121)
        if index <= self.scheduler.infer_steps - 2:
            records[index + 1] = self.calculate_should_calc(img, vec, weights)
        return img, vec

    def clear(self):
        self.previous_modulated_input = None
        self.previous_residual = None


def _t_embed(t, device):
    """pre_infer.py:62-64: cos|sin of t * exp(-ln(1e4) j/128), fp32 → bf16, [1, 256] (256 values: host-side glue)."""
    freqs = torch.exp(-math.log(10000) * torch.arange(start=0, end=128, dtype=torch.float32, device=device) / 128)
    args = t.reshape(1, 1).float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1).to(BF16)


class HunyuanPreInfer:
    """reference: hunyuan/infer/pre_infer.py:6-154 (t2v)."""

    def __init__(self, config):
        self.config = config
        self.heads_num = config.get("heads_num", 24)

    def set_scheduler(self, scheduler):
        self.scheduler = scheduler

    @staticmethod
    def _mlp(a, b, x):
        return b.apply(a.apply(x, epilogue=lib.EPI_SILU))

    def infer(self, weights, inputs):
        sch = self.scheduler
        x, t = sch.latents, sch.timesteps[sch.step_index]
        te = inputs["text_encoder_output"]
        text_states, text_mask, text_states_2 = te["text_encoder_1_text_states"], te["text_encoder_1_attention_mask"], te["text_encoder_2_text_states"]
        dev = x.device
        time_out = self._mlp(weights.time_in_mlp_0, weights.time_in_mlp_2, _t_embed(t, dev))
        img_out = weights.img_in_proj.apply(x.to(BF16)).flatten(2).transpose(1, 2).squeeze(0)  # pre_infer.py:72-75 → [S, D] (views of the GEMM output)
        txt_out = self.infer_text_in(weights, text_states, text_mask, t)
        vec = time_out + self._mlp(weights.vector_in_in_layer, weights.vector_in_out_layer, text_states_2)
        vec = vec + self._mlp(weights.guidance_in_mlp_0, weights.guidance_in_mlp_2, _t_embed(sch.guidance, dev))
        n_img, n_txt = img_out.shape[0], txt_out.shape[0]
        n_valid = int(text_mask.sum())
        cu_seqlens_qkv = torch.tensor([0, n_valid + n_img, n_txt + n_img], dtype=torch.int32)
        return img_out, txt_out, vec, cu_seqlens_qkv, n_img + n_txt, (sch.freqs_cos, sch.freqs_sin)

    def infer_text_in(self, weights, text_states, text_mask, t):
        """pre_infer.py:72-145: conditioning vector c, input embedding, two token-refiner blocks.  The boolean
        attention mask (valid rows attend valid keys; every row may attend key 0) is realised as one dense attention
        over the n valid tokens plus a broadcast of v[0] to the padded rows (softmax over a single allowed key)."""
        dev = text_states.device
        H = self.heads_num
        c = self._mlp(weights.txt_in_t_embedder_mlp_0, weights.txt_in_t_embedder_mlp_2, _t_embed(t, dev))
        mask_float = text_mask.float().unsqueeze(-1).to(BF16)
        ctx = (text_states * mask_float).sum(dim=1) / mask_float.sum(dim=1)  # masked mean over <= 256 tokens: glue
        c = c + self._mlp(weights.txt_in_c_embedder_linear_1, weights.txt_in_c_embedder_linear_2, ctx)
        x = weights.txt_in_input_embedder.apply(text_states[0])
        n = int(text_mask.sum())
        if not bool(text_mask[0, :n].all()):
            raise NotImplementedError("token refiner: the valid text tokens must be left-aligned (tokenizer padding side 'right')")
        c_silu = lib.activation(c, 3)
        D = x.shape[1]
        for j in range(2):
            g = lambda s: getattr(weights, f"txt_in_individual_token_refiner_blocks_{j}_{s}")  # noqa: E731
            gate_msa, gate_mlp = g("adaLN_modulation_1").apply(c_silu).chunk(2, dim=1)
            qkv = g("self_attn_qkv").apply(g("norm1").apply(x))
            attn = torch.empty((x.shape[0], D), dtype=BF16, device=dev)
            lib.attention(qkv[:n, :D], qkv[:n, D : 2 * D], qkv[:n, 2 * D :], H, 128, out=attn[:n])
            if n < x.shape[0]:
                attn[n:] = qkv[0:1, 2 * D :]  # padded rows: softmax over the single allowed key 0 → v[0]
            x = g("self_attn_proj").apply(attn, epilogue=lib.EPI_RESIDUAL, resid=x.clone(), gate=gate_msa)
            h = g("mlp_fc1").apply(g("norm2").apply(x), epilogue=lib.EPI_SILU)
            x = g("mlp_fc2").apply(h, epilogue=lib.EPI_RESIDUAL, resid=x, gate=gate_mlp)
        return x


class HunyuanPostInfer:
    """reference: hunyuan/infer/post_infer.py:4-33."""

    def __init__(self, config):
        self.config = config

    def set_scheduler(self, scheduler):
        self.scheduler = scheduler

    def infer(self, weights, img, vec):
        shift, scale = weights.final_layer_adaLN_modulation_1.apply(lib.activation(vec, 3)).chunk(2, dim=1)
        out = lib.layernorm(img, scale=scale, shift=shift, eps=1e-6)
        out = weights.final_layer_linear.apply(out)  # fp32 [S, 64]
        _, _, ot, oh, ow = self.scheduler.latents.shape
        tt, th, tw = ot, oh // 2, ow // 2
        out = out.reshape(1, tt, th, tw, 16, 1, 2, 2)
        out = torch.einsum("nthwcopq->nctohpwq", out)  # unpatchify: a permutation
        return out.reshape(1, 16, tt, th * 2, tw * 2)


class HunyuanScheduler:
    """reference: schedulers/hunyuan/scheduler.py:237-319 (t2v): shift-7 sigmas, embedded guidance 6.0, bf16 RoPE
    tables for (T, H/2, W/2) with dims [16, 56, 56] and theta 256, Euler step in fp32."""

    def __init__(self, config, device="cuda"):
        self.config, self.device = config, torch.device(device)
        self.infer_steps = config["infer_steps"]
        self.shift, self.embedded_guidance_scale = 7.0, 6.0
        sig = torch.linspace(1, 0, self.infer_steps + 1)
        self.sigmas = (self.shift * sig) / (1 + (self.shift - 1) * sig)
        self.timesteps = (self.sigmas[:-1] * 1000).to(dtype=torch.float32, device=self.device)
        self.step_index, self.latents, self.noise_pred = 0, None, None
        self.caching_records = [True] * self.infer_steps  # schedulers/scheduler.py:11 (read and written by the TeaCache driver)

    def prepare(self, latents):
        """latents: [1,16,T,H,W] (the reference draws them from a device generator, scheduler.py:262-264; parity runs feed a file)."""
        self.latents = latents.to(self.device)
        self.guidance = torch.tensor([self.embedded_guidance_scale], dtype=BF16, device=self.device) * 1000.0
        _, _, t, h, w = self.latents.shape
        self.freqs_cos, self.freqs_sin = (f.to(self.device) for f in rope_tables([t, h // 2, w // 2]))

    def step_pre(self, step_index):
        self.step_index = step_index

    def step_post(self):
        dt = (self.sigmas[self.step_index + 1] - self.sigmas[self.step_index]).item()
        self.latents = self.latents.to(torch.float32) + self.noise_pred.to(torch.float32) * dt


def rope_tables(rope_sizes, rope_dim_list=(16, 56, 56), theta=256.0):
    """get_nd_rotary_pos_embed(use_real=True) as called at scheduler.py:299-306 → bf16 cos, sin [T*H*W, 128] (built once
    per run on the host: load-time table, not per-step math)."""
    axes = [torch.linspace(0, n, n + 1, dtype=torch.float32)[:n] for n in rope_sizes]
    grid = torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=0)
    cos, sin = [], []
    for a, d in enumerate(rope_dim_list):
        freqs = 1.0 / (theta ** (torch.arange(0, d, 2)[: d // 2].float() / d))
        f = torch.outer(grid[a].reshape(-1), freqs)
        cos.append(f.cos().repeat_interleave(2, dim=1))
        sin.append(f.sin().repeat_interleave(2, dim=1))
    return torch.cat(cos, dim=1).to(BF16).contiguous(), torch.cat(sin, dim=1).to(BF16).contiguous()


class HunyuanModel:
    """reference: hunyuan/model.py:23-158 (weights from a name→tensor dict already on the device)."""

    def __init__(self, config, weight_dict):
        self.config = config
        self.pre_weight, self.post_weight, self.transformer_weights = HunyuanPreWeights(config), HunyuanPostWeights(config), HunyuanTransformerWeights(config)
        for w in (self.pre_weight, self.post_weight, self.transformer_weights):
            w.load(weight_dict)
        fc = config.get("feature_caching", "NoCaching")  # reference: hunyuan/model.py:55-66
        if fc not in ("NoCaching", "Tea"):
            raise NotImplementedError(f"feature_caching={fc}: only 'NoCaching' and 'Tea' are built")
        tr_cls = HunyuanTransformerInferTeaCaching if fc == "Tea" else HunyuanTransformerInfer
        self.pre_infer, self.post_infer, self.transformer_infer = HunyuanPreInfer(config), HunyuanPostInfer(config), tr_cls(config)

    def set_scheduler(self, scheduler):
        self.scheduler = scheduler
        for m in (self.pre_infer, self.post_infer, self.transformer_infer):
            m.set_scheduler(scheduler)

    def infer(self, inputs):
        pre = self.pre_infer.infer(self.pre_weight, inputs)
        img, vec = self.transformer_infer.infer(self.transformer_weights, *pre)
        self.scheduler.noise_pred = self.post_infer.infer(self.post_weight, img, vec)


def default_config(dims, infer_steps=50, **overrides):
    cfg = dict(task="t2v", mm_config={}, do_mm_calib=False, cpu_offload=False, attention_type="hip_flash", feature_caching="NoCaching", infer_steps=infer_steps,
               heads_num=dims["heads"], hidden_size=dims["hidden"], mlp_hidden_dim=dims["mlp"], double_blocks_num=dims["double_blocks"],
               single_blocks_num=dims["single_blocks"], hip_ref_rounding=False)
    cfg.update(overrides)
    return cfg
