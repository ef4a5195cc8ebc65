"""HIP-backed operator objects behind the reference's operator API.

Every class honours the reference's duck-typed protocol — ctor signature, `set_config`, `load(weight_dict)`,
`apply(...)`, `to_cuda/to_cpu`, `state_dict`, `clear`, `_calculate_size` — and is registered under new keys,
so an unchanged LightX2V config selects them by string:

    mm_config.mm_type      = "Hip-bf16" | "W-fp8-channel-sym-A-fp8-channel-sym-dynamic-Hip"
    self_attn_1_type / cross_attn_1_type / attention_type = "hip_flash"
    RMS / LN registries    : key "hip"   (also aliased as "Default"/"sgl-kernel" inside this package, since the
                             reference's weight classes hard-code those keys: transformer_weights.py:127,159,167)

apply() never falls back to torch math: tensors must live on a gfx950 device.
"""
import torch

from . import lib
from .registry import ATTN_WEIGHT_REGISTER, CONV3D_WEIGHT_REGISTER, LN_WEIGHT_REGISTER, MM_WEIGHT_REGISTER, RMS_WEIGHT_REGISTER, TENSOR_REGISTER


def _to(t, device, non_blocking=False):
    return None if t is None else t.to(device, non_blocking=non_blocking)


class _Movable:
    _tensor_attrs = ()

    def set_config(self, config=None):
        if config is not None:
            self.config = config

    def to_cuda(self, non_blocking=False):
        for a in self._tensor_attrs:
            if getattr(self, a, None) is not None:
                setattr(self, a, _to(getattr(self, a), "cuda", non_blocking))

    def to_cpu(self, non_blocking=False):
        for a in self._tensor_attrs:
            if getattr(self, a, None) is not None:
                setattr(self, a, _to(getattr(self, a), "cpu", non_blocking))

    def clear(self):
        for a in self._tensor_attrs:
            setattr(self, a, None)

    def _calculate_size(self):
        return sum(getattr(self, a).numel() * getattr(self, a).element_size() for a in self._tensor_attrs if getattr(self, a, None) is not None)


# ------------------------------------------------------------------------------------------------ MM
@MM_WEIGHT_REGISTER("Hip-bf16")
class MMWeightHip(_Movable):
    """reference: common/ops/mm/mm_weight.py:70-96 (`Default`).  The checkpoint's [N,K] tensor stays as it
    is in HBM (the reference stores the `.t()` view, :76); the kernel consumes K-contiguous rows of both
    operands.  Extra (optional) keyword arguments expose the fused epilogues to the fused block driver."""

    _tensor_attrs = ("weight", "bias")
    accepts_blocked = True  # apply() takes a 3-D K-blocked input / N-blocked `out` (lib.gemm: the Ulysses exchange buffers in place)

    def __init__(self, weight_name, bias_name, lazy_load=False, lazy_load_file=None):
        self.weight_name, self.bias_name = weight_name, bias_name
        self.lazy_load, self.lazy_load_file = lazy_load, lazy_load_file
        self.config = {}
        self.weight = self.bias = None

    def load(self, weight_dict):
        self.weight = weight_dict[self.weight_name]
        if self.weight.dim() > 2:  # patch-embedding conv kernel [D,C,1,2,2] used as a [D, C*4] matrix
            self.weight = self.weight.reshape(self.weight.shape[0], -1)
        self.weight = self.weight.contiguous()
        self.bias = weight_dict[self.bias_name] if self.bias_name is not None else None

    def load_from_disk(self):
        self.weight = self.lazy_load_file.get_tensor(self.weight_name).to(torch.bfloat16)
        self.bias = self.lazy_load_file.get_tensor(self.bias_name).to(torch.bfloat16) if self.bias_name is not None else None

    def apply(self, input_tensor, epilogue=lib.EPI_NONE, resid=None, gate=None, out=None, row_slice=None):
        """`row_slice` (a slice over the N output channels) runs the layer on that block of weight rows only — weight, bias (and, in the
        quantised classes, the per-channel scales) sliced together; used where one checkpoint tensor feeds two consumers with different
        epilogues (HunyuanVideo's linear1 = [qkv | mlp], hunyuan/infer/transformer_infer.py:329-334)."""
        w, b = self.weight, self.bias
        if row_slice is not None:
            w, b = w[row_slice], (None if b is None else b[row_slice])
        return lib.gemm(input_tensor, w, b, epilogue=epilogue, resid=resid, gate=gate, out=out)

    def apply_vt(self, input_tensor, num_heads):
        """The layer's output as V^T [H, ceil(M/64), 128, 64] (the attention kernel's operand) — from the GEMM epilogue when the shape takes
        the single-stream kernel, else GEMM + transpose; same bits either way."""
        return lib.gemm_vt(input_tensor, self.weight, self.bias, num_heads)

    def state_dict(self, destination=None):
        destination = {} if destination is None else destination
        destination[self.weight_name] = self.weight.cpu().detach().clone()
        if self.bias is not None:
            destination[self.bias_name] = self.bias.cpu().detach().clone()
        return destination


@MM_WEIGHT_REGISTER("W-fp8-channel-sym-A-fp8-channel-sym-dynamic-Hip")
class MMWeightFp8Hip(_Movable):
    """reference: mm_weight.py:111-284 (template) + :287-319 (Vllm) / :532-589 (Sgl): e4m3fn weight [N,K] with
    fp32 per-out-channel scale `<name>.weight_scale` [N,1] (converter format, tools/convert/converter.py:294-339)
    or `weight_auto_quant` from bf16 (:167-173); per-token dynamic activation quant (:236-245); scaled GEMM."""

    _tensor_attrs = ("weight", "weight_scale", "bias")
    # apply() takes the Ulysses exchange buffers in place (round 5): an N-blocked 3-D `out` goes to x2v_gemm_fp8_blocked, a K-blocked 3-D bf16
    # input is de-blocked by its quantisation pass (lib.quant_fp8_rowwise -> x2v_quant_fp8_rowwise_blocked) and multiplied row-major
    accepts_blocked = True

    def __init__(self, weight_name, bias_name, lazy_load=False, lazy_load_file=None):
        self.weight_name, self.bias_name = weight_name, bias_name
        self.weight_scale_name = weight_name.removesuffix(".weight") + ".weight_scale"
        self.lazy_load, self.lazy_load_file = lazy_load, lazy_load_file
        self.config = {}
        self.weight = self.weight_scale = self.bias = None

    def load(self, weight_dict):
        w = weight_dict[self.weight_name]
        if self.config.get("weight_auto_quant", False) or w.dtype != torch.float8_e4m3fn:
            wf = w.to(torch.float32)
            # quant_utils.py:46-48.  A tensor divisor on purpose: dividing by a Python scalar becomes a multiplication by its rounded
            # reciprocal on the device, one ulp off for most rows — and w / scale hits exact e4m3 ties often (bf16 weights)
            scale = wf.abs().amax(dim=1, keepdim=True).clamp(min=1e-5) / torch.tensor(448.0, dtype=torch.float32, device=wf.device)
            self.weight = (wf / scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).contiguous()
            self.weight_scale = scale.to(torch.float32)
        else:
            self.weight = w.contiguous()
            self.weight_scale = weight_dict[self.weight_scale_name].float()
        self.bias = weight_dict[self.bias_name] if self.bias_name is not None else None

    def load_from_disk(self):
        """mm_weight.py:146-165 (lazy-load path of the quantised template): e4m3 weight + fp32 `<name>.weight_scale` from the open file."""
        self.weight = self.lazy_load_file.get_tensor(self.weight_name).contiguous()
        self.weight_scale = self.lazy_load_file.get_tensor(self.weight_scale_name).float()
        self.bias = self.lazy_load_file.get_tensor(self.bias_name).to(torch.bfloat16) if self.bias_name is not None else None

    def quantize_input(self, input_tensor):
        """Per-token dynamic e4m3 quantisation of an activation (mm_weight.py:236-245): (codes, fp32 scales).  The fused block driver calls
        it once per LayerNorm output and hands the pair to every projection that consumes that tensor (`apply(..., quantized=)`)."""
        return lib.quant_fp8_rowwise(input_tensor)

    @staticmethod
    def layernorm_quantize(x, weight=None, bias=None, scale=None, shift=None, eps=1e-6):
        """LayerNorm (+affine, +modulate) and `quantize_input` of its output in one kernel (bit-identical to the two in sequence)."""
        return lib.layernorm_quant_fp8(x, weight, bias, scale, shift, eps)

    def apply(self, input_tensor, epilogue=lib.EPI_NONE, resid=None, gate=None, out=None, row_slice=None, quantized=None):
        xq, sx = self.quantize_input(input_tensor) if quantized is None else quantized
        w, sw, b = self.weight, self.weight_scale, self.bias
        if row_slice is not None:
            w, sw, b = w[row_slice], sw[row_slice], (None if b is None else b[row_slice])
        if out is not None and out.dim() == 3:  # N-blocked y: a seq->head send buffer [N, S/N, (H/N) d]
            return lib.gemm_fp8_blocked(xq, sx, w, sw, b, epilogue=epilogue, out=out)
        return lib.gemm_fp8(xq, sx, w, sw, b, epilogue=epilogue, resid=resid, gate=gate, out=out)

    def state_dict(self, destination=None):
        destination = {} if destination is None else destination
        destination[self.weight_name] = self.weight.cpu().detach().clone()
        destination[self.weight_scale_name] = self.weight_scale.cpu().detach().clone()
        if self.bias is not None:
            destination[self.bias_name] = self.bias.cpu().detach().clone()
        return destination


# ------------------------------------------------------------------------------------------------ norms
@MM_WEIGHT_REGISTER("W-mxfp8-A-mxfp8-dynamic-Hip")
class MMWeightMxfp8Hip(_Movable):
    """MXFP8 linear layer (OCP microscaling: e4m3 elements, one e8m0 scale per 32 K elements for weights and activations alike) on
    gfx950's block-scaled MFMA — the operator-API form of the reference's lightx2v_kernel pair `scaled_fp8_quant` +
    `cutlass_scaled_mxfp8_mm` (lightx2v_kernel/python/lightx2v_kernel/gemm.py:73-97; test_mxfp8_quant.py:20-37 is its usage pattern:
    quantise the weight once, the activation per call, alpha = 1, bias fused).  The reference ships the kernels but no MMWeight
    class for them; the config key follows its naming scheme (mm_weight.py:287-589).  Weights arrive as bf16/fp16/fp32 [N, K] and are
    quantised at load (the `weight_auto_quant` path of the other quantised classes, :167-173)."""

    _tensor_attrs = ("weight", "weight_scale", "bias")

    def __init__(self, weight_name, bias_name, lazy_load=False, lazy_load_file=None):
        self.weight_name, self.bias_name = weight_name, bias_name
        self.lazy_load, self.lazy_load_file = lazy_load, lazy_load_file
        self.config = {}
        self.weight = self.weight_scale = self.bias = None

    def load(self, weight_dict):
        w = weight_dict[self.weight_name]
        if not w.is_cuda:
            raise lib.X2VError(f"{self.weight_name}: MXFP8 weights are quantised by the HIP kernel at load — tensor is on {w.device}")
        self.weight, self.weight_scale = lib.quant_mxfp8(w.to(torch.bfloat16).contiguous())
        self.bias = weight_dict[self.bias_name] if self.bias_name is not None else None

    def quantize_input(self, input_tensor):
        return lib.quant_mxfp8(input_tensor)

    def apply(self, input_tensor, epilogue=lib.EPI_NONE, resid=None, gate=None, out=None, row_slice=None, quantized=None):
        xq, sx = self.quantize_input(input_tensor) if quantized is None else quantized
        w, sw, b = self.weight, self.weight_scale, self.bias
        if row_slice is not None:  # scale table [K/128, N, 4]: slice its row axis; the kernel wants it contiguous
            w, sw, b = w[row_slice], sw[:, row_slice].contiguous(), (None if b is None else b[row_slice])
        return lib.gemm_mxfp8(xq, sx, w, sw, bias=b, epilogue=epilogue, resid=resid, gate=gate, out=out)

    def state_dict(self, destination=None):
        destination = {} if destination is None else destination
        destination[self.weight_name] = self.weight.cpu().detach().clone()
        destination[self.weight_name + "_scale"] = self.weight_scale.cpu().detach().clone()
        if self.bias is not None:
            destination[self.bias_name] = self.bias.cpu().detach().clone()
        return destination


@RMS_WEIGHT_REGISTER("hip")
class RMSWeightHip(_Movable):
    """reference: common/ops/norm/rms_norm_weight.py:53-118.  `round_mode`: fp32 statistics (what
    sgl_kernel.rmsnorm, the reference's GPU path, computes) or the torch bf16 chain of the CPU fallback."""

    _tensor_attrs = ("weight",)

    def __init__(self, weight_name, lazy_load=False, lazy_load_file=None, eps=1e-6):
        self.weight_name, self.eps = weight_name, eps
        self.lazy_load, self.lazy_load_file = lazy_load, lazy_load_file
        self.config = {}
        self.weight = None
        self.round_mode = lib.ROUND_FP32

    def load(self, weight_dict):
        if not self.lazy_load:
            self.weight = weight_dict[self.weight_name]

    def load_from_disk(self):
        self.weight = self.lazy_load_file.get_tensor(self.weight_name).to(torch.bfloat16)

    def apply(self, input_tensor):
        return lib.rmsnorm(input_tensor, self.weight, self.eps, round_mode=self.round_mode)

    def state_dict(self, destination=None):
        destination = {} if destination is None else destination
        destination[self.weight_name] = self.weight.cpu().detach().clone()
        return destination


@LN_WEIGHT_REGISTER("hip")
class LNWeightHip(_Movable):
    """reference: common/ops/norm/layer_norm_weight.py:78-111; `apply(x, scale=, shift=)` additionally fuses the
    adaLN modulate that follows every no-affine LN in the block (transformer_infer.py:329-334,481-484)."""

    _tensor_attrs = ("weight", "bias")

    def __init__(self, weight_name=None, bias_name=None, lazy_load=False, lazy_load_file=None, eps=1e-6):
        self.weight_name, self.bias_name, self.eps = weight_name, bias_name, eps
        self.lazy_load, self.lazy_load_file = lazy_load, lazy_load_file
        self.config = {}
        self.weight = self.bias = None

    def load(self, weight_dict):
        if not self.lazy_load:
            self.weight = weight_dict[self.weight_name] if self.weight_name is not None else None
            self.bias = weight_dict[self.bias_name] if self.bias_name is not None else None

    def apply(self, input_tensor, scale=None, shift=None):
        return lib.layernorm(input_tensor, self.weight, self.bias, scale, shift, self.eps)

    def state_dict(self, destination=None):
        destination = {} if destination is None else destination
        if self.weight is not None:
            destination[self.weight_name] = self.weight.cpu().detach().clone()
        if self.bias is not None:
            destination[self.bias_name] = self.bias.cpu().detach().clone()
        return destination


# ------------------------------------------------------------------------------------------------ attention
_CU_CACHE = {}  # id(tensor) -> (tensor, version, boundaries); entries PIN their tensor


def _segments(cu):
    """cu_seqlens (int tensor on any device, list, or None) → python list of boundaries.  A device tensor costs one host read.  The fused
    block loop hands the SAME tensor object to every layer, so the boundaries are cached per tensor object: an entry holds a reference to
    its tensor (so neither its id nor its storage address can be recycled by the caching allocator while the entry lives) and is valid only
    for that very object at the same version counter.  A caller that builds fresh cu tensors per call (the reference's op-by-op loop,
    wan/infer/transformer_infer.py:73-77) always misses and pays the 8-byte read — never a stale hit.  At most 8 tensors are pinned."""
    if cu is None:
        return None
    if not torch.is_tensor(cu):
        return [int(c) for c in cu]
    if not cu.is_cuda:
        return cu.tolist()
    hit = _CU_CACHE.get(id(cu))
    if hit is not None and hit[0] is cu and hit[1] == cu._version:
        return hit[2]
    while len(_CU_CACHE) >= 8:
        _CU_CACHE.pop(next(iter(_CU_CACHE)))
    bounds = cu.tolist()
    _CU_CACHE[id(cu)] = (cu, cu._version, bounds)
    return bounds


def hip_flash(q, k, v, cu_seqlens_q=None, cu_seqlens_kv=None, max_seqlen_q=None, max_seqlen_kv=None, model_cls=None, variant=0):
    """Functional form (reference: lightx2v/attentions/common/flash_attn2.py:8, dispatcher attentions/__init__.py:8-20) with
    flash_attn_varlen_func's contract (attn_weight.py:76-97): q [total_q, H, d], k/v [total_k, H, d]; segment i attends
    q[cu_q[i]:cu_q[i+1]] over k/v[cu_kv[i]:cu_kv[i+1]] — one launch per non-empty segment (Wan: one segment; HunyuanVideo builds
    [0, img + valid text, img + all text], hunyuan/infer/pre_infer.py:50-56).  Returns [max_seqlen_q, H*d] (`.reshape(max_seqlen_q, -1)`
    of the reference, :95; max_seqlen_q defaults to total_q); query rows that belong to no segment are zero.  Keys past the last
    boundary (a padded token buffer) are not attended, as with flash-attn."""
    H, d = q.shape[1], q.shape[2]
    total_q = q.shape[0]
    cq, ck = _segments(cu_seqlens_q), _segments(cu_seqlens_kv)
    if cq is None:
        cq = [0, total_q]
    if ck is None:
        ck = cq if k.shape[0] == total_q else [0, k.shape[0]]
    if len(cq) != len(ck) or len(cq) < 2:
        raise lib.X2VError(f"hip_flash: cu_seqlens_q ({len(cq)} entries) and cu_seqlens_kv ({len(ck)}) must describe the same number of sequences")
    if cq[-1] > total_q or ck[-1] > k.shape[0] or any(b < a for a, b in zip(cq[:-1], cq[1:])) or any(b < a for a, b in zip(ck[:-1], ck[1:])):
        raise lib.X2VError(f"hip_flash: cu_seqlens {cq} / {ck} do not fit q [{total_q}] / k [{k.shape[0]}] rows")
    rows = total_q if max_seqlen_q is None else int(max_seqlen_q)
    if rows * H * d != total_q * H * d:
        raise lib.X2VError(f"hip_flash: max_seqlen_q={rows} does not reshape a [{total_q}, {H}, {d}] result (the reference's .reshape(max_seqlen_q, -1))")
    out = torch.empty((total_q, H * d), dtype=q.dtype, device=q.device)
    if cq[0] > 0:
        out[: cq[0]].zero_()
    if cq[-1] < total_q:
        out[cq[-1] :].zero_()
    for (qa, qb), (ka, kb) in zip(zip(cq[:-1], cq[1:]), zip(ck[:-1], ck[1:])):
        if qb == qa:
            continue
        if kb == ka:
            raise lib.X2VError("hip_flash: a sequence with queries but no keys")
        lib.attention(q[qa:qb], k[ka:kb], v[ka:kb], num_heads=H, head_dim=d, out=out[qa:qb], variant=variant)
    return out


@ATTN_WEIGHT_REGISTER("hip_flash")
class HipFlashAttnWeight:
    """reference: common/ops/attn/attn_weight.py:71-126 (flash_attn2/3 keys), :209-239 (torch_sdpa): q,k,v [tokens,H,d] →
    [max_seqlen_q, H*d]; any number of sequences through cu_seqlens (see hip_flash)."""

    def __init__(self):
        self.config = {}

    def load(self, weight_dict):
        pass

    def set_config(self, config=None):
        if config is not None:
            self.config = config

    def apply(self, q, k, v, cu_seqlens_q=None, cu_seqlens_kv=None, max_seqlen_q=None, max_seqlen_kv=None, model_cls=None, mask_map=None):
        return hip_flash(q, k, v, cu_seqlens_q, cu_seqlens_kv, max_seqlen_q, max_seqlen_kv, model_cls)

    def to_cpu(self, non_blocking=False):
        pass

    def to_cuda(self, non_blocking=False):
        pass

    def state_dict(self, destination=None):
        return {} if destination is None else destination


# ------------------------------------------------------------------------------------------------ tensors / conv
@TENSOR_REGISTER("Default")
class DefaultTensor(_Movable):
    """reference: common/ops/tensor/tensor.py:6-47."""

    _tensor_attrs = ("tensor",)

    def __init__(self, tensor_name, lazy_load=False, lazy_load_file=None):
        self.tensor_name = tensor_name
        self.lazy_load, self.lazy_load_file = lazy_load, lazy_load_file
        self.tensor = None

    def load(self, weight_dict):
        if not self.lazy_load:
            self.tensor = weight_dict[self.tensor_name]

    def load_from_disk(self):
        self.tensor = self.lazy_load_file.get_tensor(self.tensor_name).to(torch.bfloat16)

    def state_dict(self, destination=None):
        destination = {} if destination is None else destination
        destination[self.tensor_name] = self.tensor.cpu().detach().clone()
        return destination


@CONV3D_WEIGHT_REGISTER("hip_patch")
class PatchEmbedConv3dHip(MMWeightHip):
    """reference: common/ops/conv/conv3d.py:29-75 as used by pre_infer.py:57 — a Conv3d whose kernel equals its stride (1,2,2) with no
    padding is a GEMM over non-overlapping patches: x[S, C*1*2*2] . W[D, C*4]^T + b.  `apply` returns the reference's layout
    [1, D, T, H/2, W/2] as a VIEW of the GEMM's token-major [S, D] result, so the caller's `flatten(2).transpose(1, 2)`
    (pre_infer.py:59) lands back on the contiguous [1, S, D] tensor without a copy."""

    def __init__(self, weight_name, bias_name, stride=(1, 2, 2), padding=0, dilation=1, groups=1):
        super().__init__(weight_name, bias_name)
        self.stride = (stride,) * 3 if isinstance(stride, int) else tuple(stride)
        if padding not in (0, (0, 0, 0)) or dilation not in (1, (1, 1, 1)) or groups != 1:
            raise lib.X2VError("hip_patch: only the patch-embedding form (no padding, no dilation, groups = 1) is a GEMM")

    def load(self, weight_dict):
        w = weight_dict[self.weight_name]
        if w.dim() == 5 and tuple(w.shape[2:]) != self.stride:
            raise lib.X2VError(f"hip_patch: kernel {tuple(w.shape[2:])} != stride {self.stride}: not a patch embedding")
        super().load(weight_dict)
        # the GEMM kernels advance K in 64-element tiles: a patch of C*pt*ph*pw values that is not a multiple of 64 (i2v: 36 channels x 4 = 144) is
        # zero-padded on both operands — the padding adds exact zeros to the fp32 sums, the result does not change
        self._k, self._ckpt_shape = self.weight.shape[1], tuple(w.shape)
        if self._k % 64:
            self.weight = torch.nn.functional.pad(self.weight, (0, 64 - self._k % 64)).contiguous()

    def state_dict(self, destination=None):
        """The checkpoint's tensor back: the K padding of `load` is an operand detail of the GEMM, not part of the parameter — without this
        an i2v round trip exported [D, 192] under patch_embedding.weight instead of [D, 36, 1, 2, 2]."""
        destination = super().state_dict(destination)
        destination[self.weight_name] = destination[self.weight_name][:, : self._k].reshape(self._ckpt_shape).contiguous()
        return destination

    def apply_tokens(self, input_tensor):
        """[1, C, T, H, W] → token-major [S, D] (what the fused driver consumes)."""
        # patches [T*(H/2)*(W/2), C*4] in (c, pt, ph, pw) order = the conv kernel's layout
        _, c, t, h, w = input_tensor.shape
        pt, ph, pw = self.stride
        x = input_tensor.reshape(c, t // pt, pt, h // ph, ph, w // pw, pw).permute(1, 3, 5, 0, 2, 4, 6)
        x = x.reshape((t // pt) * (h // ph) * (w // pw), c * pt * ph * pw)
        if x.shape[1] != self.weight.shape[1]:
            x = torch.nn.functional.pad(x, (0, self.weight.shape[1] - x.shape[1]))
        return lib.gemm(x.contiguous(), self.weight, self.bias)

    def apply(self, input_tensor):
        _, _, t, h, w = input_tensor.shape
        pt, ph, pw = self.stride
        y = self.apply_tokens(input_tensor)  # [S, D]
        return y.t().reshape(1, y.shape[1], t // pt, h // ph, w // pw)


# the reference's weight classes look norms up under these keys; inside this package they resolve to HIP
RMS_WEIGHT_REGISTER["sgl-kernel"] = RMSWeightHip
RMS_WEIGHT_REGISTER["Default"] = RMSWeightHip
LN_WEIGHT_REGISTER["Default"] = LNWeightHip
MM_WEIGHT_REGISTER["Default"] = MMWeightHip
