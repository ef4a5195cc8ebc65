"""ctypes binding of the C-ABI (include/x2v.h) + thin torch-tensor wrappers.

PyTorch is plumbing here: it owns device memory (caching allocator) and the stream; every wrapper passes
raw `data_ptr()`s, element strides and `torch.cuda.current_stream().cuda_stream` to libx2v_hip.so.
The library is REQUIRED: importing this module without the built .so, or calling an op on a non-gfx950
device, raises — there is no eager/PyTorch fallback on the product path.
"""
import ctypes
import os

import math

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("X2V_LIB_PATH") or os.path.join(_HERE, "libx2v_hip.so")  # X2V_LIB_PATH: an alternative build of the same sources (tools/build_variant.sh, A/B runs)

EPI_NONE, EPI_GELU_TANH, EPI_RESIDUAL, EPI_SILU = 0, 1, 2, 3
ACT_GELU_ERF = 4  # x2v.h X2V_ACT_GELU_ERF (activation() only: exact GELU of the i2v CLIP-feature MLP)
ROUND_FP32, ROUND_REF = 0, 1

_c_void_p, _i64, _i32, _f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float

# name -> argtypes (restype int unless listed in _RESTYPES); mirrors include/x2v.h one to one
PROTOTYPES = {
    "x2v_init": [_i32],
    "x2v_last_error": [],
    "x2v_version": [],
    "x2v_device_info": [_i32, ctypes.POINTER(_i32), ctypes.POINTER(_i32), ctypes.c_char_p, _i32],
    "x2v_switches": [ctypes.c_char_p, _i32],
    "x2v_quant_fp8_rowwise_blocked": [_c_void_p, _i64, _i32, _i64, _c_void_p, _i64, _c_void_p, _i64, _i32, _c_void_p],
    "x2v_rmsnorm_bf16": [_c_void_p, _i64, _c_void_p, _c_void_p, _i64, _i64, _i32, _f32, _i32, _c_void_p],
    "x2v_layernorm_bf16": [_c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _i64, _i64, _i32, _f32, _c_void_p],
    "x2v_layernorm_bf16_variant": [_c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _i64, _i64, _i32, _f32, _i32, _c_void_p],
    "x2v_rmsnorm_rope_scaled_bf16_variant": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _i64, _i32, _i64, _i32, _i32, _i32, _f32, _i32, _f32, _i32, _c_void_p],
    "x2v_rmsnorm_rope_bf16": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _i64, _i32, _i64, _i32, _i32, _i32, _f32, _i32, _c_void_p],
    "x2v_rmsnorm_rope_scaled_bf16": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _i64, _i32, _i64, _i32, _i32, _i32, _f32, _i32, _f32, _c_void_p],
    "x2v_gemm_bf16_vt": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _c_void_p, _i64, _i64, _i32, _i32, _c_void_p],
    "x2v_headnorm_rope_blocked_bf16": [_c_void_p, _c_void_p, _i64, _i32, _i64, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _i64, _i32, _i64, _f32, _i32, _f32, _c_void_p],
    "x2v_headnorm_rope_bf16": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _i64, _i32, _i64, _f32, _i32, _f32, _c_void_p],
    "x2v_gate_residual_bf16": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _i32, _c_void_p],
    "x2v_activation_bf16": [_c_void_p, _c_void_p, _i64, _i32, _c_void_p],
    "x2v_gemm_bf16": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _c_void_p, _i64, _i64, _i32, _i32, _i32, _c_void_p, _i64, _c_void_p, _c_void_p],
    "x2v_gemm_bf16_variant": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _c_void_p, _i64, _i64, _i32, _i32, _i32, _c_void_p, _i64, _c_void_p, _i32, _c_void_p],
    "x2v_gemm_bf16_blocked": [_c_void_p, _i64, _i32, _i64, _c_void_p, _i64, _c_void_p, _c_void_p, _i64, _i32, _i64, _i64, _i32, _i32, _i32, _c_void_p, _i64, _c_void_p, _c_void_p],
    "x2v_gemm_fp8_blocked": [_c_void_p, _i64, _i32, _i64, _c_void_p, _c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _i64, _i32, _i64, _i64, _i32, _i32, _i32, _c_void_p, _i64,
                             _c_void_p, _c_void_p],
    "x2v_rmsnorm_rope_blocked_bf16": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _i64, _i32, _i64, _i64, _i32, _i64, _i32, _i32, _i32, _f32,
                                      _i32, _f32, _c_void_p],
    "x2v_gemm_kernel_choice": [_i64, _i32, _i32, _i64, _i64, _i32],
    "x2v_attn_fwd_bf16": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _i64, _i64, _i32, _i32, _f32, _c_void_p],
    "x2v_transpose_heads_bf16": [_c_void_p, _i64, _c_void_p, _i64, _i64, _i32, _c_void_p],
    "x2v_attn_fwd_bf16_vt_batched": [_c_void_p, _i64, _i64, _c_void_p, _i64, _i64, _c_void_p, _i64, _i64, _c_void_p, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _f32, _i32, _c_void_p],
    "x2v_attn_vt_launch_plan": [_i64, _i64, _i32, _i32, _i32],
    "x2v_mfma_probe_bf16": [_i32, ctypes.POINTER(_f32), _c_void_p],
    "x2v_attn_fwd_bf16_vt": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _i64, _i64, _i32, _i32, _f32, _i32, _c_void_p],
    "x2v_attn_fwd_bf16_variant": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _i64, _i64, _i32, _i32, _f32, _i32, _c_void_p],
    "x2v_quant_fp8_rowwise": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _i32, _c_void_p],
    "x2v_layernorm_quant_fp8": [_c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _i64, _c_void_p, _i64, _i32, _f32, _c_void_p],
    "x2v_quant_mxfp8_bf16": [_c_void_p, _i64, _c_void_p, _i64, _c_void_p, _i64, _i32, _c_void_p],
    "x2v_gemm_mxfp8_variant": [_c_void_p, _i64, _c_void_p, _c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _i64, _i64, _i32, _i32, _i32, _c_void_p],
    "x2v_gemm_mxfp8_epi": [_c_void_p, _i64, _c_void_p, _c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _i64, _i64, _i32, _i32, _i32, _c_void_p, _i64, _c_void_p, _c_void_p],
    "x2v_gemm_mxfp8": [_c_void_p, _i64, _c_void_p, _c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _i64, _i64, _i32, _i32, _c_void_p],
    "x2v_gemm_fp8": [_c_void_p, _i64, _c_void_p, _c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _i64, _i64, _i32, _i32, _i32, _c_void_p, _i64, _c_void_p, _c_void_p],
    "x2v_gemm_fp8_variant": [_c_void_p, _i64, _c_void_p, _c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _i64, _i64, _i32, _i32, _i32, _c_void_p, _i64, _c_void_p, _i32, _c_void_p],
    "x2v_unipc_step_f32": [_c_void_p, _c_void_p, _c_void_p, _i32, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, ctypes.POINTER(_f32), _i32, _i32, _i64,
                           _c_void_p],
    "x2v_distill_step_f32": [_c_void_p, _c_void_p, _f32, _c_void_p, _i32, _c_void_p, _f32, _f32, _f32, _c_void_p, _c_void_p, _i64, _c_void_p],
    "x2v_sinusoid_embed_bf16": [_c_void_p, _c_void_p, _i32, _i32, _c_void_p],
    "x2v_causal_conv3d_f32": [_c_void_p, _c_void_p, _i32, _c_void_p, _c_void_p, _c_void_p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _c_void_p],
    "x2v_vae_conv_f32": [_c_void_p, _i64, _i64, _i64, _c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _c_void_p],
    "x2v_vae_prep_f32": [_c_void_p, _c_void_p, _i32, _i32, _i32, _i32, _c_void_p, _c_void_p, _c_void_p, _i32, _i32, _i64, _i64, _c_void_p],
    "x2v_softmax_rows_f32": [_c_void_p, _i64, _i64, _i32, _f32, _c_void_p],
    "x2v_vae_conv_f16": [_c_void_p, _i64, _i64, _i64, _c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _c_void_p],
    "x2v_vae_conv_f16_cached": [_c_void_p, _c_void_p, _i64, _i64, _i64, _c_void_p, _i64, _c_void_p, _c_void_p, _c_void_p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _c_void_p],
    "x2v_vae_conv_f16_cached_ok": [_i32, _i32, _i32, _i32, _i32, _i32],
    "x2v_vae_prep_f16": [_c_void_p, _c_void_p, _i32, _i32, _i32, _i32, _c_void_p, _c_void_p, _c_void_p, _i32, _i32, _i64, _i64, _i64, _c_void_p],
    "x2v_vae_prep_split_f16": [_c_void_p, _c_void_p, _i32, _i32, _i32, _i32, _c_void_p, _c_void_p, _c_void_p, _i32, _i32, _i64, _i64, _i64, _c_void_p],
    "x2v_vae_prep_ex_f16": [_c_void_p, _c_void_p, _i32, _i32, _i32, _i32, _c_void_p, _c_void_p, _i32, _i32, _i32, _i32, _i64, _i64, _c_void_p],
    "x2v_vae_prep_ex_f32": [_c_void_p, _c_void_p, _i32, _i32, _i32, _i32, _c_void_p, _c_void_p, _i32, _i32, _i32, _i32, _i64, _i64, _c_void_p],
    "x2v_vae_replicate_border_f32": [_c_void_p, _i32, _i32, _i32, _i32, _i32, _i32, _c_void_p],
    "x2v_groupnorm_affine_f32": [_c_void_p, _i64, _i32, _i32, _c_void_p, _c_void_p, _f32, _c_void_p, _c_void_p, _c_void_p, _c_void_p],
    "x2v_softmax_rows_causal_f32": [_c_void_p, _i64, _i64, _i32, _f32, _i32, _i32, _c_void_p],
    "x2v_blend_axis_f32": [_c_void_p, _c_void_p, _i64, _i32, _i32, _i64, _i64, _i64, _i32, _c_void_p],
}
_RESTYPES = {"x2v_last_error": ctypes.c_char_p, "x2v_version": ctypes.c_char_p}


class X2VError(RuntimeError):
    pass


def load_library(path=LIB_PATH):
    if not os.path.exists(path):
        raise X2VError(f"{path} is missing: build it with `python -m lightx2v_amd.build` (hipcc, gfx950). There is no fallback path.")
    lib = ctypes.CDLL(path)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library disagree
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, ctypes.c_int)
    return lib


_lib = load_library()
_inited = set()


def version():
    return _lib.x2v_version().decode()


def switches():
    """The library's process-wide A/B switches (effective values, latched from the environment at first use) as a dict — x2v_switches."""
    buf = ctypes.create_string_buffer(256)
    _check(_lib.x2v_switches(buf, 256), "x2v_switches")
    return {k: int(v) for k, v in (kv.split("=") for kv in buf.value.decode().split())}


def _check(rc, what):
    if rc != 0:
        raise X2VError(f"{what} failed ({rc}): {_lib.x2v_last_error().decode()}")


def init(device_index=None):
    """Verify the device is gfx950 (raises otherwise)."""
    if not torch.cuda.is_available():
        raise X2VError("no HIP device visible: the x2v HIP path has no CPU fallback")
    idx = torch.cuda.current_device() if device_index is None else device_index
    if idx not in _inited:
        _check(_lib.x2v_init(idx), "x2v_init")
        _inited.add(idx)
    return idx


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _row2d(t, name):
    if t.dim() != 2 or t.stride(1) != 1:
        raise X2VError(f"{name}: expected a 2-D tensor with unit inner stride, got shape {tuple(t.shape)} strides {t.stride()}")
    if not t.is_cuda:
        raise X2VError(f"{name}: tensor is on {t.device}; the x2v HIP path has no CPU fallback")
    return t


def _bf16(t, name):
    if t.dtype != torch.bfloat16:
        raise X2VError(f"{name}: expected bfloat16, got {t.dtype}")
    return t


def _vec(t, name, n=None, dtype=torch.bfloat16):
    """A per-channel operand (bias / norm weight / gate / scale row): None passes through; otherwise a contiguous device vector of
    `dtype` with `n` elements (any leading unit dims) — the C side only sees a pointer, so a wrong dtype would be silent."""
    if t is None:
        return None
    if t.dtype != dtype:
        raise X2VError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_cuda:
        raise X2VError(f"{name}: tensor is on {t.device}; the x2v HIP path has no CPU fallback")
    if n is not None and t.numel() != n:
        raise X2VError(f"{name}: expected {n} elements, got shape {tuple(t.shape)}")
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------------------------------
def rmsnorm(x, weight, eps=1e-6, out=None, round_mode=ROUND_FP32):
    shape = x.shape
    x2 = _row2d(_bf16(x.reshape(-1, shape[-1]) if x.dim() != 2 else x, "x"), "x")
    out2 = torch.empty((x2.shape[0], x2.shape[1]), dtype=torch.bfloat16, device=x.device) if out is None else _row2d(out.reshape(-1, shape[-1]) if out.dim() != 2 else out, "out")
    init()
    if x2.shape[0] == 0:  # empty shard: torch hands out a null data_ptr, nothing to launch
        return out2.view(shape) if out is None else out
    _check(_lib.x2v_rmsnorm_bf16(_p(x2), x2.stride(0), _p(_vec(weight, "rmsnorm weight", x2.shape[1])), _p(out2), out2.stride(0), x2.shape[0], x2.shape[1], eps, round_mode, _stream()), "rmsnorm")
    return out2.view(shape) if out is None else out


def layernorm(x, weight=None, bias=None, scale=None, shift=None, eps=1e-6, out=None, variant=0):
    """LN(x)[*w+b] then optional adaLN `* (1 + scale) + shift` (scale/shift: [D] or [1,D] bf16).  variant: x2v_layernorm_bf16_variant."""
    x2 = _row2d(_bf16(x, "x"), "x")
    out2 = torch.empty_like(x2) if out is None else _row2d(out, "out")
    D = x2.shape[1]
    if (scale is None) != (shift is None):
        raise X2VError("layernorm: scale and shift must be given together")
    weight, bias = _vec(weight, "layernorm weight", D), _vec(bias, "layernorm bias", D)
    scale, shift = _vec(scale, "layernorm scale", D), _vec(shift, "layernorm shift", D)
    init()
    if x2.shape[0] == 0:
        return out2
    _check(
        _lib.x2v_layernorm_bf16_variant(_p(x2), x2.stride(0), _p(weight), _p(bias), _p(scale), _p(shift), _p(out2), out2.stride(0), x2.shape[0], x2.shape[1], eps, variant,
                                        _stream()),
        "layernorm",
    )
    return out2


ATTN_Q_PRESCALED = 0x100
ATTN_STAGGER = 0x200  # x2v.h X2V_ATTN_VT_STAGGER: query block b starts its key walk (b mod 8) tiles in (an opt-in; wan.SELF_ATTN_STAGGER decides for the Wan drivers)
ATTN_ONE_WALK = 0x400  # x2v.h X2V_ATTN_VT_ONE_WALK: never the persistent short-walk form (A/B runs, the bit-equality test of the two forms)
ATTN_FAST = 12  # "ping-pong" kernel on a pre-transposed V (x2v_transpose_heads_bf16 + x2v_attn_fwd_bf16_vt); used with
#                 ATTN_Q_PRESCALED by the fused block drivers.  attention() does the transposition itself for this variant.
ATTN_PRESCALE = 1.4426950408889634 / math.sqrt(128.0)  # softmax scale * log2(e) for head_dim 128


def rmsnorm_rope_(q, k, wq, wk, rope_cs, grid, num_heads, s0=0, eps=1e-6, round_mode=ROUND_FP32, q_out_scale=1.0, variant=0):
    """In place: q,k [S, H*128] ← RoPE3D(RMSNorm(q|k)); q additionally * q_out_scale inside its final rounding."""
    q2, k2 = _row2d(_bf16(q, "q"), "q"), _row2d(_bf16(k, "k"), "k")
    if rope_cs.dtype != torch.float32 or tuple(rope_cs.shape) != (1024, 64, 2) or not rope_cs.is_contiguous():
        raise X2VError("rope_cs must be a contiguous float32 [1024,64,2] (cos,sin) table")
    gf, gh, gw = grid
    wq, wk = _vec(wq, "norm_q weight", q2.shape[1]), _vec(wk, "norm_k weight", k2.shape[1])
    init()
    if q2.shape[0] == 0:
        return q, k
    _check(
        _lib.x2v_rmsnorm_rope_scaled_bf16_variant(_p(q2), q2.stride(0), _p(k2), k2.stride(0), _p(wq), _p(wk), _p(rope_cs), q2.shape[0], num_heads, s0, gf, gh, gw, eps,
                                                  round_mode, q_out_scale, variant, _stream()),
        "rmsnorm_rope",
    )
    return q, k


def rmsnorm_rope_blocked(q, k, wq, wk, rope_cs, grid, num_heads, q_out, k_out, s0=0, eps=1e-6, round_mode=ROUND_FP32, q_out_scale=1.0):
    """x2v_rmsnorm_rope_blocked_bf16: RoPE3D(RMSNorm(q|k)) written out of place into N-blocked buffers q_out / k_out [B, S, H*128/B]
    (the Ulysses seq->head send buffers)."""
    q2, k2 = _row2d(_bf16(q, "q"), "q"), _row2d(_bf16(k, "k"), "k")
    if rope_cs.dtype != torch.float32 or tuple(rope_cs.shape) != (1024, 64, 2) or not rope_cs.is_contiguous():
        raise X2VError("rope_cs must be a contiguous float32 [1024,64,2] (cos,sin) table")
    cb, cbs, ldo = _blocks3d(_bf16(q_out, "q_out"), "q_out")
    if tuple(k_out.shape) != tuple(q_out.shape) or k_out.stride() != q_out.stride() or k_out.dtype != torch.bfloat16 or not k_out.is_cuda:
        raise X2VError("rmsnorm_rope_blocked: q_out and k_out must have the same shape, strides and dtype")
    if q_out.shape[0] * cb != num_heads * 128 or q_out.shape[1] != q2.shape[0]:
        raise X2VError(f"rmsnorm_rope_blocked: outputs {tuple(q_out.shape)} do not hold [{q2.shape[0]}, {num_heads * 128}]")
    gf, gh, gw = grid
    wq, wk = _vec(wq, "norm_q weight", q2.shape[1]), _vec(wk, "norm_k weight", k2.shape[1])
    init()
    if q2.shape[0] == 0:
        return q_out, k_out
    _check(
        _lib.x2v_rmsnorm_rope_blocked_bf16(_p(q2), q2.stride(0), _p(k2), k2.stride(0), _p(wq), _p(wk), _p(rope_cs), _p(q_out), _p(k_out), ldo, cb, cbs, q2.shape[0], num_heads, s0, gf,
                                           gh, gw, eps, round_mode, q_out_scale, _stream()),
        "rmsnorm_rope_blocked",
    )
    return q_out, k_out


def gate_residual_(x, y, gate=None):
    x2, y2 = _row2d(_bf16(x, "x"), "x"), _row2d(_bf16(y, "y"), "y")
    gate = _vec(gate, "gate", x2.shape[1])
    init()
    if x2.shape[0] == 0:
        return x
    _check(_lib.x2v_gate_residual_bf16(_p(x2), x2.stride(0), _p(y2), y2.stride(0), _p(gate), x2.shape[0], x2.shape[1], _stream()), "gate_residual")
    return x


def activation(x, act):
    xc = _bf16(x, "x").contiguous()
    out = torch.empty_like(xc)
    init()
    _check(_lib.x2v_activation_bf16(_p(xc), _p(out), xc.numel(), act, _stream()), "activation")
    return out


def _blocks3d(t, name):
    """A block-strided operand [B, M, c] (unit inner stride): returns (block columns, block stride, row stride)."""
    if t.dim() != 3 or t.stride(2) != 1 or not t.is_cuda:
        raise X2VError(f"{name}: a blocked operand is a 3-D device tensor [blocks, rows, cols] with unit inner stride, got {tuple(t.shape)} / {t.stride()}")
    return t.shape[2], t.stride(0), t.stride(1)


def gemm_blocked(x, weight_nk, bias=None, epilogue=EPI_NONE, out=None):
    """x2v_gemm_bf16_blocked.  `x`: [M, K] or K-blocked [B, M, K/B]; `out`: None / [M, N] or N-blocked [B', M, N/B'] (preallocated) — the
    Ulysses exchange buffers read / written in place.  Returns `out`."""
    w2 = _row2d(_bf16(weight_nk, "weight"), "weight")
    N, K = w2.shape
    if x.dim() == 3:
        kb, kbs, ldx = _blocks3d(_bf16(x, "x"), "x")
        M = x.shape[1]
        if kb * x.shape[0] != K:
            raise X2VError(f"gemm_blocked: x blocks {tuple(x.shape)} do not make K={K}")
    else:
        x2 = _row2d(_bf16(x, "x"), "x")
        kb, kbs, ldx, M = 0, 0, x2.stride(0), x2.shape[0]
        if x2.shape[1] != K:
            raise X2VError(f"gemm_blocked: x [M,{x2.shape[1]}] vs weight [{N},{K}]")
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    if out.dim() == 3:
        nb, nbs, ldy = _blocks3d(_bf16(out, "out"), "out")
        if nb * out.shape[0] != N or out.shape[1] != M:
            raise X2VError(f"gemm_blocked: out blocks {tuple(out.shape)} do not make [{M}, {N}]")
    else:
        o2 = _row2d(_bf16(out, "out"), "out")
        nb, nbs, ldy = 0, 0, o2.stride(0)
        if tuple(o2.shape) != (M, N):
            raise X2VError(f"gemm_blocked: out is {tuple(o2.shape)}, expected {(M, N)}")
    if epilogue == EPI_RESIDUAL:
        raise X2VError("gemm_blocked: use gemm(..., resid=) for the residual epilogue (x may be 3-D there)")
    bias = _vec(bias, "gemm bias", N)
    init()
    if M == 0:
        return out
    _check(_lib.x2v_gemm_bf16_blocked(_p(x), ldx, kb, kbs, _p(w2), w2.stride(0), _p(bias), _p(out), ldy, nb, nbs, M, N, K, epilogue, None, 0, None, _stream()), "gemm_bf16_blocked")
    return out


def gemm(x, weight_nk, bias=None, epilogue=EPI_NONE, resid=None, gate=None, out=None, variant=0):
    """y = epi(x @ weight_nk.T + bias); weight_nk is the checkpoint's [N,K] tensor.  variant: see x2v_gemm_bf16_variant.
    A 3-D `x` ([B, M, K/B], K-blocked) or 3-D `out` ([B, M, N/B], N-blocked) goes through x2v_gemm_bf16_blocked."""
    if x.dim() == 3 or (out is not None and out.dim() == 3):
        if epilogue != EPI_RESIDUAL:
            return gemm_blocked(x, weight_nk, bias, epilogue, out)
        # residual epilogue on a K-blocked x (the Ulysses output projection): y = resid, row-major
        w2 = _row2d(_bf16(weight_nk, "weight"), "weight")
        N, K = w2.shape
        kb, kbs, ldx = _blocks3d(_bf16(x, "x"), "x")
        M = x.shape[1]
        if kb * x.shape[0] != K or resid is None:
            raise X2VError(f"gemm: K-blocked x {tuple(x.shape)} vs weight [{N},{K}] (residual epilogue needs resid)")
        out2 = _row2d(_bf16(resid if out is None else out, "out"), "out")
        r2 = _row2d(_bf16(resid, "resid"), "resid")
        if tuple(out2.shape) != (M, N):
            raise X2VError(f"gemm: out is {tuple(out2.shape)}, expected {(M, N)}")
        init()
        if M == 0:
            return out2
        _check(_lib.x2v_gemm_bf16_blocked(_p(x), ldx, kb, kbs, _p(w2), w2.stride(0), _p(_vec(bias, "gemm bias", N)), _p(out2), out2.stride(0), 0, 0, M, N, K, epilogue, _p(r2),
                                          r2.stride(0), _p(_vec(gate, "gemm gate", N)), _stream()), "gemm_bf16_blocked")
        return out2
    x2, w2 = _row2d(_bf16(x, "x"), "x"), _row2d(_bf16(weight_nk, "weight"), "weight")
    M, K = x2.shape
    N = w2.shape[0]
    if w2.shape[1] != K:
        raise X2VError(f"gemm: x [M,{K}] vs weight [{N},{w2.shape[1]}]")
    if epilogue == EPI_RESIDUAL:
        if resid is None:
            raise X2VError("gemm: residual epilogue needs resid")
        out2 = _row2d(resid if out is None else out, "out")
        r2 = _row2d(_bf16(resid, "resid"), "resid")
        gate = _vec(gate, "gemm gate", N)
    else:
        out2 = torch.empty((M, N), dtype=torch.bfloat16, device=x.device) if out is None else _row2d(out, "out")
        r2 = None
    bias = _vec(bias, "gemm bias", N)
    if _bf16(out2, "out").shape != (M, N):
        raise X2VError(f"gemm: out is {tuple(out2.shape)}, expected {(M, N)}")
    init()
    if M == 0:
        return out2
    _check(
        _lib.x2v_gemm_bf16_variant(_p(x2), x2.stride(0), _p(w2), w2.stride(0), _p(bias), _p(out2), out2.stride(0), M, N, K, epilogue, _p(r2), 0 if r2 is None else r2.stride(0), _p(gate), variant, _stream()),
        "gemm_bf16",
    )
    return out2


def attention_batched(q, k, vt, num_heads, batch, rows_per_seq, seq_len, out=None, prescaled=False, scale=0.0, all_rows_query=True, one_launch=None, timed=None, stagger=False):
    """`batch` independent self-attentions over stacked rows (x2v_attn_fwd_bf16_vt_batched): q, k, out are [batch * rows_per_seq, H*128]
    row-major (any token stride), sequence b in rows [b * rows_per_seq, b * rows_per_seq + seq_len); vt = V^T [H, batch * rows_per_seq / 64,
    128, 64] over the stacked rows (gemm_vt / transpose_heads of the stacked v; rows_per_seq % 64 == 0).  Keys are the first seq_len rows of
    a sequence's slot; with all_rows_query (default) every row of the slot is a query, so every row of `out` is written (the padding rows
    of a stacked activation buffer stay finite), else only the first seq_len.
    one_launch: True = all sequences in one launch (grid z), False = one launch per sequence, None = by size — measured on MI355X: one
    launch wins when a sequence alone already fills the chip many times over (Wan-14B 720p, 11 840 workgroups each: -0.85 %), and loses
    when it does not (Wan-1.3B 480p, 960 workgroups each: +2 %, the two sequences' K/V share the L2s).  Same results either way.
    timed: optional callable(fn) that runs fn() — called once per kernel launch (bench.py's per-launch HIP-event timer).
    stagger: X2V_ATTN_VT_STAGGER (see x2v.h): same for every launch form here, since a row's query block index does not depend on it."""
    q2, k2 = _row2d(_bf16(q, "q"), "q"), _row2d(_bf16(k, "k"), "k")
    rows = batch * rows_per_seq
    if rows_per_seq % 64 or not 0 < seq_len <= rows_per_seq or q2.shape[0] != rows or k2.shape[0] != rows or min(q2.shape[1], k2.shape[1]) < num_heads * 128:
        raise X2VError(f"attention_batched: q {tuple(q2.shape)} k {tuple(k2.shape)} do not hold {batch} sequences of {rows_per_seq} (multiple of 64) rows x {num_heads} heads")
    if vt.dtype != torch.bfloat16 or not vt.is_cuda or not vt.is_contiguous() or tuple(vt.shape) != (num_heads, rows // 64, 128, 64):
        raise X2VError(f"attention_batched: vt must be the contiguous bf16 [H, rows/64, 128, 64] tensor over the stacked rows, got {tuple(vt.shape)}")
    out2 = torch.empty((rows, num_heads * 128), dtype=torch.bfloat16, device=q.device) if out is None else _row2d(_bf16(out, "out"), "out")
    if out2.shape[0] != rows or out2.shape[1] < num_heads * 128:
        raise X2VError(f"attention_batched: out is {tuple(out2.shape)}")
    init()
    sq = rows_per_seq if all_rows_query else seq_len
    if one_launch is None:
        one_launch = ((sq + 255) // 256) * num_heads >= 4096
    qb, kb, ob, vb = rows_per_seq * q2.stride(0), rows_per_seq * k2.stride(0), rows_per_seq * out2.stride(0), rows_per_seq * 128
    for b0, nb in ([(0, batch)] if one_launch else [(b, 1) for b in range(batch)]):
        def launch(b0=b0, nb=nb):
            _check(_lib.x2v_attn_fwd_bf16_vt_batched(q2.data_ptr() + 2 * b0 * qb, q2.stride(0), qb, k2.data_ptr() + 2 * b0 * kb, k2.stride(0), kb, vt.data_ptr() + 2 * b0 * vb, rows,
                                                     vb, out2.data_ptr() + 2 * b0 * ob, out2.stride(0), ob, sq, seq_len, num_heads, nb, 128, scale, int(bool(prescaled)) | (2 if stagger else 0), _stream()),
                   "attn_fwd_vt_batched")

        timed(launch) if timed is not None else launch()
    return out2


def gemm_vt(x, weight_nk, bias, num_heads):
    """V^T [H, ceil(M/64), 128, 64] of v = x @ weight_nk.T + bias (what transpose_heads(gemm(...)) returns, same bits): one kernel when the
    shape takes the single-stream 256x256 GEMM (x2v_gemm_bf16_vt), the two-kernel sequence otherwise."""
    x2, w2 = _row2d(_bf16(x, "x"), "x"), _row2d(_bf16(weight_nk, "weight"), "weight")
    M, K = x2.shape
    N = w2.shape[0]
    if w2.shape[1] != K or N != num_heads * 128:
        raise X2VError(f"gemm_vt: x [M,{K}] vs weight [{N},{w2.shape[1]}] for {num_heads} heads of 128")
    init()
    if M == 0 or (_lib.x2v_gemm_kernel_choice(M, N, K, x2.stride(0), w2.stride(0), 0) & 0xff) != 3:
        return transpose_heads(gemm(x2, w2, bias), num_heads)
    ldvt = (M + 63) // 64 * 64
    vt = torch.empty((num_heads, ldvt // 64, 128, 64), dtype=torch.bfloat16, device=x.device)
    _check(_lib.x2v_gemm_bf16_vt(_p(x2), x2.stride(0), _p(w2), w2.stride(0), _p(_vec(bias, "gemm bias", N)), _p(vt), ldvt, M, N, K, _stream()), "gemm_bf16_vt")
    return vt


def gemm_kernel_choice(M, N, K, ldx=None, ldw=None, fp8=False, with_form=False):
    """Tile family variant 0 launches for this shape (x2v_gemm_kernel_choice): 1 = 128x128 kernel, 2 = the 256x256 fp8 kernels, 3 = the 256x256
    single-stream kernel (bf16).  with_form: (family, continuous) — continuous = the continuous-pipeline form (gemm256c / gemm256c8) runs for a
    row-major y, else one tile per workgroup / ping-pong (same bits)."""
    rc = _lib.x2v_gemm_kernel_choice(M, N, K, K if ldx is None else ldx, K if ldw is None else ldw, int(fp8))
    if rc < 0:
        raise X2VError(f"gemm_kernel_choice: bad shape M={M} N={N} K={K}")
    return (rc & 0xff, bool(rc & 0x100)) if with_form else rc & 0xff


def attn_vt_launch_plan(Sq, Sk, num_heads, batch=1, stagger=False, one_walk=False, with_short=False):
    """(xcd_remap, staggered_walk) that x2v_attn_fwd_bf16_vt(_batched) takes for this shape (x2v_attn_vt_launch_plan; host-only);
    with_short: (xcd_remap, staggered_walk, persistent_short_walk)."""
    rc = _lib.x2v_attn_vt_launch_plan(Sq, Sk, num_heads, batch, (2 if stagger else 0) | (4 if one_walk else 0))
    if rc < 0:
        raise X2VError(f"attn_vt_launch_plan: bad shape Sq={Sq} Sk={Sk} H={num_heads} B={batch}")
    return (bool(rc & 1), bool(rc & 0x100), bool(rc & 0x200)) if with_short else (bool(rc & 1), bool(rc & 0x100))


def mfma_probe(milliseconds=1500):
    """TFLOP/s this board sustains on bare 16x16x32 bf16 MFMAs right now (x2v_mfma_probe_bf16): the box calibration bench.py prints."""
    init()
    out = _f32(0.0)
    _check(_lib.x2v_mfma_probe_bf16(int(milliseconds), ctypes.byref(out), _stream()), "mfma_probe")
    return float(out.value)


def transpose_heads(v, num_heads):
    """v [Sk, H*128] (any token stride) → V^T [H, ceil(Sk/64), 128, 64] (per head and 64-key tile a contiguous [dv][key] block)
    with the key padding zero-filled: the operand of the pre-transposed-V attention kernel."""
    v2 = _row2d(_bf16(v, "v"), "v")
    Sk = v2.shape[0]
    ldvt = (Sk + 63) // 64 * 64
    vt = torch.empty((num_heads, ldvt // 64, 128, 64), dtype=torch.bfloat16, device=v.device)
    init()
    _check(_lib.x2v_transpose_heads_bf16(_p(v2), v2.stride(0), _p(vt), ldvt, Sk, num_heads, _stream()), "transpose_heads")
    return vt


def attention(q, k, v, num_heads, head_dim=128, scale=0.0, out=None, variant=0, vt=None):
    """q [Sq, H*d] (or [Sq,H,d]), k/v [Sk, H*d]; any token stride (fused-QKV views are fine).
    variant ATTN_FAST runs on V^T: pass `vt` (transpose_heads(v, H)) or let this call transpose v."""

    def as2d(t, name):
        if t.dim() == 3:
            if t.stride(2) != 1 or t.stride(1) != t.shape[2]:
                raise X2VError(f"attention: {name} heads must be packed (stride {t.stride()})")
            t = t.as_strided((t.shape[0], t.shape[1] * t.shape[2]), (t.stride(0), 1))
        return _row2d(_bf16(t, name), name)

    if v is None:  # V^T only (gemm_vt): the pre-transposed-V kernel never reads row-major v
        if vt is None or (variant & 0xFF) != ATTN_FAST:
            raise X2VError("attention: v may be omitted only together with vt= and variant ATTN_FAST")
        v = k
    q2, k2, v2 = as2d(q, "q"), as2d(k, "k"), as2d(v, "v")
    Sq, Sk = q2.shape[0], k2.shape[0]
    out2 = torch.empty((Sq, num_heads * head_dim), dtype=torch.bfloat16, device=q.device) if out is None else _row2d(_bf16(out, "out"), "out")
    if out2.shape[0] != Sq or v2.shape[0] != Sk or min(q2.shape[1], k2.shape[1], v2.shape[1], out2.shape[1]) < num_heads * head_dim:
        raise X2VError(f"attention: q {tuple(q2.shape)} k {tuple(k2.shape)} v {tuple(v2.shape)} out {tuple(out2.shape)} do not describe {num_heads} heads of {head_dim}")
    if Sk == 0:
        raise X2VError("attention: no keys (softmax over an empty set)")
    init()
    if Sq == 0:
        return out2
    if (variant & 0xFF) == ATTN_FAST:
        if vt is None:
            vt = transpose_heads(v2, num_heads)
        elif vt.dtype != torch.bfloat16 or not vt.is_cuda or not vt.is_contiguous() or tuple(vt.shape) != (num_heads, (Sk + 63) // 64, 128, 64):
            raise X2VError(f"attention: vt must be the contiguous bf16 [H, ceil(Sk/64), 128, 64] tensor of transpose_heads, got {tuple(vt.shape)}")
        _check(
            _lib.x2v_attn_fwd_bf16_vt(_p(q2), q2.stride(0), _p(k2), k2.stride(0), _p(vt), vt.shape[1] * 64, _p(out2), out2.stride(0), Sq, Sk, num_heads, head_dim, scale,
                                      (1 if (variant & ATTN_Q_PRESCALED) else 0) | (2 if (variant & ATTN_STAGGER) else 0) | (4 if (variant & ATTN_ONE_WALK) else 0), _stream()),
            "attn_fwd_vt",
        )
        return out2
    _check(
        _lib.x2v_attn_fwd_bf16_variant(_p(q2), q2.stride(0), _p(k2), k2.stride(0), _p(v2), v2.stride(0), _p(out2), out2.stride(0), Sq, Sk, num_heads, head_dim, scale, variant, _stream()),
        "attn_fwd",
    )
    return out2


def quant_fp8_rowwise(x):
    """(e4m3 codes [M, K] row-major, fp32 scales [M, 1]) of a bf16 x [M, K] — or of a K-blocked x [B, M, K/B] (the Ulysses head->seq receive
    buffer; x2v_quant_fp8_rowwise_blocked: the codes come out row-major, the de-blocking rides in the quantisation pass)."""
    if x.dim() == 3:
        kb, kbs, ldx = _blocks3d(_bf16(x, "x"), "x")
        M, K = x.shape[1], kb * x.shape[0]
        x2 = x
    else:
        x2 = _row2d(_bf16(x, "x"), "x")
        (M, K), kb, kbs, ldx = x2.shape, 0, 0, x2.stride(0)
    xq = torch.empty((M, K), dtype=torch.float8_e4m3fn, device=x.device)
    s = torch.empty((M, 1), dtype=torch.float32, device=x.device)
    init()
    if M == 0:
        return xq, s
    _check(_lib.x2v_quant_fp8_rowwise_blocked(_p(x2), ldx, kb, kbs, _p(xq), xq.stride(0), _p(s), M, K, _stream()), "quant_fp8_rowwise")
    return xq, s


def layernorm_quant_fp8(x, weight=None, bias=None, scale=None, shift=None, eps=1e-6):
    """(e4m3 codes [M, D], fp32 scales [M, 1]) of LN(x)[*w+b][*(1+scale)+shift] — x2v_layernorm_quant_fp8, bit-identical to
    quant_fp8_rowwise(layernorm(...)).  Rows of <= 512 elements go through the two kernels."""
    x2 = _row2d(_bf16(x, "x"), "x")
    M, D = x2.shape
    if D <= 512:
        return quant_fp8_rowwise(layernorm(x2, weight, bias, scale, shift, eps))
    if (scale is None) != (shift is None):
        raise X2VError("layernorm_quant_fp8: scale and shift must be given together")
    weight, bias = _vec(weight, "layernorm weight", D), _vec(bias, "layernorm bias", D)
    scale, shift = _vec(scale, "layernorm scale", D), _vec(shift, "layernorm shift", D)
    xq = torch.empty((M, D), dtype=torch.float8_e4m3fn, device=x.device)
    s = torch.empty((M, 1), dtype=torch.float32, device=x.device)
    init()
    if M == 0:
        return xq, s
    _check(_lib.x2v_layernorm_quant_fp8(_p(x2), x2.stride(0), _p(weight), _p(bias), _p(scale), _p(shift), _p(xq), xq.stride(0), _p(s), M, D, eps, _stream()), "layernorm_quant_fp8")
    return xq, s


def quant_mxfp8(x):
    """bf16 [M, K] → (e4m3 [M, K], e8m0 scale bytes [K/128, M, 4]) — see include/x2v.h::x2v_quant_mxfp8_bf16.
    `mx_scales_rowmajor` turns the scale tensor into the logical [M, K/32] table."""
    x2 = _row2d(_bf16(x, "x"), "x")
    M, K = x2.shape
    q = torch.empty((M, K), dtype=torch.float8_e4m3fn, device=x.device)
    sc = torch.empty((max(K // 128, 1), M, 4), dtype=torch.uint8, device=x.device)
    init()
    if M == 0:
        return q, sc
    _check(_lib.x2v_quant_mxfp8_bf16(_p(x2), x2.stride(0), _p(q), q.stride(0), _p(sc), M, K, _stream()), "quant_mxfp8")
    return q, sc


def mx_scales_rowmajor(sc):
    """[K/128, rows, 4] (kernel layout) → logical [rows, K/32]."""
    return sc.permute(1, 0, 2).reshape(sc.shape[1], -1)


def mx_scales_tiled(sc_rm):
    """logical [rows, K/32] → [K/128, rows, 4] (kernel layout)."""
    rows, kb = sc_rm.shape
    return sc_rm.reshape(rows, kb // 4, 4).permute(1, 0, 2).contiguous()


def gemm_mxfp8(a, sa, b, sb, alpha=None, bias=None, out=None, variant=0, epilogue=EPI_NONE, resid=None, gate=None):
    """alpha * deq(a)[M,K] @ deq(b)[N,K]^T + bias → bf16 [M,N]; scales uint8/e8m0 [K/128, rows, 4] as produced by quant_mxfp8;
    alpha: fp32 device tensor or None."""
    M, K = a.shape
    N = b.shape[0]
    for t, name in ((a, "a"), (b, "b")):
        if t.dtype not in (torch.float8_e4m3fn, torch.uint8) or not t.is_cuda or t.stride(1) != 1:
            raise X2VError(f"gemm_mxfp8: {name} must be a CUDA e4m3 (or raw uint8) matrix with unit inner stride")
    for t, name, rows in ((sa, "scales_a", M), (sb, "scales_b", N)):
        if t.element_size() != 1 or not t.is_cuda or not t.is_contiguous() or tuple(t.shape) != (K // 128, rows, 4):
            raise X2VError(f"gemm_mxfp8: {name} must be the contiguous CUDA byte tensor [K/128, {rows}, 4] of quant_mxfp8, got {tuple(t.shape)}")
    if b.shape[1] != K or K % 128:
        raise X2VError(f"gemm_mxfp8: shapes a{tuple(a.shape)} b{tuple(b.shape)} do not agree (K % 128 == 0)")
    if alpha is not None and (alpha.dtype != torch.float32 or not alpha.is_cuda):
        raise X2VError("gemm_mxfp8: alpha must be a float32 CUDA tensor")
    if bias is not None:
        bias = _bf16(bias, "bias").reshape(-1)
        if bias.numel() != N:
            raise X2VError("gemm_mxfp8: bias must have N elements")
    if epilogue != EPI_NONE:
        if epilogue == EPI_RESIDUAL:
            out2 = _row2d(resid if out is None else out, "out")
            r2 = _row2d(_bf16(resid, "resid"), "resid")
            gate = _vec(gate, "gemm_mxfp8 gate", N)
        else:
            out2 = torch.empty((M, N), dtype=torch.bfloat16, device=a.device) if out is None else _row2d(out, "out")
            r2 = None
        init()
        _check(
            _lib.x2v_gemm_mxfp8_epi(_p(a), a.stride(0), _p(sa), _p(b), b.stride(0), _p(sb), _p(bias), _p(alpha), _p(out2), out2.stride(0), M, N, K, epilogue, _p(r2),
                                    0 if r2 is None else r2.stride(0), _p(gate), _stream()),
            "gemm_mxfp8_epi",
        )
        return out2
    out2 = torch.empty((M, N), dtype=torch.bfloat16, device=a.device) if out is None else _row2d(out, "out")
    init()
    _check(
        _lib.x2v_gemm_mxfp8_variant(_p(a), a.stride(0), _p(sa), _p(b), b.stride(0), _p(sb), _p(bias), _p(alpha), _p(out2), out2.stride(0), M, N, K, variant, _stream()),
        "gemm_mxfp8",
    )
    return out2


def gemm_fp8(xq, sx, wq_nk, sw, bias=None, epilogue=EPI_NONE, resid=None, gate=None, out=None, variant=0):
    M, K = xq.shape
    N = wq_nk.shape[0]
    if xq.dtype != torch.float8_e4m3fn or wq_nk.dtype != torch.float8_e4m3fn:
        raise X2VError("gemm_fp8: operands must be float8_e4m3fn (OCP; gfx950)")
    if epilogue == EPI_RESIDUAL:
        out2 = _row2d(resid if out is None else out, "out")
        r2 = _row2d(_bf16(resid, "resid"), "resid")
        gate = _vec(gate, "gemm_fp8 gate", N)
    else:
        out2 = torch.empty((M, N), dtype=torch.bfloat16, device=xq.device) if out is None else _row2d(out, "out")
        r2 = None
    if wq_nk.shape[1] != K or not xq.is_cuda or not wq_nk.is_cuda or xq.stride(1) != 1 or wq_nk.stride(1) != 1:
        raise X2VError(f"gemm_fp8: xq {tuple(xq.shape)} / wq {tuple(wq_nk.shape)} must be device matrices with unit inner stride and equal K")
    sw, sx = _vec(sw, "gemm_fp8 weight scales", N, torch.float32), _vec(sx, "gemm_fp8 activation scales", M, torch.float32)
    bias = _vec(bias, "gemm_fp8 bias", N)
    init()
    if M == 0:
        return out2
    _check(
        _lib.x2v_gemm_fp8_variant(_p(xq), xq.stride(0), _p(sx), _p(wq_nk), wq_nk.stride(0), _p(sw), _p(bias), _p(out2), out2.stride(0), M, N, K, epilogue, _p(r2), 0 if r2 is None else r2.stride(0), _p(gate), variant, _stream()),
        "gemm_fp8",
    )
    return out2


def gemm_fp8_blocked(xq, sx, wq_nk, sw, bias=None, epilogue=EPI_NONE, out=None, resid=None, gate=None):
    """x2v_gemm_fp8_blocked: the w8a8 GEMM on block-strided operands (see gemm_blocked).  `xq`: e4m3 codes [M, K] or K-blocked [B, M, K/B]
    (the codes of the ROW's quantisation, `sx` [M] stays per row); `out`: None / [M, N] or N-blocked [B', M, N/B'] (preallocated); the residual
    epilogue needs a row-major `resid` (= the output).  Returns the output tensor."""
    if wq_nk.dtype != torch.float8_e4m3fn or xq.dtype != torch.float8_e4m3fn or wq_nk.dim() != 2 or wq_nk.stride(1) != 1 or not wq_nk.is_cuda or not xq.is_cuda:
        raise X2VError("gemm_fp8_blocked: operands must be float8_e4m3fn device tensors (weight [N, K] with unit inner stride)")
    N, K = wq_nk.shape
    if xq.dim() == 3:
        kb, kbs, ldx = _blocks3d(xq, "xq")
        M = xq.shape[1]
        if kb * xq.shape[0] != K:
            raise X2VError(f"gemm_fp8_blocked: xq blocks {tuple(xq.shape)} do not make K={K}")
    else:
        if xq.dim() != 2 or xq.stride(1) != 1 or xq.shape[1] != K:
            raise X2VError(f"gemm_fp8_blocked: xq {tuple(xq.shape)} vs weight [{N},{K}]")
        kb, kbs, ldx, M = 0, 0, xq.stride(0), xq.shape[0]
    r2 = None
    if epilogue == EPI_RESIDUAL:
        if resid is None:
            raise X2VError("gemm_fp8_blocked: residual epilogue needs resid")
        r2 = _row2d(_bf16(resid, "resid"), "resid")
        out = resid if out is None else out
        gate = _vec(gate, "gemm_fp8_blocked gate", N)
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=xq.device)
    if out.dim() == 3:
        nb, nbs, ldy = _blocks3d(_bf16(out, "out"), "out")
        if nb * out.shape[0] != N or out.shape[1] != M:
            raise X2VError(f"gemm_fp8_blocked: out blocks {tuple(out.shape)} do not make [{M}, {N}]")
    else:
        o2 = _row2d(_bf16(out, "out"), "out")
        nb, nbs, ldy = 0, 0, o2.stride(0)
        if tuple(o2.shape) != (M, N):
            raise X2VError(f"gemm_fp8_blocked: out is {tuple(o2.shape)}, expected {(M, N)}")
    sw, sx = _vec(sw, "gemm_fp8_blocked weight scales", N, torch.float32), _vec(sx, "gemm_fp8_blocked activation scales", M, torch.float32)
    bias = _vec(bias, "gemm_fp8_blocked bias", N)
    init()
    if M == 0:
        return out
    _check(
        _lib.x2v_gemm_fp8_blocked(_p(xq), ldx, kb, kbs, _p(sx), _p(wq_nk), wq_nk.stride(0), _p(sw), _p(bias), _p(out), ldy, nb, nbs, M, N, K, epilogue, _p(r2),
                                  0 if r2 is None else r2.stride(0), _p(gate) if epilogue == EPI_RESIDUAL else None, _stream()),
        "gemm_fp8_blocked",
    )
    return out


def sinusoid_embed(t, dim):
    t = t.reshape(-1).to(torch.int64)
    out = torch.empty((t.numel(), dim), dtype=torch.bfloat16, device=t.device)
    init()
    _check(_lib.x2v_sinusoid_embed_bf16(_p(t), _p(out), t.numel(), dim, _stream()), "sinusoid_embed")
    return out


def causal_conv3d(x, weight, bias=None, cache=None):
    """x [T,H,W,Cin] fp32 channels-last, weight [Cout,kt,kh,kw,Cin], cache [nc,H,W,Cin] or None."""
    if x.dtype != torch.float32 or not x.is_contiguous() or not weight.is_contiguous():
        raise X2VError("causal_conv3d: x and weight must be contiguous float32")
    T, H, W, Cin = x.shape
    Cout, kt, kh, kw, _ = weight.shape
    nc = 0 if cache is None else cache.shape[0]
    out = torch.empty((T, H, W, Cout), dtype=torch.float32, device=x.device)
    init()
    _check(_lib.x2v_causal_conv3d_f32(_p(x), _p(cache), nc, _p(weight), _p(bias), _p(out), T, H, W, Cin, Cout, kt, kh, kw, _stream()), "causal_conv3d")
    return out


VCONV_CLAMP, VCONV_TSPLIT = 1, 2
VCONV_PER_TAP, VCONV_HALO64, VCONV_ZERO_TAIL32 = 4, 8, 16  # x2v_vae_conv_f16 only: kernel choice for A/B runs; the last 32 channels of Cin are zero padding


def _f32c(t, name):
    if t is not None and (t.dtype != torch.float32 or not t.is_cuda):
        raise X2VError(f"{name}: expected a float32 device tensor")
    return t


def _f32dense(t, name, dim=None, numel=None):
    """None passes through; otherwise a contiguous float32 device tensor (of rank `dim` / `numel` elements when given)."""
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
        raise X2VError(f"{name}: expected a contiguous float32 device tensor, got {t.dtype} {tuple(t.shape)} strides {t.stride()} on {t.device}")
    if dim is not None and t.dim() != dim:
        raise X2VError(f"{name}: expected rank {dim}, got shape {tuple(t.shape)}")
    if numel is not None and t.numel() != numel:
        raise X2VError(f"{name}: expected {numel} elements, got shape {tuple(t.shape)}")
    return t


def _dev_view(t, name, dtypes):
    if t.dtype not in dtypes or not t.is_cuda:
        raise X2VError(f"{name}: expected a device tensor of {' / '.join(str(d) for d in dtypes)}, got {t.dtype} on {t.device}")
    return t


def vae_conv(xp, strides, weight, out, T, H, W, bias=None, resid=None, flags=0, w_row_stride=None, cin=None):
    """Implicit-GEMM convolution over a zero-bordered buffer (x2v_vae_conv_f32).  `xp`: tensor VIEW whose first element is
    what tap (0,0,0) of output pixel (0,0,0) reads; strides = (frame, row, pixel) in floats; weight [Cout,kt,kh,kw,Cin]
    (or [Cout, K] with kt=kh=kw=1); out [T,H,W,Cout] (preallocated, contiguous)."""
    _f32c(xp, "vae_conv x"), _f32c(weight, "vae_conv weight"), _f32c(out, "vae_conv out")
    if weight.dim() == 5:
        Cout, kt, kh, kw, Cin = weight.shape
    else:
        Cout, kt, kh, kw = weight.shape[0], 1, 1, 1
        Cin = weight.shape[1] if cin is None else cin
    wrs = weight.stride(0) if w_row_stride is None else w_row_stride
    fs, rs, ps = strides
    init()
    _check(_lib.x2v_vae_conv_f32(_p(xp), fs, rs, ps, _p(weight), wrs, _p(bias), _p(resid), _p(out), T, H, W, Cin, Cout, kt, kh, kw, flags, _stream()), "vae_conv")
    return out


_VCONV16_FORCE = {"": 0, "halo64": VCONV_HALO64, "pertap": VCONV_PER_TAP}[os.environ.get("X2V_VAE_CONV16", "")]  # A/B runs: force one of the older 3x3 kernels


def vae_conv16_cached_ok(W, weight, flags=0):
    """Whether vae_conv16(..., cache=...) is available for this weight [Cout,kt,kh,kw,Cin] and image width (x2v_vae_conv_f16_cached_ok)."""
    Cout, kt, kh, kw, Cin = weight.shape
    init()
    return kt > 1 and _lib.x2v_vae_conv_f16_cached_ok(W, Cin, Cout, kh, kw, flags | _VCONV16_FORCE) == 1


def vae_conv16(xp, strides, weight, out, T, H, W, bias=None, resid=None, flags=0, cache=None):
    """x2v_vae_conv_f16: fp16 operand buffer `xp` (strides in halves) and fp16 weight [Cout,kt,kh,kw,Cin], fp32 bias / resid / out.
    cache: the kt - 1 leading input frames in a tensor of their own ([kt-1, H+2, W+2, Cin], xp's layout; x2v_vae_conv_f16_cached) — xp's own leading frames are not read."""
    if xp.dtype != torch.float16 or weight.dtype != torch.float16 or not xp.is_cuda or not weight.is_contiguous():
        raise X2VError("vae_conv16: operand buffer and weight must be CUDA float16 (weight contiguous)")
    _f32c(out, "vae_conv16 out")
    Cout, kt, kh, kw, Cin = weight.shape
    fs, rs, ps = strides
    init()
    flags |= _VCONV16_FORCE
    if cache is not None:
        if cache.dtype != torch.float16 or not cache.is_cuda or not cache.is_contiguous() or cache.shape[0] != kt - 1 or cache[0].numel() != fs:
            raise X2VError("vae_conv16: cache must be a contiguous CUDA float16 tensor of kt - 1 frames in xp's layout")
        _check(_lib.x2v_vae_conv_f16_cached(_p(xp), _p(cache), fs, rs, ps, _p(weight), weight.stride(0), _p(bias), _p(resid), _p(out), T, H, W, Cin, Cout, kt, kh, kw, flags, _stream()), "vae_conv16 (cached)")
        return out
    _check(_lib.x2v_vae_conv_f16(_p(xp), fs, rs, ps, _p(weight), weight.stride(0), _p(bias), _p(resid), _p(out), T, H, W, Cin, Cout, kt, kh, kw, flags, _stream()), "vae_conv16")
    return out


def vae_prep(x, y_view, y_strides, gamma=None, a=None, b=None, silu=False, upsample=False, split=False):
    """x [T,H,W,C] contiguous fp32 -> y_view (first element = destination of pixel (0,0,0)); y_strides = (frame, row) in floats.
    split (fp16 y_view only): write the hi/lo split [hi | hi * 2^-12 | lo] (3*C channels per pixel, x2v_vae_prep_split_f16; weights [hi | lo * 2^12 | hi])."""
    T, H, W, C = _f32dense(x, "vae_prep x", 4).shape
    for nm, t in (("gamma", gamma), ("a", a), ("b", b)):
        _f32dense(t, f"vae_prep {nm}", numel=C)
    _dev_view(y_view, "vae_prep y", (torch.float32, torch.float16))
    init()
    if y_view.dtype == torch.float16:  # operand buffer of the 16-bit convolution: channel axis possibly padded (pixel stride from the view)
        fn = _lib.x2v_vae_prep_split_f16 if split else _lib.x2v_vae_prep_f16
        _check(fn(_p(x), _p(y_view), T, H, W, C, _p(gamma), _p(a), _p(b), int(silu), int(upsample), y_strides[0], y_strides[1], y_view.stride(2), _stream()), "vae_prep_f16")
        return
    if split:
        raise X2VError("vae_prep: split needs an fp16 destination")
    _check(_lib.x2v_vae_prep_f32(_p(x), _p(y_view), T, H, W, C, _p(gamma), _p(a), _p(b), int(silu), int(upsample), y_strides[0], y_strides[1], _stream()), "vae_prep")


def softmax_rows_(s, scale):
    _f32c(s, "softmax_rows")
    M, N = s.shape
    init()
    _check(_lib.x2v_softmax_rows_f32(_p(s), s.stride(0), M, N, float(scale), _stream()), "softmax_rows")
    return s


def headnorm_rope_(q, k, wq, wk, cos, sin, num_heads, l_rope, eps=1e-6, round_mode=ROUND_FP32, q_out_scale=1.0):
    """In place on q, k [L, H*128] views (unit inner stride, any token stride): per-head RMSNorm + real RoPE on the
    first l_rope tokens (x2v_headnorm_rope_bf16)."""
    _row2d(_bf16(q, "headnorm_rope q"), "headnorm_rope q"), _row2d(_bf16(k, "headnorm_rope k"), "headnorm_rope k")
    L = q.shape[0]
    if q.shape != k.shape or q.shape[1] != num_heads * 128:
        raise X2VError(f"headnorm_rope: q {tuple(q.shape)} and k {tuple(k.shape)} must both be [L, {num_heads}*128]")
    if not 0 <= l_rope <= L:
        raise X2VError(f"headnorm_rope: l_rope={l_rope} outside [0, {L}]")
    wq, wk = _vec(wq, "headnorm_rope wq", 128), _vec(wk, "headnorm_rope wk", 128)
    cos, sin = _vec(cos, "headnorm_rope cos"), _vec(sin, "headnorm_rope sin")
    if l_rope and (cos is None or sin is None or cos.numel() < l_rope * 128 or sin.numel() < l_rope * 128):
        raise X2VError(f"headnorm_rope: cos/sin must be bf16 tables of at least [{l_rope}, 128]")
    init()
    _check(_lib.x2v_headnorm_rope_bf16(_p(q), q.stride(0), _p(k), k.stride(0), _p(wq), _p(wk), _p(cos), _p(sin), L, num_heads, l_rope, eps, round_mode, q_out_scale, _stream()),
           "headnorm_rope")


def headnorm_rope_blocked_(q, k, wq, wk, cos, sin, num_heads, l_rope, eps=1e-6, round_mode=ROUND_FP32, q_out_scale=1.0):
    """headnorm_rope_ in place on head-blocked q, k [N, L, (H/N)*128] (contiguous: the send buffers of the Ulysses exchange)."""
    for t, nm in ((q, "q"), (k, "k")):
        if t.dim() != 3 or t.dtype != torch.bfloat16 or not t.is_cuda or not t.is_contiguous():
            raise X2VError(f"headnorm_rope_blocked {nm}: expected a contiguous bf16 device tensor [N, L, (H/N)*128], got {t.dtype} {tuple(t.shape)}")
    nb, L, bc = q.shape
    if q.shape != k.shape or bc % 128 or nb * bc != num_heads * 128:
        raise X2VError(f"headnorm_rope_blocked: q {tuple(q.shape)} / k {tuple(k.shape)} do not hold {num_heads} heads of 128 in equal blocks")
    if not 0 <= l_rope <= L:
        raise X2VError(f"headnorm_rope_blocked: l_rope={l_rope} outside [0, {L}]")
    wq, wk = _vec(wq, "headnorm_rope wq", 128), _vec(wk, "headnorm_rope wk", 128)
    cos, sin = _vec(cos, "headnorm_rope cos"), _vec(sin, "headnorm_rope sin")
    if l_rope and (cos is None or sin is None or cos.numel() < l_rope * 128 or sin.numel() < l_rope * 128):
        raise X2VError(f"headnorm_rope_blocked: cos/sin must be bf16 tables of at least [{l_rope}, 128]")
    init()
    _check(_lib.x2v_headnorm_rope_blocked_bf16(_p(q), _p(k), bc, bc // 128, L * bc, _p(wq), _p(wk), _p(cos), _p(sin), L, num_heads, l_rope, eps, round_mode, q_out_scale, _stream()),
           "headnorm_rope_blocked")


def vae_prep_ex(x, y_view, y_strides, mul=None, add=None, silu=False, clamp01=False, up_hw=False, up_t=False):
    """fp32 x [T,H,W,C] → y_view (fp32, or fp16 for the 16-bit convolution's operand buffer; strides in elements of y)."""
    T, H, W, C = _f32dense(x, "vae_prep_ex x", 4).shape
    _f32dense(mul, "vae_prep_ex mul", numel=C), _f32dense(add, "vae_prep_ex add", numel=C)
    _dev_view(y_view, "vae_prep_ex y", (torch.float32, torch.float16))
    init()
    if y_view.dtype == torch.float16:
        _check(_lib.x2v_vae_prep_ex_f16(_p(x), _p(y_view), T, H, W, C, _p(mul), _p(add), int(silu), int(clamp01), int(up_hw), int(up_t), y_strides[0], y_strides[1], _stream()),
               "vae_prep_ex_f16")
        return
    _check(_lib.x2v_vae_prep_ex_f32(_p(x), _p(y_view), T, H, W, C, _p(mul), _p(add), int(silu), int(clamp01), int(up_hw), int(up_t), y_strides[0], y_strides[1], _stream()),
           "vae_prep_ex")


def vae_replicate_border_(buf, lead, pad):
    frames, hp, wp, c = buf.shape
    if buf.dtype == torch.float16:  # the kernel moves whole pixels: a pixel of C halves is C/2 floats
        c //= 2
    init()
    _check(_lib.x2v_vae_replicate_border_f32(_p(buf), frames, lead, hp, wp, c, pad, _stream()), "vae_replicate_border")


def groupnorm_affine(x, groups, gamma, beta, eps=1e-6):
    """x [..., C] contiguous fp32 → (mul[C], add[C]) such that GroupNorm(x)[..., c] = x[..., c]*mul[c] + add[c]."""
    c = _f32dense(x, "groupnorm_affine x").shape[-1]
    if groups <= 0 or c % groups:
        raise X2VError(f"groupnorm_affine: {c} channels do not split into {groups} groups")
    _f32dense(gamma, "groupnorm_affine gamma", numel=c), _f32dense(beta, "groupnorm_affine beta", numel=c)
    npix = x.numel() // c
    ws = torch.empty(2 * groups, dtype=torch.float64, device=x.device)
    mul, add = torch.empty(c, dtype=torch.float32, device=x.device), torch.empty(c, dtype=torch.float32, device=x.device)
    init()
    _check(_lib.x2v_groupnorm_affine_f32(_p(x), npix, c, groups, _p(gamma), _p(beta), eps, _p(ws), _p(mul), _p(add), _stream()), "groupnorm_affine")
    return mul, add


def softmax_rows_causal_(s, scale, hw, n_keys=None):
    _dev_view(s, "softmax_rows_causal", (torch.float32,))
    if s.dim() != 2 or s.stride(1) != 1:
        raise X2VError(f"softmax_rows_causal: expected a 2-D fp32 tensor with unit inner stride, got {tuple(s.shape)} strides {s.stride()}")
    M, N = s.shape
    if hw <= 0 or not 0 <= (N if n_keys is None else n_keys) <= N:
        raise X2VError(f"softmax_rows_causal: hw={hw}, n_keys={n_keys} for {N} columns")
    init()
    _check(_lib.x2v_softmax_rows_causal_f32(_p(s), s.stride(0), M, N, float(scale), hw, N if n_keys is None else n_keys, _stream()), "softmax_rows_causal")
    return s


def blend_axis_(a, b, axis, extent):
    """b[.., idx, ..] = a[.., na-extent+idx, ..]*(1-idx/extent) + b*(idx/extent) along `axis` of two contiguous fp32 tensors
    that agree in every other dimension."""
    _f32dense(a, "blend_axis a"), _f32dense(b, "blend_axis b")
    if a.dim() != b.dim():
        raise X2VError(f"blend_axis: ranks differ: {tuple(a.shape)} vs {tuple(b.shape)}")
    axis = axis % a.dim()
    if any(x_ != y_ for i_, (x_, y_) in enumerate(zip(a.shape, b.shape)) if i_ != axis):
        raise X2VError(f"blend_axis: {tuple(a.shape)} and {tuple(b.shape)} must agree outside axis {axis}")
    outer = 1
    for d in a.shape[:axis]:
        outer *= d
    inner = 1
    for d in a.shape[axis + 1 :]:
        inner *= d
    na, nb = a.shape[axis], b.shape[axis]
    init()
    _check(_lib.x2v_blend_axis_f32(_p(a), _p(b), outer, na, nb, inner, na * inner, nb * inner, min(extent, na, nb), _stream()), "blend_axis")
    return b


def _f32flat(t, name, n):
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous() or t.numel() != n:
        raise X2VError(f"{name}: expected a contiguous float32 device tensor of {n} elements, got {t.dtype} {tuple(t.shape)} on {t.device}")
    return t


def _latents(t, name, n):
    if t.dtype not in (torch.float32, torch.bfloat16) or not t.is_cuda or not t.is_contiguous() or t.numel() != n:
        raise X2VError(f"{name}: expected a contiguous float32 / bfloat16 device tensor of {n} elements")
    return t


def unipc_step(cond, uncond, latents, last_sample, m0, m1, coef, order_c, order_p, want_noise_pred=False):
    """x2v_unipc_step_f32: (noise_pred | None, x0, corrected sample, next latents) — CFG combine + WanScheduler.step_post in one launch.
    `coef`: the 12 fp32 coefficients of include/x2v.h as Python floats."""
    n = cond.numel()
    _f32flat(cond, "cond", n), _f32flat(uncond, "uncond", n), _latents(latents, "latents", n)
    _f32flat(last_sample, "last_sample", n), _f32flat(m0, "m0", n), _f32flat(m1, "m1", n)
    if len(coef) != 12:
        raise X2VError("unipc_step: coef must hold 12 values")
    new = lambda: torch.empty(cond.shape, dtype=torch.float32, device=cond.device)  # noqa: E731
    noise_pred = new() if want_noise_pred else None
    x0, sample, lat = new(), new(), new()
    init()
    carr = (_f32 * 12)(*[float(c) for c in coef])
    _check(
        _lib.x2v_unipc_step_f32(_p(cond), _p(uncond), _p(latents), int(latents.dtype == torch.bfloat16), _p(last_sample), _p(m0), _p(m1), _p(noise_pred), _p(x0), _p(sample),
                                _p(lat), carr, order_c, order_p, n, _stream()),
        "unipc_step",
    )
    return noise_pred, x0, sample, lat


def distill_step(cond, uncond, guide, latents, noise, sigma, one_minus_next, sigma_next, want_noise_pred=False):
    """x2v_distill_step_f32: (noise_pred | None, next latents in the dtype of `latents`)."""
    n = cond.numel()
    _f32flat(cond, "cond", n), _f32flat(uncond, "uncond", n), _latents(latents, "latents", n), _f32flat(noise, "noise", n)
    noise_pred = torch.empty(cond.shape, dtype=torch.float32, device=cond.device) if want_noise_pred else None
    out = torch.empty(cond.shape, dtype=latents.dtype, device=cond.device)
    init()
    _check(
        _lib.x2v_distill_step_f32(_p(cond), _p(uncond), float(guide), _p(latents), int(latents.dtype == torch.bfloat16), _p(noise), float(sigma), float(one_minus_next),
                                  float(sigma_next), _p(noise_pred), _p(out), n, _stream()),
        "distill_step",
    )
    return noise_pred, out
