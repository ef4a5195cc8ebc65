"""MXFP8 quantise + GEMM with the call signatures of the reference's `lightx2v_kernel.gemm` module
(lightx2v_kernel/python/lightx2v_kernel/gemm.py:73-83 `scaled_fp8_quant`, :93-97 `cutlass_scaled_mxfp8_mm`), on gfx950's block-scaled
MFMA.  One difference a caller can see: the scale tensor is the `[K/128, rows, 4]` e8m0 table of this GEMM (K-tile major), not the
reference's `[ceil128(rows), ceil4(K/32)]` sm120-swizzled int32 view — either way it is the consumer's format and callers pass it
straight back to the GEMM (`lib.mx_scales_rowmajor` gives the logical [rows, K/32] table).  K must be a multiple of 128.

    a_q, a_s = scaled_fp8_quant(activation)          # bf16 [m, k] -> e4m3 bytes [m, k], e8m0 [k/128, m, 4]
    w_q, w_s = scaled_fp8_quant(weight)              # [n, k]
    y = cutlass_scaled_mxfp8_mm(a_q, w_q, a_s, w_s, alpha=alpha, bias=bias)   # bf16 [m, n]
"""
import torch

from . import lib

_E8M0 = getattr(torch, "float8_e8m0fnu", None)


def scaled_fp8_quant(input: torch.Tensor):
    q, sc = lib.quant_mxfp8(input)
    return q.view(torch.uint8), (sc.view(_E8M0) if _E8M0 is not None else sc)


def cutlass_scaled_mxfp8_mm(mat_a, mat_b, scales_a, scales_b, alpha, bias=None):
    sa = scales_a.view(torch.uint8) if scales_a.dtype != torch.uint8 else scales_a
    sb = scales_b.view(torch.uint8) if scales_b.dtype != torch.uint8 else scales_b
    return lib.gemm_mxfp8(mat_a, sa, mat_b, sb, alpha=alpha, bias=None if bias is None else bias.reshape(-1))


scaled_mxfp8_mm = cutlass_scaled_mxfp8_mm  # the name without the CUDA library in it
