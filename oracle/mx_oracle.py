"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's MXFP8 quantiser and block-scaled GEMM (never imported by the product).

reference: lightx2v_kernel/csrc/gemm/mxfp8_quant_kernels_sm120.cu:139-196 (cvt_warp_fp16_to_fp8)
    vecMax = max |x| over 32 consecutive K elements (taken in the input type, bf16);  SFValue = vecMax / 448.0f;
    scale byte = __nv_cvt_float_to_e8m0(SFValue, __NV_SATFINITE, cudaRoundPosInf): the smallest power of two >= SFValue, 2^(byte-127);
    outputScale = 1 / 2^(byte-127) (rcp.approx of a power of two is exact);  q = cvt.rn.satfinite.e4m3(float(x) * outputScale).
  lightx2v_kernel/csrc/gemm/mxfp8_scaled_mm_kernels_sm120.cu:60-66,150-160: D = alpha * (A . B^T) + bias per column, bf16 out, fp32 accumulate
  (OCP MX: each product term is (a * 2^sa) * (b * 2^sb) over its 32-wide block).

PARITY UNPINNED against a run of the reference: its kernels are sm120 CUDA (CUTLASS) and cannot execute here, and the reference's own
tests hold no golden vectors for them — only the acceptance bound `error(mm_pred, linear(a, w, bias)) < 1e-2`
(lightx2v_kernel/test/mxfp8_mxfp8/test_mxfp8_quant.py:37), which tests/test_gpu_mx.py applies to the HIP path at the same shapes.
One documented divergence: for an all-zero block the CUDA code multiplies 0 by rcp.approx.ftz(2^-127) = inf and stores NaN bytes; here
(and in the HIP kernel) such a block is zero elements with scale byte 0.
"""
import numpy as np
import torch


def e8m0_ceil(sf: np.ndarray) -> np.ndarray:
    """fp32 array (>= 0) → biased exponent byte of the smallest power of two >= sf, saturating at 254; 0 → 0."""
    bits = sf.astype(np.float32).view(np.uint32)
    ex = (bits >> 23).astype(np.int64)
    man = (bits & 0x7FFFFF).astype(np.int64)
    byte = np.where(ex == 0, (man > 0x400000).astype(np.int64), ex + (man != 0))
    return np.minimum(byte, 254).astype(np.uint8)


def quant_mxfp8(x: torch.Tensor):
    """x bf16 [M, K] → (q float8_e4m3fn [M, K], scale bytes uint8 [M, K/32])."""
    M, K = x.shape
    xb = x.to(torch.float32).reshape(M, K // 32, 32)
    vmax = xb.abs().amax(dim=-1)
    sf = (vmax / torch.tensor(448.0, dtype=torch.float32)).numpy()
    byte = e8m0_ceil(sf)
    inv = np.ldexp(np.float64(1.0), 127 - byte.astype(np.int64))  # exact power of two
    scaled = (xb.double() * torch.from_numpy(inv).unsqueeze(-1)).float()  # exact: power-of-two scaling of a bf16 value
    q = scaled.reshape(M, K).to(torch.float8_e4m3fn)
    return q, torch.from_numpy(byte)


def dequant(q: torch.Tensor, sc: torch.Tensor) -> torch.Tensor:
    M, K = q.shape
    mul = torch.from_numpy(np.ldexp(np.float64(1.0), sc.numpy().astype(np.int64) - 127))
    return (q.float().double().reshape(M, K // 32, 32) * mul.unsqueeze(-1)).reshape(M, K)


def gemm_mxfp8(a, sa, b, sb, alpha=1.0, bias=None) -> torch.Tensor:
    """float64 evaluation of alpha * deq(a) @ deq(b)^T + bias, rounded once to bf16."""
    y = float(alpha) * (dequant(a, sa) @ dequant(b, sb).T)
    if bias is not None:
        y = y + bias.double().reshape(1, -1)
    return y.to(torch.bfloat16)


def snr_error(pred: torch.Tensor, real: torch.Tensor) -> float:
    """lightx2v_kernel/python/lightx2v_kernel/utils.py:44-72 `error`: sum (pred-real)^2 / (sum real^2 + 1e-7) over all elements."""
    p, r = pred.flatten().float(), real.flatten().float()
    return (torch.pow(p - r, 2).sum() / (torch.pow(r, 2).sum() + 1e-7)).item()
