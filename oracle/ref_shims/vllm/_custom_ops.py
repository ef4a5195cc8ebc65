"""TEST INFRASTRUCTURE ONLY.  The two vLLM kernels the reference's w8a8-fp8 class calls, RESTATED from vLLM's published behaviour
(csrc/quantization/fp8/common.cu `dynamic_per_token_scaled_fp8_quant`, csrc/quantization/cutlass_w8a8 `cutlass_scaled_mm`):

  scaled_fp8_quant(x, None, scale_ub=None, use_per_token_if_dynamic=True)      (mm_weight.py:236-238)
      scale[m] = max(absmax(x[m, :]) / 448, 1 / (448 * 512))   fp32 [M, 1]
      q[m, k]  = e4m3fn_rne(clamp(float(x[m, k]) / scale[m], -448, 448))
  torch.ops._C.cutlass_scaled_mm(out, a, b, a_scales, b_scales, bias)          (mm_weight.py:310-318)
      out = ((float(a) @ float(b)) * a_scales * b_scales^T + bias)  -> out.dtype, a [M, K] e4m3, b [K, N] e4m3 (the .t() view of [N, K])

Fixtures generated through these stubs pin the reference's own glue — weight auto-quantisation (FloatQuantizer per channel,
utils/quant_utils.py:41-53,155-161), scale/bias handling, transposes, output allocation — and leave the two kernels' arithmetic marked
as restated (the reference holds no golden vectors for them)."""
import torch

_FP8_MAX = 448.0


def scaled_fp8_quant(input, scale=None, num_token_padding=None, scale_ub=None, use_per_token_if_dynamic=False):
    if scale is not None or not use_per_token_if_dynamic or scale_ub is not None:
        raise NotImplementedError("stub covers the reference's only call form: dynamic per-token scales")
    xf = input.float()
    s = torch.clamp(xf.abs().amax(dim=1, keepdim=True) / _FP8_MAX, min=1.0 / (_FP8_MAX * 512.0))
    q = torch.clamp(xf / s, -_FP8_MAX, _FP8_MAX).to(torch.float8_e4m3fn)
    return q, s


_lib = torch.library.Library("_C", "FRAGMENT")
_lib.define("cutlass_scaled_mm(Tensor(a!) out, Tensor a, Tensor b, Tensor a_scales, Tensor b_scales, Tensor? bias) -> ()")


def _cutlass_scaled_mm(out, a, b, a_scales, b_scales, bias):
    y = (a.float() @ b.float()) * a_scales.float() * b_scales.float().reshape(1, -1)
    if bias is not None:
        y = y + bias.float()
    out.copy_(y.to(out.dtype))


_lib.impl("cutlass_scaled_mm", _cutlass_scaled_mm, "CompositeExplicitAutograd")
