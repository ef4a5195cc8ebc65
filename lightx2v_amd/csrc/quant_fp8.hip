// Per-token dynamic e4m3fn quantisation of activations (w8a8 path, BASELINE config #4).
// HBM-bound: reads M*K bf16, writes M*K bytes + M floats.  One workgroup per row, row held in registers.
#include <type_traits>

#include "x2v_common.h"

namespace x2v {

template <int CH>
__global__ __launch_bounds__(256) void quant_fp8_rowwise_kernel(const unsigned short* __restrict__ x, int64_t ldx, unsigned char* __restrict__ xq,
                                                                int64_t ldq, float* __restrict__ scale, int K, int kblock, int64_t kblock_stride) {
  __shared__ float red[4];
  const int t = threadIdx.x;
  const int64_t row = blockIdx.x;
  const unsigned short* xr = x + row * ldx;
  float v[CH][8];
  bool ok[CH];
  float amax = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int e = (c * 256 + t) * 8;
    ok[c] = e < K;
    if (ok[c]) {
      // K-blocked x (kblock > 0; the Ulysses head->seq receive buffer [N, S/N, (H/N) d]): element e of the row sits in block e / kblock
      const int64_t off = kblock > 0 ? (int64_t)(e / kblock) * kblock_stride + e % kblock : e;
      unpack8(*reinterpret_cast<const uint4*>(xr + off), v[c]);
#pragma unroll
      for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[c][j]));
    }
  }
  amax = block_max<4>(amax, red);
  // reference: mm_weight.py:236-245 → vllm dynamic per-token quant: scale = max(amax / 448, 1 / (448 * 512))
  const float s = fmaxf(amax / 448.0f, 1.0f / (448.0f * 512.0f));
  if (t == 0) scale[row] = s;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    if (!ok[c]) continue;
    const int e = (c * 256 + t) * 8;
    float q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = fminf(fmaxf(v[c][j] / s, -448.f), 448.f);
    unsigned lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(q[4], q[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(q[6], q[7], hi, true);
    uint2 o;
    o.x = lo;
    o.y = hi;
    *reinterpret_cast<uint2*>(xq + row * ldq + e) = o;
  }
}

}  // namespace x2v

using namespace x2v;

extern "C" __attribute__((visibility("default"))) int x2v_quant_fp8_rowwise_blocked(const void* x, int64_t ldx, int x_kblock, int64_t x_kblock_stride, void* xq, int64_t ldq,
                                                                                   float* scale, int64_t M, int K, void* stream) {
  X2V_REQUIRE(x && xq && scale, X2V_E_ARG, "quant_fp8: null pointer");
  X2V_REQUIRE(K > 0 && K % 8 == 0 && K <= 16384, X2V_E_SHAPE, "quant_fp8: K=%d must be a multiple of 8 and <= 16384", K);
  X2V_REQUIRE(ldx % 8 == 0 && ldq % 8 == 0 && aligned16(x) && ((uintptr_t)xq % 8) == 0, X2V_E_ALIGN, "quant_fp8: row alignment");
  X2V_REQUIRE(x_kblock == 0 || (x_kblock > 0 && x_kblock % 8 == 0 && K % x_kblock == 0 && x_kblock_stride % 8 == 0 && ldx >= x_kblock), X2V_E_SHAPE,
              "quant_fp8: x K-block of %d elements must be a multiple of 8 dividing K=%d, block stride a multiple of 8", x_kblock, K);
  if (M <= 0) return X2V_OK;
  const int kblock = x_kblock;
  const int64_t kblock_stride = x_kblock_stride;
  const int ch = (K / 8 + 255) / 256;
  hipStream_t st = (hipStream_t)stream;
  auto xs = (const unsigned short*)x;
  auto qs = (unsigned char*)xq;
  switch (ch) {
    case 1: hipLaunchKernelGGL((quant_fp8_rowwise_kernel<1>), dim3((unsigned)M), dim3(256), 0, st, xs, ldx, qs, ldq, scale, K, kblock, kblock_stride); break;
    case 2: hipLaunchKernelGGL((quant_fp8_rowwise_kernel<2>), dim3((unsigned)M), dim3(256), 0, st, xs, ldx, qs, ldq, scale, K, kblock, kblock_stride); break;
    case 3: hipLaunchKernelGGL((quant_fp8_rowwise_kernel<3>), dim3((unsigned)M), dim3(256), 0, st, xs, ldx, qs, ldq, scale, K, kblock, kblock_stride); break;
    case 4: hipLaunchKernelGGL((quant_fp8_rowwise_kernel<4>), dim3((unsigned)M), dim3(256), 0, st, xs, ldx, qs, ldq, scale, K, kblock, kblock_stride); break;
    default: hipLaunchKernelGGL((quant_fp8_rowwise_kernel<8>), dim3((unsigned)M), dim3(256), 0, st, xs, ldx, qs, ldq, scale, K, kblock, kblock_stride); break;
  }
  X2V_LAUNCH_CHECK("quant_fp8 launch");
  return X2V_OK;
}

extern "C" __attribute__((visibility("default"))) int x2v_quant_fp8_rowwise(const void* x, int64_t ldx, void* xq, int64_t ldq, float* scale, int64_t M, int K, void* stream) {
  return x2v_quant_fp8_rowwise_blocked(x, ldx, 0, 0, xq, ldq, scale, M, K, stream);
}
