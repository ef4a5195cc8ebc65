// Causal Conv3d (Wan VAE decoder) as an implicit GEMM on the fp32-input MFMA.
//
// Out[pixel][cout] = sum over (tap, cin) X[pixel + tap offset][cin] * W[cout][tap][cin], fp32 throughout:
// v_mfma_f32_32x32x2_f32 is an exact fp32 fma chain at the fp32 vector rate (157 TFLOP/s peak), so results
// keep the reference's fp32 numerics (vae.py decodes in fp32) while the reduction runs on the matrix pipe.
//
// v1 structure: workgroup = 64 pixels (consecutive in the h*W+w order of one frame) x 64 output channels,
// 4 waves x one 32x32 MFMA tile; per (tap, 16-channel chunk) the shifted input patch [64][16] and the weight
// slab [64][16] are staged in LDS (pitch 17 floats: conflict-free fragment reads), zero-filled where the
// tap falls outside the frame / before the first cached frame.  Algorithmic work: 2*T*H*W*Cout*Cin*taps FLOP.
#include "x2v_common.h"

namespace x2v {

constexpr int CV_PIX = 64, CV_CO = 64, CV_KC = 16, CV_LD = 17;

__global__ __launch_bounds__(256) void causal_conv3d_kernel(const float* __restrict__ x, const float* __restrict__ cache, int nc,
                                                            const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y, int T,
                                                            int Hh, int Ww, int Cin, int Cout, int kt, int kh, int kw) {
  __shared__ float xs[CV_PIX * CV_LD];
  __shared__ float ws[CV_CO * CV_LD];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fl = lane & 31, fh = lane >> 5;
  const int wr = wid >> 1, wc = wid & 1;
  const int HW = Hh * Ww;
  const int tiles_per_frame = (HW + CV_PIX - 1) / CV_PIX;
  const int frame = blockIdx.x / tiles_per_frame;
  const int p0 = (blockIdx.x % tiles_per_frame) * CV_PIX;
  const int co0 = blockIdx.y * CV_CO;

  // staging role: row = tid/4 (pixel or cout), 4 consecutive channels at (tid%4)*4
  const int srow = tid >> 2, sc4 = (tid & 3) * 4;
  const int pix = p0 + srow;
  const int ph = pix / Ww, pw = pix % Ww;
  const bool pix_ok = pix < HW;
  const int co_s = co0 + srow;
  const bool co_ok = co_s < Cout;
  const int taps = kt * kh * kw;
  const int64_t frame_elems = (int64_t)HW * Cin;

  f32x16_t acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;

  for (int tap = 0; tap < taps; ++tap) {
    const int dt = tap / (kh * kw), dh = (tap / kw) % kh, dw = tap % kw;
    const int tt = frame + dt - (kt - 1);  // source frame relative to x; negative -> cache / zero pad
    const int hh = ph + dh - kh / 2, ww = pw + dw - kw / 2;
    const float* src = nullptr;
    if (pix_ok && hh >= 0 && hh < Hh && ww >= 0 && ww < Ww) {
      if (tt >= 0)
        src = x + (int64_t)tt * frame_elems + ((int64_t)hh * Ww + ww) * Cin;
      else if (nc + tt >= 0)
        src = cache + (int64_t)(nc + tt) * frame_elems + ((int64_t)hh * Ww + ww) * Cin;
    }
    const float* wsrc = co_ok ? w + ((int64_t)co_s * taps + tap) * Cin : nullptr;
    for (int c0 = 0; c0 < Cin; c0 += CV_KC) {
      float4 xv = make_float4(0.f, 0.f, 0.f, 0.f), wv = make_float4(0.f, 0.f, 0.f, 0.f);
      const int c = c0 + sc4;
      if (c < Cin) {
        if (src != nullptr) xv = *reinterpret_cast<const float4*>(src + c);
        if (wsrc != nullptr) wv = *reinterpret_cast<const float4*>(wsrc + c);
      }
      __syncthreads();  // previous chunk's fragment reads are done
      float* xd = xs + srow * CV_LD + sc4;
      float* wd = ws + srow * CV_LD + sc4;
      xd[0] = xv.x; xd[1] = xv.y; xd[2] = xv.z; xd[3] = xv.w;
      wd[0] = wv.x; wd[1] = wv.y; wd[2] = wv.z; wd[3] = wv.w;
      __syncthreads();
      const float* xa = xs + (wr * 32 + fl) * CV_LD + fh;
      const float* wb = ws + (wc * 32 + fl) * CV_LD + fh;
#pragma unroll
      for (int kk = 0; kk < CV_KC / 2; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[2 * kk], wb[2 * kk], acc, 0, 0, 0);
    }
  }
  // D[i = pixel (A rows)][j = cout (B cols)]: lane column j = fl, rows i = (r&3) + 8*(r>>2) + 4*fh
  const int co = co0 + wc * 32 + fl;
  if (co < Cout) {
    const float bv = bias != nullptr ? bias[co] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = p0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
      if (p < HW) y[((int64_t)frame * HW + p) * Cout + co] = acc[r] + bv;
    }
  }
}

}  // namespace x2v

using namespace x2v;

extern "C" __attribute__((visibility("default"))) int x2v_causal_conv3d_f32(const float* x, const float* cache, int cache_frames, const float* w, const float* bias, float* y, int T, int Hh,
                                     int Ww, int Cin, int Cout, int kt, int kh, int kw, void* stream) {
  X2V_REQUIRE(x && w && y, X2V_E_ARG, "conv3d: null pointer");
  X2V_REQUIRE(T > 0 && Hh > 0 && Ww > 0 && Cin > 0 && Cout > 0, X2V_E_SHAPE, "conv3d: bad shape");
  X2V_REQUIRE(kt >= 1 && kt <= 3 && (kh == 1 || kh == 3) && (kw == 1 || kw == 3), X2V_E_SHAPE, "conv3d: kernel %dx%dx%d unsupported", kt, kh, kw);
  X2V_REQUIRE(Cin % 4 == 0, X2V_E_SHAPE, "conv3d: Cin=%d must be a multiple of 4", Cin);
  X2V_REQUIRE(cache_frames >= 0 && cache_frames <= kt - 1, X2V_E_SHAPE, "conv3d: cache_frames=%d must be in [0, kt-1]", cache_frames);
  X2V_REQUIRE(cache_frames == 0 || cache != nullptr, X2V_E_ARG, "conv3d: cache pointer missing");
  X2V_REQUIRE(aligned16(x) && aligned16(w) && aligned16(cache), X2V_E_ALIGN, "conv3d: pointers must be 16-byte aligned");
  const int64_t tiles = (int64_t)T * (((int64_t)Hh * Ww + CV_PIX - 1) / CV_PIX);
  X2V_REQUIRE(tiles < (1ll << 31), X2V_E_SHAPE, "conv3d: too many tiles");
  dim3 grid((unsigned)tiles, (unsigned)((Cout + CV_CO - 1) / CV_CO));
  hipLaunchKernelGGL(causal_conv3d_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, cache, cache_frames, w, bias, y, T, Hh, Ww, Cin, Cout, kt, kh, kw);
  X2V_LAUNCH_CHECK("conv3d launch");
  return X2V_OK;
}
