// Row-wise HBM-bound kernels of the DiT block: RMSNorm, LayerNorm(+adaLN modulate), fused
// RMSNorm+3-axis RoPE on q/k, gate-residual, activations, timestep sinusoid.
//
// Roofline: each is one read + one write of an [M,D] bf16 tensor (2*M*D*2 bytes) against ~6.3 TB/s
// achievable HBM.  Structure: a row lives entirely in registers between its single 16-byte-vector read
// and its single write (two-pass statistics cost no extra HBM traffic); NW waves cooperate on one row
// (NW=4 for the model dim, NW=1 for short rows so a 256-thread block carries 4 rows).
#include <algorithm>
#include <type_traits>

#include "x2v_common.h"

namespace x2v {

// ------------------------------------------------------------------------------------------------
// Row holder: CH 16-byte chunks per lane, lanes of the row's NW waves interleaved chunk-wise so every
// wave-instruction reads NW... 64 consecutive chunks (1 KiB) — fully coalesced.
template <int CH, int NW>
struct RowRegs {
  float v[CH][8];
  bool ok[CH];
  __device__ __forceinline__ void load(const unsigned short* row, int D, int t) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int e = (c * NW * 64 + t) * 8;
      ok[c] = e < D;
      if (ok[c]) {
        uint4 u = *reinterpret_cast<const uint4*>(row + e);
        unpack8(u, v[c]);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[c][j] = 0.f;
      }
    }
  }
};

template <int CH, int NW, int ROUND>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const unsigned short* __restrict__ x, int64_t ldx, const unsigned short* __restrict__ w,
                                                      unsigned short* __restrict__ y, int64_t ldy, int64_t M, int D, float eps) {
  __shared__ float red[4];
  constexpr int ROWS = 4 / NW;
  const int t = threadIdx.x % (NW * 64);
  const int64_t row = (int64_t)blockIdx.x * ROWS + threadIdx.x / (NW * 64);
  const bool live = row < M;
  RowRegs<CH, NW> r;
  if (live) r.load(x + row * ldx, D, t);
  float ss = 0.f;
  if (live) {
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float p = r.v[c][j] * r.v[c][j];
        ss += (ROUND == X2V_ROUND_REF) ? rbf(p) : p;  // torch: x.pow(2) is a bf16 tensor
      }
  }
  ss = (NW == 1) ? wave_sum(ss) : block_sum<4>(ss, red);
  if (!live) return;
  float rs;
  if (ROUND == X2V_ROUND_REF) {
    float mean = rbf(ss / (float)D);  // .mean(-1): fp32 accumulate, bf16 result
    float tt = rbf(mean + eps);       // + eps   → bf16
    rs = rbf(1.0f / sqrtf(tt));       // rsqrt   → bf16
  } else {
    rs = 1.0f / sqrtf(ss / (float)D + eps);
  }
  unsigned short* yr = y + row * ldy;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    if (!r.ok[c]) continue;
    const int e = (c * NW * 64 + t) * 8;
    float wv[8], o[8];
    unpack8(*reinterpret_cast<const uint4*>(w + e), wv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (ROUND == X2V_ROUND_REF)
        o[j] = rbf(r.v[c][j] * rs) * wv[j];  // (x * rstd) → bf16, then * weight → bf16 (pack8 rounds)
      else
        o[j] = r.v[c][j] * rs * wv[j];
    }
    *reinterpret_cast<uint4*>(yr + e) = pack8(o);
  }
}

// LayerNorm (+ optional affine, + optional adaLN modulate), reference rounding chain.
template <int CH, int NW>
__global__ __launch_bounds__(256) void layernorm_kernel(const unsigned short* __restrict__ x, int64_t ldx, const unsigned short* __restrict__ w,
                                                        const unsigned short* __restrict__ b, const unsigned short* __restrict__ scale,
                                                        const unsigned short* __restrict__ shift, unsigned short* __restrict__ y, int64_t ldy,
                                                        int64_t M, int D, float eps) {
  __shared__ float red[4];
  constexpr int ROWS = 4 / NW;
  const int t = threadIdx.x % (NW * 64);
  const int64_t row = (int64_t)blockIdx.x * ROWS + threadIdx.x / (NW * 64);
  const bool live = row < M;
  RowRegs<CH, NW> r;
  if (live) r.load(x + row * ldx, D, t);
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += r.v[c][j];
  s = (NW == 1) ? wave_sum(s) : block_sum<4>(s, red);
  const float mean = s / (float)D;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c)
    if (r.ok[c]) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float d = r.v[c][j] - mean;
        q += d * d;
      }
    }
  q = (NW == 1) ? wave_sum(q) : block_sum<4>(q, red);
  if (!live) return;
  const float rstd = 1.0f / sqrtf(q / (float)D + eps);
  unsigned short* yr = y + row * ldy;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    if (!r.ok[c]) continue;
    const int e = (c * NW * 64 + t) * 8;
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (r.v[c][j] - mean) * rstd;
    if (w != nullptr) {
      float wv[8];
      unpack8(*reinterpret_cast<const uint4*>(w + e), wv);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] *= wv[j];
    }
    if (b != nullptr) {
      float bv[8];
      unpack8(*reinterpret_cast<const uint4*>(b + e), bv);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += bv[j];
    }
    if (scale != nullptr) {  // norm_out.mul_(1 + scale).add_(shift): three bf16 roundings
      float sc[8], sh[8];
      unpack8(*reinterpret_cast<const uint4*>(scale + e), sc);
      unpack8(*reinterpret_cast<const uint4*>(shift + e), sh);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float ln = rbf(o[j]);
        float m = rbf(ln * rbf(1.0f + sc[j]));
        o[j] = m + sh[j];
      }
    }
    *reinterpret_cast<uint4*>(yr + e) = pack8(o);
  }
}

// LayerNorm (+affine, +adaLN modulate) fused with the per-token dynamic e4m3 quantisation of its output (w8a8 path): what the reference
// runs as LNWeight.apply + mul_/add_ + scaled_fp8_quant in front of the q / k / v (or ffn_0) projections (mm_weight.py:236-245) — the
// bf16 activation never goes to HBM, and the one quantised copy serves every projection that consumes it.  The normalised values are
// rounded to bf16 exactly where layernorm_kernel rounds them (same expression order), so codes and scales are bit-identical to
// x2v_layernorm_bf16 followed by x2v_quant_fp8_rowwise.  One block per row.
template <int CH>
__global__ __launch_bounds__(256) void layernorm_fp8_kernel(const unsigned short* __restrict__ x, int64_t ldx, const unsigned short* __restrict__ w,
                                                            const unsigned short* __restrict__ b, const unsigned short* __restrict__ scale,
                                                            const unsigned short* __restrict__ shift, unsigned char* __restrict__ xq, int64_t ldq,
                                                            float* __restrict__ sx, int D, float eps) {
  __shared__ float red[4];
  const int t = threadIdx.x;
  const int64_t row = blockIdx.x;
  RowRegs<CH, 4> r;
  r.load(x + row * ldx, D, t);
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += r.v[c][j];
  s = block_sum<4>(s, red);
  const float mean = s / (float)D;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c)
    if (r.ok[c]) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float d = r.v[c][j] - mean;
        q += d * d;
      }
    }
  q = block_sum<4>(q, red);
  const float rstd = 1.0f / sqrtf(q / (float)D + eps);
  float amax = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    if (!r.ok[c]) continue;
    const int e = (c * 256 + t) * 8;
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (r.v[c][j] - mean) * rstd;
    if (w != nullptr) {
      float wv[8];
      unpack8(*reinterpret_cast<const uint4*>(w + e), wv);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] *= wv[j];
    }
    if (b != nullptr) {
      float bv[8];
      unpack8(*reinterpret_cast<const uint4*>(b + e), bv);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += bv[j];
    }
    if (scale != nullptr) {
      float sc[8], sh[8];
      unpack8(*reinterpret_cast<const uint4*>(scale + e), sc);
      unpack8(*reinterpret_cast<const uint4*>(shift + e), sh);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float ln = rbf(o[j]);
        float m = rbf(ln * rbf(1.0f + sc[j]));
        o[j] = m + sh[j];
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      r.v[c][j] = rbf(o[j]);  // the bf16 tensor the reference quantises
      amax = fmaxf(amax, fabsf(r.v[c][j]));
    }
  }
  amax = block_max<4>(amax, red);
  const float qs = fmaxf(amax / 448.0f, 1.0f / (448.0f * 512.0f));  // quant_fp8_rowwise_kernel's scale rule
  if (t == 0) sx[row] = qs;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    if (!r.ok[c]) continue;
    const int e = (c * 256 + t) * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fminf(fmaxf(r.v[c][j] / qs, -448.f), 448.f);
    unsigned lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
    *reinterpret_cast<uint2*>(xq + row * ldq + e) = make_uint2(lo, hi);
  }
}

// Where the q|k norm+RoPE kernels write a row: in place (qo == nullptr), or out of place into N-blocked buffers — column e of token
// `row` at (e / cbw) * cbs + row * ldo + e % cbw: the [N_ranks][S/N][(H/N) d] send buffer of the Ulysses seq->head exchange
// (x2v_rmsnorm_rope_blocked_bf16), so no transposing copy stands between this kernel and the all-to-all.
struct RopeOut {
  unsigned short* qo = nullptr;
  unsigned short* ko = nullptr;
  int64_t ldo = 0, cbs = 0;
  int cbw = 0;
  __device__ __forceinline__ unsigned short* dst(int which, int64_t row, int e, unsigned short* inplace_row) const {
    if (qo == nullptr) return inplace_row + e;
    return (which == 0 ? qo : ko) + (int64_t)(e / cbw) * cbs + row * ldo + e % cbw;
  }
};

// One complex rotation (a + i b) * (co + i si), then the optional output scale — written with explicit fused multiply-adds so every
// kernel that rotates (per-row and streaming forms) rounds identically whatever the optimiser would contract on its own.
__device__ __forceinline__ void rope_pair(float a, float bb, float co, float si, float oscale, float& o0, float& o1) {
  o0 = __builtin_fmaf(a, co, -(bb * si)) * oscale;
  o1 = __builtin_fmaf(a, si, bb * co) * oscale;
}

// Fused q/k RMSNorm over the full model dim + 3-axis RoPE.  blockIdx.y selects q (0) or k (1).
// One block per token row; D = H*128 so every 16-byte chunk holds 4 (re,im) pairs of one head.
template <int CH, int ROUND>
__global__ __launch_bounds__(256) void rmsnorm_rope_kernel(unsigned short* __restrict__ q, int64_t ldq, unsigned short* __restrict__ k, int64_t ldk,
                                                           const unsigned short* __restrict__ wq, const unsigned short* __restrict__ wk,
                                                           const float2* __restrict__ cs, int64_t S, int D, int64_t s0, int gf, int gh, int gw,
                                                           float eps, float q_out_scale, RopeOut ro) {
  __shared__ float red[4];
  const int t = threadIdx.x;
  const int64_t row = blockIdx.x;
  const float oscale = blockIdx.y == 0 ? q_out_scale : 1.f;  // folded into q's single final rounding (attention prescale)
  unsigned short* base = (blockIdx.y == 0 ? q + row * ldq : k + row * ldk);
  const unsigned short* w = blockIdx.y == 0 ? wq : wk;
  RowRegs<CH, 4> r;
  r.load(base, D, t);
  float rs = 1.f;
  if (w != nullptr) {
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float p = r.v[c][j] * r.v[c][j];
        ss += (ROUND == X2V_ROUND_REF) ? rbf(p) : p;
      }
    ss = block_sum<4>(ss, red);
    if (ROUND == X2V_ROUND_REF) {
      float mean = rbf(ss / (float)D);
      rs = rbf(1.0f / sqrtf(rbf(mean + eps)));
    } else {
      rs = 1.0f / sqrtf(ss / (float)D + eps);
    }
  }
  // grid position of this token (global index s0+row); beyond the grid → identity rotation
  const int64_t g = s0 + row;
  const bool rot = g < (int64_t)gf * gh * gw;
  const int pw = (int)(g % gw), ph = (int)((g / gw) % gh), pf = (int)(g / ((int64_t)gw * gh));
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    if (!r.ok[c]) continue;
    const int e = (c * 256 + t) * 8;
    float xn[8], o[8];
    if (w != nullptr) {
      float wv[8];
      unpack8(*reinterpret_cast<const uint4*>(w + e), wv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (ROUND == X2V_ROUND_REF)
          xn[j] = rbf(rbf(r.v[c][j] * rs) * wv[j]);
        else
          xn[j] = rbf(r.v[c][j] * rs * wv[j]);  // the norm's bf16 output feeds RoPE in the reference
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) xn[j] = r.v[c][j];
    }
    const int pair0 = (e & 127) >> 1;  // complex index within the head: 0..63, 4 pairs per chunk
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int ci = pair0 + p;
      float co = 1.f, si = 0.f;
      if (rot) {
        const int pos = ci < 22 ? pf : (ci < 43 ? ph : pw);
        const float2 f = cs[pos * 64 + ci];
        co = f.x;
        si = f.y;
      }
      rope_pair(xn[2 * p], xn[2 * p + 1], co, si, oscale, o[2 * p], o[2 * p + 1]);
    }
    *reinterpret_cast<uint4*>(ro.dst(blockIdx.y, row, e, base)) = pack8(o);
  }
}

// ------------------------------------------------------------------------------------------------
// Streaming forms of the two kernels above for long inputs (M >> #CUs; the DiT's [S, D] activations).
// One block per row pays the whole load -> reduce -> reduce -> store chain once per 20 KB and re-fetches / re-unpacks the
// per-channel operands for every row; with ~11 VALU operations per element (three bf16 rounding points) these kernels sit at
// 62-70 % of the achievable bandwidth, issue-bound as much as latency-bound.  Here a block is persistent (grid = what fits the
// chip at once), walks rows blockIdx.x, +gridDim.x, ..., issues the NEXT row's 16-byte loads before the current row's
// reductions, and keeps the per-channel operands (affine / modulation rows with 1 + scale already rounded, norm weights) in
// registers and the token's rotation factors in LDS, fetched once per block / per token instead of once per row.  (A deeper
// prefetch — two rows ahead — measured slower: more registers, fewer resident blocks, and the hoisted operand math gone.)  The arithmetic (order of every sum and rounding) is exactly that of the kernels above: the two forms
// give bit-identical outputs, which tools/x2v_check and tests/test_gpu_edge.py assert.
template <int CH>
__device__ __forceinline__ void load_row_raw(uint4 (&dst)[CH], const unsigned short* row, int D, int t) {
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int e = (c * 256 + t) * 8;
    if (e < D) dst[c] = *reinterpret_cast<const uint4*>(row + e);
    else dst[c] = make_uint4(0u, 0u, 0u, 0u);
  }
}
template <int CH, bool AFF, bool MOD>
__global__ __launch_bounds__(256) void layernorm_stream_kernel(const unsigned short* __restrict__ x, int64_t ldx, const unsigned short* __restrict__ w,
                                                               const unsigned short* __restrict__ b, const unsigned short* __restrict__ scale,
                                                               const unsigned short* __restrict__ shift, unsigned short* __restrict__ y, int64_t ldy,
                                                               int64_t M, int D, float eps) {
  __shared__ float red[4];
  const int t = threadIdx.x;
  // per-channel operands, resident for every row of this block (absent w -> 1.0, absent b -> 0: unused placeholders); the optimiser
  // hoists their unpacking and the rounding of 1 + scale out of the row loop, which is where most of the gain over the per-row form
  // comes from: at 6 TB/s these kernels are as much VALU-issue bound (~11 operations per element) as they are latency bound
  uint4 wv[AFF ? CH : 1], bv[AFF ? CH : 1], scv[MOD ? CH : 1], shv[MOD ? CH : 1];
  bool ok[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int e = (c * 256 + t) * 8;
    ok[c] = e < D;
    if constexpr (AFF) {
      wv[c] = (ok[c] && w != nullptr) ? *reinterpret_cast<const uint4*>(w + e) : make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
      bv[c] = (ok[c] && b != nullptr) ? *reinterpret_cast<const uint4*>(b + e) : make_uint4(0u, 0u, 0u, 0u);
    }
    if constexpr (MOD) {
      scv[c] = ok[c] ? *reinterpret_cast<const uint4*>(scale + e) : make_uint4(0u, 0u, 0u, 0u);
      shv[c] = ok[c] ? *reinterpret_cast<const uint4*>(shift + e) : make_uint4(0u, 0u, 0u, 0u);
    }
  }
  const int64_t stride = gridDim.x;
  int64_t row = blockIdx.x;
  uint4 cur[CH], nxt[CH];
  if (row < M) load_row_raw<CH>(cur, x + row * ldx, D, t);
  for (; row < M; row += stride) {
    if (row + stride < M) load_row_raw<CH>(nxt, x + (row + stride) * ldx, D, t);  // in flight across this row's reductions
    float v[CH][8];
#pragma unroll
    for (int c = 0; c < CH; ++c) unpack8(cur[c], v[c]);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[c][j];
    s = block_sum<4>(s, red);
    const float mean = s / (float)D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (ok[c]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float d = v[c][j] - mean;
          q += d * d;
        }
      }
    q = block_sum<4>(q, red);
    const float rstd = 1.0f / sqrtf(q / (float)D + eps);
    unsigned short* yr = y + row * ldy;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (!ok[c]) continue;
      const int e = (c * 256 + t) * 8;
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (v[c][j] - mean) * rstd;
      if constexpr (AFF) {
        float wf[8], bf[8];
        unpack8(wv[c], wf);
        unpack8(bv[c], bf);
        if (w != nullptr) {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] *= wf[j];
        }
        if (b != nullptr) {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] += bf[j];
        }
      }
      if constexpr (MOD) {  // norm_out.mul_(1 + scale).add_(shift): three bf16 roundings
        float sc[8], sh[8];
        unpack8(scv[c], sc);
        unpack8(shv[c], sh);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float ln = rbf(o[j]);
          float m = rbf(ln * rbf(1.0f + sc[j]));
          o[j] = m + sh[j];
        }
      }
      *reinterpret_cast<uint4*>(yr + e) = pack8(o);
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) cur[c] = nxt[c];
  }
}

// q and k rows of one token are handled back to back by the same block, so the token's 64 rotation factors are gathered once
// (64 lanes, into LDS) instead of 4 scalar 8-byte gathers per 16-byte chunk for q and again for k.
template <int CH, int ROUND>
__global__ __launch_bounds__(256) void rmsnorm_rope_stream_kernel(unsigned short* __restrict__ q, int64_t ldq, unsigned short* __restrict__ k, int64_t ldk,
                                                                  const unsigned short* __restrict__ wq, const unsigned short* __restrict__ wk,
                                                                  const float2* __restrict__ cs, int64_t S, int D, int64_t s0, int gf, int gh, int gw,
                                                                  float eps, float q_out_scale, RopeOut ro) {
  __shared__ float red[4];
  __shared__ __attribute__((aligned(16))) float2 tab[2][64];  // by row parity: staging row n+1 never races readers of row n
  const int t = threadIdx.x;
  const bool has_w = wq != nullptr;
  uint4 wqv[CH], wkv[CH];
  bool ok[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int e = (c * 256 + t) * 8;
    ok[c] = e < D;
    wqv[c] = (ok[c] && has_w) ? *reinterpret_cast<const uint4*>(wq + e) : make_uint4(0u, 0u, 0u, 0u);
    wkv[c] = (ok[c] && has_w) ? *reinterpret_cast<const uint4*>(wk + e) : make_uint4(0u, 0u, 0u, 0u);
  }
  const int64_t stride = gridDim.x;
  const int64_t ntok = (int64_t)gf * gh * gw;
  int64_t row = blockIdx.x;
  uint4 cur[CH], nxt[CH];
  if (row < S) load_row_raw<CH>(cur, q + row * ldq, D, t);
  int par = 0;
  for (; row < S; row += stride, par ^= 1) {
    if (t < 64) {  // this token's rotation factors; beyond the grid -> identity rotation
      const int64_t g = s0 + row;
      float2 f = make_float2(1.f, 0.f);
      if (g < ntok) {
        const int pw = (int)(g % gw), ph = (int)((g / gw) % gh), pf = (int)(g / ((int64_t)gw * gh));
        const int pos = t < 22 ? pf : (t < 43 ? ph : pw);
        f = cs[pos * 64 + t];
      }
      tab[par][t] = f;
    }
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      unsigned short* base = which == 0 ? q + row * ldq : k + row * ldk;
      // the next item (this token's k row, then the next token's q row) is in flight across this item's reduction
      if (which == 0) load_row_raw<CH>(nxt, k + row * ldk, D, t);
      else if (row + stride < S) load_row_raw<CH>(nxt, q + (row + stride) * ldq, D, t);
      const float oscale = which == 0 ? q_out_scale : 1.f;
      float v[CH][8];
#pragma unroll
      for (int c = 0; c < CH; ++c) unpack8(cur[c], v[c]);
      float rs = 1.f;
      if (has_w) {
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float p = v[c][j] * v[c][j];
            ss += (ROUND == X2V_ROUND_REF) ? rbf(p) : p;
          }
        ss = block_sum<4>(ss, red);  // its barriers also publish tab[par]
        if (ROUND == X2V_ROUND_REF) {
          float mean = rbf(ss / (float)D);
          rs = rbf(1.0f / sqrtf(rbf(mean + eps)));
        } else {
          rs = 1.0f / sqrtf(ss / (float)D + eps);
        }
      } else if (which == 0) {
        __syncthreads();  // publish tab[par]
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (!ok[c]) continue;
        const int e = (c * 256 + t) * 8;
        float xn[8], o[8];
        if (has_w) {
          float wf[8];
          unpack8(which == 0 ? wqv[c] : wkv[c], wf);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (ROUND == X2V_ROUND_REF)
              xn[j] = rbf(rbf(v[c][j] * rs) * wf[j]);
            else
              xn[j] = rbf(v[c][j] * rs * wf[j]);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) xn[j] = v[c][j];
        }
        const int pair0 = (e & 127) >> 1;
        const float4 f01 = *reinterpret_cast<const float4*>(&tab[par][pair0]);
        const float4 f23 = *reinterpret_cast<const float4*>(&tab[par][pair0 + 2]);
        const float co[4] = {f01.x, f01.z, f23.x, f23.z}, si[4] = {f01.y, f01.w, f23.y, f23.w};
#pragma unroll
        for (int p = 0; p < 4; ++p) rope_pair(xn[2 * p], xn[2 * p + 1], co[p], si[p], oscale, o[2 * p], o[2 * p + 1]);
        *reinterpret_cast<uint4*>(ro.dst(which, row, e, base)) = pack8(o);
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) cur[c] = nxt[c];
    }
  }
}

__global__ __launch_bounds__(256) void gate_residual_kernel(unsigned short* __restrict__ x, int64_t ldx, const unsigned short* __restrict__ y,
                                                            int64_t ldy, const unsigned short* __restrict__ gate, int64_t M, int D) {
  const int cpr = D / 8;
  const int64_t total = M * (int64_t)cpr;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / cpr;
    const int e = (int)(i % cpr) * 8;
    float xv[8], yv[8], o[8];
    unpack8(*reinterpret_cast<const uint4*>(x + row * ldx + e), xv);
    unpack8(*reinterpret_cast<const uint4*>(y + row * ldy + e), yv);
    if (gate != nullptr) {
      float gv[8];
      unpack8(*reinterpret_cast<const uint4*>(gate + e), gv);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = xv[j] + rbf(yv[j] * gv[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = xv[j] + yv[j];
    }
    *reinterpret_cast<uint4*>(x + row * ldx + e) = pack8(o);
  }
}

template <int ACT>
__device__ __forceinline__ float act_f(float x) {
  if constexpr (ACT == X2V_EPI_GELU_TANH) return gelu_tanh_f(x);
  else if constexpr (ACT == X2V_ACT_GELU_ERF) return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f));  // torch gelu(approximate="none") on a bf16 tensor: fp32 math
  else return silu_f(x);
}

template <int ACT>
__global__ __launch_bounds__(256) void activation_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, int64_t n) {
  const int64_t nv = n / 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    float v[8];
    unpack8(reinterpret_cast<const uint4*>(x)[i], v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = act_f<ACT>(v[j]);
    reinterpret_cast<uint4*>(y)[i] = pack8(v);
  }
  // tail (n % 8) handled by the first block
  if (blockIdx.x == 0) {
    for (int64_t i = nv * 8 + threadIdx.x; i < n; i += blockDim.x) {
      float f = bf2f(x[i]);
      y[i] = f2bf(act_f<ACT>(f));
    }
  }
}

__global__ void sinusoid_kernel(const int64_t* __restrict__ t, unsigned short* __restrict__ y, int n, int dim) {
  const int half = dim / 2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n * half; i += gridDim.x * blockDim.x) {
    const int r = i / half, j = i % half;
    // float64 as the reference (utils.py:165-171): pos * 10000^(-j/half)
    const double f = pow(10000.0, -((double)j / (double)half));
    const double a = (double)t[r] * f;
    y[(int64_t)r * dim + j] = f2bf((float)cos(a));
    y[(int64_t)r * dim + half + j] = f2bf((float)sin(a));
  }
}

// ------------------------------------------------------------------------------------------------
// HunyuanVideo q/k path: per-head RMSNorm over d = 128 followed (for the first `l_rope` tokens = the image
// tokens) by the real-valued RoPE  x*cos + rotate_half(x)*sin  with bf16 cos/sin tables [l_rope, 128] —
// replaces RMSWeightSgl.apply on [L,H,128] (rms_norm_weight.py:102-113) + utils_bf16.apply_rotary_emb
// (hunyuan/infer/utils_bf16.py:5-31), in place on the q and k column blocks of the fused QKV GEMM output.
// 16 lanes x 16 bytes per (token, head) row, 16 rows per 256-thread block; blockIdx.y selects q / k.
// Rounding: X2V_ROUND_REF reproduces the reference's bf16 chain (pow, mean, +eps, rsqrt, *rstd, *w; then the
// three roundings of the rotary form); X2V_ROUND_FP32 keeps fp32 through the norm (sgl_kernel.rmsnorm) and
// rounds once before the (always bf16-chained) rotary step.
template <int ROUND>
__global__ __launch_bounds__(256) void headnorm_rope_kernel(unsigned short* __restrict__ q, int64_t ldq, unsigned short* __restrict__ k, int64_t ldk,
                                                            const unsigned short* __restrict__ wq, const unsigned short* __restrict__ wk,
                                                            const unsigned short* __restrict__ cosb, const unsigned short* __restrict__ sinb, int64_t L,
                                                            int H, int64_t l_rope, float eps, float q_out_scale, int hpb, int64_t cbs) {
  const int sub = threadIdx.x & 15;
  const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (row >= L * H) return;  // whole 16-lane groups leave together; the shuffles below stay inside a group
  const int64_t tok = row / H;
  const int head = (int)(row - tok * H);
  // head-blocked operand (hpb < H): heads [j hpb, (j+1) hpb) of every token form the matrix [L][hpb*128] (token stride ld) that starts
  // j * cbs elements into the buffer — the [N_ranks][S/N][(H/N) d] send buffer of the Ulysses exchange
  const int hb = head / hpb;
  unsigned short* base = (blockIdx.y == 0 ? q + tok * ldq : k + tok * ldk) + hb * cbs;
  const unsigned short* w = blockIdx.y == 0 ? wq : wk;
  unsigned short* p = base + (head - hb * hpb) * 128 + sub * 8;
  const float oscale = (blockIdx.y == 0 && ROUND != X2V_ROUND_REF) ? q_out_scale : 1.f;
  float v[8], wv[8];
  unpack8(*reinterpret_cast<const uint4*>(p), v);
  if (w != nullptr) {
    unpack8(*reinterpret_cast<const uint4*>(w + sub * 8), wv);
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float pw = v[j] * v[j];
      ss += (ROUND == X2V_ROUND_REF) ? rbf(pw) : pw;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    float rs;
    if (ROUND == X2V_ROUND_REF) {
      const float mean = rbf(ss / 128.f);
      rs = rbf(1.0f / sqrtf(rbf(mean + eps)));
    } else {
      rs = 1.0f / sqrtf(ss / 128.f + eps);
    }
    // q_out_scale (attention prescale, FP32 mode only): folded into the value's last rounding — for un-rotated (text)
    // rows that is the norm's rounding, for rotated rows the rotary sum's
    const bool scale_here = ROUND != X2V_ROUND_REF && tok >= l_rope;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      v[j] = (ROUND == X2V_ROUND_REF) ? rbf(rbf(v[j] * rs) * wv[j]) : (scale_here ? v[j] * rs * wv[j] * oscale : rbf(v[j] * rs * wv[j]));
  }
  if (tok < l_rope) {
    float c[8], sn[8];
    unpack8(*reinterpret_cast<const uint4*>(cosb + tok * 128 + sub * 8), c);
    unpack8(*reinterpret_cast<const uint4*>(sinb + tok * 128 + sub * 8), sn);
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      const float re = v[j], im = v[j + 1];
      v[j] = (rbf(re * c[j]) + rbf(-im * sn[j])) * oscale;
      v[j + 1] = (rbf(im * c[j + 1]) + rbf(re * sn[j + 1])) * oscale;
    }
  }
  *reinterpret_cast<uint4*>(p) = pack8(v);
}

}  // namespace x2v

using namespace x2v;

namespace {
int chunks_for(int D, int nw) { return (D / 8 + nw * 64 - 1) / (nw * 64); }
}  // namespace

// Dispatch a runtime chunk count to a compile-time CH (generic lambda receives std::integral_constant).
template <typename F>
static int dispatch_ch(int ch, int D, F&& f) {
  switch (ch) {
    case 1: f(std::integral_constant<int, 1>{}); return X2V_OK;
    case 2: f(std::integral_constant<int, 2>{}); return X2V_OK;
    case 3: f(std::integral_constant<int, 3>{}); return X2V_OK;
    case 4: f(std::integral_constant<int, 4>{}); return X2V_OK;
    case 5: case 6: case 7: case 8: f(std::integral_constant<int, 8>{}); return X2V_OK;
    default: ::x2v::set_error("row too long: D=%d (max 16384)", D); return X2V_E_SHAPE;
  }
}

// Blocks of a 256-thread kernel that are resident on the whole chip at once (occupancy x CUs), queried once per kernel: the grid of
// the persistent "stream" kernels.  0 = query failed (callers fall back to the one-block-per-row form).
template <auto KERNEL>
static int resident_blocks() {
  static int cached = 0;
  if (cached == 0) {
    int dev = 0, nb = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)KERNEL, 256, 0) != hipSuccess || nb <= 0)
      return 0;
    cached = nb * prop.multiProcessorCount;
  }
  return cached;
}

extern "C" __attribute__((visibility("default"))) int x2v_rmsnorm_bf16(const void* x, int64_t ldx, const void* w, void* y, int64_t ldy, int64_t M, int D, float eps, int round_mode,
                                void* stream) {
  X2V_REQUIRE(x && w && y, X2V_E_ARG, "rmsnorm: null pointer");
  X2V_REQUIRE(D > 0 && D % 8 == 0 && D <= 16384, X2V_E_SHAPE, "rmsnorm: D=%d must be a multiple of 8 and <= 16384", D);
  X2V_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && aligned16(x) && aligned16(y) && aligned16(w), X2V_E_ALIGN, "rmsnorm: rows must be 16-byte aligned");
  X2V_REQUIRE(round_mode == X2V_ROUND_FP32 || round_mode == X2V_ROUND_REF, X2V_E_ARG, "rmsnorm: bad round_mode %d", round_mode);
  if (M <= 0) return X2V_OK;
  auto xs = (const unsigned short*)x, ws = (const unsigned short*)w;
  auto ys = (unsigned short*)y;
  hipStream_t st = (hipStream_t)stream;
  if (D <= 512) {  // one wave per row, 4 rows per block
    const unsigned grid = (unsigned)((M + 3) / 4);
    if (round_mode == X2V_ROUND_REF)
      hipLaunchKernelGGL((rmsnorm_kernel<1, 1, X2V_ROUND_REF>), dim3(grid), dim3(256), 0, st, xs, ldx, ws, ys, ldy, M, D, eps);
    else
      hipLaunchKernelGGL((rmsnorm_kernel<1, 1, X2V_ROUND_FP32>), dim3(grid), dim3(256), 0, st, xs, ldx, ws, ys, ldy, M, D, eps);
  } else {
    const int ch = chunks_for(D, 4);
    const unsigned grid = (unsigned)M;
    int rc = dispatch_ch(ch, D, [&](auto chc) {
      constexpr int CH = decltype(chc)::value;
      if (round_mode == X2V_ROUND_REF)
        hipLaunchKernelGGL((rmsnorm_kernel<CH, 4, X2V_ROUND_REF>), dim3(grid), dim3(256), 0, st, xs, ldx, ws, ys, ldy, M, D, eps);
      else
        hipLaunchKernelGGL((rmsnorm_kernel<CH, 4, X2V_ROUND_FP32>), dim3(grid), dim3(256), 0, st, xs, ldx, ws, ys, ldy, M, D, eps);
    });
    if (rc != X2V_OK) return rc;
  }
  X2V_LAUNCH_CHECK("rmsnorm launch");
  return X2V_OK;
}

extern "C" __attribute__((visibility("default"))) int x2v_layernorm_bf16_variant(const void* x, int64_t ldx, const void* w, const void* b, const void* scale, const void* shift,
                                                                                 void* y, int64_t ldy, int64_t M, int D, float eps, int variant, void* stream) {
  X2V_REQUIRE(x && y, X2V_E_ARG, "layernorm: null pointer");
  X2V_REQUIRE(variant >= 0 && variant <= 2, X2V_E_ARG, "layernorm: unknown variant %d", variant);
  X2V_REQUIRE((scale == nullptr) == (shift == nullptr), X2V_E_ARG, "layernorm: scale and shift must be given together");
  X2V_REQUIRE(D > 0 && D % 8 == 0 && D <= 16384, X2V_E_SHAPE, "layernorm: D=%d must be a multiple of 8 and <= 16384", D);
  X2V_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && aligned16(x) && aligned16(y) && aligned16(w) && aligned16(b) && aligned16(scale) && aligned16(shift),
              X2V_E_ALIGN, "layernorm: rows must be 16-byte aligned");
  if (M <= 0) return X2V_OK;
  auto xs = (const unsigned short*)x;
  auto ws = (const unsigned short*)w, bs = (const unsigned short*)b, scs = (const unsigned short*)scale, shs = (const unsigned short*)shift;
  auto ys = (unsigned short*)y;
  hipStream_t st = (hipStream_t)stream;
  if (D <= 512) {
    X2V_REQUIRE(variant != 2, X2V_E_SHAPE, "layernorm: the streaming kernel covers 512 < D <= 8192 (D=%d)", D);
    const unsigned grid = (unsigned)((M + 3) / 4);
    hipLaunchKernelGGL((layernorm_kernel<1, 1>), dim3(grid), dim3(256), 0, st, xs, ldx, ws, bs, scs, shs, ys, ldy, M, D, eps);
  } else {
    const int ch = chunks_for(D, 4);
    const unsigned grid = (unsigned)M;
    const bool aff = ws != nullptr || bs != nullptr, mod = scs != nullptr;
    bool streamed = false;
    int rc = dispatch_ch(ch, D, [&](auto chc) {
      constexpr int CH = decltype(chc)::value;
      if constexpr (CH <= 4) {
        // persistent form for long inputs (bit-identical results; see layernorm_stream_kernel)
        auto go = [&](auto kern, int resident) {
          if (variant == 1 || resident <= 0 || (variant == 0 && M < 2 * (int64_t)resident)) return;
          hipLaunchKernelGGL(kern, dim3((unsigned)std::min<int64_t>(M, resident)), dim3(256), 0, st, xs, ldx, ws, bs, scs, shs, ys, ldy, M, D, eps);
          streamed = true;
        };
        if (aff && mod) go(layernorm_stream_kernel<CH, true, true>, resident_blocks<layernorm_stream_kernel<CH, true, true>>());
        else if (aff) go(layernorm_stream_kernel<CH, true, false>, resident_blocks<layernorm_stream_kernel<CH, true, false>>());
        else if (mod) go(layernorm_stream_kernel<CH, false, true>, resident_blocks<layernorm_stream_kernel<CH, false, true>>());
        else go(layernorm_stream_kernel<CH, false, false>, resident_blocks<layernorm_stream_kernel<CH, false, false>>());
      }
      if (!streamed) hipLaunchKernelGGL((layernorm_kernel<CH, 4>), dim3(grid), dim3(256), 0, st, xs, ldx, ws, bs, scs, shs, ys, ldy, M, D, eps);
    });
    if (rc != X2V_OK) return rc;
    X2V_REQUIRE(variant != 2 || streamed, X2V_E_SHAPE, "layernorm: the streaming kernel covers 512 < D <= 8192 (D=%d)", D);
  }
  X2V_LAUNCH_CHECK("layernorm launch");
  return X2V_OK;
}

extern "C" __attribute__((visibility("default"))) int x2v_layernorm_quant_fp8(const void* x, int64_t ldx, const void* w, const void* b, const void* scale, const void* shift,
                                                                              void* xq, int64_t ldq, float* sx, int64_t M, int D, float eps, void* stream) {
  X2V_REQUIRE(x && xq && sx, X2V_E_ARG, "layernorm_quant_fp8: null pointer");
  X2V_REQUIRE((scale == nullptr) == (shift == nullptr), X2V_E_ARG, "layernorm_quant_fp8: scale and shift must be given together");
  X2V_REQUIRE(D > 512 && D % 8 == 0 && D <= 16384, X2V_E_SHAPE, "layernorm_quant_fp8: D=%d must be a multiple of 8 in (512, 16384] (smaller rows: call the two kernels)", D);
  X2V_REQUIRE(ldx % 8 == 0 && ldq % 8 == 0 && aligned16(x) && ((uintptr_t)xq % 8) == 0 && aligned16(w) && aligned16(b) && aligned16(scale) && aligned16(shift), X2V_E_ALIGN,
              "layernorm_quant_fp8: row alignment");
  if (M <= 0) return X2V_OK;
  const int ch = chunks_for(D, 4);
  int rc = dispatch_ch(ch, D, [&](auto chc) {
    constexpr int CH = decltype(chc)::value;
    hipLaunchKernelGGL((layernorm_fp8_kernel<CH>), dim3((unsigned)M), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, ldx, (const unsigned short*)w,
                       (const unsigned short*)b, (const unsigned short*)scale, (const unsigned short*)shift, (unsigned char*)xq, ldq, sx, D, eps);
  });
  if (rc != X2V_OK) return rc;
  X2V_LAUNCH_CHECK("layernorm_quant_fp8 launch");
  return X2V_OK;
}

extern "C" __attribute__((visibility("default"))) int x2v_layernorm_bf16(const void* x, int64_t ldx, const void* w, const void* b, const void* scale, const void* shift, void* y,
                                  int64_t ldy, int64_t M, int D, float eps, void* stream) {
  return x2v_layernorm_bf16_variant(x, ldx, w, b, scale, shift, y, ldy, M, D, eps, 0, stream);
}

extern "C" __attribute__((visibility("default"))) int x2v_rmsnorm_rope_bf16(void* q, int64_t ldq, void* k, int64_t ldk, const void* wq, const void* wk, const void* rope_cs, int64_t S,
                                     int H, int64_t s0, int gf, int gh, int gw, float eps, int round_mode, void* stream) {
  return x2v_rmsnorm_rope_scaled_bf16(q, ldq, k, ldk, wq, wk, rope_cs, S, H, s0, gf, gh, gw, eps, round_mode, 1.0f, stream);
}

static int rmsnorm_rope_impl(void* q, int64_t ldq, void* k, int64_t ldk, const void* wq, const void* wk, const void* rope_cs, int64_t S, int H, int64_t s0, int gf, int gh,
                             int gw, float eps, int round_mode, float q_out_scale, int variant, RopeOut ro, void* stream) {
  X2V_REQUIRE(q && k && rope_cs, X2V_E_ARG, "rmsnorm_rope: null pointer");
  X2V_REQUIRE(variant >= 0 && variant <= 2, X2V_E_ARG, "rmsnorm_rope: unknown variant %d", variant);
  X2V_REQUIRE(q_out_scale > 0.f, X2V_E_ARG, "rmsnorm_rope: q_out_scale must be positive");
  X2V_REQUIRE((wq == nullptr) == (wk == nullptr), X2V_E_ARG, "rmsnorm_rope: wq and wk must be given together");
  const int D = H * 128;
  X2V_REQUIRE(H > 0 && D <= 16384, X2V_E_SHAPE, "rmsnorm_rope: H=%d out of range", H);
  X2V_REQUIRE(gf > 0 && gh > 0 && gw > 0 && gf <= 1024 && gh <= 1024 && gw <= 1024, X2V_E_SHAPE, "rmsnorm_rope: grid (%d,%d,%d) out of range", gf, gh, gw);
  X2V_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && aligned16(q) && aligned16(k) && aligned16(wq) && aligned16(wk), X2V_E_ALIGN,
              "rmsnorm_rope: rows must be 16-byte aligned");
  X2V_REQUIRE(round_mode == X2V_ROUND_FP32 || round_mode == X2V_ROUND_REF, X2V_E_ARG, "rmsnorm_rope: bad round_mode %d", round_mode);
  if (S <= 0) return X2V_OK;
  const int ch = chunks_for(D, 4);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)S, 2);
  auto qs = (unsigned short*)q, ks = (unsigned short*)k;
  auto wqs = (const unsigned short*)wq, wks = (const unsigned short*)wk;
  auto cs = (const float2*)rope_cs;
  bool streamed = false;
  int rc = dispatch_ch(ch, D, [&](auto chc) {
    constexpr int CH = decltype(chc)::value;
    if constexpr (CH <= 4) {
      // persistent form for long inputs (bit-identical results; see rmsnorm_rope_stream_kernel)
      auto go = [&](auto kern, int resident) {
        if (variant == 1 || resident <= 0 || (variant == 0 && S < 2 * (int64_t)resident)) return;
        hipLaunchKernelGGL(kern, dim3((unsigned)std::min<int64_t>(S, resident)), dim3(256), 0, st, qs, ldq, ks, ldk, wqs, wks, cs, S, D, s0, gf, gh, gw, eps, q_out_scale, ro);
        streamed = true;
      };
      if (round_mode == X2V_ROUND_REF) go(rmsnorm_rope_stream_kernel<CH, X2V_ROUND_REF>, resident_blocks<rmsnorm_rope_stream_kernel<CH, X2V_ROUND_REF>>());
      else go(rmsnorm_rope_stream_kernel<CH, X2V_ROUND_FP32>, resident_blocks<rmsnorm_rope_stream_kernel<CH, X2V_ROUND_FP32>>());
    }
    if (streamed) return;
    if (round_mode == X2V_ROUND_REF)
      hipLaunchKernelGGL((rmsnorm_rope_kernel<CH, X2V_ROUND_REF>), grid, dim3(256), 0, st, qs, ldq, ks, ldk, wqs, wks, cs, S, D, s0, gf, gh, gw, eps, q_out_scale, ro);
    else
      hipLaunchKernelGGL((rmsnorm_rope_kernel<CH, X2V_ROUND_FP32>), grid, dim3(256), 0, st, qs, ldq, ks, ldk, wqs, wks, cs, S, D, s0, gf, gh, gw, eps, q_out_scale, ro);
  });
  if (rc != X2V_OK) return rc;
  X2V_REQUIRE(variant != 2 || streamed, X2V_E_SHAPE, "rmsnorm_rope: the streaming kernel covers D <= 8192 (D=%d)", D);
  X2V_LAUNCH_CHECK("rmsnorm_rope launch");
  return X2V_OK;
}

extern "C" __attribute__((visibility("default"))) int x2v_rmsnorm_rope_scaled_bf16_variant(void* q, int64_t ldq, void* k, int64_t ldk, const void* wq, const void* wk,
                                                                                           const void* rope_cs, int64_t S, int H, int64_t s0, int gf, int gh, int gw, float eps,
                                                                                           int round_mode, float q_out_scale, int variant, void* stream) {
  return rmsnorm_rope_impl(q, ldq, k, ldk, wq, wk, rope_cs, S, H, s0, gf, gh, gw, eps, round_mode, q_out_scale, variant, RopeOut(), stream);
}

extern "C" __attribute__((visibility("default"))) int x2v_rmsnorm_rope_blocked_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* wq, const void* wk,
                                                                                    const void* rope_cs, void* q_out, void* k_out, int64_t ldo, int block_cols,
                                                                                    int64_t block_stride, int64_t S, int H, int64_t s0, int gf, int gh, int gw, float eps,
                                                                                    int round_mode, float q_out_scale, void* stream) {
  X2V_REQUIRE(q_out && k_out, X2V_E_ARG, "rmsnorm_rope_blocked: null output");
  X2V_REQUIRE(block_cols > 0 && block_cols % 8 == 0 && (H * 128) % block_cols == 0 && ldo % 8 == 0 && ldo >= block_cols && block_stride % 8 == 0 && aligned16(q_out) &&
                  aligned16(k_out),
              X2V_E_SHAPE, "rmsnorm_rope_blocked: block_cols=%d must divide H*128 and be a multiple of 8; ldo / block_stride multiples of 8", block_cols);
  RopeOut ro;
  ro.qo = (unsigned short*)q_out;
  ro.ko = (unsigned short*)k_out;
  ro.ldo = ldo;
  ro.cbw = block_cols;
  ro.cbs = block_stride;
  return rmsnorm_rope_impl((void*)q, ldq, (void*)k, ldk, wq, wk, rope_cs, S, H, s0, gf, gh, gw, eps, round_mode, q_out_scale, 0, ro, stream);
}

extern "C" __attribute__((visibility("default"))) int x2v_rmsnorm_rope_scaled_bf16(void* q, int64_t ldq, void* k, int64_t ldk, const void* wq, const void* wk, const void* rope_cs,
                                                                                   int64_t S, int H, int64_t s0, int gf, int gh, int gw, float eps, int round_mode,
                                                                                   float q_out_scale, void* stream) {
  return x2v_rmsnorm_rope_scaled_bf16_variant(q, ldq, k, ldk, wq, wk, rope_cs, S, H, s0, gf, gh, gw, eps, round_mode, q_out_scale, 0, stream);
}

extern "C" __attribute__((visibility("default"))) int x2v_gate_residual_bf16(void* x, int64_t ldx, const void* y, int64_t ldy, const void* gate, int64_t M, int D, void* stream) {
  X2V_REQUIRE(x && y, X2V_E_ARG, "gate_residual: null pointer");
  X2V_REQUIRE(D > 0 && D % 8 == 0, X2V_E_SHAPE, "gate_residual: D=%d must be a multiple of 8", D);
  X2V_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && aligned16(x) && aligned16(y) && aligned16(gate), X2V_E_ALIGN, "gate_residual: rows must be 16-byte aligned");
  if (M <= 0) return X2V_OK;
  const int64_t total = M * (int64_t)(D / 8);
  const unsigned grid = (unsigned)((total + 255) / 256 < 2048 * 4 ? (total + 255) / 256 : 2048 * 4);
  hipLaunchKernelGGL(gate_residual_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (unsigned short*)x, ldx, (const unsigned short*)y, ldy,
                     (const unsigned short*)gate, M, D);
  X2V_LAUNCH_CHECK("gate_residual launch");
  return X2V_OK;
}

extern "C" __attribute__((visibility("default"))) int x2v_activation_bf16(const void* x, void* y, int64_t n, int act, void* stream) {
  X2V_REQUIRE(x && y, X2V_E_ARG, "activation: null pointer");
  X2V_REQUIRE(act == X2V_EPI_GELU_TANH || act == X2V_EPI_SILU || act == X2V_ACT_GELU_ERF, X2V_E_ARG, "activation: unknown act %d", act);
  X2V_REQUIRE(aligned16(x) && aligned16(y), X2V_E_ALIGN, "activation: pointers must be 16-byte aligned");
  if (n <= 0) return X2V_OK;
  const int64_t nv = n / 8;
  const unsigned grid = (unsigned)((nv + 255) / 256 < 8192 ? ((nv + 255) / 256 > 0 ? (nv + 255) / 256 : 1) : 8192);
  if (act == X2V_EPI_GELU_TANH)
    hipLaunchKernelGGL((activation_kernel<X2V_EPI_GELU_TANH>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, (unsigned short*)y, n);
  else if (act == X2V_ACT_GELU_ERF)
    hipLaunchKernelGGL((activation_kernel<X2V_ACT_GELU_ERF>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, (unsigned short*)y, n);
  else
    hipLaunchKernelGGL((activation_kernel<X2V_EPI_SILU>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, (unsigned short*)y, n);
  X2V_LAUNCH_CHECK("activation launch");
  return X2V_OK;
}

extern "C" __attribute__((visibility("default"))) int x2v_sinusoid_embed_bf16(const int64_t* t, void* y, int n, int dim, void* stream) {
  X2V_REQUIRE(t && y, X2V_E_ARG, "sinusoid: null pointer");
  X2V_REQUIRE(n > 0 && dim > 0 && dim % 2 == 0, X2V_E_SHAPE, "sinusoid: n=%d dim=%d", n, dim);
  const int total = n * (dim / 2);
  hipLaunchKernelGGL(sinusoid_kernel, dim3((total + 127) / 128), dim3(128), 0, (hipStream_t)stream, t, (unsigned short*)y, n, dim);
  X2V_LAUNCH_CHECK("sinusoid launch");
  return X2V_OK;
}

static int headnorm_rope_impl(void* q, int64_t ldq, void* k, int64_t ldk, const void* wq, const void* wk, const void* cos_tab, const void* sin_tab, int64_t L, int H,
                              int64_t l_rope, float eps, int round_mode, float q_out_scale, int hpb, int64_t cbs, void* stream) {
  X2V_REQUIRE(q_out_scale > 0.f, X2V_E_ARG, "headnorm_rope: q_out_scale must be positive");
  X2V_REQUIRE(q && k, X2V_E_ARG, "headnorm_rope: null pointer");
  X2V_REQUIRE(L > 0 && H > 0 && l_rope >= 0 && l_rope <= L, X2V_E_SHAPE, "headnorm_rope: bad shape L=%lld H=%d l_rope=%lld", (long long)L, H, (long long)l_rope);
  X2V_REQUIRE(l_rope == 0 || (cos_tab && sin_tab), X2V_E_ARG, "headnorm_rope: rope tables missing");
  X2V_REQUIRE(hpb > 0 && H % hpb == 0 && cbs >= 0 && cbs % 8 == 0, X2V_E_SHAPE, "headnorm_rope: bad head blocking (H=%d, heads per block %d, block stride %lld)", H, hpb,
              (long long)cbs);
  X2V_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldq >= (int64_t)hpb * 128 && ldk >= (int64_t)hpb * 128 && aligned16(q) && aligned16(k) && aligned16(wq) && aligned16(wk) &&
                  aligned16(cos_tab) && aligned16(sin_tab),
              X2V_E_ALIGN, "headnorm_rope: rows must be 16-byte aligned");
  const int64_t rows = L * H;
  X2V_REQUIRE((rows + 15) / 16 < (1ll << 31), X2V_E_SHAPE, "headnorm_rope: too many rows");
  dim3 grid((unsigned)((rows + 15) / 16), 2);
  if (round_mode == X2V_ROUND_REF)
    hipLaunchKernelGGL((headnorm_rope_kernel<X2V_ROUND_REF>), grid, dim3(256), 0, (hipStream_t)stream, (unsigned short*)q, ldq, (unsigned short*)k, ldk,
                       (const unsigned short*)wq, (const unsigned short*)wk, (const unsigned short*)cos_tab, (const unsigned short*)sin_tab, L, H, l_rope, eps, q_out_scale, hpb,
                       cbs);
  else
    hipLaunchKernelGGL((headnorm_rope_kernel<X2V_ROUND_FP32>), grid, dim3(256), 0, (hipStream_t)stream, (unsigned short*)q, ldq, (unsigned short*)k, ldk,
                       (const unsigned short*)wq, (const unsigned short*)wk, (const unsigned short*)cos_tab, (const unsigned short*)sin_tab, L, H, l_rope, eps, q_out_scale, hpb,
                       cbs);
  X2V_LAUNCH_CHECK("headnorm_rope launch");
  return X2V_OK;
}

extern "C" __attribute__((visibility("default"))) int x2v_headnorm_rope_bf16(void* q, int64_t ldq, void* k, int64_t ldk, const void* wq, const void* wk, const void* cos_tab,
                                                                             const void* sin_tab, int64_t L, int H, int64_t l_rope, float eps, int round_mode,
                                                                             float q_out_scale, void* stream) {
  return headnorm_rope_impl(q, ldq, k, ldk, wq, wk, cos_tab, sin_tab, L, H, l_rope, eps, round_mode, q_out_scale, H, 0, stream);
}

extern "C" __attribute__((visibility("default"))) int x2v_headnorm_rope_blocked_bf16(void* q, void* k, int64_t ld, int heads_per_block, int64_t block_stride, const void* wq,
                                                                                     const void* wk, const void* cos_tab, const void* sin_tab, int64_t L, int H, int64_t l_rope,
                                                                                     float eps, int round_mode, float q_out_scale, void* stream) {
  return headnorm_rope_impl(q, ld, k, ld, wq, wk, cos_tab, sin_tab, L, H, l_rope, eps, round_mode, q_out_scale, heads_per_block, block_stride, stream);
}
