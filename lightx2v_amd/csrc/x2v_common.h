// Shared device helpers for the gfx950 kernels (wave64, MFMA fragment types, bf16 conversion).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "x2v.h"

namespace x2v {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x8_t __attribute__((ext_vector_type(8)));

constexpr int kWave = 64;  // CDNA wavefront width

// ---- error plumbing (host) -------------------------------------------------------------------------
void set_error(const char* fmt, ...);
// Process-wide A/B switches read ONCE from the environment (latched at first use; x2v_switches reports the effective values so that a bench line
// or a test can record which kernels a process really ran):
int gemm_continuous_switch();      // X2V_GEMM_CONTINUOUS      default 1: bf16 256x256 GEMMs take the continuous pipeline (gemm256c.hip) where the shape allows
int gemm_fp8_continuous_switch();  // X2V_GEMM_FP8_CONTINUOUS  default 2: w8a8 likewise (gemm256c8.hip), block-strided operands included; 1 = row-major only, 0 = ping-pong kernel
int attn_map_switch();             // X2V_ATTN_MAP             default -1: the launcher's rule; 0 / 1 force the plain / XCD-aware work mapping
int attn_rot_switch();             // X2V_ATTN_ROT             default -1: the caller's flag; 0 / 1 force the key walk from tile 0 / staggered
int check_hip(hipError_t e, const char* what);
bool aligned16(const void* p);
// Raise a kernel's dynamic-LDS cap to `bytes` on the current device (once per kernel and device; thread-safe).
int ensure_dynamic_lds(const void* kernel, int bytes, const char* what);

#define X2V_REQUIRE(cond, code, ...) \
  do {                               \
    if (!(cond)) {                   \
      ::x2v::set_error(__VA_ARGS__); \
      return (code);                 \
    }                                \
  } while (0)

#define X2V_LAUNCH_CHECK(what) \
  do {                                                              \
    int _rc = ::x2v::check_hip(hipGetLastError(), what);            \
    if (_rc != X2V_OK) return _rc;                                  \
  } while (0)

// Block addressing of the GEMM kernels (x2v_gemm_bf16_blocked): x may be stored as K-blocks — K-tile kt (128 bytes of a row) at
// (kt / a_kpb) * a_cbs + (kt % a_kpb) * 128 bytes from the row's start — and y as N-blocks — column n at (n / y_cbw) * y_cbs + n % y_cbw
// elements from the row's start.  These are the [N_ranks][S/N][(H/N)d] buffers of the Ulysses exchanges read / written in place.
// a_kpb <= 0 / y_cbw <= 0: plain row-major.
struct GemmBlocking {
  int a_kpb = 0;        // K-tiles (128 B) per x block
  unsigned a_cbs = 0;   // bytes between x blocks
  int y_cbw = 0;        // columns per y block (a multiple of 8)
  int64_t y_cbs = 0;    // elements between y blocks
};

// Batch strides and launch form of the pre-transposed-V attention kernel (attn.hip; tools/probes/attn_pc.hip takes the same struct)
struct AttnBatch {
  int64_t q, k, vt, o;  // elements from one sequence of the batch to the next (0: single sequence)
  int xcd_remap;        // bit 0: XCD-aware head-major work mapping (see the kernel), bit 8: staggered key walk; 0: the grid as dispatched, walk from tile 0
};

// ---- bf16 <-> fp32 (device) -------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float(((unsigned)u) << 16); }
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// round-to-nearest-even via the hardware v_cvt_pk_bf16_f32 (what `(__bf16)x` lowers to on gfx950)
__device__ __forceinline__ unsigned short f2bf(float x) {
  __bf16 b = (__bf16)x;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  bf16x2_t p;
  p[0] = (__bf16)lo;
  p[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned, p);
}
// round an fp32 value to bf16 precision and back (one reference rounding point)
__device__ __forceinline__ float rbf(float x) { return bf2f(f2bf(x)); }

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = bf_lo(v.x); f[1] = bf_hi(v.x);
  f[2] = bf_lo(v.y); f[3] = bf_hi(v.y);
  f[4] = bf_lo(v.z); f[5] = bf_hi(v.z);
  f[6] = bf_lo(v.w); f[7] = bf_hi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 v;
  v.x = pack_bf2(f[0], f[1]);
  v.y = pack_bf2(f[2], f[3]);
  v.z = pack_bf2(f[4], f[5]);
  v.w = pack_bf2(f[6], f[7]);
  return v;
}

// ---- activations (fp32 math on a bf16-rounded input, as torch does for bf16 tensors) ------------------
__device__ __forceinline__ float gelu_tanh_f(float x) {
  // torch: 0.5*x*(1+tanh(sqrt(2/pi)*(x+0.044715 x^3)))
  // 0.5*(1+tanh(u)) == 1/(1+exp(-2u)): one v_exp_f32 + one v_rcp_f32 instead of tanhf's long polynomial path
  // (fp32 relative error ~3e-7, far below the bf16 rounding that follows; limits: x -> -inf gives -0, +inf gives x)
  const float kBeta = 0.7978845608028654f, kKappa = 0.044715f;
  const float inner = kBeta * (x + kKappa * x * x * x);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * inner));
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

// ---- wave / block reductions -------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// Block-wide sum for blockDim.x = NW*64; `red` is NW floats of LDS. Every thread gets the total.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  if (NW == 1) return v;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();  // protect `red` from the previous use
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) t += red[i];
  return t;
}
template <int NW>
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  if (NW == 1) return v;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) t = fmaxf(t, red[i]);
  return t;
}

// XCD-aware remap of a 1-D block id: the dispatcher places block b on XCD b % 8 (observed, speed only);
// give every XCD a contiguous chunk of the logical tile space so neighbouring tiles share one L2.
// Bijective for any grid size (cdna_hip_programming.md §5 "XCD swizzle must be bijective").
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
  const unsigned nx = 8;
  const unsigned q = nblk / nx, r = nblk % nx;
  const unsigned xcd = bid % nx, idx = bid / nx;
  const unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

}  // namespace x2v
