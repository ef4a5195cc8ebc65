// y[M,N] = epi(x[M,K] . W[N,K]^T + bias) — bf16 (or w8a8 e4m3fn) in, fp32 MFMA accumulate, bf16 out.
// The fp8 variant shares the whole structure: a 128-byte operand row per stage is 64 bf16 or 128 fp8 k-values;
// it issues v_mfma_scale_f32_32x32x64_f8f6f4 with unit (2^0) block scales — the only fp8 MFMA that runs at
// the 2x rate on gfx950 — and applies the per-token x per-channel fp32 scales in the epilogue.
//
// Bound: MFMA (bf16 dense peak ~2.5 PFLOP/s); algorithmic work 2*M*N*K FLOP per launch.
//
// Structure (v1, "128^2 tile / 2-barrier" of the CDNA4 playbook):
//   * 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles of
//     v_mfma_f32_32x32x16_bf16), BK = 64 (one 128-byte line per operand row per stage).
//   * both operands are K-contiguous in HBM ([M,K] activations, [N,K] checkpoint weights), so A and B tiles
//     are staged identically: LDS-DMA (global_load_lds, 16 B/lane, 1 KiB per wave-instruction) into a
//     lane-linear [128 rows][128 B] image, double buffered (64 KiB -> 2 workgroups per CU).
//   * bank conflicts: the 16-byte chunk c of row r is stored at chunk c ^ ((r>>1)&7).  The permutation is
//     applied on the per-lane GLOBAL source address (the DMA destination is lane-linear by construction)
//     and again on the ds_read_b128 address; with ds_read_b128's 16-lane service groups this makes every
//     group hit 16 distinct 16-byte slots of the 256-byte bank row.
//   * operands are passed to the MFMA swapped (W fragment as A, x fragment as B) so that each lane's
//     accumulator registers hold 4 CONSECUTIVE output columns of one row -> 8-byte packed LDS writes in
//     the epilogue, which goes through LDS to emit full 16-byte row-contiguous global stores (and, for the
//     residual epilogue, 16-byte loads of the residual/gate).
//   * 1-D grid, XCD-aware remap + grouped (8 m-tiles) ordering so the 64 tiles resident on one XCD share
//     A/B panels through that XCD's L2.
//   * M and N tails: source rows are clamped (loads stay in bounds), stores are masked.  K % 64 == 0.
#include "x2v_common.h"

namespace x2v {

constexpr int GB_M = 128, GB_N = 128, GB_K = 64;        // GB_K counts bf16 elements; fp8 stages 128 per row
constexpr int G_ROW_BYTES = 128;                         // operand bytes per row per stage
constexpr int G_STAGE_BYTES = (GB_M + GB_N) * G_ROW_BYTES;  // 32 KiB
constexpr int G_LDS_BYTES = 2 * G_STAGE_BYTES;           // 64 KiB
constexpr int G_EPI_LD = 272;                            // bytes per epilogue row (128 bf16 + 16 pad)

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}

// swizzled byte offset of 16-byte chunk `c` of row `r` in a [rows][128 B] tile
__device__ __forceinline__ int swz_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

template <bool FP8, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const char* __restrict__ A, int64_t lda_bytes, const char* __restrict__ W, int64_t ldw_bytes,
                                                      const unsigned short* __restrict__ bias, unsigned short* __restrict__ Y, int64_t ldy, int64_t M,
                                                      int N, int nk, const unsigned short* __restrict__ resid, int64_t ldr,
                                                      const unsigned short* __restrict__ gate, const float* __restrict__ sx,
                                                      const float* __restrict__ sw, int ntm, int ntn, GemmBlocking gb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int a_kpb = gb.a_kpb > 0 && gb.a_kpb < nk ? gb.a_kpb : nk;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- tile coordinates (XCD chunking + grouped ordering)
  const unsigned nblk = (unsigned)ntm * (unsigned)ntn;
  const unsigned v = xcd_remap(blockIdx.x, nblk);
  constexpr unsigned GM = 8;
  const unsigned per_group = GM * (unsigned)ntn;
  const unsigned group = v / per_group, in_g = v % per_group;
  const unsigned first_m = group * GM;
  const unsigned gsz = min((unsigned)ntm - first_m, GM);
  const int tm = (int)(first_m + in_g % gsz), tn = (int)(in_g / gsz);
  const int64_t m0 = (int64_t)tm * GB_M;
  const int n0 = tn * GB_N;

  // ---- per-lane staging source pointers: wave `wid` stages rows [wid*32, wid*32+32) of A and of B,
  //      4 wave-instructions each; instruction i covers 8 rows x 8 chunks.
  const int srow = lane >> 3, spos = lane & 7;
  const char* a_src[4];
  const char* b_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wid * 32 + i * 8 + srow;
    const int c = spos ^ ((r >> 1) & 7);
    int64_t gm = m0 + r;
    gm = gm < M ? gm : M - 1;
    int gn = n0 + r;
    gn = gn < N ? gn : N - 1;
    a_src[i] = A + gm * lda_bytes + c * 16;
    b_src[i] = W + (int64_t)gn * ldw_bytes + c * 16;
  }
  auto stage = [&](int s, int kt) {
    char* as = smem + s * G_STAGE_BYTES + wid * (32 * 128);
    char* bs = as + GB_M * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(a_src[i] + ((int64_t)(kt / a_kpb) * gb.a_cbs + (int64_t)(kt % a_kpb) * G_ROW_BYTES), as + i * 1024);  // K-blocked x
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(b_src[i] + (int64_t)kt * G_ROW_BYTES, bs + i * 1024);
  };

  // ---- MFMA fragment read offsets (bytes within a tile image); wave (wr, wc) owns rows wr*64.., cols wc*64..
  const int wr = wid >> 1, wc = wid & 1;
  const int fl = lane & 31, fh = lane >> 5;
  // chunk order: bf16 step ks (K=16) reads chunk ks*2+fh; fp8 step s (K=64) reads chunks s*4+fh*2+{0,1}.
  // In both cases A and B use the same (half-wave, element) -> k map, which is all an MFMA requires.
  int a_off[2][4], b_off[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = FP8 ? ((ks >> 1) * 4 + fh * 2 + (ks & 1)) : (ks * 2 + fh);
      a_off[i][ks] = swz_off(wr * 64 + i * 32 + fl, c);
      b_off[i][ks] = swz_off(wc * 64 + i * 32 + fl, c) + GB_M * 128;
    }

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const char* base = smem + cur * G_STAGE_BYTES;
    if constexpr (!FP8) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8_t xa[2], wb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          xa[i] = *reinterpret_cast<const bf16x8_t*>(base + a_off[i][ks]);
          wb[i] = *reinterpret_cast<const bf16x8_t*>(base + b_off[i][ks]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        i32x8_t xa[2], wb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const i32x4_t a0 = *reinterpret_cast<const i32x4_t*>(base + a_off[i][s2 * 2]);
          const i32x4_t a1 = *reinterpret_cast<const i32x4_t*>(base + a_off[i][s2 * 2 + 1]);
          const i32x4_t b0 = *reinterpret_cast<const i32x4_t*>(base + b_off[i][s2 * 2]);
          const i32x4_t b1 = *reinterpret_cast<const i32x4_t*>(base + b_off[i][s2 * 2 + 1]);
          xa[i] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
          wb[i] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
        constexpr int kOne = 0x7f7f7f7f;  // e8m0 2^0 block scales
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wb[j], xa[i], acc[i][j], 0, 0, 0, kOne, 0, kOne);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue phase 1: acc (+bias, activation) -> bf16 -> LDS [128][G_EPI_LD]
  //      acc[i][j][r]: output row m = wr*64+i*32+fl, col n = wc*64+j*32 + (r&3) + 8*(r>>2) + 4*fh
  //      per-column operands (bias, fp8 channel scales) are fetched up front in one batch: one L2 round trip
  uint2 bv[2][4];
  float4 swv[2][4];
  float sxv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      int gn = n0 + wc * 64 + j * 32 + 8 * g + 4 * fh;
      gn = gn + 3 < N ? gn : (N >= 4 ? N - 4 : 0);
      bv[j][g] = make_uint2(0u, 0u);
      if (bias != nullptr) bv[j][g] = *reinterpret_cast<const uint2*>(bias + gn);
      if constexpr (FP8) swv[j][g] = *reinterpret_cast<const float4*>(sw + gn);
    }
  if constexpr (FP8) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int64_t gm = m0 + wr * 64 + i * 32 + fl;
      gm = gm < M ? gm : M - 1;
      sxv[i] = sx[gm];
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ml = wr * 64 + i * 32 + fl;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nl = wc * 64 + j * 32 + 8 * g + 4 * fh;
        float vv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) vv[e] = acc[i][j][4 * g + e];
        if constexpr (FP8) {
          vv[0] = vv[0] * sxv[i] * swv[j][g].x;
          vv[1] = vv[1] * sxv[i] * swv[j][g].y;
          vv[2] = vv[2] * sxv[i] * swv[j][g].z;
          vv[3] = vv[3] * sxv[i] * swv[j][g].w;
        }
        vv[0] += bf_lo(bv[j][g].x);
        vv[1] += bf_hi(bv[j][g].x);
        vv[2] += bf_lo(bv[j][g].y);
        vv[3] += bf_hi(bv[j][g].y);
        if (EPI == X2V_EPI_GELU_TANH) {
#pragma unroll
          for (int e = 0; e < 4; ++e) vv[e] = gelu_tanh_f(rbf(vv[e]));
        } else if (EPI == X2V_EPI_SILU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) vv[e] = silu_f(rbf(vv[e]));
        }
        uint2 pk;
        pk.x = pack_bf2(vv[0], vv[1]);
        pk.y = pack_bf2(vv[2], vv[3]);
        *reinterpret_cast<uint2*>(smem + ml * G_EPI_LD + nl * 2) = pk;
      }
    }
  }
  __syncthreads();
  // ---- epilogue phase 2: 16-byte row-contiguous stores (16 lanes per 256-byte output row)
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int id = it * 256 + tid;
    const int row = id >> 4, cc = id & 15;
    const int64_t gm = m0 + row;
    const int gn = n0 + cc * 8;
    if (gm < M && gn < N) {
      uint4 o = *reinterpret_cast<const uint4*>(smem + row * G_EPI_LD + cc * 16);
      if (EPI == X2V_EPI_RESIDUAL) {
        float yv[8], xv[8], ov[8];
        unpack8(o, yv);
        unpack8(*reinterpret_cast<const uint4*>(resid + gm * ldr + gn), xv);
        if (gate != nullptr) {
          float gv[8];
          unpack8(*reinterpret_cast<const uint4*>(gate + gn), gv);
#pragma unroll
          for (int e = 0; e < 8; ++e) ov[e] = xv[e] + rbf(yv[e] * gv[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) ov[e] = xv[e] + yv[e];
        }
        o = pack8(ov);
      }
      const int64_t ycol = gb.y_cbw > 0 ? (int64_t)(gn / gb.y_cbw) * gb.y_cbs + gn % gb.y_cbw : gn;  // N-blocked y
      *reinterpret_cast<uint4*>(Y + gm * ldy + ycol) = o;
    }
  }
}

}  // namespace x2v

namespace x2v {
// gemm256.hip: the 256x256-tile ping-pong kernel for large shapes
template <bool FP8>
int gemm256_dispatch(int epilogue, const void* x, int64_t ldxb, const void* w, int64_t ldwb, const void* bias, void* y, int64_t ldy, int64_t M, int N, int nk,
                     const void* resid, int64_t ldr, const void* gate, const float* sx, const float* sw, int gm_tiles, hipStream_t st, GemmBlocking gb);
// gemm256s.hip: the same tile as one software-pipelined wave per SIMD (bf16)
int gemm256s_dispatch(int epilogue, const void* x, int64_t ldxb, const void* w, int64_t ldwb, const void* bias, void* y, int64_t ldy, int64_t M, int N, int nk,
                      const void* resid, int64_t ldr, const void* gate, int gm_tiles, hipStream_t st, GemmBlocking gb);
int gemm256s_vt_dispatch(const void* x, int64_t ldxb, const void* w, int64_t ldwb, const void* bias, void* vt, int64_t ldvt, int64_t M, int N, int nk, hipStream_t st);
// gemm256c.hip: the single-stream kernel as a continuous pipeline over output tiles (persistent workgroups, register-direct epilogue)
bool gemm256c_ok(int nk, const GemmBlocking& gb);
int gemm256c_dispatch(int epilogue, const void* x, int64_t ldxb, const void* w, int64_t ldwb, const void* bias, void* y, int64_t ldy, int64_t M, int N, int nk,
                      const void* resid, int64_t ldr, const void* gate, int gm_tiles, hipStream_t st, GemmBlocking gb);
// gemm256c8.hip: the same continuous pipeline for the w8a8 operator (e4m3 operands, per-token / per-channel scales)
int gemm256c8_dispatch(int epilogue, const void* x, int64_t ldxb, const void* w, int64_t ldwb, const void* bias, void* y, int64_t ldy, int64_t M, int N, int nk,
                       const void* resid, int64_t ldr, const void* gate, const float* sx, const float* sw, int gm_tiles, hipStream_t st, GemmBlocking gb);
}  // namespace x2v

using namespace x2v;

template <bool FP8, int EPI>
static int launch_gemm(const void* x, int64_t ldx_bytes, const void* w, int64_t ldw_bytes, const void* bias, void* y, int64_t ldy, int64_t M, int N, int nk,
                       const void* resid, int64_t ldr, const void* gate, const float* sx, const float* sw, hipStream_t st, GemmBlocking gb) {
  const int ntm = (int)((M + GB_M - 1) / GB_M), ntn = (N + GB_N - 1) / GB_N;
  {
    int rc = ensure_dynamic_lds((const void*)gemm_kernel<FP8, EPI>, G_LDS_BYTES, "gemm attr");
    if (rc != X2V_OK) return rc;
  }
  hipLaunchKernelGGL((gemm_kernel<FP8, EPI>), dim3((unsigned)ntm * (unsigned)ntn), dim3(256), G_LDS_BYTES, st, (const char*)x, ldx_bytes, (const char*)w,
                     ldw_bytes, (const unsigned short*)bias, (unsigned short*)y, ldy, M, N, nk, (const unsigned short*)resid, ldr,
                     (const unsigned short*)gate, sx, sw, ntm, ntn, gb);
  X2V_LAUNCH_CHECK("gemm launch");
  return X2V_OK;
}

// The shape rule of variant 0 (one place: the dispatcher and x2v_gemm_kernel_choice both ask it): the 256^2 kernel wants at least
// ~one full round of tiles (256 CUs), a K loop longer than its pipeline, and 32-bit buffer offsets over a 256-row tile.
// bf16 takes the single-stream form of the 256^2 tile (3), fp8 / mxfp8 the ping-pong form (2).
#ifndef X2V_GEMM256_BF16_KERNEL
#define X2V_GEMM256_BF16_KERNEL 3
#endif
// The 256x256 kernels address an operand tile through a 32-bit buffer descriptor from the tile's first row: 255 rows + the K span of one row
// (K-blocked x: (K blocks - 1) block strides + one block) must stay below 4 GiB, else the range check would wrap and valid elements read as
// zero.  Such shapes take the 128x128 kernel (64-bit addressing).
static bool spans_fit_256(int nk, int64_t ldxb, int64_t ldwb, const GemmBlocking& gb) {
  const int64_t a_kpb = gb.a_kpb > 0 && gb.a_kpb < nk ? gb.a_kpb : nk;
  const int64_t a_span = a_kpb < nk ? ((nk - 1) / a_kpb) * (int64_t)gb.a_cbs + a_kpb * 128 : (int64_t)nk * 128;
  return ldxb < (1 << 24) && ldwb < (1 << 24) && 255 * ldxb + a_span < (1ll << 32) && 255 * ldwb + (int64_t)nk * 128 < (1ll << 32);
}

int x2v::gemm_continuous_switch() {
  static const int v = [] { const char* e = getenv("X2V_GEMM_CONTINUOUS"); return e == nullptr ? 1 : (atoi(e) != 0 ? 1 : 0); }();
  return v;
}
int x2v::gemm_fp8_continuous_switch() {
  static const int v = [] { const char* e = getenv("X2V_GEMM_FP8_CONTINUOUS"); return e == nullptr ? 2 : atoi(e); }();
  return v;
}

static int choose_kernel(int64_t M, int N, int nk, int64_t ldxb, int64_t ldwb, bool fp8, const GemmBlocking& gb = GemmBlocking()) {
  const bool fits256 = spans_fit_256(nk, ldxb, ldwb, gb);
  const int64_t tiles256 = ((M + 255) / 256) * (int64_t)((N + 255) / 256);
  return (fits256 && tiles256 >= 192 && nk >= 8) ? (fp8 ? 2 : X2V_GEMM256_BF16_KERNEL) : 1;
}

// variant: 0 = choose by shape, 1 = 128x128 kernel, 2 = 256x256 ping-pong kernel, 3 = 256x256 single-stream kernel (bf16) in the form the
// dispatcher prefers (continuous pipeline where the shape allows it), 4 = its one-output-tile-per-workgroup form (gemm256s.hip), 5 = its continuous
// form (gemm256c.hip; X2V_E_SHAPE where it does not apply); bits 8..15 = m-tiles per scheduling group of the 256x256 kernel (0 = default).
// X2V_GEMM_CONTINUOUS=0 in the environment turns the continuous form off for variants 0 / 3 (whole-model A/B runs).
template <bool FP8>
static int dispatch_epi(int epilogue, const void* x, int64_t ldxb, const void* w, int64_t ldwb, const void* bias, void* y, int64_t ldy, int64_t M, int N,
                        int nk, const void* resid, int64_t ldr, const void* gate, const float* sx, const float* sw, int variant, hipStream_t st,
                        GemmBlocking gb = GemmBlocking()) {
  int kind = variant & 0xff;
  const int gm_tiles = (variant >> 8) & 0xff;
  const bool fits256 = spans_fit_256(nk, ldxb, ldwb, gb);
  const int form = (kind == 4 || kind == 5) ? kind : 0;  // a forced form of the single-stream kernel
  if (form) kind = 3;
  if ((kind == 2 || kind == 3) && !fits256) {
    set_error("gemm: leading dimension / K-block span too large for the 256x256 kernels (32-bit tile addressing)");
    return X2V_E_SHAPE;
  }
  if (kind == 3 && FP8 && form != 5) {
    set_error("gemm: the one-tile-per-workgroup single-stream 256x256 kernel is bf16 only (fp8: variant 2 = ping-pong, 5 = continuous single-stream)");
    return X2V_E_ARG;
  }
  const int chosen = kind == 0 ? choose_kernel(M, N, nk, ldxb, ldwb, FP8, gb) : kind;
  if constexpr (FP8) {
    // gemm256c8.hip: the continuous single-stream pipeline for w8a8 — bit-equal with gemm256.hip's fp8 mode in 84 / 84 cases at first contact and
    // +2..5 % (plain), +1.5..2.5 % (GELU), +4..10 % (residual) at the w8a8 step's shapes (profiles/r04_call16_*).  Variant 0 takes it where the shape
    // allows, block-strided (Ulysses) operands included since round 5 (23 / 23 blocked cases bit-equal with the ping-pong kernel on both kernels,
    // profiles/r05_call1_*; tests/test_gpu_bench_shapes.py::test_gemm_fp8_blocked_*).  X2V_GEMM_FP8_CONTINUOUS=1 keeps blocked operands on the
    // ping-pong kernel, =0 turns the continuous form off (whole-model A/B runs); variant 5 forces it, variant 2 forces the ping-pong kernel.
    if (form == 5 || chosen == 2) {
      const int fp8_continuous_mode = gemm_fp8_continuous_switch();
      const bool unblocked = gb.a_kpb <= 0 && gb.y_cbw <= 0;
      const bool fp8_continuous_on = fp8_continuous_mode >= 2 || (fp8_continuous_mode == 1 && unblocked);
      const int64_t y_cols_span = gb.y_cbw > 0 ? (int64_t)((N - 1) / gb.y_cbw) * gb.y_cbs + gb.y_cbw : (int64_t)N;
      const bool can_c = gemm256c_ok(nk, gb) && N % 256 == 0 && (255 * ldy + y_cols_span) * 2 < 0x80000000ll && (resid == nullptr || (ldr == ldy && gb.y_cbw <= 0));
      if (form == 5 && !can_c) {
        set_error("gemm_fp8: the continuous single-stream kernel needs an even number of K tiles >= 4, N %% 256 == 0, y blocks that are multiples of 128 columns and resid with y's row stride (nk=%d, N=%d)", nk, N);
        return X2V_E_SHAPE;
      }
      if (form == 5 || (kind == 0 && fp8_continuous_on && can_c))
        return gemm256c8_dispatch(epilogue, x, ldxb, w, ldwb, bias, y, ldy, M, N, nk, resid, ldr, gate, sx, sw, gm_tiles, st, gb);
    }
  }
  if constexpr (!FP8) {
    // the single-stream kernel addresses its output (and residual) tile with 32-bit offsets from the tile's first row
    const int64_t y_cols_span = gb.y_cbw > 0 ? (int64_t)((N - 1) / gb.y_cbw) * gb.y_cbs + gb.y_cbw : (int64_t)N;
    const bool y32 = (255 * ldy + y_cols_span) * 2 < (1ll << 32) && (resid == nullptr || (255 * ldr + (int64_t)N) * 2 < (1ll << 32));
    if (chosen == 3 && !y32) {
      if (kind == 3) {
        set_error("gemm: output leading dimension / block stride too large for the single-stream 256x256 kernel (32-bit tile addressing)");
        return X2V_E_SHAPE;
      }
      return dispatch_epi<FP8>(epilogue, x, ldxb, w, ldwb, bias, y, ldy, M, N, nk, resid, ldr, gate, sx, sw, 2 | (gm_tiles << 8), st, gb);
    }
    if (chosen == 3) {
      // continuous form: its epilogue addresses the output / residual tile through descriptors of 2^31 bytes (offset 0x80000000 is the "no such
      // row" mark), so the tile spans must stay below that
      const bool continuous_on = gemm_continuous_switch() != 0;
      const bool can_c = gemm256c_ok(nk, gb) && N % 256 == 0 && (255 * ldy + y_cols_span) * 2 < 0x80000000ll && (resid == nullptr || (ldr == ldy && gb.y_cbw <= 0));  // residual tile addressed with y's offsets
      if (form == 5 && !can_c) {
        set_error("gemm: the continuous single-stream kernel needs an even number of K tiles >= 4, N %% 256 == 0, y blocks that are multiples of 128 columns and resid with y's row stride (nk=%d, N=%d)", nk, N);
        return X2V_E_SHAPE;
      }
      if (form == 5 || (form == 0 && continuous_on && can_c)) return gemm256c_dispatch(epilogue, x, ldxb, w, ldwb, bias, y, ldy, M, N, nk, resid, ldr, gate, gm_tiles, st, gb);
      return gemm256s_dispatch(epilogue, x, ldxb, w, ldwb, bias, y, ldy, M, N, nk, resid, ldr, gate, gm_tiles, st, gb);
    }
  }
  if (chosen == 2) {
    return gemm256_dispatch<FP8>(epilogue, x, ldxb, w, ldwb, bias, y, ldy, M, N, nk, resid, ldr, gate, sx, sw, gm_tiles, st, gb);
  }
  switch (epilogue) {
    case X2V_EPI_NONE: return launch_gemm<FP8, X2V_EPI_NONE>(x, ldxb, w, ldwb, bias, y, ldy, M, N, nk, nullptr, 0, nullptr, sx, sw, st, gb);
    case X2V_EPI_GELU_TANH: return launch_gemm<FP8, X2V_EPI_GELU_TANH>(x, ldxb, w, ldwb, bias, y, ldy, M, N, nk, nullptr, 0, nullptr, sx, sw, st, gb);
    case X2V_EPI_SILU: return launch_gemm<FP8, X2V_EPI_SILU>(x, ldxb, w, ldwb, bias, y, ldy, M, N, nk, nullptr, 0, nullptr, sx, sw, st, gb);
    case X2V_EPI_RESIDUAL: return launch_gemm<FP8, X2V_EPI_RESIDUAL>(x, ldxb, w, ldwb, bias, y, ldy, M, N, nk, resid, ldr, gate, sx, sw, st, gb);
    default: set_error("gemm: unknown epilogue %d", epilogue); return X2V_E_ARG;
  }
}

static int check_common(const char* who, const void* y, int64_t ldy, int64_t M, int N, const void* bias, int epilogue, const void* resid, int64_t ldr,
                        const void* gate) {
  X2V_REQUIRE(M >= 0 && N > 0, X2V_E_SHAPE, "%s: bad shape M=%lld N=%d", who, (long long)M, N);
  X2V_REQUIRE(N % 8 == 0, X2V_E_SHAPE, "%s: N=%d must be a multiple of 8", who, N);
  X2V_REQUIRE(ldy % 8 == 0 && ldy >= N && aligned16(y), X2V_E_ALIGN, "%s: output rows must be 16-byte aligned", who);
  X2V_REQUIRE(bias == nullptr || ((uintptr_t)bias % 8) == 0, X2V_E_ALIGN, "%s: bias must be 8-byte aligned", who);
  X2V_REQUIRE((M + GB_M - 1) / GB_M * (int64_t)((N + GB_N - 1) / GB_N) < (1ll << 31), X2V_E_SHAPE, "%s: too many tiles", who);
  if (epilogue == X2V_EPI_RESIDUAL) {
    X2V_REQUIRE(resid != nullptr, X2V_E_ARG, "%s: residual epilogue needs resid", who);
    X2V_REQUIRE(ldr % 8 == 0 && ldr >= N && aligned16(resid) && aligned16(gate), X2V_E_ALIGN, "%s: resid/gate alignment", who);
  }
  return X2V_OK;
}

extern "C" __attribute__((visibility("default"))) int x2v_gemm_bf16_variant(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias, void* y, int64_t ldy, int64_t M, int N, int K,
                             int epilogue, const void* resid, int64_t ldr, const void* gate, int variant, void* stream) {
  X2V_REQUIRE(x && w && y, X2V_E_ARG, "gemm_bf16: null pointer");
  X2V_REQUIRE(K > 0 && K % GB_K == 0, X2V_E_SHAPE, "gemm_bf16: K=%d must be a positive multiple of %d", K, GB_K);
  X2V_REQUIRE(ldx % 8 == 0 && ldw % 8 == 0 && aligned16(x) && aligned16(w), X2V_E_ALIGN, "gemm_bf16: operand rows must be 16-byte aligned");
  X2V_REQUIRE(ldx >= K && ldw >= K, X2V_E_SHAPE, "gemm_bf16: leading dimension smaller than K");
  int rc = check_common("gemm_bf16", y, ldy, M, N, bias, epilogue, resid, ldr, gate);
  if (rc != X2V_OK) return rc;
  if (M == 0) return X2V_OK;
  return dispatch_epi<false>(epilogue, x, ldx * 2, w, ldw * 2, bias, y, ldy, M, N, K / GB_K, resid, ldr, gate, nullptr, nullptr, variant, (hipStream_t)stream);
}

static int check_blocking(const char* who, int K_bytes_per_row, int64_t x_kblock_bytes, int64_t x_kblock_stride_bytes, int N, int y_nblock, int64_t y_nblock_stride, int epilogue,
                          GemmBlocking* gb) {
  if (x_kblock_bytes > 0) {
    X2V_REQUIRE(x_kblock_bytes % 128 == 0 && K_bytes_per_row % x_kblock_bytes == 0 && x_kblock_stride_bytes % 16 == 0 && x_kblock_stride_bytes > 0 &&
                    x_kblock_stride_bytes < (1ll << 31),
                X2V_E_SHAPE, "%s: x K-block of %lld bytes must be a multiple of 128 dividing the row; block stride a multiple of 16 bytes", who, (long long)x_kblock_bytes);
    gb->a_kpb = (int)(x_kblock_bytes / 128);
    gb->a_cbs = (unsigned)x_kblock_stride_bytes;
  }
  if (y_nblock > 0) {
    X2V_REQUIRE(y_nblock % 8 == 0 && N % y_nblock == 0 && y_nblock_stride % 8 == 0, X2V_E_SHAPE, "%s: y N-block of %d columns must be a multiple of 8 dividing N=%d", who, y_nblock, N);
    X2V_REQUIRE(epilogue != X2V_EPI_RESIDUAL, X2V_E_ARG, "%s: the residual epilogue reads y's layout from resid: not available with an N-blocked y", who);
    gb->y_cbw = y_nblock;
    gb->y_cbs = y_nblock_stride;
  }
  return X2V_OK;
}

extern "C" __attribute__((visibility("default"))) int x2v_gemm_bf16_blocked(const void* x, int64_t ldx, int x_kblock, int64_t x_kblock_stride, const void* w, int64_t ldw,
                                                                            const void* bias, void* y, int64_t ldy, int y_nblock, int64_t y_nblock_stride, int64_t M, int N,
                                                                            int K, int epilogue, const void* resid, int64_t ldr, const void* gate, void* stream) {
  X2V_REQUIRE(x && w && y, X2V_E_ARG, "gemm_bf16_blocked: null pointer");
  X2V_REQUIRE(K > 0 && K % GB_K == 0, X2V_E_SHAPE, "gemm_bf16_blocked: K=%d must be a positive multiple of %d", K, GB_K);
  X2V_REQUIRE(ldx % 8 == 0 && ldw % 8 == 0 && aligned16(x) && aligned16(w), X2V_E_ALIGN, "gemm_bf16_blocked: operand rows must be 16-byte aligned");
  X2V_REQUIRE(ldx >= (x_kblock > 0 ? x_kblock : K) && ldw >= K, X2V_E_SHAPE, "gemm_bf16_blocked: leading dimension smaller than the row");
  int rc = check_common("gemm_bf16_blocked", y, ldy, M, y_nblock > 0 ? y_nblock : N, bias, epilogue, resid, ldr, gate);
  if (rc != X2V_OK) return rc;
  GemmBlocking gb;
  rc = check_blocking("gemm_bf16_blocked", K * 2, (int64_t)x_kblock * 2, x_kblock_stride * 2, N, y_nblock, y_nblock_stride, epilogue, &gb);
  if (rc != X2V_OK) return rc;
  if (M == 0) return X2V_OK;
  return dispatch_epi<false>(epilogue, x, ldx * 2, w, ldw * 2, bias, y, ldy, M, N, K / GB_K, resid, ldr, gate, nullptr, nullptr, 0, (hipStream_t)stream, gb);
}

extern "C" __attribute__((visibility("default"))) int x2v_gemm_fp8_blocked(const void* xq, int64_t ldx, int x_kblock, int64_t x_kblock_stride, const float* sx, const void* wq,
                                                                           int64_t ldw, const float* sw, const void* bias, void* y, int64_t ldy, int y_nblock,
                                                                           int64_t y_nblock_stride, int64_t M, int N, int K, int epilogue, const void* resid, int64_t ldr,
                                                                           const void* gate, void* stream) {
  X2V_REQUIRE(xq && wq && y && sx && sw, X2V_E_ARG, "gemm_fp8_blocked: null pointer");
  X2V_REQUIRE(K > 0 && K % 128 == 0, X2V_E_SHAPE, "gemm_fp8_blocked: K=%d must be a positive multiple of 128", K);
  X2V_REQUIRE(ldx % 16 == 0 && ldw % 16 == 0 && aligned16(xq) && aligned16(wq) && aligned16(sw), X2V_E_ALIGN, "gemm_fp8_blocked: operand rows must be 16-byte aligned");
  X2V_REQUIRE(ldx >= (x_kblock > 0 ? x_kblock : K) && ldw >= K, X2V_E_SHAPE, "gemm_fp8_blocked: leading dimension smaller than the row");
  int rc = check_common("gemm_fp8_blocked", y, ldy, M, y_nblock > 0 ? y_nblock : N, bias, epilogue, resid, ldr, gate);
  if (rc != X2V_OK) return rc;
  GemmBlocking gb;
  rc = check_blocking("gemm_fp8_blocked", K, x_kblock, x_kblock_stride, N, y_nblock, y_nblock_stride, epilogue, &gb);
  if (rc != X2V_OK) return rc;
  if (M == 0) return X2V_OK;
  return dispatch_epi<true>(epilogue, xq, ldx, wq, ldw, bias, y, ldy, M, N, K / 128, resid, ldr, gate, sx, sw, 0, (hipStream_t)stream, gb);
}

extern "C" __attribute__((visibility("default"))) int x2v_gemm_bf16(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias, void* y, int64_t ldy, int64_t M, int N, int K,
                             int epilogue, const void* resid, int64_t ldr, const void* gate, void* stream) {
  return x2v_gemm_bf16_variant(x, ldx, w, ldw, bias, y, ldy, M, N, K, epilogue, resid, ldr, gate, 0, stream);
}

extern "C" __attribute__((visibility("default"))) int x2v_gemm_fp8_variant(const void* xq, int64_t ldx, const float* sx, const void* wq, int64_t ldw, const float* sw, const void* bias, void* y,
                            int64_t ldy, int64_t M, int N, int K, int epilogue, const void* resid, int64_t ldr, const void* gate, int variant, void* stream) {
  X2V_REQUIRE(xq && wq && y && sx && sw, X2V_E_ARG, "gemm_fp8: null pointer");
  X2V_REQUIRE(K > 0 && K % 128 == 0, X2V_E_SHAPE, "gemm_fp8: K=%d must be a positive multiple of 128", K);
  X2V_REQUIRE(ldx % 16 == 0 && ldw % 16 == 0 && aligned16(xq) && aligned16(wq) && aligned16(sw), X2V_E_ALIGN, "gemm_fp8: operand rows must be 16-byte aligned");
  X2V_REQUIRE(ldx >= K && ldw >= K, X2V_E_SHAPE, "gemm_fp8: leading dimension smaller than K");
  int rc = check_common("gemm_fp8", y, ldy, M, N, bias, epilogue, resid, ldr, gate);
  if (rc != X2V_OK) return rc;
  if (M == 0) return X2V_OK;
  return dispatch_epi<true>(epilogue, xq, ldx, wq, ldw, bias, y, ldy, M, N, K / 128, resid, ldr, gate, sx, sw, variant, (hipStream_t)stream);
}

extern "C" __attribute__((visibility("default"))) int x2v_gemm_fp8(const void* xq, int64_t ldx, const float* sx, const void* wq, int64_t ldw, const float* sw, const void* bias, void* y,
                            int64_t ldy, int64_t M, int N, int K, int epilogue, const void* resid, int64_t ldr, const void* gate, void* stream) {
  return x2v_gemm_fp8_variant(xq, ldx, sx, wq, ldw, sw, bias, y, ldy, M, N, K, epilogue, resid, ldr, gate, 0, stream);
}

// v projection with V^T output (the operand of x2v_attn_fwd_bf16_vt) straight from the GEMM epilogue: only the single-stream 256x256 kernel
// has this output mode, so shapes the dispatcher would give to another kernel are refused (X2V_E_SHAPE: run x2v_gemm_bf16 +
// x2v_transpose_heads_bf16 instead; x2v_gemm_kernel_choice tells beforehand).
extern "C" __attribute__((visibility("default"))) int x2v_gemm_bf16_vt(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias, void* vt, int64_t ldvt, int64_t M,
                                                                       int N, int K, void* stream) {
  X2V_REQUIRE(x && w && vt, X2V_E_ARG, "gemm_bf16_vt: null pointer");
  X2V_REQUIRE(K > 0 && K % GB_K == 0, X2V_E_SHAPE, "gemm_bf16_vt: K=%d must be a positive multiple of %d", K, GB_K);
  X2V_REQUIRE(M > 0 && N > 0 && N % 128 == 0, X2V_E_SHAPE, "gemm_bf16_vt: N=%d must be whole heads of 128", N);
  X2V_REQUIRE(ldvt % 64 == 0 && ldvt >= M, X2V_E_SHAPE, "gemm_bf16_vt: ldvt must be a multiple of 64 and >= M");
  X2V_REQUIRE(ldx % 8 == 0 && ldw % 8 == 0 && ldx >= K && ldw >= K && aligned16(x) && aligned16(w) && aligned16(vt) && (bias == nullptr || ((uintptr_t)bias % 2) == 0),
              X2V_E_ALIGN, "gemm_bf16_vt: operand rows must be 16-byte aligned");
  X2V_REQUIRE(choose_kernel(M, N, K / GB_K, ldx * 2, ldw * 2, false) == 3, X2V_E_SHAPE,
              "gemm_bf16_vt: this shape is not dispatched to the single-stream 256x256 kernel (use x2v_gemm_bf16 + x2v_transpose_heads_bf16)");
  return gemm256s_vt_dispatch(x, ldx * 2, w, ldw * 2, bias, vt, ldvt, M, N, K / GB_K, (hipStream_t)stream);
}

extern "C" __attribute__((visibility("default"))) int x2v_gemm_kernel_choice(int64_t M, int N, int K, int64_t ldx, int64_t ldw, int fp8) {
  if (M <= 0 || N <= 0 || K <= 0) return X2V_E_SHAPE;
  const int nk = fp8 ? K / 128 : K / GB_K;
  const int tile = fp8 ? choose_kernel(M, N, nk, ldx, ldw, true) : choose_kernel(M, N, nk, ldx * 2, ldw * 2, false);
  // bit 8: variant 0 runs the CONTINUOUS-pipeline form of that tile family (gemm256c.hip / gemm256c8.hip) for a row-major y (ldy == N) and a
  // residual of y's stride — what the parity tests and the bench line record
  const bool cont = (tile == 3 && !fp8 && gemm_continuous_switch() != 0) || (tile == 2 && fp8 && gemm_fp8_continuous_switch() >= 1);
  const bool can_c = gemm256c_ok(nk, GemmBlocking()) && N % 256 == 0 && (255 * (int64_t)N + N) * 2 < 0x80000000ll;
  return tile | ((cont && can_c) ? 0x100 : 0);
}
