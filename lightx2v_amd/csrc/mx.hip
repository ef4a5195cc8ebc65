// MXFP8 (OCP microscaling: e4m3 elements, one e8m0 power-of-two scale per 32 consecutive K elements) quantisation and GEMM on
// gfx950's native block-scaled matrix instruction — the counterpart of the reference's lightx2v_kernel package
//   scaled_fp8_quant            lightx2v_kernel/python/lightx2v_kernel/gemm.py:73-83, csrc/gemm/mxfp8_quant_kernels_sm120.cu:139-196
//   cutlass_scaled_mxfp8_mm     gemm.py:93-97, csrc/gemm/mxfp8_scaled_mm_kernels_sm120.cu:60-66,150-160  (D = alpha * A.B^T + bias[n], bf16)
//
// Scale layout: sc[K/128][rows][4] bytes — K-tile major, then operand row, then the 4 k blocks of that 128-wide K-tile.  Like the
// reference's [m/128][k/128][32][4][4] swizzle (the operand format of the sm120 tensor core) it is the consumer's format:
// v_mfma_scale_f32_32x32x64_f8f6f4 takes, per lane, the scale byte of that lane's operand row and 32-wide k block from a VGPR, so a
// GEMM wave needs one dword per row and K-tile, and with this layout the 32 rows of a fragment are 128 contiguous bytes (one cache
// line per load instead of 32 with a row-major [rows][K/32] table: 1.6-1.85 -> 2.4-2.7 PFLOP/s on the 14B shapes, profiles/r01_mxfp8_*).
//
// quant: HBM-bound — reads M*K bf16, writes M*K bytes + M*K/32 scale bytes (algorithmic bytes 3.03 per element).
// GEMM:  MFMA-bound — 2*M*N*K FLOP per launch against the 5 PFLOP/s fp8 peak; the 128x128-tile / two-barrier structure of gemm.hip
//        (first MX version: correctness and the instruction's data path; the 256x256 ping-pong form of gemm256.hip is the next step).
#include "x2v_common.h"

namespace x2v {

// ------------------------------------------------------------------------------------------------ quantisation
// One lane = 8 consecutive elements (16 B in, 8 B out), 4 lanes = one 32-element scale block.
//   vecMax = max |x|;  SF = vecMax / 448;  scale byte = e8m0(SF) rounded towards +inf (smallest power of two >= SF), saturating;
//   q = e4m3fn_rne(x * 2^-(byte-127))   — |x| * 2^-e <= 448 by construction, so the conversion never saturates.
// An all-zero block gives byte 0 (2^-127) and zero elements (the reference's 0 * rcp(flushed denormal) = NaN there is not reproduced).
__device__ __forceinline__ unsigned e8m0_ceil(float sf) {
  const unsigned bits = __float_as_uint(sf);
  const unsigned ex = bits >> 23, man = bits & 0x7fffffu;
  unsigned byte;
  if (ex == 0) byte = man > 0x400000u ? 1u : 0u;  // denormal SF: 2^-127 covers (0, 2^-127], 2^-126 the rest
  else byte = ex + (man != 0u ? 1u : 0u);
  return byte > 254u ? 254u : byte;
}

__global__ __launch_bounds__(256) void quant_mxfp8_kernel(const unsigned short* __restrict__ x, int64_t ldx, unsigned char* __restrict__ q, int64_t ldq,
                                                          unsigned* __restrict__ sc, int64_t M, int K) {
  const int kc = K >> 3;  // 16-byte chunks per row
  const int64_t total = M * kc;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total + 3; id += (int64_t)gridDim.x * 256) {
    // the 16 lanes of one K-tile of one row (16 consecutive ids) are always all in range or all out of range (kc % 16 == 0)
    const bool ok = id < total;
    const int64_t row = ok ? id / kc : 0;
    const int c = ok ? (int)(id - row * kc) : 0;
    float v[8];
    float amax = 0.f;
    if (ok) {
      unpack8(*reinterpret_cast<const uint4*>(x + row * ldx + c * 8), v);
#pragma unroll
      for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[j]));
    }
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    if (!ok) continue;
    const unsigned byte = e8m0_ceil(amax / 448.0f);
    const unsigned e = 254u - byte;
    const float inv = __uint_as_float(e >= 1u ? (e << 23) : 0x00400000u);  // 2^(127 - byte)
    unsigned lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * inv, v[5] * inv, hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * inv, v[7] * inv, hi, true);
    *reinterpret_cast<uint2*>(q + row * ldq + c * 8) = make_uint2(lo, hi);
    // the 4 scale bytes of a row's K-tile sit in lanes c, c+4, c+8, c+12 (16 lanes = 128 elements): gather them into one dword
    unsigned dw = byte;
    dw |= (unsigned)__shfl_down((int)byte, 4, 64) << 8;
    dw |= (unsigned)__shfl_down((int)byte, 8, 64) << 16;
    dw |= (unsigned)__shfl_down((int)byte, 12, 64) << 24;
    if ((c & 15) == 0) sc[(int64_t)(c >> 4) * M + row] = dw;
  }
}

// ------------------------------------------------------------------------------------------------ GEMM
constexpr int MX_M = 128, MX_N = 128;
constexpr int MX_STAGE_BYTES = (MX_M + MX_N) * 128;  // one K-tile (128 fp8 per operand row): 32 KiB
constexpr int MX_LDS_BYTES = 2 * MX_STAGE_BYTES;
constexpr int MX_EPI_LD = 272;

typedef __attribute__((address_space(3))) void* mx_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* mx_gbl_ptr_t;

__device__ __forceinline__ int mx_swz(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

// y[M,N] = alpha * (A[M,K] . B[N,K]^T) + bias[n], A/B e4m3 with per-(row, 32-k) e8m0 scales.  Same tile structure as gemm_kernel<true>.
__global__ __launch_bounds__(256, 2) void gemm_mxfp8_kernel(const char* __restrict__ A, int64_t lda, const unsigned* __restrict__ SA,
                                                            const char* __restrict__ B, int64_t ldb, const unsigned* __restrict__ SB,
                                                            const unsigned short* __restrict__ bias, const float* __restrict__ alpha_p,
                                                            unsigned short* __restrict__ Y, int64_t ldy, int64_t M, int N, int nk, int ntm, int ntn) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned nblk = (unsigned)ntm * (unsigned)ntn;
  const unsigned v = xcd_remap(blockIdx.x, nblk);
  constexpr unsigned GM = 8;
  const unsigned per_group = GM * (unsigned)ntn;
  const unsigned group = v / per_group, in_g = v % per_group;
  const unsigned first_m = group * GM;
  const unsigned gsz = min((unsigned)ntm - first_m, GM);
  const int tm = (int)(first_m + in_g % gsz), tn = (int)(in_g / gsz);
  const int64_t m0 = (int64_t)tm * MX_M;
  const int n0 = tn * MX_N;

  // operand staging by LDS-DMA: wave `wid` stages rows [wid*32, wid*32+32) of A and of B, 8 rows x 8 chunks per instruction
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
    a_src[i] = A + gm * lda + c * 16;
    b_src[i] = B + (int64_t)gn * ldb + c * 16;
  }
  auto stage = [&](int s, int kt) {
    char* as = smem + s * MX_STAGE_BYTES + wid * (32 * 128);
    char* bs = as + MX_M * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((mx_gbl_ptr_t)(a_src[i] + (int64_t)kt * 128), (mx_lds_ptr_t)(as + i * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((mx_gbl_ptr_t)(b_src[i] + (int64_t)kt * 128), (mx_lds_ptr_t)(bs + i * 1024), 16, 0, 0);
  };

  const int wr = wid >> 1, wc = wid & 1;
  const int fl = lane & 31, fh = lane >> 5;
  // fragment chunks in the instruction's own k order (it matters once the hardware applies block scales; probed with
  // tools/x2v_check-style unit tests, see tests/test_gpu_mx.py): of a lane's 32 bytes the first 16 are k = fh*16 .. +15 (scale block 0
  // of the 64-wide step), the last 16 are k = 32 + fh*16 .. +15 (block 1); block b takes its scale from the lanes of half b.
  // MFMA s of the K-tile therefore reads 16-byte chunks s*4 + fh and s*4 + 2 + fh.
  int a_off[2][4], b_off[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = (ks >> 1) * 4 + (ks & 1) * 2 + fh;
      a_off[i][ks] = mx_swz(wr * 64 + i * 32 + fl, c);
      b_off[i][ks] = mx_swz(wc * 64 + i * 32 + fl, c) + MX_M * 128;
    }
  // block scales: one dword per operand row and K-tile = the 4 scale bytes of its k blocks.  The instruction takes byte `opsel` of the
  // scale VGPR of every lane; lanes fh = 1 (k block s*2 + 1) shift their dword down by one byte so that opsel = 2 s serves both halves.
  const unsigned* sa_row[2];
  const unsigned* sb_row[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int64_t gm = m0 + wr * 64 + i * 32 + fl;
    gm = gm < M ? gm : M - 1;
    int gn = n0 + wc * 64 + i * 32 + fl;
    gn = gn < N ? gn : N - 1;
    sa_row[i] = SA + gm;  // [K-tile][row] dwords
    sb_row[i] = SB + gn;
  }
  const int sh = fh * 8;
  unsigned sa_cur[2], sb_cur[2], sa_nxt[2] = {0u, 0u}, sb_nxt[2] = {0u, 0u};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    sa_cur[i] = sa_row[i][0] >> sh;
    sb_cur[i] = sb_row[i][0] >> sh;
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
    if (kt + 1 < nk) {
      stage(cur ^ 1, kt + 1);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        sa_nxt[i] = sa_row[i][(int64_t)(kt + 1) * M];
        sb_nxt[i] = sb_row[i][(int64_t)(kt + 1) * N];
      }
    }
    const char* base = smem + cur * MX_STAGE_BYTES;
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
      // first matrix operand = B (weight) fragment, second = A (activation) fragment: the accumulator's lane index is the output row
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (s2 == 0)
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wb[j], xa[i], acc[i][j], 0, 0, 0, (int)sb_cur[j], 0, (int)sa_cur[i]);
          else
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wb[j], xa[i], acc[i][j], 0, 0, 2, (int)sb_cur[j], 2, (int)sa_cur[i]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      sa_cur[i] = sa_nxt[i] >> sh;
      sb_cur[i] = sb_nxt[i] >> sh;
    }
    __syncthreads();
  }

  // epilogue: alpha, bias -> bf16 -> LDS -> 16-byte row-contiguous stores
  const float alpha = alpha_p != nullptr ? *alpha_p : 1.0f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ml = wr * 64 + i * 32 + fl;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nl = wc * 64 + j * 32 + 8 * g + 4 * fh;
        int gn = n0 + nl;
        gn = gn + 3 < N ? gn : (N >= 4 ? N - 4 : 0);
        uint2 bv = make_uint2(0u, 0u);
        if (bias != nullptr) bv = *reinterpret_cast<const uint2*>(bias + gn);
        uint2 pk;
        pk.x = pack_bf2(acc[i][j][4 * g + 0] * alpha + bf_lo(bv.x), acc[i][j][4 * g + 1] * alpha + bf_hi(bv.x));
        pk.y = pack_bf2(acc[i][j][4 * g + 2] * alpha + bf_lo(bv.y), acc[i][j][4 * g + 3] * alpha + bf_hi(bv.y));
        *reinterpret_cast<uint2*>(smem + ml * MX_EPI_LD + nl * 2) = pk;
      }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int id = it * 256 + tid;
    const int row = id >> 4, cc = id & 15;
    const int64_t gm = m0 + row;
    const int gn = n0 + cc * 8;
    if (gm < M && gn < N) *reinterpret_cast<uint4*>(Y + gm * ldy + gn) = *reinterpret_cast<const uint4*>(smem + row * MX_EPI_LD + cc * 16);
  }
}

// gemm256.hip: the 256x256-tile ping-pong kernel with hardware block scales (large shapes)
int gemm256_mx_dispatch(int epilogue, const void* a, int64_t lda, const void* sa, const void* b, int64_t ldb, const void* sb, const void* bias, const float* alpha,
                        void* y, int64_t ldy, int64_t M, int N, int nk, const void* resid, int64_t ldr, const void* gate, hipStream_t st);

}  // namespace x2v

using namespace x2v;

extern "C" __attribute__((visibility("default"))) int x2v_quant_mxfp8_bf16(const void* x, int64_t ldx, void* q, int64_t ldq, void* scales, int64_t M, int K,
                                                                          void* stream) {
  X2V_REQUIRE(x && q && scales, X2V_E_ARG, "quant_mxfp8: null pointer");
  X2V_REQUIRE(K > 0 && K % 128 == 0, X2V_E_SHAPE, "quant_mxfp8: K=%d must be a multiple of 128 (4 scale blocks = one K-tile of the GEMM)", K);
  X2V_REQUIRE(ldx % 8 == 0 && ldq % 8 == 0 && ldx >= K && ldq >= K && aligned16(x) && ((uintptr_t)q % 8) == 0 && ((uintptr_t)scales % 4) == 0, X2V_E_ALIGN,
              "quant_mxfp8: rows must be 16-byte (input) / 8-byte (output) aligned, the scale table 4-byte aligned");
  if (M <= 0) return X2V_OK;
  const int64_t total = M * (K / 8);
  const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(quant_mxfp8_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, ldx, (unsigned char*)q, ldq,
                     (unsigned*)scales, M, K);
  X2V_LAUNCH_CHECK("quant_mxfp8 launch");
  return X2V_OK;
}

static int gemm_mxfp8_impl(const void* a, int64_t lda, const void* sa, const void* b, int64_t ldb, const void* sb, const void* bias, const float* alpha, void* y,
                           int64_t ldy, int64_t M, int N, int K, int variant, void* stream, int epilogue = X2V_EPI_NONE, const void* resid = nullptr,
                           int64_t ldr = 0, const void* gate = nullptr) {
  X2V_REQUIRE(a && sa && b && sb && y, X2V_E_ARG, "gemm_mxfp8: null pointer");
  X2V_REQUIRE(M > 0 && N > 0 && K > 0 && K % 128 == 0 && N % 8 == 0, X2V_E_SHAPE, "gemm_mxfp8: M=%lld N=%d K=%d (K %% 128 == 0, N %% 8 == 0)", (long long)M, N, K);
  X2V_REQUIRE(lda % 16 == 0 && ldb % 16 == 0 && lda >= K && ldb >= K && ldy % 8 == 0 && ldy >= N && aligned16(a) && aligned16(b) && aligned16(y) &&
                  ((uintptr_t)sa % 4) == 0 && ((uintptr_t)sb % 4) == 0,
              X2V_E_ALIGN, "gemm_mxfp8: operand rows 16-byte aligned, scale tables 4-byte aligned");
  X2V_REQUIRE((int64_t)(K / 128) * std::max<int64_t>(M, N) * 4 < (1ll << 32), X2V_E_SHAPE, "gemm_mxfp8: scale table of 4 GiB or more");
  // same kernel choice as the other GEMMs (gemm.hip): the 256x256 ping-pong kernel when it fills the chip, else 128x128 tiles
  const int64_t tiles256 = ((M + 255) / 256) * ((N + 255) / 256);
  // the fused epilogues (activation / gated residual) live in the 256x256 kernel only: any shape with one goes there
  const bool big = variant == 2 || epilogue != X2V_EPI_NONE || (variant == 0 && tiles256 >= 192 && K / 128 >= 8 && lda < (1 << 24) && ldb < (1 << 24));
  if (epilogue == X2V_EPI_RESIDUAL)
    X2V_REQUIRE(resid != nullptr && ldr % 8 == 0 && ldr >= N && aligned16(resid) && (gate == nullptr || aligned16(gate)), X2V_E_ALIGN,
                "gemm_mxfp8: residual epilogue needs a 16-byte aligned residual (and gate)");
  if (big) return gemm256_mx_dispatch(epilogue, a, lda, sa, b, ldb, sb, bias, alpha, y, ldy, M, N, K / 128, resid, ldr, gate, (hipStream_t)stream);
  {
    int rc = ensure_dynamic_lds((const void*)gemm_mxfp8_kernel, MX_LDS_BYTES, "gemm_mxfp8 attr");
    if (rc != X2V_OK) return rc;
  }
  const int ntm = (int)((M + MX_M - 1) / MX_M), ntn = (N + MX_N - 1) / MX_N;
  hipLaunchKernelGGL(gemm_mxfp8_kernel, dim3((unsigned)ntm * (unsigned)ntn), dim3(256), MX_LDS_BYTES, (hipStream_t)stream, (const char*)a, lda,
                     (const unsigned*)sa, (const char*)b, ldb, (const unsigned*)sb, (const unsigned short*)bias, alpha, (unsigned short*)y, ldy, M,
                     N, K / 128, ntm, ntn);
  X2V_LAUNCH_CHECK("gemm_mxfp8 launch");
  return X2V_OK;
}

extern "C" __attribute__((visibility("default"))) int x2v_gemm_mxfp8(const void* a, int64_t lda, const void* sa, const void* b, int64_t ldb, const void* sb,
                                                                    const void* bias, const float* alpha, void* y, int64_t ldy, int64_t M, int N, int K,
                                                                    void* stream) {
  return gemm_mxfp8_impl(a, lda, sa, b, ldb, sb, bias, alpha, y, ldy, M, N, K, 0, stream);
}

// variant: 0 = automatic, 1 = 128x128-tile kernel, 2 = 256x256-tile ping-pong kernel (tests and A/B measurements)
extern "C" __attribute__((visibility("default"))) int x2v_gemm_mxfp8_variant(const void* a, int64_t lda, const void* sa, const void* b, int64_t ldb, const void* sb,
                                                                            const void* bias, const float* alpha, void* y, int64_t ldy, int64_t M, int N, int K,
                                                                            int variant, void* stream) {
  X2V_REQUIRE(variant >= 0 && variant <= 2, X2V_E_ARG, "gemm_mxfp8: unknown variant %d", variant);
  return gemm_mxfp8_impl(a, lda, sa, b, ldb, sb, bias, alpha, y, ldy, M, N, K, variant, stream);
}

// With the fused epilogues of x2v_gemm_bf16 / x2v_gemm_fp8 (X2V_EPI_*): what the block drivers call for an MXFP8 linear layer.
extern "C" __attribute__((visibility("default"))) int x2v_gemm_mxfp8_epi(const void* a, int64_t lda, const void* sa, const void* b, int64_t ldb, const void* sb,
                                                                        const void* bias, const float* alpha, void* y, int64_t ldy, int64_t M, int N, int K,
                                                                        int epilogue, const void* resid, int64_t ldr, const void* gate, void* stream) {
  return gemm_mxfp8_impl(a, lda, sa, b, ldb, sb, bias, alpha, y, ldy, M, N, K, 0, stream, epilogue, resid, ldr, gate);
}
