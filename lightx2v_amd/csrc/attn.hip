// Dense non-causal attention forward, head_dim 128, bf16 in/out, fp32 online softmax.
//
// Bound: MFMA.  Algorithmic work per launch = 4 * Sq * Sk * H * 128 FLOP (QK^T + PV); algorithmic bytes = q, k, v, o once.
//
// Two kernels (earlier generations — the register-staged v1, the x2-unrolled v3, v4, the in-phase v6 and their timing probes — live in
// git history; DESIGN.md §4.1 keeps their measurements):
//   * attn_fwd_pipe_kernel ("v2"): the general entry on row-major q/k/v (x2v_attn_fwd_bf16): what cross-attention, the text refiner and
//     the reference-rounding mode launch.  Per-wave software pipeline, K by LDS-DMA, V^T fragments by ds_read_b64_tr_b16.
//   * attn_fwd_v9_kernel ("ping-pong"): what the fused block drivers launch for self-attention (x2v_attn_fwd_bf16_vt) on a
//     pre-transposed V and a q that already carries scale*log2(e).
// Common structure:
//   * one workgroup = 8 waves x 32 query rows of ONE head; K/V tiles of 64 keys are staged once per workgroup in LDS and shared
//     by all waves; double buffered.
//   * "swapped" QK^T: S^T = K . Q^T, K fragment as the MFMA's A operand (v2: 32x32x16, a lane owns ONE query column and 32 of the tile's
//     64 keys; ping-pong: 16x16x32, a lane owns one query column in each of two groups and 16 of the 64 keys), so max, exp2, row sum and the
//     rescale of O are lane-local up to one (v2) / three (ping-pong) cross-lane swaps of the row maxima per tile.
//   * P never leaves registers: the accumulator layout of S^T IS a valid B-operand layout for the PV MFMA as long as the V^T fragment
//     enumerates keys in the same order — the reduction index of an MFMA may be permuted freely if both operands agree.
//   * LDS image of K: [64 keys][256 B] with the 16-byte chunk index XORed by a hash of the key (conflict-free ds_read_b128 over its 16-lane
//     service groups); the swizzle is applied on the per-lane DMA SOURCE address.
//   * q-block-fastest grid: co-resident workgroups walk the same head's K/V stream close together, so an XCD's L2 serves a K/V tile to
//     several of its CUs from one fill (launch_attn_vt has the measured story of the XCD-aware alternative).
#include "x2v_common.h"

namespace x2v {

constexpr int AT_D = 128;
constexpr int AT_KV = 64;
constexpr int AT_K_BYTES = AT_KV * 256;     // 16 KiB
constexpr int AT_VSUB = 2080;               // v2: bytes per V sub-tile ([64][16] bf16 = 2048 + 32 pad: conflict-free 16-byte staging writes
                                            // AND paired (sub-tile T, T+4) transpose reads)
constexpr int AT_V_BYTES = 8 * AT_VSUB;     // 16640
constexpr int AT_BUF_BYTES = AT_K_BYTES + AT_V_BYTES;
constexpr int AT_LDS_BYTES = 2 * AT_BUF_BYTES;  // 66,048 B


typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

__device__ __forceinline__ bf16x8_t tr_frag(const char* p0, const char* p1) {
  s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0));
  s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p1));
  typedef short s16x8_t __attribute__((ext_vector_type(8)));
  s16x8_t c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8_t, c);
}

// single-instruction max helpers: hipcc otherwise inserts canonicalising v_max before fmaxf on MFMA outputs
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float vmax2(float a, float b) {
  float d;
  asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

// key index (within a 32-key MFMA tile) held in accumulator register r of half-wave `hi`
__device__ __forceinline__ int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
// ------------------------------------------------------------------------------------------------
// v2: software-pipelined variant.  PMC on v1 showed per-wave VALU-active ~= MFMA-busy (1142 vs 1024 cycles per
// tile) and a per-SIMD tile time equal to their SUM over the two co-resident waves: the waves run the same
// phase in lock-step (barrier per tile), so the matrix pipe idles during everyone's softmax.  Here every wave
// interleaves the two itself: while the softmax of tile t runs on the VALU, the QK^T MFMAs of tile t+1
// (independent accumulators) are in flight; then the O rescale of dv-tile T+1 runs under the PV MFMAs of T.
//   * K is staged by LDS-DMA (global_load_lds, swizzle applied on the per-lane source address) two tiles
//     ahead, V by registers one tile ahead; both target buffers are free for the whole iteration.
//   * the half-wave max exchange uses v_permlane32_swap (VALU) instead of ds_bpermute (LDS round trip).
//   * RESCALE_THR > 0: skip the O rescale while the running max grows by <= THR (base-2 domain): P stays
//     <= 2^THR, exact in the final O/l normalisation; the branch is wave-uniform.
typedef __attribute__((address_space(3))) void* at_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* at_gbl_ptr_t;

template <int NW, int RESCALE_THR>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_pipe_kernel(const unsigned short* __restrict__ Q, int64_t ldq,
                                                                   const unsigned short* __restrict__ Kp, int64_t ldk,
                                                                   const unsigned short* __restrict__ Vp, int64_t ldv, unsigned short* __restrict__ O,
                                                                   int64_t ldo, int64_t Sq, int64_t Sk, float scale_log2e, unsigned k_bytes,
                                                                   unsigned v_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)  // device-only builtins (buffer resources): the host pass only needs the launch stub
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = NW * 64;
  constexpr int QB = NW * 32;
  constexpr int CPT = (AT_KV * 16) / NT;   // V chunks per thread per tile
  constexpr int KDMA = 16 / NW;            // K LDS-DMA wave-instructions per wave per tile (1 KiB each)
  constexpr int K_OFF = 0, V_OFF = 2 * AT_K_BYTES;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fl = lane & 31, hi = lane >> 5;
  const int head = blockIdx.y;
  const int64_t q0 = (int64_t)blockIdx.x * QB + wid * 32;
  const unsigned short* Kh = Kp + (int64_t)head * AT_D;
  const unsigned short* Vh = Vp + (int64_t)head * AT_D;

  bf16x8_t qf[8];
  {
    int64_t qr = q0 + fl;
    qr = qr < Sq ? qr : Sq - 1;
    const unsigned short* qp = Q + qr * ldq + (int64_t)head * AT_D + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp + ks * 16);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(qf[ks]));
  }

  // K by LDS-DMA, V to registers, both through raw buffer loads: per-lane byte offsets are loop-invariant, the
  // tile offset travels in an SGPR (soffset), and rows past Sk read as zero (hardware bounds check) — no
  // per-tile 64-bit address VALU, no clamping.  Descriptors are wave-uniform (kernel arguments + blockIdx).
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)Kh, 0, k_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)Vh, 0, v_bytes, 0x00020000);
  const unsigned k_tile_bytes = (unsigned)(AT_KV * ldk * 2), v_tile_bytes = (unsigned)(AT_KV * ldv * 2);
  unsigned k_voff[KDMA];
#pragma unroll
  for (int j = 0; j < KDMA; ++j) {
    const int krow = (wid * KDMA + j) * 4 + (lane >> 4);
    k_voff[j] = (unsigned)(krow * ldk * 2) + (unsigned)(((lane & 15) ^ (krow & 15)) << 4);
  }
#define AT_DMA_K(T_, BUF_)                                                                                    \
  _Pragma("unroll") for (int j = 0; j < KDMA; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(                  \
      rk, (at_lds_ptr_t)(smem + K_OFF + (BUF_) * AT_K_BYTES + (wid * KDMA + j) * 1024), 16, k_voff[j], (unsigned)(T_) * k_tile_bytes, 0, 0);
  i32x4_t vst[CPT];
  int v_wr[CPT];
  unsigned v_voff[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int id = i * NT + tid;
    const int key = id >> 4, c = id & 15;
    v_wr[i] = V_OFF + (c >> 1) * AT_VSUB + key * 32 + (c & 1) * 16;
    v_voff[i] = (unsigned)(key * ldv * 2) + (unsigned)(c << 4);
  }
#define AT_LOAD_V(T_) \
  _Pragma("unroll") for (int i = 0; i < CPT; ++i) vst[i] = __builtin_amdgcn_raw_buffer_load_b128(rv, v_voff[i], (unsigned)(T_) * v_tile_bytes, 0);
#define AT_WRITE_V(BUF_)                                                                                       \
  _Pragma("unroll") for (int i = 0; i < CPT; ++i) *reinterpret_cast<i32x4_t*>(smem + (BUF_) * AT_V_BYTES + v_wr[i]) = vst[i];

  // K fragment offsets: row (u*32+fl)*256 + ((ks*2+hi) ^ (fl&15))*16 = kaddr[ks] + u*8192 (+ buffer offset, immediate)
  int kaddr[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) kaddr[ks] = fl * 256 + (((hi ^ (fl & 15)) << 4) ^ (ks << 5));
  const int L = lane & 15;
  const int v_rd_base = ((fl >> 4) * 4) * AT_VSUB + (4 * hi + (L >> 2)) * 32 + (L & 3) * 8;

  f32x16_t oacc[4];
#pragma unroll
  for (int T = 0; T < 4; ++T)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[T][e] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const int nt = (int)((Sk + AT_KV - 1) / AT_KV);

#define AT_QK(DST_, BUF_)                                                                                     \
  {                                                                                                            \
    const char* kb_ = smem + K_OFF + (BUF_) * AT_K_BYTES;                                                      \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) _Pragma("unroll") for (int e = 0; e < 16; ++e) DST_[u][e] = 0.f; \
    _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) _Pragma("unroll") for (int u = 0; u < 2; ++u) {           \
      const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(kb_ + u * 8192 + kaddr[ks]);                      \
      DST_[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], DST_[u], 0, 0, 0);                         \
    }                                                                                                          \
  }

  // ---- prologue: K(0), K(1) by DMA; V(0) by registers; S(0)
  AT_DMA_K(0, 0)
  if (nt > 1) {
    AT_DMA_K(1, 1)
  }
  AT_LOAD_V(0)
  AT_WRITE_V(0)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  f32x16_t sc[2];
  AT_QK(sc, 0)
  // every wave must be done reading K(0) before iteration 0 re-targets K buffer 0 with the DMA of K(2)
  // (without this barrier a fast wave's DMA could land under a slow wave's prologue QK^T: rare, small errors)
  __syncthreads();

  // One tile, written as an explicit 32-slot software pipeline (the compiler's own interleave of two
  // independent streams proved unreliable): every slot = {fragment read for the NEXT slot, one MFMA, a fixed
  // chunk of VALU work}, pinned by sched_barrier(0).
  //   slots  0..15 (phase 1): MFMA = S(t+1) += K(t+1) Q^T;  VALU = softmax of S(t): 4 slots of row-max, 1 slot of
  //                           max exchange + the lazy-rescale decision, 11 slots of exp2 + bf16 pack
  //   slots 16..31 (phase 2): MFMA = O[T] += V(t)^T P^T;     VALU = row-sum adds
  // SC_/SN_ are the two score buffers (they ping-pong: the loop is unrolled x2, nothing is copied), KN_/VB_ the
  // compile-time LDS buffer indices of K(t+1) / V(t) (every LDS address = loop-invariant VGPR + immediate).
  // LAST_ = true (peeled final tile): key masking, no next-tile MFMAs, no prefetch.
#define AT_SB() __builtin_amdgcn_sched_barrier(0)
#define AT_TILE(LAST_, SC_, SN_, KN_, VB_)                                                                     \
  {                                                                                                            \
    const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  \
    if ((LAST_) && (int64_t)(t + 1) * AT_KV > Sk) {                                                            \
      const int left = (int)(Sk - (int64_t)t * AT_KV);                                                         \
      _Pragma("unroll") for (int u = 0; u < 2; ++u) _Pragma("unroll") for (int r = 0; r < 16; ++r)             \
        if (u * 32 + acc_row(r, hi) >= left) SC_[u][r] = -1e30f;                                               \
    }                                                                                                          \
    const char* kb_ = smem + K_OFF + (KN_) * AT_K_BYTES;                                                       \
    bf16x8_t kf_n = *reinterpret_cast<const bf16x8_t*>(kb_ + kaddr[0]);                                        \
    float pm[4];                                                                                               \
    unsigned pw[16];                                                                                           \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                                           \
      const int ks = i >> 1, u = i & 1;                                                                        \
      if (!(LAST_)) {                                                                                          \
        const bf16x8_t kf_c = kf_n;                                                                            \
        if (i < 15) kf_n = *reinterpret_cast<const bf16x8_t*>(kb_ + ((i + 1) & 1) * 8192 + kaddr[(i + 1) >> 1]); \
        SN_[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf_c, qf[ks], ks == 0 ? zero16 : SN_[u], 0, 0, 0);    \
      }                                                                                                        \
      if (i < 4) { /* row max of 8 scores: 3 v_max3 + 1 v_max */                                               \
        const int uu = i >> 1, r0 = (i & 1) * 8;                                                               \
        float m_ = vmax3(SC_[uu][r0], SC_[uu][r0 + 1], SC_[uu][r0 + 2]);                                       \
        m_ = vmax3(m_, SC_[uu][r0 + 3], SC_[uu][r0 + 4]);                                                      \
        m_ = vmax3(m_, SC_[uu][r0 + 5], SC_[uu][r0 + 6]);                                                      \
        pm[i] = vmax2(m_, SC_[uu][r0 + 7]);                                                                    \
      } else if (i == 4) {                                                                                     \
        float mx = vmax2(vmax3(pm[0], pm[1], pm[2]), pm[3]);                                                   \
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);    \
        mx = vmax2(__uint_as_float(sw[0]), __uint_as_float(sw[1]));                                            \
        const float ms = mx * scale_log2e;                                                                     \
        if (RESCALE_THR < 0 || __any(ms - m_run > (float)RESCALE_THR)) {                                       \
          /* lazy rescale: taken only when some row's max grew by more than THR (base-2); otherwise the old    \
             max is kept (P <= 2^THR, exact after the final O/l) and O is not touched this tile */             \
          const float m_new = fmaxf(m_run, ms);                                                                \
          const float al = __builtin_amdgcn_exp2f(m_run - m_new);                                              \
          m_run = m_new;                                                                                       \
          l_run *= al;                                                                                         \
          _Pragma("unroll") for (int T = 0; T < 4; ++T) _Pragma("unroll") for (int e = 0; e < 16; ++e) oacc[T][e] *= al; \
        }                                                                                                      \
      } else { /* slots 5..15: 3 (last: 2) elements of exp2, packed to bf16 pairs as they complete */          \
        const int e0 = (i - 5) * 3, e1 = (e0 + 3 < 32) ? e0 + 3 : 32;                                          \
        _Pragma("unroll") for (int e = e0; e < e1; ++e) {                                                      \
          SC_[e >> 4][e & 15] = __builtin_amdgcn_exp2f(SC_[e >> 4][e & 15] * scale_log2e - m_run);             \
          if (e & 1) pw[e >> 1] = pack_bf2(SC_[e >> 4][(e & 15) - 1], SC_[e >> 4][e & 15]);                    \
        }                                                                                                      \
      }                                                                                                        \
      AT_SB();                                                                                                 \
    }                                                                                                          \
    const char* vb = smem + V_OFF + (VB_) * AT_V_BYTES + v_rd_base;                                            \
    bf16x8_t vf_n = tr_frag(vb, vb + 8 * 32);                                                                  \
    _Pragma("unroll") for (int j = 0; j < 16; ++j) {                                                           \
      const int T = j >> 2, uh = j & 3; /* uh = u*2 + h2: keys (uh*16) .. +16 of the tile */                   \
      const bf16x8_t vf_c = vf_n;                                                                              \
      if (j < 15) {                                                                                            \
        const char* p0 = vb + ((j + 1) >> 2) * AT_VSUB + (((j + 1) & 3) * 16) * 32;                            \
        vf_n = tr_frag(p0, p0 + 8 * 32);                                                                       \
      }                                                                                                        \
      i32x4_t pq = {(int)pw[uh * 4 + 0], (int)pw[uh * 4 + 1], (int)pw[uh * 4 + 2], (int)pw[uh * 4 + 3]};       \
      oacc[T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf_c, __builtin_bit_cast(bf16x8_t, pq), oacc[T], 0, 0, 0); \
      l_run += SC_[j >> 3][(j & 7) * 2] + SC_[j >> 3][(j & 7) * 2 + 1];                                        \
      AT_SB();                                                                                                 \
    }                                                                                                          \
  }
  // (a x2 ping-pong unroll with compile-time buffer indices was tried: it removes the score copy below and the
  //  LDS address adds, but hipcc then spills ~80 registers in the loop — net loss; kept single-bodied.)
  f32x16_t sd[2];
  int t = 0;
  for (; t < nt - 1; ++t) {
    if (t + 2 < nt) {
      AT_DMA_K(t + 2, t & 1)
    }
    AT_LOAD_V(t + 1)
    AT_SB();
    AT_TILE(false, sc, sd, (t + 1) & 1, t & 1)
    AT_WRITE_V((t + 1) & 1)
#pragma unroll
    for (int u = 0; u < 2; ++u) sc[u] = sd[u];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  AT_TILE(true, sc, sd, (t + 1) & 1, t & 1)

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int64_t qrow = q0 + fl;
  if (qrow < Sq) {
    unsigned short* op = O + qrow * ldo + (int64_t)head * AT_D;
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int i0 = 8 * g + 4 * hi;
        const int dv = (i0 < 16) ? (16 * T + i0) : (64 + 16 * T + (i0 - 16));
        uint2 pk;
        pk.x = pack_bf2(oacc[T][4 * g + 0] * inv, oacc[T][4 * g + 1] * inv);
        pk.y = pack_bf2(oacc[T][4 * g + 2] * inv, oacc[T][4 * g + 3] * inv);
        *reinterpret_cast<uint2*>(op + dv) = pk;
      }
  }
#endif
}

#undef AT_DMA_K
#undef AT_LOAD_V
#undef AT_WRITE_V
#undef AT_QK
#undef AT_SB
#undef AT_TILE

// ------------------------------------------------------------------------------------------------
// "ping-pong" kernel (generation v9; what the fused drivers launch through x2v_attn_fwd_bf16_vt): every wave alternates a pure matrix half-step
// {QK^T(t) then PV(t-1): 64 MFMAs + their 32 LDS fragment reads} with a pure vector half-step {softmax(t): row max, exp2, row sum, bf16 pack},
// and the two waves of a SIMD (wid and wid + NW/2) run half a step apart — while one occupies the matrix pipe the other occupies the VALU port.
// In phase (the v2 pipeline above) both waves want the VALU during the softmax (exp2 is quarter rate) and both want the matrix pipe during PV;
// the measured decomposition showed the parts adding up rather than overlapping.
//   * V is consumed pre-transposed (V^T [H][S/64][128 dv][64 keys], written by the v projection's epilogue x2v_gemm_bf16_vt or by
//     x2v_transpose_heads_bf16), so both operand tiles arrive by LDS-DMA into swizzled images and both fragments are plain ds_read_b128;
//   * one barrier per half-step; data movement is by global half-step g, identical for all waves: even g = 2t issues K(t+1) and V^T(t) (both
//     double buffered), the odd half-step that follows ends with s_waitcnt vmcnt(0).  Only the waves in their vector half-step issue the DMA
//     (an LDS-DMA issue costs ~60 cycles between bare MFMAs, about half in VALU-only gaps);
//   * scores leave the MFMA already relative to the running max (C = -m_run, known before QK^T(t) starts because softmax(t-1) is the same wave's
//     previous half-step), q carries scale*log2(e) (PRESCALED: folded in by the producer), so P = exp2(S') with no per-score FMA; rescale stays
//     lazy (cold branch, threshold RESCALE_THR in base-2 units); the first MFMA of every score chain is an asm statement with an early-clobber
//     destination so that hipcc never computes into (and then restores) the -m tuple;
//   * a straight-line loop per role (early / late waves), unrolled x2 so both LDS buffer indices are compile-time and every fragment address is
//     register + immediate; one score buffer: nothing spills at 2 waves per SIMD (224 VGPRs).
// Register geometry — v_mfma_f32_16x16x32_bf16.  At the board's power limit the 16x16x32 shape delivers more FLOP per joule than 32x32x16
// (tools/probes/mfma_power_probe: 1582 vs 1529-1540 TFLOP/s with attention's fragment traffic at 32 rows per wave), the same effect that moved
// the GEMMs (gemm256.hip): the round-2 kernel (same protocol on 32x32x16, 32 MFMAs per tile) ran 4.8 % slower on the same box
// (profiles/r03_attn_v9_map_rot_matrix_and_pmc.txt; its source is in git history).
//   * a wave's 32 query rows are two groups g of 16; lane = (c = lane & 15, qd = lane >> 4).
//   * S^T = K . Q^T per 16-key sub-tile kt (4 per tile): A = K fragment (row = key kappa(kt, c), k-slot 8 qd + e <-> d = 32 ks + 8 qd + e),
//     B = Q fragment (col = query 16 g + c, same k-slots): 4 kt x 2 g x 4 ks = 32 MFMAs; a K fragment read feeds both groups.
//     The output lane (c, qd) holds rows 4 qd + r of sub-tile kt, i.e. keys kappa(kt, 4 qd + r).
//   * kappa(2 j + b, i) = 32 j + 8 (i >> 2) + 4 b + (i & 3): lane (c, qd) then owns, over the sub-tile pair (2 j, 2 j + 1), the 8 CONSECUTIVE keys
//     32 j + 8 qd + e — exactly the k-slots of a 16-byte V^T fragment (row dv, keys 32 j + 8 qd ..), so P stays in registers as the B operand
//     of O^T += V^T . P: 8 dv tiles x 2 g x 2 j = 32 MFMAs, a V^T fragment read feeds both groups.  64 half-size MFMAs per tile for the
//     32 of v8, the same 32 ds_read_b128.
//   * K's LDS image keeps [64 keys][256 B] but its 16-byte chunk index is XORed with h(key) = (key & 3) | ((key >> 1) & 12): the 16 rows a
//     16-lane group reads are kappa(kt, 0..15), whose h values are exactly 0..15 (h(kappa(kt, c)) = c) — conflict-free; (key & 15), v8's
//     hash, would collide two-fold.  The DMA pieces of a wave are chosen so that h is the same for all of them (one voffset register).
//   * a query's 64 scores of a tile sit in 4 lanes (the 4 qd of its column): row maxima of both groups cross lanes with 3 swaps
//     (permlane16_swap, permlane32_swap, permlane16_swap) and 2 max.
template <int NW, int RESCALE_THR, bool PRESCALED, bool ROT>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_v9_kernel(const unsigned short* __restrict__ Q, int64_t ldq,
                                                                   const unsigned short* __restrict__ Kp, int64_t ldk,
                                                                   const unsigned short* __restrict__ VTp, int64_t ldvt, unsigned short* __restrict__ O,
                                                                   int64_t ldo, int64_t Sq, int64_t Sk, float scale_log2e, unsigned k_bytes,
                                                                   unsigned v_bytes, AttnBatch bs) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(NW == 8, "DMA piece assignment below is written for 4 issuing waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // Work mapping.  The launch grid is (query blocks, heads, sequences) and the dispatcher hands workgroup number L = x + gx (y + gy z) to XCD
  // L % 8 (observed placement; a speed assumption only, MI355X_MICROARCH.md "Workgroup dispatch").  Taken as it comes, the 32 workgroups
  // resident on an XCD (one per CU: 224 VGPRs = two waves per SIMD = the workgroup's eight waves) belong to ~1-2 heads whose K / V^T streams all eight 4 MiB L2s pull through together.  XCD-aware form
  // (bit 0 of bs.xcd_remap): XCD c owns the contiguous range c of the (sequence, head, query block) list (bijective for any count), so
  // its resident workgroups are consecutive query blocks of ONE head walking the same K / V^T tiles at about the same time.  Measured
  // (profiles/r03_attn_*): L2 hit rate 73-79 % -> 96 %, L2<->fabric traffic 99-123 GB -> 16 GB per Wan-14B 720p launch — and the launch is
  // 4-8 % SLOWER there, with any key-walk rotation and with K stored head-contiguous as well: eight XCDs on eight different heads keep
  // 8 x 38.7 MB of K / V^T in flight, more than the 256 MB Infinity Cache holds, while the plain grid keeps the whole chip on ~2 heads
  // (77 MB, cache-resident) and is not bound by fabric traffic to begin with.  The mapping pays where the heads in flight do fit (Ulysses rank
  // of the same video, 5 heads: +4 %; Wan-1.3B 480p: +1.5 %): the launcher's rule (launch_attn_vt) turns it on exactly there.
  int qblk = blockIdx.x, head = blockIdx.y, seq = blockIdx.z;
  if (bs.xcd_remap & 1) {
    const unsigned gx = gridDim.x, gy = gridDim.y;
    const unsigned nwg = gx * gy * gridDim.z;
    const unsigned L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned xcd = L & 7u, q8 = nwg >> 3, r8 = nwg & 7u;
    const unsigned id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (L >> 3);
    qblk = (int)(id % gx);
    const unsigned hz = id / gx;
    head = (int)(hz % gy);
    seq = (int)(hz / gy);
  }
  Q += (int64_t)seq * bs.q;
  Kp += (int64_t)seq * bs.k;
  VTp += (int64_t)seq * bs.vt;
  O += (int64_t)seq * bs.o;
  constexpr int K_OFF = 0, V_OFF = 2 * AT_K_BYTES;
  // (static wave priorities, a deeper fragment queue, the late waves' first fragments read in front of their barrier, and bare s_barrier instead of
  //  __syncthreads() were all measured in round 5 and move nothing: HISTORY.md §R5.)
  constexpr int DEPTH = 4;  // fragment reads in flight ahead of their MFMAs
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, qd = lane >> 4;
  const int64_t q0 = (int64_t)qblk * (NW * 32) + wid * 32;
  const unsigned short* Kh = Kp + (int64_t)head * AT_D;
  const unsigned short* Vh = VTp + (int64_t)head * AT_D * ldvt;

  bf16x8_t qf[2][4];  // [query group][k-step of 32 head-dim values]
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    int64_t qr = q0 + 16 * g + c16;
    qr = qr < Sq ? qr : Sq - 1;
    const unsigned short* qp = Q + qr * ldq + (int64_t)head * AT_D + qd * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(qp + ks * 32);
      if constexpr (!PRESCALED) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (__bf16)((float)v[e] * scale_log2e);
      }
      qf[g][ks] = v;
    }
  }
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[g][ks]));

  // LDS-DMA, 1 KiB pieces (4 K rows / 8 V^T rows), issued by the upper half of the waves during their vector half-step (as v8).  K piece pc
  // holds rows 4 pc .. 4 pc + 3; wave wl takes pc = 2 wl + b0 + 8 b3 (b0, b3 in {0,1}): for all of them h(row) = (row & 3) | (wl << 2), so ONE
  // per-lane offset serves its four pieces; the piece's rows travel in the scalar offset.
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)Kh, 0, k_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)Vh, 0, v_bytes, 0x00020000);
  constexpr int NDW = NW / 2;
  constexpr int NPC = 16 / NDW;
  const int wl = wid & (NW / 2 - 1);
  const unsigned k_tile_bytes = (unsigned)(AT_KV * ldk * 2), k_row_bytes = (unsigned)(ldk * 2), v_piece_bytes = (unsigned)(8 * NDW * 128);
  const int krr = lane >> 4, vrow_w = wl * 8 + (lane >> 3);
  const unsigned k_voff = (unsigned)((8 * wl + krr) * ldk * 2) + (unsigned)(((lane & 15) ^ (krr | (wl << 2))) << 4);
  const unsigned v_voff = (unsigned)(vrow_w * 128) + (unsigned)(((lane & 7) ^ ((vrow_w >> 1) & 7)) << 4);
#define A9_DMA_K(SOFF_, BUF_)                                                                                                            \
  _Pragma("unroll") for (int j = 0; j < NPC; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(                                            \
      rk, (at_lds_ptr_t)(smem + K_OFF + (BUF_) * AT_K_BYTES + ((wl << 1) | (j & 1) | ((j >> 1) << 3)) * 1024), 16, k_voff,              \
      (SOFF_) + (unsigned)(4 * (j & 1) + 32 * (j >> 1)) * k_row_bytes, 0, 0);
#define A9_DMA_V(SOFF_, BUF_)                                                                                                            \
  _Pragma("unroll") for (int j = 0; j < NPC; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(                                            \
      rv, (at_lds_ptr_t)(smem + V_OFF + (BUF_) * AT_K_BYTES + (wl + NDW * j) * 1024), 16, v_voff, (SOFF_) + j * v_piece_bytes, 0, 0);

  // fragment offsets.  K: row kappa(kt, c) = 8 (c >> 2) + (c & 3) [+ 32 j + 4 b as an immediate], chunk (4 ks + qd) ^ c.  V^T: row 16 T + c
  // [16 T rows as an immediate], chunk (4 j + qd) ^ ((c >> 1) & 7).
  int kbase[4], vbase[2];
  const int krow_rd = 8 * (c16 >> 2) + (c16 & 3);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kbase[ks] = krow_rd * 256 + ((((ks << 2) | qd) ^ c16) << 4);
#pragma unroll
  for (int j = 0; j < 2; ++j) vbase[j] = c16 * 128 + ((((j << 2) | qd) ^ ((c16 >> 1) & 7)) << 4);

  f32x4_t oacc[8][2], sc[4][2], negm[2];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      negm[g][e] = 0.f;
#pragma unroll
      for (int T = 0; T < 8; ++T) oacc[T][g][e] = 0.f;
    }
  float m_run[2] = {0.f, 0.f};
  float lsum[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  bool force = true;  // first tile: adopt its row max in either direction
  unsigned pw[2][2][4];  // [key group j][query group g]: 8 bf16 probabilities = the B operand of a PV MFMA
  const int nt = (int)((Sk + AT_KV - 1) / AT_KV);
  // Key-tile walk.  With ROT (flag X2V_ATTN_VT_STAGGER) query block b starts (b mod 8) tiles in: co-resident workgroups then ask for a tile at
  // slightly different times (one takes the L2 miss, its followers hit) instead of all at the same instant: +1.3 % on the plain grid at Wan-14B
  // 720p in 2-step runs, -0.9 % at sustained load (round 4, A-B-A-B step times): opt-in, no driver sets it now.  Only the first nt - 1 (full) tiles rotate — logical tile t < nt - 1 is physical tile (t + rot) mod (nt - 1) — and the physical last
  // tile, the one that may hold fewer than 64 keys, stays last, so the key mask lives in the final softmax only (a mask test in every tile's
  // softmax costs 37 spilled SGPRs).  The walk is kept as running byte offsets of the next K / V^T tile to fetch.  rot depends on (qblk, nt)
  // only: a query row's result does not depend on how a launch is batched or mapped, but it does on which 256-row block of the launch it is in.
  int rot = ROT ? (qblk & 7) : 0;
  if (rot >= nt - 1) rot = 0;
  rot = __builtin_amdgcn_readfirstlane(rot);
  const unsigned k_last = (unsigned)(nt - 1) * k_tile_bytes, v_last = (unsigned)(nt - 1) * AT_K_BYTES;
  unsigned kso = (unsigned)rot * k_tile_bytes, vso = (unsigned)rot * AT_K_BYTES;
#define A9_NEXT_K() if constexpr (ROT) { kso += k_tile_bytes; kso = kso == k_last ? 0u : kso; }
#define A9_NEXT_V() if constexpr (ROT) { vso += AT_K_BYTES; vso = vso == v_last ? 0u : vso; }
#define A9_KOFF(T_) (ROT ? ((T_) == nt - 1 ? k_last : kso) : (unsigned)(T_) * k_tile_bytes)
#define A9_VOFF(T_) (ROT ? ((T_) == nt - 1 ? v_last : vso) : (unsigned)(T_) * AT_K_BYTES)

#define A9_SB() __builtin_amdgcn_sched_barrier(0)
  // matrix half-step: fragment slots n = 0..15 (K: k-step n >> 2, sub-tile n & 3) and 16..31 (V^T: key group (n - 16) >> 3, dv tile (n - 16) & 7),
  // each fragment read DEPTH slots ahead and consumed by the two MFMAs of its slot (query groups 0 and 1).
#define A9_FRAG(N_)                                                                                                      \
  ((N_) < 16 ? *reinterpret_cast<const bf16x8_t*>(kb_ + (((N_) & 3) >> 1) * 8192 + ((N_) & 1) * 1024 + kbase[(N_) >> 2]) \
             : *reinterpret_cast<const bf16x8_t*>(vb + (((N_) - 16) & 7) * 2048 + vbase[((N_) - 16) >> 3]))
  bf16x8_t fr[DEPTH];
#define A9_MATRIX(N0_, N1_, VB_, KB_)                                                                                   \
  {                                                                                                                      \
    const char* vb = smem + V_OFF + (VB_) * AT_K_BYTES;                                                                  \
    const char* kb_ = smem + K_OFF + (KB_) * AT_K_BYTES;                                                                 \
    __builtin_amdgcn_s_setprio(1);                                                                                       \
    _Pragma("unroll") for (int d = 0; d < DEPTH; ++d) fr[d] = A9_FRAG((N0_) + d);                                        \
    _Pragma("unroll") for (int n = (N0_); n < (N1_); ++n) {                                                              \
      const bf16x8_t f_ = fr[(n - (N0_)) % DEPTH];                                                                       \
      if (n < 16) {                                                                                                      \
        const int ks = n >> 2, kt = n & 3;                                                                               \
        _Pragma("unroll") for (int g = 0; g < 2; ++g) {                                                                  \
          if (ks == 0) /* D early-clobber: the -m tuple stays a pure input */                                            \
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(sc[kt][g]) : "v"(f_), "v"(qf[g][0]), "v"(negm[g])); \
          else                                                                                                           \
            sc[kt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f_, qf[g][ks], sc[kt][g], 0, 0, 0);                       \
        }                                                                                                                \
      } else {                                                                                                           \
        const int T = (n - 16) & 7, j = (n - 16) >> 3;                                                                   \
        _Pragma("unroll") for (int g = 0; g < 2; ++g) {                                                                  \
          i32x4_t pq = {(int)pw[j][g][0], (int)pw[j][g][1], (int)pw[j][g][2], (int)pw[j][g][3]};                         \
          oacc[T][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f_, __builtin_bit_cast(bf16x8_t, pq), oacc[T][g], 0, 0, 0); \
        }                                                                                                                \
      }                                                                                                                  \
      if (n + DEPTH < (N1_)) fr[(n - (N0_)) % DEPTH] = A9_FRAG(n + DEPTH);                                               \
      A9_SB();                                                                                                           \
    }                                                                                                                    \
    __builtin_amdgcn_s_setprio(0);                                                                                       \
  }
  // vector half-step: softmax of the tile in sc -> packed bf16 P in pw.  Register r of sc[kt][g] is key 32 (kt >> 1) + 8 qd + 4 (kt & 1) + r.
#define A9_SOFTMAX(LAST_)                                                                                                \
  {                                                                                                                      \
    if ((LAST_) && (int64_t)(t + 1) * AT_KV > Sk) {                                                                      \
      const int left = (int)(Sk - (int64_t)t * AT_KV);                                                                   \
      _Pragma("unroll") for (int kt = 0; kt < 4; ++kt) _Pragma("unroll") for (int r = 0; r < 4; ++r)                     \
        if (32 * (kt >> 1) + 8 * qd + 4 * (kt & 1) + r >= left) { sc[kt][0][r] = -1e30f; sc[kt][1][r] = -1e30f; }        \
    }                                                                                                                    \
    float mg[2];                                                                                                         \
    _Pragma("unroll") for (int g = 0; g < 2; ++g) {                                                                      \
      const float a_ = vmax3(sc[0][g][0], sc[0][g][1], sc[0][g][2]), b_ = vmax3(sc[0][g][3], sc[1][g][0], sc[1][g][1]);   \
      const float c_ = vmax3(sc[1][g][2], sc[1][g][3], sc[2][g][0]), d_ = vmax3(sc[2][g][1], sc[2][g][2], sc[2][g][3]);   \
      const float e_ = vmax3(sc[3][g][0], sc[3][g][1], sc[3][g][2]);                                                     \
      mg[g] = vmax3(vmax3(a_, b_, c_), vmax3(d_, e_, sc[3][g][3]), -3.0e38f);                                            \
    }                                                                                                                    \
    { /* across the four lanes (qd) of a query column, both groups at once: rows {A01, B01, A23, B23} -> {A, B, A, B} -> A | B */ \
      auto s1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(mg[0]), __float_as_uint(mg[1]), false, false);         \
      const float c1 = vmax2(__uint_as_float(s1[0]), __uint_as_float(s1[1]));                                            \
      auto s2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(c1), __float_as_uint(c1), false, false);               \
      const float c2 = vmax2(__uint_as_float(s2[0]), __uint_as_float(s2[1]));                                            \
      auto s3 = __builtin_amdgcn_permlane16_swap(__float_as_uint(c2), __float_as_uint(c2), false, false);               \
      mg[0] = __uint_as_float(s3[0]);                                                                                    \
      mg[1] = __uint_as_float(s3[1]);                                                                                    \
    }                                                                                                                    \
    if (force || __any(vmax2(mg[0], mg[1]) > (float)RESCALE_THR)) { /* cold: some row's max grew by more than THR (or first tile) */ \
      _Pragma("unroll") for (int g = 0; g < 2; ++g) {                                                                    \
        const float d = force ? mg[g] : fmaxf(mg[g], 0.f);                                                               \
        m_run[g] += d;                                                                                                   \
        if (!force) {                                                                                                    \
          const float al = __builtin_amdgcn_exp2f(-d);                                                                   \
          lsum[g][0] *= al;                                                                                              \
          lsum[g][1] *= al;                                                                                              \
          _Pragma("unroll") for (int T = 0; T < 8; ++T) _Pragma("unroll") for (int e = 0; e < 4; ++e) oacc[T][g][e] *= al; \
        }                                                                                                                \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) negm[g][e] = -m_run[g];                                            \
        _Pragma("unroll") for (int kt = 0; kt < 4; ++kt) _Pragma("unroll") for (int e = 0; e < 4; ++e) sc[kt][g][e] -= d; \
      }                                                                                                                  \
      force = false;                                                                                                     \
    }                                                                                                                    \
    _Pragma("unroll") for (int g = 0; g < 2; ++g) _Pragma("unroll") for (int kt = 0; kt < 4; ++kt) _Pragma("unroll") for (int e = 0; e < 4; e += 2) { \
      const float p0 = __builtin_amdgcn_exp2f(sc[kt][g][e]), p1 = __builtin_amdgcn_exp2f(sc[kt][g][e + 1]);             \
      lsum[g][0] += p0;                                                                                                  \
      lsum[g][1] += p1;                                                                                                  \
      pw[kt >> 1][g][2 * (kt & 1) + (e >> 1)] = pack_bf2(p0, p1);                                                        \
    }                                                                                                                    \
    { /* the row sums belong to THIS half-step */                                                                        \
      asm volatile("" : "+v"(lsum[0][0]), "+v"(lsum[0][1]), "+v"(lsum[1][0]), "+v"(lsum[1][1]));                         \
      A9_SB();                                                                                                           \
    }                                                                                                                    \
  }

  // ---- prologue: K(0)
  if (wid >= NW / 2) {
    A9_DMA_K(A9_KOFF(0), 0)
    A9_NEXT_K()
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int t = 0;
#define A9_ISSUE(TG_) /* even half-step g = 2 TG_ */  \
  if ((TG_) + 1 < nt) {                               \
    A9_DMA_K(A9_KOFF((TG_) + 1), ((TG_) + 1) & 1)     \
    A9_NEXT_K()                                       \
  }                                                   \
  if ((TG_) < nt) {                                   \
    A9_DMA_V(A9_VOFF(TG_), (TG_) & 1)                 \
    A9_NEXT_V()                                       \
  }                                                   \
  A9_SB();
  // __syncthreads() carries a workgroup-scope release fence, for which hipcc drains every outstanding VMEM operation of the wave — LDS-DMA pieces
  // included — in front of the barrier: the late waves' pieces of an even half-step land within that half-step and the explicit wait behind the odd
  // half-step is a no-op.  The bare instruction (pieces in flight through the odd half-step) runs at the same speed (HISTORY.md §R5), so the fenced form
  // stays: the compiler — not a clobber list — owns the LDS ordering.
#define A9_BARRIER() __syncthreads();
#define A9_BAR_EVEN() A9_BARRIER()
#define A9_BAR_ODD()                                   \
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     \
  A9_BARRIER()
  if (wid < NW / 2) {
    A9_MATRIX(0, 16, 0, 0)
    A9_BAR_EVEN()
    while (t < nt - 1) {
      A9_SOFTMAX(false)
      A9_BAR_ODD()
      A9_MATRIX(0, 32, 0, 1)
      A9_BAR_EVEN()
      if (++t >= nt - 1) break;
      A9_SOFTMAX(false)
      A9_BAR_ODD()
      A9_MATRIX(0, 32, 1, 0)
      A9_BAR_EVEN()
      ++t;
    }
    A9_SOFTMAX(true)
    A9_BAR_ODD()
    A9_MATRIX(16, 32, t & 1, 0)
    A9_BAR_EVEN()
    A9_BAR_ODD()
  } else {
    A9_ISSUE(0)
    A9_BAR_EVEN()
    A9_MATRIX(0, 16, 0, 0)
    A9_BAR_ODD()
    while (t < nt - 1) {
      A9_ISSUE(t + 1)
      A9_SOFTMAX(false)
      A9_BAR_EVEN()
      A9_MATRIX(0, 32, 0, 1)
      A9_BAR_ODD()
      if (++t >= nt - 1) break;
      A9_ISSUE(t + 1)
      A9_SOFTMAX(false)
      A9_BAR_EVEN()
      A9_MATRIX(0, 32, 1, 0)
      A9_BAR_ODD()
      ++t;
    }
    A9_SOFTMAX(true)
    A9_BAR_EVEN()
    A9_MATRIX(16, 32, t & 1, 0)
    A9_BAR_ODD()
  }
#undef A9_ISSUE
#undef A9_NEXT_K
#undef A9_NEXT_V
#undef A9_KOFF
#undef A9_VOFF
#undef A9_BAR_EVEN
#undef A9_BARRIER
#undef A9_BAR_ODD
  // (A9_SOFTMAX, A9_MATRIX, A9_FRAG, A9_SB, A9_DMA_K, A9_DMA_V stay defined for the persistent form below)

#pragma unroll
  for (int g = 0; g < 2; ++g) {
    float l_run = lsum[g][0] + lsum[g][1];
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_run;
    const int64_t qrow = q0 + 16 * g + c16;
    if (qrow < Sq) {
      unsigned short* op = O + qrow * ldo + (int64_t)head * AT_D + 4 * qd;
#pragma unroll
      for (int T = 0; T < 8; ++T) {
        uint2 pk;
        pk.x = pack_bf2(oacc[T][g][0] * inv, oacc[T][g][1] * inv);
        pk.y = pack_bf2(oacc[T][g][2] * inv, oacc[T][g][3] * inv);
        *reinterpret_cast<uint2*>(op + 16 * T) = pk;
      }
    }
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// Persistent short-walk form of the ping-pong kernel ("p9").  Cross-attention (Wan-14B 720p: 75 600 query rows x 512 text keys x 40 heads) is 11 840
// walks of 8 key tiles.  As one-walk workgroups (v9 above) every walk exposes its prologue (q rows and K(0) from memory), its tail (PV of the last tile,
// normalise, store) and the workgroup turnover: 1.02 ms per launch = 0.31 of the bf16 peak against 0.56 on the 1182-tile self-attention walk
// (profiles/r05_final_rocprof_kernel_stats_wan14b_720p.csv).  Here a workgroup takes every gridDim.x-th item of the (sequence, head, query block)
// list and runs them as ONE tile stream through the same half-step protocol:
//   * tile tau of the stream is key tile tau mod nt of block tau / nt; the K / V^T double buffers, the DMA cadence (K(tau + 2) and V^T(tau + 1) issued
//     by the late waves under softmax(tau)) and the two-role schedule do not notice block boundaries — the first QK^T of block b + 1 shares its matrix
//     half-step with the last PV of block b;
//   * the q rows of block b + 1 arrive by LDS-DMA (issued by the EARLY waves in the walk of block b, pieces spread over the vector half-steps of tiles
//     1 .. nt-3, into a [256 rows][256 B] image whose 16-byte chunk index is XORed with row & 15: conflict-free fragment reads) and are read into the
//     fragment registers behind softmax(nt - 1) of block b, when the last QK^T of block b has issued;
//   * block b's output is normalised and stored in front of softmax(0) of block b + 1 — a vector half-step, the other role's waves are in their matrix
//     half-step — and the accumulators, row sums and running maxima start again from v9's initial state.
// Every query row sees the arithmetic of v9 in v9's order (same 256-row blocks, same wave -> row mapping, same lazy-rescale decisions): bit-identical
// output (tests/test_gpu_ops.py::test_attention_persistent_short_walk).  Whole key tiles only (Sk % 64 == 0: no mask), 4 <= nt: the launcher's rule (attn_vt_plan bit 9).
typedef unsigned int at_u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int at_u32x4_t __attribute__((ext_vector_type(4)));
struct P9Item {
  int qblk, head, seq;
};

template <int RESCALE_THR, bool PRESCALED>
__global__ __launch_bounds__(512, 2) void attn_fwd_p9_kernel(const unsigned short* __restrict__ Q, int64_t ldq, const unsigned short* __restrict__ Kp, int64_t ldk,
                                                               const unsigned short* __restrict__ VTp, int64_t ldvt, unsigned short* __restrict__ O, int64_t ldo,
                                                               int64_t Sq, int nt, float scale_log2e, unsigned k_bytes, unsigned v_bytes, AttnBatch bs, int nqb, int H,
                                                               unsigned n_items) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NW = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int K_OFF = 0, V_OFF = 2 * AT_K_BYTES, Q_OFF = 4 * AT_K_BYTES;  // K x 2 | V^T x 2 | the next block's q rows (64 KiB)
  constexpr int DEPTH = 4;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, qd = lane >> 4;
  // Work: items blockIdx.x, + gridDim.x, + 2 gridDim.x, .. of the (sequence, head, query block) list — at any time the workgroups of the chip are on
  // ~gridDim.x consecutive query blocks, i.e. one or two heads, whose K / V^T (256 KB per head at 512 keys) every XCD's L2 then holds, as with v9's
  // plain grid.  (Contiguous ranges per workgroup put all 40 heads in flight at once: 10 MB of K / V^T through each 4 MB L2 — measured 20-25 % more
  // time per tile than v9, profiles/r06_call15_*.)
  if (blockIdx.x >= n_items) return;
  const int64_t Sk = (int64_t)nt * AT_KV;  // whole tiles: A9_SOFTMAX's key mask (LAST_) is never compiled in
  const int t = 0;
  (void)Sk;
  (void)t;
  const int dq = (int)(gridDim.x % (unsigned)nqb), dh = (int)(gridDim.x / (unsigned)nqb);
  auto item_next = [&](P9Item& it) {
    it.qblk += dq;
    it.head += dh;
    if (it.qblk >= nqb) {
      it.qblk -= nqb;
      ++it.head;
    }
    while (it.head >= H) {
      it.head -= H;
      ++it.seq;
    }
  };
  auto item_prev = [&](P9Item& it) {
    it.qblk -= dq;
    it.head -= dh;
    if (it.qblk < 0) {
      it.qblk += nqb;
      --it.head;
    }
    while (it.head < 0) {
      it.head += H;
      --it.seq;
    }
  };
  P9Item cur;
  {
    const unsigned hz = blockIdx.x / (unsigned)nqb;
    cur.qblk = (int)(blockIdx.x - hz * (unsigned)nqb);
    cur.head = (int)(hz % (unsigned)H);
    cur.seq = (int)(hz / (unsigned)H);
  }
  const unsigned n_mine = (n_items - blockIdx.x + gridDim.x - 1) / gridDim.x;  // >= 1

  // q rows of the FIRST block: memory -> registers (as v9); every later block's come through the LDS image
  bf16x8_t qf[2][4];
  {
    const unsigned short* Qb = Q + (int64_t)cur.seq * bs.q + (int64_t)cur.head * AT_D;
    const int64_t q0 = (int64_t)cur.qblk * (NW * 32) + wid * 32;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      int64_t qr = q0 + 16 * g + c16;
      qr = qr < Sq ? qr : Sq - 1;
      const unsigned short* qp = Qb + qr * ldq + qd * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(qp + ks * 32);
        if constexpr (!PRESCALED) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (__bf16)((float)v[e] * scale_log2e);
        }
        qf[g][ks] = v;
      }
    }
  }
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[g][ks]));

  // LDS-DMA pieces and fragment offsets: v9's
  constexpr int NDW = NW / 2;
  constexpr int NPC = 16 / NDW;
  const int wl = wid & (NW / 2 - 1);
  const unsigned k_tile_bytes = (unsigned)(AT_KV * ldk * 2), k_row_bytes = (unsigned)(ldk * 2), v_piece_bytes = (unsigned)(8 * NDW * 128);
  const int krr = lane >> 4, vrow_w = wl * 8 + (lane >> 3);
  const unsigned k_voff = (unsigned)((8 * wl + krr) * ldk * 2) + (unsigned)(((lane & 15) ^ (krr | (wl << 2))) << 4);
  const unsigned v_voff = (unsigned)(vrow_w * 128) + (unsigned)(((lane & 7) ^ ((vrow_w >> 1) & 7)) << 4);
  // q piece j of late wave wl: rows 16 j + 4 wl + (lane >> 4) of the block (row & 15 = 4 wl + (lane >> 4) for all of them: one per-lane offset)
  const unsigned q_voff = (unsigned)((4 * wl + krr) * ldq * 2) + (unsigned)(((lane & 15) ^ (4 * wl + krr)) << 4);
  const unsigned q_row16_bytes = (unsigned)(16 * ldq * 2);
  int kbase[4], vbase[2];
  const int krow_rd = 8 * (c16 >> 2) + (c16 & 3);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kbase[ks] = krow_rd * 256 + ((((ks << 2) | qd) ^ c16) << 4);
#pragma unroll
  for (int j = 0; j < 2; ++j) vbase[j] = c16 * 128 + ((((j << 2) | qd) ^ ((c16 >> 1) & 7)) << 4);
  const unsigned o_row_bytes = (unsigned)(ldo * 2), o_row16_bytes = 16u * o_row_bytes;
  // per-lane offsets of the block hand-over are recomputed where they are used, from a lane index the compiler cannot see through: hoisted out of the
  // loop they are eight more live registers, and the loop is at the 256-register limit of two waves per SIMD
#define P9_LANE()                      \
  int ln_ = lane;                      \
  asm volatile("" : "+v"(ln_));        \
  const int c_ = ln_ & 15, q_ = ln_ >> 4;

  f32x4_t oacc[8][2], sc[4][2], negm[2];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      negm[g][e] = 0.f;
#pragma unroll
      for (int T = 0; T < 8; ++T) oacc[T][g][e] = 0.f;
    }
  float m_run[2] = {0.f, 0.f};
  float lsum[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  bool force = true;
  unsigned pw[2][2][4];
  bf16x8_t fr[DEPTH];

  // ---- the end of the stream: normalise and store the last block's rows from registers (buffer stores: one loop-invariant per-lane offset, rows
  //      past Sq fall outside the descriptor's range and are dropped)
#define P9_OUT_DESC()                                                                                                    \
  P9Item pv = cur;                                                                                                       \
  item_prev(pv);                                                                                                         \
  const int64_t r0_ = (int64_t)pv.qblk * (NW * 32);                                                                      \
  const int64_t rows_ = Sq - r0_ < (int64_t)(NW * 32) ? Sq - r0_ : (int64_t)(NW * 32);                                   \
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(                                                   \
      (void*)(O + (int64_t)pv.seq * bs.o + (int64_t)pv.head * AT_D + r0_ * ldo), 0, (unsigned)((rows_ - 1) * ldo * 2 + AT_D * 2), 0x00020000);
#define P9_STORE()                                                                                                       \
  {                                                                                                                      \
    P9_OUT_DESC()                                                                                                        \
    P9_LANE()                                                                                                            \
    const unsigned o_voff = (unsigned)(wid * 32 + c_) * o_row_bytes + (unsigned)(8 * q_);                                \
    _Pragma("unroll") for (int g = 0; g < 2; ++g) {                                                                      \
      float l_run = lsum[g][0] + lsum[g][1];                                                                             \
      l_run += __shfl_xor(l_run, 16, 64);                                                                                \
      l_run += __shfl_xor(l_run, 32, 64);                                                                                \
      const float inv = 1.0f / l_run;                                                                                    \
      _Pragma("unroll") for (int T = 0; T < 8; ++T) {                                                                    \
        at_u32x2_t pk;                                                                                                   \
        pk.x = pack_bf2(oacc[T][g][0] * inv, oacc[T][g][1] * inv);                                                       \
        pk.y = pack_bf2(oacc[T][g][2] * inv, oacc[T][g][3] * inv);                                                       \
        __builtin_amdgcn_raw_buffer_store_b64(pk, ro, o_voff, (unsigned)g * o_row16_bytes + 32u * T, 0);                  \
      }                                                                                                                  \
    }                                                                                                                    \
  }
  // ---- block hand-over inside the stream: every wave normalises and stores its rows from registers in front of softmax(0) of the next block, then
  //      starts again from v9's initial accumulator state.  (An LDS-staged form — all waves write the block into the q image, the early role
  //      flushes whole 256-byte rows, the late role's memory counter never sees a store — ran at 0.80 ms against this form's 0.85 on the Wan-14B
  //      launch but returned wrong rows for the second query group of every wave, with every piece of it correct in isolation; not shipped:
  //      profiles/r06_cross_attn_persistent_form.txt.)
#define P9_HANDOVER()                                                                                                    \
  {                                                                                                                      \
    P9_STORE()                                                                                                           \
    _Pragma("unroll") for (int g = 0; g < 2; ++g) {                                                                      \
      lsum[g][0] = 0.f;                                                                                                  \
      lsum[g][1] = 0.f;                                                                                                  \
      _Pragma("unroll") for (int T = 0; T < 8; ++T) _Pragma("unroll") for (int e = 0; e < 4; ++e) oacc[T][g][e] = 0.f;   \
    }                                                                                                                    \
    A9_SB();                                                                                                             \
  }
  // behind softmax(nt - 1): the next block starts from running maximum 0 (first tile: adopt), its q fragments come out of the LDS image
#define P9_NEXT_BLOCK()                                                                                                  \
  {                                                                                                                      \
    _Pragma("unroll") for (int g = 0; g < 2; ++g) {                                                                      \
      m_run[g] = 0.f;                                                                                                    \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) negm[g][e] = 0.f;                                                    \
    }                                                                                                                    \
    force = true;                                                                                                        \
    if (left > 1) {                                                                                                      \
      P9_LANE()                                                                                                          \
      const char* const qr_ = smem + Q_OFF + (wid * 32 + c_) * 256; /* + 4096 g, chunk (4 ks + qd) ^ c16 */              \
      _Pragma("unroll") for (int g = 0; g < 2; ++g) _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                   \
        bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(qr_ + g * 4096 + ((((ks << 2) | q_) ^ c_) << 4));                \
        if constexpr (!PRESCALED) {                                                                                      \
          _Pragma("unroll") for (int e = 0; e < 8; ++e) v[e] = (__bf16)((float)v[e] * scale_log2e);                      \
        }                                                                                                                \
        qf[g][ks] = v;                                                                                                   \
      }                                                                                                                  \
      _Pragma("unroll") for (int g = 0; g < 2; ++g) _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[g][ks])); \
    }                                                                                                                    \
    A9_SB();                                                                                                             \
  }
  // One form for every tile: behind the stream's last tile the QK^T half multiplies whatever the idle K buffer holds into the dead score registers
  // (32 MFMAs per workgroup; a second, PV-only form of the half-step in the loop costs 100+ spilled registers).
#define P9_MATRIX_STEP(P_) A9_MATRIX(0, 32, P_, (P_) ^ 1)
#define P9_ADVANCE()     \
  if (++tc == nt) {      \
    tc = 0;              \
    have_prev = true;    \
    item_next(cur);      \
    --left;              \
  }
  // bare barrier: __syncthreads() carries a release fence for which hipcc drains every LDS-DMA piece of the wave (v9's note).  LDS ordering by hand:
  // fragment reads are consumed by the half-step's MFMAs before the barrier that frees their buffer; DMA pieces are waited for (vmcnt) by the
  // issuing wave in front of the barrier that publishes them; the q image is read into registers that the next half-step's MFMAs consume.
#define P9_BAR() asm volatile("s_barrier" ::: "memory");

  unsigned left = n_mine;  // blocks of this workgroup not finished yet (the current one included)
  int tc = 0;              // key tile of the current block whose softmax comes next
  bool have_prev = false;

  if (wid < NW / 2) {
    // ---- early role: matrix half-steps on even global half-steps; all q / o traffic.
    // q pieces of the next block: wave wl's 16 pieces in the vector half-steps of tiles 1 .. nt - 3 (the late role reads the image for the current
    // block half a step behind this role's softmax(0)), waited for once, behind softmax(nt - 2): the barrier that follows is in front of both roles'
    // reads (behind their softmax(nt - 1)).  Issued by THIS role because it never waits on its memory counter inside a walk: in the late role's
    // queue the pieces (HBM latency) sat in front of K / V^T tiles (L2 latency) that are waited for every step — 0.93 ms per Wan-14B launch; counted
    // waits that leave them in flight: 0.91; here, with the round-robin item order: 0.85 (v9: 0.93-0.94; profiles/r06_cross_attn_persistent_form.txt).
    const int qps = (16 + nt - 4) / (nt - 3);
    __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)Q, 0, 0u, 0x00020000);
#define P9_ISSUE_Q()                                                                                                                  \
  if (left > 1 && tc >= 1 && (tc - 1) * qps < 16) {                                                                                   \
    if (tc == 1) {                                                                                                                    \
      P9Item nx = cur;                                                                                                                \
      item_next(nx);                                                                                                                  \
      const int64_t r0 = (int64_t)nx.qblk * (NW * 32);                                                                                \
      const int64_t rows = Sq - r0 < (int64_t)(NW * 32) ? Sq - r0 : (int64_t)(NW * 32);                                               \
      rq = __builtin_amdgcn_make_buffer_rsrc((void*)(Q + (int64_t)nx.seq * bs.q + (int64_t)nx.head * AT_D + r0 * ldq), 0,            \
                                             (unsigned)((rows - 1) * ldq * 2 + AT_D * 2), 0x00020000);                                \
    }                                                                                                                                 \
    const int j0 = (tc - 1) * qps, j1 = j0 + qps < 16 ? j0 + qps : 16;                                                                \
    for (int j = j0; j < j1; ++j)                                                                                                     \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (at_lds_ptr_t)(smem + Q_OFF + j * 4096 + wl * 1024), 16, q_voff, (unsigned)j * q_row16_bytes, 0, 0); \
    A9_SB();                                                                                                                          \
  }
#define P9_VECTOR_EARLY()                                             \
  P9_ISSUE_Q()                                                        \
  if (tc == 0 && have_prev) {                                         \
    P9_HANDOVER()                                                     \
  }                                                                   \
  A9_SOFTMAX(false)                                                                   \
  if (tc == nt - 2 && left > 1) {                                     \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  \
  }                                                                   \
  if (tc == nt - 1) {                                                 \
    P9_NEXT_BLOCK()                                                   \
  }
    P9_BAR()  // K(0): the late waves' prologue
    A9_MATRIX(0, 16, 0, 0)
    P9_BAR()
    for (;;) {
      P9_VECTOR_EARLY()
      P9_BAR()
      P9_MATRIX_STEP(0)
      P9_BAR()
      P9_ADVANCE()
      if (left == 0) break;
      P9_VECTOR_EARLY()
      P9_BAR()
      P9_MATRIX_STEP(1)
      P9_BAR()
      P9_ADVANCE()
      if (left == 0) break;
    }
    P9_BAR()
#undef P9_ISSUE_Q
#undef P9_VECTOR_EARLY
  } else {
    // ---- late role: vector half-steps and the K / V^T LDS-DMA on even global half-steps.
    // Cursors: K(tau + 2) / V^T(tau + 1) of the stream, as (descriptor of the block's head, tile within the block, block).
    P9Item ik = cur;  // V^T runs one tile behind K: when it wraps into a block, K's cursor is already in that block
    int tk = 0, tv = 0;
    unsigned k_left = n_mine * (unsigned)nt, v_left = k_left;
    __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)(Kp + (int64_t)ik.seq * bs.k + (int64_t)ik.head * AT_D), 0, k_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(VTp + (int64_t)ik.seq * bs.vt + (int64_t)ik.head * AT_D * ldvt), 0, v_bytes, 0x00020000);
#define P9_ISSUE_K(BUF_)                                                                                                              \
  if (k_left) {                                                                                                                       \
    A9_DMA_K((unsigned)tk * k_tile_bytes, BUF_)                                                                                       \
    --k_left;                                                                                                                         \
    if (++tk == nt) {                                                                                                                 \
      tk = 0;                                                                                                                         \
      item_next(ik);                                                                                                                  \
      rk = __builtin_amdgcn_make_buffer_rsrc((void*)(Kp + (int64_t)ik.seq * bs.k + (int64_t)ik.head * AT_D), 0, k_left ? k_bytes : 0u, 0x00020000); \
    }                                                                                                                                 \
  }
#define P9_ISSUE_V(BUF_)                                                                                                              \
  if (v_left) {                                                                                                                       \
    A9_DMA_V((unsigned)tv * (unsigned)AT_K_BYTES, BUF_)                                                                               \
    --v_left;                                                                                                                         \
    if (++tv == nt) {                                                                                                                 \
      tv = 0;                                                                                                                         \
      rv = __builtin_amdgcn_make_buffer_rsrc((void*)(VTp + (int64_t)ik.seq * bs.vt + (int64_t)ik.head * AT_D * ldvt), 0, v_left ? v_bytes : 0u, 0x00020000); \
    }                                                                                                                                 \
  }
#define P9_VECTOR_LATE()                                              \
  if (tc == 0 && have_prev) {                                         \
    P9_HANDOVER()                                                     \
  }                                                                   \
  A9_SOFTMAX(false)                                                   \
  if (tc == nt - 1) {                                                 \
    P9_NEXT_BLOCK()                                                   \
  }
#define P9_BAR_ODD()                                 \
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   \
  P9_BAR()
    P9_ISSUE_K(0)  // K(0)
    P9_BAR_ODD()
    P9_ISSUE_K(1)  // K(1), V^T(0)
    P9_ISSUE_V(0)
    A9_SB();
    P9_BAR()
    A9_MATRIX(0, 16, 0, 0)
    P9_BAR_ODD()
    for (;;) {
      P9_ISSUE_K(0)  // tau even: K(tau + 2) -> buffer 0, V^T(tau + 1) -> buffer 1
      P9_ISSUE_V(1)
      A9_SB();
      P9_VECTOR_LATE()
      P9_BAR()
      P9_MATRIX_STEP(0)
      P9_BAR_ODD()
      P9_ADVANCE()
      if (left == 0) break;
      P9_ISSUE_K(1)
      P9_ISSUE_V(0)
      A9_SB();
      P9_VECTOR_LATE()
      P9_BAR()
      P9_MATRIX_STEP(1)
      P9_BAR_ODD()
      P9_ADVANCE()
      if (left == 0) break;
    }
#undef P9_ISSUE_K
#undef P9_ISSUE_V
#undef P9_VECTOR_LATE
#undef P9_BAR_ODD
  }
  P9_STORE()
#undef P9_LANE
#undef P9_OUT_DESC
#undef P9_STORE
#undef P9_HANDOVER
#undef P9_NEXT_BLOCK
#undef P9_MATRIX_STEP
#undef P9_ADVANCE
#undef P9_BAR
#endif
}
#undef A9_SOFTMAX
#undef A9_MATRIX
#undef A9_FRAG
#undef A9_SB
#undef A9_DMA_K
#undef A9_DMA_V

// V [Sk, H*128] (token stride ldv) -> V^T [H][ldvt/64][128][64] bf16 (per head and 64-key tile a contiguous 16 KiB [dv][key] block: one
// attention tile = one linear stream, like K's) with keys >= Sk zero-filled up to ldvt (a multiple of 64).  One tile per block, through LDS.
__global__ __launch_bounds__(256) void transpose_heads_kernel(const unsigned short* __restrict__ V, int64_t ldv, unsigned short* __restrict__ VT, int64_t ldvt,
                                                              int64_t Sk) {
  __shared__ unsigned short tile[64][128 + 2];
  const int head = blockIdx.y;
  const int64_t k0 = (int64_t)blockIdx.x * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = i * 256 + threadIdx.x;
    const int row = id >> 4, c = id & 15;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (k0 + row < Sk) v = *reinterpret_cast<const uint4*>(V + (k0 + row) * ldv + (int64_t)head * 128 + c * 8);
    unsigned short* d = &tile[row][c * 8];
    d[0] = (unsigned short)(v.x & 0xffff); d[1] = (unsigned short)(v.x >> 16);
    d[2] = (unsigned short)(v.y & 0xffff); d[3] = (unsigned short)(v.y >> 16);
    d[4] = (unsigned short)(v.z & 0xffff); d[5] = (unsigned short)(v.z >> 16);
    d[6] = (unsigned short)(v.w & 0xffff); d[7] = (unsigned short)(v.w >> 16);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = i * 256 + threadIdx.x;
    const int dv = id >> 3, kc = id & 7;
    unsigned w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = (unsigned)tile[kc * 8 + 2 * e][dv] | ((unsigned)tile[kc * 8 + 2 * e + 1][dv] << 16);
    *reinterpret_cast<uint4*>(VT + (((int64_t)head * (ldvt >> 6) + blockIdx.x) * 128 + dv) * 64 + kc * 8) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

}  // namespace x2v
using namespace x2v;

template <int NW, int THR>
static int launch_attn_pipe(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo, int64_t Sq,
                            int64_t Sk, int H, float scale, hipStream_t st) {
  // buffer descriptors address 32 bits: the per-head K/V views must stay below 4 GiB
  const int64_t kb = (Sk - 1) * ldk * 2 + AT_D * 2, vb = (Sk - 1) * ldv * 2 + AT_D * 2;
  X2V_REQUIRE(kb < (1ll << 32) - (int64_t)AT_KV * ldk * 2 && vb < (1ll << 32) - (int64_t)AT_KV * ldv * 2, X2V_E_SHAPE,
              "attn: K/V view spans >= 4 GiB (Sk=%lld, ld=%lld/%lld)", (long long)Sk, (long long)ldk, (long long)ldv);
  int rc = ensure_dynamic_lds((const void*)attn_fwd_pipe_kernel<NW, THR>, AT_LDS_BYTES, "attn attr");
  if (rc != X2V_OK) return rc;
  const int QB = NW * 32;
  dim3 grid((unsigned)((Sq + QB - 1) / QB), (unsigned)H);
  hipLaunchKernelGGL((attn_fwd_pipe_kernel<NW, THR>), grid, dim3(NW * 64), AT_LDS_BYTES, st, (const unsigned short*)q, ldq, (const unsigned short*)k, ldk,
                     (const unsigned short*)v, ldv, (unsigned short*)o, ldo, Sq, Sk, scale * 1.4426950408889634f, (unsigned)kb, (unsigned)vb);
  X2V_LAUNCH_CHECK("attn launch");
  return X2V_OK;
}

// The launch form of the pre-transposed-V kernel for a shape: bit 0 = XCD-aware head-major work mapping, bit 8 = staggered key walk.
// Host-only and deterministic in its arguments (plus the X2V_ATTN_MAP / X2V_ATTN_ROT overrides for A/B runs); exported as
// x2v_attn_vt_launch_plan so that a parity test can assert WHICH kernel branch the shape it compared with the oracle took.
namespace x2v {
int attn_map_switch() {
  static const int v = [] { const char* e = getenv("X2V_ATTN_MAP"); return e ? atoi(e) : -1; }();
  return v;
}
int attn_rot_switch() {
  static const int v = [] { const char* e = getenv("X2V_ATTN_ROT"); return e ? atoi(e) : -1; }();
  return v;
}
int attn_short_switch() {  // X2V_ATTN_SHORT=0: never the persistent short-walk form (A/B runs)
  static const int v = [] { const char* e = getenv("X2V_ATTN_SHORT"); return e ? atoi(e) : -1; }();
  return v;
}
}  // namespace x2v
static int attn_cu_count() {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
    return n;
  }();
  return cus;
}
static int attn_vt_plan(int64_t Sq, int64_t Sk, int H, int B, bool stagger, int q_rows_per_wg, bool one_walk = false) {
  const int map_env = attn_map_switch(), rot_env = attn_rot_switch();
  const uint64_t nwg = (uint64_t)((Sq + q_rows_per_wg - 1) / q_rows_per_wg) * (uint64_t)H * (uint64_t)B;
  const int rot_mode = rot_env >= 0 ? rot_env : ((stagger && Sk >= 16 * AT_KV) ? 1 : 0);
  // bit 9: the persistent short-walk form (attn_fwd_p9_kernel) — whole key tiles, 4..32 of them, at least two walks per CU (the chip's 256: a
  // shape rule, not a device query, so that the plan is a function of its arguments), walk from tile 0
  if (attn_short_switch() != 0 && !one_walk && !rot_mode && Sk % AT_KV == 0 && Sk >= 4 * AT_KV && Sk <= 32 * AT_KV && nwg >= 512 && nwg < (1ull << 31)) return 0x200;
  const int64_t heads_in_flight = (int64_t)H * B < 8 ? (int64_t)H * B : 8;
  const bool mall_resident = heads_in_flight * Sk * (2 * AT_D * 2) <= (224ll << 20);
  const int map_mode = map_env >= 0 ? map_env : ((nwg >= 512 && mall_resident) ? 1 : 0);
  return (map_mode ? 1 : 0) | (rot_mode ? 0x100 : 0);
}

template <int NW, int THR, bool PRESCALED>
static int launch_attn_vt(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt, int64_t ldvt, void* o, int64_t ldo, int64_t Sq, int64_t Sk,
                          int H, float scale, hipStream_t st, int flags, int B = 1, AttnBatch bs = AttnBatch{0, 0, 0, 0, 0}) {
  const bool stagger = (flags & 2) != 0;
  // buffer ranges from a head's (and sequence's) first element: K rows of this sequence, the V^T blocks of its ceil(Sk/64) key tiles
  const int64_t kb = (Sk - 1) * ldk * 2 + AT_D * 2, vb = ((Sk + AT_KV - 1) / AT_KV) * (int64_t)AT_D * AT_KV * 2;
  X2V_REQUIRE(kb < (1ll << 32) - (int64_t)AT_KV * ldk * 2 && vb < (1ll << 32), X2V_E_SHAPE, "attn: K view / V^T head block spans >= 4 GiB");
  dim3 grid((unsigned)((Sq + NW * 32 - 1) / (NW * 32)), (unsigned)H, (unsigned)B);
  // Work mapping and key-walk rotation (see the kernels), measured on MI355X (profiles/r03_attn_v9_map_rot_matrix_and_pmc.txt):
  //   * the XCD-aware head-major mapping lifts the L2 hit rate from 73-79 % to 96 % and cuts the L2<->fabric traffic 6-8x at every shape, but it
  //     only PAYS while the K / V^T streams of the 8 heads then in flight fit the 256 MB Infinity Cache (Ulysses rank of Wan-14B 720p, 5 heads:
  //     +4 %; Wan-1.3B 480p: +1.5 %); with 40 heads at 720p (8 x 38.7 MB in flight) it is 4-8 % SLOWER than the plain grid, whatever the
  //     rotation — there the plain grid, which keeps all XCDs on the same ~2 heads, stays.
  //   * the stagger of the walk (rot mode 1) was worth +1.3 % on the plain grid in 2-step runs and costs 0.9 % at sustained load
  //     (profiles/r04_call12_*); it changes the summation order per query block; opt-in per call (flag X2V_ATTN_VT_STAGGER), no driver sets it.
  // X2V_ATTN_MAP / X2V_ATTN_ROT force a mode (A/B runs).
  const int plan = attn_vt_plan(Sq, Sk, H, B, stagger, NW * 32, (flags & 4) != 0);
  if (plan & 0x200) {
    const int nqb = (int)((Sq + NW * 32 - 1) / (NW * 32));
    const unsigned n_items = (unsigned)nqb * (unsigned)H * (unsigned)B;
    const int64_t qbytes = (int64_t)(NW * 32 - 1) * ldq * 2 + AT_D * 2;
    X2V_REQUIRE(qbytes < (1ll << 32), X2V_E_SHAPE, "attn: a 256-row q block spans >= 4 GiB");
    auto kern = attn_fwd_p9_kernel<THR, PRESCALED>;
    int rc = ensure_dynamic_lds((const void*)kern, 8 * AT_K_BYTES, "attn p9 attr");
    if (rc != X2V_OK) return rc;
    const unsigned cus = (unsigned)attn_cu_count();
    hipLaunchKernelGGL(kern, dim3(n_items < cus ? n_items : cus), dim3(NW * 64), 8 * AT_K_BYTES, st, (const unsigned short*)q, ldq, (const unsigned short*)k, ldk,
                       (const unsigned short*)vt, ldvt, (unsigned short*)o, ldo, Sq, (int)(Sk / AT_KV), scale * 1.4426950408889634f, (unsigned)kb, (unsigned)vb, bs, nqb, H,
                       n_items);
    X2V_LAUNCH_CHECK("attn p9 launch");
    return X2V_OK;
  }
  const int rot_mode = plan & 0x100;
  bs.xcd_remap = plan;
  auto kern = rot_mode ? attn_fwd_v9_kernel<NW, THR, PRESCALED, true> : attn_fwd_v9_kernel<NW, THR, PRESCALED, false>;
  int rc = ensure_dynamic_lds((const void*)kern, 4 * AT_K_BYTES, "attn attr");
  if (rc != X2V_OK) return rc;
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), 4 * AT_K_BYTES, st, (const unsigned short*)q, ldq, (const unsigned short*)k, ldk,
                     (const unsigned short*)vt, ldvt, (unsigned short*)o, ldo, Sq, Sk, scale * 1.4426950408889634f, (unsigned)kb, (unsigned)vb, bs);
  X2V_LAUNCH_CHECK("attn launch");
  return X2V_OK;
}

extern "C" __attribute__((visibility("default"))) int x2v_transpose_heads_bf16(const void* v, int64_t ldv, void* vt, int64_t ldvt, int64_t Sk, int H, void* stream) {
  X2V_REQUIRE(v && vt, X2V_E_ARG, "transpose_heads: null pointer");
  X2V_REQUIRE(Sk > 0 && H > 0 && H <= 65535 && ldvt % 64 == 0 && ldvt >= Sk && ldv >= (int64_t)H * AT_D && ldv % 8 == 0, X2V_E_SHAPE,
              "transpose_heads: bad shape (ldvt must be a multiple of 64 and >= Sk)");
  X2V_REQUIRE(aligned16(v) && aligned16(vt), X2V_E_ALIGN, "transpose_heads: 16-byte alignment");
  hipLaunchKernelGGL(transpose_heads_kernel, dim3((unsigned)(ldvt / 64), (unsigned)H), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)v, ldv,
                     (unsigned short*)vt, ldvt, Sk);
  X2V_LAUNCH_CHECK("transpose_heads launch");
  return X2V_OK;
}

extern "C" __attribute__((visibility("default"))) int x2v_attn_vt_launch_plan(int64_t Sq, int64_t Sk, int H, int B, int flags) {
  X2V_REQUIRE(Sq > 0 && Sk > 0 && H > 0 && H <= 65535 && B > 0 && B <= 65535 && (flags & ~7) == 0, X2V_E_SHAPE, "attn_vt_launch_plan: bad shape Sq=%lld Sk=%lld H=%d B=%d flags=%d",
              (long long)Sq, (long long)Sk, H, B, flags);
  return attn_vt_plan(Sq, Sk, H, B, (flags & 2) != 0, 8 * 32, (flags & 4) != 0);
}

extern "C" __attribute__((visibility("default"))) int x2v_attn_fwd_bf16_vt(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt, int64_t ldvt, void* o, int64_t ldo,
                                                                           int64_t Sq, int64_t Sk, int H, int head_dim, float scale, int flags, void* stream) {
  if (Sq == 0 && Sk > 0 && H > 0) return X2V_OK;  // no query rows: nothing to write (empty shard)
  X2V_REQUIRE(q && k && vt && o, X2V_E_ARG, "attn_vt: null pointer");
  X2V_REQUIRE(head_dim == AT_D, X2V_E_SHAPE, "attn_vt: head_dim=%d (only 128 is built)", head_dim);
  X2V_REQUIRE(Sq > 0 && Sk > 0 && H > 0 && H <= 65535, X2V_E_SHAPE, "attn_vt: bad shape Sq=%lld Sk=%lld H=%d", (long long)Sq, (long long)Sk, H);
  X2V_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 64 == 0 && ldvt >= Sk && ldo % 4 == 0 && aligned16(q) && aligned16(k) && aligned16(vt) && aligned16(o), X2V_E_ALIGN,
              "attn_vt: rows must be 16-byte aligned, ldvt a multiple of 64");
  X2V_REQUIRE(ldq >= (int64_t)H * AT_D && ldk >= (int64_t)H * AT_D && ldo >= (int64_t)H * AT_D, X2V_E_SHAPE, "attn_vt: token stride smaller than H*128");
  if (scale <= 0.f) scale = 0.08838834764831845f;
  X2V_REQUIRE((flags & ~7) == 0, X2V_E_ARG, "attn_vt: flags = X2V_ATTN_VT_PRESCALED | X2V_ATTN_VT_STAGGER | X2V_ATTN_VT_ONE_WALK");
  hipStream_t st = (hipStream_t)stream;
  return (flags & 1) ? launch_attn_vt<8, 8, true>(q, ldq, k, ldk, vt, ldvt, o, ldo, Sq, Sk, H, scale, st, flags)
                     : launch_attn_vt<8, 8, false>(q, ldq, k, ldk, vt, ldvt, o, ldo, Sq, Sk, H, scale, st, flags);
}

// B independent sequences in one launch (grid z): sequence b reads q / k / V^T and writes o at b * {q,k,vt,o}_bstride elements from the base
// pointers; all share Sq, Sk, H and the strides.  With V^T laid out [H][ldvt/64][128][64] over the tokens of ALL sequences (x2v_gemm_bf16_vt on
// the stacked rows), vt_bstride = rows_per_sequence * 128 and rows_per_sequence % 64 == 0.
extern "C" __attribute__((visibility("default"))) int x2v_attn_fwd_bf16_vt_batched(const void* q, int64_t ldq, int64_t q_bstride, const void* k, int64_t ldk, int64_t k_bstride,
                                                                                   const void* vt, int64_t ldvt, int64_t vt_bstride, void* o, int64_t ldo, int64_t o_bstride,
                                                                                   int64_t Sq, int64_t Sk, int H, int B, int head_dim, float scale, int flags, void* stream) {
  if (Sq == 0 && Sk > 0 && H > 0 && B > 0) return X2V_OK;
  X2V_REQUIRE(q && k && vt && o, X2V_E_ARG, "attn_vt_batched: null pointer");
  X2V_REQUIRE(head_dim == AT_D, X2V_E_SHAPE, "attn_vt_batched: head_dim=%d (only 128 is built)", head_dim);
  X2V_REQUIRE(Sq > 0 && Sk > 0 && H > 0 && H <= 65535 && B > 0 && B <= 65535, X2V_E_SHAPE, "attn_vt_batched: bad shape Sq=%lld Sk=%lld H=%d B=%d", (long long)Sq, (long long)Sk, H, B);
  X2V_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 64 == 0 && ldo % 4 == 0 && aligned16(q) && aligned16(k) && aligned16(vt) && aligned16(o) && q_bstride % 8 == 0 &&
                  k_bstride % 8 == 0 && vt_bstride % (64 * AT_D) == 0 && o_bstride % 8 == 0,
              X2V_E_ALIGN, "attn_vt_batched: rows must be 16-byte aligned, ldvt a multiple of 64, vt_bstride whole 64-key blocks");
  X2V_REQUIRE(ldq >= (int64_t)H * AT_D && ldk >= (int64_t)H * AT_D && ldo >= (int64_t)H * AT_D && ldvt >= (B - 1) * (vt_bstride / AT_D) + Sk, X2V_E_SHAPE,
              "attn_vt_batched: token stride smaller than H*128, or ldvt smaller than the stacked sequences");
  X2V_REQUIRE((flags & ~7) == 0, X2V_E_ARG, "attn_vt_batched: flags = X2V_ATTN_VT_PRESCALED | X2V_ATTN_VT_STAGGER | X2V_ATTN_VT_ONE_WALK");
  if (scale <= 0.f) scale = 0.08838834764831845f;
  const AttnBatch bs{q_bstride, k_bstride, vt_bstride, o_bstride, 0};
  hipStream_t st = (hipStream_t)stream;
  return (flags & 1) ? launch_attn_vt<8, 8, true>(q, ldq, k, ldk, vt, ldvt, o, ldo, Sq, Sk, H, scale, st, flags, B, bs)
                     : launch_attn_vt<8, 8, false>(q, ldq, k, ldk, vt, ldvt, o, ldo, Sq, Sk, H, scale, st, flags, B, bs);
}

// variant: 0 = default (= 6); lazy-rescale threshold of the pipelined kernel: 4 = eager rescale (every tile), 5 = threshold 4, 6 = threshold 8
// (base-2 units) — the three must agree to rounding (validation of the rare rescale branch)
extern "C" __attribute__((visibility("default"))) int x2v_attn_fwd_bf16_variant(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo, int64_t Sq,
                                         int64_t Sk, int H, int head_dim, float scale, int variant, void* stream) {
  if (Sq == 0 && Sk > 0 && H > 0) return X2V_OK;  // no query rows: nothing to write (empty shard)
  X2V_REQUIRE(q && k && v && o, X2V_E_ARG, "attn: null pointer");
  X2V_REQUIRE(head_dim == AT_D, X2V_E_SHAPE, "attn: head_dim=%d (only 128 is built)", head_dim);
  X2V_REQUIRE(Sq > 0 && Sk > 0 && H > 0 && H <= 65535, X2V_E_SHAPE, "attn: bad shape Sq=%lld Sk=%lld H=%d", (long long)Sq, (long long)Sk, H);
  X2V_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0 && aligned16(q) && aligned16(k) && aligned16(v) && aligned16(o), X2V_E_ALIGN,
              "attn: rows must be 16-byte aligned");
  X2V_REQUIRE(ldq >= (int64_t)H * AT_D && ldk >= (int64_t)H * AT_D && ldv >= (int64_t)H * AT_D && ldo >= (int64_t)H * AT_D, X2V_E_SHAPE,
              "attn: token stride smaller than H*128");
  if (scale <= 0.f) scale = 0.08838834764831845f;  // 1/sqrt(128)
  hipStream_t st = (hipStream_t)stream;
  switch (variant) {
    case 0:
    case 6: return launch_attn_pipe<8, 8>(q, ldq, k, ldk, v, ldv, o, ldo, Sq, Sk, H, scale, st);
    case 4: return launch_attn_pipe<8, -1>(q, ldq, k, ldk, v, ldv, o, ldo, Sq, Sk, H, scale, st);
    case 5: return launch_attn_pipe<8, 4>(q, ldq, k, ldk, v, ldv, o, ldo, Sq, Sk, H, scale, st);
    default: set_error("attn: unknown variant %d", variant); return X2V_E_ARG;
  }
}

extern "C" __attribute__((visibility("default"))) int x2v_attn_fwd_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo, int64_t Sq,
                                 int64_t Sk, int H, int head_dim, float scale, void* stream) {
  return x2v_attn_fwd_bf16_variant(q, ldq, k, ldk, v, ldv, o, ldo, Sq, Sk, H, head_dim, scale, 0, stream);
}
