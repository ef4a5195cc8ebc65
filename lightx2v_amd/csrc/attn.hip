// Dense non-causal attention forward, head_dim 128, bf16 in/out, fp32 online softmax.
//
// Bound: MFMA.  Algorithmic work per launch = 4 * Sq * Sk * H * 128 FLOP (QK^T + PV).
//
// Structure (v1):
//   * one workgroup = NW waves x 32 query rows of ONE head; K/V tiles of 64 keys are staged once per
//     workgroup in LDS (register-staged, issue-early / write-late so HBM latency hides under the MFMAs of
//     the current tile) and shared by all waves; double buffered, one barrier per tile.
//   * "swapped" QK^T: S^T = K . Q^T with v_mfma_f32_32x32x16_bf16, K fragment as the A operand.  Each lane
//     then owns ONE query column (lane&31) and 32 of the tile's 64 keys, so the whole online softmax
//     (max, exp2, row sum, rescale of O) is lane-local; the two half-waves exchange one max per tile.
//   * P never leaves registers: the accumulator layout of S^T (per lane: keys {0-3,8-11,..}+4*half) IS a valid
//     B-operand layout for the PV MFMA as long as the V^T fragment enumerates keys in the same order — the
//     reduction index of an MFMA may be permuted freely if both operands agree.  V^T fragments come from
//     row-major V tiles through the gfx950 transpose read ds_read_b64_tr_b16 (two per fragment).
//   * O^T accumulates as 4 MFMA tiles of [32 dv][32 queries] per wave (64 accumulator registers).
//   * LDS images: K [64 keys][256 B] with the 16-byte chunk index XORed by (key & 15) (conflict-free
//     ds_read_b128 over its 16-lane service groups); V as 8 sub-tiles [64 keys][16 cols] (32-byte rows, so a
//     16-lane transpose read touches 128 contiguous bytes) with a 2080-byte sub-tile pitch chosen so both
//     the 16-byte staging writes and the paired (sub-tile T, T+4) transpose reads are conflict-free.
//   * q-block-fastest grid: co-resident workgroups walk the same head's K/V stream in near lock-step, so
//     each XCD's L2 serves a K/V tile to its 32 CUs from one fill.
#include "x2v_common.h"

namespace x2v {

constexpr int AT_D = 128;
constexpr int AT_KV = 64;
constexpr int AT_K_BYTES = AT_KV * 256;     // 16 KiB
constexpr int AT_VSUB = 2080;               // bytes per V sub-tile ([64][16] bf16 = 2048 + 32 pad)
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

// key index (within a 32-key MFMA tile) held in accumulator register r of half-wave `hi`
__device__ __forceinline__ int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

template <int NW, bool SAFE_V>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_kernel(const unsigned short* __restrict__ Q, int64_t ldq, const unsigned short* __restrict__ Kp,
                                                              int64_t ldk, const unsigned short* __restrict__ Vp, int64_t ldv,
                                                              unsigned short* __restrict__ O, int64_t ldo, int64_t Sq, int64_t Sk, float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = NW * 64;
  constexpr int QB = NW * 32;           // queries per workgroup
  constexpr int CPT = (AT_KV * 16) / NT;  // 16-byte chunks per thread per operand tile (4 for NW=4, 2 for NW=8)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int fl = lane & 31, hi = lane >> 5;
  const int head = blockIdx.y;
  const int64_t q0 = (int64_t)blockIdx.x * QB + wid * 32;

  const unsigned short* Kh = Kp + (int64_t)head * AT_D;
  const unsigned short* Vh = Vp + (int64_t)head * AT_D;

  // ---- Q fragments (B operand): lane (query fl, half hi) holds d = ks*16 + hi*8 .. +8
  bf16x8_t qf[8];
  {
    int64_t qr = q0 + fl;
    qr = qr < Sq ? qr : Sq - 1;
    const unsigned short* qp = Q + qr * ldq + (int64_t)head * AT_D + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp + ks * 16);
  }

  // ---- staging assignment: chunk id = i*NT + tid -> key = id/16, c = id%16.  Staging registers are
  //      plain named vectors (no arrays captured by reference: hipcc would demote those to scratch).
  i32x4_t kst[CPT], vst[CPT];
  int k_wr[CPT], v_wr[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int id = i * NT + tid;
    const int key = id >> 4, c = id & 15;
    k_wr[i] = key * 256 + ((c ^ (key & 15)) << 4);
    v_wr[i] = AT_K_BYTES + (c >> 1) * AT_VSUB + key * 32 + (c & 1) * 16;
  }
#define AT_ISSUE_LOADS(T_)                                                         \
  _Pragma("unroll") for (int i = 0; i < CPT; ++i) {                                \
    const int id = i * NT + tid;                                                   \
    int64_t key = (int64_t)(T_) * AT_KV + (id >> 4);                               \
    key = key < Sk ? key : Sk - 1;                                                 \
    const int c = id & 15;                                                         \
    kst[i] = *reinterpret_cast<const i32x4_t*>(Kh + key * ldk + c * 8);            \
    vst[i] = *reinterpret_cast<const i32x4_t*>(Vh + key * ldv + c * 8);            \
  }
#define AT_WRITE_STAGE(BUF_)                                                       \
  {                                                                                \
    char* b_ = smem + (BUF_) * AT_BUF_BYTES;                                       \
    _Pragma("unroll") for (int i = 0; i < CPT; ++i) {                              \
      *reinterpret_cast<i32x4_t*>(b_ + k_wr[i]) = kst[i];                          \
      *reinterpret_cast<i32x4_t*>(b_ + v_wr[i]) = vst[i];                          \
    }                                                                              \
  }

  // ---- fragment read offsets
  int k_rd[2][8];  // K A-operand: key = t*32 + fl, chunk = ks*2 + hi
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int key = t * 32 + fl;
      k_rd[t][ks] = key * 256 + (((ks * 2 + hi) ^ (key & 15)) << 4);
    }
  // V^T A-operand for dv tile T: lanes fl<16 read sub-tile T, fl>=16 read sub-tile T+4; within the 16-lane
  // group lane L supplies row (L>>2) / column group (L&3) of the [4 keys][16 cols] block.
  const int L = lane & 15;
  const int v_rd_base = AT_K_BYTES + ((fl >> 4) * 4) * AT_VSUB + (4 * hi + (L >> 2)) * 32 + (L & 3) * 8;

  f32x16_t oacc[4];
#pragma unroll
  for (int T = 0; T < 4; ++T)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[T][e] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  const int nt = (int)((Sk + AT_KV - 1) / AT_KV);
  AT_ISSUE_LOADS(0)
  AT_WRITE_STAGE(0)
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const char* kb = smem + (t & 1) * AT_BUF_BYTES;
    if (t + 1 < nt) {
      AT_ISSUE_LOADS(t + 1)
    }

    // ---- S^T = K Q^T : two [32 keys][32 queries] tiles
    f32x16_t st[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int e = 0; e < 16; ++e) st[u][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(kb + k_rd[u][ks]);
        st[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], st[u], 0, 0, 0);
      }
    }
    // ---- mask keys beyond Sk (last tile only; wave-uniform branch)
    if ((int64_t)(t + 1) * AT_KV > Sk) {
      const int left = (int)(Sk - (int64_t)t * AT_KV);  // valid keys in this tile, 1..63
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (u * 32 + acc_row(r, hi) >= left) st[u][r] = -1e30f;
    }
    // ---- online softmax (base-2 domain): lane-local over its 32 keys, one exchange across half-waves
    float mx = st[0][0];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[u][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx * scale_log2e);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
    bf16x8_t pb[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float p[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        p[r] = __builtin_amdgcn_exp2f(st[u][r] * scale_log2e - m_new);
        psum += p[r];
      }
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int e = 0; e < 8; ++e) pb[u][h2][e] = (__bf16)p[h2 * 8 + e];
    }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[T][e] *= alpha;

    // ---- O^T += V^T P^T
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        // keys of this k16 step: u*32 + h2*16 + {0-3, 8-11} + 4*hi
#pragma unroll
        for (int T = 0; T < 4; ++T) {
          bf16x8_t vf;
          if (!SAFE_V) {
            const char* p0 = kb + v_rd_base + T * AT_VSUB + (u * 32 + h2 * 16) * 32;
            vf = tr_frag(p0, p0 + 8 * 32);
          } else {
            const int sub = T + (fl >> 4) * 4, col = fl & 15;
            const unsigned short* vs = reinterpret_cast<const unsigned short*>(kb + AT_K_BYTES + sub * AT_VSUB);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const int key = u * 32 + h2 * 16 + (e & 3) + 8 * (e >> 2) + 4 * hi;
              unsigned short raw = vs[key * 16 + col];
              vf[e] = __builtin_bit_cast(__bf16, raw);
            }
          }
          oacc[T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[u][h2], oacc[T], 0, 0, 0);
        }
      }

    if (t + 1 < nt) AT_WRITE_STAGE((t + 1) & 1)
    __syncthreads();
  }

  // ---- epilogue: O[q][dv] = O^T / l
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int64_t qrow = q0 + fl;
  if (qrow < Sq) {
    unsigned short* op = O + qrow * ldo + (int64_t)head * AT_D;
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int i0 = 8 * g + 4 * hi;  // row index within the [32 dv] tile (4 consecutive rows i0..i0+3)
        const int dv = (i0 < 16) ? (16 * T + i0) : (64 + 16 * T + (i0 - 16));
        uint2 pk;
        pk.x = pack_bf2(oacc[T][4 * g + 0] * inv, oacc[T][4 * g + 1] * inv);
        pk.y = pack_bf2(oacc[T][4 * g + 2] * inv, oacc[T][4 * g + 3] * inv);
        *reinterpret_cast<uint2*>(op + dv) = pk;
      }
  }
}

#undef AT_ISSUE_LOADS
#undef AT_WRITE_STAGE

}  // namespace x2v

using namespace x2v;

template <int NW, bool SAFE_V>
static int launch_attn(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo, int64_t Sq, int64_t Sk,
                       int H, float scale, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    int rc = check_hip(hipFuncSetAttribute((const void*)attn_fwd_kernel<NW, SAFE_V>, hipFuncAttributeMaxDynamicSharedMemorySize, AT_LDS_BYTES), "attn attr");
    if (rc != X2V_OK) return rc;
    attr_set = true;
  }
  const int QB = NW * 32;
  dim3 grid((unsigned)((Sq + QB - 1) / QB), (unsigned)H);
  hipLaunchKernelGGL((attn_fwd_kernel<NW, SAFE_V>), grid, dim3(NW * 64), AT_LDS_BYTES, st, (const unsigned short*)q, ldq, (const unsigned short*)k, ldk,
                     (const unsigned short*)v, ldv, (unsigned short*)o, ldo, Sq, Sk, scale * 1.4426950408889634f);
  X2V_LAUNCH_CHECK("attn launch");
  return X2V_OK;
}

// variant: 0 = default, 1 = NW=4 tr-read, 2 = NW=8 tr-read, 3 = NW=4 scalar-V (validation path for the transpose read)
extern "C" __attribute__((visibility("default"))) int x2v_attn_fwd_bf16_variant(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo, int64_t Sq,
                                         int64_t Sk, int H, int head_dim, float scale, int variant, void* stream) {
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
    case 2: return launch_attn<8, false>(q, ldq, k, ldk, v, ldv, o, ldo, Sq, Sk, H, scale, st);
    case 1: return launch_attn<4, false>(q, ldq, k, ldk, v, ldv, o, ldo, Sq, Sk, H, scale, st);
    case 3: return launch_attn<4, true>(q, ldq, k, ldk, v, ldv, o, ldo, Sq, Sk, H, scale, st);
    default: set_error("attn: unknown variant %d", variant); return X2V_E_ARG;
  }
}

extern "C" __attribute__((visibility("default"))) int x2v_attn_fwd_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo, int64_t Sq,
                                 int64_t Sk, int H, int head_dim, float scale, void* stream) {
  return x2v_attn_fwd_bf16_variant(q, ldq, k, ldk, v, ldv, o, ldo, Sq, Sk, H, head_dim, scale, 0, stream);
}
