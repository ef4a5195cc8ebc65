// ARCHIVED EXPERIMENT (round 2), not part of libx2v_hip.so: self-attention with 64 query rows per wave, one wave per SIMD, accumulators and Q in
// hand-allocated AGPRs.  Measured 0.97-1.0x the ping-pong kernel (profiles/r02_attn_w64_vs_pingpong.log, r02_pmc_attn_pingpong_vs_w64_S75600_H5.txt):
// a single wave per SIMD cannot issue the softmax VALU work beside its own MFMAs (tools/probes/mfma_filler_probe.hip).  It was wired in as
// selector 2 of x2v_attn_fwd_bf16_vt (attn_w64_dispatch) while it was measured; compiles with  hipcc --offload-arch=gfx950 -I include -I lightx2v_amd/csrc -c.
// Self-attention forward, 64 query rows per wave, one wave per SIMD ("w64"; x2v_attn_fwd_bf16_vt kernel selector 2).
//
// Bound: MFMA.  Same operands, numerics and LDS images as the ping-pong kernel of attn.hip (V pre-transposed, q carrying
// scale*log2(e), scores leaving the MFMA relative to the running max, lazy rescale); what changes is the register blocking:
//   * a workgroup = 4 waves = 256 query rows of one head; each wave owns TWO 32-row query blocks and the whole 512-entry register
//     file of its SIMD.  O (2 x 4 tiles = a[0:127]) and Q (2 x 8 fragments = a[128:191]) live in the accumulator half and are touched
//     ONLY by the asm statements of this file, by literal register number — hipcc never sees them (given "a"-constrained C++ values it
//     parks them in the VGPR half or scratch and copies them in front of every MFMA: 175-700 spilled registers).  Two score buffers
//     (2 x 64), -m (32) and the fragments are ordinary C++ values in the VGPR half; P is packed in place into its score tuple;
//   * every K / V^T fragment read from LDS feeds TWO MFMAs (one per query block): 0.5 ds_read_b128 per MFMA instead of 1, half the LDS
//     bytes and wait instructions per FLOP, and K/V tiles are streamed once per 256 rows by 4 waves instead of 8;
//   * with a single wave per SIMD the overlap of matrix and vector work is inside the instruction stream.  Per key tile t two phases of
//     32 MFMA slots (ONE score buffer, one packed-P buffer):
//         Q(t+1): QK^T -> S(t+1), key block 0 first   carrying  exp2 / row-sum / pack of tile t's key groups 2, 3 (they read the
//                                                               key-block-1 tuples, overwritten only by slots 16..31) + 8 LDS-DMA pieces
//         P(t)  : PV(t), key groups in order          carrying  the row max of S(t+1), its rescale decision, then exp2 / sum / pack of
//                                                               tile t+1's key groups 0, 1 (into P words PV(t) has already consumed)
//     each slot = {one MFMA | <= 5 VALU | 1/2 fragment read}, pinned with sched_barrier.  The rescale of tile t+1 is decided while PV(t)
//     still accumulates into O, so it is split: scores, row sums and -m are adjusted at once, O is scaled at the head of the next Q phase
//     (after PV(t) has finished, before PV(t+1) starts);
//   * one barrier per tile (in front of the Q phase); K(t+2) and V^T(t+1) are issued during Q(t+1) (their buffers were last read before
//     that barrier) and awaited before the next one: a full tile of flight.
// AUDIT after every edit (the accumulator half is invisible to the compiler): `hipcc -S` must show .vgpr_spill_count 0,
// .private_segment_fixed_size 0 and no v_accvgpr_* / a[..] operand outside ;;#ASMSTART / ;;#ASMEND.
#include <type_traits>

#include "x2v_common.h"

namespace x2v {

constexpr int W6_D = 128, W6_KV = 64, W6_TILE_BYTES = 64 * 256;  // 16 KiB per K or V^T tile
typedef __attribute__((address_space(3))) void* w6_lds_ptr_t;

template <int B, int E, class F>
__device__ __forceinline__ void w6_for(F&& f) {  // f(integral_constant<int, i>) for i = B .. E-1, fully unrolled with constant indices
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    w6_for<B + 1, E>(f);
  }
}

__device__ __forceinline__ float w6_max3(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float w6_max2(float a, float b) {
  float d;
  asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

// ---- the accumulator half: O tile i (= query block * 4 + dv block) is a[16 i : 16 i + 15], Q fragment j (= query block * 8 + head-dim
//      step) is a[128 + 4 j : 128 + 4 j + 3]
template <int I>
__device__ __forceinline__ void w6_pv(const bf16x8_t& vf, const f32x4_t& p) {  // O_I += V^T fragment . P
  asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(vf), "v"(p), "i"(16 * I), "i"(16 * I + 15));
}
template <int J>
__device__ __forceinline__ void w6_qk_first(f32x16_t& s, const bf16x8_t& kf, const f32x16_t& negm) {  // S = K fragment . Q_J - m
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, a[%c3:%c4], %2" : "=&v"(s) : "v"(kf), "v"(negm), "i"(128 + 4 * J), "i"(128 + 4 * J + 3));
}
template <int J>
__device__ __forceinline__ void w6_qk_acc(f32x16_t& s, const bf16x8_t& kf) {  // S += K fragment . Q_J
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(s) : "v"(kf), "i"(128 + 4 * J), "i"(128 + 4 * J + 3));
}
template <int R>
__device__ __forceinline__ void w6_acc_zero() {
  asm volatile("v_accvgpr_write_b32 a[%c0], 0" ::"i"(R));
}
template <int R>
__device__ __forceinline__ void w6_acc_write(unsigned v) {
  asm volatile("v_accvgpr_write_b32 a[%c1], %0" ::"v"(v), "i"(R));
}
template <int R>
__device__ __forceinline__ float w6_acc_read() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(x) : "i"(R));
  return x;
}
template <int R>
__device__ __forceinline__ void w6_acc_scale(float al) {  // a[R] *= al (cold path)
  float tmp;
  asm volatile("v_accvgpr_read_b32 %0, a[%c2]\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a[%c2], %0" : "=&v"(tmp) : "v"(al), "i"(R));
}

template <int RESCALE_THR, bool PRESCALED, int PROBE = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_fwd_w64_kernel(
    const unsigned short* __restrict__ Q, int64_t ldq, const unsigned short* __restrict__ Kp, int64_t ldk, const unsigned short* __restrict__ VTp, int64_t ldvt,
    unsigned short* __restrict__ O, int64_t ldo, int64_t Sq, int64_t Sk, float scale_log2e, unsigned k_bytes, unsigned v_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int K_OFF = 0, V_OFF = 2 * W6_TILE_BYTES;
  // reserve a[0:191] in the kernel descriptor (the compiler itself allocates nothing there: audit rule in the header)
  asm volatile("" ::: "a0", "a127", "a128", "a191");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fl = lane & 31, hi = lane >> 5;
  const int head = blockIdx.y;
  const int64_t q0 = (int64_t)blockIdx.x * 256 + wid * 64;
  const unsigned short* Kh = Kp + (int64_t)head * W6_D;
  const unsigned short* Vh = VTp + (int64_t)head * W6_D * ldvt;

  // ---- Q fragments of the two query blocks (rows q0 + fl and q0 + 32 + fl) -> a[128:191]; O = 0 -> a[0:127]
  w6_for<0, 16>([&](auto jc) {
    constexpr int J = decltype(jc)::value, qb = J >> 3, ks = J & 7;
    int64_t qr = q0 + qb * 32 + fl;
    qr = qr < Sq ? qr : Sq - 1;
    bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(Q + qr * ldq + (int64_t)head * W6_D + hi * 8 + ks * 16);
    if constexpr (!PRESCALED) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (__bf16)((float)v[e] * scale_log2e);
    }
    const i32x4_t w = __builtin_bit_cast(i32x4_t, v);
    w6_acc_write<128 + 4 * J + 0>((unsigned)w[0]);
    w6_acc_write<128 + 4 * J + 1>((unsigned)w[1]);
    w6_acc_write<128 + 4 * J + 2>((unsigned)w[2]);
    w6_acc_write<128 + 4 * J + 3>((unsigned)w[3]);
  });
  w6_for<0, 128>([&](auto rc) { w6_acc_zero<decltype(rc)::value>(); });

  // ---- LDS-DMA: 16 K pieces (4 rows x 256 B) + 16 V^T pieces (8 rows x 128 B) per tile, 4 of each per wave (pieces wid + 4 j: their rows
  //      differ by multiples of 16, so one swizzled per-lane offset per operand serves all of them)
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)Kh, 0, k_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)Vh, 0, v_bytes, 0x00020000);
  const unsigned k_tile_bytes = (unsigned)(W6_KV * ldk * 2), k_piece_bytes = (unsigned)(16 * ldk * 2), v_piece_bytes = 32u * 128u;
  const int krow_w = wid * 4 + (lane >> 4), vrow_w = wid * 8 + (lane >> 3);
  const unsigned k_voff = (unsigned)(krow_w * ldk * 2) + (unsigned)(((lane & 15) ^ (krow_w & 15)) << 4);
  const unsigned v_voff = (unsigned)(vrow_w * 128) + (unsigned)(((lane & 7) ^ ((vrow_w >> 1) & 7)) << 4);
  auto dma_k = [&](int tile, int buf, int j) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (w6_lds_ptr_t)(smem + K_OFF + buf * W6_TILE_BYTES + (wid + 4 * j) * 1024), 16, k_voff,
                                             (unsigned)tile * k_tile_bytes + (unsigned)j * k_piece_bytes, 0, 0);
  };
  auto dma_v = [&](int tile, int buf, int j) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (w6_lds_ptr_t)(smem + V_OFF + buf * W6_TILE_BYTES + (wid + 4 * j) * 1024), 16, v_voff,
                                             (unsigned)tile * W6_TILE_BYTES + (unsigned)j * v_piece_bytes, 0, 0);
  };

  // ---- fragment read offsets (as the ping-pong kernel: K rows through the bit-2/bit-3 swap so that a half-wave's P registers are 8
  //      consecutive keys = the k order of a 16-byte V^T fragment)
  int kaddr[8], vaddr[4];
  const int krow_rd = (fl & 0x13) | ((fl & 4) << 1) | ((fl & 8) >> 1);
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) kaddr[ks] = krow_rd * 256 + (((hi ^ (krow_rd & 15)) << 4) ^ (ks << 5));
#pragma unroll
  for (int g = 0; g < 4; ++g) vaddr[g] = fl * 128 + ((((g << 1) | hi) ^ ((fl >> 1) & 7)) << 4);
  f32x16_t S[2][2], negm[2];  // scores of the tile in flight: [query block][key block]; -m as the C operand of the first QK^T step
  // packed P: [buffer][query block][16-key group].  Groups 0..2 of tile t+1 are produced while PV(t) still reads tile t's -> two buffers
  // alternating by tile parity; group 3 is produced in the following Q phase, after PV(t) has read it -> buffer 0 only
  f32x4_t pw[2][2][4];
#pragma unroll
  for (int e = 0; e < 16; ++e) negm[0][e] = negm[1][e] = 0.f;
  float m_run[2] = {0.f, 0.f}, mx[2] = {0.f, 0.f}, pend_al[2] = {1.f, 1.f};
  float ls[2][2] = {{0.f, 0.f}, {0.f, 0.f}};   // row sums (two partial chains per query block)
  float lsz[2][2] = {{0.f, 0.f}, {0.f, 0.f}};  // row-sum share of the SPECULATIVE group 0 of the next tile (folded in or redone by decide)
  float pa[2], pb[2], pm[2][4];
  unsigned long long ts[16] = {};  // PROBE 3: cycle stamps
  bool force = true;     // first tile: adopt its row max in either direction
  bool pending = false;  // O still has to be scaled by pend_al (decided while the previous PV was accumulating)
  const int nt = (int)((Sk + W6_KV - 1) / W6_KV);

#define W6_SB() __builtin_amdgcn_sched_barrier(0)

  // ---- vector work, in slot-sized pieces -------------------------------------------------------------------------------------
  // quarter C (0..31) of a tile's exp2 / row-sum / pack: unit (16-key group C>>3, query block (C>>2)&1), pair j = C&3, into P buffer PB.
  // SPEC: the quarter runs BEFORE its tile's rescale decision (scores still relative to the old max): its sums go to lsz.
  auto eq = [&](auto cc, auto pbc, auto specc) {
    constexpr int C = decltype(cc)::value, qb = (C >> 2) & 1, uh = C >> 3, j = C & 3, kb = uh >> 1, e = (uh & 1) * 8 + 2 * j;
    constexpr int PB = uh == 3 ? 0 : decltype(pbc)::value;
    constexpr bool SPEC = decltype(specc)::value;
    const float p0 = __builtin_amdgcn_exp2f(S[qb][kb][e]), p1 = __builtin_amdgcn_exp2f(S[qb][kb][e + 1]);
    if constexpr (SPEC) {
      if constexpr (j == 0) {
        lsz[qb][0] = p0;
        lsz[qb][1] = p1;
      } else {
        lsz[qb][0] += p0;
        lsz[qb][1] += p1;
      }
    } else {
      ls[qb][0] += p0;
      ls[qb][1] += p1;
    }
    pw[PB][qb][uh][j] = __uint_as_float(pack_bf2(p0, p1));
    // pin the quarter into ITS slot: pure arithmetic is otherwise sunk towards its first use (past the MFMAs it should hide under)
    if constexpr (SPEC) asm volatile("" : "+v"(pw[PB][qb][uh][j]), "+v"(lsz[qb][0]), "+v"(lsz[qb][1]));
    else asm volatile("" : "+v"(pw[PB][qb][uh][j]), "+v"(ls[qb][0]), "+v"(ls[qb][1]));
  };
  // step I (0..21) of the row max of S: query block I&1, sub-step I>>1 (0..7: groups of 8 scores, key block 0 first; 8, 9: combine;
  // 10: the half-wave exchange)
  auto mstep = [&](auto ic) {
    constexpr int I = decltype(ic)::value, qb = I & 1, st = I >> 1;
    if constexpr (st < 8) {
      constexpr int g = st >> 1, kb = g >> 1, r0 = (g & 1) * 8;
      if constexpr ((st & 1) == 0) {
        pa[qb] = w6_max3(S[qb][kb][r0], S[qb][kb][r0 + 1], S[qb][kb][r0 + 2]);
        pb[qb] = w6_max3(S[qb][kb][r0 + 3], S[qb][kb][r0 + 4], S[qb][kb][r0 + 5]);
      } else {
        const float c = w6_max2(S[qb][kb][r0 + 6], S[qb][kb][r0 + 7]);
        pm[qb][g] = w6_max3(pa[qb], pb[qb], c);
      }
    } else if constexpr (st == 8) {
      pa[qb] = w6_max2(pm[qb][0], pm[qb][1]);
      pb[qb] = w6_max2(pm[qb][2], pm[qb][3]);
    } else if constexpr (st == 9) {
      pa[qb] = w6_max2(pa[qb], pb[qb]);
    } else {
      auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(pa[qb]), __float_as_uint(pa[qb]), false, false);
      mx[qb] = w6_max2(__uint_as_float(sw[0]), __uint_as_float(sw[1]));  // row max relative to m_run
    }
  };
  // keys past Sk in the last tile never win a max and get P = 0
  auto mask = [&](int tile) {
    const int left = (int)(Sk - (int64_t)tile * W6_KV);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (u * 32 + (r & 7) + 8 * hi + 16 * (r >> 3) >= left) S[qb][u][r] = -1e30f;
  };
  // group 0 of the tile in S, from scratch (after a mask or a rescale invalidated the speculative quarters): lsz and P buffer PB
  auto redo_spec = [&](auto pbc) {
    w6_for<0, 8>([&](auto cc) { eq(cc, pbc, std::true_type{}); });
  };
  // The rescale decision for the tile whose scores are in S, whose group 0 went (SPEC) speculatively into P buffer PB (cold: some row's
  // max grew by more than THR, or first tile).  Everything but O is adjusted here; O (busy under the running PV phase) follows at the
  // head of the next Q phase (scale_o).  The speculative row-sum share is relative to the old max like every earlier contribution, so the
  // common factor covers it — but its exp2 may have overflowed, so it is recomputed with the packed words instead.
  auto decide = [&](auto pbc, auto specc) {
    constexpr bool SPEC = decltype(specc)::value;
    if (force || __any(fmaxf(mx[0], mx[1]) > (float)RESCALE_THR)) {
      w6_for<0, 2>([&](auto qc) {
        constexpr int qb = decltype(qc)::value;
        const float d = force ? mx[qb] : fmaxf(mx[qb], 0.f);
        m_run[qb] += d;
        if (!force) {
          const float al = __builtin_amdgcn_exp2f(-d);
          ls[qb][0] *= al;
          ls[qb][1] *= al;
          pend_al[qb] = al;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) negm[qb][e] = -m_run[qb];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int e = 0; e < 16; ++e) S[qb][u][e] -= d;
      });
      pending = !force;
      force = false;
      if constexpr (SPEC) redo_spec(pbc);
    }
    if constexpr (SPEC) {  // fold the (now final) group-0 share into the row sums
      ls[0][0] += lsz[0][0];
      ls[0][1] += lsz[0][1];
      ls[1][0] += lsz[1][0];
      ls[1][1] += lsz[1][1];
    }
  };
  auto scale_o = [&]() {
    if (pending) {
      asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");  // the last PV MFMAs' results before O is touched
      w6_for<0, 2>([&](auto qc) {
        constexpr int qb = decltype(qc)::value;
        const float al = pend_al[qb];
        w6_for<0, 64>([&](auto rc) { w6_acc_scale<qb * 64 + decltype(rc)::value>(al); });
      });
      asm volatile("s_nop 1" ::: "memory");
      pending = false;
    }
  };

  // Fragment stream of one loop iteration: 0..15 = K of tile t+1 in Q-phase order (key block g>>3, head-dim step g&7), 16..31 = V^T of
  // tile t in P-phase order (key group (g-16)>>2, dv block g&3); 4-deep ring, read 3 fragments ahead of use.
#define W6_FRAG(KB_, VB_, G_)                                                                                                  \
  ((G_) < 16 ? *reinterpret_cast<const bf16x8_t*>(smem + K_OFF + (KB_) * W6_TILE_BYTES + (((G_) & 15) >> 3) * 8192 + kaddr[(G_) & 7]) \
             : *reinterpret_cast<const bf16x8_t*>(smem + V_OFF + (VB_) * W6_TILE_BYTES + ((G_) & 3) * 4096 + vaddr[(((G_) - 16) & 15) >> 2]))

  // One loop iteration for tile t (P buffer PC = t & 1):
  //   Q phase: QK^T(t+1) | group 3 of tile t (even slots < 16) | the 8 DMA pieces (odd slots < 16) | from slot 16: row max of S(t+1) over key
  //            block 0 alternating with the SPECULATIVE group 0 of tile t+1 (-> P buffer PC ^ 1)
  //   P phase: PV(t) | rest of the row max (slots 0..6), decision (slot 7), groups 1, 2 of tile t+1 (16 quarters over slots 8..31)
  // KN: K buffer holding tile t+1, VC: V^T buffer holding tile t = PC; HAS_NEXT: t + 1 < nt.
  auto iter = [&](auto knc, auto vcc, auto hnc, int t) {
    constexpr int KN = decltype(knc)::value, VC = decltype(vcc)::value, PC = VC, PN = VC ^ 1;
    constexpr bool HAS_NEXT = decltype(hnc)::value;
    constexpr int G0 = HAS_NEXT ? 0 : 16;
    using pc_t = std::integral_constant<int, PC>;
    using pn_t = std::integral_constant<int, PN>;
    if constexpr (PROBE == 3) { if (t >= 200 && t < 204) ts[(t - 200) * 4 + 0] = __builtin_readcyclecounter(); }
    scale_o();
    W6_SB();
    bf16x8_t fr[4];
    fr[0] = W6_FRAG(KN, VC, G0 + 0);
    fr[1] = W6_FRAG(KN, VC, G0 + 1);
    fr[2] = W6_FRAG(KN, VC, G0 + 2);
    W6_SB();
    // ---- Q phase
    w6_for<0, 32>([&](auto nc) {
      constexpr int n = decltype(nc)::value, g = n >> 1, qb = n & 1, kb = g >> 3, ks = g & 7;
      if constexpr (HAS_NEXT) {
        if constexpr (qb == 0) fr[(g + 3) & 3] = W6_FRAG(KN, VC, g + 3);
      }
      if constexpr (PROBE != 2) {
        // group 3 of tile t (reads the key-block-1 tuples, overwritten from slot 16 on): one quarter every other slot
        if constexpr (n < 16 && (n & 1) == 0) eq(std::integral_constant<int, (24 + (n >> 1))>{}, pc_t{}, std::false_type{});
        if constexpr (HAS_NEXT && n >= 16) {
          // key block 0 of S(t+1) is complete since slot 15 (>= 2 MFMA issues ago for every register read here)
          if constexpr ((n & 1) == 0) mstep(std::integral_constant<int, ((n - 16) >> 1)>{});
          else eq(std::integral_constant<int, ((n - 17) >> 1)>{}, pn_t{}, std::true_type{});
        }
      }
      if constexpr (HAS_NEXT) {
        if constexpr (ks == 0) w6_qk_first<qb * 8>(S[qb][kb], fr[g & 3], negm[qb]);
        else w6_qk_acc<qb * 8 + ks>(S[qb][kb], fr[g & 3]);
        if constexpr (n < 16 && (n & 1) == 1 && PROBE != 1) {
          // unconditional: a tile past the end lies outside the buffer descriptor's range and lands as zeros in a buffer nobody reads
          constexpr int j = n >> 1;
          if constexpr (j < 4) dma_k(t + 2, KN ^ 1, j);
          else dma_v(t + 1, VC ^ 1, j - 4);
        }
      }
      W6_SB();
    });
    // ---- P phase
    if constexpr (PROBE == 3) { if (t >= 200 && t < 204) ts[(t - 200) * 4 + 1] = __builtin_readcyclecounter(); }
    if constexpr (HAS_NEXT) {
      if ((int64_t)(t + 2) * W6_KV > Sk) {  // ragged last tile: mask S(t+1), then redo what ran on the unmasked key block 0 (once per block)
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
        mask(t + 1);
        w6_for<0, 8>([&](auto ic) { mstep(ic); });
        redo_spec(pn_t{});
      }
    }
    w6_for<0, 32>([&](auto nc) {
      constexpr int n = decltype(nc)::value, g = 16 + (n >> 1), qb = n & 1, T = g & 3, uh = (g - 16) >> 2;
      constexpr int ring = HAS_NEXT ? g : g - 16;
      if constexpr (qb == 0 && g + 3 < 32) fr[(ring + 3) & 3] = W6_FRAG(KN, VC, g + 3);
      w6_pv<qb * 4 + T>(fr[ring & 3], pw[uh == 3 ? 0 : PC][qb][uh]);
      if constexpr (HAS_NEXT && PROBE != 2) {
        if constexpr (n < 7) {  // row max, key block 1 and the combine steps (14 steps)
          mstep(std::integral_constant<int, 8 + 2 * n>{});
          mstep(std::integral_constant<int, 8 + 2 * n + 1>{});
        }
        if constexpr (n == 7) decide(pn_t{}, std::true_type{});
        // groups 1, 2 of tile t+1 into the other P buffer: 16 quarters over slots 8..31 (two in every three slots)
        if constexpr (n >= 8 && (n - 8) % 3 != 2) eq(std::integral_constant<int, 8 + ((n - 8) / 3) * 2 + (n - 8) % 3>{}, pn_t{}, std::false_type{});
      }
      W6_SB();
    });
    if constexpr (PROBE == 3) { if (t >= 200 && t < 204) ts[(t - 200) * 4 + 2] = __builtin_readcyclecounter(); }
  };

  // ---- prologue: K(0), V^T(0), K(1) in flight; S(0), its row max, adoption of the max, groups 0..2 into P buffer 0
  {
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_k(0, 0, j);
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_v(0, 0, j);
    if (nt > 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) dma_k(1, 1, j);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    w6_for<0, 32>([&](auto nc) {
      constexpr int n = decltype(nc)::value, g = n >> 1, qb = n & 1, kb = g >> 3, ks = g & 7;
      const bf16x8_t kf = W6_FRAG(0, 0, g);
      if constexpr (ks == 0) w6_qk_first<qb * 8>(S[qb][kb], kf, negm[qb]);
      else w6_qk_acc<qb * 8 + ks>(S[qb][kb], kf);
    });
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    if ((int64_t)W6_KV > Sk) mask(0);
    w6_for<0, 22>([&](auto ic) { mstep(ic); });
    decide(std::integral_constant<int, 0>{}, std::false_type{});
    w6_for<0, 24>([&](auto cc) { eq(cc, std::integral_constant<int, 0>{}, std::false_type{}); });
  }

  // ---- tiles: t even -> K(t+1) in buffer 1, V^T(t) in buffer 0; t odd -> the other way round.  The barrier in front of each iteration
  //      orders this iteration's DMA behind every wave's reads of the buffers it re-targets.
  using c0 = std::integral_constant<int, 0>;
  using c1 = std::integral_constant<int, 1>;
  int t = 0;
  while (t + 1 < nt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (PROBE == 3) { if (t >= 200 && t < 204) ts[(t - 200) * 4 + 3] = __builtin_readcyclecounter(); }
    __builtin_amdgcn_s_barrier();
    iter(c1{}, c0{}, std::true_type{}, t);
    ++t;
    if (t + 1 >= nt) break;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (PROBE == 3) { if (t >= 200 && t < 204) ts[(t - 200) * 4 + 3] = __builtin_readcyclecounter(); }
    __builtin_amdgcn_s_barrier();
    iter(c0{}, c1{}, std::true_type{}, t);
    ++t;
  }
  // last tile t = nt - 1: its group 3, then PV
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (t & 1) iter(c0{}, c1{}, std::false_type{}, t);
  else iter(c1{}, c0{}, std::false_type{}, t);
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");

  if constexpr (PROBE == 3) {
    if (blockIdx.x == 3 && blockIdx.y == 0 && tid == 0) {
      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(O + (int64_t)Sq * ldo / 2);  // scratch in the middle of the output
      for (int i = 0; i < 16; ++i) dbg[i] = ts[i];
    }
    return;
  }
  // ---- epilogue: O / l -> bf16
  w6_for<0, 2>([&](auto qc) {
    constexpr int qb = decltype(qc)::value;
    const float l_run = ls[qb][0] + ls[qb][1];
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int64_t qrow = q0 + qb * 32 + fl;
    unsigned short* op = O + (qrow < Sq ? qrow : 0) * ldo + (int64_t)head * W6_D;
    w6_for<0, 16>([&](auto gc) {  // (dv block T, group of 4 accumulator registers g): 4 consecutive dv of this lane's query row
      constexpr int T = decltype(gc)::value >> 2, g = decltype(gc)::value & 3, base = (qb * 4 + T) * 16 + 4 * g;
      const int dv = 32 * T + 8 * g + 4 * hi;
      uint2 pk;
      pk.x = pack_bf2(w6_acc_read<base + 0>() * inv, w6_acc_read<base + 1>() * inv);
      pk.y = pack_bf2(w6_acc_read<base + 2>() * inv, w6_acc_read<base + 3>() * inv);
      if (qrow < Sq) *reinterpret_cast<uint2*>(op + dv) = pk;
    });
  });
#undef W6_FRAG
#undef W6_SB
#endif
}

template <bool PRESCALED, int PROBE = 0>
static int launch_w64(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt, int64_t ldvt, void* o, int64_t ldo, int64_t Sq, int64_t Sk, int H,
                      float scale, hipStream_t st) {
  const int64_t kb = (Sk - 1) * ldk * 2 + W6_D * 2, vb = (int64_t)W6_D * ldvt * 2;
  X2V_REQUIRE(kb < (1ll << 32) - (int64_t)W6_KV * ldk * 2 && vb < (1ll << 32), X2V_E_SHAPE, "attn: K view / V^T head block spans >= 4 GiB");
  auto kern = attn_fwd_w64_kernel<8, PRESCALED, PROBE>;
  int rc = ensure_dynamic_lds((const void*)kern, 4 * W6_TILE_BYTES, "attn w64 attr");
  if (rc != X2V_OK) return rc;
  dim3 grid((unsigned)((Sq + 255) / 256), (unsigned)H);
  hipLaunchKernelGGL(kern, grid, dim3(256), 4 * W6_TILE_BYTES, st, (const unsigned short*)q, ldq, (const unsigned short*)k, ldk, (const unsigned short*)vt, ldvt,
                     (unsigned short*)o, ldo, Sq, Sk, scale * 1.4426950408889634f, (unsigned)kb, (unsigned)vb);
  X2V_LAUNCH_CHECK("attn w64 launch");
  return X2V_OK;
}

// called by x2v_attn_fwd_bf16_vt (attn.hip) after argument validation
int attn_w64_probe(int probe, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt, int64_t ldvt, void* o, int64_t ldo, int64_t Sq, int64_t Sk, int H, float scale,
                    hipStream_t st) {  // timing probes (results invalid): 1 no DMA in the loop, 2 no softmax VALU, 3 no barrier
  switch (probe) {
    case 1: return launch_w64<false, 1>(q, ldq, k, ldk, vt, ldvt, o, ldo, Sq, Sk, H, scale, st);
    case 2: return launch_w64<false, 2>(q, ldq, k, ldk, vt, ldvt, o, ldo, Sq, Sk, H, scale, st);
    case 3: return launch_w64<false, 3>(q, ldq, k, ldk, vt, ldvt, o, ldo, Sq, Sk, H, scale, st);
    default: return X2V_E_ARG;
  }
}
int attn_w64_dispatch(bool prescaled, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt, int64_t ldvt, void* o, int64_t ldo, int64_t Sq, int64_t Sk,
                      int H, float scale, hipStream_t st) {
  return prescaled ? launch_w64<true>(q, ldq, k, ldk, vt, ldvt, o, ldo, Sq, Sk, H, scale, st) : launch_w64<false>(q, ldq, k, ldk, vt, ldvt, o, ldo, Sq, Sk, H, scale, st);
}

}  // namespace x2v
