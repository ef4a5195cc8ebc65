// PROBE, not product (round 6; DESIGN.md §4.1, HISTORY.md §R6): the PRODUCER / CONSUMER form of the pre-transposed-V attention forward that VERDICT r5
// asked to be built against a kill criterion (keep at >= +3 % over attn_fwd_v9_kernel).  It is correct (tools/x2v_check attn with variant 13: 35 / 35,
// the full-size fp32 triangles of tests/test_gpu_full_size.py) and runs at 0.975-0.99x of the ping-pong kernel: killed as a product path, kept here with
// its measurements.  Build: tools/probes/build_attn_pc.sh <tag> [-D...]; run: LD_LIBRARY_PATH=tools/probes/ab/<tag> tools/x2v_check pattn 13 75600 40.
//
// Same operands, LDS images of K / V^T, DMA pieces and MFMA geometry (v_mfma_f32_16x16x32_bf16, swapped S^T = K . Q^T, P as the B operand of
// O^T += V^T . P) as the ping-pong kernel attn_fwd_v9_kernel in attn.hip, whose header explains them.  What differs is who does what:
//
//   * a workgroup = 8 waves = 256 query rows of one head; the two waves of a SIMD (wid and wid + 4) form a PAIR that owns 64 query rows:
//       - the S-wave (wid < 4, "producer") keeps Q for the 64 rows (4 groups of 16) in registers, computes S^T = K . Q^T chunk by chunk (a chunk
//         = 16 keys = 16 MFMAs, one K fragment read feeds FOUR groups), runs the softmax of chunk c under the MFMAs of chunk c + 1 — chunk 0 of
//         tile t + 1 under the softmax of chunk 3 of tile t, over a THREE-slot K ring, so its stream has no MFMA-free tail and its first fragments
//         are requested in front of the barrier — and hands the packed bf16 probabilities to its partner through LDS (8 bytes per lane,
//         lane-linear: conflict-free);
//       - the O-wave (wid >= 4, "consumer") keeps O^T for the 64 rows (128 accumulator registers) and the row sums (an all-ones "V^T row" on the
//         matrix pipe: the sums of the ROUNDED probabilities, what O is built from), reads V^T fragments (one read feeds FOUR groups) and the P
//         fragments, and issues every LDS-DMA piece of the workgroup, one per fragment slot (in a burst they cost the whole kernel 8 %).
//     Per 64 rows and key tile: 16 KiB of K fragments + 16 KiB of V^T fragments + 8 KiB of P written + 8 KiB of P read = 48 KiB of LDS traffic
//     for the ping-pong kernel's 64 KiB — but NOT fewer LDS-pipeline cycles: a ds_write_b64 costs 6 cycles per 512 B and the consumer's paired
//     8-byte reads 8 per KiB against 4 per KiB for ds_read_b128 (MI355X_MICROARCH.md §LDS): ~1450 LDS cycles per tile and CU in both kernels.
//   * ONE barrier per key tile.  Interval t: the S-waves turn K(t) into P(t) (P slot t & 1); the O-waves multiply P(t-1) (slot (t-1) & 1) with
//     V^T(t-1) and issue the DMA of K(t+2) and V^T(t).  The O-waves hold back the MFMAs of their last two fragment slots and issue them right
//     BEHIND the barrier, under the LDS latency of their first fragment reads of the new interval.
//   * lazy running max without a max on the hot path.  Scores leave the MFMA relative to the row's running max m (C operand = -m), P = exp2(S')
//     with no per-score subtraction, no row max, no row sum and no cross-lane traffic; the hot path only keeps a packed max of the tile's bf16
//     probabilities (v_pk_max_u16: non-negative bf16 order like unsigned integers) and compares it with 2^8, the ping-pong kernel's lazy-rescale
//     bound.  When any lane of the S-wave trips (wave-uniform branch, cold), and on the first and the last (possibly ragged) tile, the tile is
//     REDONE in the textbook order: scores recomputed with C = 0, row maxima across the four lanes of a query column, m = max(m, row max), P
//     against the new m, the per-row factor exp2(m_old - m_new) recorded in LDS with a flag; the O-wave multiplies O and the row sums by it
//     before it consumes that tile's P.  132 VALU operations per 64 rows and tile for the ping-pong kernel's 212.
//
// Measured (profiles/r06_attn_pc_*; Wan-14B 720p launch, 75 600 x 40 heads, A-B-A-B on one box): ping-pong 84.1 / 84.2 ms, this kernel 84.6 / 85.1 ms
// (a slower box: 87.7-88.1 vs 89.6-90.1).  The first form (no chunk rotation, row sums as VALU adds in the S-wave, DMA in a burst) ran 92.1 ms:
// a cycle trace (X2V_PC_TRACE) showed the S-wave's stream (3360 cycles per interval) as the critical path with the O-wave waiting 385 cycles at every
// barrier; the DMA burst alone cost 7 ms.  What it does not escape: both kernels settle at ~70 % matrix-pipe busy at the clock the 1400 W limit grants
// (this one ran 2.03 GHz at 59.8 % busy in its first form against the ping-pong kernel's 1.885 GHz at 72.3 %), the S-wave waits for the partner's
// MFMA in the pipe at every one of its own (in-order issue, 16-cycle non-preemptible instructions), and the LDS pipeline is as busy as before.
// Negative on the way: v_dot2c_f32_bf16 for the row sums (+11 % time: the guide's "anti-lever beside MFMAs"), static priority 0..3 for the S-waves
// (+-0.7 %), fragment depth 6 (-1.5 %), V^T pieces first (no change).  A bug worth remembering: an asm-statement MFMA whose C operand had just been
// broadcast by v_mov read it stale (no wait states inside an asm string, cdna_hip_programming.md §5.7 item 2) — the builtin, which hipcc emits with
// D != C and no copy on gfx950, is what stays.
//
// Bound: MFMA (4 * Sq * Sk * H * 128 FLOP per launch).
#include "x2v_common.h"

namespace x2v {

namespace {

typedef __attribute__((address_space(3))) void* pc_lds_ptr_t;

constexpr int PC_D = 128;
constexpr int PC_KV = 64;
constexpr int PC_TILE = 16384;                       // one K or V^T tile image
constexpr int PC_K_OFF = 0;                          // 3 K slots (tile t in slot t mod 3: the S-waves start tile t + 1 before they finish tile t)
constexpr int PC_V_OFF = 3 * PC_TILE;                // 2 V^T slots
constexpr int PC_P_OFF = 5 * PC_TILE;                // 2 P slots x 4 pairs x 8 KiB: [slot][pair][key group j][query group g][chunk half b][lane][8 B]
constexpr int PC_P_SLOT = 4 * 8192;
constexpr int PC_R_OFF = PC_P_OFF + 2 * PC_P_SLOT;   // rescale records [slot][pair]: factor[16 c][4 g] fp32 (256 B), flag word at +256
constexpr int PC_REC = 512;
constexpr int PC_LDS_BYTES = PC_R_OFF + 2 * 4 * PC_REC;  // 151,552 B: one workgroup per CU (as the register budget already implies)

__device__ __forceinline__ float pc_max3(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// max of two packed pairs of NON-NEGATIVE bf16 values (their bit patterns order like unsigned integers): one v_pk_max_u16.  An asm statement: through
// the vector builtin hipcc re-derives the halves from the fp32 values (two more conversions and a v_perm per word).
__device__ __forceinline__ unsigned pc_pkmax(unsigned a, unsigned b) {
  unsigned d;
  asm("v_pk_max_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

// One 1 KiB LDS-DMA piece (buffer_load_dwordx4 ... lds: lane l's 16 bytes land at lds_off + 16 l).  M0 is saved and restored around the statement
// (compiler-reserved register, cdna_hip_programming.md §5.7); s_nop 4: a descriptor / offset word fresh from a VALU write; s_nop 0: M0 -> LDS-DMA.
__device__ __forceinline__ void pc_dma16(unsigned lds_off, unsigned voff, i32x4_t rsrc, unsigned soff) {
  unsigned keep;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_off), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory");
}
__device__ __forceinline__ i32x4_t pc_make_rsrc(const void* base, unsigned bytes) {
  const uint64_t a = (uint64_t)(uintptr_t)base;
  i32x4_t r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));  // stride 0
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}

#ifndef X2V_PC_DEPTH
#define X2V_PC_DEPTH 4  // fragment reads in flight ahead of their MFMAs
#endif
#ifndef X2V_PC_KNOCK
#define X2V_PC_KNOCK 0  // timing probes, results INVALID: 1 = the O-waves issue no MFMAs, 2 = the S-waves skip the softmax arithmetic, 3 = the S-waves issue no MFMAs, 4 = no P writes, 5 = no LDS-DMA in the loop, 6 = no softmax of a tile's last chunk
#endif
#ifndef X2V_PC_ASM_MFMA
#define X2V_PC_ASM_MFMA 0  // 1 = the first MFMA of a score chain as an asm statement (A/B builds)
#endif
#ifndef X2V_PC_ROWSUM
#define X2V_PC_ROWSUM 0  // the row sums of P in the O-waves: 0 = on the matrix pipe (an all-ones V^T row: 8 more MFMAs per tile), 1 = v_dot2c_f32_bf16 on the packed fragments (32 VALU per tile)
#endif
#ifndef X2V_PC_DMA_SPREAD
#define X2V_PC_DMA_SPREAD 1  // O-waves' LDS-DMA issue: 0 = eight pieces in a burst behind the held-back MFMAs, 1 = one per fragment slot 0..7 (K then V^T), 2 = one per ~1.75 slots, 3 = V^T in slots 0..3 then K, 4 = V^T right behind the barrier, K in slots 0..3 (A/B builds)
#endif
#ifndef X2V_PC_DEBUG
#define X2V_PC_DEBUG 0  // bug hunting: bit 0 = every hot tile takes the exact form as well, bit 1 = no fragment reads in front of the S-waves' barrier, bit 2 = chunk 0 recomputed at the start of its tile
#endif
#ifndef X2V_PC_PRIO
#define X2V_PC_PRIO 2  // static priority of the S-waves (their stream carries all the VALU work; the O-waves fill the matrix pipe's gaps)
#endif

template <bool PRESCALED>
__global__ __launch_bounds__(512, 2) void attn_fwd_pc_kernel(const unsigned short* __restrict__ Q, int64_t ldq, const unsigned short* __restrict__ Kp, int64_t ldk,
                                                             const unsigned short* __restrict__ VTp, int64_t ldvt, unsigned short* __restrict__ O, int64_t ldo,
                                                             int64_t Sq, int64_t Sk, float scale_log2e, unsigned k_bytes, unsigned v_bytes, AttnBatch bs) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int DEPTH = X2V_PC_DEPTH;
  // work mapping: as attn_fwd_v9_kernel (grid = query blocks of 256 rows x heads x sequences; bit 0 of bs.xcd_remap = XCD-aware head-major order)
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

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = wid & 3;
  const int c16 = lane & 15, qd = lane >> 4;
  const int64_t q0 = (int64_t)qblk * 256 + pair * 64;
  const int nt = (int)((Sk + PC_KV - 1) / PC_KV);

  // rescale flags start cleared (LDS is not initialised)
  if (tid < 8) *reinterpret_cast<int*>(smem + PC_R_OFF + tid * PC_REC + 256) = 0;

#define PC_SB() __builtin_amdgcn_sched_barrier(0)
  // end of an interval: this wave's LDS-DMA pieces have landed, its LDS writes are done, its fragment reads have returned
  // PC_ARRIVE: the barrier instruction alone; PC_BARRIER: end of an interval — this wave's LDS-DMA pieces have landed, its LDS writes are done, its
  // fragment reads have returned — then the barrier
#ifdef X2V_PC_TRACE  // timing probe (tools/probes): cycles between barriers ("stream") and at them ("wait"), per wave; the waves of workgroup 7 dump them over O
  unsigned tr_t0 = (unsigned)__builtin_readcyclecounter(), tr_stream = 0, tr_wait = 0, tr_n = 0;  // 32-bit sums: fine for one workgroup's ~4e6 cycles
#define PC_ARRIVE()                                                           \
  {                                                                           \
    const unsigned a_ = (unsigned)__builtin_readcyclecounter();               \
    asm volatile("s_barrier" ::: "memory");                                   \
    const unsigned b_ = (unsigned)__builtin_readcyclecounter();               \
    tr_stream += a_ - tr_t0;                                                  \
    tr_wait += b_ - a_;                                                       \
    tr_t0 = b_;                                                               \
    ++tr_n;                                                                   \
  }
#else
#define PC_ARRIVE() asm volatile("s_barrier" ::: "memory")
#endif
#define PC_BARRIER()                                                \
  {                                                                 \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     \
    PC_ARRIVE();                                                    \
  }

  if (wid < 4) {
    // ================================================================ S-wave (producer) ================================================================
    if (X2V_PC_PRIO) __builtin_amdgcn_s_setprio(X2V_PC_PRIO);
    const unsigned short* Qh = Q + (int64_t)head * PC_D;
    bf16x8_t qf[4][4];  // [query group][k-step of 32 head-dim values]
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      int64_t qr = q0 + 16 * g + c16;
      qr = qr < Sq ? qr : Sq - 1;
      const unsigned short* qp = Qh + qr * ldq + qd * 8;
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
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[g][ks]));

    // K fragment (chunk kt, k-step ks): row kappa(kt, c) = 32 (kt >> 1) + 8 (c >> 2) + 4 (kt & 1) + (c & 3), chunk (4 ks + qd) ^ c
    int kbase[4];
    const int krow_rd = 8 * (c16 >> 2) + (c16 & 3);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kbase[ks] = krow_rd * 256 + ((((ks << 2) | qd) ^ c16) << 4);
    char* const pw = smem + PC_P_OFF + pair * 8192 + lane * 8;
    char* const rec0 = smem + PC_R_OFF + pair * PC_REC;

    f32x4_t s[2][4], negm[4];
    float m_run[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      m_run[g] = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) negm[g][e] = 0.f;
    }
    bf16x8_t fr[DEPTH];

    // The exact (cold) form of one tile: scores with C = 0, row maxima across the four lanes of a query column, P against the new max, the rescale
    // factor of everything accumulated so far for the partner.  FIX_NEXT: chunk 0 of the NEXT tile already sits in s[0], relative to the old max.
    auto redo = [&](const char* kb, char* pb, char* rec, bool first, int left, bool fix_next) {
      const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
      f32x4_t sa[4][4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(kb + (kt >> 1) * 8192 + (kt & 1) * 1024 + kbase[ks]);
#pragma unroll
          for (int g = 0; g < 4; ++g) sa[kt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[g][ks], ks == 0 ? zero4 : sa[kt][g], 0, 0, 0);
        }
      if (left < PC_KV) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (32 * (kt >> 1) + 8 * qd + 4 * (kt & 1) + r >= left) {
#pragma unroll
              for (int g = 0; g < 4; ++g) sa[kt][g][r] = -1e30f;
            }
      }
      f32x4_t arec;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float mx = sa[0][g][0];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sa[kt][g][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // the row max itself (not a rounded-up bound): a key that dominates its row gets P = 1.0 exactly, as in the ping-pong kernel
        const float mnew = first ? mx : fmaxf(m_run[g], mx);
        const float d = mnew - m_run[g];
        arec[g] = first ? 1.0f : __builtin_amdgcn_exp2f(-d);  // first tile: nothing accumulated yet
        m_run[g] = mnew;
#pragma unroll
        for (int e = 0; e < 4; ++e) negm[g][e] = -mnew;
        if (fix_next) {
#pragma unroll
          for (int e = 0; e < 4; ++e) s[0][g][e] -= d;
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          const float p0 = __builtin_amdgcn_exp2f(sa[kt][g][0] - mnew), p1 = __builtin_amdgcn_exp2f(sa[kt][g][1] - mnew);
          const float p2 = __builtin_amdgcn_exp2f(sa[kt][g][2] - mnew), p3 = __builtin_amdgcn_exp2f(sa[kt][g][3] - mnew);
          *reinterpret_cast<uint2*>(pb + ((kt >> 1) * 4 + g) * 1024 + (kt & 1) * 512) = make_uint2(pack_bf2(p0, p1), pack_bf2(p2, p3));
        }
      }
      if (qd == 0) *reinterpret_cast<f32x4_t*>(rec + c16 * 16) = arec;
      if (lane == 0) *reinterpret_cast<int*>(rec + 256) = 1;
    };

    // softmax micro-ops of (chunk C_, slot SL_): query group SL_ >> 2, half SL_ & 1 of the group's four scores on slots (SL_ & 3) < 2, the
    // store on slot (SL_ & 3) == 2 — at most three VALU operations behind one MFMA.  pmax: running packed max of the tile's probabilities.
    float p0, p1;
    unsigned w0, w1, pmax;
#define PC_SMX(C_, SL_)                                                                                           \
  {                                                                                                               \
    constexpr int gg_ = (SL_) >> 2, qq_ = (SL_) & 3, cp_ = (C_) & 1;                                              \
    if constexpr (X2V_PC_KNOCK == 2) {                                                                            \
      if constexpr (qq_ == 2) {                                                                                   \
        *reinterpret_cast<uint2*>(pb + (((C_) >> 1) * 4 + gg_) * 1024 + ((C_) & 1) * 512) = make_uint2(__float_as_uint(s[cp_][gg_][0]), __float_as_uint(s[cp_][gg_][1])); \
        pmax = 0u;                                                                                                \
      }                                                                                                           \
    } else if constexpr (qq_ == 0) {                                                                              \
      p0 = __builtin_amdgcn_exp2f(s[cp_][gg_][0]);                                                                \
      p1 = __builtin_amdgcn_exp2f(s[cp_][gg_][1]);                                                                \
      w0 = pack_bf2(p0, p1);                                                                                      \
    } else if constexpr (qq_ == 1) {                                                                              \
      p0 = __builtin_amdgcn_exp2f(s[cp_][gg_][2]);                                                                \
      p1 = __builtin_amdgcn_exp2f(s[cp_][gg_][3]);                                                                \
      w1 = pack_bf2(p0, p1);                                                                                      \
    } else if constexpr (qq_ == 2) {                                                                              \
      if constexpr (X2V_PC_KNOCK == 4) asm volatile("" ::"v"(w0), "v"(w1));                                       \
      else *reinterpret_cast<uint2*>(pb + (((C_) >> 1) * 4 + gg_) * 1024 + ((C_) & 1) * 512) = make_uint2(w0, w1); \
      pmax = ((C_) == 0 && gg_ == 0) ? pc_pkmax(w0, w1) : pc_pkmax(pmax, pc_pkmax(w0, w1));                       \
    }                                                                                                             \
  }
    // compile-time slot dispatch (the slot index of an unrolled loop is a constant only after unrolling: switch over the 16 cases)
#define PC_SMX_DISPATCH(C_, SLV_)                                                                                                    \
  switch (SLV_) {                                                                                                                    \
    case 0: PC_SMX(C_, 0) break; case 1: PC_SMX(C_, 1) break; case 2: PC_SMX(C_, 2) break; case 3: PC_SMX(C_, 3) break;                \
    case 4: PC_SMX(C_, 4) break; case 5: PC_SMX(C_, 5) break; case 6: PC_SMX(C_, 6) break; case 7: PC_SMX(C_, 7) break;                \
    case 8: PC_SMX(C_, 8) break; case 9: PC_SMX(C_, 9) break; case 10: PC_SMX(C_, 10) break; case 11: PC_SMX(C_, 11) break;            \
    case 12: PC_SMX(C_, 12) break; case 13: PC_SMX(C_, 13) break; case 14: PC_SMX(C_, 14) break; default: PC_SMX(C_, 15) break;        \
  }
    // fragment F_ = 4 seg + ks of an interval: segments 0..2 = chunks 1..3 of the current tile (K slot at kcur), segment 3 = chunk 0 of the next (knxt)
#define PC_KFRAG(F_) \
  (*reinterpret_cast<const bf16x8_t*>(((F_) < 12 ? kcur : knxt) + (((((F_) >> 2) + 1) & 3) >> 1) * 8192 + (((((F_) >> 2) + 1) & 3) & 1) * 1024 + kbase[(F_) & 3]))
    // chunk 0 of a tile from the K slot at KB_ into s[0], no softmax beside it (behind the exact form of tile 0 only)
#define PC_A_C0(KB_)                                                                                                            \
  {                                                                                                                             \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                                          \
      const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>((KB_) + kbase[ks]);                                                \
      _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                                           \
        s[0][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[g][ks], ks == 0 ? negm[g] : s[0][g], 0, 0, 0);                 \
      }                                                                                                                         \
    }                                                                                                                           \
    PC_SB();                                                                                                                    \
  }
    // One hot interval: P slot PS_ (compile time), K slots at kcur / knxt (run time).  On entry s[0] holds chunk 0 of the tile (relative to m) and
    // fr[] the first DEPTH fragments (read in front of the barrier).  Segment sg: MFMAs of chunk sg + 1 (chunk 0 of the next tile for sg = 3)
    // into s[(sg + 1) & 1], softmax of chunk sg out of s[sg & 1] beside them.
#define PC_A_TILE(PS_)                                                                                                          \
  {                                                                                                                             \
    char* pb = pw + (PS_) * PC_P_SLOT;                                                                                          \
    if ((X2V_PC_DEBUG) & 4) { PC_A_C0(kcur) }                                                                                   \
    if ((X2V_PC_DEBUG) & 2) {                                                                                                   \
      _Pragma("unroll") for (int d = 0; d < DEPTH; ++d) fr[d] = PC_KFRAG(d);                                                    \
    }                                                                                                                           \
    _Pragma("unroll") for (int sg = 0; sg < 4; ++sg) {                                                                          \
      _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                                        \
        const bf16x8_t kf = fr[(sg * 4 + ks) % DEPTH];                                                                          \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                                         \
          if (X2V_PC_KNOCK == 3) asm volatile("" : "+v"(s[(sg + 1) & 1][g]) : "v"(kf));                                         \
          else if (X2V_PC_ASM_MFMA && ks == 0) /* first k-step, C = the -m tuple; s_nop 1: an operand may be fresh from a VALU write (round 6: without it the tuple's broadcast v_movs right in front of the statement were read stale) */ \
            asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(s[(sg + 1) & 1][g]) : "v"(kf), "v"(qf[g][0]), "v"(negm[g])); \
          else                                                                                                                  \
            s[(sg + 1) & 1][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[g][ks], ks == 0 ? negm[g] : s[(sg + 1) & 1][g], 0, 0, 0); \
          if (sg == 0) PC_SMX_DISPATCH(0, ks * 4 + g)                                                                           \
          if (sg == 1) PC_SMX_DISPATCH(1, ks * 4 + g)                                                                           \
          if (sg == 2) PC_SMX_DISPATCH(2, ks * 4 + g)                                                                           \
          if (sg == 3) PC_SMX_DISPATCH(3, ks * 4 + g)                                                                           \
          if (g == 3 && sg * 4 + ks + DEPTH < 16) fr[(sg * 4 + ks) % DEPTH] = PC_KFRAG(sg * 4 + ks + DEPTH);                    \
          PC_SB();                                                                                                              \
        }                                                                                                                       \
      }                                                                                                                         \
    }                                                                                                                           \
    /* any probability of the tile above 2^8 (bf16 0x4380), or not a number: redo the tile against its own row maxima */         \
    const unsigned pm_ = pmax > (pmax << 16) ? pmax : (pmax << 16);                                                             \
    if (__builtin_expect(((X2V_PC_DEBUG) & 1) || __any(pm_ > 0x4380ffffu) != 0, 0)) redo(kcur, pb, rec0 + (PS_) * 4 * PC_REC, false, PC_KV, true); \
    PC_SB();                                                                                                                    \
  }
    // the S-waves' barrier: P stores done, then the next interval's first fragments (their K tile landed an interval ago) requested in front of it
#define PC_A_BARRIER(PREFETCH_)                                              \
  {                                                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       \
    if ((PREFETCH_) && !((X2V_PC_DEBUG) & 2)) {                              \
      _Pragma("unroll") for (int d = 0; d < DEPTH; ++d) fr[d] = PC_KFRAG(d); \
    }                                                                        \
    PC_ARRIVE();                                                             \
  }

    const char* kcur = smem + PC_K_OFF;  // K slot of the tile whose softmax the interval runs
    const char* knxt = smem + PC_K_OFF + PC_TILE;
    PC_BARRIER();  // K(0), K(1) have landed
    redo(kcur, pw, rec0, true, nt == 1 ? (int)Sk : PC_KV, false);  // tile 0 (and the last tile: ragged key count) take the exact form
    int t = 1;
    if (nt > 1) {
      PC_A_C0(knxt)  // chunk 0 of tile 1
      kcur = knxt;
      knxt = smem + PC_K_OFF + 2 * PC_TILE;
      PC_A_BARRIER(nt > 2)
#define PC_A_ADVANCE()                                                                     \
  kcur = knxt;                                                                             \
  knxt = (knxt == smem + PC_K_OFF + 2 * PC_TILE) ? smem + PC_K_OFF : knxt + PC_TILE;
      while (t < nt - 1) {  // odd t: P slot 1
        PC_A_TILE(1)
        PC_A_ADVANCE()
        ++t;
        PC_A_BARRIER(t < nt - 1)
        if (t >= nt - 1) break;
        PC_A_TILE(0)
        PC_A_ADVANCE()
        ++t;
        PC_A_BARRIER(t < nt - 1)
      }
      redo(kcur, pw + (t & 1) * PC_P_SLOT, rec0 + (t & 1) * 4 * PC_REC, false, (int)(Sk - (int64_t)t * PC_KV), false);
    }
    PC_BARRIER();
#undef PC_A_ADVANCE
#undef PC_A_BARRIER
#undef PC_A_TILE
#undef PC_A_C0
#undef PC_SMX_DISPATCH
#undef PC_SMX
#undef PC_KFRAG
  } else {
    // ================================================================ O-wave (consumer) ================================================================
    const unsigned short* Kh = Kp + (int64_t)head * PC_D;
    const unsigned short* Vh = VTp + (int64_t)head * PC_D * ldvt;
    // Buffer descriptors as four scalar words: the LDS-DMA pieces are issued from asm statements (pc_dma16).  Through the builtin hipcc must assume
    // that a piece may write what any later ds_read reads, and drains vmcnt to 0 in front of this wave's next fragment read — right behind the
    // issue, i.e. a whole DMA latency per interval.  The only reader of a piece's bytes sits behind the NEXT barrier, whose wait is explicit.
    const i32x4_t rk = pc_make_rsrc(Kh, k_bytes), rv = pc_make_rsrc(Vh, v_bytes);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(pc_lds_ptr_t)smem);
    // LDS-DMA pieces exactly as attn_fwd_v9_kernel's issuing waves (wl = pair)
    const int wl = pair;
    const unsigned k_tile_bytes = (unsigned)(PC_KV * ldk * 2), k_row_bytes = (unsigned)(ldk * 2), v_piece_bytes = (unsigned)(8 * 4 * 128);
    const int krr = lane >> 4, vrow_w = wl * 8 + (lane >> 3);
    const unsigned k_voff = (unsigned)((8 * wl + krr) * ldk * 2) + (unsigned)(((lane & 15) ^ (krr | (wl << 2))) << 4);
    const unsigned v_voff = (unsigned)(vrow_w * 128) + (unsigned)(((lane & 7) ^ ((vrow_w >> 1) & 7)) << 4);
    // KSLOT_: byte offset of the K slot (run time, a multiple of PC_TILE); BUF_: V^T slot (compile time)
#define PC_DMA_K1(SOFF_, KSLOT_, J_) \
  pc_dma16(lds0 + PC_K_OFF + (KSLOT_) + ((wl << 1) | ((J_) & 1) | (((J_) >> 1) << 3)) * 1024, k_voff, rk, (SOFF_) + (unsigned)(4 * ((J_) & 1) + 32 * ((J_) >> 1)) * k_row_bytes);
#define PC_DMA_V1(SOFF_, BUF_, J_) pc_dma16(lds0 + PC_V_OFF + (BUF_) * PC_TILE + (wl + 4 * (J_)) * 1024, v_voff, rv, (SOFF_) + (J_) * v_piece_bytes);
#define PC_DMA_K1_RT(SOFF_, KSLOT_, J_) \
  switch (J_) { case 0: PC_DMA_K1(SOFF_, KSLOT_, 0) break; case 1: PC_DMA_K1(SOFF_, KSLOT_, 1) break; case 2: PC_DMA_K1(SOFF_, KSLOT_, 2) break; default: PC_DMA_K1(SOFF_, KSLOT_, 3) break; }
#define PC_DMA_V1_RT(SOFF_, BUF_, J_) \
  switch (J_) { case 0: PC_DMA_V1(SOFF_, BUF_, 0) break; case 1: PC_DMA_V1(SOFF_, BUF_, 1) break; case 2: PC_DMA_V1(SOFF_, BUF_, 2) break; default: PC_DMA_V1(SOFF_, BUF_, 3) break; }
#define PC_DMA_K(SOFF_, KSLOT_) PC_DMA_K1(SOFF_, KSLOT_, 0) PC_DMA_K1(SOFF_, KSLOT_, 1) PC_DMA_K1(SOFF_, KSLOT_, 2) PC_DMA_K1(SOFF_, KSLOT_, 3)
#define PC_DMA_V(SOFF_, BUF_) PC_DMA_V1(SOFF_, BUF_, 0) PC_DMA_V1(SOFF_, BUF_, 1) PC_DMA_V1(SOFF_, BUF_, 2) PC_DMA_V1(SOFF_, BUF_, 3)

    // V^T fragment (dv tile T, key group j): row 16 T + c, chunk (4 j + qd) ^ ((c >> 1) & 7)
    int vbase[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) vbase[j] = c16 * 128 + ((((j << 2) | qd) ^ ((c16 >> 1) & 7)) << 4);
    const char* const pr = smem + PC_P_OFF + pair * 8192 + lane * 8;
    char* const rec0 = smem + PC_R_OFF + pair * PC_REC;

    f32x4_t oacc[8][4], lacc[4];  // lacc: the row sums, on the matrix pipe (an all-ones "V^T row": every row of the 16 x 16 result is the sum)
    i32x4_t pf[2][4];
    bf16x8_t vfr[DEPTH], vtail[2], ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
    asm volatile("" : "+v"(ones));
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int T = 0; T < 8; ++T)
#pragma unroll
        for (int e = 0; e < 4; ++e) oacc[T][g][e] = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        lacc[g][e] = 0.f;
        pf[0][g][e] = 0;
        pf[1][g][e] = 0;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) vtail[i][e] = (__bf16)0.f;

#define PC_VFRAG(N_) (*reinterpret_cast<const bf16x8_t*>(vb + ((N_) & 7) * 2048 + vbase[(N_) >> 3]))
#define PC_LOAD_P(J_)                                                                                   \
  _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                       \
    const uint2 lo_ = *reinterpret_cast<const uint2*>(pb + ((J_) * 4 + g) * 1024);                      \
    const uint2 hi_ = *reinterpret_cast<const uint2*>(pb + ((J_) * 4 + g) * 1024 + 512);                \
    pf[J_][g] = i32x4_t{(int)lo_.x, (int)lo_.y, (int)hi_.x, (int)hi_.y};                                \
  }
#define PC_PV(ACC_, VF_, PF_)                                                                           \
  if (X2V_PC_KNOCK == 1) asm volatile("" ::"v"(VF_), "v"(PF_));                                         \
  else ACC_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(VF_, __builtin_bit_cast(bf16x8_t, PF_), ACC_, 0, 0, 0);
    // the two held-back fragment slots (14, 15 = key group 1, dv tiles 6, 7) of the previous tile
#define PC_B_TAIL()                                                                                     \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                       \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) { PC_PV(oacc[6 + i][g], vtail[i], pf[1][g]) }         \
    PC_SB();                                                                                            \
  }
    // interval u + 1: consume tile u (V^T slot PAR_, P slot PAR_), issue K(u + 3) into the K slot tile u just left and V^T(u + 1) -> V^T slot PAR_ ^ 1
#define PC_B_STEP(PAR_)                                                                                                                 \
  {                                                                                                                                     \
    PC_BARRIER();                                                                                                                       \
    const char* vb = smem + PC_V_OFF + (PAR_) * PC_TILE;                                                                                \
    const char* pb = pr + (PAR_) * PC_P_SLOT;                                                                                           \
    char* rec = rec0 + (PAR_) * 4 * PC_REC;                                                                                             \
    const int flag = *reinterpret_cast<const int*>(rec + 256);                                                                          \
    PC_LOAD_P(0)                                                                                                                        \
    _Pragma("unroll") for (int d = 0; d < DEPTH; ++d) vfr[d] = PC_VFRAG(d);                                                             \
    PC_SB();                                                                                                                            \
    if (X2V_PC_DMA_SPREAD == 4 && X2V_PC_KNOCK != 5 && u + 1 < nt) { PC_DMA_V((unsigned)(u + 1) * PC_TILE, (PAR_) ^ 1) }                \
    PC_B_TAIL()                                                                                                                         \
    if (!X2V_PC_DMA_SPREAD && X2V_PC_KNOCK != 5 && u + 3 < nt) { PC_DMA_K((unsigned)(u + 3) * k_tile_bytes, kslot) }                    \
    if (!X2V_PC_DMA_SPREAD && X2V_PC_KNOCK != 5 && u + 1 < nt) { PC_DMA_V((unsigned)(u + 1) * PC_TILE, (PAR_) ^ 1) }                    \
    const unsigned kslot_now = kslot;                                                                                                   \
    kslot = kslot == 2u * PC_TILE ? 0u : kslot + PC_TILE;                                                                               \
    PC_SB();                                                                                                                            \
    if (__builtin_amdgcn_readfirstlane(flag) != 0) { /* cold: the S-wave raised the running max of some rows at this tile */            \
      const f32x4_t al = *reinterpret_cast<const f32x4_t*>(rec + c16 * 16);                                                             \
      *reinterpret_cast<int*>(rec + 256) = 0;                                                                                           \
      _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                                                   \
        _Pragma("unroll") for (int T = 0; T < 8; ++T) _Pragma("unroll") for (int e = 0; e < 4; ++e) oacc[T][g][e] *= al[g];             \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) lacc[g][e] *= al[g];                                                              \
      }                                                                                                                                 \
    }                                                                                                                                   \
    PC_SB();                                                                                                                            \
    _Pragma("unroll") for (int n = 0; n < 14; ++n) {                                                                                    \
      const bf16x8_t vf = vfr[n % DEPTH];                                                                                               \
      _Pragma("unroll") for (int g = 0; g < 4; ++g) { PC_PV(oacc[n & 7][g], vf, pf[n >> 3][g]) }                                        \
      if ((n & 7) < 4) { /* row sums of key group n >> 3, query group n & 7 */                                                          \
        if (X2V_PC_ROWSUM == 0) { PC_PV(lacc[n & 7], ones, pf[n >> 3][n & 7]) }                                                         \
        else { _Pragma("unroll") for (int e = 0; e < 4; ++e) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(lacc[n & 7][0]) : "v"(pf[n >> 3][n & 7][e]), "v"(0x3f803f80u)); } \
      }                                                                                                                                 \
      if (n + DEPTH < 14) vfr[n % DEPTH] = PC_VFRAG(n + DEPTH);                                                                         \
      else if (n + DEPTH < 16) vtail[n + DEPTH - 14] = PC_VFRAG(n + DEPTH);                                                             \
      if (n == 2) { PC_LOAD_P(1) }                                                                                                      \
      if (X2V_PC_DMA_SPREAD && X2V_PC_KNOCK != 5) { /* piece p of the wave's eight (K 0..3, V^T 4..7) behind fragment slot p (mode 1) or (7 p) / 4 (mode 2) */ \
        _Pragma("unroll") for (int pc_ = 0; pc_ < 8; ++pc_) {                                                                           \
          if (n == (X2V_PC_DMA_SPREAD == 2 ? (7 * pc_) / 4 : X2V_PC_DMA_SPREAD == 3 ? (pc_ + 4) % 8 : X2V_PC_DMA_SPREAD == 4 ? pc_ % 4 : pc_)) { \
            if (pc_ < 4 && u + 3 < nt) { PC_DMA_K1_RT((unsigned)(u + 3) * k_tile_bytes, kslot_now, pc_) }                               \
            if (pc_ >= 4 && X2V_PC_DMA_SPREAD != 4 && u + 1 < nt) { PC_DMA_V1_RT((unsigned)(u + 1) * PC_TILE, (PAR_) ^ 1, pc_ - 4) }    \
          }                                                                                                                             \
        }                                                                                                                               \
      }                                                                                                                                 \
      PC_SB();                                                                                                                          \
    }                                                                                                                                   \
  }

    // prologue: K(0), K(1); interval 0: K(2), V^T(0)
    PC_DMA_K(0u, 0u)
    if (nt > 1) { PC_DMA_K(k_tile_bytes, (unsigned)PC_TILE) }
    PC_BARRIER();
    if (nt > 2) { PC_DMA_K(2u * k_tile_bytes, 2u * PC_TILE) }
    PC_DMA_V(0u, 0)
    unsigned kslot = 0u;  // K slot (byte offset) that tile u leaves free in interval u + 1
    int u = 0;
    while (true) {
      PC_B_STEP(0)
      if (++u >= nt) break;
      PC_B_STEP(1)
      if (++u >= nt) break;
    }
    PC_B_TAIL()
#undef PC_B_STEP
#undef PC_B_TAIL
#undef PC_PV
#undef PC_LOAD_P
#undef PC_VFRAG
#undef PC_DMA_K
#undef PC_DMA_V
#undef PC_DMA_K1
#undef PC_DMA_K1_RT
#undef PC_DMA_V1_RT
#undef PC_DMA_V1

    // epilogue: O / l
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float l = lacc[g][0];
      if (X2V_PC_ROWSUM != 0) {  // per-lane partial sums (this lane's 8 keys of every 32): the four lanes of a query column add up
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
      }
      const float inv = 1.0f / l;
      const int64_t qrow = q0 + 16 * g + c16;
      if (qrow < Sq) {
        unsigned short* op = O + qrow * ldo + (int64_t)head * PC_D + 4 * qd;
#pragma unroll
        for (int T = 0; T < 8; ++T) {
          uint2 pk;
          pk.x = pack_bf2(oacc[T][g][0] * inv, oacc[T][g][1] * inv);
          pk.y = pack_bf2(oacc[T][g][2] * inv, oacc[T][g][3] * inv);
          *reinterpret_cast<uint2*>(op + 16 * T) = pk;
        }
      }
    }
  }
#ifdef X2V_PC_TRACE
  if (blockIdx.x == 7 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0) {
    float* dbg = reinterpret_cast<float*>(O) + wid * 4;  // garbage over the first rows of O: probe builds only
    dbg[0] = (float)tr_stream;
    dbg[1] = (float)tr_wait;
    dbg[2] = (float)tr_n;
    dbg[3] = (float)((unsigned)__builtin_readcyclecounter() - tr_t0);
  }
#endif
#undef PC_SB
#undef PC_BARRIER
#undef PC_ARRIVE
#endif
}

}  // namespace
}  // namespace x2v
using namespace x2v;

// The probe's own entry (NOT part of include/x2v.h): x2v_attn_fwd_bf16_vt's operands and checks; flags bit 0 = q carries scale * log2(e),
// bit 2 = XCD-aware head-major work mapping.  tools/x2v_check resolves it with dlsym from a library built by tools/probes/build_attn_pc.sh.
extern "C" __attribute__((visibility("default"))) int x2v_probe_attn_pc(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt, int64_t ldvt, void* o, int64_t ldo,
                                                                        int64_t Sq, int64_t Sk, int H, int head_dim, float scale, int flags, void* stream) {
  if (Sq == 0 && Sk > 0 && H > 0) return X2V_OK;
  X2V_REQUIRE(q && k && vt && o, X2V_E_ARG, "probe_attn_pc: null pointer");
  X2V_REQUIRE(head_dim == PC_D && Sq > 0 && Sk > 0 && H > 0 && H <= 65535, X2V_E_SHAPE, "probe_attn_pc: bad shape");
  X2V_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 64 == 0 && ldvt >= Sk && ldo % 4 == 0 && aligned16(q) && aligned16(k) && aligned16(vt) && aligned16(o), X2V_E_ALIGN,
              "probe_attn_pc: rows must be 16-byte aligned, ldvt a multiple of 64");
  if (scale <= 0.f) scale = 0.08838834764831845f;
  const int64_t kb = (Sk - 1) * ldk * 2 + PC_D * 2, vb = ((Sk + PC_KV - 1) / PC_KV) * (int64_t)PC_D * PC_KV * 2;
  X2V_REQUIRE(kb < (1ll << 32) - (int64_t)PC_KV * ldk * 2 && vb < (1ll << 32), X2V_E_SHAPE, "probe_attn_pc: K view / V^T head block spans >= 4 GiB");
  dim3 grid((unsigned)((Sq + 255) / 256), (unsigned)H, 1u);
  auto kern = (flags & 1) ? attn_fwd_pc_kernel<true> : attn_fwd_pc_kernel<false>;
  int rc = ensure_dynamic_lds((const void*)kern, PC_LDS_BYTES, "attn_pc attr");
  if (rc != X2V_OK) return rc;
  const AttnBatch bs{0, 0, 0, 0, (flags & 4) ? 1 : 0};
  hipLaunchKernelGGL(kern, grid, dim3(512), PC_LDS_BYTES, (hipStream_t)stream, (const unsigned short*)q, ldq, (const unsigned short*)k, ldk, (const unsigned short*)vt, ldvt,
                     (unsigned short*)o, ldo, Sq, Sk, scale * 1.4426950408889634f, (unsigned)kb, (unsigned)vb, bs);
  X2V_LAUNCH_CHECK("attn_pc launch");
  return X2V_OK;
}
