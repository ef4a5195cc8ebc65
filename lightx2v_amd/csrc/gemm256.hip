// y[M,N] = epi(x[M,K] . W[N,K]^T + bias) — the large-shape GEMM: 256x256 output tile, 8 waves, ping-pong phases.
// Same contract, operand layouts, epilogues and rounding points as gemm.hip (which stays the small-shape kernel).
//
// Bound: MFMA (bf16 dense peak ~2.5 PFLOP/s, fp8 ~5); algorithmic work 2*M*N*K FLOP per launch.
//
// Why a second structure: the 128^2 / two-barriers-per-K-step kernel drains its LDS-DMA queue (vmcnt(0)) in front of
// every barrier and tops out near 900 TFLOP/s.  Here
//   * one 512-thread workgroup per CU owns a 256x256 tile; wave (wr, wc) of the 2x4 wave grid computes the 4x2 set of
//     32x32 MFMA tiles {rows mi*64 + wr*32, cols nj*128 + wc*32}, i.e. every wave touches every 128-row "half" of the
//     A tile (x rows) and of the B tile (W rows) — so a half-tile is consumed by all waves in the same phase;
//   * a K-tile (64 bf16 / 128 fp8 = one 128-byte line per operand row) is staged as four 16 KiB half-tiles
//     B0 | A_a | B1 | A_b by LDS-DMA (buffer_load ... lds, 16 B/lane) into a ring of 8 slots (two K-tiles), LEAD
//     half-tiles ahead of their use; the swizzle  chunk ^= (row>>1)&7  is applied on the per-lane SOURCE offset and
//     on the ds_read_b128 address (the DMA destination is lane-linear);
//   * a K-tile is four phases of 16 MFMAs (bf16: 16x16x32, two per 32x32 quadrant pair; fp8: 4 of the MX-scaled 32x32x64) per wave:
//         q0: read A_a      -> acc[0..1][0] += A_a.B0        q1: read B1 -> acc[0..1][1] += A_a.B1
//         q2: read A_b      -> acc[2..3][0] += A_b.B0        q3: read next tile's B0 -> acc[2..3][1] += A_b.B1
//     each phase = {fragment ds_reads + one half-tile of DMA issue | s_barrier | MFMAs | counted vmcnt | s_barrier};
//     the wr = 1 waves run one barrier behind the wr = 0 waves, so on every SIMD (which hosts one wave of each
//     group) one wave is in its MFMA segment while the other is in its load segment;
//   * loads stay in flight across barriers: the only waits are `s_waitcnt vmcnt(2*KW)` (never 0 before the tail), raw
//     s_barrier and lgkmcnt — nothing the compiler would drain.
//   Hazard accounting (n = global phase index, half g is issued in phase g - LEAD and read in phase g - 1):
//     RAW: a wave's wait at the end of phase m covers its halves issued in phases <= m - KW; a wr=0 wave reading in
//          phase n+1 sits behind the wr=1 waves' wait of phase n-1  =>  g - LEAD <= n - 1 - KW with g = n + 2, i.e.
//          LEAD >= KW + 3.
//     WAR: the slot of half g last held g - 8, read in phase g - 9 = n + LEAD - 9 <= n - 2  =>  LEAD <= 7.
//   * M / N tails: the buffer descriptors are based at the tile's first row with num_records covering only its valid
//     rows, so rows past M (or N) read as zero through the hardware bounds check; stores are masked.
//   * epilogue through LDS ([256][528 B]) so global stores are 16 bytes per lane, 512 bytes contiguous per row.
#include "x2v_common.h"

namespace x2v {

constexpr int T_M = 256, T_N = 256;
constexpr int T_HALF_BYTES = 128 * 128;     // one half-tile: 128 operand rows x 128 B
constexpr int T_EPI_LD = 528;               // bytes per epilogue row (256 bf16 + 16 pad)
constexpr int T_LDS_BYTES = 256 * T_EPI_LD; // 135168 >= 8 * T_HALF_BYTES

typedef __attribute__((address_space(3))) void* t_lds_ptr_t;

template <bool FP8, int EPI, int LEAD, int KW, bool MX>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const char* __restrict__ A, int64_t lda_bytes, const char* __restrict__ W, int64_t ldw_bytes,
                                                         const unsigned short* __restrict__ bias, unsigned short* __restrict__ Y, int64_t ldy, int64_t M,
                                                         int N, int nk, const unsigned short* __restrict__ resid, int64_t ldr,
                                                         const unsigned short* __restrict__ gate, const float* __restrict__ sx,
                                                         const float* __restrict__ sw, int ntm, int ntn, int gm_tiles,
                                                         const unsigned* __restrict__ SA, const unsigned* __restrict__ SB, GemmBlocking gb) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(LEAD >= KW + 3 && LEAD <= 7 && LEAD >= 4, "see the hazard accounting in the header comment (and the A-half K offsets ak1/ak2)");
  static_assert(!MX || FP8, "MX: block-scaled fp8");
  // bf16 runs on v_mfma_f32_16x16x32_bf16: at the board's power limit (where this kernel lives) the 16x16 shape delivers ~11 % more FLOP/s
  // than 32x32x16 for the same LDS fragment traffic (tools/probes/mfma_power_probe.hip, profiles/r02_mfma_power_probe.log)
  constexpr bool MI16 = !FP8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  const int fl = lane & 31, fh = lane >> 5;

  // ---- tile coordinates: XCD chunking + grouped ordering (gm_tiles m-tiles x all n-tiles per group)
  const unsigned nblk = (unsigned)ntm * (unsigned)ntn;
  const unsigned v = xcd_remap(blockIdx.x, nblk);
  const unsigned GM = (unsigned)gm_tiles;
  const unsigned per_group = GM * (unsigned)ntn;
  const unsigned group = v / per_group, in_g = v % per_group;
  const unsigned first_m = group * GM;
  const unsigned gsz = min((unsigned)ntm - first_m, GM);
  const int tm = (int)(first_m + in_g % gsz), tn = (int)(in_g / gsz);
  const int64_t m0 = (int64_t)tm * T_M;
  const int n0 = tn * T_N;

  // ---- buffer descriptors over this tile's valid rows (wave-uniform: kernel arguments + blockIdx only)
  const unsigned row_bytes = (unsigned)nk * 128u;
  const int rows_a = (int)min((int64_t)T_M, M - m0), rows_w = min(T_N, N - n0);
  // K-blocked x (GemmBlocking, x2v_common.h): K-tile kt of a row lives at (kt / a_kpb) * a_cbs + (kt % a_kpb) * 128 bytes; the range covers
  // the last block (rows past M of an earlier block alias the next block's rows: their results are never stored)
  const int a_kpb = gb.a_kpb > 0 && gb.a_kpb < nk ? gb.a_kpb : nk;
  const unsigned a_span = a_kpb < nk ? (unsigned)((nk - 1) / a_kpb) * gb.a_cbs + (unsigned)a_kpb * 128u : row_bytes;
  const __amdgpu_buffer_rsrc_t ra =
      __builtin_amdgcn_make_buffer_rsrc((void*)(A + m0 * lda_bytes), 0, (unsigned)((rows_a - 1) * lda_bytes) + a_span, 0x00020000);
  // byte offsets of K-tiles t+1 and t+2 of x (the two tiles whose A halves the steady state issues) and their index within a K-block
  unsigned ak1 = 0, ak2 = 0;
  int akc1 = 0, akc2 = 0;
#define T_AK_NEXT(OFF_, CNT_) { if (++(CNT_) == a_kpb) { (CNT_) = 0; (OFF_) += gb.a_cbs - (unsigned)(a_kpb - 1) * 128u; } else (OFF_) += 128u; }
  T_AK_NEXT(ak1, akc1)
  ak2 = ak1;
  akc2 = akc1;
  T_AK_NEXT(ak2, akc2)
  const unsigned ak_first = ak1;  // tile 1, for the prologue
  const __amdgpu_buffer_rsrc_t rw =
      __builtin_amdgcn_make_buffer_rsrc((void*)(W + (int64_t)n0 * ldw_bytes), 0, (unsigned)((rows_w - 1) * ldw_bytes) + row_bytes, 0x00020000);

  // ---- per-lane DMA source offsets: wave `wid` stages rows [wid*16, wid*16+16) of a half (2 wave-instructions of
  //      8 rows x 8 chunks); [h*2+i] = half h (rows +128), instruction i.  The K offset travels in soffset.
  unsigned a_voff[4], w_voff[4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (wid * 2 + i) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      a_voff[h * 2 + i] = (unsigned)((h * 128 + r) * lda_bytes) + (unsigned)(c << 4);
      w_voff[h * 2 + i] = (unsigned)((h * 128 + r) * ldw_bytes) + (unsigned)(c << 4);
    }
  // half J of a K-tile: 0 = B0 (W rows 0..127), 1 = A_a (x rows 0..127), 2 = B1 (W rows 128..255), 3 = A_b
#define T_ISSUE1(J_, SLOT_, KOFF_, I_)                                                                                         \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(((J_) & 1) ? ra : rw, (t_lds_ptr_t)(smem + (SLOT_) * T_HALF_BYTES + (wid * 2 + (I_)) * 1024), 16, \
                                           ((J_) & 1) ? a_voff[((J_) >> 1) * 2 + (I_)] : w_voff[((J_) >> 1) * 2 + (I_)], (unsigned)(KOFF_), 0, 0);
#define T_ISSUE(J_, SLOT_, KOFF_) { T_ISSUE1(J_, SLOT_, KOFF_, 0) T_ISSUE1(J_, SLOT_, KOFF_, 1) }

  // ---- fragment read addresses: row (32-row block base + fl), 16-byte chunk index kpart(ks) ^ lanepart
  //      bf16: step ks (K=16) reads chunk ks*2+fh; fp8: MFMA s (K=64) reads chunks s*4+fh and s*4+2+fh (ks = s*2+e) — the
  //      instruction's own k order (first 16 bytes of a lane = k fh*16.., last 16 = k 32+fh*16..), which matters once the
  //      hardware applies block scales (MX); with unit scales any map shared by A and B would do.
  int rd_a[4], rd_b[4];
  if constexpr (MI16) {
    // 16x16x32: a lane reads row r16 of a 16-row block, 16-byte chunk ks*4 + g16 (k = 8 g16 .. 8 g16 + 7 of the 32-wide step); slot b*2 + ks =
    // 16-row block b of the wave's 32 rows, k-step ks.  With the same chunk ^= (row>>1)&7 image the four lane groups of a ds_read_b128 still
    // cover sixteen distinct 16-byte columns of the 256-byte bank line (rows of equal parity never share a chunk within a group).
    const int r16 = lane & 15, g16 = lane >> 4;
    const int swz = (r16 >> 1) & 7;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int o = (b * 16 + r16) * 128 + ((((ks << 2) | g16) ^ swz) << 4);
        rd_a[b * 2 + ks] = o + wr * 4096;
        rd_b[b * 2 + ks] = o + wc * 4096;
      }
  } else {
    const int swz = (fl >> 1) & 7;
    const int lp = fh ^ swz;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kp = FP8 ? (((ks >> 1) << 2) | ((ks & 1) << 1)) : (ks << 1);
      const int o = fl * 128 + ((lp ^ kp) << 4);
      rd_a[ks] = o + wr * 4096;
      rd_b[ks] = o + wc * 4096;
    }
  }

  // accumulators of the wave's 4 x 2 set of 32x32 regions: one 32x32 MFMA tile each (fp8), or 2 x 2 tiles of 16x16 (bf16): acc4[mi][nj][xh*2+wh]
  f32x16_t acc[4][2];
  f32x4_t acc4[4][2][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        acc[i][j][e] = 0.f;
        acc4[i][j][e >> 2][e & 3] = 0.f;
      }
  i32x4_t xa[2][4], wb0[4], wb1[4];

  // ---- MX block scales: per operand row and K-tile one dword = the 4 e8m0 bytes of its 32-wide k blocks, tables [K/128][rows]
  //      (mx.hip).  A lane needs the dwords of its 4 x rows (mi) and 2 W rows (nj) — 32 consecutive rows per load = one 128-byte
  //      line; they are fetched one K-tile ahead by plain buffer loads issued in phase 0 BEFORE that phase's DMA pair, so the counted
  //      vmcnt waits stay exact: phases 0..KW-1 allow NSC more loads in flight.  Lanes of the upper half (k block 2s+1) shift their
  //      dword down one byte so that opsel = 2s selects the right byte in both halves.
  constexpr int NSC = MX ? 6 : 0;
  unsigned sx_cur[4] = {0, 0, 0, 0}, sw_cur[2] = {0, 0}, sx_nxt[4] = {0, 0, 0, 0}, sw_nxt[2] = {0, 0};
  __amdgpu_buffer_rsrc_t rsa = ra, rsb = rw;
  unsigned sa_voff = 0, sb_voff = 0;
  if constexpr (MX) {
    // whole tables behind the descriptors; rows of a partial last tile read a neighbour's scales (their outputs are never stored) or,
    // past the end of the table, zero
    rsa = __builtin_amdgcn_make_buffer_rsrc((void*)SA, 0, (unsigned)((int64_t)nk * M * 4), 0x00020000);
    rsb = __builtin_amdgcn_make_buffer_rsrc((void*)SB, 0, (unsigned)((int64_t)nk * N * 4), 0x00020000);
    sa_voff = (unsigned)((m0 + wr * 32 + fl) * 4);
    sb_voff = (unsigned)((n0 + wc * 32 + fl) * 4);
  }
#define T_SC_LOAD(T_)                                                                                                          \
  {                                                                                                                            \
    _Pragma("unroll") for (int mi_ = 0; mi_ < 4; ++mi_)                                                                        \
      sx_nxt[mi_] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsa, sa_voff, (unsigned)((int64_t)(T_) * M * 4) + (unsigned)(mi_ * 256), 0); \
    _Pragma("unroll") for (int nj_ = 0; nj_ < 2; ++nj_)                                                                        \
      sw_nxt[nj_] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsb, sb_voff, (unsigned)((int64_t)(T_) * N * 4) + (unsigned)(nj_ * 512), 0); \
  }
#define T_SC_ADOPT()                                                                                                           \
  {                                                                                                                            \
    _Pragma("unroll") for (int mi_ = 0; mi_ < 4; ++mi_) sx_cur[mi_] = sx_nxt[mi_] >> (fh * 8);                                 \
    _Pragma("unroll") for (int nj_ = 0; nj_ < 2; ++nj_) sw_cur[nj_] = sw_nxt[nj_] >> (fh * 8);                                 \
  }

  const int total = nk * 4;  // half-tiles in this tile's K loop

#define T_RD(DST_, SLOT_, RD_, EXTRA_) \
  _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_) DST_[ks_] = *reinterpret_cast<const i32x4_t*>(smem + (SLOT_) * T_HALF_BYTES + (EXTRA_) + RD_[ks_]);

  // 8 MFMAs (fp8: 4) of one phase
#define T_MFMA(MI0_, NJ_, WB_, I0_, I1_)                                                                                                 \
  if constexpr (MI16) {                                                                                                        \
    _Pragma("unroll") for (int idx_ = (I0_); idx_ < (I1_); ++idx_) {                                                            \
      const int ks_ = idx_ >> 3, wh_ = (idx_ >> 2) & 1, i_ = (idx_ >> 1) & 1, xh_ = idx_ & 1;                                  \
      acc4[(MI0_) + i_][NJ_][xh_ * 2 + wh_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                         \
          __builtin_bit_cast(bf16x8_t, WB_[wh_ * 2 + ks_]), __builtin_bit_cast(bf16x8_t, xa[i_][xh_ * 2 + ks_]), acc4[(MI0_) + i_][NJ_][xh_ * 2 + wh_], 0, 0, 0); \
    }                                                                                                                          \
  } else if constexpr (!FP8) {                                                                                                 \
    _Pragma("unroll") for (int idx_ = (I0_) / 2; idx_ < (I1_) / 2; ++idx_) {                                                                   \
      const int ks_ = idx_ >> 1, i_ = idx_ & 1;                                                                                \
      acc[(MI0_) + i_][NJ_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, WB_[ks_]),                  \
                                                                      __builtin_bit_cast(bf16x8_t, xa[i_][ks_]), acc[(MI0_) + i_][NJ_], 0, 0, 0); \
    }                                                                                                                          \
  } else {                                                                                                                     \
    constexpr int kOne = 0x7f7f7f7f; /* e8m0 2^0 block scales */                                                               \
    _Pragma("unroll") for (int idx_ = (I0_) / 4; idx_ < (I1_) / 4; ++idx_) {                                                   \
      const int s_ = idx_ >> 1, i_ = idx_ & 1;                                                                                 \
      const i32x8_t wf_ = __builtin_shufflevector(WB_[2 * s_], WB_[2 * s_ + 1], 0, 1, 2, 3, 4, 5, 6, 7);                       \
      const i32x8_t xf_ = __builtin_shufflevector(xa[i_][2 * s_], xa[i_][2 * s_ + 1], 0, 1, 2, 3, 4, 5, 6, 7);                 \
      if constexpr (!MX)                                                                                                       \
        acc[(MI0_) + i_][NJ_] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf_, xf_, acc[(MI0_) + i_][NJ_], 0, 0, 0, kOne, 0, kOne); \
      else if (s_ == 0)                                                                                                        \
        acc[(MI0_) + i_][NJ_] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf_, xf_, acc[(MI0_) + i_][NJ_], 0, 0, 0, (int)sw_cur[NJ_], 0, (int)sx_cur[(MI0_) + i_]); \
      else                                                                                                                     \
        acc[(MI0_) + i_][NJ_] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf_, xf_, acc[(MI0_) + i_][NJ_], 0, 0, 2, (int)sw_cur[NJ_], 2, (int)sx_cur[(MI0_) + i_]); \
    }                                                                                                                          \
  }

#define T_MFMA_Q(Q_, I0_, I1_)                                                                                                 \
  if constexpr ((Q_) == 0) { T_MFMA(0, 0, wb0, I0_, I1_) }                                                                     \
  else if constexpr ((Q_) == 1) { T_MFMA(0, 1, wb1, I0_, I1_) }                                                                \
  else if constexpr ((Q_) == 2) { T_MFMA(2, 0, wb0, I0_, I1_) }                                                                \
  else { T_MFMA(2, 1, wb1, I0_, I1_) }

#define T_PIN(MI0_, NJ_)                                                                                                       \
  if constexpr (MI16)                                                                                                          \
    asm volatile("" : "+v"(acc4[(MI0_)][NJ_][0]), "+v"(acc4[(MI0_)][NJ_][1]), "+v"(acc4[(MI0_)][NJ_][2]), "+v"(acc4[(MI0_)][NJ_][3]), \
                 "+v"(acc4[(MI0_) + 1][NJ_][0]), "+v"(acc4[(MI0_) + 1][NJ_][1]), "+v"(acc4[(MI0_) + 1][NJ_][2]), "+v"(acc4[(MI0_) + 1][NJ_][3])); \
  else                                                                                                                         \
    asm volatile("" : "+v"(acc[(MI0_)][NJ_]), "+v"(acc[(MI0_) + 1][NJ_]));

  // One phase.  TB_ = K-tile parity (compile time), Q_ = phase within the tile, t = K-tile index (run time);
  // CHK_ = 0: steady state (every phase issues its half-tile; counted wait), 1: tail (issue / wait by run-time test).
#define T_PHASE(TB_, Q_, CHK_)                                                                                                 \
  {                                                                                                                            \
    if constexpr ((Q_) == 0) {                                                                                                 \
      T_RD(xa[0], (TB_) * 4 + 1, rd_a, 0) T_RD(xa[1], (TB_) * 4 + 1, rd_a, 8192)                                               \
    } else if constexpr ((Q_) == 1) {                                                                                          \
      T_RD(wb1, (TB_) * 4 + 2, rd_b, 0)                                                                                        \
    } else if constexpr ((Q_) == 2) {                                                                                          \
      T_RD(xa[0], (TB_) * 4 + 3, rd_a, 0) T_RD(xa[1], (TB_) * 4 + 3, rd_a, 8192)                                               \
    } else {                                                                                                                   \
      T_RD(wb0, ((TB_) ^ 1) * 4 + 0, rd_b, 0)                                                                                  \
    }                                                                                                                          \
    const bool live_ = (CHK_) ? (4 * t + (Q_) + LEAD < total) : true;                                                          \
    if constexpr (MX && (Q_) == 0) {                                                                                           \
      if ((CHK_) ? (t + 1 < nk) : true) T_SC_LOAD(t + 1)                                                                        \
    }                                                                                                                          \
    if (live_) T_ISSUE(((Q_) + LEAD) & 3, ((TB_) * 4 + (Q_) + LEAD) & 7, ((((Q_) + LEAD) & 1) ? ((((Q_) + LEAD) / 4) == 1 ? ak1 : ak2) : (unsigned)((t + ((Q_) + LEAD) / 4) * 128)))          \
    __builtin_amdgcn_sched_barrier(0);                                                                                         \
    __builtin_amdgcn_s_barrier();                                                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                         \
    /* MFMAs are pure to the optimizer (it may sink them past barriers, to the loop latch): pin this phase's two          \
       accumulators in place with empty asm statements on both sides of the cluster */                                       \
    T_PIN(((Q_) >> 1) * 2, (Q_) == 1 || (Q_) == 3)                                                                             \
    __builtin_amdgcn_s_setprio(1);                                                                                             \
    T_MFMA_Q(Q_, 0, 16)                                                                                                        \
    __builtin_amdgcn_s_setprio(0);                                                                                             \
    T_PIN(((Q_) >> 1) * 2, (Q_) == 1 || (Q_) == 3)                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                                         \
    /* steady state: the newest KW phases' loads may stay in flight = 2 DMA pieces each, plus the NSC scale loads of phase 0 */ \
    if (live_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * KW + ((CHK_) == 0 && (Q_) < KW ? NSC : 0)) : "memory");            \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                      \
    if constexpr (MX && (Q_) == 3) T_SC_ADOPT()                                                                                \
    /* (issuing the last 4 or 8 MFMAs of the cluster behind this barrier, so that the other wave of the SIMD starts while they   \
       drain, measured 8-10 % SLOWER: profiles/r02_gemm256_mi16_trail_ab.log) */                                                \
    __builtin_amdgcn_s_barrier();                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                                         \
  }

#define T_AK_ADVANCE() { ak1 = ak2; akc1 = akc2; T_AK_NEXT(ak2, akc2) }
  // ---- prologue: halves 0 .. LEAD-1 in flight, wait for B0 and A_a of K-tile 0, read B0
  {
    const int t = 0;
    (void)t;
    if constexpr (MX) T_SC_LOAD(0)  // oldest loads in flight: complete after the prologue wait below
#pragma unroll
    for (int g = 0; g < LEAD; ++g)
      if (g < total) T_ISSUE(g & 3, g & 7, ((g & 1) ? ((g >> 2) ? ak_first : 0u) : (unsigned)((g >> 2) * 128)))
    // = the steady-state wait of "phase -1": leaves KW halves in flight, so B1 of K-tile 0 (read in phase 1 by the
    // wr = 0 waves, which see no later wait of the wr = 1 waves) has landed too
    if (total >= LEAD) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * KW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (MX) T_SC_ADOPT()
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    T_RD(wb0, 0, rd_b, 0)
  }
  if (wr == 1) __builtin_amdgcn_s_barrier();  // the wr = 1 waves run one barrier behind (ping-pong)
  __builtin_amdgcn_sched_barrier(0);

  // steady state: K-tiles [0, nmain) — every phase of a tile t <= nk - 3 still has a half-tile to issue
  const int nmain = nk > 2 ? ((nk - 2) & ~1) : 0;
  for (int tt = 0; tt < nmain; tt += 2) {
    {
      const int t = tt;
      T_PHASE(0, 0, 0) T_PHASE(0, 1, 0) T_PHASE(0, 2, 0) T_PHASE(0, 3, 0)
      T_AK_ADVANCE()
    }
    {
      const int t = tt + 1;
      T_PHASE(1, 0, 0) T_PHASE(1, 1, 0) T_PHASE(1, 2, 0) T_PHASE(1, 3, 0)
      T_AK_ADVANCE()
    }
  }
  for (int tt = nmain; tt < nk; tt += 2) {
    {
      const int t = tt;
      T_PHASE(0, 0, 1) T_PHASE(0, 1, 1) T_PHASE(0, 2, 1) T_PHASE(0, 3, 1)
      T_AK_ADVANCE()
    }
    if (tt + 1 < nk) {
      const int t = tt + 1;
      T_PHASE(1, 0, 1) T_PHASE(1, 1, 1) T_PHASE(1, 2, 1) T_PHASE(1, 3, 1)
      T_AK_ADVANCE()
    }
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();  // re-align the two groups; every wave is past its last LDS read
  __builtin_amdgcn_sched_barrier(0);
#undef T_PHASE
#undef T_AK_ADVANCE
#undef T_AK_NEXT
#undef T_PIN
#undef T_MFMA
#undef T_MFMA_Q
#undef T_RD
#undef T_ISSUE
#undef T_ISSUE1
#undef T_SC_LOAD
#undef T_SC_ADOPT

  // ---- epilogue phase 1: acc (+scales, bias, activation) -> bf16 -> LDS [256][T_EPI_LD]
  //      32x32 tiles: acc[mi][nj][4g+e]  = tile row mi*64 + wr*32 + fl,               tile col nj*128 + wc*32 + 8*g + 4*fh + e
  //      16x16 tiles: acc4[mi][nj][g][e] = tile row mi*64 + wr*32 + (g>>1)*16 + r16,  tile col nj*128 + wc*32 + (g&1)*16 + 4*g16 + e
  const int e_r16 = lane & 15, e_g16 = lane >> 4;
  //      All per-column operands (bias, fp8 channel scales) are fetched up front in one batch: one L2 round trip.
  uint2 bv[2][4];
  float4 swv[2][4];
  float sxv[4];
#pragma unroll
  for (int nj = 0; nj < 2; ++nj)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      int gn = n0 + nj * 128 + wc * 32 + (MI16 ? (g & 1) * 16 + 4 * e_g16 : 8 * g + 4 * fh);
      gn = gn + 3 < N ? gn : (N >= 4 ? N - 4 : 0);
      bv[nj][g] = make_uint2(0u, 0u);
      if (bias != nullptr) bv[nj][g] = *reinterpret_cast<const uint2*>(bias + gn);
      if constexpr (FP8 && !MX) swv[nj][g] = *reinterpret_cast<const float4*>(sw + gn);
    }
  float mx_alpha = 1.0f;
  if constexpr (MX) {
    if (sx != nullptr) mx_alpha = *sx;  // MX: `sx` carries the device pointer to alpha (x2v_gemm_mxfp8)
  } else if constexpr (FP8) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      int64_t gmr = m0 + mi * 64 + wr * 32 + fl;
      gmr = gmr < M ? gmr : M - 1;
      sxv[mi] = sx[gmr];
    }
  }
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ml = mi * 64 + wr * 32 + (MI16 ? (g >> 1) * 16 + e_r16 : fl);
        const int nl = nj * 128 + wc * 32 + (MI16 ? (g & 1) * 16 + 4 * e_g16 : 8 * g + 4 * fh);
        float vv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) vv[e] = MI16 ? acc4[mi][nj][g][e] : acc[mi][nj][4 * g + e];
        if constexpr (MX) {
#pragma unroll
          for (int e = 0; e < 4; ++e) vv[e] *= mx_alpha;
        } else if constexpr (FP8) {
          vv[0] = vv[0] * sxv[mi] * swv[nj][g].x;
          vv[1] = vv[1] * sxv[mi] * swv[nj][g].y;
          vv[2] = vv[2] * sxv[mi] * swv[nj][g].z;
          vv[3] = vv[3] * sxv[mi] * swv[nj][g].w;
        }
        vv[0] += bf_lo(bv[nj][g].x);
        vv[1] += bf_hi(bv[nj][g].x);
        vv[2] += bf_lo(bv[nj][g].y);
        vv[3] += bf_hi(bv[nj][g].y);
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
        *reinterpret_cast<uint2*>(smem + ml * T_EPI_LD + nl * 2) = pk;
      }
    }
  }
  __syncthreads();
  // ---- epilogue phase 2: 16-byte stores, 32 lanes per 512-byte output row
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int id = it * 512 + tid;
    const int row = id >> 5, cc = id & 31;
    const int64_t gmr = m0 + row;
    const int gn = n0 + cc * 8;
    if (gmr < M && gn < N) {
      // N-blocked y (GemmBlocking): column block gn / y_cbw starts y_cbs elements after the previous one
      const int64_t ycol = gb.y_cbw > 0 ? (int64_t)(gn / gb.y_cbw) * gb.y_cbs + gn % gb.y_cbw : gn;
      uint4 o = *reinterpret_cast<const uint4*>(smem + row * T_EPI_LD + cc * 16);
      if (EPI == X2V_EPI_RESIDUAL) {
        float yv[8], xv[8], ov[8];
        unpack8(o, yv);
        unpack8(*reinterpret_cast<const uint4*>(resid + gmr * ldr + gn), xv);
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
      *reinterpret_cast<uint4*>(Y + gmr * ldy + ycol) = o;
    }
  }
#endif
}

template <bool FP8, int EPI, bool MX = false>
static int launch_gemm256(const void* x, int64_t ldx_bytes, const void* w, int64_t ldw_bytes, const void* bias, void* y, int64_t ldy, int64_t M, int N,
                          int nk, const void* resid, int64_t ldr, const void* gate, const float* sx, const float* sw, int gm_tiles, hipStream_t st,
                          const void* sa = nullptr, const void* sb = nullptr, GemmBlocking gb = GemmBlocking()) {
  constexpr int LEAD = 6, KW = 3;
  if (gm_tiles <= 0) gm_tiles = 4;  // m-tiles per scheduling group
  const int ntm = (int)((M + T_M - 1) / T_M), ntn = (N + T_N - 1) / T_N;
  {
    int rc = ensure_dynamic_lds((const void*)gemm256_kernel<FP8, EPI, LEAD, KW, MX>, T_LDS_BYTES, "gemm256 attr");
    if (rc != X2V_OK) return rc;
  }
  hipLaunchKernelGGL((gemm256_kernel<FP8, EPI, LEAD, KW, MX>), dim3((unsigned)ntm * (unsigned)ntn), dim3(512), T_LDS_BYTES, st, (const char*)x, ldx_bytes,
                     (const char*)w, ldw_bytes, (const unsigned short*)bias, (unsigned short*)y, ldy, M, N, nk, (const unsigned short*)resid, ldr,
                     (const unsigned short*)gate, sx, sw, ntm, ntn, gm_tiles, (const unsigned*)sa, (const unsigned*)sb, gb);
  X2V_LAUNCH_CHECK("gemm256 launch");
  return X2V_OK;
}

// Called by gemm.hip's dispatcher (arguments already validated there).  ld*_bytes < 16 MiB is required (32-bit
// buffer offsets over a 256-row tile) and checked by the caller.
template <bool FP8>
int gemm256_dispatch(int epilogue, const void* x, int64_t ldxb, const void* w, int64_t ldwb, const void* bias, void* y, int64_t ldy, int64_t M, int N, int nk,
                     const void* resid, int64_t ldr, const void* gate, const float* sx, const float* sw, int gm_tiles, hipStream_t st, GemmBlocking gb) {
  switch (epilogue) {
    case X2V_EPI_NONE: return launch_gemm256<FP8, X2V_EPI_NONE>(x, ldxb, w, ldwb, bias, y, ldy, M, N, nk, nullptr, 0, nullptr, sx, sw, gm_tiles, st, nullptr, nullptr, gb);
    case X2V_EPI_GELU_TANH: return launch_gemm256<FP8, X2V_EPI_GELU_TANH>(x, ldxb, w, ldwb, bias, y, ldy, M, N, nk, nullptr, 0, nullptr, sx, sw, gm_tiles, st, nullptr, nullptr, gb);
    case X2V_EPI_SILU: return launch_gemm256<FP8, X2V_EPI_SILU>(x, ldxb, w, ldwb, bias, y, ldy, M, N, nk, nullptr, 0, nullptr, sx, sw, gm_tiles, st, nullptr, nullptr, gb);
    case X2V_EPI_RESIDUAL: return launch_gemm256<FP8, X2V_EPI_RESIDUAL>(x, ldxb, w, ldwb, bias, y, ldy, M, N, nk, resid, ldr, gate, sx, sw, gm_tiles, st, nullptr, nullptr, gb);
    default: set_error("gemm: unknown epilogue %d", epilogue); return X2V_E_ARG;
  }
}
// MXFP8 (mx.hip): e4m3 operands with e8m0 block-scale tables [K/128][rows][4]; alpha = device pointer or null.  Arguments validated by the caller.
int gemm256_mx_dispatch(int epilogue, const void* a, int64_t lda, const void* sa, const void* b, int64_t ldb, const void* sb, const void* bias, const float* alpha,
                        void* y, int64_t ldy, int64_t M, int N, int nk, const void* resid, int64_t ldr, const void* gate, hipStream_t st) {
  switch (epilogue) {
    case X2V_EPI_NONE: return launch_gemm256<true, X2V_EPI_NONE, true>(a, lda, b, ldb, bias, y, ldy, M, N, nk, nullptr, 0, nullptr, alpha, nullptr, 4, st, sa, sb);
    case X2V_EPI_GELU_TANH: return launch_gemm256<true, X2V_EPI_GELU_TANH, true>(a, lda, b, ldb, bias, y, ldy, M, N, nk, nullptr, 0, nullptr, alpha, nullptr, 4, st, sa, sb);
    case X2V_EPI_SILU: return launch_gemm256<true, X2V_EPI_SILU, true>(a, lda, b, ldb, bias, y, ldy, M, N, nk, nullptr, 0, nullptr, alpha, nullptr, 4, st, sa, sb);
    case X2V_EPI_RESIDUAL: return launch_gemm256<true, X2V_EPI_RESIDUAL, true>(a, lda, b, ldb, bias, y, ldy, M, N, nk, resid, ldr, gate, alpha, nullptr, 4, st, sa, sb);
    default: set_error("gemm_mxfp8: unknown epilogue %d", epilogue); return X2V_E_ARG;
  }
}

template int gemm256_dispatch<false>(int, const void*, int64_t, const void*, int64_t, const void*, void*, int64_t, int64_t, int, int, const void*, int64_t,
                                     const void*, const float*, const float*, int, hipStream_t, GemmBlocking);
template int gemm256_dispatch<true>(int, const void*, int64_t, const void*, int64_t, const void*, void*, int64_t, int64_t, int, int, const void*, int64_t,
                                    const void*, const float*, const float*, int, hipStream_t, GemmBlocking);

}  // namespace x2v
