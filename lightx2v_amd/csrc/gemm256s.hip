// y[M,N] = epi(x[M,K] . W[N,K]^T + bias), bf16 — the large-shape GEMM as ONE wave per SIMD running a single software-pipelined stream.
// Same contract, operand layouts, epilogues and rounding points as gemm.hip / gemm256.hip.
//
// Bound: MFMA (bf16 dense peak ~2.5 PFLOP/s; at this board's 1400 W limit a bare 16x16x32 MFMA loop with GEMM-like LDS traffic sustains
// ~1.8 PFLOP/s, tools/probes/mfma_power_probe.hip).  Algorithmic work 2*M*N*K FLOP per launch.
//
// Why a third structure.  gemm256.hip keeps two waves per SIMD and hands the matrix pipe from one to the other through a barrier every
// 16 MFMAs; the pipe idles over every hand-off (85 % MFMA-busy) and each wave's 128x64 tile reads 0.75 KiB of LDS per MFMA-equivalent.
// Here a 256-thread workgroup owns the same 256x256 output tile with four waves of 128x128 (2 x 2), one per SIMD:
//   * 64 accumulator tiles of 16x16 per wave = the whole accumulator half of the register file (a[0:255]), addressed by literal
//     register number from the asm MFMA statements of this file only — hipcc never sees them (audit rule below);
//   * no hand-off: the wave's own stream is {MFMA | at most one ds_read_b128 or one LDS-DMA issue} per slot, 128 slots per 64-wide K tile,
//     pinned with sched_barrier; fragments for k-step s+1 are read while k-step s multiplies (0.5 KiB of LDS per MFMA-equivalent);
//   * two LDS stages of {W tile | x tile} (2 x 64 KiB, [256 rows][128 B] images with the chunk ^= (row>>1)&7 swizzle on the DMA source
//     offset and on the fragment address).  Per K tile t (stage t&1), in slots:
//        0..30   reads of k-step 1 of tile t        36  lgkmcnt(0) + s_barrier "stage t&1 is free"
//        37..    LDS-DMA of tile t+2 into stage t&1, one piece every 7th slot, running on into the first slots of tile t+1
//        94      counted vmcnt + s_barrier "tile t+1 has landed everywhere"      96..126  reads of k-step 0 of tile t+1
//     (slot constants below).
//   * M / N tails, K-blocked x and N-blocked y (GemmBlocking), the LDS-staged epilogue: as gemm256.hip.
// AUDIT after every edit (the accumulator half is invisible to the compiler): `hipcc -S` must show .vgpr_spill_count 0,
// .private_segment_fixed_size 0 and no v_accvgpr_* / a[..] operand outside ;;#ASMSTART / ;;#ASMEND.
#include <type_traits>

#include "x2v_common.h"

namespace x2v {

constexpr int S_M = 256, S_N = 256;
constexpr int S_OP_BYTES = 256 * 128;          // one operand tile of one stage
constexpr int S_STAGE_BYTES = 2 * S_OP_BYTES;  // W tile | x tile
constexpr int S_EPI_LD = 528;                  // bytes per epilogue row (256 bf16 + 16 pad)
constexpr int S_LDS_BYTES = 256 * S_EPI_LD;    // 135168 >= 2 stages (131072)
// MFMA slots of a K tile (128 per wave) at which the other instructions of the stream sit:
//   S_LATE0 + S_STEP i    the last 16 - S_EARLY LDS-DMA pieces of tile t+1            0, 2, .., 30   fragment reads of k-step 1
//   S_FREE                lgkmcnt(0) + barrier "this tile's stage is free"
//   S_FREE + 1 + S_STEP i the first S_EARLY pieces of tile t+2
//   S_READY               vmcnt + barrier "tile t+1 has landed"                      S_READY + 2, + 4, ..   fragment reads of k-step 0 of t+1
// The 64 pieces a workgroup moves per tile keep the CU's texture path busy for half of the tile's 2048 cycles: issued in a burst
// (all four waves right behind the first barrier) they queue up and stall the issuing waves; spread over the tile they cost ~nothing.
constexpr int S_STEP = 7, S_FREE = 36, S_READY = 94, S_LATE0 = 3;
constexpr int S_EARLY = (127 - S_FREE - 1) / S_STEP + 1 < 16 ? (127 - S_FREE - 1) / S_STEP + 1 : 16;  // pieces of tile t+2 that fit behind S_FREE
static_assert(S_LATE0 + (16 - S_EARLY - 1) * S_STEP < S_FREE && S_READY + 2 + 30 <= 127, "slot plan");

typedef __attribute__((address_space(3))) void* s_lds_ptr_t;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

// Every asm statement that touches the accumulator half names ALL of it as clobbered: hipcc must never park a value of its own in an AGPR
// across one of them.  (Round 3: with only a0 / a255 named once at kernel entry, the register allocator put part of the hoisted residual
// chunks into a1..a8 — `v_accvgpr_write` outside the asm blocks — and two accumulator tiles per wave were overwritten: the AUDIT rule in
// the header exists for exactly this.)
#define S_AGPRS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"

template <int B, int E, class F>
__device__ __forceinline__ void s_for(F&& f) {  // f(integral_constant<int, i>) for i = B .. E-1, fully unrolled with constant indices
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    s_for<B + 1, E>(f);
  }
}

// accumulator tile I (= x block * 8 + W block) is a[4 I : 4 I + 3]
template <int I>
__device__ __forceinline__ void s_mfma(const bf16x8_t& wf, const bf16x8_t& xf) {
  asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(wf), "v"(xf), "i"(4 * I), "i"(4 * I + 3) : S_AGPRS);
}
template <int R>
__device__ __forceinline__ void s_acc_zero() {
  asm volatile("v_accvgpr_write_b32 a[%c0], 0" ::"i"(R) : S_AGPRS);
}
template <int R>
__device__ __forceinline__ float s_acc_read() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(x) : "i"(R) : S_AGPRS);
  return x;
}

// VT (epilogue NONE only): y is written TRANSPOSED per head and 64-token block — V^T [N/128][ldy/64][128][64], the operand layout of the
// ping-pong attention kernel (x2v_transpose_heads_bf16's output), tokens in [M, ldy) zero-filled.  The MFMA operands swap roles, so a lane
// owns four consecutive TOKENS of one output channel and the staging / store code stays 8- and 16-byte wide; every (m, n) sums the same
// products in the same order as in the row-major form (bit-identical values).
template <int EPI, bool VT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm256s_kernel(
    const char* __restrict__ A, int64_t lda_bytes, const char* __restrict__ W, int64_t ldw_bytes, const unsigned short* __restrict__ bias,
    unsigned short* __restrict__ Y, int64_t ldy, int64_t M, int N, int nk, const unsigned short* __restrict__ resid, int64_t ldr,
    const unsigned short* __restrict__ gate, int ntm, int ntn, int gm_tiles, GemmBlocking gb) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  asm volatile("" ::: S_AGPRS);  // the accumulator half belongs to the asm statements of this kernel
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const int r16 = lane & 15, g16 = lane >> 4;

  // ---- tile coordinates: XCD chunking + grouped ordering (gm_tiles m-tiles x all n-tiles per group), as gemm256.hip
  const unsigned nblk = (unsigned)ntm * (unsigned)ntn;
  const unsigned v = xcd_remap(blockIdx.x, nblk);
  const unsigned GM = (unsigned)gm_tiles;
  const unsigned per_group = GM * (unsigned)ntn;
  const unsigned group = v / per_group, in_g = v % per_group;
  const unsigned first_m = group * GM;
  const unsigned gsz = min((unsigned)ntm - first_m, GM);
  const int tm = (int)(first_m + in_g % gsz), tn = (int)(in_g / gsz);
  const int64_t m0 = (int64_t)tm * S_M;
  const int n0 = tn * S_N;

  // ---- buffer descriptors over this tile's valid rows: rows past M / N read as zero through the bounds check
  const unsigned row_bytes = (unsigned)nk * 128u;
  const int rows_a = (int)min((int64_t)S_M, M - m0), rows_w = min(S_N, N - n0);
  const int a_kpb = gb.a_kpb > 0 && gb.a_kpb < nk ? gb.a_kpb : nk;  // K tiles per K block of x (GemmBlocking)
  const unsigned a_span = a_kpb < nk ? (unsigned)((nk - 1) / a_kpb) * gb.a_cbs + (unsigned)a_kpb * 128u : row_bytes;
  const __amdgpu_buffer_rsrc_t ra =
      __builtin_amdgcn_make_buffer_rsrc((void*)(A + m0 * lda_bytes), 0, (unsigned)((rows_a - 1) * lda_bytes) + a_span, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw =
      __builtin_amdgcn_make_buffer_rsrc((void*)(W + (int64_t)n0 * ldw_bytes), 0, (unsigned)((rows_w - 1) * ldw_bytes) + row_bytes, 0x00020000);
  // byte offsets of x's K tiles t+1 and t+2 within a row (K-blocked x: GemmBlocking) and the index of tile t+2 within its K block
  unsigned ak1 = 0, ak2 = 0;
  int akc2 = 0;
#define S_AK_NEXT(OFF_, CNT_) { if (++(CNT_) == a_kpb) { (CNT_) = 0; (OFF_) += gb.a_cbs - (unsigned)(a_kpb - 1) * 128u; } else (OFF_) += 128u; }
  S_AK_NEXT(ak2, akc2)
  ak1 = ak2;
  S_AK_NEXT(ak2, akc2)

  // ---- LDS-DMA: wave `wid` stages rows [64 wid, 64 wid + 64) of both operand tiles as 8 pieces of 8 rows (1 KiB, lane-linear in LDS).
  //      Piece i = 2 j + par: row 64 wid + 16 j + 8 par + (lane>>3); its swizzle (row>>1)&7 = ((lane>>4) + 4 par) & 7 does not depend on j,
  //      so two per-lane offsets per operand serve all pieces and 16 j rows travel in the scalar offset with the K offset.
  unsigned a_voff[2], w_voff[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int r = wid * 64 + par * 8 + (lane >> 3);
    const int c = (lane & 7) ^ (((lane >> 4) + 4 * par) & 7);
    a_voff[par] = (unsigned)(r * lda_bytes) + (unsigned)(c << 4);
    w_voff[par] = (unsigned)(r * ldw_bytes) + (unsigned)(c << 4);
  }
  const unsigned a_j = (unsigned)(16 * lda_bytes), w_j = (unsigned)(16 * ldw_bytes);
  // piece P_ in 0..15 of a K tile: 0..7 = W pieces, 8..15 = x pieces
#define S_DMA(P_, STAGE_, KW_, KA_)                                                                                                     \
  {                                                                                                                                    \
    constexpr int i_ = (P_) & 7;                                                                                                       \
    if constexpr ((P_) < 8)                                                                                                            \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (s_lds_ptr_t)(smem + (STAGE_) * S_STAGE_BYTES + wid * 8192 + i_ * 1024), 16, w_voff[i_ & 1],         \
                                               (unsigned)(KW_) + (unsigned)(i_ >> 1) * w_j, 0, 0);                                     \
    else                                                                                                                               \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (s_lds_ptr_t)(smem + (STAGE_) * S_STAGE_BYTES + S_OP_BYTES + wid * 8192 + i_ * 1024), 16,            \
                                               a_voff[i_ & 1], (unsigned)(KA_) + (unsigned)(i_ >> 1) * a_j, 0, 0);                     \
  }

  // ---- fragment addresses (16x16x32: row r16 of a 16-row block, 16-byte chunk ks*4 + g16), block offsets travel as immediates
  int rd_x[2], rd_w[2];
  {
    const int swz = (r16 >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int o = r16 * 128 + ((((ks << 2) | g16) ^ swz) << 4);
      rd_x[ks] = o + S_OP_BYTES + wr * 16384;
      rd_w[ks] = o + wc * 16384;
    }
  }
  bf16x8_t fx[2][8], fw[2][8];
  // fragment R_ in 0..15 of k-step KS_ of the tile in stage STAGE_; order x0, W0..W7, x1..x7 (the first MFMA of a k-step needs x0 and W0)
#define S_READ(R_, STAGE_, KS_)                                                                                                         \
  {                                                                                                                                    \
    if constexpr ((R_) == 0) fx[KS_][0] = *reinterpret_cast<const bf16x8_t*>(smem + (STAGE_) * S_STAGE_BYTES + rd_x[KS_]);              \
    else if constexpr ((R_) <= 8) fw[KS_][(R_) - 1] = *reinterpret_cast<const bf16x8_t*>(smem + (STAGE_) * S_STAGE_BYTES + ((R_) - 1) * 2048 + rd_w[KS_]); \
    else fx[KS_][(R_) - 8] = *reinterpret_cast<const bf16x8_t*>(smem + (STAGE_) * S_STAGE_BYTES + ((R_) - 8) * 2048 + rd_x[KS_]);       \
  }
#define S_SB() __builtin_amdgcn_sched_barrier(0)

  s_for<0, 256>([&](auto rc) { s_acc_zero<decltype(rc)::value>(); });

  // ---- prologue: tile 0 and the first S_EARLY pieces of tile 1 in flight, tile 0 landed, k-step 0 of tile 0 in registers
  s_for<0, 16>([&](auto pc) { S_DMA(decltype(pc)::value, 0, 0u, 0u) });
  if (nk > 1) {
    s_for<0, S_EARLY>([&](auto pc) { S_DMA(decltype(pc)::value, 1, 128u, ak1) });
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S_EARLY) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  S_SB();
  s_for<0, 16>([&](auto rc) { S_READ(decltype(rc)::value, 0, 0) });
  S_SB();

  // One K tile.  ST = its stage; CHK = 0: steady state (tiles t+1 and t+2 exist), 1: tail (run-time tests).
  auto tile = [&](auto stc, auto chkc, int t) {
    constexpr int ST = decltype(stc)::value;
    constexpr bool CHK = decltype(chkc)::value != 0;
    const bool has1 = CHK ? (t + 1 < nk) : true, has2 = CHK ? (t + 2 < nk) : true;
    const unsigned kw1 = (unsigned)(t + 1) * 128u, kw2 = (unsigned)(t + 2) * 128u;
    s_for<0, 128>([&](auto nc) {
      constexpr int n = decltype(nc)::value, ks = n >> 6, xb = (n >> 3) & 7, wb = n & 7;
      if constexpr (VT) s_mfma<xb * 8 + wb>(fx[ks][xb], fw[ks][wb]);
      else s_mfma<xb * 8 + wb>(fw[ks][wb], fx[ks][xb]);
      if constexpr (n < 32 && (n & 1) == 0) S_READ(n >> 1, ST, 1)  // k-step 1 of this tile
      // the last 16 - S_EARLY pieces of tile t+1 (its stage was freed by the previous tile's first barrier)
      if constexpr (n >= S_LATE0 && (n - S_LATE0) % S_STEP == 0 && (n - S_LATE0) / S_STEP < 16 - S_EARLY) {
        if (has1) S_DMA(S_EARLY + (n - S_LATE0) / S_STEP, ST ^ 1, kw1, ak1)
      }
      if constexpr (n == S_FREE) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every fragment of this tile is in registers: the stage may be overwritten
        __builtin_amdgcn_s_barrier();
      }
      // the first S_EARLY pieces of tile t+2 into this tile's stage
      if constexpr (n > S_FREE && (n - S_FREE - 1) % S_STEP == 0 && (n - S_FREE - 1) / S_STEP < S_EARLY) {
        if (has2) S_DMA((n - S_FREE - 1) / S_STEP, ST, kw2, ak2)
      }
      if constexpr (n == S_READY) {
        if (has1) {
          // tile t+1 has landed; the pieces of tile t+2 issued so far in this tile may stay in flight
          constexpr int newer = (S_READY - S_FREE - 1) / S_STEP + 1 < S_EARLY ? (S_READY - S_FREE - 1) / S_STEP + 1 : S_EARLY;
          if (has2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(newer) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
        }
      }
      if constexpr (n > S_READY + 1 && (n & 1) == 0) {
        if (has1) S_READ((n - S_READY - 2) >> 1, ST ^ 1, 0)  // k-step 0 of the next tile
      }
      S_SB();
    });
    ak1 = ak2;
    if (has2) S_AK_NEXT(ak2, akc2)
  };
  using c0 = std::integral_constant<int, 0>;
  using c1 = std::integral_constant<int, 1>;
  int t = 0;
  const int nmain = nk > 2 ? ((nk - 2) & ~1) : 0;  // tiles [0, nmain): t + 2 < nk throughout
  for (; t < nmain; t += 2) {
    tile(c0{}, c0{}, t);
    tile(c1{}, c0{}, t + 1);
  }
  for (; t < nk; t += 2) {
    tile(c0{}, c1{}, t);
    if (t + 1 < nk) tile(c1{}, c1{}, t + 1);
  }
#undef S_READ
#undef S_DMA
#undef S_AK_NEXT
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");  // the last MFMAs' results before the accumulator reads below
  __builtin_amdgcn_s_barrier();                                 // every wave is past its last fragment read: LDS becomes the staging area
  S_SB();

  if constexpr (VT) {
    static_assert(EPI == X2V_EPI_NONE, "V^T output: plain epilogue only");
    // ---- phase 1: tile (xb, wb), register e: channel wc*128 + wb*16 + r16, token wr*128 + xb*16 + 4*g16 + e  ->  LDS [256 channels][S_EPI_LD]
    float bvt[8];
#pragma unroll
    for (int wb = 0; wb < 8; ++wb) {
      const int gn = n0 + wc * 128 + wb * 16 + r16;
      bvt[wb] = (bias != nullptr && gn < N) ? bf2f(bias[gn]) : 0.f;
    }
    s_for<0, 64>([&](auto ic) {
      constexpr int I = decltype(ic)::value, xb = I >> 3, wb = I & 7;
      const int nl = wc * 128 + wb * 16 + r16, ml = wr * 128 + xb * 16 + 4 * g16;
      uint2 pk;
      pk.x = pack_bf2(s_acc_read<4 * I + 0>() + bvt[wb], s_acc_read<4 * I + 1>() + bvt[wb]);
      pk.y = pack_bf2(s_acc_read<4 * I + 2>() + bvt[wb], s_acc_read<4 * I + 3>() + bvt[wb]);
      *reinterpret_cast<uint2*>(smem + nl * S_EPI_LD + ml * 2) = pk;
    });
    __syncthreads();
    // ---- phase 2: a channel's 256 tokens = four 128-byte runs of V^T; 16-byte stores, tokens past M zeroed, blocks past ldy skipped
#pragma unroll 4
    for (int it = 0; it < 32; ++it) {
      const int id = it * 256 + tid;
      const int ch = id >> 5, cc = id & 31;  // channel within the tile, 8-token chunk
      const int gn = n0 + ch;
      const int64_t tok = m0 + cc * 8;
      if (gn < N && tok < ldy) {
        uint4 o = *reinterpret_cast<const uint4*>(smem + ch * S_EPI_LD + cc * 16);
        if (tok + 8 > M) {
          unsigned short h[8];
          *reinterpret_cast<uint4*>(h) = o;
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (tok + e >= M) h[e] = 0;
          o = *reinterpret_cast<const uint4*>(h);
        }
        const int64_t blocks = ldy >> 6;  // 64-token blocks per head
        unsigned short* dst = Y + (((int64_t)(gn >> 7) * blocks + (tok >> 6)) * 128 + (gn & 127)) * 64 + (tok & 63);
        *reinterpret_cast<uint4*>(dst) = o;
      }
    }
  } else {
  // ---- epilogue addressing: the output tile (and the residual tile) through buffer descriptors whose base is the tile's first element; a thread's
    //      chunk (row tid>>5 + 8 it, columns 8 (tid&31)..) is a per-thread 32-bit offset + a per-iteration SCALAR offset 8 it rows — no 64-bit
    //      address registers per iteration (they cost the residual variant 9 spilled VGPRs once its loads were hoisted).  Rows / columns past M / N
    //      are predicated off below; the descriptor range only has to cover the tile.
    const int erow = tid >> 5, ecc = tid & 31;
    const int egn = n0 + ecc * 8;
    const int64_t eycol = gb.y_cbw > 0 ? (int64_t)(egn / gb.y_cbw) * gb.y_cbs + egn % gb.y_cbw : egn;  // N-blocked y (GemmBlocking)
    const bool ecol_ok = egn < N;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(Y + m0 * ldy), 0, 0xffffffffu, 0x00020000);
    const unsigned y_voff = (unsigned)((erow * ldy + eycol) * 2), y_step = (unsigned)(8 * ldy * 2);
  // ---- epilogue phase 0 (residual epilogue): this thread's 32 residual chunks (the rows / columns it will store in phase 2) are requested NOW,
    //      into the registers the fragments no longer need, so that ONE memory latency runs under the accumulator -> LDS pass instead of eight
    //      dependent round trips inside the store loop (the residual variant ran 8 % below the plain one per FLOP: ~10 us of a 117 us tile).
    u32x4_t rres[32];
    if constexpr (EPI == X2V_EPI_RESIDUAL) {
      const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(resid + m0 * ldr), 0, 0xffffffffu, 0x00020000);
      const unsigned r_voff = (unsigned)((erow * ldr + egn) * 2), r_step = (unsigned)(8 * ldr * 2);
  #pragma unroll
      for (int it = 0; it < 32; ++it) {
        rres[it] = u32x4_t{0u, 0u, 0u, 0u};
        if (ecol_ok && m0 + erow + 8 * it < M) rres[it] = __builtin_amdgcn_raw_buffer_load_b128(rr, r_voff, (unsigned)it * r_step, 0);
      }
    }
  // ---- epilogue phase 1: acc (+bias, activation) -> bf16 -> LDS [256][S_EPI_LD]
    //      tile (xb, wb), register e: tile row wr*128 + xb*16 + r16, tile col wc*128 + wb*16 + 4*g16 + e
    uint2 bv[8];
  #pragma unroll
    for (int wb = 0; wb < 8; ++wb) {
      int gn = n0 + wc * 128 + wb * 16 + 4 * g16;
      gn = gn + 3 < N ? gn : (N >= 4 ? N - 4 : 0);
      bv[wb] = make_uint2(0u, 0u);
      if (bias != nullptr) bv[wb] = *reinterpret_cast<const uint2*>(bias + gn);
    }
    s_for<0, 64>([&](auto ic) {
      constexpr int I = decltype(ic)::value, xb = I >> 3, wb = I & 7;
      const int ml = wr * 128 + xb * 16 + r16, nl = wc * 128 + wb * 16 + 4 * g16;
      float vv[4] = {s_acc_read<4 * I + 0>(), s_acc_read<4 * I + 1>(), s_acc_read<4 * I + 2>(), s_acc_read<4 * I + 3>()};
      vv[0] += bf_lo(bv[wb].x);
      vv[1] += bf_hi(bv[wb].x);
      vv[2] += bf_lo(bv[wb].y);
      vv[3] += bf_hi(bv[wb].y);
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
      *reinterpret_cast<uint2*>(smem + ml * S_EPI_LD + nl * 2) = pk;
    });
    __syncthreads();
    // ---- epilogue phase 2: 16-byte stores, 32 lanes per 512-byte output row
    if constexpr (EPI == X2V_EPI_RESIDUAL) {
      float gv[8];  // per-column gate chunk: the same 8 columns in every iteration of this thread
      {
        uint4 g4 = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
        if (gate != nullptr && ecol_ok) g4 = *reinterpret_cast<const uint4*>(gate + egn);
        unpack8(g4, gv);
      }
  #pragma unroll
      for (int it = 0; it < 32; ++it) {
        if (ecol_ok && m0 + erow + 8 * it < M) {
          float yv[8], xv[8], ov[8];
          unpack8(*reinterpret_cast<const uint4*>(smem + (erow + 8 * it) * S_EPI_LD + ecc * 16), yv);
          unpack8(__builtin_bit_cast(uint4, rres[it]), xv);
          if (gate != nullptr) {
  #pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = xv[e] + rbf(yv[e] * gv[e]);
          } else {
  #pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = xv[e] + yv[e];
          }
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, pack8(ov)), ry, y_voff, (unsigned)it * y_step, 0);
        }
      }
    } else {
  #pragma unroll 4
      for (int it = 0; it < 32; ++it) {
        if (ecol_ok && m0 + erow + 8 * it < M)
          __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4_t*>(smem + (erow + 8 * it) * S_EPI_LD + ecc * 16), ry, y_voff, (unsigned)it * y_step, 0);
      }
    }
}
#undef S_SB
#endif
}

template <int EPI, bool VT = false>
static int launch_gemm256s(const void* x, int64_t ldx_bytes, const void* w, int64_t ldw_bytes, const void* bias, void* y, int64_t ldy, int64_t M, int N, int nk,
                           const void* resid, int64_t ldr, const void* gate, int gm_tiles, hipStream_t st, GemmBlocking gb) {
  if (gm_tiles <= 0) gm_tiles = 4;
  const int ntm = (int)((M + S_M - 1) / S_M), ntn = (N + S_N - 1) / S_N;
  int rc = ensure_dynamic_lds((const void*)gemm256s_kernel<EPI, VT>, S_LDS_BYTES, "gemm256s attr");
  if (rc != X2V_OK) return rc;
  hipLaunchKernelGGL((gemm256s_kernel<EPI, VT>), dim3((unsigned)ntm * (unsigned)ntn), dim3(256), S_LDS_BYTES, st, (const char*)x, ldx_bytes, (const char*)w, ldw_bytes,
                     (const unsigned short*)bias, (unsigned short*)y, ldy, M, N, nk, (const unsigned short*)resid, ldr, (const unsigned short*)gate, ntm, ntn, gm_tiles, gb);
  X2V_LAUNCH_CHECK("gemm256s launch");
  return X2V_OK;
}

// Called by gemm.hip's dispatcher (arguments already validated there; ld*_bytes < 16 MiB checked by the caller).
int gemm256s_dispatch(int epilogue, const void* x, int64_t ldxb, const void* w, int64_t ldwb, const void* bias, void* y, int64_t ldy, int64_t M, int N, int nk,
                      const void* resid, int64_t ldr, const void* gate, int gm_tiles, hipStream_t st, GemmBlocking gb) {
  switch (epilogue) {
    case X2V_EPI_NONE: return launch_gemm256s<X2V_EPI_NONE>(x, ldxb, w, ldwb, bias, y, ldy, M, N, nk, nullptr, 0, nullptr, gm_tiles, st, gb);
    case X2V_EPI_GELU_TANH: return launch_gemm256s<X2V_EPI_GELU_TANH>(x, ldxb, w, ldwb, bias, y, ldy, M, N, nk, nullptr, 0, nullptr, gm_tiles, st, gb);
    case X2V_EPI_SILU: return launch_gemm256s<X2V_EPI_SILU>(x, ldxb, w, ldwb, bias, y, ldy, M, N, nk, nullptr, 0, nullptr, gm_tiles, st, gb);
    case X2V_EPI_RESIDUAL: return launch_gemm256s<X2V_EPI_RESIDUAL>(x, ldxb, w, ldwb, bias, y, ldy, M, N, nk, resid, ldr, gate, gm_tiles, st, gb);
    default: set_error("gemm: unknown epilogue %d", epilogue); return X2V_E_ARG;
  }
}

// V^T-producing form (x2v_gemm_bf16_vt): y = V^T [N/128][ldvt/64][128][64]; `ldvt` travels in the ldy argument
int gemm256s_vt_dispatch(const void* x, int64_t ldxb, const void* w, int64_t ldwb, const void* bias, void* vt, int64_t ldvt, int64_t M, int N, int nk, hipStream_t st) {
  return launch_gemm256s<X2V_EPI_NONE, true>(x, ldxb, w, ldwb, bias, vt, ldvt, M, N, nk, nullptr, 0, nullptr, 0, st, GemmBlocking());
}

}  // namespace x2v
