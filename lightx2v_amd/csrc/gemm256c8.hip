// y[M,N] = epi((xq[M,K] . Wq[N,K]^T) * sx[m] * sw[n] + bias), e4m3 operands with per-token / per-channel fp32 scales (w8a8) — gemm256c.hip's CONTINUOUS
// single-stream pipeline for the fp8 operator.  Same contract, operand layouts, epilogues, rounding points, MFMA (v_mfma_scale_f32_32x32x64_f8f6f4 with
// unit block scales) and k order as the fp8 instantiation of gemm256.hip: bit-equal results (tools/gemm_fp8_continuous_check.py,
// tests/test_gpu_bench_shapes.py::test_gemm_fp8_continuous_pipeline_equals_ping_pong).
// Replaces the cutlass_scaled_mm / fp8_scaled_mm call of common/ops/mm/mm_weight.py:287-319 at the large shapes, behind x2v_gemm_fp8[_variant|_blocked].
//
// Bound: MFMA (fp8 dense peak ~5 PFLOP/s; at the board's 1400 W limit a bare 32x32x64 loop with this LDS fragment traffic sustains ~3.9 PFLOP/s,
// profiles/r02_mfma_power_probe.log PROBE_SET=2).  Algorithmic work 2*M*N*K FLOP per launch.
//
// Why: gemm256.hip's two-groups-of-four-waves ping-pong reaches 2.6-2.85 PFLOP/s in the w8a8 step, 68-73 % of what the
// chip sustains for this instruction mix; its prologue and epilogue (dequantise -> LDS -> barrier -> stores) are un-overlapped and weigh twice what
// they weigh in bf16 because an fp8 K tile (128 k) takes the time of a bf16 one (64 k).  The bf16 answer (gemm256s -> gemm256c) carries over because
// the BYTES are the same: a K tile is one 128-byte line per operand row either way, so the LDS images, the LDS-DMA pieces, the cursors that run on into
// the next output tile and the slot plan are gemm256c.hip's; what changes is
//   * the matrix instruction: a wave's 128 x 128 part = 4 x 4 accumulator tiles of 32 x 32 (16 AGPRs each); a K tile = 2 k-steps of 16 MFMAs of 16
//     passes (2048 matrix cycles, as in bf16), so the 128 "slots" of gemm256c's plan become 4 instruction positions behind each of the 32 MFMAs;
//   * fragments: 32 bytes per lane (row fl = lane & 31 of a 32-row block, 16-byte chunks 4 s + fh and 4 s + 2 + fh of k-step s: the instruction's own
//     k order, as gemm256.hip) = two ds_read_b128 into the halves of an 8-register operand; 16 reads per k-step, the same count as bf16;
//   * the epilogue dequantises in the accumulator layout (value * sx[row] * sw[col] + bias, then the activation, then bf16 — gemm256.hip's order) and
//     transposes through the wave's private 4 KiB strip in half-blocks of 32 rows x 64 columns: 8-byte pieces in, 16-byte row-major pieces out (8 rows
//     x 128 contiguous bytes = whole lines per store instruction).  The walk is column-half-major so that only one half's per-column operands (8 x float4
//     scales + 8 x 4 bf16 bias) have to be requested while the fragment registers are still in use: the first half's ride in the LAST K tile's slots, the second half's
//     are requested when the epilogue starts and arrive under the first half's work.
// Shapes: as gemm256c (even number of K tiles >= 4, N a multiple of 256, y blocks multiples of 128 columns, residual with y's row stride); the rest
// stays on gemm256.hip.  First contact on MI355X (profiles/r04_call16_*): 84 / 84 equality cases bit-equal, +2..5 % plain, +1.5..2.5 % GELU, +4..10 %
// residual at the w8a8 step's shapes (2.55-2.87 PFLOP/s).  The dispatcher's default for row-major operands; block-strided operands (Ulysses buffers) keep
// the ping-pong kernel until that path has met a GPU (X2V_GEMM_FP8_CONTINUOUS=2; =0: never).  Variant 5 of x2v_gemm_fp8_variant forces this kernel.
// AUDIT after every edit (the accumulator half is invisible to the compiler): `hipcc -S` must show .vgpr_spill_count 0,
// .private_segment_fixed_size 0 and no v_accvgpr_* / a[..] operand outside ;;#ASMSTART / ;;#ASMEND.
#include <type_traits>

#include "x2v_common.h"

namespace x2v {
namespace {

constexpr int C8_M = 256, C8_N = 256;
constexpr int C8_OP_BYTES = 256 * 128;           // one operand tile of one stage
constexpr int C8_STAGE_BYTES = 2 * C8_OP_BYTES;  // W tile | x tile
constexpr int C8_LDS_BYTES = 2 * C8_STAGE_BYTES;  // 131072: the two stages
constexpr int C8_STRIP_BYTES = 32 * 128;          // per wave: the epilogue's transposition strip, one half-block (32 rows x 64 bf16) at a time
constexpr int C8_LDS_TOTAL = C8_LDS_BYTES + 4 * C8_STRIP_BYTES;
// Instruction positions of a K tile: n = 0..127, MFMA n / 4 sits at n % 4 == 0 (gemm256c.hip's slot plan, one bf16 MFMA slot = a quarter of an fp8 one):
//   C8_LATE0 + C8_STEP i    the last 16 - C8_EARLY LDS-DMA pieces of tile t+1          0, 2, .., 30   fragment reads of k-step 1
//   C8_FREE                 lgkmcnt(0) + barrier "this tile's stage is free"
//   C8_FREE + 1 + C8_STEP i the first C8_EARLY pieces of tile t+2
//   C8_READY                vmcnt + barrier "tile t+1 has landed"                     C8_READY + 2, + 4, ..   fragment reads of k-step 0 of t+1
//   (LAST K tile of an output tile only) one epilogue-operand buffer load in each position of [C8_X0, C8_READY) that holds no LDS-DMA piece
constexpr int C8_STEP = 7, C8_FREE = 36, C8_READY = 94, C8_LATE0 = 3;
constexpr int C8_EARLY = (127 - C8_FREE - 1) / C8_STEP + 1 < 16 ? (127 - C8_FREE - 1) / C8_STEP + 1 : 16;
constexpr int C8_NEWER = (C8_READY - C8_FREE - 1) / C8_STEP + 1 < C8_EARLY ? (C8_READY - C8_FREE - 1) / C8_STEP + 1 : C8_EARLY;
static_assert(C8_LATE0 + (16 - C8_EARLY - 1) * C8_STEP < C8_FREE && C8_READY + 2 + 30 <= 127, "slot plan");
// fragments of k-step 1 are last read at position 30 and the stage is declared free at C8_FREE: every ds_read of the tile is issued before the barrier
static_assert(30 < C8_FREE, "slot plan: fragment reads before the stage is freed");
constexpr int C8_X0 = C8_FREE + 2;
constexpr bool c8_dma_slot(int n) { return n > C8_FREE && (n - C8_FREE - 1) % C8_STEP == 0 && (n - C8_FREE - 1) / C8_STEP < C8_EARLY; }
constexpr int c8_xload_index(int n) {  // which epilogue-operand load sits at position n (-1: none)
  if (n < C8_X0 || n >= C8_READY || c8_dma_slot(n)) return -1;
  int idx = 0;
  for (int s = C8_X0; s < n; ++s)
    if (!c8_dma_slot(s)) ++idx;
  return idx;
}
constexpr int c8_xload_slots() {
  int c = 0;
  for (int s = C8_X0; s < C8_READY; ++s)
    if (!c8_dma_slot(s)) ++c;
  return c;
}
// sx (4) + sw of a column half (8) + bias of a column half (8) [+ gate of a column half (1) + the residual chunks of the first half-block (4)]
constexpr int C8_NX_PLAIN = 4 + 8 + 8, C8_NX_RES = C8_NX_PLAIN + 1 + 4;
static_assert(c8_xload_slots() >= C8_NX_RES, "the LAST K tile has a position for every epilogue-operand load");
static_assert(C8_NEWER + C8_NX_RES <= 63, "vmcnt immediate");

typedef __attribute__((address_space(3))) void* c8_lds_ptr_t;
typedef unsigned int c8_u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int c8_u32x4_t __attribute__((ext_vector_type(4)));
typedef float c8_f32x4_t __attribute__((ext_vector_type(4)));

// Every asm statement that touches the accumulator half names ALL of it as clobbered (gemm256s.hip: the round-3 incident).
#define C8_AGPRS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"

template <int B, int E, class F>
__device__ __forceinline__ void c8_for(F&& f) {  // f(integral_constant<int, i>) for i = B .. E-1, fully unrolled with constant indices
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    c8_for<B + 1, E>(f);
  }
}

// accumulator tile I (= x block * 4 + W block, 32 x 32) is a[16 I : 16 I + 15].  The UNSCALED encoding v_mfma_f32_32x32x64_f8f6f4 (8 bytes; both
// operands e4m3 by its default cbsz / blgp): per-channel w8a8 has no block scales, and the scaled encoding with unit e8m0 scales (16 bytes + a
// scale register read per MFMA; what gemm256.hip's builtin emits) gives the same bits — 84 / 84 equality cases — and is 0.5-1.0 % slower at every
// shape and epilogue measured (profiles/r05_call1_*: 13824->5120 plain 2905 -> 2933 TFLOP/s).
template <int I>
__device__ __forceinline__ void c8_mfma(const i32x8_t& wf, const i32x8_t& xf) {
  asm volatile("v_mfma_f32_32x32x64_f8f6f4 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(wf), "v"(xf), "i"(16 * I), "i"(16 * I + 15) : C8_AGPRS);
}
template <int I>
__device__ __forceinline__ void c8_mfma_first(const i32x8_t& wf, const i32x8_t& xf) {  // first k-step of an output tile: C = 0
  asm volatile("v_mfma_f32_32x32x64_f8f6f4 a[%c2:%c3], %0, %1, 0" ::"v"(wf), "v"(xf), "i"(16 * I), "i"(16 * I + 15) : C8_AGPRS);
}
template <int R>
__device__ __forceinline__ float c8_acc_read() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(x) : "i"(R) : C8_AGPRS);
  return x;
}

// LDS-DMA cursor: which K tile of which output tile a piece belongs to.  Descriptors are wave-uniform (SGPRs).
struct C8Cursor {
  __amdgpu_buffer_rsrc_t ra, rw;
  unsigned kw;  // byte offset of the K tile within a W row
  unsigned ka;  // byte offset of the K tile within an x row (K-blocked x: GemmBlocking)
  int kc;       // index of the K tile within its K block of x
  int k;        // index of the K tile within the output tile
};

template <int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm256c8_kernel(
    const char* __restrict__ A, int64_t lda_bytes, const char* __restrict__ W, int64_t ldw_bytes, const unsigned short* __restrict__ bias, unsigned short* Y,
    int64_t ldy, int64_t M, int N, int nk, const unsigned short* resid, int64_t ldr, const unsigned short* __restrict__ gate, const float* __restrict__ sx,
    const float* __restrict__ sw, int ntm, int ntn, int gm_tiles, GemmBlocking gb) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  asm volatile("" ::: C8_AGPRS);  // the accumulator half belongs to the asm statements of this kernel
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const int fl = lane & 31, fh = lane >> 5;

  // ---- this workgroup's output tiles (gemm256c.hip): positions v, v + vstep, .. < vend of the grouped tile order; XCD x (= blockIdx % 8) owns the
  //      contiguous chunk x of that order, its workgroups take the chunk's positions round-robin
  const unsigned nblk = (unsigned)ntm * (unsigned)ntn;
  unsigned v, vstep, vend;
  if (gridDim.x == nblk) {
    v = xcd_remap(blockIdx.x, nblk);
    vstep = 1u;
    vend = v + 1u;
  } else {
    const unsigned x = blockIdx.x & 7u, j = blockIdx.x >> 3, per = gridDim.x >> 3;
    const unsigned q = nblk >> 3, r = nblk & 7u;
    const unsigned base = x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q;
    v = base + j;
    vstep = per;
    vend = base + q + (x < r ? 1u : 0u);
  }
  const unsigned GM = (unsigned)gm_tiles;
  const unsigned per_group = GM * (unsigned)ntn;
  const unsigned row_bytes = (unsigned)nk * 128u;
  const int a_kpb = gb.a_kpb > 0 && gb.a_kpb < nk ? gb.a_kpb : nk;  // K tiles per K block of x (GemmBlocking)
  const unsigned a_span = a_kpb < nk ? (unsigned)((nk - 1) / a_kpb) * gb.a_cbs + (unsigned)a_kpb * 128u : row_bytes;
  const unsigned a_wrap = gb.a_cbs - (unsigned)(a_kpb - 1) * 128u;  // from the last K tile of an x block to the first of the next

  auto coords = [&](unsigned p, int& tm, int& tn) {  // grouped ordering: gm_tiles m-tiles x all n-tiles per group, as gemm256.hip
    const unsigned group = p / per_group, in_g = p % per_group;
    const unsigned first_m = group * GM;
    const unsigned gsz = min((unsigned)ntm - first_m, GM);
    tm = (int)(first_m + in_g % gsz);
    tn = (int)(in_g / gsz);
  };
  // buffer descriptors over a tile's valid rows: rows past M / N read as zero through the bounds check; `live` false: an empty range
  auto operands = [&](int tm, int tn, bool live, __amdgpu_buffer_rsrc_t& ra, __amdgpu_buffer_rsrc_t& rw) {
    const int64_t m0 = (int64_t)tm * C8_M;
    const int n0 = tn * C8_N;
    const int rows_a = (int)min((int64_t)C8_M, M - m0), rows_w = min(C8_N, N - n0);
    ra = __builtin_amdgcn_make_buffer_rsrc((void*)(A + m0 * lda_bytes), 0, live ? (unsigned)((rows_a - 1) * lda_bytes) + a_span : 0u, 0x00020000);
    rw = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (int64_t)n0 * ldw_bytes), 0, live ? (unsigned)((rows_w - 1) * ldw_bytes) + row_bytes : 0u, 0x00020000);
  };
  auto advance = [&](C8Cursor& c, const __amdgpu_buffer_rsrc_t& nra, const __amdgpu_buffer_rsrc_t& nrw) {  // to the next K tile of the pipeline
    if (++c.k == nk) {
      c.k = 0;
      c.kw = 0u;
      c.ka = 0u;
      c.kc = 0;
      c.ra = nra;
      c.rw = nrw;
    } else {
      c.kw += 128u;
      if (++c.kc == a_kpb) {
        c.kc = 0;
        c.ka += a_wrap;
      } else {
        c.ka += 128u;
      }
    }
  };

  // ---- LDS-DMA (gemm256c.hip): wave `wid` stages rows [64 wid, 64 wid + 64) of both operand tiles as 8 pieces of 8 rows (1 KiB, lane-linear in LDS);
  //      the image of a row is its eight 16-byte chunks at chunk ^ ((row >> 1) & 7)
  unsigned a_voff[2], w_voff[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int r = wid * 64 + par * 8 + (lane >> 3);
    const int c = (lane & 7) ^ (((lane >> 4) + 4 * par) & 7);
    a_voff[par] = (unsigned)(r * lda_bytes) + (unsigned)(c << 4);
    w_voff[par] = (unsigned)(r * ldw_bytes) + (unsigned)(c << 4);
  }
  const unsigned a_j = (unsigned)(16 * lda_bytes), w_j = (unsigned)(16 * ldw_bytes);
  // piece P_ in 0..15 of the K tile cursor CUR_ points at: 0..7 = W pieces, 8..15 = x pieces
#define C8_DMA(P_, STAGE_, CUR_)                                                                                                                \
  {                                                                                                                                            \
    constexpr int i_ = (P_) & 7;                                                                                                               \
    if constexpr ((P_) < 8)                                                                                                                    \
      __builtin_amdgcn_raw_ptr_buffer_load_lds((CUR_).rw, (c8_lds_ptr_t)(smem + (STAGE_) * C8_STAGE_BYTES + wid * 8192 + i_ * 1024), 16, w_voff[i_ & 1],        \
                                               (CUR_).kw + (unsigned)(i_ >> 1) * w_j, 0, 0);                                                   \
    else                                                                                                                                       \
      __builtin_amdgcn_raw_ptr_buffer_load_lds((CUR_).ra, (c8_lds_ptr_t)(smem + (STAGE_) * C8_STAGE_BYTES + C8_OP_BYTES + wid * 8192 + i_ * 1024), 16,          \
                                               a_voff[i_ & 1], (CUR_).ka + (unsigned)(i_ >> 1) * a_j, 0, 0);                                   \
  }

  // ---- fragment addresses (32x32x64: row fl of a 32-row block; half e of k-step s = 16-byte chunk 4 s + 2 e + fh), block offsets travel as immediates
  int rd_x[2][2], rd_w[2][2];
  {
    const int swz = (fl >> 1) & 7;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int o = fl * 128 + ((((s << 2) | (e << 1) | fh) ^ swz) << 4);
        rd_x[s][e] = o + C8_OP_BYTES + wr * 16384;
        rd_w[s][e] = o + wc * 16384;
      }
  }
  // fragment halves: fx[ks][xb][e], fw[ks][wb][e]; an MFMA operand is the 8-register pair (e = 0, 1)
  i32x4_t fx[2][4][2], fw[2][4][2];
  // read R_ in 0..15 of k-step KS_ of the tile in stage STAGE_: fragment R_ / 2 (order x0, W0..W3, x1..x3: the first MFMA of a k-step needs x0, W0), half R_ % 2
#define C8_READ(R_, STAGE_, KS_)                                                                                                                \
  {                                                                                                                                            \
    constexpr int f_ = (R_) >> 1, e_ = (R_) & 1;                                                                                               \
    if constexpr (f_ == 0) fx[KS_][0][e_] = *reinterpret_cast<const i32x4_t*>(smem + (STAGE_) * C8_STAGE_BYTES + rd_x[KS_][e_]);                \
    else if constexpr (f_ <= 4) fw[KS_][f_ - 1][e_] = *reinterpret_cast<const i32x4_t*>(smem + (STAGE_) * C8_STAGE_BYTES + (f_ - 1) * 4096 + rd_w[KS_][e_]); \
    else fx[KS_][f_ - 4][e_] = *reinterpret_cast<const i32x4_t*>(smem + (STAGE_) * C8_STAGE_BYTES + (f_ - 4) * 4096 + rd_x[KS_][e_]);           \
  }
#define C8_SB() __builtin_amdgcn_sched_barrier(0)

  // ---- epilogue of the CURRENT output tile (see the header).  Half-block (ch, xb) = rows [32 xb, 32 xb + 32) x columns [64 ch, 64 ch + 64) of the
  //      wave's 128 x 128 part, walked ch-major.
  //      Phase A (accumulator layout: lane = row fl, columns 8 g + 4 fh + e of W block wb): dequantise, bias, activation, bf16; the 4 values of a
  //      (wb, g) = 8 bytes = unit u = ((wb & 1) * 4 + g) * 2 + fh of the strip row, stored at unit u ^ (fl >> 1): 32 lanes of one fh hit 64 banks once.
  //      Phase B (row-major: lane = row 8 i + (lane >> 3), 16-byte chunk c8 = lane & 7 of the 128-byte half-row): the two units of chunk c8 sit in
  //      chunk c8 ^ (row >> 2), swapped when (row >> 1) & 1; one ds_read_b128, a conditional swap, (residual), one 16-byte store — the store
  //      instruction covers 8 rows x 128 contiguous bytes.  LDS executes a wave's instructions in order, so phase A of the next half-block may
  //      overwrite the strip right behind phase B's reads.
  //      Addressing as gemm256c.hip: vector offset = lane part or the "row does not exist" mark 0x80000000, scalar offset = column base + rows.
  c8_u32x2_t e_bias[2][8];  // [column half][j]
  c8_f32x4_t e_sw[2][8];
  float e_sx[4];
  c8_u32x4_t e_gate4[2], e_res[2][4];
  __amdgpu_buffer_rsrc_t r_y, r_res, r_bias, r_gate, r_sw, r_sx;
  char* const strip = smem + C8_LDS_BYTES + wid * C8_STRIP_BYTES;
  const int l8 = lane >> 3, c8 = lane & 7;
  const unsigned lane_off = (unsigned)((wr * 128 + l8) * ldy * 2) + (unsigned)(16 * c8);  // phase B: bytes from the tile's first row / the half-block's first column
  const int wa0 = fl * 128 + ((fh ^ ((fl >> 1) & 15)) << 3);                                // phase A: unit fh ^ (fl >> 1) of row fl
  const int rb0 = l8 * 128 + ((c8 ^ (l8 >> 2)) << 4);                                       // phase B: chunk c8 ^ (row >> 2) of row l8 (+ 8 i rows: ^ (i << 5), + 1024 i)
  const bool swap_b = ((l8 >> 1) & 1) != 0;
  unsigned s_col = 0u;   // the wave's first column in the output / residual row, bytes (wave-uniform)
  unsigned s_bias = 0u;  // the wave's first column in bias / gate (bf16), bytes; scales (fp32): twice that
  int rows_left = 0;     // valid rows of the current tile below row wr*128 + (lane>>3): phase-B row 32 xb + 8 i of the lane exists iff it is < rows_left
  const unsigned y_row = (unsigned)(ldy * 2);  // one row of y (and of the residual: ldr == ldy, y row-major — dispatcher), bytes
  constexpr bool RES = EPI == X2V_EPI_RESIDUAL;
  constexpr int NXLOAD = RES ? C8_NX_RES : C8_NX_PLAIN;

  auto epilogue_setup = [&](int tm, int tn) {
    const int64_t m0 = (int64_t)tm * C8_M;
    const int gn0 = tn * C8_N + wc * 128;  // first column of this wave (wave-uniform)
    r_y = __builtin_amdgcn_make_buffer_rsrc((void*)(Y + m0 * ldy), 0, 0x80000000u, 0x00020000);
    r_bias = __builtin_amdgcn_make_buffer_rsrc((void*)bias, 0, bias != nullptr ? (unsigned)N * 2u : 0u, 0x00020000);
    r_sw = __builtin_amdgcn_make_buffer_rsrc((void*)sw, 0, (unsigned)N * 4u, 0x00020000);
    // rows past M read scale 0: their results are never stored
    r_sx = __builtin_amdgcn_make_buffer_rsrc((void*)(sx + m0 + wr * 128), 0, (unsigned)(max((int64_t)0, min((int64_t)128, M - m0 - wr * 128)) * 4), 0x00020000);
    unsigned col = (unsigned)gn0;
    if (gb.y_cbw > 0) {  // N-blocked y: column n at (n / y_cbw) * y_cbs + n % y_cbw elements from the row's start
      const unsigned qb = (unsigned)gn0 / (unsigned)gb.y_cbw;
      col = qb * (unsigned)gb.y_cbs + ((unsigned)gn0 - qb * (unsigned)gb.y_cbw);
    }
    s_col = col * 2u;
    s_bias = (unsigned)gn0 * 2u;
    rows_left = (int)min((int64_t)C8_M, M - m0) - wr * 128 - l8;
    if constexpr (RES) {
      r_res = __builtin_amdgcn_make_buffer_rsrc((void*)(resid + m0 * ldy), 0, 0x80000000u, 0x00020000);
      r_gate = __builtin_amdgcn_make_buffer_rsrc((void*)gate, 0, gate != nullptr ? (unsigned)N * 2u : 0u, 0x00020000);
    }
  };
  auto row_voff = [&](int row8) { return row8 < rows_left ? lane_off : 0x80000000u; };  // phase-B vector offset of local row `row8` (= 32 xb + 8 i)
  // half-block hb = 4 ch + xb in walk order; residual chunk i of it: the 16 bytes phase B's store i of that half-block will overwrite
  auto res_load = [&](auto hbc, auto ic) {
    constexpr int hb = decltype(hbc)::value, ch = hb >> 2, xb = hb & 3, i = decltype(ic)::value;
    e_res[hb & 1][i] = __builtin_bit_cast(c8_u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r_res, row_voff(32 * xb + 8 * i), s_col + (unsigned)(ch * 128) + (unsigned)(32 * xb + 8 * i) * y_row, 0));
  };
  // per-column operands of column half ch: scales and bias in phase-A layout (tile wb = 2 ch + (j >> 2), g = j & 3: columns 32 wb + 8 g + 4 fh ..+3)
  auto sw_load = [&](auto chc, auto jc) {
    constexpr int ch = decltype(chc)::value, j = decltype(jc)::value;
    e_sw[ch][j] = __builtin_bit_cast(c8_f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r_sw, (unsigned)(16 * fh), 2u * s_bias + (unsigned)((ch * 64 + j * 8) * 4), 0));
  };
  auto bias_load = [&](auto chc, auto jc) {
    constexpr int ch = decltype(chc)::value, j = decltype(jc)::value;
    e_bias[ch][j] = __builtin_bit_cast(c8_u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(r_bias, (unsigned)(8 * fh), s_bias + (unsigned)((ch * 64 + j * 8) * 2), 0));
  };
  auto gate_load = [&](auto chc) {
    constexpr int ch = decltype(chc)::value;
    e_gate4[ch] = __builtin_bit_cast(c8_u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r_gate, (unsigned)(16 * c8), s_bias + (unsigned)(ch * 128), 0));
  };
  using k0 = std::integral_constant<int, 0>;
  using k1 = std::integral_constant<int, 1>;
  // epilogue-operand load J of the current output tile (LAST K tile): sx, then column half 0's scales and bias, [gate half 0, residual of half-block 0]
  auto xload = [&](auto jc) {
    constexpr int J = decltype(jc)::value;
    if constexpr (J < 4) {
      e_sx[J] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_sx, (unsigned)(4 * fl), (unsigned)(J * 128), 0));
    } else if constexpr (J < 12) {
      sw_load(k0{}, std::integral_constant<int, J - 4>{});
    } else if constexpr (J < 20) {
      bias_load(k0{}, std::integral_constant<int, J - 12>{});
    } else if constexpr (RES && J == 20) {
      gate_load(k0{});
    } else if constexpr (RES && J < C8_NX_RES) {
      res_load(k0{}, std::integral_constant<int, J - 21>{});
    }
  };

  // One K tile of the pipeline (gemm256c.hip's `tile`).  ST = its LDS stage; FIRST: K tile 0 of an output tile; LAST: its last K tile.
  auto tile = [&](auto stc, auto firstc, auto lastc, const C8Cursor& c1, const C8Cursor& c2) {
    constexpr int ST = decltype(stc)::value;
    constexpr bool FIRST = decltype(firstc)::value != 0, LAST = decltype(lastc)::value != 0;
    c8_for<0, 128>([&](auto nc) {
      constexpr int n = decltype(nc)::value;
      if constexpr ((n & 3) == 0) {
        constexpr int m = n >> 2, ks = m >> 4, xb = (m >> 2) & 3, wb = m & 3;
        const i32x8_t wf = __builtin_shufflevector(fw[ks][wb][0], fw[ks][wb][1], 0, 1, 2, 3, 4, 5, 6, 7);
        const i32x8_t xf = __builtin_shufflevector(fx[ks][xb][0], fx[ks][xb][1], 0, 1, 2, 3, 4, 5, 6, 7);
        if constexpr (FIRST && ks == 0) c8_mfma_first<xb * 4 + wb>(wf, xf);
        else c8_mfma<xb * 4 + wb>(wf, xf);
      }
      if constexpr (n < 32 && (n & 1) == 0) C8_READ(n >> 1, ST, 1)  // k-step 1 of this tile
      // the last 16 - C8_EARLY pieces of tile t+1 (its stage was freed by the previous tile's first barrier)
      if constexpr (n >= C8_LATE0 && (n - C8_LATE0) % C8_STEP == 0 && (n - C8_LATE0) / C8_STEP < 16 - C8_EARLY) C8_DMA(C8_EARLY + (n - C8_LATE0) / C8_STEP, ST ^ 1, c1)
      if constexpr (n == C8_FREE) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every fragment of this tile is in registers: the stage may be overwritten
        __builtin_amdgcn_s_barrier();
      }
      // the first C8_EARLY pieces of tile t+2 into this tile's stage
      if constexpr (c8_dma_slot(n)) C8_DMA((n - C8_FREE - 1) / C8_STEP, ST, c2)
      if constexpr (LAST && c8_xload_index(n) >= 0 && c8_xload_index(n) < NXLOAD) xload(std::integral_constant<int, c8_xload_index(n)>{});
      if constexpr (n == C8_READY) {
        // tile t+1 has landed; younger loads may stay in flight: the pieces of tile t+2 issued so far in this tile (+ the epilogue operands)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C8_NEWER + (LAST ? NXLOAD : 0)) : "memory");
        __builtin_amdgcn_s_barrier();
      }
      if constexpr (!LAST && n > C8_READY + 1 && (n & 1) == 0) C8_READ((n - C8_READY - 2) >> 1, ST ^ 1, 0)  // k-step 0 of the next tile
      C8_SB();
    });
  };

  // Phase A of accumulator tile I = 4 xb + wb: value 4 g + e of the lane = row 32 xb + fl, column 32 wb + 8 g + 4 fh + e of the wave's part
  auto epi_phase_a = [&](auto ic) {
    constexpr int I = decltype(ic)::value, xb = I >> 2, wb = I & 3, ch = wb >> 1;
    c8_for<0, 4>([&](auto gc) {
      constexpr int g = decltype(gc)::value, j = (wb & 1) * 4 + g;
      float vv[4] = {c8_acc_read<16 * I + 4 * g + 0>(), c8_acc_read<16 * I + 4 * g + 1>(), c8_acc_read<16 * I + 4 * g + 2>(), c8_acc_read<16 * I + 4 * g + 3>()};
      // gemm256.hip's statements, in its order
      vv[0] = vv[0] * e_sx[xb] * e_sw[ch][j].x;
      vv[1] = vv[1] * e_sx[xb] * e_sw[ch][j].y;
      vv[2] = vv[2] * e_sx[xb] * e_sw[ch][j].z;
      vv[3] = vv[3] * e_sx[xb] * e_sw[ch][j].w;
      vv[0] += bf_lo(e_bias[ch][j].x);
      vv[1] += bf_hi(e_bias[ch][j].x);
      vv[2] += bf_lo(e_bias[ch][j].y);
      vv[3] += bf_hi(e_bias[ch][j].y);
      if (EPI == X2V_EPI_GELU_TANH) {
#pragma unroll
        for (int e = 0; e < 4; ++e) vv[e] = gelu_tanh_f(rbf(vv[e]));
      } else if (EPI == X2V_EPI_SILU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) vv[e] = silu_f(rbf(vv[e]));
      }
      *reinterpret_cast<c8_u32x2_t*>(strip + (wa0 ^ (j << 4))) = c8_u32x2_t{pack_bf2(vv[0], vv[1]), pack_bf2(vv[2], vv[3])};
    });
  };
  // Phase B, instruction i of half-block hb: local rows 32 xb + 8 i + (lane >> 3), the lane's 8 columns 64 ch + 8 (lane & 7)..: strip -> (residual) -> memory.
  // The four strip reads of a half-block are issued back to back (one LDS latency per half-block, not four), then combined and stored.
  c8_u32x4_t e_raw[4];
  auto epi_read_b = [&](auto ic) {
    constexpr int i = decltype(ic)::value;
    e_raw[i] = *reinterpret_cast<const c8_u32x4_t*>(strip + ((rb0 ^ (i << 5)) + i * 1024));
  };
  auto epi_phase_b = [&](auto hbc, auto ic) {
    constexpr int hb = decltype(hbc)::value, ch = hb >> 2, xb = hb & 3, i = decltype(ic)::value;
    const c8_u32x4_t raw = e_raw[i];
    c8_u32x4_t yv4 = swap_b ? c8_u32x4_t{raw.z, raw.w, raw.x, raw.y} : raw;
    if constexpr (RES) {
      float yv[8], xv[8], gv[8], ov[8];
      unpack8(__builtin_bit_cast(uint4, yv4), yv);
      unpack8(__builtin_bit_cast(uint4, e_res[hb & 1][i]), xv);
      unpack8(__builtin_bit_cast(uint4, e_gate4[ch]), gv);
#pragma unroll
      for (int e = 0; e < 8; ++e) ov[e] = xv[e] + rbf(yv[e] * gv[e]);
      yv4 = __builtin_bit_cast(c8_u32x4_t, pack8(ov));
    }
    __builtin_amdgcn_raw_buffer_store_b128(yv4, r_y, row_voff(32 * xb + 8 * i), s_col + (unsigned)(ch * 128) + (unsigned)(32 * xb + 8 * i) * y_row, 0);
  };
  // Walk order (software-pipelined by one half-block so that no LDS round trip is waited for): A(0) R(0) | A(1) B(0) R(1) | .. | A(7) B(6) R(7) | B(7),
  // A = phase A (accumulators -> strip), R = the four strip reads, B = combine + store.  R(hb - 1) is in flight while A(hb) overwrites the strip: LDS
  // executes a wave's instructions in order, so the reads return the old contents.
  auto epilogue = [&]() {
    // column half 1's operands fly under half 0's four half-blocks (the fragment registers are free here)
    c8_for<0, 8>([&](auto jc) { sw_load(k1{}, jc); });
    c8_for<0, 8>([&](auto jc) { bias_load(k1{}, jc); });
    if constexpr (RES) {
      if (gate != nullptr) {
        gate_load(k1{});
      } else {  // x + y: gate 1.0 gives the same bits (y is already bf16)
        e_gate4[0] = c8_u32x4_t{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
        e_gate4[1] = e_gate4[0];
      }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");  // the last MFMAs' results before the accumulator reads below
    C8_SB();
    c8_for<0, 9>([&](auto hbc) {
      constexpr int hb = decltype(hbc)::value;
      if constexpr (hb < 8) {
        constexpr int ch = hb >> 2, xb = hb & 3;
        c8_for<0, 2>([&](auto wc2) { epi_phase_a(std::integral_constant<int, xb * 4 + ch * 2 + decltype(wc2)::value>{}); });
        C8_SB();  // one half-block at a time: left alone, the scheduler hoists the unpacking of every operand and spills
      }
      if constexpr (hb > 0) {
        c8_for<0, 4>([&](auto ic) { epi_phase_b(std::integral_constant<int, hb - 1>{}, ic); });
        C8_SB();
      }
      if constexpr (hb < 8) {
        if constexpr (RES && hb < 7) {  // the next half-block's residual chunks (their buffer was last read by B(hb - 1) above)
          c8_for<0, 4>([&](auto ic) { res_load(std::integral_constant<int, hb + 1>{}, ic); });
        }
        c8_for<0, 4>([&](auto ic) { epi_read_b(ic); });
        C8_SB();
      }
    });
  };

  // ---- pipeline start: K tile 0 of the first output tile and the first C8_EARLY pieces of its K tile 1 in flight, tile 0 landed
  int tm, tn;
  coords(v, tm, tn);
  C8Cursor cu1, cu2;
  {
    __amdgpu_buffer_rsrc_t ra, rw;
    operands(tm, tn, true, ra, rw);
    cu1 = C8Cursor{ra, rw, 0u, 0u, 0, 0};
    c8_for<0, 16>([&](auto pc) { C8_DMA(decltype(pc)::value, 0, cu1) });
    advance(cu1, ra, rw);  // K tile 1 (nk >= 4: no wrap here)
    c8_for<0, C8_EARLY>([&](auto pc) { C8_DMA(decltype(pc)::value, 1, cu1) });
    cu2 = cu1;
    advance(cu2, ra, rw);  // K tile 2
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C8_EARLY) : "memory");
  }
  __builtin_amdgcn_s_barrier();
  C8_SB();

  for (;;) {
    // current output tile (tm, tn); the one behind it, whose K tiles the cursors run into near the end of this one
    const bool has_next = v + vstep < vend;
    int ntm_ = tm, ntn_ = tn;
    if (has_next) coords(v + vstep, ntm_, ntn_);
    __amdgpu_buffer_rsrc_t nra, nrw;
    operands(ntm_, ntn_, has_next, nra, nrw);
    epilogue_setup(tm, tn);
    c8_for<0, 16>([&](auto rc) { C8_READ(decltype(rc)::value, 0, 0) });  // k-step 0 of K tile 0 (stage 0: nk is even)
    C8_SB();
#define C8_STEP_CURSORS()     \
  {                           \
    cu1 = cu2;                \
    advance(cu2, nra, nrw);   \
  }
    tile(k0{}, k1{}, k0{}, cu1, cu2);  // K tile 0
    C8_STEP_CURSORS()
    tile(k1{}, k0{}, k0{}, cu1, cu2);  // K tile 1
    C8_STEP_CURSORS()
    for (int t = 2; t < nk - 2; t += 2) {
      tile(k0{}, k0{}, k0{}, cu1, cu2);
      C8_STEP_CURSORS()
      tile(k1{}, k0{}, k0{}, cu1, cu2);
      C8_STEP_CURSORS()
    }
    tile(k0{}, k0{}, k0{}, cu1, cu2);  // K tile nk - 2: cu2 already points at K tile 0 of the next output tile
    C8_STEP_CURSORS()
    tile(k1{}, k0{}, k1{}, cu1, cu2);  // K tile nk - 1 (LAST)
    C8_STEP_CURSORS()
#undef C8_STEP_CURSORS
    epilogue();
    if (!has_next) break;
    v += vstep;
    tm = ntm_;
    tn = ntn_;
  }
  // the pieces issued for the (non-existent) K tiles behind the last output tile read an empty range; let them retire before the LDS goes away
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef C8_READ
#undef C8_DMA
#undef C8_SB
#endif
}

template <int EPI>
int launch_gemm256c8(const void* x, int64_t ldx_bytes, const void* w, int64_t ldw_bytes, const void* bias, void* y, int64_t ldy, int64_t M, int N, int nk,
                     const void* resid, int64_t ldr, const void* gate, const float* sx, const float* sw, int gm_tiles, hipStream_t st, GemmBlocking gb) {
  if (gm_tiles <= 0) gm_tiles = 4;
  const int ntm = (int)((M + C8_M - 1) / C8_M), ntn = (N + C8_N - 1) / C8_N;
  int rc = ensure_dynamic_lds((const void*)gemm256c8_kernel<EPI>, C8_LDS_TOTAL, "gemm256c8 attr");
  if (rc != X2V_OK) return rc;
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
    return n & ~7;  // whole XCD octets
  }();
  const unsigned nblk = (unsigned)ntm * (unsigned)ntn;
  const unsigned grid = nblk > (unsigned)cus ? (unsigned)cus : nblk;
  hipLaunchKernelGGL((gemm256c8_kernel<EPI>), dim3(grid), dim3(256), C8_LDS_TOTAL, st, (const char*)x, ldx_bytes, (const char*)w, ldw_bytes, (const unsigned short*)bias,
                     (unsigned short*)y, ldy, M, N, nk, (const unsigned short*)resid, ldr, (const unsigned short*)gate, sx, sw, ntm, ntn, gm_tiles, gb);
  X2V_LAUNCH_CHECK("gemm256c8 launch");
  return X2V_OK;
}

}  // namespace

// Called by gemm.hip's dispatcher (arguments validated there; shape conditions: gemm256c_ok + the caller's N % 256 == 0, residual stride and 2^31 tile spans).
int gemm256c8_dispatch(int epilogue, const void* x, int64_t ldxb, const void* w, int64_t ldwb, const void* bias, void* y, int64_t ldy, int64_t M, int N, int nk,
                       const void* resid, int64_t ldr, const void* gate, const float* sx, const float* sw, int gm_tiles, hipStream_t st, GemmBlocking gb) {
  switch (epilogue) {
    case X2V_EPI_NONE: return launch_gemm256c8<X2V_EPI_NONE>(x, ldxb, w, ldwb, bias, y, ldy, M, N, nk, nullptr, 0, nullptr, sx, sw, gm_tiles, st, gb);
    case X2V_EPI_GELU_TANH: return launch_gemm256c8<X2V_EPI_GELU_TANH>(x, ldxb, w, ldwb, bias, y, ldy, M, N, nk, nullptr, 0, nullptr, sx, sw, gm_tiles, st, gb);
    case X2V_EPI_SILU: return launch_gemm256c8<X2V_EPI_SILU>(x, ldxb, w, ldwb, bias, y, ldy, M, N, nk, nullptr, 0, nullptr, sx, sw, gm_tiles, st, gb);
    case X2V_EPI_RESIDUAL: return launch_gemm256c8<X2V_EPI_RESIDUAL>(x, ldxb, w, ldwb, bias, y, ldy, M, N, nk, resid, ldr, gate, sx, sw, gm_tiles, st, gb);
    default: set_error("gemm_fp8: unknown epilogue %d", epilogue); return X2V_E_ARG;
  }
}

}  // namespace x2v
