// y[M,N] = epi(x[M,K] . W[N,K]^T + bias), bf16 — gemm256s.hip's single-stream 256x256 kernel as a CONTINUOUS pipeline over output tiles.
// Same contract, operand layouts, epilogues, rounding points, MFMA and k order as gemm.hip / gemm256.hip / gemm256s.hip: bit-equal results
// (asserted in tests/test_gpu_bench_shapes.py::test_gemm_continuous_pipeline_equals_one_tile_per_workgroup against the one-tile-per-workgroup kernel).
//
// Bound: MFMA (bf16 dense peak ~2.5 PFLOP/s; at this board's 1400 W limit a bare 16x16x32 MFMA loop with GEMM-like LDS traffic sustains
// ~1.8 PFLOP/s, tools/probes/mfma_power_probe.hip).  Algorithmic work 2*M*N*K FLOP per launch.
//
// What this file changes.  A one-workgroup-per-CU kernel has nothing on its matrix pipe during its own prologue
// (first operand tiles in flight: a full memory latency) and epilogue (accumulators -> LDS -> barrier -> 16-byte stores): ~5 us of a 117 us
// output tile at K = 5120, ~20 % of a tile at K = 1536.  Here
//   * a workgroup is PERSISTENT: it walks the output tiles of its XCD's chunk of the grouped tile order (the same assignment an in-order
//     dispatch of gemm256s gives), grid = CUs;
//   * the K loop is ONE software pipeline over (output tile, K tile) pairs: the "tile t+1 / t+2" LDS-DMA cursors of gemm256s's slot plan simply
//     run on into the next output tile with that tile's buffer descriptors, so after the first output tile there is no prologue — when an output
//     tile's last K tile retires, K tile 0 of the next one has landed in LDS and K tile 1 is in flight;
//   * the epilogue needs no workgroup barrier and no tile-wide LDS staging — LDS is busy holding the next output tile's operands: each wave
//     transposes its own 128 x 128 part through a private 4 KiB strip, 16 rows at a time, and stores whole 128-byte lines (see "epilogue" below);
//   * the per-column epilogue operands (bias, gate) and the first residual chunks are requested inside the LAST K tile's slot stream (one buffer
//     load in an otherwise empty slot), the following residual chunks one x block ahead of their use;
//   * the first k-step of an output tile issues its 64 MFMAs with C = 0 (inline constant) instead of 256 v_accvgpr_write.
// Measured (profiles/r04_gemm_continuous_ab.txt): +5..12 % at K = 1536 (a 25 us tile), -0.3..+1.8 % at K = 5120 / 13824, where the board's power cap and
// not the per-tile overhead sets the rate.  The first form of the epilogue — 8-byte stores straight from the accumulator layout, four partial writes per
// 128-byte line — was 5 % (K = 5120) to 13 % (K = 1536) slower than not storing at all; hence the transposition strip.
// Slot plan of a K tile: gemm256s.hip's (constants below), LDS images and swizzle: gemm256s.hip's.  Needs an even number of K tiles >= 4 (every
// output tile then starts in LDS stage 0) and N a multiple of 256; other shapes, the V^T output mode and y blocks that are not multiples of a
// wave's 128 columns stay on gemm256s.
// vmcnt: the counted waits written here (LDS-DMA landed) have only LOADS younger than the pieces they wait for; older stores of the previous
// output tile's epilogue only make them stricter.  Waits for register-returning loads (bias, gate, residual) are the compiler's.
// AUDIT after every edit (the accumulator half is invisible to the compiler): `hipcc -S` must show .vgpr_spill_count 0,
// .private_segment_fixed_size 0 and no v_accvgpr_* / a[..] operand outside ;;#ASMSTART / ;;#ASMEND.
#include <type_traits>

#include "x2v_common.h"

namespace x2v {

#ifndef C_STORE_AUX
#define C_STORE_AUX 0  // cache policy of the output stores (A/B builds: 2 = non-temporal)
#endif
#ifndef C_DMA_AUX_A
#define C_DMA_AUX_A 0  // cache policy of the x-operand LDS-DMA (A/B builds)
#endif
constexpr int C_M = 256, C_N = 256;
constexpr int C_OP_BYTES = 256 * 128;          // one operand tile of one stage
constexpr int C_STAGE_BYTES = 2 * C_OP_BYTES;  // W tile | x tile
constexpr int C_LDS_BYTES = 2 * C_STAGE_BYTES;  // 131072: the two stages
constexpr int C_STRIP_BYTES = 16 * 256;         // per wave: the epilogue's transposition strip, one x block (16 rows x 128 bf16) at a time
constexpr int C_LDS_TOTAL = C_LDS_BYTES + 4 * C_STRIP_BYTES;
// MFMA slots of a K tile (128 per wave) at which the other instructions of the stream sit — gemm256s.hip's plan:
//   C_LATE0 + C_STEP i    the last 16 - C_EARLY LDS-DMA pieces of tile t+1            0, 2, .., 30   fragment reads of k-step 1
//   C_FREE                lgkmcnt(0) + barrier "this tile's stage is free"
//   C_FREE + 1 + C_STEP i the first C_EARLY pieces of tile t+2
//   C_READY               vmcnt + barrier "tile t+1 has landed"                      C_READY + 2, + 4, ..   fragment reads of k-step 0 of t+1
//   (LAST K tile of an output tile only) one epilogue-operand buffer load in each slot of [C_X0, C_READY) that holds no LDS-DMA piece
// (Round 6, profiles/r06_gemm_vs_hipblaslt_pmc_and_knockouts.txt: pieces every 5 / 6 slots, READY at 106 with a read per slot, k-step-1 reads a slot apart, FREE at 44 and a
//  W-major k-step order are all nil or slower at the step's shapes; builds without the vmcnt / lgkmcnt waits below — invalid results — gain 0.1-0.6 %: the plan is at its optimum.)
constexpr int C_STEP = 7, C_FREE = 36, C_READY = 94, C_LATE0 = 3;
constexpr int C_EARLY = (127 - C_FREE - 1) / C_STEP + 1 < 16 ? (127 - C_FREE - 1) / C_STEP + 1 : 16;  // pieces of tile t+2 that fit behind C_FREE
constexpr int C_NEWER = (C_READY - C_FREE - 1) / C_STEP + 1 < C_EARLY ? (C_READY - C_FREE - 1) / C_STEP + 1 : C_EARLY;  // of them issued before C_READY
static_assert(C_LATE0 + (16 - C_EARLY - 1) * C_STEP < C_FREE && C_READY + 2 + 30 <= 127, "slot plan");
constexpr int C_X0 = C_FREE + 2;  // first slot that may carry an epilogue-operand load
constexpr bool c_dma_slot(int n) { return n > C_FREE && (n - C_FREE - 1) % C_STEP == 0 && (n - C_FREE - 1) / C_STEP < C_EARLY; }
constexpr int c_xload_index(int n) {  // which epilogue-operand load sits in slot n (-1: none)
  if (n < C_X0 || n >= C_READY || c_dma_slot(n)) return -1;
  int idx = 0;
  for (int s = C_X0; s < n; ++s)
    if (!c_dma_slot(s)) ++idx;
  return idx;
}
constexpr int c_xload_slots() {
  int c = 0;
  for (int s = C_X0; s < C_READY; ++s)
    if (!c_dma_slot(s)) ++c;
  return c;
}
static_assert(c_xload_slots() >= 13, "bias (8) + gate (1) + the residual chunks of one x block (4) need a slot each");

typedef __attribute__((address_space(3))) void* c_lds_ptr_t;
typedef unsigned int c_u32x2_t __attribute__((ext_vector_type(2)));

// Every asm statement that touches the accumulator half names ALL of it as clobbered (gemm256s.hip: the round-3 incident).
#define C_AGPRS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"

template <int B, int E, class F>
__device__ __forceinline__ void c_for(F&& f) {  // f(integral_constant<int, i>) for i = B .. E-1, fully unrolled with constant indices
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    c_for<B + 1, E>(f);
  }
}

// accumulator tile I (= x block * 8 + W block) is a[4 I : 4 I + 3]
template <int I>
__device__ __forceinline__ void c_mfma(const bf16x8_t& wf, const bf16x8_t& xf) {
  asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(wf), "v"(xf), "i"(4 * I), "i"(4 * I + 3) : C_AGPRS);
}
template <int I>
__device__ __forceinline__ void c_mfma_first(const bf16x8_t& wf, const bf16x8_t& xf) {  // first k-step of an output tile: C = 0
  asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, 0" ::"v"(wf), "v"(xf), "i"(4 * I), "i"(4 * I + 3) : C_AGPRS);
}
template <int R>
__device__ __forceinline__ float c_acc_read() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(x) : "i"(R) : C_AGPRS);
  return x;
}

// LDS-DMA cursor: which K tile of which output tile a piece belongs to.  Descriptors are wave-uniform (SGPRs).
struct CCursor {
  __amdgpu_buffer_rsrc_t ra, rw;
  unsigned kw;  // byte offset of the K tile within a W row
  unsigned ka;  // byte offset of the K tile within an x row (K-blocked x: GemmBlocking)
  int kc;       // index of the K tile within its K block of x
  int k;        // index of the K tile within the output tile
};

template <int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm256c_kernel(
    const char* __restrict__ A, int64_t lda_bytes, const char* __restrict__ W, int64_t ldw_bytes, const unsigned short* __restrict__ bias, unsigned short* Y,
    int64_t ldy, int64_t M, int N, int nk, const unsigned short* resid, int64_t ldr, const unsigned short* __restrict__ gate, int ntm, int ntn, int gm_tiles,
    GemmBlocking gb) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  asm volatile("" ::: C_AGPRS);  // the accumulator half belongs to the asm statements of this kernel
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const int r16 = lane & 15, g16 = lane >> 4;

  // ---- this workgroup's output tiles: positions v, v + vstep, .. < vend of the grouped tile order; XCD x (= blockIdx % 8, the dispatcher's
  //      placement) owns the contiguous chunk x of that order, its workgroups take the chunk's positions round-robin
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

  // tile coordinates of position p (grouped ordering: gm_tiles m-tiles x all n-tiles per group, as gemm256.hip)
  auto coords = [&](unsigned p, int& tm, int& tn) {
    const unsigned group = p / per_group, in_g = p % per_group;
    const unsigned first_m = group * GM;
    const unsigned gsz = min((unsigned)ntm - first_m, GM);
    tm = (int)(first_m + in_g % gsz);
    tn = (int)(in_g / gsz);
  };
  // buffer descriptors over a tile's valid rows: rows past M / N read as zero through the bounds check; `live` false: an empty range (every
  // piece reads as zero) — what the cursors point at behind the workgroup's last output tile
  auto operands = [&](int tm, int tn, bool live, __amdgpu_buffer_rsrc_t& ra, __amdgpu_buffer_rsrc_t& rw) {
    const int64_t m0 = (int64_t)tm * C_M;
    const int n0 = tn * C_N;
    const int rows_a = (int)min((int64_t)C_M, M - m0), rows_w = min(C_N, N - n0);
    ra = __builtin_amdgcn_make_buffer_rsrc((void*)(A + m0 * lda_bytes), 0, live ? (unsigned)((rows_a - 1) * lda_bytes) + a_span : 0u, 0x00020000);
    rw = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (int64_t)n0 * ldw_bytes), 0, live ? (unsigned)((rows_w - 1) * ldw_bytes) + row_bytes : 0u, 0x00020000);
  };
  auto advance = [&](CCursor& c, const __amdgpu_buffer_rsrc_t& nra, const __amdgpu_buffer_rsrc_t& nrw) {  // to the next K tile of the pipeline
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
  // piece P_ in 0..15 of the K tile cursor CUR_ points at: 0..7 = W pieces, 8..15 = x pieces
#define C_DMA(P_, STAGE_, CUR_)                                                                                                                 \
  {                                                                                                                                            \
    constexpr int i_ = (P_) & 7;                                                                                                               \
    if constexpr ((P_) < 8)                                                                                                                    \
      __builtin_amdgcn_raw_ptr_buffer_load_lds((CUR_).rw, (c_lds_ptr_t)(smem + (STAGE_) * C_STAGE_BYTES + wid * 8192 + i_ * 1024), 16, w_voff[i_ & 1],           \
                                               (CUR_).kw + (unsigned)(i_ >> 1) * w_j, 0, 0);                                                   \
    else                                                                                                                                       \
      __builtin_amdgcn_raw_ptr_buffer_load_lds((CUR_).ra, (c_lds_ptr_t)(smem + (STAGE_) * C_STAGE_BYTES + C_OP_BYTES + wid * 8192 + i_ * 1024), 16,              \
                                               a_voff[i_ & 1], (CUR_).ka + (unsigned)(i_ >> 1) * a_j, 0, C_DMA_AUX_A);                         \
  }

  // ---- fragment addresses (16x16x32: row r16 of a 16-row block, 16-byte chunk ks*4 + g16), block offsets travel as immediates
  int rd_x[2], rd_w[2];
  {
    const int swz = (r16 >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int o = r16 * 128 + ((((ks << 2) | g16) ^ swz) << 4);
      rd_x[ks] = o + C_OP_BYTES + wr * 16384;
      rd_w[ks] = o + wc * 16384;
    }
  }
  bf16x8_t fx[2][8], fw[2][8];
  // fragment R_ in 0..15 of k-step KS_ of the tile in stage STAGE_; order x0, W0..W7, x1..x7 (the first MFMA of a k-step needs x0 and W0)
#define C_READ(R_, STAGE_, KS_)                                                                                                                 \
  {                                                                                                                                            \
    if constexpr ((R_) == 0) fx[KS_][0] = *reinterpret_cast<const bf16x8_t*>(smem + (STAGE_) * C_STAGE_BYTES + rd_x[KS_]);                      \
    else if constexpr ((R_) <= 8) fw[KS_][(R_) - 1] = *reinterpret_cast<const bf16x8_t*>(smem + (STAGE_) * C_STAGE_BYTES + ((R_) - 1) * 2048 + rd_w[KS_]); \
    else fx[KS_][(R_) - 8] = *reinterpret_cast<const bf16x8_t*>(smem + (STAGE_) * C_STAGE_BYTES + ((R_) - 8) * 2048 + rd_x[KS_]);               \
  }
#define C_SB() __builtin_amdgcn_sched_barrier(0)

  // ---- epilogue of the CURRENT output tile.  LDS holds the next output tile's operands, so the accumulators cannot be staged tile-wide as in
  //      gemm256s; going straight from the accumulator layout to memory (8 bytes per lane, four partial writes per 128-byte line) was measured
  //      5 % (K = 5120) to 13 % (K = 1536) slower than not storing at all (profiles/r04_gemm_continuous_*).  So each WAVE transposes its own
  //      128 x 128 part through a private 4 KiB strip of the LDS the two stages leave free — 16 rows (one x block) at a time, no workgroup
  //      barrier: phase A writes the 8 accumulator tiles of an x block (+ bias, activation, bf16) as 8-byte pieces into a [16 rows][256 B] image
  //      (16-byte chunk index XOR row: conflict-free both ways), phase B reads it back row-major, 16 bytes per lane, and issues 4 stores of
  //      4 rows x 256 contiguous bytes (whole 128-byte lines), combining with the residual chunk of the same shape first.  LDS executes a wave's
  //      instructions in order, so phase A of the next x block may overwrite the strip right behind phase B's reads.
  //      Addressing: N is a multiple of 256 and y blocks (GemmBlocking) are whole multiples of a wave's 128 columns (dispatcher), so a wave's 128
  //      columns are contiguous in memory and start at a wave-uniform offset; a phase-B access of x block xb, instruction i is
  //      vector offset = lane part (row wr*128 + (lane>>4), columns 8 (lane&15).. of the wave) or the "row does not exist" mark 0x80000000 (outside
  //      every descriptor's range: the bounds check drops the access)  +  scalar offset = wave's column base + (16 xb + 4 i) rows.
  //      The per-column operands (bias in phase-A layout, gate in phase-B layout) and the residual chunks of x block 0 are requested inside the
  //      LAST K tile's slot stream; the residual chunks of x block xb + 1 while x block xb is processed.
  typedef unsigned int c_u32x4_t __attribute__((ext_vector_type(4)));
  c_u32x2_t e_bias[8];
  c_u32x4_t e_gate4, e_res[2][4];
  __amdgpu_buffer_rsrc_t r_y, r_res, r_bias, r_gate;
  char* const strip = smem + C_LDS_BYTES + wid * C_STRIP_BYTES;
  const int l4 = lane >> 4, c16 = lane & 15;
  const unsigned lane_off = (unsigned)((wr * 128 + l4) * ldy * 2) + (unsigned)(16 * c16);  // phase B: bytes from the tile's first row / the wave's first column
  unsigned s_col = 0u;    // the wave's first column in the output / residual row, bytes (wave-uniform)
  unsigned s_bias = 0u;   // the wave's first column in bias / gate, bytes
  int rows_left = 0;      // valid rows of the current tile below row wr*128 + (lane>>4): phase-B row 16 xb + 4 i of the lane exists iff it is < rows_left
  const unsigned y_row = (unsigned)(ldy * 2);  // one row of y (and of the residual: ldr == ldy, y row-major — dispatcher), bytes
  constexpr bool RES = EPI == X2V_EPI_RESIDUAL;
  constexpr int NXLOAD = RES ? 8 + 1 + 4 : 8;  // bias (8) [+ gate (1) + the residual chunks of x block 0 (4)]

  // Set up the epilogue addressing of output tile (tm, tn).  Runs when the tile becomes current (scalar instructions + one vector subtract).
  auto epilogue_setup = [&](int tm, int tn) {
    const int64_t m0 = (int64_t)tm * C_M;
    const int gn0 = tn * C_N + wc * 128;  // first column of this wave (wave-uniform)
    r_y = __builtin_amdgcn_make_buffer_rsrc((void*)(Y + m0 * ldy), 0, 0x80000000u, 0x00020000);
    r_bias = __builtin_amdgcn_make_buffer_rsrc((void*)bias, 0, bias != nullptr ? (unsigned)N * 2u : 0u, 0x00020000);
    unsigned col = (unsigned)gn0;
    if (gb.y_cbw > 0) {  // N-blocked y: column n at (n / y_cbw) * y_cbs + n % y_cbw elements from the row's start
      const unsigned qb = (unsigned)gn0 / (unsigned)gb.y_cbw;
      col = qb * (unsigned)gb.y_cbs + ((unsigned)gn0 - qb * (unsigned)gb.y_cbw);
    }
    s_col = col * 2u;
    s_bias = (unsigned)gn0 * 2u;
    rows_left = (int)min((int64_t)C_M, M - m0) - wr * 128 - l4;
    if constexpr (RES) {
      r_res = __builtin_amdgcn_make_buffer_rsrc((void*)(resid + m0 * ldy), 0, 0x80000000u, 0x00020000);
      r_gate = __builtin_amdgcn_make_buffer_rsrc((void*)gate, 0, gate != nullptr ? (unsigned)N * 2u : 0u, 0x00020000);
    }
  };
  auto row_voff = [&](int row16) { return row16 < rows_left ? lane_off : 0x80000000u; };  // phase-B vector offset of local row `row16` (= 16 xb + 4 i)
  auto res_load = [&](auto xbc, auto ic) {  // residual chunk i of x block xb: the 16 bytes phase B's store i of that block will overwrite
    constexpr int xb = decltype(xbc)::value, i = decltype(ic)::value;
    e_res[xb & 1][i] = __builtin_bit_cast(c_u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r_res, row_voff(16 * xb + 4 * i), s_col + (unsigned)(16 * xb + 4 * i) * y_row, 0));
  };
  // epilogue-operand load J_ of the current output tile (LAST K tile)
  auto xload = [&](auto jc) {
    constexpr int J = decltype(jc)::value;
    if constexpr (J < 8) {
      e_bias[J] = __builtin_bit_cast(c_u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(r_bias, (unsigned)(8 * g16), s_bias + (unsigned)(J * 32), 0));
    } else if constexpr (RES && J == 8) {
      e_gate4 = __builtin_bit_cast(c_u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r_gate, (unsigned)(16 * c16), s_bias, 0));
    } else if constexpr (RES && J < 8 + 1 + 4) {
      res_load(std::integral_constant<int, 0>{}, std::integral_constant<int, J - 9>{});
    }
  };

  // One K tile of the pipeline.  ST = its LDS stage; FIRST: K tile 0 of an output tile (k-step 0 starts the accumulators from 0);
  // LAST: the output tile's last K tile (epilogue-operand loads ride in its slots; the k-step-0 fragments of the next output tile are read
  // behind the epilogue instead of here — the epilogue needs the registers).  c1 / c2: cursors of the pipeline's next / next-but-one K tile.
  auto tile = [&](auto stc, auto firstc, auto lastc, const CCursor& c1, const CCursor& c2) {
    constexpr int ST = decltype(stc)::value;
    constexpr bool FIRST = decltype(firstc)::value != 0, LAST = decltype(lastc)::value != 0;
    c_for<0, 128>([&](auto nc) {
      constexpr int n = decltype(nc)::value, ks = n >> 6, xb = (n >> 3) & 7, wb = n & 7;
      if constexpr (FIRST && ks == 0) c_mfma_first<xb * 8 + wb>(fw[ks][wb], fx[ks][xb]);
      else c_mfma<xb * 8 + wb>(fw[ks][wb], fx[ks][xb]);
      if constexpr (n < 32 && (n & 1) == 0) C_READ(n >> 1, ST, 1)  // k-step 1 of this tile
      // the last 16 - C_EARLY pieces of tile t+1 (its stage was freed by the previous tile's first barrier)
      if constexpr (n >= C_LATE0 && (n - C_LATE0) % C_STEP == 0 && (n - C_LATE0) / C_STEP < 16 - C_EARLY) C_DMA(C_EARLY + (n - C_LATE0) / C_STEP, ST ^ 1, c1)
      if constexpr (n == C_FREE) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every fragment of this tile is in registers: the stage may be overwritten
        __builtin_amdgcn_s_barrier();
      }
      // the first C_EARLY pieces of tile t+2 into this tile's stage
      if constexpr (c_dma_slot(n)) C_DMA((n - C_FREE - 1) / C_STEP, ST, c2)
      if constexpr (LAST && c_xload_index(n) >= 0 && c_xload_index(n) < NXLOAD) xload(std::integral_constant<int, c_xload_index(n)>{});
      if constexpr (n == C_READY) {
        // tile t+1 has landed; younger loads may stay in flight: the pieces of tile t+2 issued so far in this tile (+ the epilogue operands)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C_NEWER + (LAST ? NXLOAD : 0)) : "memory");
        __builtin_amdgcn_s_barrier();
      }
      if constexpr (!LAST && n > C_READY + 1 && (n & 1) == 0) C_READ((n - C_READY - 2) >> 1, ST ^ 1, 0)  // k-step 0 of the next tile
      C_SB();
    });
  };

  // Phase A of accumulator tile I: value e of the lane = row wr*128 + xb*16 + r16, column wc*128 + wb*16 + 4*g16 + e of the output tile: bias,
  // activation, bf16, 8 bytes into the strip's image of x block xb.
  auto epi_phase_a = [&](auto ic) {
    constexpr int I = decltype(ic)::value, wb = I & 7;
    float vv[4] = {c_acc_read<4 * I + 0>(), c_acc_read<4 * I + 1>(), c_acc_read<4 * I + 2>(), c_acc_read<4 * I + 3>()};
    vv[0] += bf_lo(e_bias[wb].x);
    vv[1] += bf_hi(e_bias[wb].x);
    vv[2] += bf_lo(e_bias[wb].y);
    vv[3] += bf_hi(e_bias[wb].y);
    if (EPI == X2V_EPI_GELU_TANH) {
#pragma unroll
      for (int e = 0; e < 4; ++e) vv[e] = gelu_tanh_f(rbf(vv[e]));
    } else if (EPI == X2V_EPI_SILU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) vv[e] = silu_f(rbf(vv[e]));
    }
    // [row r16][256 B], 16-byte chunk (2 wb + (g16 >> 1)) ^ r16, half g16 & 1
    const int wa = r16 * 256 + (((((g16 >> 1) ^ r16) << 4)) ^ (wb * 32)) + (g16 & 1) * 8;
    *reinterpret_cast<c_u32x2_t*>(strip + wa) = c_u32x2_t{pack_bf2(vv[0], vv[1]), pack_bf2(vv[2], vv[3])};
  };
  // Phase B, instruction i of x block xb: local rows 16 xb + 4 i + (lane >> 4), this lane's 8 columns 8 (lane & 15)..: strip -> (residual) -> memory
  auto epi_phase_b = [&](auto xbc, auto ic) {
    constexpr int xb = decltype(xbc)::value, i = decltype(ic)::value;
    const int ra = (4 * i + l4) * 256 + ((c16 ^ (4 * i + l4)) << 4);
    c_u32x4_t yv4 = *reinterpret_cast<const c_u32x4_t*>(strip + ra);
    if constexpr (RES) {
      float yv[8], xv[8], gv[8], ov[8];
      unpack8(__builtin_bit_cast(uint4, yv4), yv);
      unpack8(__builtin_bit_cast(uint4, e_res[xb & 1][i]), xv);
      unpack8(__builtin_bit_cast(uint4, e_gate4), gv);
#pragma unroll
      for (int e = 0; e < 8; ++e) ov[e] = xv[e] + rbf(yv[e] * gv[e]);
      yv4 = __builtin_bit_cast(c_u32x4_t, pack8(ov));
    }
    __builtin_amdgcn_raw_buffer_store_b128(yv4, r_y, row_voff(16 * xb + 4 * i), s_col + (unsigned)(16 * xb + 4 * i) * y_row, C_STORE_AUX);
  };
  auto epilogue = [&]() {
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");  // the last MFMAs' results before the accumulator reads below
    if constexpr (RES) {
      if (gate == nullptr) e_gate4 = c_u32x4_t{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};  // x + y: gate 1.0 gives the same bits (y is already bf16)
    }
    // (The walk software-pipelined by one x block — gemm256c8.hip's form: A(xb + 1) under the strip reads of block xb — was measured in round 5 and is
    //  a wash in bf16: 5120->13824 plain +1..2 %, GELU -1 %, 13824->5120 +0.8 %, K = 1536 inside the noise; profiles/r05_call1_*.  Not kept.)
    c_for<0, 8>([&](auto xbc) {
      constexpr int xb = decltype(xbc)::value;
      if constexpr (RES && xb < 7) {  // the next x block's residual chunks fly under this block's arithmetic
        c_for<0, 4>([&](auto ic) { res_load(std::integral_constant<int, xb + 1>{}, ic); });
      }
      c_for<0, 8>([&](auto wbc) { epi_phase_a(std::integral_constant<int, xb * 8 + decltype(wbc)::value>{}); });
      C_SB();  // one x block at a time: left alone, the scheduler hoists the unpacking of every operand and spills
      c_for<0, 4>([&](auto ic) { epi_phase_b(xbc, ic); });
      C_SB();
    });
  };

  // ---- pipeline start: K tile 0 of the first output tile and the first C_EARLY pieces of its K tile 1 in flight, tile 0 landed, k-step 0 in registers
  using c0 = std::integral_constant<int, 0>;
  using c1t = std::integral_constant<int, 1>;
  int tm, tn;
  coords(v, tm, tn);
  CCursor cu1, cu2;
  {
    __amdgpu_buffer_rsrc_t ra, rw;
    operands(tm, tn, true, ra, rw);
    cu1 = CCursor{ra, rw, 0u, 0u, 0, 0};
    c_for<0, 16>([&](auto pc) { C_DMA(decltype(pc)::value, 0, cu1) });
    advance(cu1, ra, rw);  // K tile 1 (nk >= 4: no wrap here)
    c_for<0, C_EARLY>([&](auto pc) { C_DMA(decltype(pc)::value, 1, cu1) });
    cu2 = cu1;
    advance(cu2, ra, rw);  // K tile 2
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C_EARLY) : "memory");
  }
  __builtin_amdgcn_s_barrier();
  C_SB();

  for (;;) {
    // current output tile (tm, tn); the one behind it, whose K tiles the cursors run into near the end of this one
    const bool has_next = v + vstep < vend;
    int ntm_ = tm, ntn_ = tn;
    if (has_next) coords(v + vstep, ntm_, ntn_);
    __amdgpu_buffer_rsrc_t nra, nrw;
    operands(ntm_, ntn_, has_next, nra, nrw);
    epilogue_setup(tm, tn);
    c_for<0, 16>([&](auto rc) { C_READ(decltype(rc)::value, 0, 0) });  // k-step 0 of K tile 0 (stage 0: nk is even)
    C_SB();
#define C_STEP_CURSORS()      \
  {                           \
    cu1 = cu2;                \
    advance(cu2, nra, nrw);   \
  }
    tile(c0{}, c1t{}, c0{}, cu1, cu2);  // K tile 0
    C_STEP_CURSORS()
    tile(c1t{}, c0{}, c0{}, cu1, cu2);  // K tile 1
    C_STEP_CURSORS()
    for (int t = 2; t < nk - 2; t += 2) {
      tile(c0{}, c0{}, c0{}, cu1, cu2);
      C_STEP_CURSORS()
      tile(c1t{}, c0{}, c0{}, cu1, cu2);
      C_STEP_CURSORS()
    }
    tile(c0{}, c0{}, c0{}, cu1, cu2);  // K tile nk - 2: cu2 already points at K tile 0 of the next output tile
    C_STEP_CURSORS()
    tile(c1t{}, c0{}, c1t{}, cu1, cu2);  // K tile nk - 1 (LAST)
    C_STEP_CURSORS()
#undef C_STEP_CURSORS
    epilogue();
    if (!has_next) break;
    v += vstep;
    tm = ntm_;
    tn = ntn_;
  }
  // the pieces issued for the (non-existent) K tiles behind the last output tile read an empty range; let them retire before the LDS goes away
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef C_READ
#undef C_DMA
#undef C_SB
#endif
}

template <int EPI>
static int launch_gemm256c(const void* x, int64_t ldx_bytes, const void* w, int64_t ldw_bytes, const void* bias, void* y, int64_t ldy, int64_t M, int N, int nk,
                           const void* resid, int64_t ldr, const void* gate, int gm_tiles, hipStream_t st, GemmBlocking gb) {
  if (gm_tiles <= 0) gm_tiles = 4;
  const int ntm = (int)((M + C_M - 1) / C_M), ntn = (N + C_N - 1) / C_N;
  int rc = ensure_dynamic_lds((const void*)gemm256c_kernel<EPI>, C_LDS_TOTAL, "gemm256c attr");
  if (rc != X2V_OK) return rc;
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
    return n & ~7;  // whole XCD octets
  }();
  const unsigned nblk = (unsigned)ntm * (unsigned)ntn;
  const unsigned grid = nblk > (unsigned)cus ? (unsigned)cus : nblk;
  hipLaunchKernelGGL((gemm256c_kernel<EPI>), dim3(grid), dim3(256), C_LDS_TOTAL, st, (const char*)x, ldx_bytes, (const char*)w, ldw_bytes, (const unsigned short*)bias,
                     (unsigned short*)y, ldy, M, N, nk, (const unsigned short*)resid, ldr, (const unsigned short*)gate, ntm, ntn, gm_tiles, gb);
  X2V_LAUNCH_CHECK("gemm256c launch");
  return X2V_OK;
}

// Shapes the continuous kernel takes (the others stay on gemm256s): an even number of K tiles >= 4; y blocks that are whole multiples of a
// wave's 128 columns (the caller adds: N a multiple of 256, a residual with y's row stride, tile spans below the epilogue descriptors' range).
bool gemm256c_ok(int nk, const GemmBlocking& gb) { return nk >= 4 && (nk & 1) == 0 && (gb.y_cbw <= 0 || gb.y_cbw % 128 == 0); }

// Called by gemm.hip's dispatcher (arguments already validated there; ld*_bytes < 16 MiB and the 32-bit tile spans checked by the caller).
int gemm256c_dispatch(int epilogue, const void* x, int64_t ldxb, const void* w, int64_t ldwb, const void* bias, void* y, int64_t ldy, int64_t M, int N, int nk,
                      const void* resid, int64_t ldr, const void* gate, int gm_tiles, hipStream_t st, GemmBlocking gb) {
  switch (epilogue) {
    case X2V_EPI_NONE: return launch_gemm256c<X2V_EPI_NONE>(x, ldxb, w, ldwb, bias, y, ldy, M, N, nk, nullptr, 0, nullptr, gm_tiles, st, gb);
    case X2V_EPI_GELU_TANH: return launch_gemm256c<X2V_EPI_GELU_TANH>(x, ldxb, w, ldwb, bias, y, ldy, M, N, nk, nullptr, 0, nullptr, gm_tiles, st, gb);
    case X2V_EPI_SILU: return launch_gemm256c<X2V_EPI_SILU>(x, ldxb, w, ldwb, bias, y, ldy, M, N, nk, nullptr, 0, nullptr, gm_tiles, st, gb);
    case X2V_EPI_RESIDUAL: return launch_gemm256c<X2V_EPI_RESIDUAL>(x, ldxb, w, ldwb, bias, y, ldy, M, N, nk, resid, ldr, gate, gm_tiles, st, gb);
    default: set_error("gemm: unknown epilogue %d", epilogue); return X2V_E_ARG;
  }
}

}  // namespace x2v
