/*
 * x2v.h — C-ABI of libx2v_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the LightX2V DiT
 * denoise-step hot path.  This is the drop-in boundary: plain pointers, sizes and a
 * hipStream_t; no torch types.  Each entry point names the reference call site it replaces
 * (paths relative to /root/reference/lightx2v/).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless stated; bf16 tensors are passed as `const void*`/`void*`
 *     (raw bfloat16 bits, little-endian); fp8 tensors are OCP e4m3fn bytes (gfx950 format, not fnuz)
 *   - `ld*` / strides are in ELEMENTS; rows must be 16-byte aligned (ld % 8 == 0 for bf16)
 *   - `stream` is a hipStream_t passed as void* (the Python shim passes
 *     torch.cuda.current_stream().cuda_stream); every call is asynchronous, re-entrant, never
 *     synchronises the device and keeps no mutable global state besides the thread-local error string
 *   - return value: X2V_OK (0) or a negative X2V_E_* code; x2v_last_error() then describes it.
 *     No exceptions or exit() cross the ABI (reference convention: Python exceptions / TORCH_CHECK →
 *     RuntimeError, lightx2v_kernel/csrc/gemm/mxfp8_scaled_mm_kernels_sm120.cu:13-25; the shim raises
 *     RuntimeError from the code)
 */
#ifndef X2V_H
#define X2V_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define X2V_OK 0
#define X2V_E_SHAPE (-1)  /* unsupported shape / size constraint violated */
#define X2V_E_ALIGN (-2)  /* pointer or leading dimension not 16-byte aligned */
#define X2V_E_ARCH (-3)   /* device is not gfx950 */
#define X2V_E_HIP (-4)    /* a HIP runtime call failed */
#define X2V_E_ARG (-5)    /* null pointer / bad enum */

/* GEMM epilogues (x2v_gemm_bf16 / x2v_gemm_fp8) */
#define X2V_EPI_NONE 0      /* y = bf16(acc + bias) */
#define X2V_EPI_GELU_TANH 1 /* y = bf16(gelu_tanh(bf16(acc + bias)))          (transformer_infer.py:488-492) */
#define X2V_EPI_RESIDUAL 2  /* y = bf16(resid + bf16(bf16(acc + bias) * gate)) (transformer_infer.py:402,468,503);
                               gate may be NULL (plain add); y may alias resid */
#define X2V_EPI_SILU 3      /* y = bf16(silu(bf16(acc + bias)))                (pre_infer.py:74-76) */

/* rounding model of the norm kernels */
#define X2V_ROUND_FP32 0 /* fp32 statistics, one final rounding (what sgl_kernel.rmsnorm does on the reference's GPU path) */
#define X2V_ROUND_REF 1  /* reproduce the reference's bf16 torch chain rounding for rounding (rms_norm_weight.py:111-113) */

/* Library / device. x2v_init checks that `device` is gfx950 and caches its properties. */
int x2v_init(int device);
const char* x2v_last_error(void);
const char* x2v_version(void);
int x2v_device_info(int device, int* cu_count, int* lds_bytes_per_cu, char* arch_name, int arch_name_len);

/* The process-wide A/B switches this library reads ONCE from the environment, as "NAME=value NAME=value ..." (effective values, defaults
 * included): X2V_GEMM_CONTINUOUS, X2V_GEMM_FP8_CONTINUOUS (kernel forms, same bits), X2V_ATTN_MAP, X2V_ATTN_ROT (launch forms of
 * x2v_attn_fwd_bf16_vt; ROT changes the key-walk order, i.e. low bits).  The reference has no counterpart (its kernel choice is the config
 * string, utils/registry_factory.py:47-56); bench.py prints the string into its JSON line so that a stray variable on a rank is visible. */
int x2v_switches(char* buf, int buf_len);

/* y[M,D] = x * rsqrt(mean(x^2) + eps) * w     — replaces RMSWeight/RMSWeightSgl.apply
 * (common/ops/norm/rms_norm_weight.py:53-118; sgl_kernel.rmsnorm :102-108).  D % 8 == 0, D <= 16384.
 * y may alias x. */
int x2v_rmsnorm_bf16(const void* x, int64_t ldx, const void* w, void* y, int64_t ldy, int64_t M, int D, float eps, int round_mode,
                     void* stream);

/* y[M,D] = LN(x; w,b,eps) [* (1 + scale) + shift]   — replaces LNWeight.apply (layer_norm_weight.py:78-111)
 * fused with the adaLN modulate `norm_out.mul_(1 + scale).add_(shift)` (transformer_infer.py:329-334,
 * 481-484; post_infer.py:24-28).  w,b (affine, norm3) and scale,shift (modulate, [D] bf16) are each
 * optional (NULL).  Rounding follows the reference chain: bf16 after LN, after the multiply, after the add. */
int x2v_layernorm_bf16(const void* x, int64_t ldx, const void* w, const void* b, const void* scale, const void* shift, void* y, int64_t ldy,
                       int64_t M, int D, float eps, void* stream);

/* Same, selecting the kernel form (tuning / validation hook): 0 = by shape (what x2v_layernorm_bf16 does: the persistent
 * "streaming" kernel — next row prefetched under the current row's reductions, per-channel operands resident in registers —
 * when 512 < D <= 8192 and M is at least twice the resident grid, else one block per row), 1 = one block per row,
 * 2 = streaming (X2V_E_SHAPE outside 512 < D <= 8192).  The forms are bit-identical. */
int x2v_layernorm_bf16_variant(const void* x, int64_t ldx, const void* w, const void* b, const void* scale, const void* shift, void* y, int64_t ldy,
                               int64_t M, int D, float eps, int variant, void* stream);

/* In-place q,k <- RoPE3D(RMSNorm_D(q|k)) for Wan self-attention — replaces rms_norm_weight.py apply on
 * q and k (transformer_infer.py:341-342) + compute_freqs/apply_rotary_emb (wan/infer/utils.py:7-20,
 * 107-115).  q,k: [S, H*128] bf16 with token stride ldq/ldk; wq,wk: [H*128] bf16 (NULL = skip the norm);
 * rope_cs: float2 (cos,sin) table [1024][64] = the reference's `freqs` [1024,64] (pre_infer.py:12-19),
 * columns [0,22) t-axis, [22,43) h-axis, [43,64) w-axis; token s has grid position
 * ((s0+s) / (gh*gw), ((s0+s) / gw) % gh, (s0+s) % gw); tokens with s0+s >= gf*gh*gw are rotated by 1
 * (compute_freqs_dist's ones padding, utils.py:78-83).  s0 = first global token of this rank's shard. */
int x2v_rmsnorm_rope_bf16(void* q, int64_t ldq, void* k, int64_t ldk, const void* wq, const void* wk, const void* rope_cs, int64_t S, int H,
                          int64_t s0, int gf, int gh, int gw, float eps, int round_mode, void* stream);

/* Same, with q additionally multiplied by q_out_scale inside its single final rounding (k untouched).  Used by the fused
 * block driver to hand the attention kernel a q that already carries softmax_scale * log2(e)
 * (X2V_ATTN_Q_PRESCALED): numerically one rounding of the scaled value instead of one rounding of the unscaled one. */
int x2v_rmsnorm_rope_scaled_bf16(void* q, int64_t ldq, void* k, int64_t ldk, const void* wq, const void* wk, const void* rope_cs, int64_t S, int H,
                                 int64_t s0, int gf, int gh, int gw, float eps, int round_mode, float q_out_scale, void* stream);

/* Same, selecting the kernel form (tuning / validation hook; bit-identical results): 0 = by shape, 1 = one block per (token, q|k)
 * row, 2 = persistent streaming kernel (a token's q and k rows back to back, its 64 rotation factors gathered once into LDS, the
 * next row prefetched under the reduction; D <= 8192). */
int x2v_rmsnorm_rope_scaled_bf16_variant(void* q, int64_t ldq, void* k, int64_t ldk, const void* wq, const void* wk, const void* rope_cs, int64_t S,
                                         int H, int64_t s0, int gf, int gh, int gw, float eps, int round_mode, float q_out_scale, int variant,
                                         void* stream);

/* x2v_rmsnorm_rope_scaled_bf16 out of place into N-blocked outputs: q/k [S, H*128] are read, column e of token s is written to
 * q_out / k_out[(e / block_cols) * block_stride + s * ldo + e % block_cols] — the seq->head send buffer [N_ranks][S/N][(H/N) d] of the
 * Ulysses exchange (block_cols = (H/N)*128), replacing the reference's view + transpose + contiguous copy (all2all.py:29-33). */
int x2v_rmsnorm_rope_blocked_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* wq, const void* wk, const void* rope_cs, void* q_out, void* k_out,
                                  int64_t ldo, int block_cols, int64_t block_stride, int64_t S, int H, int64_t s0, int gf, int gh, int gw, float eps, int round_mode,
                                  float q_out_scale, void* stream);

/* In-place per-head RMSNorm (d = 128) of q and k [L, H*128] (token strides ldq/ldk, e.g. the column blocks of a fused
 * QKV GEMM output), followed for tokens < l_rope by the real-valued RoPE x*cos + rotate_half(x)*sin with bf16 tables
 * cos/sin [l_rope, 128] — replaces RMSWeightSgl.apply on [L,H,128] (rms_norm_weight.py:102-113; hunyuan
 * transformer_infer.py:271-272,289-290,338-339) + hunyuan/infer/utils_bf16.apply_rotary_emb (:5-31).
 * wq/wk [128] bf16 (NULL = no norm); tokens >= l_rope (the text tokens) are only normalised.  q_out_scale (1 = none;
 * ignored with X2V_ROUND_REF) multiplies q inside its final rounding, see x2v_rmsnorm_rope_scaled_bf16. */
int x2v_headnorm_rope_bf16(void* q, int64_t ldq, void* k, int64_t ldk, const void* wq, const void* wk, const void* cos_tab, const void* sin_tab, int64_t L,
                           int H, int64_t l_rope, float eps, int round_mode, float q_out_scale, void* stream);

/* x2v_headnorm_rope_bf16 in place on HEAD-BLOCKED q and k: heads [j*heads_per_block, (j+1)*heads_per_block) of all L tokens form the matrix
 * [L][heads_per_block*128] (token stride ld) that starts j*block_stride elements into the buffer — the [N_ranks][S/N][(H/N) d] send buffer of
 * the Ulysses exchange, written in that layout by x2v_gemm_bf16_blocked (replaces the reference's view + transpose + contiguous copy of
 * q/k/v in front of all_to_all, attentions/distributed/ulysses/attn.py:36-46 / comm/all2all.py:29-33). */
int x2v_headnorm_rope_blocked_bf16(void* q, void* k, int64_t ld, int heads_per_block, int64_t block_stride, const void* wq, const void* wk, const void* cos_tab,
                                   const void* sin_tab, int64_t L, int H, int64_t l_rope, float eps, int round_mode, float q_out_scale, void* stream);

/* x[M,D] = bf16(x + bf16(y * gate)) (gate [D] bf16, NULL = plain add) — replaces `x.add_(y * gate)`
 * (transformer_infer.py:402,468,503) for callers that do not fuse it into the GEMM epilogue. */
int x2v_gate_residual_bf16(void* x, int64_t ldx, const void* y, int64_t ldy, const void* gate, int64_t M, int D, void* stream);

/* y = act(x) elementwise on [n] bf16: act 1 = gelu-tanh (transformer_infer.py:492), 3 = silu (pre_infer.py:74,76), X2V_ACT_GELU_ERF = the exact
 * GELU of the i2v CLIP-feature MLP (pre_infer.py:106: gelu(approximate="none")). */
#define X2V_ACT_GELU_ERF 4
int x2v_activation_bf16(const void* x, void* y, int64_t n, int act, void* stream);

/* y[M,N] = epi(x[M,K] . W[N,K]^T + bias[N])   — replaces MMWeight.apply = torch.addmm(bias, x, W.t())
 * (common/ops/mm/mm_weight.py:70-96); W is the checkpoint's [N,K] row-major tensor (the reference's
 * `.t()` view, :76).  bf16 in, fp32 MFMA accumulate, bf16 out.  K % 64 == 0, N % 8 == 0, any M >= 1.
 * resid/gate only for X2V_EPI_RESIDUAL (resid [M,N] ld = ldr; gate [N] or NULL). */
int x2v_gemm_bf16(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias, void* y, int64_t ldy, int64_t M, int N, int K,
                  int epilogue, const void* resid, int64_t ldr, const void* gate, void* stream);

/* Same, selecting the kernel (tuning / validation hook).  variant & 0xff: 0 = by shape (what x2v_gemm_bf16 does:
 * the 256x256-tile ping-pong kernel of gemm256.hip when the grid fills the chip, else the 128x128 kernel of
 * gemm.hip), 1 = 128x128 kernel, 2 = 256x256 ping-pong kernel (two waves per SIMD; fp8's and mxfp8's large-shape kernel), 3 = 256x256
 * single-stream kernel (one software-pipelined wave per SIMD; bf16 only, its large-shape kernel) in the form the dispatcher prefers, 4 = its
 * one-output-tile-per-workgroup form (gemm256s.hip), 5 = its continuous-pipeline form (gemm256c.hip: persistent workgroups, the K loop runs on
 * into the next output tile, epilogue straight from the accumulators; needs K a multiple of 128 and >= 256, N a multiple of 256, y blocks
 * that are multiples of 128 columns and resid with y's row stride, else X2V_E_SHAPE; same bits as form 4); (variant >> 8) & 0xff = m-tiles per
 * scheduling group of the 256x256 kernels (0 = default). */
int x2v_gemm_bf16_variant(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias, void* y, int64_t ldy, int64_t M, int N, int K,
                          int epilogue, const void* resid, int64_t ldr, const void* gate, int variant, void* stream);

/* x2v_gemm_bf16 on block-strided operands — what lets the Ulysses exchange buffers (attentions/distributed/comm/all2all.py:6-89) be GEMM
 * operands in place instead of being transposed by copies (all2all.py:29-33, :70-75,87):
 *   x K-blocked (x_kblock > 0): element k of row m at x[(k / x_kblock) * x_kblock_stride + m * ldx + k % x_kblock] — the received
 *     head->seq buffer [N_ranks][S/N][(H/N) d] read as the [S/N, H d] input of the output projection (x_kblock = (H/N) d, a multiple
 *     of 64 elements dividing K; ldx >= x_kblock);
 *   y N-blocked (y_nblock > 0): column n of row m at y[(n / y_nblock) * y_nblock_stride + m * ldy + n % y_nblock] — the q/k/v projection
 *     writing straight into the seq->head send buffer [N_ranks][S/N][(H/N) d] (y_nblock a multiple of 8 dividing N; ldy >= y_nblock;
 *     not with X2V_EPI_RESIDUAL).
 * 0 disables either blocking (plain row-major).  The kernel is chosen by shape as in x2v_gemm_bf16. */
int x2v_gemm_bf16_blocked(const void* x, int64_t ldx, int x_kblock, int64_t x_kblock_stride, const void* w, int64_t ldw, const void* bias, void* y, int64_t ldy,
                          int y_nblock, int64_t y_nblock_stride, int64_t M, int N, int K, int epilogue, const void* resid, int64_t ldr, const void* gate, void* stream);

/* v = x . W^T + bias written as V^T [N/128][ldvt/64][128][64] with tokens in [M, ldvt) zero-filled — x2v_gemm_bf16 followed by
 * x2v_transpose_heads_bf16 in one kernel (same bits): the self-attention v projection (transformer_infer.py:345-347) handing its result to
 * x2v_attn_fwd_bf16_vt without the transposing pass.  Only for shapes that x2v_gemm_kernel_choice maps to kernel 3 (else X2V_E_SHAPE). */
int x2v_gemm_bf16_vt(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias, void* vt, int64_t ldvt, int64_t M, int N, int K, void* stream);

/* Which kernel variant 0 of x2v_gemm_bf16_variant (fp8 = 0) / x2v_gemm_fp8_variant (fp8 = 1) launches for this shape.  Low byte = tile family:
 * 1 = the 128x128 kernel, 2 = the 256x256 fp8 kernels, 3 = the 256x256 single-stream kernel (bf16).  Bit 8 (0x100) = the CONTINUOUS-pipeline
 * form of that family (gemm256c.hip / gemm256c8.hip) for a row-major y and a residual of y's stride; clear = one output tile per workgroup
 * (gemm256s.hip) / the ping-pong kernel (gemm256.hip) — same bits either way, asserted by the equality tests.  Negative = X2V_E_SHAPE.
 * Host-only; lets a parity test and the bench line record WHICH kernel was compared with the oracle / timed (the X2V_GEMM_*CONTINUOUS
 * switches are folded in). */
int x2v_gemm_kernel_choice(int64_t M, int N, int K, int64_t ldx, int64_t ldw, int fp8);

/* Dense non-causal attention, head_dim 128: o[Sq, H*128] = softmax(q k^T * scale) v per head —
 * replaces FlashAttn2Weight/FlashAttn3Weight/TorchSDPAWeight.apply (common/ops/attn/attn_weight.py:71-126,
 * 209-239) for one sequence (cu_seqlens = [0, S]).  q/k/v: token stride in elements (ldq/ldk/ldv), head h
 * at column offset h*128; fp32 softmax, bf16 P, fp32 accumulate.  scale <= 0 selects 1/sqrt(128).
 * Sq == 0 (an empty query shard) is a no-op returning X2V_OK; Sk == 0 is X2V_E_SHAPE (softmax over nothing). */
int x2v_attn_fwd_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo, int64_t Sq,
                      int64_t Sk, int H, int head_dim, float scale, void* stream);

/* Same, selecting the lazy-rescale threshold of the online softmax (validation hook for the rare, data-dependent rescale branch: the three
 * must agree to rounding): 0 = default (= 6), 4 = rescale O on every tile, 5 = when a row max grew by more than 4 (base-2 units), 6 = by
 * more than 8.  X2V_ATTN_Q_PRESCALED is a flag of the Python layer's `variant` word (lib.attention); at the C-ABI the pre-scaled-q form is
 * X2V_ATTN_VT_PRESCALED in x2v_attn_fwd_bf16_vt's `flags`. */
#define X2V_ATTN_Q_PRESCALED 0x100
#define X2V_ATTN_LOG2E_SCALE(head_dim_rsqrt) ((head_dim_rsqrt) * 1.4426950408889634f)
int x2v_attn_fwd_bf16_variant(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo, int64_t Sq,
                              int64_t Sk, int H, int head_dim, float scale, int variant, void* stream);

/* V [Sk, H*128] (token stride ldv) -> V^T [H][ldvt/64][128][64] (per head and 64-key tile a contiguous [dv][key] block) with
 * keys >= Sk zero-filled (ldvt % 64 == 0, ldvt >= Sk): the operand layout of x2v_attn_fwd_bf16_vt. */
int x2v_transpose_heads_bf16(const void* v, int64_t ldv, void* vt, int64_t ldvt, int64_t Sk, int H, void* stream);

/* x2v_attn_fwd_bf16 on a pre-transposed V (x2v_transpose_heads_bf16) — the "ping-pong" kernel the fused block drivers launch for
 * self-attention (transformer_infer.py:369-379): V^T is staged by LDS-DMA and read as plain 16-byte fragments, the softmax scale * log2(e)
 * lives in q.  flags: X2V_ATTN_VT_PRESCALED — q already carries scale*log2(e) (x2v_rmsnorm_rope_scaled_bf16 / x2v_headnorm_rope_bf16 folded it
 * into q's one rounding; without it the kernel multiplies and re-rounds q itself); X2V_ATTN_VT_STAGGER — query block b starts its walk over the
 * key tiles (b mod 8) tiles in (the online softmax does not care where the walk starts; the L2 does: +1.3 % at Wan-14B 720p in 2-step runs, but
 * -0.9 % at sustained load, profiles/r04_call12_*: the fused drivers stopped setting it in round 4).  The result of a
 * query row then depends on which 256-row block of the launch it sits in (another fp32 summation order, same tolerance): callers that compare
 * bits across differently partitioned launches (the Ulysses driver) leave it off.  X2V_ATTN_VT_ONE_WALK — never the persistent short-walk form
 * (below): one workgroup per 256-row query block, as for long walks (A/B runs and the bit-equality test of the two forms).
 * Short walks — cross-attention over the text context, transformer_infer.py:424-455: 4..32 whole 64-key tiles and >= 512 query blocks x heads —
 * run as a persistent launch: a workgroup walks a contiguous range of the (sequence, head, query block) list as one tile stream, the next
 * block's q rows prefetched through LDS, the previous block's output stored under the next block's first tile.  Same bits as the one-walk form. */
#define X2V_ATTN_VT_PRESCALED 1
#define X2V_ATTN_VT_STAGGER 2
#define X2V_ATTN_VT_ONE_WALK 4
int x2v_attn_fwd_bf16_vt(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt, int64_t ldvt, void* o, int64_t ldo, int64_t Sq, int64_t Sk,
                         int H, int head_dim, float scale, int flags, void* stream);

/* Which launch form x2v_attn_fwd_bf16_vt / _batched takes for a shape (host-only, no GPU needed): bit 0 = the XCD-aware head-major work mapping
 * (on while the K / V^T streams of the <= 8 heads in flight fit the Infinity Cache and the launch has >= 512 workgroups — e.g. the Ulysses
 * rank's 5 heads x 75 600 keys; off for 40 heads at 720p), bit 8 = the staggered key walk (X2V_ATTN_VT_STAGGER and Sk >= 16 tiles), bit 9 = the
 * persistent short-walk form (then bits 0 and 8 are clear).  `flags` as x2v_attn_fwd_bf16_vt's.  Lets a parity
 * test assert that the kernel branch it compared with the oracle is the one a model's shapes take — the launches replace
 * attentions/distributed/ulysses/attn.py:68-80 (per rank) and transformer_infer.py:369-379 (single GPU).  Negative = X2V_E_SHAPE. */
int x2v_attn_vt_launch_plan(int64_t Sq, int64_t Sk, int H, int B, int flags);

/* x2v_attn_fwd_bf16_vt over B independent sequences in ONE launch: sequence b uses q / k / V^T / o at b * {q,k,vt,o}_bstride elements from the
 * base pointers (same Sq, Sk, H, strides).  The fused Wan driver runs the conditional and unconditional forwards of a CFG step
 * (models/networks/wan/model.py:197-226) as one pass over both token sets; this is their self-attention — 2 x 11 840 workgroups fill the 256 CUs'
 * last round better than two launches do.  V^T is one array [H][ldvt/64][128][64] over the stacked tokens of all sequences (each sequence padded to
 * a multiple of 64 rows), so vt_bstride = rows_per_sequence * 128. */
int x2v_attn_fwd_bf16_vt_batched(const void* q, int64_t ldq, int64_t q_bstride, const void* k, int64_t ldk, int64_t k_bstride, const void* vt, int64_t ldvt,
                                 int64_t vt_bstride, void* o, int64_t ldo, int64_t o_bstride, int64_t Sq, int64_t Sk, int H, int B, int head_dim, float scale,
                                 int flags, void* stream);

/* Per-token dynamic fp8 quantisation: s[m] = amax(|x[m,:]|)/448, xq = e4m3fn(x / s) — replaces
 * vllm ops.scaled_fp8_quant(use_per_token_if_dynamic=True) / sgl_kernel.sgl_per_token_quant_fp8
 * (mm_weight.py:236-245).  xq [M,K] bytes (ld = ldq), scale fp32 [M]. */
int x2v_quant_fp8_rowwise(const void* x, int64_t ldx, void* xq, int64_t ldq, float* scale, int64_t M, int K, void* stream);

/* x2v_quant_fp8_rowwise reading a K-BLOCKED x (element k of row m at x[(k / x_kblock) * x_kblock_stride + m * ldx + k % x_kblock]: the Ulysses
 * head->seq receive buffer [N_ranks][S/N][(H/N) d], attentions/distributed/ulysses/attn.py:82-91 after its transposing copy) and writing ROW-MAJOR
 * codes — the quantisation pass reads every element once anyway, so under w8a8 the de-blocking costs nothing and the output projection
 * (mm_weight.py:236-245 + :287-319) runs the plain row-major x2v_gemm_fp8.  x_kblock = 0: plain row-major x (= x2v_quant_fp8_rowwise). */
int x2v_quant_fp8_rowwise_blocked(const void* x, int64_t ldx, int x_kblock, int64_t x_kblock_stride, void* xq, int64_t ldq, float* scale, int64_t M, int K,
                                  void* stream);

/* x2v_layernorm_bf16 fused with x2v_quant_fp8_rowwise on its output: xq [M,D] e4m3 codes (ld = ldq) and sx [M] fp32 scales of the
 * normalised (+affine, +modulated) row, bit-identical to the two calls in sequence — the w8a8 path's LayerNorm -> scaled_fp8_quant in front
 * of the q/k/v and ffn_0 projections (transformer_infer.py:329-342,481-492 with mm_weight.py:236-245), without the bf16 round trip through
 * HBM and quantised once for every projection that reads it.  512 < D <= 16384. */
int x2v_layernorm_quant_fp8(const void* x, int64_t ldx, const void* w, const void* b, const void* scale, const void* shift, void* xq, int64_t ldq, float* sx,
                            int64_t M, int D, float eps, void* stream);

/* y[M,N] = epi((xq[M,K] . wq[N,K]^T) * sx[m] * sw[n] + bias[n]) → bf16 — replaces
 * torch.ops._C.cutlass_scaled_mm / sgl_kernel.fp8_scaled_mm (mm_weight.py:310-318,551-558,581-588).
 * e4m3fn operands, fp32 MFMA accumulate.  K % 128 == 0, N % 8 == 0. */
int x2v_gemm_fp8(const void* xq, int64_t ldx, const float* sx, const void* wq, int64_t ldw, const float* sw, const void* bias, void* y,
                 int64_t ldy, int64_t M, int N, int K, int epilogue, const void* resid, int64_t ldr, const void* gate, void* stream);

/* Same with the kernel selector of x2v_gemm_bf16_variant: 0 = by shape, 1 = 128x128 kernel, 2 = 256x256 ping-pong kernel (gemm256.hip, the large-shape
 * default), 5 = the continuous single-stream pipeline (gemm256c8.hip: gemm256c's structure on v_mfma_scale_f32_32x32x64_f8f6f4; needs K a multiple
 * of 256 and >= 512, N a multiple of 256, y blocks that are multiples of 128 columns and resid with y's row stride, else X2V_E_SHAPE; meant to give
 * variant 2's bits: 84 / 84 cases equal on MI355X, tools/gemm_fp8_continuous_check.py).  Variant 0 takes the continuous form where the shape
 * allows and the operands are row-major (X2V_GEMM_FP8_CONTINUOUS=0: never, =2: also for x2v_gemm_fp8_blocked's operands).  3 / 4 are bf16 only
 * (X2V_E_ARG). */
int x2v_gemm_fp8_variant(const void* xq, int64_t ldx, const float* sx, const void* wq, int64_t ldw, const float* sw, const void* bias, void* y,
                         int64_t ldy, int64_t M, int N, int K, int epilogue, const void* resid, int64_t ldr, const void* gate, int variant,
                         void* stream);

/* x2v_gemm_fp8 on block-strided operands (see x2v_gemm_bf16_blocked; x_kblock in e4m3 elements = bytes, a multiple of 128; sx stays [M]). */
int x2v_gemm_fp8_blocked(const void* xq, int64_t ldx, int x_kblock, int64_t x_kblock_stride, const float* sx, const void* wq, int64_t ldw, const float* sw,
                         const void* bias, void* y, int64_t ldy, int y_nblock, int64_t y_nblock_stride, int64_t M, int N, int K, int epilogue, const void* resid,
                         int64_t ldr, const void* gate, void* stream);

/* MXFP8 (OCP microscaling) activation / weight quantisation — replaces lightx2v_kernel.gemm.scaled_fp8_quant
 * (lightx2v_kernel/python/lightx2v_kernel/gemm.py:73-83, csrc/gemm/mxfp8_quant_kernels_sm120.cu:139-196): per 32 consecutive K
 * elements one e8m0 scale = the smallest power of two >= max|x|/448, elements = e4m3fn_rne(x / scale).
 * x bf16 [M,K]; q bytes [M,K] (ld = ldq); scales: bytes [K/128][M][4] — K-tile major, row, the 4 k blocks of that K-tile: the
 * layout x2v_gemm_mxfp8 consumes (as the reference's 128x4 swizzle is the sm120 tensor core's; logical element (m, kb) sits at
 * ((kb/4)*M + m)*4 + kb%4).  K % 128 == 0. */
int x2v_quant_mxfp8_bf16(const void* x, int64_t ldx, void* q, int64_t ldq, void* scales, int64_t M, int K, void* stream);

/* y[M,N] = alpha * (deq(a)[M,K] . deq(b)[N,K]^T) + bias[n] → bf16, deq = e4m3 element * 2^(scale byte - 127) of its row and
 * 32-wide k block — replaces lightx2v_kernel.gemm.cutlass_scaled_mxfp8_mm (gemm.py:93-97,
 * csrc/gemm/mxfp8_scaled_mm_kernels_sm120.cu:60-66,150-160).  sa [K/128][M][4], sb [K/128][N][4] as written by
 * x2v_quant_mxfp8_bf16; alpha: device pointer to one fp32 (NULL = 1), bias bf16 [N] or NULL.
 * Runs on v_mfma_scale_f32_32x32x64_f8f6f4 with the block scales applied by the matrix instruction.  K % 128 == 0, N % 8 == 0. */
int x2v_gemm_mxfp8(const void* a, int64_t lda, const void* sa, const void* b, int64_t ldb, const void* sb, const void* bias, const float* alpha, void* y,
                   int64_t ldy, int64_t M, int N, int K, void* stream);

/* x2v_gemm_mxfp8 with the fused epilogues of x2v_gemm_bf16 / x2v_gemm_fp8 (X2V_EPI_*; y = epi(alpha * a.b^T + bias)): what the fused
 * block drivers call for a linear layer held in MXFP8 (operator class MMWeightMxfp8Hip). */
int x2v_gemm_mxfp8_epi(const void* a, int64_t lda, const void* sa, const void* b, int64_t ldb, const void* sb, const void* bias, const float* alpha, void* y,
                       int64_t ldy, int64_t M, int N, int K, int epilogue, const void* resid, int64_t ldr, const void* gate, void* stream);

/* Same with a kernel selector: 0 = automatic (as x2v_gemm_mxfp8), 1 = 128x128-tile kernel, 2 = 256x256-tile ping-pong kernel. */
int x2v_gemm_mxfp8_variant(const void* a, int64_t lda, const void* sa, const void* b, int64_t ldb, const void* sb, const void* bias, const float* alpha,
                           void* y, int64_t ldy, int64_t M, int N, int K, int variant, void* stream);

/* Timestep sinusoid: y[n, dim] bf16 = [cos(t*f_j) | sin(t*f_j)], f_j = 10000^(-j/(dim/2)) computed in
 * float64 then rounded — replaces sinusoidal_embedding_1d (wan/infer/utils.py:161-172).  t: int64 [n]. */
int x2v_sinusoid_embed_bf16(const int64_t* t, void* y, int n, int dim, void* stream);

/* One denoise step's sampler update on the fp32 latent tensor as a single elementwise kernel — replaces the CFG combine
 * `noise_pred = uncond + guide * (cond - uncond)` (models/networks/wan/model.py:218) followed by WanScheduler.step_post
 * (models/schedulers/wan/scheduler.py:322-360: x0 = sample - sigma_i * noise_pred; UniPC-bh2 corrector :224-320 when order_c > 0;
 * predictor :130-222), solver order <= 2.  Every product / sum / quotient is rounded separately, in the reference's order (the result is
 * bit-identical to the reference's op sequence on CPU tensors).
 *   cond, uncond (NULL = no CFG): fp32 [n];  latents: the sample (bf16 if latents_bf16 — the reference's DTYPE=BF16 step_pre — else fp32);
 *   last_sample, m0, m1: the previous step's corrected sample and the x0 predictions of steps i-1, i-2 (fp32 [n]; NULL where unused);
 *   outputs (fp32 [n], must not alias the inputs): noise_pred (NULL = not wanted), x0_out (this step's x0 prediction), sample_out (the
 *   corrected sample = next step's last_sample), latents_out (the predictor's output = next latents);
 *   coef: HOST array of 12 floats {guide, sigma_i, c_a = sigma_t/sigma_s0, c_b = alpha_t*h_phi_1, c_c = alpha_t*B_h, c_rk = r_1,
 *   c_rho0 = rhos_c[0], c_rhol = rhos_c[-1], p_a, p_b, p_c, p_rk} (corrector c_*, predictor p_*), each computed as the reference
 *   computes it; order_c in {0 (no corrector), 1, 2}, order_p in {1, 2}. */
int x2v_unipc_step_f32(const float* cond, const float* uncond, const void* latents, int latents_bf16, const float* last_sample, const float* m0,
                       const float* m1, float* noise_pred, float* x0_out, float* sample_out, float* latents_out, const float* coef, int order_c, int order_p,
                       int64_t n, void* stream);

/* The 4-step-distilled scheduler's update (schedulers/wan/step_distill/scheduler.py:40-56) with the optional CFG combine in front:
 * x0 = latents - sigma * noise_pred;  noise != NULL (not the last step): x0 = one_minus_next * x0 + sigma_next * noise;
 * latents_out = x0 in the dtype of `latents` (bf16 if latents_bf16).  latents_out may alias latents. */
int x2v_distill_step_f32(const float* cond, const float* uncond, float guide, const void* latents, int latents_bf16, const float* noise, float sigma,
                         float one_minus_next, float sigma_next, float* noise_pred, void* latents_out, int64_t n, void* stream);

/* Causal Conv3d on channels-last fp32 activations — replaces CausalConv3d.forward
 * (models/video_encoders/hf/wan/vae.py:19-44; with kt = 1 also the decoder's Conv2d, vae.py:70-118).
 * Time axis input = [zeros(kt-1-cache_frames) | cache (cache_frames frames) | x (T frames)], output T frames;
 * zero 'same' padding spatially (kh/2, kw/2).  fp32 in/out, fp32-input MFMA (exact fp32 fma chain).
 * x:  [T, Hh, Ww, Cin]   cache: [cache_frames, Hh, Ww, Cin] or NULL when cache_frames == 0
 * w:  [Cout, kt, kh, kw, Cin]   bias: [Cout] or NULL   y: [T, Hh, Ww, Cout].  Cin % 4 == 0. */
int x2v_causal_conv3d_f32(const float* x, const float* cache, int cache_frames, const float* w, const float* bias, float* y, int T, int Hh, int Ww,
                          int Cin, int Cout, int kt, int kh, int kw, void* stream);

/* ---- Wan VAE decoder (models/video_encoders/hf/wan/vae.py), fp32, channels-last --------------------------------
 * Convolution inputs live in zero-bordered buffers [lead + T][H + 2*ph][W + 2*pw][C] whose `lead` = kt-1 leading
 * frames are the reference's per-conv feature cache (CACHE_T = 2, vae.py:16); see lightx2v_amd/csrc/vae.hip. */

#define X2V_VCONV_CLAMP 1  /* clamp the result to [-1, 1] (WanVAE.decode: .clamp_(-1, 1), vae.py:951-955) */
#define X2V_VCONV_TSPLIT 2 /* Cout = 2C: channel block j of frame t goes to frame 2t + j (Resample upsample3d, vae.py:136-138) */

/* y[t,h,w,co] = bias[co] + resid[t,h,w,co] + sum_{dt,dh,dw,c} xp[(t+dt)*fs + (h+dh)*rs + (w+dw)*ps + c] * w[co*wrs + ((dt*kh+dh)*kw+dw)*Cin + c]
 * — replaces CausalConv3d.forward (vae.py:19-44), the decoder's nn.Conv2d (vae.py:87-95) and, with kt=kh=kw=1, its
 * 1x1 convolutions and the attention block's GEMMs (vae.py:226-262).  `xp` points at the element tap (0,0,0) of output
 * pixel (0,0,0) reads (strides in floats; borders must be zero).  fp32 in/out on the fp32-input MFMA.
 * Cin % 16 == 0; strides % 4 == 0; bias/resid optional; resid has y's layout; y is [T,H,W,Cout] (or [2T,H,W,Cout/2]
 * with X2V_VCONV_TSPLIT). */
int x2v_vae_conv_f32(const float* xp, int64_t x_frame_stride, int64_t x_row_stride, int64_t x_px_stride, const float* w, int64_t w_row_stride,
                     const float* bias, const float* resid, float* y, int T, int Hh, int Ww, int Cin, int Cout, int kt, int kh, int kw, int flags,
                     void* stream);

/* Pixel-wise producer of conv input buffers — replaces RMS_norm (vae.py:47-59) + nn.SiLU (ResidualBlock, head),
 * Upsample nearest-exact x2 (vae.py:62-67) and the latent un-normalisation z / scale[1] + scale[0] (vae.py:716-719):
 *   v = x[t,h,w,:];  gamma != NULL: v = v / max(||v||_2, 1e-12) * sqrt(C) * gamma;  else v = v / a + b (a, b optional);
 *   silu: v *= sigmoid(v);  written to y + t*y_frame_stride + h*y_row_stride + w*C (2x2 replicated when upsample). */
int x2v_vae_prep_f32(const float* x, float* y, int T, int Hh, int Ww, int C, const float* gamma, const float* a, const float* b, int silu, int upsample,
                     int64_t y_frame_stride, int64_t y_row_stride, void* stream);

/* In-place s[M,N] = softmax(scale * s) rows, fp32 — the softmax of F.scaled_dot_product_attention in the VAE
 * AttentionBlock (vae.py:249-253).  N % 4 == 0. */
int x2v_softmax_rows_f32(float* s, int64_t ld, int64_t M, int N, float scale, void* stream);

/* ---- HunyuanVideo VAE decode (video_encoders/hf/autoencoder_kl_causal_3d/), fp32, channels-last ------------------- */

/* Pixel-wise producer: v = x*mul[c] + add[c] (mul/add optional; GroupNorm applied as a per-channel affine), optional SiLU
 * and clamp to [0,1]; nearest x2 upsampling in H,W (up_hw) and/or in T where frame 0 is not duplicated (up_t) — replaces
 * GroupNorm+SiLU in ResnetBlockCausal3D (unet_causal_3d_blocks.py:377-410), UpsampleCausal3D.forward (:168-187) and the
 * final (x/2+0.5).clamp(0,1) of VideoEncoderKLCausal3DModel.decode (model.py:41).  Output addressing as x2v_vae_prep_f32. */
int x2v_vae_prep_ex_f32(const float* x, float* y, int T, int Hh, int Ww, int C, const float* mul, const float* add, int silu, int clamp01, int up_hw, int up_t,
                        int64_t y_frame_stride, int64_t y_row_stride, void* stream);

/* 16-bit-operand form of x2v_vae_conv_f32 for the HunyuanVideo VAE, which the reference runs in fp16 (hunyuan_runner.py:40): xp and w
 * are fp16 (strides in halves, Cin % 64 == 0), accumulation fp32 on v_mfma_f32_32x32x16_f16, bias / residual / output fp32 — the residual
 * stream and the normalisation statistics keep fp32, only the convolution operands are rounded.  Same flags and layouts as the fp32 entry, plus
 * 4 = the per-tap kernel and 8 = the 64-pixel halo kernel (kernel choice for A/B runs and tests: by default 3x3 kernels with Cout % 96 == 0 take the
 * 128-pixel x 96-cout kernel on v_mfma_f32_16x16x32_f16, whose reduction order differs in rounding) and 16 = the last 32 channels of Cin are zero
 * padding in both operands (skipped where the kernel steps in 32 channels; a no-op for the results). */
int x2v_vae_conv_f16(const void* xp, int64_t x_frame_stride, int64_t x_row_stride, int64_t x_px_stride, const void* w, int64_t w_row_stride, const float* bias,
                     const float* resid, float* y, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw, int flags, void* stream);

/* x2v_vae_conv_f16 with the causal convolution's feature cache (the reference's feat_cache entry, vae.py:16,199-214: the last kt - 1 input frames of the
 * previous chunk) in a buffer of its own: input frames 0 .. kt-2 are read from `cache` ([kt-1][H+2][W+2][Cin], xp's strides), the rest from xp, whose own
 * leading kt - 1 frames are not read — a frame buffer shared by several convolutions then needs no copy of the cache in front of it.  Only where
 * x2v_vae_conv_f16_cached_ok(...) == 1 (3x3 kernels with Cout % 96 == 0, Cout % 128 == 0 or Cout <= 16: the 128-pixel kernel); bit-identical to x2v_vae_conv_f16 on a buffer
 * that carries the same frames. */
int x2v_vae_conv_f16_cached(const void* xp, const void* cache, int64_t x_frame_stride, int64_t x_row_stride, int64_t x_px_stride, const void* w, int64_t w_row_stride,
                            const float* bias, const float* resid, float* y, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw, int flags, void* stream);
int x2v_vae_conv_f16_cached_ok(int W, int Cin, int Cout, int kh, int kw, int flags);

/* x2v_vae_prep_f32 writing fp16 with an explicit pixel stride y_px_stride >= C (the Wan decoder's 96-channel stage pads its operand
 * buffers to 128 channels for x2v_vae_conv_f16's 64-channel K steps; pad channels are never written and stay zero). */
int x2v_vae_prep_f16(const float* x, void* y, int T, int H, int W, int C, const float* gamma, const float* a, const float* b, int silu, int upsample,
                     int64_t y_frame_stride, int64_t y_row_stride, int64_t y_px_stride, void* stream);

/* x2v_vae_prep_f16 writing the hi/lo fp16 split of its result (x = hi + lo, hi = fp16(x), lo = fp16(x - hi): ~22 mantissa bits) as channels
 * [hi | hi * 2^-12 | lo] (3*C halves per pixel, y_px_stride >= 3*C; hi saturates at the fp16 range instead of overflowing).  With weights laid out
 * [hi | lo * 2^12 | hi] along Cin (the power-of-two pair keeps the weights' lo halves normal fp16 numbers), x2v_vae_conv_f16 accumulates
 * xh.wh + xh.wl + xl.wh in fp32: the Wan VAE's fp32 convolutions (vae.py:794) at fp32-grade accuracy on the 16-bit matrix instruction. */
int x2v_vae_prep_split_f16(const float* x, void* y, int T, int Hh, int Ww, int C, const float* gamma, const float* a, const float* b, int silu, int upsample,
                           int64_t y_frame_stride, int64_t y_row_stride, int64_t y_px_stride, void* stream);

/* x2v_vae_prep_ex_f32 writing fp16: fills the operand buffer of x2v_vae_conv_f16 (y strides in halves, C % 8 == 0). */
int x2v_vae_prep_ex_f16(const float* x, void* y, int T, int H, int W, int C, const float* mul, const float* add, int silu, int clamp01, int up_hw, int up_t,
                        int64_t y_frame_stride, int64_t y_row_stride, void* stream);

/* Fill the borders of a conv input buffer [frames][Hp][Wp][C] by replication: spatial borders (width `pad`) from the
 * nearest interior pixel, the `lead` leading frames from frame `lead` — F.pad(mode="replicate") of CausalConv3d
 * (unet_causal_3d_blocks.py:84-91). */
int x2v_vae_replicate_border_f32(float* buf, int frames, int lead, int Hp, int Wp, int C, int pad, void* stream);

/* GroupNorm over x [npix, C] (G groups of consecutive channels, statistics over all pixels) reduced to a per-channel
 * affine: mul[c] = rstd_g*gamma[c], add[c] = beta[c] - mean_g*mul[c] (apply with x2v_vae_prep_ex_f32) — replaces
 * torch.nn.GroupNorm (unet_causal_3d_blocks.py:318,341; vae.py conv_norm_out; the attention block's group_norm).
 * workspace: 2*G doubles (device).  fp64 accumulation. */
int x2v_groupnorm_affine_f32(const float* x, int64_t npix, int C, int G, const float* gamma, const float* beta, float eps, double* workspace, float* mul,
                             float* add, void* stream);

/* In-place s[M,N] = softmax(scale*s) over the frame-causal prefix: row i sees keys j < min(n_keys, (i/hw + 1)*hw), the
 * rest (incl. padding columns n_keys..N) becomes 0 (prepare_causal_attention_mask, unet_causal_3d_blocks.py:48-63, in
 * UNetMidBlockCausal3D.forward :629-634). */
int x2v_softmax_rows_causal_f32(float* s, int64_t ld, int64_t M, int N, float scale, int hw, int n_keys, void* stream);

/* Linear cross-fade of overlapping tiles along one axis, tensors viewed as [outer][axis][inner]:
 * b[idx] = a[na-extent+idx]*(1-idx/extent) + b[idx]*(idx/extent), idx < extent — blend_v / blend_h / blend_t
 * (autoencoder_kl_causal_3d.py:347-364). */
int x2v_blend_axis_f32(const float* a, float* b, int64_t outer, int na, int nb, int64_t inner, int64_t a_outer_stride, int64_t b_outer_stride, int extent,
                       void* stream);

/* Box calibration (measurement plumbing; no reference counterpart): runs bare v_mfma_f32_16x16x32_bf16 loops (operands in registers,
 * 8 waves per CU, every CU) for `milliseconds` on `stream` and returns in *tflops what the board sustained over the second half of that time.
 * bench.py calls it before and after its timed region so that a roofline fraction measured on one box can be compared with another's
 * (boxes of one pool differ by several percent under the 1400 W board limit).  Synchronises the stream. */
int x2v_mfma_probe_bf16(int milliseconds, float* tflops, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* X2V_H */
