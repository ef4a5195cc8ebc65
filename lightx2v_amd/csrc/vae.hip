// Wan VAE decoder kernels (fp32, channels-last): implicit-GEMM convolution over a zero-bordered input buffer on the
// fp32-input MFMA, the pixel-wise RMS_norm + SiLU (+ nearest 2x upsample, + per-channel affine) producer that writes
// such buffers, and the row softmax of the decoder's single-head attention block.
//
// reference: models/video_encoders/hf/wan/vae.py — CausalConv3d :19-44, RMS_norm :47-59, Upsample :62-67,
// Resample :70-159, ResidualBlock :185-223, AttentionBlock :226-262, Decoder3d :377-489, WanVAE_.decode :713-738.
// The reference decodes in fp32 (vae.py:794); v_mfma_f32_32x32x2_f32 is an exact fp32 fma chain at the fp32 vector
// rate (157 TFLOP/s dense peak), so the reduction runs on the matrix pipe without changing the numerics class.
//
// Data layout (MI355X-first, 288 GB): every convolution input lives in its own persistent buffer
//   [lead + T frames][H + 2*ph][W + 2*pw][C]  fp32, channels-last, borders zero,
// where the `lead` = kt-1 leading frames ARE the reference's per-conv feature cache (CACHE_T = 2 frames, vae.py:16;
// the cache update rule "last two frames of [cache | x]" is a 2-frame move inside the buffer).  With the halo
// materialised, the input address of (output pixel, tap, channel) is  pixel_base + tap_offset + channel: the first
// term is a loop-invariant per-lane VGPR, the rest is wave-uniform and travels in the buffer-load soffset — the K
// loop issues LDS-DMA with zero address arithmetic and needs no border predicates.
//
// conv kernel: workgroup = 4 waves = 256 output pixels (consecutive in h*W+w order of one frame) x 32*NF output
// channels; wave = 64 pixels x 32*NF channels = 2 x NF MFMA tiles (acc 32*NF VGPRs); K loop over (tap, KC-channel
// slab): A slab [256 px][KC] and B slab [32*NF couts][KC] staged by LDS-DMA, double buffered, 16-byte chunks
// XOR-swizzled on the source offset and on the ds_read_b128 address (same involution as gemm256.hip).  One
// ds_read_b128 feeds 4 MFMAs (lane (fl, fh) holds k = 8j + 4fh + e for MFMA e), so LDS traffic is negligible; a
// K step carries 32*NF fp32 MFMAs of 64 cycles per wave, which hides the next slab's DMA behind one barrier.
// Bound: MFMA fp32 (157 TFLOP/s); algorithmic work 2 * T*H*W * Cout * Cin * taps FLOP per launch.
#include <algorithm>

#include "x2v_common.h"

namespace x2v {

typedef __attribute__((address_space(3))) void* v_lds_ptr_t;

constexpr int VC_PIX = 256;
constexpr unsigned VC_OOB = 0x80000000u;  // voffset of a masked row: beyond any descriptor range we build (< 2 GiB)

// flags of x2v_vae_conv_f32
constexpr int VCF_CLAMP = 1;   // clamp the result to [-1, 1] (WanVAE.decode, vae.py:951-955)
constexpr int VCF_TSPLIT = 2;  // Cout = 2C: channel block j of frame t -> frame 2t + j (Resample upsample3d, vae.py:136-138)

template <int NF, int KC>
__global__ __launch_bounds__(256, 2) void vae_conv_kernel(const float* __restrict__ xp, int64_t x_frame_stride, int64_t x_row_stride, int64_t x_px_stride,
                                                       const float* __restrict__ w, int64_t w_row_stride, const float* __restrict__ bias,
                                                       const float* __restrict__ resid, float* __restrict__ y, int T, int Hh, int Ww, int Cin, int Cout,
                                                       int kt, int kh, int kw, int flags, int ncol) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ROWB = KC * 4;                   // bytes per staged row
  constexpr int CPR = KC / 4;                    // 16-byte chunks per row
  constexpr int RPI = 64 / CPR;                  // rows per wave-instruction
  constexpr int BN = 32 * NF;
  constexpr int A_BYTES = VC_PIX * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
  constexpr int A_INSTR = 64 / RPI;              // DMA instructions per wave for its 64 A rows
  constexpr int B_INSTR = (BN / 4 + RPI - 1) / RPI;  // ... for its BN/4 B rows (rounded up; extra rows masked)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fl = lane & 31, fh = lane >> 5;
  const int HW = Hh * Ww;
  const int tiles_per_frame = (HW + VC_PIX - 1) / VC_PIX;
  const unsigned v = xcd_remap(blockIdx.x, gridDim.x);
  const int ptile = (int)(v / (unsigned)ncol), ctile = (int)(v % (unsigned)ncol);
  const int frame = ptile / tiles_per_frame;
  const int p0 = (ptile % tiles_per_frame) * VC_PIX;
  const int co0 = ctile * BN;
  const int taps = kt * kh * kw;

  // descriptors: input = kt frames starting at this output frame; weights = rows co0 .. Cout
  const int64_t fbytes = x_frame_stride * 4;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(xp + (int64_t)frame * x_frame_stride), 0, (unsigned)(fbytes * kt), 0x00020000);
  const int wrows = min(BN, Cout - co0);
  const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(w + (int64_t)co0 * w_row_stride), 0, (unsigned)(((int64_t)(wrows - 1) * w_row_stride + (int64_t)taps * Cin) * 4), 0x00020000);

  // per-lane DMA source offsets (bytes): A rows = this wave's 64 pixels, B rows = this wave's quarter of the couts
  constexpr int swz_shift = (KC == 32) ? 1 : 2;  // rows sharing one 256-byte bank row
  unsigned a_voff[A_INSTR], b_voff[B_INSTR];
#pragma unroll
  for (int i = 0; i < A_INSTR; ++i) {
    const int r = wid * 64 + i * RPI + lane / CPR;  // tile row = pixel
    const int c = (lane % CPR) ^ ((r >> swz_shift) & (CPR - 1));
    const int p = p0 + r;
    const int ph = p / Ww, pw = p - ph * Ww;
    a_voff[i] = p < HW ? (unsigned)(((int64_t)ph * x_row_stride + (int64_t)pw * x_px_stride) * 4) + (unsigned)(c << 4) : VC_OOB;
  }
#pragma unroll
  for (int i = 0; i < B_INSTR; ++i) {
    const int rl = i * RPI + lane / CPR;            // row within this wave's share
    const int r = wid * (BN / 4) + rl;
    const int c = (lane % CPR) ^ ((r >> swz_shift) & (CPR - 1));
    b_voff[i] = (rl < BN / 4 && r < wrows) ? (unsigned)((int64_t)r * w_row_stride * 4) + (unsigned)(c << 4) : VC_OOB;
  }
  const int kchunks = Cin / KC;
  const int nsteps = taps * kchunks;
  auto stage = [&](int s, int step) {
    const int tap = step / kchunks, kc = step - tap * kchunks;
    const int dt = tap / (kh * kw), dh = (tap / kw) % kh, dw = tap % kw;
    const unsigned xso = (unsigned)(((int64_t)dt * x_frame_stride + (int64_t)dh * x_row_stride + (int64_t)dw * x_px_stride + (int64_t)kc * KC) * 4);
    const unsigned wso = (unsigned)(((int64_t)tap * Cin + (int64_t)kc * KC) * 4);
    char* as = smem + s * STAGE + wid * (64 * ROWB);
    char* bs = smem + s * STAGE + A_BYTES + wid * ((BN / 4) * ROWB);
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (v_lds_ptr_t)(as + i * 1024), 16, a_voff[i], xso, 0, 0);
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i)
      if ((i + 1) * RPI <= BN / 4 || lane / CPR + i * RPI < BN / 4)  // the last instruction may cover fewer rows
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rwt, (v_lds_ptr_t)(bs + i * 1024), 16, b_voff[i], wso, 0, 0);
  };

  // fragment read offsets: row (32-row block + fl), chunk (j*2 + fh) ^ swizzle
  int rd[KC / 8];
#pragma unroll
  for (int j = 0; j < KC / 8; ++j) rd[j] = fl * ROWB + ((((j << 1) | fh) ^ ((fl >> swz_shift) & (CPR - 1))) << 4);

  f32x16_t acc[2][NF];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int step = 0; step < nsteps; ++step) {
    const int cur = step & 1;
    if (step + 1 < nsteps) stage(cur ^ 1, step + 1);
    const char* ab = smem + cur * STAGE + wid * (64 * ROWB);
    const char* bb = smem + cur * STAGE + A_BYTES;
#pragma unroll
    for (int j = 0; j < KC / 8; ++j) {
      f32x4_t xa[2], wb[NF];
#pragma unroll
      for (int i = 0; i < 2; ++i) xa[i] = *reinterpret_cast<const f32x4_t*>(ab + i * 32 * ROWB + rd[j]);
#pragma unroll
      for (int n = 0; n < NF; ++n) wb[n] = *reinterpret_cast<const f32x4_t*>(bb + n * 32 * ROWB + rd[j]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int n = 0; n < NF; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[n][e], xa[i][e], acc[i][n], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue.  acc[i][n][r]: pixel p0 + wid*64 + i*32 + fl, cout co0 + n*32 + (r&3) + 8*(r>>2) + 4*fh
  const bool vec_ok = (Cout & 3) == 0;
  const int csplit = (flags & VCF_TSPLIT) ? Cout / 2 : Cout;  // channels per output pixel
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = p0 + wid * 64 + i * 32 + fl;
    if (p >= HW) continue;
#pragma unroll
    for (int n = 0; n < NF; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = co0 + n * 32 + 8 * g + 4 * fh;
        if (co >= Cout) continue;
        float vv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) vv[e] = acc[i][n][4 * g + e];
        int64_t oidx;
        if (flags & VCF_TSPLIT) {
          const int hi = co >= csplit ? 1 : 0;
          oidx = ((int64_t)(2 * frame + hi) * HW + p) * csplit + (co - hi * csplit);
        } else {
          oidx = ((int64_t)frame * HW + p) * Cout + co;
        }
        if (vec_ok) {
          if (bias != nullptr) {
            const float4 b4 = *reinterpret_cast<const float4*>(bias + co);
            vv[0] += b4.x; vv[1] += b4.y; vv[2] += b4.z; vv[3] += b4.w;
          }
          if (resid != nullptr) {
            const float4 r4 = *reinterpret_cast<const float4*>(resid + oidx);
            vv[0] += r4.x; vv[1] += r4.y; vv[2] += r4.z; vv[3] += r4.w;
          }
          if (flags & VCF_CLAMP) {
#pragma unroll
            for (int e = 0; e < 4; ++e) vv[e] = fminf(fmaxf(vv[e], -1.f), 1.f);
          }
          *reinterpret_cast<float4*>(y + oidx) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (co + e < Cout) {
              float o = vv[e] + (bias != nullptr ? bias[co + e] : 0.f) + (resid != nullptr ? resid[oidx + e] : 0.f);
              if (flags & VCF_CLAMP) o = fminf(fmaxf(o, -1.f), 1.f);
              y[oidx + e] = o;
            }
        }
      }
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// 16-bit-input variant of the implicit-GEMM convolution: activations and weights in fp16 (the precision the reference runs the
// HunyuanVideo VAE in, hunyuan_runner.py:40), fp32 accumulate on v_mfma_f32_32x32x16_f16, fp32 bias / residual / output — the residual
// stream and the GroupNorm statistics stay fp32, only the operands of the big convolutions are rounded.  Same tile (256 pixels x 32 NF
// couts per workgroup, wave = 64 px), same LDS-DMA staging and swizzle as the fp32 kernel; a K step is one tap x 64 channels
// (128-byte rows) = 4 k-steps of 16: 8 NF MFMAs of 32 cycles per wave and step against 12 + NF LDS-DMA pieces per workgroup —
// MFMA-bound no longer, the staging path sets the pace (measured in DESIGN.md §4.4).
typedef _Float16 vc_half8_t __attribute__((ext_vector_type(8)));

template <int NF>
__global__ __launch_bounds__(256, 2) void vae_conv16_kernel(const _Float16* __restrict__ xp, int64_t x_frame_stride, int64_t x_row_stride, int64_t x_px_stride,
                                                         const _Float16* __restrict__ w, int64_t w_row_stride, const float* __restrict__ bias,
                                                         const float* __restrict__ resid, float* __restrict__ y, int T, int Hh, int Ww, int Cin, int Cout,
                                                         int kt, int kh, int kw, int flags, int ncol) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KC = 64;                         // channels per K step
  constexpr int ROWB = KC * 2;                   // bytes per staged row
  constexpr int CPR = 8, RPI = 8;                // 16-byte chunks per row, rows per wave-instruction
  constexpr int BN = 32 * NF;
  constexpr int A_BYTES = VC_PIX * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
  constexpr int A_INSTR = 64 / RPI;
  constexpr int B_INSTR = (BN / 4 + RPI - 1) / RPI;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fl = lane & 31, fh = lane >> 5;
  const int HW = Hh * Ww;
  const int tiles_per_frame = (HW + VC_PIX - 1) / VC_PIX;
  const unsigned v = xcd_remap(blockIdx.x, gridDim.x);
  const int ptile = (int)(v / (unsigned)ncol), ctile = (int)(v % (unsigned)ncol);
  const int frame = ptile / tiles_per_frame;
  const int p0 = (ptile % tiles_per_frame) * VC_PIX;
  const int co0 = ctile * BN;
  const int taps = kt * kh * kw;

  const int64_t fbytes = x_frame_stride * 2;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(xp + (int64_t)frame * x_frame_stride), 0, (unsigned)(fbytes * kt), 0x00020000);
  const int wrows = min(BN, Cout - co0);
  const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(w + (int64_t)co0 * w_row_stride), 0, (unsigned)(((int64_t)(wrows - 1) * w_row_stride + (int64_t)taps * Cin) * 2), 0x00020000);

  unsigned a_voff[A_INSTR], b_voff[B_INSTR];
#pragma unroll
  for (int i = 0; i < A_INSTR; ++i) {
    const int r = wid * 64 + i * RPI + lane / CPR;
    const int c = (lane % CPR) ^ ((r >> 1) & (CPR - 1));
    const int p = p0 + r;
    const int ph = p / Ww, pw = p - ph * Ww;
    a_voff[i] = p < HW ? (unsigned)(((int64_t)ph * x_row_stride + (int64_t)pw * x_px_stride) * 2) + (unsigned)(c << 4) : VC_OOB;
  }
#pragma unroll
  for (int i = 0; i < B_INSTR; ++i) {
    const int rl = i * RPI + lane / CPR;
    const int r = wid * (BN / 4) + rl;
    const int c = (lane % CPR) ^ ((r >> 1) & (CPR - 1));
    b_voff[i] = (rl < BN / 4 && r < wrows) ? (unsigned)((int64_t)r * w_row_stride * 2) + (unsigned)(c << 4) : VC_OOB;
  }
  const int kchunks = Cin / KC;
  const int nsteps = taps * kchunks;
  auto stage = [&](int s, int step) {
    const int tap = step / kchunks, kc = step - tap * kchunks;
    const int dt = tap / (kh * kw), dh = (tap / kw) % kh, dw = tap % kw;
    const unsigned xso = (unsigned)(((int64_t)dt * x_frame_stride + (int64_t)dh * x_row_stride + (int64_t)dw * x_px_stride + (int64_t)kc * KC) * 2);
    const unsigned wso = (unsigned)(((int64_t)tap * Cin + (int64_t)kc * KC) * 2);
    char* as = smem + s * STAGE + wid * (64 * ROWB);
    char* bs = smem + s * STAGE + A_BYTES + wid * ((BN / 4) * ROWB);
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (v_lds_ptr_t)(as + i * 1024), 16, a_voff[i], xso, 0, 0);
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i)
      if ((i + 1) * RPI <= BN / 4 || lane / CPR + i * RPI < BN / 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rwt, (v_lds_ptr_t)(bs + i * 1024), 16, b_voff[i], wso, 0, 0);
  };

  // fragment read offsets: row (32-row block + fl), k-step ks reads chunk (ks*2 + fh) ^ swizzle (8 halves = the lane's k values)
  int rd[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) rd[ks] = fl * ROWB + ((((ks << 1) | fh) ^ ((fl >> 1) & (CPR - 1))) << 4);

  f32x16_t acc[2][NF];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int step = 0; step < nsteps; ++step) {
    const int cur = step & 1;
    if (step + 1 < nsteps) stage(cur ^ 1, step + 1);
    const char* ab = smem + cur * STAGE + wid * (64 * ROWB);
    const char* bb = smem + cur * STAGE + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      vc_half8_t xa[2], wb[NF];
#pragma unroll
      for (int i = 0; i < 2; ++i) xa[i] = *reinterpret_cast<const vc_half8_t*>(ab + i * 32 * ROWB + rd[ks]);
#pragma unroll
      for (int n = 0; n < NF; ++n) wb[n] = *reinterpret_cast<const vc_half8_t*>(bb + n * 32 * ROWB + rd[ks]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < NF; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[n], xa[i], acc[i][n], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // epilogue: identical to the fp32 kernel (fp32 bias / residual / output)
  const bool vec_ok = (Cout & 3) == 0;
  const int csplit = (flags & VCF_TSPLIT) ? Cout / 2 : Cout;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = p0 + wid * 64 + i * 32 + fl;
    if (p >= HW) continue;
#pragma unroll
    for (int n = 0; n < NF; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = co0 + n * 32 + 8 * g + 4 * fh;
        if (co >= Cout) continue;
        float vv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) vv[e] = acc[i][n][4 * g + e];
        int64_t oidx;
        if (flags & VCF_TSPLIT) {
          const int hi = co >= csplit ? 1 : 0;
          oidx = ((int64_t)(2 * frame + hi) * HW + p) * csplit + (co - hi * csplit);
        } else {
          oidx = ((int64_t)frame * HW + p) * Cout + co;
        }
        if (vec_ok) {
          if (bias != nullptr) {
            const float4 b4 = *reinterpret_cast<const float4*>(bias + co);
            vv[0] += b4.x; vv[1] += b4.y; vv[2] += b4.z; vv[3] += b4.w;
          }
          if (resid != nullptr) {
            const float4 r4 = *reinterpret_cast<const float4*>(resid + oidx);
            vv[0] += r4.x; vv[1] += r4.y; vv[2] += r4.z; vv[3] += r4.w;
          }
          if (flags & VCF_CLAMP) {
#pragma unroll
            for (int e = 0; e < 4; ++e) vv[e] = fminf(fmaxf(vv[e], -1.f), 1.f);
          }
          *reinterpret_cast<float4*>(y + oidx) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (co + e < Cout) {
              float o = vv[e] + (bias != nullptr ? bias[co + e] : 0.f) + (resid != nullptr ? resid[oidx + e] : 0.f);
              if (flags & VCF_CLAMP) o = fminf(fmaxf(o, -1.f), 1.f);
              y[oidx + e] = o;
            }
        }
      }
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// Pixel-wise producer of convolution input buffers:
//   v = x[t,h,w,:];  norm: v = v / max(||v||_2, 1e-12) * sqrt(C) * gamma   (RMS_norm, vae.py:47-59: F.normalize * scale * gamma)
//   else affine: v = v / a + b (per channel, either may be NULL)                 (z un-normalisation, vae.py:716-719)
//   silu: v = v * sigmoid(v)                                                     (nn.SiLU in ResidualBlock / head)
//   up: the result is written to the 2x2 output pixels (2h+{0,1}, 2w+{0,1})       (Upsample nearest-exact x2, vae.py:62-67,87-95)
// Output addressing: y + t*y_frame_stride + h*y_row_stride + w*C (the caller passes y already offset to the
// interior origin of a zero-bordered buffer).  One group of LPP lanes per pixel, 16-byte accesses.
template <int LPP, typename OT = float>
__global__ __launch_bounds__(256) void vae_prep_kernel(const float* __restrict__ x, OT* __restrict__ y, int64_t npix, int Hh, int Ww, int C,
                                                       const float* __restrict__ gamma, const float* __restrict__ a, const float* __restrict__ b,
                                                       int silu, int up, int64_t y_frame_stride, int64_t y_row_stride, int64_t y_px_stride, int split = 0) {
  constexpr int GPB = 256 / LPP;  // pixel groups per block
  const int g = threadIdx.x / LPP, l = threadIdx.x % LPP;
  const int nch = C / 4;
  const float rms_scale = sqrtf((float)C);
  for (int64_t pix = (int64_t)blockIdx.x * GPB + g; pix < npix; pix += (int64_t)gridDim.x * GPB) {
    const float* xr = x + pix * C;
    float4 v[4];
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c4 = l + k * LPP;
      v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c4 < nch) {
        v[k] = *reinterpret_cast<const float4*>(xr + c4 * 4);
        ss += v[k].x * v[k].x + v[k].y * v[k].y + v[k].z * v[k].z + v[k].w * v[k].w;
      }
    }
    float inv = 1.f;
    if (gamma != nullptr) {
#pragma unroll
      for (int o = LPP / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
      inv = rms_scale / fmaxf(sqrtf(ss), 1e-12f);
    }
    const int64_t t = pix / ((int64_t)Hh * Ww);
    const int rem = (int)(pix - t * (int64_t)Hh * Ww);
    const int h = rem / Ww, wq = rem - h * Ww;
    // y_px_stride >= C: the fp16 operand buffers pad the channel axis to a multiple of 64 (pad channels stay zero)
    OT* yb = y + t * y_frame_stride + (int64_t)(up ? 2 * h : h) * y_row_stride + (int64_t)(up ? 2 * wq : wq) * y_px_stride;
    struct alignas(sizeof(OT) * 4) Out4 { OT e[4]; };
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c4 = l + k * LPP;
      if (c4 >= nch) continue;
      float o[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
      if (gamma != nullptr) {
        const float4 gm = *reinterpret_cast<const float4*>(gamma + c4 * 4);
        o[0] = o[0] * inv * gm.x; o[1] = o[1] * inv * gm.y; o[2] = o[2] * inv * gm.z; o[3] = o[3] * inv * gm.w;
      } else {
        if (a != nullptr) {
          const float4 av = *reinterpret_cast<const float4*>(a + c4 * 4);
          o[0] /= av.x; o[1] /= av.y; o[2] /= av.z; o[3] /= av.w;
        }
        if (b != nullptr) {
          const float4 bv = *reinterpret_cast<const float4*>(b + c4 * 4);
          o[0] += bv.x; o[1] += bv.y; o[2] += bv.z; o[3] += bv.w;
        }
      }
      if (silu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = o[e] / (1.f + __expf(-o[e]));
      }
      if constexpr (sizeof(OT) == 2) {
        if (split) {  // hi = fp16(x) must stay finite: beyond the fp16 range hi saturates at 65504 and lo carries the remainder at fp16 precision (2^-12 relative instead of inf)
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = fminf(fmaxf(o[e], -131008.f), 131008.f);
        }
      }
      Out4 ov = {{(OT)o[0], (OT)o[1], (OT)o[2], (OT)o[3]}};
      if constexpr (sizeof(OT) == 2) {
        if (split) {
#pragma unroll
          for (int e = 0; e < 4; ++e) ov.e[e] = (OT)fminf(fmaxf(o[e], -65504.f), 65504.f);
        }
      }
      *reinterpret_cast<Out4*>(yb + c4 * 4) = ov;
      if (up) {
        *reinterpret_cast<Out4*>(yb + y_px_stride + c4 * 4) = ov;
        *reinterpret_cast<Out4*>(yb + y_row_stride + c4 * 4) = ov;
        *reinterpret_cast<Out4*>(yb + y_row_stride + y_px_stride + c4 * 4) = ov;
      }
      if constexpr (sizeof(OT) == 2) {
        if (split) {
          // hi/lo operand split (x = hi + lo to ~22 mantissa bits): channels [hi | hi * 2^-12 | lo], to meet weights laid out
          // [hi | lo * 2^12 | hi] — the 16-bit convolution then accumulates xh.wh + xh.wl + xl.wh in fp32 (the xl.wl term is below fp32
          // resolution).  The power-of-two pair on the middle plane keeps the weights' lo halves NORMAL fp16 numbers (|w_lo| <= 2^-12 |w|
          // would be subnormal for every |w| < 1/4, i.e. nearly all of a conv kernel: ~17 instead of 22 bits, and dependent on the matrix
          // unit not flushing subnormals); x_hi * 2^-12 only has to carry the 11 bits a 2^-11-sized correction term needs.
          const Out4 lv = {{(OT)(o[0] - (float)ov.e[0]), (OT)(o[1] - (float)ov.e[1]), (OT)(o[2] - (float)ov.e[2]), (OT)(o[3] - (float)ov.e[3])}};
          const Out4 mv = {{(OT)((float)ov.e[0] * 0.000244140625f), (OT)((float)ov.e[1] * 0.000244140625f), (OT)((float)ov.e[2] * 0.000244140625f),
                            (OT)((float)ov.e[3] * 0.000244140625f)}};
#pragma unroll
          for (int q = 0; q < (up ? 4 : 1); ++q) {
            OT* yq = yb + (q & 1) * y_px_stride + (q >> 1) * y_row_stride;
            *reinterpret_cast<Out4*>(yq + C + c4 * 4) = mv;
            *reinterpret_cast<Out4*>(yq + 2 * C + c4 * 4) = lv;
          }
        }
      }
    }
  }
}

// In-place row softmax of s[M][N] fp32 with a pre-scale: s = softmax(scale * s) — the attention block's
// F.scaled_dot_product_attention (vae.py:249-253) split as GEMM / softmax / GEMM.  One 256-thread block per row.
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ s, int64_t lds_, int64_t M, int N, float scale) {
  __shared__ float red[4];
  float* row = s + (int64_t)blockIdx.x * lds_;
  float mx = -3.0e38f;
  for (int i = threadIdx.x * 4; i < N; i += 1024) {
    const float4 v = *reinterpret_cast<const float4*>(row + i);
    mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
  }
  mx = block_max<4>(mx, red) * scale;
  float sum = 0.f;
  for (int i = threadIdx.x * 4; i < N; i += 1024) {
    float4 v = *reinterpret_cast<const float4*>(row + i);
    v.x = __expf(v.x * scale - mx); v.y = __expf(v.y * scale - mx); v.z = __expf(v.z * scale - mx); v.w = __expf(v.w * scale - mx);
    sum += (v.x + v.y) + (v.z + v.w);
    *reinterpret_cast<float4*>(row + i) = v;
  }
  sum = block_sum<4>(sum, red);
  const float inv = 1.f / sum;
  for (int i = threadIdx.x * 4; i < N; i += 1024) {
    float4 v = *reinterpret_cast<const float4*>(row + i);
    v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
    *reinterpret_cast<float4*>(row + i) = v;
  }
}

// ------------------------------------------------------------------------------------------------
// HunyuanVideo VAE (AutoencoderKLCausal3D) helpers — reference: video_encoders/hf/autoencoder_kl_causal_3d/
// unet_causal_3d_blocks.py (CausalConv3d replicate padding :65-91, UpsampleCausal3D :94-197, ResnetBlockCausal3D
// :261-420, UNetMidBlockCausal3D :526-640) and autoencoder_kl_causal_3d.py (tiled decode + blends :347-518).

// General pixel-wise producer (superset of vae_prep_kernel's affine path): v = x*mul[c] + add[c] (GroupNorm applied as a
// per-channel affine), optional SiLU, optional clamp to [0,1], nearest upsampling x2 in H,W and/or in T where the FIRST
// frame is not duplicated (UpsampleCausal3D.forward :168-187: frame 0 -> 0, frame t>=1 -> 2t-1 and 2t).
template <int LPP, typename OT = float>
__global__ __launch_bounds__(256) void vae_prep_ex_kernel(const float* __restrict__ x, OT* __restrict__ y, int64_t npix, int Hh, int Ww, int C,
                                                          const float* __restrict__ mul, const float* __restrict__ add, int silu, int clamp01, int up_hw,
                                                          int up_t, int64_t y_frame_stride, int64_t y_row_stride) {
  constexpr int GPB = 256 / LPP;
  const int g = threadIdx.x / LPP, l = threadIdx.x % LPP;
  const int nch = C / 4;
  for (int64_t pix = (int64_t)blockIdx.x * GPB + g; pix < npix; pix += (int64_t)gridDim.x * GPB) {
    const int64_t t = pix / ((int64_t)Hh * Ww);
    const int rem = (int)(pix - t * (int64_t)Hh * Ww);
    const int h = rem / Ww, wq = rem - h * Ww;
    const int64_t t0 = up_t ? (t == 0 ? 0 : 2 * t - 1) : t;
    const int nt = (up_t && t > 0) ? 2 : 1;
    for (int c4 = l; c4 < nch; c4 += LPP) {
      const float4 v = *reinterpret_cast<const float4*>(x + pix * C + c4 * 4);
      float o[4] = {v.x, v.y, v.z, v.w};
      if (mul != nullptr) {
        const float4 m = *reinterpret_cast<const float4*>(mul + c4 * 4);
        o[0] *= m.x; o[1] *= m.y; o[2] *= m.z; o[3] *= m.w;
      }
      if (add != nullptr) {
        const float4 a = *reinterpret_cast<const float4*>(add + c4 * 4);
        o[0] += a.x; o[1] += a.y; o[2] += a.z; o[3] += a.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (silu) o[e] = o[e] / (1.f + __expf(-o[e]));
        if (clamp01) o[e] = fminf(fmaxf(o[e], 0.f), 1.f);
      }
      // 4 channels per lane: a float4 or, for the fp16 conv operand buffers, 4 halves (strides are in elements of the output type)
      struct alignas(sizeof(OT) * 4) Out4 { OT e[4]; };
      const Out4 ov = {{(OT)o[0], (OT)o[1], (OT)o[2], (OT)o[3]}};
      for (int dt = 0; dt < nt; ++dt) {
        OT* yb = y + (t0 + dt) * y_frame_stride + (int64_t)(up_hw ? 2 * h : h) * y_row_stride + (int64_t)(up_hw ? 2 * wq : wq) * C + c4 * 4;
        *reinterpret_cast<Out4*>(yb) = ov;
        if (up_hw) {
          *reinterpret_cast<Out4*>(yb + C) = ov;
          *reinterpret_cast<Out4*>(yb + y_row_stride) = ov;
          *reinterpret_cast<Out4*>(yb + y_row_stride + C) = ov;
        }
      }
    }
  }
}

// Replicate padding of a conv input buffer [lead + T][H + 2p][W + 2p][C]: spatial borders copy the nearest interior pixel,
// the `lead` leading frames copy frame `lead` (F.pad(..., mode="replicate") with (p, p, p, p, kt-1, 0), :84-91).
__global__ __launch_bounds__(256) void vae_replicate_border_kernel(float* __restrict__ buf, int frames, int lead, int Hp, int Wp, int C, int pad) {
  const int nch = C / 4;
  const int64_t total = (int64_t)frames * Hp * Wp * nch;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % nch);
    int64_t r = i / nch;
    const int w = (int)(r % Wp);
    r /= Wp;
    const int h = (int)(r % Hp);
    const int f = (int)(r / Hp);
    const int hs = min(max(h, pad), Hp - 1 - pad), ws = min(max(w, pad), Wp - 1 - pad), fs = max(f, lead);
    if (hs == h && ws == w && fs == f) continue;  // interior pixel of a real frame
    const float4 v = *reinterpret_cast<const float4*>(buf + (((int64_t)fs * Hp + hs) * Wp + ws) * C + c4 * 4);
    *reinterpret_cast<float4*>(buf + (((int64_t)f * Hp + h) * Wp + w) * C + c4 * 4) = v;
  }
}

// GroupNorm statistics over a channels-last tensor [npix][C] with G groups of C/G consecutive channels: per-group
// sum and sum of squares accumulated in fp64 (block partials in LDS, one fp64 atomicAdd pair per group per block).
__global__ __launch_bounds__(256) void groupnorm_stats_kernel(const float* __restrict__ x, int64_t npix, int C, int G, double* __restrict__ acc) {
  __shared__ double part[2 * 64];
  if (threadIdx.x < 2 * G) part[threadIdx.x] = 0.0;
  __syncthreads();
  const int nch = C / 4, cpg = C / G;
  const int64_t total = npix * nch;
  // a thread keeps its channel chunk fixed across iterations when the stride is a multiple of nch; accumulate locally per visited group
  double s = 0.0, q = 0.0;
  int cur = -1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % nch);
    const int grp = (c4 * 4) / cpg;
    if (grp != cur) {
      if (cur >= 0) {
        atomicAdd(&part[2 * cur], s);
        atomicAdd(&part[2 * cur + 1], q);
      }
      cur = grp;
      s = q = 0.0;
    }
    const float4 v = *reinterpret_cast<const float4*>(x + i * 4);
    s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
    q += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  if (cur >= 0) {
    atomicAdd(&part[2 * cur], s);
    atomicAdd(&part[2 * cur + 1], q);
  }
  __syncthreads();
  if (threadIdx.x < 2 * G) atomicAdd(&acc[threadIdx.x], part[threadIdx.x]);
}

// per-channel affine of GroupNorm from the accumulated sums: mul[c] = rstd_g * gamma[c], add[c] = beta[c] - mean_g * mul[c]
__global__ void groupnorm_finalize_kernel(const double* __restrict__ acc, const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ mul,
                                          float* __restrict__ add, int C, int G, double count, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int g = c / (C / G);
  const double mean = acc[2 * g] / count;
  const double var = fmax(acc[2 * g + 1] / count - mean * mean, 0.0);
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float m = rstd * gamma[c];
  mul[c] = m;
  add[c] = beta[c] - (float)mean * m;
}

// In-place row softmax with a frame-causal prefix mask (prepare_causal_attention_mask :48-63): row i (frame i / hw) sees keys
// j < (i/hw + 1) * hw; masked entries become 0.
__global__ __launch_bounds__(256) void softmax_rows_causal_kernel(float* __restrict__ s, int64_t lds_, int N, float scale, int hw, int n_keys) {
  __shared__ float red[4];
  float* row = s + (int64_t)blockIdx.x * lds_;
  const int valid = min(n_keys, (int)((blockIdx.x / hw + 1) * hw));  // n_keys <= N: columns past it are padding
  float mx = -3.0e38f;
  for (int i = threadIdx.x; i < valid; i += 256) mx = fmaxf(mx, row[i]);
  mx = block_max<4>(mx, red) * scale;
  float sum = 0.f;
  for (int i = threadIdx.x; i < valid; i += 256) {
    const float e = __expf(row[i] * scale - mx);
    row[i] = e;
    sum += e;
  }
  sum = block_sum<4>(sum, red);
  const float inv = 1.f / sum;
  for (int i = threadIdx.x; i < N; i += 256) row[i] = i < valid ? row[i] * inv : 0.f;
}

// Linear cross-fade of two overlapping tiles along one axis (blend_v / blend_h / blend_t :347-364):
// b[idx] = a[na - extent + idx] * (1 - idx/extent) + b[idx] * (idx/extent) for idx < extent along the axis.
// Both tensors are addressed as [outer][axis][inner] with their own axis lengths; outer/inner extents are common.
__global__ __launch_bounds__(256) void blend_axis_kernel(const float* __restrict__ a, float* __restrict__ b, int64_t outer, int na, int nb, int64_t inner,
                                                         int64_t a_outer_stride, int64_t b_outer_stride, int extent) {
  const int64_t total = outer * extent * inner;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t in = i % inner;
    const int idx = (int)((i / inner) % extent);
    const int64_t o = i / (inner * extent);
    const float wb = (float)idx / (float)extent;
    const float av = a[o * a_outer_stride + (int64_t)(na - extent + idx) * inner + in];
    float* bp = b + o * b_outer_stride + (int64_t)idx * inner + in;
    *bp = av * (1.f - wb) + *bp * wb;
  }
}


// ------------------------------------------------------------------------------------------------
// Halo-tiled form of the 16-bit convolution for 3x3 spatial kernels (kt = 1 or 3).  vae_conv16_kernel stages a fresh 256-pixel
// input block for every tap — 48 KB from the L2 per 4.2 MFLOP, and the PMC profile shows it pinned at the L2's aggregate request
// rate (11 TB/s, matrix pipe 32 % busy).  Here a workgroup owns an 8 x 32 pixel tile of one output frame and stages, per input
// frame tap dt and 64-channel slab, the 10 x 34 pixel HALO block once (44 KB); the nine spatial taps then read their shifted
// 8 x 32 windows out of that one LDS image, so only the weights (16 KB per tap) stream per step: 187 KB per 37.7 MFLOP, 2.3x fewer
// L2 bytes per FLOP.  Same MFMA tiling as vae_conv16_kernel (wave = 2 image rows x 32 px = two 32-row MFMA blocks, 32 NF couts).
//   * LDS: halo image [352 rows][128 B] x 2 (row r = hy*34 + hx, chunk ^= (r >> 1) & 7 applied on the DMA source as everywhere),
//     weight slab [32 NF][128 B] x 2.  Fragment row of lane fl for tap (dh, dw), image row 2 wid + mi: r = (2 wid + mi + dh)*34 + dw + fl;
//     the swizzle depends on r, so the 8 fragment addresses are recomputed per tap (a few VALU against 8 NF MFMAs).
//   * schedule: step = (slab, tap), one barrier per step, 4 compute waves + 4 loader waves (one of each per SIMD).  The weight slabs
//     travel through a RING of VH_NB = 4 LDS slots: at step s a loader issues the weights of step s + 3 (and, on a slab's first tap, the
//     next slab's halo: 11 pieces per loader wave, padded rows masked) and waits only until the weights of step s + 1 have landed — a
//     counted vmcnt that leaves the two younger weight slabs (and, for its first three steps, the halo) in flight.  Round 5: with a
//     two-slot ring every step waited for weights issued ONE step earlier, i.e. a step lasted an L2 round trip (~2200 cycles against the
//     step's 1024 MFMA cycles: the 45 % matrix-pipe busy of the round-1 profile), and __syncthreads() in front of the barrier drained
//     the halo as well (hipcc's VMEM drain for the release fence) — both found in the ISA.
// the bare instruction, not __syncthreads(): its release fence makes hipcc drain the wave's whole VMEM queue in front of the barrier (+2.8 % on the decode)
#define VH_BARRIER() asm volatile("s_barrier" ::: "memory")
#ifndef X2V_VH_RING
#define X2V_VH_RING 4  // weight-slab ring slots (A/B builds: 2 = the one-step-ahead form of rounds 1-4)
#endif
#ifndef X2V_VH_PREFETCH
#define X2V_VH_PREFETCH (X2V_VH_RING >= 4)  // the compute waves read step s + 1's first fragments in front of step s's barrier (A/B builds: 0)
#endif
#ifndef X2V_VH_INTERLEAVE
#define X2V_VH_INTERLEAVE 1  // one fragment read behind every MFMA (slot form; A/B builds: 0 = read blocks between MFMA blocks)
#endif
#ifndef X2V_VH_DIST
#define X2V_VH_DIST 1  // slot form: k-steps of fragment read-ahead.  2 (three fragment sets, step loop unrolled by 3, 239 VGPRs) was measured in round 5
                       // and is SLOWER: 2.62 vs 2.38 s per 720p x 81f decode on one box (profiles/r05_call10_*) — the read pipeline's depth is not the limiter either
#endif
constexpr int VH_NB = X2V_VH_RING;
constexpr int VH_NEED = X2V_VH_PREFETCH ? 2 : 1;  // at step s the loaders wait for the weights of step s + VH_NEED
static_assert(VH_NB == 2 || VH_NB == 4, "ring of 2 (one step ahead) or 4 (three steps ahead)");
static_assert(VH_NEED <= VH_NB - 1, "the cross-step fragment prefetch needs the 4-slot ring");
constexpr int VH_TH = 8, VH_TW = 32, VH_HW = VH_TW + 2, VH_ROWS = 352, VH_A_BYTES = VH_ROWS * 128, VH_A_PIECES = VH_ROWS / 8 / 4;

template <int NF>
__global__ __launch_bounds__(512, 2) void vae_conv16h_kernel(const _Float16* __restrict__ xp, int64_t x_frame_stride, int64_t x_row_stride, int64_t x_px_stride,
                                                          const _Float16* __restrict__ w, int64_t w_row_stride, const float* __restrict__ bias,
                                                          const float* __restrict__ resid, float* __restrict__ y, int T, int Hh, int Ww, int Cin, int Cout,
                                                          int kt, int flags, int ncol, int tiles_x, int tiles_y) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ROWB = 128, CPR = 8, RPI = 8;
  constexpr int BN = 32 * NF;
  constexpr int B_BYTES = BN * ROWB;
  constexpr int B_INSTR = (BN / 4 + RPI - 1) / RPI;
  constexpr int B_OFF = 2 * VH_A_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // 8 waves, two per SIMD: waves 0..3 compute (MFMAs + fragment reads only), waves 4..7 move data (every LDS-DMA piece of the
  // workgroup).  An LDS-DMA issue costs 60-185 cycles of a wave's issue stream; with the four compute waves issuing their own 5-12
  // pieces per step the matrix pipe sat idle two thirds of the time — a partner wave per SIMD that does nothing else hides it.
  const bool loader = wave >= 4;
  const int wid = wave & 3;
  const int fl = lane & 31, fh = lane >> 5;
  // persistent: a workgroup walks tiles blockIdx.x, + gridDim.x, ...; after a tile's last barrier nothing reads LDS any more, so the
  // loader waves stage the next tile's first halo and weights while the compute waves are still in the previous tile's fp32 epilogue
  // (128 KB out + 128 KB residual in per tile: 13-23 % of a tile's life, unhidden when a tile is a workgroup)
  const unsigned ntiles = (unsigned)T * (unsigned)tiles_y * (unsigned)tiles_x * (unsigned)ncol;
  for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  const unsigned v = xcd_remap(tile, ntiles);
  const int ctile = (int)(v % (unsigned)ncol);
  unsigned pt = v / (unsigned)ncol;
  const int tx = (int)(pt % (unsigned)tiles_x);
  pt /= (unsigned)tiles_x;
  const int ty = (int)(pt % (unsigned)tiles_y);
  const int frame = (int)(pt / (unsigned)tiles_y);
  const int y0 = ty * VH_TH, x0 = tx * VH_TW;
  const int co0 = ctile * BN;
  const int taps = kt * 9;

  const int64_t fbytes = x_frame_stride * 2;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(xp + (int64_t)frame * x_frame_stride), 0, (unsigned)(fbytes * kt), 0x00020000);
  const int wrows = min(BN, Cout - co0);
  const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(w + (int64_t)co0 * w_row_stride), 0, (unsigned)(((int64_t)(wrows - 1) * w_row_stride + (int64_t)taps * Cin) * 2), 0x00020000);

  // halo DMA: wave wid moves pieces wid*11 .. wid*11+10 (8 halo rows each); source = padded-buffer pixel (y0 + hy, x0 + hx)
  unsigned a_voff[VH_A_PIECES], b_voff[B_INSTR];
#pragma unroll
  for (int i = 0; i < VH_A_PIECES; ++i) {
    const int r = (wid * VH_A_PIECES + i) * 8 + lane / CPR;
    const int c = (lane % CPR) ^ ((r >> 1) & (CPR - 1));
    const int hy = r / VH_HW, hx = r - hy * VH_HW;
    a_voff[i] = r < (VH_TH + 2) * VH_HW ? (unsigned)(((int64_t)(y0 + hy) * x_row_stride + (int64_t)(x0 + hx) * x_px_stride) * 2) + (unsigned)(c << 4) : VC_OOB;
  }
#pragma unroll
  for (int i = 0; i < B_INSTR; ++i) {
    const int rl = i * RPI + lane / CPR;
    const int r = wid * (BN / 4) + rl;
    const int c = (lane % CPR) ^ ((r >> 1) & (CPR - 1));
    b_voff[i] = (rl < BN / 4 && r < wrows) ? (unsigned)((int64_t)r * w_row_stride * 2) + (unsigned)(c << 4) : VC_OOB;
  }
  const int kchunks = Cin / 64;
  const int nslabs = kt * kchunks;  // (dt, kc)
  auto stage_a = [&](int buf, int slab, int p0 = 0, int p1 = VH_A_PIECES) {  // this loader wave's halo pieces [p0, p1) of a slab
    const int dt = slab / kchunks, kc = slab - dt * kchunks;
    const unsigned xso = (unsigned)(((int64_t)dt * x_frame_stride + (int64_t)kc * 64) * 2);
    char* as = smem + buf * VH_A_BYTES + wid * (VH_A_PIECES * 1024);
#pragma unroll
    for (int i = 0; i < VH_A_PIECES; ++i)
      if (i >= p0 && i < p1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (v_lds_ptr_t)(as + i * 1024), 16, a_voff[i], xso, 0, 0);
  };
  auto stage_b = [&](int buf, int slab, int tap9) {
    const int dt = slab / kchunks, kc = slab - dt * kchunks;
    const unsigned wso = (unsigned)(((int64_t)(dt * 9 + tap9) * Cin + (int64_t)kc * 64) * 2);
    char* bs = smem + B_OFF + buf * B_BYTES + wid * ((BN / 4) * ROWB);
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i)
      if ((i + 1) * RPI <= BN / 4 || lane / CPR + i * RPI < BN / 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rwt, (v_lds_ptr_t)(bs + i * 1024), 16, b_voff[i], wso, 0, 0);
  };

  int rd_b[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) rd_b[ks] = fl * ROWB + ((((ks << 1) | fh) ^ ((fl >> 1) & (CPR - 1))) << 4);

  f32x16_t acc[2][NF];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nsteps = nslabs * 9;
  constexpr int AHEAD = VH_NB - 1;  // weights of step s + AHEAD are issued at step s
  auto stage_b_step = [&](int st) {
    const int sl = st / 9;
    stage_b(st & (VH_NB - 1), sl, st - sl * 9);
  };
  if (loader) {
#pragma unroll
    for (int j = 0; j < AHEAD; ++j) stage_b_step(j);  // nsteps >= 9 > AHEAD
    stage_a(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (loader) {
    // same step structure as the compute waves (one barrier per step).  Step s: issue the weights of step s + AHEAD into their ring slot (its last
    // reader was step s - 1, behind the previous barrier) and, on a slab's first tap, the next slab's halo; then wait until the weights of step
    // s + VH_NEED have landed (VH_NEED = 2: the compute waves read step s + 1's first fragments in front of THIS step's barrier, so those weights were
    // published one barrier earlier).  LDS-DMA pieces retire in issue order, so that is "all but the pieces issued after them": AHEAD - VH_NEED weight
    // slabs (fewer at the tile's end) and, on the first taps, the halo issued on tap 0 (behind that step's weights) — then the count drops and the halo
    // is drained, long before the next slab's first fragments are read.
    // Halo pieces per tap: all eleven on tap 0 (spreading the burst over six taps made the decode 5.7 % slower, profiles/r05_call11_*).
    // These two functions are what the counted waits below are computed from: they must mirror the issue statements of the tap loop exactly
    // (stage_b_step issues B_INSTR pieces, stage_a issues the halo's VH_A_PIECES on a slab's first tap).
    auto halo_pieces = [&](int j) -> int {  // pieces this wave issues in step j (behind that step's weights)
      if (j < 0) return 0;
      const int sl = j / 9, tp = j - sl * 9;
      if (sl + 1 >= nslabs) return 0;
      return tp == 0 ? VH_A_PIECES : 0;
    };
    auto weight_pieces = [&](int j) -> int { return j + AHEAD < nsteps ? B_INSTR : 0; };
    auto vmcnt_wait = [&](int n) {
      switch (n) {
#define VH_WAITCASE(N_) case N_: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); break;
        VH_WAITCASE(1) VH_WAITCASE(2) VH_WAITCASE(3) VH_WAITCASE(4) VH_WAITCASE(5) VH_WAITCASE(6) VH_WAITCASE(7) VH_WAITCASE(8) VH_WAITCASE(9) VH_WAITCASE(10)
        VH_WAITCASE(11) VH_WAITCASE(12) VH_WAITCASE(13) VH_WAITCASE(14) VH_WAITCASE(15) VH_WAITCASE(16) VH_WAITCASE(17) VH_WAITCASE(18) VH_WAITCASE(19)
#undef VH_WAITCASE
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      }
    };
    int st = 0;
    for (int slab = 0; slab < nslabs; ++slab) {
#pragma unroll 1
      for (int tap9 = 0; tap9 < 9; ++tap9, ++st) {
        if (weight_pieces(st)) stage_b_step(st + AHEAD);
        const int hp = halo_pieces(st);
        if (hp) {
          stage_a((slab + 1) & 1, slab + 1, 0, hp);
        }
        // what must have landed: the weights of step st + VH_NEED, issued first thing in step j0 = st + VH_NEED - AHEAD.  Pieces retire in issue order,
        // so everything issued behind them may stay in flight: step j0's halo pieces, then weights + halo pieces of steps j0 + 1 .. st.  (At the tile's
        // end, where no such weights exist, everything is drained; a slab's last halo pieces are behind a later step's weights within two steps,
        // long before the next slab's first fragments are read.)
        int allowed = 0;
        if (st + VH_NEED < nsteps) {
          const int j0 = st + VH_NEED - AHEAD;
          allowed = halo_pieces(j0);
#pragma unroll
          for (int j = 1; j <= AHEAD - VH_NEED; ++j) allowed += weight_pieces(j0 + j) + halo_pieces(j0 + j);
        }
        vmcnt_wait(allowed);
        VH_BARRIER();
      }
    }
    continue;
  }
  // Compute waves.  Fragments of k-step ks + 1 are read before the MFMAs of k-step ks are issued (pinned: left to itself hipcc reads a k-step's
  // fragments right in front of its MFMAs and the LDS latency is paid four times per step) — and, with VH_NEED = 2, the first fragments of step
  // s + 1 before the MFMAs of step s's last k-step, i.e. in front of the barrier: the pipeline of fragment reads runs through the whole tile and no
  // step starts by waiting an LDS round trip (round 5; PMC before: matrix pipe 50.5 % busy, ~1500 cycles per 768-cycle step).
  struct Step {
    const char *ab, *bb;
    int ra[2], sw[2];
  };
  auto step_addr = [&](int sl, int tp, int stn) {
    Step a;
    const int dh = tp / 3, dw = tp - dh * 3;
    a.ab = smem + (sl & 1) * VH_A_BYTES;
    a.bb = smem + B_OFF + (stn & (VH_NB - 1)) * B_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (2 * wid + i + dh) * VH_HW + dw + fl;
      a.ra[i] = r * ROWB;
      a.sw[i] = (r >> 1) & (CPR - 1);
    }
    return a;
  };
  // SLOT form (NF < 4): MFMA m of a k-step is followed by fragment read m of the k-step DIST ahead (2 + NF reads for 2 NF MFMAs), pinned with
  // sched_barrier; the reads run DIST k-steps ahead through DIST + 1 fragment sets — across the step boundary, i.e. in front of the barrier (VH_NEED = 2:
  // the next step's weights were published one barrier earlier).  Round 5 (profiles/r05_call8_* .. r05_call11_*): against the BLOCK form (all reads of a
  // k-step, then all its MFMAs — the read / address instructions between two MFMA blocks drain the matrix pipe four times per step; still what NF = 4 runs,
  // where 128 accumulators leave no room for more) the slot form is worth +3.5 % of the 720p decode; DIST = 2 (three sets, step loop unrolled by 3, 239 VGPRs)
  // is 9 % SLOWER than DIST = 1.  Knock-out probes of that round (a build without MFMAs / without DMA / without barriers, results invalid by construction,
  // git history: commit "VAE timing probes") put the decode at 75 % / 85 % / 97 % of its time: no single resource is the limit, the step is too small.
  constexpr bool SLOT = X2V_VH_INTERLEAVE != 0 && NF < 4;
  constexpr bool PF = X2V_VH_PREFETCH != 0 && NF < 4;  // reads run on across the step boundary (NF = 4: no registers for it — every step starts with its own first reads)
  constexpr int DIST = SLOT && PF ? X2V_VH_DIST : 1;  // k-steps of read-ahead
  constexpr int NSET = DIST + 1;                                   // fragment sets
  constexpr int UNR = NSET == 3 ? 3 : 1;                           // a step has 4 k-steps: the set phase advances by 4 mod NSET per step
  static_assert(DIST == 1 || DIST == 2, "one or two k-steps of fragment read-ahead");
  vc_half8_t xa[NSET][2], wb[NSET][NF];
  // read r of k-step KS of the step addressed by A into fragment set S: r = 0: xa[0], 1 .. NF: wb[0 .. NF-1], NF + 1: xa[1]
  auto frag_read = [&](int S, int KS, const Step& A, int r) {
    if (r == 0 || r == NF + 1) {
      const int ii = r == 0 ? 0 : 1;
      xa[S][ii] = *reinterpret_cast<const vc_half8_t*>(A.ab + A.ra[ii] + ((((KS << 1) | fh) ^ A.sw[ii]) << 4));
    } else {
      wb[S][r - 1] = *reinterpret_cast<const vc_half8_t*>(A.bb + (r - 1) * 32 * ROWB + rd_b[KS]);
    }
  };
  Step cur = step_addr(0, 0, 0);
  if (PF) {  // the pipeline's head: the first DIST k-steps of the tile
#pragma unroll
    for (int d = 0; d < DIST; ++d)
#pragma unroll
      for (int r = 0; r < 2 + NF; ++r) frag_read(d, d, cur, r);
  }
  int slab = 0, tap9 = 0;
#pragma unroll 1
  for (int st = 0; st < nsteps; st += UNR) {  // nsteps is a multiple of 9
#pragma unroll
    for (int p = 0; p < UNR; ++p) {
      int nslab = slab, ntap = tap9 + 1;
      if (ntap == 9) {
        ntap = 0;
        ++nslab;
      }
      // (behind the tile's last step the "next step" addresses still lie inside the LDS images: those reads are harmless and unused — no branch)
      const Step nxt = step_addr(nslab, ntap, st + p + 1);
      if (!PF) {
#pragma unroll
        for (int r = 0; r < 2 + NF; ++r) frag_read(0, 0, cur, r);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int cs = (4 * p + ks) % NSET, ns = (4 * p + ks + DIST) % NSET;  // sets consumed / filled by this k-step
        const int kn = ks + DIST;                                           // the k-step read now: of this step, or of the next one
        const bool rd = kn < 4 || PF;
        if constexpr (SLOT) {
#pragma unroll
          for (int m = 0; m < 2 * NF; ++m) {
            const int i = m / NF, n = m - i * NF;
            acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[cs][n], xa[cs][i], acc[i][n], 0, 0, 0);
            // slot m carries read m (the last slot every read that is left: NF = 1 has 3 reads for 2 MFMAs)
#pragma unroll
            for (int r = m; r < (m == 2 * NF - 1 ? 2 + NF : m + 1); ++r)
              if (r < 2 + NF && rd) frag_read(ns, kn & 3, kn < 4 ? cur : nxt, r);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          if (rd) {
#pragma unroll
            for (int r = 0; r < 2 + NF; ++r) frag_read(ns, kn & 3, kn < 4 ? cur : nxt, r);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int n = 0; n < NF; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[cs][n], xa[cs][i], acc[i][n], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      VH_BARRIER();
      cur = nxt;
      slab = nslab;
      tap9 = ntap;
    }
  }

  // epilogue.  acc[i][n][r]: pixel (y0 + 2 wid + i, x0 + fl), cout co0 + n*32 + (r&3) + 8*(r>>2) + 4*fh
  const bool vec_ok = (Cout & 3) == 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int py = y0 + 2 * wid + i, px = x0 + fl;
    if (py >= Hh || px >= Ww) continue;
    const int64_t pbase = ((int64_t)frame * Hh + py) * Ww + px;
#pragma unroll
    for (int n = 0; n < NF; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = co0 + n * 32 + 8 * g + 4 * fh;
        if (co >= Cout) continue;
        float vv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) vv[e] = acc[i][n][4 * g + e];
        const int64_t oidx = pbase * Cout + co;
        if (vec_ok) {
          if (bias != nullptr) {
            const float4 b4 = *reinterpret_cast<const float4*>(bias + co);
            vv[0] += b4.x; vv[1] += b4.y; vv[2] += b4.z; vv[3] += b4.w;
          }
          if (resid != nullptr) {
            const float4 r4 = *reinterpret_cast<const float4*>(resid + oidx);
            vv[0] += r4.x; vv[1] += r4.y; vv[2] += r4.z; vv[3] += r4.w;
          }
          if (flags & VCF_CLAMP) {
#pragma unroll
            for (int e = 0; e < 4; ++e) vv[e] = fminf(fmaxf(vv[e], -1.f), 1.f);
          }
          *reinterpret_cast<float4*>(y + oidx) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (co + e < Cout) {
              float o = vv[e] + (bias != nullptr ? bias[co + e] : 0.f) + (resid != nullptr ? resid[oidx + e] : 0.f);
              if (flags & VCF_CLAMP) o = fminf(fmaxf(o, -1.f), 1.f);
              y[oidx + e] = o;
            }
        }
      }
  }
  }  // persistent tile loop
#endif
}

bool vae_conv16g_ok(int Ww, int Cin, int Cout);
int vae_conv16g_dispatch(const void* xp, const void* cache, int64_t fs, int64_t rs, int64_t ps, const void* w, int64_t wrs, const float* bias, const float* resid, float* y,
                         int T, int Hh, int Ww, int Cin, int Cout, int kt, int flags, int cin_zero_tail, hipStream_t st);

}  // namespace x2v

using namespace x2v;

template <int NF, int KC>
static int launch_vconv(const float* xp, int64_t fs, int64_t rs, int64_t ps, const float* w, int64_t wrs, const float* bias, const float* resid, float* y,
                        int T, int Hh, int Ww, int Cin, int Cout, int kt, int kh, int kw, int flags, hipStream_t st) {
  constexpr int lds = 2 * (VC_PIX + 32 * NF) * KC * 4;
  {
    int rc = ensure_dynamic_lds((const void*)vae_conv_kernel<NF, KC>, lds, "vae conv attr");
    if (rc != X2V_OK) return rc;
  }
  const int64_t ptiles = (int64_t)T * (((int64_t)Hh * Ww + VC_PIX - 1) / VC_PIX);
  const int ncol = (Cout + 32 * NF - 1) / (32 * NF);
  X2V_REQUIRE(ptiles * ncol < (1ll << 31), X2V_E_SHAPE, "vae_conv: too many tiles");
  hipLaunchKernelGGL((vae_conv_kernel<NF, KC>), dim3((unsigned)(ptiles * ncol)), dim3(256), lds, st, xp, fs, rs, ps, w, wrs, bias, resid, y, T, Hh, Ww, Cin, Cout,
                     kt, kh, kw, flags, ncol);
  X2V_LAUNCH_CHECK("vae_conv launch");
  return X2V_OK;
}

extern "C" __attribute__((visibility("default"))) int x2v_vae_conv_f32(const float* xp, int64_t x_frame_stride, int64_t x_row_stride, int64_t x_px_stride,
                                                                       const float* w, int64_t w_row_stride, const float* bias, const float* resid, float* y,
                                                                       int T, int Hh, int Ww, int Cin, int Cout, int kt, int kh, int kw, int flags,
                                                                       void* stream) {
  X2V_REQUIRE(xp && w && y, X2V_E_ARG, "vae_conv: null pointer");
  X2V_REQUIRE(T > 0 && Hh > 0 && Ww > 0 && Cin > 0 && Cout > 0, X2V_E_SHAPE, "vae_conv: bad shape");
  X2V_REQUIRE(kt >= 1 && kt <= 3 && kh >= 1 && kh <= 3 && kw >= 1 && kw <= 3, X2V_E_SHAPE, "vae_conv: kernel %dx%dx%d unsupported", kt, kh, kw);
  X2V_REQUIRE(Cin % 16 == 0, X2V_E_SHAPE, "vae_conv: Cin=%d must be a multiple of 16", Cin);
  X2V_REQUIRE(x_px_stride % 4 == 0 && x_row_stride % 4 == 0 && x_frame_stride % 4 == 0 && w_row_stride % 4 == 0 && x_px_stride >= Cin &&
                  w_row_stride >= (int64_t)kt * kh * kw * Cin,
              X2V_E_ALIGN, "vae_conv: strides must be multiples of 4 floats and cover the extents");
  X2V_REQUIRE(aligned16(xp) && aligned16(w) && aligned16(y) && aligned16(resid) && aligned16(bias), X2V_E_ALIGN, "vae_conv: pointers must be 16-byte aligned");
  X2V_REQUIRE(x_frame_stride * 4 * kt < (1ll << 31) && w_row_stride * 4 * 128 < (1ll << 31), X2V_E_SHAPE,
              "vae_conv: a kt-frame input window / 128 weight rows must stay below 2 GiB (32-bit buffer offsets)");
  X2V_REQUIRE(!(flags & VCF_TSPLIT) || (Cout % 8 == 0 && resid == nullptr), X2V_E_ARG, "vae_conv: time-split output needs Cout %% 8 == 0 and no residual");
  hipStream_t st = (hipStream_t)stream;
#define X2V_VC(NF_, KC_) return launch_vconv<NF_, KC_>(xp, x_frame_stride, x_row_stride, x_px_stride, w, w_row_stride, bias, resid, y, T, Hh, Ww, Cin, Cout, kt, kh, kw, flags, st)
  const bool k32 = Cin % 32 == 0;
  if (Cout <= 32) {
    if (k32) X2V_VC(1, 32); else X2V_VC(1, 16);
  } else if (Cout % 128 != 0 && Cout % 96 == 0) {
    if (k32) X2V_VC(3, 32); else X2V_VC(3, 16);
  } else if (Cout <= 64) {
    if (k32) X2V_VC(2, 32); else X2V_VC(2, 16);
  } else {
    if (k32) X2V_VC(4, 32); else X2V_VC(4, 16);
  }
#undef X2V_VC
}

extern "C" __attribute__((visibility("default"))) int x2v_vae_prep_f32(const float* x, float* y, int T, int Hh, int Ww, int C, const float* gamma,
                                                                       const float* a, const float* b, int silu, int upsample, int64_t y_frame_stride,
                                                                       int64_t y_row_stride, void* stream) {
  X2V_REQUIRE(x && y, X2V_E_ARG, "vae_prep: null pointer");
  X2V_REQUIRE(T > 0 && Hh > 0 && Ww > 0 && C > 0 && C % 4 == 0 && C <= 1024, X2V_E_SHAPE, "vae_prep: bad shape (C %% 4 == 0, C <= 1024)");
  X2V_REQUIRE(y_frame_stride % 4 == 0 && y_row_stride % 4 == 0 && aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(a) && aligned16(b), X2V_E_ALIGN,
              "vae_prep: 16-byte alignment");
  const int64_t npix = (int64_t)T * Hh * Ww;
  hipStream_t st = (hipStream_t)stream;
  if (C <= 128) {
    const int64_t blocks = std::min<int64_t>((npix + 7) / 8, 65536 * 4);
    hipLaunchKernelGGL((vae_prep_kernel<32>), dim3((unsigned)blocks), dim3(256), 0, st, x, y, npix, Hh, Ww, C, gamma, a, b, silu, upsample, y_frame_stride, y_row_stride,
                       (int64_t)C);
  } else {
    const int64_t blocks = std::min<int64_t>((npix + 3) / 4, 65536 * 4);
    hipLaunchKernelGGL((vae_prep_kernel<64>), dim3((unsigned)blocks), dim3(256), 0, st, x, y, npix, Hh, Ww, C, gamma, a, b, silu, upsample, y_frame_stride, y_row_stride,
                       (int64_t)C);
  }
  X2V_LAUNCH_CHECK("vae_prep launch");
  return X2V_OK;
}

extern "C" __attribute__((visibility("default"))) int x2v_softmax_rows_f32(float* s, int64_t ld, int64_t M, int N, float scale, void* stream) {
  X2V_REQUIRE(s, X2V_E_ARG, "softmax_rows: null pointer");
  X2V_REQUIRE(M > 0 && N > 0 && N % 4 == 0 && ld % 4 == 0 && ld >= N && M < (1ll << 31), X2V_E_SHAPE, "softmax_rows: bad shape (N %% 4 == 0)");
  X2V_REQUIRE(aligned16(s), X2V_E_ALIGN, "softmax_rows: 16-byte alignment");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)M), dim3(256), 0, (hipStream_t)stream, s, ld, M, N, scale);
  X2V_LAUNCH_CHECK("softmax_rows launch");
  return X2V_OK;
}

extern "C" __attribute__((visibility("default"))) int x2v_vae_prep_ex_f32(const float* x, float* y, int T, int Hh, int Ww, int C, const float* mul, const float* add, int silu,
                                                                          int clamp01, int up_hw, int up_t, int64_t y_frame_stride, int64_t y_row_stride,
                                                                          void* stream) {
  X2V_REQUIRE(x && y, X2V_E_ARG, "vae_prep_ex: null pointer");
  X2V_REQUIRE(T > 0 && Hh > 0 && Ww > 0 && C > 0 && C % 4 == 0, X2V_E_SHAPE, "vae_prep_ex: bad shape (C %% 4 == 0)");
  X2V_REQUIRE(y_frame_stride % 4 == 0 && y_row_stride % 4 == 0 && aligned16(x) && aligned16(y) && aligned16(mul) && aligned16(add), X2V_E_ALIGN,
              "vae_prep_ex: 16-byte alignment");
  const int64_t npix = (int64_t)T * Hh * Ww;
  hipStream_t st = (hipStream_t)stream;
  if (C <= 128) {
    const int64_t blocks = std::min<int64_t>((npix + 7) / 8, 65536 * 4);
    hipLaunchKernelGGL((vae_prep_ex_kernel<32>), dim3((unsigned)blocks), dim3(256), 0, st, x, y, npix, Hh, Ww, C, mul, add, silu, clamp01, up_hw, up_t, y_frame_stride, y_row_stride);
  } else {
    const int64_t blocks = std::min<int64_t>((npix + 3) / 4, 65536 * 4);
    hipLaunchKernelGGL((vae_prep_ex_kernel<64>), dim3((unsigned)blocks), dim3(256), 0, st, x, y, npix, Hh, Ww, C, mul, add, silu, clamp01, up_hw, up_t, y_frame_stride, y_row_stride);
  }
  X2V_LAUNCH_CHECK("vae_prep_ex launch");
  return X2V_OK;
}

// x2v_vae_prep_f32 writing fp16 with an explicit pixel stride (channel axis padded to a multiple of 64 for x2v_vae_conv_f16); strides in halves
static int vae_prep_f16_impl(const float* x, void* y, int T, int Hh, int Ww, int C, const float* gamma, const float* a, const float* b, int silu, int upsample,
                             int64_t y_frame_stride, int64_t y_row_stride, int64_t y_px_stride, int split, void* stream) {
  X2V_REQUIRE(x && y, X2V_E_ARG, "vae_prep_f16: null pointer");
  X2V_REQUIRE(T > 0 && Hh > 0 && Ww > 0 && C > 0 && C % 4 == 0 && C <= 1024, X2V_E_SHAPE, "vae_prep_f16: bad shape (C %% 4 == 0, C <= 1024)");
  X2V_REQUIRE(y_frame_stride % 8 == 0 && y_row_stride % 8 == 0 && y_px_stride % 8 == 0 && y_px_stride >= (split ? 3 : 1) * C && aligned16(x) && aligned16(y) &&
                  aligned16(gamma) && aligned16(a) && aligned16(b),
              X2V_E_ALIGN, "vae_prep_f16: 16-byte alignment (strides multiples of 8 halves, pixel stride >= the channels written)");
  const int64_t npix = (int64_t)T * Hh * Ww;
  hipStream_t st = (hipStream_t)stream;
  auto launch = [&](auto lppc) {
    constexpr int LPP = decltype(lppc)::value;
    const int64_t blocks = std::min<int64_t>((npix + 256 / LPP - 1) / (256 / LPP), 65536 * 4);
    hipLaunchKernelGGL((vae_prep_kernel<LPP, _Float16>), dim3((unsigned)blocks), dim3(256), 0, st, x, (_Float16*)y, npix, Hh, Ww, C, gamma, a, b, silu, upsample,
                       y_frame_stride, y_row_stride, y_px_stride, split);
  };
  // lanes per pixel: a lane takes up to four 4-channel chunks LPP apart.  The decoder's widths are 3 x 2^n chunks (96 / 192 / 384 channels = 24 / 48 / 96):
  // a third of them per pixel keeps every lane busy (the power-of-two choice below idles a quarter of each wave)
  const int nch = C / 4;
  static const bool thirds = [] { const char* e = getenv("X2V_VAE_PREP_POW2"); return e == nullptr || atoi(e) == 0; }();  // A/B: 1 = the power-of-two lane groups
  if (thirds && nch == 24) launch(std::integral_constant<int, 8>{});
  else if (thirds && nch == 48) launch(std::integral_constant<int, 16>{});
  else if (thirds && nch == 96) launch(std::integral_constant<int, 32>{});
  else if (C <= 128) launch(std::integral_constant<int, 32>{});
  else launch(std::integral_constant<int, 64>{});
  X2V_LAUNCH_CHECK("vae_prep_f16 launch");
  return X2V_OK;
}

extern "C" __attribute__((visibility("default"))) int x2v_vae_prep_f16(const float* x, void* y, int T, int Hh, int Ww, int C, const float* gamma, const float* a,
                                                                       const float* b, int silu, int upsample, int64_t y_frame_stride, int64_t y_row_stride,
                                                                       int64_t y_px_stride, void* stream) {
  return vae_prep_f16_impl(x, y, T, Hh, Ww, C, gamma, a, b, silu, upsample, y_frame_stride, y_row_stride, y_px_stride, 0, stream);
}

// The same pass writing the hi/lo split of its result, channels [hi | hi * 2^-12 | lo] (3 C halves per pixel): operand of x2v_vae_conv_f16 with weights
// [hi | lo * 2^12 | hi] — fp32-grade convolution (~22 mantissa bits per operand) on the 16-bit matrix instruction
extern "C" __attribute__((visibility("default"))) int x2v_vae_prep_split_f16(const float* x, void* y, int T, int Hh, int Ww, int C, const float* gamma, const float* a,
                                                                             const float* b, int silu, int upsample, int64_t y_frame_stride, int64_t y_row_stride,
                                                                             int64_t y_px_stride, void* stream) {
  return vae_prep_f16_impl(x, y, T, Hh, Ww, C, gamma, a, b, silu, upsample, y_frame_stride, y_row_stride, y_px_stride, 1, stream);
}

// fp32 in, fp16 out: the operand buffer of x2v_vae_conv_f16 (strides in halves)
extern "C" __attribute__((visibility("default"))) int x2v_vae_prep_ex_f16(const float* x, void* y, int T, int Hh, int Ww, int C, const float* mul, const float* add, int silu,
                                                                          int clamp01, int up_hw, int up_t, int64_t y_frame_stride, int64_t y_row_stride, void* stream) {
  X2V_REQUIRE(x && y, X2V_E_ARG, "vae_prep_ex_f16: null pointer");
  X2V_REQUIRE(T > 0 && Hh > 0 && Ww > 0 && C > 0 && C % 8 == 0, X2V_E_SHAPE, "vae_prep_ex_f16: bad shape (C %% 8 == 0)");
  X2V_REQUIRE(y_frame_stride % 8 == 0 && y_row_stride % 8 == 0 && aligned16(x) && aligned16(y) && aligned16(mul) && aligned16(add), X2V_E_ALIGN,
              "vae_prep_ex_f16: 16-byte alignment");
  const int64_t npix = (int64_t)T * Hh * Ww;
  hipStream_t st = (hipStream_t)stream;
  if (C <= 128) {
    const int64_t blocks = std::min<int64_t>((npix + 7) / 8, 65536 * 4);
    hipLaunchKernelGGL((vae_prep_ex_kernel<32, _Float16>), dim3((unsigned)blocks), dim3(256), 0, st, x, (_Float16*)y, npix, Hh, Ww, C, mul, add, silu, clamp01, up_hw, up_t,
                       y_frame_stride, y_row_stride);
  } else {
    const int64_t blocks = std::min<int64_t>((npix + 3) / 4, 65536 * 4);
    hipLaunchKernelGGL((vae_prep_ex_kernel<64, _Float16>), dim3((unsigned)blocks), dim3(256), 0, st, x, (_Float16*)y, npix, Hh, Ww, C, mul, add, silu, clamp01, up_hw, up_t,
                       y_frame_stride, y_row_stride);
  }
  X2V_LAUNCH_CHECK("vae_prep_ex_f16 launch");
  return X2V_OK;
}

template <int NF>
static int launch_vconv16(const void* xp, int64_t fs, int64_t rs, int64_t ps, const void* w, int64_t wrs, const float* bias, const float* resid, float* y, int T, int Hh,
                          int Ww, int Cin, int Cout, int kt, int kh, int kw, int flags, hipStream_t st) {
  constexpr int lds = 2 * (VC_PIX + 32 * NF) * 128;
  {
    int rc = ensure_dynamic_lds((const void*)vae_conv16_kernel<NF>, lds, "vae conv16 attr");
    if (rc != X2V_OK) return rc;
  }
  const int64_t ptiles = (int64_t)T * (((int64_t)Hh * Ww + VC_PIX - 1) / VC_PIX);
  const int ncol = (Cout + 32 * NF - 1) / (32 * NF);
  X2V_REQUIRE(ptiles * ncol < (1ll << 31), X2V_E_SHAPE, "vae_conv_f16: too many tiles");
  hipLaunchKernelGGL((vae_conv16_kernel<NF>), dim3((unsigned)(ptiles * ncol)), dim3(256), lds, st, (const _Float16*)xp, fs, rs, ps, (const _Float16*)w, wrs, bias, resid,
                     y, T, Hh, Ww, Cin, Cout, kt, kh, kw, flags, ncol);
  X2V_LAUNCH_CHECK("vae_conv_f16 launch");
  return X2V_OK;
}

template <int NF>
static int launch_vconv16h(const void* xp, int64_t fs, int64_t rs, int64_t ps, const void* w, int64_t wrs, const float* bias, const float* resid, float* y, int T, int Hh,
                           int Ww, int Cin, int Cout, int kt, int flags, hipStream_t st) {
  constexpr int lds = 2 * VH_A_BYTES + VH_NB * 32 * NF * 128;  // NF = 4, ring of 4: 152 KiB of the CU's 160
  {
    int rc = ensure_dynamic_lds((const void*)vae_conv16h_kernel<NF>, lds, "vae conv16h attr");
    if (rc != X2V_OK) return rc;
  }
  const int tiles_x = (Ww + VH_TW - 1) / VH_TW, tiles_y = (Hh + VH_TH - 1) / VH_TH;
  const int ncol = (Cout + 32 * NF - 1) / (32 * NF);
  const int64_t blocks = (int64_t)T * tiles_x * tiles_y * ncol;
  X2V_REQUIRE(blocks < (1ll << 31), X2V_E_SHAPE, "vae_conv_f16: too many tiles");
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return check_hip(hipErrorUnknown, "vae conv16h device query");
    n_cu = prop.multiProcessorCount;
  }
  const unsigned grid = (unsigned)std::min<int64_t>(blocks, n_cu);  // one workgroup per CU (120-152 KiB of LDS each)
  hipLaunchKernelGGL((vae_conv16h_kernel<NF>), dim3(grid), dim3(512), lds, st, (const _Float16*)xp, fs, rs, ps, (const _Float16*)w, wrs, bias, resid, y, T,
                     Hh, Ww, Cin, Cout, kt, flags, ncol, tiles_x, tiles_y);
  X2V_LAUNCH_CHECK("vae_conv_f16 (halo) launch");
  return X2V_OK;
}

static int vae_conv_f16_impl(const void* xp, const void* cache, int64_t x_frame_stride, int64_t x_row_stride, int64_t x_px_stride, const void* w, int64_t w_row_stride,
                             const float* bias, const float* resid, float* y, int T, int Hh, int Ww, int Cin, int Cout, int kt, int kh, int kw, int flags, void* stream);

extern "C" __attribute__((visibility("default"))) int x2v_vae_conv_f16(const void* xp, int64_t x_frame_stride, int64_t x_row_stride, int64_t x_px_stride, const void* w,
                                                                       int64_t w_row_stride, const float* bias, const float* resid, float* y, int T, int Hh, int Ww,
                                                                       int Cin, int Cout, int kt, int kh, int kw, int flags, void* stream) {
  return vae_conv_f16_impl(xp, nullptr, x_frame_stride, x_row_stride, x_px_stride, w, w_row_stride, bias, resid, y, T, Hh, Ww, Cin, Cout, kt, kh, kw, flags, stream);
}

// 1 if x2v_vae_conv_f16_cached takes this shape (the 128-pixel kernel of vae16g.hip does), else 0: the caller then puts the cache frames in front of xp itself.
extern "C" __attribute__((visibility("default"))) int x2v_vae_conv_f16_cached_ok(int Ww, int Cin, int Cout, int kh, int kw, int flags) {
  return (kh == 3 && kw == 3 && !(flags & (VCF_TSPLIT | 4 | 8)) && vae_conv16g_ok(Ww, Cin, Cout)) ? 1 : 0;
}

extern "C" __attribute__((visibility("default"))) int x2v_vae_conv_f16_cached(const void* xp, const void* cache, int64_t x_frame_stride, int64_t x_row_stride,
                                                                              int64_t x_px_stride, const void* w, int64_t w_row_stride, const float* bias, const float* resid,
                                                                              float* y, int T, int Hh, int Ww, int Cin, int Cout, int kt, int kh, int kw, int flags, void* stream) {
  X2V_REQUIRE(cache != nullptr && aligned16(cache), X2V_E_ARG, "vae_conv_f16_cached: cache must be a 16-byte aligned pointer");
  X2V_REQUIRE(x2v_vae_conv_f16_cached_ok(Ww, Cin, Cout, kh, kw, flags) == 1, X2V_E_SHAPE,
              "vae_conv_f16_cached: 3x3 kernels with Cout %% 96 == 0 or Cout <= 16 only (x2v_vae_conv_f16_cached_ok)");
  return vae_conv_f16_impl(xp, cache, x_frame_stride, x_row_stride, x_px_stride, w, w_row_stride, bias, resid, y, T, Hh, Ww, Cin, Cout, kt, kh, kw, flags, stream);
}

static int vae_conv_f16_impl(const void* xp, const void* cache, int64_t x_frame_stride, int64_t x_row_stride, int64_t x_px_stride, const void* w, int64_t w_row_stride,
                             const float* bias, const float* resid, float* y, int T, int Hh, int Ww, int Cin, int Cout, int kt, int kh, int kw, int flags, void* stream) {
  X2V_REQUIRE(xp && w && y, X2V_E_ARG, "vae_conv_f16: null pointer");
  X2V_REQUIRE(T > 0 && Hh > 0 && Ww > 0 && Cin > 0 && Cout > 0, X2V_E_SHAPE, "vae_conv_f16: bad shape");
  X2V_REQUIRE(kt >= 1 && kt <= 3 && kh >= 1 && kh <= 3 && kw >= 1 && kw <= 3, X2V_E_SHAPE, "vae_conv_f16: kernel %dx%dx%d unsupported", kt, kh, kw);
  X2V_REQUIRE(Cin % 64 == 0, X2V_E_SHAPE, "vae_conv_f16: Cin=%d must be a multiple of 64 (one 128-byte row per K step)", Cin);
  X2V_REQUIRE(x_px_stride % 8 == 0 && x_row_stride % 8 == 0 && x_frame_stride % 8 == 0 && w_row_stride % 8 == 0 && x_px_stride >= Cin &&
                  w_row_stride >= (int64_t)kt * kh * kw * Cin,
              X2V_E_ALIGN, "vae_conv_f16: strides must be multiples of 8 halves and cover the extents");
  X2V_REQUIRE(aligned16(xp) && aligned16(w) && aligned16(y) && aligned16(resid) && aligned16(bias), X2V_E_ALIGN, "vae_conv_f16: pointers must be 16-byte aligned");
  X2V_REQUIRE(x_frame_stride * 2 * kt < (1ll << 31) && w_row_stride * 2 * 128 < (1ll << 31), X2V_E_SHAPE,
              "vae_conv_f16: a kt-frame input window / 128 weight rows must stay below 2 GiB (32-bit buffer offsets)");
  X2V_REQUIRE(!(flags & VCF_TSPLIT) || (Cout % 8 == 0 && resid == nullptr), X2V_E_ARG, "vae_conv_f16: time-split output needs Cout %% 8 == 0 and no residual");
  hipStream_t st = (hipStream_t)stream;
  X2V_REQUIRE((flags & ~31) == 0 && (!(flags & 16) || Cin >= 64), X2V_E_ARG, "vae_conv_f16: flags = 1 clamp | 2 time-split | 4 per-tap kernel | 8 64-pixel halo kernel | 16 zero tail");
  // 3x3 spatial kernels take a halo-tiled kernel (2.3x fewer L2 bytes per FLOP than a fresh pixel block per tap) unless the image is narrower than half a tile
  // or the caller asks for the per-tap kernel (flag 4: A/B measurements and tests): the 128-pixel x 96-cout kernel of vae16g.hip where Cout is a multiple of 96
  // (every 3x3 convolution of the Wan decoder but its 3-channel head), else — or with flag 8 — the 64-pixel kernel below.
  // Flag 16: the last 32 channels of Cin are zero padding in both operands (the split mode's 3 x 96 = 288 channels in a 320-channel buffer); the 32-channel-slab
  // kernel skips them, the others multiply the zeros.
  if (kh == 3 && kw == 3 && !(flags & (VCF_TSPLIT | 4 | 8)) && vae_conv16g_ok(Ww, Cin, Cout))
    return vae_conv16g_dispatch(xp, cache, x_frame_stride, x_row_stride, x_px_stride, w, w_row_stride, bias, resid, y, T, Hh, Ww, Cin, Cout, kt, flags, (flags & 16) ? 32 : 0, st);
  if (kh == 3 && kw == 3 && !(flags & (VCF_TSPLIT | 4)) && Ww >= 16) {
#define X2V_VC16H(NF_) return launch_vconv16h<NF_>(xp, x_frame_stride, x_row_stride, x_px_stride, w, w_row_stride, bias, resid, y, T, Hh, Ww, Cin, Cout, kt, flags, st)
    if (Cout <= 32) X2V_VC16H(1);
    else if (Cout % 128 != 0 && Cout % 96 == 0) X2V_VC16H(3);
    else if (Cout <= 64) X2V_VC16H(2);
    else X2V_VC16H(4);
#undef X2V_VC16H
  }
#define X2V_VC16(NF_) return launch_vconv16<NF_>(xp, x_frame_stride, x_row_stride, x_px_stride, w, w_row_stride, bias, resid, y, T, Hh, Ww, Cin, Cout, kt, kh, kw, flags, st)
  if (Cout <= 32) X2V_VC16(1);
  else if (Cout % 128 != 0 && Cout % 96 == 0) X2V_VC16(3);
  else if (Cout <= 64) X2V_VC16(2);
  else X2V_VC16(4);
#undef X2V_VC16
}

extern "C" __attribute__((visibility("default"))) int x2v_vae_replicate_border_f32(float* buf, int frames, int lead, int Hp, int Wp, int C, int pad, void* stream) {
  X2V_REQUIRE(buf, X2V_E_ARG, "vae_replicate_border: null pointer");
  X2V_REQUIRE(frames > lead && lead >= 0 && pad >= 0 && Hp > 2 * pad && Wp > 2 * pad && C > 0 && C % 4 == 0, X2V_E_SHAPE, "vae_replicate_border: bad shape");
  X2V_REQUIRE(aligned16(buf), X2V_E_ALIGN, "vae_replicate_border: 16-byte alignment");
  const int64_t total = (int64_t)frames * Hp * Wp * (C / 4);
  const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 65536 * 4);
  hipLaunchKernelGGL(vae_replicate_border_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, buf, frames, lead, Hp, Wp, C, pad);
  X2V_LAUNCH_CHECK("vae_replicate_border launch");
  return X2V_OK;
}

extern "C" __attribute__((visibility("default"))) int x2v_groupnorm_affine_f32(const float* x, int64_t npix, int C, int G, const float* gamma, const float* beta, float eps,
                                                                               double* workspace, float* mul, float* add, void* stream) {
  X2V_REQUIRE(x && gamma && beta && workspace && mul && add, X2V_E_ARG, "groupnorm_affine: null pointer");
  X2V_REQUIRE(npix > 0 && C > 0 && G > 0 && G <= 64 && C % G == 0 && (C / G) % 4 == 0, X2V_E_SHAPE, "groupnorm_affine: C=%d G=%d (C/G %% 4 == 0, G <= 64)", C, G);
  X2V_REQUIRE(aligned16(x), X2V_E_ALIGN, "groupnorm_affine: 16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  int rc = check_hip(hipMemsetAsync(workspace, 0, sizeof(double) * 2 * G, st), "groupnorm memset");
  if (rc != X2V_OK) return rc;
  const int64_t total = npix * (C / 4);
  const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 2048);
  hipLaunchKernelGGL(groupnorm_stats_kernel, dim3(grid), dim3(256), 0, st, x, npix, C, G, workspace);
  hipLaunchKernelGGL(groupnorm_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, workspace, gamma, beta, mul, add, C, G, (double)npix * (C / G), eps);
  X2V_LAUNCH_CHECK("groupnorm_affine launch");
  return X2V_OK;
}

extern "C" __attribute__((visibility("default"))) int x2v_softmax_rows_causal_f32(float* s, int64_t ld, int64_t M, int N, float scale, int hw, int n_keys, void* stream) {
  X2V_REQUIRE(s, X2V_E_ARG, "softmax_rows_causal: null pointer");
  X2V_REQUIRE(M > 0 && N > 0 && hw > 0 && ld >= N && n_keys > 0 && n_keys <= N && M < (1ll << 31), X2V_E_SHAPE, "softmax_rows_causal: bad shape");
  hipLaunchKernelGGL(softmax_rows_causal_kernel, dim3((unsigned)M), dim3(256), 0, (hipStream_t)stream, s, ld, N, scale, hw, n_keys);
  X2V_LAUNCH_CHECK("softmax_rows_causal launch");
  return X2V_OK;
}

extern "C" __attribute__((visibility("default"))) int x2v_blend_axis_f32(const float* a, float* b, int64_t outer, int na, int nb, int64_t inner, int64_t a_outer_stride,
                                                                         int64_t b_outer_stride, int extent, void* stream) {
  X2V_REQUIRE(a && b, X2V_E_ARG, "blend_axis: null pointer");
  X2V_REQUIRE(outer > 0 && inner > 0 && extent >= 0 && extent <= na && extent <= nb, X2V_E_SHAPE, "blend_axis: bad shape");
  if (extent == 0) return X2V_OK;
  const int64_t total = outer * extent * inner;
  const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 65536 * 4);
  hipLaunchKernelGGL(blend_axis_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, b, outer, na, nb, inner, a_outer_stride, b_outer_stride, extent);
  X2V_LAUNCH_CHECK("blend_axis launch");
  return X2V_OK;
}
