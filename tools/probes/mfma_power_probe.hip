// Microbenchmark: SUSTAINED matrix throughput of the whole chip (256 workgroups, seconds long, i.e. at the board's power limit) for
// inner loops that differ only in MFMA shape, per-wave tile and LDS fragment traffic.  The denoise step's GEMM and attention kernels
// run at the 1400 W cap with the engine clock throttled (profiles/r01_power_clock_probe.txt), so what sets their speed is energy per
// FLOP, not issue slots; this probe prices the candidates:
//   SHAPE 0 = v_mfma_f32_32x32x16_bf16, 1 = v_mfma_f32_16x16x32_bf16
//   NA x NB fragments per wave and k-step (wave tile = NA*R x NB*R rows, R = 32 or 16)
//   MODE 0 = operands stay in registers (no LDS), 1 = A and B fragments from LDS every k-step (GEMM), 2 = only B from LDS (attention:
//          Q / P live in registers)
//   NW waves per workgroup (one workgroup per CU: 4 = one wave per SIMD, 8 = two)
// LDS holds random bf16 (data toggling is part of the power).  Prints TFLOP/s over the last second of a ~2.5 s run.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

template <int SHAPE>
struct Acc { using type = f32x16_t; };
template <>
struct Acc<1> { using type = f32x4_t; };

template <int SHAPE, int NA, int NB, int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void k(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 4 slots x {A 16 KiB | B 16 KiB}
  for (int i = threadIdx.x; i < 131072 / 16; i += NW * 64) reinterpret_cast<uint4*>(smem)[i] = src[i];
  __syncthreads();
  using acc_t = typename Acc<SHAPE>::type;
  constexpr int KS = SHAPE == 0 ? 4 : 2;      // k-steps per 64-wide K tile
  constexpr int RB = SHAPE == 0 ? 4096 : 2048;  // bytes per fragment row block ([R rows][128 B])
  const int lane = threadIdx.x & 63;
  int row, chunk0;
  if (SHAPE == 0) { row = lane & 31; chunk0 = lane >> 5; } else { row = lane & 15; chunk0 = lane >> 4; }
  const int swz = (row >> 1) & 7;
  int rd[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) rd[ks] = row * 128 + (((chunk0 + ks * (SHAPE == 0 ? 2 : 4)) ^ swz) << 4);
  acc_t acc[NA][NB];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int e = 0; e < (SHAPE == 0 ? 16 : 4); ++e) acc[i][j][e] = 0.f;
  bf16x8_t fa[2][NA], fb[2][NB];
#define RD_A(BUF_, SLOT_, KS_) _Pragma("unroll") for (int i = 0; i < NA; ++i) fa[BUF_][i] = *reinterpret_cast<const bf16x8_t*>(smem + (SLOT_) * 32768 + ((i * RB) & 16383) + rd[KS_]);
#define RD_B(BUF_, SLOT_, KS_) _Pragma("unroll") for (int j = 0; j < NB; ++j) fb[BUF_][j] = *reinterpret_cast<const bf16x8_t*>(smem + (SLOT_) * 32768 + 16384 + ((j * RB) & 16383) + rd[KS_]);
  RD_A(0, 0, 0) RD_B(0, 0, 0) RD_A(1, 1, 1) RD_B(1, 1, 1)
  for (int it = 0; it < iters; ++it) {
    const int slot = it & 3, nslot = (it + 1) & 3;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (MODE == 1) {
        if (ks + 1 < KS) { RD_A(nxt, slot, ks + 1) } else { RD_A(nxt, nslot, 0) }
      }
      if (MODE >= 1) {
        if (ks + 1 < KS) { RD_B(nxt, slot, ks + 1) } else { RD_B(nxt, nslot, 0) }
      }
      const int ca = MODE == 1 ? cur : 0, cb = MODE >= 1 ? cur : 0;
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          // register-resident operands rotate so that consecutive MFMAs never repeat an operand pair
          const bf16x8_t av = fa[MODE == 1 ? ca : ((i + ks) & 1)][MODE == 1 ? i : (i + ks) % NA];
          const bf16x8_t bv = fb[MODE >= 1 ? cb : ((j + ks) & 1)][MODE >= 1 ? j : (j + ks) % NB];
          if constexpr (SHAPE == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv, av, acc[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv, av, acc[i][j], 0, 0, 0);
        }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) s += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * (NW * 64) + threadIdx.x] = s;
}

template <int SHAPE, int NA, int NB, int MODE, int NW>
void run(const uint4* src, float* out, const char* what) {
  auto kern = k<SHAPE, NA, NB, MODE, NW>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  constexpr int KS = SHAPE == 0 ? 4 : 2;
  const double flop_per_iter = 256.0 * NW * KS * NA * NB * (SHAPE == 0 ? 32.0 * 32 * 16 : 16.0 * 16 * 32) * 2.0;
  const int iters = (int)(2.0e13 / flop_per_iter) + 1;  // ~15-20 ms per launch at 1-1.4 PFLOP/s
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float ms = 0.f;
  int n = 0;
  // warm: ~1.5 s; timed: ~1 s
  for (int phase = 0; phase < 2; ++phase) {
    const double budget = phase == 0 ? 1500.0 : 1000.0;
    double spent = 0.0;
    n = 0;
    (void)hipEventRecord(e0, 0);
    while (spent < budget) {
      for (int r = 0; r < 8; ++r) kern<<<256, NW * 64, 131072, 0>>>(src, out, iters);
      n += 8;
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
      spent = ms;
    }
  }
  const double tf = flop_per_iter * iters * n / (ms * 1e-3) / 1e12;
  const double reads = MODE == 0 ? 0.0 : (MODE == 1 ? (double)(NA + NB) : (double)NB) / (NA * NB);
  printf("%-46s %s NA=%d NB=%d waves/CU=%d lds_reads/MFMA=%.3f (KiB/MFMA-cycle-equiv %.3f): %7.1f TFLOP/s  = busy x clock %.3f GHz\n", what,
         SHAPE == 0 ? "32x32x16" : "16x16x32", NA, NB, NW, reads, reads * (SHAPE == 0 ? 1.0 : 2.0), tf, tf / 2500.0 * 2.4);
  fflush(stdout);
  if (hipGetLastError() != hipSuccess) printf("  (launch error)\n");
}

// fp8 (e4m3, unit block scales): SHAPE 0 = v_mfma_scale_f32_32x32x64_f8f6f4, 1 = v_mfma_scale_f32_16x16x128_f8f6f4; a fragment = 32 bytes per lane
typedef int i32x8_t __attribute__((ext_vector_type(8)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));
template <int SHAPE, int NA, int NB, int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void k8(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < 131072 / 16; i += NW * 64) reinterpret_cast<uint4*>(smem)[i] = src[i];
  __syncthreads();
  using acc_t = typename Acc<SHAPE>::type;
  constexpr int KS = SHAPE == 0 ? 2 : 1;        // k-steps per 128-wide K tile (one 128-byte row)
  constexpr int RB = SHAPE == 0 ? 4096 : 2048;
  const int lane = threadIdx.x & 63;
  int row, g;
  if (SHAPE == 0) { row = lane & 31; g = lane >> 5; } else { row = lane & 15; g = lane >> 4; }
  const int swz = (row >> 1) & 7;
  int rd[KS][2];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int h = 0; h < 2; ++h) rd[ks][h] = row * 128 + ((((SHAPE == 0 ? ks * 4 + g * 2 : g * 2) + h) ^ swz) << 4);
  acc_t acc[NA][NB];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int e = 0; e < (SHAPE == 0 ? 16 : 4); ++e) acc[i][j][e] = 0.f;
  i32x8_t fa[2][NA], fb[2][NB];
  auto rdf = [&](int base, int ks) {
    const i32x4_t lo = *reinterpret_cast<const i32x4_t*>(smem + base + rd[ks][0]), hi = *reinterpret_cast<const i32x4_t*>(smem + base + rd[ks][1]);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
#define RD8_A(BUF_, SLOT_, KS_) _Pragma("unroll") for (int i = 0; i < NA; ++i) fa[BUF_][i] = rdf((SLOT_) * 32768 + ((i * RB) & 16383), KS_);
#define RD8_B(BUF_, SLOT_, KS_) _Pragma("unroll") for (int j = 0; j < NB; ++j) fb[BUF_][j] = rdf((SLOT_) * 32768 + 16384 + ((j * RB) & 16383), KS_);
  RD8_A(0, 0, 0) RD8_B(0, 0, 0) RD8_A(1, 1, KS - 1) RD8_B(1, 1, KS - 1)
  for (int it = 0; it < iters; ++it) {
    const int slot = it & 3, nslot = (it + 1) & 3;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int cur = (KS == 1 ? it : ks) & 1, nxt = cur ^ 1;
      if (MODE == 1) {
        if (ks + 1 < KS) { RD8_A(nxt, slot, ks + 1) RD8_B(nxt, slot, ks + 1) } else { RD8_A(nxt, nslot, 0) RD8_B(nxt, nslot, 0) }
      }
      const int c = MODE == 1 ? cur : 0;
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          const i32x8_t av = fa[MODE == 1 ? c : ((i + ks) & 1)][MODE == 1 ? i : (i + ks) % NA];
          const i32x8_t bv = fb[MODE == 1 ? c : ((j + ks) & 1)][MODE == 1 ? j : (j + ks) % NB];
          if constexpr (SHAPE == 0) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bv, av, acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
          else acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bv, av, acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
    }
  }
  float sacc = 0.f;
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) sacc += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * (NW * 64) + threadIdx.x] = sacc;
}

template <int SHAPE, int NA, int NB, int MODE, int NW>
void run8(const uint4* src, float* out, const char* what) {
  auto kern = k8<SHAPE, NA, NB, MODE, NW>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  constexpr int KS = SHAPE == 0 ? 2 : 1;
  const double flop_per_iter = 256.0 * NW * KS * NA * NB * (SHAPE == 0 ? 32.0 * 32 * 64 : 16.0 * 16 * 128) * 2.0;
  const int iters = (int)(4.0e13 / flop_per_iter) + 1;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float ms = 0.f;
  int n = 0;
  for (int phase = 0; phase < 2; ++phase) {
    const double budget = phase == 0 ? 1500.0 : 1000.0;
    double spent = 0.0;
    n = 0;
    (void)hipEventRecord(e0, 0);
    while (spent < budget) {
      for (int r = 0; r < 8; ++r) kern<<<256, NW * 64, 131072, 0>>>(src, out, iters);
      n += 8;
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
      spent = ms;
    }
  }
  const double tf = flop_per_iter * iters * n / (ms * 1e-3) / 1e12;
  printf("%-46s %s NA=%d NB=%d waves/CU=%d %s: %7.1f TFLOP/s  = busy x clock %.3f GHz\n", what, SHAPE == 0 ? "fp8 32x32x64 " : "fp8 16x16x128", NA, NB, NW,
         MODE == 1 ? "fragments from LDS" : "operands in registers", tf, tf / 5000.0 * 2.4);
  fflush(stdout);
}

int main() {
  uint4* src;
  float* out;
  (void)hipMalloc(&src, 131072);
  (void)hipMalloc(&out, 256 * 512 * 4);
  unsigned short* h = (unsigned short*)malloc(131072);
  srand(1);
  for (int i = 0; i < 65536; ++i) {  // bf16 values in (-2, 2) with random mantissas
    const unsigned sign = (rand() & 1) << 15, exp = 120 + (rand() % 8), man = rand() & 127;
    h[i] = (unsigned short)(sign | (exp << 7) | man);
  }
  (void)hipMemcpy(src, h, 131072, hipMemcpyHostToDevice);
  if (getenv("PROBE_SET") == nullptr || atoi(getenv("PROBE_SET")) == 0) {
  run<0, 4, 2, 1, 8>(src, out, "gemm256 today (wave tile 128x64)");
  run<0, 4, 2, 0, 8>(src, out, "  same, operands in registers");
  run<0, 4, 4, 1, 4>(src, out, "wave tile 128x128, one wave per SIMD");
  run<0, 4, 4, 0, 4>(src, out, "  same, operands in registers");
  run<1, 8, 8, 1, 4>(src, out, "hipBLASLt-like (MI16x16, wave tile 128x128)");
  run<1, 8, 8, 0, 4>(src, out, "  same, operands in registers");
  run<1, 8, 4, 1, 8>(src, out, "MI16x16, wave tile 128x64, two waves per SIMD");
  run<0, 1, 2, 2, 8>(src, out, "attention ping-pong (1 read per MFMA)");
  run<0, 2, 2, 2, 4>(src, out, "attention 64 rows per wave (0.5 per MFMA)");
  run<0, 2, 4, 2, 8>(src, out, "attention 64 rows, two waves per SIMD");
  run<0, 1, 2, 0, 8>(src, out, "  attention shape, operands in registers");
  } else if (atoi(getenv("PROBE_SET")) == 2) {
  run8<0, 4, 2, 1, 8>(src, out, "fp8 gemm256 today (wave tile 128x64)");
  run8<1, 8, 4, 1, 8>(src, out, "fp8 MI16x16x128, wave tile 128x64");
  run8<0, 4, 2, 0, 8>(src, out, "  32x32x64, operands in registers");
  run8<1, 8, 4, 0, 8>(src, out, "  16x16x128, operands in registers");
  run8<0, 4, 2, 1, 8>(src, out, "fp8 gemm256 today again (drift check)");
  } else {
  run<0, 4, 2, 1, 8>(src, out, "gemm256 today (wave tile 128x64)");
  run<1, 8, 4, 1, 8>(src, out, "MI16x16, wave tile 128x64, two waves per SIMD");
  run<1, 8, 4, 0, 8>(src, out, "  same, operands in registers");
  run<0, 4, 2, 0, 8>(src, out, "  32x32, operands in registers");
  run<1, 4, 4, 1, 8>(src, out, "MI16x16, wave tile 64x64 (more LDS traffic)");
  run<0, 2, 2, 1, 8>(src, out, "32x32, wave tile 64x64 (more LDS traffic)");
  run<1, 2, 4, 2, 8>(src, out, "attention ping-pong with MI16x16 (32 q rows)");
  run<0, 1, 2, 2, 8>(src, out, "attention ping-pong 32x32 (1 read per MFMA)");
  run<1, 4, 4, 2, 8>(src, out, "attention MI16x16, 64 q rows per wave");
  run<0, 4, 2, 1, 8>(src, out, "gemm256 today again (drift check)");
  }
  return 0;
}
