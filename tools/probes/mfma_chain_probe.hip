// Microbenchmark: cycles per v_mfma_f32_32x32x16_bf16 when NC independent accumulator chains are issued round-robin by ONE wave per SIMD
// (asm MFMAs: no compiler-inserted wait states), optionally with F plain VALU fillers per MFMA.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
template <int NC, int F>
__global__ __launch_bounds__(256) void k(const bf16x8_t* in, float* out, unsigned long long* cyc, int iters) {
  bf16x8_t a = in[threadIdx.x], b = in[threadIdx.x + 256];
  f32x16_t acc[NC];
  for (int c = 0; c < NC; ++c)
    for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
  float f0 = threadIdx.x, f1 = 1.f;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16 / NC; ++r)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b));
#pragma unroll
        for (int f = 0; f < F; ++f) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f0) : "v"(f1));
      }
  }
  asm volatile("s_nop 7\n\ts_nop 7");
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = f0;
  for (int c = 0; c < NC; ++c) s += acc[c][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NC, int F>
void run(const bf16x8_t* in, float* out, unsigned long long* cyc) {
  const int iters = 2000;
  k<NC, F><<<256, 256>>>(in, out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long h;
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("chains=%d fillers/mfma=%d: %.1f ticks per MFMA\n", NC, F, (double)h / (iters * 16.0));
}
int main() {
  bf16x8_t* in; float* out; unsigned long long* cyc;
  hipMalloc(&in, 512 * 16); hipMemset(in, 0x3c, 512 * 16); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  run<1, 0>(in, out, cyc); run<2, 0>(in, out, cyc); run<4, 0>(in, out, cyc); run<8, 0>(in, out, cyc);
  run<2, 2>(in, out, cyc); run<2, 4>(in, out, cyc); run<8, 2>(in, out, cyc); run<8, 4>(in, out, cyc); run<8, 5>(in, out, cyc); run<8, 6>(in, out, cyc); run<8, 8>(in, out, cyc);
  return 0;
}
