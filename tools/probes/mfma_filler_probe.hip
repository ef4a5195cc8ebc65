// Microbenchmark: cycles per v_mfma_f32_32x32x16_bf16 for ONE wave per SIMD when each MFMA gap carries E independent v_exp_f32, A independent
// v_add_f32, C v_cvt_pk_bf16_f32 and L ds_read_b128 (asm MFMAs and fillers: nothing is reordered or padded by the compiler).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
template <int E, int A, int C, int L, int LAG = 0>
__global__ __launch_bounds__(256) void k(const bf16x8_t* in, float* out, unsigned long long* cyc, int iters) {
  __shared__ f32x4_t lds[1024];
  lds[threadIdx.x] = f32x4_t{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  bf16x8_t a = in[threadIdx.x], b = in[threadIdx.x + 256];
  f32x16_t acc[8];
  for (int c = 0; c < 8; ++c)
    for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
  float x[8], y[8], z[4] = {0, 0, 0, 0}, w[4] = {0, 0, 0, 0};
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.001f + i; y[i] = i; }
  f32x4_t ld[2];
  const unsigned lp = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(&lds[0]) + (threadIdx.x & 255) * 16;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b));
#pragma unroll
      for (int f = 0; f < E; ++f) asm volatile("v_exp_f32 %0, %1" : "=v"(y[(2 * c + f) & 7]) : "v"(x[(2 * c + f) & 7]));
#pragma unroll
      for (int f = 0; f < A; ++f) {
        if (LAG) asm volatile("v_add_f32 %0, %0, %1" : "+v"(z[f & 3]) : "v"(y[(2 * c + f + 6) & 7]));  // consumes the PREVIOUS gap's exp results
        else asm volatile("v_add_f32 %0, %0, %1" : "+v"(z[f & 3]) : "v"(y[(2 * c + f) & 7]));           // consumes this gap's exp results at once
      }
#pragma unroll
      for (int f = 0; f < C; ++f) {
        if (LAG) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[c & 3]) : "v"(y[(2 * c + 6) & 7]), "v"(y[(2 * c + 7) & 7]));
        else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[c & 3]) : "v"(y[(2 * c) & 7]), "v"(y[(2 * c + 1) & 7]));
      }
      if (L > 0 && (c % (2 / (L > 2 ? 2 : L))) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[c & 1]) : "v"(lp + (c & 3) * 4096));
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
  }
  asm volatile("s_nop 7\n\ts_nop 7");
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += x[i] + y[i];
  for (int i = 0; i < 4; ++i) s += z[i] + w[i];
  for (int c = 0; c < 8; ++c) s += acc[c][0];
  s += ld[0][0] + ld[1][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int E, int A, int C, int L, int LAG = 0>
void run(const bf16x8_t* in, float* out, unsigned long long* cyc) {
  const int iters = 2000;
  k<E, A, C, L, LAG><<<256, 256>>>(in, out, cyc, iters);
  (void)hipDeviceSynchronize();
  unsigned long long h;
  (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("per MFMA: exp=%d add=%d cvt=%d lag=%d lds=%.1f : %.1f ticks per MFMA\n", E, A, C, LAG, L == 0 ? 0.0 : (L == 1 ? 0.5 : 1.0), (double)h / (iters * 8.0));
}
int main() {
  bf16x8_t* in; float* out; unsigned long long* cyc;
  (void)hipMalloc(&in, 512 * 16); (void)hipMemset(in, 0x3c, 512 * 16); (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 8);
  run<0, 0, 0, 0>(in, out, cyc);
  run<1, 0, 0, 0>(in, out, cyc); run<2, 0, 0, 0>(in, out, cyc); run<3, 0, 0, 0>(in, out, cyc); run<4, 0, 0, 0>(in, out, cyc);
  run<0, 2, 0, 0>(in, out, cyc); run<0, 4, 0, 0>(in, out, cyc); run<0, 6, 0, 0>(in, out, cyc);
  run<2, 2, 0, 0>(in, out, cyc); run<2, 2, 1, 0>(in, out, cyc); run<2, 2, 1, 1>(in, out, cyc); run<2, 1, 1, 1>(in, out, cyc); run<2, 3, 1, 1>(in, out, cyc);
  run<0, 0, 0, 1>(in, out, cyc); run<0, 0, 0, 2>(in, out, cyc);
  run<2, 2, 1, 0, 1>(in, out, cyc); run<2, 2, 1, 1, 1>(in, out, cyc); run<2, 2, 0, 0, 1>(in, out, cyc); run<1, 1, 1, 1, 1>(in, out, cyc); run<1, 1, 0, 1, 1>(in, out, cyc);
  return 0;
}
