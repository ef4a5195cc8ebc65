// Box calibration: what THIS board sustains on the matrix pipe right now, measured in-process by the benchmark that reports the roofline.
//
// MI355X boxes of one pool differ by several percent on the same binary (the 1400 W board limit decides the clock a matrix kernel is granted:
// DESIGN.md §4), so a roofline fraction quoted against the nominal 2.5 PFLOP/s cannot be compared between two runs unless both also say what
// the bare instruction reached on their box.  x2v_mfma_probe_bf16 runs `v_mfma_f32_16x16x32_bf16` back to back on every CU — operands in
// registers, no LDS, no memory traffic, 8 waves per CU — for a given number of milliseconds and returns the TFLOP/s of the second half of that
// time (the first half brings the board to its power-limited steady state).  It is the "16x16x32, operands in registers" row of
// tools/probes/mfma_power_probe.hip (2061 TFLOP/s on the round-2 box) exported through the C-ABI so that bench.py can call it before and after
// its timed region.  Not part of the reference's operator surface: measurement plumbing.
#include "x2v_common.h"

namespace x2v {

constexpr int PR_NA = 8, PR_NB = 4, PR_NW = 8;

__global__ __launch_bounds__(PR_NW * 64) void mfma_probe_kernel(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int lane = threadIdx.x & 63;
  bf16x8_t fa[PR_NA], fb[PR_NB];
#pragma unroll
  for (int i = 0; i < PR_NA; ++i) fa[i] = __builtin_bit_cast(bf16x8_t, src[(i * 64 + lane) & 1023]);
#pragma unroll
  for (int j = 0; j < PR_NB; ++j) fb[j] = __builtin_bit_cast(bf16x8_t, src[((PR_NA + j) * 64 + lane) & 1023]);
  f32x4_t acc[PR_NA][PR_NB];
#pragma unroll
  for (int i = 0; i < PR_NA; ++i)
#pragma unroll
    for (int j = 0; j < PR_NB; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < PR_NB; ++j)
#pragma unroll
        for (int i = 0; i < PR_NA; ++i)  // operand pairs rotate so that consecutive MFMAs never repeat one (data toggling is part of the power)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[(j + ks) % PR_NB], fa[(i + ks) % PR_NA], acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < PR_NA; ++i)
#pragma unroll
    for (int j = 0; j < PR_NB; ++j) s += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * (PR_NW * 64) + threadIdx.x] = s;
#endif
}

}  // namespace x2v

using namespace x2v;

extern "C" __attribute__((visibility("default"))) int x2v_mfma_probe_bf16(int milliseconds, float* tflops, void* stream) {
  X2V_REQUIRE(tflops != nullptr && milliseconds >= 10 && milliseconds <= 20000, X2V_E_ARG, "mfma_probe: 10 <= milliseconds <= 20000 and a result pointer");
  hipStream_t st = (hipStream_t)stream;
  int dev = 0, cus = 0;
  int rc = check_hip(hipGetDevice(&dev), "mfma_probe: hipGetDevice");
  if (rc != X2V_OK) return rc;
  rc = check_hip(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev), "mfma_probe: CU count");
  if (rc != X2V_OK) return rc;
  uint4* src = nullptr;
  float* out = nullptr;
  rc = check_hip(hipMalloc(&src, 1024 * sizeof(uint4)), "mfma_probe: hipMalloc");
  if (rc != X2V_OK) return rc;
  rc = check_hip(hipMalloc(&out, (size_t)cus * PR_NW * 64 * sizeof(float)), "mfma_probe: hipMalloc");
  if (rc != X2V_OK) { (void)hipFree(src); return rc; }
  {  // bf16 values in (-2, 2) with random mantissas (xorshift; the values only have to toggle bits)
    unsigned short h[8192];
    unsigned s = 0x9e3779b9u;
    for (int i = 0; i < 8192; ++i) {
      s ^= s << 13; s ^= s >> 17; s ^= s << 5;
      h[i] = (unsigned short)(((s & 1u) << 15) | ((120u + ((s >> 1) & 7u)) << 7) | ((s >> 4) & 127u));
    }
    rc = check_hip(hipMemcpyAsync(src, h, sizeof(h), hipMemcpyHostToDevice, st), "mfma_probe: upload");
    if (rc == X2V_OK) rc = check_hip(hipStreamSynchronize(st), "mfma_probe: upload");
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (rc == X2V_OK) rc = check_hip(hipEventCreate(&e0), "mfma_probe: event");
  if (rc == X2V_OK) rc = check_hip(hipEventCreate(&e1), "mfma_probe: event");
  const double flop_per_iter = (double)cus * PR_NW * 2 * PR_NA * PR_NB * (16.0 * 16 * 32) * 2.0;
  const int iters = (int)(2.0e13 / flop_per_iter) + 1;  // ~10 ms per launch at 2 PFLOP/s
  double tf = 0.0;
  for (int phase = 0; phase < 2 && rc == X2V_OK; ++phase) {
    const double budget = 0.5 * milliseconds;
    float ms = 0.f;
    int n = 0;
    rc = check_hip(hipEventRecord(e0, st), "mfma_probe: record");
    while (rc == X2V_OK && ms < budget) {
      for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(mfma_probe_kernel, dim3((unsigned)cus), dim3(PR_NW * 64), 0, st, (const uint4*)src, out, iters);
      n += 4;
      rc = check_hip(hipGetLastError(), "mfma_probe: launch");
      if (rc == X2V_OK) rc = check_hip(hipEventRecord(e1, st), "mfma_probe: record");
      if (rc == X2V_OK) rc = check_hip(hipEventSynchronize(e1), "mfma_probe: sync");
      if (rc == X2V_OK) rc = check_hip(hipEventElapsedTime(&ms, e0, e1), "mfma_probe: elapsed");
    }
    if (rc == X2V_OK && ms > 0.f) tf = flop_per_iter * iters * n / (ms * 1e-3) / 1e12;
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(src);
  (void)hipFree(out);
  if (rc == X2V_OK) *tflops = (float)tf;
  return rc;
}
