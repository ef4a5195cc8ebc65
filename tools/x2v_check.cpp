// GPU-side self-test and micro-benchmark of libx2v_hip.so that needs no Python/torch (starts in
// milliseconds on a fresh box).  Test infrastructure — not part of the product path.
//
//   x2v_check probe            hardware-semantics probes (MFMA layouts, ds_read_b64_tr_b16, LDS-DMA)
//   x2v_check norm|gemm|attn|fp8|conv|misc     numerics vs straightforward CPU references
//   x2v_check bench            GEMM / attention / norm throughput at the BASELINE shapes
//
// CPU references here are plain fp32/fp64 loops over the same bf16 inputs (the torch oracle lives in
// oracle/ and is exercised by tests/).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <string>
#include <vector>

#include "x2v.h"

#define HIP_OK(x)                                                                       \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);     \
      exit(2);                                                                          \
    }                                                                                   \
  } while (0)
#define X2V_OKAY(x)                                                                     \
  do {                                                                                  \
    int r_ = (x);                                                                       \
    if (r_ != X2V_OK) {                                                                 \
      printf("x2v error %d (%s) at %s:%d\n", r_, x2v_last_error(), __FILE__, __LINE__); \
      exit(3);                                                                          \
    }                                                                                   \
  } while (0)

static int g_fail = 0;

// ---------------------------------------------------------------- host bf16 helpers
static inline uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffff) > 0x7f800000) return 0x7fc0;
  u += 0x7fff + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}
static inline float bf2f(uint16_t h) {
  uint32_t u = ((uint32_t)h) << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static inline float rbf(float f) { return bf2f(f2bf(f)); }

struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 2654435761u + 88172645463325252ull) {}
  uint32_t next() {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return (uint32_t)(s >> 32);
  }
  float uni() { return (next() >> 8) * (1.0f / 16777216.0f); }  // [0,1)
  float normal() {
    float u1 = uni() + 1e-7f, u2 = uni();
    return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
  }
};

static std::vector<uint16_t> rand_bf(size_t n, Rng& r, float std, float mean = 0.f) {
  std::vector<uint16_t> v(n);
  for (size_t i = 0; i < n; ++i) v[i] = f2bf(mean + std * r.normal());
  return v;
}

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  explicit DevBuf(size_t n_) : n(n_) { HIP_OK(hipMalloc(&p, std::max<size_t>(n * sizeof(T), 16))); }
  DevBuf(const std::vector<T>& h) : n(h.size()) {
    HIP_OK(hipMalloc(&p, std::max<size_t>(n * sizeof(T), 16)));
    HIP_OK(hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
  }
  ~DevBuf() { (void)hipFree(p); }
  std::vector<T> host() const {
    std::vector<T> h(n);
    HIP_OK(hipMemcpy(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost));
    return h;
  }
  DevBuf(const DevBuf&) = delete;
};

struct ErrStat {
  double max_abs = 0, max_rel = 0, sum_abs = 0;
  size_t n = 0, bad = 0;
};
// compares with |a-b| <= atol + rtol*|b|
static ErrStat compare(const std::vector<float>& got, const std::vector<float>& ref, double atol, double rtol) {
  ErrStat e;
  e.n = ref.size();
  for (size_t i = 0; i < ref.size(); ++i) {
    double d = fabs((double)got[i] - (double)ref[i]);
    if (!(d == d)) d = 1e30;  // NaN
    e.max_abs = std::max(e.max_abs, d);
    e.max_rel = std::max(e.max_rel, d / (fabs((double)ref[i]) + 1e-6));
    e.sum_abs += d;
    if (d > atol + rtol * fabs((double)ref[i])) e.bad++;
  }
  return e;
}
static void report(const char* name, const ErrStat& e, double max_bad_frac = 0.0) {
  const bool ok = (double)e.bad <= max_bad_frac * (double)e.n;
  if (!ok) g_fail++;
  printf("%-58s %s  max_abs=%.3e mean_abs=%.3e bad=%zu/%zu\n", name, ok ? "PASS" : "FAIL", e.max_abs, e.sum_abs / std::max<size_t>(e.n, 1), e.bad, e.n);
}
static std::vector<float> to_f(const std::vector<uint16_t>& v) {
  std::vector<float> f(v.size());
  for (size_t i = 0; i < v.size(); ++i) f[i] = bf2f(v[i]);
  return f;
}

// ---------------------------------------------------------------- probes
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));

// D = A(32x16) * B(16x32) with A[i][k] = (i == k0sel? ...) — we instead feed one-hot patterns and read back
__global__ void probe_mfma32(float* out) {
  // A[i][k] = i*100 + k (as bf16-exact small ints), B[k][j] = (k == 3 && j == lane&31 ...) -> use B = one-hot at
  // k = KSEL so D[i][j] = A[i][KSEL] for every j.  Then out tells which (lane, reg) holds row i, and the A
  // operand k-layout: we set A per lane as A[lane&31][(lane>>5)*8 + e] = value(i,k).
  const int lane = threadIdx.x;
  for (int ksel = 0; ksel < 16; ++ksel) {
    bf16x8_t a, b;
    for (int e = 0; e < 8; ++e) {
      const int i = lane & 31, k = (lane >> 5) * 8 + e;
      a[e] = (__bf16)(float)(i * 16 + k);  // <= 511, exact in bf16? 9 bits needed -> not exact above 256; use i + 32*k/… below
      a[e] = (__bf16)(float)(i + 32 * (k & 7));  // exact (<=255); k&7 = e, the half is identified by ksel sweep
      b[e] = (__bf16)((k == ksel) ? 1.0f : 0.0f);
    }
    f32x16_t c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[(ksel * 64 + lane) * 16 + r] = c[r];
  }
}
__global__ void probe_mfma32_b(float* out) {
  // B probe: A one-hot row selector: A[i][k] = (k == 0), B[k][j] = j + 1 for k == 0 else 0 -> D[i][j] = j+1
  const int lane = threadIdx.x;
  bf16x8_t a, b;
  for (int e = 0; e < 8; ++e) {
    const int k = (lane >> 5) * 8 + e;
    a[e] = (__bf16)((k == 0) ? 1.0f : 0.0f);
    b[e] = (__bf16)((k == 0) ? (float)((lane & 31) + 1) : 0.0f);
  }
  f32x16_t c = {};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) out[lane * 16 + r] = c[r];
}
__global__ void probe_tr(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (short)i;
  __syncthreads();
  // lane L of each 16-lane group points at row (L>>2) (row pitch 64 elements), column group (L&3)*4, group g at row block g*4
  const int lane = threadIdx.x, L = lane & 15, g = lane >> 4;
  const short* p = lds + (g * 4 + (L >> 2)) * 64 + (L & 3) * 4;
  s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
__global__ void probe_glds(const int* src, int* out) {
  __shared__ __attribute__((aligned(16))) int lds[64 * 4 * 2];
  for (int i = threadIdx.x; i < 512; i += 64) lds[i] = -1;
  __syncthreads();
  // lane l loads 16 B from src + (63-l)*4 ints (reversed) into lds base + 1024 B
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (63 - threadIdx.x) * 4),
                                   (__attribute__((address_space(3))) void*)(lds + 256), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}

// raw buffer loads: which offsets take part in the hardware bounds check?  src holds 64 ints = 256 B = num_records.
// out[0] in-range load, out[1] voffset past the end, out[2] voffset in range + soffset past the end,
// out[3] as [2] through the LDS-DMA form (LDS word preset to -7), out[4] voffset past the end through LDS-DMA.
__global__ void probe_buffer_bounds(const int* src, int* out) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) int lds[64 * 2];
  lds[threadIdx.x] = -7;
  lds[64 + threadIdx.x] = -7;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 256, 0x00020000);
  const int a = __builtin_amdgcn_raw_buffer_load_b32(r, 16, 0, 0);
  const int b = __builtin_amdgcn_raw_buffer_load_b32(r, 256 + 16, 0, 0);
  const int c = __builtin_amdgcn_raw_buffer_load_b32(r, 16, 256, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 4, 16, 256, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 64), 4, 256 + 16, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = a;
    out[1] = b;
    out[2] = c;
    out[3] = lds[0];
    out[4] = lds[64];
  }
#endif
}

static void run_probe() {
  {
    // 128 ints allocated, descriptor covers the first 64: in-range word 4 = 104, the word soffset reaches = 168
    std::vector<int> src(128);
    for (int i = 0; i < 128; ++i) src[i] = 100 + i;
    DevBuf<int> s(src), out(8);
    hipLaunchKernelGGL(probe_buffer_bounds, dim3(1), dim3(64), 0, 0, s.p, out.p);
    HIP_OK(hipDeviceSynchronize());
    auto h = out.host();
    printf("probe raw buffer bounds: in-range=%d (want 104); voffset OOB -> %d (want 0); soffset OOB -> %d (0 = soffset IS bounds-checked, 168 = it is NOT);\n"
           "      LDS-DMA soffset OOB -> %d (-7 = no write, 0 = zero written, 168 = not checked); LDS-DMA voffset OOB -> %d\n",
           h[0], h[1], h[2], h[3], h[4]);
    if (h[0] != 104 || h[1] != 0) g_fail++;
    // the kernels rely on: voffset OOB reads return 0 and LDS-DMA voffset OOB writes 0 (M/N/Sk tails)
    if (h[4] != 0) g_fail++;
  }
  {
    DevBuf<float> out(16 * 64 * 16);
    hipLaunchKernelGGL(probe_mfma32, dim3(1), dim3(64), 0, 0, out.p);
    HIP_OK(hipDeviceSynchronize());
    auto h = out.host();
    // expectation: D[i][j] = A[i][ksel]; value encodes i + 32*(ksel&7). Output layout claim: col j = lane&31,
    // row i = (r&3) + 8*(r>>2) + 4*(lane>>5).
    int bad = 0;
    for (int ksel = 0; ksel < 16; ++ksel)
      for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 16; ++r) {
          const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const float exp = (float)(i + 32 * (ksel & 7));
          if (h[(ksel * 64 + lane) * 16 + r] != exp) bad++;
        }
    printf("probe mfma_32x32x16 C layout + A/B k-order (lane>>5)*8+e : %s (mismatches %d)\n", bad ? "MISMATCH" : "OK", bad);
    if (bad) {
      g_fail++;
      printf("  dump ksel=0 lane0: ");
      for (int r = 0; r < 16; ++r) printf("%g ", h[r]);
      printf("\n  dump ksel=9 lane0: ");
      for (int r = 0; r < 16; ++r) printf("%g ", h[(9 * 64) * 16 + r]);
      printf("\n  dump ksel=0 lane33: ");
      for (int r = 0; r < 16; ++r) printf("%g ", h[(33) * 16 + r]);
      printf("\n");
    }
  }
  {
    DevBuf<float> out(64 * 16);
    hipLaunchKernelGGL(probe_mfma32_b, dim3(1), dim3(64), 0, 0, out.p);
    HIP_OK(hipDeviceSynchronize());
    auto h = out.host();
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
      for (int r = 0; r < 16; ++r)
        if (h[lane * 16 + r] != (float)((lane & 31) + 1)) bad++;
    printf("probe mfma_32x32x16 column index = lane&31 of the B operand     : %s (mismatches %d)\n", bad ? "MISMATCH" : "OK", bad);
    if (bad) g_fail++;
  }
  {
    DevBuf<short> out(64 * 4);
    hipLaunchKernelGGL(probe_tr, dim3(1), dim3(64), 0, 0, out.p);
    HIP_OK(hipDeviceSynchronize());
    auto h = out.host();
    // claim: lane c of group g receives elements (row g*4 + j, col c) for j = 0..3 -> value (g*4+j)*64 + c
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
      for (int j = 0; j < 4; ++j) {
        const int g = lane >> 4, c = lane & 15;
        if (h[lane * 4 + j] != (short)((g * 4 + j) * 64 + c)) bad++;
      }
    printf("probe ds_read_b64_tr_b16: lane c gets column c of its group's [4][16] block : %s (mismatches %d)\n", bad ? "MISMATCH" : "OK", bad);
    if (bad) {
      g_fail++;
      for (int lane = 0; lane < 64; lane += 1) {
        printf("  lane %2d:", lane);
        for (int j = 0; j < 4; ++j) printf(" (r%d,c%d)", h[lane * 4 + j] / 64, h[lane * 4 + j] % 64);
        printf("\n");
      }
    }
  }
  {
    std::vector<int> src(256);
    for (int i = 0; i < 256; ++i) src[i] = i;
    DevBuf<int> s(src), out(512);
    hipLaunchKernelGGL(probe_glds, dim3(1), dim3(64), 0, 0, s.p, out.p);
    HIP_OK(hipDeviceSynchronize());
    auto h = out.host();
    // claim: LDS[base + lane*16 B] = src chunk (63-lane)
    int bad = 0;
    for (int i = 0; i < 256; ++i)
      if (h[i] != -1) bad++;
    for (int l = 0; l < 64; ++l)
      for (int e = 0; e < 4; ++e)
        if (h[256 + l * 4 + e] != (63 - l) * 4 + e) bad++;
    printf("probe global_load_lds 16B: LDS dest = wave base + lane*16, per-lane global source : %s (mismatches %d)\n", bad ? "MISMATCH" : "OK", bad);
    if (bad) {
      g_fail++;
      printf("  first words at base: ");
      for (int i = 256; i < 272; ++i) printf("%d ", h[i]);
      printf("\n");
    }
  }
}

// ---------------------------------------------------------------- norm tests
static void ref_layernorm(const std::vector<uint16_t>& x, int64_t M, int D, const uint16_t* w, const uint16_t* b, const uint16_t* scale, const uint16_t* shift,
                          std::vector<float>& out) {
  out.resize((size_t)M * D);
  for (int64_t m = 0; m < M; ++m) {
    double s = 0;
    for (int d = 0; d < D; ++d) s += bf2f(x[m * D + d]);
    const double mean = s / D;
    double q = 0;
    for (int d = 0; d < D; ++d) {
      double t = bf2f(x[m * D + d]) - mean;
      q += t * t;
    }
    const double rstd = 1.0 / sqrt(q / D + 1e-6);
    for (int d = 0; d < D; ++d) {
      float o = (float)((bf2f(x[m * D + d]) - mean) * rstd);
      if (w) o *= bf2f(w[d]);
      if (b) o += bf2f(b[d]);
      o = rbf(o);
      if (scale) {
        o = rbf(o * rbf(1.0f + bf2f(scale[d])));
        o = rbf(o + bf2f(shift[d]));
      }
      out[m * D + d] = o;
    }
  }
}
static void ref_rmsnorm(const std::vector<uint16_t>& x, int64_t M, int D, const uint16_t* w, int mode, std::vector<float>& out) {
  out.resize((size_t)M * D);
  for (int64_t m = 0; m < M; ++m) {
    if (mode == X2V_ROUND_REF) {
      float ss = 0;
      for (int d = 0; d < D; ++d) {
        float v = bf2f(x[m * D + d]);
        ss += rbf(v * v);
      }
      float mean = rbf(ss / D), tt = rbf(mean + 1e-6f), rs = rbf(1.0f / sqrtf(tt));
      for (int d = 0; d < D; ++d) out[m * D + d] = rbf(rbf(bf2f(x[m * D + d]) * rs) * bf2f(w[d]));
    } else {
      double ss = 0;
      for (int d = 0; d < D; ++d) {
        double v = bf2f(x[m * D + d]);
        ss += v * v;
      }
      const float rs = (float)(1.0 / sqrt(ss / D + 1e-6));
      for (int d = 0; d < D; ++d) out[m * D + d] = rbf(bf2f(x[m * D + d]) * rs * bf2f(w[d]));
    }
  }
}

static void run_norm() {
  Rng rng(1);
  const int Ds[] = {128, 256, 1536, 3072, 5120, 13824};
  for (int D : Ds) {
    const int64_t M = 37;
    auto x = rand_bf((size_t)M * D, rng, 2.0f, 0.3f);
    auto w = rand_bf(D, rng, 0.1f, 1.0f), b = rand_bf(D, rng, 0.1f), sc = rand_bf(D, rng, 0.3f), sh = rand_bf(D, rng, 0.3f);
    DevBuf<uint16_t> dx(x), dw(w), db(b), dsc(sc), dsh(sh), dy((size_t)M * D);
    std::vector<float> ref;
    char name[128];
    for (int mode = 0; mode < 2; ++mode) {
      X2V_OKAY(x2v_rmsnorm_bf16(dx.p, D, dw.p, dy.p, D, M, D, 1e-6f, mode, nullptr));
      HIP_OK(hipDeviceSynchronize());
      ref_rmsnorm(x, M, D, w.data(), mode, ref);
      snprintf(name, sizeof name, "rmsnorm D=%d mode=%s", D, mode ? "ref-chain" : "fp32");
      report(name, compare(to_f(dy.host()), ref, 1e-6, 0.0079), 0.002);  // 1 bf16 ulp on <=0.2% (summation order)
    }
    X2V_OKAY(x2v_layernorm_bf16(dx.p, D, nullptr, nullptr, nullptr, nullptr, dy.p, D, M, D, 1e-6f, nullptr));
    HIP_OK(hipDeviceSynchronize());
    ref_layernorm(x, M, D, nullptr, nullptr, nullptr, nullptr, ref);
    snprintf(name, sizeof name, "layernorm D=%d plain", D);
    report(name, compare(to_f(dy.host()), ref, 1e-6, 0.0079), 0.002);
    X2V_OKAY(x2v_layernorm_bf16(dx.p, D, nullptr, nullptr, dsc.p, dsh.p, dy.p, D, M, D, 1e-6f, nullptr));
    HIP_OK(hipDeviceSynchronize());
    ref_layernorm(x, M, D, nullptr, nullptr, sc.data(), sh.data(), ref);
    snprintf(name, sizeof name, "layernorm D=%d modulate", D);
    report(name, compare(to_f(dy.host()), ref, 4e-3, 0.0079), 0.002);
    X2V_OKAY(x2v_layernorm_bf16(dx.p, D, dw.p, db.p, nullptr, nullptr, dy.p, D, M, D, 1e-6f, nullptr));
    HIP_OK(hipDeviceSynchronize());
    ref_layernorm(x, M, D, w.data(), b.data(), nullptr, nullptr, ref);
    snprintf(name, sizeof name, "layernorm D=%d affine", D);
    report(name, compare(to_f(dy.host()), ref, 1e-3, 0.0079), 0.002);
  }
  // gate residual
  {
    const int64_t M = 77;
    const int D = 1536;
    auto x = rand_bf((size_t)M * D, rng, 1.f), y = rand_bf((size_t)M * D, rng, 1.f), g = rand_bf(D, rng, 0.5f);
    DevBuf<uint16_t> dx(x), dy(y), dg(g);
    X2V_OKAY(x2v_gate_residual_bf16(dx.p, D, dy.p, D, dg.p, M, D, nullptr));
    HIP_OK(hipDeviceSynchronize());
    std::vector<float> ref((size_t)M * D);
    for (int64_t i = 0; i < M * D; ++i) ref[i] = rbf(bf2f(x[i]) + rbf(bf2f(y[i]) * bf2f(g[i % D])));
    report("gate_residual (bit-exact)", compare(to_f(dx.host()), ref, 0, 0));
  }
  // activations
  {
    const int64_t n = 8 * 1000 + 3;
    auto x = rand_bf(n, rng, 2.f);
    DevBuf<uint16_t> dx(x), dy(n);
    X2V_OKAY(x2v_activation_bf16(dx.p, dy.p, n, X2V_EPI_GELU_TANH, nullptr));
    HIP_OK(hipDeviceSynchronize());
    std::vector<float> ref(n);
    for (int64_t i = 0; i < n; ++i) {
      double v = bf2f(x[i]);
      ref[i] = rbf((float)(0.5 * v * (1.0 + tanh(0.7978845608028654 * (v + 0.044715 * v * v * v)))));
    }
    report("gelu_tanh", compare(to_f(dy.host()), ref, 1e-6, 0.0079), 0.01);
    X2V_OKAY(x2v_activation_bf16(dx.p, dy.p, n, X2V_EPI_SILU, nullptr));
    HIP_OK(hipDeviceSynchronize());
    for (int64_t i = 0; i < n; ++i) {
      double v = bf2f(x[i]);
      ref[i] = rbf((float)(v / (1.0 + exp(-v))));
    }
    report("silu", compare(to_f(dy.host()), ref, 1e-6, 0.0079), 0.01);
  }
  // sinusoid
  {
    std::vector<int64_t> t = {999, 727, 3, 0};
    DevBuf<int64_t> dt(t);
    DevBuf<uint16_t> dy(4 * 256);
    X2V_OKAY(x2v_sinusoid_embed_bf16(dt.p, dy.p, 4, 256, nullptr));
    HIP_OK(hipDeviceSynchronize());
    std::vector<float> ref(4 * 256);
    for (int r = 0; r < 4; ++r)
      for (int j = 0; j < 128; ++j) {
        double a = (double)t[r] * pow(10000.0, -(double)j / 128.0);
        ref[r * 256 + j] = rbf((float)cos(a));
        ref[r * 256 + 128 + j] = rbf((float)sin(a));
      }
    report("sinusoid_embed", compare(to_f(dy.host()), ref, 1e-6, 0.0079), 0.01);
  }
}

// ---------------------------------------------------------------- rope
static void run_rope() {
  Rng rng(5);
  const int H = 3, D = H * 128;
  const int gf = 3, gh = 4, gw = 6;
  const int64_t S = gf * gh * gw + 8;  // 8 padded tokens (identity rotation)
  auto q = rand_bf((size_t)S * D, rng, 1.5f), k = rand_bf((size_t)S * D, rng, 1.5f);
  auto wq = rand_bf(D, rng, 0.1f, 1.f), wk = rand_bf(D, rng, 0.1f, 1.f);
  std::vector<float> cs(1024 * 64 * 2);
  // reference table (pre_infer.py:12-19): dims 44|42|42 -> 22|21|21 complex columns
  const int dimsz[3] = {44, 42, 42}, cols[3] = {22, 21, 21};
  for (int pos = 0; pos < 1024; ++pos) {
    int col = 0;
    for (int a = 0; a < 3; ++a)
      for (int j = 0; j < cols[a]; ++j, ++col) {
        double f = pos / pow(10000.0, (double)(2 * j) / dimsz[a]);
        cs[(pos * 64 + col) * 2] = (float)cos(f);
        cs[(pos * 64 + col) * 2 + 1] = (float)sin(f);
      }
  }
  for (int mode = 0; mode < 2; ++mode) {
    DevBuf<uint16_t> dq(q), dk(k), dwq(wq), dwk(wk);
    DevBuf<float> dcs(cs);
    X2V_OKAY(x2v_rmsnorm_rope_bf16(dq.p, D, dk.p, D, dwq.p, dwk.p, dcs.p, S, H, 0, gf, gh, gw, 1e-6f, mode, nullptr));
    HIP_OK(hipDeviceSynchronize());
    std::vector<float> nq, nk, rq((size_t)S * D), rk((size_t)S * D);
    ref_rmsnorm(q, S, D, wq.data(), mode, nq);
    ref_rmsnorm(k, S, D, wk.data(), mode, nk);
    for (int64_t s = 0; s < S; ++s) {
      const bool rot = s < gf * gh * gw;
      const int pw = s % gw, ph = (s / gw) % gh, pf = s / (gw * gh);
      for (int h = 0; h < H; ++h)
        for (int c = 0; c < 64; ++c) {
          double co = 1, si = 0;
          if (rot) {
            const int pos = c < 22 ? pf : (c < 43 ? ph : pw);
            co = cs[(pos * 64 + c) * 2];
            si = cs[(pos * 64 + c) * 2 + 1];
          }
          const size_t i0 = s * D + h * 128 + 2 * c;
          rq[i0] = rbf((float)(nq[i0] * co - nq[i0 + 1] * si));
          rq[i0 + 1] = rbf((float)(nq[i0] * si + nq[i0 + 1] * co));
          rk[i0] = rbf((float)(nk[i0] * co - nk[i0 + 1] * si));
          rk[i0 + 1] = rbf((float)(nk[i0] * si + nk[i0 + 1] * co));
        }
    }
    report(mode ? "rmsnorm_rope q (ref-chain)" : "rmsnorm_rope q (fp32)", compare(to_f(dq.host()), rq, 2e-3, 0.0079), 0.004);
    report(mode ? "rmsnorm_rope k (ref-chain)" : "rmsnorm_rope k (fp32)", compare(to_f(dk.host()), rk, 2e-3, 0.0079), 0.004);
  }
}

// ---------------------------------------------------------------- gemm
static void ref_gemm(const std::vector<uint16_t>& x, const std::vector<uint16_t>& w, const uint16_t* bias, int64_t M, int N, int K, int epi,
                     const uint16_t* resid, const uint16_t* gate, std::vector<float>& out) {
  out.resize((size_t)M * N);
  std::vector<float> xf = to_f(x), wf = to_f(w);
  for (int64_t m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      double acc = 0;
      const float* xr = &xf[m * K];
      const float* wr = &wf[(size_t)n * K];
      for (int k = 0; k < K; ++k) acc += (double)xr[k] * wr[k];
      float v = (float)acc + (bias ? bf2f(bias[n]) : 0.f);
      v = rbf(v);
      if (epi == X2V_EPI_GELU_TANH) {
        double t = v;
        v = rbf((float)(0.5 * t * (1.0 + tanh(0.7978845608028654 * (t + 0.044715 * t * t * t)))));
      } else if (epi == X2V_EPI_SILU) {
        double t = v;
        v = rbf((float)(t / (1.0 + exp(-t))));
      } else if (epi == X2V_EPI_RESIDUAL) {
        float g = gate ? rbf(v * bf2f(gate[n])) : v;
        v = rbf(bf2f(resid[m * N + n]) + g);
      }
      out[m * N + n] = v;
    }
}

static void run_gemm() {
  Rng rng(7);
  struct Shape {
    int64_t M;
    int N, K;
  } shapes[] = {{128, 128, 64}, {256, 256, 512}, {200, 384, 256}, {1, 1536, 256}, {77, 64, 1536}, {515, 136, 192}, {300, 1536, 1536}, {700, 520, 128}};
  for (int variant : {1, 2})
  for (auto sh : shapes) {
    auto x = rand_bf((size_t)sh.M * sh.K, rng, 1.0f), w = rand_bf((size_t)sh.N * sh.K, rng, 1.0f / sqrtf((float)sh.K));
    auto b = rand_bf(sh.N, rng, 0.2f), res = rand_bf((size_t)sh.M * sh.N, rng, 1.0f), g = rand_bf(sh.N, rng, 0.5f);
    DevBuf<uint16_t> dx(x), dw(w), db(b), dg(g), dy((size_t)sh.M * sh.N);
    std::vector<float> ref;
    char name[160];
    const int epis[] = {X2V_EPI_NONE, X2V_EPI_GELU_TANH, X2V_EPI_SILU, X2V_EPI_RESIDUAL};
    const char* en[] = {"none", "gelu", "silu", "residual+gate"};
    for (int e = 0; e < 4; ++e) {
      DevBuf<uint16_t> dres(res);
      X2V_OKAY(x2v_gemm_bf16_variant(dx.p, sh.K, dw.p, sh.K, db.p, epis[e] == X2V_EPI_RESIDUAL ? dres.p : dy.p, sh.N, sh.M, sh.N, sh.K, epis[e], dres.p, sh.N,
                                     dg.p, variant, nullptr));
      HIP_OK(hipDeviceSynchronize());
      ref_gemm(x, w, b.data(), sh.M, sh.N, sh.K, epis[e], res.data(), g.data(), ref);
      snprintf(name, sizeof name, "gemm_bf16 v%d M=%lld N=%d K=%d epi=%s", variant, (long long)sh.M, sh.N, sh.K, en[e]);
      auto got = to_f(epis[e] == X2V_EPI_RESIDUAL ? dres.host() : dy.host());
      report(name, compare(got, ref, 2e-3, 0.0079), 0.002);
    }
    // no-bias
    X2V_OKAY(x2v_gemm_bf16_variant(dx.p, sh.K, dw.p, sh.K, nullptr, dy.p, sh.N, sh.M, sh.N, sh.K, X2V_EPI_NONE, nullptr, 0, nullptr, variant, nullptr));
    HIP_OK(hipDeviceSynchronize());
    ref_gemm(x, w, nullptr, sh.M, sh.N, sh.K, X2V_EPI_NONE, nullptr, nullptr, ref);
    snprintf(name, sizeof name, "gemm_bf16 v%d M=%lld N=%d K=%d nobias", variant, (long long)sh.M, sh.N, sh.K);
    report(name, compare(to_f(dy.host()), ref, 2e-3, 0.0079), 0.002);
  }
  // transposition / layout check with asymmetric integer data: exact result required
  {
    const int64_t M = 130;
    const int N = 264, K = 128;
    std::vector<uint16_t> x((size_t)M * K), w((size_t)N * K);
    for (int64_t m = 0; m < M; ++m)
      for (int k = 0; k < K; ++k) x[m * K + k] = f2bf((float)((m * 3 + k) % 5 - 2));
    for (int n = 0; n < N; ++n)
      for (int k = 0; k < K; ++k) w[(size_t)n * K + k] = f2bf((float)((n + 2 * k) % 7 - 3));
    DevBuf<uint16_t> dx(x), dw(w), dy((size_t)M * N);
    std::vector<float> ref;
    ref_gemm(x, w, nullptr, M, N, K, X2V_EPI_NONE, nullptr, nullptr, ref);
    for (int variant : {1, 2}) {
      X2V_OKAY(x2v_gemm_bf16_variant(dx.p, K, dw.p, K, nullptr, dy.p, N, M, N, K, X2V_EPI_NONE, nullptr, 0, nullptr, variant, nullptr));
      HIP_OK(hipDeviceSynchronize());
      report(variant == 1 ? "gemm_bf16 v1 asymmetric integer data (exact)" : "gemm_bf16 v2 asymmetric integer data (exact)", compare(to_f(dy.host()), ref, 0, 0));
    }
  }
  // race screen for the 256x256 ping-pong pipeline: a long-K integer problem (exact in fp32) repeated, every run bit-identical
  // to the first and equal to the CPU result
  {
    const int64_t M = 1100;
    const int N = 1304, K = 2048;
    std::vector<uint16_t> x((size_t)M * K), w((size_t)N * K);
    for (int64_t m = 0; m < M; ++m)
      for (int k = 0; k < K; ++k) x[m * K + k] = f2bf((float)(((m * 7 + k * 3) % 9) - 4));
    for (int n = 0; n < N; ++n)
      for (int k = 0; k < K; ++k) w[(size_t)n * K + k] = f2bf((float)(((n * 5 + 2 * k) % 7) - 3) * 0.5f);
    DevBuf<uint16_t> dx(x), dw(w), dy((size_t)M * N);
    std::vector<float> ref;
    ref_gemm(x, w, nullptr, M, N, K, X2V_EPI_NONE, nullptr, nullptr, ref);
    int bad_runs = 0;
    ErrStat worst{};
    for (int run = 0; run < 20; ++run) {
      X2V_OKAY(x2v_gemm_bf16_variant(dx.p, K, dw.p, K, nullptr, dy.p, N, M, N, K, X2V_EPI_NONE, nullptr, 0, nullptr, 2 | ((1 + run % 4) << 8), nullptr));
      HIP_OK(hipDeviceSynchronize());
      ErrStat e = compare(to_f(dy.host()), ref, 0, 0);
      if (e.bad) { ++bad_runs; worst = e; }
    }
    ErrStat e = worst;
    if (!bad_runs) e = compare(to_f(dy.host()), ref, 0, 0);
    report("gemm_bf16 v2 race screen: 20 runs of M=1100 N=1304 K=2048 integer data (exact)", e);
  }
}

// ---------------------------------------------------------------- attention
static void ref_attn(const std::vector<uint16_t>& q, const std::vector<uint16_t>& k, const std::vector<uint16_t>& v, int64_t Sq, int64_t Sk, int H,
                     std::vector<float>& out) {
  const int d = 128;
  out.assign((size_t)Sq * H * d, 0.f);
  std::vector<double> s(Sk);
  const double scale = 1.0 / sqrt(128.0);
  for (int h = 0; h < H; ++h)
    for (int64_t i = 0; i < Sq; ++i) {
      double mx = -1e300;
      for (int64_t j = 0; j < Sk; ++j) {
        double a = 0;
        for (int e = 0; e < d; ++e) a += (double)bf2f(q[(i * H + h) * d + e]) * bf2f(k[(j * H + h) * d + e]);
        s[j] = a * scale;
        mx = std::max(mx, s[j]);
      }
      double l = 0;
      for (int64_t j = 0; j < Sk; ++j) {
        s[j] = exp(s[j] - mx);
        l += s[j];
      }
      for (int e = 0; e < d; ++e) {
        double o = 0;
        for (int64_t j = 0; j < Sk; ++j) o += s[j] * bf2f(v[(j * H + h) * d + e]);
        out[(i * H + h) * d + e] = (float)(o / l);
      }
    }
}

// variants 12, 13 = pre-transposed V path (x2v_transpose_heads_bf16 + x2v_attn_fwd_bf16_vt, kernel selector 0 / 1); everything else goes to the variant entry
struct AttnVt {
  DevBuf<uint16_t>* vt = nullptr;
  int64_t ldvt = 0;
  ~AttnVt() { delete vt; }
  void prepare(const void* v, int64_t ldv, int64_t Sk, int H) {
    ldvt = (Sk + 63) / 64 * 64;
    delete vt;
    vt = new DevBuf<uint16_t>((size_t)H * 128 * ldvt);
    X2V_OKAY(x2v_transpose_heads_bf16(v, ldv, vt->p, ldvt, Sk, H, nullptr));
  }
};
// variant 13: the producer / consumer probe (tools/probes/attn_pc.hip) where the loaded library carries it (tools/probes/build_attn_pc.sh)
typedef int (*probe_attn_fn)(const void*, int64_t, const void*, int64_t, const void*, int64_t, void*, int64_t, int64_t, int64_t, int, int, float, int, void*);
static probe_attn_fn probe_attn_pc() {
  static probe_attn_fn fn = (probe_attn_fn)dlsym(RTLD_DEFAULT, "x2v_probe_attn_pc");
  return fn;
}
static void attn_any(int variant, AttnVt& t, const void* q, const void* k, const void* v, void* o, int64_t Sq, int64_t Sk, int H) {
  if (variant == 13) {
    if (!probe_attn_pc()) {
      fprintf(stderr, "variant 13 needs a library built by tools/probes/build_attn_pc.sh (x2v_probe_attn_pc not found)\n");
      exit(2);
    }
    X2V_OKAY(probe_attn_pc()(q, H * 128, k, H * 128, t.vt->p, t.ldvt, o, H * 128, Sq, Sk, H, 128, 0.f, 0, nullptr));
  } else if (variant == 12)  // the ping-pong kernel on a pre-transposed V
    X2V_OKAY(x2v_attn_fwd_bf16_vt(q, H * 128, k, H * 128, t.vt->p, t.ldvt, o, H * 128, Sq, Sk, H, 128, 0.f, 0, nullptr));
  else
    X2V_OKAY(x2v_attn_fwd_bf16_variant(q, H * 128, k, H * 128, v, H * 128, o, H * 128, Sq, Sk, H, 128, 0.f, variant, nullptr));
}

static void run_attn() {
  Rng rng(11);
  struct Shape {
    int64_t Sq, Sk;
    int H;
  } shapes[] = {{64, 64, 1}, {256, 256, 2}, {200, 333, 2}, {300, 40, 3}, {515, 512, 1}, {33, 1000, 2}};
  for (auto sh : shapes) {
    auto q = rand_bf((size_t)sh.Sq * sh.H * 128, rng, 1.0f), k = rand_bf((size_t)sh.Sk * sh.H * 128, rng, 1.0f), v = rand_bf((size_t)sh.Sk * sh.H * 128, rng, 1.0f);
    // spike one key against one query so the online-softmax rescale path fires mid-sequence
    if (sh.Sk > 100)
      for (int e = 0; e < 128; ++e) k[(size_t)(90 * sh.H) * 128 + e] = f2bf(4.0f * bf2f(q[(size_t)(5 * sh.H) * 128 + e]));
    DevBuf<uint16_t> dq(q), dk(k), dv(v), dout((size_t)sh.Sq * sh.H * 128);
    std::vector<float> ref;
    ref_attn(q, k, v, sh.Sq, sh.Sk, sh.H, ref);
    AttnVt vt;
    vt.prepare(dv.p, sh.H * 128, sh.Sk, sh.H);
    for (int variant : {0, 4, 5, 6, 12, 13}) {
      if (variant == 13 && !probe_attn_pc()) continue;
      HIP_OK(hipMemset(dout.p, 0xff, dout.n * 2));
      attn_any(variant, vt, dq.p, dk.p, dv.p, dout.p, sh.Sq, sh.Sk, sh.H);
      HIP_OK(hipDeviceSynchronize());
      char name[160];
      snprintf(name, sizeof name, "attn Sq=%lld Sk=%lld H=%d variant=%d", (long long)sh.Sq, (long long)sh.Sk, sh.H, variant);
      report(name, compare(to_f(dout.host()), ref, 6e-3, 0.016), 0.0);  // bf16 P + bf16 output
    }
  }
  // every score strongly negative (keys anti-aligned with the queries): the running max must be adopted downwards on
  // the first tile or exp2 underflows and l = 0; plus a late positive spike far above everything before it
  {
    const int64_t S = 300;
    const int H = 2;
    auto q = rand_bf((size_t)S * H * 128, rng, 1.0f), v = rand_bf((size_t)S * H * 128, rng, 1.0f);
    std::vector<uint16_t> k(q.size());
    for (int64_t s = 0; s < S; ++s)
      for (int h = 0; h < H; ++h)
        for (int e = 0; e < 128; ++e) {
          const float base = bf2f(q[(size_t)(7 * H + h) * 128 + e]);
          k[(size_t)(s * H + h) * 128 + e] = f2bf((s == 250 ? 6.0f : -12.0f) * base + 0.05f * rng.normal());
        }
    DevBuf<uint16_t> dq(q), dk(k), dv(v), dout((size_t)S * H * 128);
    std::vector<float> ref;
    ref_attn(q, k, v, S, S, H, ref);
    AttnVt vt;
    vt.prepare(dv.p, H * 128, S, H);
    for (int variant : {4, 6, 12, 13}) {
      if (variant == 13 && !probe_attn_pc()) continue;
      HIP_OK(hipMemset(dout.p, 0xff, dout.n * 2));
      attn_any(variant, vt, dq.p, dk.p, dv.p, dout.p, S, S, H);
      HIP_OK(hipDeviceSynchronize());
      char name[160];
      snprintf(name, sizeof name, "attn anti-aligned keys + late spike variant=%d", variant);
      // v3/v4 round Q * scale * log2(e) to bf16 once more (relative 2^-9); keys 12x larger than any real (normalised) key
      // amplify that to ~0.03 in the exponent where the spike and the 299 other keys carry comparable weight
      report(name, compare(to_f(dout.host()), ref, variant >= 12 ? 6e-2 : 6e-3, 0.016), 0.0);
    }
  }
  // strided (fused-QKV style) views: ld = 3*H*128
  {
    const int64_t S = 130;
    const int H = 2, ld = 3 * H * 128;
    auto qkv = rand_bf((size_t)S * ld, rng, 1.0f);
    std::vector<uint16_t> q((size_t)S * H * 128), k(q.size()), v(q.size());
    for (int64_t s = 0; s < S; ++s)
      for (int c = 0; c < H * 128; ++c) {
        q[s * H * 128 + c] = qkv[s * ld + c];
        k[s * H * 128 + c] = qkv[s * ld + H * 128 + c];
        v[s * H * 128 + c] = qkv[s * ld + 2 * H * 128 + c];
      }
    DevBuf<uint16_t> d(qkv), dout((size_t)S * H * 128);
    X2V_OKAY(x2v_attn_fwd_bf16(d.p, ld, d.p + H * 128, ld, d.p + 2 * H * 128, ld, dout.p, H * 128, S, S, H, 128, 0.f, nullptr));
    HIP_OK(hipDeviceSynchronize());
    std::vector<float> ref;
    ref_attn(q, k, v, S, S, H, ref);
    report("attn strided qkv views", compare(to_f(dout.host()), ref, 6e-3, 0.016), 0.0);
  }
}

// ---------------------------------------------------------------- fp8
static float e4m3_to_f(uint8_t b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v;
  if (e == 0)
    v = ldexpf((float)m, -9);
  else if (e == 15 && m == 7)
    v = NAN;
  else
    v = ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -v : v;
}
static void run_fp8() {
  Rng rng(13);
  const int64_t M = 200;
  const int K = 512, N = 264;
  auto x = rand_bf((size_t)M * K, rng, 1.5f);
  DevBuf<uint16_t> dx(x);
  DevBuf<uint8_t> dq((size_t)M * K);
  DevBuf<float> ds(M);
  X2V_OKAY(x2v_quant_fp8_rowwise(dx.p, K, dq.p, K, ds.p, M, K, nullptr));
  HIP_OK(hipDeviceSynchronize());
  auto hq = dq.host();
  auto hs = ds.host();
  // check scale and dequantised error
  std::vector<float> got((size_t)M * K), ref((size_t)M * K), sref(M), sgot(M);
  for (int64_t m = 0; m < M; ++m) {
    float amax = 0;
    for (int k = 0; k < K; ++k) amax = std::max(amax, fabsf(bf2f(x[m * K + k])));
    sref[m] = amax / 448.f;
    sgot[m] = hs[m];
    for (int k = 0; k < K; ++k) {
      got[m * K + k] = e4m3_to_f(hq[m * K + k]) * hs[m];
      ref[m * K + k] = bf2f(x[m * K + k]);
    }
  }
  report("quant_fp8 per-token scale", compare(sgot, sref, 0, 1e-6));
  report("quant_fp8 dequantised value within half an e4m3 ulp", compare(got, ref, 2e-3 * 1.5 * 4, 0.0625), 0.0);
  // fp8 gemm vs fp64 on the quantised operands
  std::vector<uint8_t> wq((size_t)N * K);
  std::vector<float> sw(N);
  for (int n = 0; n < N; ++n) {
    sw[n] = 0.001f + 0.002f * rng.uni();
    for (int k = 0; k < K; ++k) {
      uint8_t b = (uint8_t)(rng.next() & 0xff);
      if ((b & 0x7f) == 0x7f) b &= 0xfe;  // no NaN
      if (((b >> 3) & 15) > 11) b &= 0xbf;  // keep magnitudes moderate
      wq[(size_t)n * K + k] = b;
    }
  }
  auto bias = rand_bf(N, rng, 0.2f);
  DevBuf<uint8_t> dwq(wq);
  DevBuf<float> dsw(sw);
  DevBuf<uint16_t> db(bias), dy((size_t)M * N);
  std::vector<float> yref((size_t)M * N);
  for (int64_t m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      double acc = 0;
      for (int k = 0; k < K; ++k) acc += (double)e4m3_to_f(hq[m * K + k]) * e4m3_to_f(wq[(size_t)n * K + k]);
      yref[m * N + n] = rbf((float)(acc * hs[m] * sw[n]) + bf2f(bias[n]));
    }
  for (int variant : {1, 2}) {
    X2V_OKAY(x2v_gemm_fp8_variant(dq.p, K, ds.p, dwq.p, K, dsw.p, db.p, dy.p, N, M, N, K, X2V_EPI_NONE, nullptr, 0, nullptr, variant, nullptr));
    HIP_OK(hipDeviceSynchronize());
    report(variant == 1 ? "gemm_fp8 v1 (MX-scaled MFMA, unit block scales) vs fp64" : "gemm_fp8 v2 (256x256 kernel) vs fp64",
           compare(to_f(dy.host()), yref, 2e-3, 0.0079), 0.002);
  }
}

// ---------------------------------------------------------------- conv
static void run_conv() {
  Rng rng(17);
  struct C {
    int T, H, W, Cin, Cout, kt, kh, kw, nc;
  } cases[] = {{2, 6, 10, 16, 24, 3, 3, 3, 2}, {1, 9, 7, 32, 3, 3, 3, 3, 1}, {3, 5, 5, 20, 70, 3, 3, 3, 0}, {2, 8, 8, 16, 16, 1, 3, 3, 0}, {2, 4, 6, 16, 32, 3, 1, 1, 2}, {1, 4, 4, 16, 16, 1, 1, 1, 0}};
  for (auto c : cases) {
    const size_t nx = (size_t)c.T * c.H * c.W * c.Cin, ncache = (size_t)c.nc * c.H * c.W * c.Cin, nw = (size_t)c.Cout * c.kt * c.kh * c.kw * c.Cin;
    std::vector<float> x(nx), cache(ncache), w(nw), b(c.Cout);
    for (auto& f : x) f = rng.normal();
    for (auto& f : cache) f = rng.normal();
    for (auto& f : w) f = rng.normal() * 0.1f;
    for (auto& f : b) f = rng.normal();
    DevBuf<float> dx(x), dc(cache), dw(w), db(b), dy((size_t)c.T * c.H * c.W * c.Cout);
    X2V_OKAY(x2v_causal_conv3d_f32(dx.p, c.nc ? dc.p : nullptr, c.nc, dw.p, db.p, dy.p, c.T, c.H, c.W, c.Cin, c.Cout, c.kt, c.kh, c.kw, nullptr));
    HIP_OK(hipDeviceSynchronize());
    std::vector<float> ref((size_t)c.T * c.H * c.W * c.Cout);
    for (int t = 0; t < c.T; ++t)
      for (int h = 0; h < c.H; ++h)
        for (int ww = 0; ww < c.W; ++ww)
          for (int co = 0; co < c.Cout; ++co) {
            double acc = b[co];
            for (int dt = 0; dt < c.kt; ++dt)
              for (int dh = 0; dh < c.kh; ++dh)
                for (int dw_ = 0; dw_ < c.kw; ++dw_) {
                  const int tt = t + dt - (c.kt - 1), hh = h + dh - c.kh / 2, w2 = ww + dw_ - c.kw / 2;
                  if (hh < 0 || hh >= c.H || w2 < 0 || w2 >= c.W) continue;
                  const float* src = nullptr;
                  if (tt >= 0)
                    src = &x[(((size_t)tt * c.H + hh) * c.W + w2) * c.Cin];
                  else if (c.nc + tt >= 0)
                    src = &cache[(((size_t)(c.nc + tt) * c.H + hh) * c.W + w2) * c.Cin];
                  if (!src) continue;
                  const float* wp = &w[((((size_t)co * c.kt + dt) * c.kh + dh) * c.kw + dw_) * c.Cin];
                  for (int ci = 0; ci < c.Cin; ++ci) acc += (double)src[ci] * wp[ci];
                }
            ref[(((size_t)t * c.H + h) * c.W + ww) * c.Cout + co] = (float)acc;
          }
    char name[160];
    snprintf(name, sizeof name, "causal_conv3d T=%d %dx%d Cin=%d Cout=%d k=%dx%dx%d cache=%d", c.T, c.H, c.W, c.Cin, c.Cout, c.kt, c.kh, c.kw, c.nc);
    report(name, compare(dy.host(), ref, 2e-4, 1e-4));
  }
}

// ---------------------------------------------------------------- error-path checks (no GPU work)
static void run_errors() {
  int bad = 0;
  bad += x2v_gemm_bf16((void*)16, 64, (void*)16, 64, nullptr, (void*)16, 64, 4, 64, 63, 0, nullptr, 0, nullptr, nullptr) != X2V_E_SHAPE;
  bad += x2v_gemm_bf16(nullptr, 64, (void*)16, 64, nullptr, (void*)16, 64, 4, 64, 64, 0, nullptr, 0, nullptr, nullptr) != X2V_E_ARG;
  bad += x2v_gemm_bf16((void*)8, 64, (void*)16, 64, nullptr, (void*)16, 64, 4, 64, 64, 0, nullptr, 0, nullptr, nullptr) != X2V_E_ALIGN;
  bad += x2v_attn_fwd_bf16((void*)16, 128, (void*)16, 128, (void*)16, 128, (void*)16, 128, 4, 4, 1, 64, 0.f, nullptr) != X2V_E_SHAPE;
  bad += x2v_rmsnorm_bf16((void*)16, 8, (void*)16, (void*)16, 8, 1, 12, 1e-6f, 0, nullptr) != X2V_E_SHAPE;
  bad += strlen(x2v_last_error()) == 0;
  printf("%-58s %s\n", "error codes (shape/arg/align) and x2v_last_error", bad ? "FAIL" : "PASS");
  if (bad) g_fail++;
}

// ---------------------------------------------------------------- bench
static double time_ms(int iters, const std::function<void()>& f) {
  hipEvent_t a, b;
  HIP_OK(hipEventCreate(&a));
  HIP_OK(hipEventCreate(&b));
  f();
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipEventRecord(a, 0));
  for (int i = 0; i < iters; ++i) f();
  HIP_OK(hipEventRecord(b, 0));
  HIP_OK(hipEventSynchronize(b));
  float ms = 0;
  HIP_OK(hipEventElapsedTime(&ms, a, b));
  HIP_OK(hipEventDestroy(a));
  HIP_OK(hipEventDestroy(b));
  return ms / iters;
}

static void fill_random(DevBuf<uint16_t>& d, Rng& rng, float std) {
  // random bf16 fill generated on the host in chunks (uniform-ish normal); reused pattern is fine for timing
  std::vector<uint16_t> h(std::min<size_t>(d.n, 1 << 22));
  for (auto& v : h) v = f2bf(std * rng.normal());
  for (size_t off = 0; off < d.n; off += h.size())
    HIP_OK(hipMemcpy(d.p + off, h.data(), std::min(h.size(), d.n - off) * 2, hipMemcpyHostToDevice));
}

static void run_bench(bool big) {
  Rng rng(23);
  struct G {
    const char* name;
    int64_t M;
    int N, K, epi;
  };
  std::vector<G> gs = {{"1.3B qkv/o  S=20280 D=1536", 20280, 1536, 1536, X2V_EPI_NONE},
                       {"1.3B ffn0   S=20280 1536->8960 gelu", 20280, 8960, 1536, X2V_EPI_GELU_TANH},
                       {"1.3B ffn2   S=20280 8960->1536 resid", 20280, 1536, 8960, X2V_EPI_RESIDUAL},
                       {"sq 4096^3", 4096, 4096, 4096, X2V_EPI_NONE},
                       {"sq 8192^3", 8192, 8192, 8192, X2V_EPI_NONE}};
  if (big) {
    gs.push_back({"14B qkv/o   S=75600 D=5120", 75600, 5120, 5120, X2V_EPI_NONE});
    gs.push_back({"14B ffn0    S=75600 5120->13824 gelu", 75600, 13824, 5120, X2V_EPI_GELU_TANH});
    gs.push_back({"14B ffn2    S=75600 13824->5120 resid", 75600, 5120, 13824, X2V_EPI_RESIDUAL});
  }
  for (auto g : gs) {
    DevBuf<uint16_t> x((size_t)g.M * g.K), w((size_t)g.N * g.K), b(g.N), y((size_t)g.M * g.N), gate(g.N);
    fill_random(x, rng, 1.f);
    fill_random(w, rng, 0.02f);
    fill_random(b, rng, 0.02f);
    fill_random(gate, rng, 0.5f);
    fill_random(y, rng, 1.f);
    const int iters = g.M * (double)g.N * g.K > 1e13 ? 3 : 10;
    for (int variant : {1, 2 | (4 << 8), 2 | (8 << 8)}) {
      double ms = time_ms(iters, [&] {
        X2V_OKAY(x2v_gemm_bf16_variant(x.p, g.K, w.p, g.K, b.p, y.p, g.N, g.M, g.N, g.K, g.epi, g.epi == X2V_EPI_RESIDUAL ? y.p : nullptr, g.N,
                                       g.epi == X2V_EPI_RESIDUAL ? gate.p : nullptr, variant, nullptr));
      });
      const double tf = 2.0 * g.M * g.N * g.K / (ms * 1e-3) / 1e12;
      printf("BENCH gemm_bf16 v=%d gm=%d %-42s %9.3f ms  %8.1f TFLOP/s  (%.1f%% of 2500)\n", variant & 255, (variant >> 8) & 255, g.name, ms, tf, tf / 25.0);
    }
  }
  if (big) {  // fp8 w8a8 GEMMs (config #4) on the 14B shapes
    struct F { const char* name; int64_t M; int N, K; } fs[] = {{"14B qkv/o   S=75600 D=5120", 75600, 5120, 5120}, {"14B ffn0    5120->13824", 75600, 13824, 5120},
                                                              {"14B ffn2    13824->5120", 75600, 5120, 13824}};
    for (auto f : fs) {
      DevBuf<uint8_t> xq((size_t)f.M * f.K), wq((size_t)f.N * f.K);
      DevBuf<float> sx(f.M), sw(f.N);
      DevBuf<uint16_t> b(f.N), y((size_t)f.M * f.N);
      {
        std::vector<uint8_t> h(1 << 22);
        for (auto& v : h) { v = (uint8_t)(rng.next() & 0xff); if ((v & 0x7f) >= 0x78) v &= 0xbf; }
        for (size_t off = 0; off < xq.n; off += h.size()) HIP_OK(hipMemcpy(xq.p + off, h.data(), std::min(h.size(), xq.n - off), hipMemcpyHostToDevice));
        for (size_t off = 0; off < wq.n; off += h.size()) HIP_OK(hipMemcpy(wq.p + off, h.data(), std::min(h.size(), wq.n - off), hipMemcpyHostToDevice));
        std::vector<float> s1(f.M, 0.01f), s2(f.N, 0.001f);
        HIP_OK(hipMemcpy(sx.p, s1.data(), s1.size() * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(sw.p, s2.data(), s2.size() * 4, hipMemcpyHostToDevice));
      }
      fill_random(b, rng, 0.02f);
      for (int variant : {1, 2}) {
        double ms = time_ms(3, [&] { X2V_OKAY(x2v_gemm_fp8_variant(xq.p, f.K, sx.p, wq.p, f.K, sw.p, b.p, y.p, f.N, f.M, f.N, f.K, X2V_EPI_NONE, nullptr, 0, nullptr, variant, nullptr)); });
        const double tf = 2.0 * f.M * f.N * f.K / (ms * 1e-3) / 1e12;
        printf("BENCH gemm_fp8 v=%d %-42s %9.3f ms  %8.1f TFLOP/s  (%.1f%% of 5000)\n", variant, f.name, ms, tf, tf / 50.0);
      }
    }
  }
  struct A {
    const char* name;
    int64_t Sq, Sk;
    int H;
  };
  std::vector<A> as = {{"1.3B self  S=20280 H=12", 20280, 20280, 12}, {"1.3B cross S=20280x512 H=12", 20280, 512, 12}, {"N=2048 H=64", 2048, 2048, 64}, {"N=8192 H=16", 8192, 8192, 16}};
  if (big) {
    as.push_back({"14B self   S=75600 H=40 (1 GPU)", 75600, 75600, 40});
    as.push_back({"14B self   S=75600 H=5  (Ulysses rank)", 75600, 75600, 5});
    as.push_back({"14B cross  S=75600x512 H=40", 75600, 512, 40});
  }
  for (auto a : as) {
    DevBuf<uint16_t> q((size_t)a.Sq * a.H * 128), k((size_t)a.Sk * a.H * 128), v((size_t)a.Sk * a.H * 128), o((size_t)a.Sq * a.H * 128);
    fill_random(q, rng, 1.f);
    fill_random(k, rng, 1.f);
    fill_random(v, rng, 1.f);
    AttnVt vt;
    vt.prepare(v.p, a.H * 128, a.Sk, a.H);
    {
      double ms = time_ms(5, [&] { X2V_OKAY(x2v_transpose_heads_bf16(v.p, a.H * 128, vt.vt->p, vt.ldvt, a.Sk, a.H, nullptr)); });
      printf("BENCH transpose_heads %-40s %9.3f ms  %8.1f GB/s\n", a.name, ms, 4.0 * a.Sk * a.H * 128 / ms / 1e6);
    }
    for (int variant : {6, 12, 13}) {
      const double flop = 4.0 * a.Sq * a.Sk * a.H * 128;
      const int iters = flop > 2e13 ? 1 : 5;
      double ms = time_ms(iters, [&] { attn_any(variant, vt, q.p, k.p, v.p, o.p, a.Sq, a.Sk, a.H); });
      const double tf = flop / (ms * 1e-3) / 1e12;
      printf("BENCH attn variant=%d %-40s %9.3f ms  %8.1f TFLOP/s  (%.1f%% of 2500)\n", variant, a.name, ms, tf, tf / 25.0);
    }
  }
  // HBM-bound rows
  {
    const int64_t M = big ? 75600 : 20280;
    const int D = big ? 5120 : 1536;
    DevBuf<uint16_t> x((size_t)M * D), y((size_t)M * D), w(D), sc(D), sh(D);
    fill_random(x, rng, 1.f);
    fill_random(w, rng, 1.f);
    fill_random(sc, rng, .1f);
    fill_random(sh, rng, .1f);
    const double bytes = 2.0 * M * D * 2;
    double ms = time_ms(10, [&] { X2V_OKAY(x2v_rmsnorm_bf16(x.p, D, w.p, y.p, D, M, D, 1e-6f, 0, nullptr)); });
    printf("BENCH rmsnorm M=%lld D=%d            %9.3f ms  %8.1f GB/s (%.1f%% of 6290 achievable)\n", (long long)M, D, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 62.9);
    ms = time_ms(10, [&] { X2V_OKAY(x2v_layernorm_bf16(x.p, D, nullptr, nullptr, sc.p, sh.p, y.p, D, M, D, 1e-6f, nullptr)); });
    printf("BENCH layernorm+modulate M=%lld D=%d %9.3f ms  %8.1f GB/s (%.1f%%)\n", (long long)M, D, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 62.9);
    std::vector<float> cs(1024 * 64 * 2, 0.5f);
    DevBuf<float> dcs(cs);
    ms = time_ms(10, [&] { X2V_OKAY(x2v_rmsnorm_rope_bf16(x.p, D, y.p, D, w.p, w.p, dcs.p, M, D / 128, 0, 21, 45, 80, 1e-6f, 0, nullptr)); });
    printf("BENCH rmsnorm+rope (q,k) M=%lld D=%d  %9.3f ms  %8.1f GB/s (%.1f%%)\n", (long long)M, D, ms, 2 * bytes / ms / 1e6, 2 * bytes / ms / 1e6 / 62.9);
  }
}

// HBM-bound row kernels at the DiT activation shapes: the one-block-per-row and the persistent streaming forms must agree bit for
// bit (same arithmetic, different schedule), and both are timed against the achievable HBM rate.  `x2v_check hbm`
static ErrStat compare_bits(const DevBuf<uint16_t>& a, const DevBuf<uint16_t>& b) {
  const auto ha = a.host(), hb = b.host();
  ErrStat e;
  e.n = ha.size();
  if (memcmp(ha.data(), hb.data(), ha.size() * 2) != 0)
    for (size_t i = 0; i < ha.size(); ++i) e.bad += ha[i] != hb[i];
  e.max_abs = (double)e.bad;
  return e;
}

static void run_hbm() {
  Rng rng(41);
  // all = every operand combination is compared (small shapes); otherwise only the combinations the DiT block uses
  struct Shape { int64_t M; int D; int gf, gh, gw; bool all; } shapes[] = {{1000, 5120, 3, 17, 20, true},  {3000, 1536, 3, 25, 40, true}, {2100, 3072, 3, 25, 28, true},
                                                                           {777, 8192, 3, 16, 16, true},   {75600, 5120, 21, 45, 80, false}, {20280, 1536, 13, 30, 52, false}};
  for (auto sh : shapes) {
    const int64_t M = sh.M;
    const int D = sh.D;
    DevBuf<uint16_t> x((size_t)M * D), y1((size_t)M * D), y2((size_t)M * D), w(D), b(D), sc(D), shf(D), q1((size_t)M * D), k1((size_t)M * D), q2((size_t)M * D), k2((size_t)M * D);
    fill_random(x, rng, 1.f);
    fill_random(w, rng, 1.f);
    fill_random(b, rng, .1f);
    fill_random(sc, rng, .1f);
    fill_random(shf, rng, .1f);
    std::vector<float> cs(1024 * 64 * 2);
    for (size_t i = 0; i < cs.size(); i += 2) {
      const double a = 0.37 * (double)(i / 2 % 977);
      cs[i] = (float)cos(a);
      cs[i + 1] = (float)sin(a);
    }
    DevBuf<float> dcs(cs);
    const double bytes = 2.0 * M * D * 2;
    char name[160];
    struct Ln { const char* what; const uint16_t *w, *b, *sc, *sh; } lns[] = {{"modulate", nullptr, nullptr, sc.p, shf.p}, {"affine", w.p, b.p, nullptr, nullptr},
                                                                             {"plain", nullptr, nullptr, nullptr, nullptr}, {"affine+modulate", w.p, b.p, sc.p, shf.p}};
    for (auto ln : lns) {
      if (!sh.all && ln.what[0] == 'p') continue;
      if (!sh.all && ln.w && ln.sc) continue;
      X2V_OKAY(x2v_layernorm_bf16_variant(x.p, D, ln.w, ln.b, ln.sc, ln.sh, y1.p, D, M, D, 1e-6f, 1, nullptr));
      X2V_OKAY(x2v_layernorm_bf16_variant(x.p, D, ln.w, ln.b, ln.sc, ln.sh, y2.p, D, M, D, 1e-6f, 2, nullptr));
      HIP_OK(hipDeviceSynchronize());
      snprintf(name, sizeof name, "layernorm %s M=%lld D=%d: streaming == per-row (bit-exact)", ln.what, (long long)M, D);
      report(name, compare_bits(y2, y1));
      if (sh.all) continue;
      for (int variant : {1, 2}) {
        double ms = time_ms(10, [&] { X2V_OKAY(x2v_layernorm_bf16_variant(x.p, D, ln.w, ln.b, ln.sc, ln.sh, y1.p, D, M, D, 1e-6f, variant, nullptr)); });
        printf("BENCH layernorm %-16s v=%d M=%lld D=%d %9.3f ms  %8.1f GB/s (%.1f%% of 6290 achievable)\n", ln.what, variant, (long long)M, D, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 62.9);
      }
    }
    for (int mode = 0; mode < 2; ++mode) {
      for (int with_w = 1; with_w >= 0; --with_w) {
        if (!sh.all && (mode == 1 || with_w == 0)) continue;
        HIP_OK(hipMemcpy(q1.p, x.p, (size_t)M * D * 2, hipMemcpyDeviceToDevice));
        HIP_OK(hipMemcpy(q2.p, x.p, (size_t)M * D * 2, hipMemcpyDeviceToDevice));
        HIP_OK(hipMemset(k1.p, 0, (size_t)M * D * 2));
        HIP_OK(hipMemset(k2.p, 0, (size_t)M * D * 2));
        HIP_OK(hipMemcpy(k1.p, x.p + 8, ((size_t)M * D - 8) * 2, hipMemcpyDeviceToDevice));  // k = x shifted by one chunk: different data than q
        HIP_OK(hipMemcpy(k2.p, x.p + 8, ((size_t)M * D - 8) * 2, hipMemcpyDeviceToDevice));
        const uint16_t* ww = with_w ? w.p : nullptr;
        const uint16_t* wk = with_w ? b.p : nullptr;
        // s0 = 7: the last 7 tokens fall beyond the grid (identity rotation)
        X2V_OKAY(x2v_rmsnorm_rope_scaled_bf16_variant(q1.p, D, k1.p, D, ww, wk, dcs.p, M, D / 128, 7, sh.gf, sh.gh, sh.gw, 1e-6f, mode, 0.1275f, 1, nullptr));
        X2V_OKAY(x2v_rmsnorm_rope_scaled_bf16_variant(q2.p, D, k2.p, D, ww, wk, dcs.p, M, D / 128, 7, sh.gf, sh.gh, sh.gw, 1e-6f, mode, 0.1275f, 2, nullptr));
        HIP_OK(hipDeviceSynchronize());
        snprintf(name, sizeof name, "rmsnorm+rope M=%lld D=%d mode=%d norm=%d: streaming == per-row (bit-exact, q)", (long long)M, D, mode, with_w);
        report(name, compare_bits(q2, q1));
        snprintf(name, sizeof name, "rmsnorm+rope M=%lld D=%d mode=%d norm=%d: streaming == per-row (bit-exact, k)", (long long)M, D, mode, with_w);
        report(name, compare_bits(k2, k1));
      }
    }
    if (sh.all) continue;
    for (int variant : {1, 2}) {
      double ms = time_ms(10, [&] { X2V_OKAY(x2v_rmsnorm_rope_scaled_bf16_variant(q1.p, D, k1.p, D, w.p, b.p, dcs.p, M, D / 128, 0, sh.gf, sh.gh, sh.gw, 1e-6f, 0, 0.1275f, variant, nullptr)); });
      printf("BENCH rmsnorm+rope (q,k) v=%d M=%lld D=%d %9.3f ms  %8.1f GB/s (%.1f%%)\n", variant, (long long)M, D, ms, 2 * bytes / ms / 1e6, 2 * bytes / ms / 1e6 / 62.9);
    }
    double ms = time_ms(10, [&] { X2V_OKAY(x2v_rmsnorm_bf16(x.p, D, w.p, y1.p, D, M, D, 1e-6f, 0, nullptr)); });
    printf("BENCH rmsnorm M=%lld D=%d            %9.3f ms  %8.1f GB/s (%.1f%%)\n", (long long)M, D, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 62.9);
    ms = time_ms(10, [&] { X2V_OKAY(x2v_gate_residual_bf16(y1.p, D, x.p, D, sc.p, M, D, nullptr)); });
    printf("BENCH gate_residual M=%lld D=%d       %9.3f ms  %8.1f GB/s (%.1f%%; 3 streams)\n", (long long)M, D, ms, 1.5 * bytes / ms / 1e6, 1.5 * bytes / ms / 1e6 / 62.9);
  }
}


// layout diagnosis of the pre-transposed-V attention kernels (`x2v_check dattn`): query r is aimed at key perm(r) (k = 4 q, every other score
// ~N(0,1)), V encodes first the key index, then the dv index — the printed tables show which key / which dv column really arrived where.
static void run_dattn() {
  Rng rng(5);
  const int H = 1;
  for (int64_t S : {64, 128, 200}) {
    auto q = rand_bf((size_t)S * 128, rng, 1.0f);
    std::vector<uint16_t> k(q.size()), v1(q.size()), v2(q.size());
    for (int64_t j = 0; j < S; ++j) {
      const int64_t r = (j * 37 + 11) % S;  // key j is the target of query r
      for (int e = 0; e < 128; ++e) {
        k[j * 128 + e] = f2bf(4.0f * bf2f(q[r * 128 + e]));
        v1[j * 128 + e] = f2bf((float)(j % 256));
        v2[j * 128 + e] = f2bf((float)e);
      }
    }
    DevBuf<uint16_t> dq(q), dk(k), dv1(v1), dv2(v2), dout((size_t)S * 128);
    for (int pass = 0; pass < 2; ++pass) {
      AttnVt vt;
      vt.prepare(pass ? dv2.p : dv1.p, 128, S, H);
      HIP_OK(hipMemset(dout.p, 0xff, dout.n * 2));
      attn_any(12, vt, dq.p, dk.p, nullptr, dout.p, S, S, H);
      HIP_OK(hipDeviceSynchronize());
      auto o = to_f(dout.host());
      int bad = 0;
      for (int64_t r = 0; r < S; ++r) {
        int64_t jt = -1;
        for (int64_t j = 0; j < S; ++j)
          if ((j * 37 + 11) % S == r) jt = j;
        for (int e = 0; e < 128; ++e) {
          const float want = pass ? (float)e : (float)(jt % 256), got = o[r * 128 + e];
          if (fabsf(got - want) > 0.51f && bad < 24) {
            printf("  dattn S=%lld pass=%d: o[%lld][%d] = %g, want %g (target key %lld)\n", (long long)S, pass, (long long)r, e, got, want, (long long)jt);
            ++bad;
          } else if (fabsf(got - want) > 0.51f) ++bad;
        }
      }
      char name[96];
      snprintf(name, sizeof name, "dattn S=%lld %s", (long long)S, pass ? "dv routing" : "key routing");
      ErrStat e;
      e.n = o.size();
      e.bad = bad;
      report(name, e);
    }
  }
}

// single-kernel loops for rocprofv3 --pmc passes: `x2v_check pattn <variant> <S> <H> [iters]`, `x2v_check pgemm <M> <N> <K> [iters]`
static void run_single(int argc, char** argv) {
  Rng rng(31);
  const std::string mode = argv[1];
  if (mode == "pattn") {
    const int variant = atoi(argv[2]);
    const int64_t S = atoll(argv[3]);
    const int H = atoi(argv[4]);
    const int iters = argc > 5 ? atoi(argv[5]) : 3;
    DevBuf<uint16_t> q((size_t)S * H * 128), k((size_t)S * H * 128), v((size_t)S * H * 128), o((size_t)S * H * 128);
    fill_random(q, rng, 1.f);
    fill_random(k, rng, 1.f);
    fill_random(v, rng, 1.f);
    AttnVt vt;
    vt.prepare(v.p, H * 128, S, H);
    double ms = time_ms(iters, [&] { attn_any(variant, vt, q.p, k.p, v.p, o.p, S, S, H); });
    printf("pattn variant=%d S=%lld H=%d: %.3f ms %.1f TFLOP/s\n", variant, (long long)S, H, ms, 4.0 * S * S * H * 128 / ms / 1e9);
    if (getenv("X2V_DUMP_TRACE")) {  // probe builds of attn_pc.hip (-DX2V_PC_TRACE) leave per-wave cycle sums in the first bytes of o
      std::vector<float> t(32);
      HIP_OK(hipMemcpy(t.data(), o.p, 32 * sizeof(float), hipMemcpyDeviceToHost));
      for (int w = 0; w < 8; ++w)
        printf("  wave %d: stream %.0f wait %.0f cycles over %.0f barriers (per interval: %.0f + %.0f), tail %.0f\n", w, t[w * 4], t[w * 4 + 1], t[w * 4 + 2], t[w * 4] / t[w * 4 + 2],
               t[w * 4 + 1] / t[w * 4 + 2], t[w * 4 + 3]);
    }
  } else {
    const int64_t M = atoll(argv[2]);
    const int N = atoi(argv[3]), K = atoi(argv[4]);
    const int iters = argc > 5 ? atoi(argv[5]) : 3;
    DevBuf<uint16_t> x((size_t)M * K), w((size_t)N * K), b(N), y((size_t)M * N);
    fill_random(x, rng, 1.f);
    fill_random(w, rng, 0.02f);
    fill_random(b, rng, 0.02f);
    const int variant = argc > 6 ? atoi(argv[6]) : 0;
    const int epi = argc > 7 ? atoi(argv[7]) : X2V_EPI_NONE;  // 2 = gate-residual (y doubles as resid), 1 = GELU
    DevBuf<uint16_t> gate(N);
    fill_random(gate, rng, 0.5f);
    if (epi == X2V_EPI_RESIDUAL) fill_random(y, rng, 1.f);
    double ms = time_ms(iters, [&] {
      X2V_OKAY(x2v_gemm_bf16_variant(x.p, K, w.p, K, b.p, y.p, N, M, N, K, epi, epi == X2V_EPI_RESIDUAL ? y.p : nullptr, N, epi == X2V_EPI_RESIDUAL ? gate.p : nullptr, variant, nullptr));
    });
    printf("pgemm variant=%d epi=%d M=%lld N=%d K=%d: %.3f ms %.1f TFLOP/s\n", variant, epi, (long long)M, N, K, ms, 2.0 * M * N * K / ms / 1e9);
  }
}

// `x2v_check pattnb <S> <H> [iters]`: the self-attention launch with K stored HEAD-BLOCKED [H][S][128] (a head's keys contiguous: row stride 256 B
// instead of H * 256 B), expressed through the batched entry (one "sequence" per head, H = 1 inside) — an experiment on whether the strided K
// rows are what the XCD-aware mapping stumbles over.
static void run_pattnb(int argc, char** argv) {
  Rng rng(31);
  const int64_t S = atoll(argv[2]);
  const int H = atoi(argv[3]);
  const int iters = argc > 4 ? atoi(argv[4]) : 3;
  const int64_t Sp = (S + 63) / 64 * 64;
  DevBuf<uint16_t> q((size_t)S * H * 128), kb((size_t)H * S * 128), v((size_t)S * H * 128), o((size_t)S * H * 128);
  fill_random(q, rng, 1.f);
  fill_random(kb, rng, 1.f);
  fill_random(v, rng, 1.f);
  AttnVt vt;
  vt.prepare(v.p, H * 128, S, H);
  double ms = time_ms(iters, [&] {
    X2V_OKAY(x2v_attn_fwd_bf16_vt_batched(q.p, H * 128, 128, kb.p, 128, S * 128, vt.vt->p, (int64_t)H * Sp, Sp * 128, o.p, H * 128, 128, S, S, 1, H, 128, 0.f, 0, nullptr));
  });
  printf("pattnb (K head-blocked) S=%lld H=%d: %.3f ms %.1f TFLOP/s\n", (long long)S, H, ms, 4.0 * S * S * H * 128 / ms / 1e9);
}

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "all";
  if (mode == "pattn" || mode == "pgemm") {
    X2V_OKAY(x2v_init(0));
    run_single(argc, argv);
    return 0;
  }
  if (mode == "pattnb") {
    X2V_OKAY(x2v_init(0));
    run_pattnb(argc, argv);
    return 0;
  }
  X2V_OKAY(x2v_init(0));
  char arch[64];
  int cus = 0, lds = 0;
  X2V_OKAY(x2v_device_info(0, &cus, &lds, arch, sizeof arch));
  printf("%s on %s, %d CUs, %d B LDS/CU — mode %s\n", x2v_version(), arch, cus, lds, mode.c_str());
  if (mode == "probe" || mode == "all") run_probe();
  if (mode == "misc" || mode == "all") run_errors();
  if (mode == "norm" || mode == "all") run_norm();
  if (mode == "rope" || mode == "all") run_rope();
  if (mode == "gemm" || mode == "all") run_gemm();
  if (mode == "attn" || mode == "all") run_attn();
  if (mode == "dattn") run_dattn();
  if (mode == "fp8" || mode == "all") run_fp8();
  if (mode == "conv" || mode == "all") run_conv();
  if (mode == "hbm") run_hbm();
  if (mode == "bench") run_bench(false);
  if (mode == "benchbig") run_bench(true);
  printf("x2v_check %s: %s (%d failing checks)\n", mode.c_str(), g_fail ? "FAILED" : "ALL PASS", g_fail);
  return g_fail ? 1 : 0;
}
