// Micro-benchmark (round 5, review item 2a): VALU cost of the exact-erf GELU of the FFN-up epilogue in three formulations, register-resident,
// 8 waves per CU on every CU (the epilogue's occupancy), and their error against erf in double.
//   0: the product's form -- Abramowitz-Stegun 7.1.26: t = rcp(1 + p a), degree-5 polynomial in t, exp2(-a^2 log2e / 2): 2 transcendentals
//   1: no transcendental: phi(a) = a erfc(a / sqrt2) / 2 as a degree-16 polynomial on the clamped argument [0, 5.5] (Chebyshev interpolant,
//      Horner in t = a (2 / 5.5) - 1; |error| <= 2.5e-7 in fp32 arithmetic), gelu(x) = max(x, 0) - phi(|x|)
//   2: the same polynomial evaluated as two interleaved halves in t^2 (Estrin-style even / odd split: shorter dependency chain, same operation count)
// hipcc --offload-arch=gfx950 -O3 tools/micro/gelu_rate.hip -o tools/micro/gelu_rate
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 gelu_as(f32x2 x) {
  const float a0 = fabsf(x.x), a1 = fabsf(x.y);
  const f32x2 t = {__builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, a0, 1.0f)), __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, a1, 1.0f))};
  f32x2 p = __builtin_elementwise_fma(t, (f32x2)(0.5f * 1.061405429f), (f32x2)(-0.5f * 1.453152027f));
  p = __builtin_elementwise_fma(p, t, (f32x2)(0.5f * 1.421413741f));
  p = __builtin_elementwise_fma(p, t, (f32x2)(-0.5f * 0.284496736f));
  p = __builtin_elementwise_fma(p, t, (f32x2)(0.5f * 0.254829592f));
  p = p * t;
  const f32x2 w = (x * x) * (f32x2)(-0.5f * 1.4426950408889634f);
  const f32x2 e = {__builtin_amdgcn_exp2f(w.x), __builtin_amdgcn_exp2f(w.y)};
  const f32x2 h = p * e;
  return (f32x2){fmaf(-a0, h.x, fmaxf(x.x, 0.0f)), fmaf(-a1, h.y, fmaxf(x.y, 0.0f))};
}
__device__ constexpr float PC[17] = {8.194348896e-03f, -6.057822884e-02f, 1.912661398e-01f, -3.086717827e-01f, 1.788887135e-01f, 2.603210509e-01f,
                                     -6.278478086e-01f, 4.601556816e-01f, 1.495049154e-01f, -5.463876933e-01f, 2.762417609e-01f, 2.338470182e-01f,
                                     -2.616878095e-01f, -3.841818385e-02f, 1.018768775e-01f, -2.678953359e-04f, -1.643713241e-02f};
__device__ __forceinline__ f32x2 gelu_poly(f32x2 x) {
  const f32x2 a = {fminf(fabsf(x.x), 5.5f), fminf(fabsf(x.y), 5.5f)};
  const f32x2 t = __builtin_elementwise_fma(a, (f32x2)(2.0f / 5.5f), (f32x2)(-1.0f));
  f32x2 p = (f32x2)(PC[16]);
#pragma unroll
  for (int k = 15; k >= 0; --k) p = __builtin_elementwise_fma(p, t, (f32x2)(PC[k]));
  return (f32x2){fmaxf(x.x, 0.0f) - p.x, fmaxf(x.y, 0.0f) - p.y};
}
__device__ __forceinline__ f32x2 gelu_poly2(f32x2 x) {
  const f32x2 a = {fminf(fabsf(x.x), 5.5f), fminf(fabsf(x.y), 5.5f)};
  const f32x2 t = __builtin_elementwise_fma(a, (f32x2)(2.0f / 5.5f), (f32x2)(-1.0f));
  const f32x2 u = t * t;
  f32x2 pe = (f32x2)(PC[16]), po = (f32x2)(PC[15]);
#pragma unroll
  for (int k = 14; k >= 0; k -= 2) pe = __builtin_elementwise_fma(pe, u, (f32x2)(PC[k]));
#pragma unroll
  for (int k = 13; k >= 1; k -= 2) po = __builtin_elementwise_fma(po, u, (f32x2)(PC[k]));
  const f32x2 p = __builtin_elementwise_fma(po, t, pe);
  return (f32x2){fmaxf(x.x, 0.0f) - p.x, fmaxf(x.y, 0.0f) - p.y};
}

template <int V>
__global__ __launch_bounds__(512, 2) void rate_kernel(float* out, int iters) {
  f32x2 v[32];                                            // 64 values per lane, as one quarter of a tile's accumulators
  for (int i = 0; i < 32; ++i) v[i] = (f32x2){(float)((threadIdx.x * 37 + i * 11) % 997) * 0.012f - 6.0f, (float)((threadIdx.x * 53 + i * 7) % 991) * 0.012f - 6.0f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      f32x2 r = V == 0 ? gelu_as(v[i]) : V == 1 ? gelu_poly(v[i]) : gelu_poly2(v[i]);
      v[i] = r * (f32x2)(3.0f) - (f32x2)(1.5f);           // keeps the values spread and the chain alive (2 extra packed operations per pair, all variants)
    }
  }
  float s = 0.f;
  for (int i = 0; i < 32; ++i) s += v[i].x + v[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int V>
__global__ void err_kernel(const float* x, float* y, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { f32x2 r = V == 0 ? gelu_as((f32x2){x[i], x[i]}) : V == 1 ? gelu_poly((f32x2){x[i], x[i]}) : gelu_poly2((f32x2){x[i], x[i]}); y[i] = r.x; }
}

int main() {
  int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  float* out; hipMalloc(&out, (size_t)ncu * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep)
    for (int v = 0; v < 3; ++v) {
      auto launch = [&](int n) {
        if (v == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(ncu), dim3(512), 0, 0, out, n);
        else if (v == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(ncu), dim3(512), 0, 0, out, n);
        else hipLaunchKernelGGL(rate_kernel<2>, dim3(ncu), dim3(512), 0, 0, out, n);
      };
      launch(100);
      hipEventRecord(e0); launch(iters); hipEventRecord(e1);
      float ms; hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      const double vals = (double)iters * 64 * 512;       // per CU
      printf("variant %d: %8.3f ms  -> %.3f ns per 65 536 values per CU-tile-equivalent: %.2f us  (%.1f ps per value per CU)\n", v, ms, ms * 1e6 / vals * 65536.0 / 1e3 * 1e3,
             ms * 1e3 / vals * 65536.0 * 2, ms * 1e9 / vals);
    }
  // error against erf in double over [-8, 8]
  const int n = 1 << 20;
  std::vector<float> hx(n), hy(n);
  for (int i = 0; i < n; ++i) hx[i] = -8.0f + 16.0f * i / (n - 1);
  float *dx, *dy; hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4);
  hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
  for (int v = 0; v < 3; ++v) {
    if (v == 0) hipLaunchKernelGGL(err_kernel<0>, dim3(n / 256), dim3(256), 0, 0, dx, dy, n);
    else if (v == 1) hipLaunchKernelGGL(err_kernel<1>, dim3(n / 256), dim3(256), 0, 0, dx, dy, n);
    else hipLaunchKernelGGL(err_kernel<2>, dim3(n / 256), dim3(256), 0, 0, dx, dy, n);
    hipMemcpy(hy.data(), dy, n * 4, hipMemcpyDeviceToHost);
    double me = 0, mr = 0;
    for (int i = 0; i < n; ++i) {
      const double x = hx[i], ref = 0.5 * x * (1.0 + erf(x / sqrt(2.0)));
      const double e = fabs((double)hy[i] - ref);
      if (e > me) me = e;
      if (fabs(ref) > 1e-3 && e / fabs(ref) > mr) mr = e / fabs(ref);
    }
    printf("variant %d: max |error| %.3e, max relative error where |gelu| > 1e-3: %.3e\n", v, me, mr);
  }
  return 0;
}
