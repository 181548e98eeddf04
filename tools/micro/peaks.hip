// Micro-benchmark: what MI355X actually sustains -- dense fp16 MFMA rate (register-resident, no memory), the shader clock
// during that load, and HBM copy / read bandwidth.  Standalone: hipcc --offload-arch=gfx950 -O3 -o peaks peaks.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(512, 2) void mfma_kernel(float* out, int iters, long long* clk) {
  h16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[4 * blockIdx.x] = c1 - c0; clk[4 * blockIdx.x + 1] = w1 - w0; clk[4 * blockIdx.x + 2] = w0; clk[4 * blockIdx.x + 3] = w1; }
}

// one wave per SIMD, 64 accumulators kept in AGPRs by inline asm (the shape of gemm_w4's K loop without any memory traffic)
__global__ __launch_bounds__(256) void mfma_agpr_kernel(float* out, int iters, long long* clk) {
  h16x8 a[8], b[8];
  for (int q = 0; q < 8; ++q) for (int i = 0; i < 8; ++i) { a[q][i] = (_Float16)(0.001f * (threadIdx.x + i + q)); b[q][i] = (_Float16)(0.002f * (threadIdx.x - i - q)); }
  f32x4 acc[8][8];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(a[i]), "v"(b[j]));
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[4 * blockIdx.x] = c1 - c0; clk[4 * blockIdx.x + 1] = w1 - w0; clk[4 * blockIdx.x + 2] = w0; clk[4 * blockIdx.x + 3] = w1; }
}

__global__ void copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void read_kernel(const float4* __restrict__ src, float* out, size_t n) {
  float4 s = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = src[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
  if (s.x + s.y + s.z + s.w == 123.456f) out[0] = 1.f;
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); return ms; }

int main() {
  int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float* out; hipMalloc(&out, (size_t)ncu * 512 * 4);
  long long* clk; hipMalloc(&clk, (size_t)ncu * 32);
  std::vector<long long> h(4 * ncu);
  printf("CUs: %d\n", ncu);
  for (int iters : {2000, 20000, 200000}) {          // ~0.1 ms, ~1 ms, ~10 ms of MFMA: the clock settles with duration
    constexpr int NACC = 8;
    hipLaunchKernelGGL(mfma_kernel<NACC>, dim3(ncu), dim3(512), 0, 0, out, 100, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_kernel<NACC>, dim3(ncu), dim3(512), 0, 0, out, iters, clk);
    hipEventRecord(e1);
    const float ms = time_ms(e0, e1);
    hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    double mhz = 0; long long wmin = h[2], wmax = h[3], smax = h[2]; double dur = 0;
    for (int i = 0; i < ncu; ++i) { mhz += (double)h[4 * i] / ((double)h[4 * i + 1] / 100.0); if (h[4*i+2] < wmin) wmin = h[4*i+2]; if (h[4*i+2] > smax) smax = h[4*i+2]; if (h[4*i+3] > wmax) wmax = h[4*i+3]; dur += h[4*i+1] / 100.0; }
    mhz /= ncu; dur /= ncu;
    printf("  workgroups: mean duration %.1f us, start spread %.1f us, kernel span %.1f us\n", dur, (smax - wmin) / 100.0, (wmax - wmin) / 100.0);
    const double flops = (double)ncu * 8 /*waves*/ * iters * NACC * 2.0 * 16 * 16 * 32;
    const double clk_per_mfma = ms * 1e3 * mhz / ((double)iters * NACC * 2 /*waves per SIMD*/);   // over the whole launch: the SIMD serves its older wave first
    printf("MFMA f32_16x16x32_f16, %7d iters: %8.3f ms  %7.1f TFLOP/s  shader clock %6.0f MHz  %.2f clk per MFMA per SIMD  (peak at that clock %.0f TFLOP/s)\n",
           iters, ms, flops / ms / 1e9, mhz, clk_per_mfma, ncu * 4 * 1024.0 * mhz * 1e6 / 1e12);
  }
  for (int iters : {400, 4000}) {
    hipLaunchKernelGGL(mfma_agpr_kernel, dim3(ncu), dim3(256), 0, 0, out, 10, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_agpr_kernel, dim3(ncu), dim3(256), 0, 0, out, iters, clk);
    hipEventRecord(e1);
    const float ms = time_ms(e0, e1);
    hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    double mhz = 0; for (int i = 0; i < ncu; ++i) mhz += (double)h[4 * i] / ((double)h[4 * i + 1] / 100.0); mhz /= ncu;
    const double flops = (double)ncu * 4 * iters * 64 * 2.0 * 16 * 16 * 32;
    printf("MFMA one wave/SIMD, 64 AGPR accumulators, %5d iters: %8.3f ms  %7.1f TFLOP/s  shader clock %6.0f MHz  %.2f clk per MFMA per SIMD\n",
           iters, ms, flops / ms / 1e9, mhz, ms * 1e3 * mhz / ((double)iters * 64));
  }
  const size_t bytes = (size_t)4 << 30;
  float4 *src, *dst; hipMalloc(&src, bytes); hipMalloc(&dst, bytes);
  hipMemset(src, 1, bytes); hipMemset(dst, 0, bytes);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL(copy_kernel, dim3(ncu * 8), dim3(512), 0, 0, src, dst, bytes / 16); hipEventRecord(e1);
    const float ms = time_ms(e0, e1);
    printf("HBM copy 4 GiB -> 4 GiB: %.3f ms  %.2f TB/s (read + write)\n", ms, 2.0 * bytes / ms / 1e9);
    hipEventRecord(e0); hipLaunchKernelGGL(read_kernel, dim3(ncu * 8), dim3(512), 0, 0, src, out, bytes / 16); hipEventRecord(e1);
    const float ms2 = time_ms(e0, e1);
    printf("HBM read 4 GiB: %.3f ms  %.2f TB/s\n", ms2, bytes / ms2 / 1e9);
  }
  return 0;
}
