// Micro-benchmark: register-resident dense fp16 MFMA rate and shader clock for the two gfx950 shapes (16x16x32 and 32x32x16:
// the same flops per clock on paper, half the operand-register reads per flop for 32x32).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void mfma_kernel(float* out, int iters, long long* clk, int zeros) {
  h16x8 a[2], b[2];
  unsigned s = threadIdx.x * 2654435761u + 12345u;
  for (int q = 0; q < 2; ++q)
    for (int i = 0; i < 8; ++i) {
      s = s * 1664525u + 1013904223u; a[q][i] = (_Float16)(((int)(s >> 16) - 32768) * (1.0f / 65536.0f));
      s = s * 1664525u + 1013904223u; b[q][i] = (_Float16)(((int)(s >> 16) - 32768) * (1.0f / 65536.0f));
    }
  i32x8 a8[2], b8[2];
  for (int q = 0; q < 2; ++q)
    for (int i = 0; i < 8; ++i) { s = s * 1664525u + 1013904223u; a8[q][i] = (int)(s & 0x77777777u); s = s * 1664525u + 1013904223u; b8[q][i] = (int)(s & 0x77777777u); }
  if (zeros) {                                            // all-zero operands: what the clock does when the multipliers do not toggle
    for (int q = 0; q < 2; ++q) for (int i = 0; i < 8; ++i) { a[q][i] = (_Float16)0.f; b[q][i] = (_Float16)0.f; a8[q][i] = 0; b8[q][i] = 0; }
  }
  f32x4 acc4[16];
  f32x16 acc16[4];
  for (int i = 0; i < 16; ++i) acc4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc16[i][j] = 0.f;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if (SHAPE == 16) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 1], b[(i >> 1) & 1], acc4[i], 0, 0, 0);
    } else if (SHAPE == 128) {                          // e4m3 x e4m3, K = 128: 4x the flops of one 16x16x32 f16 MFMA
#pragma unroll
      for (int i = 0; i < 16; ++i) acc4[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[i & 1], b8[(i >> 1) & 1], acc4[i], 0, 0, 0, 127, 0, 127);
    } else if (SHAPE == 129) {                          // fp4 x fp4 (cbsz = blgp = 4), K = 128: the weight-correction pass's instruction
#pragma unroll
      for (int i = 0; i < 16; ++i) acc4[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[i & 1], b8[(i >> 1) & 1], acc4[i], 4, 4, 0, 127, 0, 127);
    } else if (SHAPE == 65) {                           // fp4 x fp4, 32x32x64
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc16[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i & 1], b8[(i >> 1) & 1], acc16[i], 4, 4, 0, 127, 0, 127);
    } else {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc16[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 1], b[(i >> 1) & 1], acc16[i], 0, 0, 0);
    }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float r = 0.f;
  for (int i = 0; i < 16; ++i) r += acc4[i][0] + acc4[i][3];
  for (int i = 0; i < 4; ++i) r += acc16[i][0] + acc16[i][15];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

int main() {
  int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float* out; hipMalloc(&out, (size_t)ncu * 512 * 4);
  long long* clk; hipMalloc(&clk, (size_t)ncu * 16);
  std::vector<long long> h(2 * ncu);
  // zeros = 1 reconciles this file with /opt/skills/guides/MI355X_MICROARCH.md (2495 TFLOP/s "measured", 32x32x16): the guide's own DVFS note
  // says zero-filled inputs clock ~20 % higher than random ones.  Random operands are what a GEMM sees.
  for (int zeros = 0; zeros < 2; ++zeros)
  for (int rep = 0; rep < 3; ++rep)
    for (int shape : {16, 32, 128, 129, 65}) {
      const int iters = 20000;                        // x 16 (or 8) MFMAs: 262144 flop-units per wave either way
      auto launch = [&](int n) {
        if (shape == 16) hipLaunchKernelGGL(mfma_kernel<16>, dim3(ncu), dim3(512), 0, 0, out, n, clk, zeros);
        else if (shape == 128) hipLaunchKernelGGL(mfma_kernel<128>, dim3(ncu), dim3(512), 0, 0, out, n, clk, zeros);
        else if (shape == 129) hipLaunchKernelGGL(mfma_kernel<129>, dim3(ncu), dim3(512), 0, 0, out, n, clk, zeros);
        else if (shape == 65) hipLaunchKernelGGL(mfma_kernel<65>, dim3(ncu), dim3(512), 0, 0, out, n, clk, zeros);
        else hipLaunchKernelGGL(mfma_kernel<32>, dim3(ncu), dim3(512), 0, 0, out, n, clk, zeros);
      };
      launch(1000);
      hipEventRecord(e0); launch(iters); hipEventRecord(e1);
      float ms; hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
      double mhz = 0; for (int i = 0; i < ncu; ++i) mhz += (double)h[2 * i] / ((double)h[2 * i + 1] / 100.0); mhz /= ncu;
      const double flops = (double)ncu * 8 * iters * 16 * 2.0 * 16 * 16 * (shape >= 128 ? 128 : shape == 65 ? 128 : 32);   // (32x32x64 x 8 per iteration = 16x16x128 x 16)
      printf("%s %s: %8.3f ms  %7.1f TFLOP/s  shader clock %6.0f MHz  (peak at that clock %.0f TFLOP/s)\n",
             zeros ? "[zero operands]  " : "[random operands]", shape == 16 ? "v_mfma_f32_16x16x32_f16" : shape == 128 ? "v_mfma_scale_f32_16x16x128 e4m3" : shape == 129 ? "v_mfma_scale_f32_16x16x128 fp4" : shape == 65 ? "v_mfma_scale_f32_32x32x64 fp4" : "v_mfma_f32_32x32x16_f16", ms, flops / ms / 1e9, mhz, ncu * 4 * 1024.0 * mhz * 1e6 / 1e12);
    }
  return 0;
}
