// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 on gfx950 (for a future e4m3 "lo pass" of the split-activation GEMMs): operand layout and scale
// semantics.  Hypothesis: lane l holds row (l % 16) and the 32 consecutive K bytes 32*(l/16) .. 32*(l/16)+31 of a 16 x 128 e4m3 operand (8 VGPRs);
// D as for 16x16x32_f16: lane l holds D[4*(l/16) + r][l % 16], r = 0..3; E8M0 scales multiply the products by 2^(sa-127) * 2^(sb-127).
// hipcc --offload-arch=gfx950 -O3 -o mfma_f8_probe mfma_f8_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const uint8_t* A, const uint8_t* B, float* D, int sa, int sb) {
  const int l = threadIdx.x, row = l % 16, kb = l / 16;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = ((const int*)(A + row * 128 + kb * 32))[i]; b[i] = ((const int*)(B + row * 128 + kb * 32))[i]; }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0 /*A: fp8 e4m3*/, 0 /*B: fp8 e4m3*/, 0, sa, 0, sb);
  for (int r = 0; r < 4; ++r) D[(4 * kb + r) * 16 + row] = c[r];
}

static float e4m3_to_f(uint8_t v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -f : f;
}

int main() {
  std::vector<uint8_t> A(16 * 128), B(16 * 128);
  unsigned s = 12345;
  for (auto* v : {&A, &B})
    for (auto& x : *v) { s = s * 1664525u + 1013904223u; uint8_t b = (s >> 24) & 0xff; if (((b >> 3) & 15) == 15 && (b & 7) == 7) b ^= 1; if (((b >> 3) & 15) > 9) b &= ~0x40; x = b; }
  uint8_t *dA, *dB; float* dD;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 256 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  for (int sc : {127, 120}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, sc, 127);
    std::vector<float> D(256);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    // hypothesis 1: D[i][j] = sum_k A[j][k] * B[i][k]  (first operand = columns, as with the f16 shape in this repo) ; hypothesis 2: transposed
    double e1 = 0, e2 = 0, mx = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double r1 = 0, r2 = 0;
        for (int k = 0; k < 128; ++k) { r1 += (double)e4m3_to_f(A[j * 128 + k]) * e4m3_to_f(B[i * 128 + k]); r2 += (double)e4m3_to_f(A[i * 128 + k]) * e4m3_to_f(B[j * 128 + k]); }
        const double sc2 = ldexp(1.0, sc - 127);
        e1 = fmax(e1, fabs(D[i * 16 + j] - r1 * sc2)); e2 = fmax(e2, fabs(D[i * 16 + j] - r2 * sc2)); mx = fmax(mx, fabs(r1));
      }
    printf("scale_a = %d: max |D - ref| with D[i][j] = sum_k A[j][k] B[i][k]: %.3e ; with D[i][j] = sum_k A[i][k] B[j][k]: %.3e  (max |ref| %.3e)\n", sc, e1, e2, mx);
  }
  return 0;
}
