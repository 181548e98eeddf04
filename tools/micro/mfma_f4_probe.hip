// Probe of the MX-fp4 path of v_mfma_scale_f32_16x16x128_f8f6f4 on gfx950 (cbsz = blgp = 4) for the 4-bit "lo pass" of the strict mode:
//  (1) operand layout: lane l holds row (l % 16) and the 32 consecutive K elements 32*(l/16) .. +31 as 16 bytes (element 2j = low nibble of byte j);
//  (2) scales: per LANE E8M0 byte (row l%16, K block l/16) picked by op_sel, products * 2^(sa-127) * 2^(sb-127) -- tested with scales that vary
//      by row and by K block on both operands;
//  (3) v_cvt_scalef32_pk_fp4_f32: rounding and the direction of the scale.
// hipcc --offload-arch=gfx950 -O3 -o mfma_f4_probe mfma_f4_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const uint8_t* A, const uint8_t* B, const uint8_t* SA, const uint8_t* SB, float* D) {
  const int l = threadIdx.x, row = l % 16, kb = l / 16;
  const i32x4 a4 = *(const i32x4*)(A + row * 64 + kb * 16), b4 = *(const i32x4*)(B + row * 64 + kb * 16);
  const i32x8 a = __builtin_shufflevector(a4, a4, 0, 1, 2, 3, -1, -1, -1, -1), b = __builtin_shufflevector(b4, b4, 0, 1, 2, 3, -1, -1, -1, -1);
  // scale VGPRs: byte 2 of sa_reg / byte 1 of sb_reg carry the scale (op_sel 2 / 1), the other bytes are poison
  const int sa = 0x11003322 | ((int)SA[row * 4 + kb] << 16), sb = 0x55660077 | ((int)SB[row * 4 + kb] << 8);
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4 /*A: fp4*/, 4 /*B: fp4*/, 2, sa, 1, sb);
  for (int r = 0; r < 4; ++r) D[(4 * kb + r) * 16 + row] = c[r];
}

__global__ void cvt_probe(const float* x, const float* sc, uint32_t* out, int n) {
  const int i = threadIdx.x;
  if (i < n) out[i] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(0u, x[2 * i], x[2 * i + 1], sc[i], 0);
}

static const float F4[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
static float f4(uint8_t c) { float v = F4[c & 7]; return (c & 8) ? -v : v; }

int main() {
  std::vector<uint8_t> A(16 * 64), B(16 * 64), SA(64), SB(64);
  unsigned s = 777;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 16) & 0xffff; };
  for (auto& x : A) x = rnd() & 0xff;
  for (auto& x : B) x = rnd() & 0xff;
  for (auto& x : SA) x = 120 + rnd() % 12;
  for (auto& x : SB) x = 118 + rnd() % 12;
  uint8_t *dA, *dB, *dSA, *dSB; float* dD;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dSA, 64); hipMalloc(&dSB, 64); hipMalloc(&dD, 1024);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  hipMemcpy(dSA, SA.data(), 64, hipMemcpyHostToDevice); hipMemcpy(dSB, SB.data(), 64, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dSA, dSB, dD);
  std::vector<float> D(256);
  hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
  auto el = [&](const std::vector<uint8_t>& M, int r, int k) { const uint8_t b = M[r * 64 + k / 2]; return f4((k & 1) ? (b >> 4) : (b & 15)); };
  double e1 = 0, e2 = 0, mx = 0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double r1 = 0, r2 = 0;
      for (int k = 0; k < 128; ++k) {
        r1 += (double)el(A, i, k) * el(B, j, k) * ldexp(1.0, SA[i * 4 + k / 32] - 127) * ldexp(1.0, SB[j * 4 + k / 32] - 127);   // D[i][j]: first operand = rows
        r2 += (double)el(A, j, k) * el(B, i, k) * ldexp(1.0, SA[j * 4 + k / 32] - 127) * ldexp(1.0, SB[i * 4 + k / 32] - 127);
      }
      e1 = fmax(e1, fabs(D[i * 16 + j] - r1)); e2 = fmax(e2, fabs(D[i * 16 + j] - r2)); mx = fmax(mx, fabs(r1));
    }
  printf("fp4 MFMA, per-lane block scales: max |D - ref|: D[i][j] = sum A[i][k] B[j][k]: %.3e ; D[i][j] = sum A[j][k] B[i][k]: %.3e (max |ref| %.3e)\n", e1, e2, mx);

  // converter
  const int n = 32;
  std::vector<float> x(2 * n), sc(n);
  for (int i = 0; i < n; ++i) { x[2 * i] = (i - 14) * 0.25f; x[2 * i + 1] = -(i - 3) * 0.125f; sc[i] = 1.0f; }
  x[60] = 3.0f; x[61] = 0.75f; sc[30] = 4.0f;      // scale 4: 3 / 4 = 0.75 -> 1.0 (tie to even 0.5|1.0?) or 3 * 4 = 12 -> 6
  x[62] = 3.0f; x[63] = 0.76f; sc[31] = 0.25f;
  float *dx, *dsc; uint32_t* dout;
  hipMalloc(&dx, 8 * n); hipMalloc(&dsc, 4 * n); hipMalloc(&dout, 4 * n);
  hipMemcpy(dx, x.data(), 8 * n, hipMemcpyHostToDevice); hipMemcpy(dsc, sc.data(), 4 * n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(cvt_probe, dim3(1), dim3(64), 0, 0, dx, dsc, dout, n);
  std::vector<uint32_t> out(n);
  hipMemcpy(out.data(), dout, 4 * n, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i)
    printf("cvt_scalef32_pk_fp4(%7.3f, %7.3f; scale %5.2f) -> 0x%08x = (%5.2f, %5.2f)\n", x[2 * i], x[2 * i + 1], sc[i], out[i], f4(out[i] & 15), f4((out[i] >> 4) & 15));
  return 0;
}
