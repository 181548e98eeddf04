// Micro-benchmark: how many bytes per clock does ONE CU pull from L2 / fabric when all CUs pull at once, by path -- LDS-DMA (global_load_lds_dwordx4,
// what the GEMM's K loop uses) vs plain global_load_dwordx4 into VGPRs (+ ds_write_b128), and mixes of the two.  The access pattern is the GEMM's:
// 1 KiB per wave instruction = 8 rows x 128 B at a row stride of 2 KiB (K = 1024 halfs), half of the instructions from a 67 MB "activation" matrix
// (rows private to the CU: streams from HBM / Infinity Cache), half from an 8 MB "weight" matrix (the same rows for every CU: L2 hits).
// One 512-thread workgroup per CU, one "K-tile" (8 instructions per wave = 64 KiB per CU) in flight behind the one being waited for.
//   hipcc --offload-arch=gfx950 -O3 -o cu_load_path cu_load_path.hip && ./cu_load_path
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GLDS16(gptr, ldsptr) __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gptr), (void __attribute__((address_space(3)))*)(ldsptr), 16, 0, 0)

template <int ND, int NV>
__global__ __launch_bounds__(512) void pull_kernel(const char* __restrict__ A, const char* __restrict__ W, int iters, long long* clk, float* sink, int rows_a) {
  __shared__ __attribute__((aligned(16))) char lds[2 * 65536 + 16384];   // two K-tile parities (+ pad: one workgroup per CU)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NI = ND + NV;                                            // 1 KiB instructions per wave and K-tile (8 = the GEMM)
  // instruction j of wave w covers rows (w * NI + j) * 8 .. +7 of the tile's 64 * NI rows; first half A rows, second half W rows
  const size_t stride = 2048;
  const int rows_half = rows_a;                                          // rows of the tile that are "activation" rows (0 .. 64 NI)
  size_t off[NI];
  const char* base[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int r = (wave * NI + j) * 8 + (lane >> 3);
    const bool isA = r < rows_half;
    base[j] = isA ? A + (size_t)blockIdx.x * rows_half * stride : W;
    off[j] = (size_t)(isA ? r : r - rows_half) * stride + (lane & 7) * 16;
  }
  f32x4 reg[NV > 0 ? NV : 1];
  float acc = 0.f;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    const int kt = it & 15;                                             // 16 K-tiles of 128 B along a row, then the same rows again (a new "tile")
    char* par = lds + (it & 1) * 65536;
    // wait for the previous K-tile (everything but nothing: it was issued one iteration ago), park its VGPR share in LDS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (it > 0) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        asm volatile("" : "+v"(reg[j]));
        *(f32x4*)(lds + ((it - 1) & 1) * 65536 + ((wave * NI + ND + j) * 64 + lane) * 16) = reg[j];
      }
    }
    __syncthreads();
    if (it > 1) acc += *(const float*)(lds + (it & 1) * 65536 + tid * 4);   // a token read of the parity about to be overwritten
#pragma unroll
    for (int j = 0; j < ND; ++j) GLDS16(base[j] + off[j] + kt * 128, par + (wave * NI + j) * 1024);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const char* p = base[ND + j] + off[ND + j] + kt * 128;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(reg[j]) : "v"(p) : "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long c1 = clock64(), w1 = wall_clock64();
#pragma unroll
  for (int j = 0; j < NV; ++j) { asm volatile("" : "+v"(reg[j])); acc += reg[j][0]; }
  sink[blockIdx.x * 512 + tid] = acc;
  if (tid == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int ND, int NV>
static void run(const char* A, const char* W, int ncu, long long* clk, float* sink, int grid = 0, int a_eighths = 4) {
  const int all_cu = ncu;
  if (grid) ncu = grid;
  const int rows_a = 64 * (ND + NV) * a_eighths / 8;
  const int iters = 16 * 64;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((pull_kernel<ND, NV>), dim3(ncu), dim3(512), 0, 0, A, W, 64, clk, sink, rows_a);
  hipEventRecord(e0);
  hipLaunchKernelGGL((pull_kernel<ND, NV>), dim3(ncu), dim3(512), 0, 0, A, W, iters, clk, sink, rows_a);
  hipEventRecord(e1);
  float ms; hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(2 * ncu);
  hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0; for (int i = 0; i < ncu; ++i) { cyc += (double)h[2 * i]; wall += (double)h[2 * i + 1]; }
  cyc /= ncu; wall /= ncu;
  const double bytes = (double)iters * 8 * (ND + NV) * 1024;            // per CU
  (void)all_cu;
  printf("%3d workgroups, %d/8 private rows | DMA %d + VGPR %d KiB per wave and K-tile (%3d KiB per CU): %7.1f us  %6.2f B/clk/CU  %6.1f GB/s/CU  %6.2f TB/s chip  (%.0f MHz)\n",
         ncu, a_eighths, ND, NV, 8 * (ND + NV), ms * 1e3, bytes / cyc, bytes / (ms * 1e-3) / 1e9, bytes * ncu / (ms * 1e-3) / 1e12, cyc / (wall / 100.0));
}

int main() {
  int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  char *A, *W; float* sink; long long* clk;
  const size_t abytes = (size_t)ncu * 512 * 2048 + (1 << 20), wbytes = 16 << 20;
  hipMalloc(&A, abytes); hipMalloc(&W, wbytes); hipMalloc(&sink, (size_t)ncu * 512 * 4); hipMalloc(&clk, (size_t)ncu * 16);
  hipMemset(A, 1, abytes); hipMemset(W, 2, wbytes);
  for (int rep = 0; rep < 2; ++rep) {
    run<8, 0>(A, W, ncu, clk, sink);
    run<7, 1>(A, W, ncu, clk, sink);
    run<4, 4>(A, W, ncu, clk, sink);
    run<0, 8>(A, W, ncu, clk, sink);
    run<4, 0>(A, W, ncu, clk, sink);
  }
  for (int ae : {0, 4, 8})                                             // all shared (L2 hits) / half / all private (streaming)
    for (int grid : {256, 128, 64, 32, 8}) run<8, 0>(A, W, ncu, clk, sink, grid, ae);
  for (int grid : {256, 64, 8}) run<0, 8>(A, W, ncu, clk, sink, grid, 0);
  return 0;
}
