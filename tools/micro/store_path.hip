// Micro-benchmark: what does the SHAPE of a 16-byte-per-lane store instruction cost?  The GEMM epilogues write a 256 x 256 output tile per CU from
// MFMA accumulator layout: one wave instruction covers 16 rows x 64 B (fp16 outputs after the permlane swap, fp32 + residual alike) -- half of a
// 128-byte line per row, the other half coming with the next instruction.  Alternatives need a lane exchange first (rows l15 <-> l15 ^ 8): 8 rows x
// 128 B (full lines) or 4 rows x 256 B per instruction.  This kernel times the three shapes (stores only, and read-modify-write like the residual
// epilogue) with every CU writing its own 256-row x RB-byte tile of a [rows, row_stride] matrix, one 512-thread workgroup per CU.
//   hipcc --offload-arch=gfx950 -O3 -o store_path store_path.hip && ./store_path
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// SHAPE: rows per instruction = 16 / 8 / 4 (segment = 64 / 128 / 256 B).  RMW: load + add + store (fp32 residual pass) instead of store only.
// RMW with the loads of NB instructions issued before the first store (the in-order vmcnt counter makes "load next, store previous" wait for the
// PREVIOUS STORE'S acknowledgement at every step)
template <int RPI, int NB>
__global__ __launch_bounds__(512) void rmw_batched_kernel(char* __restrict__ out, size_t row_stride, int tile_bytes_per_row, int iters, long long* wall) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int SEG = 1024 / RPI;
  const int r_in = lane / (SEG / 16), c_in = (lane % (SEG / 16)) * 16;
  char* base = out + (size_t)blockIdx.x * 256 * row_stride + (size_t)(wave * 32) * row_stride;
  const int nrow_steps = 32 / RPI, ncol_steps = tile_bytes_per_row / SEG;
  const int total = nrow_steps * ncol_steps;
  f32x4 v = {1.f, 2.f, 3.f, (float)lane};
  const long long w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    for (int s0 = 0; s0 < total; s0 += NB) {
      f32x4 buf[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int st = s0 + j, cs = st / nrow_steps, rs = st - cs * nrow_steps;
        buf[j] = *(const f32x4*)(base + (size_t)(rs * RPI + r_in) * row_stride + (size_t)it * tile_bytes_per_row + cs * SEG + c_in);
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int st = s0 + j, cs = st / nrow_steps, rs = st - cs * nrow_steps;
        *(f32x4*)(base + (size_t)(rs * RPI + r_in) * row_stride + (size_t)it * tile_bytes_per_row + cs * SEG + c_in) = buf[j] + v;
      }
    }
    v[0] += 1.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long w1 = wall_clock64();
  if (tid == 0) wall[blockIdx.x] = w1 - w0;
}

template <int RPI, bool RMW>
__global__ __launch_bounds__(512) void store_kernel(char* __restrict__ out, size_t row_stride, int tile_bytes_per_row, int iters, long long* wall) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int SEG = 1024 / RPI;                        // contiguous bytes per row and instruction
  const int r_in = lane / (SEG / 16), c_in = (lane % (SEG / 16)) * 16;
  // the tile: 256 rows x tile_bytes_per_row; wave w owns rows [w * 32, w * 32 + 32); instructions sweep its rows, then the columns
  char* base = out + (size_t)blockIdx.x * 256 * row_stride + (size_t)(wave * 32) * row_stride;
  const int nrow_steps = 32 / RPI, ncol_steps = tile_bytes_per_row / SEG;
  f32x4 v = {1.f, 2.f, 3.f, (float)lane};
  const long long w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    for (int cs = 0; cs < ncol_steps; ++cs)
      for (int rs = 0; rs < nrow_steps; ++rs) {
        char* p = base + (size_t)(rs * RPI + r_in) * row_stride + (size_t)it * tile_bytes_per_row + cs * SEG + c_in;   // every pass a new column block: each line is touched once
        if (RMW) {
          f32x4 r = *(const f32x4*)p;
          *(f32x4*)p = r + v;
        } else {
          *(f32x4*)p = v;
        }
      }
    v[0] += 1.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long w1 = wall_clock64();
  if (tid == 0) wall[blockIdx.x] = w1 - w0;
}

template <int RPI, bool RMW>
static void run(const char* tag, char* buf, size_t row_stride, int tile_bytes_per_row, int iters, long long* dwall, int ncu) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((store_kernel<RPI, RMW>), dim3(ncu), dim3(512), 0, 0, buf, row_stride, tile_bytes_per_row, iters, dwall);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((store_kernel<RPI, RMW>), dim3(ncu), dim3(512), 0, 0, buf, row_stride, tile_bytes_per_row, iters, dwall);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> w(ncu);
  hipMemcpy(w.data(), dwall, ncu * sizeof(long long), hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto x : w) mean += (double)x;
  mean = mean / ncu * 0.01;                              // us (100 MHz)
  const double bytes = 256.0 * tile_bytes_per_row * iters * (RMW ? 2 : 1);
  printf("%-34s: %8.1f us per launch, %6.2f us per tile pass, %6.1f GB/s per CU, %6.2f TB/s chip (R+W)\n", tag, ms * 1e3, mean / iters,
         bytes / (mean * 1e-6) / 1e9, bytes * ncu / (ms * 1e-3) / 1e12);
}

template <int RPI, int NB>
static void run_b(const char* tag, char* buf, size_t row_stride, int tile_bytes_per_row, int iters, long long* dwall, int ncu) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((rmw_batched_kernel<RPI, NB>), dim3(ncu), dim3(512), 0, 0, buf, row_stride, tile_bytes_per_row, iters, dwall);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((rmw_batched_kernel<RPI, NB>), dim3(ncu), dim3(512), 0, 0, buf, row_stride, tile_bytes_per_row, iters, dwall);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> w(ncu);
  hipMemcpy(w.data(), dwall, ncu * sizeof(long long), hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto x : w) mean += (double)x;
  mean = mean / ncu * 0.01;
  const double bytes = 256.0 * tile_bytes_per_row * iters * 2;
  printf("%-34s: %8.1f us per launch, %6.2f us per tile pass, %6.1f GB/s per CU, %6.2f TB/s chip (R+W)\n", tag, ms * 1e3, mean / iters,
         bytes / (mean * 1e-6) / 1e9, bytes * ncu / (ms * 1e-3) / 1e12);
}

int main() {
  int ncu = 256;
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  const size_t rows = (size_t)ncu * 256;
  long long* dwall;
  hipMalloc(&dwall, ncu * sizeof(long long));
  // (a) fp16 output of a 256 x 256 tile: 512 B per row, row stride 8192 B (N = 4096 halfs); working set 256 CUs x 2 MiB
  // (b) fp32 + residual: 1024 B per row, row stride 4096 B (N = 1024 floats)
  for (int cfg = 0; cfg < 2; ++cfg) {
    const size_t stride = cfg == 0 ? 8192 : 4096;
    const int tb = cfg == 0 ? 512 : 1024;
    char* buf;
    hipMalloc(&buf, rows * stride);
    hipMemset(buf, 0, rows * stride);
    printf("== %s: tile 256 rows x %d B, row stride %zu B\n", cfg == 0 ? "fp16 tile (stores only)" : "fp32 tile", tb, stride);
    for (int rep = 0; rep < 2; ++rep) {
      const int it = (int)(stride / tb);                   // passes that cover every column block of the rows exactly once
      run<16, false>("store 16 rows x  64 B / instr", buf, stride, tb, it, dwall, ncu);
      run<8, false>("store  8 rows x 128 B / instr", buf, stride, tb, it, dwall, ncu);
      run<4, false>("store  4 rows x 256 B / instr", buf, stride, tb, it, dwall, ncu);
      if (cfg == 1) {
        run<16, true>("RMW   16 rows x  64 B / instr", buf, stride, tb, it, dwall, ncu);
        run<8, true>("RMW    8 rows x 128 B / instr", buf, stride, tb, it, dwall, ncu);
        run<4, true>("RMW    4 rows x 256 B / instr", buf, stride, tb, it, dwall, ncu);
        run_b<16, 4>("RMW 16x64B, 4 loads then 4 stores", buf, stride, tb, it, dwall, ncu);
        run_b<16, 8>("RMW 16x64B, 8 loads then 8 stores", buf, stride, tb, it, dwall, ncu);
        run_b<16, 16>("RMW 16x64B, 16 loads then 16 stores", buf, stride, tb, it, dwall, ncu);
        run_b<16, 32>("RMW 16x64B, 32 loads then 32 stores", buf, stride, tb, it, dwall, ncu);
        run_b<8, 16>("RMW 8x128B, 16 loads then 16 stores", buf, stride, tb, it, dwall, ncu);
      }
    }
    hipFree(buf);
  }
  return 0;
}
