// Which lane-address patterns of ds_read_b128 are free of LDS bank conflicts on gfx950?  Times a loop of independent 16-byte fragment reads for
// a few layouts of a 16-row fragment (lane = (row l15 = lane & 15, K group g = lane >> 4)):
//   0  rows of 128 B, slot (ks * 4 + g) ^ ((row >> 1) & 7)            -- the fp16 K-tiles' layout (PMC: 0 conflicts)
//   1  rows of  64 B, chunk g ^ ((row >> 2) & 3)                       -- the mini-tiles' first layout (PMC: 2-way conflicts)
//   2  rows of  64 B, no swizzle
//   3  rows of  64 B, chunk g ^ ((row >> 1) & 3)
//   4  rows of  64 B, chunk g ^ (row & 3)
//   5  rows of  64 B, chunk (g + (row >> 2)) & 3 ... (rotation)
//   6  rows of  64 B, chunk g ^ ((row >> 2) & 3) ^ ((row & 1) << 1)
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/lds_b128.hip -o tools/micro/lds_b128 ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int P>
__global__ __launch_bounds__(512) void k(int iters, int* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, g = lane >> 4;
  for (int i = threadIdx.x; i < 16384; i += 512) ((int*)smem)[i] = i;
  __syncthreads();
  int off[8];
  for (int t = 0; t < 8; ++t) {
    const int base = (wave * 8 + t) * 1024 % 65536;
    int o;
    if (P == 0) o = l15 * 128 + ((((t & 1) * 4 + g) ^ ((l15 >> 1) & 7)) * 16);
    else if (P == 7) o = l15 * 128 + ((((t & 1) * 4 + g) ^ (l15 & 7)) * 16);
    else if (P == 8) o = l15 * 128 + ((((t & 1) * 4 + g) ^ ((l15 >> 2) & 3) ^ ((l15 & 1) << 2)) * 16);
    else if (P == 9) o = l15 * 128 + (((t & 1) * 4 + g) * 16);
    else if (P == 10) o = l15 * 128 + ((((t & 1) * 4 + g) ^ ((l15 >> 1) & 3)) * 16);
    else {
      int c = g;
      if (P == 1) c = g ^ ((l15 >> 2) & 3);
      if (P == 3) c = g ^ ((l15 >> 1) & 3);
      if (P == 4) c = g ^ (l15 & 3);
      if (P == 5) c = (g + (l15 >> 2)) & 3;
      if (P == 6) c = g ^ ((l15 >> 2) & 3) ^ ((l15 & 1) << 1);
      o = l15 * 64 + c * 16;
    }
    off[t] = (base + o) & 65535 & ~15;
  }
  i32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 8; ++t) { const i32x4 v = *(const i32x4*)(smem + off[t]); acc += v; }
    asm volatile("" : "+v"(acc));
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 0x7fffffff) out[0] = 1;
}

template <int P>
static void run(const char* name) {
  int* out; hipMalloc(&out, 4);
  hipFuncSetAttribute((const void*)k<P>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<P><<<256, 512, 65536>>>(100, out);
  hipEventRecord(a);
  k<P><<<256, 512, 65536>>>(20000, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("pattern %d (%s): %.3f ms for 20000 x 8 reads per wave, 8 waves per CU = %.2f ns per wave-read per CU\n", P, name, ms, ms * 1e6 / (20000.0 * 8 * 8));
}
int main() {
  run<0>("128-B rows, fp16 K-tile layout"); run<1>("64-B rows, chunk ^ (row >> 2)"); run<2>("64-B rows, no swizzle"); run<3>("64-B rows, chunk ^ (row >> 1)");
  run<7>("128-B rows, slot ^ row"); run<8>("128-B rows, slot ^ (row >> 2) ^ ((row & 1) << 2)"); run<9>("128-B rows, no swizzle"); run<10>("128-B rows, slot ^ ((row >> 1) & 3)");
  run<4>("64-B rows, chunk ^ row"); run<5>("64-B rows, rotate by row >> 2"); run<6>("64-B rows, chunk ^ (row >> 2) ^ ((row & 1) << 1)");
  return 0;
}
