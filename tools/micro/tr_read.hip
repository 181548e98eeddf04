// Micro-probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds u16 value = element index.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(uint16_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int lane = threadIdx.x;
  unsigned addr;
  if (mode == 0) addr = lane * 8;                                   // contiguous 8 B per lane
  else if (mode == 1) addr = (lane & 15) * 8 + (lane >> 4) * 512;   // each 16-lane group on its own 128-B block
  else addr = ((lane & 15) >> 2) * 128 + ((lane & 3) * 8) + (lane >> 4) * 1024;  // 4 rows (stride 128 B) x 4 lanes of 8 B per group
  addr += (unsigned)(uintptr_t)lds;   // LDS base is 0 for the first static array; keep it general
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[lane * 4 + 0] = v.x & 0xffff; out[lane * 4 + 1] = v.x >> 16; out[lane * 4 + 2] = v.y & 0xffff; out[lane * 4 + 3] = v.y >> 16;
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  uint16_t h[256];
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) { printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); }
  }
  return 0;
}
