// Which CUs serve a launch on a given stream?  Every workgroup spins ~20 us (so that the launch spreads over every CU the stream may use) and
// records the XCC it ran on and its HW_ID register (CU / SH / SE).  Used by tools/cumask_probe.py to read the bit -> CU map of
// hipExtStreamCreateWithCUMask on MI355X.
// build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/micro/cu_census.hip -o tools/micro/libcu_census.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void census_kernel(uint32_t* out, long long spin_ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);    // HW_REG_XCC_ID[3:0]
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hw;
  }
}

extern "C" int cu_census(uint32_t* out, int blocks, int spin_us, void* stream) {
  hipLaunchKernelGGL(census_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, out, (long long)spin_us * 100);
  return (int)hipGetLastError();
}
