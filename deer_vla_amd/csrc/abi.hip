// Library identification for libdeer_hip.so (include/deer_hip.h).
#include "common.h"
extern "C" const char* deer_hip_arch(void) { return "gfx950"; }
extern "C" int deer_hip_abi_version(void) { return 1; }

// Keeps the stream busy for ~`us` microseconds (wall_clock64 ticks at 100 MHz).  Used by bench.py's profiling pass
// to let the host run ahead of the GPU so that event brackets do not contain host launch gaps.
__global__ void spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
extern "C" int deer_spin_us(int us, void* stream) {
  if (us <= 0 || us > 1000000) return DEER_ERR_SHAPE;
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), (long long)us * 100);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}
