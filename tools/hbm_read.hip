// HBM read-rate probe shaped like the skinny GEMM's weight stream: every wave pulls a contiguous run of 1 KiB fragments.
// build: hipcc --offload-arch=gfx950 -O3 tools/hbm_read.hip -o tools/hbm_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int UN>
__global__ __launch_bounds__(256) void rd(const u32x4* __restrict__ src, unsigned* __restrict__ out, int frags_per_wave, int rot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long w = (long)blockIdx.x * 4 + wave;
  const u32x4* p = src + w * frags_per_wave * 64 + lane;
  extern __shared__ unsigned sm[];
  unsigned acc = 0;
  if (rot == 77) sm[threadIdx.x] = 1;
  const int r = rot ? (int)(w & (UN - 1)) : 0;
  for (int kt = 0; kt < frags_per_wave; kt += UN) {
    u32x4 v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) v[u] = __builtin_nontemporal_load(p + (long)(kt + ((u + r) & (UN - 1))) * 64);
#pragma unroll
    for (int u = 0; u < UN; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
int main(int argc, char** argv) {
  const long bytes = (argc > 1 ? atol(argv[1]) : 32) << 20;
  const int fpw = argc > 2 ? atoi(argv[2]) : 16;
  const int lds = argc > 3 ? atoi(argv[3]) : 0;
  const int rep_each = argc > 4 ? atoi(argv[4]) : 1;   // read every buffer this many times back to back (MALL/L2 reuse probe)
  hipFuncSetAttribute(reinterpret_cast<const void*>(&rd<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const long pool = 1L << 30;
  char* buf; unsigned* out;
  hipMalloc(&buf, pool + bytes); hipMalloc(&out, 4);
  hipMemset(buf, 1, pool + bytes);
  const int ncopy = (int)(pool / bytes);
  const long waves = bytes / (1024L * fpw);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  {  // same launches as graph nodes (what the engine replays)
    hipStream_t st; hipStreamCreate(&st);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < ncopy; ++i)
      hipLaunchKernelGGL(rd<8>, dim3(waves / 4), dim3(256), lds, st, reinterpret_cast<const u32x4*>(buf + (long)(i / rep_each) * bytes), out, fpw, 0);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("GRAPH rep %d lds %d bytes %ld MB frags/wave %d unroll 8: %.2f us/launch  %.2f TB/s\n", rep_each, lds, bytes >> 20, fpw, 1e3 * ms / ncopy, bytes / (1e3 * ms / ncopy) / 1e6);
  }
  for (int un : {8, 16}) for (int rot : {0, 1}) {
    if (fpw % un) continue;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      for (int i = 0; i < ncopy; ++i) {
        const u32x4* s = reinterpret_cast<const u32x4*>(buf + (long)i * bytes);
        if (un == 8) hipLaunchKernelGGL(rd<8>, dim3(waves / 4), dim3(256), 0, 0, s, out, fpw, rot);
        else hipLaunchKernelGGL(rd<16>, dim3(waves / 4), dim3(256), 0, 0, s, out, fpw, rot);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("bytes %ld MB frags/wave %d unroll %d rot %d: %.2f us/launch  %.2f TB/s\n", bytes >> 20, fpw, un, rot, 1e3 * ms / ncopy, bytes / (1e3 * ms / ncopy) / 1e6);
    }
  }
  return 0;
}
