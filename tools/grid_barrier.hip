// Cost of a device-wide barrier inside one persistent kernel vs a kernel boundary (the alternative the engine uses).
// build: hipcc --offload-arch=gfx950 -O3 tools/grid_barrier.hip -o tools/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// flat barrier: one agent-scope atomic per workgroup on a single counter, spin on it
__global__ __launch_bounds__(256) void flat(unsigned* cnt, int iters, float* sink, const float* data, int touch) {
  const unsigned nb = gridDim.x;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (touch) {   // a little real traffic between barriers: write then (after the barrier) read a neighbour's line
      sink[(size_t)blockIdx.x * 256 + threadIdx.x] = acc + it;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      __atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE);     // agent scope by default for global atomics in HIP
      const unsigned target = (unsigned)(it + 1) * nb;
      while (__atomic_load_n(cnt, __ATOMIC_ACQUIRE) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    if (touch) acc += sink[(size_t)((blockIdx.x + 37) % nb) * 256 + threadIdx.x];
  }
  if (acc == 1.2345f) sink[0] = acc;
}

// two-level: per-XCD counter (workgroups round-robin over 8 XCDs: xcd = blockIdx.x & 7), then one global counter
__global__ __launch_bounds__(256) void twolevel(unsigned* cnt, int iters, float* sink) {
  const unsigned nb = gridDim.x, xcd = blockIdx.x & 7, per = nb >> 3;
  unsigned* local = cnt + 64 * (1 + xcd);
  unsigned* flag = cnt + 64 * 10;
  for (int it = 0; it < iters; ++it) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned t = __atomic_fetch_add(local, 1u, __ATOMIC_ACQ_REL);
      if (t == (unsigned)(it + 1) * per - 1) {                      // last of this XCD
        const unsigned g = __atomic_fetch_add(cnt, 1u, __ATOMIC_ACQ_REL);
        if (g == (unsigned)(it + 1) * 8 - 1) __atomic_store_n(flag, (unsigned)(it + 1), __ATOMIC_RELEASE);
      }
      while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) < (unsigned)(it + 1)) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
  }
}

__global__ void empty(float* p) { if (p == nullptr) p[0] = 1; }

int main(int argc, char** argv) {
  unsigned* cnt; float* sink;
  hipMalloc(&cnt, 64 * 16 * 4); hipMalloc(&sink, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  float ms;
  for (int nb : {64, 128, 256, 512}) {
    for (int touch : {0, 1}) {
      hipMemset(cnt, 0, 64 * 16 * 4);
      hipLaunchKernelGGL(flat, dim3(nb), dim3(256), 0, 0, cnt, 10, sink, sink, touch);
      hipDeviceSynchronize();
      hipMemset(cnt, 0, 64 * 16 * 4);
      hipEventRecord(e0);
      hipLaunchKernelGGL(flat, dim3(nb), dim3(256), 0, 0, cnt, iters, sink, sink, touch);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      printf("flat     blocks %4d touch %d: %.2f us / barrier\n", nb, touch, 1e3 * ms / iters);
    }
    hipMemset(cnt, 0, 64 * 16 * 4);
    hipEventRecord(e0);
    hipLaunchKernelGGL(twolevel, dim3(nb), dim3(256), 0, 0, cnt, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("twolevel blocks %4d        : %.2f us / barrier\n", nb, 1e3 * ms / iters);
  }
  // kernel boundary under graph replay for comparison
  hipStream_t st; hipStreamCreate(&st);
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(empty, dim3(256), dim3(256), 0, st, sink);
  hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, st); hipStreamSynchronize(st);
  hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  printf("graph replay of 1000 empty 256-block kernels: %.2f us / kernel\n", ms);
  return 0;
}
