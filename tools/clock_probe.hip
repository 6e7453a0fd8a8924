// What shader clock does the chip sustain while its matrix pipes are busy?  (a 257x256x1024 frame
// tile takes 21 us with 64 workgroups and 30 us with 256.)  The hwmon / rocm-smi sclk shows the DPM level (2406 MHz throughout,
// tools/dvfs_probe.py); this probe reads the clock the WAVES see: every workgroup runs a register-only MFMA loop (no memory traffic)
// and brackets it with s_memtime (shader-clock ticks) and s_memrealtime (constant 100 MHz):
//     effective clock = d(memtime) / d(memrealtime) * 100 MHz.
// Legs: v_mfma_f32_16x16x32_bf16 and v_mfma_f32_32x32x16_bf16 (equal FLOPs per iteration), 64 ... 512 workgroups of 4 and 16 waves, operands random bf16 or zero, each leg sustained for ~0.5 s
// (back-to-back launches) so that the power controller is in steady state; the last launches are reported.
// build: hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o tools/clock_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <chrono>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void burn(const uint4* __restrict__ seed, int iters, int mfma_on, unsigned long long* __restrict__ out, float* sink) {
  const int tid = threadIdx.x;
  uint4 sa = seed[(blockIdx.x * blockDim.x + tid) & 4095], sb = seed[(blockIdx.x * blockDim.x + tid + 977) & 4095];
  bf16x8 a, b;
  __builtin_memcpy(&a, &sa, 16);
  __builtin_memcpy(&b, &sb, 16);
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const unsigned long long c0 = __builtin_readcyclecounter();        // s_memtime
  const unsigned long long r0 = wall_clock64();                       // s_memrealtime, 100 MHz
  if (mfma_on == 1) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
  } else if (mfma_on == 2) {                                           // the same FLOPs per iteration as 4 x 32x32x16 (a quarter of the operand reads per MAC)
    f32x16 big[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) big[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) big[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, big[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i][0] = big[i][0] + big[i][5] + big[i][10] + big[i][15];
  } else {
    float x = acc[0][0] + tid;
    for (int it = 0; it < iters * 16; ++it) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
    acc[0][0] = x;
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  const unsigned long long r1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) *sink = s;                                      // keeps the accumulators live
  if (tid == 0) {
    out[2 * blockIdx.x] = c1 - c0;
    out[2 * blockIdx.x + 1] = r1 - r0;
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 0.5;
  uint4* seed[2];
  std::vector<uint4> h(4096);
  srand(1);
  for (auto& v : h) {                                                  // random bf16 in about [-2, 2): random sign, exponent 120..127, random mantissa
    uint32_t w[4];
    for (int i = 0; i < 4; ++i) {
      auto one = []() { return (uint32_t)(((rand() & 1) << 15) | ((120 + (rand() & 7)) << 7) | (rand() & 127)); };
      w[i] = one() | (one() << 16);
    }
    v = uint4{w[0], w[1], w[2], w[3]};
  }
  CK(hipMalloc(&seed[0], 4096 * 16));
  CK(hipMalloc(&seed[1], 4096 * 16));
  CK(hipMemcpy(seed[0], h.data(), 4096 * 16, hipMemcpyHostToDevice));
  CK(hipMemset(seed[1], 0, 4096 * 16));
  unsigned long long* out;
  float* sink;
  CK(hipMalloc(&out, 2 * 1024 * 8));
  CK(hipMalloc(&sink, 4));
  std::vector<unsigned long long> ho(2 * 1024);
  printf("%-6s %-5s %5s %6s | %10s %10s %9s | %s\n", "pipe", "data", "wgs", "waves", "us/launch", "clk MHz", "min MHz", "MFMA rate of the launch, TFLOP/s (bf16 dense)");
  for (int mfma_on = 2; mfma_on >= 0; --mfma_on)
    for (int zero = 0; zero < 2; ++zero)
      for (int waves : {4, 16})
        for (int wgs : {64, 128, 192, 256, 512}) {
          if (!mfma_on && (zero || waves == 16)) continue;
          if (mfma_on == 2 && (wgs == 128 || wgs == 192)) continue;
          const int iters = 20000 / (waves / 4) / (wgs > 256 ? 2 : 1);   // ~100-200 us per launch
          auto t0 = std::chrono::steady_clock::now();
          int n = 0;
          double us = 0;
          while (true) {
            hipLaunchKernelGGL(burn, dim3(wgs), dim3(64 * waves), 0, 0, seed[zero], iters, mfma_on, out, sink);
            ++n;
            if ((n & 15) == 0) {
              CK(hipDeviceSynchronize());
              if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > secs) break;
            }
          }
          hipEvent_t e0, e1;
          CK(hipEventCreate(&e0));
          CK(hipEventCreate(&e1));
          CK(hipEventRecord(e0, 0));
          for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(burn, dim3(wgs), dim3(64 * waves), 0, 0, seed[zero], iters, mfma_on, out, sink);
          CK(hipEventRecord(e1, 0));
          CK(hipDeviceSynchronize());
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          us = ms * 1e3 / 8;
          CK(hipMemcpy(ho.data(), out, 2 * wgs * 8, hipMemcpyDeviceToHost));
          double sum = 0, mn = 1e30;
          for (int i = 0; i < wgs; ++i) {
            const double mhz = (double)ho[2 * i] / (double)ho[2 * i + 1] * 100.0;
            sum += mhz;
            if (mhz < mn) mn = mhz;
          }
          const double flop = mfma_on ? 2.0 * 16 * 16 * 32 * 8.0 * iters * waves * wgs : 0;
          printf("%-6s %-5s %5d %6d | %10.1f %10.0f %9.0f | %.0f\n", mfma_on == 2 ? "mfma32" : mfma_on ? "mfma16" : "valu", zero ? "zero" : "rand", wgs, waves, us, sum / wgs, mn, flop / us / 1e6);
          fflush(stdout);
        }
  return 0;
}
