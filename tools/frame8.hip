// PROTOTYPE (measurement only, not in libdeer_hip.so): the frame-tile GEMM of csrc/gemm_bigm.hip on 8 waves per workgroup.
// the 16-wave kernel runs the matrix pipe 33 % of the time and its waves are parked 52 % of theirs; 16 waves leave 128
// VGPRs per wave, which excludes bigger wave tiles.  Here: 4 x 2 waves, wave tile 64 rows x (TN x 16) columns (BN = 32 TN: 256 or 192),
// 34 (26) MFMAs per K-step and wave behind 13 (11) fragment reads instead of 17 behind 9, the fragment reads of a K-step interleaved with
// its MFMAs (one W fragment ahead: sched_group_barrier), MUBUF LDS-DMA so that the compiler counts lgkmcnt, bf16 result staged through LDS.
// One camera frame (257 rows) per row tile, the 17th MFMA row tile dealt out two 16x16 tiles per wave.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/frame8.hip -o tools/frame8 -ldl
// run (GPU box, from the repo root): tools/frame8 [frames=16] [N=4096] [K=1024]   -> us per launch next to the library's auto tile, max |diff|
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lptr_t;

__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {       // round to nearest even, like csrc/common.h
  uint32_t a = __float_as_uint(lo), b = __float_as_uint(hi);
  a += 0x7fffu + ((a >> 16) & 1u);
  b += 0x7fffu + ((b >> 16) & 1u);
  return (a >> 16) | (b & 0xffff0000u);
}
__device__ __forceinline__ void dma16(const void* base, unsigned voff, unsigned soff, void* lds) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000), (lptr_t*)lds, 16, voff,
                                           soff, 0, 0);
}
template <int N_>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

template <int TN, int D>
__global__ __launch_bounds__(512) void frame8_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W, int ldw,
                                                      bf16_t* __restrict__ C, int ldc, int M, int N, int K) {
  constexpr int NW = 8, BN = 32 * TN, CH = 17 + BN / 16, STAGE = CH * 1024, XT = BN / 16 / 8 + ((BN / 16) % 8 ? 1 : 0);
  constexpr int CPW = (CH + NW - 1) / NW, N_HI = CH - (CPW - 1) * NW;
  static_assert(D * STAGE <= 160 * 1024, "LDS");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int c = lane & 15, g = lane >> 4;
  const int tiles_n = N / BN, rows_n = gridDim.x / tiles_n;
  int rt, ct;
  {                                                          // XCD blocks (gemm_bigm.hip)
    int gr = 0, gc = 0;
    long best = 1L << 60;
    for (int e = 0; e < 4; ++e) {
      const int r_ = 1 << e, c_ = 8 >> e;
      if (rows_n % r_ == 0 && tiles_n % c_ == 0) {
        const long cost = (long)(rows_n / r_) * 257 + (long)(tiles_n / c_) * BN;
        if (cost < best) { best = cost; gr = r_; gc = c_; }
      }
    }
    const int bid = blockIdx.x;
    if (gr != 0) {
      const int xcd = bid & 7, idx = bid >> 3, bc = tiles_n / gc, br = rows_n / gr;
      rt = (xcd / gc) * br + idx / bc;
      ct = (xcd % gc) * bc + idx % bc;
    } else { rt = bid / tiles_n; ct = bid % tiles_n; }
  }
  const int m0 = rt * 257, n0 = ct * BN;
  const int rows_valid = min(257, M - m0);

  const int lr = lane >> 2;
  const int ls = ((lane & 3) ^ ((0x1320 >> (((lr >> 2) & 3) * 4)) & 3)) * 8;
  const bf16_t* base[CPW];
  unsigned vo[CPW];
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    const int q = min(wave + i * NW, CH - 1);
    const bool is_a = q < 17;
    base[i] = is_a ? A : W;
    vo[i] = is_a ? (unsigned)(((long)min(m0 + q * 16 + lr, M - 1) * lda + ls) * 2) : (unsigned)(((long)min(n0 + (q - 17) * 16 + lr, N - 1) * ldw + ls) * 2);
  }
  const int nk = K >> 5;
  f32x4 acc[TN][4], accx[XT];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int x = 0; x < XT; ++x) accx[x] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr_sw = (g ^ ((0x1320 >> (((c >> 2) & 3) * 4)) & 3)) << 4;
  const int a_off = (wm * 64 + c) * 64 + fr_sw;
  const int x_off = (256 + c) * 64 + fr_sw;
  const int w_off = 17 * 1024 + (wn * TN * 16 + c) * 64 + fr_sw;
  // the dealt-out row tile: column tiles xi = wm * XT + x of this wave's column group (x < XT), if xi < TN
  auto run = [&](auto cpw_tag) {
    constexpr int CPWL = decltype(cpw_tag)::value;
    auto issue = [&](int t) {
      const int k0 = min(t, nk - 1) << 5;
      unsigned char* st = smem + (t % D) * STAGE;
#pragma unroll
      for (int i = 0; i < CPWL; ++i) dma16(base[i], vo[i], k0 * 2, st + (wave + i * NW) * 1024);
    };
#pragma unroll
    for (int t = 0; t < D - 1; ++t) issue(t);
    for (int kt = 0; kt < nk; ++kt) {
      wait_vmcnt<(D - 2) * CPWL>();
      __builtin_amdgcn_s_barrier();
      issue(kt + D - 1);
      const unsigned char* st = smem + (kt % D) * STAGE;
      bf16x8 af[4], wf[TN];
#pragma unroll
      for (int j = 0; j < 4; ++j) af[j] = *reinterpret_cast<const bf16x8*>(st + a_off + j * 1024);
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(st + w_off + i * 1024);
      const bf16x8 afx = *reinterpret_cast<const bf16x8*>(st + x_off);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int x = 0; x < XT; ++x) {
        bf16x8 wx = wf[0];
#pragma unroll
        for (int e = 1; e < TN; ++e) wx = (wm * XT + x == e) ? wf[e] : wx;
        accx[x] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wx, afx, accx[x], 0, 0, 0);
      }
      // A fragments + the first TWO W fragments, then per W fragment: its 4 MFMAs, the read of the fragment after next (reads run one
      // group of MFMAs ahead); the dealt-out tiles last
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // W fragment i+2 ... and finally the extra A fragment (no-op once exhausted)
      }
      __builtin_amdgcn_sched_group_barrier(0x008, XT, 0);
    }
    wait_vmcnt<0>();
  };
  if (wave < N_HI) run(std::integral_constant<int, CPW>{});
  else run(std::integral_constant<int, CPW - 1>{});

  constexpr int CPITCH = BN * 2 + 16;
  static_assert(272 * CPITCH <= 160 * 1024, "C staging");
  __syncthreads();
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int n = (wn * TN + i) * 16 + g * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 a = acc[i][j];
      *reinterpret_cast<uint2*>(smem + (wm * 64 + j * 16 + c) * CPITCH + n * 2) = uint2{pack2bf(a[0], a[1]), pack2bf(a[2], a[3])};
    }
  }
#pragma unroll
  for (int x = 0; x < XT; ++x) {
    const int xi = wm * XT + x;
    if (xi < TN) {
      const f32x4 a = accx[x];
      *reinterpret_cast<uint2*>(smem + (256 + c) * CPITCH + ((wn * TN + xi) * 16 + g * 4) * 2) = uint2{pack2bf(a[0], a[1]), pack2bf(a[2], a[3])};
    }
  }
  __syncthreads();
  constexpr int PPR = BN / 8;
  const int pieces = rows_valid * PPR;
  bf16_t* Cb = C + (long)m0 * ldc + n0;
  for (int p = tid; p < pieces; p += 512) {
    const int r = p / PPR, cp = p - r * PPR;
    *reinterpret_cast<uint4*>(Cb + (long)r * ldc + cp * 8) = *reinterpret_cast<const uint4*>(smem + r * CPITCH + cp * 16);
  }
}

typedef int (*gemm_fn)(const void*, int, long, const void*, int, const float*, void*, int, long, int, int, int, int, int, const float*, int, const int*, void*);

template <int TN, int D>
static float time_frame8(const bf16_t* A, const std::vector<bf16_t*>& Ws, bf16_t* C, int M, int N, int K, int reps) {
  constexpr int BN = 32 * TN;
  constexpr int ring = D * (17 + BN / 16) * 1024, cst = 272 * (BN * 2 + 16), smem = ring > cst ? ring : cst;
  hipFuncSetAttribute((const void*)frame8_kernel<TN, D>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  const int tiles = (M / 257) * (N / BN);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (size_t i = 0; i < Ws.size(); ++i) frame8_kernel<TN, D><<<tiles, 512, smem>>>(A, K, Ws[i], K, C, N, M, N, K);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r)
    for (size_t i = 0; i < Ws.size(); ++i) frame8_kernel<TN, D><<<tiles, 512, smem>>>(A, K, Ws[i], K, C, N, M, N, K);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return 1e3f * ms / (reps * Ws.size());
}

int main(int argc, char** argv) {
  const int frames = argc > 1 ? atoi(argv[1]) : 16, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 1024;
  const int M = frames * 257, NCOPY = 24;
  void* lib = dlopen("deer_vla_amd/lib/libdeer_hip.so", RTLD_NOW);
  gemm_fn gemm = lib ? (gemm_fn)dlsym(lib, "deer_gemm_bf16_nt") : nullptr;
  std::vector<bf16_t> hA((size_t)M * K), hW((size_t)N * K);
  uint32_t s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((int)(s >> 9) - (1 << 22)) / (float)(1 << 22); };
  auto tobf = [](float f) { union { float f; uint32_t u; } v; v.f = f; v.u += 0x7fffu + ((v.u >> 16) & 1u); return (bf16_t)(v.u >> 16); };
  for (auto& v : hA) v = tobf(rnd());
  bf16_t *A, *C, *C2;
  hipMalloc(&A, hA.size() * 2); hipMalloc(&C, (size_t)M * N * 2); hipMalloc(&C2, (size_t)M * N * 2);
  hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
  std::vector<bf16_t*> Ws(NCOPY);
  for (int i = 0; i < NCOPY; ++i) {
    for (auto& v : hW) v = tobf(rnd() * 0.03f);
    hipMalloc(&Ws[i], hW.size() * 2);
    hipMemcpy(Ws[i], hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
  }
  float lib_us = 0.f;
  if (gemm) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < NCOPY; ++i) gemm(A, K, 0, Ws[i], K, nullptr, C2, N, 0, M, N, K, 1, 0, nullptr, 0, nullptr, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 4; ++r)
      for (int i = 0; i < NCOPY; ++i) gemm(A, K, 0, Ws[i], K, nullptr, C2, N, 0, M, N, K, 1, 0, nullptr, 0, nullptr, nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    lib_us = 1e3f * ms / (4 * NCOPY);
  }
  float us = 0.f;
  const char* what = "";
  if (N % 256 == 0 && argc <= 4) { us = time_frame8<8, 4>(A, Ws, C, M, N, K, 4); what = "257x256, 8 waves, 4 stages"; }
  else if (N % 192 == 0) { us = time_frame8<6, 4>(A, Ws, C, M, N, K, 4); what = "257x192, 8 waves, 4 stages"; }
  else { printf("N must be a multiple of 256 or 192\n"); return 1; }
  // both paths last multiplied with weight copy NCOPY-1: same K order per output element -> identical bits expected
  std::vector<bf16_t> h1((size_t)M * N), h2((size_t)M * N);
  hipMemcpy(h1.data(), C, h1.size() * 2, hipMemcpyDeviceToHost);
  hipMemcpy(h2.data(), C2, h2.size() * 2, hipMemcpyDeviceToHost);
  size_t diff = 0;
  for (size_t i = 0; i < h1.size(); ++i) diff += h1[i] != h2[i];
  printf("M=%d N=%d K=%d | frame8 (%s) %.1f us  %.0f TFLOP/s | library auto tile %.1f us | elements that differ from the library: %zu of %zu\n",
         M, N, K, what, us, 2.0 * M * N * K / us / 1e6, lib_us, gemm ? diff : (size_t)0, h1.size());
  return 0;
}
