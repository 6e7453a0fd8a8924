// Camera-frame preprocessing on the GPU: uint8 HWC frame -> CLIP-normalised bf16 / f32 CHW tensor, i.e. the reference's
// ``preprocess_image`` (robot_flamingo/data/data.py:898-902) with open_clip's eval transform (factory.py:109-112):
//   Resize(S, bicubic, shorter side) -> CenterCrop(S) -> ToTensor (/255) -> Normalize(CLIP mean, std).
// The reference does this with PIL on the host (200x200 static camera and 84x84 gripper camera, both UPSCALED to 224);
// PIL's resampler is restated here exactly, because one uint8 step is 0.0145 after normalisation - more than the 1e-2 the
// parity gate allows:
//   * separable, horizontal pass first then vertical, each pass ROUNDED to uint8 (Pillow ImagingResample);
//   * bicubic kernel with a = -0.5, support 2 * max(scale, 1), taps [int(c - sup + .5), int(c + sup + .5)) around
//     c = (x + .5) * scale, weights normalised to sum 1 in double precision;
//   * 8-bit fixed point: k_int = (int)(k * 2^22 +- .5), pixel = clip8((2^21 + sum k_int * p) >> 22).
// Two launches (one per pass) for a batch of N equally sized frames; HBM traffic is the frame itself (tens of KB): latency-bound.
#include "common.h"
#include <cmath>

#define PP_BITS 22
#define PP_MAX_TAPS 32   // support 2*scale: covers down-scaling by up to ~7x

__device__ __forceinline__ double pp_bicubic(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0;
  if (x < 2.0) return (((x - 5.0) * x + 8.0) * x - 4.0) * a;
  return 0.0;
}

// taps of output coordinate `o`: first input index and integer coefficients; returns the tap count
__device__ __forceinline__ int pp_taps(int o, int in_size, int out_size, int* xmin_out, int* kk) {
  const double scale = (double)in_size / (double)out_size;
  const double fs = scale < 1.0 ? 1.0 : scale;
  const double support = 2.0 * fs, ss = 1.0 / fs;
  const double center = (o + 0.5) * scale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  int n = xmax - xmin;
  if (n > PP_MAX_TAPS) n = PP_MAX_TAPS;
  double k[PP_MAX_TAPS];
  double ww = 0.0;
  for (int x = 0; x < n; ++x) {
    k[x] = pp_bicubic((x + xmin - center + 0.5) * ss);
    ww += k[x];
  }
  for (int x = 0; x < n; ++x) {
    const double v = ww != 0.0 ? k[x] / ww : k[x];
    kk[x] = (int)(v < 0.0 ? v * (double)(1 << PP_BITS) - 0.5 : v * (double)(1 << PP_BITS) + 0.5);
  }
  *xmin_out = xmin;
  return n;
}

__device__ __forceinline__ unsigned char pp_clip8(int v) {
  v >>= PP_BITS;
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: src [N][H][W][3] u8 -> tmp [N][H][OW][3] u8
__global__ void pp_horizontal_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ tmp, int N, int H, int W, int OW) {
  const long total = (long)N * H * OW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW);
    const long row = i / OW;                               // n*H + y
    int kk[PP_MAX_TAPS], xmin;
    const int n = pp_taps(ox, W, OW, &xmin, kk);
    const unsigned char* p = src + (row * W + xmin) * 3;
    int s0 = 1 << (PP_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < n; ++x) {
      s0 += p[3 * x] * kk[x]; s1 += p[3 * x + 1] * kk[x]; s2 += p[3 * x + 2] * kk[x];
    }
    unsigned char* q = tmp + (row * OW + ox) * 3;
    q[0] = pp_clip8(s0); q[1] = pp_clip8(s1); q[2] = pp_clip8(s2);
  }
}

// vertical pass + center crop + ToTensor + Normalize: tmp [N][H][OW][3] u8 -> out [N][3][S][S] (bf16 and/or f32)
template <bool F16>
__global__ void pp_vertical_kernel(const unsigned char* __restrict__ tmp, bf16_t* __restrict__ out_bf, float* __restrict__ out_f32, int N, int H, int OW,
                                   int OH, int S, int crop_x, int crop_y, float m0, float m1, float m2, float r0, float r1, float r2) {
  const long total = (long)N * S * S;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % S), y = (int)((i / S) % S), n = (int)(i / ((long)S * S));
    int kk[PP_MAX_TAPS], ymin;
    const int nt = pp_taps(y + crop_y, H, OH, &ymin, kk);
    const unsigned char* p = tmp + (((long)n * H + ymin) * OW + (x + crop_x)) * 3;
    int s0 = 1 << (PP_BITS - 1), s1 = s0, s2 = s0;
    for (int t = 0; t < nt; ++t) {
      const unsigned char* q = p + (long)t * OW * 3;
      s0 += q[0] * kk[t]; s1 += q[1] * kk[t]; s2 += q[2] * kk[t];
    }
    const float v0 = ((float)pp_clip8(s0) / 255.0f - m0) / r0, v1 = ((float)pp_clip8(s1) / 255.0f - m1) / r1,
                v2 = ((float)pp_clip8(s2) / 255.0f - m2) / r2;
    const long o = ((long)n * 3 * S + y) * S + x, plane = (long)S * S;
    if (out_bf != nullptr) { out_bf[o] = f2x<F16>(v0); out_bf[o + plane] = f2x<F16>(v1); out_bf[o + 2 * plane] = f2x<F16>(v2); }
    if (out_f32 != nullptr) { out_f32[o] = v0; out_f32[o + plane] = v1; out_f32[o + 2 * plane] = v2; }
  }
}

// src: uint8 [N][H][W][3] (device); tmp: uint8 scratch of N*H*OW*3 bytes, OW = max(S, round(W * S / min(W, H))); out_bf16 / out_f32:
// [N][3][S][S] (either may be NULL).  mean / std: host float[3].
template <bool F16>
static int preprocess_frames(const unsigned char* src, int N, int H, int W, int S, const float* mean, const float* std_, unsigned char* tmp,
                                      void* out_bf16, float* out_f32, void* stream) {
  if (src == nullptr || tmp == nullptr || N <= 0 || H <= 0 || W <= 0 || S <= 0 || mean == nullptr || std_ == nullptr ||
      (out_bf16 == nullptr && out_f32 == nullptr))
    return DEER_ERR_SHAPE;
  // torchvision Resize(S): the SHORTER side becomes S, the other keeps the aspect ratio (int(S * long / short)); open_clip then
  // center-crops S x S
  int OW, OH;
  if (W <= H) { OW = S; OH = (int)((long)S * H / W); } else { OH = S; OW = (int)((long)S * W / H); }
  const double sx = (double)W / OW, sy = (double)H / OH;
  if (2.0 * (sx < 1.0 ? 1.0 : sx) * 2.0 + 1.0 > PP_MAX_TAPS || 2.0 * (sy < 1.0 ? 1.0 : sy) * 2.0 + 1.0 > PP_MAX_TAPS) return DEER_ERR_SHAPE;
  const int crop_x = (int)nearbyint((OW - S) / 2.0), crop_y = (int)nearbyint((OH - S) / 2.0);    // torchvision center_crop: int(round(.)), half to even
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long t1 = (long)N * H * OW, t2 = (long)N * S * S;
  hipLaunchKernelGGL(pp_horizontal_kernel, dim3((unsigned)((t1 + 255) / 256 > 4096 ? 4096 : (t1 + 255) / 256)), dim3(256), 0, st, src, tmp, N, H, W, OW);
  hipLaunchKernelGGL(pp_vertical_kernel<F16>, dim3((unsigned)((t2 + 255) / 256 > 4096 ? 4096 : (t2 + 255) / 256)), dim3(256), 0, st, tmp,
                     reinterpret_cast<bf16_t*>(out_bf16), out_f32, N, H, OW, OH, S, crop_x, crop_y, mean[0], mean[1], mean[2], std_[0], std_[1], std_[2]);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

extern "C" int deer_preprocess_frames(const unsigned char* src, int N, int H, int W, int S, const float* mean, const float* std_, unsigned char* tmp,
                                      void* out_bf16, float* out_f32, void* stream) {
  return preprocess_frames<false>(src, N, H, W, S, mean, std_, tmp, out_bf16, out_f32, stream);
}
// the same with the 16-bit output in IEEE fp16 (round 6: the frame format of a precision = "fp16" engine; every 8-bit pixel level stays distinct)
extern "C" int deer_preprocess_frames_f16(const unsigned char* src, int N, int H, int W, int S, const float* mean, const float* std_, unsigned char* tmp,
                                          void* out_f16, float* out_f32, void* stream) {
  return preprocess_frames<true>(src, N, H, W, S, mean, std_, tmp, out_f16, out_f32, stream);
}

// scratch bytes deer_preprocess_frames needs for N frames of H x W (host only)
extern "C" long deer_preprocess_scratch_bytes(int N, int H, int W, int S) {
  const int OW = W <= H ? S : (int)((long)S * W / H);
  return (long)N * H * OW * 3;
}
