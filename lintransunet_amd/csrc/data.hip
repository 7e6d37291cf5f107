// Data side of the training scripts: dataset/CT_pancreas_ids.py:143-173 -- HU clip [-91, 250],
// (x - 86.9) / 39.4, (D,H,W) -> (H,W,D) -- and the crop + flip that follow (monai RandCropByPosNegLabeld / RandFlipd applied at
// host-chosen centres), as streaming kernels: a volume is uploaded once and every patch is cut on the device.
#include "common.h"

#define GRID_STRIDE(i, n) \
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)
static unsigned dgrid(long long n) {
  long long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

// img f32 [H][W][D] = (clamp(raw [D][H][W], lo, hi) - mean) / std ;  lab u8 [H][W][D] = rawlab [D][H][W] (nullable pair)
__global__ void ct_preprocess_kernel(const float* __restrict__ raw, float* __restrict__ img, const uint8_t* __restrict__ rawlab,
                                     uint8_t* __restrict__ lab, int D, int H, int W, float lo, float hi, float mean, float std) {
  const long long n = (long long)D * H * W;
  GRID_STRIDE(i, n) {                      // i enumerates the OUTPUT [H][W][D]
    const int d = (int)(i % D);
    const long long hw = i / D;
    const long long src = (long long)d * H * W + hw;
    if (raw != nullptr) {
      float v = raw[src];
      v = v < lo ? lo : v;
      v = v > hi ? hi : v;
      img[i] = (v - mean) / std;
    }
    if (rawlab != nullptr) lab[i] = rawlab[src];
  }
}

// patches [n][h][w][d] cut from vol [H][W][D] at desc [n][5] = (h0, w0, d0, flip_h, flip_w); flips mirror the patch axes
template <typename T>
__global__ void crop_flip_kernel(const T* __restrict__ vol, T* __restrict__ out, const int* __restrict__ desc, int n, int H, int W,
                                 int D, int h, int w, int d) {
  const long long per = (long long)h * w * d, total = per * n;
  GRID_STRIDE(i, total) {
    const int k = (int)(i / per);
    long long r = i - (long long)k * per;
    const int z = (int)(r % d); r /= d;
    int y = (int)(r % w);
    int x = (int)(r / w);
    if (desc[5 * k + 3]) x = h - 1 - x;
    if (desc[5 * k + 4]) y = w - 1 - y;
    out[i] = vol[((long long)(desc[5 * k] + x) * W + (desc[5 * k + 1] + y)) * D + (desc[5 * k + 2] + z)];
  }
}

extern "C" int ltu_ct_preprocess(const float* raw, float* img, const uint8_t* rawlab, uint8_t* lab, int D, int H, int W, float lo,
                                 float hi, float mean, float std, ltu_stream_t s) {
  const long long n = (long long)D * H * W;
  if (n <= 0) return LTU_OK;
  if (std == 0.f || (raw == nullptr) != (img == nullptr) || (rawlab == nullptr) != (lab == nullptr)) return LTU_E_ARG;
  hipLaunchKernelGGL(ct_preprocess_kernel, dim3(dgrid(n)), dim3(256), 0, (hipStream_t)s, raw, img, rawlab, lab, D, H, W, lo, hi, mean, std);
  return ltu_check_launch();
}
extern "C" int ltu_crop_flip(const void* vol, void* out, const int* desc, int n, int H, int W, int D, int h, int w, int d, int elem_bytes,
                             ltu_stream_t s) {
  if (n <= 0) return LTU_OK;
  if (h > H || w > W || d > D) return LTU_E_SHAPE;
  const long long total = (long long)n * h * w * d;
  if (elem_bytes == 4)
    hipLaunchKernelGGL((crop_flip_kernel<float>), dim3(dgrid(total)), dim3(256), 0, (hipStream_t)s, (const float*)vol, (float*)out, desc, n, H, W, D, h, w, d);
  else if (elem_bytes == 1)
    hipLaunchKernelGGL((crop_flip_kernel<uint8_t>), dim3(dgrid(total)), dim3(256), 0, (hipStream_t)s, (const uint8_t*)vol, (uint8_t*)out, desc, n, H, W, D, h, w, d);
  else
    return LTU_E_DTYPE;
  return ltu_check_launch();
}
