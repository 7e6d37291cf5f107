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

// ---- augmentations of IdPosPanCTDataset (dataset/CT_pancreas_ids.py:112-134) on batches of patches [n][H][W][D] f32 ----------
// RandRotated -> monai Rotate: out[p] = trilinear(in, M p) in voxel coordinates, border padding (grid_sample semantics: the
// coordinate is clamped to [0, size-1], corners outside contribute nothing).  mats [n][12]: rows of the 3x4 matrix.
__device__ __forceinline__ float tri_sample(const float* __restrict__ v, int H, int W, int D, float x, float y, float z) {
  x = fminf(fmaxf(x, 0.f), (float)(H - 1));
  y = fminf(fmaxf(y, 0.f), (float)(W - 1));
  z = fminf(fmaxf(z, 0.f), (float)(D - 1));
  const float fx = floorf(x), fy = floorf(y), fz = floorf(z);
  const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
  const float tx = x - fx, ty = y - fy, tz = z - fz;
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int xi = x0 + a, yi = y0 + b, zi = z0 + c;
        const float w = (a ? tx : 1.f - tx) * (b ? ty : 1.f - ty) * (c ? tz : 1.f - tz);
        if (xi < H && yi < W && zi < D) acc += w * v[((long long)xi * W + yi) * D + zi];
      }
  return acc;
}
__global__ void __launch_bounds__(256) affine_sample_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            const float* __restrict__ mats, int H, int W, int D) {
  const int k = blockIdx.z, i = blockIdx.y;
  const float* m = mats + 12 * k;
  const long long per = (long long)H * W * D;
  const float* v = in + k * per;
  const float fi = (float)i;
  const float bx = m[0] * fi + m[3], by = m[4] * fi + m[7], bz = m[8] * fi + m[11];
  for (int t = blockIdx.x * 256 + threadIdx.x; t < W * D; t += gridDim.x * 256) {
    const int j = t / D, d = t - j * D;
    const float fj = (float)j, fd = (float)d;
    out[k * per + ((long long)i * W + j) * D + d] =
        tri_sample(v, H, W, D, bx + m[1] * fj + m[2] * fd, by + m[5] * fj + m[6] * fd, bz + m[9] * fj + m[10] * fd);
  }
}

// RandZoomd -> monai Zoom(keep_size=True): F.interpolate(scale_factor = zoom, trilinear, align_corners=True) to
// Z = floor(size * zoom) per axis (zsize [n][3], computed on the host in double as torch does), then a centred edge-pad (zoom < 1) or centre crop (zoom > 1) back to [H][W][D], fused:
// out[p] = interp(clamp(p -/+ half, 0, Z-1)),  half = |size - Z| / 2,  source = q * (size-1)/(Z-1), taps floor and floor+1.
struct ZoomAxis { int Z, off; float scale; };        // q = clamp(p + off, 0, Z-1)
__device__ __forceinline__ ZoomAxis zoom_axis(int size, int Z) {
  ZoomAxis a;
  a.Z = Z < 1 ? 1 : Z;
  const int diff = size - a.Z, half = (diff < 0 ? -diff : diff) / 2;
  a.off = diff > 0 ? -half : half;
  a.scale = a.Z > 1 ? (float)(size - 1) / (float)(a.Z - 1) : 0.f;
  return a;
}
__device__ __forceinline__ void zoom_tap(const ZoomAxis& a, int p, int size, int* i0, int* i1, float* lam) {
  int q = p + a.off;
  q = q < 0 ? 0 : (q > a.Z - 1 ? a.Z - 1 : q);
  const float src = a.scale * (float)q;
  int x0 = (int)src;
  if (x0 > size - 1) x0 = size - 1;
  *i0 = x0;
  *i1 = x0 + (x0 < size - 1 ? 1 : 0);
  *lam = src - (float)x0;
}
__global__ void __launch_bounds__(256) zoom_sample_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          const int* __restrict__ zsize, int H, int W, int D) {
  const int k = blockIdx.z, i = blockIdx.y;
  const ZoomAxis ah = zoom_axis(H, zsize[3 * k]), aw = zoom_axis(W, zsize[3 * k + 1]), ad = zoom_axis(D, zsize[3 * k + 2]);
  const long long per = (long long)H * W * D;
  const float* v = in + k * per;
  int h0, h1; float lh;
  zoom_tap(ah, i, H, &h0, &h1, &lh);
  for (int t = blockIdx.x * 256 + threadIdx.x; t < W * D; t += gridDim.x * 256) {
    const int j = t / D, d = t - j * D;
    int w0, w1, d0, d1; float lw, ld;
    zoom_tap(aw, j, W, &w0, &w1, &lw);
    zoom_tap(ad, d, D, &d0, &d1, &ld);
    const float* r00 = v + ((long long)h0 * W + w0) * D, *r01 = v + ((long long)h0 * W + w1) * D;
    const float* r10 = v + ((long long)h1 * W + w0) * D, *r11 = v + ((long long)h1 * W + w1) * D;
    // same association as upsample_trilinear3d: depth-wise lerps first, then w, then h
    const float c00 = (1.f - ld) * r00[d0] + ld * r00[d1], c01 = (1.f - ld) * r01[d0] + ld * r01[d1];
    const float c10 = (1.f - ld) * r10[d0] + ld * r10[d1], c11 = (1.f - ld) * r11[d0] + ld * r11[d1];
    const float c0 = (1.f - lw) * c00 + lw * c01, c1 = (1.f - lw) * c10 + lw * c11;
    out[k * per + ((long long)i * W + j) * D + d] = (1.f - lh) * c0 + lh * c1;
  }
}

// RandAdjustContrastd -> monai AdjustContrast: ((x - min) / (range + 1e-7))^gamma * range + min over the whole patch.
// Order-preserving integer keys let the extrema come from integer atomics.
__device__ __forceinline__ int f2key(float f) { const int b = __float_as_int(f); return b >= 0 ? b : b ^ 0x7fffffff; }
__device__ __forceinline__ float key2f(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }
__global__ void minmax_init_kernel(int* __restrict__ mm, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { mm[2 * i] = 0x7fffffff; mm[2 * i + 1] = (int)0x80000000; }
}
__global__ void __launch_bounds__(256) minmax_kernel(const float* __restrict__ in, int* __restrict__ mm, long long per) {
  const int k = blockIdx.y;
  const float* v = in + k * per;
  float lo = INFINITY, hi = -INFINITY;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long long)gridDim.x * 256) {
    const float x = v[i];
    lo = fminf(lo, x); hi = fmaxf(hi, x);
  }
  lo = -wave_max(-lo); hi = wave_max(hi);
  __shared__ float sl[4], sh[4];
  if ((threadIdx.x & 63) == 0) { sl[threadIdx.x >> 6] = lo; sh[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    lo = fminf(fminf(sl[0], sl[1]), fminf(sl[2], sl[3]));
    hi = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    atomicMin(mm + 2 * k, f2key(lo));
    atomicMax(mm + 2 * k + 1, f2key(hi));
  }
}
__global__ void __launch_bounds__(256) contrast_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                       const float* __restrict__ gamma, const int* __restrict__ mm, long long per) {
  const int k = blockIdx.y;
  const float g = gamma[k];
  const float lo = key2f(mm[2 * k]), range = key2f(mm[2 * k + 1]) - lo;
  const float inv = 1.f / (range + 1e-7f);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long long)gridDim.x * 256) {
    const float x = in[k * per + i];
    // t^g = exp2(g log2 t) on the hardware transcendentals (t in [0,1]; log2(0) = -inf gives 0); libm powf costs ~100 instructions
    out[k * per + i] = g > 0.f ? __builtin_amdgcn_exp2f(g * __builtin_amdgcn_logf((x - lo) * inv)) * range + lo : x;
  }
}

extern "C" int ltu_affine_sample(const float* in, float* out, const float* mats, int n, int H, int W, int D, ltu_stream_t s) {
  if (n <= 0) return LTU_OK;
  if (H < 1 || W < 1 || D < 1 || H > 65535 || n > 65535 || (long long)W * D >= (1LL << 31)) return LTU_E_SHAPE;
  const unsigned gx = (unsigned)(((long long)W * D + 1023) / 1024);
  hipLaunchKernelGGL(affine_sample_kernel, dim3(gx, H, n), dim3(256), 0, (hipStream_t)s, in, out, mats, H, W, D);
  return ltu_check_launch();
}
extern "C" int ltu_zoom_sample(const float* in, float* out, const int* zsize, int n, int H, int W, int D, ltu_stream_t s) {
  if (n <= 0) return LTU_OK;
  if (H < 1 || W < 1 || D < 1 || H > 65535 || n > 65535 || (long long)W * D >= (1LL << 31)) return LTU_E_SHAPE;
  const unsigned gx = (unsigned)(((long long)W * D + 1023) / 1024);
  hipLaunchKernelGGL(zoom_sample_kernel, dim3(gx, H, n), dim3(256), 0, (hipStream_t)s, in, out, zsize, H, W, D);
  return ltu_check_launch();
}
extern "C" int ltu_adjust_contrast(const float* in, float* out, const float* gamma, int* minmax_ws, int n, long long per,
                                   ltu_stream_t s) {
  if (n <= 0 || per <= 0) return LTU_OK;
  if (n > 65535) return LTU_E_SHAPE;
  long long blocks = (per + 4095) / 4096;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(minmax_init_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)s, minmax_ws, n);
  hipLaunchKernelGGL(minmax_kernel, dim3((unsigned)blocks, n), dim3(256), 0, (hipStream_t)s, in, minmax_ws, per);
  hipLaunchKernelGGL(contrast_kernel, dim3((unsigned)blocks, n), dim3(256), 0, (hipStream_t)s, in, out, gamma, minmax_ws, per);
  return ltu_check_launch();
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
