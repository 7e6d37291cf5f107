// Dynamic-ROI machinery for gfx950: per-sample box finder, separable sampling plans, and the
// crop-and-warp / warp-back bilinear resamplers (forward + adjoint), all on device with no host sync.
//
// Reference: model/Unet_3Dblock.py 821-873 + 37-49 (box), 51-82 (piecewise-linear index maps),
// 985-1039 / 1080-1117 (F.grid_sample bilinear, zeros padding, align_corners=True on every depth slice).
// The sampling grid is an outer product of a 1-D map along H and a 1-D map along W, so a "plan" holds,
// per axis and sample, for every destination index its two source taps (gather form, used forward) and
// for every source index the list of destinations that touch it (CSR form, used by the adjoint - no
// atomics).  The index arithmetic repeats the reference's fp32 operation order without fma contraction.
#include "common.h"

#define ROI_MAX_LEN 512

struct AxisPlan {
  int* src0;     // [B][ND]      first source tap (clamped into range)
  float* wt;     // [B][ND][2]   weights of taps src0, src0+1 (0 where the tap is outside the source)
  int* cnt;      // [B][NS]
  int* lidx;     // [B][NS][L]   destinations touching this source index
  float* lw;     // [B][NS][L]
  int ND, NS, L;
};

__device__ __forceinline__ float fsub(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float fadd(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float fmul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float fdiv(float a, float b) { return __fdiv_rn(a, b); }

// model/Unet_3Dblock.py:51-64
__device__ float map_fwd(float idx, float x0, float x1, int span, int roi, int eval_roi) {
  const float k2 = fdiv(fsub(x1, x0), (float)(roi - 1));
  const float k1 = fdiv(fadd(fsub((float)span, x1), x0), (float)(eval_roi - roi));
  float pos = fadd(fmul(idx, k2), fmul(x0, fsub(1.f, fdiv(k2, k1))));
  const float r = fdiv(k1, k2);
  if (pos <= x0) pos = fadd(fmul(pos, r), fmul(x0, fsub(1.f, r)));
  if (pos >= x1) pos = fadd(fmul(pos, r), fmul(x1, fsub(1.f, r)));
  return fsub(fdiv(fmul(pos, 2.f), (float)span), 1.f);
}
// model/Unet_3Dblock.py:66-82
__device__ float map_back(float idx, float x0, float x1, int span, int roi, int eval_roi) {
  const float k2 = fdiv((float)roi, fsub(x1, x0));
  const float k1 = fdiv((float)(eval_roi - roi), fadd(fsub((float)span, x1), x0));
  const float p0 = fmul(x0, k1);
  const float p1 = fsub((float)eval_roi, fmul(fsub((float)span, x1), k1));
  float pos = fadd(fmul(idx, k2), fmul(p0, fsub(1.f, fdiv(k2, k1))));
  const float r = fdiv(k1, k2);
  if (pos <= p0) pos = fadd(fmul(pos, r), fmul(p0, fsub(1.f, r)));
  if (pos >= p1) pos = fadd(fmul(pos, r), fmul(p1, fsub(1.f, r)));
  return fsub(fdiv(fmul(pos, 2.f), (float)eval_roi), 1.f);
}

// grid_sample tap of a normalised coordinate (align_corners=True, zeros padding).  src0 may be -1
// (then only tap src0+1 = 0 can carry weight); taps outside the source get weight 0.
__device__ void make_tap(float g, int size, int* src0, float* w0, float* w1) {
  const float ix = fmul(fdiv(fadd(g, 1.f), 2.f), (float)(size - 1));
  const float fl = floorf(ix);
  if (!(fl >= -1.f && fl <= (float)(size - 1))) {   // entirely outside, or NaN
    *src0 = 0; *w0 = 0.f; *w1 = 0.f;
    return;
  }
  const int i0 = (int)fl;
  float a = fsub(fadd(fl, 1.f), ix), b = fsub(ix, fl);
  if (i0 < 0) a = 0.f;
  if (i0 + 1 > size - 1) b = 0.f;
  *src0 = i0; *w0 = a; *w1 = b;
}

__device__ void quantiles(const int* hist, int len, float* lo, float* hi, float* center) {
  int total = 0;
  for (int i = 0; i < len; ++i) total += hist[i];
  if (total == 0) {
    const float mid = (float)len / 2.f;
    *lo = mid - 1.f; *hi = mid + 1.f; *center = mid;
    return;
  }
  int l = len, h = len, m = len, run = 0;
  const float ft = (float)total;
  for (int i = 0; i < len; ++i) {
    run += hist[i];
    const float ratio = fdiv((float)run, ft);
    if (l == len && ratio >= 0.001f) l = i;
    if (h == len && ratio > 0.999f) h = i;
    if (m == len && ratio > 0.5f) m = i;
  }
  *lo = (float)l; *hi = (float)h; *center = (float)m;
}

__device__ void fit_extent(float lo, float hi, float center, int full, int min_len, float* out_lo, float* out_hi) {
  const float size = hi - lo;
  float a = lo, b = hi;
  if (size < (float)min_len) {
    const float half = (float)min_len / 2.f;
    a = fmaxf(center - half, 0.f);
    b = fminf(center + half, (float)full);
  }
  if (size > (float)(full - min_len)) {
    const float half = (float)(full - min_len) / 2.f;
    a = fmaxf(center - half, 0.f);
    b = fminf(center + half, (float)full);
  }
  *out_lo = a; *out_hi = b;
}

struct RoiArgs {
  const float* prob;   // [B][H][W][D][C] class probabilities; foreground = 1 - prob[..., 0]
  int B, H, W, D, C;
  float thr;
  int h_roi, w_roi, eval_h, eval_w, min_h, min_w;
  float* box;          // [B][6]
  AxisPlan fh, fw, bh, bw;
};

__device__ void build_axis(const AxisPlan& p, int b, bool forward, float x0, float x1, int span, int roi, int eval_roi) {
  for (int i = threadIdx.x; i < p.ND; i += blockDim.x) {
    const float g = forward ? map_fwd((float)i, x0, x1, span, roi, eval_roi) : map_back((float)i, x0, x1, span, roi, eval_roi);
    int s0; float w0, w1;
    make_tap(g, p.NS, &s0, &w0, &w1);
    p.src0[(long long)b * p.ND + i] = s0;
    p.wt[((long long)b * p.ND + i) * 2] = w0;
    p.wt[((long long)b * p.ND + i) * 2 + 1] = w1;
  }
  __syncthreads();
  for (int x = threadIdx.x; x < p.NS; x += blockDim.x) {
    int n = 0;
    int* li = p.lidx + ((long long)b * p.NS + x) * p.L;
    float* lw = p.lw + ((long long)b * p.NS + x) * p.L;
    for (int i = 0; i < p.ND; ++i) {
      const int s0 = p.src0[(long long)b * p.ND + i];
      const float w0 = p.wt[((long long)b * p.ND + i) * 2], w1 = p.wt[((long long)b * p.ND + i) * 2 + 1];
      if (s0 == x && w0 != 0.f && n < p.L) { li[n] = i; lw[n] = w0; ++n; }
      if (s0 + 1 == x && w1 != 0.f && n < p.L) { li[n] = i; lw[n] = w1; ++n; }
    }
    p.cnt[(long long)b * p.NS + x] = n;
  }
  __syncthreads();
}

// foreground histograms over h and w: grid (nblk, B); hist [B][2][ROI_MAX_LEN] zero on entry.  A block counts a contiguous
// range of voxels in LDS and flushes its non-zero bins (a range touches few h rows) with integer atomics.
__global__ void __launch_bounds__(256) roi_hist_kernel(const float* __restrict__ prob, int H, int W, int D, int C, float thr,
                                                       int* __restrict__ hist) {
  __shared__ int hx[ROI_MAX_LEN], hy[ROI_MAX_LEN];
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < ROI_MAX_LEN; i += blockDim.x) { hx[i] = 0; hy[i] = 0; }
  __syncthreads();
  const long long n = (long long)H * W * D;
  const long long per = (n + gridDim.x - 1) / gridDim.x;
  const long long i0 = (long long)blockIdx.x * per;
  const long long i1 = i0 + per < n ? i0 + per : n;
  const float* pb = prob + (long long)b * n * C;
  for (long long i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
    const float fg = fsub(1.f, pb[i * C]);
    if (fg >= thr) {
      const long long hw = i / D;
      atomicAdd(&hx[(int)(hw / W)], 1);
      atomicAdd(&hy[(int)(hw % W)], 1);
    }
  }
  __syncthreads();
  int* hb = hist + (long long)b * 2 * ROI_MAX_LEN;
  for (int i = threadIdx.x; i < ROI_MAX_LEN; i += blockDim.x) {
    if (hx[i]) atomicAdd(hb + i, hx[i]);
    if (hy[i]) atomicAdd(hb + ROI_MAX_LEN + i, hy[i]);
  }
}

__global__ void roi_plan_kernel(const RoiArgs a, const int* __restrict__ hist) {
  __shared__ int hx[ROI_MAX_LEN], hy[ROI_MAX_LEN];
  __shared__ float bx[4];
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < ROI_MAX_LEN; i += blockDim.x) {
    hx[i] = hist[(long long)b * 2 * ROI_MAX_LEN + i];
    hy[i] = hist[(long long)b * 2 * ROI_MAX_LEN + ROI_MAX_LEN + i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float lo, hi, c;
    quantiles(hx, a.H, &lo, &hi, &c);
    fit_extent(lo, hi, c, a.H, a.min_h, &bx[0], &bx[2]);
    quantiles(hy, a.W, &lo, &hi, &c);
    fit_extent(lo, hi, c, a.W, a.min_w, &bx[1], &bx[3]);
    float* o = a.box + b * 6;
    if (blockIdx.y == 0) {
    o[0] = bx[0]; o[1] = bx[1]; o[2] = 0.f; o[3] = bx[2]; o[4] = bx[3]; o[5] = (float)(a.D - 1);
    }
  }
  __syncthreads();
  const float x0 = bx[0], y0 = bx[1], x1 = bx[2], y1 = bx[3];
  // grid (B, 4): every workgroup derives the box (cheap) and builds one of the four axis plans
  if (blockIdx.y == 0) build_axis(a.fh, b, true, x0, x1, a.H - 1, a.h_roi, a.eval_h);
  else if (blockIdx.y == 1) build_axis(a.fw, b, true, y0, y1, a.W - 1, a.w_roi, a.eval_w);
  else if (blockIdx.y == 2) build_axis(a.bh, b, false, x0, x1, a.H - 1, a.h_roi, a.eval_h);
  else build_axis(a.bw, b, false, y0, y1, a.W - 1, a.w_roi, a.eval_w);
}

// plan buffers: ints  = [src0 B*ND | cnt B*NS | lidx B*NS*L],  floats = [wt B*ND*2 | lw B*NS*L],  L = 2*ND
static void carve_plan(AxisPlan& p, int B, int ND, int NS, int* ibuf, float* fbuf, long long* ioff, long long* foff) {
  p.ND = ND; p.NS = NS; p.L = 2 * ND;
  p.src0 = ibuf + *ioff; *ioff += (long long)B * ND;
  p.cnt = ibuf + *ioff; *ioff += (long long)B * NS;
  p.lidx = ibuf + *ioff; *ioff += (long long)B * NS * p.L;
  p.wt = fbuf + *foff; *foff += (long long)B * ND * 2;
  p.lw = fbuf + *foff; *foff += (long long)B * NS * p.L;
}
static void carve_all(RoiArgs& a, int B, int H, int W, int eval_h, int eval_w, int* ibuf, float* fbuf, long long* ni, long long* nf) {
  long long io = 0, fo = 0;
  carve_plan(a.fh, B, eval_h, H, ibuf, fbuf, &io, &fo);
  carve_plan(a.fw, B, eval_w, W, ibuf, fbuf, &io, &fo);
  // the warp-back reads the un-embedded grid, whose H/W are 2*ceil(eval/2) (stride-2 embed, nearest x2 un-embed);
  // the reference normalises by eval and lets grid_sample un-normalise by the actual size (Unet_3Dblock.py:1101-1112)
  carve_plan(a.bh, B, H, 2 * ((eval_h + 1) / 2), ibuf, fbuf, &io, &fo);
  carve_plan(a.bw, B, W, 2 * ((eval_w + 1) / 2), ibuf, fbuf, &io, &fo);
  *ni = io; *nf = fo;
}

extern "C" int ltu_roi_plan_size(int B, int H, int W, int roi_size, long long* n_int, long long* n_float, long long* n_hist) {
  RoiArgs a;
  const int eval_h = (int)(1.2 * roi_size), eval_w = (int)(eval_h * 0.6);
  carve_all(a, B, H, W, eval_h, eval_w, nullptr, nullptr, n_int, n_float);
  *n_hist = (long long)B * 2 * ROI_MAX_LEN;      // foreground histograms: a separate, zero-filled scratch buffer
  return LTU_OK;
}

// hist: B * 2 * ROI_MAX_LEN ints, ZERO on entry (caller-provided scratch, like the loss sums).  It used to be the tail of plan_i,
// cleared here with hipMemsetAsync; inside a captured HIP graph that memset node was not reliably ordered before the histogram
// kernel on replays (the plan then saw an empty histogram and fell back to the empty-mask box), so the library issues no memsets.
extern "C" int ltu_roi_plan(const float* prob, int B, int H, int W, int D, int C, int roi_size, float thr, float* box, int* plan_i,
                            float* plan_f, int* hist, ltu_stream_t s) {
  if (H > ROI_MAX_LEN || W > ROI_MAX_LEN) return LTU_E_SHAPE;
  RoiArgs a;
  a.prob = prob; a.B = B; a.H = H; a.W = W; a.D = D; a.C = C; a.thr = thr; a.box = box;
  a.h_roi = roi_size;
  a.w_roi = (int)(roi_size * 0.6);
  a.eval_h = (int)(1.2 * roi_size);
  a.eval_w = (int)(a.eval_h * 0.6);
  a.min_h = a.eval_h / 2;
  a.min_w = a.eval_w / 2;
  long long ni, nf;
  carve_all(a, B, H, W, a.eval_h, a.eval_w, plan_i, plan_f, &ni, &nf);
  if (hist == nullptr) return LTU_E_ARG;
  const long long n = (long long)H * W * D;
  const int nblk = (int)(n / 4096 < 1 ? 1 : (n / 4096 > 256 ? 256 : n / 4096));
  hipLaunchKernelGGL(roi_hist_kernel, dim3(nblk, B), dim3(256), 0, (hipStream_t)s, prob, H, W, D, C, thr, hist);
  hipLaunchKernelGGL(roi_plan_kernel, dim3(B, 4), dim3(256), 0, (hipStream_t)s, a, (const int*)hist);
  return ltu_check_launch();
}

// ------------------------------------------------------------------------------------------------ resamplers
// dst[b,i,j,d,:] = sum_{a,e} wh[i][a] ww[j][e] src[b, h0(i)+a, w0(j)+e, d, :]
// grid (blocks, B * ND_h): the (b, i) row of the destination comes from blockIdx.y (its h taps are hoisted), threads walk
// (j, d, channel quad) with 32-bit arithmetic.
template <typename T>
__global__ void __launch_bounds__(256) plan_gather_kernel(const T* __restrict__ src, T* __restrict__ dst, const AxisPlan ph,
                                                          const AxisPlan pw, int B, int D, int C) {
  const unsigned cv = C / 4;
  const unsigned b = blockIdx.y / (unsigned)ph.ND, i = blockIdx.y - b * (unsigned)ph.ND;
  const int h0 = ph.src0[b * ph.ND + i];
  const float wh0 = ph.wt[(b * ph.ND + i) * 2], wh1 = ph.wt[(b * ph.ND + i) * 2 + 1];
  const int hs0 = min(max(h0, 0), ph.NS - 1), hs1 = min(max(h0 + 1, 0), ph.NS - 1);
  const unsigned n = (unsigned)pw.ND * D * cv;
  const T* sb = src + (long long)b * ph.NS * pw.NS * D * C;
  T* db = dst + (long long)blockIdx.y * n * 4;
  for (unsigned t = blockIdx.x * 256u + threadIdx.x; t < n; t += gridDim.x * 256u) {
    const unsigned r = t / cv, v = t - r * cv;
    const unsigned j = r / (unsigned)D, d = r - j * (unsigned)D;
    const int w0 = pw.src0[b * pw.ND + j];
    const float ww0 = pw.wt[(b * pw.ND + j) * 2], ww1 = pw.wt[(b * pw.ND + j) * 2 + 1];
    const int ws0 = min(max(w0, 0), pw.NS - 1), ws1 = min(max(w0 + 1, 0), pw.NS - 1);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float wgt[4] = {wh0 * ww0, wh0 * ww1, wh1 * ww0, wh1 * ww1};
    const int hh[4] = {hs0, hs0, hs1, hs1}, wv[4] = {ws0, ws1, ws0, ws1};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (wgt[q] == 0.f) continue;
      const float4 x = Vec4<T>::load(sb + (((long long)hh[q] * pw.NS + wv[q]) * D + d) * C + v * 4);
      acc.x += x.x * wgt[q]; acc.y += x.y * wgt[q]; acc.z += x.z * wgt[q]; acc.w += x.w * wgt[q];
    }
    Vec4<T>::store(db + (long long)t * 4, acc);
  }
}

// adjoint: dsrc[b,x,y,d,:] = sum_{(i,wi) in list_h(x)} sum_{(j,wj) in list_w(y)} wi wj ddst[b,i,j,d,:]
// grid (blocks, B * NS_h): row (b, x) from blockIdx.y, its h list hoisted.
template <typename T>
__global__ void __launch_bounds__(256) plan_scatter_kernel(const T* __restrict__ ddst, T* __restrict__ dsrc, const AxisPlan ph,
                                                           const AxisPlan pw, int B, int D, int C) {
  const unsigned cv = C / 4;
  const unsigned b = blockIdx.y / (unsigned)ph.NS, x = blockIdx.y - b * (unsigned)ph.NS;
  const int nh = ph.cnt[b * ph.NS + x];
  const int* lih = ph.lidx + (long long)(b * ph.NS + x) * ph.L;
  const float* lwh = ph.lw + (long long)(b * ph.NS + x) * ph.L;
  const unsigned n = (unsigned)pw.NS * D * cv;
  const T* gb = ddst + (long long)b * ph.ND * pw.ND * D * C;
  T* ob = dsrc + (long long)blockIdx.y * n * 4;
  for (unsigned t = blockIdx.x * 256u + threadIdx.x; t < n; t += gridDim.x * 256u) {
    const unsigned r = t / cv, v = t - r * cv;
    const unsigned y = r / (unsigned)D, d = r - y * (unsigned)D;
    const int nw = pw.cnt[b * pw.NS + y];
    const int* liw = pw.lidx + (long long)(b * pw.NS + y) * pw.L;
    const float* lww = pw.lw + (long long)(b * pw.NS + y) * pw.L;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int a = 0; a < nh; ++a) {
      const int i = lih[a];
      const float wi = lwh[a];
      for (int e = 0; e < nw; ++e) {
        const float w = wi * lww[e];
        const float4 q = Vec4<T>::load(gb + (((long long)i * pw.ND + liw[e]) * D + d) * C + v * 4);
        acc.x += q.x * w; acc.y += q.y * w; acc.z += q.z * w; acc.w += q.w * w;
      }
    }
    Vec4<T>::store(ob + (long long)t * 4, acc);
  }
}

static unsigned rgrid(long long n) {
  long long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

// which: 0 = image -> ROI grid (roi_alignment2), 1 = ROI grid -> image (post_processing2); adjoint != 0 runs the transpose.
extern "C" int ltu_roi_resample(const void* in, void* out, int* plan_i, float* plan_f, int which, int adjoint, int B, int H, int W,
                                int D, int C, int roi_size, int dtype, ltu_stream_t s) {
  if (C % 4) return LTU_E_SHAPE;
  RoiArgs a;
  const int eval_h = (int)(1.2 * roi_size), eval_w = (int)(eval_h * 0.6);
  long long ni, nf;
  carve_all(a, B, H, W, eval_h, eval_w, plan_i, plan_f, &ni, &nf);
  const AxisPlan& ph = which == 0 ? a.fh : a.bh;
  const AxisPlan& pw = which == 0 ? a.fw : a.bw;
  LTU_DISPATCH_T(dtype, {
    if (!adjoint) {
      const long long n = (long long)pw.ND * D * (C / 4);
      if (n >= (1LL << 31) || (long long)B * ph.ND > 65535) return LTU_E_SHAPE;
      hipLaunchKernelGGL((plan_gather_kernel<T>), dim3((unsigned)((n + 511) / 512), B * ph.ND), dim3(256), 0, (hipStream_t)s,
                         (const T*)in, (T*)out, ph, pw, B, D, C);
    } else {
      const long long n = (long long)pw.NS * D * (C / 4);
      if (n >= (1LL << 31) || (long long)B * ph.NS > 65535) return LTU_E_SHAPE;
      hipLaunchKernelGGL((plan_scatter_kernel<T>), dim3((unsigned)((n + 511) / 512), B * ph.NS), dim3(256), 0, (hipStream_t)s,
                         (const T*)in, (T*)out, ph, pw, B, D, C);
    }
  });
  return ltu_check_launch();
}

// ------------------------------------------------------------------------------------------------ trilinear x2
// nn.Upsample(scale (2,2,2) | (2,2,1), trilinear, align_corners=True)  (model/Unet_3Dblock.py:1341-1345)
// src coordinate of output o along an axis: o * (in-1)/(out-1) in fp32, taps floor and floor+1 (clamped).
// `scale` = (float)(in-1)/(float)(out-1) (0 when out == 1), computed once on the host with the same fp32 division.
__device__ __forceinline__ void tri_tap(int o, int in, int out, float scale, int* i0, int* i1, float* l0, float* l1) {
  if (in == out) { *i0 = o; *i1 = o; *l0 = 1.f; *l1 = 0.f; return; }
  const float src = scale * (float)o;
  int a = (int)src;
  if (a > in - 1) a = in - 1;
  const int p = a < in - 1 ? 1 : 0;
  const float lam = src - (float)a;
  *i0 = a; *i1 = a + p; *l1 = lam; *l0 = 1.f - lam;
}
struct TriScale { float h, w, d, ih, iw, id; };      // forward scales and their inverses (out-1)/(in-1) per axis

// grid (blocks, B*Ho): the (b, h) plane comes from blockIdx.y, threads walk (w, d, channel quad) with 32-bit arithmetic
// (a first version decomposed a 64-bit linear index and divided floats per output vector: VALU-bound).
template <typename T>
__global__ void __launch_bounds__(256) trilinear_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int D,
                                                            int C, int Ho, int Wo, int Do, TriScale sc) {
  const unsigned cv = C / 4;
  const unsigned b = blockIdx.y / (unsigned)Ho, h = blockIdx.y - b * (unsigned)Ho;
  int hs[2], ws[2], dsv[2];
  float lh[2], lw[2], ld[2];
  tri_tap((int)h, H, Ho, sc.h, &hs[0], &hs[1], &lh[0], &lh[1]);
  const unsigned n = (unsigned)Wo * Do * cv;
  const T* xb = x + (long long)b * H * W * D * C;
  T* yb = y + ((long long)b * Ho + h) * n * 4;
  for (unsigned t = blockIdx.x * 256u + threadIdx.x; t < n; t += gridDim.x * 256u) {
    const unsigned r = t / cv, v = t - r * cv;
    const unsigned w = r / (unsigned)Do, d = r - w * (unsigned)Do;
    tri_tap((int)w, W, Wo, sc.w, &ws[0], &ws[1], &lw[0], &lw[1]);
    tri_tap((int)d, D, Do, sc.d, &dsv[0], &dsv[1], &ld[0], &ld[1]);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const float wgt = lh[a] * lw[e] * ld[f];
          if (wgt == 0.f) continue;
          const float4 q = Vec4<T>::load(xb + (((long long)hs[a] * W + ws[e]) * D + dsv[f]) * C + v * 4);
          acc.x += q.x * wgt; acc.y += q.y * wgt; acc.z += q.z * wgt; acc.w += q.w * wgt;
        }
    Vec4<T>::store(yb + (long long)t * 4, acc);
  }
}

// adjoint in gather form: candidates o with src(o) in (i-1, i+1), tested with the forward's own tap function
__device__ __forceinline__ int tri_cands(int i, int in, int out, float scale, float inv, int* os, float* wsum) {
  if (in == out) { os[0] = i; wsum[0] = 1.f; return 1; }
  int n = 0;
  int lo = (int)floorf((float)(i - 1) * inv) - 1, hi = (int)ceilf((float)(i + 1) * inv) + 1;
  if (lo < 0) lo = 0;
  if (hi > out - 1) hi = out - 1;
  for (int o = lo; o <= hi && n < 8; ++o) {
    int i0, i1; float l0, l1;
    tri_tap(o, in, out, scale, &i0, &i1, &l0, &l1);
    float w = 0.f;
    if (i0 == i) w += l0;
    if (i1 == i) w += l1;
    if (w != 0.f) { os[n] = o; wsum[n] = w; ++n; }
  }
  return n;
}

// grid (blocks, B*H)
template <typename T>
__global__ void __launch_bounds__(256) trilinear_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ dy2, T* __restrict__ dx, int B,
                                                            int H, int W, int D, int C, int Ho, int Wo, int Do, TriScale sc) {
  const unsigned cv = C / 4;
  const unsigned b = blockIdx.y / (unsigned)H, h = blockIdx.y - b * (unsigned)H;
  int oh[8], ow[8], od[8];
  float wh[8], ww[8], wd[8];
  const int nh = tri_cands((int)h, H, Ho, sc.h, sc.ih, oh, wh);
  const unsigned n = (unsigned)W * D * cv;
  const T* gb = dy + (long long)b * Ho * Wo * Do * C;
  const T* gb2 = dy2 != nullptr ? dy2 + (long long)b * Ho * Wo * Do * C : nullptr;      // second consumer's gradient: summed on load
  T* xb = dx + ((long long)b * H + h) * n * 4;
  for (unsigned t = blockIdx.x * 256u + threadIdx.x; t < n; t += gridDim.x * 256u) {
    const unsigned r = t / cv, v = t - r * cv;
    const unsigned w = r / (unsigned)D, d = r - w * (unsigned)D;
    const int nw = tri_cands((int)w, W, Wo, sc.w, sc.iw, ow, ww), nd = tri_cands((int)d, D, Do, sc.d, sc.id, od, wd);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int a = 0; a < nh; ++a)
      for (int e = 0; e < nw; ++e)
        for (int f = 0; f < nd; ++f) {
          const float wgt = wh[a] * ww[e] * wd[f];
          const long long off = (((long long)oh[a] * Wo + ow[e]) * Do + od[f]) * C + v * 4;
          float4 q = Vec4<T>::load(gb + off);
          if (gb2 != nullptr) { const float4 q2 = Vec4<T>::load(gb2 + off); q.x += q2.x; q.y += q2.y; q.z += q2.z; q.w += q2.w; }
          acc.x += q.x * wgt; acc.y += q.y * wgt; acc.z += q.z * wgt; acc.w += q.w * wgt;
        }
    Vec4<T>::store(xb + (long long)t * 4, acc);
  }
}

// The same with the (w, d) candidate lists of the plane built once per workgroup in LDS (W, D <= 256): the kernel above spends
// ~150 instructions per output vector re-deriving them.
#define TRI_TAB 256
template <typename T>
__global__ void __launch_bounds__(256) trilinear_bwd_tab_kernel(const T* __restrict__ dy, const T* __restrict__ dy2, T* __restrict__ dx,
                                                                int B, int H, int W, int D, int C, int Ho, int Wo, int Do, TriScale sc) {
  __shared__ int t_n[2][TRI_TAB];
  __shared__ int t_o[2][TRI_TAB][8];
  __shared__ float t_w[2][TRI_TAB][8];
  for (int idx = threadIdx.x; idx < W + D; idx += 256) {
    const int ax = idx < W ? 0 : 1, i = ax ? idx - W : idx;
    int os[8]; float ws[8];
    const int n = ax ? tri_cands(i, D, Do, sc.d, sc.id, os, ws) : tri_cands(i, W, Wo, sc.w, sc.iw, os, ws);
    t_n[ax][i] = n;
#pragma unroll
    for (int q = 0; q < 8; ++q) { t_o[ax][i][q] = q < n ? os[q] : 0; t_w[ax][i][q] = q < n ? ws[q] : 0.f; }
  }
  const unsigned cv = C / 4;
  const unsigned b = blockIdx.y / (unsigned)H, h = blockIdx.y - b * (unsigned)H;
  int oh[8];
  float wh[8];
  const int nh = tri_cands((int)h, H, Ho, sc.h, sc.ih, oh, wh);
  __syncthreads();
  const unsigned n = (unsigned)W * D * cv;
  const T* gb = dy + (long long)b * Ho * Wo * Do * C;
  const T* gb2 = dy2 != nullptr ? dy2 + (long long)b * Ho * Wo * Do * C : nullptr;
  T* xb = dx + ((long long)b * H + h) * n * 4;
  for (unsigned t = blockIdx.x * 256u + threadIdx.x; t < n; t += gridDim.x * 256u) {
    const unsigned r = t / cv, v = t - r * cv;
    const unsigned w = r / (unsigned)D, d = r - w * (unsigned)D;
    const int nw = t_n[0][w], nd = t_n[1][d];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int a = 0; a < nh; ++a)
      for (int e = 0; e < nw; ++e) {
        const float whw = wh[a] * t_w[0][w][e];
        const long long roff = ((long long)oh[a] * Wo + t_o[0][w][e]) * Do * C + v * 4;
        for (int f = 0; f < nd; ++f) {
          const float wgt = whw * t_w[1][d][f];
          const long long off = roff + (long long)t_o[1][d][f] * C;
          float4 q = Vec4<T>::load(gb + off);
          if (gb2 != nullptr) { const float4 q2 = Vec4<T>::load(gb2 + off); q.x += q2.x; q.y += q2.y; q.z += q2.z; q.w += q2.w; }
          acc.x += q.x * wgt; acc.y += q.y * wgt; acc.z += q.z * wgt; acc.w += q.w * wgt;
        }
      }
    Vec4<T>::store(xb + (long long)t * 4, acc);
  }
}

// Separable form of the adjoint: trilinear interpolation is a product of three 1-D linear interpolations, so its transpose is
// three 1-D transposes applied one axis at a time (depth only when it was upsampled), each with at most 5 candidates per output
// instead of ~64 (4 x 4 x 4) gathers in the kernels above - those are bound by the number of L1 requests, not by bytes.
// View per pass: in [outer][L_fine][inner] -> out [outer][L_coarse][inner], `inner` contiguous.  grid (outer * L_coarse, blocks).
// 16-byte vectors of the storage type (8 bf16 / 4 fp32): accumulate NV floats
template <typename T> struct TriVec;
template <> struct TriVec<float> {
  static constexpr int NV = 4;
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct TriVec<bf16_t> {
  static constexpr int NV = 8;
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[2 * k] = __uint_as_float(w[k] << 16); v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u); }
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
  }
};
// one output row (outer, l) per blockIdx.x; the candidate count n is uniform in the workgroup, so the body is instantiated per
// n (all loads of a vector issued together, none wasted on empty slots)
template <typename T, bool HAS2, int N>
__device__ __forceinline__ void tri_adj1d_body(const T* ib, const T* ib2, T* ob, const int (&os)[8], const float (&ws)[8], long long inner,
                                               long long nvec) {
  constexpr int NV = TriVec<T>::NV;
  for (long long v = (long long)blockIdx.y * 256 + threadIdx.x; v < nvec; v += (long long)gridDim.y * 256) {
    float q[N][NV], q2[N][NV];
#pragma unroll
    for (int k = 0; k < N; ++k) {
      TriVec<T>::load(ib + (long long)os[k] * inner + v * NV, q[k]);
      if constexpr (HAS2) TriVec<T>::load(ib2 + (long long)os[k] * inner + v * NV, q2[k]);
    }
    float acc[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) acc[e] = 0.f;
#pragma unroll
    for (int k = 0; k < N; ++k)
#pragma unroll
      for (int e = 0; e < NV; ++e) acc[e] += (HAS2 ? q[k][e] + q2[k][e] : q[k][e]) * ws[k];
    TriVec<T>::store(ob + v * NV, acc);
  }
}
struct TriEntry { int n; int o[8]; float w[8]; int pad[15]; };       // 128 bytes
// entries [0, H) for the height axis, [H, H+W) width, [H+W, H+W+D) depth
__device__ __forceinline__ void tri_table_entry(int i, TriEntry* __restrict__ t, int H, int W, int D, int Ho, int Wo, int Do, TriScale sc) {
  if (i >= H + W + D) return;
  int os[8];
  float ws[8];
  int n;
  if (i < H) n = tri_cands(i, H, Ho, sc.h, sc.ih, os, ws);
  else if (i < H + W) n = tri_cands(i - H, W, Wo, sc.w, sc.iw, os, ws);
  else n = tri_cands(i - H - W, D, Do, sc.d, sc.id, os, ws);
  t[i].n = n;
  for (int q = 0; q < 8; ++q) { t[i].o[q] = q < n ? os[q] : os[0]; t[i].w[q] = q < n ? ws[q] : 0.f; }
}
template <typename T, bool HAS2>
__global__ void __launch_bounds__(256) tri_adj1d_kernel(const T* __restrict__ in, const T* __restrict__ in2, T* __restrict__ out,
                                                        int L_fine, int L_coarse, long long inner, const TriEntry* __restrict__ table) {
  const unsigned row = blockIdx.x;
  const unsigned outer = row / (unsigned)L_coarse, l = row - outer * (unsigned)L_coarse;
  // the candidate list of the row comes from a table built once per call (tri_table_kernel): deriving it is a serial search
  // with dynamically indexed arrays, ~150 instructions and scratch traffic - more than the few vectors a thread moves
  const TriEntry* te = table + l;
  int os[8];
  float ws[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { os[q] = te->o[q]; ws[q] = te->w[q]; }
  const int n = te->n;
  const long long nvec = inner / TriVec<T>::NV;
  const T* ib = in + (long long)outer * L_fine * inner;
  const T* ib2 = HAS2 ? in2 + (long long)outer * L_fine * inner : nullptr;
  T* ob = out + ((long long)outer * L_coarse + l) * inner;
  if (n <= 1) tri_adj1d_body<T, HAS2, 1>(ib, ib2, ob, os, ws, inner, nvec);
  else if (n == 2) tri_adj1d_body<T, HAS2, 2>(ib, ib2, ob, os, ws, inner, nvec);
  else if (n == 3) tri_adj1d_body<T, HAS2, 3>(ib, ib2, ob, os, ws, inner, nvec);
  else if (n == 4) tri_adj1d_body<T, HAS2, 4>(ib, ib2, ob, os, ws, inner, nvec);
  else if (n == 5) tri_adj1d_body<T, HAS2, 5>(ib, ib2, ob, os, ws, inner, nvec);
  else tri_adj1d_body<T, HAS2, 8>(ib, ib2, ob, os, ws, inner, nvec);
}

// Two adjacent output rows per workgroup: their candidate lists overlap (x2: rows 2l-1 .. 2l+2 and 2l+1 .. 2l+4), so the union is
// 5-6 input rows instead of 8 - a third fewer 16-byte loads (and sums of the two gradient tensors) per output vector.  The merged
// lists come from a table as well (TriPair: union rows, the weights of either output, zero where a row does not contribute).
struct TriPair { int n; int o[8]; float w0[8]; float w1[8]; int pad[7]; };     // 128 bytes; n < 0: union longer than 8, rows done singly
// entries [0, PH) for the height axis, [PH, PH + PW) width, then depth; P* = (L + 1) / 2
__device__ __forceinline__ void tri_pair_table_entry(int i, TriPair* __restrict__ t, int H, int W, int D, int Ho, int Wo, int Do, TriScale sc) {
  const int PH = (H + 1) / 2, PW = (W + 1) / 2, PD = (D + 1) / 2;
  if (i >= PH + PW + PD) return;
  int L, Lo, pr;
  float scale, inv;
  if (i < PH) { L = H; Lo = Ho; pr = i; scale = sc.h; inv = sc.ih; }
  else if (i < PH + PW) { L = W; Lo = Wo; pr = i - PH; scale = sc.w; inv = sc.iw; }
  else { L = D; Lo = Do; pr = i - PH - PW; scale = sc.d; inv = sc.id; }
  int o0[8], o1[8];
  float w0[8], w1[8];
  const int n0 = tri_cands(2 * pr, L, Lo, scale, inv, o0, w0);
  const int n1 = 2 * pr + 1 < L ? tri_cands(2 * pr + 1, L, Lo, scale, inv, o1, w1) : 0;
  TriPair e;
  int a = 0, b = 0, n = 0;
  bool over = false;
  for (int q = 0; q < 8; ++q) { e.o[q] = o0[0]; e.w0[q] = 0.f; e.w1[q] = 0.f; }
  while (a < n0 || b < n1) {
    if (n == 8) { over = true; break; }
    const int oa = a < n0 ? o0[a] : 0x7fffffff, ob = b < n1 ? o1[b] : 0x7fffffff;
    const int o = oa < ob ? oa : ob;
    e.o[n] = o;
    if (oa == o) { e.w0[n] = w0[a]; ++a; }
    if (ob == o) { e.w1[n] = w1[b]; ++b; }
    ++n;
  }
  e.n = over ? -1 : n;
  for (int q = 0; q < 7; ++q) e.pad[q] = 0;
  t[i] = e;
}
// both tables of a call in ONE launch (they used to be two 8-9 us one-workgroup kernels at the head of every adjoint, on the backward chain):
// workgroups [0, nb1) fill the row table, the rest the pair table (pt == nullptr: none)
__global__ void __launch_bounds__(128) tri_tables_kernel(TriEntry* __restrict__ t, TriPair* __restrict__ pt, int nb1, int H, int W, int D, int Ho,
                                                         int Wo, int Do, TriScale sc) {
  if ((int)blockIdx.x < nb1) tri_table_entry(blockIdx.x * 128 + threadIdx.x, t, H, W, D, Ho, Wo, Do, sc);
  else tri_pair_table_entry(((int)blockIdx.x - nb1) * 128 + threadIdx.x, pt, H, W, D, Ho, Wo, Do, sc);
}
template <typename T, bool HAS2, int N>
__device__ __forceinline__ void tri_adj1d_pair_body(const T* ib, const T* ib2, T* ob0, T* ob1, const int (&os)[8], const float (&w0)[8],
                                                    const float (&w1)[8], long long inner, long long nvec) {
  constexpr int NV = TriVec<T>::NV;
  for (long long v = (long long)blockIdx.y * 256 + threadIdx.x; v < nvec; v += (long long)gridDim.y * 256) {
    float q[N][NV], q2[N][NV];
#pragma unroll
    for (int k = 0; k < N; ++k) {
      TriVec<T>::load(ib + (long long)os[k] * inner + v * NV, q[k]);
      if constexpr (HAS2) TriVec<T>::load(ib2 + (long long)os[k] * inner + v * NV, q2[k]);
    }
    float a0[NV], a1[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) { a0[e] = 0.f; a1[e] = 0.f; }
#pragma unroll
    for (int k = 0; k < N; ++k)
#pragma unroll
      for (int e = 0; e < NV; ++e) {
        const float x = HAS2 ? q[k][e] + q2[k][e] : q[k][e];
        a0[e] += x * w0[k];
        a1[e] += x * w1[k];
      }
    TriVec<T>::store(ob0 + v * NV, a0);
    if (ob1 != nullptr) TriVec<T>::store(ob1 + v * NV, a1);
  }
}
// grid (outer * pairs, blocks): rows 2 pair and 2 pair + 1 of one outer slice
template <typename T, bool HAS2>
__global__ void __launch_bounds__(256) tri_adj1d_pair_kernel(const T* __restrict__ in, const T* __restrict__ in2, T* __restrict__ out,
                                                             int L_fine, int L_coarse, long long inner, const TriEntry* __restrict__ table,
                                                             const TriPair* __restrict__ ptable) {
  const unsigned npair = ((unsigned)L_coarse + 1u) >> 1;
  const unsigned outer = blockIdx.x / npair, pr = blockIdx.x - outer * npair;
  const int l0 = 2 * (int)pr, l1 = l0 + 1;
  const long long nvec = inner / TriVec<T>::NV;
  const T* ib = in + (long long)outer * L_fine * inner;
  const T* ib2 = HAS2 ? in2 + (long long)outer * L_fine * inner : nullptr;
  T* ob0 = out + ((long long)outer * L_coarse + l0) * inner;
  T* ob1 = l1 < L_coarse ? ob0 + inner : nullptr;
  const TriPair* tp = ptable + pr;
  const int n = tp->n;
  if (n < 0) {                               // union too long (never for x2): the two rows one after the other
    for (int which = 0; which < 2; ++which) {
      const int l = which == 0 ? l0 : l1;
      if (l >= L_coarse) break;
      const TriEntry* te = table + l;
      int os[8];
      float ws[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) { os[q] = te->o[q]; ws[q] = te->w[q]; }
      tri_adj1d_body<T, HAS2, 8>(ib, ib2, which == 0 ? ob0 : ob1, os, ws, inner, nvec);
    }
    return;
  }
  int os[8];
  float w0[8], w1[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { os[q] = tp->o[q]; w0[q] = tp->w0[q]; w1[q] = tp->w1[q]; }
  if (n <= 2) tri_adj1d_pair_body<T, HAS2, 2>(ib, ib2, ob0, ob1, os, w0, w1, inner, nvec);
  else if (n == 3) tri_adj1d_pair_body<T, HAS2, 3>(ib, ib2, ob0, ob1, os, w0, w1, inner, nvec);
  else if (n == 4) tri_adj1d_pair_body<T, HAS2, 4>(ib, ib2, ob0, ob1, os, w0, w1, inner, nvec);
  else if (n == 5) tri_adj1d_pair_body<T, HAS2, 5>(ib, ib2, ob0, ob1, os, w0, w1, inner, nvec);
  else if (n == 6) tri_adj1d_pair_body<T, HAS2, 6>(ib, ib2, ob0, ob1, os, w0, w1, inner, nvec);
  else tri_adj1d_pair_body<T, HAS2, 8>(ib, ib2, ob0, ob1, os, w0, w1, inner, nvec);
}

// Short rows (the depth pass: inner = C elements, 4-16 vectors): one pair of output rows per workgroup leaves 4-16 of its 256
// lanes busy (the 64x64x128 x 32 depth pass: 262 144 workgroups, 154 us for 167 MB).  Here a workgroup carries 256 / nvec
// pairs; every lane reads its own pair's table entry.  The union lists are padded with (row o[0], weight 0) entries, so the
// loads are unconditional (8 per tensor, all requested before the first add); a pair whose union is longer than 8 does its
// two rows singly from the per-row table.
template <typename T, bool HAS2>
__global__ void __launch_bounds__(256) tri_adj1d_pair_rows_kernel(const T* __restrict__ in, const T* __restrict__ in2, T* __restrict__ out,
                                                                  int L_fine, int L_coarse, int inner, int nvec, long long total,
                                                                  const TriEntry* __restrict__ table, const TriPair* __restrict__ ptable) {
  constexpr int NV = TriVec<T>::NV;
  const int rpb = 256 / nvec;
  const int sub = (int)threadIdx.x / nvec, v = (int)threadIdx.x - sub * nvec;
  const long long idx = (long long)blockIdx.x * rpb + sub;
  if (sub >= rpb || idx >= total) return;
  const unsigned npair = ((unsigned)L_coarse + 1u) >> 1;
  const long long outer = idx / npair;
  const int pr = (int)(idx - outer * npair);
  const int l0 = 2 * pr, l1 = l0 + 1;
  const T* ib = in + outer * L_fine * inner + v * NV;
  const T* ib2 = HAS2 ? in2 + outer * L_fine * inner + v * NV : nullptr;
  T* ob0 = out + (outer * L_coarse + l0) * inner + v * NV;
  const TriPair* tp = ptable + pr;
  const int n = tp->n;
  if (n < 0) {
    for (int which = 0; which < 2; ++which) {
      const int l = which == 0 ? l0 : l1;
      if (l >= L_coarse) break;
      const TriEntry* te = table + l;
      float acc[NV];
#pragma unroll
      for (int e = 0; e < NV; ++e) acc[e] = 0.f;
      for (int k = 0; k < 8; ++k) {
        float q[NV];
        TriVec<T>::load(ib + (long long)te->o[k] * inner, q);
        if constexpr (HAS2) {
          float q2[NV];
          TriVec<T>::load(ib2 + (long long)te->o[k] * inner, q2);
#pragma unroll
          for (int e = 0; e < NV; ++e) q[e] += q2[e];
        }
#pragma unroll
        for (int e = 0; e < NV; ++e) acc[e] += q[e] * te->w[k];
      }
      TriVec<T>::store(ob0 + (long long)which * inner, acc);
    }
    return;
  }
  int os[8];
  float w0[8], w1[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { os[q] = tp->o[q]; w0[q] = tp->w0[q]; w1[q] = tp->w1[q]; }
  float q[8][NV], q2[8][NV];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    TriVec<T>::load(ib + (long long)os[k] * inner, q[k]);
    if constexpr (HAS2) TriVec<T>::load(ib2 + (long long)os[k] * inner, q2[k]);
  }
  float a0[NV], a1[NV];
#pragma unroll
  for (int e = 0; e < NV; ++e) { a0[e] = 0.f; a1[e] = 0.f; }
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int e = 0; e < NV; ++e) {
      const float x = HAS2 ? q[k][e] + q2[k][e] : q[k][e];
      a0[e] += x * w0[k];
      a1[e] += x * w1[k];
    }
  TriVec<T>::store(ob0, a0);
  if (l1 < L_coarse) TriVec<T>::store(ob0 + inner, a1);
}

static float tri_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }
static float tri_inv(int in, int out) { return (float)(out - 1) / (float)(in - 1 > 0 ? in - 1 : 1); }

template <typename T>
static void launch_tri_adj1d(const T* in, const T* in2, T* out, long long outer, int L_fine, int L_coarse, long long inner,
                             const TriEntry* table, const TriPair* ptable, hipStream_t st) {
  const long long nv = inner / TriVec<T>::NV;
  unsigned gx = (unsigned)((nv + 255) / 256);
  if (gx > 64) gx = 64;
  if (gx < 1) gx = 1;
  if (ptable != nullptr && L_fine != L_coarse && nv <= 64 && inner < (1 << 20) && !ltu_knob("LTU_TRI_NO_ROWS", 0)) {
    const long long total = outer * ((L_coarse + 1) / 2);
    const int rpb = 256 / (int)nv;
    const dim3 rgrid((unsigned)((total + rpb - 1) / rpb));
    if (in2 != nullptr)
      hipLaunchKernelGGL((tri_adj1d_pair_rows_kernel<T, true>), rgrid, dim3(256), 0, st, in, in2, out, L_fine, L_coarse, (int)inner, (int)nv, total, table, ptable);
    else
      hipLaunchKernelGGL((tri_adj1d_pair_rows_kernel<T, false>), rgrid, dim3(256), 0, st, in, (const T*)nullptr, out, L_fine, L_coarse, (int)inner, (int)nv, total, table, ptable);
    return;
  }
  if (ptable != nullptr && L_fine != L_coarse) {
    const dim3 pgrid((unsigned)(outer * ((L_coarse + 1) / 2)), gx);
    if (in2 != nullptr) hipLaunchKernelGGL((tri_adj1d_pair_kernel<T, true>), pgrid, dim3(256), 0, st, in, in2, out, L_fine, L_coarse, inner, table, ptable);
    else hipLaunchKernelGGL((tri_adj1d_pair_kernel<T, false>), pgrid, dim3(256), 0, st, in, (const T*)nullptr, out, L_fine, L_coarse, inner, table, ptable);
    return;
  }
  const dim3 grid((unsigned)(outer * L_coarse), gx);
  if (in2 != nullptr) hipLaunchKernelGGL((tri_adj1d_kernel<T, true>), grid, dim3(256), 0, st, in, in2, out, L_fine, L_coarse, inner, table);
  else hipLaunchKernelGGL((tri_adj1d_kernel<T, false>), grid, dim3(256), 0, st, in, (const T*)nullptr, out, L_fine, L_coarse, inner, table);
}

extern "C" long long ltu_trilinear_adjoint_ws_elems(int B, int H, int W, int D, int C, int sd) {
  // depth pass output [B][2H][2W][D][C] (only when sd == 2) + width pass output [B][2H][W][D][C], in elements of the storage type,
  // + the candidate tables (H + W + D entries of 128 bytes, counted as 4-byte elements at most)
  const long long t2 = (long long)B * 2 * H * W * D * C;
  return (sd == 2 ? 2 * t2 + t2 : t2) + (long long)(H + W + D + 1) * 64 + (long long)((H + 1) / 2 + (W + 1) / 2 + (D + 1) / 2 + 2) * 64;
}

extern "C" int ltu_trilinear_adjoint(const void* dy, const void* dy2, void* dx, void* ws, long long ws_elems, int B, int H, int W, int D, int C,
                                     int sd, int dtype, ltu_stream_t s) {
  if (C % 4 || (dtype == LTU_BF16 && C % 8) || (sd != 1 && sd != 2) || ws == nullptr) return LTU_E_SHAPE;
  if (ws_elems < ltu_trilinear_adjoint_ws_elems(B, H, W, D, C, sd)) return LTU_E_ARG;
  const int Ho = 2 * H, Wo = 2 * W, Do = sd * D;
  if ((long long)B * Ho * Wo * D >= (1LL << 31)) return LTU_E_SHAPE;
  hipStream_t st = (hipStream_t)s;
  TriScale sc;
  sc.h = tri_scale(H, Ho); sc.w = tri_scale(W, Wo); sc.d = tri_scale(D, Do);
  sc.ih = tri_inv(H, Ho); sc.iw = tri_inv(W, Wo); sc.id = tri_inv(D, Do);
  LTU_DISPATCH_T(dtype, {
    const long long t2n = (long long)B * Ho * W * D * C;
    T* t2 = (T*)ws;                                              // [B][Ho][W][D][C]
    T* t1 = t2 + t2n;                                            // [B][Ho][Wo][D][C] (sd == 2 only)
    T* tend = sd == 2 ? t1 + 2 * t2n : t1;
    TriEntry* table = reinterpret_cast<TriEntry*>((reinterpret_cast<uintptr_t>(tend) + 127) & ~(uintptr_t)127);
    TriPair* ptable = ltu_knob("LTU_TRI_NO_PAIR", 0) ? nullptr : reinterpret_cast<TriPair*>(table + (H + W + D + 1));
    const int PH = (H + 1) / 2, PW = (W + 1) / 2, PDn = (D + 1) / 2;
    const int nb1 = (H + W + D + 127) / 128, nb2 = ptable != nullptr ? (PH + PW + PDn + 127) / 128 : 0;
    hipLaunchKernelGGL(tri_tables_kernel, dim3((unsigned)(nb1 + nb2)), dim3(128), 0, st, table, ptable, nb1, H, W, D, Ho, Wo, Do, sc);
    const T* src = (const T*)dy;
    const T* src2 = (const T*)dy2;
    if (sd == 2) {                                               // depth: [B Ho Wo][Do -> D][C]
      launch_tri_adj1d<T>(src, src2, t1, (long long)B * Ho * Wo, Do, D, C, table + H + W, ptable ? ptable + PH + PW : nullptr, st);
      src = t1; src2 = nullptr;
    }
    launch_tri_adj1d<T>(src, src2, t2, (long long)B * Ho, Wo, W, (long long)D * C, table + H, ptable ? ptable + PH : nullptr, st);       // width: [B Ho][Wo -> W][D C]
    launch_tri_adj1d<T>(t2, (const T*)nullptr, (T*)dx, B, Ho, H, (long long)W * D * C, table, ptable, st);        // height: [B][Ho -> H][W D C]
  });
  return ltu_check_launch();
}

extern "C" int ltu_trilinear_up(const void* in, const void* in2, void* out, int adjoint, int B, int H, int W, int D, int C, int sd,
                                int dtype, ltu_stream_t s) {
  if (in2 != nullptr && !adjoint) return LTU_E_ARG;      // a second (summed) input exists for gradients only
  if (C % 4 || (sd != 1 && sd != 2)) return LTU_E_SHAPE;
  const int Ho = 2 * H, Wo = 2 * W, Do = sd * D;
  if ((long long)Wo * Do * (C / 4) >= (1LL << 31) || (long long)B * Ho >= 65536) return LTU_E_SHAPE;
  TriScale sc;
  sc.h = tri_scale(H, Ho); sc.w = tri_scale(W, Wo); sc.d = tri_scale(D, Do);
  sc.ih = tri_inv(H, Ho); sc.iw = tri_inv(W, Wo); sc.id = tri_inv(D, Do);
  LTU_DISPATCH_T(dtype, {
    if (!adjoint) {
      const long long n = (long long)Wo * Do * (C / 4);
      const unsigned gx = (unsigned)((n + 511) / 512 < 1 ? 1 : (n + 511) / 512);
      hipLaunchKernelGGL((trilinear_fwd_kernel<T>), dim3(gx, B * Ho), dim3(256), 0, (hipStream_t)s, (const T*)in, (T*)out, B, H, W,
                         D, C, Ho, Wo, Do, sc);
    } else {
      const long long n = (long long)W * D * (C / 4);
      const unsigned gx = (unsigned)((n + 511) / 512 < 1 ? 1 : (n + 511) / 512);
      if (W <= TRI_TAB && D <= TRI_TAB)
        hipLaunchKernelGGL((trilinear_bwd_tab_kernel<T>), dim3(gx, B * H), dim3(256), 0, (hipStream_t)s, (const T*)in, (const T*)in2, (T*)out, B, H,
                           W, D, C, Ho, Wo, Do, sc);
      else
        hipLaunchKernelGGL((trilinear_bwd_kernel<T>), dim3(gx, B * H), dim3(256), 0, (hipStream_t)s, (const T*)in, (const T*)in2, (T*)out,
                           B, H, W, D, C, Ho, Wo, Do, sc);
    }
  });
  return ltu_check_launch();
}
