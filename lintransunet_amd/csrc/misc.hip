// Library version + the batched weight-operand preparation (one launch per step for the whole model).
#include <math.h>

#include "common.h"

extern "C" int ltu_version(void) { return 2; }

// One descriptor per prepared operand.  kinds:
//   0  cast        dst[i] = src[i]                                  R*C elements
//   1  transpose   dst[c*p0 + p1 + r] = src[r*C + c]                src [R][C]  (p0 = ld of dst, p1 = column offset)
//   2  pack_f      conv [R=Co][C=Ci][27] -> dst [p0=CoP][27][p1=CiP] zero padded
//   3  pack_d      conv [R=Co][C=Ci][27] -> dst [p1=CiP][27][p0=CoP] zero padded
//   4  copy_f32    dst[i] = src[i] as fp32 whatever the operand dtype (padded biases)      R*C elements
//   5  upconv_f    conv [R=Co][C=Ci][27] -> sub-pixel forward operand  dst [8 classes][p0=CoP][8 slots][p1=CiP]
//   6  upconv_d    conv [R=Co][C=Ci][27] -> sub-pixel data-gradient operand  dst [p1=CiP][64][p0=CoP]   (see upconv.hip)
struct WPrep {
  const float* src;
  void* dst;
  int kind, R, C, p0, p1, pad;
};

template <typename TW>
__global__ void weight_prep_kernel(const WPrep* __restrict__ table) {
  const WPrep d = table[blockIdx.y];
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  TW* dst = reinterpret_cast<TW*>(d.dst);
  if (d.kind == 0) {
    const long long n = (long long)d.R * d.C;
    for (long long i = t0; i < n; i += stride) st1<TW>(dst + i, d.src[i]);
  } else if (d.kind == 4) {
    const long long n = (long long)d.R * d.C;
    for (long long i = t0; i < n; i += stride) reinterpret_cast<float*>(d.dst)[i] = d.src[i];
  } else if (d.kind == 1) {
    const long long n = (long long)d.R * d.C;
    for (long long i = t0; i < n; i += stride) {       // i enumerates the DESTINATION (coalesced writes; sources are L2-resident)
      const int r = (int)(i % d.R), c = (int)(i / d.R);
      st1<TW>(dst + (long long)c * d.p0 + d.p1 + r, d.src[(long long)r * d.C + c]);
    }
  } else if (d.kind == 5 || d.kind == 6) {
    const int CoP = d.p0, CiP = d.p1;
    const long long n = 64LL * CoP * CiP;
    for (long long i = t0; i < n; i += stride) {
      int cls, sl, co, ci;
      if (d.kind == 5) { ci = (int)(i % CiP); sl = (int)((i / CiP) % 8); co = (int)((i / (8LL * CiP)) % CoP); cls = (int)(i / (8LL * CiP * CoP)); }
      else { co = (int)(i % CoP); const int cs = (int)((i / CoP) % 64); ci = (int)(i / (64LL * CoP)); cls = cs >> 3; sl = cs & 7; }
      float v = 0.f;
      if (co < d.R && ci < d.C) {
        // per axis: class p, slot a -> taps {0} | {1,2} (p = 0) or {0,1} | {2} (p = 1)
        int lo[3], hi[3];
        for (int a = 0; a < 3; ++a) {
          const int p = (cls >> (2 - a)) & 1, sa = (sl >> (2 - a)) & 1;
          if (p == 0) { lo[a] = sa == 0 ? 0 : 1; hi[a] = sa == 0 ? 0 : 2; }
          else { lo[a] = sa == 0 ? 0 : 2; hi[a] = sa == 0 ? 1 : 2; }
        }
        const float* w = d.src + ((long long)co * d.C + ci) * 27;
        for (int th = lo[0]; th <= hi[0]; ++th)
          for (int tw = lo[1]; tw <= hi[1]; ++tw)
            for (int td = lo[2]; td <= hi[2]; ++td) v += w[(th * 3 + tw) * 3 + td];
      }
      st1<TW>(dst + i, v);
    }
  } else {
    const int CoP = d.p0, CiP = d.p1;
    const long long n = (long long)CoP * 27 * CiP;
    for (long long i = t0; i < n; i += stride) {
      int co, ci, t;
      if (d.kind == 2) { ci = (int)(i % CiP); t = (int)((i / CiP) % 27); co = (int)(i / ((long long)CiP * 27)); }
      else { co = (int)(i % CoP); t = (int)((i / CoP) % 27); ci = (int)(i / ((long long)CoP * 27)); }
      const float v = (co < d.R && ci < d.C) ? d.src[((long long)co * d.C + ci) * 27 + t] : 0.f;
      st1<TW>(dst + i, v);
    }
  }
}

extern "C" int ltu_weight_prep(const void* table, int n, int out_dtype, ltu_stream_t s) {
  if (n <= 0) return LTU_OK;
  LTU_DISPATCH_T(out_dtype, {
    hipLaunchKernelGGL((weight_prep_kernel<T>), dim3(256, n), dim3(256), 0, (hipStream_t)s, (const WPrep*)table);
  });
  return ltu_check_launch();
}

// ------------------------------------------------------------------------------------------------ optimizer
// AdamW step of train3D.py:193 (torch.optim.AdamW, decoupled weight decay, bias-corrected moments) on flat fp32 buffers:
//   p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
// `gscale` multiplies the gradient on load (1 / accumulation count or loss-scale inverse).  One launch per bucket.
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             long long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2, float gscale) {
  const long long nv = n >> 2;
  const float step_size = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2), decay = 1.f - lr * wd;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
    float4 pv = reinterpret_cast<float4*>(p)[i], mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gg = f4at(gv, k) * gscale;
      const float mm = b1 * f4at(mv, k) + (1.f - b1) * gg;
      const float v2 = b2 * f4at(vv, k) + (1.f - b2) * gg * gg;
      f4at(mv, k) = mm; f4at(vv, k) = v2;
      f4at(pv, k) = f4at(pv, k) * decay - step_size * mm / (sqrtf(v2) * inv_sqrt_bc2 + eps);
    }
    reinterpret_cast<float4*>(p)[i] = pv; reinterpret_cast<float4*>(m)[i] = mv; reinterpret_cast<float4*>(v)[i] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {        // tail
    const long long i = (nv << 2) + threadIdx.x;
    const float gg = g[i] * gscale;
    const float mm = b1 * m[i] + (1.f - b1) * gg;
    const float v2 = b2 * v[i] + (1.f - b2) * gg * gg;
    m[i] = mm; v[i] = v2;
    p[i] = p[i] * decay - step_size * mm / (sqrtf(v2) * inv_sqrt_bc2 + eps);
  }
}

extern "C" int ltu_adamw(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                         float weight_decay, long long step, float grad_scale, ltu_stream_t s) {
  if (n <= 0) return LTU_OK;
  if (step < 1) return LTU_E_ARG;
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return LTU_E_ARG;
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  long long blocks = ((n >> 2) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, p, g, m, v, n, lr, beta1, beta2, eps,
                     weight_decay, bc1, bc2, grad_scale);
  return ltu_check_launch();
}
