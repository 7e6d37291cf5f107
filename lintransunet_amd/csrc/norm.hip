// InstanceNorm3d(+LeakyReLU +residual +dropout) and residual LayerNorm for gfx950, fwd + bwd.
// All kernels are HBM-streaming: 16-byte (fp32) / 8-byte (bf16) vector accesses along the channel
// axis of channels-last tensors, fp32 statistics, wavefront (64-lane) reductions.
//
// InstanceNorm (model/Unet_3Dblock.py:312,316,526,531,593 and 204,210,380,427): per (sample, channel)
// mean / biased variance over the S voxels, eps 1e-5, no affine.
//   stats   : sums[b][c] = { shift, sum(x-shift), sum((x-shift)^2) }   shift = x[b,0,c]  (fp32 atomics)
//   apply   : y = drop(act((x-mean)*rstd)) + res
//   bwd     : g = dy*dropmask*act'(xhat);  s1 = sum g, s2 = sum g*xhat;  dx = rstd*(g - s1/S - xhat*s2/S)
// LayerNorm (model/trans_block.py:205-206, 209-210, eps 1e-6): y = LN(x + drop(r)) * gamma + beta; the
// pre-norm sum z = x + drop(r) overwrites r (it is what the backward pass needs).
#include <stdlib.h>
#include <string.h>

#include "common.h"

#define IN_EPS 1e-5f

struct InStat {
  float mean, rstd;
};
__device__ __forceinline__ InStat in_stat(const float* sums, float invS) {
  const float m1 = sums[1] * invS;
  float var = sums[2] * invS - m1 * m1;
  var = fmaxf(var, 0.f);
  InStat s;
  s.mean = sums[0] + m1;
  s.rstd = rsqrtf(var + IN_EPS);
  return s;
}

// ---- two-stage reductions ---------------------------------------------------------------------
// Per-block partial sums used to be added with fp32 atomics: a few hundred blocks adding to the same few addresses form a
// serial chain at the memory side (~25 ns per link: 25 us for 1024 blocks).  With a workspace the blocks store their partials
// ([group][part][n]) and this kernel folds them.  mode 0: out[g*n + i];  mode 1 (InstanceNorm sums [B][C][3]):
// out[(g*n/2 + i/2)*3 + 1 + (i&1)];  mode 2 (LayerNorm): i even -> out[i/2], odd -> out2[i/2].
#define LTU_NORM_WS_FLOATS (1 << 20)
__global__ void __launch_bounds__(1024) reduce_parts_kernel(const float* __restrict__ part, int nparts, int n, float* __restrict__ out,
                                                            float* __restrict__ out2, int mode) {
  __shared__ float red[32][33];
  const int el = threadIdx.x & 31, zq = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + el, g = blockIdx.y;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (i < n) {
    const float* p = part + (long long)g * nparts * n + i;
    int z = zq;
    for (; z + 96 < nparts; z += 128) {
      a0 += p[(long long)z * n]; a1 += p[(long long)(z + 32) * n]; a2 += p[(long long)(z + 64) * n]; a3 += p[(long long)(z + 96) * n];
    }
    for (; z < nparts; z += 32) a0 += p[(long long)z * n];
  }
  red[zq][el] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (zq != 0 || i >= n) return;
  float v = 0.f;
#pragma unroll
  for (int q = 0; q < 32; ++q) v += red[q][el];
  if (mode == 0) out[(long long)g * n + i] += v;
  else if (mode == 1) out[((long long)g * (n >> 1) + (i >> 1)) * 3 + 1 + (i & 1)] += v;
  else if (i & 1) out2[i >> 1] += v;
  else out[i >> 1] += v;
}
static void launch_reduce_parts(const float* part, int nparts, int n, int groups, float* out, float* out2, int mode, hipStream_t st) {
  if (ltu_knob("LTU_DBG_NO_STAGE2", 0)) return;
  hipLaunchKernelGGL(reduce_parts_kernel, dim3(cdiv(n, 32), groups), dim3(1024), 0, st, part, nparts, n, out, out2, mode);
}
extern "C" long long ltu_norm_ws_floats(void) { return LTU_NORM_WS_FLOATS; }

// sum over the workgroup's row groups of red[g][n] at column i: 8 independent LDS reads in flight (the plain loop is a chain of up
// to 64 dependent-latency reads executed by 32 of the 256 threads at the end of every statistics workgroup)
__device__ __forceinline__ float in_fold_rowgroups(const float* red, int nrg, int n, int i) {
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int g = 0;
  for (; g + 8 <= nrg; g += 8) {
    const float* p = red + (long long)g * n + i;
    const float v0 = p[0], v1 = p[n], v2 = p[2 * n], v3 = p[3 * n], v4 = p[4 * n], v5 = p[5 * n], v6 = p[6 * n], v7 = p[7 * n];
    a0 += v0 + v4; a1 += v1 + v5; a2 += v2 + v6; a3 += v3 + v7;
  }
  for (; g < nrg; ++g) a0 += red[(long long)g * n + i];
  return (a0 + a1) + (a2 + a3);
}

// grid (nchunks, B), block 256.  x [B][S][C]; sums [B][C][3] must be zero on entry.
template <typename T>
__global__ void instnorm_stats_kernel(const T* __restrict__ x, float* __restrict__ sums, float* __restrict__ ws, long long S,
                                      int C, int rows_per_block) {
  extern __shared__ float red[];   // [rowgroups][C][2]
  const int b = blockIdx.y;
  const int cv = C / 4;                       // vectors per row
  const int tid = threadIdx.x;
  const int v = tid % cv, rg = tid / cv, nrg = blockDim.x / cv;
  const T* xb = x + (long long)b * S * C;
  float4 shift = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 a1 = shift, a2 = shift;
  if (rg < nrg) {
    shift = Vec4<T>::load(xb + v * 4);
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > S) r1 = S;
    for (long long r = r0 + rg; r < r1; r += nrg) {
      float4 t = Vec4<T>::load(xb + r * C + v * 4);
      t.x -= shift.x; t.y -= shift.y; t.z -= shift.z; t.w -= shift.w;
      a1.x += t.x; a1.y += t.y; a1.z += t.z; a1.w += t.w;
      a2.x += t.x * t.x; a2.y += t.y * t.y; a2.z += t.z * t.z; a2.w += t.w * t.w;
    }
    float* dst = red + ((long long)rg * C + v * 4) * 2;
    dst[0] = a1.x; dst[1] = a2.x; dst[2] = a1.y; dst[3] = a2.y; dst[4] = a1.z; dst[5] = a2.z; dst[6] = a1.w; dst[7] = a2.w;
  }
  __syncthreads();
  for (int i = tid; i < C * 2; i += blockDim.x) {
    const float acc = in_fold_rowgroups(red, nrg, C * 2, i);
    const int c = i >> 1, which = i & 1;
    float* s = sums + ((long long)b * C + c) * 3;
    if (ws != nullptr) ws[((long long)b * gridDim.x + blockIdx.x) * C * 2 + i] = acc;
    else atomicAdd(s + 1 + which, acc);
    if (blockIdx.x == 0 && which == 0) s[0] = ld1<T>(xb + c);
  }
}

// y = drop(act(xhat)) + res.   grid-stride over 4-vectors.
template <typename T>
__global__ void instnorm_apply_kernel(const T* __restrict__ x, const float* __restrict__ sums, const T* __restrict__ res,
                                      T* __restrict__ y, long long S, int C, int B, int act, float slope, float p,
                                      uint64_t seed, const uint64_t* step) {
  const long long nvec = (long long)B * S * C / 4;
  const float invS = 1.f / (float)S;
  const DropCfg dc = make_drop(p, seed, step);
  const long long per_b = S * C / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / per_b);
    const int c = (int)((i * 4) % C);
    float4 v = Vec4<T>::load(x + i * 4);
    float4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const InStat st = in_stat(sums + ((long long)b * C + c + k) * 3, invS);
      float h = (f4at(v, k) - st.mean) * st.rstd;
      if (act == LTU_ACT_LRELU) h = h > 0.f ? h : h * slope;
      f4at(o, k) = h;
    }
    o = drop4(dc, (uint64_t)i, o);
    if (res) {
      const float4 r = Vec4<T>::load(res + i * 4);
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    Vec4<T>::store(y + i * 4, o);
  }
}

// The same with the channel statistics hoisted out of the loop: grid (blocks, B) and a loop stride that is a multiple of the
// channel count (C | 1024), so a thread meets one channel quad only.  The kernel above recomputes 4 in_stat()s (12 loads, 4
// rsqrt) and two 64-bit divisions per 4 elements, which made it VALU-bound at 3 TB/s.
template <typename T>
__global__ void __launch_bounds__(256) instnorm_apply_fixedc_kernel(const T* __restrict__ x, const float* __restrict__ sums,
                                                                    const T* __restrict__ res, T* __restrict__ y, long long S,
                                                                    int C, int act, float slope, float p, uint64_t seed,
                                                                    const uint64_t* step) {
  const int b = blockIdx.y;
  const long long per_b = S * C / 4;
  const float invS = 1.f / (float)S;
  const DropCfg dc = make_drop(p, seed, step);
  const int c = (threadIdx.x * 4) % C;
  float mean[4], rstd[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const InStat st = in_stat(sums + ((long long)b * C + c + k) * 3, invS);
    mean[k] = st.mean; rstd[k] = st.rstd;
  }
  const long long base = (long long)b * per_b;
  for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < per_b; j += (long long)gridDim.x * 256) {
    const long long i = base + j;
    const float4 v = Vec4<T>::load(x + i * 4);
    float4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float h = (f4at(v, k) - mean[k]) * rstd[k];
      if (act == LTU_ACT_LRELU) h = h > 0.f ? h : h * slope;
      f4at(o, k) = h;
    }
    o = drop4(dc, (uint64_t)i, o);
    if (res) {
      const float4 r = Vec4<T>::load(res + i * 4);
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    Vec4<T>::store(y + i * 4, o);
  }
}

// A tensor with several consumers receives one gradient per consumer.  Instead of a stand-alone add pass (one more read-modify-
// write of an activation-sized tensor) the backward kernels of the PRODUCER take up to three gradient tensors and sum on load.
template <typename T>
__device__ __forceinline__ float4 load_grad3(const T* __restrict__ dy, const T* __restrict__ dy2, const T* __restrict__ dy3, long long e) {
  float4 g = Vec4<T>::load(dy + e);
  if (dy2 != nullptr) { const float4 t = Vec4<T>::load(dy2 + e); g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w; }
  if (dy3 != nullptr) { const float4 t = Vec4<T>::load(dy3 + e); g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w; }
  return g;
}

template <typename T>
__global__ void __launch_bounds__(256) instnorm_bwd_apply_fixedc_kernel(const T* __restrict__ dy, const T* __restrict__ dy2,
                                                                        const T* __restrict__ dy3, const T* __restrict__ x,
                                                                        const float* __restrict__ sums,
                                                                        const float* __restrict__ bsums, T* __restrict__ dx,
                                                                        long long S, int C, int act, float slope, float p,
                                                                        uint64_t seed, const uint64_t* step) {
  const int b = blockIdx.y;
  const long long per_b = S * C / 4;
  const float invS = 1.f / (float)S;
  const DropCfg dc = make_drop(p, seed, step);
  const int c = (threadIdx.x * 4) % C;
  float mean[4], rstd[4], b0[4], b1[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const InStat st = in_stat(sums + ((long long)b * C + c + k) * 3, invS);
    mean[k] = st.mean; rstd[k] = st.rstd;
    b0[k] = bsums[((long long)b * C + c + k) * 2] * invS;
    b1[k] = bsums[((long long)b * C + c + k) * 2 + 1] * invS;
  }
  const long long base = (long long)b * per_b;
  for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < per_b; j += (long long)gridDim.x * 256) {
    const long long i = base + j;
    const float4 xv = Vec4<T>::load(x + i * 4);
    const float4 g = load_grad3<T>(dy, dy2, dy3, i * 4);
    const float4 mk = dropmask4(dc, (uint64_t)i);
    float4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float h = (f4at(xv, k) - mean[k]) * rstd[k];
      float gg = f4at(g, k) * f4at(mk, k);
      if (act == LTU_ACT_LRELU && h <= 0.f) gg *= slope;
      f4at(o, k) = rstd[k] * (gg - b0[k] - h * b1[k]);
    }
    Vec4<T>::store(dx + i * 4, o);
  }
}

// backward reductions: bsums[b][c][2] += { sum g, sum g*xhat },  g = dy*mask*act'(xhat)
template <typename T>
__global__ void instnorm_bwd_stats_kernel(const T* __restrict__ dy, const T* __restrict__ dy2, const T* __restrict__ dy3,
                                          const T* __restrict__ x, const float* __restrict__ sums,
                                          float* __restrict__ bsums, float* __restrict__ ws, long long S, int C,
                                          int rows_per_block, int act, float slope, float p, uint64_t seed, const uint64_t* step) {
  extern __shared__ float red[];
  const int b = blockIdx.y;
  const int cv = C / 4;
  const int tid = threadIdx.x;
  const int v = tid % cv, rg = tid / cv, nrg = blockDim.x / cv;
  const float invS = 1.f / (float)S;
  const DropCfg dc = make_drop(p, seed, step);
  float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f), a2 = a1;
  if (rg < nrg) {
    InStat st[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) st[k] = in_stat(sums + ((long long)b * C + v * 4 + k) * 3, invS);
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > S) r1 = S;
    for (long long r = r0 + rg; r < r1; r += nrg) {
      const long long e = ((long long)b * S + r) * C + v * 4;
      const float4 xv = Vec4<T>::load(x + e);
      float4 g = load_grad3<T>(dy, dy2, dy3, e);
      const float4 mk = dropmask4(dc, (uint64_t)(e >> 2));
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float h = (f4at(xv, k) - st[k].mean) * st[k].rstd;
        float gg = f4at(g, k) * f4at(mk, k);
        if (act == LTU_ACT_LRELU && h <= 0.f) gg *= slope;
        f4at(a1, k) += gg;
        f4at(a2, k) += gg * h;
      }
    }
    float* dst = red + ((long long)rg * C + v * 4) * 2;
    dst[0] = a1.x; dst[1] = a2.x; dst[2] = a1.y; dst[3] = a2.y; dst[4] = a1.z; dst[5] = a2.z; dst[6] = a1.w; dst[7] = a2.w;
  }
  __syncthreads();
  for (int i = tid; i < C * 2; i += blockDim.x) {
    const float acc = in_fold_rowgroups(red, nrg, C * 2, i);
    if (ws != nullptr) ws[((long long)b * gridDim.x + blockIdx.x) * C * 2 + i] = acc;
    else atomicAdd(bsums + (long long)b * C * 2 + i, acc);
  }
}

template <typename T>
__global__ void instnorm_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ dy2, const T* __restrict__ dy3,
                                          const T* __restrict__ x, const float* __restrict__ sums,
                                          const float* __restrict__ bsums, T* __restrict__ dx, long long S, int C, int B,
                                          int act, float slope, float p, uint64_t seed, const uint64_t* step) {
  const long long nvec = (long long)B * S * C / 4;
  const float invS = 1.f / (float)S;
  const DropCfg dc = make_drop(p, seed, step);
  const long long per_b = S * C / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / per_b);
    const int c = (int)((i * 4) % C);
    const float4 xv = Vec4<T>::load(x + i * 4);
    const float4 g = load_grad3<T>(dy, dy2, dy3, i * 4);
    const float4 mk = dropmask4(dc, (uint64_t)i);
    float4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const InStat st = in_stat(sums + ((long long)b * C + c + k) * 3, invS);
      const float* bs = bsums + ((long long)b * C + c + k) * 2;
      const float h = (f4at(xv, k) - st.mean) * st.rstd;
      float gg = f4at(g, k) * f4at(mk, k);
      if (act == LTU_ACT_LRELU && h <= 0.f) gg *= slope;
      f4at(o, k) = st.rstd * (gg - bs[0] * invS - h * bs[1] * invS);
    }
    Vec4<T>::store(dx + i * 4, o);
  }
}

// ------------------------------------------------------------------------------------------------ LayerNorm
// One row (d <= 256 channels) per group of d/4 lanes; groups never straddle a wavefront.
// z = x + drop(r) is written over r; y = (z-mean)*rstd*gamma + beta; stat[row] = {mean, rstd}
// V = 4-element vectors per lane (d = 4 V G): V = 2 from d = 128 up means 16-byte accesses for bf16 and half the shuffle
// steps; a workgroup walks `iters` row groups so that a launch is a few thousand workgroups at most.
template <typename T, int G, int V>
__global__ void __launch_bounds__(256) layernorm_fwd_kernel(const T* __restrict__ x, T* __restrict__ r,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            T* __restrict__ y, float* __restrict__ stat, long long M, float eps,
                                                            float p, uint64_t seed, const uint64_t* step, int iters) {
  constexpr int d = G * 4 * V, E = 4 * V;
  const int gl = threadIdx.x % G;
  const int rows_per_block = 256 / G;
  const DropCfg dc = make_drop(p, seed, step);
  float gm[E], bt[E];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const float4 a = *reinterpret_cast<const float4*>(gamma + gl * E + 4 * v), c = *reinterpret_cast<const float4*>(beta + gl * E + 4 * v);
    gm[4 * v] = a.x; gm[4 * v + 1] = a.y; gm[4 * v + 2] = a.z; gm[4 * v + 3] = a.w;
    bt[4 * v] = c.x; bt[4 * v + 1] = c.y; bt[4 * v + 2] = c.z; bt[4 * v + 3] = c.w;
  }
  for (int it = 0; it < iters; ++it) {
    const long long row = ((long long)blockIdx.x * iters + it) * rows_per_block + threadIdx.x / G;
    const bool ok = row < M;
    float z[E];
#pragma unroll
    for (int k = 0; k < E; ++k) z[k] = 0.f;
    const long long e = row * d + gl * E;
    if (ok) {
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const float4 xv = Vec4<T>::load(x + e + 4 * v);
        float4 rv = Vec4<T>::load(r + e + 4 * v);
        rv = drop4(dc, (uint64_t)((e >> 2) + v), rv);
        z[4 * v] = xv.x + rv.x; z[4 * v + 1] = xv.y + rv.y; z[4 * v + 2] = xv.z + rv.z; z[4 * v + 3] = xv.w + rv.w;
        Vec4<T>::store(r + e + 4 * v, make_float4(z[4 * v], z[4 * v + 1], z[4 * v + 2], z[4 * v + 3]));
      }
    }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < E; ++k) sum += z[k];
    const float mean = group_sum<G>(sum) / (float)d;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < E; ++k) { z[k] -= mean; sq += z[k] * z[k]; }
    const float var = group_sum<G>(sq) / (float)d;
    const float rstd = rsqrtf(var + eps);
    if (ok) {
#pragma unroll
      for (int v = 0; v < V; ++v)
        Vec4<T>::store(y + e + 4 * v, make_float4(z[4 * v] * rstd * gm[4 * v] + bt[4 * v], z[4 * v + 1] * rstd * gm[4 * v + 1] + bt[4 * v + 1],
                                                   z[4 * v + 2] * rstd * gm[4 * v + 2] + bt[4 * v + 2],
                                                   z[4 * v + 3] * rstd * gm[4 * v + 3] + bt[4 * v + 3]));
      if (gl == 0) {
        stat[row * 2] = mean;
        stat[row * 2 + 1] = rstd;
      }
    }
  }
}

// dz = rstd*(g*gamma - mean_d(g*gamma) - xhat*mean_d(g*gamma*xhat));  dr = dz*dropmask;
// dgamma += sum_rows g*xhat, dbeta += sum_rows g   (block partials through LDS, then fp32 atomics)
// V = 4-element vectors per lane (a lane owns 4V consecutive elements of the row): V = 2 halves the cross-lane reduction
// steps and the instruction count per byte for d >= 128.
template <typename T, int V>
__device__ __forceinline__ void ln_load(const T* p, float (&o)[4 * V]) {
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const float4 t = Vec4<T>::load(p + 4 * v);
    o[4 * v] = t.x; o[4 * v + 1] = t.y; o[4 * v + 2] = t.z; o[4 * v + 3] = t.w;
  }
}
template <typename T, int G, int V>
__global__ void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ dy2, const T* __restrict__ z, const float* __restrict__ stat,
                                     const float* __restrict__ gamma, T* __restrict__ dz, T* __restrict__ dr,
                                     float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ ws, long long M,
                                     int rows_per_block, float p, uint64_t seed, const uint64_t* step) {
  extern __shared__ float red[];   // [rowgroups][d][2]
  constexpr int E = 4 * V, d = G * E;
  const int gl = threadIdx.x % G, rg = threadIdx.x / G, nrg = blockDim.x / G;
  const DropCfg dc = make_drop(p, seed, step);
  float gm[E], ag[E], ab[E];
#pragma unroll
  for (int k = 0; k < E; ++k) { gm[k] = gamma[gl * E + k]; ag[k] = 0.f; ab[k] = 0.f; }
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  long long r1 = r0 + rows_per_block;
  if (r1 > M) r1 = M;
  // software pipeline: the next row's operands are in flight while this row goes through its two cross-lane reductions
  float gn[E], zn[E];
  float2 stn = make_float2(0.f, 0.f);
  auto fetch = [&](long long row) {
#pragma unroll
    for (int k = 0; k < E; ++k) { gn[k] = 0.f; zn[k] = 0.f; }
    if (row < r1) {
      ln_load<T, V>(dy + row * d + gl * E, gn);
      if (dy2 != nullptr) {
        float t2[E];
        ln_load<T, V>(dy2 + row * d + gl * E, t2);
#pragma unroll
        for (int k = 0; k < E; ++k) gn[k] += t2[k];
      }
      ln_load<T, V>(z + row * d + gl * E, zn);
      stn = *reinterpret_cast<const float2*>(stat + row * 2);
    }
  };
  fetch(r0 + rg);
  for (long long rb = r0; rb < r1; rb += nrg) {
    const long long row = rb + rg;
    const bool ok = row < r1;
    float g[E], h[E];
    const float mean = stn.x, rstd = ok ? stn.y : 0.f;
#pragma unroll
    for (int k = 0; k < E; ++k) { g[k] = gn[k]; h[k] = ok ? (zn[k] - mean) * rstd : 0.f; }
    fetch(row + nrg);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < E; ++k) {
      if (ok) { ab[k] += g[k]; ag[k] += g[k] * h[k]; }
      g[k] *= gm[k];                       // g * gamma from here on
      s1 += g[k]; s2 += g[k] * h[k];
    }
    const float m1 = group_sum<G>(s1) / (float)d;
    const float m2 = group_sum<G>(s2) / (float)d;
    if (ok) {
      const long long e = row * d + gl * E;
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const float4 o = make_float4(rstd * (g[4 * v] - m1 - h[4 * v] * m2), rstd * (g[4 * v + 1] - m1 - h[4 * v + 1] * m2),
                                     rstd * (g[4 * v + 2] - m1 - h[4 * v + 2] * m2), rstd * (g[4 * v + 3] - m1 - h[4 * v + 3] * m2));
        Vec4<T>::store(dz + e + 4 * v, o);
        if (dr != dz) Vec4<T>::store(dr + e + 4 * v, drop4(dc, (uint64_t)((e + 4 * v) >> 2), o));
      }
    }
  }
  float* dst = red + ((long long)rg * d + gl * E) * 2;
#pragma unroll
  for (int k = 0; k < E; ++k) { dst[2 * k] = ag[k]; dst[2 * k + 1] = ab[k]; }
  __syncthreads();
  for (int i = threadIdx.x; i < d * 2; i += blockDim.x) {
    float acc = 0.f;
    for (int g = 0; g < nrg; ++g) acc += red[(long long)g * d * 2 + i];
    if (ws != nullptr) ws[(long long)blockIdx.x * d * 2 + i] = acc;
    else atomicAdd(((i & 1) ? dbeta : dgamma) + (i >> 1), acc);
  }
}

// ------------------------------------------------------------------------------------------------ host side
static int stats_rows(long long S, int B, int* nchunks) {
  int chunks = -1;
  chunks = ltu_knob_pos("LTU_IN_CHUNKS", 2048);
  long long want = chunks / (B > 0 ? B : 1);
  if (want < 1) want = 1;
  long long rows = (S + want - 1) / want;
  if (rows < 64) rows = 64;
  *nchunks = (int)((S + rows - 1) / rows);
  return (int)rows;
}

extern "C" int ltu_instnorm_stats(const void* x, float* sums, float* ws, int B, long long S, int C, int dtype, ltu_stream_t s) {
  if (C % 4 != 0 || C > 1024 || 256 % (C / 4) != 0) return LTU_E_SHAPE;
  int nchunks;
  const int rows = stats_rows(S, B, &nchunks);
  const int block = 256;
  const int nrg = block / (C / 4);
  if (nrg < 1) return LTU_E_SHAPE;
  const size_t lds = (size_t)nrg * C * 2 * sizeof(float);
  LTU_DISPATCH_T(dtype, {
    if ((long long)nchunks * B * C * 2 > LTU_NORM_WS_FLOATS) ws = nullptr;
    hipLaunchKernelGGL((instnorm_stats_kernel<T>), dim3(nchunks, B), dim3(block), lds, (hipStream_t)s, (const T*)x, sums, ws, S, C, rows);
    if (ws != nullptr) launch_reduce_parts(ws, nchunks, C * 2, B, sums, nullptr, 1, (hipStream_t)s);
  });
  return ltu_check_launch();
}

static unsigned stream_grid(long long nvec) {
  long long blocks = (nvec + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

static unsigned per_sample_grid(long long nvec_per_sample, int B) {
  long long blocks = (nvec_per_sample + 255) / 256;
  const long long cap = 4096 / (B > 0 ? B : 1) > 1 ? 4096 / (B > 0 ? B : 1) : 1;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

extern "C" int ltu_instnorm_apply(const void* x, const float* sums, const void* res, void* y, int B, long long S, int C,
                                  int act, float slope, float p, uint64_t seed, const uint64_t* step, int dtype, ltu_stream_t s) {
  if (C % 4 != 0) return LTU_E_SHAPE;
  const long long nvec = (long long)B * S * C / 4;
  const bool fixedc = 1024 % C == 0;           // the loop stride (256 vectors) is then a multiple of the channel count
  LTU_DISPATCH_T(dtype, {
    if (fixedc)
      hipLaunchKernelGGL((instnorm_apply_fixedc_kernel<T>), dim3(per_sample_grid(S * C / 4, B), B), dim3(256), 0, (hipStream_t)s,
                         (const T*)x, sums, (const T*)res, (T*)y, S, C, act, slope, p, seed, step);
    else
      hipLaunchKernelGGL((instnorm_apply_kernel<T>), dim3(stream_grid(nvec)), dim3(256), 0, (hipStream_t)s, (const T*)x, sums,
                         (const T*)res, (T*)y, S, C, B, act, slope, p, seed, step);
  });
  return ltu_check_launch();
}

extern "C" int ltu_instnorm_bwd(const void* dy, const void* dy2, const void* dy3, const void* x, const float* sums, float* bsums,
                                float* ws, void* dx, int B,
                                long long S, int C, int act, float slope, float p, uint64_t seed, const uint64_t* step, int dtype,
                                ltu_stream_t s) {
  if (C % 4 != 0 || C > 1024 || 256 % (C / 4) != 0) return LTU_E_SHAPE;
  int nchunks;
  const int rows = stats_rows(S, B, &nchunks);
  const int block = 256;
  const int nrg = block / (C / 4);
  if (nrg < 1) return LTU_E_SHAPE;
  const size_t lds = (size_t)nrg * C * 2 * sizeof(float);
  const long long nvec = (long long)B * S * C / 4;
  LTU_DISPATCH_T(dtype, {
    if ((long long)nchunks * B * C * 2 > LTU_NORM_WS_FLOATS) ws = nullptr;
    hipLaunchKernelGGL((instnorm_bwd_stats_kernel<T>), dim3(nchunks, B), dim3(block), lds, (hipStream_t)s, (const T*)dy,
                       (const T*)dy2, (const T*)dy3, (const T*)x, sums, bsums, ws, S, C, rows, act, slope, p, seed, step);
    if (ws != nullptr) launch_reduce_parts(ws, nchunks, C * 2, B, bsums, nullptr, 0, (hipStream_t)s);
    if (1024 % C == 0)
      hipLaunchKernelGGL((instnorm_bwd_apply_fixedc_kernel<T>), dim3(per_sample_grid(S * C / 4, B), B), dim3(256), 0,
                         (hipStream_t)s, (const T*)dy, (const T*)dy2, (const T*)dy3, (const T*)x, sums, bsums, (T*)dx, S, C, act, slope,
                         p, seed, step);
    else
      hipLaunchKernelGGL((instnorm_bwd_apply_kernel<T>), dim3(stream_grid(nvec)), dim3(256), 0, (hipStream_t)s, (const T*)dy,
                         (const T*)dy2, (const T*)dy3, (const T*)x, sums, bsums, (T*)dx, S, C, B, act, slope, p, seed, step);
  });
  return ltu_check_launch();
}

#define LN_DISPATCH_G(d, ...)                                \
  do {                                                       \
    if ((d) == 32) { constexpr int G = 8; __VA_ARGS__ }      \
    else if ((d) == 64) { constexpr int G = 16; __VA_ARGS__ } \
    else if ((d) == 128) { constexpr int G = 32; __VA_ARGS__ } \
    else if ((d) == 256) { constexpr int G = 64; __VA_ARGS__ } \
    else return LTU_E_SHAPE;                                 \
  } while (0)

// backward: 8 elements per lane from d = 128 up
#define LN_DISPATCH_GV(d, ...)                                                  \
  do {                                                                          \
    if ((d) == 32) { constexpr int G = 8, V = 1; __VA_ARGS__ }                  \
    else if ((d) == 64) { constexpr int G = 16, V = 1; __VA_ARGS__ }            \
    else if ((d) == 128) { constexpr int G = 16, V = 2; __VA_ARGS__ }           \
    else if ((d) == 256) { constexpr int G = 32, V = 2; __VA_ARGS__ }           \
    else return LTU_E_SHAPE;                                                    \
  } while (0)

extern "C" int ltu_layernorm_fwd(const void* x, void* r, const float* gamma, const float* beta, void* y, float* stat,
                                 long long M, int d, float eps, float p, uint64_t seed, const uint64_t* step, int dtype, ltu_stream_t s) {
  LTU_DISPATCH_T(dtype, {
    LN_DISPATCH_GV(d, {
      const int rows = 256 / G;
      const long long groups = (M + rows - 1) / rows;
      const int iters = (int)((groups + 4095) / 4096 < 1 ? 1 : (groups + 4095) / 4096);
      hipLaunchKernelGGL((layernorm_fwd_kernel<T, G, V>), dim3((unsigned)((groups + iters - 1) / iters)), dim3(256), 0, (hipStream_t)s,
                         (const T*)x, (T*)r, gamma, beta, (T*)y, stat, M, eps, p, seed, step, iters);
    });
  });
  return ltu_check_launch();
}

extern "C" int ltu_layernorm_bwd(const void* dy, const void* dy2, const void* z, const float* stat, const float* gamma, void* dz, void* dr,
                                 float* dgamma, float* dbeta, float* ws, ltu_reduce_job* defer, long long M, int d, float p,
                                 uint64_t seed, const uint64_t* step, int dtype, ltu_stream_t s) {
  if (defer != nullptr) defer->part = nullptr;
  LTU_DISPATCH_T(dtype, {
    LN_DISPATCH_GV(d, {
      const int nrg = 256 / G;
      long long rows = (M + 1023) / 1024;        // 1024 workgroups (swept 512 / 1024 / 2048; the callers size the workspace for <= 2048)
      if (rows < nrg) rows = nrg;
      rows = (rows + nrg - 1) / nrg * nrg;
      const size_t lds = (size_t)nrg * d * 2 * sizeof(float);
      const int nblk = cdiv(M, rows);
      if ((long long)nblk * d * 2 > LTU_NORM_WS_FLOATS) ws = nullptr;
      hipLaunchKernelGGL((layernorm_bwd_kernel<T, G, V>), dim3(nblk), dim3(256), lds, (hipStream_t)s, (const T*)dy, (const T*)dy2,
                         (const T*)z, stat, gamma, (T*)dz, (T*)dr, dgamma, dbeta, ws, M, (int)rows, p, seed, step);
      if (ws != nullptr && defer != nullptr) {
        defer->part = ws; defer->nsplit = nblk; defer->n = d * 2; defer->k = 1; defer->nseg = 1; defer->mode = 1;
        defer->out[0] = dgamma; defer->out[1] = dbeta; defer->out[2] = nullptr;
        defer->outb[0] = defer->outb[1] = defer->outb[2] = nullptr;
      } else if (ws != nullptr) {
        launch_reduce_parts(ws, nblk, d * 2, 1, dgamma, dbeta, 2, (hipStream_t)s);
      }
    });
  });
  return ltu_check_launch();
}

// ------------------------------------------------------------------------------------------------ deferred second stages
// One launch folds up to 8 pending two-stage reductions (grid.y = job).  A workgroup owns 32 consecutive output elements;
// its 32 thread groups each sum every 32nd split (coalesced 128-byte reads, 4 loads in flight), then combine through LDS.
struct ReduceBatch {
  ltu_reduce_job j[8];
};
__global__ void __launch_bounds__(1024) reduce_batch_kernel(const ReduceBatch b) {
  __shared__ float red[32][33];
  const ltu_reduce_job& jb = b.j[blockIdx.y];
  const long long nk = (long long)jb.n * jb.k;
  const long long total = jb.mode == 0 ? nk + jb.n : jb.n;
  const int el = threadIdx.x & 31, zq = threadIdx.x >> 5;
  const long long e = (long long)blockIdx.x * 32 + el;
  if ((long long)blockIdx.x * 32 >= total) return;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (e < total) {
    const float* p;
    long long zs;
    if (jb.mode == 0 && e >= nk) { p = jb.part + (long long)jb.nsplit * nk + (e - nk); zs = jb.n; }
    else { p = jb.part + e; zs = jb.mode == 0 ? nk : jb.n; }
    int z = zq;
    for (; z + 96 < jb.nsplit; z += 128) { a0 += p[z * zs]; a1 += p[(z + 32) * zs]; a2 += p[(z + 64) * zs]; a3 += p[(z + 96) * zs]; }
    for (; z < jb.nsplit; z += 32) a0 += p[z * zs];
  }
  red[zq][el] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (zq != 0 || e >= total) return;
  float v = 0.f;
#pragma unroll
  for (int q = 0; q < 32; ++q) v += red[q][el];
  if (jb.mode == 1) {
    float* o = (e & 1) ? jb.out[1] : jb.out[0];
    o[e >> 1] += v;
    return;
  }
  const int nper = jb.n / jb.nseg;
  if (e < nk) {
    const int n = (int)(e / jb.k), k = (int)(e - (long long)n * jb.k);
    const int seg = n / nper;
    float* o = seg == 0 ? jb.out[0] : (seg == 1 ? jb.out[1] : jb.out[2]);
    o[(long long)(n - seg * nper) * jb.k + k] += v;
  } else {
    const int n = (int)(e - nk);
    const int seg = n / nper;
    float* o = seg == 0 ? jb.outb[0] : (seg == 1 ? jb.outb[1] : jb.outb[2]);
    if (o != nullptr) o[n - seg * nper] += v;
  }
}

extern "C" int ltu_reduce_batch(const ltu_reduce_job* jobs, int njobs, ltu_stream_t s) {
  for (int i0 = 0; i0 < njobs; i0 += 8) {
    ReduceBatch b;
    memset(&b, 0, sizeof(b));
    int cnt = 0;
    long long maxblk = 0;
    for (int i = i0; i < njobs && cnt < 8; ++i) {
      const ltu_reduce_job& j = jobs[i];
      if (j.part == nullptr) continue;
      if (j.nseg < 1 || j.nseg > 3 || j.n % j.nseg) return LTU_E_ARG;
      b.j[cnt++] = j;
      const long long total = j.mode == 0 ? (long long)j.n * j.k + j.n : j.n;
      const long long nb = (total + 31) / 32;
      if (nb > maxblk) maxblk = nb;
    }
    if (cnt == 0) continue;
    hipLaunchKernelGGL(reduce_batch_kernel, dim3((unsigned)maxblk, cnt), dim3(1024), 0, (hipStream_t)s, b);
  }
  return ltu_check_launch();
}
