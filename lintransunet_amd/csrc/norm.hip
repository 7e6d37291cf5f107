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
// NB = loads per thread, all requested before the first add: as a loop of 4 loads per trip a fold over 1 024 partials was eight
// dependent L2 round trips - most of the 5 us of a launch that moves a few KB (54 of them per training step).
template <int NB>
__global__ void __launch_bounds__(1024) reduce_parts_kernel(const float* __restrict__ part, int nparts, int n, float* __restrict__ out,
                                                            float* __restrict__ out2, int mode) {
  __shared__ float red[32][33];
  const int el = threadIdx.x & 31, zq = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + el, g = blockIdx.y;
  float acc = 0.f;
  if (i < n) {
    const float* p = part + (long long)g * nparts * n + i;
    for (int z0 = zq; z0 < nparts; z0 += 32 * NB) {         // one trip unless nparts > 32 NB
      float v[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int z = z0 + 32 * k;
        v[k] = p[(long long)(z < nparts ? z : z0) * n];
      }
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const float t = z0 + 32 * k < nparts ? v[k] : 0.f;
        if ((k & 3) == 0) a0 += t; else if ((k & 3) == 1) a1 += t; else if ((k & 3) == 2) a2 += t; else a3 += t;
      }
      acc += (a0 + a1) + (a2 + a3);
    }
  }
  red[zq][el] = acc;
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
  const dim3 grid(cdiv(n, 32), groups);
  if (nparts <= 128) hipLaunchKernelGGL(reduce_parts_kernel<4>, grid, dim3(1024), 0, st, part, nparts, n, out, out2, mode);
  else if (nparts <= 256) hipLaunchKernelGGL(reduce_parts_kernel<8>, grid, dim3(1024), 0, st, part, nparts, n, out, out2, mode);
  else if (nparts <= 512) hipLaunchKernelGGL(reduce_parts_kernel<16>, grid, dim3(1024), 0, st, part, nparts, n, out, out2, mode);
  else hipLaunchKernelGGL(reduce_parts_kernel<32>, grid, dim3(1024), 0, st, part, nparts, n, out, out2, mode);
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

// ---- VW-wide vector access: 4 elements (8 / 16 bytes) or 8 elements (16 bytes of bf16: twice the bytes in flight per thread;
// the statistics and apply passes keep one or two loads per thread in flight and are latency-, not bandwidth-limited) ----------
template <typename T, int VW>
__device__ __forceinline__ void ldv(const T* p, float (&f)[VW]) {
  if constexpr (VW == 8 && sizeof(T) == 2) {
    const uint4 r = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  } else {
#pragma unroll
    for (int q = 0; q < VW / 4; ++q) {
      const float4 t = Vec4<T>::load(p + 4 * q);
      f[4 * q] = t.x; f[4 * q + 1] = t.y; f[4 * q + 2] = t.z; f[4 * q + 3] = t.w;
    }
  }
}
template <typename T, int VW>
__device__ __forceinline__ void stv(T* p, const float (&f)[VW]) {
  if constexpr (VW == 8 && sizeof(T) == 2) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
  } else {
#pragma unroll
    for (int q = 0; q < VW / 4; ++q) Vec4<T>::store(p + 4 * q, make_float4(f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]));
  }
}
// keep-mask * scale of the VW elements starting at element index e (a multiple of VW): the groups of four of drop4
template <int VW>
__device__ __forceinline__ void dropmaskv(const DropCfg& dc, long long e, float (&m)[VW]) {
#pragma unroll
  for (int q = 0; q < VW / 4; ++q) {
    const float4 t = dropmask4(dc, (uint64_t)((e >> 2) + q));
    m[4 * q] = t.x; m[4 * q + 1] = t.y; m[4 * q + 2] = t.z; m[4 * q + 3] = t.w;
  }
}
template <typename T, int VW>
__device__ __forceinline__ void load_gradv(const T* __restrict__ dy, const T* __restrict__ dy2, const T* __restrict__ dy3, long long e, float (&g)[VW]) {
  ldv<T, VW>(dy + e, g);
  if (dy2 != nullptr) {
    float t[VW];
    ldv<T, VW>(dy2 + e, t);
#pragma unroll
    for (int k = 0; k < VW; ++k) g[k] += t[k];
  }
  if (dy3 != nullptr) {
    float t[VW];
    ldv<T, VW>(dy3 + e, t);
#pragma unroll
    for (int k = 0; k < VW; ++k) g[k] += t[k];
  }
}

// ---- second stage inside the consumer ----------------------------------------------------------------------------------------
// The statistics kernels leave per-chunk partial sums [nparts][n] (n = 2 C); instead of a fold launch of their own (5 us for a few
// KB, 46 of them per training step) every workgroup of the APPLY kernel that follows folds them itself - same code, same order, so
// all workgroups hold bit-identical statistics - and workgroup 0 of each sample publishes them for the backward pass.  Bounded by
// IN_FOLD_MAX floats per sample so that the redundant reads stay a few KB per workgroup (the chunk count is chosen accordingly).
#define IN_FOLD_MAX 8192
#define IN_FOLD_LDS 1024
// lds[0 .. n) = sum over parts; n a power of two in [4, IN_FOLD_LDS], nparts * n <= IN_FOLD_MAX; all 256 threads call this; ends
// with a barrier.  A thread owns one 4-column group and every G-th part: at most IN_FOLD_MAX / 1024 = 8 float4 loads, ALL requested
// before the first add (as a loop of dependent L2 round trips this prologue cost more than the fold launch it replaces).
__device__ __forceinline__ void in_fold_parts(const float* __restrict__ part, int nparts, int n, float* lds, int tid) {
  const int W = n >> 2, G = 256 / W;            // W column quads (<= 256), G part groups
  const int cq = tid % W, g = tid / W;
  constexpr int NL = IN_FOLD_MAX / 1024;
  float4 v[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const int z = g + k * G;
    v[k] = *reinterpret_cast<const float4*>(part + (long long)(z < nparts ? z : 0) * n + cq * 4);
  }
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < NL; ++k)
    if (g + k * G < nparts) { a.x += v[k].x; a.y += v[k].y; a.z += v[k].z; a.w += v[k].w; }
  // groups of a wave that share cq sit W lanes apart: fold them across lanes first when W < 64, then across waves through LDS
  if (W < 64) {
    for (int o = W; o < 64; o <<= 1) {
      a.x += __shfl_xor(a.x, o); a.y += __shfl_xor(a.y, o); a.z += __shfl_xor(a.z, o); a.w += __shfl_xor(a.w, o);
    }
  }
  const int lane = tid & 63, wave = tid >> 6;
  const int rows = W < 64 ? 4 : G;              // partial rows left: one per wave, or one per group
  if (W < 64) { if (lane < W) *reinterpret_cast<float4*>(lds + wave * n + lane * 4) = a; }
  else *reinterpret_cast<float4*>(lds + g * n + cq * 4) = a;
  __syncthreads();
  float r = 0.f;
  if (tid < n) {                                 // n <= 1024: up to 4 columns per thread
    for (int q = 0; q < rows; ++q) r += lds[q * n + tid];
  }
  float r1 = 0.f, r2 = 0.f, r3 = 0.f;
  if (n > 256) {
    for (int q = 0; q < rows; ++q) {
      if (tid + 256 < n) r1 += lds[q * n + tid + 256];
      if (tid + 512 < n) r2 += lds[q * n + tid + 512];
      if (tid + 768 < n) r3 += lds[q * n + tid + 768];
    }
  }
  __syncthreads();
  if (tid < n) lds[tid] = r;
  if (n > 256) {
    if (tid + 256 < n) lds[tid + 256] = r1;
    if (tid + 512 < n) lds[tid + 512] = r2;
    if (tid + 768 < n) lds[tid + 768] = r3;
  }
  __syncthreads();
}

// grid (nchunks, B), block 256.  x [B][S][C]; sums [B][C][3] must be zero on entry.
template <typename T, int VW>
__global__ void __launch_bounds__(256) instnorm_stats_kernel(const T* __restrict__ x, float* __restrict__ sums, float* __restrict__ ws, long long S,
                                                              int C, int rows_per_block) {
  extern __shared__ float red[];   // [rowgroups][C][2]
  const int b = blockIdx.y;
  const int cv = C / VW;                      // vectors per row
  const int tid = threadIdx.x;
  const int v = tid % cv, rg = tid / cv, nrg = blockDim.x / cv;
  const T* xb = x + (long long)b * S * C;
  float shift[VW], a1[VW], a2[VW];
#pragma unroll
  for (int k = 0; k < VW; ++k) { shift[k] = 0.f; a1[k] = 0.f; a2[k] = 0.f; }
  if (rg < nrg) {
    ldv<T, VW>(xb + v * VW, shift);
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > S) r1 = S;
    long long r = r0 + rg;
    for (; r + nrg < r1; r += 2 * nrg) {       // two rows per step: both loads in flight
      float t[VW], u[VW];
      ldv<T, VW>(xb + r * C + v * VW, t);
      ldv<T, VW>(xb + (r + nrg) * C + v * VW, u);
#pragma unroll
      for (int k = 0; k < VW; ++k) {
        t[k] -= shift[k]; u[k] -= shift[k];
        a1[k] += t[k] + u[k];
        a2[k] += t[k] * t[k] + u[k] * u[k];
      }
    }
    if (r < r1) {
      float t[VW];
      ldv<T, VW>(xb + r * C + v * VW, t);
#pragma unroll
      for (int k = 0; k < VW; ++k) { t[k] -= shift[k]; a1[k] += t[k]; a2[k] += t[k] * t[k]; }
    }
    float* dst = red + ((long long)rg * C + v * VW) * 2;
#pragma unroll
    for (int k = 0; k < VW; ++k) { dst[2 * k] = a1[k]; dst[2 * k + 1] = a2[k]; }
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
// channel count (C | 256 VW), so a thread meets one channel vector only.  The kernel above recomputes 4 in_stat()s (12 loads, 4
// rsqrt) and two 64-bit divisions per 4 elements, which made it VALU-bound at 3 TB/s.
// FOLD: `sums` holds only the shifts; the sums themselves are folded here from the statistics kernel's partials `ws`
// [B][nchunks][2 C] and published to `sums` by workgroup 0 of each sample (see in_fold_parts).
template <typename T, int VW, bool FOLD>
__global__ void __launch_bounds__(256) instnorm_apply_fixedc_kernel(const T* __restrict__ x, float* __restrict__ sums,
                                                                    const T* __restrict__ res, T* __restrict__ y, long long S,
                                                                    int C, int act, float slope, float p, uint64_t seed,
                                                                    const uint64_t* step, const float* __restrict__ ws, int nchunks) {
  __shared__ float fold[FOLD ? IN_FOLD_LDS : 1];
  const int b = blockIdx.y;
  const long long per_b = S * C / VW;
  const float invS = 1.f / (float)S;
  const DropCfg dc = make_drop(p, seed, step);
  const int c = (threadIdx.x * VW) % C;
  float mean[VW], rstd[VW];
  if constexpr (FOLD) {
    in_fold_parts(ws + (long long)b * nchunks * C * 2, nchunks, C * 2, fold, threadIdx.x);
#pragma unroll
    for (int k = 0; k < VW; ++k) {
      const float m1 = fold[(c + k) * 2] * invS;
      const float var = fmaxf(fold[(c + k) * 2 + 1] * invS - m1 * m1, 0.f);
      mean[k] = sums[((long long)b * C + c + k) * 3] + m1;
      rstd[k] = rsqrtf(var + IN_EPS);
    }
    if (blockIdx.x == 0)
      for (int i = threadIdx.x; i < 2 * C; i += 256) sums[((long long)b * C + (i >> 1)) * 3 + 1 + (i & 1)] = fold[i];
  } else {
#pragma unroll
    for (int k = 0; k < VW; ++k) {
      const InStat st = in_stat(sums + ((long long)b * C + c + k) * 3, invS);
      mean[k] = st.mean; rstd[k] = st.rstd;
    }
  }
  const long long base = (long long)b * per_b;
  const long long stride = (long long)gridDim.x * 256;
  auto one = [&](long long i, const float (&v)[VW], const float (&r)[VW]) {
    float o[VW], mk[VW];
    dropmaskv<VW>(dc, i * VW, mk);
#pragma unroll
    for (int k = 0; k < VW; ++k) {
      float h = (v[k] - mean[k]) * rstd[k];
      if (act == LTU_ACT_LRELU) h = h > 0.f ? h : h * slope;
      o[k] = h * mk[k] + r[k];
    }
    stv<T, VW>(y + i * VW, o);
  };
  long long j = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; j + stride < per_b; j += 2 * stride) {          // two vectors per step: their loads are in flight together
    const long long i0 = base + j, i1 = i0 + stride;
    float v0[VW], v1[VW], r0[VW], r1[VW];
    ldv<T, VW>(x + i0 * VW, v0);
    ldv<T, VW>(x + i1 * VW, v1);
    if (res) { ldv<T, VW>(res + i0 * VW, r0); ldv<T, VW>(res + i1 * VW, r1); }
    else {
#pragma unroll
      for (int k = 0; k < VW; ++k) { r0[k] = 0.f; r1[k] = 0.f; }
    }
    one(i0, v0, r0);
    one(i1, v1, r1);
  }
  if (j < per_b) {
    const long long i0 = base + j;
    float v0[VW], r0[VW];
    ldv<T, VW>(x + i0 * VW, v0);
    if (res) ldv<T, VW>(res + i0 * VW, r0);
    else {
#pragma unroll
      for (int k = 0; k < VW; ++k) r0[k] = 0.f;
    }
    one(i0, v0, r0);
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

// FOLD: bsums are folded here from the partials of instnorm_bwd_stats_kernel (see in_fold_parts) and published by workgroup 0
template <typename T, int VW, bool FOLD>
__global__ void __launch_bounds__(256) instnorm_bwd_apply_fixedc_kernel(const T* __restrict__ dy, const T* __restrict__ dy2,
                                                                        const T* __restrict__ dy3, const T* __restrict__ x,
                                                                        const float* __restrict__ sums,
                                                                        float* __restrict__ bsums, T* __restrict__ dx,
                                                                        long long S, int C, int act, float slope, float p,
                                                                        uint64_t seed, const uint64_t* step, const float* __restrict__ ws,
                                                                        int nchunks) {
  __shared__ float fold[FOLD ? IN_FOLD_LDS : 1];
  const int b = blockIdx.y;
  const long long per_b = S * C / VW;
  const float invS = 1.f / (float)S;
  const DropCfg dc = make_drop(p, seed, step);
  const int c = (threadIdx.x * VW) % C;
  float mean[VW], rstd[VW], b0[VW], b1[VW];
  if constexpr (FOLD) {
    in_fold_parts(ws + (long long)b * nchunks * C * 2, nchunks, C * 2, fold, threadIdx.x);
    if (blockIdx.x == 0)
      for (int i = threadIdx.x; i < 2 * C; i += 256) bsums[(long long)b * C * 2 + i] = fold[i];
  }
#pragma unroll
  for (int k = 0; k < VW; ++k) {
    const InStat st = in_stat(sums + ((long long)b * C + c + k) * 3, invS);
    mean[k] = st.mean; rstd[k] = st.rstd;
    if constexpr (FOLD) {
      b0[k] = fold[(c + k) * 2] * invS;
      b1[k] = fold[(c + k) * 2 + 1] * invS;
    } else {
      b0[k] = bsums[((long long)b * C + c + k) * 2] * invS;
      b1[k] = bsums[((long long)b * C + c + k) * 2 + 1] * invS;
    }
  }
  const long long base = (long long)b * per_b;
  for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < per_b; j += (long long)gridDim.x * 256) {
    const long long i = base + j;
    float xv[VW], g[VW], mk[VW], o[VW];
    ldv<T, VW>(x + i * VW, xv);
    load_gradv<T, VW>(dy, dy2, dy3, i * VW, g);
    dropmaskv<VW>(dc, i * VW, mk);
#pragma unroll
    for (int k = 0; k < VW; ++k) {
      const float h = (xv[k] - mean[k]) * rstd[k];
      float gg = g[k] * mk[k];
      if (act == LTU_ACT_LRELU && h <= 0.f) gg *= slope;
      o[k] = rstd[k] * (gg - b0[k] - h * b1[k]);
    }
    stv<T, VW>(dx + i * VW, o);
  }
}

// backward reductions: bsums[b][c][2] += { sum g, sum g*xhat },  g = dy*mask*act'(xhat)
template <typename T, int VW>
__global__ void __launch_bounds__(256) instnorm_bwd_stats_kernel(const T* __restrict__ dy, const T* __restrict__ dy2, const T* __restrict__ dy3,
                                          const T* __restrict__ x, const float* __restrict__ sums,
                                          float* __restrict__ bsums, float* __restrict__ ws, long long S, int C,
                                          int rows_per_block, int act, float slope, float p, uint64_t seed, const uint64_t* step) {
  extern __shared__ float red[];
  const int b = blockIdx.y;
  const int cv = C / VW;
  const int tid = threadIdx.x;
  const int v = tid % cv, rg = tid / cv, nrg = blockDim.x / cv;
  const float invS = 1.f / (float)S;
  const DropCfg dc = make_drop(p, seed, step);
  float a1[VW], a2[VW];
#pragma unroll
  for (int k = 0; k < VW; ++k) { a1[k] = 0.f; a2[k] = 0.f; }
  if (rg < nrg) {
    float mean[VW], rstd[VW];
#pragma unroll
    for (int k = 0; k < VW; ++k) {
      const InStat st = in_stat(sums + ((long long)b * C + v * VW + k) * 3, invS);
      mean[k] = st.mean; rstd[k] = st.rstd;
    }
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > S) r1 = S;
    auto add = [&](long long e, const float (&xv)[VW], const float (&g)[VW]) {
      float mk[VW];
      dropmaskv<VW>(dc, e, mk);
#pragma unroll
      for (int k = 0; k < VW; ++k) {
        const float h = (xv[k] - mean[k]) * rstd[k];
        float gg = g[k] * mk[k];
        if (act == LTU_ACT_LRELU && h <= 0.f) gg *= slope;
        a1[k] += gg;
        a2[k] += gg * h;
      }
    };
    long long r = r0 + rg;
    for (; r + nrg < r1; r += 2 * nrg) {        // two rows per step: their loads are in flight together
      const long long e0 = ((long long)b * S + r) * C + v * VW, e1 = e0 + (long long)nrg * C;
      float x0[VW], g0[VW], x1[VW], g1[VW];
      ldv<T, VW>(x + e0, x0);
      ldv<T, VW>(x + e1, x1);
      load_gradv<T, VW>(dy, dy2, dy3, e0, g0);
      load_gradv<T, VW>(dy, dy2, dy3, e1, g1);
      add(e0, x0, g0);
      add(e1, x1, g1);
    }
    if (r < r1) {
      const long long e0 = ((long long)b * S + r) * C + v * VW;
      float x0[VW], g0[VW];
      ldv<T, VW>(x + e0, x0);
      load_gradv<T, VW>(dy, dy2, dy3, e0, g0);
      add(e0, x0, g0);
    }
    float* dst = red + ((long long)rg * C + v * VW) * 2;
#pragma unroll
    for (int k = 0; k < VW; ++k) { dst[2 * k] = a1[k]; dst[2 * k + 1] = a2[k]; }
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

// ---- small tensors: statistics + apply in ONE launch ---------------------------------------------------------------------------
// At the coarsest level (4x4x32 x 256 per sample: 256 KB) the three launches of a normalisation (partial sums, fold, apply) are
// 4.7-6 us each - launch ramp and dependent L2 round trips, not bytes - and so are the three of its backward (7-16 + 5 + 5-8 us).
// Here one workgroup owns (sample, 16-byte channel group): it holds its part of the tensor in registers (all loads requested before
// the first use), reduces through LDS and writes the result - no partials, no second kernel: 5.6 us forward, 10.7 us backward in
// the step.  The 16 bytes per voxel row mean every 128-byte line is fetched by 8 workgroups, which is what limits the scheme to
// the smallest level: at 1 MB per sample (8x8x64 x 128: 32 workgroups of 1 024 threads pulling 512 KB of lines each through one
// CU's L1) the launch took 14-18 us forward and 23-40 us backward - no better than the three it replaces - so the default limit
// is 256 KB (LTU_IN_SMALL_KB).  Same shift (first voxel), same statistics layout and the same dropout indexing as the streaming
// kernels; the order of the fp32 sums differs (tests: tolerance, not bit equality).
template <typename T, int VW>
__device__ __forceinline__ uint4 in_raw_load(const T* p) { return *reinterpret_cast<const uint4*>(p); }
template <typename T, int VW>
__device__ __forceinline__ void in_raw_cvt(const uint4 r, float (&f)[VW]) {
  if constexpr (sizeof(T) == 2) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  } else {
    f[0] = __uint_as_float(r.x); f[1] = __uint_as_float(r.y); f[2] = __uint_as_float(r.z); f[3] = __uint_as_float(r.w);
  }
}
// block sums of 2 VW values per thread: lds [nwaves][2 VW] + [2 VW]; every thread returns with the totals in a[]
template <int NV>
__device__ __forceinline__ void in_block_sums(float (&a)[NV], float* lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    // pinned: hipcc otherwise contracts the caller's last multiply into the first add of the reduction (fma(d, d, neighbour)), the
    // lanes of a wave then hold totals that differ in the last bit, and two instantiations of one kernel disagree with each other
    asm volatile("" : "+v"(a[k]));
    a[k] = wave_sum(a[k]);
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) lds[wave * NV + k] = a[k];
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    float t = 0.f;
    for (int w = 0; w < nw; ++w) t += lds[w * NV + threadIdx.x];
    lds[nw * NV + threadIdx.x] = t;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) a[k] = lds[nw * NV + k];
}

// grid (C / VW, B); block 256 or 1024; NIT = ceil(S / blockDim.x)
template <typename T, int VW, int NIT, bool APPLY>
__global__ void __launch_bounds__(1024) instnorm_small_fwd_kernel(const T* __restrict__ x, float* __restrict__ sums, const T* __restrict__ res,
                                                                  T* __restrict__ y, int S, int C, int act, float slope, float p,
                                                                  uint64_t seed, const uint64_t* step) {
  __shared__ float lds[17 * 2 * VW];
  const int b = blockIdx.y, c0 = blockIdx.x * VW, nthr = blockDim.x;
  const long long eb = (long long)b * S * C + c0;         // element index of (b, voxel 0, c0)
  uint4 xr[NIT], rr[NIT];
  const uint4 sr = in_raw_load<T, VW>(x + eb);
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int r = (int)threadIdx.x + it * nthr;
    xr[it] = r < S ? in_raw_load<T, VW>(x + eb + (long long)r * C) : sr;       // a row past the end contributes x - shift = 0
    if constexpr (APPLY) rr[it] = (res != nullptr && r < S) ? in_raw_load<T, VW>(res + eb + (long long)r * C) : make_uint4(0u, 0u, 0u, 0u);
  }
  float shift[VW], a[2 * VW];
  in_raw_cvt<T, VW>(sr, shift);
#pragma unroll
  for (int k = 0; k < 2 * VW; ++k) a[k] = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    float t[VW];
    in_raw_cvt<T, VW>(xr[it], t);
#pragma unroll
    for (int k = 0; k < VW; ++k) { const float d = t[k] - shift[k]; a[2 * k] += d; a[2 * k + 1] += d * d; }
  }
  in_block_sums<2 * VW>(a, lds);
  if (threadIdx.x < VW) {
    float* sp = sums + ((long long)b * C + c0 + threadIdx.x) * 3;
    sp[0] = shift[threadIdx.x]; sp[1] = a[2 * threadIdx.x]; sp[2] = a[2 * threadIdx.x + 1];
  }
  if constexpr (APPLY) {
    const float invS = 1.f / (float)S;
    const DropCfg dc = make_drop(p, seed, step);
    float mean[VW], rstd[VW];
#pragma unroll
    for (int k = 0; k < VW; ++k) {
      const float m1 = a[2 * k] * invS;
      const float var = fmaxf(a[2 * k + 1] * invS - m1 * m1, 0.f);
      mean[k] = shift[k] + m1;
      rstd[k] = rsqrtf(var + IN_EPS);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int r = (int)threadIdx.x + it * nthr;
      if (r >= S) break;
      const long long e = eb + (long long)r * C;
      float t[VW], rs[VW], mk[VW], o[VW];
      in_raw_cvt<T, VW>(xr[it], t);
      in_raw_cvt<T, VW>(rr[it], rs);
      dropmaskv<VW>(dc, e, mk);
#pragma unroll
      for (int k = 0; k < VW; ++k) {
        float h = (t[k] - mean[k]) * rstd[k];
        if (act == LTU_ACT_LRELU) h = h > 0.f ? h : h * slope;
        o[k] = h * mk[k] + rs[k];
      }
      stv<T, VW>(y + e, o);
    }
  }
}

template <typename T, int VW, int NIT>
__global__ void __launch_bounds__(1024) instnorm_small_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ dy2, const T* __restrict__ dy3,
                                                                  const T* __restrict__ x, const float* __restrict__ sums,
                                                                  float* __restrict__ bsums, T* __restrict__ dx, int S, int C, int act,
                                                                  float slope, float p, uint64_t seed, const uint64_t* step) {
  __shared__ float lds[17 * 2 * VW];
  const int b = blockIdx.y, c0 = blockIdx.x * VW, nthr = blockDim.x;
  const long long eb = (long long)b * S * C + c0;
  const float invS = 1.f / (float)S;
  const DropCfg dc = make_drop(p, seed, step);
  uint4 xr[NIT], g1[NIT], g2[NIT], g3[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int r = (int)threadIdx.x + it * nthr;
    const long long e = eb + (long long)(r < S ? r : 0) * C;
    xr[it] = in_raw_load<T, VW>(x + e);
    g1[it] = in_raw_load<T, VW>(dy + e);
    if (dy2 != nullptr) g2[it] = in_raw_load<T, VW>(dy2 + e);
    if (dy3 != nullptr) g3[it] = in_raw_load<T, VW>(dy3 + e);
  }
  float mean[VW], rstd[VW];
#pragma unroll
  for (int k = 0; k < VW; ++k) {
    const InStat st = in_stat(sums + ((long long)b * C + c0 + k) * 3, invS);
    mean[k] = st.mean; rstd[k] = st.rstd;
  }
  float gg[NIT][VW], a[2 * VW];
#pragma unroll
  for (int k = 0; k < 2 * VW; ++k) a[k] = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int r = (int)threadIdx.x + it * nthr;
    float xv[VW], t[VW], mk[VW];
    in_raw_cvt<T, VW>(xr[it], xv);
    in_raw_cvt<T, VW>(g1[it], gg[it]);
    if (dy2 != nullptr) {
      in_raw_cvt<T, VW>(g2[it], t);
#pragma unroll
      for (int k = 0; k < VW; ++k) gg[it][k] += t[k];
    }
    if (dy3 != nullptr) {
      in_raw_cvt<T, VW>(g3[it], t);
#pragma unroll
      for (int k = 0; k < VW; ++k) gg[it][k] += t[k];
    }
    dropmaskv<VW>(dc, eb + (long long)(r < S ? r : 0) * C, mk);
#pragma unroll
    for (int k = 0; k < VW; ++k) {
      const float h = (xv[k] - mean[k]) * rstd[k];
      float v = gg[it][k] * mk[k];
      if (act == LTU_ACT_LRELU && h <= 0.f) v *= slope;
      if (r >= S) v = 0.f;
      gg[it][k] = v;
      a[2 * k] += v;
      a[2 * k + 1] += v * h;
    }
  }
  in_block_sums<2 * VW>(a, lds);
  if (threadIdx.x < 2 * VW) bsums[((long long)b * C + c0) * 2 + threadIdx.x] = a[threadIdx.x];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int r = (int)threadIdx.x + it * nthr;
    if (r >= S) break;
    float xv[VW], o[VW];
    in_raw_cvt<T, VW>(xr[it], xv);
#pragma unroll
    for (int k = 0; k < VW; ++k) {
      const float h = (xv[k] - mean[k]) * rstd[k];
      o[k] = rstd[k] * (gg[it][k] - a[2 * k] * invS - h * a[2 * k + 1] * invS);
    }
    stv<T, VW>(dx + eb + (long long)r * C, o);
  }
}

// the fused small-tensor path applies when a sample is at most LTU_IN_SMALL_KB KB (default 256), S <= 4096 and the channels split into
// 16-byte groups; returns the thread count (256 / 1024) and the trip count per thread, or 0
static int in_small_plan(int dtype, long long S, int C, int* nit) {
  const int vw = dtype == LTU_BF16 ? 8 : 4, esz = dtype == LTU_BF16 ? 2 : 4;
  if (C % vw || S > 4096 || S < 1 || S * C * esz > (long long)ltu_knob_pos("LTU_IN_SMALL_KB", 256) * 1024 || ltu_knob("LTU_NO_IN_SMALL", 0)) return 0;
  const int nthr = S <= 1024 ? 256 : 1024;
  const int n = (int)((S + nthr - 1) / nthr);
  *nit = n <= 1 ? 1 : n <= 2 ? 2 : 4;
  return nthr;
}
#define IN_SMALL_DISPATCH(nit, ...)                       \
  do {                                                    \
    if ((nit) == 1) { constexpr int NIT = 1; __VA_ARGS__ } \
    else if ((nit) == 2) { constexpr int NIT = 2; __VA_ARGS__ } \
    else { constexpr int NIT = 4; __VA_ARGS__ }           \
  } while (0)

// ------------------------------------------------------------------------------------------------ host side
static int stats_rows(long long S, int B, int* nchunks) {
  int chunks = -1;
  chunks = ltu_knob_pos("LTU_IN_CHUNKS", 1024);
  long long want = chunks / (B > 0 ? B : 1);
  if (want < 1) want = 1;
  long long rows = (S + want - 1) / want;
  if (rows < 64) rows = 64;
  *nchunks = (int)((S + rows - 1) / rows);
  return (int)rows;
}

// vector width of the InstanceNorm kernels.  8 elements (16 bytes of bf16) is built and tested but OFF: at 34 / 17 / 4 MB the
// statistics pass took 12.7 / 11.2 / 2.6 us against 6.5 / 4.9 / 1.3 (half the loop trips, the same epilogue) and the three-tensor
// backward 41 / 32 / 13.5 against 37 / 25 / 12 (tools/sweep_in.sh): these passes already stream at 4-5 TB/s at 8 bytes per lane.
// Both variants below (LTU_IN_VW8, LTU_IN_FOLD) lost their measurements and are compiled only into an experiments build
// (make EXPERIMENTS=1 -> -DLTU_EXPERIMENTS); the product library carries the 8-byte two-stage kernels alone.
static int in_vw(int dtype, int C) {
#ifdef LTU_EXPERIMENTS
  return (dtype == LTU_BF16 && C % 8 == 0 && 256 % (C / 8) == 0 && ltu_knob("LTU_IN_VW8", 0)) ? 8 : 4;
#else
  return 4;
#endif
}
// Second stage folded by the apply kernel (in_fold_parts): chunk count per sample such that the partials of a sample stay within
// IN_FOLD_MAX floats.  Returns false when the shape does not qualify (the two-stage path with its own fold launch runs then).
static bool in_fold_plan(long long S, int B, int C, int vw, const float* ws, int* nchunks, int* rows) {
  // OFF by default (LTU_IN_FOLD=1 enables): measured per tensor size 34 / 17 / 4 / 1 MB (tools/sweep_in.sh), forward 25.0 / 19.2 /
  // 11.5 / 7.3 us against 27.9 / 20.3 / 10.2 / 8.0 with the fold launch, backward 40.1 / 29.6 / 15.4 / 9.4 against 36.8 / 25.5 / 11.8 /
  // 10.1: bounding the partials to a few KB per sample leaves the statistics kernels 4x fewer workgroups, which costs the
  // mid-sized tensors and the heavier backward statistics more than the 5 us launch this removes.
#ifndef LTU_EXPERIMENTS
  return false;
#endif
  if (ws == nullptr || !ltu_knob("LTU_IN_FOLD", 0)) return false;
  if ((256 * vw) % C != 0 || 2 * C > IN_FOLD_LDS || C < 4) return false;
  long long want = IN_FOLD_MAX / (2 * C);
  const long long cap = ltu_knob_pos("LTU_IN_CHUNKS", 1024) / (B > 0 ? B : 1);
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  long long r = (S + want - 1) / want;
  if (r < 64) r = 64;
  *nchunks = (int)((S + r - 1) / r);
  *rows = (int)r;
  return (long long)*nchunks * B * C * 2 <= LTU_NORM_WS_FLOATS;
}
#ifdef LTU_EXPERIMENTS
#define IN_DISPATCH_VW(vw, ...)                    \
  do {                                             \
    if ((vw) == 8) { constexpr int VW = 8; __VA_ARGS__ } \
    else { constexpr int VW = 4; __VA_ARGS__ }     \
  } while (0)
#else
#define IN_DISPATCH_VW(vw, ...)                    \
  do {                                             \
    constexpr int VW = 4; __VA_ARGS__              \
  } while (0)
#endif

static unsigned stream_grid(long long nvec) {
  long long blocks = (nvec + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

static unsigned per_sample_grid(long long nvec_per_sample, int B, int total = 4096) {
  long long blocks = (nvec_per_sample + 255) / 256;
  const long long cap = total / (B > 0 ? B : 1) > 1 ? total / (B > 0 ? B : 1) : 1;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

template <typename T, int VW>
static void launch_in_stats(const void* x, float* sums, float* ws, int B, long long S, int C, int nchunks, int rows, hipStream_t st) {
  const size_t lds = (size_t)(256 / (C / VW)) * C * 2 * sizeof(float);
  hipLaunchKernelGGL((instnorm_stats_kernel<T, VW>), dim3(nchunks, B), dim3(256), lds, st, (const T*)x, sums, ws, S, C, rows);
}
template <typename T, int VW, bool FOLD>
static void launch_in_apply(const void* x, float* sums, const void* res, void* y, int B, long long S, int C, int act, float slope, float p,
                            uint64_t seed, const uint64_t* step, const float* ws, int nchunks, hipStream_t st) {
  const unsigned gx = per_sample_grid(S * C / VW, B, FOLD ? ltu_knob_pos("LTU_IN_FOLD_BLOCKS", 2048) : 4096);
  hipLaunchKernelGGL((instnorm_apply_fixedc_kernel<T, VW, FOLD>), dim3(gx, B), dim3(256), 0, st, (const T*)x, sums, (const T*)res, (T*)y,
                     S, C, act, slope, p, seed, step, ws, nchunks);
}

extern "C" int ltu_instnorm_stats(const void* x, float* sums, float* ws, long long ws_floats, int B, long long S, int C, int dtype,
                                  ltu_stream_t s) {
  if (C % 4 != 0 || C > 1024 || 256 % (C / 4) != 0) return LTU_E_SHAPE;
  if (ws != nullptr && ws_floats < LTU_NORM_WS_FLOATS) return LTU_E_ARG;
  {
    int nit = 0;
    const int nthr = in_small_plan(dtype, S, C, &nit);
    if (nthr) {
      LTU_DISPATCH_T(dtype, {
        constexpr int VW = sizeof(T) == 2 ? 8 : 4;
        IN_SMALL_DISPATCH(nit, {
          hipLaunchKernelGGL((instnorm_small_fwd_kernel<T, VW, NIT, false>), dim3(C / VW, B), dim3(nthr), 0, (hipStream_t)s, (const T*)x, sums,
                             (const T*)nullptr, (T*)nullptr, (int)S, C, 0, 0.f, 0.f, (uint64_t)0, (const uint64_t*)nullptr);
        });
      });
      return ltu_check_launch();
    }
  }
  int nchunks;
  const int rows = stats_rows(S, B, &nchunks);
  const int vw = in_vw(dtype, C);
  (void)vw;
  LTU_DISPATCH_T(dtype, {
    if ((long long)nchunks * B * C * 2 > LTU_NORM_WS_FLOATS) ws = nullptr;
    IN_DISPATCH_VW(vw, { launch_in_stats<T, VW>(x, sums, ws, B, S, C, nchunks, rows, (hipStream_t)s); });
    if (ws != nullptr) launch_reduce_parts(ws, nchunks, C * 2, B, sums, nullptr, 1, (hipStream_t)s);
  });
  return ltu_check_launch();
}

extern "C" int ltu_instnorm_apply(const void* x, const float* sums, const void* res, void* y, int B, long long S, int C,
                                  int act, float slope, float p, uint64_t seed, const uint64_t* step, int dtype, ltu_stream_t s) {
  if (C % 4 != 0) return LTU_E_SHAPE;
  const long long nvec = (long long)B * S * C / 4;
  const int vw = (256 * in_vw(dtype, C)) % C == 0 ? in_vw(dtype, C) : 4;
  const bool fixedc = (256 * vw) % C == 0;     // the loop stride (256 vectors) is then a multiple of the channel count
  LTU_DISPATCH_T(dtype, {
    if (fixedc)
      IN_DISPATCH_VW(vw, {
        launch_in_apply<T, VW, false>(x, const_cast<float*>(sums), res, y, B, S, C, act, slope, p, seed, step, nullptr, 0, (hipStream_t)s);
      });
    else
      hipLaunchKernelGGL((instnorm_apply_kernel<T>), dim3(stream_grid(nvec)), dim3(256), 0, (hipStream_t)s, (const T*)x, sums,
                         (const T*)res, (T*)y, S, C, B, act, slope, p, seed, step);
  });
  return ltu_check_launch();
}

// statistics + apply in one call: sums [B][C][3] (zero on entry) receives the statistics for the backward pass.  When the shape
// qualifies the statistics kernel's partials are folded by the apply kernel itself (no fold launch in between).
extern "C" int ltu_instnorm_fwd(const void* x, float* sums, float* ws, long long ws_floats, const void* res, void* y, int B, long long S, int C,
                                int act, float slope, float p, uint64_t seed, const uint64_t* step, int dtype, ltu_stream_t s) {
  if (C % 4 != 0 || C > 1024 || 256 % (C / 4) != 0) return LTU_E_SHAPE;
  if (ws != nullptr && ws_floats < LTU_NORM_WS_FLOATS) return LTU_E_ARG;
  {
    int nit = 0;
    const int nthr = in_small_plan(dtype, S, C, &nit);
    if (nthr) {
      LTU_DISPATCH_T(dtype, {
        constexpr int VW = sizeof(T) == 2 ? 8 : 4;
        IN_SMALL_DISPATCH(nit, {
          hipLaunchKernelGGL((instnorm_small_fwd_kernel<T, VW, NIT, true>), dim3(C / VW, B), dim3(nthr), 0, (hipStream_t)s, (const T*)x, sums,
                             (const T*)res, (T*)y, (int)S, C, act, slope, p, seed, step);
        });
      });
      return ltu_check_launch();
    }
  }
  const int vw = in_vw(dtype, C);
  int nchunks = 0, rows = 0;
  if (!in_fold_plan(S, B, C, vw, ws, &nchunks, &rows)) {
    const int rc = ltu_instnorm_stats(x, sums, ws, ws_floats, B, S, C, dtype, s);
    return rc != LTU_OK ? rc : ltu_instnorm_apply(x, sums, res, y, B, S, C, act, slope, p, seed, step, dtype, s);
  }
#ifdef LTU_EXPERIMENTS
  LTU_DISPATCH_T(dtype, {
    IN_DISPATCH_VW(vw, {
      launch_in_stats<T, VW>(x, sums, ws, B, S, C, nchunks, rows, (hipStream_t)s);
      launch_in_apply<T, VW, true>(x, sums, res, y, B, S, C, act, slope, p, seed, step, ws, nchunks, (hipStream_t)s);
    });
  });
#endif
  return ltu_check_launch();
}

extern "C" int ltu_instnorm_bwd(const void* dy, const void* dy2, const void* dy3, const void* x, const float* sums, float* bsums,
                                float* ws, long long ws_floats, void* dx, int B,
                                long long S, int C, int act, float slope, float p, uint64_t seed, const uint64_t* step, int dtype,
                                ltu_stream_t s) {
  if (C % 4 != 0 || C > 1024 || 256 % (C / 4) != 0) return LTU_E_SHAPE;
  if (ws != nullptr && ws_floats < LTU_NORM_WS_FLOATS) return LTU_E_ARG;
  {
    int nit = 0;
    const int nthr = in_small_plan(dtype, S, C, &nit);
    if (nthr) {
      LTU_DISPATCH_T(dtype, {
        constexpr int VW = sizeof(T) == 2 ? 8 : 4;
        IN_SMALL_DISPATCH(nit, {
          hipLaunchKernelGGL((instnorm_small_bwd_kernel<T, VW, NIT>), dim3(C / VW, B), dim3(nthr), 0, (hipStream_t)s, (const T*)dy, (const T*)dy2,
                             (const T*)dy3, (const T*)x, sums, bsums, (T*)dx, (int)S, C, act, slope, p, seed, step);
        });
      });
      return ltu_check_launch();
    }
  }
  const int vw = in_vw(dtype, C);
  int nchunks = 0, rows = 0;
  const bool fold = in_fold_plan(S, B, C, vw, ws, &nchunks, &rows);
  if (!fold) rows = stats_rows(S, B, &nchunks);
  const long long nvec = (long long)B * S * C / 4;
  hipStream_t st = (hipStream_t)s;
  LTU_DISPATCH_T(dtype, {
    if (!fold && (long long)nchunks * B * C * 2 > LTU_NORM_WS_FLOATS) ws = nullptr;
    IN_DISPATCH_VW(vw, {
      const size_t lds = (size_t)(256 / (C / VW)) * C * 2 * sizeof(float);
      hipLaunchKernelGGL((instnorm_bwd_stats_kernel<T, VW>), dim3(nchunks, B), dim3(256), lds, st, (const T*)dy, (const T*)dy2,
                         (const T*)dy3, (const T*)x, sums, bsums, ws, S, C, rows, act, slope, p, seed, step);
#ifdef LTU_EXPERIMENTS
      if (fold) {
        hipLaunchKernelGGL((instnorm_bwd_apply_fixedc_kernel<T, VW, true>),
                           dim3(per_sample_grid(S * C / VW, B, ltu_knob_pos("LTU_IN_FOLD_BLOCKS", 2048)), B), dim3(256), 0, st, (const T*)dy,
                           (const T*)dy2, (const T*)dy3, (const T*)x, sums, bsums, (T*)dx, S, C, act, slope, p, seed, step, ws, nchunks);
      } else
#endif
      {
        if (ws != nullptr) launch_reduce_parts(ws, nchunks, C * 2, B, bsums, nullptr, 0, st);
        if ((256 * VW) % C == 0)
          hipLaunchKernelGGL((instnorm_bwd_apply_fixedc_kernel<T, VW, false>), dim3(per_sample_grid(S * C / VW, B), B), dim3(256), 0, st,
                             (const T*)dy, (const T*)dy2, (const T*)dy3, (const T*)x, sums, bsums, (T*)dx, S, C, act, slope, p, seed, step,
                             nullptr, 0);
        else
          hipLaunchKernelGGL((instnorm_bwd_apply_kernel<T>), dim3(stream_grid(nvec)), dim3(256), 0, st, (const T*)dy,
                             (const T*)dy2, (const T*)dy3, (const T*)x, sums, bsums, (T*)dx, S, C, B, act, slope, p, seed, step);
      }
    });
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
                                 float* dgamma, float* dbeta, float* ws, long long ws_floats, ltu_reduce_job* defer, long long M, int d,
                                 float p, uint64_t seed, const uint64_t* step, int dtype, ltu_stream_t s) {
  if (defer != nullptr) defer->part = nullptr;
  LTU_DISPATCH_T(dtype, {
    LN_DISPATCH_GV(d, {
      const int nrg = 256 / G;
      long long rows = (M + 1023) / 1024;        // 1024 workgroups (swept 512 / 1024 / 2048; the callers size the workspace for <= 2048)
      if (rows < nrg) rows = nrg;
      rows = (rows + nrg - 1) / nrg * nrg;
      const size_t lds = (size_t)nrg * d * 2 * sizeof(float);
      const int nblk = cdiv(M, rows);
      if (ws != nullptr && (long long)nblk * d * 2 > ws_floats) return LTU_E_ARG;     // [nblk][2 d] partial rows
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
  if (jb.mode == 1 && (jb.n & 31) == 0) {
    // LayerNorm gamma / beta sums of the chain kernels: one partial row per 32-token block, i.e. thousands of splits of a few hundred
    // floats.  A thread owns a float4 column and every 128th split and requests 8 of them at a time (as 4 scalar loads per trip
    // the 7 176 splits of the largest level were 56 dependent round trips: 21 us for 7 MB).
    __shared__ float4 red4[16][8];
    const int c4 = threadIdx.x & 7, zg = threadIdx.x >> 3;
    const float* p = jb.part + (long long)blockIdx.x * 32 + c4 * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z0 = zg; z0 < jb.nsplit; z0 += 128 * 8) {
      float4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int z = z0 + 128 * k;
        v[k] = *reinterpret_cast<const float4*>(p + (long long)(z < jb.nsplit ? z : z0) * jb.n);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (z0 + 128 * k < jb.nsplit) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
    }
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {         // the 8 split groups of a wave
      acc.x += __shfl_xor(acc.x, o); acc.y += __shfl_xor(acc.y, o); acc.z += __shfl_xor(acc.z, o); acc.w += __shfl_xor(acc.w, o);
    }
    if ((threadIdx.x & 63) < 8) red4[threadIdx.x >> 6][c4] = acc;
    __syncthreads();
    if (threadIdx.x >= 32) return;
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) v += reinterpret_cast<const float*>(&red4[q][0])[threadIdx.x];
    float* o = (e & 1) ? jb.out[1] : jb.out[0];
    o[e >> 1] += v;
    return;
  }
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
