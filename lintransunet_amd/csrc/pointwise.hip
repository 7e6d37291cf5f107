// Streaming (HBM-bound) pointwise / small-stencil kernels of the LinTransUNet hot path for gfx950:
// window (un)embedding, weight repacking, GELU+dropout, class softmax heads, the attention gate,
// the depthwise positional conv and the nearest-upsampling adjoint.  Channels-last, 4-wide vectors.
#include "common.h"

static unsigned sgrid(long long n, int per_block = 256) {
  long long blocks = (n + per_block - 1) / per_block;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}
#define GRID_STRIDE(i, n) \
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)

// ------------------------------------------------------------------------------------------------ window embed
// x f32 [B,1,H,W,D] -> y T [B,H/2,W/2,D,8]  (model/Unet_3Dblock.py:123-136; channels 4..7 zero)
template <typename T>
__global__ void window_embed_kernel(const float* __restrict__ x, T* __restrict__ y, int B, int H, int W, int D) {
  const int h2 = H / 2, w2 = W / 2;
  const long long n = (long long)B * h2 * w2 * D;
  GRID_STRIDE(i, n) {
    const int d = (int)(i % D);
    long long t = i / D;
    const int w = (int)(t % w2);
    t /= w2;
    const int h = (int)(t % h2);
    const int b = (int)(t / h2);
    const float* xb = x + (((long long)b * H + 2 * h) * W + 2 * w) * D + d;
    float4 v = make_float4(xb[0], xb[D], xb[(long long)W * D], xb[(long long)W * D + D]);   // (kh,kw) = 00,01,10,11
    Vec4<T>::store(y + i * 8, v);
    Vec4<T>::store(y + i * 8 + 4, make_float4(0.f, 0.f, 0.f, 0.f));
  }
}

extern "C" int ltu_window_embed(const float* x, void* y, int dtype, int B, int H, int W, int D, ltu_stream_t s) {
  if (H % 2 || W % 2) return LTU_E_SHAPE;
  const long long n = (long long)B * (H / 2) * (W / 2) * D;
  LTU_DISPATCH_T(dtype, { hipLaunchKernelGGL((window_embed_kernel<T>), dim3(sgrid(n)), dim3(256), 0, (hipStream_t)s, x, (T*)y, B, H, W, D); });
  return ltu_check_launch();
}

// ------------------------------------------------------------------------------------------------ weight repacking
// w [Co][Ci][27] -> wf [CoP][27][CiP], wd [CiP][27][CoP]  (zero padded), stored as TW (fp32 or bf16)
template <typename TW>
__global__ void pack_conv_kernel(const float* __restrict__ w, TW* __restrict__ wf, TW* __restrict__ wd, int Co, int Ci,
                                 int CoP, int CiP) {
  const long long n = (long long)CoP * 27 * CiP;
  GRID_STRIDE(i, n) {
    const int ci = (int)(i % CiP);
    const int t = (int)((i / CiP) % 27);
    const int co = (int)(i / ((long long)CiP * 27));
    const float v = (co < Co && ci < Ci) ? w[((long long)co * Ci + ci) * 27 + t] : 0.f;
    if (wf) st1<TW>(wf + i, v);
    if (wd) st1<TW>(wd + ((long long)ci * 27 + t) * CoP + co, v);
  }
}
__global__ void unpack_conv_kernel(const float* __restrict__ dwf, float* __restrict__ dw, int Co, int Ci, int CiP) {
  const long long n = (long long)Co * Ci * 27;
  GRID_STRIDE(i, n) {
    const int t = (int)(i % 27);
    const int ci = (int)((i / 27) % Ci);
    const int co = (int)(i / (27LL * Ci));
    dw[i] = dwf[((long long)co * 27 + t) * CiP + ci];
  }
}
template <typename TW>
__global__ void transpose_kernel(const float* __restrict__ in, TW* __restrict__ out, int R, int C, int ldo, int col_off) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const int r = r0 + j, c = c0 + threadIdx.x;
    if (r < R && c < C) tile[j][threadIdx.x] = in[(long long)r * C + c];
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const int c = c0 + j, r = r0 + threadIdx.x;
    if (r < R && c < C) st1<TW>(out + (long long)c * ldo + col_off + r, tile[threadIdx.x][j]);
  }
}
template <typename TW>
__global__ void cast_kernel(const float* __restrict__ in, TW* __restrict__ out, long long n) {
  GRID_STRIDE(i, n) st1<TW>(out + i, in[i]);
}

extern "C" int ltu_pack_conv_weight(const float* w, void* wf, void* wd, int Co, int Ci, int CoP, int CiP, int out_dtype,
                                    ltu_stream_t s) {
  if (CoP < Co || CiP < Ci) return LTU_E_SHAPE;
  LTU_DISPATCH_T(out_dtype, {
    hipLaunchKernelGGL((pack_conv_kernel<T>), dim3(sgrid((long long)CoP * 27 * CiP)), dim3(256), 0, (hipStream_t)s, w, (T*)wf,
                       (T*)wd, Co, Ci, CoP, CiP);
  });
  return ltu_check_launch();
}
extern "C" int ltu_unpack_conv_wgrad(const float* dwf, float* dw, int Co, int Ci, int CiP, ltu_stream_t s) {
  hipLaunchKernelGGL(unpack_conv_kernel, dim3(sgrid((long long)Co * Ci * 27)), dim3(256), 0, (hipStream_t)s, dwf, dw, Co, Ci, CiP);
  return ltu_check_launch();
}
extern "C" int ltu_transpose_f32(const float* in, void* out, int R, int C, int ldo, int col_off, int out_dtype, ltu_stream_t s) {
  LTU_DISPATCH_T(out_dtype, {
    hipLaunchKernelGGL((transpose_kernel<T>), dim3(cdiv(C, 32), cdiv(R, 32)), dim3(32, 8), 0, (hipStream_t)s, in, (T*)out, R, C,
                       ldo, col_off);
  });
  return ltu_check_launch();
}
extern "C" int ltu_cast_f32(const float* in, void* out, long long n, int out_dtype, ltu_stream_t s) {
  LTU_DISPATCH_T(out_dtype, { hipLaunchKernelGGL((cast_kernel<T>), dim3(sgrid(n)), dim3(256), 0, (hipStream_t)s, in, (T*)out, n); });
  return ltu_check_launch();
}

// ------------------------------------------------------------------------------------------------ nearest x2 adjoint
template <typename T>
__global__ void sumpool2_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int D, int C) {
  const int cv = C / 4;
  const long long n = (long long)B * H * W * D * cv;
  GRID_STRIDE(i, n) {
    const int v = (int)(i % cv);
    long long t = i / cv;
    const int d = (int)(t % D); t /= D;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const int b = (int)(t / H);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const long long vox = (((long long)b * 2 * H + 2 * h + (k >> 2)) * 2 * W + 2 * w + ((k >> 1) & 1)) * 2 * D + 2 * d + (k & 1);
      const float4 q = Vec4<T>::load(x + vox * C + v * 4);
      a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
    }
    Vec4<T>::store(y + i * 4, a);
  }
}
extern "C" int ltu_sumpool2(const void* x, void* y, int B, int H, int W, int D, int C, int dtype, ltu_stream_t s) {
  if (C % 4) return LTU_E_SHAPE;
  const long long n = (long long)B * H * W * D * (C / 4);
  LTU_DISPATCH_T(dtype, { hipLaunchKernelGGL((sumpool2_kernel<T>), dim3(sgrid(n)), dim3(256), 0, (hipStream_t)s, (const T*)x, (T*)y, B, H, W, D, C); });
  return ltu_check_launch();
}

// ------------------------------------------------------------------------------------------------ GELU + dropout
// h = drop(gelu(u))  (model/trans_block.py:208);  du = dh * mask * gelu'(u)
template <typename T>
__global__ void gelu_drop_fwd_kernel(const T* __restrict__ u, T* __restrict__ h, long long nvec, float p, uint64_t seed, const uint64_t* step) {
  const DropCfg dc = make_drop(p, seed, step);
  GRID_STRIDE(i, nvec) {
    float4 v = Vec4<T>::load(u + i * 4);
    v = make_float4(gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w));
    Vec4<T>::store(h + i * 4, drop4(dc, (uint64_t)i, v));
  }
}
template <typename T>
__global__ void gelu_drop_bwd_kernel(const T* __restrict__ dh, const T* __restrict__ u, T* __restrict__ du, long long nvec,
                                     float p, uint64_t seed, const uint64_t* step) {
  const DropCfg dc = make_drop(p, seed, step);
  GRID_STRIDE(i, nvec) {
    const float4 v = Vec4<T>::load(u + i * 4);
    const float4 g = Vec4<T>::load(dh + i * 4);
    const float4 m = dropmask4(dc, (uint64_t)i);
    Vec4<T>::store(du + i * 4, make_float4(g.x * m.x * gelu_erf_grad(v.x), g.y * m.y * gelu_erf_grad(v.y),
                                           g.z * m.z * gelu_erf_grad(v.z), g.w * m.w * gelu_erf_grad(v.w)));
  }
}
extern "C" int ltu_gelu_dropout_fwd(const void* u, void* h, long long n, float p, uint64_t seed, const uint64_t* step, int dtype, ltu_stream_t s) {
  if (n % 4) return LTU_E_SHAPE;
  LTU_DISPATCH_T(dtype, { hipLaunchKernelGGL((gelu_drop_fwd_kernel<T>), dim3(sgrid(n / 4)), dim3(256), 0, (hipStream_t)s, (const T*)u, (T*)h, n / 4, p, seed, step); });
  return ltu_check_launch();
}
extern "C" int ltu_gelu_dropout_bwd(const void* dh, const void* u, void* du, long long n, float p, uint64_t seed, const uint64_t* step, int dtype,
                                    ltu_stream_t s) {
  if (n % 4) return LTU_E_SHAPE;
  LTU_DISPATCH_T(dtype, { hipLaunchKernelGGL((gelu_drop_bwd_kernel<T>), dim3(sgrid(n / 4)), dim3(256), 0, (hipStream_t)s, (const T*)dh, (const T*)u, (T*)du, n / 4, p, seed, step); });
  return ltu_check_launch();
}

// ------------------------------------------------------------------------------------------------ class softmax heads
// mask head (model/Unet_3Dblock.py:1380-1381): logits T [M][CP] (CP = padded conv width) -> probs f32 [M][C]
template <typename T>
__global__ void head_softmax_fwd_kernel(const T* __restrict__ z, float* __restrict__ p, long long M, int C, int CP) {
  GRID_STRIDE(m, M) {
    float v[4];
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) { v[c] = ld1<T>(z + m * CP + c); mx = fmaxf(mx, v[c]); }
    float sum = 0.f;
    for (int c = 0; c < C; ++c) { v[c] = expf(v[c] - mx); sum += v[c]; }
    for (int c = 0; c < C; ++c) p[m * C + c] = v[c] / sum;
  }
}
template <typename T>
__global__ void head_softmax_bwd_kernel(const float* __restrict__ dp, const float* __restrict__ p, T* __restrict__ dz,
                                        long long M, int C, int CP) {
  GRID_STRIDE(m, M) {
    float dot = 0.f;
    for (int c = 0; c < C; ++c) dot += dp[m * C + c] * p[m * C + c];
    for (int c = 0; c < CP; ++c) st1<T>(dz + m * CP + c, c < C ? p[m * C + c] * (dp[m * C + c] - dot) : 0.f);
  }
}
extern "C" int ltu_head_softmax_fwd(const void* z, float* p, long long M, int C, int CP, int dtype, ltu_stream_t s) {
  if (C > 4 || CP < C) return LTU_E_SHAPE;
  LTU_DISPATCH_T(dtype, { hipLaunchKernelGGL((head_softmax_fwd_kernel<T>), dim3(sgrid(M)), dim3(256), 0, (hipStream_t)s, (const T*)z, p, M, C, CP); });
  return ltu_check_launch();
}
extern "C" int ltu_head_softmax_bwd(const float* dp, const float* p, void* dz, long long M, int C, int CP, int dtype,
                                    ltu_stream_t s) {
  if (C > 4 || CP < C) return LTU_E_SHAPE;
  LTU_DISPATCH_T(dtype, { hipLaunchKernelGGL((head_softmax_bwd_kernel<T>), dim3(sgrid(M)), dim3(256), 0, (hipStream_t)s, dp, p, (T*)dz, M, C, CP); });
  return ltu_check_launch();
}

// final head (model/Unet_3Dblock.py:1392-1394): z T [B,h,w,D,4C] -> window un-embedding + softmax over classes
// -> probs f32 [B,2h,2w,D,C];  channel c*4 + kh*2 + kw of voxel (h,w) is class c of voxel (2h+kh, 2w+kw).
template <typename T>
__global__ void final_softmax_fwd_kernel(const T* __restrict__ z, float* __restrict__ p, int B, int h, int w, int D, int C) {
  const long long n = (long long)B * h * w * D;
  GRID_STRIDE(i, n) {
    const int d = (int)(i % D);
    long long t = i / D;
    const int ww = (int)(t % w); t /= w;
    const int hh = (int)(t % h);
    const int b = (int)(t / h);
    float v[16];
    for (int k = 0; k < 4 * C; k += 4) {
      const float4 q = Vec4<T>::load(z + i * 4 * C + k);
      v[k] = q.x; v[k + 1] = q.y; v[k + 2] = q.z; v[k + 3] = q.w;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float mx = -INFINITY, e[4], sum = 0.f;
      for (int c = 0; c < C; ++c) mx = fmaxf(mx, v[c * 4 + q]);
      for (int c = 0; c < C; ++c) { e[c] = expf(v[c * 4 + q] - mx); sum += e[c]; }
      float* o = p + ((((long long)b * 2 * h + 2 * hh + (q >> 1)) * 2 * w + 2 * ww + (q & 1)) * D + d) * C;
      for (int c = 0; c < C; ++c) o[c] = e[c] / sum;
    }
  }
}
template <typename T>
__global__ void final_softmax_bwd_kernel(const float* __restrict__ dp, const float* __restrict__ p, T* __restrict__ dz, int B,
                                         int h, int w, int D, int C) {
  const long long n = (long long)B * h * w * D;
  GRID_STRIDE(i, n) {
    const int d = (int)(i % D);
    long long t = i / D;
    const int ww = (int)(t % w); t /= w;
    const int hh = (int)(t % h);
    const int b = (int)(t / h);
    float v[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long long o = ((((long long)b * 2 * h + 2 * hh + (q >> 1)) * 2 * w + 2 * ww + (q & 1)) * D + d) * C;
      float dot = 0.f;
      for (int c = 0; c < C; ++c) dot += dp[o + c] * p[o + c];
      for (int c = 0; c < C; ++c) v[c * 4 + q] = p[o + c] * (dp[o + c] - dot);
    }
    for (int k = 0; k < 4 * C; k += 4) Vec4<T>::store(dz + i * 4 * C + k, make_float4(v[k], v[k + 1], v[k + 2], v[k + 3]));
  }
}
extern "C" int ltu_final_softmax_fwd(const void* z, float* p, int B, int h, int w, int D, int C, int dtype, ltu_stream_t s) {
  if (C < 1 || C > 4) return LTU_E_SHAPE;
  LTU_DISPATCH_T(dtype, { hipLaunchKernelGGL((final_softmax_fwd_kernel<T>), dim3(sgrid((long long)B * h * w * D)), dim3(256), 0, (hipStream_t)s, (const T*)z, p, B, h, w, D, C); });
  return ltu_check_launch();
}
extern "C" int ltu_final_softmax_bwd(const float* dp, const float* p, void* dz, int B, int h, int w, int D, int C, int dtype,
                                     ltu_stream_t s) {
  if (C < 1 || C > 4) return LTU_E_SHAPE;
  LTU_DISPATCH_T(dtype, { hipLaunchKernelGGL((final_softmax_bwd_kernel<T>), dim3(sgrid((long long)B * h * w * D)), dim3(256), 0, (hipStream_t)s, dp, p, (T*)dz, B, h, w, D, C); });
  return ltu_check_launch();
}

// eval branch (model/trans_3DUnet.py:199-202): one-hot of the arg-max class (first maximum)
__global__ void onehot_argmax_kernel(const float* __restrict__ p, float* __restrict__ o, long long M, int C) {
  GRID_STRIDE(m, M) {
    int best = 0;
    float bv = p[m * C];
    for (int c = 1; c < C; ++c)
      if (p[m * C + c] > bv) { bv = p[m * C + c]; best = c; }
    for (int c = 0; c < C; ++c) o[m * C + c] = c == best ? 1.f : 0.f;
  }
}
extern "C" int ltu_onehot_argmax(const float* p, float* o, long long M, int C, ltu_stream_t s) {
  hipLaunchKernelGGL(onehot_argmax_kernel, dim3(sgrid(M)), dim3(256), 0, (hipStream_t)s, p, o, M, C);
  return ltu_check_launch();
}

// ------------------------------------------------------------------------------------------------ attention gate
// model/Unet_3Dblock.py:217-221 + 1385:  a = sigmoid(psi . relu(IN(u1) + IN(u2)) + b);  out = skip * a
// u1,u2 [B][S][C] are the 1x1x1 conv outputs, sums1/sums2 their InstanceNorm sums ({shift,s1,s2} per (b,c)).
// A voxel row is handled by G = C/4 lanes.
#define IN_EPS 1e-5f
__device__ __forceinline__ void in_stat2(const float* sums, float invS, float& mean, float& rstd) {
  const float m1 = sums[1] * invS;
  const float var = fmaxf(sums[2] * invS - m1 * m1, 0.f);
  mean = sums[0] + m1;
  rstd = rsqrtf(var + IN_EPS);
}

template <typename T, int G>
__global__ void gate_fwd_kernel(const T* __restrict__ u1, const T* __restrict__ u2, const float* __restrict__ sums1,
                                const float* __restrict__ sums2, const float* __restrict__ psi_w, const float* __restrict__ psi_b,
                                const T* __restrict__ skip, float* __restrict__ a_out, T* __restrict__ out, int B, long long S) {
  const int C = G * 4;
  const int gl = threadIdx.x % G;
  const long long rows = (long long)B * S;
  const float invS = 1.f / (float)S;
  const float4 pw = *reinterpret_cast<const float4*>(psi_w + gl * 4);
  const float pb = psi_b[0];
  const long long rpb = blockDim.x / G;
  for (long long r0 = (long long)blockIdx.x * rpb; r0 < rows; r0 += (long long)gridDim.x * rpb) {
    const long long row = r0 + threadIdx.x / G;
    const bool ok = row < rows;
    float part = 0.f;
    float4 sk = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) {
      const int b = (int)(row / S);
      const long long e = row * C + gl * 4;
      const float4 a = Vec4<T>::load(u1 + e), c = Vec4<T>::load(u2 + e);
      sk = Vec4<T>::load(skip + e);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float m1, r1, m2, r2;
        in_stat2(sums1 + ((long long)b * C + gl * 4 + k) * 3, invS, m1, r1);
        in_stat2(sums2 + ((long long)b * C + gl * 4 + k) * 3, invS, m2, r2);
        const float r = fmaxf((f4at(a, k) - m1) * r1 + (f4at(c, k) - m2) * r2, 0.f);
        part += r * f4at(pw, k);
      }
    }
    const float sdot = group_sum<G>(part) + pb;
    const float av = 1.f / (1.f + expf(-sdot));
    if (ok) {
      if (gl == 0) a_out[row] = av;
      Vec4<T>::store(out + row * C + gl * 4, make_float4(sk.x * av, sk.y * av, sk.z * av, sk.w * av));
    }
  }
}

// backward stage 1: dskip = dout*a; ds = (sum_c dout*skip) a (1-a);  accumulate
//   dpsi_w[c] += ds*r_c, dpsi_b += ds, bs1[b][c] += {dr, dr*h1}, bs2[b][c] += {dr, dr*h2},  dr = ds*w_c*[r_c>0]
template <typename T, int G>
__global__ void gate_bwd_reduce_kernel(const T* __restrict__ dout, const T* __restrict__ u1, const T* __restrict__ u2,
                                       const float* __restrict__ sums1, const float* __restrict__ sums2,
                                       const float* __restrict__ psi_w, const T* __restrict__ skip, const float* __restrict__ a_in,
                                       T* __restrict__ dskip, float* __restrict__ ds_out, float* __restrict__ dpsi_w,
                                       float* __restrict__ dpsi_b, float* __restrict__ bs1, float* __restrict__ bs2, int B,
                                       long long S, int rows_per_block) {
  extern __shared__ float red[];   // [rowgroups][C][5]
  const int C = G * 4;
  const int gl = threadIdx.x % G, rg = threadIdx.x / G, nrg = blockDim.x / G;
  const int b = blockIdx.y;
  const float invS = 1.f / (float)S;
  const float4 pw = *reinterpret_cast<const float4*>(psi_w + gl * 4);
  float m1[4], r1[4], m2[4], r2[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    in_stat2(sums1 + ((long long)b * C + gl * 4 + k) * 3, invS, m1[k], r1[k]);
    in_stat2(sums2 + ((long long)b * C + gl * 4 + k) * 3, invS, m2[k], r2[k]);
  }
  float acc[4][5];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int q = 0; q < 5; ++q) acc[k][q] = 0.f;
  float accb = 0.f;
  const long long s0 = (long long)blockIdx.x * rows_per_block;
  long long s1 = s0 + rows_per_block;
  if (s1 > S) s1 = S;
  for (long long sb = s0; sb < s1; sb += nrg) {
    const long long sv = sb + rg;
    const bool ok = sv < s1;
    const long long row = (long long)b * S + sv;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f), sk = g, a = g, c = g;
    float av = 0.f;
    if (ok) {
      const long long e = row * C + gl * 4;
      g = Vec4<T>::load(dout + e);
      sk = Vec4<T>::load(skip + e);
      a = Vec4<T>::load(u1 + e);
      c = Vec4<T>::load(u2 + e);
      av = a_in[row];
    }
    const float da = group_sum<G>(g.x * sk.x + g.y * sk.y + g.z * sk.z + g.w * sk.w);
    const float ds = da * av * (1.f - av);
    if (ok) {
      Vec4<T>::store(dskip + row * C + gl * 4, make_float4(g.x * av, g.y * av, g.z * av, g.w * av));
      if (gl == 0) { ds_out[row] = ds; accb += ds; }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float h1 = (f4at(a, k) - m1[k]) * r1[k], h2 = (f4at(c, k) - m2[k]) * r2[k];
        const float r = fmaxf(h1 + h2, 0.f);
        const float dr = r > 0.f ? ds * f4at(pw, k) : 0.f;
        acc[k][0] += ds * r;
        acc[k][1] += dr;
        acc[k][2] += dr * h1;
        acc[k][3] += dr * h2;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float* dst = red + ((long long)rg * C + gl * 4 + k) * 5;
    dst[0] = acc[k][0]; dst[1] = acc[k][1]; dst[2] = acc[k][2]; dst[3] = acc[k][3];
    dst[4] = (k == 0 && gl == 0) ? accb : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * 5; i += blockDim.x) {
    float v = 0.f;
    for (int q = 0; q < nrg; ++q) v += red[(long long)q * C * 5 + i];
    const int c = i / 5, which = i % 5;
    if (which == 0) atomicAdd(dpsi_w + c, v);
    else if (which == 1) { atomicAdd(bs1 + ((long long)b * C + c) * 2, v); atomicAdd(bs2 + ((long long)b * C + c) * 2, v); }
    else if (which == 2) atomicAdd(bs1 + ((long long)b * C + c) * 2 + 1, v);
    else if (which == 3) atomicAdd(bs2 + ((long long)b * C + c) * 2 + 1, v);
    else if (c == 0) atomicAdd(dpsi_b, v);
  }
}

// backward stage 2: du1 = rstd1 (dr - bs1[0]/S - h1 bs1[1]/S), du2 likewise
template <typename T, int G>
__global__ void gate_bwd_apply_kernel(const T* __restrict__ u1, const T* __restrict__ u2, const float* __restrict__ sums1,
                                      const float* __restrict__ sums2, const float* __restrict__ psi_w,
                                      const float* __restrict__ ds_in, const float* __restrict__ bs1, const float* __restrict__ bs2,
                                      T* __restrict__ du1, T* __restrict__ du2, int B, long long S) {
  const int C = G * 4;
  const long long nvec = (long long)B * S * G;
  const float invS = 1.f / (float)S;
  GRID_STRIDE(i, nvec) {
    const int gl = (int)(i % G);
    const long long row = i / G;
    const int b = (int)(row / S);
    const float4 a = Vec4<T>::load(u1 + i * 4), c = Vec4<T>::load(u2 + i * 4);
    const float4 pw = *reinterpret_cast<const float4*>(psi_w + gl * 4);
    const float ds = ds_in[row];
    float4 o1, o2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float m1, r1, m2, r2;
      const long long bc = (long long)b * C + gl * 4 + k;
      in_stat2(sums1 + bc * 3, invS, m1, r1);
      in_stat2(sums2 + bc * 3, invS, m2, r2);
      const float h1 = (f4at(a, k) - m1) * r1, h2 = (f4at(c, k) - m2) * r2;
      const float dr = (h1 + h2) > 0.f ? ds * f4at(pw, k) : 0.f;
      f4at(o1, k) = r1 * (dr - bs1[bc * 2] * invS - h1 * bs1[bc * 2 + 1] * invS);
      f4at(o2, k) = r2 * (dr - bs2[bc * 2] * invS - h2 * bs2[bc * 2 + 1] * invS);
    }
    Vec4<T>::store(du1 + i * 4, o1);
    Vec4<T>::store(du2 + i * 4, o2);
  }
}

#define GATE_DISPATCH_G(C, ...)                               \
  do {                                                        \
    if ((C) == 8) { constexpr int G = 2; __VA_ARGS__ }        \
    else if ((C) == 16) { constexpr int G = 4; __VA_ARGS__ }  \
    else if ((C) == 32) { constexpr int G = 8; __VA_ARGS__ }  \
    else if ((C) == 64) { constexpr int G = 16; __VA_ARGS__ } \
    else if ((C) == 128) { constexpr int G = 32; __VA_ARGS__ } \
    else if ((C) == 256) { constexpr int G = 64; __VA_ARGS__ } \
    else return LTU_E_SHAPE;                                  \
  } while (0)

extern "C" int ltu_gate_fwd(const void* u1, const void* u2, const float* sums1, const float* sums2, const float* psi_w,
                            const float* psi_b, const void* skip, float* a_out, void* out, int B, long long S, int C, int dtype,
                            ltu_stream_t s) {
  LTU_DISPATCH_T(dtype, {
    GATE_DISPATCH_G(C, {
      const long long rows = (long long)B * S;
      hipLaunchKernelGGL((gate_fwd_kernel<T, G>), dim3(sgrid(rows, 256 / G)), dim3(256), 0, (hipStream_t)s, (const T*)u1,
                         (const T*)u2, sums1, sums2, psi_w, psi_b, (const T*)skip, a_out, (T*)out, B, S);
    });
  });
  return ltu_check_launch();
}

extern "C" int ltu_gate_bwd(const void* dout, const void* u1, const void* u2, const float* sums1, const float* sums2,
                            const float* psi_w, const void* skip, const float* a_in, void* dskip, float* ds_ws, float* dpsi_w,
                            float* dpsi_b, float* bs1, float* bs2, void* du1, void* du2, int B, long long S, int C, int dtype,
                            ltu_stream_t s) {
  LTU_DISPATCH_T(dtype, {
    GATE_DISPATCH_G(C, {
      const int nrg = 256 / G;
      long long want = 2048 / (B > 0 ? B : 1);
      if (want < 1) want = 1;
      long long rows = (S + want - 1) / want;
      if (rows < nrg) rows = nrg;
      rows = (rows + nrg - 1) / nrg * nrg;
      const size_t lds = (size_t)nrg * C * 5 * sizeof(float);
      hipLaunchKernelGGL((gate_bwd_reduce_kernel<T, G>), dim3(cdiv(S, rows), B), dim3(256), lds, (hipStream_t)s, (const T*)dout,
                         (const T*)u1, (const T*)u2, sums1, sums2, psi_w, (const T*)skip, a_in, (T*)dskip, ds_ws, dpsi_w, dpsi_b,
                         bs1, bs2, B, S, (int)rows);
      hipLaunchKernelGGL((gate_bwd_apply_kernel<T, G>), dim3(sgrid((long long)B * S * G)), dim3(256), 0, (hipStream_t)s,
                         (const T*)u1, (const T*)u2, sums1, sums2, psi_w, ds_ws, bs1, bs2, (T*)du1, (T*)du2, B, S);
    });
  });
  return ltu_check_launch();
}

// ------------------------------------------------------------------------------------------------ positional conv
// model/trans_block.py:94-96 on the grid of Unet_3Dblock.py:267-270: y = chandrop(x + dwconv3x3x3(x) + bias).
// x [B,H,W,D,C]; the reference runs the conv on a [B,C,D,H,W] view, so weight w[c][kd][kh][kw] multiplies the
// neighbour at offset (kh-1, kw-1, kd-1) of the (H,W,D) lattice.  Channel dropout (nn.Dropout3d) draws one
// keep/drop per (sample, channel).
__device__ __forceinline__ void dw_tap(int t, int& dh, int& dw, int& dd) {
  dd = t / 9 - 1;
  dh = (t / 3) % 3 - 1;
  dw = t % 3 - 1;
}

template <typename T>
__global__ void dwconv_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                  T* __restrict__ y, int B, int H, int W, int D, int C, float p, uint64_t seed, const uint64_t* step) {
  extern __shared__ float wl[];   // [27][C]
  for (int i = threadIdx.x; i < 27 * C; i += blockDim.x) wl[(i % 27) * C + i / 27] = w[i];
  __syncthreads();
  const int cv = C / 4;
  const long long n = (long long)B * H * W * D * cv;
  const DropCfg dc = make_drop(p, seed, step);
  GRID_STRIDE(i, n) {
    const int v = (int)(i % cv);
    long long t = i / cv;
    const int d = (int)(t % D); t /= D;
    const int ww = (int)(t % W); t /= W;
    const int hh = (int)(t % H);
    const int b = (int)(t / H);
    float4 acc = *reinterpret_cast<const float4*>(bias + v * 4);
    for (int tp = 0; tp < 27; ++tp) {
      int dh, dw, dd;
      dw_tap(tp, dh, dw, dd);
      const int h2 = hh + dh, w2 = ww + dw, d2 = d + dd;
      if ((unsigned)h2 >= (unsigned)H || (unsigned)w2 >= (unsigned)W || (unsigned)d2 >= (unsigned)D) continue;
      const float4 q = Vec4<T>::load(x + ((((long long)b * H + h2) * W + w2) * D + d2) * C + v * 4);
      const float4 k = *reinterpret_cast<const float4*>(&wl[tp * C + v * 4]);
      acc.x += q.x * k.x; acc.y += q.y * k.y; acc.z += q.z * k.z; acc.w += q.w * k.w;
    }
    const float4 c0 = Vec4<T>::load(x + i * 4);
    acc.x += c0.x; acc.y += c0.y; acc.z += c0.z; acc.w += c0.w;
    Vec4<T>::store(y + i * 4, drop4(dc, (uint64_t)(((long long)b * C + v * 4) >> 2), acc));
  }
}

// dx = g' + sum_t w[t] g'(vox - off_t),  g' = dy * chanmask
template <typename T>
__global__ void dwconv_bwd_data_kernel(const T* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx, int B, int H,
                                       int W, int D, int C, float p, uint64_t seed, const uint64_t* step) {
  extern __shared__ float wl[];
  for (int i = threadIdx.x; i < 27 * C; i += blockDim.x) wl[(i % 27) * C + i / 27] = w[i];
  __syncthreads();
  const int cv = C / 4;
  const long long n = (long long)B * H * W * D * cv;
  const DropCfg dc = make_drop(p, seed, step);
  GRID_STRIDE(i, n) {
    const int v = (int)(i % cv);
    long long t = i / cv;
    const int d = (int)(t % D); t /= D;
    const int ww = (int)(t % W); t /= W;
    const int hh = (int)(t % H);
    const int b = (int)(t / H);
    float4 acc = Vec4<T>::load(dy + i * 4);
    for (int tp = 0; tp < 27; ++tp) {
      int dh, dw, dd;
      dw_tap(tp, dh, dw, dd);
      const int h2 = hh - dh, w2 = ww - dw, d2 = d - dd;
      if ((unsigned)h2 >= (unsigned)H || (unsigned)w2 >= (unsigned)W || (unsigned)d2 >= (unsigned)D) continue;
      const float4 q = Vec4<T>::load(dy + ((((long long)b * H + h2) * W + w2) * D + d2) * C + v * 4);
      const float4 k = *reinterpret_cast<const float4*>(&wl[tp * C + v * 4]);
      acc.x += q.x * k.x; acc.y += q.y * k.y; acc.z += q.z * k.z; acc.w += q.w * k.w;
    }
    const float4 m = dropmask4(dc, (uint64_t)(((long long)b * C + v * 4) >> 2));
    Vec4<T>::store(dx + i * 4, make_float4(acc.x * m.x, acc.y * m.y, acc.z * m.z, acc.w * m.w));
  }
}

// dw[c][t] += sum_vox g'[vox][c] x[vox+off_t][c];  db[c] += sum g'.   grid (chunks, B); a thread owns 4 channels.
template <typename T>
__global__ void dwconv_bwd_weight_kernel(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ dwt,
                                         float* __restrict__ db, int B, int H, int W, int D, int C, int rows_per_block, float p,
                                         uint64_t seed, const uint64_t* step) {
  extern __shared__ float red[];   // [C][28]
  for (int i = threadIdx.x; i < C * 28; i += blockDim.x) red[i] = 0.f;
  __syncthreads();
  const int cv = C / 4;
  const int v = threadIdx.x % cv, rg = threadIdx.x / cv, nrg = blockDim.x / cv;
  const int b = blockIdx.y;
  const DropCfg dc = make_drop(p, seed, step);
  const float4 m = dropmask4(dc, (uint64_t)(((long long)b * C + v * 4) >> 2));
  float acc[28][4];
#pragma unroll
  for (int t = 0; t < 28; ++t)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[t][k] = 0.f;
  const long long S = (long long)H * W * D;
  const long long s0 = (long long)blockIdx.x * rows_per_block;
  long long s1 = s0 + rows_per_block;
  if (s1 > S) s1 = S;
  if (rg < nrg) {
    for (long long sv = s0 + rg; sv < s1; sv += nrg) {
      const int d = (int)(sv % D);
      const int ww = (int)((sv / D) % W);
      const int hh = (int)(sv / ((long long)D * W));
      float4 g = Vec4<T>::load(dy + ((long long)b * S + sv) * C + v * 4);
      g.x *= m.x; g.y *= m.y; g.z *= m.z; g.w *= m.w;
      acc[27][0] += g.x; acc[27][1] += g.y; acc[27][2] += g.z; acc[27][3] += g.w;
#pragma unroll
      for (int tp = 0; tp < 27; ++tp) {
        int dh, dw, dd;
        dw_tap(tp, dh, dw, dd);
        const int h2 = hh + dh, w2 = ww + dw, d2 = d + dd;
        if ((unsigned)h2 >= (unsigned)H || (unsigned)w2 >= (unsigned)W || (unsigned)d2 >= (unsigned)D) continue;
        const float4 q = Vec4<T>::load(x + ((((long long)b * H + h2) * W + w2) * D + d2) * C + v * 4);
        acc[tp][0] += g.x * q.x; acc[tp][1] += g.y * q.y; acc[tp][2] += g.z * q.z; acc[tp][3] += g.w * q.w;
      }
    }
#pragma unroll
    for (int tp = 0; tp < 28; ++tp)
#pragma unroll
      for (int k = 0; k < 4; ++k) atomicAdd(&red[(v * 4 + k) * 28 + tp], acc[tp][k]);   // LDS atomics
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * 28; i += blockDim.x) {
    const int c = i / 28, tp = i % 28;
    if (tp < 27) atomicAdd(dwt + (long long)c * 27 + tp, red[i]);
    else atomicAdd(db + c, red[i]);
  }
}

extern "C" int ltu_dwconv_fwd(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int D, int C,
                              float p, uint64_t seed, const uint64_t* step, int dtype, ltu_stream_t s) {
  if (C % 4) return LTU_E_SHAPE;
  const long long n = (long long)B * H * W * D * (C / 4);
  LTU_DISPATCH_T(dtype, { hipLaunchKernelGGL((dwconv_fwd_kernel<T>), dim3(sgrid(n)), dim3(256), 27 * C * sizeof(float), (hipStream_t)s, (const T*)x, w, bias, (T*)y, B, H, W, D, C, p, seed, step); });
  return ltu_check_launch();
}
extern "C" int ltu_dwconv_bwd(const void* dy, const void* x, const float* w, void* dx, float* dwt, float* db, int B, int H,
                              int W, int D, int C, float p, uint64_t seed, const uint64_t* step, int dtype, ltu_stream_t s) {
  if (C % 4 || 256 % (C / 4)) return LTU_E_SHAPE;
  const long long n = (long long)B * H * W * D * (C / 4);
  const long long S = (long long)H * W * D;
  long long want = 512 / (B > 0 ? B : 1);
  if (want < 1) want = 1;
  long long rows = (S + want - 1) / want;
  if (rows < 32) rows = 32;
  LTU_DISPATCH_T(dtype, {
    hipLaunchKernelGGL((dwconv_bwd_data_kernel<T>), dim3(sgrid(n)), dim3(256), 27 * C * sizeof(float), (hipStream_t)s, (const T*)dy, w, (T*)dx, B, H, W, D, C, p, seed, step);
    hipLaunchKernelGGL((dwconv_bwd_weight_kernel<T>), dim3(cdiv(S, rows), B), dim3(256), (size_t)C * 28 * sizeof(float), (hipStream_t)s, (const T*)dy, (const T*)x, dwt, db, B, H, W, D, C, (int)rows, p, seed, step);
  });
  return ltu_check_launch();
}
