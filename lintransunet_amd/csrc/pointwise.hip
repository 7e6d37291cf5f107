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
// bf16 fast path: 16-byte accesses (8 elements), two independent loads in flight per thread.  The dropout groups are the same
// 4-element groups as above (2i and 2i+1), so either kernel regenerates the other's mask.
__device__ __forceinline__ void unpack8(const uint4& r, float4& a, float4& b) {
  a = make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
  b = make_float4(__uint_as_float(r.z << 16), __uint_as_float(r.z & 0xffff0000u), __uint_as_float(r.w << 16), __uint_as_float(r.w & 0xffff0000u));
}
__device__ __forceinline__ uint4 pack8(const float4& a, const float4& b) {
  return make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
}
template <bool BWD>
__device__ __forceinline__ uint4 gelu8(const DropCfg& dc, long long i, const uint4& ur, const uint4& gr) {
  float4 a, b;
  unpack8(ur, a, b);
  if (!BWD) {
    a = make_float4(gelu_erf(a.x), gelu_erf(a.y), gelu_erf(a.z), gelu_erf(a.w));
    b = make_float4(gelu_erf(b.x), gelu_erf(b.y), gelu_erf(b.z), gelu_erf(b.w));
    return pack8(drop4(dc, (uint64_t)(2 * i), a), drop4(dc, (uint64_t)(2 * i + 1), b));
  }
  float4 ga, gb;
  unpack8(gr, ga, gb);
  const float4 ma = dropmask4(dc, (uint64_t)(2 * i)), mb = dropmask4(dc, (uint64_t)(2 * i + 1));
  a = make_float4(ga.x * ma.x * gelu_erf_grad(a.x), ga.y * ma.y * gelu_erf_grad(a.y), ga.z * ma.z * gelu_erf_grad(a.z),
                  ga.w * ma.w * gelu_erf_grad(a.w));
  b = make_float4(gb.x * mb.x * gelu_erf_grad(b.x), gb.y * mb.y * gelu_erf_grad(b.y), gb.z * mb.z * gelu_erf_grad(b.z),
                  gb.w * mb.w * gelu_erf_grad(b.w));
  return pack8(a, b);
}
template <bool BWD>
__global__ void __launch_bounds__(256) gelu_drop_bf16x8_kernel(const uint4* __restrict__ u, const uint4* __restrict__ dh,
                                                               uint4* __restrict__ out, long long n8, float p, uint64_t seed,
                                                               const uint64_t* step) {
  const DropCfg dc = make_drop(p, seed, step);
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (; i + stride < n8; i += 2 * stride) {
    const uint4 u0 = u[i], u1 = u[i + stride];
    const uint4 g0 = BWD ? dh[i] : z, g1 = BWD ? dh[i + stride] : z;
    out[i] = gelu8<BWD>(dc, i, u0, g0);
    out[i + stride] = gelu8<BWD>(dc, i + stride, u1, g1);
  }
  if (i < n8) out[i] = gelu8<BWD>(dc, i, u[i], BWD ? dh[i] : z);
}
static unsigned gelu8_grid(long long n8) {
  int per = -1;
  per = ltu_knob("LTU_GELU_PER_BLOCK", 512) >= 256 ? ltu_knob("LTU_GELU_PER_BLOCK", 512) : 512;
  long long blocks = (n8 + per - 1) / per;      // 512: two 16-byte pieces per thread
  if (blocks > 16384) blocks = 16384;
  return (unsigned)(blocks < 1 ? 1 : blocks);
}

extern "C" int ltu_gelu_dropout_fwd(const void* u, void* h, long long n, float p, uint64_t seed, const uint64_t* step, int dtype, ltu_stream_t s) {
  if (n % 4) return LTU_E_SHAPE;
  if (dtype == LTU_BF16 && n % 8 == 0 && (((uintptr_t)u | (uintptr_t)h) & 15) == 0) {
    hipLaunchKernelGGL((gelu_drop_bf16x8_kernel<false>), dim3(gelu8_grid(n / 8)), dim3(256), 0, (hipStream_t)s, (const uint4*)u,
                       (const uint4*)u, (uint4*)h, n / 8, p, seed, step);
    return ltu_check_launch();
  }
  LTU_DISPATCH_T(dtype, { hipLaunchKernelGGL((gelu_drop_fwd_kernel<T>), dim3(sgrid(n / 4)), dim3(256), 0, (hipStream_t)s, (const T*)u, (T*)h, n / 4, p, seed, step); });
  return ltu_check_launch();
}
extern "C" int ltu_gelu_dropout_bwd(const void* dh, const void* u, void* du, long long n, float p, uint64_t seed, const uint64_t* step, int dtype,
                                    ltu_stream_t s) {
  if (n % 4) return LTU_E_SHAPE;
  if (dtype == LTU_BF16 && n % 8 == 0 && (((uintptr_t)u | (uintptr_t)dh | (uintptr_t)du) & 15) == 0) {
    hipLaunchKernelGGL((gelu_drop_bf16x8_kernel<true>), dim3(gelu8_grid(n / 8)), dim3(256), 0, (hipStream_t)s, (const uint4*)u,
                       (const uint4*)dh, (uint4*)du, n / 8, p, seed, step);
    return ltu_check_launch();
  }
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
// compile-time class count, CP % 4 == 0: the row's first four logits as ONE vector load (instead of C two-byte loads in a run-time
// loop), the probabilities as one store for C = 2
template <typename T, int C>
__global__ void head_softmax_fwd_vec_kernel(const T* __restrict__ z, float* __restrict__ p, long long M, int CP) {
  GRID_STRIDE(m, M) {
    const float4 q = Vec4<T>::load(z + m * CP);
    const float v[4] = {q.x, q.y, q.z, q.w};
    float mx = -INFINITY, e[C], sum = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, v[c]);
#pragma unroll
    for (int c = 0; c < C; ++c) { e[c] = expf(v[c] - mx); sum += e[c]; }
    if constexpr (C == 2) *reinterpret_cast<float2*>(p + m * 2) = make_float2(e[0] / sum, e[1] / sum);
    else {
#pragma unroll
      for (int c = 0; c < C; ++c) p[m * C + c] = e[c] / sum;
    }
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
// bf16 rows of CP = 8k padded logits: one 16-byte store per 8 channels (the fused conv pairs pad the heads to 16 / 32 columns)
template <int CT>
__global__ void head_softmax_bwd_bf16x8_kernel(const float* __restrict__ dp, const float* __restrict__ p, uint4* __restrict__ dz,
                                               long long M, int C_rt, int CP) {
  const int C = CT > 0 ? CT : C_rt;
  const int v8 = CP / 8;
  GRID_STRIDE(i, M * v8) {
    const long long m = i / v8;
    uint4 o = make_uint4(0u, 0u, 0u, 0u);
    if (i - m * v8 == 0) {
      float dot = 0.f, g[4] = {0.f, 0.f, 0.f, 0.f}, a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (CT == 2) {
        const float2 av = *reinterpret_cast<const float2*>(dp + m * 2), bv = *reinterpret_cast<const float2*>(p + m * 2);
        a[0] = av.x; a[1] = av.y; b[0] = bv.x; b[1] = bv.y;
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < C) { a[c] = dp[m * C + c]; b[c] = p[m * C + c]; }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) if (c < C) dot += a[c] * b[c];
#pragma unroll
      for (int c = 0; c < 4; ++c) if (c < C) g[c] = b[c] * (a[c] - dot);
      o.x = pack_bf16x2(g[0], g[1]);
      o.y = pack_bf16x2(g[2], g[3]);
    }
    dz[i] = o;
  }
}
extern "C" int ltu_head_softmax_fwd(const void* z, float* p, long long M, int C, int CP, int dtype, ltu_stream_t s) {
  if (C > 4 || CP < C) return LTU_E_SHAPE;
  LTU_DISPATCH_T(dtype, {
    if (CP % 4 == 0 && C == 2) hipLaunchKernelGGL((head_softmax_fwd_vec_kernel<T, 2>), dim3(sgrid(M)), dim3(256), 0, (hipStream_t)s, (const T*)z, p, M, CP);
    else if (CP % 4 == 0 && C == 3) hipLaunchKernelGGL((head_softmax_fwd_vec_kernel<T, 3>), dim3(sgrid(M)), dim3(256), 0, (hipStream_t)s, (const T*)z, p, M, CP);
    else hipLaunchKernelGGL((head_softmax_fwd_kernel<T>), dim3(sgrid(M)), dim3(256), 0, (hipStream_t)s, (const T*)z, p, M, C, CP);
  });
  return ltu_check_launch();
}
extern "C" int ltu_head_softmax_bwd(const float* dp, const float* p, void* dz, long long M, int C, int CP, int dtype,
                                    ltu_stream_t s) {
  if (C > 4 || CP < C) return LTU_E_SHAPE;
  if (dtype == LTU_BF16 && CP % 8 == 0 && ((uintptr_t)dz & 15) == 0) {
    if (C == 2) hipLaunchKernelGGL(head_softmax_bwd_bf16x8_kernel<2>, dim3(sgrid(M * (CP / 8))), dim3(256), 0, (hipStream_t)s, dp, p, (uint4*)dz, M, C, CP);
    else if (C == 3) hipLaunchKernelGGL(head_softmax_bwd_bf16x8_kernel<3>, dim3(sgrid(M * (CP / 8))), dim3(256), 0, (hipStream_t)s, dp, p, (uint4*)dz, M, C, CP);
    else hipLaunchKernelGGL(head_softmax_bwd_bf16x8_kernel<0>, dim3(sgrid(M * (CP / 8))), dim3(256), 0, (hipStream_t)s, dp, p, (uint4*)dz, M, C, CP);
    return ltu_check_launch();
  }
  LTU_DISPATCH_T(dtype, { hipLaunchKernelGGL((head_softmax_bwd_kernel<T>), dim3(sgrid(M)), dim3(256), 0, (hipStream_t)s, dp, p, (T*)dz, M, C, CP); });
  return ltu_check_launch();
}

// final head (model/Unet_3Dblock.py:1392-1394): z T [B,h,w,D,4C] -> window un-embedding + softmax over classes
// -> probs f32 [B,2h,2w,D,C];  channel c*4 + kh*2 + kw of voxel (h,w) is class c of voxel (2h+kh, 2w+kw).
// CT: compile-time class count (0 = run-time C).  With a run-time C the class loops stay loops: one scalar load per trip, each
// waited for where it stands (16 dependent round trips per thread in the backward kernel).
template <typename T, int CT>
__global__ void final_softmax_fwd_kernel(const T* __restrict__ z, float* __restrict__ p, int B, int h, int w, int D, int C_rt, int CP) {
  const int C = CT > 0 ? CT : C_rt;
  const long long n = (long long)B * h * w * D;
  GRID_STRIDE(i, n) {
    const int d = (int)(i % D);
    long long t = i / D;
    const int ww = (int)(t % w); t /= w;
    const int hh = (int)(t % h);
    const int b = (int)(t / h);
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; k += 4) {
      if (k < 4 * C) {
        const float4 q = Vec4<T>::load(z + i * CP + k);
        v[k] = q.x; v[k + 1] = q.y; v[k + 2] = q.z; v[k + 3] = q.w;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float mx = -INFINITY, e[4], sum = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) if (c < C) mx = fmaxf(mx, v[c * 4 + q]);
#pragma unroll
      for (int c = 0; c < 4; ++c) if (c < C) { e[c] = expf(v[c * 4 + q] - mx); sum += e[c]; }
      float* o = p + ((((long long)b * 2 * h + 2 * hh + (q >> 1)) * 2 * w + 2 * ww + (q & 1)) * D + d) * C;
      if constexpr (CT == 2) *reinterpret_cast<float2*>(o) = make_float2(e[0] / sum, e[1] / sum);
      else {
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < C) o[c] = e[c] / sum;
      }
    }
  }
}
template <typename T, int CT>
__global__ void final_softmax_bwd_kernel(const float* __restrict__ dp, const float* __restrict__ p, T* __restrict__ dz, int B,
                                         int h, int w, int D, int C_rt, int CP) {
  const int C = CT > 0 ? CT : C_rt;
  const long long n = (long long)B * h * w * D;
  GRID_STRIDE(i, n) {
    const int d = (int)(i % D);
    long long t = i / D;
    const int ww = (int)(t % w); t /= w;
    const int hh = (int)(t % h);
    const int b = (int)(t / h);
    float v[16], gv[4][4], pv[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {                 // all loads of the four fine voxels first
      const long long o = ((((long long)b * 2 * h + 2 * hh + (q >> 1)) * 2 * w + 2 * ww + (q & 1)) * D + d) * C;
      if constexpr (CT == 2) {
        const float2 a = *reinterpret_cast<const float2*>(dp + o), c2 = *reinterpret_cast<const float2*>(p + o);
        gv[q][0] = a.x; gv[q][1] = a.y; pv[q][0] = c2.x; pv[q][1] = c2.y;
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < C) { gv[q][c] = dp[o + c]; pv[q][c] = p[o + c]; }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) if (c < C) dot += gv[q][c] * pv[q][c];
#pragma unroll
      for (int c = 0; c < 4; ++c) if (c < C) v[c * 4 + q] = pv[q][c] * (gv[q][c] - dot);
    }
#pragma unroll
    for (int k = 0; k < 16; k += 4)
      if (k < 4 * C) Vec4<T>::store(dz + i * CP + k, make_float4(v[k], v[k + 1], v[k + 2], v[k + 3]));
    for (int k = 4 * C; k < CP; k += 4) Vec4<T>::store(dz + i * CP + k, make_float4(0.f, 0.f, 0.f, 0.f));      // padded conv columns
  }
}
extern "C" int ltu_final_softmax_fwd(const void* z, float* p, int B, int h, int w, int D, int C, int CP, int dtype, ltu_stream_t s) {
  if (C < 1 || C > 4 || CP < 4 * C || CP % 4) return LTU_E_SHAPE;
  LTU_DISPATCH_T(dtype, {
    const dim3 grid(sgrid((long long)B * h * w * D));
    if (C == 2) hipLaunchKernelGGL((final_softmax_fwd_kernel<T, 2>), grid, dim3(256), 0, (hipStream_t)s, (const T*)z, p, B, h, w, D, C, CP);
    else if (C == 3) hipLaunchKernelGGL((final_softmax_fwd_kernel<T, 3>), grid, dim3(256), 0, (hipStream_t)s, (const T*)z, p, B, h, w, D, C, CP);
    else hipLaunchKernelGGL((final_softmax_fwd_kernel<T, 0>), grid, dim3(256), 0, (hipStream_t)s, (const T*)z, p, B, h, w, D, C, CP);
  });
  return ltu_check_launch();
}
extern "C" int ltu_final_softmax_bwd(const float* dp, const float* p, void* dz, int B, int h, int w, int D, int C, int CP,
                                     int dtype, ltu_stream_t s) {
  if (C < 1 || C > 4 || CP < 4 * C || CP % 4) return LTU_E_SHAPE;
  LTU_DISPATCH_T(dtype, {
    const dim3 grid(sgrid((long long)B * h * w * D));
    if (C == 2) hipLaunchKernelGGL((final_softmax_bwd_kernel<T, 2>), grid, dim3(256), 0, (hipStream_t)s, dp, p, (T*)dz, B, h, w, D, C, CP);
    else if (C == 3) hipLaunchKernelGGL((final_softmax_bwd_kernel<T, 3>), grid, dim3(256), 0, (hipStream_t)s, dp, p, (T*)dz, B, h, w, D, C, CP);
    else hipLaunchKernelGGL((final_softmax_bwd_kernel<T, 0>), grid, dim3(256), 0, (hipStream_t)s, dp, p, (T*)dz, B, h, w, D, C, CP);
  });
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
                                       float* __restrict__ dpsi_b, float* __restrict__ bs1, float* __restrict__ bs2,
                                       float* __restrict__ ws, int B, long long S, int rows_per_block) {
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
    if (ws != nullptr) {      // two-stage: plain store of the block's partial, folded by gate_fold_kernel
      ws[((long long)b * gridDim.x + blockIdx.x) * (C * 5) + i] = v;
      continue;
    }
    const int c = i / 5, which = i % 5;
    if (which == 0) atomicAdd(dpsi_w + c, v);
    else if (which == 1) { atomicAdd(bs1 + ((long long)b * C + c) * 2, v); atomicAdd(bs2 + ((long long)b * C + c) * 2, v); }
    else if (which == 2) atomicAdd(bs1 + ((long long)b * C + c) * 2 + 1, v);
    else if (which == 3) atomicAdd(bs2 + ((long long)b * C + c) * 2 + 1, v);
    else if (c == 0) atomicAdd(dpsi_b, v);
  }
}

// folds the per-block partials ws[b][part][C*5] of gate_bwd_reduce_kernel (a few thousand blocks adding to the same handful of
// addresses with atomics is a serial chain at the memory side).  grid (cdiv(C*5,32), B), 1024 threads.
__global__ void __launch_bounds__(1024) gate_fold_kernel(const float* __restrict__ ws, int nparts, int C, float* __restrict__ dpsi_w,
                                                         float* __restrict__ dpsi_b, float* __restrict__ bs1,
                                                         float* __restrict__ bs2) {
  __shared__ float red[32][33];
  const int n = C * 5;
  const int el = threadIdx.x & 31, zq = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + el, b = blockIdx.y;
  float a0 = 0.f, a1 = 0.f;
  if (i < n) {
    const float* p = ws + (long long)b * nparts * n + i;
    int z = zq;
    for (; z + 32 < nparts; z += 64) { a0 += p[(long long)z * n]; a1 += p[(long long)(z + 32) * n]; }
    for (; z < nparts; z += 32) a0 += p[(long long)z * n];
  }
  red[zq][el] = a0 + a1;
  __syncthreads();
  if (zq != 0 || i >= n) return;
  float v = 0.f;
#pragma unroll
  for (int q = 0; q < 32; ++q) v += red[q][el];
  const int c = i / 5, which = i % 5;
  if (which == 0) atomicAdd(dpsi_w + c, v);
  else if (which == 1) { bs1[((long long)b * C + c) * 2] += v; bs2[((long long)b * C + c) * 2] += v; }
  else if (which == 2) bs1[((long long)b * C + c) * 2 + 1] += v;
  else if (which == 3) bs2[((long long)b * C + c) * 2 + 1] += v;
  else if (c == 0) atomicAdd(dpsi_b, v);
}

// backward stage 2: du1 = rstd1 (dr - bs1[0]/S - h1 bs1[1]/S), du2 likewise.  grid (blocks, B); the loop stride is a multiple of G,
// so a thread keeps its channel quad and the 20 per-channel constants (two statistics, two backward sums, psi) are computed once -
// recomputed per vector (8 rsqrt, 28 cached loads per 4 elements) this pass ran at 2.6 TB/s, VALU-bound.
template <typename T, int G>
__global__ void __launch_bounds__(256) gate_bwd_apply_kernel(const T* __restrict__ u1, const T* __restrict__ u2, const float* __restrict__ sums1,
                                      const float* __restrict__ sums2, const float* __restrict__ psi_w,
                                      const float* __restrict__ ds_in, const float* __restrict__ bs1, const float* __restrict__ bs2,
                                      T* __restrict__ du1, T* __restrict__ du2, int B, long long S) {
  constexpr int C = G * 4;
  const int b = blockIdx.y;
  const int gl = threadIdx.x % G;
  const float invS = 1.f / (float)S;
  const float4 pw = *reinterpret_cast<const float4*>(psi_w + gl * 4);
  float m1[4], r1[4], m2[4], r2[4], a1[4], b1[4], a2[4], b2[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long bc = (long long)b * C + gl * 4 + k;
    in_stat2(sums1 + bc * 3, invS, m1[k], r1[k]);
    in_stat2(sums2 + bc * 3, invS, m2[k], r2[k]);
    a1[k] = bs1[bc * 2] * invS; b1[k] = bs1[bc * 2 + 1] * invS;
    a2[k] = bs2[bc * 2] * invS; b2[k] = bs2[bc * 2 + 1] * invS;
  }
  const long long per_b = S * G, base = (long long)b * per_b;
  for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < per_b; j += (long long)gridDim.x * 256) {
    const long long i = base + j;
    const float4 a = Vec4<T>::load(u1 + i * 4), c = Vec4<T>::load(u2 + i * 4);
    const float ds = ds_in[i / G];
    float4 o1, o2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float h1 = (f4at(a, k) - m1[k]) * r1[k], h2 = (f4at(c, k) - m2[k]) * r2[k];
      const float dr = (h1 + h2) > 0.f ? ds * f4at(pw, k) : 0.f;
      f4at(o1, k) = r1[k] * (dr - a1[k] - h1 * b1[k]);
      f4at(o2, k) = r2[k] * (dr - a2[k] - h2 * b2[k]);
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
                            float* dpsi_b, float* bs1, float* bs2, float* ws, long long ws_floats, void* du1, void* du2, int B, long long S,
                            int C, int dtype, ltu_stream_t s) {
  if (ws != nullptr && ws_floats < ltu_norm_ws_floats()) return LTU_E_ARG;
  // ws: ltu_norm_ws_floats() floats of workspace for the two-stage reduction, or NULL (atomics)
  LTU_DISPATCH_T(dtype, {
    GATE_DISPATCH_G(C, {
      const int nrg = 256 / G;
      long long cap = (1 << 20) / (C * 5);
      if (cap > 2048) cap = 2048;
      long long want = cap / (B > 0 ? B : 1);
      if (want < 1) want = 1;
      long long rows = (S + want - 1) / want;
      if (rows < nrg) rows = nrg;
      rows = (rows + nrg - 1) / nrg * nrg;
      const size_t lds = (size_t)nrg * C * 5 * sizeof(float);
      hipLaunchKernelGGL((gate_bwd_reduce_kernel<T, G>), dim3(cdiv(S, rows), B), dim3(256), lds, (hipStream_t)s, (const T*)dout,
                         (const T*)u1, (const T*)u2, sums1, sums2, psi_w, (const T*)skip, a_in, (T*)dskip, ds_ws, dpsi_w, dpsi_b,
                         bs1, bs2, ws, B, S, (int)rows);
      if (ws != nullptr)
        hipLaunchKernelGGL(gate_fold_kernel, dim3(cdiv(C * 5, 32), B), dim3(1024), 0, (hipStream_t)s, ws, (int)cdiv(S, rows), C,
                           dpsi_w, dpsi_b, bs1, bs2);
      long long ablocks = (S * G + 255) / 256;
      const long long acap = 4096 / (B > 0 ? B : 1) > 1 ? 4096 / (B > 0 ? B : 1) : 1;
      if (ablocks > acap) ablocks = acap;
      hipLaunchKernelGGL((gate_bwd_apply_kernel<T, G>), dim3((unsigned)ablocks, B), dim3(256), 0, (hipStream_t)s,
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
// tap t = kd*9 + kh*3 + kw of the reference weight reads the neighbour (kh-1, kw-1, kd-1).

// A workgroup walks 4x4x8 bricks of output voxels for one 128-byte channel chunk (64 bf16 / 32 fp32
// channels): the 6x6x10 halo brick is staged in LDS once and the 27 taps are LDS reads at constant offsets (27x fewer
// vector-memory requests than gathering from global, no per-tap address arithmetic).  A thread owns one channel quad and
// RUNS of 4 consecutive outputs along d: for each of the 9 (h, w) tap rows it reads the 6 halo values under the run once and
// uses each of them for up to three taps - 54 LDS reads and unpacks per 4 outputs instead of 108 (one output at a time the
// kernel was LDS-read- and unpack-bound at 0.09 of the HBM peak).  Its 27 x 4 weights (or weight-gradient accumulators) stay
// in registers across bricks.  Halo rows have a d pitch of 11 voxels: the four runs a wave reads together (2 w x 2 d-halves)
// then start on 128-byte rows of both parities, i.e. on both halves of the 64 LDS banks.
//   MODE 0: y = mask * (x + conv(x) + bias)      MODE 1: dx = mask * (dy + conv_flipped(dy))
//   MODE 2: dw[c][t] += sum g' x[v + off_t], db[c] += sum g'   (g' = mask * dy; block-level LDS reduction, then atomics)
#define DWH_VOX 360
#define DWH_PD 11           // d pitch of the LDS halo image (10 voxels + 1)
#define DWH_ROWS (36 * DWH_PD)
#define DWH_RUN 4
// 16 bytes of T (4 floats / 8 bf16) plus the same of a second tensor: the backward modes take the gradients of two consumers
// of the layer's output and sum them while staging (no stand-alone add pass)
template <typename T>
__device__ __forceinline__ uint4 add16(uint4 a, uint4 b);
template <>
__device__ __forceinline__ uint4 add16<float>(uint4 a, uint4 b) {
  return make_uint4(__float_as_uint(__uint_as_float(a.x) + __uint_as_float(b.x)), __float_as_uint(__uint_as_float(a.y) + __uint_as_float(b.y)),
                    __float_as_uint(__uint_as_float(a.z) + __uint_as_float(b.z)), __float_as_uint(__uint_as_float(a.w) + __uint_as_float(b.w)));
}
template <>
__device__ __forceinline__ uint4 add16<bf16_t>(uint4 a, uint4 b) {
  const uint32_t av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
  uint32_t o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    o[i] = pack_bf16x2(__uint_as_float(av[i] << 16) + __uint_as_float(bv[i] << 16),
                       __uint_as_float(av[i] & 0xffff0000u) + __uint_as_float(bv[i] & 0xffff0000u));
  return make_uint4(o[0], o[1], o[2], o[3]);
}
// x2 / g2 (nullable): second gradient tensors summed onto x (MODE 1) / g (MODE 2)
template <typename T, int MODE>
__global__ void __launch_bounds__(256, 2) dwconv_halo_kernel(const T* __restrict__ x, const T* __restrict__ x2, const T* __restrict__ g,
                                                          const T* __restrict__ g2,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          T* __restrict__ y, float* __restrict__ dwt, float* __restrict__ db, int B,
                                                          int H, int W, int D, int C, int bricks, int bricks_per_block, float p,
                                                          uint64_t seed, const uint64_t* step, float* __restrict__ part) {
  constexpr int CC = 128 / (int)sizeof(T);          // channels per chunk (one 128-byte LDS row per voxel)
  constexpr int QV = CC / 4;                        // channel quads per voxel
  constexpr int NV = 256 / QV;                      // run slots
  constexpr int R = DWH_RUN, PD = DWH_PD;
  constexpr int NJ = (128 / R) / NV;                // runs per thread and brick
  static_assert(NJ >= 1 && NJ * NV * R == 128, "runs tile the brick");
  __shared__ __attribute__((aligned(16))) T halo[DWH_ROWS * CC];
  __shared__ __attribute__((aligned(16))) T gl[MODE == 2 ? 128 * CC : 4];
  const int tid = threadIdx.x;
  const int cq = tid % QV, vs = tid / QV;
  const int c = blockIdx.x * CC + cq * 4;
  const bool cok = c < C;
  const int nbh = (H + 3) / 4, nbw = (W + 3) / 4, nbd = (D + 7) / 8;
  const DropCfg dc = make_drop(p, seed, step);
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);

  float4 wr[27];                                    // weights (MODE 0/1) or gradient accumulators (MODE 2)
  float4 bs = z;                                    // bias (MODE 0) or bias-gradient accumulator (MODE 2)
  if (MODE != 2 && cok) {                           // w[c..c+3][0..26] is one contiguous 432-byte run: 27 vector loads
    float flat[108];
#pragma unroll
    for (int i = 0; i < 27; ++i) {
      const float4 r = *reinterpret_cast<const float4*>(w + (long long)c * 27 + i * 4);
      flat[4 * i] = r.x; flat[4 * i + 1] = r.y; flat[4 * i + 2] = r.z; flat[4 * i + 3] = r.w;
    }
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const int ts = MODE == 1 ? 26 - t : t;
      wr[t] = make_float4(flat[ts], flat[27 + ts], flat[54 + ts], flat[81 + ts]);
    }
  } else {
#pragma unroll
    for (int t = 0; t < 27; ++t) wr[t] = z;
  }
  if (MODE == 0 && cok && bias != nullptr) bs = *reinterpret_cast<const float4*>(bias + c);

  int brick = blockIdx.y * bricks_per_block;
  int brick_end = brick + bricks_per_block;
  if (brick_end > bricks) brick_end = bricks;
  for (; brick < brick_end; ++brick) {
    int t = brick;
    const int bd = t % nbd; t /= nbd;
    const int bw = t % nbw; t /= nbw;
    const int bh = t % nbh;
    const int b = t / nbh;
    const int h0 = bh * 4, w0 = bw * 4, d0 = bd * 8;
    __syncthreads();                                // previous brick fully consumed
    // All of a thread's 16-byte pieces are requested before the first one is used, from clamped addresses (padding is selected
    // to zero on the way to LDS): as a loop with a test around each load every piece was a memory round trip of its own - 12 in a
    // row per brick, which is where this kernel's time went (0.7 TB/s).
    {
      constexpr int NH = (DWH_VOX * 8 + 255) / 256;
      constexpr int EPP = 16 / (int)sizeof(T);      // elements per piece
      constexpr int GP = MODE == 1 ? NH / 2 : NH;   // pieces per batch: two tensors (MODE 1) = two batches of 6 + 6 loads
      static_assert(NH % GP == 0, "whole batches");
      const bool two = MODE == 1 && x2 != nullptr;
#pragma unroll
      for (int p0 = 0; p0 < NH; p0 += GP) {
        uint4 hv1[GP], hv2[MODE == 1 ? GP : 1];
        unsigned inmask = 0;
#pragma unroll
        for (int q = 0; q < GP; ++q) {
          const int idx = tid + (p0 + q) * 256;
          const int hv = idx >> 3, part = idx & 7;  // 8 x 16 bytes per voxel row
          const int hd = hv % 10, hw = (hv / 10) % 6, hh = hv / 60;
          const int h = h0 - 1 + hh, ww = w0 - 1 + hw, d = d0 - 1 + hd;
          const int cc = blockIdx.x * CC + part * EPP;
          const bool in = idx < DWH_VOX * 8 && (unsigned)h < (unsigned)H && (unsigned)ww < (unsigned)W && (unsigned)d < (unsigned)D && cc < C;
          const long long off = in ? ((((long long)b * H + h) * W + ww) * D + d) * C + cc : 0;
          hv1[q] = *reinterpret_cast<const uint4*>(x + off);
          if (MODE == 1 && two) hv2[q] = *reinterpret_cast<const uint4*>(x2 + off);
          inmask |= in ? 1u << q : 0u;
        }
#pragma unroll
        for (int q = 0; q < GP; ++q) {
          const int idx = tid + (p0 + q) * 256;
          if (idx < DWH_VOX * 8) {
            const int hv = idx >> 3, part = idx & 7;
            const int hd = hv % 10, hw = (hv / 10) % 6, hh = hv / 60;
            uint4 v = hv1[q];
            if (MODE == 1 && two) v = add16<T>(v, hv2[q]);
            if (!((inmask >> q) & 1u)) v = make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4*>(reinterpret_cast<char*>(halo) + ((hh * 6 + hw) * PD + hd) * 128 + part * 16) = v;
          }
        }
      }
      if (MODE == 2) {                              // the brick's output gradients: 128 voxels x 128 bytes
        uint4 gv1[4], gv2[4];
        unsigned gmask = 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int idx = tid + p * 256;
          const int ov = idx >> 3, part = idx & 7;
          const int h = h0 + (ov >> 5), ww = w0 + ((ov >> 3) & 3), d = d0 + (ov & 7);
          const int cc = blockIdx.x * CC + part * EPP;
          const bool in = h < H && ww < W && d < D && cc < C;
          const long long off = in ? ((((long long)b * H + h) * W + ww) * D + d) * C + cc : 0;
          gv1[p] = *reinterpret_cast<const uint4*>(g + off);
          if (g2 != nullptr) gv2[p] = *reinterpret_cast<const uint4*>(g2 + off);
          gmask |= in ? 1u << p : 0u;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int idx = tid + p * 256;
          uint4 v = gv1[p];
          if (g2 != nullptr) v = add16<T>(v, gv2[p]);
          if (!((gmask >> p) & 1u)) v = make_uint4(0u, 0u, 0u, 0u);
          *reinterpret_cast<uint4*>(reinterpret_cast<char*>(gl) + (idx >> 3) * 128 + (idx & 7) * 16) = v;
        }
      }
    }
    __syncthreads();
    if (!cok) continue;
    const float4 m = dropmask4(dc, (uint64_t)(((long long)b * C + c) >> 2));
#pragma unroll 1
    for (int j = 0; j < NJ; ++j) {
      const int rn = vs + NV * j;                   // run (oh, ow, od0 .. od0 + 3)
      const int oh = rn >> 3, ow = (rn >> 1) & 3, od0 = (rn & 1) * R;
      const int h = h0 + oh, ww = w0 + ow, d = d0 + od0;
      const T* hp = halo + ((oh * 6 + ow) * PD + od0) * CC + cq * 4;
      const long long e = ((((long long)b * H + h) * W + ww) * D + d) * C + c;
      if (MODE == 2) {
        float4 gv[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          gv[r] = Vec4<T>::load(gl + ((oh << 5) + (ow << 3) + od0 + r) * CC + cq * 4);      // zero outside the volume
          gv[r].x *= m.x; gv[r].y *= m.y; gv[r].z *= m.z; gv[r].w *= m.w;
          bs.x += gv[r].x; bs.y += gv[r].y; bs.z += gv[r].z; bs.w += gv[r].w;
        }
#pragma unroll
        for (int th = 0; th < 3; ++th)
#pragma unroll
          for (int tw = 0; tw < 3; ++tw) {
            float4 q[R + 2];
#pragma unroll
            for (int i = 0; i < R + 2; ++i) q[i] = Vec4<T>::load(hp + ((th * 6 + tw) * PD + i) * CC);
#pragma unroll
            for (int td = 0; td < 3; ++td) {
              float4& a = wr[td * 9 + th * 3 + tw];
#pragma unroll
              for (int r = 0; r < R; ++r) {
                a.x += gv[r].x * q[r + td].x; a.y += gv[r].y * q[r + td].y; a.z += gv[r].z * q[r + td].z; a.w += gv[r].w * q[r + td].w;
              }
            }
            // one tap row at a time (the accumulators are pinned here in program order): with all 54 reads hoisted above the
            // arithmetic the kernel needs 300 registers and loses the second wave per SIMD
#pragma unroll
            for (int td = 0; td < 3; ++td) {
              float4& a = wr[td * 9 + th * 3 + tw];
              asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w));
            }
          }
      } else {
        float4 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = bs;
#pragma unroll
        for (int th = 0; th < 3; ++th)
#pragma unroll
          for (int tw = 0; tw < 3; ++tw) {
            float4 q[R + 2];
#pragma unroll
            for (int i = 0; i < R + 2; ++i) q[i] = Vec4<T>::load(hp + ((th * 6 + tw) * PD + i) * CC);
            if (th == 1 && tw == 1) {               // the identity term: the run's own voxels
#pragma unroll
              for (int r = 0; r < R; ++r) { acc[r].x += q[r + 1].x; acc[r].y += q[r + 1].y; acc[r].z += q[r + 1].z; acc[r].w += q[r + 1].w; }
            }
#pragma unroll
            for (int td = 0; td < 3; ++td) {
              const float4 wv = wr[td * 9 + th * 3 + tw];
#pragma unroll
              for (int r = 0; r < R; ++r) {
                acc[r].x += wv.x * q[r + td].x; acc[r].y += wv.y * q[r + td].y; acc[r].z += wv.z * q[r + td].z; acc[r].w += wv.w * q[r + td].w;
              }
            }
            // one tap row at a time: left alone, hipcc sinks all the arithmetic into the guarded store block below, behind all 54
            // reads (300 registers: no second wave per SIMD)
#pragma unroll
            for (int r = 0; r < R; ++r) asm volatile("" : "+v"(acc[r].x), "+v"(acc[r].y), "+v"(acc[r].z), "+v"(acc[r].w));
          }
        if (h < H && ww < W) {
#pragma unroll
          for (int r = 0; r < R; ++r)
            if (d + r < D) Vec4<T>::store(y + e + (long long)r * C, make_float4(acc[r].x * m.x, acc[r].y * m.y, acc[r].z * m.z, acc[r].w * m.w));
        }
      }
    }
  }
  if (MODE == 2) {
    // reduce over the voxel slots: lanes l, l+16, l+32, l+48 of a wave share a channel quad (bf16; l, l+8, .. for fp32) ->
    // cross-lane adds, then the 4 wave sums meet in LDS (the halo region is free now).  No LDS atomics: they cost ~200
    // cycles per wave instruction on this chip.
    __syncthreads();
    float* wsum = reinterpret_cast<float*>(halo);       // [4 waves][28][CC]
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int tp = 0; tp < 28; ++tp) {
      float4 v = tp < 27 ? wr[tp] : bs;
      if constexpr (QV == 16) {            // lanes 16 and 32 apart hold the same channel quad: row / half exchanges, no LDS crossbar
        v.x = xhalf_combine<LtuAdd>(xrow_combine<LtuAdd>(v.x)); v.y = xhalf_combine<LtuAdd>(xrow_combine<LtuAdd>(v.y));
        v.z = xhalf_combine<LtuAdd>(xrow_combine<LtuAdd>(v.z)); v.w = xhalf_combine<LtuAdd>(xrow_combine<LtuAdd>(v.w));
      } else {
#pragma unroll
        for (int o = QV; o < 64; o <<= 1) {
          v.x += __shfl_xor(v.x, o); v.y += __shfl_xor(v.y, o); v.z += __shfl_xor(v.z, o); v.w += __shfl_xor(v.w, o);
        }
      }
      if (lane < QV) *reinterpret_cast<float4*>(wsum + (wave * 28 + tp) * CC + lane * 4) = v;
    }
    __syncthreads();
    // consecutive lanes -> consecutive gradient addresses: dwt[chunk channels][27] is one contiguous run
    for (int f = tid; f < 27 * CC; f += 256) {
      const int cl = f / 27, tp = f - cl * 27;
      if (blockIdx.x * CC + cl >= C) continue;
      const float v = wsum[tp * CC + cl] + wsum[(28 + tp) * CC + cl] + wsum[(56 + tp) * CC + cl] + wsum[(84 + tp) * CC + cl];
      // two-stage (part != nullptr): this workgroup's sums go to part[block][chunk][28 CC] and dwconv_fold_kernel adds the blocks
      // in a fixed order; the atomics of the fallback make these two gradients differ in the last bits from run to run
      if (part != nullptr) part[((long long)blockIdx.y * gridDim.x + blockIdx.x) * 28 * CC + f] = v;
      else atomicAdd(dwt + (long long)blockIdx.x * CC * 27 + f, v);
    }
    if (tid < CC && blockIdx.x * CC + tid < C) {
      const float v = wsum[27 * CC + tid] + wsum[(28 + 27) * CC + tid] + wsum[(56 + 27) * CC + tid] + wsum[(84 + 27) * CC + tid];
      if (part != nullptr) part[((long long)blockIdx.y * gridDim.x + blockIdx.x) * 28 * CC + 27 * CC + tid] = v;
      else atomicAdd(db + blockIdx.x * CC + tid, v);
    }
  }
}

// dwt[chunk channels][27] / db[chunk channels] += sum over blocks of part[block][chunk][28 CC]; grid (nchunk, 28 CC / 64), block 256:
// 64 consecutive outputs x 4 thread groups that share the blocks
__global__ void __launch_bounds__(256) dwconv_fold_kernel(const float* __restrict__ part, int nblk, int nchunk, int CC, int C,
                                                          float* __restrict__ dwt, float* __restrict__ db) {
  __shared__ float red[3][64];
  const int chunk = blockIdx.x, l = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int f = blockIdx.y * 64 + l;                     // index inside the chunk's 28 CC run
  float a0 = 0.f, a1 = 0.f;
  if (f < 28 * CC) {
    const float* p = part + (long long)chunk * 28 * CC + f;
    const long long bs = (long long)nchunk * 28 * CC;
    int z = grp;
    for (; z + 12 < nblk; z += 16) { a0 += p[z * bs] + p[(z + 8) * bs]; a1 += p[(z + 4) * bs] + p[(z + 12) * bs]; }
    for (; z < nblk; z += 4) a0 += p[z * bs];
  }
  const float a = a0 + a1;
  if (grp > 0) red[grp - 1][l] = a;
  __syncthreads();
  if (grp != 0 || f >= 28 * CC) return;
  const float v = a + red[0][l] + red[1][l] + red[2][l];
  if (f < 27 * CC) {
    if (chunk * CC + f / 27 < C) dwt[(long long)chunk * CC * 27 + f] += v;
  } else if (chunk * CC + (f - 27 * CC) < C) {
    db[chunk * CC + (f - 27 * CC)] += v;
  }
}

template <typename T, int MODE>
static void launch_dwconv_halo(const void* x, const void* x2, const void* g, const void* g2, const float* w, const float* bias, void* y,
                               float* dwt, float* db, int B, int H, int W, int D, int C, float p, uint64_t seed, const uint64_t* step,
                               hipStream_t st, float* part = nullptr) {
  constexpr int CC = 128 / (int)sizeof(T);
  const int nchunk = cdiv(C, CC);
  const long long bricks = (long long)B * ((H + 3) / 4) * ((W + 3) / 4) * ((D + 7) / 8);
  long long nblk = (MODE == 2 ? ltu_knob_pos("LTU_DW_BLOCKS2", 512) : ltu_knob_pos("LTU_DW_BLOCKS", 512)) / nchunk;      // fewer, longer-lived workgroups for the reduction
  if (nblk < 1) nblk = 1;
  if (nblk > bricks) nblk = bricks;
  const int bpb = (int)((bricks + nblk - 1) / nblk);
  nblk = (bricks + bpb - 1) / bpb;
  hipLaunchKernelGGL((dwconv_halo_kernel<T, MODE>), dim3(nchunk, (unsigned)nblk), dim3(256), 0, st, (const T*)x, (const T*)x2, (const T*)g,
                     (const T*)g2, w, bias, (T*)y, dwt, db, B, H, W, D, C, (int)bricks, bpb, p, seed, step, part);
  if (MODE == 2 && part != nullptr)
    hipLaunchKernelGGL(dwconv_fold_kernel, dim3(nchunk, (unsigned)cdiv(28 * CC, 64)), dim3(256), 0, st, part, (int)nblk, nchunk, CC, C, dwt, db);
}
// blocks x chunks x 28 CC floats of the weight-gradient pass (same geometry as launch_dwconv_halo<T, 2>)
extern "C" long long ltu_dwconv_bwd_ws_floats(int B, int H, int W, int D, int C, int dtype) {
  const int CC = dtype == LTU_BF16 ? 64 : 32;
  const int nchunk = cdiv(C, CC);
  const long long bricks = (long long)B * ((H + 3) / 4) * ((W + 3) / 4) * ((D + 7) / 8);
  long long nblk = ltu_knob_pos("LTU_DW_BLOCKS2", 512) / nchunk;
  if (nblk < 1) nblk = 1;
  if (nblk > bricks) nblk = bricks;
  const int bpb = (int)((bricks + nblk - 1) / nblk);
  nblk = (bricks + bpb - 1) / bpb;
  return nblk * nchunk * 28LL * CC;
}

extern "C" int ltu_dwconv_fwd(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int D, int C,
                              float p, uint64_t seed, const uint64_t* step, int dtype, ltu_stream_t s) {
  if (C % 4 || (dtype == LTU_BF16 && C % 8)) return LTU_E_SHAPE;
  LTU_DISPATCH_T(dtype, { launch_dwconv_halo<T, 0>(x, nullptr, nullptr, nullptr, w, bias, y, nullptr, nullptr, B, H, W, D, C, p, seed, step, (hipStream_t)s); });
  return ltu_check_launch();
}
extern "C" int ltu_dwconv_bwd(const void* dy, const void* dy2, const void* x, const float* w, void* dx, float* dwt, float* db, float* ws,
                              long long ws_floats, int B, int H, int W, int D, int C, float p, uint64_t seed, const uint64_t* step, int dtype,
                              ltu_stream_t s) {
  if (C % 4 || (dtype == LTU_BF16 && C % 8)) return LTU_E_SHAPE;
  if (ws != nullptr && dwt != nullptr && ws_floats < ltu_dwconv_bwd_ws_floats(B, H, W, D, C, dtype)) return LTU_E_ARG;
  LTU_DISPATCH_T(dtype, {
    // either half may be left out (dx NULL: weight / bias gradient only; dwt NULL: data gradient only): the weight gradient is off
    // the data-gradient chain and the caller may issue it later, on another stream
    if (dx != nullptr)
      launch_dwconv_halo<T, 1>(dy, dy2, nullptr, nullptr, w, nullptr, dx, nullptr, nullptr, B, H, W, D, C, p, seed, step, (hipStream_t)s);
    if (dwt != nullptr)
      launch_dwconv_halo<T, 2>(x, nullptr, dy, dy2, w, nullptr, nullptr, dwt, db, B, H, W, D, C, p, seed, step, (hipStream_t)s, ws);
  });
  return ltu_check_launch();
}
