// Row-block chain kernels for the post-attention half of a transformer layer (model/trans_block.py:203-211) on the SMALL token
// levels (a few thousand to ~20 000 tokens), bf16 storage.
//
//   z1 = x + drop(a Wo^T + bo);  t1 = LN1(z1);  u = t1 W1^T + b1;  h = drop(gelu(u));  z2 = t1 + drop(h W2^T + b2);  y = LN2(z2)
//
// Launched op by op (projection, LayerNorm, projection + GELU, projection, LayerNorm) every one of these is a 5-20 us kernel whose
// time is pipeline fill and drain, not bytes: the three small levels cost 40 % of the step for 20 % of the tokens.  Every op of
// the chain is per token, so a workgroup can carry a block of 32 token rows through all of it (67 KB of LDS: two workgroups per CU):
//   * the activations of the block stay in LDS between the stages (bf16, padded rows);
//   * the weights are read straight from L2 into MFMA operand registers: they are prepared once per step in fragment order
//     (weight-prep kind 8 / 9: a wave's 16 x 32 operand is one coalesced 1 KiB load), are shared by all workgroups and never
//     pass through LDS, so the GEMM stages need no barriers of their own;
//   * the product is formed transposed (D[n][m] = W X^T, v_mfma_f32_32x32x16_bf16): a lane owns 4 consecutive output columns of
//     one token, i.e. one 8-byte bf16 quad - the unit of the dropout hash, of the LDS stores and of the global stores.
// Rounding points follow the op-by-op path (each projection output is rounded to bf16 before the next op reads it), so both
// paths agree to bf16 rounding noise of the LayerNorm sums.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 tl_bf16x8;

#define TL_ROWS 32

struct TailArgs {
  const uint16_t* a;      // [M][d] attention output
  const uint16_t* x;      // [M][d] layer input (residual)
  const uint16_t* wo;     // fragment-ordered bf16 weights (kind 8): out [d][d], linear1 [2d][d], linear2 [d][2d]
  const uint16_t* w1;
  const uint16_t* w2;
  const float* bo;
  const float* b1;
  const float* b2;
  const float* g1;
  const float* be1;
  const float* g2;
  const float* be2;
  uint16_t* z1;           // [M][d]   pre-norm sums (LayerNorm backward)
  uint16_t* t1;           // [M][d]   LN1 output
  uint16_t* u;            // [M][2d]  FFN pre-activation
  uint16_t* h;            // [M][2d]  dropout(gelu(u))
  uint16_t* z2;           // [M][d]
  uint16_t* y;            // [M][d]   layer output
  float* stat1;           // [M][2]   mean, rstd
  float* stat2;
  long long M;
  float eps, p;
  uint64_t seed1, seedg, seed2;               // dropout sites: after the out projection, after GELU, after linear2
  const uint64_t* step;
};

__device__ __forceinline__ tl_bf16x8 as_bf16x8(uint4 v) { return __builtin_bit_cast(tl_bf16x8, v); }

// One column tile of a GEMM stage: acc[g][..] = sum_k W[ct*32 + ..][k] X[(rg0+g)*32 + ..][k] (transposed product, see above).
// X: LDS, bf16, row stride LDX elements; W: fragment order in global memory.  Operand loads run U reduction steps ahead of the
// matrix instructions.
template <int NRG>
__device__ __forceinline__ void tl_tile(const uint16_t* __restrict__ W, int KS, int ct, int rg0, const uint16_t* Xl, int LDX,
                                        int lane, f32x16 (&acc)[NRG]) {
  constexpr int U = 4;
  const int li = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int g = 0; g < NRG; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
  const uint4* wp = reinterpret_cast<const uint4*>(W) + (size_t)ct * KS * 64 + lane;
  const uint16_t* xr[NRG];
#pragma unroll
  for (int g = 0; g < NRG; ++g) xr[g] = Xl + ((rg0 + g) * 32 + li) * LDX + 8 * lh;
  uint4 wv[U];
#pragma unroll
  for (int i = 0; i < U; ++i) wv[i] = wp[(size_t)i * 64];
  for (int k0 = 0; k0 < KS; k0 += U) {
    uint4 wc[U], xv[NRG][U];
#pragma unroll
    for (int i = 0; i < U; ++i) {
      wc[i] = wv[i];
#pragma unroll
      for (int g = 0; g < NRG; ++g) xv[g][i] = *reinterpret_cast<const uint4*>(xr[g] + (k0 + i) * 16);
    }
    if (k0 + U < KS) {
#pragma unroll
      for (int i = 0; i < U; ++i) wv[i] = wp[(size_t)(k0 + U + i) * 64];
    }
#pragma unroll
    for (int i = 0; i < U; ++i)
#pragma unroll
      for (int g = 0; g < NRG; ++g)
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(wc[i]), as_bf16x8(xv[g][i]), acc[g], 0, 0, 0);
  }
}

__device__ __forceinline__ uint2 pack_quad(float a, float b, float c, float d) {
  uint2 r;
  r.x = pack_bf16x2(a, b);
  r.y = pack_bf16x2(c, d);
  return r;
}
__device__ __forceinline__ float4 unpack_quad(uint2 v) {
  return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                     __uint_as_float(v.y & 0xffff0000u));
}

// (RES_LDS / Y_LDS are template flags rather than run-time null checks: hipcc 7.2 crashes in its inliner on the latter.)
// LayerNorm stage over the block: row = res + drop(r), normalised.  r: LDS (bf16, stride LD); res: LDS (stride LD) or global
// (stride D); z and y go to global memory, y also to LDS `yl` (nullable).  A row is handled by D/4 lanes (one quad each).
template <int D, bool RES_LDS, bool Y_LDS>
__device__ __forceinline__ void tl_layernorm(const uint16_t* rl, const uint16_t* res_l, const uint16_t* res_g, uint16_t* yl, int LD,
                                             const float* __restrict__ gamma, const float* __restrict__ beta, uint16_t* zg,
                                             uint16_t* yg, float* stat, long long row0, long long M, float eps, const DropCfg& dc,
                                             int tid) {
  constexpr int G = D / 4;                 // lanes per row (64 or 32)
  constexpr int RPP = (D >= 256 ? 512 : 256) / G;      // rows per pass (the workgroup has D / 32 waves, at most 8)
  const int gl = tid % G;
  const float4 gm = *reinterpret_cast<const float4*>(gamma + gl * 4), bt = *reinterpret_cast<const float4*>(beta + gl * 4);
  for (int r = tid / G; r < TL_ROWS; r += RPP) {
    const long long row = row0 + r;
    const bool ok = row < M;
    float4 rv = unpack_quad(*reinterpret_cast<const uint2*>(rl + r * LD + gl * 4));
    float4 xv;
    if constexpr (RES_LDS) xv = unpack_quad(*reinterpret_cast<const uint2*>(res_l + r * LD + gl * 4));
    else xv = ok ? unpack_quad(*reinterpret_cast<const uint2*>(res_g + row * D + gl * 4)) : make_float4(0.f, 0.f, 0.f, 0.f);
    rv = drop4(dc, (unsigned long long)((row * D + gl * 4) >> 2), rv);
    float z[4] = {xv.x + rv.x, xv.y + rv.y, xv.z + rv.z, xv.w + rv.w};
    if (ok) *reinterpret_cast<uint2*>(zg + row * D + gl * 4) = pack_quad(z[0], z[1], z[2], z[3]);
    const float mean = group_sum<G>(z[0] + z[1] + z[2] + z[3]) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { z[k] -= mean; sq += z[k] * z[k]; }
    const float rstd = rsqrtf(group_sum<G>(sq) / (float)D + eps);
    const uint2 yv = pack_quad(z[0] * rstd * gm.x + bt.x, z[1] * rstd * gm.y + bt.y, z[2] * rstd * gm.z + bt.z, z[3] * rstd * gm.w + bt.w);
    if constexpr (Y_LDS) *reinterpret_cast<uint2*>(yl + r * LD + gl * 4) = yv;
    if (ok) {
      *reinterpret_cast<uint2*>(yg + row * D + gl * 4) = yv;
      if (gl == 0) { stat[row * 2] = mean; stat[row * 2 + 1] = rstd; }
    }
  }
}

template <int D>
__global__ void __launch_bounds__(D >= 256 ? 512 : 256) tail_fwd_kernel(const TailArgs ta) {
  constexpr int LD = D + 8, LDH = 2 * D + 8;          // padded rows: +16 bytes rotates the banks from row to row
  constexpr int NW = D >= 256 ? 8 : 4, NTHR = NW * 64;
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  uint16_t* XA = smem;                                // [32][LD]   a, later t1
  uint16_t* TB = XA + TL_ROWS * LD;                   // [32][LD]   out-projection result, later linear2 result
  uint16_t* HB = TB + TL_ROWS * LD;                   // [32][LDH]  h
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const long long row0 = (long long)blockIdx.x * TL_ROWS;
  const DropCfg dc1 = make_drop(ta.p, ta.seed1, ta.step), dcg = make_drop(ta.p, ta.seedg, ta.step), dc2 = make_drop(ta.p, ta.seed2, ta.step);

  // stage 0: the block's rows of the attention output
  for (int i = tid; i < TL_ROWS * (D / 8); i += NTHR) {
    const int r = i / (D / 8), c = (i % (D / 8)) * 8;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (row0 + r < ta.M) v = *reinterpret_cast<const uint4*>(ta.a + (row0 + r) * D + c);
    *reinterpret_cast<uint4*>(XA + r * LD + c) = v;
  }
  __syncthreads();

  constexpr int NT1 = D / 32, NT2 = 2 * D / 32;       // column tiles; wave w takes tiles w, w + NW, ..
  // stage 1: out projection -> TB (bf16)
  for (int ct = wave; ct < NT1; ct += NW) {
    f32x16 acc[1];
    tl_tile<1>(ta.wo, D / 16, ct, 0, XA, LD, lane, acc);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = ct * 32 + 8 * q + 4 * lh;
      const float4 b = *reinterpret_cast<const float4*>(ta.bo + n);
      *reinterpret_cast<uint2*>(TB + li * LD + n) =
          pack_quad(acc[0][4 * q] + b.x, acc[0][4 * q + 1] + b.y, acc[0][4 * q + 2] + b.z, acc[0][4 * q + 3] + b.w);
    }
  }
  __syncthreads();
  // stage 2: z1 = x + drop(o), t1 = LN1(z1) -> XA (the attention rows are no longer needed)
  tl_layernorm<D, false, true>(TB, nullptr, ta.x, XA, LD, ta.g1, ta.be1, ta.z1, ta.t1, ta.stat1, row0, ta.M, ta.eps, dc1, tid);
  __syncthreads();
  // stage 3: u = t1 W1^T + b1 (global), h = drop(gelu(u)) (global + HB)
  for (int ct = wave; ct < NT2; ct += NW) {
    f32x16 acc[1];
    tl_tile<1>(ta.w1, D / 16, ct, 0, XA, LD, lane, acc);
    const long long row = row0 + li;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = ct * 32 + 8 * q + 4 * lh;
      const float4 b = *reinterpret_cast<const float4*>(ta.b1 + n);
      const uint2 uq = pack_quad(acc[0][4 * q] + b.x, acc[0][4 * q + 1] + b.y, acc[0][4 * q + 2] + b.z, acc[0][4 * q + 3] + b.w);
      const float4 uf = unpack_quad(uq);          // GELU of the bf16-rounded pre-activation, as the stand-alone kernel computes it
      const float4 hv = drop4(dcg, (unsigned long long)((row * (2 * D) + n) >> 2),
                              make_float4(gelu_erf(uf.x), gelu_erf(uf.y), gelu_erf(uf.z), gelu_erf(uf.w)));
      const uint2 hq = pack_quad(hv.x, hv.y, hv.z, hv.w);
      *reinterpret_cast<uint2*>(HB + li * LDH + n) = hq;
      if (row < ta.M) {
        *reinterpret_cast<uint2*>(ta.u + row * (2 * D) + n) = uq;
        *reinterpret_cast<uint2*>(ta.h + row * (2 * D) + n) = hq;
      }
    }
  }
  __syncthreads();
  // stage 4: linear2 -> TB
  for (int ct = wave; ct < NT1; ct += NW) {
    f32x16 acc[1];
    tl_tile<1>(ta.w2, 2 * D / 16, ct, 0, HB, LDH, lane, acc);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = ct * 32 + 8 * q + 4 * lh;
      const float4 b = *reinterpret_cast<const float4*>(ta.b2 + n);
      *reinterpret_cast<uint2*>(TB + li * LD + n) =
          pack_quad(acc[0][4 * q] + b.x, acc[0][4 * q + 1] + b.y, acc[0][4 * q + 2] + b.z, acc[0][4 * q + 3] + b.w);
    }
  }
  __syncthreads();
  // stage 5: z2 = t1 + drop(f), y = LN2(z2)
  tl_layernorm<D, true, false>(TB, XA, nullptr, nullptr, LD, ta.g2, ta.be2, ta.z2, ta.y, ta.stat2, row0, ta.M, ta.eps, dc2, tid);
}

extern "C" int ltu_layer_tail_fwd(const void* a, const void* x, const void* wo, const void* w1, const void* w2, const float* bo,
                                  const float* b1, const float* b2, const float* g1, const float* be1, const float* g2,
                                  const float* be2, void* z1, void* t1, void* u, void* h, void* z2, void* y, float* stat1,
                                  float* stat2, long long M, int d, float eps, float p, uint64_t seed1, uint64_t seedg,
                                  uint64_t seed2, const uint64_t* step, int dtype, ltu_stream_t s) {
  if (dtype != LTU_BF16) return LTU_E_DTYPE;
  if (d != 128 && d != 256) return LTU_E_SHAPE;
  if (M <= 0) return LTU_OK;
  TailArgs ta;
  ta.a = (const uint16_t*)a; ta.x = (const uint16_t*)x;
  ta.wo = (const uint16_t*)wo; ta.w1 = (const uint16_t*)w1; ta.w2 = (const uint16_t*)w2;
  ta.bo = bo; ta.b1 = b1; ta.b2 = b2; ta.g1 = g1; ta.be1 = be1; ta.g2 = g2; ta.be2 = be2;
  ta.z1 = (uint16_t*)z1; ta.t1 = (uint16_t*)t1; ta.u = (uint16_t*)u; ta.h = (uint16_t*)h; ta.z2 = (uint16_t*)z2; ta.y = (uint16_t*)y;
  ta.stat1 = stat1; ta.stat2 = stat2;
  ta.M = M; ta.eps = eps; ta.p = p; ta.seed1 = seed1; ta.seedg = seedg; ta.seed2 = seed2; ta.step = step;
  const unsigned blocks = cdiv(M, TL_ROWS);
  const size_t lds = (size_t)TL_ROWS * (2 * (d + 8) + (2 * d + 8)) * sizeof(uint16_t);
  if (d == 256) {
    static LtuDevOnce once;
    if (once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_fwd_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((tail_fwd_kernel<256>), dim3(blocks), dim3(512), lds, (hipStream_t)s, ta);
  } else {
    static LtuDevOnce once;
    if (once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_fwd_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((tail_fwd_kernel<128>), dim3(blocks), dim3(256), lds, (hipStream_t)s, ta);
  }
  return ltu_check_launch();
}
