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
  int dbg;                // ablation (tools/bench_tail.py): 1 = weight operands are not loaded
};

__device__ __forceinline__ tl_bf16x8 as_bf16x8(uint4 v) { return __builtin_bit_cast(tl_bf16x8, v); }

// One column tile of a GEMM stage: acc[g][..] = sum_k W[ct*32 + ..][k] X[(rg0+g)*32 + ..][k] (transposed product, see above).
// X: LDS, bf16, row stride LDX elements; W: fragment order in global memory (served by L2: every workgroup streams the same
// weights).  KS (reduction steps of 16) is a compile-time constant: the loop is fully unrolled around a rotating queue of U weight
// fragments, so U loads stay in flight per wave all the time (a counted vmcnt per step instead of a drain per group of steps).
template <int NRG, int KS>
__device__ __forceinline__ void tl_tile(const uint16_t* __restrict__ W, int ct, int rg0, const uint16_t* Xl, int LDX, int lane,
                                        f32x16 (&acc)[NRG], int dbg = 0) {
  constexpr int U = KS < 8 ? KS : 8;
  const int li = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int g = 0; g < NRG; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
  const uint4* wp = reinterpret_cast<const uint4*>(W) + (size_t)ct * KS * 64 + lane;
  const uint16_t* xr[NRG];
#pragma unroll
  for (int g = 0; g < NRG; ++g) xr[g] = Xl + ((rg0 + g) * 32 + li) * LDX + 8 * lh;
  uint4 wq[U];
  if (dbg & 1) {            // ablation: one fragment, re-used for every step
    const uint4 w0 = wp[0];
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
      for (int g = 0; g < NRG; ++g) {
        const uint4 xv = *reinterpret_cast<const uint4*>(xr[g] + k * 16);
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(w0), as_bf16x8(xv), acc[g], 0, 0, 0);
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < U; ++i) wq[i] = wp[(size_t)i * 64];
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    const uint4 w = wq[k % U];
    if (k + U < KS) wq[k % U] = wp[(size_t)(k + U) * 64];
#pragma unroll
    for (int g = 0; g < NRG; ++g) {
      const uint4 xv = *reinterpret_cast<const uint4*>(xr[g] + k * 16);
      acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(w), as_bf16x8(xv), acc[g], 0, 0, 0);
    }
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

  // stage 0: the block's rows of the attention output -> XA, and of the residual input -> HB (free until stage 3): the residual
  // is needed only by the LayerNorm of stage 2, but loading it there would expose one more global round trip per workgroup
  for (int i = tid; i < TL_ROWS * (D / 8); i += NTHR) {
    const int r = i / (D / 8), c = (i % (D / 8)) * 8;
    uint4 v = make_uint4(0u, 0u, 0u, 0u), xv = v;
    if (row0 + r < ta.M) {
      v = *reinterpret_cast<const uint4*>(ta.a + (row0 + r) * D + c);
      xv = *reinterpret_cast<const uint4*>(ta.x + (row0 + r) * D + c);
    }
    *reinterpret_cast<uint4*>(XA + r * LD + c) = v;
    *reinterpret_cast<uint4*>(HB + r * LD + c) = xv;
  }
  __syncthreads();

  constexpr int NT1 = D / 32, NT2 = 2 * D / 32;       // column tiles; wave w takes tiles w, w + NW, ..
  // stage 1: out projection -> TB (bf16)
  for (int ct = wave; ct < NT1; ct += NW) {
    f32x16 acc[1];
    tl_tile<1, D / 16>(ta.wo, ct, 0, XA, LD, lane, acc, ta.dbg);
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
  tl_layernorm<D, true, true>(TB, HB, nullptr, XA, LD, ta.g1, ta.be1, ta.z1, ta.t1, ta.stat1, row0, ta.M, ta.eps, dc1, tid);
  __syncthreads();
  // stage 3: u = t1 W1^T + b1 (global), h = drop(gelu(u)) (global + HB)
  for (int ct = wave; ct < NT2; ct += NW) {
    f32x16 acc[1];
    tl_tile<1, D / 16>(ta.w1, ct, 0, XA, LD, lane, acc, ta.dbg);
    const long long row = row0 + li;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = ct * 32 + 8 * q + 4 * lh;
      const float4 b = *reinterpret_cast<const float4*>(ta.b1 + n);
      const uint2 uq = pack_quad(acc[0][4 * q] + b.x, acc[0][4 * q + 1] + b.y, acc[0][4 * q + 2] + b.z, acc[0][4 * q + 3] + b.w);
      const float4 uf = unpack_quad(uq);          // GELU of the bf16-rounded pre-activation, as the stand-alone kernel computes it
      float4 hv = uf;
      if (!(ta.dbg & 2))
        hv = drop4(dcg, (unsigned long long)((row * (2 * D) + n) >> 2),
                   make_float4(gelu_erf(uf.x), gelu_erf(uf.y), gelu_erf(uf.z), gelu_erf(uf.w)));
      const uint2 hq = pack_quad(hv.x, hv.y, hv.z, hv.w);
      *reinterpret_cast<uint2*>(HB + li * LDH + n) = hq;
      if (row < ta.M && !(ta.dbg & 4)) {
        *reinterpret_cast<uint2*>(ta.u + row * (2 * D) + n) = uq;
        *reinterpret_cast<uint2*>(ta.h + row * (2 * D) + n) = hq;
      }
    }
  }
  __syncthreads();
  // stage 4: linear2 -> TB
  for (int ct = wave; ct < NT1; ct += NW) {
    f32x16 acc[1];
    tl_tile<1, 2 * D / 16>(ta.w2, ct, 0, HB, LDH, lane, acc, ta.dbg);
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

// ------------------------------------------------------------------------------------------------ backward chain
// dy (+ dy2) -> LN2 backward -> dr2 -> (W2) -> dh -> GELU / dropout backward -> du -> (W1) -> dt1 -> LN1 backward (dt1 + dz2)
// -> dz1 (residual gradient of the layer input), dr1 -> (Wo) -> da (gradient of the attention output).
// dr2, du, dr1 are also the G operands of the three weight gradients and are written out; the per-workgroup column sums of the
// two LayerNorms (gamma / beta gradients) go to lnws2 / lnws1 [nblocks][2 d] (interleaved gamma, beta) for the batched fold.
struct TailBwdArgs {
  const uint16_t* dy;     // [M][d] gradient of the layer output
  const uint16_t* dy2;    // second gradient of the layer output (nullable)
  const uint16_t* z2;
  const uint16_t* z1;
  const uint16_t* u;      // [M][2d]
  const float* stat2;
  const float* stat1;
  const float* g2;        // LayerNorm weights
  const float* g1;
  const uint16_t* w2t;    // fragment-ordered transposes (kind 9): linear2 -> outputs 2d over d, linear1 -> d over 2d, out -> d over d
  const uint16_t* w1t;
  const uint16_t* wot;
  uint16_t* dr2;          // [M][d]
  uint16_t* du;           // [M][2d]
  uint16_t* dr1;          // [M][d]
  uint16_t* dz1;          // [M][d]
  uint16_t* da;           // [M][d]
  float* lnws2;
  float* lnws1;
  long long M;
  float p;
  uint64_t seed1, seedg, seed2;
  const uint64_t* step;
};

// LayerNorm backward over the block's rows.  g = gl (LDS, nullable) + gg1 (global, nullable) + gg2 (global, nullable); writes dz to
// dzl (LDS, nullable) / dzg (global, nullable), dr = dz * dropout mask to drl (LDS) and drg (global); column sums to lnws.
template <int D, int NTHR, bool G_LDS, bool G2_LDS, bool DZ_LDS, bool DZ_GLOBAL>
__device__ __forceinline__ void tl_layernorm_bwd(const uint16_t* gl_, const uint16_t* gl2_, const uint16_t* gg1, const uint16_t* gg2,
                                                 const uint16_t* __restrict__ zg, const float* __restrict__ stat,
                                                 const float* __restrict__ gamma, uint16_t* dzl, uint16_t* dzg, uint16_t* drl,
                                                 uint16_t* drg, float* __restrict__ lnws, float* red, int LD, long long row0,
                                                 long long M, const DropCfg& dc, int tid) {
  constexpr int G = D / 4, RPP = NTHR / G;
  const int gl = tid % G, rgp = tid / G;
  const float4 gm = *reinterpret_cast<const float4*>(gamma + gl * 4);
  float ag[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
  for (int r = rgp; r < TL_ROWS; r += RPP) {
    const long long row = row0 + r;
    const bool ok = row < M;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (G_LDS) g = unpack_quad(*reinterpret_cast<const uint2*>(gl_ + r * LD + gl * 4));
    if constexpr (G2_LDS) {
      const float4 t = unpack_quad(*reinterpret_cast<const uint2*>(gl2_ + r * LD + gl * 4));
      g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
    }
    if (gg1 != nullptr && ok) {
      const float4 t = unpack_quad(*reinterpret_cast<const uint2*>(gg1 + row * D + gl * 4));
      g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
    }
    if (gg2 != nullptr && ok) {
      const float4 t = unpack_quad(*reinterpret_cast<const uint2*>(gg2 + row * D + gl * 4));
      g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
    }
    float4 zv = make_float4(0.f, 0.f, 0.f, 0.f);
    float mean = 0.f, rstd = 0.f;
    if (ok) {
      zv = unpack_quad(*reinterpret_cast<const uint2*>(zg + row * D + gl * 4));
      mean = stat[row * 2]; rstd = stat[row * 2 + 1];
    }
    const float gv[4] = {g.x, g.y, g.z, g.w}, gmv[4] = {gm.x, gm.y, gm.z, gm.w};
    float h[4] = {(zv.x - mean) * rstd, (zv.y - mean) * rstd, (zv.z - mean) * rstd, (zv.w - mean) * rstd};
    float gg[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (ok) { ab[k] += gv[k]; ag[k] += gv[k] * h[k]; }
      gg[k] = gv[k] * gmv[k];
      s1 += gg[k]; s2 += gg[k] * h[k];
    }
    const float m1 = group_sum<G>(s1) / (float)D, m2 = group_sum<G>(s2) / (float)D;
    const float4 dz = make_float4(rstd * (gg[0] - m1 - h[0] * m2), rstd * (gg[1] - m1 - h[1] * m2), rstd * (gg[2] - m1 - h[2] * m2),
                                  rstd * (gg[3] - m1 - h[3] * m2));
    const uint2 dzq = pack_quad(dz.x, dz.y, dz.z, dz.w);
    const float4 dr = drop4(dc, (unsigned long long)((row * D + gl * 4) >> 2), dz);
    const uint2 drq = pack_quad(dr.x, dr.y, dr.z, dr.w);
    if constexpr (DZ_LDS) *reinterpret_cast<uint2*>(dzl + r * LD + gl * 4) = dzq;
    *reinterpret_cast<uint2*>(drl + r * LD + gl * 4) = drq;
    if (ok) {
      if constexpr (DZ_GLOBAL) *reinterpret_cast<uint2*>(dzg + row * D + gl * 4) = dzq;
      *reinterpret_cast<uint2*>(drg + row * D + gl * 4) = drq;
    }
  }
  // column sums of the block: row groups meet in LDS (red: [RPP][2 D] floats)
  float* dst = red + ((long long)rgp * D + gl * 4) * 2;
#pragma unroll
  for (int k = 0; k < 4; ++k) { dst[2 * k] = ag[k]; dst[2 * k + 1] = ab[k]; }
  __syncthreads();
  for (int i = tid; i < 2 * D; i += NTHR) {
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < RPP; ++q) acc += red[(long long)q * 2 * D + i];
    lnws[(long long)blockIdx.x * 2 * D + i] = acc;
  }
}

template <int D>
__global__ void __launch_bounds__(D >= 256 ? 512 : 256) tail_bwd_kernel(const TailBwdArgs ta) {
  constexpr int LD = D + 8, LDH = 2 * D + 8;
  constexpr int NW = D >= 256 ? 8 : 4, NTHR = NW * 64;
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  uint16_t* RB = smem;                                // [32][LD]   dr2, later dt1 and (in place) dr1
  uint16_t* ZB = RB + TL_ROWS * LD;                   // [32][LD]   dz2
  uint16_t* UB = ZB + TL_ROWS * LD;                   // [32][LDH]  du
  float* red = reinterpret_cast<float*>(UB + TL_ROWS * LDH);      // [NTHR / (D/4)][2 D] column-sum scratch
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const long long row0 = (long long)blockIdx.x * TL_ROWS;
  const DropCfg dc1 = make_drop(ta.p, ta.seed1, ta.step), dcg = make_drop(ta.p, ta.seedg, ta.step), dc2 = make_drop(ta.p, ta.seed2, ta.step);
  constexpr int NT1 = D / 32, NT2 = 2 * D / 32;

  // stage 0: LayerNorm 2 backward: dz2 -> ZB, dr2 -> RB + global
  tl_layernorm_bwd<D, NTHR, false, false, true, false>(nullptr, nullptr, ta.dy, ta.dy2, ta.z2, ta.stat2, ta.g2, ZB, nullptr, RB, ta.dr2,
                                                      ta.lnws2, red, LD, row0, ta.M, dc2, tid);
  __syncthreads();
  // stage 1: dh = dr2 W2, du = dh * dropout mask * gelu'(u) -> UB + global
  for (int ct = wave; ct < NT2; ct += NW) {
    f32x16 acc[1];
    tl_tile<1, D / 16>(ta.w2t, ct, 0, RB, LD, lane, acc);
    const long long row = row0 + li;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = ct * 32 + 8 * q + 4 * lh;
      const float4 dh = unpack_quad(pack_quad(acc[0][4 * q], acc[0][4 * q + 1], acc[0][4 * q + 2], acc[0][4 * q + 3]));   // bf16-rounded, as stored by the op-by-op path
      float4 uf = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < ta.M) uf = unpack_quad(*reinterpret_cast<const uint2*>(ta.u + row * (2 * D) + n));
      const float4 mk = dropmask4(dcg, (unsigned long long)((row * (2 * D) + n) >> 2));
      const uint2 dq = pack_quad(dh.x * mk.x * gelu_erf_grad(uf.x), dh.y * mk.y * gelu_erf_grad(uf.y), dh.z * mk.z * gelu_erf_grad(uf.z),
                                 dh.w * mk.w * gelu_erf_grad(uf.w));
      *reinterpret_cast<uint2*>(UB + li * LDH + n) = dq;
      if (row < ta.M) *reinterpret_cast<uint2*>(ta.du + row * (2 * D) + n) = dq;
    }
  }
  __syncthreads();
  // stage 2: dt1 = du W1 -> RB (dr2 is no longer needed)
  for (int ct = wave; ct < NT1; ct += NW) {
    f32x16 acc[1];
    tl_tile<1, 2 * D / 16>(ta.w1t, ct, 0, UB, LDH, lane, acc);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = ct * 32 + 8 * q + 4 * lh;
      *reinterpret_cast<uint2*>(RB + li * LD + n) = pack_quad(acc[0][4 * q], acc[0][4 * q + 1], acc[0][4 * q + 2], acc[0][4 * q + 3]);
    }
  }
  __syncthreads();
  // stage 3: LayerNorm 1 backward on dt1 + dz2: dz1 -> global, dr1 -> RB (in place) + global
  tl_layernorm_bwd<D, NTHR, true, true, false, true>(RB, ZB, nullptr, nullptr, ta.z1, ta.stat1, ta.g1, nullptr, ta.dz1, RB, ta.dr1, ta.lnws1,
                                                    red, LD, row0, ta.M, dc1, tid);
  __syncthreads();
  // stage 4: da = dr1 Wo
  for (int ct = wave; ct < NT1; ct += NW) {
    f32x16 acc[1];
    tl_tile<1, D / 16>(ta.wot, ct, 0, RB, LD, lane, acc);
    const long long row = row0 + li;
    if (row < ta.M) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = ct * 32 + 8 * q + 4 * lh;
        *reinterpret_cast<uint2*>(ta.da + row * D + n) = pack_quad(acc[0][4 * q], acc[0][4 * q + 1], acc[0][4 * q + 2], acc[0][4 * q + 3]);
      }
    }
  }
}

extern "C" long long ltu_layer_tail_blocks(long long M) { return (M + TL_ROWS - 1) / TL_ROWS; }

extern "C" int ltu_layer_tail_bwd(const void* dy, const void* dy2, const void* z2, const void* z1, const void* u, const float* stat2,
                                  const float* stat1, const float* g2, const float* g1, const void* w2t, const void* w1t,
                                  const void* wot, void* dr2, void* du, void* dr1, void* dz1, void* da, float* lnws2, float* lnws1,
                                  long long M, int d, float p, uint64_t seed1, uint64_t seedg, uint64_t seed2, const uint64_t* step,
                                  int dtype, ltu_stream_t s) {
  if (dtype != LTU_BF16) return LTU_E_DTYPE;
  if (d != 128 && d != 256) return LTU_E_SHAPE;
  if (M <= 0) return LTU_OK;
  TailBwdArgs ta;
  ta.dy = (const uint16_t*)dy; ta.dy2 = (const uint16_t*)dy2; ta.z2 = (const uint16_t*)z2; ta.z1 = (const uint16_t*)z1;
  ta.u = (const uint16_t*)u; ta.stat2 = stat2; ta.stat1 = stat1; ta.g2 = g2; ta.g1 = g1;
  ta.w2t = (const uint16_t*)w2t; ta.w1t = (const uint16_t*)w1t; ta.wot = (const uint16_t*)wot;
  ta.dr2 = (uint16_t*)dr2; ta.du = (uint16_t*)du; ta.dr1 = (uint16_t*)dr1; ta.dz1 = (uint16_t*)dz1; ta.da = (uint16_t*)da;
  ta.lnws2 = lnws2; ta.lnws1 = lnws1;
  ta.M = M; ta.p = p; ta.seed1 = seed1; ta.seedg = seedg; ta.seed2 = seed2; ta.step = step;
  const unsigned blocks = cdiv(M, TL_ROWS);
  const int rpp = (d >= 256 ? 512 : 256) / (d / 4);
  const size_t lds = (size_t)TL_ROWS * (2 * (d + 8) + (2 * d + 8)) * sizeof(uint16_t) + (size_t)rpp * 2 * d * sizeof(float);
  if (d == 256) {
    static LtuDevOnce once;
    if (once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_bwd_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((tail_bwd_kernel<256>), dim3(blocks), dim3(512), lds, (hipStream_t)s, ta);
  } else {
    static LtuDevOnce once;
    if (once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_bwd_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((tail_bwd_kernel<128>), dim3(blocks), dim3(256), lds, (hipStream_t)s, ta);
  }
  return ltu_check_launch();
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
  ta.dbg = ltu_knob("LTU_TAIL_DBG", 0);
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
