// Row-block chain kernels for the post-attention half of a transformer layer (model/trans_block.py:203-211), bf16 storage, at every
// token level of the step (1 024 ... 114 816 tokens; d = 128 / 256).
//
//   z1 = x + drop(a Wo^T + bo);  t1 = LN1(z1);  u = t1 W1^T + b1;  h = drop(gelu(u));  z2 = t1 + drop(h W2^T + b2);  y = LN2(z2)
//
// Launched op by op (projection, LayerNorm, projection + GELU, projection, LayerNorm) every one of these is a 5-20 us kernel at
// the small levels - pipeline fill and drain, not bytes - and a full read + write of the activations at the large one.  Every op of
// the chain is per token, so a workgroup can carry a block of 32 token rows through all of it (38 KB of LDS at d = 128, 75 KB at
// d = 256):
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
#ifndef TL_QDEPTH
#define TL_QDEPTH 8
#endif

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
  uint16_t* u;            // [M][2d]  FFN pre-activation (u_mode 0) or dropout mask * gelu'(u) (u_mode 1: what the backward chain multiplies by)
  uint16_t* h;            // [M][2d]  dropout(gelu(u))
  uint16_t* z2;           // [M][d]
  uint16_t* y;            // [M][d]   layer output
  float* stat1;           // [M][2]   mean, rstd
  float* stat2;
  long long M;
  float eps, p;
  uint64_t seed1, seedg, seed2;               // dropout sites: after the out projection, after GELU, after linear2
  const uint64_t* step;
  // fused attention phase B (ATTN kernels): q rows come from qkv [M][3d], the per-(sample, head) context from ctx [B*H][32][32];
  // the attention output goes to a_out [M][d] (the out-projection weight gradient reads it) and the row statistics to qstat
  const uint16_t* qkv;
  const float* ctx;
  uint16_t* a_out;
  float* qstat;           // [M][H][2] = (row max, 1 / (row sum * sqrt(32)))
  int ntok;               // tokens per sample (a multiple of 32: a row block never straddles samples)
  // fused q|k|v projection of the NEXT layer (QKV kernels): qkv_next [M][3d] = y Wqkv^T + b, Wqkv = cat(Wq, Wk, Wv) of the next layer
  // in fragment order ([3d][d], kind 8); nothing but per-token ops lies between LayerNorm 2 and that projection
  // (model/trans_block.py:156-158 follows :210 of the previous layer)
  const uint16_t* wq;
  const float* bq[3];
  uint16_t* qkv_next;
  int u_mode;
  int dbg;                // ablation (tools/bench_tail.py): 2 = no GELU / dropout arithmetic, 4 = u and h are not stored
};

__device__ __forceinline__ tl_bf16x8 as_bf16x8(uint4 v) { return __builtin_bit_cast(tl_bf16x8, v); }

// One column tile of a GEMM stage: acc[..] = sum_k W[ct*32 + ..][k] X[..][k] (transposed product, see above).
// X: LDS, bf16, row stride LDX elements; W: fragment order in global memory (served by L2: every workgroup streams the same
// weights).  KS (reduction steps of 16) is a compile-time constant: the loop is fully unrolled around a rotating queue of U weight
// fragments.  The queue is filled by tl_issue - which the caller places BEFORE the global stores of the previous tile's epilogue:
// gfx950 counts loads and stores in one in-order counter (vmcnt), so a load issued after a store cannot be waited for without also
// waiting for the store's acknowledgement - and drained by tl_run.  Every step is fenced with sched_barrier: left alone, hipcc
// sinks each weight load next to its MFMA (one load in flight per wave, an L2 round trip per MFMA).
template <int KS, int QD = TL_QDEPTH>
struct TlQueue {
  static constexpr int U = KS < QD ? KS : QD;
  uint4 q[U];
};

template <int KS, int QD = TL_QDEPTH>
__device__ __forceinline__ void tl_issue(const uint16_t* __restrict__ W, int ct, int lane, TlQueue<KS, QD>& wq) {
  const uint4* wp = reinterpret_cast<const uint4*>(W) + (size_t)ct * KS * 64 + lane;
#pragma unroll
  for (int i = 0; i < TlQueue<KS, QD>::U; ++i) wq.q[i] = wp[(size_t)i * 64];
  __builtin_amdgcn_sched_barrier(0);
}

template <int KS, int QD = TL_QDEPTH>
__device__ __forceinline__ void tl_run(const uint16_t* __restrict__ W, int ct, const uint16_t* Xl, int LDX, int lane, TlQueue<KS, QD>& wq,
                                       f32x16& acc) {
  constexpr int U = TlQueue<KS, QD>::U;
  const int li = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const uint4* wp = reinterpret_cast<const uint4*>(W) + (size_t)ct * KS * 64 + lane;
  const uint16_t* xr = Xl + li * LDX + 8 * lh;
  uint4 xv = *reinterpret_cast<const uint4*>(xr);
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    const uint4 w = wq.q[k % U];
    if (k + U < KS) wq.q[k % U] = wp[(size_t)(k + U) * 64];
    uint4 xn = xv;
    if (k + 1 < KS) xn = *reinterpret_cast<const uint4*>(xr + (k + 1) * 16);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(w), as_bf16x8(xv), acc, 0, 0, 0);
    asm volatile("" : "+v"(acc));          // the MFMA has no side effect of its own: without this it floats past the fences
    xv = xn;
    __builtin_amdgcn_sched_barrier(0);
  }
}

#ifndef TL_LN_E128
#define TL_LN_E128 8
#endif
#ifndef TL_LN_E256
#define TL_LN_E256 8
#endif
#ifndef TL_LN_E256F
#define TL_LN_E256F TL_LN_E256      // forward stages (no column-sum scratch: 16 is possible there)
#endif
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

// Q quads (4 bf16 each) of one lane <-> memory: 16-byte accesses for pairs of quads
template <int Q>
__device__ __forceinline__ void st_quads(uint16_t* p, const uint2 (&v)[Q]) {
  if constexpr (Q == 1) *reinterpret_cast<uint2*>(p) = v[0];
  else {
#pragma unroll
    for (int q = 0; q < Q; q += 2) *reinterpret_cast<uint4*>(p + 4 * q) = make_uint4(v[q].x, v[q].y, v[q + 1].x, v[q + 1].y);
  }
}
template <int Q>
__device__ __forceinline__ void ld_quads(const uint16_t* p, uint2 (&v)[Q]) {
  if constexpr (Q == 1) v[0] = *reinterpret_cast<const uint2*>(p);
  else {
#pragma unroll
    for (int q = 0; q < Q; q += 2) {
      const uint4 t = *reinterpret_cast<const uint4*>(p + 4 * q);
      v[q] = make_uint2(t.x, t.y); v[q + 1] = make_uint2(t.z, t.w);
    }
  }
}
// (RES_LDS / Y_LDS are template flags rather than run-time null checks: hipcc 7.2 crashes in its inliner on the latter.)
// LayerNorm stage over the block: row = res + drop(r), normalised.  r: LDS (bf16, stride LD); res: LDS (stride LD) or global
// (stride D); z and y go to global memory, y also to LDS `yl` (nullable).  A row is handled by D/4 lanes (one quad each).
template <int D, bool RES_LDS, bool Y_LDS>
__device__ __forceinline__ void tl_layernorm(const uint16_t* rl, const uint16_t* res_l, const uint16_t* res_g, uint16_t* yl, int LD,
                                             const float* __restrict__ gamma, const float* __restrict__ beta, uint16_t* zg,
                                             uint16_t* yg, float* stat, long long row0, long long M, float eps, const DropCfg& dc,
                                             int tid) {
  // E elements per lane: 8 (d = 128: 16 lanes per row = one DPP row, the two reductions of a row stay inside it; d = 256: 32 lanes,
  // one row exchange instead of two), 16-byte LDS and global accesses, half the row passes of the 4-element form.  Measured
  // (tools/bench_tail.py): forward 91.3 -> 86.8 us, backward 95.4 -> 85.3 us at 114 816 x 128; 47.0 -> 44.3 | 41.4 -> 40.0 at
  // 21 504 x 256, 29.3 -> 27.6 | 26.0 -> 25.2 at 8 640 x 256, 15.9 -> 14.8 | 13.8 -> 13.2 at 1 024 x 256.
  constexpr int E = D == 128 ? TL_LN_E128 : TL_LN_E256F, Q = E / 4;
  constexpr int G = D / E;                 // lanes per row (64, 32 or 16)
  constexpr int RPP = (D >= 256 ? 512 : 256) / G;      // rows per pass (the workgroup has D / 32 waves, at most 8)
  const int gl = tid % G;
  float gm[E], bt[E];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const float4 a = *reinterpret_cast<const float4*>(gamma + gl * E + 4 * q), c = *reinterpret_cast<const float4*>(beta + gl * E + 4 * q);
    gm[4 * q] = a.x; gm[4 * q + 1] = a.y; gm[4 * q + 2] = a.z; gm[4 * q + 3] = a.w;
    bt[4 * q] = c.x; bt[4 * q + 1] = c.y; bt[4 * q + 2] = c.z; bt[4 * q + 3] = c.w;
  }
  for (int r = tid / G; r < TL_ROWS; r += RPP) {
    const long long row = row0 + r;
    const bool ok = row < M;
    float z[E];
    uint2 zq[Q];
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      float4 rv = unpack_quad(*reinterpret_cast<const uint2*>(rl + r * LD + gl * E + 4 * q));
      float4 xv;
      if constexpr (RES_LDS) xv = unpack_quad(*reinterpret_cast<const uint2*>(res_l + r * LD + gl * E + 4 * q));
      else xv = ok ? unpack_quad(*reinterpret_cast<const uint2*>(res_g + row * D + gl * E + 4 * q)) : make_float4(0.f, 0.f, 0.f, 0.f);
      rv = drop4(dc, (unsigned long long)(((row * D + gl * E) >> 2) + q), rv);
      z[4 * q] = xv.x + rv.x; z[4 * q + 1] = xv.y + rv.y; z[4 * q + 2] = xv.z + rv.z; z[4 * q + 3] = xv.w + rv.w;
      zq[q] = pack_quad(z[4 * q], z[4 * q + 1], z[4 * q + 2], z[4 * q + 3]);
      sum += (z[4 * q] + z[4 * q + 1]) + (z[4 * q + 2] + z[4 * q + 3]);
    }
    if (ok) st_quads<Q>(zg + row * D + gl * E, zq);
    const float mean = group_sum<G>(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < E; ++k) { z[k] -= mean; sq += z[k] * z[k]; }
    const float rstd = rsqrtf(group_sum<G>(sq) / (float)D + eps);
    uint2 yv[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q)
      yv[q] = pack_quad(z[4 * q] * rstd * gm[4 * q] + bt[4 * q], z[4 * q + 1] * rstd * gm[4 * q + 1] + bt[4 * q + 1],
                        z[4 * q + 2] * rstd * gm[4 * q + 2] + bt[4 * q + 2], z[4 * q + 3] * rstd * gm[4 * q + 3] + bt[4 * q + 3]);
    if constexpr (Y_LDS) st_quads<Q>(yl + r * LD + gl * E, yv);
    if (ok) st_quads<Q>(yg + row * D + gl * E, yv);
    if (ok && gl == 0) { stat[row * 2] = mean; stat[row * 2 + 1] = rstd; }
  }
}

// OCC = waves per SIMD the kernel is built for.  5 (d = 128 only, no q|k|v stage): the "slim" layout - TB shares HB's memory (the
// out-projection result is dead before u is written; the linear2 result is written after a barrier behind the last read of h), the
// residual rows x are read from global memory in LayerNorm 1 instead of waiting in LDS, and the weight queue is 6 deep: 30.7 KB of
// LDS and <= 96 registers = five workgroups per CU instead of four.  Measured (round 3): 91-93 us either way at 114 816 x 128 and
// +0.07 ms per step - the kernel is instruction-issue-bound, a fifth wave per SIMD has nothing to hide.  Kept behind
// LTU_TAIL_FWD_OCC=5, off by default.
template <int D, bool ATTN, bool QKV, int OCC>
__global__ void __launch_bounds__(D >= 256 ? 512 : 256, OCC) tail_fwd_kernel(const TailArgs ta) {
  constexpr bool SLIM = OCC >= 5;
  constexpr int QD = SLIM ? 6 : TL_QDEPTH;
  static_assert(!SLIM || (D == 128 && !QKV), "the slim layout exists for d = 128 without the fused q|k|v stage");
  constexpr int LD = D + 8, LDH = 2 * D + 8;          // padded rows: +16 bytes rotates the banks from row to row
  constexpr int NW = D >= 256 ? 8 : 4, NTHR = NW * 64;
  constexpr int NT1 = D / 32, NT2 = 2 * D / 32;       // column tiles of the d-wide and the 2d-wide stages
  static_assert(NT1 == NW && NT2 == 2 * NW, "one d-wide tile and two 2d-wide tiles per wave");
  constexpr int KS1 = D / 16, KS2 = 2 * D / 16;       // reduction steps over d and over 2d
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  uint16_t* XA = smem;                                // [32][LD]   a, later t1
  uint16_t* TB = XA + TL_ROWS * LD;                   // [32][LD]   out-projection result, later linear2 result
  uint16_t* HB = SLIM ? TB : TB + TL_ROWS * LD;       // [32][LDH]  x (stage 0-2), then h
  float* PB = reinterpret_cast<float*>(HB + TL_ROWS * LDH);   // bo[D] b1[2D] b2[D] g1[D] be1[D] g2[D] be2[D]: no parameter is loaded
                                                              // from global memory behind a store (see tl_issue)
  constexpr int NPAR = QKV ? 11 * D : 8 * D;                  // QKV: + the next layer's q | k | v biases [3D]
  float* QS = PB + NPAR;                                      // ATTN: [32][H][2] row statistics of the block
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const long long row0 = (long long)blockIdx.x * TL_ROWS;
  const DropCfg dc1 = make_drop(ta.p, ta.seed1, ta.step), dcg = make_drop(ta.p, ta.seedg, ta.step), dc2 = make_drop(ta.p, ta.seed2, ta.step);

  // stage 0: the wave's out-projection weights first (they do not depend on the rows), then the block's rows of the attention
  // output -> XA and of the residual input -> HB (free until stage 3: loading it in stage 2 would expose one more global round
  // trip per workgroup), and the parameters -> PB
  TlQueue<KS1, QD> q1;
  tl_issue<KS1, QD>(ta.wo, wave, lane, q1);
  {
    static_assert(8 * D == 4 * NTHR, "one float4 of parameters per thread");
    const int e = tid * 4;
    const float* src = e < D ? ta.bo + e : e < 3 * D ? ta.b1 + (e - D) : e < 4 * D ? ta.b2 + (e - 3 * D) : e < 5 * D ? ta.g1 + (e - 4 * D)
                       : e < 6 * D ? ta.be1 + (e - 5 * D) : e < 7 * D ? ta.g2 + (e - 6 * D) : ta.be2 + (e - 7 * D);
    *reinterpret_cast<float4*>(PB + e) = *reinterpret_cast<const float4*>(src);
    if constexpr (QKV) {
      if (e < 3 * D) {
        const float* sq = e < D ? ta.bq[0] + e : e < 2 * D ? ta.bq[1] + (e - D) : ta.bq[2] + (e - 2 * D);
        *reinterpret_cast<float4*>(PB + 8 * D + e) = *reinterpret_cast<const float4*>(sq);
      }
    }
  }
  for (int i = tid; i < TL_ROWS * (D / 8); i += NTHR) {
    const int r = i / (D / 8), c = (i % (D / 8)) * 8;
    uint4 v = make_uint4(0u, 0u, 0u, 0u), xv = v;
    if (row0 + r < ta.M) {
      if constexpr (ATTN) v = *reinterpret_cast<const uint4*>(ta.qkv + (row0 + r) * (3 * D) + c);      // the q third of the row
      else v = *reinterpret_cast<const uint4*>(ta.a + (row0 + r) * D + c);
      if constexpr (!SLIM) xv = *reinterpret_cast<const uint4*>(ta.x + (row0 + r) * D + c);
    }
    *reinterpret_cast<uint4*>(XA + r * LD + c) = v;
    if constexpr (!SLIM) *reinterpret_cast<uint4*>(HB + r * LD + c) = xv;
  }
  if constexpr (ATTN) {
    // Phase B of the linear attention (linattn.hip: linattn_apply_rows) on the block's q rows, wave = head: row softmax over the
    // head's 32 channels, out^T[j][t] = sum_i ctx[i][j] qs[t][i] with A = ctx^T; the lane's quads of token li go back into XA in
    // place - XA then holds the attention output `a`, which stage 1 consumes - and leave for global memory in whole rows below.
    constexpr int H = D / 32;
    const long long bh = (row0 / ta.ntok) * H + wave;
    const float* cx = ta.ctx + bh * 1024;
    tl_bf16x8 ca[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      tl_bf16x8 t;
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = (__bf16)cx[(16 * lh + 8 * u + e) * 32 + li];
      ca[u] = t;
    }
    __syncthreads();                       // the q rows are in XA
    const uint16_t* qr = XA + li * LD + wave * 32 + 16 * lh;
    const uint4 v0 = *reinterpret_cast<const uint4*>(qr), v1 = *reinterpret_cast<const uint4*>(qr + 8);
    const uint32_t wv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    float a[16];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[2 * k] = __uint_as_float(wv[k] << 16); a[2 * k + 1] = __uint_as_float(wv[k] & 0xffff0000u); }
    float mx = a[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) mx = fmaxf(mx, a[k]);
    mx = xhalf_combine<LtuMax>(mx);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { a[k] = __expf(a[k] - mx); sum += a[k]; }
    sum = xhalf_combine<LtuAdd>(sum);
    const float inv = 0.17677669529663688110f / sum;      // 1 / (sqrt(32) * row sum)
    if (lh == 0) *reinterpret_cast<float2*>(QS + (li * H + wave) * 2) = make_float2(mx, inv);
    tl_bf16x8 p0, p1;
#pragma unroll
    for (int k = 0; k < 8; ++k) { p0[k] = (__bf16)(a[k] * inv); p1[k] = (__bf16)(a[8 + k] * inv); }
    f32x16 oa;
#pragma unroll
    for (int r = 0; r < 16; ++r) oa[r] = 0.f;
    oa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[0], p0, oa, 0, 0, 0);
    oa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[1], p1, oa, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<uint2*>(XA + li * LD + wave * 32 + 8 * q + 4 * lh) = pack_quad(oa[4 * q], oa[4 * q + 1], oa[4 * q + 2], oa[4 * q + 3]);
    __syncthreads();                       // `a` complete in XA
    for (int i = tid; i < TL_ROWS * (D / 8); i += NTHR) {
      const int r = i / (D / 8), c = (i % (D / 8)) * 8;
      if (row0 + r < ta.M) *reinterpret_cast<uint4*>(ta.a_out + (row0 + r) * D + c) = *reinterpret_cast<const uint4*>(XA + r * LD + c);
    }
    for (int i = tid; i < TL_ROWS * H / 2; i += NTHR) {       // 32 rows x H pairs = TL_ROWS * H * 2 floats, 4 per thread
      const int r = (i * 4) / (2 * H);
      if (row0 + r < ta.M) *reinterpret_cast<float4*>(ta.qstat + row0 * (2 * H) + i * 4) = *reinterpret_cast<const float4*>(QS + i * 4);
    }
  } else {
    __syncthreads();
  }
  const float* bo = PB, *b1 = PB + D, *b2 = PB + 3 * D, *g1 = PB + 4 * D, *be1 = PB + 5 * D, *g2 = PB + 6 * D, *be2 = PB + 7 * D;

  // stage 1: out projection -> TB (bf16)
  f32x16 acc;
  tl_run<KS1, QD>(ta.wo, wave, XA, LD, lane, q1, acc);
  tl_issue<KS1, QD>(ta.w1, wave, lane, q1);           // first linear1 tile: in flight across the LayerNorm stage
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int n = wave * 32 + 8 * q + 4 * lh;
    const float4 b = *reinterpret_cast<const float4*>(bo + n);
    *reinterpret_cast<uint2*>(TB + li * LD + n) = pack_quad(acc[4 * q] + b.x, acc[4 * q + 1] + b.y, acc[4 * q + 2] + b.z, acc[4 * q + 3] + b.w);
  }
  __syncthreads();
  // stage 2: z1 = x + drop(o), t1 = LN1(z1) -> XA (the attention rows are no longer needed)
  tl_layernorm<D, !SLIM, true>(TB, HB, ta.x, XA, LD, g1, be1, ta.z1, ta.t1, ta.stat1, row0, ta.M, ta.eps, dc1, tid);
  __syncthreads();
  // stage 3a: u = t1 W1^T + b1 -> HB (bf16); the wave's tiles are wave and wave + NW
  TlQueue<KS2, QD> q2;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int ct = wave + t * NW;
    tl_run<KS1, QD>(ta.w1, ct, XA, LD, lane, q1, acc);
    if (t == 0) tl_issue<KS1, QD>(ta.w1, wave + NW, lane, q1);
    else tl_issue<KS2, QD>(ta.w2, wave, lane, q2);    // linear2 operands: requested before any store of stage 3b
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = ct * 32 + 8 * q + 4 * lh;
      const float4 b = *reinterpret_cast<const float4*>(b1 + n);
      *reinterpret_cast<uint2*>(HB + li * LDH + n) = pack_quad(acc[4 * q] + b.x, acc[4 * q + 1] + b.y, acc[4 * q + 2] + b.z, acc[4 * q + 3] + b.w);
    }
  }
  __syncthreads();
  // stage 3b: row-contiguous pass over the block: u -> global, h = drop(gelu(u)) -> global and (in place) HB.  The accumulator
  // layout gives a lane one token, i.e. a wave store would touch 32 rows with 16 bytes each; here a wave writes whole 512-byte
  // or 1 KiB row segments (the vector L1 works per line: bytes per line, not bytes, set the store rate).
  {
    constexpr int CPR = 2 * D / 8;                    // 16-byte chunks per row
    for (int i = tid; i < TL_ROWS * CPR; i += NTHR) {
      const int r = i / CPR, c = (i % CPR) * 8;
      const long long row = row0 + r;
      const uint4 uv = *reinterpret_cast<const uint4*>(HB + r * LDH + c);
      const float4 u0 = unpack_quad(make_uint2(uv.x, uv.y)), u1 = unpack_quad(make_uint2(uv.z, uv.w));
      float4 h0 = u0, h1 = u1;
      uint4 sv = uv;                                  // what the backward pass gets: u, or (u_mode 1) mask * gelu'(u) - the
      if (!(ta.dbg & 2)) {                            // Gaussian terms are shared with gelu(u), the backward chain then needs
        const unsigned long long g4 = (unsigned long long)((row * (2 * D) + c) >> 2);      // no exp / rcp / hash at all
        const float4 m0 = dropmask4(dcg, g4), m1 = dropmask4(dcg, g4 + 1);
        const float uu[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
        const float mm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
        float hh[8], gg[8];
#pragma unroll
        for (int k = 0; k < 8; k += 2) {                // pairs: packed fp32 multiplies / fmas (gelu_cdf2)
          const ltu_f2 x2 = {uu[k], uu[k + 1]}, m2 = {mm[k], mm[k + 1]};
          ltu_f2 e2;
          const ltu_f2 cdf = gelu_cdf2(x2, e2);
          const ltu_f2 h2 = x2 * cdf * m2;
          const ltu_f2 g2v = (cdf + x2 * 0.39894228040143267794f * e2) * m2;
          hh[k] = h2.x; hh[k + 1] = h2.y;
          gg[k] = g2v.x; gg[k + 1] = g2v.y;
        }
        h0 = make_float4(hh[0], hh[1], hh[2], hh[3]); h1 = make_float4(hh[4], hh[5], hh[6], hh[7]);
        if (ta.u_mode) {
          const uint2 s0 = pack_quad(gg[0], gg[1], gg[2], gg[3]), s1 = pack_quad(gg[4], gg[5], gg[6], gg[7]);
          sv = make_uint4(s0.x, s0.y, s1.x, s1.y);
        }
      }
      const uint2 a0 = pack_quad(h0.x, h0.y, h0.z, h0.w), a1 = pack_quad(h1.x, h1.y, h1.z, h1.w);
      const uint4 hv = make_uint4(a0.x, a0.y, a1.x, a1.y);
      *reinterpret_cast<uint4*>(HB + r * LDH + c) = hv;
      if (row < ta.M && !(ta.dbg & 4)) {
        *reinterpret_cast<uint4*>(ta.u + row * (2 * D) + c) = sv;
        *reinterpret_cast<uint4*>(ta.h + row * (2 * D) + c) = hv;
      }
    }
  }
  __syncthreads();
  // stage 4: linear2 -> TB
  tl_run<KS2, QD>(ta.w2, wave, HB, LDH, lane, q2, acc);
  if constexpr (SLIM) __syncthreads();                // TB is HB's memory: every wave has read its h fragments
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int n = wave * 32 + 8 * q + 4 * lh;
    const float4 b = *reinterpret_cast<const float4*>(b2 + n);
    *reinterpret_cast<uint2*>(TB + li * LD + n) = pack_quad(acc[4 * q] + b.x, acc[4 * q + 1] + b.y, acc[4 * q + 2] + b.z, acc[4 * q + 3] + b.w);
  }
  __syncthreads();
  // stage 5: z2 = t1 + drop(f), y = LN2(z2); QKV: y also replaces t1 in XA (a lane overwrites exactly the quad it has just read)
  if constexpr (QKV) tl_issue<KS1, QD>(ta.wq, wave, lane, q1);   // first q|k|v tile: in flight across the LayerNorm stage
  tl_layernorm<D, true, QKV>(TB, XA, nullptr, XA, LD, g2, be2, ta.z2, ta.y, ta.stat2, row0, ta.M, ta.eps, dc2, tid);
  if constexpr (QKV) {
    // stage 6: the next layer's q | k | v projection of the block: three column tiles per wave (wave, wave + NW, wave + 2 NW of the
    // 3D / 32), staged in the TB | HB region ([32][3D + 16]: both are free now) and stored in whole rows
    constexpr int LDQ = 3 * D + 16;
    static_assert(TL_ROWS * LDQ == TL_ROWS * (LD + LDH), "the q|k|v staging tile is exactly TB + HB");
    uint16_t* QB = TB;
    const float* bq = PB + 8 * D;
    __syncthreads();                       // y complete in XA, TB (linear2 result) consumed
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int ct = wave + t * NW;
      tl_run<KS1, QD>(ta.wq, ct, XA, LD, lane, q1, acc);
      if (t < 2) tl_issue<KS1, QD>(ta.wq, wave + (t + 1) * NW, lane, q1);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = ct * 32 + 8 * q + 4 * lh;
        const float4 b = *reinterpret_cast<const float4*>(bq + n);
        *reinterpret_cast<uint2*>(QB + li * LDQ + n) = pack_quad(acc[4 * q] + b.x, acc[4 * q + 1] + b.y, acc[4 * q + 2] + b.z, acc[4 * q + 3] + b.w);
      }
    }
    __syncthreads();
    constexpr int CPRQ = 3 * D / 8;
    for (int i = tid; i < TL_ROWS * CPRQ; i += NTHR) {
      const int r = i / CPRQ, c = (i % CPRQ) * 8;
      if (row0 + r < ta.M) *reinterpret_cast<uint4*>(ta.qkv_next + (row0 + r) * (3 * D) + c) = *reinterpret_cast<const uint4*>(QB + r * LDQ + c);
    }
  }
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
  const uint16_t* u;      // [M][2d]  as written by the forward kernel (see TailArgs::u)
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
  int u_mode;
};

// LayerNorm backward over the block's rows.  g = gl (LDS, nullable) + gg1 (global, nullable) + gg2 (global, nullable); writes dz to
// dzl (LDS, nullable) / dzg (global, nullable), dr = dz * dropout mask to drl (LDS) and drg (global); column sums to lnws.
template <int D, int NTHR, bool G_LDS, bool G2_LDS, bool DZ_LDS, bool DZ_GLOBAL>
__device__ __forceinline__ void tl_layernorm_bwd(const uint16_t* gl_, const uint16_t* gl2_, const uint16_t* gg1, const uint16_t* gg2,
                                                 const uint16_t* __restrict__ zg, const float* __restrict__ stat,
                                                 const float* __restrict__ gamma, uint16_t* dzl, uint16_t* dzg, uint16_t* drl,
                                                 uint16_t* drg, float* __restrict__ lnws, float* red, int LD, long long row0,
                                                 long long M, const DropCfg& dc, int tid) {
  // E elements per lane (see tl_layernorm)
  constexpr int E = D == 128 ? TL_LN_E128 : TL_LN_E256, Q = E / 4;
  constexpr int G = D / E, RPP = NTHR / G, NR = TL_ROWS / RPP;
  const int gl = tid % G, rgp = tid / G;
  float gmv[E], ag[E], ab[E];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const float4 t = *reinterpret_cast<const float4*>(gamma + gl * E + 4 * q);
    gmv[4 * q] = t.x; gmv[4 * q + 1] = t.y; gmv[4 * q + 2] = t.z; gmv[4 * q + 3] = t.w;
  }
#pragma unroll
  for (int k = 0; k < E; ++k) { ag[k] = 0.f; ab[k] = 0.f; }
  // all global operands of the thread's NR rows first (one round trip, not one per row), then the arithmetic
  uint2 zq[NR][Q], q1[NR][Q], q2[NR][Q];
  float2 st[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const long long row = row0 + rgp + i * RPP;
#pragma unroll
    for (int q = 0; q < Q; ++q) zq[i][q] = q1[i][q] = q2[i][q] = make_uint2(0u, 0u);
    st[i] = make_float2(0.f, 0.f);
    if (row < M) {
      ld_quads<Q>(zg + row * D + gl * E, zq[i]);
      if (gg1 != nullptr) ld_quads<Q>(gg1 + row * D + gl * E, q1[i]);
      if (gg2 != nullptr) ld_quads<Q>(gg2 + row * D + gl * E, q2[i]);
      st[i] = *reinterpret_cast<const float2*>(stat + row * 2);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int r = rgp + i * RPP;
    const long long row = row0 + r;
    const bool ok = row < M;
    const float mean = st[i].x, rstd = st[i].y;
    float gv[E], h[E], gg[E], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (G_LDS) g = unpack_quad(*reinterpret_cast<const uint2*>(gl_ + r * LD + gl * E + 4 * q));
      if constexpr (G2_LDS) {
        const float4 t = unpack_quad(*reinterpret_cast<const uint2*>(gl2_ + r * LD + gl * E + 4 * q));
        g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
      }
      if (gg1 != nullptr) {
        const float4 t = unpack_quad(q1[i][q]);
        g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
      }
      if (gg2 != nullptr) {
        const float4 t = unpack_quad(q2[i][q]);
        g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
      }
      const float4 zv = unpack_quad(zq[i][q]);
      gv[4 * q] = g.x; gv[4 * q + 1] = g.y; gv[4 * q + 2] = g.z; gv[4 * q + 3] = g.w;
      h[4 * q] = (zv.x - mean) * rstd; h[4 * q + 1] = (zv.y - mean) * rstd; h[4 * q + 2] = (zv.z - mean) * rstd; h[4 * q + 3] = (zv.w - mean) * rstd;
    }
#pragma unroll
    for (int k = 0; k < E; ++k) {
      if (ok) { ab[k] += gv[k]; ag[k] += gv[k] * h[k]; }
      gg[k] = gv[k] * gmv[k];
      s1 += gg[k]; s2 += gg[k] * h[k];
    }
    const float m1 = group_sum<G>(s1) / (float)D, m2 = group_sum<G>(s2) / (float)D;
    uint2 dzq[Q], drq[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const float4 dz = make_float4(rstd * (gg[4 * q] - m1 - h[4 * q] * m2), rstd * (gg[4 * q + 1] - m1 - h[4 * q + 1] * m2),
                                    rstd * (gg[4 * q + 2] - m1 - h[4 * q + 2] * m2), rstd * (gg[4 * q + 3] - m1 - h[4 * q + 3] * m2));
      dzq[q] = pack_quad(dz.x, dz.y, dz.z, dz.w);
      const float4 dr = drop4(dc, (unsigned long long)(((row * D + gl * E) >> 2) + q), dz);
      drq[q] = pack_quad(dr.x, dr.y, dr.z, dr.w);
    }
    if constexpr (DZ_LDS) st_quads<Q>(dzl + r * LD + gl * E, dzq);
    st_quads<Q>(drl + r * LD + gl * E, drq);
    if (ok) {
      if constexpr (DZ_GLOBAL) st_quads<Q>(dzg + row * D + gl * E, dzq);
      st_quads<Q>(drg + row * D + gl * E, drq);
    }
  }
  // column sums of the block: row groups meet in LDS (red: [RPP][2 D] floats)
  float* dst = red + ((long long)rgp * D + gl * E) * 2;
#pragma unroll
  for (int k = 0; k < E; ++k) { dst[2 * k] = ag[k]; dst[2 * k + 1] = ab[k]; }
  __syncthreads();
  for (int i = tid; i < 2 * D; i += NTHR) {
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < RPP; ++q) acc += red[(long long)q * 2 * D + i];
    lnws[(long long)blockIdx.x * 2 * D + i] = acc;
  }
}

// WPS: waves per SIMD the register allocator has to fit (d = 128: 4 = four workgroups per CU at a 128-register budget, which spills
// 9 registers; 3 = 168 registers, no spills, three workgroups per CU.  d = 256 runs 512-thread workgroups, two per CU: 4)
template <int D, int UMODE, int WPS, bool PRE_U>
__global__ void __launch_bounds__(D >= 256 ? 512 : 256, WPS) tail_bwd_kernel(const TailBwdArgs ta) {
  constexpr int LD = D + 8, LDH = 2 * D + 8;
  constexpr int NW = D >= 256 ? 8 : 4, NTHR = NW * 64;
  constexpr int NT1 = D / 32, NT2 = 2 * D / 32;
  static_assert(NT1 == NW && NT2 == 2 * NW, "one d-wide tile and two 2d-wide tiles per wave");
  constexpr int KS1 = D / 16, KS2 = 2 * D / 16;
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  uint16_t* RB = smem;                                // [32][LD]   dr2, later dt1 and (in place) dr1
  uint16_t* ZB = RB + TL_ROWS * LD;                   // [32][LD]   dz2
  uint16_t* UB = ZB + TL_ROWS * LD;                   // [32][LDH]  du
  float* red = reinterpret_cast<float*>(UB);          // [NTHR / (D/4)][2 D] column-sum scratch of the two LayerNorm stages: du is
                                                      // not yet written in stage 0 and no longer needed in stage 3
  static_assert((NTHR / (D / (D == 128 ? TL_LN_E128 : TL_LN_E256))) * 2 * D * sizeof(float) <= TL_ROWS * LDH * sizeof(uint16_t), "column-sum scratch fits in UB");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const long long row0 = (long long)blockIdx.x * TL_ROWS;
  const DropCfg dc1 = make_drop(ta.p, ta.seed1, ta.step), dcg = make_drop(ta.p, ta.seedg, ta.step), dc2 = make_drop(ta.p, ta.seed2, ta.step);

  // the wave's first linear2^T operands: in flight across the LayerNorm stage
  TlQueue<KS1> q1;
  TlQueue<KS2> q2;
  // stage 0: LayerNorm 2 backward: dz2 -> ZB, dr2 -> RB + global
  tl_layernorm_bwd<D, NTHR, false, false, true, false>(nullptr, nullptr, ta.dy, ta.dy2, ta.z2, ta.stat2, ta.g2, ZB, nullptr, RB, ta.dr2,
                                                      ta.lnws2, red, LD, row0, ta.M, dc2, tid);
  tl_issue<KS1>(ta.w2t, wave, lane, q1);
  // the thread's 16-byte chunks of the pre-activations u for stage 1b (row-contiguous): in flight across stage 1a (d = 128; at
  // d = 256, where the budget is 128 registers, holding them costs 16 registers = spills: they are fetched in stage 1b instead)
  constexpr int CPR2 = 2 * D / 8, NCH2 = TL_ROWS * CPR2 / NTHR;
  uint4 uch[PRE_U ? NCH2 : 1];
  if constexpr (PRE_U) {
#pragma unroll
    for (int j = 0; j < NCH2; ++j) {
      const int i = tid + j * NTHR, r = i / CPR2, c = (i % CPR2) * 8;
      uch[j] = make_uint4(0u, 0u, 0u, 0u);
      if (row0 + r < ta.M) uch[j] = *reinterpret_cast<const uint4*>(ta.u + (row0 + r) * (2 * D) + c);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  // stage 1a: dh = dr2 W2 -> UB (bf16, as stored by the op-by-op path); the wave's tiles are wave and wave + NW
  f32x16 acc;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int ct = wave + t * NW;
    tl_run<KS1>(ta.w2t, ct, RB, LD, lane, q1, acc);
    if (t == 0) tl_issue<KS1>(ta.w2t, wave + NW, lane, q1);
    else tl_issue<KS2>(ta.w1t, wave, lane, q2);       // linear1^T operands: requested before any store of stage 1b
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = ct * 32 + 8 * q + 4 * lh;
      *reinterpret_cast<uint2*>(UB + li * LDH + n) = pack_quad(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    }
  }
  __syncthreads();
  // stage 1b: row-contiguous pass: du = dh * dropout mask * gelu'(u) -> global and (in place) UB (see stage 3b of the forward kernel)
#pragma unroll
  for (int j = 0; j < NCH2; ++j) {
    const int i = tid + j * NTHR, r = i / CPR2, c = (i % CPR2) * 8;
    const long long rw = row0 + r;
    const uint4 dv = *reinterpret_cast<const uint4*>(UB + r * LDH + c);
    const float4 d0 = unpack_quad(make_uint2(dv.x, dv.y)), d1 = unpack_quad(make_uint2(dv.z, dv.w));
    uint4 uv;
    if constexpr (PRE_U) uv = uch[j];
    else uv = rw < ta.M ? *reinterpret_cast<const uint4*>(ta.u + rw * (2 * D) + c) : make_uint4(0u, 0u, 0u, 0u);
    const float4 u0 = unpack_quad(make_uint2(uv.x, uv.y)), u1 = unpack_quad(make_uint2(uv.z, uv.w));
    uint2 a0, a1;
    if constexpr (UMODE == 1) {         // the forward kernel left mask * gelu'(u)
      a0 = pack_quad(d0.x * u0.x, d0.y * u0.y, d0.z * u0.z, d0.w * u0.w);
      a1 = pack_quad(d1.x * u1.x, d1.y * u1.y, d1.z * u1.z, d1.w * u1.w);
    } else {
      const unsigned long long g4 = (unsigned long long)((rw * (2 * D) + c) >> 2);
      const float4 m0 = dropmask4(dcg, g4), m1 = dropmask4(dcg, g4 + 1);
      a0 = pack_quad(d0.x * m0.x * gelu_erf_grad(u0.x), d0.y * m0.y * gelu_erf_grad(u0.y), d0.z * m0.z * gelu_erf_grad(u0.z),
                     d0.w * m0.w * gelu_erf_grad(u0.w));
      a1 = pack_quad(d1.x * m1.x * gelu_erf_grad(u1.x), d1.y * m1.y * gelu_erf_grad(u1.y), d1.z * m1.z * gelu_erf_grad(u1.z),
                     d1.w * m1.w * gelu_erf_grad(u1.w));
    }
    const uint4 dq = make_uint4(a0.x, a0.y, a1.x, a1.y);
    *reinterpret_cast<uint4*>(UB + r * LDH + c) = dq;
    if (rw < ta.M) *reinterpret_cast<uint4*>(ta.du + rw * (2 * D) + c) = dq;
  }
  __syncthreads();
  // stage 2: dt1 = du W1 -> RB (dr2 is no longer needed)
  tl_run<KS2>(ta.w1t, wave, UB, LDH, lane, q2, acc);
  tl_issue<KS1>(ta.wot, wave, lane, q1);              // the out-projection^T operands: in flight across the LayerNorm stage
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int n = wave * 32 + 8 * q + 4 * lh;
    *reinterpret_cast<uint2*>(RB + li * LD + n) = pack_quad(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
  }
  __syncthreads();
  // stage 3: LayerNorm 1 backward on dt1 + dz2: dz1 -> global, dr1 -> RB (in place) + global
  tl_layernorm_bwd<D, NTHR, true, true, false, true>(RB, ZB, nullptr, nullptr, ta.z1, ta.stat1, ta.g1, nullptr, ta.dz1, RB, ta.dr1, ta.lnws1,
                                                    red, LD, row0, ta.M, dc1, tid);
  __syncthreads();
  // stage 4: da = dr1 Wo -> ZB (dz2 is no longer needed), then row-contiguous stores
  tl_run<KS1>(ta.wot, wave, RB, LD, lane, q1, acc);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int n = wave * 32 + 8 * q + 4 * lh;
    *reinterpret_cast<uint2*>(ZB + li * LD + n) = pack_quad(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
  }
  __syncthreads();
  {
    constexpr int CPR = D / 8;
    for (int i = tid; i < TL_ROWS * CPR; i += NTHR) {
      const int r = i / CPR, c = (i % CPR) * 8;
      if (row0 + r < ta.M) *reinterpret_cast<uint4*>(ta.da + (row0 + r) * D + c) = *reinterpret_cast<const uint4*>(ZB + r * LD + c);
    }
  }
}

extern "C" long long ltu_layer_tail_blocks(long long M) { return (M + TL_ROWS - 1) / TL_ROWS; }

extern "C" int ltu_layer_tail_bwd(const void* dy, const void* dy2, const void* z2, const void* z1, const void* u, const float* stat2,
                                  const float* stat1, const float* g2, const float* g1, const void* w2t, const void* w1t,
                                  const void* wot, void* dr2, void* du, void* dr1, void* dz1, void* da, float* lnws2, float* lnws1,
                                  long long lnws_floats, long long M, int d, float p, uint64_t seed1, uint64_t seedg, uint64_t seed2, const uint64_t* step,
                                  int u_mode, int dtype, ltu_stream_t s) {
  if (dtype != LTU_BF16) return LTU_E_DTYPE;
  if (d != 128 && d != 256) return LTU_E_SHAPE;
  if (u_mode != 0 && u_mode != 1) return LTU_E_ARG;
  if (M <= 0) return LTU_OK;
  if (lnws_floats < ltu_layer_tail_blocks(M) * 2 * d) return LTU_E_ARG;       // capacity of EACH of lnws2 / lnws1
  TailBwdArgs ta;
  ta.u_mode = u_mode;
  ta.dy = (const uint16_t*)dy; ta.dy2 = (const uint16_t*)dy2; ta.z2 = (const uint16_t*)z2; ta.z1 = (const uint16_t*)z1;
  ta.u = (const uint16_t*)u; ta.stat2 = stat2; ta.stat1 = stat1; ta.g2 = g2; ta.g1 = g1;
  ta.w2t = (const uint16_t*)w2t; ta.w1t = (const uint16_t*)w1t; ta.wot = (const uint16_t*)wot;
  ta.dr2 = (uint16_t*)dr2; ta.du = (uint16_t*)du; ta.dr1 = (uint16_t*)dr1; ta.dz1 = (uint16_t*)dz1; ta.da = (uint16_t*)da;
  ta.lnws2 = lnws2; ta.lnws1 = lnws1;
  ta.M = M; ta.p = p; ta.seed1 = seed1; ta.seedg = seedg; ta.seed2 = seed2; ta.step = step;
  const unsigned blocks = cdiv(M, TL_ROWS);
  const size_t lds = (size_t)TL_ROWS * (2 * (d + 8) + (2 * d + 8)) * sizeof(uint16_t);
  auto launch = [&](auto kern, unsigned threads) {
    static LtuDevOnce once;                // one latch per kernel instantiation (the lambda body is instantiated per `kern` type)
    if (once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, (hipStream_t)s, ta);
  };
  if (d == 256) {
    if (ltu_knob("LTU_TAIL_BWD_PREU", 0)) { if (u_mode) launch(&tail_bwd_kernel<256, 1, 4, true>, 512); else launch(&tail_bwd_kernel<256, 0, 4, true>, 512); }
    else { if (u_mode) launch(&tail_bwd_kernel<256, 1, 4, false>, 512); else launch(&tail_bwd_kernel<256, 0, 4, false>, 512); }
  } else if (ltu_knob("LTU_TAIL_BWD_WPS", 4) == 3) {
    if (u_mode) launch(&tail_bwd_kernel<128, 1, 3, true>, 256); else launch(&tail_bwd_kernel<128, 0, 3, true>, 256);
  } else if (ltu_knob("LTU_TAIL_BWD_PREU", 1) == 0) {       // d = 128: u prefetched across stage 1a since the 8-element LayerNorm
    if (u_mode) launch(&tail_bwd_kernel<128, 1, 4, false>, 256); else launch(&tail_bwd_kernel<128, 0, 4, false>, 256);      // stages left room (127 registers, no spills): 88.7 -> 84.3 us
  } else {
    if (u_mode) launch(&tail_bwd_kernel<128, 1, 4, true>, 256); else launch(&tail_bwd_kernel<128, 0, 4, true>, 256);
  }
  return ltu_check_launch();
}

extern "C" int ltu_layer_tail_fwd(const void* a, const void* x, const void* wo, const void* w1, const void* w2, const float* bo,
                                  const float* b1, const float* b2, const float* g1, const float* be1, const float* g2,
                                  const float* be2, void* z1, void* t1, void* u, void* h, void* z2, void* y, float* stat1,
                                  float* stat2, long long M, int d, float eps, float p, uint64_t seed1, uint64_t seedg,
                                  uint64_t seed2, const uint64_t* step, int u_mode, const void* qkv, const float* ctx, float* qstat,
                                  int ntok, const void* wq_next, const float* bq0, const float* bq1, const float* bq2,
                                  void* qkv_next, int dtype, ltu_stream_t s) {
  if (dtype != LTU_BF16) return LTU_E_DTYPE;
  if (d != 128 && d != 256) return LTU_E_SHAPE;
  if (u_mode != 0 && u_mode != 1) return LTU_E_ARG;
  if (qkv != nullptr && (ctx == nullptr || qstat == nullptr || ntok <= 0 || ntok % TL_ROWS || M % ntok)) return LTU_E_ARG;
  if (qkv_next != nullptr && (wq_next == nullptr || bq0 == nullptr || bq1 == nullptr || bq2 == nullptr)) return LTU_E_ARG;
  if (M <= 0) return LTU_OK;
  TailArgs ta;
  ta.u_mode = u_mode;
  ta.qkv = (const uint16_t*)qkv; ta.ctx = ctx; ta.qstat = qstat; ta.ntok = ntok; ta.a_out = (uint16_t*)const_cast<void*>(a);
  ta.a = (const uint16_t*)a; ta.x = (const uint16_t*)x;
  ta.wo = (const uint16_t*)wo; ta.w1 = (const uint16_t*)w1; ta.w2 = (const uint16_t*)w2;
  ta.bo = bo; ta.b1 = b1; ta.b2 = b2; ta.g1 = g1; ta.be1 = be1; ta.g2 = g2; ta.be2 = be2;
  ta.z1 = (uint16_t*)z1; ta.t1 = (uint16_t*)t1; ta.u = (uint16_t*)u; ta.h = (uint16_t*)h; ta.z2 = (uint16_t*)z2; ta.y = (uint16_t*)y;
  ta.stat1 = stat1; ta.stat2 = stat2;
  ta.M = M; ta.eps = eps; ta.p = p; ta.seed1 = seed1; ta.seedg = seedg; ta.seed2 = seed2; ta.step = step;
  ta.dbg = ltu_knob("LTU_TAIL_DBG", 0);
  ta.wq = (const uint16_t*)wq_next; ta.bq[0] = bq0; ta.bq[1] = bq1; ta.bq[2] = bq2; ta.qkv_next = (uint16_t*)qkv_next;
  const bool qn = qkv_next != nullptr;
#ifdef LTU_EXPERIMENTS
  const bool slim = d == 128 && !qn && ltu_knob("LTU_TAIL_FWD_OCC", 4) >= 5;
#else
  constexpr bool slim = false;               // the five-workgroups-per-CU layout lost its measurement: experiments build only
#endif
  const unsigned blocks = cdiv(M, TL_ROWS);
  const size_t lds = (size_t)TL_ROWS * ((slim ? 1 : 2) * (d + 8) + (2 * d + 8)) * sizeof(uint16_t) + (size_t)(qn ? 11 : 8) * d * sizeof(float) +
                     (qkv != nullptr ? (size_t)TL_ROWS * (d / 32) * 2 * sizeof(float) : 0);
  auto launch = [&](auto kern, unsigned threads) {
    static LtuDevOnce once;                // one latch per kernel instantiation
    if (once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, (hipStream_t)s, ta);
  };
  if (d == 256) {
    if (qkv != nullptr) { if (qn) launch(&tail_fwd_kernel<256, true, true, 4>, 512); else launch(&tail_fwd_kernel<256, true, false, 4>, 512); }
    else { if (qn) launch(&tail_fwd_kernel<256, false, true, 4>, 512); else launch(&tail_fwd_kernel<256, false, false, 4>, 512); }
#ifdef LTU_EXPERIMENTS
  } else if (slim) {
    if (qkv != nullptr) launch(&tail_fwd_kernel<128, true, false, 5>, 256); else launch(&tail_fwd_kernel<128, false, false, 5>, 256);
#endif
  } else {
    if (qkv != nullptr) { if (qn) launch(&tail_fwd_kernel<128, true, true, 4>, 256); else launch(&tail_fwd_kernel<128, true, false, 4>, 256); }
    else { if (qn) launch(&tail_fwd_kernel<128, false, true, 4>, 256); else launch(&tail_fwd_kernel<128, false, false, 4>, 256); }
  }
  return ltu_check_launch();
}
