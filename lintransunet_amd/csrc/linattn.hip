// Softmax-feature linear attention core for gfx950 (model/trans_block.py:41-67), forward + backward.
//
//   qs = softmax_i(q[n,:]) / sqrt(32)          (per token, over the 32 head channels)
//   P  = softmax_n(k[:,i])                      (per head channel, over ALL tokens of the sample)
//   ctx[i][j] = sum_n P[n][i] v[n][j]           (32x32 per (sample, head))
//   out[n][j] = sum_i qs[n][i] ctx[i][j]
//
// q,k,v live interleaved in one projection buffer qkv [B*N][3d] (q | k | v, head h = columns
// h*32..h*32+31 of each third).  A workgroup is H waves (H = d/32 heads); wave w owns head w, so a
// workgroup streams whole token rows (coalesced 2d- or 3d-wide) and no cross-wave reduction exists.
//
// Forward  phase A  (kv_partial):  stream K,V once; running column max / sum (online softmax over N)
//                                  and the fp32 32x32 K^T V accumulator per wave, rescaled as the max
//                                  moves; one partial (m, s, acc) per (sample, split, head).
//          combine  (kv_combine):  merge the splits -> colmax, colsum, ctx.
//          phase B  (apply):       stream Q, row softmax in registers, 32x32 context product on the
//                                  f32 matrix cores, write out; saves per-(token,head) row max and
//                                  scaled inverse sum for the backward pass.
// Backward pass 1   (dctx_partial/combine): dctx = qs^T dOut (reduction over tokens), and
//                                  t[i] = sum_j dctx[i][j] ctx[i][j]  (the column-softmax Jacobian term:
//                                  sum_n P[n][i] dP[n][i] collapses to this, so no second pass over N).
//          pass 2   (bwd_apply):   per token: dq, dk, dv (three 32x32 products).
//
// The matrix products use v_mfma_f32_32x32x2_f32 (exact fp32): the core is HBM-bound (AI = 16 FLOP/B),
// fp32 MFMA time stays below the streaming time for both storage types.
#include "common.h"

#define DK 32

__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32, 64); }

// ------------------------------------------------------------------------------------------------ phase A
// grid (nsplit, B), block H*64.  part layout: [B][nsplit][H][32 (m) + 32 (s) + 1024 (accT[j][i])]
#define PART_STRIDE (64 + 1024)

template <typename T, int TOK>
__global__ void linattn_kv_partial(const T* __restrict__ qkv, float* __restrict__ part, int N, int d, int tokens_per_split) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [TOK][2d] : k | v rows
  const int H = d / DK;
  const int b = blockIdx.y, sp = blockIdx.x, nsplit = gridDim.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int n_begin = sp * tokens_per_split;
  int n_end = n_begin + tokens_per_split;
  if (n_end > N) n_end = N;
  const int row = 2 * d;                       // floats per staged token
  const T* base = qkv + (long long)b * N * 3 * d + d;   // start of k|v part of token 0

  float m_run = -INFINITY, s_run = 0.f;        // column i = li (both halves hold a copy; s is per-half partial)
  f32x16 acc;                                  // accT[j][i]: row j by register, column i = li
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  for (int n0 = n_begin; n0 < n_end; n0 += TOK) {
    __syncthreads();
    // stage TOK token rows (k|v) as fp32; rows past n_end are left unread (masked below)
    const int nvec = TOK * row / 4;
    for (int i = tid; i < nvec; i += blockDim.x) {
      const int t = i / (row / 4), c = (i % (row / 4)) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n0 + t < n_end) v = Vec4<T>::load(base + (long long)(n0 + t) * 3 * d + c);
      *reinterpret_cast<float4*>(&smem[t * row + c]) = v;
    }
    __syncthreads();
    const int ntok = min(TOK, n_end - n0);
    const float* ks = smem + wave * DK;        // k of this head: ks[t*row + i]
    const float* vs = smem + d + wave * DK;
    // column max over this tile: lane (i, half) scans tokens half, half+2, ...
    float mt = -INFINITY;
    for (int t = lh; t < ntok; t += 2) mt = fmaxf(mt, ks[t * row + li]);
    mt = fmaxf(mt, xhalf(mt));
    const float m_new = fmaxf(m_run, mt);
    const float alpha = __expf(m_run - m_new);          // exp(-inf) = 0 on the first tile
    m_run = m_new;
    s_run *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] *= alpha;
    // accT[j][i] += sum_t v[t][j] * p[t][i];  MFMA k-step = 2 tokens (lane half picks the token)
#pragma unroll 4
    for (int t0 = 0; t0 < TOK; t0 += 2) {
      const int t = t0 + lh;
      float p = 0.f, vv = 0.f;
      if (t < ntok) {
        p = __expf(ks[t * row + li] - m_new);
        vv = vs[t * row + li];
      }
      s_run += p;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, p, acc, 0, 0, 0);
    }
  }
  const float s_tot = s_run + xhalf(s_run);
  float* out = part + (((long long)b * nsplit + sp) * H + wave) * PART_STRIDE;
  if (lh == 0) {
    out[li] = m_run;
    out[32 + li] = s_tot;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int j = (r & 3) + 8 * (r >> 2) + 4 * lh;
    out[64 + j * 32 + li] = acc[r];
  }
}

// grid (B*H), block 1024: thread (i = tid>>5, j = tid&31).  stats [B*H][64] = colmax | colsum; ctx [B*H][32][32]
__global__ void linattn_kv_combine(const float* __restrict__ part, float* __restrict__ stats, float* __restrict__ ctx,
                                   int nsplit, int H) {
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int i = threadIdx.x & 31, j = threadIdx.x >> 5;     // consecutive lanes read consecutive i of accT[j][i]
  const float* p0 = part + ((long long)b * nsplit * H + h) * PART_STRIDE;
  const long long sstride = (long long)H * PART_STRIDE;
  float m = -INFINITY;
  for (int s = 0; s < nsplit; ++s) m = fmaxf(m, p0[s * sstride + i]);
  float ssum = 0.f, a = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float* p = p0 + s * sstride;
    const float f = __expf(p[i] - m);
    ssum += p[32 + i] * f;
    a += p[64 + j * 32 + i] * f;
  }
  ctx[(long long)bh * 1024 + i * 32 + j] = a / ssum;
  if (j == 0) {
    stats[bh * 64 + i] = m;
    stats[bh * 64 + 32 + i] = ssum;
  }
}

// ------------------------------------------------------------------------------------------------ phase B
// grid (ceil(N/TOKB), B), block H*64; each block handles TOKB tokens in tiles of 32.
// out [B*N][d];  qstat [B*N][H][2] = (row max, 1/(rowsum*sqrt(32)))
template <typename T>
__global__ void linattn_apply(const T* __restrict__ qkv, const float* __restrict__ ctx, T* __restrict__ out,
                              float* __restrict__ qstat, int N, int d, int tokb) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [32][d+1] q tile, reused for the out tile
  const int H = d / DK;
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int ldq = d + 1;
  // B operand: ctx[i = 2s+half][j = li]
  float cb[16];
  const float* cx = ctx + ((long long)b * H + wave) * 1024;
#pragma unroll
  for (int s = 0; s < 16; ++s) cb[s] = cx[(2 * s + lh) * 32 + li];
  const float rs = 0.17677669529663688110f;   // 1/sqrt(32)

  const int n_begin = blockIdx.x * tokb;
  const int n_end = min(N, n_begin + tokb);
  for (int n0 = n_begin; n0 < n_end; n0 += 32) {
    __syncthreads();
    for (int i = tid; i < 32 * d / 4; i += blockDim.x) {
      const int t = i / (d / 4), c = (i % (d / 4)) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n0 + t < n_end) v = Vec4<T>::load(qkv + ((long long)b * N + n0 + t) * 3 * d + c);
      float* dst = &smem[t * ldq + c];
      dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
    __syncthreads();
    // lane (tok = li, half): elements i = 2s+half of its token's head row
    float a[16];
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      a[s] = smem[li * ldq + wave * DK + 2 * s + lh];
      mx = fmaxf(mx, a[s]);
    }
    mx = fmaxf(mx, xhalf(mx));
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      a[s] = __expf(a[s] - mx);
      sum += a[s];
    }
    sum += xhalf(sum);
    const float inv = rs / sum;
    if (lh == 0 && n0 + li < n_end) {
      float* qs = qstat + (((long long)b * N + n0 + li) * H + wave) * 2;
      qs[0] = mx;
      qs[1] = inv;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s] * inv, cb[s], acc, 0, 0, 0);
    __syncthreads();   // everyone is done reading the q tile
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = (r & 3) + 8 * (r >> 2) + 4 * lh;
      smem[t * ldq + wave * DK + li] = acc[r];
    }
    __syncthreads();
    for (int i = tid; i < 32 * d / 4; i += blockDim.x) {
      const int t = i / (d / 4), c = (i % (d / 4)) * 4;
      if (n0 + t < n_end) {
        const float* src = &smem[t * ldq + c];
        Vec4<T>::store(out + ((long long)b * N + n0 + t) * d + c, make_float4(src[0], src[1], src[2], src[3]));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward pass 1
// dctxT[j][i] = sum_n dO[n][j] * qs[n][i];  same split structure as phase A.
// part layout: [B][nsplit][H][1024 (accT[j][i])]
template <typename T, int TOK>
__global__ void linattn_dctx_partial(const T* __restrict__ qkv, const T* __restrict__ dout, const float* __restrict__ qstat,
                                     float* __restrict__ part, int N, int d, int tokens_per_split) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [TOK][2d]: q | dO rows, then [TOK][H][2] stats
  const int H = d / DK;
  const int b = blockIdx.y, sp = blockIdx.x, nsplit = gridDim.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int n_begin = sp * tokens_per_split;
  const int n_end = min(N, n_begin + tokens_per_split);
  const int row = 2 * d;
  float* st = smem + TOK * row;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int n0 = n_begin; n0 < n_end; n0 += TOK) {
    __syncthreads();
    for (int i = tid; i < TOK * d / 4; i += blockDim.x) {
      const int t = i / (d / 4), c = (i % (d / 4)) * 4;
      float4 q = make_float4(0.f, 0.f, 0.f, 0.f), g = q;
      if (n0 + t < n_end) {
        q = Vec4<T>::load(qkv + ((long long)b * N + n0 + t) * 3 * d + c);
        g = Vec4<T>::load(dout + ((long long)b * N + n0 + t) * d + c);
      }
      *reinterpret_cast<float4*>(&smem[t * row + c]) = q;
      *reinterpret_cast<float4*>(&smem[t * row + d + c]) = g;
    }
    for (int i = tid; i < TOK * H * 2; i += blockDim.x) {
      const int t = i / (H * 2);
      st[i] = (n0 + t < n_end) ? qstat[((long long)b * N + n0) * H * 2 + i] : 0.f;
    }
    __syncthreads();
    const int ntok = min(TOK, n_end - n0);
#pragma unroll 4
    for (int t0 = 0; t0 < TOK; t0 += 2) {
      const int t = t0 + lh;
      float qv = 0.f, gv = 0.f;
      if (t < ntok) {
        qv = __expf(smem[t * row + wave * DK + li] - st[(t * H + wave) * 2]) * st[(t * H + wave) * 2 + 1];
        gv = smem[t * row + d + wave * DK + li];
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(gv, qv, acc, 0, 0, 0);
    }
  }
  float* out = part + (((long long)b * nsplit + sp) * H + wave) * 1024;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int j = (r & 3) + 8 * (r >> 2) + 4 * lh;
    out[j * 32 + li] = acc[r];
  }
}

// grid (B*H), block 1024: dctx[bh][i][j] = sum_s partT[s][j][i];  tvec[bh][i] = sum_j dctx[i][j]*ctx[i][j]
__global__ void linattn_dctx_combine(const float* __restrict__ part, const float* __restrict__ ctx, float* __restrict__ dctx,
                                     float* __restrict__ tvec, int nsplit, int H) {
  __shared__ float sm[32][33];
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const float* p0 = part + ((long long)b * nsplit * H + h) * 1024;
  {
    const int ii = threadIdx.x & 31, jj = threadIdx.x >> 5;   // coalesced over the split partials partT[jj][ii]
    float a = 0.f;
    for (int s = 0; s < nsplit; ++s) a += p0[(long long)s * H * 1024 + jj * 32 + ii];
    sm[jj][ii] = a;
  }
  __syncthreads();
  const int i = threadIdx.x >> 5, j = threadIdx.x & 31;
  const float a = sm[j][i];
  dctx[(long long)bh * 1024 + i * 32 + j] = a;
  float t = a * ctx[(long long)bh * 1024 + i * 32 + j];
  t = group_sum<32>(t);
  if (j == 0) tvec[bh * 32 + i] = t;
}

// ------------------------------------------------------------------------------------------------ backward pass 2
// per 32-token tile and head (tokens on the lanes, channel index in the registers):
//   dqsT[i][t] = sum_j ctx[i][j] dO[t][j]      dq[t][i] = qs[t][i] (dqsT[i][t] - sum_i' p[t][i'] dqsT[i'][t])
//   dvT[j][t]  = sum_i dctx[i][j] P[t][i]
//   dPT[i][t]  = sum_j dctx[i][j] v[t][j]      dk[t][i] = P[t][i] (dPT[i][t] - tvec[i])
// writes dqkv [B*N][3d] (dq | dk | dv)
template <typename T>
__global__ void linattn_bwd_apply(const T* __restrict__ qkv, const T* __restrict__ dout, const float* __restrict__ ctx,
                                  const float* __restrict__ dctx, const float* __restrict__ stats,
                                  const float* __restrict__ tvec, const float* __restrict__ qstat, T* __restrict__ dqkv,
                                  int N, int d, int tokb) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [32][3d+1] qkv tile (reused for dqkv), [32][d+1] dO tile
  const int H = d / DK;
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int ld3 = 3 * d + 1, ld1 = d + 1;
  float* gt = smem + 32 * ld3;
  const long long bh = (long long)b * H + wave;
  // A operands (rows on lanes):  ctxA[s] = ctx[i=li][j=2s+half];  dcA[s] = dctx[i=li][j=2s+half];
  //                              dcT[s]  = dctx[i=2s+half][j=li]
  float ctxA[16], dcA[16], dcT[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    ctxA[s] = ctx[bh * 1024 + li * 32 + 2 * s + lh];
    dcA[s] = dctx[bh * 1024 + li * 32 + 2 * s + lh];
    dcT[s] = dctx[bh * 1024 + (2 * s + lh) * 32 + li];
  }
  const float* cst = stats + bh * 64;
  const float* tv = tvec + bh * 32;

  const int n_begin = blockIdx.x * tokb;
  const int n_end = min(N, n_begin + tokb);
  for (int n0 = n_begin; n0 < n_end; n0 += 32) {
    __syncthreads();
    for (int i = tid; i < 32 * d; i += blockDim.x) {      // i over (t, 4-vector); 3d/4 + d/4 = d vectors per token
      const int t = i / d, v4 = i % d;
      const bool ok = n0 + t < n_end;
      if (v4 < 3 * d / 4) {
        const int c = v4 * 4;
        float4 v = ok ? Vec4<T>::load(qkv + ((long long)b * N + n0 + t) * 3 * d + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        float* dst = &smem[t * ld3 + c];
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
      } else {
        const int c = (v4 - 3 * d / 4) * 4;
        float4 v = ok ? Vec4<T>::load(dout + ((long long)b * N + n0 + t) * d + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        float* dst = &gt[t * ld1 + c];
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
      }
    }
    __syncthreads();
    const bool tok_ok = n0 + li < n_end;
    float rmax = 0.f, rinv = 0.f;
    if (tok_ok) {
      const float* qs = qstat + (((long long)b * N + n0 + li) * H + wave) * 2;
      rmax = qs[0];
      rinv = qs[1];
    }
    const float* qrow = &smem[li * ld3 + wave * DK];
    const float* krow = qrow + d;
    const float* vrow = qrow + 2 * d;
    const float* grow = &gt[li * ld1 + wave * DK];

    // B operands with the token on the lane: element index kk = 2s+half
    f32x16 aq, av, ak;
#pragma unroll
    for (int r = 0; r < 16; ++r) { aq[r] = 0.f; av[r] = 0.f; ak[r] = 0.f; }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int kk = 2 * s + lh;
      const float gj = grow[kk];                                           // dO[t][j=kk]
      const float pk = __expf(krow[kk] - cst[kk]) / cst[32 + kk];          // P[t][i=kk]
      const float vj = vrow[kk];                                           // v[t][j=kk]
      aq = __builtin_amdgcn_mfma_f32_32x32x2f32(ctxA[s], gj, aq, 0, 0, 0);  // dqsT[i][t]
      av = __builtin_amdgcn_mfma_f32_32x32x2f32(dcT[s], pk, av, 0, 0, 0);   // dvT[j][t]   (A[row=j][kk=i] = dctx[i][j])
      ak = __builtin_amdgcn_mfma_f32_32x32x2f32(dcA[s], vj, ak, 0, 0, 0);   // dPT[i][t]
    }
    // registers now hold rows idx(r) = (r&3) + 8*(r>>2) + 4*half for token li
    float dot = 0.f;
    float qsv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
      qsv[r] = __expf(qrow[i] - rmax) * rinv;                 // qs[t][i] = p/sqrt(32)
      dot += qsv[r] * aq[r];
    }
    dot += xhalf(dot);
    dot *= 5.65685424949238019521f;                           // sum_i p_i dqs_i = sqrt(32) * sum_i qs_i dqs_i
    float dqv[16], dkv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
      dqv[r] = qsv[r] * (aq[r] - dot);
      const float pk = __expf(krow[i] - cst[i]) / cst[32 + i];
      dkv[r] = pk * (ak[r] - tv[i]);
    }
    __syncthreads();   // all waves finished reading the staged tile
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
      smem[li * ld3 + wave * DK + i] = dqv[r];
      smem[li * ld3 + d + wave * DK + i] = dkv[r];
      smem[li * ld3 + 2 * d + wave * DK + i] = av[r];
    }
    __syncthreads();
    for (int i = tid; i < 32 * 3 * d / 4; i += blockDim.x) {
      const int t = i / (3 * d / 4), c = (i % (3 * d / 4)) * 4;
      if (n0 + t < n_end) {
        const float* src = &smem[t * ld3 + c];
        Vec4<T>::store(dqkv + ((long long)b * N + n0 + t) * 3 * d + c, make_float4(src[0], src[1], src[2], src[3]));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
static int pick_splits(int B, int N, int* tokens_per_split) {
  int want = 1024 / (B > 0 ? B : 1);
  if (want < 1) want = 1;
  int tps = (N + want - 1) / want;
  if (tps < 64) tps = 64;
  tps = (tps + 31) / 32 * 32;
  *tokens_per_split = tps;
  return (N + tps - 1) / tps;
}

extern "C" int ltu_linattn_splits(int B, int N) {
  int tps;
  return pick_splits(B, N, &tps);
}

extern "C" int ltu_linattn_fwd(const void* qkv, void* out, float* ctx, float* colstats, float* qstat, float* part_ws,
                               int B, int N, int d, int dtype, ltu_stream_t s) {
  if (d % 32 != 0 || d < 32 || d > 512) return LTU_E_SHAPE;
  const int H = d / 32;
  int tps;
  const int nsplit = pick_splits(B, N, &tps);
  hipStream_t st = (hipStream_t)s;
  constexpr int TOK = 32;
  const size_t lds_a = (size_t)TOK * 2 * d * sizeof(float);
  const size_t lds_b = (size_t)32 * (d + 1) * sizeof(float);
  const int tokb = 128;
  LTU_DISPATCH_T(dtype, {
    hipLaunchKernelGGL((linattn_kv_partial<T, TOK>), dim3(nsplit, B), dim3(H * 64), lds_a, st, (const T*)qkv, part_ws, N, d, tps);
    hipLaunchKernelGGL(linattn_kv_combine, dim3(B * H), dim3(1024), 0, st, part_ws, colstats, ctx, nsplit, H);
    hipLaunchKernelGGL((linattn_apply<T>), dim3(cdiv(N, tokb), B), dim3(H * 64), lds_b, st, (const T*)qkv, ctx, (T*)out, qstat, N, d, tokb);
  });
  return ltu_check_launch();
}

extern "C" int ltu_linattn_bwd(const void* qkv, const void* dout, const float* ctx, const float* colstats,
                               const float* qstat, void* dqkv, float* dctx, float* tvec, float* part_ws, int B, int N,
                               int d, int dtype, ltu_stream_t s) {
  if (d % 32 != 0 || d < 32 || d > 256) return LTU_E_SHAPE;
  const int H = d / 32;
  int tps;
  const int nsplit = pick_splits(B, N, &tps);
  hipStream_t st = (hipStream_t)s;
  constexpr int TOK = 32;
  const size_t lds_a = (size_t)(TOK * 2 * d + TOK * H * 2) * sizeof(float);
  const size_t lds_b = (size_t)(32 * (3 * d + 1) + 32 * (d + 1)) * sizeof(float);
  const int tokb = 64;
  LTU_DISPATCH_T(dtype, {
    hipLaunchKernelGGL((linattn_dctx_partial<T, TOK>), dim3(nsplit, B), dim3(H * 64), lds_a, st, (const T*)qkv, (const T*)dout, qstat, part_ws, N, d, tps);
    hipLaunchKernelGGL(linattn_dctx_combine, dim3(B * H), dim3(1024), 0, st, part_ws, ctx, dctx, tvec, nsplit, H);
    hipLaunchKernelGGL((linattn_bwd_apply<T>), dim3(cdiv(N, tokb), B), dim3(H * 64), lds_b, st, (const T*)qkv, (const T*)dout, ctx, dctx, colstats, tvec, qstat, (T*)dqkv, N, d, tokb);
  });
  return ltu_check_launch();
}
