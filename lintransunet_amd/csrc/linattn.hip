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
// All four streaming kernels share one structure: 32-token tiles staged in LDS, the NEXT tile's 16-byte global
// loads issued (fully unrolled, compile-time counts) before the current tile is consumed, so HBM latency
// overlaps the matrix work inside a workgroup.  fp32 storage: fp32 tiles, v_mfma_f32_32x32x2_f32 (exact);
// bf16 storage: v_mfma_f32_32x32x16_bf16, and in the two reductions over tokens (kv_partial, dctx_partial) the
// tiles stay bf16 and the fragments come from transposing LDS reads.  The core is HBM-bound (AI = 16 FLOP/B).
#include "common.h"

#define DK 32
#define TOK 32
#define PART_STRIDE (64 + 1024)
#ifndef LA_TR_PAD
#define LA_TR_PAD 32          // row padding (bf16 elements) of the token tiles read with ds_read_b64_tr_b16
#endif

// value of the other wave half (lane ^ 32) combined with this lane's: sum / max in both lanes (v_permlane32_swap, no LDS crossbar)
__device__ __forceinline__ float xhalf_sum(float v) { return xhalf_combine<LtuAdd>(v); }
__device__ __forceinline__ float xhalf_max(float v) { return xhalf_combine<LtuMax>(v); }

// bf16 storage: the 32x32 products run on v_mfma_f32_32x32x16_bf16 (2 instructions of 8 passes per product instead of 16
// fp32 instructions of 16 passes, which had made the per-token kernels matrix-core-bound).  Lane (li, lh) supplies the 8
// k-elements 16*lh + 8*u + e of k-step u, i.e. elements 8u..8u+7 of the same per-lane arrays the fp32 path walks with s.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
__device__ __forceinline__ bf16x8 pack8(const float* v) {
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = (__bf16)v[e];
  return r;
}
template <typename T>
struct IsBf16 { static constexpr bool value = false; };
template <>
struct IsBf16<bf16_t> { static constexpr bool value = true; };

// ---- 16-byte global vector <-> fp32 LDS --------------------------------------------------------------------
template <typename T>
struct GVec;
template <>
struct GVec<float> {
  static constexpr int W = 4;
  typedef float4 reg;
  static __device__ __forceinline__ reg load(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ reg zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
  static __device__ __forceinline__ void to_lds(float* dst, reg v) { *reinterpret_cast<float4*>(dst) = v; }
  static __device__ __forceinline__ void from_lds(float* gdst, const float* src) {
    *reinterpret_cast<float4*>(gdst) = *reinterpret_cast<const float4*>(src);
  }
  static __device__ __forceinline__ reg pack_lds(const float* src) { return *reinterpret_cast<const float4*>(src); }
  static __device__ __forceinline__ void store_g(float* gdst, reg v) { *reinterpret_cast<float4*>(gdst) = v; }
};
template <>
struct GVec<bf16_t> {
  static constexpr int W = 8;
  typedef uint4 reg;
  static __device__ __forceinline__ reg load(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
  static __device__ __forceinline__ reg zero() { return make_uint4(0u, 0u, 0u, 0u); }
  static __device__ __forceinline__ void to_lds(float* dst, reg v) {
    float4 a, b;
    a.x = __uint_as_float(v.x << 16); a.y = __uint_as_float(v.x & 0xffff0000u);
    a.z = __uint_as_float(v.y << 16); a.w = __uint_as_float(v.y & 0xffff0000u);
    b.x = __uint_as_float(v.z << 16); b.y = __uint_as_float(v.z & 0xffff0000u);
    b.z = __uint_as_float(v.w << 16); b.w = __uint_as_float(v.w & 0xffff0000u);
    *reinterpret_cast<float4*>(dst) = a;
    *reinterpret_cast<float4*>(dst + 4) = b;
  }
  static __device__ __forceinline__ void from_lds(bf16_t* gdst, const float* src) {
    const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
    uint4 v;
    v.x = pack_bf16x2(a.x, a.y);
    v.y = pack_bf16x2(a.z, a.w);
    v.z = pack_bf16x2(b.x, b.y);
    v.w = pack_bf16x2(b.z, b.w);
    *reinterpret_cast<uint4*>(gdst) = v;
  }
  static __device__ __forceinline__ reg pack_lds(const float* src) {
    const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
    return make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
  }
  static __device__ __forceinline__ void store_g(bf16_t* gdst, reg v) { *reinterpret_cast<uint4*>(gdst) = v; }
};

// Stages TOK rows of ROWE elements (row r at gbase + (n0 + r)*gstride + goff) into LDS rows of stride LDSROW floats.
template <typename T, int D, int ROWE, int LDSROW>
struct RowTile {
  static constexpr int NTHR = 2 * D;
  static constexpr int W = GVec<T>::W;
  static constexpr int VPR = ROWE / W;                 // vectors per row
  static constexpr int NV = (TOK * VPR + NTHR - 1) / NTHR;
  typename GVec<T>::reg r[NV];
  __device__ __forceinline__ void load(const T* gbase, long long gstride, int goff, int n0, int n_end, int tid) {
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      const int idx = tid + p * NTHR;
      const int t = idx / VPR, c = (idx % VPR) * W;
      r[p] = (idx < TOK * VPR && n0 + t < n_end) ? GVec<T>::load(gbase + (long long)(n0 + t) * gstride + goff + c) : GVec<T>::zero();
    }
  }
  __device__ __forceinline__ void store(float* lds, int tid) const {
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      const int idx = tid + p * NTHR;
      const int t = idx / VPR, c = (idx % VPR) * W;
      if (idx < TOK * VPR) GVec<T>::to_lds(lds + t * LDSROW + c, r[p]);
    }
  }
  // the 16-byte pieces as they are (bf16 tiles read back with the transposing ds_read_b64_tr_b16; LDSROW in bf16 elements)
  __device__ __forceinline__ void store_raw(uint16_t* lds, int tid) const {
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      const int idx = tid + p * NTHR;
      const int t = idx / VPR, c = (idx % VPR) * W;
      if (idx < TOK * VPR) *reinterpret_cast<typename GVec<T>::reg*>(lds + t * LDSROW + c) = r[p];
    }
  }
};

// bf16 token-major tile [TOK][PITCH] -> MFMA operand fragment: lane (li, lh) receives rows (tokens) 16 u + 8 lh .. + 7 of column
// col0 + li.  Two transposing reads (4 rows x 16 columns per 16-lane group each).  A 32-lane half reads 4 consecutive rows x 64
// bytes per instruction: conflict-free when the row pitch is 64 bytes modulo 256 (tools/lds_conflicts.py), i.e. PITCH = row + 32.
typedef __attribute__((ext_vector_type(4))) short la_s16x4;
typedef __attribute__((address_space(3))) la_s16x4 la_lds_s16x4;
__device__ __forceinline__ int tr_lane_offset(int lane, int pitch) {
  const int gq = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  return (8 * (gq >> 1) + tq) * pitch + 16 * (gq & 1) + 4 * tp;
}
__device__ __forceinline__ bf16x8 tr_frag(const uint16_t* p, int pitch) {
  union { struct { la_s16x4 l, h; } s; bf16x8 v; } u;
  u.s.l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((la_lds_s16x4*)p);
  u.s.h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((la_lds_s16x4*)(p + 4 * pitch));
  return u.v;
}

// ------------------------------------------------------------------------------------------------ phase A
// grid (nsplit, B), block 2D.  part layout: [B][nsplit][H][32 (m) + 32 (s) + 1024 (accT[j][i])]
template <typename T, int D>
__global__ void __launch_bounds__(2 * D) linattn_kv_partial(const T* __restrict__ qkv, float* __restrict__ part, int N,
                                                           int tokens_per_split) {
  constexpr int H = D / DK, ROW = 2 * D;
  // fp32 storage: one fp32 tile; bf16 storage: two bf16 tiles [TOK][ROW + 32] (double-buffered: one barrier per tile)
  constexpr int PITCH = ROW + LA_TR_PAD;
  constexpr int SMEM_FLOATS = IsBf16<T>::value ? (2 * TOK * PITCH) / 2 : TOK * ROW;
  __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];   // k | v rows
  const int b = blockIdx.y, sp = blockIdx.x, nsplit = gridDim.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int n_begin = sp * tokens_per_split;
  const int n_end = min(N, n_begin + tokens_per_split);
  const T* base = qkv + (long long)b * N * 3 * D;

  float m_run = -INFINITY, s_run = 0.f;        // column i = li (both halves hold a copy; s is a per-half partial)
  f32x16 acc;                                  // accT[j][i]: row j by register, column i = li
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  if constexpr (IsBf16<T>::value) {
    // The tile stays bf16 in LDS and the operand fragments come from transposing reads: 8 LDS reads per tile and wave instead
    // of 48 scalar ones, half the LDS bytes, no conversion when the tile is written, and with two buffers a single barrier
    // per tile.  The arithmetic is the fp32 path's, operation for operation (bf16 -> fp32 is exact).
    uint16_t* sm = reinterpret_cast<uint16_t*>(smem);
    RowTile<T, D, ROW, PITCH> tile;
    const int koff = tr_lane_offset(lane, PITCH) + wave * DK, voff = koff + D;
    tile.load(base, 3 * D, D, n_begin, n_end, tid);
    tile.store_raw(sm, tid);
    __syncthreads();
    int buf = 0;
    for (int n0 = n_begin; n0 < n_end; n0 += TOK) {
      const bool more = n0 + TOK < n_end;
      if (more) tile.load(base, 3 * D, D, n0 + TOK, n_end, tid);
      const uint16_t* S = sm + buf * (TOK * PITCH);
      bf16x8 kf[2], vf[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        kf[u] = tr_frag(S + 16 * u * PITCH + koff, PITCH);
        vf[u] = tr_frag(S + 16 * u * PITCH + voff, PITCH);
      }
      const int ntok = min(TOK, n_end - n0);
      float kx[2][8];
      float mt = -INFINITY;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          kx[u][e] = (float)kf[u][e];
          if (16 * u + 8 * lh + e < ntok) mt = fmaxf(mt, kx[u][e]);
        }
      mt = xhalf_max(mt);
      const float m_new = fmaxf(m_run, mt);
      const float alpha = __expf(m_run - m_new);          // exp(-inf) = 0 on the first tile
      m_run = m_new;
      s_run *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] *= alpha;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float pe[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          pe[e] = 16 * u + 8 * lh + e < ntok ? __expf(kx[u][e] - m_new) : 0.f;
          s_run += pe[e];
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[u], pack8(pe), acc, 0, 0, 0);
      }
      if (more) tile.store_raw(sm + (buf ^ 1) * (TOK * PITCH), tid);
      __syncthreads();                          // the next tile is in place; everybody has left the current one
      buf ^= 1;
    }
  } else {
  RowTile<T, D, ROW, ROW> tile;
  tile.load(base, 3 * D, D, n_begin, n_end, tid);
  for (int n0 = n_begin; n0 < n_end; n0 += TOK) {
    __syncthreads();                            // previous tile fully consumed
    tile.store(smem, tid);
    __syncthreads();
    if (n0 + TOK < n_end) tile.load(base, 3 * D, D, n0 + TOK, n_end, tid);
    const int ntok = min(TOK, n_end - n0);
    const float* ks = smem + wave * DK;        // k of this head: ks[t*ROW + i]
    const float* vs = smem + D + wave * DK;
    float mt = -INFINITY;
#pragma unroll 4
    for (int t = lh; t < TOK; t += 2)
      if (t < ntok) mt = fmaxf(mt, ks[t * ROW + li]);
    mt = xhalf_max(mt);
    const float m_new = fmaxf(m_run, mt);
    const float alpha = __expf(m_run - m_new);          // exp(-inf) = 0 on the first tile
    m_run = m_new;
    s_run *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] *= alpha;
    // accT[j][i] += sum_t v[t][j] * p[t][i];  MFMA k-step = 2 tokens (lane half picks the token)
    if constexpr (IsBf16<T>::value) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {             // k-step = 16 tokens: lane half lh takes tokens 16u + 8lh + e
        float pe[8], ve[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int t = 16 * u + 8 * lh + e;
          pe[e] = 0.f; ve[e] = 0.f;
          if (t < ntok) {
            pe[e] = __expf(ks[t * ROW + li] - m_new);
            ve[e] = vs[t * ROW + li];
          }
          s_run += pe[e];
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8(ve), pack8(pe), acc, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int t0 = 0; t0 < TOK; t0 += 2) {
        const int t = t0 + lh;
        float p = 0.f, vv = 0.f;
        if (t < ntok) {
          p = __expf(ks[t * ROW + li] - m_new);
          vv = vs[t * ROW + li];
        }
        s_run += p;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, p, acc, 0, 0, 0);
      }
    }
  }
  }
  const float s_tot = xhalf_sum(s_run);
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

// Merge of (m, s, accT) partials, two levels so that more than B*H workgroups share the work:
//   level 1: grid (ngroups, B*H): partials [g*group, (g+1)*group) of `part` -> one partial in `part2`
//   level 2: grid (1, B*H) with final != 0: the ngroups partials -> colmax, colsum, ctx (normalised)
// block 1024: thread (i = tid&31, j = tid>>5) (coalesced over accT[j][i]).
#define KVC_GROUP 16
__global__ void __launch_bounds__(1024) linattn_kv_combine(const float* __restrict__ part, int nsplit, int group,
                                                           float* __restrict__ part2, float* __restrict__ stats,
                                                           float* __restrict__ ctx, int H, int final) {
  const int bh = blockIdx.y, b = bh / H, h = bh % H, g = blockIdx.x;
  const int i = threadIdx.x & 31, j = threadIdx.x >> 5;
  const float* p0 = part + ((long long)b * nsplit * H + h) * PART_STRIDE;
  const long long sstride = (long long)H * PART_STRIDE;
  const int s0 = g * group, s1 = min(nsplit, s0 + group);
  // at most KVC_GROUP splits per workgroup: every load of a pass is issued before the first use (two round trips in all, not one
  // per four splits)
  float mv[KVC_GROUP];
#pragma unroll
  for (int q = 0; q < KVC_GROUP; ++q) mv[q] = s0 + q < s1 ? p0[(s0 + q) * sstride + i] : -INFINITY;
  float m = mv[0];
#pragma unroll
  for (int q = 1; q < KVC_GROUP; ++q) m = fmaxf(m, mv[q]);
  float sv[KVC_GROUP], av[KVC_GROUP];
#pragma unroll
  for (int q = 0; q < KVC_GROUP; ++q) {
    const float* p = p0 + (s0 + q) * sstride;
    sv[q] = s0 + q < s1 ? p[32 + i] : 0.f;
    av[q] = s0 + q < s1 ? p[64 + j * 32 + i] : 0.f;
  }
  float ssum = 0.f, a = 0.f;
#pragma unroll
  for (int q = 0; q < KVC_GROUP; ++q) {
    const float f = __expf(mv[q] - m);          // exp(-inf) = 0 for the absent splits
    ssum += sv[q] * f;
    a += av[q] * f;
  }
  if (final) {
    ctx[(long long)bh * 1024 + i * 32 + j] = a / ssum;
    if (j == 0) {
      stats[bh * 64 + i] = m;
      stats[bh * 64 + 32 + i] = ssum;
    }
  } else {
    float* o = part2 + (((long long)b * gridDim.x + g) * H + h) * PART_STRIDE;
    o[64 + j * 32 + i] = a;
    if (j == 0) {
      o[i] = m;
      o[32 + i] = ssum;
    }
  }
}

// The same merge as ONE launch (two dependent 5 us launches were pure latency at every level): grid (8, B*H), block 256.  Workgroup w
// produces accT rows j = 4w .. 4w+3 (128 elements) from ALL nsplit partials; every workgroup recomputes the 32 column maxima and
// sums itself (2 x nsplit x 128 B from L2) and keeps the nsplit x 32 rescale factors exp(m_s - m) in LDS.
template <int NB>
__global__ void __launch_bounds__(256) linattn_kv_combine1(const float* __restrict__ part, int nsplit, float* __restrict__ stats,
                                                           float* __restrict__ ctx, int H) {
  extern __shared__ __attribute__((aligned(16))) float fs[];       // [nsplit][32]
  __shared__ float red[8][32];
  __shared__ float4 red4[7][32];
  const int bh = blockIdx.y, b = bh / H, h = bh % H;
  const float* p0 = part + ((long long)b * nsplit * H + h) * PART_STRIDE;
  const long long sstride = (long long)H * PART_STRIDE;
  const int l = threadIdx.x & 31, grp = threadIdx.x >> 5;
  // Up to 256 splits (32 per thread) the three passes are three BATCHES of loads: the maxima and sums of a thread's splits are
  // requested together and stay in registers for the rescale factors, then all of its accT quads are requested together.  As three
  // loops of 8 loads in flight each pass was four dependent L2 round trips (12.7 us at 256 splits for 9 MB).
  const bool fast = nsplit <= 8 * NB;            // NB = batch depth, chosen by the host from the split count
  float m = -INFINITY, ss = 0.f;
  const int e0 = blockIdx.x * 128 + l * 4, c0 = e0 & 31;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (fast) {
    float pm[NB], ps[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int sp = grp + 8 * k, sc = sp < nsplit ? sp : 0;
      pm[k] = p0[sc * sstride + l];
      ps[k] = p0[sc * sstride + 32 + l];
    }
#pragma unroll
    for (int k = 0; k < NB; ++k)
      if (grp + 8 * k < nsplit) m = fmaxf(m, pm[k]);
    red[grp][l] = m;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) m = fmaxf(m, red[q][l]);
    __syncthreads();
    float4 v[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {             // the accT quads: requested before the exponentials below are needed
      const int sp = grp + 8 * k, sc = sp < nsplit ? sp : 0;
      v[k] = *reinterpret_cast<const float4*>(p0 + sc * sstride + 64 + e0);
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int sp = grp + 8 * k;
      if (sp < nsplit) {
        const float f = __expf(pm[k] - m);
        fs[sp * 32 + l] = f;
        ss += ps[k] * f;
      }
    }
    red[grp][l] = ss;
    __syncthreads();                           // also: every factor is in LDS
    ss = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) ss += red[q][l];        // same order in every thread and workgroup: one value per column
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int sp = grp + 8 * k;
      if (sp < nsplit) {
        const float4 f = *reinterpret_cast<const float4*>(fs + sp * 32 + c0);
        a.x += v[k].x * f.x; a.y += v[k].y * f.y; a.z += v[k].z * f.z; a.w += v[k].w * f.w;
      }
    }
  } else {
#pragma unroll 8
    for (int s = grp; s < nsplit; s += 8) m = fmaxf(m, p0[s * sstride + l]);
    red[grp][l] = m;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) m = fmaxf(m, red[q][l]);
    __syncthreads();
#pragma unroll 8
    for (int s = grp; s < nsplit; s += 8) {
      const float f = __expf(p0[s * sstride + l] - m);
      fs[s * 32 + l] = f;
      ss += p0[s * sstride + 32 + l] * f;
    }
    red[grp][l] = ss;
    __syncthreads();                             // also: every factor is in LDS
    ss = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) ss += red[q][l];          // same order in every thread and workgroup: one value per column
    // accT slice: thread (grp, l) takes the float4 at element e0 = 128 w + 4 l of splits grp, grp + 8, ...; its 4 columns are 4 (l & 7) ..
#pragma unroll 8
    for (int s = grp; s < nsplit; s += 8) {
      const float4 v = *reinterpret_cast<const float4*>(p0 + s * sstride + 64 + e0);
      const float4 f = *reinterpret_cast<const float4*>(fs + s * 32 + c0);
      a.x += v.x * f.x; a.y += v.y * f.y; a.z += v.z * f.z; a.w += v.w * f.w;
    }
  }
  if (grp > 0) red4[grp - 1][l] = a;
  // column sums of this thread's 4 columns (threads with l < 8 of group 0 hold all 32 between them after the exchange below)
  __shared__ float scol[32];
  if (grp == 0) scol[l] = ss;
  __syncthreads();
  if (grp != 0) return;
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    const float4 v = red4[q][l];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  const int j = e0 >> 5;                        // accT[j][c0 .. c0+3] -> ctx[c][j] / colsum[c]
  float* o = ctx + (long long)bh * 1024 + j;
  o[c0 * 32] = a.x / scol[c0]; o[(c0 + 1) * 32] = a.y / scol[c0 + 1]; o[(c0 + 2) * 32] = a.z / scol[c0 + 2]; o[(c0 + 3) * 32] = a.w / scol[c0 + 3];
  if (blockIdx.x == 0) {
    stats[bh * 64 + l] = m;
    stats[bh * 64 + 32 + l] = ss;
  }
}

// ------------------------------------------------------------------------------------------------ phase B
// grid (ceil(N/tokb), B), block 2D; each block handles tokb tokens in tiles of 32.
// out [B*N][d];  qstat [B*N][H][2] = (row max, 1/(rowsum*sqrt(32)))
//
// MFMA k-index convention of the per-token kernels: in k-step s the lane half `lh` supplies channel
// ch(s, lh) = 16*lh + s (any bijection works as long as both operands use it).  A lane therefore needs 16
// CONTIGUOUS channels of its token row, fetched with four 128-bit LDS reads from 16-byte aligned rows of
// stride d+4 floats (lane l starts at bank 4*l mod 64: conflict-free for ds_read_b128).
template <typename T, int D>
__global__ void __launch_bounds__(2 * D) linattn_apply(const T* __restrict__ qkv, const float* __restrict__ ctx, T* __restrict__ out,
                                                      float* __restrict__ qstat, int N, int tokb) {
  constexpr int H = D / DK, LDQ = D + 4, NTHR = 2 * D, W = GVec<T>::W;
  __shared__ __attribute__((aligned(16))) float smem[TOK * LDQ];   // q tile, reused for the out tile
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  float cb[16];                                  // B operand: ctx[i = ch(s,lh)][j = li]
  const float* cx = ctx + ((long long)b * H + wave) * 1024;
#pragma unroll
  for (int s = 0; s < 16; ++s) cb[s] = cx[(16 * lh + s) * 32 + li];
  const bf16x8 cbb[2] = {pack8(cb), pack8(cb + 8)};
  const float rs = 0.17677669529663688110f;      // 1/sqrt(32)
  const T* base = qkv + (long long)b * N * 3 * D;

  const int n_begin = blockIdx.x * tokb;
  const int n_end = min(N, n_begin + tokb);
  RowTile<T, D, D, LDQ> tile;
  tile.load(base, 3 * D, 0, n_begin, n_end, tid);
  for (int n0 = n_begin; n0 < n_end; n0 += TOK) {
    __syncthreads();
    tile.store(smem, tid);
    __syncthreads();
    if (n0 + TOK < n_end) tile.load(base, 3 * D, 0, n0 + TOK, n_end, tid);
    float a[16];
    const float* qrow = &smem[li * LDQ + wave * DK + 16 * lh];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const float4 v = *reinterpret_cast<const float4*>(qrow + 4 * q4);
      a[4 * q4] = v.x; a[4 * q4 + 1] = v.y; a[4 * q4 + 2] = v.z; a[4 * q4 + 3] = v.w;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < 16; ++s) mx = fmaxf(mx, a[s]);
    mx = xhalf_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      a[s] = __expf(a[s] - mx);
      sum += a[s];
    }
    sum = xhalf_sum(sum);
    const float inv = rs / sum;
    if (lh == 0 && n0 + li < n_end) {
      float* qs = qstat + (((long long)b * N + n0 + li) * H + wave) * 2;
      qs[0] = mx;
      qs[1] = inv;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if constexpr (IsBf16<T>::value) {
#pragma unroll
      for (int s = 0; s < 16; ++s) a[s] *= inv;
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8(a), cbb[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8(a + 8), cbb[1], acc, 0, 0, 0);
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s] * inv, cb[s], acc, 0, 0, 0);
    }
    __syncthreads();   // everyone is done reading the q tile
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = (r & 3) + 8 * (r >> 2) + 4 * lh;
      smem[t * LDQ + wave * DK + li] = acc[r];
    }
    __syncthreads();
    constexpr int VPR = D / W;
#pragma unroll
    for (int p = 0; p < (TOK * VPR + NTHR - 1) / NTHR; ++p) {
      const int idx = tid + p * NTHR;
      const int t = idx / VPR, c = (idx % VPR) * W;
      if (idx < TOK * VPR && n0 + t < n_end) GVec<T>::from_lds(out + ((long long)b * N + n0 + t) * D + c, &smem[t * LDQ + c]);
    }
  }
}

// bf16 storage, wave-private form of phase B.  Everything phase B does is per token with one constant 32x32 matrix per
// (sample, head), so no two waves have to meet: a wave owns 32 tokens x 4 heads (256 bytes of every token row), stages them in
// its own 8.5 KB of LDS with row-contiguous 16-byte loads (a wave instruction covers 4 whole row segments: the vector L1 works
// per line, and a lane-per-token access would touch 32 lines for 512 bytes), and walks the 4 heads: lane (t = li, lh) reads 16
// channels of its token, row softmax with one cross-half exchange, the product transposed - outT[j][t] = sum_i ctx[i][j] qs[t][i],
// A = ctx^T held in registers for all 4 heads - so that the lane receives quads of its own token, which go back into the tile in
// place; the tile leaves as it came, in whole row segments.  No barrier (LDS operations of one wave execute in order), the next
// tile's loads in flight during the arithmetic.  The fp32-LDS kernel above (4 barriers per tile, 240 workgroups) ran at 2 TB/s.
#define LAW_HEADS 4                              // heads per wave tile
#define LAW_LD (LAW_HEADS * DK + 8)              // LDS row stride in bf16 elements (272 bytes: 16-byte aligned, banks rotate by 4 per row)

template <int D>
__global__ void __launch_bounds__(256) linattn_apply_rows(const uint16_t* __restrict__ qkv, const float* __restrict__ ctx,
                                                         uint16_t* __restrict__ out, float* __restrict__ qstat, int N, int tiles_per_wave) {
  constexpr int H = D / DK, CG = H / LAW_HEADS;          // column groups of 4 heads per token row
  __shared__ __attribute__((aligned(16))) uint16_t smem[4 * TOK * LAW_LD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lh = lane >> 5;
  uint16_t* tile = smem + wave * TOK * LAW_LD;
  // work item of the wave: (sample b, column group cg, first tile)
  const int b = blockIdx.y;
  const int gw = blockIdx.x * 4 + wave;                  // global wave index within the sample
  const int cg = gw % CG;
  const int n_begin = (gw / CG) * tiles_per_wave * TOK;
  const int n_end = min(N, n_begin + tiles_per_wave * TOK);
  if (n_begin >= N) return;
  // A operands: ctx^T[j = li][i = 16 lh + 8 u + e] of the 4 heads
  bf16x8 ca[LAW_HEADS][2];
#pragma unroll
  for (int hh = 0; hh < LAW_HEADS; ++hh) {
    const float* cx = ctx + ((long long)b * H + cg * LAW_HEADS + hh) * 1024;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = cx[(16 * lh + 8 * u + e) * 32 + li];
      ca[hh][u] = pack8(t);
    }
  }
  const float rs = 0.17677669529663688110f;      // 1/sqrt(32)
  // row-contiguous chunk map: instruction p covers rows 4p .. 4p+3, 16 lanes x 16 bytes each
  const int cr = lane >> 4, cc = (lane & 15) * 8;
  const uint16_t* qb = qkv + (long long)b * N * 3 * D + cg * (LAW_HEADS * DK) + cc;
  uint16_t* ob = out + (long long)b * N * D + cg * (LAW_HEADS * DK) + cc;
  uint4 nx[8];
  auto fetch = [&](int n0) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int n = n0 + 4 * p + cr;
      nx[p] = make_uint4(0u, 0u, 0u, 0u);
      if (n < n_end) nx[p] = *reinterpret_cast<const uint4*>(qb + (long long)n * 3 * D);
    }
  };
  fetch(n_begin);
  for (int n0 = n_begin; n0 < n_end; n0 += TOK) {
#pragma unroll
    for (int p = 0; p < 8; ++p) *reinterpret_cast<uint4*>(tile + (4 * p + cr) * LAW_LD + cc) = nx[p];
    if (n0 + TOK < n_end) fetch(n0 + TOK);
    float2 st[LAW_HEADS];
#pragma unroll
    for (int hh = 0; hh < LAW_HEADS; ++hh) {
      const uint16_t* qr = tile + li * LAW_LD + hh * DK + 16 * lh;
      const uint4 v0 = *reinterpret_cast<const uint4*>(qr), v1 = *reinterpret_cast<const uint4*>(qr + 8);
      const uint32_t w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      float a[16];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        a[2 * k] = __uint_as_float(w[k] << 16);
        a[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
      }
      float mx = a[0];
#pragma unroll
      for (int s = 1; s < 16; ++s) mx = fmaxf(mx, a[s]);
      mx = xhalf_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        a[s] = __expf(a[s] - mx);
        sum += a[s];
      }
      sum = xhalf_sum(sum);
      const float inv = rs / sum;
      st[hh] = make_float2(mx, inv);
#pragma unroll
      for (int s = 0; s < 16; ++s) a[s] *= inv;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[hh][0], pack8(a), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[hh][1], pack8(a + 8), acc, 0, 0, 0);
      // the lane's quads of token li, head hh: channels 8 q + 4 lh (in place: this head's q values have been read by every lane)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint2 v;
        v.x = pack_bf16x2(acc[4 * q], acc[4 * q + 1]);
        v.y = pack_bf16x2(acc[4 * q + 2], acc[4 * q + 3]);
        *reinterpret_cast<uint2*>(tile + li * LAW_LD + hh * DK + 8 * q + 4 * lh) = v;
      }
    }
    // per-token softmax statistics of the 4 heads: 32 contiguous bytes per token
    if (lh == 0 && n0 + li < n_end) {
      float4* qs = reinterpret_cast<float4*>(qstat + (((long long)b * N + n0 + li) * H + cg * LAW_HEADS) * 2);
      qs[0] = make_float4(st[0].x, st[0].y, st[1].x, st[1].y);
      qs[1] = make_float4(st[2].x, st[2].y, st[3].x, st[3].y);
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int n = n0 + 4 * p + cr;
      const uint4 v = *reinterpret_cast<const uint4*>(tile + (4 * p + cr) * LAW_LD + cc);
      if (n < n_end) *reinterpret_cast<uint4*>(ob + (long long)n * D) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward pass 1
// dctxT[j][i] = sum_n dO[n][j] * qs[n][i];  same split structure as phase A.
// part layout: [B][nsplit][H][1024 (accT[j][i])]
template <typename T, int D>
__global__ void __launch_bounds__(2 * D) linattn_dctx_partial(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                             const float* __restrict__ qstat, float* __restrict__ part, int N,
                                                             int tokens_per_split) {
  constexpr int H = D / DK;
  extern __shared__ __attribute__((aligned(16))) float smem[];   // q tile | dO tile | row stats
  float* sq = smem;
  float* sg = smem + TOK * D;
  float* st = smem + 2 * TOK * D;
  const int b = blockIdx.y, sp = blockIdx.x, nsplit = gridDim.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int n_begin = sp * tokens_per_split;
  const int n_end = min(N, n_begin + tokens_per_split);
  const T* qb = qkv + (long long)b * N * 3 * D;
  const T* gb = dout + (long long)b * N * D;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  RowTile<T, D, D, D> tq, tg;
  // the row statistics travel with the tile (one float per thread): a load issued after the prefetch below would have to wait
  // for the whole prefetch as well (vector-memory loads retire in order)
  static_assert(TOK * H * 2 <= 2 * D, "one statistics float per thread");
  auto load_stat = [&](int n0) {
    return (tid < TOK * H * 2 && n0 + tid / (H * 2) < n_end) ? qstat[((long long)b * N + n0) * H * 2 + tid] : 0.f;
  };
  if constexpr (IsBf16<T>::value) {
    // bf16 tiles + transposing reads, double-buffered (see linattn_kv_partial): LDS [2][q tile | dO tile][TOK][D + 32], then
    // [2][TOK * H * 2] row statistics
    constexpr int PQ = D + LA_TR_PAD;
    uint16_t* sm = reinterpret_cast<uint16_t*>(smem);
    float* stf = reinterpret_cast<float*>(sm + 4 * TOK * PQ);
    RowTile<T, D, D, PQ> bq, bg;
    const int off = tr_lane_offset(lane, PQ) + wave * DK;
    bq.load(qb, 3 * D, 0, n_begin, n_end, tid);
    bg.load(gb, D, 0, n_begin, n_end, tid);
    float stn = load_stat(n_begin);
    bq.store_raw(sm, tid);
    bg.store_raw(sm + TOK * PQ, tid);
    if (tid < TOK * H * 2) stf[tid] = stn;
    __syncthreads();
    int buf = 0;
    for (int n0 = n_begin; n0 < n_end; n0 += TOK) {
      const bool more = n0 + TOK < n_end;
      if (more) {
        bq.load(qb, 3 * D, 0, n0 + TOK, n_end, tid);
        bg.load(gb, D, 0, n0 + TOK, n_end, tid);
        stn = load_stat(n0 + TOK);
      }
      const uint16_t* S = sm + buf * (2 * TOK * PQ);
      const float* ST = stf + buf * (TOK * H * 2);
      const int ntok = min(TOK, n_end - n0);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const bf16x8 qf = tr_frag(S + 16 * u * PQ + off, PQ), gf = tr_frag(S + (TOK + 16 * u) * PQ + off, PQ);
        float qe[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int t = 16 * u + 8 * lh + e;
          const float2 rs = *reinterpret_cast<const float2*>(ST + (t * H + wave) * 2);
          qe[e] = t < ntok ? __expf((float)qf[e] - rs.x) * rs.y : 0.f;
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gf, pack8(qe), acc, 0, 0, 0);
      }
      if (more) {
        uint16_t* Sn = sm + (buf ^ 1) * (2 * TOK * PQ);
        bq.store_raw(Sn, tid);
        bg.store_raw(Sn + TOK * PQ, tid);
        if (tid < TOK * H * 2) stf[(buf ^ 1) * (TOK * H * 2) + tid] = stn;
      }
      __syncthreads();
      buf ^= 1;
    }
  } else {
  tq.load(qb, 3 * D, 0, n_begin, n_end, tid);
  tg.load(gb, D, 0, n_begin, n_end, tid);
  float stn = load_stat(n_begin);
  for (int n0 = n_begin; n0 < n_end; n0 += TOK) {
    __syncthreads();
    tq.store(sq, tid);
    tg.store(sg, tid);
    if (tid < TOK * H * 2) st[tid] = stn;
    __syncthreads();
    if (n0 + TOK < n_end) {
      tq.load(qb, 3 * D, 0, n0 + TOK, n_end, tid);
      tg.load(gb, D, 0, n0 + TOK, n_end, tid);
      stn = load_stat(n0 + TOK);
    }
    const int ntok = min(TOK, n_end - n0);
    if constexpr (IsBf16<T>::value) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float qe[8], ge[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int t = 16 * u + 8 * lh + e;
          qe[e] = 0.f; ge[e] = 0.f;
          if (t < ntok) {
            qe[e] = __expf(sq[t * D + wave * DK + li] - st[(t * H + wave) * 2]) * st[(t * H + wave) * 2 + 1];
            ge[e] = sg[t * D + wave * DK + li];
          }
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8(ge), pack8(qe), acc, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int t0 = 0; t0 < TOK; t0 += 2) {
        const int t = t0 + lh;
        float qv = 0.f, gv = 0.f;
        if (t < ntok) {
          qv = __expf(sq[t * D + wave * DK + li] - st[(t * H + wave) * 2]) * st[(t * H + wave) * 2 + 1];
          gv = sg[t * D + wave * DK + li];
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(gv, qv, acc, 0, 0, 0);
      }
    }
  }
  }
  float* out = part + (((long long)b * nsplit + sp) * H + wave) * 1024;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int j = (r & 3) + 8 * (r >> 2) + 4 * lh;
    out[j * 32 + li] = acc[r];
  }
}

// grid (8, B*H), block 256: dctx[bh][i][j] = sum_s partT[s][j][i].  Workgroup w folds partT rows j = 4w .. 4w+3 (32 float4 outputs
// x 8 thread groups that share the splits).  One workgroup per (sample, head) pulled 0.9 MB through a single CU (8.5 us); the
// Jacobian term tvec[i] = sum_j dctx[i][j] ctx[i][j] moved into linattn_bwd_apply's prologue (its lanes hold both rows).
// NB = splits per thread requested together (the host picks it from the split count: one batch up to 8 NB splits)
template <int NB>
__global__ void __launch_bounds__(256) linattn_dctx_combine(const float* __restrict__ part, float* __restrict__ dctx, int nsplit, int H) {
  __shared__ float4 red4[7][32];
  const int bh = blockIdx.y, b = bh / H, h = bh % H;
  const float* p0 = part + ((long long)b * nsplit * H + h) * 1024 + blockIdx.x * 128;
  const int l = threadIdx.x & 31, grp = threadIdx.x >> 5;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s0 = grp; s0 < nsplit; s0 += 8 * NB) {
    float4 v[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int sp = s0 + 8 * k;
      v[k] = *reinterpret_cast<const float4*>(p0 + (long long)(sp < nsplit ? sp : s0) * H * 1024 + l * 4);
    }
#pragma unroll
    for (int k = 0; k < NB; ++k)
      if (s0 + 8 * k < nsplit) { a.x += v[k].x; a.y += v[k].y; a.z += v[k].z; a.w += v[k].w; }
  }
  if (grp > 0) red4[grp - 1][l] = a;
  __syncthreads();
  if (grp != 0) return;
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    const float4 v = red4[q][l];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  const int e = blockIdx.x * 128 + l * 4, jj = e >> 5, ii = e & 31;      // partT[jj][ii .. ii+3] -> dctx[ii ..][jj]
  float* o = dctx + (long long)bh * 1024 + jj;
  o[ii * 32] = a.x; o[(ii + 1) * 32] = a.y; o[(ii + 2) * 32] = a.z; o[(ii + 3) * 32] = a.w;
}

// ------------------------------------------------------------------------------------------------ backward pass 2
// per 32-token tile and head (tokens on the lanes, channel index in the registers):
//   dqsT[i][t] = sum_j ctx[i][j] dO[t][j]      dq[t][i] = qs[t][i] (dqsT[i][t] - sum_i' p[t][i'] dqsT[i'][t])
//   dvT[j][t]  = sum_i dctx[i][j] P[t][i]
//   dPT[i][t]  = sum_j dctx[i][j] v[t][j]      dk[t][i] = P[t][i] (dPT[i][t] - tvec[i])
// writes dqkv [B*N][3d] (dq | dk | dv).  Same k-index convention and LDS row format as linattn_apply.
template <typename T, int D>
__global__ void __launch_bounds__(2 * D, 2) linattn_bwd_apply(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                          const float* __restrict__ ctx, const float* __restrict__ dctx,
                                                          const float* __restrict__ stats, const float* __restrict__ tvec,
                                                          const float* __restrict__ qstat, T* __restrict__ dqkv, int N, int tokb) {
  constexpr int H = D / DK, LD3 = 3 * D + 4, LD1 = D + 4, NTHR = 2 * D, W = GVec<T>::W;
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [32][LD3] qkv tile (reused for dqkv), [32][LD1] dO, [H][96] stats
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  float* gt = smem + TOK * LD3;
  float* cs = gt + TOK * LD1 + wave * 96;      // this head's column max | 1/colsum | tvec
  const long long bh = (long long)b * H + wave;
  if (lane < 32) {
    cs[lane] = stats[bh * 64 + lane];
    cs[32 + lane] = 1.f / stats[bh * 64 + 32 + lane];
  }
  // A operands (rows on lanes), k-index ch(s,lh) = 16*lh + s:
  //   ctxA[s] = ctx[i=li][j=ch];  dcA[s] = dctx[i=li][j=ch];  dcT[s] = dctx[i=ch][j=li]
  float ctxA[16], dcA[16], dcT[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    ctxA[s] = ctx[bh * 1024 + li * 32 + 16 * lh + s];
    dcA[s] = dctx[bh * 1024 + li * 32 + 16 * lh + s];
    dcT[s] = dctx[bh * 1024 + (16 * lh + s) * 32 + li];
  }
  {
    // tvec[i] = sum_j dctx[i][j] ctx[i][j] (the column-softmax Jacobian term), from the fp32 values: lane (i = li, lh) holds 16 j
    float t = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) t += ctxA[s] * dcA[s];
    t = xhalf_sum(t);
    if (lane < 32) cs[64 + lane] = t;
  }
  const bf16x8 ctxB[2] = {pack8(ctxA), pack8(ctxA + 8)}, dcTB[2] = {pack8(dcT), pack8(dcT + 8)};
  // dk = P (dP - tvec) cancels almost completely (the key gradients are ~1e-3 of the others), and tvec comes from the fp32
  // dctx: dP must use the same values.  dctx is therefore split into three bf16 pieces (8 + 8 + 8 mantissa bits = exact), v is
  // exact in bf16 already, so the three products accumulate to the fp32 result at 6 bf16 instructions instead of 16 fp32 ones.
  bf16x8 dcAB[3][2];
  {
    float r1[16], r2[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      r1[s] = dcA[s] - (float)(__bf16)dcA[s];
      r2[s] = r1[s] - (float)(__bf16)r1[s];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) { dcAB[0][u] = pack8(dcA + 8 * u); dcAB[1][u] = pack8(r1 + 8 * u); dcAB[2][u] = pack8(r2 + 8 * u); }
  }
  const T* qb = qkv + (long long)b * N * 3 * D;
  const T* gb = dout + (long long)b * N * D;

  const int n_begin = blockIdx.x * tokb;
  const int n_end = min(N, n_begin + tokb);
  RowTile<T, D, 3 * D, LD3> tq;
  RowTile<T, D, D, LD1> tg;
  // (row max, scaled inverse sum) of token n0 + li travel with the tile prefetch: loaded after it they would wait for it
  // (unconditional, from a clamped row: a test around the load makes hipcc wait for it - and for the whole tile prefetch issued
  // just before it - where it stands)
  auto load_qs = [&](int n0) {
    return *reinterpret_cast<const float2*>(qstat + (((long long)b * N + min(n0 + li, n_end - 1)) * H + wave) * 2);
  };
  // Order inside an iteration: the next tile's loads are requested before the arithmetic; the tile's outputs are read back from
  // the staging into registers, the NEXT tile goes to LDS (its loads are older than any pending store), and only then are the
  // outputs stored.  vmcnt is one in-order counter and hipcc waits vmcnt(0) at the loop head: stores issued at the end of an
  // iteration had their acknowledgement waited for before every tile, with one wave per SIMD to hide it.
  tq.load(qb, 3 * D, 0, n_begin, n_end, tid);
  tg.load(gb, D, 0, n_begin, n_end, tid);
  float2 qsn = load_qs(n_begin);
  tq.store(smem, tid);
  tg.store(gt, tid);
  __syncthreads();
  for (int n0 = n_begin; n0 < n_end; n0 += TOK) {
    const float rmax = qsn.x, rinv = qsn.y;
    const bool more = n0 + TOK < n_end;
    if (more) {
      tq.load(qb, 3 * D, 0, n0 + TOK, n_end, tid);
      tg.load(gb, D, 0, n0 + TOK, n_end, tid);
      qsn = load_qs(n0 + TOK);
    }
    const float* qrow = &smem[li * LD3 + wave * DK];
    const float* krow = qrow + D;
    const float* vrow = qrow + 2 * D;
    const float* grow = &gt[li * LD1 + wave * DK];

    // B operands with the token on the lane: channels 16*lh + s
    float gj[16], kk[16], vj[16];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const float4 a = *reinterpret_cast<const float4*>(grow + 16 * lh + 4 * q4);
      const float4 c = *reinterpret_cast<const float4*>(krow + 16 * lh + 4 * q4);
      const float4 e = *reinterpret_cast<const float4*>(vrow + 16 * lh + 4 * q4);
      gj[4 * q4] = a.x; gj[4 * q4 + 1] = a.y; gj[4 * q4 + 2] = a.z; gj[4 * q4 + 3] = a.w;
      kk[4 * q4] = c.x; kk[4 * q4 + 1] = c.y; kk[4 * q4 + 2] = c.z; kk[4 * q4 + 3] = c.w;
      vj[4 * q4] = e.x; vj[4 * q4 + 1] = e.y; vj[4 * q4 + 2] = e.z; vj[4 * q4 + 3] = e.w;
    }
    f32x16 aq, av, ak;
#pragma unroll
    for (int r = 0; r < 16; ++r) { aq[r] = 0.f; av[r] = 0.f; ak[r] = 0.f; }
    if constexpr (IsBf16<T>::value) {
#pragma unroll
      for (int s = 0; s < 16; ++s) kk[s] = __expf(kk[s] - cs[16 * lh + s]) * cs[32 + 16 * lh + s];      // P[t][i=ch]
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        aq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ctxB[u], pack8(gj + 8 * u), aq, 0, 0, 0);
        av = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dcTB[u], pack8(kk + 8 * u), av, 0, 0, 0);
        const bf16x8 vb = pack8(vj + 8 * u);
#pragma unroll
        for (int pc = 2; pc >= 0; --pc) ak = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dcAB[pc][u], vb, ak, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int ch = 16 * lh + s;
        const float pk = __expf(kk[s] - cs[ch]) * cs[32 + ch];                  // P[t][i=ch]
        aq = __builtin_amdgcn_mfma_f32_32x32x2f32(ctxA[s], gj[s], aq, 0, 0, 0);  // dqsT[i][t]
        av = __builtin_amdgcn_mfma_f32_32x32x2f32(dcT[s], pk, av, 0, 0, 0);      // dvT[j][t]   (A[row=j][kk=i] = dctx[i][j])
        ak = __builtin_amdgcn_mfma_f32_32x32x2f32(dcA[s], vj[s], ak, 0, 0, 0);   // dPT[i][t]
      }
    }
    // registers hold rows idx(r) = (r&3) + 8*(r>>2) + 4*half of token li: 4 groups of 4 contiguous channels
    float dot = 0.f;
    float qsv[16], pkv[16];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int i0 = 8 * g4 + 4 * lh;
      const float4 qv = *reinterpret_cast<const float4*>(qrow + i0);
      const float4 kv = *reinterpret_cast<const float4*>(krow + i0);
      const float qa[4] = {qv.x, qv.y, qv.z, qv.w}, ka[4] = {kv.x, kv.y, kv.z, kv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g4 + e;
        qsv[r] = __expf(qa[e] - rmax) * rinv;                 // qs[t][i] = p/sqrt(32)
        pkv[r] = __expf(ka[e] - cs[i0 + e]) * cs[32 + i0 + e];
        dot += qsv[r] * aq[r];
      }
    }
    dot = xhalf_sum(dot);
    dot *= 5.65685424949238019521f;                           // sum_i p_i dqs_i = sqrt(32) * sum_i qs_i dqs_i
    __syncthreads();   // all waves finished reading the staged tile
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int i0 = 8 * g4 + 4 * lh;
      float4 oq, ok4, ov;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g4 + e;
        f4at(oq, e) = qsv[r] * (aq[r] - dot);
        f4at(ok4, e) = pkv[r] * (ak[r] - cs[64 + i0 + e]);
        f4at(ov, e) = av[r];
      }
      *reinterpret_cast<float4*>(&smem[li * LD3 + wave * DK + i0]) = oq;
      *reinterpret_cast<float4*>(&smem[li * LD3 + D + wave * DK + i0]) = ok4;
      *reinterpret_cast<float4*>(&smem[li * LD3 + 2 * D + wave * DK + i0]) = ov;
    }
    __syncthreads();
    constexpr int VPR = 3 * D / W, NOUT = (TOK * VPR + NTHR - 1) / NTHR;
    typename GVec<T>::reg ovec[NOUT];
#pragma unroll
    for (int p = 0; p < NOUT; ++p) {
      const int idx = min(tid + p * NTHR, TOK * VPR - 1);
      ovec[p] = GVec<T>::pack_lds(&smem[(idx / VPR) * LD3 + (idx % VPR) * W]);
    }
    __syncthreads();                       // the staging has been read: the region takes the next tile
    if (more) {
      tq.store(smem, tid);
      tg.store(gt, tid);
    }
#pragma unroll
    for (int p = 0; p < NOUT; ++p) {
      const int idx = tid + p * NTHR;
      const int t = idx / VPR, c = (idx % VPR) * W;
      if (idx < TOK * VPR && n0 + t < n_end) GVec<T>::store_g(dqkv + ((long long)b * N + n0 + t) * 3 * D + c, ovec[p]);
    }
    __syncthreads();                       // the next tile is in place
  }
}

// ------------------------------------------------------------------------------------------------ host side
static int pick_splits(int B, int N, int d, int* tokens_per_split) {
  // ~256..512 streaming workgroups in total, whole 32-token tiles each.  d = 256 (512-thread workgroups, a 35 KB partial per split):
  // 256 - with the one-batch merge fewer, longer splits win there (21 504 x 256: forward 25.8 -> 22.5 us, backward 36.9 -> 34.8;
  // 8 640 x 256: 22.3 -> 17.8, 28.3 -> 26.2); d <= 128 keeps 512 (114 816 x 128: 37.1 | 71.0 against 38.6 | 75.3 at 256).
  int total = -1;
  total = ltu_knob_pos("LTU_LA_SPLITS", d >= 256 ? 256 : 512);
  int want = total / (B > 0 ? B : 1);
  if (want < 1) want = 1;
  if (want > KVC_GROUP * KVC_GROUP) want = KVC_GROUP * KVC_GROUP;      // the merge is two levels of at most KVC_GROUP partials
  int tps = (N + want - 1) / want;
  tps = (tps + 31) / 32 * 32;
  if (tps < 32) tps = 32;
  *tokens_per_split = tps;
  return (N + tps - 1) / tps;
}
// tokens per workgroup of the per-token kernels: ~512 workgroups, whole tiles
static int pick_tokb(int B, int N, int d) {
  // d <= 128: two workgroups fit a CU (66 KB of LDS, <= 256 VGPRs) -> 512 workgroups; d = 256: one (132 KB) -> 256.  Swept 128 .. 1024.
  const int blocks = ltu_knob_pos("LTU_LA_TOKB_BLOCKS", d <= 128 ? 512 : 256);
  long long per = ((long long)B * N + blocks - 1) / blocks;
  int tokb = (int)((per + 31) / 32 * 32);
  if (tokb < 32) tokb = 32;
  if (tokb > 512) tokb = 512;
  return tokb;
}

extern "C" int ltu_linattn_splits(int B, int N) {
  int tps;
  return pick_splits(B, N, 128, &tps);      // the largest count any d uses: an upper bound for workspace sizing
}

#define LA_DISPATCH_D(d, ...)                                   \
  do {                                                          \
    if ((d) == 32) { constexpr int D = 32; __VA_ARGS__ }        \
    else if ((d) == 64) { constexpr int D = 64; __VA_ARGS__ }   \
    else if ((d) == 128) { constexpr int D = 128; __VA_ARGS__ } \
    else if ((d) == 256) { constexpr int D = 256; __VA_ARGS__ } \
    else return LTU_E_SHAPE;                                    \
  } while (0)

extern "C" int ltu_linattn_fwd(const void* qkv, void* out, float* ctx, float* colstats, float* qstat, float* part_ws,
                               long long ws_floats, int B, int N, int d, int dtype, ltu_stream_t s);
// floats of part_ws that ltu_linattn_fwd / _ctx / _bwd need for this shape (the same split count the launches pick)
extern "C" long long ltu_linattn_ws_floats(int B, int N, int d) {
  int tps;
  const long long ns = pick_splits(B, N, d, &tps), H = d / 32;
  return (long long)B * (ns + (ns + KVC_GROUP - 1) / KVC_GROUP + 2) * H * PART_STRIDE;
}
// phase A alone (online softmax over the tokens + context, merged): for callers that run phase B themselves
// (ltu_layer_tail_fwd with its qkv argument: the chain kernel applies the context to its own row block)
extern "C" int ltu_linattn_ctx(const void* qkv, float* ctx, float* colstats, float* part_ws, long long ws_floats, int B, int N, int d,
                               int dtype, ltu_stream_t s) {
  return ltu_linattn_fwd(qkv, nullptr, ctx, colstats, nullptr, part_ws, ws_floats, B, N, d, dtype, s);
}
extern "C" int ltu_linattn_fwd(const void* qkv, void* out, float* ctx, float* colstats, float* qstat, float* part_ws,
                               long long ws_floats, int B, int N, int d, int dtype, ltu_stream_t s) {
  const int H = d / 32;
  int tps;
  const int nsplit = pick_splits(B, N, d, &tps);
  if ((long long)B * (nsplit + (nsplit + KVC_GROUP - 1) / KVC_GROUP + 2) * H * PART_STRIDE > ws_floats) return LTU_E_ARG;
  hipStream_t st = (hipStream_t)s;
  const int tokb = pick_tokb(B, N, d);
  LTU_DISPATCH_T(dtype, {
    LA_DISPATCH_D(d, {
      hipLaunchKernelGGL((linattn_kv_partial<T, D>), dim3(nsplit, B), dim3(2 * D), 0, st, (const T*)qkv, part_ws, N, tps);
      // two-level merge of the split partials (level-1 results live behind the level-0 partials in part_ws)
      const int group = KVC_GROUP, ngroups = (nsplit + group - 1) / group;
      float* part2 = part_ws + (size_t)B * nsplit * H * PART_STRIDE;
      if (nsplit * 32 * sizeof(float) <= 48 * 1024 && !ltu_knob("LTU_LA_TWO_LEVEL", 0)) {
        const size_t fsb = (size_t)nsplit * 32 * sizeof(float);
        if (nsplit <= 32) hipLaunchKernelGGL(linattn_kv_combine1<4>, dim3(8, B * H), dim3(256), fsb, st, part_ws, nsplit, colstats, ctx, H);
        else if (nsplit <= 64) hipLaunchKernelGGL(linattn_kv_combine1<8>, dim3(8, B * H), dim3(256), fsb, st, part_ws, nsplit, colstats, ctx, H);
        else if (nsplit <= 128) hipLaunchKernelGGL(linattn_kv_combine1<16>, dim3(8, B * H), dim3(256), fsb, st, part_ws, nsplit, colstats, ctx, H);
        else hipLaunchKernelGGL(linattn_kv_combine1<32>, dim3(8, B * H), dim3(256), fsb, st, part_ws, nsplit, colstats, ctx, H);
      } else if (ngroups > 1) {
        hipLaunchKernelGGL(linattn_kv_combine, dim3(ngroups, B * H), dim3(1024), 0, st, part_ws, nsplit, group, part2, colstats, ctx, H, 0);
        hipLaunchKernelGGL(linattn_kv_combine, dim3(1, B * H), dim3(1024), 0, st, part2, ngroups, ngroups, nullptr, colstats, ctx, H, 1);
      } else {
        hipLaunchKernelGGL(linattn_kv_combine, dim3(1, B * H), dim3(1024), 0, st, part_ws, nsplit, nsplit, nullptr, colstats, ctx, H, 1);
      }
      if (out == nullptr) {
        // phase A only (ltu_linattn_ctx)
      } else if constexpr (IsBf16<T>::value && D >= 128) {
        // wave-private form: a wave per (4 heads, token chunk); ~LTU_LA_APPLY_WAVES waves in flight
        constexpr int CG = D / DK / LAW_HEADS;
        const long long tiles = cdiv(N, TOK);
        long long tpw = cdiv((long long)B * tiles * CG, (long long)ltu_knob_pos("LTU_LA_APPLY_WAVES", 4096));
        if (tpw < 1) tpw = 1;
        const long long waves = cdiv(tiles, tpw) * CG;              // per sample
        hipLaunchKernelGGL((linattn_apply_rows<D>), dim3((unsigned)cdiv(waves, 4), B), dim3(256), 0, st, (const uint16_t*)qkv, ctx, (uint16_t*)out, qstat, N, (int)tpw);
      } else {
        hipLaunchKernelGGL((linattn_apply<T, D>), dim3(cdiv(N, tokb), B), dim3(2 * D), 0, st, (const T*)qkv, ctx, (T*)out, qstat, N, tokb);
      }
    });
  });
  return ltu_check_launch();
}

extern "C" int ltu_linattn_bwd(const void* qkv, const void* dout, const float* ctx, const float* colstats,
                               const float* qstat, void* dqkv, float* dctx, float* tvec, float* part_ws, long long ws_floats, int B,
                               int N, int d, int dtype, ltu_stream_t s) {
  const int H = d / 32;
  int tps;
  const int nsplit = pick_splits(B, N, d, &tps);
  if ((long long)B * nsplit * H * 1024 > ws_floats) return LTU_E_ARG;
  hipStream_t st = (hipStream_t)s;
  const size_t lds_b = (size_t)(32 * (3 * d + 4) + 32 * (d + 4) + H * 96) * sizeof(float);
  const int tokb = pick_tokb(B, N, d);
  LTU_DISPATCH_T(dtype, {
    LA_DISPATCH_D(d, {
      const size_t lds_p = IsBf16<T>::value ? (size_t)4 * TOK * (D + LA_TR_PAD) * 2 + (size_t)2 * TOK * H * 2 * sizeof(float)
                                            : (size_t)(2 * TOK * D + TOK * H * 2) * sizeof(float);
      hipLaunchKernelGGL((linattn_dctx_partial<T, D>), dim3(nsplit, B), dim3(2 * D), lds_p, st, (const T*)qkv, (const T*)dout, qstat, part_ws, N, tps);
      if (nsplit <= 32) hipLaunchKernelGGL(linattn_dctx_combine<4>, dim3(8, B * H), dim3(256), 0, st, part_ws, dctx, nsplit, H);
      else if (nsplit <= 64) hipLaunchKernelGGL(linattn_dctx_combine<8>, dim3(8, B * H), dim3(256), 0, st, part_ws, dctx, nsplit, H);
      else if (nsplit <= 128) hipLaunchKernelGGL(linattn_dctx_combine<16>, dim3(8, B * H), dim3(256), 0, st, part_ws, dctx, nsplit, H);
      else hipLaunchKernelGGL(linattn_dctx_combine<32>, dim3(8, B * H), dim3(256), 0, st, part_ws, dctx, nsplit, H);
      hipLaunchKernelGGL((linattn_bwd_apply<T, D>), dim3(cdiv(N, tokb), B), dim3(2 * D), lds_b, st, (const T*)qkv, (const T*)dout, ctx, dctx, colstats, tvec, qstat, (T*)dqkv, N, tokb);
    });
  });
  return ltu_check_launch();
}
