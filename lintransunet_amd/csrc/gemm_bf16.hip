// bf16 matrix-core versions of the tap-table implicit GEMM (see gemm.hip for the formulation).
//
//   NT:  Y = gather(A) . W^T     v_mfma_f32_32x32x16_bf16, LDS tiles [rows][32 k] bf16 with 80-byte rows
//                                 (conflict-free ds_read_b128 fragments), register-staged double buffering,
//                                 output staged through LDS so that HBM stores are whole 16-byte vectors.
//   TN:  dW += G^T . gather(A)   both operands are "reduction-major" in memory; tiles are stored as loaded
//                                 ([m][n] / [m][k]) and the MFMA fragments are fetched with the transposing
//                                 LDS read ds_read_b64_tr_b16 (4 rows x 16 columns per 16-lane group).
// Activations and weight operands are bf16, accumulation fp32, bias / weight gradients fp32.
#include "gemm_desc.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ RowCoord split_row_b(const IGemmArgs& g, long long m) {
  RowCoord r;
  r.d = (int)(m % g.rd);
  long long t = m / g.rd;
  r.w = (int)(t % g.rw);
  t /= g.rw;
  r.h = (int)(t % g.rh);
  r.b = (int)(t / g.rh);
  return r;
}

// 8 consecutive bf16 channels of the A operand at (row, k); zero outside the source / beyond K
__device__ __forceinline__ uint4 gather_a8(const IGemmArgs& g, bool row_ok, const RowCoord& rc, int k) {
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  if (!row_ok || k >= g.K) return z;
  const int slot = k / g.C;
  const int c = k - slot * g.C;
  const Tap tp = g.tap[slot];
  int h = rc.h * g.mh + tp.dh, w = rc.w * g.mw + tp.dw, d = rc.d * g.md + tp.dd;
  if ((unsigned)h >= (unsigned)g.sh || (unsigned)w >= (unsigned)g.sw || (unsigned)d >= (unsigned)g.sd) return z;
  int ph = g.sh, pw = g.sw, pd = g.sd;
  if (g.ups) {
    h >>= 1; w >>= 1; d >>= 1;
    ph >>= 1; pw >>= 1; pd >>= 1;
  }
  const long long vox = (((long long)rc.b * ph + h) * pw + w) * pd + d;
  const uint16_t* p = c < g.c0 ? reinterpret_cast<const uint16_t*>(g.a0) + vox * g.lda0 + c
                               : reinterpret_cast<const uint16_t*>(g.a1) + vox * g.lda1 + (c - g.c0);
  return *reinterpret_cast<const uint4*>(p);
}

__device__ __forceinline__ uint4 gather_w8(const IGemmArgs& g, int n, int k) {
  if (n >= g.N || k >= g.K) return make_uint4(0u, 0u, 0u, 0u);
  const int slot = k / g.C;
  const int c = k - slot * g.C;
  const int nper = g.N / g.nseg;
  const int seg = n / nper;
  const uint16_t* base = reinterpret_cast<const uint16_t*>(g.w[seg]) + (long long)(n - seg * nper) * g.wrow;
  return *reinterpret_cast<const uint4*>(base + (int)g.tap[slot].wt * g.C + c);
}

__device__ __forceinline__ long long out_voxel_b(const IGemmArgs& g, long long m) {
  if (g.out_identity) return m;
  RowCoord rc = split_row_b(g, m);
  return (((long long)rc.b * g.oh + (rc.h * g.omh + g.ooh)) * g.ow + (rc.w * g.omw + g.oow)) * g.od +
         (rc.d * g.omd + g.ood);
}

// ------------------------------------------------------------------------------------------------ NT
template <int WM, int WN, int TM, int TN>
__global__ void __launch_bounds__(WM* WN * 64) igemm_nt_bf16_kernel(const IGemmArgs g) {
  constexpr int NT = WM * WN * 64, BM = WM * TM * 32, BN = WN * TN * 32, BK = 32, LDK = 40;
  constexpr int LA = (BM * 4 + NT - 1) / NT, LB = (BN * 4 + NT - 1) / NT;
  constexpr int LDC = BN + 8;                                   // staging row stride (bf16 elements)
  constexpr int TILE_ELEMS = 2 * (BM + BN) * LDK;
  constexpr int STAGE_ELEMS = BM * LDC;
  constexpr int SMEM_ELEMS = TILE_ELEMS > STAGE_ELEMS ? TILE_ELEMS : STAGE_ELEMS;
  __shared__ __attribute__((aligned(16))) uint16_t smem[SMEM_ELEMS];
  uint16_t* As = smem;                       // [2][BM][LDK]
  uint16_t* Bs = smem + 2 * BM * LDK;        // [2][BN][LDK]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const long long m_blk = (long long)blockIdx.x * BM;
  const int n_blk = blockIdx.y * BN;

  RowCoord rc[LA];
  bool rok[LA];
#pragma unroll
  for (int p = 0; p < LA; ++p) {
    const int idx = tid + p * NT;
    const long long m = m_blk + (idx >> 2);
    rok[p] = (idx < BM * 4) && (m < g.M);
    rc[p] = split_row_b(g, rok[p] ? m : 0);
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  uint4 ra[LA], rb[LB];
  const int nkt = (g.K + BK - 1) / BK;

  auto load_tile = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int p = 0; p < LA; ++p) {
      const int idx = tid + p * NT;
      ra[p] = gather_a8(g, rok[p], rc[p], k0 + (idx & 3) * 8);
    }
#pragma unroll
    for (int p = 0; p < LB; ++p) {
      const int idx = tid + p * NT;
      rb[p] = (idx < BN * 4) ? gather_w8(g, n_blk + (idx >> 2), k0 + (idx & 3) * 8) : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int p = 0; p < LA; ++p) {
      const int idx = tid + p * NT;
      if (idx < BM * 4) *reinterpret_cast<uint4*>(&As[(buf * BM + (idx >> 2)) * LDK + (idx & 3) * 8]) = ra[p];
    }
#pragma unroll
    for (int p = 0; p < LB; ++p) {
      const int idx = tid + p * NT;
      if (idx < BN * 4) *reinterpret_cast<uint4*>(&Bs[(buf * BN + (idx >> 2)) * LDK + (idx & 3) * 8]) = rb[p];
    }
  };

  load_tile(0);
  store_tile(0);
  __syncthreads();
  const int li = lane & 31, lh = lane >> 5;
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) load_tile(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        a[i] = *reinterpret_cast<const bf16x8*>(&As[(buf * BM + (wm * TM + i) * 32 + li) * LDK + ks * 16 + lh * 8]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        b[j] = *reinterpret_cast<const bf16x8*>(&Bs[(buf * BN + (wn * TN + j) * 32 + li) * LDK + ks * 16 + lh * 8]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nkt) store_tile(buf ^ 1);
    __syncthreads();
  }

  // epilogue: bias, convert, stage the BM x BN tile in LDS, then whole-vector stores
  uint16_t* Cs = smem;   // [BM][LDC]
  const int nper = g.N / g.nseg;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nl = (wn * TN + j) * 32 + li;
    const int n = n_blk + nl;
    float bv = 0.f;
    if (n < g.N) {
      const int seg = n / nper;
      if (g.bias[seg]) bv = g.bias[seg][n - seg * nper];
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ml = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        Cs[ml * LDC + nl] = f32_to_bf16(acc[i][j][r] + bv);
      }
  }
  __syncthreads();
  constexpr int CPR = BN / 4;                 // 8-byte chunks per row
  for (int idx = tid; idx < BM * CPR; idx += NT) {
    const int ml = idx / CPR, nl = (idx % CPR) * 4;
    const long long m = m_blk + ml;
    const int n = n_blk + nl;
    if (m >= g.M || n >= g.N) continue;
    uint2 v = *reinterpret_cast<const uint2*>(&Cs[ml * LDC + nl]);
    uint16_t* dst;
    if (n < g.n0) dst = reinterpret_cast<uint16_t*>(g.o0) + out_voxel_b(g, m) * g.ldo0 + n;
    else dst = reinterpret_cast<uint16_t*>(g.o1) + out_voxel_b(g, m) * g.ldo1 + (n - g.n0);
    if (g.accum) {
      const uint2 o = *reinterpret_cast<const uint2*>(dst);
      const float s0 = bf16_to_f32((uint16_t)(v.x & 0xffff)) + bf16_to_f32((uint16_t)(o.x & 0xffff));
      const float s1 = bf16_to_f32((uint16_t)(v.x >> 16)) + bf16_to_f32((uint16_t)(o.x >> 16));
      const float s2 = bf16_to_f32((uint16_t)(v.y & 0xffff)) + bf16_to_f32((uint16_t)(o.y & 0xffff));
      const float s3 = bf16_to_f32((uint16_t)(v.y >> 16)) + bf16_to_f32((uint16_t)(o.y >> 16));
      v.x = (uint32_t)f32_to_bf16(s0) | ((uint32_t)f32_to_bf16(s1) << 16);
      v.y = (uint32_t)f32_to_bf16(s2) | ((uint32_t)f32_to_bf16(s3) << 16);
    }
    *reinterpret_cast<uint2*>(dst) = v;
  }
}

int launch_nt_bf16(const IGemmArgs& g, hipStream_t st) {
  if (g.M <= 0 || g.N <= 0) return LTU_OK;
  if (g.C % 8 || g.c0 % 8 || g.lda0 % 8 || g.lda1 % 8 || g.wrow % 8 || g.N % 4 || g.n0 % 4 || g.ldo0 % 4 || g.ldo1 % 4)
    return LTU_E_SHAPE;
  if (g.N > 64) {
    dim3 grid(cdiv(g.M, 128), cdiv(g.N, 128));
    hipLaunchKernelGGL((igemm_nt_bf16_kernel<2, 2, 2, 2>), grid, dim3(256), 0, st, g);
  } else if (g.N > 32) {
    dim3 grid(cdiv(g.M, 128), 1);
    hipLaunchKernelGGL((igemm_nt_bf16_kernel<4, 1, 1, 2>), grid, dim3(256), 0, st, g);
  } else {
    dim3 grid(cdiv(g.M, 128), 1);
    hipLaunchKernelGGL((igemm_nt_bf16_kernel<4, 1, 1, 1>), grid, dim3(256), 0, st, g);
  }
  return ltu_check_launch();
}

// ------------------------------------------------------------------------------------------------ TN
// tile: BNn (n) x BKk (k) of dW, BR = 32 reduction rows per iteration.  Gs[BR][BNn+32], Xs[BR][BKk+32] bf16
// (the 64-byte pad makes the 4-row transposing reads of a 32-lane half hit 4 disjoint bank ranges).
template <int WM, int WN, int TM, int TN>
__global__ void __launch_bounds__(WM* WN * 64) wgrad_tn_bf16_kernel(const WGradArgs wa) {
  const IGemmArgs& g = wa.g;
  constexpr int NT = WM * WN * 64, BNn = WM * TM * 32, BKk = WN * TN * 32, BR = 32;
  constexpr int LDG = BNn + 32, LDX = BKk + 32;
  constexpr int LG = (BR * BNn / 8 + NT - 1) / NT, LX = (BR * BKk / 8 + NT - 1) / NT;
  __shared__ __attribute__((aligned(16))) uint16_t Gs[2][BR][LDG];
  __shared__ __attribute__((aligned(16))) uint16_t Xs[2][BR][LDX];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int k_blk = blockIdx.x * BKk, n_blk = blockIdx.y * BNn;
  const long long m_begin = (long long)blockIdx.z * wa.rows_per_split;
  long long m_end = m_begin + wa.rows_per_split;
  if (m_end > g.M) m_end = g.M;
  if (m_begin >= m_end) return;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  uint4 rg[LG], rx[LX];
  float bsum = 0.f;
  const int niter = (int)((m_end - m_begin + BR - 1) / BR);

  auto load_tile = [&](int it) {
    const long long m0 = m_begin + (long long)it * BR;
#pragma unroll
    for (int p = 0; p < LG; ++p) {
      const int idx = tid + p * NT;
      const int row = idx / (BNn / 8), nq = (idx % (BNn / 8)) * 8;
      const long long m = m0 + row;
      const int n = n_blk + nq;
      rg[p] = (idx < BR * BNn / 8 && m < m_end && n < g.N)
                  ? *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(wa.grad) + m * wa.ldg + n)
                  : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int p = 0; p < LX; ++p) {
      const int idx = tid + p * NT;
      const int row = idx / (BKk / 8), kq = (idx % (BKk / 8)) * 8;
      const long long m = m0 + row;
      const bool ok = idx < BR * BKk / 8 && m < m_end;
      const RowCoord rc = split_row_b(g, ok ? m : 0);
      rx[p] = gather_a8(g, ok, rc, k_blk + kq);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int p = 0; p < LG; ++p) {
      const int idx = tid + p * NT;
      if (idx < BR * BNn / 8) *reinterpret_cast<uint4*>(&Gs[buf][idx / (BNn / 8)][(idx % (BNn / 8)) * 8]) = rg[p];
    }
#pragma unroll
    for (int p = 0; p < LX; ++p) {
      const int idx = tid + p * NT;
      if (idx < BR * BKk / 8) *reinterpret_cast<uint4*>(&Xs[buf][idx / (BKk / 8)][(idx % (BKk / 8)) * 8]) = rx[p];
    }
  };

  load_tile(0);
  store_tile(0);
  __syncthreads();
  // transposing-read lane geometry: 16-lane group gq -> columns 16*(gq&1).., rows 8*(gq>>1)..; lane 4q+p -> row q, cols 4p..4p+3
  const int gq = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  const int tcol = 16 * (gq & 1) + 4 * tp;
  const int trow = 8 * (gq >> 1) + tq;
  const int lh = lane >> 5, li = lane & 31;
  const bool do_bias = wa.db != nullptr && blockIdx.x == 0;
  for (int it = 0; it < niter; ++it) {
    const int buf = it & 1;
    if (it + 1 < niter) load_tile(it + 1);
#pragma unroll
    for (int ks = 0; ks < BR / 16; ++ks) {
      bf16x8 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const uint16_t* p0 = &Gs[buf][ks * 16 + trow][(wm * TM + i) * 32 + tcol];
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 4 * LDG));
        union { struct { s16x4 l, h; } s; bf16x8 v; } u;
        u.s.l = lo; u.s.h = hi;
        a[i] = u.v;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const uint16_t* p0 = &Xs[buf][ks * 16 + trow][(wn * TN + j) * 32 + tcol];
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 4 * LDX));
        union { struct { s16x4 l, h; } s; bf16x8 v; } u;
        u.s.l = lo; u.s.h = hi;
        b[j] = u.v;
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (do_bias && tid < BNn) {
#pragma unroll 8
      for (int r = 0; r < BR; ++r) bsum += bf16_to_f32(Gs[buf][r][tid]);
    }
    if (it + 1 < niter) store_tile(buf ^ 1);
    __syncthreads();
  }

  if (do_bias && tid < BNn && n_blk + tid < g.N) atomicAdd(wa.db + n_blk + tid, bsum);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int k = k_blk + (wn * TN + j) * 32 + li;
    if (k >= g.K) continue;
    const int slot = k / g.C;
    const int c = k - slot * g.C;
    const long long wk = (long long)g.tap[slot].wt * g.C + c;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n_blk + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (n < g.N) atomicAdd(wa.dw + (long long)n * g.wrow + wk, acc[i][j][r]);
      }
  }
}

int launch_tn_bf16(WGradArgs& wa, hipStream_t st) {
  const IGemmArgs& g = wa.g;
  if (g.M <= 0 || g.N <= 0) return LTU_OK;
  if (g.C % 8 || g.c0 % 8 || g.lda0 % 8 || g.lda1 % 8 || g.N % 8 || wa.ldg % 8) return LTU_E_SHAPE;
  int bn, bk = 128;
  if (g.N > 64) bn = 128;
  else if (g.N > 32) bn = 64;
  else bn = 32;
  const unsigned nk = cdiv(g.K, bk), nn = cdiv(g.N, bn);
  long long want = 1024 / ((long long)nk * nn);
  if (want < 1) want = 1;
  long long rows = (g.M + want - 1) / want;
  if (rows < 256) rows = 256;
  rows = (rows + 31) / 32 * 32;
  wa.rows_per_split = (int)rows;
  dim3 grid(nk, nn, cdiv(g.M, rows));
  if (g.N > 64) hipLaunchKernelGGL((wgrad_tn_bf16_kernel<2, 2, 2, 2>), grid, dim3(256), 0, st, wa);
  else if (g.N > 32) hipLaunchKernelGGL((wgrad_tn_bf16_kernel<1, 4, 2, 1>), grid, dim3(256), 0, st, wa);
  else hipLaunchKernelGGL((wgrad_tn_bf16_kernel<1, 4, 1, 1>), grid, dim3(256), 0, st, wa);
  return ltu_check_launch();
}
