// Tap-table implicit GEMM for gfx950: one kernel family covers nn.Linear, 1x1x1 convs, 3x3x3 convs
// (any stride, virtual concat, fused nearest-x2 upsampling) and their data gradients; a second
// (reduction-over-rows) family covers every weight gradient.
//
//   NT:  Y[m][n]  = sum_k A[m][k] * W[n][k] (+ bias[n])       k = slot*C + c, A gathered through a tap table
//   TN:  dW[n][k] += sum_m G[m][n] * A[m][k]                   same gather, split over m, fp32 atomics
//
// Rows m are the voxels of a "row grid" [nb, rh, rw, rd]; the A element of (row, slot, c) is read at
// source voxel  row_coord * mul + off[slot]  (zero outside the source), optionally through a
// nearest-x2 upsampling (coord >> 1).  That one rule expresses forward convs (mul = stride,
// off = tap-1), stride-1 data gradients (off = 1-tap) and the parity classes of strided data
// gradients (row grid = voxels of one parity, off in {0,1}).
//
// Arithmetic: v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain) on fp32 LDS tiles; T only
// selects the HBM storage type.  Waves are 64 wide: a wave owns TM x TN tiles of 32x32.
#include <stdlib.h>

#include "gemm_desc.h"

// 4 consecutive channels of the A operand at (row, k); zero outside the source / beyond K
template <typename TA>
__device__ __forceinline__ float4 gather_a(const IGemmArgs& g, bool row_ok, const RowCoord& rc, int k) {
  float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
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
  if (c < g.c0) return Vec4<TA>::load(reinterpret_cast<const TA*>(g.a0) + vox * g.lda0 + c);
  return Vec4<TA>::load(reinterpret_cast<const TA*>(g.a1) + vox * g.lda1 + (c - g.c0));
}

__device__ __forceinline__ float4 gather_w(const IGemmArgs& g, int n, int k) {
  if (n >= g.N || k >= g.K) return make_float4(0.f, 0.f, 0.f, 0.f);
  const int slot = k / g.C;
  const int c = k - slot * g.C;
  const int nper = g.N / g.nseg;
  const int seg = n / nper;
  const float* base = reinterpret_cast<const float*>(g.w[seg]) + (long long)(n - seg * nper) * g.wrow;
  return *reinterpret_cast<const float4*>(base + (int)g.tap[slot].wt * g.C + c);
}

__device__ __forceinline__ long long out_voxel(const IGemmArgs& g, long long m) {
  if (g.out_identity) return m;
  RowCoord rc = split_row(g, m);
  return (((long long)rc.b * g.oh + (rc.h * g.omh + g.ooh)) * g.ow + (rc.w * g.omw + g.oow)) * g.od +
         (rc.d * g.omd + g.ood);
}

template <typename TA, typename TO, int WM, int WN, int TM, int TN>
__global__ void __launch_bounds__(WM* WN * 64) igemm_nt_kernel(const IGemmArgs g) {
  constexpr int NT = WM * WN * 64, BM = WM * TM * 32, BN = WN * TN * 32, BK = 16;
  constexpr int BMP = BM + 4, BNP = BN + 4;
  constexpr int LA = (BM * 4 + NT - 1) / NT, LB = (BN * 4 + NT - 1) / NT;
  __shared__ float As[2][BK][BMP];
  __shared__ float Bs[2][BK][BNP];

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
    rc[p] = split_row(g, rok[p] ? m : 0);
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[LA], rb[LB];
  const int nkt = (g.K + BK - 1) / BK;

  auto load_tile = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int p = 0; p < LA; ++p) {
      const int idx = tid + p * NT;
      ra[p] = gather_a<TA>(g, rok[p], rc[p], k0 + (idx & 3) * 4);
    }
#pragma unroll
    for (int p = 0; p < LB; ++p) {
      const int idx = tid + p * NT;
      rb[p] = (idx < BN * 4) ? gather_w(g, n_blk + (idx >> 2), k0 + (idx & 3) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int p = 0; p < LA; ++p) {
      const int idx = tid + p * NT;
      if (idx < BM * 4) {
        const int row = idx >> 2, kq = (idx & 3) * 4;
        As[buf][kq + 0][row] = ra[p].x;
        As[buf][kq + 1][row] = ra[p].y;
        As[buf][kq + 2][row] = ra[p].z;
        As[buf][kq + 3][row] = ra[p].w;
      }
    }
#pragma unroll
    for (int p = 0; p < LB; ++p) {
      const int idx = tid + p * NT;
      if (idx < BN * 4) {
        const int row = idx >> 2, kq = (idx & 3) * 4;
        Bs[buf][kq + 0][row] = rb[p].x;
        Bs[buf][kq + 1][row] = rb[p].y;
        Bs[buf][kq + 2][row] = rb[p].z;
        Bs[buf][kq + 3][row] = rb[p].w;
      }
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
    for (int ks = 0; ks < BK / 2; ++ks) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[buf][2 * ks + lh][(wm * TM + i) * 32 + li];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[buf][2 * ks + lh][(wn * TN + j) * 32 + li];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nkt) store_tile(buf ^ 1);
    __syncthreads();
  }

  // epilogue: acc reg r of lane l is D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
  const int nper = g.N / g.nseg;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n_blk + (wn * TN + j) * 32 + li;
    if (n >= g.N) continue;
    float bv = 0.f;
    {
      const int seg = n / nper;
      if (g.bias[seg]) bv = g.bias[seg][n - seg * nper];
    }
    TO* obase;
    int ld, col;
    if (n < g.n0) {
      obase = reinterpret_cast<TO*>(g.o0); ld = g.ldo0; col = n;
    } else {
      obase = reinterpret_cast<TO*>(g.o1); ld = g.ldo1; col = n - g.n0;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long m = m_blk + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < g.M) {
          TO* dst = obase + out_voxel(g, m) * ld + col;
          float val = acc[i][j][r] + bv;
          if (g.accum) val += ld1<TO>(dst);
          st1<TO>(dst, val);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// TN: dW[n][slot->wt][c] += sum_m G[m][n] * A[m][k];  db[n] += sum_m G[m][n]
template <typename TA, int WM, int WN, int TM, int TN>
__global__ void __launch_bounds__(WM* WN * 64) wgrad_tn_kernel(const WGradArgs wa) {
  const IGemmArgs& g = wa.g;
  constexpr int NT = WM * WN * 64, BNn = WM * TM * 32, BKk = WN * TN * 32, BR = 16;
  constexpr int BNP = BNn + 4, BKP = BKk + 4;
  constexpr int LG = (BR * BNn / 4 + NT - 1) / NT, LX = (BR * BKk / 4 + NT - 1) / NT;
  __shared__ float Gs[2][BR][BNP];
  __shared__ float Xs[2][BR][BKP];

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

  float4 rg[LG], rx[LX];
  float bsum = 0.f;
  const int niter = (int)((m_end - m_begin + BR - 1) / BR);

  auto load_tile = [&](int it) {
    const long long m0 = m_begin + (long long)it * BR;
#pragma unroll
    for (int p = 0; p < LG; ++p) {
      const int idx = tid + p * NT;
      const int row = idx / (BNn / 4), nq = (idx % (BNn / 4)) * 4;
      const long long m = m0 + row;
      const int n = n_blk + nq;
      rg[p] = (idx < BR * BNn / 4 && m < m_end && n < g.N)
                  ? Vec4<TA>::load(reinterpret_cast<const TA*>(wa.grad) + out_voxel(g, m) * wa.ldg + n)
                  : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int p = 0; p < LX; ++p) {
      const int idx = tid + p * NT;
      const int row = idx / (BKk / 4), kq = (idx % (BKk / 4)) * 4;
      const long long m = m0 + row;
      const bool ok = idx < BR * BKk / 4 && m < m_end;
      const RowCoord rc = split_row(g, ok ? m : 0);
      rx[p] = gather_a<TA>(g, ok, rc, k_blk + kq);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int p = 0; p < LG; ++p) {
      const int idx = tid + p * NT;
      if (idx < BR * BNn / 4) {
        const int row = idx / (BNn / 4), nq = (idx % (BNn / 4)) * 4;
        *reinterpret_cast<float4*>(&Gs[buf][row][nq]) = rg[p];
      }
    }
#pragma unroll
    for (int p = 0; p < LX; ++p) {
      const int idx = tid + p * NT;
      if (idx < BR * BKk / 4) {
        const int row = idx / (BKk / 4), kq = (idx % (BKk / 4)) * 4;
        *reinterpret_cast<float4*>(&Xs[buf][row][kq]) = rx[p];
      }
    }
  };

  load_tile(0);
  store_tile(0);
  __syncthreads();
  const int li = lane & 31, lh = lane >> 5;
  const bool do_bias = wa.db != nullptr && blockIdx.x == 0;
  for (int it = 0; it < niter; ++it) {
    const int buf = it & 1;
    if (it + 1 < niter) load_tile(it + 1);
#pragma unroll
    for (int ks = 0; ks < BR / 2; ++ks) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = Gs[buf][2 * ks + lh][(wm * TM + i) * 32 + li];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Xs[buf][2 * ks + lh][(wn * TN + j) * 32 + li];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (do_bias && tid < BNn) {
#pragma unroll
      for (int r = 0; r < BR; ++r) bsum += Gs[buf][r][tid];
    }
    if (it + 1 < niter) store_tile(buf ^ 1);
    __syncthreads();
  }

  if (do_bias && tid < BNn && n_blk + tid < (wa.t_co ? wa.t_co : g.N)) atomicAdd(wa.db + n_blk + tid, bsum);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int k = k_blk + (wn * TN + j) * 32 + li;
    if (k >= g.K) continue;
    const int slot = k / g.C;
    const int c = k - slot * g.C;
    const long long wk = (long long)g.tap[slot].wt * g.C + c;
    const int wt = g.tap[slot].wt;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n_blk + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (wa.t_co) {
          if (n < wa.t_co && c < wa.t_ci) atomicAdd(wa.dw + ((long long)n * wa.t_ci + c) * 27 + wt, acc[i][j][r]);
        } else if (n < g.N) {
          atomicAdd(wa.dw + (long long)n * g.wrow + wk, acc[i][j][r]);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
template <typename TA, typename TO>
static int launch_nt(const IGemmArgs& g, hipStream_t st) {
  if (g.M <= 0 || g.N <= 0) return LTU_OK;
  if (g.N > 64) {
    dim3 grid(cdiv(g.M, 128), cdiv(g.N, 128));
    hipLaunchKernelGGL((igemm_nt_kernel<TA, TO, 2, 2, 2, 2>), grid, dim3(256), 0, st, g);
  } else if (g.N > 32) {
    dim3 grid(cdiv(g.M, 128), 1);
    hipLaunchKernelGGL((igemm_nt_kernel<TA, TO, 4, 1, 1, 2>), grid, dim3(256), 0, st, g);
  } else {
    dim3 grid(cdiv(g.M, 128), 1);
    hipLaunchKernelGGL((igemm_nt_kernel<TA, TO, 4, 1, 1, 1>), grid, dim3(256), 0, st, g);
  }
  return ltu_check_launch();
}

template <typename TA>
static int launch_tn(WGradArgs& wa, hipStream_t st) {
  const IGemmArgs& g = wa.g;
  if (g.M <= 0 || g.N <= 0) return LTU_OK;
  int bn, bk;
  if (g.N > 64) { bn = 128; bk = 128; }
  else if (g.N > 32) { bn = 64; bk = 128; }
  else { bn = 32; bk = 128; }
  const unsigned nk = cdiv(g.K, bk), nn = cdiv(g.N, bn);
  // enough splits to fill the chip (~1024 blocks), each at least 256 rows
  long long want = 1024 / ((long long)nk * nn);
  if (want < 1) want = 1;
  long long rows = (g.M + want - 1) / want;
  if (rows < 256) rows = 256;
  rows = (rows + 15) / 16 * 16;
  wa.rows_per_split = (int)rows;
  dim3 grid(nk, nn, cdiv(g.M, rows));
  if (g.N > 64) hipLaunchKernelGGL((wgrad_tn_kernel<TA, 2, 2, 2, 2>), grid, dim3(256), 0, st, wa);
  else if (g.N > 32) hipLaunchKernelGGL((wgrad_tn_kernel<TA, 1, 4, 2, 1>), grid, dim3(256), 0, st, wa);
  else hipLaunchKernelGGL((wgrad_tn_kernel<TA, 1, 4, 1, 1>), grid, dim3(256), 0, st, wa);
  return ltu_check_launch();
}

static void dense_desc(IGemmArgs& g, long long M, int N, int K) {
  memset(&g, 0, sizeof(g));
  g.M = M; g.N = N; g.K = K; g.C = K; g.c0 = K;
  g.nseg = 1; g.wrow = K;
  g.nb = 1; g.rh = 1; g.rw = 1; g.rd = (int)M;
  g.sh = 1; g.sw = 1; g.sd = (int)M; g.ups = 0;
  g.mh = g.mw = g.md = 1;
  g.ntaps = 1;
  g.tap[0] = Tap{0, 0, 0, 0};
  g.out_identity = 1;
  g.n0 = N;
}

extern "C" int ltu_linear_fwd(const void* a, int lda, const void* const* w, int nw, const float* const* bias, void* y,
                              int ldy, int M, int N, int K, int accumulate, int dtype, ltu_stream_t s) {
  if (nw < 1 || nw > 3 || N % nw != 0 || K % 4 != 0 || lda % 4 != 0) return LTU_E_SHAPE;
  IGemmArgs g;
  dense_desc(g, M, N, K);
  g.a0 = a; g.a1 = a; g.lda0 = lda; g.lda1 = lda;
  g.nseg = nw;
  for (int i = 0; i < nw; ++i) { g.w[i] = w[i]; g.bias[i] = bias ? bias[i] : nullptr; }
  g.o0 = y; g.o1 = y; g.ldo0 = ldy; g.ldo1 = ldy;
  g.accum = accumulate;
  if (dtype == LTU_BF16 && nw == 1 && !ltu_knob("LTU_NO_PW_SMALL", 0)) {      // few-channel streaming projections (the attention gates' 1x1x1 convs)
    const int pr = launch_pw_small_bf16(a, lda, w[0], bias ? bias[0] : nullptr, y, ldy, M, N, K, accumulate, (hipStream_t)s);
    if (pr != 1) return pr;
  }
  if (dtype == LTU_BF16) return launch_nt_bf16(g, (hipStream_t)s);
  if (dtype != LTU_F32) return LTU_E_DTYPE;
  return launch_nt<float, float>(g, (hipStream_t)s);
}

// u = a . w^T + bias and h = dropout(gelu(u)) (transformer FFN, model/trans_block.py:203-208).  bf16 shapes the weight-stationary
// projection kernel handles get both from ONE launch (GELU in its epilogue); everything else runs the projection and
// ltu_gelu_dropout_fwd back to back, with identical results (the fused epilogue also applies GELU to the bf16-rounded u).
extern "C" int ltu_gelu_dropout_fwd(const void* u, void* h, long long n, float p, uint64_t seed, const uint64_t* step, int dtype, ltu_stream_t s);
extern "C" int ltu_linear_gelu_fwd(const void* a, int lda, const void* w, const float* bias, void* u, void* h, int M, int N, int K,
                                   float p, uint64_t seed, const uint64_t* step, int dtype, ltu_stream_t s) {
  if (K % 4 != 0 || lda % 4 != 0) return LTU_E_SHAPE;
  if (dtype == LTU_BF16 && !ltu_knob("LTU_NO_GELU_FUSE", 0)) {
    IGemmArgs g;
    dense_desc(g, M, N, K);
    g.a0 = a; g.a1 = a; g.lda0 = lda; g.lda1 = lda;
    g.nseg = 1; g.w[0] = w; g.bias[0] = bias;
    g.o0 = u; g.o1 = u; g.ldo0 = N; g.ldo1 = N;
    g.gelu_out = h; g.drop_p = p; g.drop_seed = seed; g.drop_step = reinterpret_cast<const unsigned long long*>(step);
    if (g.C % 8 == 0 && g.lda0 % 8 == 0 && g.N % 4 == 0) {
      const int rr = launch_nt_ring_bf16(g, (hipStream_t)s);
      if (rr != 1) return rr;
    }
  }
  const void* wl[1] = {w};
  const float* bl[1] = {bias};
  const int rc = ltu_linear_fwd(a, lda, wl, 1, bl, u, N, M, N, K, 0, dtype, s);
  if (rc) return rc;
  return ltu_gelu_dropout_fwd(u, h, (long long)M * N, p, seed, step, dtype, s);
}

extern "C" long long ltu_wgrad_ws_floats(long long M, int N, int K) {
  const long long tn = tn_geometry(M, N, K, 32).ws_floats, halo = conv_wgrad_halo_ws_floats(N, K);
  const long long ring = tn_ring_ws_floats(M, N, K);
  return tn > halo ? (tn > ring ? tn : ring) : (halo > ring ? halo : ring);
}

extern "C" int ltu_linear_wgrad(const void* grad, int ldg, const void* a, int lda, float* const* dw, float* const* db, int nw,
                                int M, int N, int K, float* ws, long long ws_floats, ltu_reduce_job* defer, int dtype, ltu_stream_t s) {
  if (defer != nullptr) defer->part = nullptr;
  if (nw < 1 || nw > 3 || N % nw != 0 || K % 4 != 0 || lda % 4 != 0 || ldg % 4 != 0 || N % 4 != 0) return LTU_E_SHAPE;
  WGradArgs wa;
  memset(&wa, 0, sizeof(wa));
  if (dtype == LTU_BF16 && ws != nullptr) {
    // one launch over all weight blocks, two-stage reduction through the workspace
    dense_desc(wa.g, M, N, K);
    wa.g.a0 = a; wa.g.a1 = a; wa.g.lda0 = lda; wa.g.lda1 = lda;
    wa.grad = grad; wa.ldg = ldg; wa.part = ws; wa.part_floats = ws_floats;
    wa.nseg_w = nw;
    for (int i = 0; i < nw; ++i) { wa.dwseg[i] = dw[i]; wa.dbseg[i] = db ? db[i] : nullptr; }
    wa.dw = dw[0]; wa.db = db ? db[0] : nullptr;
    if (defer != nullptr) {                  // ring kernel only; its second stage is left to ltu_reduce_batch
      int nsplit = 0;
      const int rr = launch_tn_ring_bf16(wa, (hipStream_t)s, &nsplit);
      if (rr == LTU_OK) {
        defer->part = ws; defer->nsplit = nsplit; defer->n = N; defer->k = K; defer->nseg = nw; defer->mode = 0;
        for (int i = 0; i < 3; ++i) { defer->out[i] = i < nw ? dw[i] : nullptr; defer->outb[i] = (i < nw && db) ? db[i] : nullptr; }
        return LTU_OK;
      }
      if (rr != 1) return rr;
    }
    return launch_tn_bf16(wa, (hipStream_t)s);
  }
  const int Ns = N / nw;
  const size_t esz = dtype == LTU_BF16 ? 2 : 4;
  for (int i = 0; i < nw; ++i) {            // atomic path: one launch per weight block
    memset(&wa, 0, sizeof(wa));
    dense_desc(wa.g, M, Ns, K);
    wa.g.a0 = a; wa.g.a1 = a; wa.g.lda0 = lda; wa.g.lda1 = lda;
    wa.grad = reinterpret_cast<const char*>(grad) + (size_t)i * Ns * esz; wa.ldg = ldg;
    wa.dw = dw[i]; wa.db = db ? db[i] : nullptr;
    int rc;
    if (dtype == LTU_BF16) rc = launch_tn_bf16(wa, (hipStream_t)s);
    else if (dtype == LTU_F32) rc = launch_tn<float>(wa, (hipStream_t)s);
    else return LTU_E_DTYPE;
    if (rc) return rc;
  }
  return LTU_OK;
}

extern "C" long long ltu_linear_wgrad_group_ws_floats(const ltu_wgrad_job* jobs, int njobs, int blocks) {
  return tn_ring_group_ws_floats(jobs, njobs, blocks);
}
extern "C" int ltu_linear_wgrad_group(const ltu_wgrad_job* jobs, int njobs, int blocks, float* ws, long long ws_floats, int dtype,
                                      ltu_stream_t s) {
  if (dtype != LTU_BF16) return LTU_E_DTYPE;
  const int rc = launch_tn_ring_group_bf16(jobs, njobs, blocks, ws, ws_floats, (hipStream_t)s);
  return rc == 1 ? LTU_E_SHAPE : rc;
}

static bool use_halo() {
  int v = -1;
  v = ltu_knob("LTU_NO_HALO", 0) ? 0 : 1;
  return v == 1;
}

// forward-conv gather description (also used by the weight gradient)
static int conv_fwd_desc(IGemmArgs& g, int B, int Hi, int Wi, int Di, int C0, int C1, int Co, int sh, int sw, int sd,
                         int ups, int* Ho, int* Wo, int* Do) {
  if ((sh != 1 && sh != 2) || (sw != 1 && sw != 2) || (sd != 1 && sd != 2)) return LTU_E_ARG;
  if (C0 % 4 != 0 || C1 % 4 != 0 || C0 <= 0) return LTU_E_SHAPE;
  memset(&g, 0, sizeof(g));
  const int Hl = ups ? 2 * Hi : Hi, Wl = ups ? 2 * Wi : Wi, Dl = ups ? 2 * Di : Di;
  *Ho = (Hl - 1) / sh + 1; *Wo = (Wl - 1) / sw + 1; *Do = (Dl - 1) / sd + 1;   // k=3, pad=1
  g.nb = B; g.rh = *Ho; g.rw = *Wo; g.rd = *Do;
  g.M = (long long)B * g.rh * g.rw * g.rd;
  g.N = Co; g.C = C0 + C1; g.c0 = C0; g.K = 27 * g.C;
  g.lda0 = C0; g.lda1 = C1;
  g.nseg = 1; g.wrow = 27 * g.C;
  g.sh = Hl; g.sw = Wl; g.sd = Dl; g.ups = ups ? 1 : 0;
  g.mh = sh; g.mw = sw; g.md = sd;
  g.ntaps = 27;
  for (int t = 0; t < 27; ++t) g.tap[t] = Tap{(int8_t)(t / 9 - 1), (int8_t)((t / 3) % 3 - 1), (int8_t)(t % 3 - 1), (int8_t)t};
  g.out_identity = 1;
  g.n0 = Co;
  return LTU_OK;
}

extern "C" int ltu_conv3d_fwd(const void* x0, const void* x1, const void* wf, const float* bias, void* y, int B, int Hi,
                              int Wi, int Di, int C0, int C1, int Co, int sh, int sw, int sd, int ups, float* ws, long long ws_floats,
                              int dtype, ltu_stream_t s) {
  IGemmArgs g;
  int Ho, Wo, Do;
  int rc = conv_fwd_desc(g, B, Hi, Wi, Di, C0, C1, Co, sh, sw, sd, ups, &Ho, &Wo, &Do);
  if (rc) return rc;
  g.a0 = x0; g.a1 = x1 ? x1 : x0;
  g.w[0] = wf; g.bias[0] = bias;
  g.o0 = y; g.o1 = y; g.ldo0 = Co; g.ldo1 = Co;
  if (dtype == LTU_BF16) {
    if (sh == 1 && sw == 1 && sd == 1 && !ups && use_halo()) {      // LDS-staged halo brick instead of 27 gathers
      HaloArgs a;
      memset(&a, 0, sizeof(a));
      a.x0 = x0; a.x1 = x1 ? x1 : x0; a.w = wf; a.bias = bias; a.o0 = y; a.o1 = y;
      a.B = B; a.H = Hi; a.W = Wi; a.D = Di;
      a.C = C0 + C1; a.c0 = C0; a.lda0 = C0; a.lda1 = C1 > 0 ? C1 : C0;
      a.N = Co; a.n0 = Co; a.ldo0 = Co; a.ldo1 = Co;
      a.part = ws; a.part_floats = ws_floats;
      const int hr = launch_conv_halo_bf16(a, (hipStream_t)s);
      if (hr != 1) return hr;
    }
    g.part = ws; g.part_floats = ws_floats;     // strided / upsampling convs on small grids split their K loop
    return launch_nt_bf16(g, (hipStream_t)s);
  }
  if (dtype != LTU_F32) return LTU_E_DTYPE;
  return launch_nt<float, float>(g, (hipStream_t)s);
}

extern "C" long long ltu_conv3d_ws_floats(int B, int H, int W, int D, int C, int Co) {
  const long long halo = conv_halo_ws_floats(B, H, W, D, C, Co), ig = igemm_nt_ws_floats((long long)B * H * W * D, Co, 27 * C);
  return halo > ig ? halo : ig;
}
extern "C" long long ltu_igemm_ws_floats(long long M, int N, int K) { return igemm_nt_ws_floats(M, N, K); }

extern "C" int ltu_conv3d_wgrad(const void* grad, const void* x0, const void* x1, float* dwf, float* db, int B, int Hi,
                                int Wi, int Di, int C0, int C1, int Co, int sh, int sw, int sd, int ups, int torch_co,
                                int torch_ci, float* ws, long long ws_floats, int dtype, ltu_stream_t s) {
  WGradArgs wa;
  memset(&wa, 0, sizeof(wa));
  int Ho, Wo, Do;
  int rc = conv_fwd_desc(wa.g, B, Hi, Wi, Di, C0, C1, Co, sh, sw, sd, ups, &Ho, &Wo, &Do);
  if (rc) return rc;
  if (Co % 4 != 0) return LTU_E_SHAPE;
  wa.g.a0 = x0; wa.g.a1 = x1 ? x1 : x0;
  wa.grad = grad; wa.ldg = Co; wa.dw = dwf; wa.db = db; wa.t_co = torch_co; wa.t_ci = torch_ci;
  wa.part = dtype == LTU_BF16 ? ws : nullptr; wa.part_floats = ws_floats; wa.nseg_w = 1;
  if (dtype == LTU_BF16 && ws != nullptr && sh == 1 && sw == 1 && sd == 1 && !ups && use_halo()) {
    WHaloArgs h;
    memset(&h, 0, sizeof(h));
    h.x0 = x0; h.x1 = x1 ? x1 : x0; h.grad = grad;
    h.B = B; h.H = Hi; h.W = Wi; h.D = Di;
    h.C = C0 + C1; h.c0 = C0; h.lda0 = C0; h.lda1 = C1 > 0 ? C1 : C0;
    h.N = Co; h.ldg = Co; h.part = ws; h.part_floats = ws_floats;
    int nsplit = 0;
    const int hr = launch_conv_wgrad_halo_bf16(h, &nsplit, (hipStream_t)s);
    if (hr == LTU_OK) {
      wa.npad = Co; wa.kpad = 27 * (C0 + C1);
      wa.bpart = ws + (long long)nsplit * wa.npad * wa.kpad;
      return launch_wgrad_reduce(wa, nsplit, (hipStream_t)s);
    }
    if (hr != 1) return hr;
  }
  if (dtype == LTU_BF16) return launch_tn_bf16(wa, (hipStream_t)s);
  if (dtype != LTU_F32) return LTU_E_DTYPE;
  return launch_tn<float>(wa, (hipStream_t)s);
}

// Data gradient.  Logical input dims (Hl,Wl,Dl); the forward output dims follow from the stride.
// For a dim of stride 2 the input positions split into parity classes:  even i receives only from
// tap 1 (at o = i/2), odd i from tap 0 (o = (i+1)/2) and tap 2 (o = (i-1)/2).
extern "C" int ltu_conv3d_dgrad(const void* grad, const void* wd, void* dx0, void* dx1, int B, int Hl, int Wl, int Dl,
                                int C0, int C1, int Co, int sh, int sw, int sd, float* ws, long long ws_floats, int dtype, ltu_stream_t s) {
  if ((sh != 1 && sh != 2) || (sw != 1 && sw != 2) || (sd != 1 && sd != 2)) return LTU_E_ARG;
  if (Co % 4 != 0) return LTU_E_SHAPE;
  if (dtype == LTU_BF16 && sh == 1 && sw == 1 && sd == 1 && use_halo()) {
    HaloArgs a;
    memset(&a, 0, sizeof(a));
    a.x0 = grad; a.x1 = grad; a.w = wd; a.bias = nullptr; a.o0 = dx0; a.o1 = dx1 ? dx1 : dx0;
    a.B = B; a.H = Hl; a.W = Wl; a.D = Dl;
    a.C = Co; a.c0 = Co; a.lda0 = Co; a.lda1 = Co;
    a.N = C0 + C1; a.n0 = C0; a.ldo0 = C0; a.ldo1 = C1 > 0 ? C1 : C0;
    a.flip = 1;
    a.part = ws; a.part_floats = ws_floats;
    const int hr = launch_conv_halo_bf16(a, (hipStream_t)s);
    if (hr != 1) return hr;
  }
  const int Ho = (Hl - 1) / sh + 1, Wo = (Wl - 1) / sw + 1, Do = (Dl - 1) / sd + 1;
  if (dtype == LTU_BF16 && sh == 2 && sw == 2 && C1 == 0 && use_halo() && !ltu_knob("LTU_NO_SDGRAD_RING", 0)) {
    // every parity class of the input grid from one LDS halo brick of the output gradient, compile-time entry table
    const int hr = launch_sdgrad_ring_bf16(grad, wd, dx0, B, Hl, Wl, Dl, C0, Co, sd, (hipStream_t)s);
    if (hr != 1) return hr;
  }
  if (dtype == LTU_BF16 && (sh == 2 || sw == 2 || sd == 2) && use_halo() && !ltu_knob("LTU_NO_CLASS_HALO", 0)) {
    // the generic class kernel (run-time entry table)
    ClassHaloArgs a;
    memset(&a, 0, sizeof(a));
    a.x = grad; a.w = wd; a.bias = nullptr; a.o0 = dx0; a.o1 = dx1 ? dx1 : dx0;
    a.B = B; a.H = Ho; a.W = Wo; a.D = Do; a.C = Co; a.lda = Co;
    a.N = C0 + C1; a.n0 = C0; a.ldo0 = C0; a.ldo1 = C1 > 0 ? C1 : C0;
    a.Hh = Hl; a.Wh = Wl; a.Dh = Dl; a.mh = sh; a.mw = sw; a.md = sd;
    a.wrow = 27 * Co;
    // per axis and parity: list of (coarse offset, tap).  stride 2: i = 2c -> tap 1 at o = c; i = 2c+1 -> tap 0 at c+1, tap 2 at c.
    // stride 1: o = i + 1 - t.
    const int st3[3] = {sh, sw, sd};
    int cnt[3][2], off[3][2][3], tap[3][2][3];
    for (int ax = 0; ax < 3; ++ax) {
      if (st3[ax] == 2) {
        cnt[ax][0] = 1; off[ax][0][0] = 0; tap[ax][0][0] = 1;
        cnt[ax][1] = 2; off[ax][1][0] = 1; tap[ax][1][0] = 0; off[ax][1][1] = 0; tap[ax][1][1] = 2;
      } else {
        cnt[ax][0] = 3; cnt[ax][1] = 0;
        for (int t = 0; t < 3; ++t) { off[ax][0][t] = 1 - t; tap[ax][0][t] = t; }
      }
    }
    int ne = 0, nc = 0;
    for (int ph = 0; ph < sh; ++ph)
      for (int pw = 0; pw < sw; ++pw)
        for (int pd = 0; pd < sd; ++pd) {
          a.cls_p[nc][0] = (int8_t)ph; a.cls_p[nc][1] = (int8_t)pw; a.cls_p[nc][2] = (int8_t)pd;
          for (int i = 0; i < cnt[0][ph]; ++i)
            for (int j = 0; j < cnt[1][pw]; ++j)
              for (int k = 0; k < cnt[2][pd]; ++k) {
                const int wt = (tap[0][ph][i] * 3 + tap[1][pw][j]) * 3 + tap[2][pd][k];
                a.ent[ne++] = ClsEntry{(int8_t)off[0][ph][i], (int8_t)off[1][pw][j], (int8_t)off[2][pd][k], (int8_t)nc, wt * Co};
              }
          ++nc;
        }
    a.nent = ne; a.ncls = nc;
    const int hr = launch_conv_class_halo_bf16(a, (hipStream_t)s);
    if (hr != 1) return hr;
  }
  const int str[3] = {sh, sw, sd};
  const int len[3] = {Hl, Wl, Dl};
  const int nclass[3] = {sh, sw, sd};
  for (int ph = 0; ph < nclass[0]; ++ph)
    for (int pw = 0; pw < nclass[1]; ++pw)
      for (int pd = 0; pd < nclass[2]; ++pd) {
        const int par[3] = {ph, pw, pd};
        IGemmArgs g;
        memset(&g, 0, sizeof(g));
        int cnt[3], toff[3][3], ttap[3][3], rows[3];
        for (int a = 0; a < 3; ++a) {
          if (str[a] == 1) {
            cnt[a] = 3;
            for (int t = 0; t < 3; ++t) { ttap[a][t] = t; toff[a][t] = 1 - t; }   // o = i + 1 - t
            rows[a] = len[a];
          } else if (par[a] == 0) {
            cnt[a] = 1; ttap[a][0] = 1; toff[a][0] = 0;                          // i = 2c: o = c
            rows[a] = (len[a] + 1) / 2;
          } else {
            cnt[a] = 2; ttap[a][0] = 0; toff[a][0] = 1; ttap[a][1] = 2; toff[a][1] = 0;  // i = 2c+1: o = c+1 | c
            rows[a] = len[a] / 2;
          }
        }
        if (rows[0] == 0 || rows[1] == 0 || rows[2] == 0) continue;
        int nt = 0;
        for (int a = 0; a < cnt[0]; ++a)
          for (int b = 0; b < cnt[1]; ++b)
            for (int c = 0; c < cnt[2]; ++c)
              g.tap[nt++] = Tap{(int8_t)toff[0][a], (int8_t)toff[1][b], (int8_t)toff[2][c],
                                (int8_t)((ttap[0][a] * 3 + ttap[1][b]) * 3 + ttap[2][c])};
        g.ntaps = nt;
        g.nb = B; g.rh = rows[0]; g.rw = rows[1]; g.rd = rows[2];
        g.M = (long long)B * rows[0] * rows[1] * rows[2];
        g.N = C0 + C1; g.C = Co; g.c0 = Co; g.K = nt * Co;
        g.lda0 = Co; g.lda1 = Co;
        g.nseg = 1; g.wrow = 27 * Co;
        g.sh = Ho; g.sw = Wo; g.sd = Do; g.ups = 0;
        g.mh = g.mw = g.md = 1;
        g.a0 = grad; g.a1 = grad;
        g.w[0] = wd;
        g.out_identity = (sh == 1 && sw == 1 && sd == 1);
        g.omh = str[0]; g.omw = str[1]; g.omd = str[2];
        g.ooh = ph; g.oow = pw; g.ood = pd;
        g.oh = Hl; g.ow = Wl; g.od = Dl;
        g.n0 = C0; g.o0 = dx0; g.ldo0 = C0; g.o1 = dx1 ? dx1 : dx0; g.ldo1 = C1 > 0 ? C1 : C0;
        int rc;
        if (dtype == LTU_F32) rc = launch_nt<float, float>(g, (hipStream_t)s);
        else if (dtype == LTU_BF16) rc = launch_nt_bf16(g, (hipStream_t)s);
        else return LTU_E_DTYPE;
        if (rc) return rc;
      }
  return LTU_OK;
}

// ---- two stride-1 convs over the same input as one (a decoder level's conv1 and its mask head: Unet_3Dblock.py:1353 +
// 1380).  Forward: output columns [0,N0) -> y0 [.., N0], the rest -> y1 [.., N1]; wf [N0+N1][27][C], bias [N0+N1].
// Data gradient: the virtual concat of g0 [.., N0] and g1 [.., N1] against wd [C][27][N0+N1] -> dx [.., C] (no add pass).
extern "C" int ltu_conv3d_pair_fwd(const void* x, const void* wf, const float* bias, void* y0, void* y1, int B, int H, int W,
                                   int D, int C, int N0, int N1, float* ws, long long ws_floats, int dtype, ltu_stream_t s) {
  if (N0 % 4 || N1 % 4 || N0 <= 0 || N1 <= 0) return LTU_E_SHAPE;
  IGemmArgs g;
  int Ho, Wo, Do;
  int rc = conv_fwd_desc(g, B, H, W, D, C, 0, N0 + N1, 1, 1, 1, 0, &Ho, &Wo, &Do);
  if (rc) return rc;
  g.a0 = x; g.a1 = x;
  g.w[0] = wf; g.bias[0] = bias;
  g.o0 = y0; g.o1 = y1; g.n0 = N0; g.ldo0 = N0; g.ldo1 = N1;
  if (dtype == LTU_BF16) {
    if (use_halo()) {
      HaloArgs a;
      memset(&a, 0, sizeof(a));
      a.x0 = x; a.x1 = x; a.w = wf; a.bias = bias; a.o0 = y0; a.o1 = y1;
      a.B = B; a.H = H; a.W = W; a.D = D;
      a.C = C; a.c0 = C; a.lda0 = C; a.lda1 = C;
      a.N = N0 + N1; a.n0 = N0; a.ldo0 = N0; a.ldo1 = N1;
      a.part = ws; a.part_floats = ws_floats;
      const int hr = launch_conv_halo_bf16(a, (hipStream_t)s);
      if (hr != 1) return hr;
    }
    return launch_nt_bf16(g, (hipStream_t)s);
  }
  if (dtype != LTU_F32) return LTU_E_DTYPE;
  return launch_nt<float, float>(g, (hipStream_t)s);
}

extern "C" int ltu_conv3d_pair_dgrad(const void* g0, const void* g1, const void* wd, void* dx, int B, int H, int W, int D, int C,
                                     int N0, int N1, float* ws, long long ws_floats, int dtype, ltu_stream_t s) {
  if (N0 % 4 || N1 % 4 || N0 <= 0 || N1 <= 0) return LTU_E_SHAPE;
  if (dtype == LTU_BF16 && use_halo()) {
    HaloArgs a;
    memset(&a, 0, sizeof(a));
    a.x0 = g0; a.x1 = g1; a.w = wd; a.bias = nullptr; a.o0 = dx; a.o1 = dx;
    a.B = B; a.H = H; a.W = W; a.D = D;
    a.C = N0 + N1; a.c0 = N0; a.lda0 = N0; a.lda1 = N1;
    a.N = C; a.n0 = C; a.ldo0 = C; a.ldo1 = C;
    a.flip = 1;
    a.part = ws; a.part_floats = ws_floats;
    const int hr = launch_conv_halo_bf16(a, (hipStream_t)s);
    if (hr != 1) return hr;
  }
  IGemmArgs g;
  memset(&g, 0, sizeof(g));
  int nt = 0;
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b)
      for (int c = 0; c < 3; ++c) g.tap[nt++] = Tap{(int8_t)(1 - a), (int8_t)(1 - b), (int8_t)(1 - c), (int8_t)((a * 3 + b) * 3 + c)};
  g.ntaps = 27;
  g.nb = B; g.rh = H; g.rw = W; g.rd = D;
  g.M = (long long)B * H * W * D;
  g.N = C; g.C = N0 + N1; g.c0 = N0; g.K = 27 * (N0 + N1);
  g.lda0 = N0; g.lda1 = N1;
  g.nseg = 1; g.wrow = 27 * (N0 + N1);
  g.sh = H; g.sw = W; g.sd = D; g.ups = 0;
  g.mh = g.mw = g.md = 1;
  g.a0 = g0; g.a1 = g1;
  g.w[0] = wd;
  g.out_identity = 1;
  g.omh = g.omw = g.omd = 1;
  g.oh = H; g.ow = W; g.od = D;
  g.n0 = C; g.o0 = dx; g.o1 = dx; g.ldo0 = C; g.ldo1 = C;
  if (dtype == LTU_F32) return launch_nt<float, float>(g, (hipStream_t)s);
  if (dtype == LTU_BF16) return launch_nt_bf16(g, (hipStream_t)s);
  return LTU_E_DTYPE;
}

// weight gradients of a conv pair from ONE pass over x (bf16 halo path): the gradient tile is the virtual concat of g0 / g1;
// rows [0, co_a) land in dwa [co_a][ci][27], rows [N0, N0 + co_b) in dwb.  Other dtypes / shapes: two ordinary calls.
extern "C" int ltu_conv3d_pair_wgrad(const void* g0, const void* g1, const void* x, float* dwa, float* dba, float* dwb,
                                     float* dbb, int B, int H, int W, int D, int C, int N0, int N1, int co_a, int co_b, int ci,
                                     float* ws, long long ws_floats, int dtype, ltu_stream_t s) {
  if (dtype == LTU_BF16 && ws != nullptr && use_halo()) {
    WGradArgs wa;
    memset(&wa, 0, sizeof(wa));
    int Ho, Wo, Do;
    int rc = conv_fwd_desc(wa.g, B, H, W, D, C, 0, N0 + N1, 1, 1, 1, 0, &Ho, &Wo, &Do);
    if (rc) return rc;
    wa.g.a0 = x; wa.g.a1 = x;
    wa.dw = dwa; wa.db = dba; wa.t_co = co_a; wa.t_ci = ci;
    wa.dw2 = dwb; wa.db2 = dbb; wa.n0_2 = N0; wa.t_co2 = co_b;
    wa.part = ws; wa.part_floats = ws_floats; wa.nseg_w = 1;
    WHaloArgs h;
    memset(&h, 0, sizeof(h));
    h.x0 = x; h.x1 = x; h.grad = g0; h.grad1 = g1; h.gn0 = N0; h.ldg1 = N1;
    h.B = B; h.H = H; h.W = W; h.D = D;
    h.C = C; h.c0 = C; h.lda0 = C; h.lda1 = C;
    h.N = N0 + N1; h.ldg = N0; h.part = ws; h.part_floats = ws_floats;
    int nsplit = 0;
    const int hr = launch_conv_wgrad_halo_bf16(h, &nsplit, (hipStream_t)s);
    if (hr == LTU_OK) {
      wa.npad = N0 + N1; wa.kpad = 27 * C;
      wa.bpart = ws + (long long)nsplit * wa.npad * wa.kpad;
      return launch_wgrad_reduce(wa, nsplit, (hipStream_t)s);
    }
    if (hr != 1) return hr;
  }
  int rc = ltu_conv3d_wgrad(g0, x, nullptr, dwa, dba, B, H, W, D, C, 0, N0, 1, 1, 1, 0, co_a, ci, ws, ws_floats, dtype, s);
  if (rc) return rc;
  return ltu_conv3d_wgrad(g1, x, nullptr, dwb, dbb, B, H, W, D, C, 0, N1, 1, 1, 1, 0, co_b, ci, ws, ws_floats, dtype, s);
}

// fp32 launchers for the other translation units (upconv.hip)
int launch_nt_f32(const IGemmArgs& g, hipStream_t st) { return launch_nt<float, float>(g, st); }
int launch_tn_f32(WGradArgs& wa, hipStream_t st) { return launch_tn<float>(wa, st); }
