// bf16 matrix-core versions of the tap-table implicit GEMM (see gemm.hip for the formulation).
//
//   NT:  Y = gather(A) . W^T     v_mfma_f32_32x32x16_bf16, LDS tiles [rows][32 k] bf16 with 80-byte rows
//                                 (conflict-free ds_read_b128 fragments), register-staged double buffering,
//                                 output staged through LDS so that HBM stores are whole 16-byte vectors.
//   TN:  dW += G^T . gather(A)   both operands are "reduction-major" in memory; tiles are stored as loaded
//                                 ([m][n] / [m][k]) and the MFMA fragments are fetched with the transposing
//                                 LDS read ds_read_b64_tr_b16 (4 rows x 16 columns per 16-lane group).
// Activations and weight operands are bf16, accumulation fp32, bias / weight gradients fp32.
#include <stdlib.h>

#include "gemm_desc.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// M < 2^31 is enforced by the launchers: 32-bit divisions only
__device__ __forceinline__ RowCoord split_row_b(const IGemmArgs& g, long long m) {
  RowCoord r;
  unsigned t = (unsigned)m;
  const unsigned q0 = t / (unsigned)g.rd;
  r.d = (int)(t - q0 * (unsigned)g.rd);
  const unsigned q1 = q0 / (unsigned)g.rw;
  r.w = (int)(q0 - q1 * (unsigned)g.rw);
  const unsigned q2 = q1 / (unsigned)g.rh;
  r.h = (int)(q1 - q2 * (unsigned)g.rh);
  r.b = (int)q2;
  return r;
}
__device__ __forceinline__ void advance_row(const IGemmArgs& g, RowCoord& r, int delta) {
  r.d += delta;
  while (r.d >= g.rd) {
    r.d -= g.rd;
    if (++r.w == g.rw) {
      r.w = 0;
      if (++r.h == g.rh) { r.h = 0; ++r.b; }
    }
  }
}
// position inside the K axis as (tap slot, channel); advanced without divisions
struct KCur {
  int slot, c;
};
__device__ __forceinline__ KCur kcur_init(const IGemmArgs& g, int k) {
  KCur q;
  q.slot = k / g.C;          // k >= K gives slot >= ntaps, i.e. "past the end"
  q.c = k - q.slot * g.C;
  return q;
}
__device__ __forceinline__ void kcur_advance(const IGemmArgs& g, KCur& q, int delta) {
  q.c += delta;
  while (q.c >= g.C && q.slot < g.ntaps) { q.c -= g.C; ++q.slot; }
}
// 8 consecutive bf16 channels of the A operand at (row, cursor); zero outside the source / beyond K
// `taps`: the workgroup's LDS copy of g.tap.  Indexing the kernel-argument copy with a per-thread slot is a vector load from
// memory whose result the address computation needs at once: its s_waitcnt vmcnt(0) also drained the data loads of the tiles
// that were supposed to stay in flight (vmcnt is one in-order counter), so every K tile paid a full memory round trip.
__device__ __forceinline__ uint4 gather_a8c(const IGemmArgs& g, const Tap* taps, bool row_ok, const RowCoord& rc, const KCur& q) {
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  if (!row_ok || q.slot >= g.ntaps) return z;
  const Tap tp = taps[q.slot];
  int h = rc.h * g.mh + tp.dh, w = rc.w * g.mw + tp.dw, d = rc.d * g.md + tp.dd;
  if ((unsigned)h >= (unsigned)g.sh || (unsigned)w >= (unsigned)g.sw || (unsigned)d >= (unsigned)g.sd) return z;
  int ph = g.sh, pw = g.sw, pd = g.sd;
  if (g.ups) {
    h >>= 1; w >>= 1; d >>= 1;
    ph >>= 1; pw >>= 1; pd >>= 1;
  }
  const long long vox = (((long long)rc.b * ph + h) * pw + w) * pd + d;
  const uint16_t* p = q.c < g.c0 ? reinterpret_cast<const uint16_t*>(g.a0) + vox * g.lda0 + q.c
                                 : reinterpret_cast<const uint16_t*>(g.a1) + vox * g.lda1 + (q.c - g.c0);
  return *reinterpret_cast<const uint4*>(p);
}

// The same as an ADDRESS + validity: the caller loads unconditionally (from g.a0 when there is nothing to read) and selects zero when
// the registers go to LDS.  With the test around the load (gather_a8c) hipcc gives every load of a tile a basic block of its own
// and waits for the earlier ones there: the tiles that were meant to be in flight behind the MFMAs arrived one round trip at a time.
__device__ __forceinline__ const uint16_t* gather_a8c_ptr(const IGemmArgs& g, const Tap* taps, bool row_ok, const RowCoord& rc, const KCur& q,
                                                          bool& valid) {
  const bool kin = q.slot < g.ntaps;
  const Tap tp = taps[kin ? q.slot : 0];
  int h = rc.h * g.mh + tp.dh, w = rc.w * g.mw + tp.dw, d = rc.d * g.md + tp.dd;
  valid = row_ok && kin && (unsigned)h < (unsigned)g.sh && (unsigned)w < (unsigned)g.sw && (unsigned)d < (unsigned)g.sd;
  int ph = g.sh, pw = g.sw, pd = g.sd;
  if (g.ups) {
    h >>= 1; w >>= 1; d >>= 1;
    ph >>= 1; pw >>= 1; pd >>= 1;
  }
  const long long vox = valid ? (((long long)rc.b * ph + h) * pw + w) * pd + d : 0;
  const int c = valid ? q.c : 0;
  return c < g.c0 ? reinterpret_cast<const uint16_t*>(g.a0) + vox * g.lda0 + c
                  : reinterpret_cast<const uint16_t*>(g.a1) + vox * g.lda1 + (c - g.c0);
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
// BK = K-extent of one LDS tile (32 / 64 / 128 bf16), NBUF = LDS tile buffers.  One tile costs one global
// round trip, so BK is chosen so that a tile carries enough matrix work to cover it (dense projections with
// K <= 128 take the whole K in ONE tile; convolutions use 64).  LDS rows are BK+8 elements: a lane group of a
// ds_read_b128 fragment read (16 rows, same 16-byte column) then covers 16 distinct 4-bank slots.
template <int WM, int WN, int TM, int TN, int BK, int NBUF>
__global__ void __launch_bounds__(WM* WN * 64) igemm_nt_bf16_kernel(const IGemmArgs g) {
  constexpr int NT = WM * WN * 64, BM = WM * TM * 32, BN = WN * TN * 32, LDK = BK + 8, CPRK = BK / 8;
  constexpr int LA = (BM * CPRK + NT - 1) / NT, LB = (BN * CPRK + NT - 1) / NT;
  constexpr int LDC = BN + 8;                                   // staging row stride (bf16 elements)
  constexpr int TILE_ELEMS = NBUF * (BM + BN) * LDK;
  constexpr int STAGE_ELEMS = BM * LDC;
  constexpr int SMEM_ELEMS = TILE_ELEMS > STAGE_ELEMS ? TILE_ELEMS : STAGE_ELEMS;
  __shared__ __attribute__((aligned(16))) uint16_t smem[SMEM_ELEMS];
  uint16_t* As = smem;                          // [NBUF][BM][LDK]
  uint16_t* Bs = smem + NBUF * BM * LDK;        // [NBUF][BN][LDK]
  __shared__ Tap s_tap[64];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 64) s_tap[tid] = g.tap[tid];
  __syncthreads();
  const int wm = wave / WN, wn = wave % WN;
  const long long m_blk = (long long)blockIdx.x * BM;
  const int n_blk = blockIdx.y * BN;

  RowCoord rc[LA];
  bool rok[LA];
#pragma unroll
  for (int p = 0; p < LA; ++p) {
    const int idx = tid + p * NT;
    const long long m = m_blk + idx / CPRK;
    rok[p] = (idx < BM * CPRK) && (m < g.M);
    rc[p] = split_row_b(g, rok[p] ? m : 0);
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int nkt = (g.K + BK - 1) / BK;
  int kt0 = 0;
  if (g.part != nullptr) {                 // split over K: this workgroup's tile range
    kt0 = (int)blockIdx.z * g.kt_per_split;
    const int kt1 = kt0 + g.kt_per_split < nkt ? kt0 + g.kt_per_split : nkt;
    nkt = kt1 > kt0 ? kt1 - kt0 : 0;
  }
  // every load of this thread sits at the same k offset inside a tile (NT % CPRK == 0): one cursor
  KCur cur = kcur_init(g, kt0 * BK + (tid % CPRK) * 8);
  const uint16_t* wrow[LB];
#pragma unroll
  for (int p = 0; p < LB; ++p) {
    const int idx = tid + p * NT;
    const int n = n_blk + idx / CPRK;
    wrow[p] = nullptr;
    if (idx < BN * CPRK && n < g.N) {
      const int nper = g.N / g.nseg;
      const int seg = n / nper;
      wrow[p] = reinterpret_cast<const uint16_t*>(g.w[seg]) + (long long)(n - seg * nper) * g.wrow;
    }
  }

  // loads the NEXT tile of the K sequence into a register set; `okm` = which of its vectors are real (bits 0.. A, bits LA.. B)
  auto load_tile = [&](uint4 (&ra)[LA], uint4 (&rb)[LB], unsigned& okm) {
    okm = 0;
#pragma unroll
    for (int p = 0; p < LA; ++p) {
      bool valid;
      const uint16_t* src = gather_a8c_ptr(g, s_tap, rok[p], rc[p], cur, valid);
      ra[p] = *reinterpret_cast<const uint4*>(src);
      okm |= valid ? 1u << p : 0u;
    }
    const bool kin = cur.slot < g.ntaps;
    const int woff = kin ? (int)s_tap[cur.slot].wt * g.C + cur.c : 0;
#pragma unroll
    for (int p = 0; p < LB; ++p) {
      const bool valid = kin && wrow[p] != nullptr;
      const uint16_t* src = valid ? wrow[p] + woff : reinterpret_cast<const uint16_t*>(g.w[0]);
      rb[p] = *reinterpret_cast<const uint4*>(src);
      okm |= valid ? 1u << (LA + p) : 0u;
    }
    kcur_advance(g, cur, BK);
  };
  // (the zero is selected per component: `ok ? ra[p] : z` on the whole vector selects an ADDRESS and sends the register sets to scratch)
  auto keep = [](uint4 v, bool ok) { return make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u); };
  auto store_tile = [&](const uint4 (&ra)[LA], const uint4 (&rb)[LB], unsigned okm, int buf) {
#pragma unroll
    for (int p = 0; p < LA; ++p) {
      const int idx = tid + p * NT;
      if (idx < BM * CPRK) *reinterpret_cast<uint4*>(&As[(buf * BM + idx / CPRK) * LDK + (idx % CPRK) * 8]) = keep(ra[p], (okm >> p) & 1u);
    }
#pragma unroll
    for (int p = 0; p < LB; ++p) {
      const int idx = tid + p * NT;
      if (idx < BN * CPRK) *reinterpret_cast<uint4*>(&Bs[(buf * BN + idx / CPRK) * LDK + (idx % CPRK) * 8]) = keep(rb[p], (okm >> (LA + p)) & 1u);
    }
  };
  const int li = lane & 31, lh = lane >> 5;
  auto compute = [&](int buf) {
    if (g.dbg & 2) return;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
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
  };

  static_assert(LA + LB <= 32, "validity bits");
  uint4 ra0[LA], rb0[LB];
  unsigned ok0 = 0;
  if (NBUF == 2) {
    // 3-stage pipeline: tile t in LDS being consumed, tile t+1 arrived in registers, tile t+2 in flight.
    uint4 ra1[LA], rb1[LB];
    unsigned ok1 = 0;
    load_tile(ra0, rb0, ok0);                     // tile 0
    store_tile(ra0, rb0, ok0, 0);
    if (nkt > 1) load_tile(ra0, rb0, ok0);        // tile 1
    if (nkt > 2) load_tile(ra1, rb1, ok1);        // tile 2
    __syncthreads();
    for (int kt = 0; kt < nkt; kt += 2) {
      compute(0);                                 // tile kt
      if (kt + 1 < nkt) {
        store_tile(ra0, rb0, ok0, 1);             // tile kt+1 (buffer 1 was released by the previous barrier)
        if (kt + 3 < nkt) load_tile(ra0, rb0, ok0);    // tile kt+3
      }
      __syncthreads();
      if (kt + 1 >= nkt) break;
      compute(1);                                 // tile kt+1
      if (kt + 2 < nkt) {
        store_tile(ra1, rb1, ok1, 0);             // tile kt+2
        if (kt + 4 < nkt) load_tile(ra1, rb1, ok1);    // tile kt+4
      }
      __syncthreads();
    }
  } else {
    load_tile(ra0, rb0, ok0);
    store_tile(ra0, rb0, ok0, 0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
      if (kt + 1 < nkt) load_tile(ra0, rb0, ok0);
      compute(0);
      if (kt + 1 < nkt) {
        __syncthreads();                          // single buffer: everyone is done reading before it is refilled
        store_tile(ra0, rb0, ok0, 0);
      }
      __syncthreads();
    }
  }

  if (g.part != nullptr) {                 // fp32 partial tile (folded by igemm_fold_kernel)
    float* pz = g.part + (long long)blockIdx.z * g.M * g.N;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n_blk + (wn * TN + j) * 32 + li;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long long m = m_blk + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m < g.M && n < g.N) pz[m * g.N + n] = acc[i][j][r];
        }
    }
    return;
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
  if (g.dbg & 1) return;
  const bool wide = !g.accum && (g.N % 8 == 0) && (g.n0 % 8 == 0) && (g.ldo0 % 8 == 0) && (g.ldo1 % 8 == 0);
  if (wide) {
    constexpr int CPR = BN / 8;               // 16-byte chunks per row
    for (int idx = tid; idx < BM * CPR; idx += NT) {
      const int ml = idx / CPR, nl = (idx % CPR) * 8;
      const long long m = m_blk + ml;
      const int n = n_blk + nl;
      if (m >= g.M || n >= g.N) continue;
      const uint4 v = *reinterpret_cast<const uint4*>(&Cs[ml * LDC + nl]);
      uint16_t* dst;
      if (n < g.n0) dst = reinterpret_cast<uint16_t*>(g.o0) + out_voxel_b(g, m) * g.ldo0 + n;
      else dst = reinterpret_cast<uint16_t*>(g.o1) + out_voxel_b(g, m) * g.ldo1 + (n - g.n0);
      *reinterpret_cast<uint4*>(dst) = v;
    }
    return;
  }
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
      v.x = pack_bf16x2(s0, s1);
      v.y = pack_bf16x2(s2, s3);
    }
    *reinterpret_cast<uint2*>(dst) = v;
  }
}

// out[m][n] = bf16(sum_z part[z][m][n] + bias[n]) for the K-split launches (out_identity, N % 4 == 0)
__global__ void __launch_bounds__(256) igemm_fold_kernel(const IGemmArgs g) {
  const int nq = g.N / 4;
  const long long total = g.M * nq;
  const int nper = g.N / g.nseg;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const long long m = t / nq;
    const int n = (int)(t - m * nq) * 4;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int seg = (n + e) / nper;
      v[e] = g.bias[seg] ? g.bias[seg][n + e - seg * nper] : 0.f;
    }
    for (int z = 0; z < g.ksplit; ++z) {
      const float4 q = *reinterpret_cast<const float4*>(g.part + ((long long)z * g.M + m) * g.N + n);
      v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
    }
    uint2 pk;
    pk.x = pack_bf16x2(v[0], v[1]);
    pk.y = pack_bf16x2(v[2], v[3]);
    uint16_t* dst = n < g.n0 ? reinterpret_cast<uint16_t*>(g.o0) + out_voxel_b(g, m) * g.ldo0 + n
                             : reinterpret_cast<uint16_t*>(g.o1) + out_voxel_b(g, m) * g.ldo1 + (n - g.n0);
    *reinterpret_cast<uint2*>(dst) = pk;
  }
}

// split geometry of the N > 64 configuration (64 x 128 tiles, BK 64): only grids that leave the chip mostly idle
static int nt_split(long long M, int N, int K, int* kps) {
  if (N <= 64) return 1;
  const long long blocks = ((M + 63) / 64) * ((N + 127) / 128);
  const int nkt = (K + 63) / 64;
  if (blocks > 300 || nkt < 16) return 1;
  int want = (int)((640 + blocks - 1) / blocks);
  if (want > 8) want = 8;
  if (want > nkt / 8) want = nkt / 8;
  if (want < 2) return 1;
  *kps = (nkt + want - 1) / want;
  return (nkt + *kps - 1) / *kps;
}
long long igemm_nt_ws_floats(long long M, int N, int K) {
  int kps = 0;
  const int ks = nt_split(M, N, K, &kps);
  return ks > 1 ? (long long)ks * M * N : 0;
}

// tuning override for experiments: LTU_NT_VARIANT = 0 (auto) | 1 (BK 32, 2 buffers) | 2 (BK 64, 2 buffers) | 3 (BK 128, 1 buffer)
static int nt_variant() { return ltu_knob("LTU_NT_VARIANT", 0); }

template <int WM, int WN, int TM, int TN>
static void launch_nt_cfg(const IGemmArgs& g, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  dim3 grid(cdiv(g.M, BM), cdiv(g.N, BN));
  int v = nt_variant();
  // measured on MI355X: BK 32 (several workgroups per CU) wins whenever the grid fills the chip more than once; a grid of
  // at most ~2 workgroups per CU is latency-bound on its serial K loop, where BK 64 halves the number of round trips
  if (v == 0) v = ((long long)grid.x * grid.y <= 512 && g.K >= 128) ? 2 : 1;
  if (v == 3 && g.K > 128) v = 2;
  if (v == 3)
    hipLaunchKernelGGL((igemm_nt_bf16_kernel<WM, WN, TM, TN, 128, 1>), grid, dim3(WM * WN * 64), 0, st, g);
  else if (v == 1)
    hipLaunchKernelGGL((igemm_nt_bf16_kernel<WM, WN, TM, TN, 32, 2>), grid, dim3(WM * WN * 64), 0, st, g);
  else
    hipLaunchKernelGGL((igemm_nt_bf16_kernel<WM, WN, TM, TN, 64, 2>), grid, dim3(WM * WN * 64), 0, st, g);
}

int launch_nt_bf16(const IGemmArgs& g_in, hipStream_t st) {
  IGemmArgs g = g_in;
  g.dbg = ltu_knob("LTU_NT_DBG", 0);
  if (g.M <= 0 || g.N <= 0) return LTU_OK;
  if (g.M >= (1LL << 31)) return LTU_E_SHAPE;
  if (g.C % 8 || g.c0 % 8 || g.lda0 % 8 || g.lda1 % 8 || g.wrow % 8 || g.N % 4 || g.n0 % 4 || g.ldo0 % 4 || g.ldo1 % 4)
    return LTU_E_SHAPE;
  {
    const int rr = launch_nt_ring_bf16(g, st);
    if (rr != 1) return rr;
  }
  int small_tile = -1;
  small_tile = ltu_knob("LTU_NT_SMALLTILE", 0);
  if (g.part != nullptr) {                   // K split (workspace given by the caller): 64 x 128 tiles, BK 64
    int kps = 0;
    const bool ok = g.out_identity && !g.accum && g.N % 4 == 0 && g.n0 % 4 == 0 && !ltu_knob("LTU_NO_NT_SPLIT", 0);
    const int ks = ok ? nt_split(g.M, g.N, g.K, &kps) : 1;
    if (ks > 1) {
      if ((long long)ks * g.M * g.N > g.part_floats) return LTU_E_ARG;         // the workspace is shorter than this split needs
      g.ksplit = ks; g.kt_per_split = kps;
      dim3 grid(cdiv(g.M, 64), cdiv(g.N, 128), ks);
      hipLaunchKernelGGL((igemm_nt_bf16_kernel<2, 2, 1, 2, 64, 2>), grid, dim3(256), 0, st, g);
      long long fb = (g.M * (g.N / 4) + 255) / 256;
      if (fb > 2048) fb = 2048;
      hipLaunchKernelGGL(igemm_fold_kernel, dim3((unsigned)fb), dim3(256), 0, st, g);
      return ltu_check_launch();
    }
    g.part = nullptr;
  }
  if (g.N > 64) {
    if (small_tile == 2) launch_nt_cfg<2, 2, 1, 1>(g, st);
    else if (small_tile == 3) launch_nt_cfg<2, 2, 2, 2>(g, st);
    else launch_nt_cfg<2, 2, 1, 2>(g, st);      // 64x128 tiles: ~100 registers, 30 KB LDS -> 5 workgroups per CU in flight
  } else if (g.N > 32) {
    launch_nt_cfg<4, 1, 1, 2>(g, st);
  } else {
    launch_nt_cfg<4, 1, 1, 1>(g, st);
  }
  return ltu_check_launch();
}

// ------------------------------------------------------------------------------------------------ TN
// tile: BNn (n) x BKk (k) of dW, BR = 32 reduction rows per iteration (40 KB LDS: 3-4 workgroups per CU; BR = 64 measured slower).  Gs[BR][BNn+32], Xs[BR][BKk+32] bf16
// (the 64-byte pad makes the 4-row transposing reads of a 32-lane half hit 4 disjoint bank ranges).
template <int WM, int WN, int TM, int TN>
__global__ void __launch_bounds__(WM* WN * 64) wgrad_tn_bf16_kernel(const WGradArgs wa) {
  const IGemmArgs& g = wa.g;
  constexpr int NT = WM * WN * 64, BNn = WM * TM * 32, BKk = WN * TN * 32, BR = 32;
  constexpr int LDG = BNn + 32, LDX = BKk + 32;
  constexpr int LG = (BR * BNn / 8 + NT - 1) / NT, LX = (BR * BKk / 8 + NT - 1) / NT;
  __shared__ __attribute__((aligned(16))) uint16_t Gs[2][BR][LDG];
  __shared__ __attribute__((aligned(16))) uint16_t Xs[2][BR][LDX];
  __shared__ Tap s_tap[64];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int k_blk = blockIdx.x * BKk, n_blk = blockIdx.y * BNn;
  const long long m_begin = (long long)blockIdx.z * wa.rows_per_split;
  long long m_end = m_begin + wa.rows_per_split;
  if (m_end > g.M) m_end = g.M;
  if (m_begin >= m_end) return;
  if (tid < 64) s_tap[tid] = g.tap[tid];
  __syncthreads();

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
  // the k position of each gather of this thread never changes; its row advances by BR per iteration
  KCur xcur[LX];
  RowCoord xrc[LX];
#pragma unroll
  for (int p = 0; p < LX; ++p) {
    const int idx = tid + p * NT;
    xcur[p] = kcur_init(g, k_blk + (idx % (BKk / 8)) * 8);
    if (k_blk + (idx % (BKk / 8)) * 8 >= g.K) xcur[p].slot = g.ntaps;
    long long m = m_begin + idx / (BKk / 8);
    xrc[p] = split_row_b(g, m < g.M ? m : 0);
  }

  auto load_tile = [&](int it) {
    const long long m0 = m_begin + (long long)it * BR;
#pragma unroll
    for (int p = 0; p < LG; ++p) {
      const int idx = tid + p * NT;
      const int row = idx / (BNn / 8), nq = (idx % (BNn / 8)) * 8;
      const long long m = m0 + row;
      const int n = n_blk + nq;
      rg[p] = (idx < BR * BNn / 8 && m < m_end && n < g.N)
                  ? *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(wa.grad) + out_voxel_b(g, m) * wa.ldg + n)
                  : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int p = 0; p < LX; ++p) {
      const int idx = tid + p * NT;
      const int row = idx / (BKk / 8);
      const long long m = m0 + row;
      const bool ok = idx < BR * BKk / 8 && m < m_end;
      rx[p] = gather_a8c(g, s_tap, ok, xrc[p], xcur[p]);
      advance_row(g, xrc[p], BR);
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
  const bool do_bias = (wa.db != nullptr || wa.part != nullptr) && blockIdx.x == 0;
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

  if (wa.part != nullptr) {
    // two-stage: plain coalesced stores of this split's tile (32 consecutive k per half-wave = 128-byte rows)
    float* pz = wa.part + (long long)blockIdx.z * wa.npad * wa.kpad;
    if (do_bias && tid < BNn) wa.bpart[(long long)blockIdx.z * wa.npad + n_blk + tid] = bsum;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int k = k_blk + (wn * TN + j) * 32 + li;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n_blk + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          pz[(long long)n * wa.kpad + k] = acc[i][j][r];
        }
    }
    return;
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
    for (int i = 0; i < TM; ++i)
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

// second stage: gradient[n][k] += sum over splits of part[z][n][k]; db[n] += sum_z bpart[z][n].
// Workgroups [0, wblocks) fold weights: each owns 64 groups of 4 consecutive k (16-byte loads; K and kpad are multiples of 4), its
// 4 thread groups sum every 4th split with independent loads in flight and combine through LDS; the remaining workgroups fold the
// bias rows.  The scatter into the PyTorch conv layout ([Co][Ci][27]: stride-27 writes) is per element.
// destination of element (n, k) in the PyTorch layouts (nullptr: padding).  The fold reads the old values of its four
// destinations FIRST, together with the partials: a `+=` at the end is a read round trip behind the previous element's store (and
// no-return float atomics, tried instead, were 1.4x slower here: four dword atomics per lane against one round of loads).
__device__ __forceinline__ float* wgrad_reduce_dst(const WGradArgs& wa, int n, int k) {
  const IGemmArgs& g = wa.g;
  if (wa.t_co) {
    const int slot = k / g.C, c = k - slot * g.C;
    if (n < wa.t_co && c < wa.t_ci) return wa.dw + ((long long)n * wa.t_ci + c) * 27 + g.tap[slot].wt;
    if (wa.dw2 != nullptr && n >= wa.n0_2 && n - wa.n0_2 < wa.t_co2 && c < wa.t_ci)
      return wa.dw2 + ((long long)(n - wa.n0_2) * wa.t_ci + c) * 27 + g.tap[slot].wt;
    return nullptr;
  }
  if (wa.nseg_w > 1) {
    const int nper = g.N / wa.nseg_w, seg = n / nper;
    return wa.dwseg[seg] + (long long)(n - seg * nper) * g.K + k;
  }
  const int slot = k / g.C, c = k - slot * g.C;
  return wa.dw + (long long)n * g.wrow + (long long)g.tap[slot].wt * g.C + c;
}
// SG = thread groups that share the splits of one output quad (256 / SG quads per workgroup): 4 for large gradients, 16 / 64 for the
// small ones with hundreds of splits (conv weights of the first levels: a few thousand elements), which would otherwise be a
// grid of a few dozen workgroups walking long serial loops.
template <int SG>
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const WGradArgs wa, int nsplit, int wblocks) {
  constexpr int NQ = 256 / SG;
  __shared__ float4 red[SG - 1][NQ];
  const IGemmArgs& g = wa.g;
  const long long zs = (long long)wa.npad * wa.kpad;
  if ((int)blockIdx.x < wblocks) {
    const int l64 = threadIdx.x % NQ, grp = threadIdx.x / NQ;
    const int kq = g.K >> 2;                                       // float4 groups per row
    const long long q = (long long)blockIdx.x * NQ + l64;
    const int n = (int)(q / kq), k = (int)(q - (long long)n * kq) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float* dst[4] = {nullptr, nullptr, nullptr, nullptr};
    float old[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < g.N) {
      if (grp == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dst[e] = wgrad_reduce_dst(wa, n, k + e);
          if (dst[e] != nullptr) old[e] = *dst[e];
        }
      }
      const float* p = wa.part + (long long)n * wa.kpad + k;
#pragma unroll 8
      for (int z = grp; z < nsplit; z += SG) {
        const float4 v = *reinterpret_cast<const float4*>(p + z * zs);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    if (grp > 0) red[grp - 1][l64] = acc;
    __syncthreads();
    if (grp != 0 || n >= g.N) return;
#pragma unroll
    for (int r = 0; r < SG - 1; ++r) {
      const float4 v = red[r][l64];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const float av[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (dst[e] != nullptr) *dst[e] = old[e] + av[e];
    return;
  }
  // bias rows: 16 outputs per workgroup, 16 thread groups share the splits (one thread per output would walk hundreds of splits
  // as a serial chain of dependent round trips: that chain, not the weight fold, set this kernel's duration)
  __shared__ float bred[16][16];
  const int bl = threadIdx.x & 15, bg = threadIdx.x >> 4;
  const int n = ((int)blockIdx.x - wblocks) * 16 + bl;
  float v = 0.f;
  if (n < g.N) {
#pragma unroll 8
    for (int z = bg; z < nsplit; z += 16) v += wa.bpart[(long long)z * wa.npad + n];
  }
  bred[bg][bl] = v;
  __syncthreads();
  if (bg != 0 || n >= g.N) return;
#pragma unroll
  for (int r = 1; r < 16; ++r) v += bred[r][bl];
  if (wa.nseg_w > 1) {
    const int nper = g.N / wa.nseg_w, seg = n / nper;
    if (wa.dbseg[seg]) wa.dbseg[seg][n - seg * nper] += v;
  } else if (wa.db && n < (wa.t_co ? wa.t_co : g.N)) {
    wa.db[n] += v;
  } else if (wa.db2 != nullptr && n >= wa.n0_2 && n - wa.n0_2 < wa.t_co2) {
    wa.db2[n - wa.n0_2] += v;
  }
}

int launch_wgrad_reduce(const WGradArgs& wa, int nsplit, hipStream_t st) {
  if (wa.g.K % 4 || wa.kpad % 4) return LTU_E_SHAPE;
  const long long quads = (long long)wa.g.N * (wa.g.K / 4);
  if (quads < 256 * 16 && nsplit >= 128) {          // a few thousand elements under hundreds of splits (first-level conv weights)
    const int wblocks = (int)((quads + 3) / 4);
    hipLaunchKernelGGL(wgrad_reduce_kernel<64>, dim3((unsigned)(wblocks + (wa.g.N + 15) / 16)), dim3(256), 0, st, wa, nsplit, wblocks);
  } else if (quads < 256 * 64 && nsplit >= 32) {
    const int wblocks = (int)((quads + 15) / 16);
    hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3((unsigned)(wblocks + (wa.g.N + 15) / 16)), dim3(256), 0, st, wa, nsplit, wblocks);
  } else {
    const int wblocks = (int)((quads + 63) / 64);
    hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3((unsigned)(wblocks + (wa.g.N + 15) / 16)), dim3(256), 0, st, wa, nsplit, wblocks);
  }
  return ltu_check_launch();
}

int launch_tn_bf16(WGradArgs& wa, hipStream_t st) {
  const IGemmArgs& g = wa.g;
  if (g.M <= 0 || g.N <= 0) return LTU_OK;
  if (g.M >= (1LL << 31)) return LTU_E_SHAPE;
  if (g.C % 8 || g.c0 % 8 || g.lda0 % 8 || g.lda1 % 8 || g.N % 8 || wa.ldg % 8) return LTU_E_SHAPE;
  {
    const int rr = launch_tn_ring_bf16(wa, st);
    if (rr != 1) return rr;
  }
  const TnGeom t = tn_geometry(g.M, g.N, g.K, 32);
  wa.rows_per_split = t.rows;
  if (wa.part != nullptr) {
    if (t.ws_floats > wa.part_floats) return LTU_E_ARG;
    wa.npad = t.nn * t.bn;
    wa.kpad = t.nk * t.bk;
    wa.bpart = wa.part + (long long)t.nsplit * wa.npad * wa.kpad;
  }
  dim3 grid(t.nk, t.nn, t.nsplit);
  if (g.N > 64) hipLaunchKernelGGL((wgrad_tn_bf16_kernel<2, 2, 2, 2>), grid, dim3(256), 0, st, wa);
  else if (g.N > 32) hipLaunchKernelGGL((wgrad_tn_bf16_kernel<1, 4, 2, 1>), grid, dim3(256), 0, st, wa);
  else hipLaunchKernelGGL((wgrad_tn_bf16_kernel<1, 4, 1, 1>), grid, dim3(256), 0, st, wa);
  if (wa.part != nullptr) return launch_wgrad_reduce(wa, t.nsplit, st);
  return ltu_check_launch();
}
