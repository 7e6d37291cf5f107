// Streaming GEMM kernels for the transformer projections (dense, tall: M = tokens >> N, K), bf16 matrix cores.
//
// These GEMMs are HBM-bound (AI ~ 100 flop/B) and, with register-staged tiles, latency-bound: a workgroup keeps only one
// 16 KB tile in flight.  Here the operands travel global -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR staging) into
// a ring of R units per workgroup, R - 1 of them in flight while one is consumed: ~80 KB in flight per CU, the amount
// HBM latency x bandwidth asks for.  One raw s_barrier and one counted s_waitcnt vmcnt(N) per unit; a unit is read only
// after the barrier that follows the wait which retired it.
//
// LDS-DMA writes lane-linear (wave base + 16 B x lane), so bank spreading is done on the SOURCE address: a 16-byte chunk c
// of row r is fetched by the lane whose linear slot is c ^ f(r), and the fragment reads apply the same involution.
#include "gemm_desc.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short rs16x4;
typedef __attribute__((address_space(3))) rs16x4 lds_rs16x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

// The LDS-DMA is issued from inline asm on purpose: hipcc then keeps no book on it and inserts no vmcnt(0) ahead of the
// fragment reads (it would, conservatively, for the builtin); completion is counted by hand below.  M0 = wave-uniform LDS
// byte address of the 1 KiB piece; lane l lands at M0 + 16 l.
__device__ __forceinline__ void glds16(const uint16_t* src, uint16_t* lds_wave_base) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void*)lds_wave_base);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
// retire all but the youngest N LDS-DMA of this wave, then meet the other waves: after it every wave's pieces of the
// oldest unit have landed and every wave has finished reading the unit before it
template <int N>
__device__ __forceinline__ void ring_sync() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// ------------------------------------------------------------------------------------------------ TN (weight gradient)
// dW[n][k] = sum_m G[m][n] X[m][k] for one 128 x 128 tile over the rows of one split.  Unit = 32 rows: G [32][128] then
// X [32][128] bf16 (256-byte rows, 16 chunks); chunk c of row r sits at slot c ^ ((r & 3) << 2), which makes the 4-row x
// 64-byte footprint of a transposing read (ds_read_b64_tr_b16) cover all 64 banks.  4 waves as 2 (n) x 2 (k), 64 x 64 each.
template <int R>
__global__ void __launch_bounds__(256) wgrad_ring_bf16_kernel(const WGradArgs wa) {
  extern __shared__ __attribute__((aligned(1024))) uint16_t smem[];      // [R][2][32][128]
  constexpr int UNIT = 2 * 32 * 128;                                     // elements
  const IGemmArgs& g = wa.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int k_blk = blockIdx.x * 128, n_blk = blockIdx.y * 128;
  const long long m_begin = (long long)blockIdx.z * wa.rows_per_split;
  long long m_end = m_begin + wa.rows_per_split;
  if (m_end > g.M) m_end = g.M;
  const int niter = m_end > m_begin ? (int)((m_end - m_begin) >> 5) : 0;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum[2] = {0.f, 0.f};

  // LDS-DMA sources: wave w fills pieces w and w + 4 (4 rows x 256 B each) of G and of X; lane -> (row = lane >> 4, slot = lane & 15)
  const int prow = 4 * wave + (lane >> 4);
  const int lchunk = (lane & 15) ^ ((lane >> 4) << 2);
  const uint16_t* gsrc = reinterpret_cast<const uint16_t*>(wa.grad) + (m_begin + prow) * wa.ldg + n_blk + lchunk * 8;
  const uint16_t* xsrc = reinterpret_cast<const uint16_t*>(g.a0) + (m_begin + prow) * g.lda0 + k_blk + lchunk * 8;
  const long long gstep16 = 16LL * wa.ldg, xstep16 = 16LL * g.lda0;
  auto issue = [&](int slot) {
    uint16_t* base = smem + slot * UNIT + wave * 512;
    glds16(gsrc, base);
    glds16(gsrc + gstep16, base + 4 * 512);
    glds16(xsrc, base + 4096);
    glds16(xsrc + xstep16, base + 4096 + 4 * 512);
    gsrc += 2 * gstep16;
    xsrc += 2 * xstep16;
  };

  // transposing-read geometry: 16-lane group gq -> columns 16 (gq & 1).., rows 8 (gq >> 1)..; lane 4q + p -> row q, cols 4p..4p+3
  const int gq = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  const int tcol = 16 * (gq & 1) + 4 * tp;
  const int trow = 8 * (gq >> 1) + tq;
  int acol[2], bcol[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    acol[i] = ((((wm * 2 + i) * 4 + (tcol >> 3)) ^ (tq << 2)) << 3) + (tcol & 7);
    bcol[i] = ((((wn * 2 + i) * 4 + (tcol >> 3)) ^ (tq << 2)) << 3) + (tcol & 7);
  }
  const bool do_bias = blockIdx.x == 0 && wn == 0;

  for (int u = 0; u < R - 1 && u < niter; ++u) issue(u);
  for (int it = 0; it < niter; ++it) {
    if (it + R - 2 < niter) ring_sync<4 * (R - 2)>(); else ring_sync<0>();
    if (it + R - 1 < niter) issue((it + R - 1) % R);
    const uint16_t* Gs = smem + (it % R) * UNIT;
    const uint16_t* Xs = Gs + 4096;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint16_t* p0 = Gs + (ks * 16 + trow) * 128 + acol[i];
        union { struct { rs16x4 l, h; } s; bf16x8 v; } u;
        u.s.l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_rs16x4*)p0);
        u.s.h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_rs16x4*)(p0 + 4 * 128));
        a[i] = u.v;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint16_t* p0 = Xs + (ks * 16 + trow) * 128 + bcol[j];
        union { struct { rs16x4 l, h; } s; bf16x8 v; } u;
        u.s.l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_rs16x4*)p0);
        u.s.h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_rs16x4*)(p0 + 4 * 128));
        b[j] = u.v;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      if (do_bias) {          // the A fragment holds 8 rows of column n = lane & 31: column sums for free
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 8; ++e) bsum[i] += (float)a[i][e];
      }
    }
  }

  const int li = lane & 31, lh = lane >> 5;
  float* pz = wa.part + (long long)blockIdx.z * wa.npad * wa.kpad;
  if (do_bias) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float t = bsum[i] + __shfl_xor(bsum[i], 32);
      if (lh == 0) wa.bpart[(long long)blockIdx.z * wa.npad + n_blk + (wm * 2 + i) * 32 + li] = t;
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int k = k_blk + (wn * 2 + j) * 32 + li;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n_blk + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        pz[(long long)n * wa.kpad + k] = acc[i][j][r];
      }
  }
}

#define TN_RING 6
static int ring_blocks() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("LTU_RING_BLOCKS"); v = (e && atoi(e) > 0) ? atoi(e) : 256; }
  return v;
}
static bool ring_enabled() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("LTU_NO_RING"); v = (e && atoi(e)) ? 0 : 1; }
  return v == 1;
}

struct RingGeom { int nk, nn, rows, nsplit; long long ws_floats; };
static RingGeom tn_ring_geometry(long long M, int N, int K) {
  RingGeom t;
  t.nk = K / 128; t.nn = N / 128;
  long long want = ring_blocks() / ((long long)t.nk * t.nn);
  if (want < 1) want = 1;
  long long rows = (M + want - 1) / want;
  if (rows < 512) rows = 512;
  rows = (rows + 31) / 32 * 32;
  t.rows = (int)rows;
  t.nsplit = (int)((M + rows - 1) / rows);
  t.ws_floats = (long long)t.nsplit * N * ((long long)K + 1);
  return t;
}
bool tn_ring_shape_ok(long long M, int N, int K) { return ring_enabled() && M % 32 == 0 && M >= 1024 && N % 128 == 0 && K % 128 == 0; }
long long tn_ring_ws_floats(long long M, int N, int K) { return tn_ring_shape_ok(M, N, K) ? tn_ring_geometry(M, N, K).ws_floats : 0; }

// dense weight gradient through the ring kernel + wgrad_reduce_kernel; returns 1 when the shape is not handled
int launch_tn_ring_bf16(WGradArgs& wa, hipStream_t st) {
  const IGemmArgs& g = wa.g;
  if (wa.part == nullptr || g.ntaps != 1 || g.K != g.C || g.c0 != g.C || !g.out_identity) return 1;
  if (!tn_ring_shape_ok(g.M, g.N, g.K) || g.lda0 % 8 || wa.ldg % 8) return 1;
  if (((uintptr_t)g.a0 | (uintptr_t)wa.grad) & 15) return 1;
  const RingGeom t = tn_ring_geometry(g.M, g.N, g.K);
  wa.rows_per_split = t.rows;
  wa.npad = g.N; wa.kpad = g.K;
  wa.bpart = wa.part + (long long)t.nsplit * wa.npad * wa.kpad;
  constexpr int smem_bytes = TN_RING * 2 * 32 * 128 * 2;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_ring_bf16_kernel<TN_RING>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              smem_bytes);
    attr_done = true;
  }
  hipLaunchKernelGGL((wgrad_ring_bf16_kernel<TN_RING>), dim3(t.nk, t.nn, t.nsplit), dim3(256), smem_bytes, st, wa);
  int rc = ltu_check_launch();
  if (rc) return rc;
  return launch_wgrad_reduce(wa, t.nsplit, st);
}
