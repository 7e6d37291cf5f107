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
#include <type_traits>

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
// 4-byte variant: lane l lands at M0 + 4 l (used for the bias row of the projection kernel)
__device__ __forceinline__ void glds4(const float* src, float* lds_wave_base) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void*)lds_wave_base);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
__device__ __attribute__((aligned(16))) float ring_zero_f32[4];          // source of absent / out-of-range bias entries
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
// One 128 x 128 tile of dW over rows [m_begin, m_end): partial tile to pz [npad][kpad] (+ bias partial row to bz [npad] from the
// workgroups with k_blk == 0).
template <int R>
__device__ __forceinline__ void wgrad_ring_tile(const uint16_t* __restrict__ grad, int ldg, const uint16_t* __restrict__ x, int lda,
                                                long long m_begin, long long m_end, int n_blk, int k_blk, float* __restrict__ pz,
                                                float* __restrict__ bz, int kpad, uint16_t* smem, bool accum = false) {
  constexpr int UNIT = 2 * 32 * 128;                                     // elements
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int niter = m_end > m_begin ? (int)((m_end - m_begin) >> 5) : 0;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum = 0.f;

  // LDS-DMA sources: wave w fills pieces w and w + 4 (4 rows x 256 B each) of G and of X; lane -> (row = lane >> 4, slot = lane & 15)
  const int prow = 4 * wave + (lane >> 4);
  const int lchunk = (lane & 15) ^ ((lane >> 4) << 2);
  const uint16_t* gsrc = grad + (m_begin + prow) * ldg + n_blk + lchunk * 8;
  const uint16_t* xsrc = x + (m_begin + prow) * lda + k_blk + lchunk * 8;
  const long long gstep16 = 16LL * ldg, xstep16 = 16LL * lda;
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
  // bias gradient = column sums of G.  Every wave sums ONE of its two G fragments (wave (wm, wn) takes tile wm * 2 + wn), with no
  // test in the loop: summed by the wn == 0 waves only, inside `if (do_bias)`, the k loop was cut into basic blocks at every step
  // (the wave index comes from threadIdx: a divergent branch for the compiler) and the two busy waves carried 32 extra VALU
  // instructions per step - 10 % of the grouped weight-gradient kernel.

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
      {                       // the A fragment holds 8 rows of column n = lane & 31: column sums from registers
        const bf16x8 ab = wn ? a[1] : a[0];
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum += (float)ab[e];
      }
    }
  }

  const int li = lane & 31, lh = lane >> 5;
  if (accum) {
    // the only split of its tile: the sums go straight into the gradient (pz / bz = the gradient block, bz nullable), no fold
    if (k_blk == 0 && bz != nullptr) {
      const float t = xhalf_combine<LtuAdd>(bsum);
      if (lh == 0) bz[n_blk + (wm * 2 + wn) * 32 + li] += t;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k_blk + (wn * 2 + j) * 32 + li;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = pz[(long long)(n_blk + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * kpad + k];
#pragma unroll
        for (int r = 0; r < 16; ++r)
          pz[(long long)(n_blk + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * kpad + k] = old[r] + acc[i][j][r];
      }
    }
    return;
  }
  if (k_blk == 0) {
    const float t = xhalf_combine<LtuAdd>(bsum);
    if (lh == 0) bz[n_blk + (wm * 2 + wn) * 32 + li] = t;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int k = k_blk + (wn * 2 + j) * 32 + li;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n_blk + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        pz[(long long)n * kpad + k] = acc[i][j][r];
      }
  }
}

template <int R>
__global__ void __launch_bounds__(256) wgrad_ring_bf16_kernel(const WGradArgs wa) {
  extern __shared__ __attribute__((aligned(1024))) uint16_t smem[];      // [R][2][32][128]
  const IGemmArgs& g = wa.g;
  const long long m_begin = (long long)blockIdx.z * wa.rows_per_split;
  long long m_end = m_begin + wa.rows_per_split;
  if (m_end > g.M) m_end = g.M;
  wgrad_ring_tile<R>(reinterpret_cast<const uint16_t*>(wa.grad), wa.ldg, reinterpret_cast<const uint16_t*>(g.a0), g.lda0, m_begin, m_end,
                     blockIdx.y * 128, blockIdx.x * 128, wa.part + (long long)blockIdx.z * wa.npad * wa.kpad,
                     wa.bpart + (long long)blockIdx.z * wa.npad, wa.kpad, smem);
}

// Several weight gradients in ONE launch (the four projections of a transformer layer: qkv, out, ffn1, ffn2).  Launched
// one by one each of them spreads its one to six 128 x 128 tiles over ~256 workgroups, i.e. 40-240 row splits whose fp32 partial
// tiles (16 MB per call) have to be written and folded; together they offer 8-32 tiles, so 8-32 splits fill the chip: a quarter of
// the partial traffic, one launch instead of four, one fold (ltu_reduce_batch) instead of four.
struct WGroupJob {
  const uint16_t* grad;
  const uint16_t* x;
  float* part;          // [nsplit][N][K] followed by the bias partials [nsplit][N]
  long long M;
  int ldg, lda, N, K, nk, rows, nsplit;
  int tile_begin;       // first tile of this job in the group's tile list; its tiles are ordered (n tile, k tile)
  float* out[3];        // direct mode: the gradient blocks themselves
  float* outb[3];
  int nper;             // rows of one gradient block (N / nw)
};
struct WGroupArgs {
  WGroupJob j[LTU_WGRAD_GROUP_MAX];
  int njobs;
  int tiles;            // tiles of all jobs
  int nsplit;           // row splits (the same for every job)
  int direct;           // nsplit == 1: every tile has one owner, which adds its sums to the gradient itself (no partials, no fold)
};
static_assert(sizeof(WGroupArgs) <= 4096, "kernel arguments");
template <int R>
__global__ void __launch_bounds__(256) wgrad_group_ring_bf16_kernel(const WGroupArgs ga) {
  extern __shared__ __attribute__((aligned(1024))) uint16_t smem[];
  // Workgroup -> (split, tile): all tiles of one row split get workgroup ids that are equal modulo 8, i.e. (with the dispatcher's
  // round-robin over the 8 XCDs) they share one XCD and its L2, and they are adjacent in dispatch order: the three column tiles
  // of the q,k,v gradient read the same rows of X, the two row tiles of the linear2 gradient the same rows of G.  Placement
  // affects speed only.  id = ((split / 8) * tiles + tile) * 8 + split % 8.
  // With fewer than 8 splits (many jobs: the layers of a small level flushed together) that mapping would launch mostly empty
  // workgroups; then id = tile * nsplit + split.
  int tg, split;
  if (ga.nsplit >= 8) {
    const int s_lo = (int)blockIdx.x & 7, q = (int)blockIdx.x >> 3;
    tg = q % ga.tiles; split = (q / ga.tiles) * 8 + s_lo;
    if (split >= ga.nsplit) return;
  } else {
    tg = (int)blockIdx.x / ga.nsplit; split = (int)blockIdx.x - tg * ga.nsplit;
  }
  int ji = 0;
  for (int t = 1; t < ga.njobs; ++t)
    if (tg >= ga.j[t].tile_begin) ji = t;
  const WGroupJob& jb = ga.j[ji];
  const int t = tg - jb.tile_begin;
  const int nb = t / jb.nk, kb = t - nb * jb.nk;
  const long long m_begin = (long long)split * jb.rows;
  long long m_end = m_begin + jb.rows;
  if (m_end > jb.M) m_end = jb.M;
  if (ga.direct) {
    const int seg = (nb * 128) / jb.nper;                   // a 128-row tile lies inside one block (nper % 128 == 0)
    float* o = seg == 0 ? jb.out[0] : (seg == 1 ? jb.out[1] : jb.out[2]);
    float* ob = seg == 0 ? jb.outb[0] : (seg == 1 ? jb.outb[1] : jb.outb[2]);
    // the tile function indexes rows of cat(dW): rebase the block's pointers by its first row
    o -= (long long)seg * jb.nper * jb.K;
    if (ob != nullptr) ob -= seg * jb.nper;
    wgrad_ring_tile<R>(jb.grad, jb.ldg, jb.x, jb.lda, m_begin, m_end, nb * 128, kb * 128, o, ob, jb.K, smem, true);
    return;
  }
  float* bpart = jb.part + (long long)jb.nsplit * jb.N * jb.K;
  wgrad_ring_tile<R>(jb.grad, jb.ldg, jb.x, jb.lda, m_begin, m_end, nb * 128, kb * 128, jb.part + (long long)split * jb.N * jb.K,
                     bpart + (long long)split * jb.N, jb.K, smem);
}

// ------------------------------------------------------------------------------------------------ NT (projection)
// Y[M][N] = A[M][K] W^T + bias for one column tile of BN = 64 TNW columns; the workgroup is PERSISTENT over row tiles.
//   * its W tile [BN][K] is fetched once and stays in LDS ("weight stationary"): the k-loop has no weight traffic at all;
//   * A streams through the ring in units of [128 rows][64 k] (16 KB), R - 1 units in flight, across tile boundaries;
//   * 128-byte LDS rows, chunk c of row r at slot c ^ ((r >> 1) & 7): conflict-free ds_read_b128 fragments;
//   * the product is formed transposed (D[n][m]) so a lane owns 4 consecutive n of one row: 8-byte staging writes into the
//     ring slot just consumed (wave-private quarter), 16-byte row-major reads, 16-byte global stores.
// vmcnt book-keeping: loads and stores retire in order, so the wait after an epilogue allows for its S stores as well
// (exact only when every store was issued, i.e. for full tiles; otherwise the stricter count is used).
template <int WM, int TNW, int R, bool GELU>
__global__ void __launch_bounds__(128 * WM) linear_ring_bf16_kernel(const IGemmArgs g, int tiles_m) {
  extern __shared__ __attribute__((aligned(1024))) uint16_t smem[];
  constexpr int NW = 2 * WM, TM = 4 / WM, BN = 64 * TNW, AUNIT = 128 * 64;
  constexpr int PW = 16 / NW;                               // LDS-DMA pieces of an A unit per wave
  constexpr int S = 2 * TM * TNW * (GELU ? 2 : 1);          // global stores per wave and tile
  const int KC = g.K >> 6;
  uint16_t* Wl = smem;                                      // [KC][BN][64]
  uint16_t* ring = smem + KC * BN * 64;                     // [R][128][64]
  float* bias_l = reinterpret_cast<float*>(ring + R * AUNIT);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int n_blk = blockIdx.y * BN;
  const int nper = g.N / g.nseg;
  DropCfg dc;                              // scalar loads only (lgkmcnt): no interference with the vmcnt book-keeping below
  if constexpr (GELU) dc = make_drop(g.drop_p, (uint64_t)g.drop_seed, reinterpret_cast<const uint64_t*>(g.drop_step));

  // The bias row travels by LDS-DMA like everything else: a plain load + ds_write + barrier here would put one full memory
  // round trip in front of the first asynchronous issue of every launch (~1 us of a 6 us fixed cost).
  for (int p = wave; p < BN / 64; p += NW) {
    const int n = n_blk + p * 64 + lane;
    const float* src = ring_zero_f32;
    if (n < g.N) {
      const int seg = n / nper;
      const float* bp = seg == 0 ? g.bias[0] : (seg == 1 ? g.bias[1] : g.bias[2]);
      if (bp != nullptr) src = bp + (n - seg * nper);
    }
    glds4(src, bias_l + p * 64);
  }

  // LDS-DMA piece = 8 rows x 128 B.  Slot = chunk ^ ((row >> 1) & 7): the 16 rows a quarter-wave reads with ds_read_b128 then
  // occupy 16 distinct 16-byte slots of the 256-byte bank row.  Every piece this wave fills has the parity of `wave`.
  const int srow = lane >> 3, lchunk = (lane & 7) ^ (((wave & 1) << 2) | (srow >> 1));
  {
    const int wpieces = KC * (BN / 8);
    for (int p = wave; p < wpieces; p += NW) {
      const int kc = p / (BN / 8), r8 = p - kc * (BN / 8);
      int n = n_blk + r8 * 8 + srow;
      if (n >= g.N) n = g.N - 1;
      const int seg = n / nper;
      const uint16_t* wp = reinterpret_cast<const uint16_t*>(seg == 0 ? g.w[0] : (seg == 1 ? g.w[1] : g.w[2]));
      glds16(wp + (long long)(n - seg * nper) * g.K + kc * 64 + lchunk * 8, Wl + (kc * BN + r8 * 8) * 64);
    }
  }

  const int ntile = (tiles_m - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int nunits = ntile * KC;
  const uint16_t* xbase = reinterpret_cast<const uint16_t*>(g.a0) + lchunk * 8;
  int it_t = blockIdx.x, it_kc = 0;
  auto issue = [&](int slot) {
    uint16_t* base = ring + slot * AUNIT + wave * 512;
#pragma unroll
    for (int s = 0; s < PW; ++s) {
      long long row = (long long)it_t * 128 + (wave + NW * s) * 8 + srow;
      if (row >= g.M) row = g.M - 1;
      glds16(xbase + row * g.lda0 + it_kc * 64, base + s * NW * 512);
    }
    if (++it_kc == KC) { it_kc = 0; it_t += gridDim.x; }
  };

  f32x16 acc[TM][TNW];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TNW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int sw = (li >> 1) & 7;
  const int a_off = (wm * 32 * TM + li) * 64, w_off = (wn * 32 * TNW + li) * 64;
  int kc = 0, tile = blockIdx.x;
  bool after_store = false;
  for (int u = 0; u < R - 1 && u < nunits; ++u) issue(u);
  // bias and weights are older than the R - 1 units just issued: wait for them only (all of it when fewer units exist)
  if (nunits >= R - 1) ring_sync<PW * (R - 1)>(); else ring_sync<0>();
  // the bias row stays in LDS and is read in the epilogue (4 x 16 bytes per column tile): as registers it cost 16 TNW VGPRs,
  // which kept the 128-column configuration above 128 registers, i.e. at one 8-wave workgroup per CU
  for (int q = 0; q < nunits; ++q) {
    if (q + R - 2 < nunits) {
      if (after_store) ring_sync<PW * (R - 2) + S>(); else ring_sync<PW * (R - 2)>();
    } else {
      ring_sync<0>();
    }
    after_store = false;
    if (q + R - 1 < nunits) issue((q + R - 1) % R);
    const uint16_t* As = ring + (q % R) * AUNIT + a_off;
    const uint16_t* Ws = Wl + kc * BN * 64 + w_off;
    // all fragment reads of the unit first, then its MFMAs (pinned: hipcc otherwise sinks each read next to its MFMA, which
    // then waits a full LDS round trip)
    bf16x8 af[4][TM], wf[4][TNW];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int co = ((ks * 2 + lh) ^ sw) << 3;
#pragma unroll
      for (int i = 0; i < TM; ++i) af[ks][i] = *reinterpret_cast<const bf16x8*>(As + i * 32 * 64 + co);
#pragma unroll
      for (int j = 0; j < TNW; ++j) wf[ks][j] = *reinterpret_cast<const bf16x8*>(Ws + j * 32 * 64 + co);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!(g.dbg & 2)) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TNW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (++kc < KC) continue;

    // ---- tile finished: stage 32 x 32 sub-tiles through this unit's slot (every wave is done with it after the barrier)
    asm volatile("s_barrier" ::: "memory");
    uint16_t* stg = ring + (q % R) * AUNIT + wave * 1024;    // wave-private [32][32] bf16, 4 chunks per row
    const bool full = (long long)(tile + 1) * 128 <= g.M && n_blk + BN <= g.N && !(g.dbg & 1);
    const long long Meff = (g.dbg & 1) ? 0 : g.M;          // ablation: no global stores
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TNW; ++j) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const float4 bv = *reinterpret_cast<const float4*>(&bias_l[wn * 32 * TNW + j * 32 + 8 * rr + 4 * lh]);
          uint2 pk;
          pk.x = pack_bf16x2(acc[i][j][4 * rr + 0] + bv.x, acc[i][j][4 * rr + 1] + bv.y);
          pk.y = pack_bf16x2(acc[i][j][4 * rr + 2] + bv.z, acc[i][j][4 * rr + 3] + bv.w);
          *reinterpret_cast<uint2*>(stg + li * 32 + ((rr ^ (li & 3)) << 3) + 4 * lh) = pk;
          acc[i][j][4 * rr + 0] = 0.f; acc[i][j][4 * rr + 1] = 0.f; acc[i][j][4 * rr + 2] = 0.f; acc[i][j][4 * rr + 3] = 0.f;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int idx = lane + 64 * s;
          const int row = idx >> 2, ch = idx & 3;
          const uint4 v = *reinterpret_cast<const uint4*>(stg + row * 32 + ((ch ^ (row & 3)) << 3));
          const long long grow = (long long)tile * 128 + wm * 32 * TM + i * 32 + row;
          const int gcol = n_blk + wn * 32 * TNW + j * 32 + ch * 8;
          uint16_t* dst = reinterpret_cast<uint16_t*>(g.o0) + grow * g.ldo0 + gcol;
          if (full) *reinterpret_cast<uint4*>(dst) = v;
          else if (grow < Meff && gcol < g.N) *reinterpret_cast<uint4*>(dst) = v;
          if constexpr (GELU) {            // h = dropout(gelu(u)) on the bf16-rounded u, as the stand-alone kernel computes it
            const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
            float f[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              f[2 * e] = gelu_erf(__uint_as_float(w4[e] << 16));
              f[2 * e + 1] = gelu_erf(__uint_as_float(w4[e] & 0xffff0000u));
            }
            const unsigned long long g4 = (unsigned long long)(grow * g.N + gcol) >> 2;
            const float4 ha = drop4(dc, g4, make_float4(f[0], f[1], f[2], f[3]));
            const float4 hb = drop4(dc, g4 + 1, make_float4(f[4], f[5], f[6], f[7]));
            const uint4 hv = make_uint4(pack_bf16x2(ha.x, ha.y), pack_bf16x2(ha.z, ha.w), pack_bf16x2(hb.x, hb.y), pack_bf16x2(hb.z, hb.w));
            uint16_t* hd = reinterpret_cast<uint16_t*>(g.gelu_out) + grow * g.N + gcol;
            if (full) *reinterpret_cast<uint4*>(hd) = hv;
            else if (grow < Meff && gcol < g.N) *reinterpret_cast<uint4*>(hd) = hv;
          }
        }
      }
    kc = 0;
    tile += gridDim.x;
    after_store = full;
  }
}

template <int WM, int TNW, int R, bool GELU>
static int launch_lin_ring_e(const IGemmArgs& g, hipStream_t st) {
  const int BN = 64 * TNW;
  const int tiles_m = (int)((g.M + 127) / 128), nt = cdiv(g.N, BN);
  int P = 256 / nt;
  if (P < 1) P = 1;
  if (P >= 8) P &= ~7;                       // column tiles of one row tile then share an XCD (ids differ by P)
  if (P > tiles_m) {                         // fewer row tiles than workgroups: one tile each (rounding DOWN to a multiple of 8
    const int up = (tiles_m + 7) & ~7;       // would leave a few tiles for a second round), a few idle workgroups instead
    P = up <= P ? up : tiles_m;
  }
  const int smem_bytes = BN * g.K * 2 + R * 16384 + BN * 4;
  static LtuDevOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_ring_bf16_kernel<WM, TNW, R, GELU>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  hipLaunchKernelGGL((linear_ring_bf16_kernel<WM, TNW, R, GELU>), dim3(P, nt), dim3(128 * WM), smem_bytes, st, g, tiles_m);
  return ltu_check_launch();
}
template <int WM, int TNW, int R>
static int launch_lin_ring(const IGemmArgs& g, hipStream_t st) {
  if (g.gelu_out != nullptr) return launch_lin_ring_e<WM, TNW, R, true>(g, st);
  return launch_lin_ring_e<WM, TNW, R, false>(g, st);
}

static bool ring_enabled();
// dense projection through the weight-stationary ring kernel; returns 1 when the shape is not handled
int launch_nt_ring_bf16(const IGemmArgs& g_in, hipStream_t st) {
  const IGemmArgs& g = g_in;
  int on = -1, w8 = -1;
  on = ltu_knob("LTU_NO_NT_RING", 0) ? 0 : 1;
  w8 = ltu_knob("LTU_NT_RING_WAVES", 8) == 4 ? 0 : 1;
  if (!on || !ring_enabled()) return 1;
  if (g.ntaps != 1 || g.K != g.C || g.c0 != g.C || !g.out_identity || g.accum || g.n0 != g.N || g.dbg) return 1;
  int rdbg = -1;                      // ablations (tools/bench_nt.py): 1 = no global stores, 2 = no MFMA
  rdbg = ltu_knob("LTU_RING_DBG", 0);
  IGemmArgs gd = g_in;
  gd.dbg = rdbg;
  const IGemmArgs& g2 = gd;
  if (g.gelu_out != nullptr && (g.nseg != 1 || g.ldo0 != g.N || ((uintptr_t)g.gelu_out & 15))) return 1;
  if (g.K % 64 || g.K > 768 || g.N % 8 || g.N < 96 || g.M < 512 || g.lda0 % 8 || g.ldo0 % 8 || g.N % g.nseg) return 1;
  uintptr_t al = (uintptr_t)g.a0 | (uintptr_t)g.o0;
  for (int i = 0; i < g.nseg; ++i) al |= (uintptr_t)g.w[i];
  if (al & 15) return 1;
  int tnw1_below = -1;
  tnw1_below = ltu_knob("LTU_NT_RING_TNW1_BELOW", 2048);     // few row tiles: narrower column tiles spread them over more workgroups
  if (w8) {
    if (g.K <= 256 && g.M < tnw1_below) return launch_lin_ring<4, 1, 4>(g2, st);
    if (g.gelu_out != nullptr && g.K <= 256) {
      // the GELU epilogue is a long VALU phase during which nothing is issued: a deeper ring keeps loads in flight across it
      int deep = -1;
      deep = ltu_knob("LTU_GELU_RING_DEEP", 1);
      if (deep && g.K <= 128) return launch_lin_ring_e<4, 2, 6, true>(g2, st);
      if (deep) return launch_lin_ring_e<4, 2, 5, true>(g2, st);
    }
    if (g.K <= 256) return launch_lin_ring<4, 2, 4>(g2, st);
    if (g.K <= 384) return launch_lin_ring<4, 2, 3>(g2, st);
    if (g.K <= 512) return launch_lin_ring<4, 1, 4>(g2, st);
    return launch_lin_ring<4, 1, 3>(g2, st);
  }
  if (g.K <= 256) return launch_lin_ring<2, 2, 4>(g2, st);
  if (g.K <= 384) return launch_lin_ring<2, 2, 3>(g2, st);
  if (g.K <= 512) return launch_lin_ring<2, 1, 4>(g2, st);
  return launch_lin_ring<2, 1, 3>(g2, st);
}

#define TN_RING 6
static int ring_blocks() {
  int v = -1;
  v = ltu_knob_pos("LTU_RING_BLOCKS", 256);
  return v;
}
static bool ring_enabled() {
  int v = -1;
  v = ltu_knob("LTU_NO_RING", 0) ? 0 : 1;
  return v == 1;
}

struct RingGeom { int nk, nn, rows, nsplit; long long ws_floats; };
static RingGeom tn_ring_geometry(long long M, int N, int K) {
  RingGeom t;
  t.nk = K / 128; t.nn = N / 128;
  long long want = ring_blocks() / ((long long)t.nk * t.nn);
  if (want < 1) want = 1;
  long long rows = (M + want - 1) / want;
  int minrows = -1;
  minrows = ltu_knob("LTU_RING_MINROWS", 0) >= 32 ? ltu_knob("LTU_RING_MINROWS", 0) : 0;
  // few rows: shorter splits put more workgroups on the serial unit loop (measured: 1 024 rows 14.8 -> 9.0 us at 128 rows per
  // split, 8 640 rows 17.7 -> 16.5 us at 256; below that the second stage grows faster than the first shrinks)
  const int mr = minrows > 0 ? minrows : (M <= 2048 ? 128 : 256);
  if (rows < mr) rows = mr;
  rows = (rows + 31) / 32 * 32;
  t.rows = (int)rows;
  t.nsplit = (int)((M + rows - 1) / rows);
  t.ws_floats = (long long)t.nsplit * N * ((long long)K + 1);
  return t;
}
bool tn_ring_shape_ok(long long M, int N, int K) { return ring_enabled() && M % 32 == 0 && M >= 1024 && N % 128 == 0 && K % 128 == 0; }
long long tn_ring_ws_floats(long long M, int N, int K) { return tn_ring_shape_ok(M, N, K) ? tn_ring_geometry(M, N, K).ws_floats : 0; }

// dense weight gradient through the ring kernel + wgrad_reduce_kernel; returns 1 when the shape is not handled.
// With `defer` set the second stage is left to the caller (ltu_reduce_batch), *nsplit_out tells it how many splits to fold.
int launch_tn_ring_bf16(WGradArgs& wa, hipStream_t st, int* nsplit_out) {
  const IGemmArgs& g = wa.g;
  if (wa.part == nullptr || g.ntaps != 1 || g.K != g.C || g.c0 != g.C || !g.out_identity) return 1;
  if (!tn_ring_shape_ok(g.M, g.N, g.K) || g.lda0 % 8 || wa.ldg % 8) return 1;
  if (((uintptr_t)g.a0 | (uintptr_t)wa.grad) & 15) return 1;
  const RingGeom t = tn_ring_geometry(g.M, g.N, g.K);
  if (t.ws_floats > wa.part_floats) return LTU_E_ARG;
  wa.rows_per_split = t.rows;
  wa.npad = g.N; wa.kpad = g.K;
  wa.bpart = wa.part + (long long)t.nsplit * wa.npad * wa.kpad;
  constexpr int smem_bytes = TN_RING * 2 * 32 * 128 * 2;
  static LtuDevOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_ring_bf16_kernel<TN_RING>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              smem_bytes);
  }
  hipLaunchKernelGGL((wgrad_ring_bf16_kernel<TN_RING>), dim3(t.nk, t.nn, t.nsplit), dim3(256), smem_bytes, st, wa);
  int rc = ltu_check_launch();
  if (rc) return rc;
  if (nsplit_out != nullptr) { *nsplit_out = t.nsplit; return LTU_OK; }
  return launch_wgrad_reduce(wa, t.nsplit, st);
}

// ---- grouped weight gradients ---------------------------------------------------------------------------------------------
// Fold of the group's partial tiles: dW[seg][n][k] += sum_split part[split][n][k], db likewise.  A workgroup owns 64 float4 outputs;
// its 4 thread groups each sum a quarter of the splits with 16-byte loads (all of a thread's loads are independent: several in
// flight), combine through LDS, and group 0 adds the result to the gradient.  The partials were written just before and are
// mostly still in L2 / MALL.
struct WFoldJob {
  const float* part;
  float* out[3];
  float* outb[3];
  int nsplit, N, K, nseg;
  int blk_begin, wblocks;      // workgroups [blk_begin, blk_begin + wblocks) fold weights, the following ones the bias rows
};
struct WFoldArgs {
  WFoldJob j[LTU_WGRAD_GROUP_MAX];
  int njobs;
};
__global__ void __launch_bounds__(256) wgroup_fold_kernel(const WFoldArgs fa) {
  __shared__ float4 red[3][64];
  int ji = 0;
  for (int t = 1; t < fa.njobs; ++t)
    if ((int)blockIdx.x >= fa.j[t].blk_begin) ji = t;
  const WFoldJob& jb = fa.j[ji];
  const int local = (int)blockIdx.x - jb.blk_begin;
  const int nper = jb.N / jb.nseg;
  const long long nk = (long long)jb.N * jb.K;
  if (local < jb.wblocks) {
    const int lane64 = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const long long e = ((long long)local * 64 + lane64) * 4;          // first of 4 consecutive k of one row n (K % 128 == 0)
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), cur = acc;
    float* o = nullptr;
    if (e < nk) {
      if (grp == 0) {                    // the old gradient values: requested with the partials, not behind them
        const int n = (int)(e / jb.K), k = (int)(e - (long long)n * jb.K);
        const int seg = n / nper;
        o = (seg == 0 ? jb.out[0] : (seg == 1 ? jb.out[1] : jb.out[2])) + (long long)(n - seg * nper) * jb.K + k;
        cur = *reinterpret_cast<const float4*>(o);
      }
      const float* p = jb.part + e;
      for (int s0 = grp; s0 < jb.nsplit; s0 += 32) {       // 8 splits per thread requested together (one trip up to 32 splits)
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int sp = s0 + 4 * k;
          v[k] = *reinterpret_cast<const float4*>(p + (long long)(sp < jb.nsplit ? sp : s0) * nk);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (s0 + 4 * k < jb.nsplit) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
      }
    }
    if (grp > 0) red[grp - 1][lane64] = acc;
    __syncthreads();
    if (grp == 0 && e < nk) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const float4 v = red[q][lane64];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      cur.x += acc.x; cur.y += acc.y; cur.z += acc.z; cur.w += acc.w;
      *reinterpret_cast<float4*>(o) = cur;
    }
  } else {
    // bias rows: 64 outputs per workgroup, the 4 thread groups share the splits (no serial walk over all of them)
    const int lane64 = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int n = (local - jb.wblocks) * 64 + lane64;
    float a = 0.f;
    if (n < jb.N) {
      const float* bp = jb.part + (long long)jb.nsplit * nk + n;
#pragma unroll 8
      for (int sp = grp; sp < jb.nsplit; sp += 4) a += bp[(long long)sp * jb.N];
    }
    float* redf = reinterpret_cast<float*>(red);
    if (grp > 0) redf[(grp - 1) * 64 + lane64] = a;
    __syncthreads();
    if (grp == 0 && n < jb.N) {
      a += redf[lane64] + redf[64 + lane64] + redf[128 + lane64];
      const int seg = n / nper;
      float* o = seg == 0 ? jb.outb[0] : (seg == 1 ? jb.outb[1] : jb.outb[2]);
      if (o != nullptr) o[n - seg * nper] += a;
    }
  }
}

static void wgroup_geometry(const ltu_wgrad_job* jobs, int njobs, int blocks, WGroupArgs& ga, long long* part_off, long long* ws_floats) {
  int tiles = 0;
  for (int i = 0; i < njobs; ++i) tiles += (jobs[i].N / 128) * (jobs[i].K / 128);
  const int budget = blocks > 0 ? blocks : ltu_knob_pos("LTU_WGROUP_BLOCKS", 256);
  long long off = 0, Mmax = 0;
  for (int i = 0; i < njobs; ++i) Mmax = jobs[i].M > Mmax ? jobs[i].M : Mmax;
  // one split count for the whole group (the jobs of a transformer layer have the same M): ~budget workgroups, splits of >= 128 rows
  long long want = budget / (tiles > 0 ? tiles : 1);
  if (want < 1) want = 1;
  long long rows0 = (Mmax + want - 1) / want;
  if (rows0 < 128) rows0 = 128;
  rows0 = (rows0 + 31) / 32 * 32;
  const int nsplit = (int)((Mmax + rows0 - 1) / rows0);
  int tb = 0;
  ga.njobs = njobs;
  ga.tiles = tiles;
  ga.nsplit = nsplit;
  ga.direct = nsplit == 1 && !ltu_knob("LTU_WGROUP_NO_DIRECT", 0);
  for (int i = 0; i < njobs && ga.direct; ++i)
    if ((jobs[i].N / jobs[i].nw) % 128) ga.direct = 0;
  for (int i = 0; i < njobs; ++i) {
    WGroupJob& j = ga.j[i];
    const long long M = jobs[i].M;
    long long rows = ((M + nsplit - 1) / nsplit + 31) / 32 * 32;
    j.M = M; j.N = jobs[i].N; j.K = jobs[i].K; j.nk = jobs[i].K / 128;
    j.ldg = jobs[i].ldg; j.lda = jobs[i].lda;
    j.rows = (int)rows;
    j.nsplit = nsplit;                       // splits past the job's rows store zero tiles
    j.tile_begin = tb;
    tb += (j.N / 128) * j.nk;
    part_off[i] = off;
    off += (long long)j.nsplit * j.N * ((long long)j.K + 1);
  }
  if (ga.direct) off = 4;                    // no partial tiles (a non-zero size keeps "0 = not handled" of the C-ABI)
  *ws_floats = off;
}
static bool wgroup_ok(const ltu_wgrad_job* jobs, int njobs) {
  if (njobs < 1 || njobs > LTU_WGRAD_GROUP_MAX || !ring_enabled() || ltu_knob("LTU_NO_WGROUP", 0)) return false;
  for (int i = 0; i < njobs; ++i) {
    const ltu_wgrad_job& j = jobs[i];
    if (!tn_ring_shape_ok(j.M, j.N, j.K) || j.lda % 8 || j.ldg % 8 || j.nw < 1 || j.nw > 3 || j.N % j.nw) return false;
    if (((uintptr_t)j.a | (uintptr_t)j.grad) & 15) return false;
  }
  return true;
}
static bool wfat_enabled();
long long wgrad_fat_group_ws_floats(const ltu_wgrad_job* jobs, int njobs, int blocks);
int launch_wgrad_fat_group_bf16(const ltu_wgrad_job* jobs, int njobs, int blocks, float* ws, long long ws_floats, hipStream_t st);
// blocks: the workgroup budget of the launch (<= 0: LTU_WGROUP_BLOCKS / 256).  The size query and the launch take the SAME argument,
// and the launch is told the capacity of `ws`: a geometry that needs more returns LTU_E_ARG instead of writing past the end.
long long tn_ring_group_ws_floats(const ltu_wgrad_job* jobs, int njobs, int blocks) {
  if (wfat_enabled()) {
    const long long n = wgrad_fat_group_ws_floats(jobs, njobs, blocks);
    if (n > 0) return n;
  }
  if (!wgroup_ok(jobs, njobs)) return 0;
  WGroupArgs ga;
  long long n = 0, off[LTU_WGRAD_GROUP_MAX];
  wgroup_geometry(jobs, njobs, blocks, ga, off, &n);
  return n;
}
// launches the grouped kernel and its fold; returns 1 when the group is not handled
int launch_tn_ring_group_bf16(const ltu_wgrad_job* jobs, int njobs, int blocks, float* ws, long long ws_floats, hipStream_t st) {
  if (wfat_enabled()) {
    const int rc = launch_wgrad_fat_group_bf16(jobs, njobs, blocks, ws, ws_floats, st);
    if (rc != 1) return rc;
  }
  if (!wgroup_ok(jobs, njobs) || ws == nullptr) return 1;
  WGroupArgs ga;
  memset(&ga, 0, sizeof(ga));
  long long n = 0, off[LTU_WGRAD_GROUP_MAX];
  wgroup_geometry(jobs, njobs, blocks, ga, off, &n);
  if (n > ws_floats) return LTU_E_ARG;
  int nblocks = 0, fblocks = 0;
  WFoldArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.njobs = njobs;
  for (int i = 0; i < njobs; ++i) {
    WGroupJob& j = ga.j[i];
    j.part = ws + off[i];
    j.grad = reinterpret_cast<const uint16_t*>(jobs[i].grad);
    j.x = reinterpret_cast<const uint16_t*>(jobs[i].a);
    j.nper = jobs[i].N / jobs[i].nw;
    for (int s = 0; s < 3; ++s) { j.out[s] = s < jobs[i].nw ? jobs[i].dw[s] : nullptr; j.outb[s] = s < jobs[i].nw ? jobs[i].db[s] : nullptr; }
    WFoldJob& f = fa.j[i];
    f.part = j.part; f.nsplit = j.nsplit; f.N = j.N; f.K = j.K; f.nseg = jobs[i].nw;
    for (int s = 0; s < 3; ++s) { f.out[s] = s < jobs[i].nw ? jobs[i].dw[s] : nullptr; f.outb[s] = s < jobs[i].nw ? jobs[i].db[s] : nullptr; }
    f.blk_begin = fblocks;
    f.wblocks = (int)(((long long)j.N * j.K / 4 + 63) / 64);
    fblocks += f.wblocks + (j.N + 63) / 64;
  }
  constexpr int smem_bytes = TN_RING * 2 * 32 * 128 * 2;
  static LtuDevOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_group_ring_bf16_kernel<TN_RING>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              smem_bytes);
  }
  nblocks = ga.nsplit >= 8 ? ((ga.nsplit + 7) / 8) * ga.tiles * 8 : ga.nsplit * ga.tiles;
  hipLaunchKernelGGL((wgrad_group_ring_bf16_kernel<TN_RING>), dim3(nblocks), dim3(256), smem_bytes, st, ga);
  if (!ga.direct) hipLaunchKernelGGL(wgroup_fold_kernel, dim3(fblocks), dim3(256), 0, st, fa);
  return ltu_check_launch();
}

#ifdef LTU_EXPERIMENTS
// ------------------------------------------------------------------------------------------------ grouped weight gradients, round 5
// EXPERIMENTS BUILD ONLY (make EXPERIMENTS=1, knob LTU_WGROUP_FAT=1): built for the round-4 verdict (one workgroup owns all column tiles
// of a job's row split), meets its micro-benchmark targets (d = 128 level at 128 workgroups: 97-101 us against 106 us, 3.6 TB/s of
// operands read once) and LOSES in the step: 12.73-12.80 ms against 12.42-12.50 ms with the kernel above, at every side width and ring
// size tried, also as the 256-register variant below (2 waves per SIMD possible, 96 KB ring).  rocprofv3 of the step: the kernel
// itself is 0.24 ms shorter per step, the main-chain kernels that run beside it lose 0.4 ms (halo / class convs, InstanceNorm
// backward, attention partials: +20-40 % each).  Not register or LDS occupancy (the polite variant loses the same) and not bytes in
// flight (64 / 96 / 144 KB rings alike): what differs is that the pipelined kernel keeps the matrix pipe and the LDS-DMA path busy
// where the kernel above idles in exposed latencies - beside a latency-bound main chain a denser side kernel is a worse neighbour
// (profiles/r05_wgroup.txt).
// "Fat tiles": ONE workgroup owns all column tiles of a job's row split that its accumulators can hold, so an operand row is read
// once - e.g. the 384-column q|k|v gradient of a d = 128 layer is three accumulator sets over ONE pass of X (the 128 x 128 tiles of
// the kernel above read X three times and, at half the machine's width beside the main chain, those re-reads did reach HBM: 1.48x
// the algorithmic bytes in profiles/r04_pmc_step.json).  Tile types (TN x TK outputs, <= 49 152 fp32 accumulators = 192
// registers per lane at one wave per SIMD): 384 x 128, 256 x 128, 128 x 256, 128 x 128; a job is cut into the type that reads the fewest
// bytes per row.  At d = 128 every projection of a layer is exactly one tile per row split (q|k|v 384 x 128, out 128 x 128, linear1
// 256 x 128, linear2 128 x 256): operands are read exactly once.  Each job gets its own split count, proportional to the bytes
// one of its tiles streams per row, so the workgroups of a launch finish together.
// The ring is sized by BYTES IN FLIGHT, not by rows: these kernels run one workgroup per CU on half of the CUs (beside the main
// chain), so what a CU keeps in flight x the CUs sets the rate (Little's law at 2-3 us of loaded HBM latency): units of 16 rows
// (8-16 KB) in a ring of 144 KB, all but two units in flight (the kernel above: 80 KB).
// LDS image of a unit: column blocks [TNB of G | TKB of X] of [ROWS][128] bf16 each, chunk c of row r at slot c ^ ((r & 3) << 2) as
// above (conflict-free transposing reads).  4 waves as 2 (n) x 2 (k).
struct WFatJob {
  const uint16_t* grad;
  const uint16_t* x;
  float* part;          // [nsplit][N][K] followed by the bias partials [nsplit][N] (unused when direct)
  float* out[3];        // the gradient blocks (direct mode)
  float* outb[3];
  long long M;
  int ldg, lda, N, K, nper, rows, wg_begin;
  short type, nt, kt, nsplit, direct, pad;
};
struct WFatArgs {
  WFatJob j[LTU_WGRAD_GROUP_MAX];
  int njobs;
};
static_assert(sizeof(WFatArgs) <= 4096, "kernel arguments");
#ifndef WFAT_RING_KB
#ifdef WFAT_TYPE0
#define WFAT_RING_KB 144
#else
#define WFAT_RING_KB 96
#endif
#endif
#define WFAT_RING_BYTES (WFAT_RING_KB * 1024)
#define WFAT_R(unit_kb) (WFAT_RING_KB / (unit_kb))

struct WFatView {     // one job as scalars (built field by field from the kernel arguments: no struct copy in private memory)
  const uint16_t* grad;
  const uint16_t* x;
  float* part;
  float* out0; float* out1; float* out2;
  float* outb0; float* outb1; float* outb2;
  int ldg, lda, N, K, nper, nsplit, direct;
};
// retire all but the youngest N LDS-DMA of this wave AND every LDS read it has issued, then meet the other waves
template <int N>
__device__ __forceinline__ void ring_sync_all() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}
// column sums of a bf16 pair, accumulated in fp32: c += lo + hi (v_dot2c_f32_bf16 against (1, 1); the builtin does not select in
// hipcc 7.2).  Not volatile: the scheduler places it in the shadow of the MFMAs.
__device__ __forceinline__ void bf16_pair_sum(float& c, uint32_t w) {
  const uint32_t ones = 0x3f803f80u;
  asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(c) : "v"(w), "v"(ones));
}

// Units of 16 rows (one MFMA k-step).  The loop is software-pipelined by hand: while the MFMAs of unit `it` run on fragments that
// are already in registers, the fragments of unit it + 1 are read from LDS into a second register set and the LDS-DMA of unit
// it + R - 1 is issued piece by piece BETWEEN the MFMAs (ablation of the first version, profiles/r05_wgroup_ablation.txt: LDS-DMA
// alone 53 us, fragment reads + MFMAs alone 55 us, together 79 us - one wave per SIMD issues in order, so whatever is not placed
// in the shadow of an MFMA is serial time).
template <int TNB, int TKB, int R, int DBG = 0>
__device__ __forceinline__ void wgrad_fat_tile(const WFatView jb, long long m_begin, long long m_end, int n0, int k0, int split,
                                               bool do_bias, uint16_t* smem) {
  constexpr int NB = TNB + TKB;
  constexpr int UNIT = 16 * NB * 128;                         // elements
  constexpr int DPW = NB;                                     // LDS-DMA pieces per wave and unit (one per column block)
  constexpr int NI = 2 * TNB, NJ = 2 * TKB;                   // 32-column groups per wave
  constexpr int NMF = NI * NJ;
  static_assert(R * UNIT * 2 <= WFAT_RING_BYTES, "ring");
  static_assert(DPW * (R - 3) <= 63 && R >= 4, "vmcnt");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int niter = m_end > m_begin ? (int)((m_end - m_begin) >> 4) : 0;

  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum[TNB];
#pragma unroll
  for (int q = 0; q < TNB; ++q) bsum[q] = 0.f;

  // LDS-DMA sources: wave w fills rows 4w .. 4w + 3 of every column block
  const int prow = 4 * wave + (lane >> 4);
  const int lchunk = (lane & 15) ^ ((lane >> 4) << 2);
  const uint16_t* gsrc = jb.grad + (m_begin + prow) * jb.ldg + n0 + lchunk * 8;
  const uint16_t* xsrc = jb.x + (m_begin + prow) * jb.lda + k0 + lchunk * 8;
  const long long gstep = 16LL * jb.ldg, xstep = 16LL * jb.lda;
  auto piece = [&](int slot, int b) {                         // column block b of the unit that goes to ring slot `slot`
    if (DBG & 2) return;
    uint16_t* base = smem + slot * UNIT + wave * 512 + b * 2048;
    if (b < TNB) glds16(gsrc + b * 128, base); else glds16(xsrc + (b - TNB) * 128, base);
  };
  auto advance = [&]() { gsrc += gstep; xsrc += xstep; };

  // transposing-read geometry (as in wgrad_ring_tile): this lane addresses row trow (and trow + 4) of the 16-row slab, 4 columns
  // at tcol of a 32-column group; group g of a block sits at chunk slots ((g ^ tq) * 4 ..) of the row
  const int gq = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  const int tcol = 16 * (gq & 1) + 4 * tp;
  const int trow = 8 * (gq >> 1) + tq;
  int aoff[NI], boff[NJ];            // element offsets of this lane's fragments inside a unit
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int gg = wm * NI + i;
    aoff[i] = (gg >> 2) * 2048 + trow * 128 + (((((gg & 3) * 4 + (tcol >> 3)) ^ (tq << 2)) << 3) + (tcol & 7));
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int gg = wn * NJ + j;
    boff[j] = (TNB + (gg >> 2)) * 2048 + trow * 128 + (((((gg & 3) * 4 + (tcol >> 3)) ^ (tq << 2)) << 3) + (tcol & 7));
  }
  auto frags = [&](bf16x8 (&a)[NI], bf16x8 (&b)[NJ], int slot) {
    if (DBG & 1) return;
    const uint16_t* U = smem + slot * UNIT;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      union { struct { rs16x4 l, h; } s; bf16x8 v; } u;
      u.s.l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_rs16x4*)(U + aoff[i]));
      u.s.h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_rs16x4*)(U + aoff[i] + 4 * 128));
      a[i] = u.v;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      union { struct { rs16x4 l, h; } s; bf16x8 v; } u;
      u.s.l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_rs16x4*)(U + boff[j]));
      u.s.h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_rs16x4*)(U + boff[j] + 4 * 128));
      b[j] = u.v;
    }
  };
  // the NMF MFMAs of a unit with the DPW pieces of a later unit issued between them (ISSUE: compile time, two code versions)
  auto mfmas = [&](const bf16x8 (&a)[NI], const bf16x8 (&b)[NJ], auto issue_tag, int slot) {
    constexpr bool ISSUE = decltype(issue_tag)::value;
    if (DBG & 1) {
      if (ISSUE) {
#pragma unroll
        for (int p = 0; p < DPW; ++p) piece(slot, p);
      }
      return;
    }
    constexpr int PER = NMF / DPW;                            // MFMAs per piece
    // bias gradient = column sums of G: the A fragment holds 8 rows of column n = lane & 31.  The two waves that share the A
    // fragments (wn = 0, 1) each sum every second one (selected dword by dword: a select between two 16-byte vectors compiles to an
    // indexed load from private memory); the sums of fragment pair q sit in the q-th group of MFMAs
    auto bias = [&](int q) {
      union { bf16x8 v; uint32_t w[4]; } u0, u1;
      u0.v = a[2 * q]; u1.v = a[2 * q + 1];
#pragma unroll
      for (int e = 0; e < 4; ++e) bf16_pair_sum(bsum[q], wn ? u1.w[e] : u0.w[e]);
    };
#pragma unroll
    for (int m = 0; m < NMF; ++m) {
      const int i = m / NJ, j = m % NJ;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      if (m % PER == PER - 1) {
        if (m / PER < TNB) bias(m / PER);
        if (ISSUE) {
          __builtin_amdgcn_sched_barrier(0);
          piece(slot, m / PER);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    static_assert(TNB <= DPW, "one bias group per piece");
  };

  bf16x8 a0[NI], b0[NJ], a1[NI], b1[NJ];
  for (int u = 0; u < R - 1 && u < niter; ++u) {
#pragma unroll
    for (int p = 0; p < DPW; ++p) piece(u, p);
    advance();
  }
  // steady state: iterations it < n_steady issue the LDS-DMA of unit it + R - 1 (into the slot of unit it - 1); two per trip so
  // that the two fragment register sets are compile-time names; no test inside
  const int n_steady = niter - (R - 1);
  int it = 0, slot_c = 0;                                    // ring slot of unit `it`
  auto nxt = [&](int sl) { return sl + 1 == R ? 0 : sl + 1; };
  auto prv = [&](int sl) { return sl == 0 ? R - 1 : sl - 1; };
  if (n_steady >= 2) {
    ring_sync_all<DPW * (R - 2)>();                           // unit 0 has landed
    frags(a0, b0, 0);
    for (; it + 1 < n_steady; it += 2) {
      // unit it + 1 has landed (this wave's pieces: vmcnt; the other waves': the barrier) and every wave holds the fragments of
      // unit `it` in registers (lgkmcnt(0) before the barrier): the slots of all units <= it may be overwritten
      ring_sync_all<DPW * (R - 3)>();
      frags(a1, b1, nxt(slot_c));
      __builtin_amdgcn_sched_barrier(0);
      mfmas(a0, b0, std::true_type{}, prv(slot_c));
      advance();
      slot_c = nxt(slot_c);
      ring_sync_all<DPW * (R - 3)>();
      frags(a0, b0, nxt(slot_c));
      __builtin_amdgcn_sched_barrier(0);
      mfmas(a1, b1, std::true_type{}, prv(slot_c));
      advance();
      slot_c = nxt(slot_c);
    }
  }
  // tail (at most R iterations): everything that is in flight is waited for, fragments are read where they are used
  for (; it < niter; ++it) {
    ring_sync_all<0>();
    frags(a0, b0, slot_c);
    if (it + R - 1 < niter) {
#pragma unroll
      for (int p = 0; p < DPW; ++p) piece(prv(slot_c), p);
      advance();
    }
    mfmas(a0, b0, std::false_type{}, 0);
    slot_c = nxt(slot_c);
  }

  const int li = lane & 31, lh = lane >> 5;
  if (do_bias) {
#pragma unroll
    for (int q = 0; q < TNB; ++q) {
      const float t = xhalf_combine<LtuAdd>(bsum[q]);
      const int n = n0 + (wm * NI + 2 * q + wn) * 32 + li;
      if (lh == 0) {
        if (jb.direct) {
          const int seg = n / jb.nper;
          float* ob = seg == 0 ? jb.outb0 : (seg == 1 ? jb.outb1 : jb.outb2);
          if (ob != nullptr) ob[n - seg * jb.nper] += t;
        } else {
          jb.part[(long long)jb.nsplit * jb.N * jb.K + (long long)split * jb.N + n] = t;
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int nb = n0 + (wm * NI + i) * 32;                   // first of this group's 32 rows of cat(dW): inside one gradient block
    float* rowbase;
    if (jb.direct) {
      const int seg = nb / jb.nper;
      rowbase = (seg == 0 ? jb.out0 : (seg == 1 ? jb.out1 : jb.out2)) + (long long)(nb - seg * jb.nper) * jb.K;
    } else {
      rowbase = jb.part + ((long long)split * jb.N + nb) * jb.K;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      float* p = rowbase + k0 + (wn * NJ + j) * 32 + li + (long long)(4 * lh) * jb.K;
      if (jb.direct) {
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = p[(long long)((r & 3) + 8 * (r >> 2)) * jb.K];
#pragma unroll
        for (int r = 0; r < 16; ++r) p[(long long)((r & 3) + 8 * (r >> 2)) * jb.K] = old[r] + acc[i][j][r];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) p[(long long)((r & 3) + 8 * (r >> 2)) * jb.K] = acc[i][j][r];
      }
    }
  }
}

struct WFatType { int tnb, tkb; };
static const WFatType WFAT_TYPES[4] = {{3, 1}, {2, 1}, {1, 2}, {1, 1}};

#ifdef WFAT_TYPE0
#define WFAT_WAVES 1
#else
#define WFAT_WAVES 2          // <= 256 registers: a second wave per SIMD (of another kernel) fits beside this one
#endif
template <int DBG>
__global__ void __launch_bounds__(256, WFAT_WAVES) wgrad_fat_group_bf16_kernel(const WFatArgs fa) {
  extern __shared__ __attribute__((aligned(1024))) uint16_t smem[];
  int ji = 0;
  for (int t = 1; t < fa.njobs; ++t)
    if ((int)blockIdx.x >= fa.j[t].wg_begin) ji = t;
  const WFatJob& J = fa.j[ji];
  const int local = (int)blockIdx.x - J.wg_begin;
  const int nt = J.nt, kt = J.kt, nsplit = J.nsplit, rows = J.rows, type = J.type;
  const long long M = J.M;
  const int tiles = nt * kt;
  // all tiles of one row split get workgroup ids that are equal modulo 8 (one XCD under round-robin placement: the tiles of a
  // multi-tile job read the same rows); wg_begin is a multiple of 8.  Placement affects speed only.
  int tile, split;
  if (nsplit >= 8) {
    const int s_lo = local & 7, q = local >> 3;
    tile = q % tiles; split = (q / tiles) * 8 + s_lo;
    if (split >= nsplit) return;
  } else {
    tile = local / nsplit; split = local - tile * nsplit;
    if (tile >= tiles) return;
  }
  WFatView v;
  v.grad = J.grad; v.x = J.x; v.part = J.part;
  v.out0 = J.out[0]; v.out1 = J.out[1]; v.out2 = J.out[2];
  v.outb0 = J.outb[0]; v.outb1 = J.outb[1]; v.outb2 = J.outb[2];
  v.ldg = J.ldg; v.lda = J.lda; v.N = J.N; v.K = J.K; v.nper = J.nper; v.nsplit = nsplit; v.direct = J.direct;
  const int nb = tile / kt, kb = tile - nb * kt;
  const long long m_begin = (long long)split * rows;
  long long m_end = m_begin + rows;
  if (m_end > M) m_end = M;
  switch (type) {
#ifdef WFAT_TYPE0
    case 0: wgrad_fat_tile<3, 1, WFAT_R(16), DBG>(v, m_begin, m_end, nb * 384, kb * 128, split, kb == 0, smem); break;
#endif
    case 1: wgrad_fat_tile<2, 1, WFAT_R(12), DBG>(v, m_begin, m_end, nb * 256, kb * 128, split, kb == 0, smem); break;
    case 2: wgrad_fat_tile<1, 2, WFAT_R(12), DBG>(v, m_begin, m_end, nb * 128, kb * 256, split, kb == 0, smem); break;
    default: wgrad_fat_tile<1, 1, WFAT_R(8), DBG>(v, m_begin, m_end, nb * 128, kb * 128, split, kb == 0, smem); break;
  }
}

static int wfat_pick(int N, int K) {
  int best = -1;
  long long bc = 0;
#ifdef WFAT_TYPE0
  const int t0 = 0;
#else
  const int t0 = 1;           // without the 384 x 128 tile (192 accumulator registers: one wave per SIMD and nothing else on the CU)
#endif
  for (int t = t0; t < 4; ++t) {
    const int TN = 128 * WFAT_TYPES[t].tnb, TK = 128 * WFAT_TYPES[t].tkb;
    if (N % TN || K % TK) continue;
    const long long c = (long long)(N / TN) * (K / TK) * (TN + TK);          // bf16 elements read per row
    if (best < 0 || c < bc) { best = t; bc = c; }
  }
  return best;
}
// the launch geometry: a function of the jobs' shapes and the workgroup budget ONLY (the size query and the launch are handed the
// same `blocks`, so they cannot disagree; the launch checks the capacity it is given all the same)
static bool wfat_geometry(const ltu_wgrad_job* jobs, int njobs, int blocks, WFatArgs& fa, long long* part_off, long long* ws_floats,
                          int* grid) {
  if (njobs < 1 || njobs > LTU_WGRAD_GROUP_MAX || !ring_enabled() || ltu_knob("LTU_NO_WGROUP", 0)) return false;
  const int budget = blocks > 0 ? blocks : ltu_knob_pos("LTU_WGROUP_BLOCKS", 256);
  double total = 0.0;
  for (int i = 0; i < njobs; ++i) {
    const ltu_wgrad_job& j = jobs[i];
    if (!tn_ring_shape_ok(j.M, j.N, j.K) || j.lda % 8 || j.ldg % 8 || j.nw < 1 || j.nw > 3 || j.N % j.nw || (j.N / j.nw) % 32) return false;
    if (((uintptr_t)j.a | (uintptr_t)j.grad) & 15) return false;
    const int t = wfat_pick(j.N, j.K);
    if (t < 0) return false;
    WFatJob& f = fa.j[i];
    f.type = (short)t;
    f.nt = (short)(j.N / (128 * WFAT_TYPES[t].tnb));
    f.kt = (short)(j.K / (128 * WFAT_TYPES[t].tkb));
    total += (double)f.nt * f.kt * (WFAT_TYPES[t].tnb + WFAT_TYPES[t].tkb) * (double)j.M;
  }
  const double target = total / budget;                       // (128-column block x row) units one workgroup should stream
  const bool no_direct = ltu_knob("LTU_WGROUP_NO_DIRECT", 0) != 0;
  long long off = 0;
  int wg = 0;
  fa.njobs = njobs;
  for (int i = 0; i < njobs; ++i) {
    const ltu_wgrad_job& j = jobs[i];
    WFatJob& f = fa.j[i];
    const double w = (double)(WFAT_TYPES[f.type].tnb + WFAT_TYPES[f.type].tkb) * (double)j.M;
    long long ns = (long long)(w / target + 0.5);
    const long long maxs = j.M / 128;                         // splits of >= 128 rows
    if (ns > maxs) ns = maxs;
    if (ns < 1) ns = 1;
    long long rows = ((j.M + ns - 1) / ns + 31) / 32 * 32;
    ns = (j.M + rows - 1) / rows;
    if (ns > 32767) return false;
    f.M = j.M; f.N = j.N; f.K = j.K; f.ldg = j.ldg; f.lda = j.lda; f.nper = j.N / j.nw;
    f.rows = (int)rows; f.nsplit = (short)ns;
    f.direct = (short)(ns == 1 && !no_direct);
    f.pad = 0;
    f.wg_begin = wg;
    const int tiles = f.nt * f.kt;
    const int n_wg = ns >= 8 ? (int)((ns + 7) / 8) * 8 * tiles : (int)ns * tiles;
    wg += (n_wg + 7) / 8 * 8;
    part_off[i] = off;
    if (!f.direct) off += ns * (long long)j.N * ((long long)j.K + 1);
  }
  *ws_floats = off > 0 ? off : 4;                             // a non-zero size keeps "0 = not handled" of the C-ABI
  *grid = wg;
  return true;
}
static bool wfat_enabled() { return ltu_knob("LTU_WGROUP_FAT", 0) != 0; }
long long wgrad_fat_group_ws_floats(const ltu_wgrad_job* jobs, int njobs, int blocks) {
  WFatArgs fa;
  long long n = 0, off[LTU_WGRAD_GROUP_MAX];
  int grid = 0;
  return wfat_geometry(jobs, njobs, blocks, fa, off, &n, &grid) ? n : 0;
}
// LTU_OK / hipError, 1 = group not handled, LTU_E_ARG = the workspace is smaller than this geometry needs
int launch_wgrad_fat_group_bf16(const ltu_wgrad_job* jobs, int njobs, int blocks, float* ws, long long ws_floats, hipStream_t st) {
  WFatArgs fa;
  memset(&fa, 0, sizeof(fa));
  long long n = 0, off[LTU_WGRAD_GROUP_MAX];
  int grid = 0;
  if (ws == nullptr || !wfat_geometry(jobs, njobs, blocks, fa, off, &n, &grid)) return 1;
  if (n > ws_floats) return LTU_E_ARG;
  WFoldArgs fo;
  memset(&fo, 0, sizeof(fo));
  int fblocks = 0, nf = 0;
  for (int i = 0; i < njobs; ++i) {
    WFatJob& j = fa.j[i];
    j.grad = reinterpret_cast<const uint16_t*>(jobs[i].grad);
    j.x = reinterpret_cast<const uint16_t*>(jobs[i].a);
    j.part = ws + off[i];
    for (int s = 0; s < 3; ++s) { j.out[s] = s < jobs[i].nw ? jobs[i].dw[s] : nullptr; j.outb[s] = s < jobs[i].nw ? jobs[i].db[s] : nullptr; }
    if (j.direct) continue;
    WFoldJob& f = fo.j[nf++];
    f.part = j.part; f.nsplit = j.nsplit; f.N = j.N; f.K = j.K; f.nseg = jobs[i].nw;
    for (int s = 0; s < 3; ++s) { f.out[s] = j.out[s]; f.outb[s] = j.outb[s]; }
    f.blk_begin = fblocks;
    f.wblocks = (int)(((long long)j.N * j.K / 4 + 63) / 64);
    fblocks += f.wblocks + (j.N + 63) / 64;
  }
  fo.njobs = nf;
  static LtuDevOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_fat_group_bf16_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, WFAT_RING_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_fat_group_bf16_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, WFAT_RING_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_fat_group_bf16_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, WFAT_RING_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_fat_group_bf16_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, WFAT_RING_BYTES);
  }
  const int dbg = ltu_knob("LTU_WFAT_DBG", 0);     // ablation: 1 = no fragment reads / MFMAs, 2 = no LDS-DMA (results invalid)
  if (dbg == 1) hipLaunchKernelGGL(wgrad_fat_group_bf16_kernel<1>, dim3(grid), dim3(256), WFAT_RING_BYTES, st, fa);
  else if (dbg == 2) hipLaunchKernelGGL(wgrad_fat_group_bf16_kernel<2>, dim3(grid), dim3(256), WFAT_RING_BYTES, st, fa);
  else if (dbg == 3) hipLaunchKernelGGL(wgrad_fat_group_bf16_kernel<3>, dim3(grid), dim3(256), WFAT_RING_BYTES, st, fa);
  else hipLaunchKernelGGL(wgrad_fat_group_bf16_kernel<0>, dim3(grid), dim3(256), WFAT_RING_BYTES, st, fa);
  if (nf > 0) hipLaunchKernelGGL(wgroup_fold_kernel, dim3(fblocks), dim3(256), 0, st, fo);
  return ltu_check_launch();
}
#else
static bool wfat_enabled() { return false; }
long long wgrad_fat_group_ws_floats(const ltu_wgrad_job*, int, int) { return 0; }
int launch_wgrad_fat_group_bf16(const ltu_wgrad_job*, int, int, float*, long long, hipStream_t) { return 1; }
#endif
