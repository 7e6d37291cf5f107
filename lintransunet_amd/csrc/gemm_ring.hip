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

static void wgroup_geometry(const ltu_wgrad_job* jobs, int njobs, WGroupArgs& ga, long long* part_off, long long* ws_floats) {
  int tiles = 0;
  for (int i = 0; i < njobs; ++i) tiles += (jobs[i].N / 128) * (jobs[i].K / 128);
  const int budget = ltu_knob_pos("LTU_WGROUP_BLOCKS", 256);
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
long long tn_ring_group_ws_floats(const ltu_wgrad_job* jobs, int njobs) {
  if (!wgroup_ok(jobs, njobs)) return 0;
  WGroupArgs ga;
  long long n = 0, off[LTU_WGRAD_GROUP_MAX];
  wgroup_geometry(jobs, njobs, ga, off, &n);
  return n;
}
// launches the grouped kernel and its fold; returns 1 when the group is not handled
int launch_tn_ring_group_bf16(const ltu_wgrad_job* jobs, int njobs, float* ws, hipStream_t st) {
  if (!wgroup_ok(jobs, njobs) || ws == nullptr) return 1;
  WGroupArgs ga;
  memset(&ga, 0, sizeof(ga));
  long long n = 0, off[LTU_WGRAD_GROUP_MAX];
  wgroup_geometry(jobs, njobs, ga, off, &n);
  int blocks = 0, fblocks = 0;
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
  blocks = ga.nsplit >= 8 ? ((ga.nsplit + 7) / 8) * ga.tiles * 8 : ga.nsplit * ga.tiles;
  hipLaunchKernelGGL((wgrad_group_ring_bf16_kernel<TN_RING>), dim3(blocks), dim3(256), smem_bytes, st, ga);
  if (!ga.direct) hipLaunchKernelGGL(wgroup_fold_kernel, dim3(fblocks), dim3(256), 0, st, fa);
  return ltu_check_launch();
}
