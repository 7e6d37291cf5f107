// Weight gradient of the stride-1 3x3x3 level convs (model/Unet_3Dblock.py:310-316, 523-531), 32-channel chunks: second generation of
// conv3_wgrad_halo_bf16_kernel (conv_halo.hip).  Same arithmetic - dW[n][tap][c] = sum over voxels of G[v][n] X[v + tap][c], K = voxel,
// both operands by transposing LDS reads, the 27 taps dealt to the 4 waves, bias sums in wave 3's spare accumulator - but:
//   * the 6x6x10 halo of X and the 128 x 32 tile of G travel global -> LDS by LDS-DMA (no register staging, no ds_write, no second
//     barrier) into a ring of three 32 KB brick buffers: two bricks are in flight while one is multiplied;
//   * a workgroup walks a LONG run of bricks (the launch is ~256 workgroups wide, not 512-768): the first generation wrote 1.1 GB of
//     fp32 partial tiles per step (e.g. 512 row splits x 110 KB for the 27 648-element gradient of a 32 -> 32 conv) and its 32 fold
//     launches read them back - most of those kernels' time was their epilogue.  Half to a quarter of the splits here.
// Partial layout, bias partials and the fold (wgrad_reduce_kernel) are the first generation's.
#include "gemm_desc.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short wr16x4;
typedef __attribute__((address_space(3))) wr16x4 lds_wr16x4;

#define WR_HROWS 368                      // 360 halo voxels padded to 23 pieces of 16 rows
#define WR_GOFF (WR_HROWS * 32)           // element offset of the G tile [128][32] inside a brick buffer
#define WR_BUF (16384)                    // elements per brick buffer: halo 11 776 + G 4 096 + one spare piece (512) = 32 KB
#define WR_NB 3

__device__ __attribute__((aligned(64))) uint32_t ltu_zero_wr[16];     // source of out-of-volume / padding rows (16 bytes per lane)

__device__ __forceinline__ void wr_glds16(const void* src, uint32_t lds_byte_addr) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_byte_addr);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wr_sync() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

__global__ void __launch_bounds__(256, 1) conv3_wgrad_halo_ring_bf16_kernel(const WHaloArgs a) {
  constexpr int NA = 7, LD = 32, HW_ = 6, HD_ = 10;
  extern __shared__ __attribute__((aligned(1024))) uint16_t smem[];      // [WR_NB][WR_BUF]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int chunk = blockIdx.x, n_blk = blockIdx.y * 32;
  const int nbh = (a.H + 3) / 4, nbw = (a.W + 3) / 4, nbd = (a.D + 7) / 8;
  const int brick_lo = blockIdx.z * a.bricks_per_split;
  int brick_hi = brick_lo + a.bricks_per_split;
  if (brick_hi > a.bricks) brick_hi = a.bricks;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;

  // ---- LDS-DMA pieces of a brick: 32 of 1 KB (16 rows of 64 bytes); wave w issues pieces w, w + 4, ...: 23 halo pieces, 8 G pieces,
  // one spare (so that every wave has 8 in flight per brick: one counted wait for all).  Lane -> (row = lane >> 2, 16-byte part = lane & 3)
  const int prow = lane >> 2, part = lane & 3;
  const int cx = chunk * 32 + part * 8;                      // channel of x this lane fetches
  const bool x_from1 = cx >= a.c0;                           // virtual concat: c0 is a multiple of 32 or equals C
  const uint16_t* xbase = x_from1 ? reinterpret_cast<const uint16_t*>(a.x1) + (cx - a.c0) : reinterpret_cast<const uint16_t*>(a.x0) + cx;
  const long long xld = x_from1 ? a.lda1 : a.lda0;
  const int ng = n_blk + part * 8;                           // gradient column this lane fetches
  const bool g_from1 = a.grad1 != nullptr && ng >= a.gn0;
  const uint16_t* gbase_p = g_from1 ? reinterpret_cast<const uint16_t*>(a.grad1) + (ng - a.gn0) : reinterpret_cast<const uint16_t*>(a.grad) + ng;
  const long long gld = g_from1 ? a.ldg1 : a.ldg;
  const bool g_ok = ng < a.N, x_ok = cx < a.C;
  const char* zsrc = reinterpret_cast<const char*>(ltu_zero_wr) + part * 16;
  // brick-independent part of a lane's 8 fetches: voxel offset relative to the brick origin and the (h, w, d) step for the bounds test
  int rel[8], pk[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int p = wave + 4 * s;
    int eh = 0, ew = 0, ed = 0, ok = 0;                      // steps from (h0 - 1, w0 - 1, d0 - 1)
    if (p < 23) {
      const int hv = p * 16 + prow;
      ed = hv % HD_, ew = (hv / HD_) % HW_, eh = hv / (HW_ * HD_);
      ok = x_ok && hv < 360;
    } else if (p < 31) {
      const int row = (p - 23) * 16 + prow;
      eh = 1 + (row >> 5), ew = 1 + ((row >> 3) & 3), ed = 1 + (row & 7);
      ok = g_ok;
    }
    rel[s] = ((eh - 1) * a.W + (ew - 1)) * a.D + (ed - 1);
    pk[s] = ok ? (eh << 16 | ew << 8 | ed) : -1;
  }
  auto issue_brick = [&](int brick, int buf) {
    int t = brick;
    const int bd = t % nbd; t /= nbd;
    const int bw = t % nbw; t /= nbw;
    const int bh = t % nbh;
    const int b = t / nbh;
    const int h0 = bh * 4 - 1, w0 = bw * 4 - 1, d0 = bd * 8 - 1;
    const long long vox0 = (((long long)b * a.H + bh * 4) * a.W + bw * 4) * a.D + bd * 8;
    const uint32_t bb = lds0 + buf * (WR_BUF * 2);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int p = wave + 4 * s;                            // piece index (wave-uniform)
      const bool isx = p < 23;
      const int h = h0 + (pk[s] >> 16), w = w0 + ((pk[s] >> 8) & 255), d = d0 + (pk[s] & 255);
      const bool in = pk[s] >= 0 && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W && (unsigned)d < (unsigned)a.D;
      const char* ptr = reinterpret_cast<const char*>((isx ? xbase : gbase_p) + (vox0 + rel[s]) * (isx ? xld : gld));
      wr_glds16(in ? ptr : zsrc, bb + p * 1024);
    }
  };

  f32x16 acc[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  int tapoff[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int tap = min(wave + 4 * i, 26);
    tapoff[i] = (((tap / 9) * HW_ + (tap / 3) % 3) * HD_ + tap % 3) * LD;
  }
  // transposing-read lane geometry (as in conv3_wgrad_halo_bf16_kernel): lane -> (row trow (+4), columns tcol..tcol+3) of a 16-row slab
  const int gq = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  const int tcol = 16 * (gq & 1) + 4 * tp;
  const int trow = 8 * (gq >> 1) + tq;
  const int gfrag = trow * LD + tcol;
  const int hfrag = ((gq >> 1) * HD_ + tq) * LD + tcol;
  const bool do_bias = chunk == 0 && wave == 3;
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

  const int nb = brick_hi - brick_lo;
  if (nb > 0) issue_brick(brick_lo, 0);
  if (nb > 1) issue_brick(brick_lo + 1, 1);
  int buf = 0;
  for (int i = 0; i < nb; ++i) {
    // brick i has landed (this wave's pieces: the counted wait; the other waves': the barrier) and every wave has left brick i - 1,
    // whose buffer takes brick i + 2
    if (i + 1 < nb) wr_sync<8>(); else wr_sync<0>();
    if (i + 2 < nb) issue_brick(brick_lo + i + 2, buf >= 1 ? buf - 1 : WR_NB - 1);
    const uint16_t* halo = smem + buf * WR_BUF;
    const uint16_t* Gs = halo + WR_GOFF;
    buf = buf + 1 == WR_NB ? 0 : buf + 1;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      union { struct { wr16x4 l, h; } s; bf16x8 v; } ua;
      const uint16_t* pg = Gs + ks * 16 * LD + gfrag;
      ua.s.l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_wr16x4*)pg);
      ua.s.h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_wr16x4*)(pg + 4 * LD));
      const int slab = ((ks >> 1) * HW_ * HD_ + (ks & 1) * 2 * HD_) * LD + hfrag;
#pragma unroll
      for (int t = 0; t < NA; ++t) {
        union { struct { wr16x4 l, h; } s; bf16x8 v; } ub;
        const uint16_t* px = halo + slab + tapoff[t];
        ub.s.l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_wr16x4*)px);
        ub.s.h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_wr16x4*)(px + 4 * LD));
        if (t == NA - 1) ub.v = do_bias ? ones : ub.v;
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, acc[t], 0, 0, 0);
      }
    }
  }

  float* pz = a.part + (long long)blockIdx.z * a.npad * a.kpad;
  if (do_bias && li == 0) {                // every column of the spare accumulator holds the sums: lanes 0 and 32 own the 2 x 16 rows
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n_blk + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (n < a.N) a.bpart[(long long)blockIdx.z * a.npad + n] = acc[NA - 1][r];
    }
  }
  const int c = chunk * 32 + li;
  if (c >= a.C) return;
#pragma unroll
  for (int t = 0; t < NA; ++t) {
    const int tap = wave + 4 * t;
    if (tap >= 27) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n_blk + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (n < a.N) pz[(long long)n * a.kpad + tap * a.C + c] = acc[t][r];
    }
  }
}

// LTU_OK after launching, 1 = shape not handled (the first generation takes it), LTU_E_ARG = the workspace is short
int launch_conv_wgrad_halo_ring_bf16(WHaloArgs a, int* nsplit_out, hipStream_t st) {
  if (a.C % 32 || a.c0 % 32 || a.N % 8 || a.N > 256 || a.lda0 % 8 || a.lda1 % 8 || a.ldg % 8) return 1;
  if (a.grad1 != nullptr && (a.gn0 % 8 || a.ldg1 % 8)) return 1;
  if (a.H < 2 || a.W < 2 || a.D < 4) return 1;
  if (((uintptr_t)a.x0 | (uintptr_t)a.x1 | (uintptr_t)a.grad | (uintptr_t)a.grad1) & 15) return 1;
  const long long bricks = (long long)a.B * ((a.H + 3) / 4) * ((a.W + 3) / 4) * ((a.D + 7) / 8);
  if (bricks >= (1LL << 31)) return 1;
  const int nchunk = a.C / 32, ntile = (int)cdiv(a.N, 32);
  long long ns = ltu_knob_pos("LTU_WHALO_RING_BLOCKS", 256) / ((long long)nchunk * ntile);
  if (ns < 1) ns = 1;
  if (ns > bricks) ns = bricks;
  a.CC = 32;
  a.bricks = (int)bricks;
  a.bricks_per_split = (int)((bricks + ns - 1) / ns);
  const int nsplit = (int)((bricks + a.bricks_per_split - 1) / a.bricks_per_split);
  a.npad = a.N;
  a.kpad = 27 * a.C;
  if ((long long)nsplit * a.npad * ((long long)a.kpad + 1) > a.part_floats) return LTU_E_ARG;
  a.bpart = a.part + (long long)nsplit * a.npad * a.kpad;
  *nsplit_out = nsplit;
  constexpr int smem_bytes = WR_NB * WR_BUF * 2;
  static LtuDevOnce attr_once;
  if (attr_once.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_wgrad_halo_ring_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
  hipLaunchKernelGGL(conv3_wgrad_halo_ring_bf16_kernel, dim3(nchunk, ntile, nsplit), dim3(256), smem_bytes, st, a);
  return ltu_check_launch();
}
