// Stride-1 3x3x3 convolution (forward and data gradient) with an LDS-staged halo brick, bf16 matrix cores.
//
// The generic implicit GEMM (gemm_bf16.hip) fetches every A tile through the vector memory pipeline once per tap:
// 27x the unique bytes, and for the high-resolution, few-channel levels of the U-Net that pipeline - not HBM, not the
// matrix cores - is the limit.  Here a workgroup owns a 4x4x8 brick of output voxels (128 rows), stages the 6x6x10
// halo brick of one 16/32-channel chunk in LDS ONCE (2.8x the unique bytes instead of 27x) and forms the MFMA A
// fragment of every tap by adding a constant tap offset to a per-lane LDS address.  Only the weights stream per tap
// (double-buffered [taps][Cout][chunk] tiles).  Output staging / 16-byte stores as in the implicit GEMM.
//
//   rows  m = voxel (h,w,d) of the brick, 32-row MFMA tiles = one h-plane (4 w x 8 d)
//   K     = 27 taps x C channels, walked chunk by chunk (outer) and tap by tap (inner)
//   N     = output channels, BN per workgroup (32 / 64 / 128)
// Data gradient = same kernel on the output gradient with mirrored tap offsets and the [Ci][27][Co] operand.
#include "gemm_desc.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define HALO_H 6
#define HALO_W 6
#define HALO_D 10
#define HALO_VOX (HALO_H * HALO_W * HALO_D)
#define LDH 40            // LDS row stride in bf16 elements (32 channels + 8 pad = 80 bytes)

template <int WM, int WN, int TM, int TN, int TS>
__global__ void __launch_bounds__(256) conv3_halo_bf16_kernel(const HaloArgs a) {
  static_assert(WM * TM == 4 && WM * WN == 4, "128-row brick on 4 waves");
  constexpr int BN = WN * TN * 32, NSTAGE = 27 / TS;
  constexpr int HALO_ELEMS = HALO_VOX * LDH, B_ELEMS = 2 * TS * BN * LDH, LDC = BN + 8, STAGE_ELEMS = 128 * LDC;
  constexpr int SMEM_ELEMS = (HALO_ELEMS + B_ELEMS) > STAGE_ELEMS ? (HALO_ELEMS + B_ELEMS) : STAGE_ELEMS;
  __shared__ __attribute__((aligned(16))) uint16_t smem[SMEM_ELEMS];
  uint16_t* halo = smem;                   // [HALO_VOX][LDH]
  uint16_t* Bs = smem + HALO_ELEMS;        // [2][TS][BN][LDH]
  constexpr int LBV = (TS * BN * 4 + 255) / 256;       // weight vectors per thread and stage (4 = max 16-byte parts per row)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  const int nbh = (a.H + 3) / 4, nbw = (a.W + 3) / 4, nbd = (a.D + 7) / 8;
  int bid = blockIdx.x;
  const int bd = bid % nbd; bid /= nbd;
  const int bw = bid % nbw; bid /= nbw;
  const int bh = bid % nbh;
  const int b = bid / nbh;
  const int h0 = bh * 4, w0 = bw * 4, d0 = bd * 8;
  const int n_blk = blockIdx.y * BN;
  const int VPV = a.CC / 8;                // 16-byte parts per voxel / weight row of one chunk
  const int nchunk = (a.C + a.CC - 1) / a.CC;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // per-lane LDS base of the A fragment rows: row r = (tile)*32 + li -> brick voxel (r>>5, (r>>3)&3, r&7)
  int arow[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = (wm * TM + i) * 32 + li;
    arow[i] = (((r >> 5) * HALO_W + ((r >> 3) & 3)) * HALO_D + (r & 7)) * LDH + lh * 8;
  }

  uint4 hreg[6];
  auto load_halo = [&](int chunk) {
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      const int idx = tid + p * 256;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (idx < HALO_VOX * VPV) {
        const int hv = idx / VPV, part = idx - hv * VPV;
        const int hd = hv % HALO_D, hw = (hv / HALO_D) % HALO_W, hh = hv / (HALO_D * HALO_W);
        const int h = h0 - 1 + hh, w = w0 - 1 + hw, d = d0 - 1 + hd;
        const int c = chunk * a.CC + part * 8;
        if ((unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W && (unsigned)d < (unsigned)a.D && c < a.C) {
          const long long vox = (((long long)b * a.H + h) * a.W + w) * a.D + d;
          const uint16_t* src = c < a.c0 ? reinterpret_cast<const uint16_t*>(a.x0) + vox * a.lda0 + c
                                         : reinterpret_cast<const uint16_t*>(a.x1) + vox * a.lda1 + (c - a.c0);
          v = *reinterpret_cast<const uint4*>(src);
        }
      }
      hreg[p] = v;
    }
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      const int idx = tid + p * 256;
      if (idx < HALO_VOX * VPV) {
        const int hv = idx / VPV, part = idx - hv * VPV;
        *reinterpret_cast<uint4*>(&halo[hv * LDH + part * 8]) = hreg[p];
      }
    }
  };
  uint4 breg[LBV];
  auto load_b = [&](int chunk, int stage) {
#pragma unroll
    for (int p = 0; p < LBV; ++p) {
      const int idx = tid + p * 256;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (idx < TS * BN * VPV) {
        const int part = idx % VPV, nl = (idx / VPV) % BN, t = idx / (VPV * BN);
        const int n = n_blk + nl, tap = stage * TS + t, c = chunk * a.CC + part * 8;
        if (n < a.N && c < a.C)
          v = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(a.w) + ((long long)n * 27 + tap) * a.C + c);
      }
      breg[p] = v;
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int p = 0; p < LBV; ++p) {
      const int idx = tid + p * 256;
      if (idx < TS * BN * VPV) {
        const int part = idx % VPV, nl = (idx / VPV) % BN, t = idx / (VPV * BN);
        *reinterpret_cast<uint4*>(&Bs[((buf * TS + t) * BN + nl) * LDH + part * 8]) = breg[p];
      }
    }
  };

  const int ksteps = a.CC / 16;
  load_halo(0);
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    __syncthreads();                       // previous chunk fully consumed (halo and both weight buffers)
    store_halo();
    load_b(chunk, 0);
    store_b(0);
    __syncthreads();
    if (chunk + 1 < nchunk) load_halo(chunk + 1);     // in flight during the 27 taps below
    for (int s = 0; s < NSTAGE; ++s) {
      const int buf = s & 1;
      if (s + 1 < NSTAGE) load_b(chunk, s + 1);
#pragma unroll
      for (int t = 0; t < TS; ++t) {
        const int tap = s * TS + t;
        int th = tap / 9, tw = (tap / 3) % 3, td = tap % 3;
        if (a.flip) { th = 2 - th; tw = 2 - tw; td = 2 - td; }
        const int tapoff = ((th * HALO_W + tw) * HALO_D + td) * LDH;
        for (int ks = 0; ks < ksteps; ++ks) {
          bf16x8 av[TM], bv[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) av[i] = *reinterpret_cast<const bf16x8*>(&halo[arow[i] + tapoff + ks * 16]);
#pragma unroll
          for (int j = 0; j < TN; ++j)
            bv[j] = *reinterpret_cast<const bf16x8*>(&Bs[((buf * TS + t) * BN + (wn * TN + j) * 32 + li) * LDH + ks * 16 + lh * 8]);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
      }
      if (s + 1 < NSTAGE) store_b(buf ^ 1);
      __syncthreads();
    }
  }

  // epilogue: bias, convert, stage the 128 x BN tile in LDS, 16-byte stores
  uint16_t* Cs = smem;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nl = (wn * TN + j) * 32 + li;
    const int n = n_blk + nl;
    const float bvv = (a.bias != nullptr && n < a.N) ? a.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ml = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        Cs[ml * LDC + nl] = f32_to_bf16(acc[i][j][r] + bvv);
      }
  }
  __syncthreads();
  constexpr int CPR = BN / 4;                 // 8-byte pieces per row (N and the split point are multiples of 4)
  for (int idx = tid; idx < 128 * CPR; idx += 256) {
    const int ml = idx / CPR, nl = (idx % CPR) * 4;
    const int n = n_blk + nl;
    const int h = h0 + (ml >> 5), w = w0 + ((ml >> 3) & 3), d = d0 + (ml & 7);
    if (n >= a.N || h >= a.H || w >= a.W || d >= a.D) continue;
    const long long vox = (((long long)b * a.H + h) * a.W + w) * a.D + d;
    const uint2 v = *reinterpret_cast<const uint2*>(&Cs[ml * LDC + nl]);
    uint16_t* dst = n < a.n0 ? reinterpret_cast<uint16_t*>(a.o0) + vox * a.ldo0 + n
                             : reinterpret_cast<uint16_t*>(a.o1) + vox * a.ldo1 + (n - a.n0);
    *reinterpret_cast<uint2*>(dst) = v;
  }
}

// returns LTU_OK after launching, or 1 when the shape is not handled here (the caller falls back to the implicit GEMM)
int launch_conv_halo_bf16(HaloArgs a, hipStream_t st) {
  if (a.C % 8 || a.c0 % 8 || a.lda0 % 8 || a.lda1 % 8 || a.N % 4 || a.n0 % 4 || a.ldo0 % 4 || a.ldo1 % 4) return 1;
  if (a.H < 2 || a.W < 2 || a.D < 4) return 1;
  a.CC = (a.C % 32 == 0 && a.c0 % 32 == 0) ? 32 : 16;
  if (a.C % 16 != 0 && a.C != 8) return 1;
  if (a.c0 % a.CC != 0 && a.c0 != a.C) return 1;
  const long long bricks = (long long)a.B * ((a.H + 3) / 4) * ((a.W + 3) / 4) * ((a.D + 7) / 8);
  if (bricks >= (1LL << 31)) return 1;
  if (a.N > 64) {
    dim3 grid((unsigned)bricks, cdiv(a.N, 128));
    hipLaunchKernelGGL((conv3_halo_bf16_kernel<2, 2, 2, 2, 1>), grid, dim3(256), 0, st, a);
  } else if (a.N > 32) {
    dim3 grid((unsigned)bricks, 1);
    hipLaunchKernelGGL((conv3_halo_bf16_kernel<4, 1, 1, 2, 3>), grid, dim3(256), 0, st, a);
  } else {
    dim3 grid((unsigned)bricks, 1);
    hipLaunchKernelGGL((conv3_halo_bf16_kernel<4, 1, 1, 1, 3>), grid, dim3(256), 0, st, a);
  }
  return ltu_check_launch();
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same convs:  dW[n][tap][c] = sum over voxels of G[v][n] * X[v + tap][c].
// A workgroup owns (32-channel chunk, 32-column gradient tile, a range of bricks).  Per brick it stages the halo brick of
// X and the 128 x 32 gradient tile once; the 27 taps are dealt round-robin to the 4 waves, each keeping its <= 7
// 32(n) x 32(c) fp32 accumulators in registers for the whole brick range.  Both MFMA operands are K(=voxel)-major in LDS,
// so they are read with the transposing ds_read_b64_tr_b16; a tap is again only a constant LDS offset.  The splits store
// their tiles to part[split][N][27*C] (+ bias partials) and wgrad_reduce_kernel (gemm_bf16.hip) folds them.
typedef __attribute__((ext_vector_type(4))) short hs16x4;
typedef __attribute__((address_space(3))) hs16x4 lds_hs16x4;
#define LDGH 40

__global__ void __launch_bounds__(256) conv3_wgrad_halo_bf16_kernel(const WHaloArgs a) {
  __shared__ __attribute__((aligned(16))) uint16_t halo[HALO_VOX * LDH];
  __shared__ __attribute__((aligned(16))) uint16_t Gs[128 * LDGH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int chunk = blockIdx.x, n_blk = blockIdx.y * 32;
  const int nbh = (a.H + 3) / 4, nbw = (a.W + 3) / 4, nbd = (a.D + 7) / 8;
  const int VPV = a.CC / 8;
  const int brick_lo = blockIdx.z * a.bricks_per_split;
  int brick_hi = brick_lo + a.bricks_per_split;
  if (brick_hi > a.bricks) brick_hi = a.bricks;

  f32x16 acc[7];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  int tapoff[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int tap = wave + 4 * i;
    tapoff[i] = (((tap / 9) * HALO_W + (tap / 3) % 3) * HALO_D + tap % 3) * LDH;
  }
  // transposing-read lane geometry (see wgrad_tn_bf16_kernel): lane -> (row trow (+4), columns tcol..tcol+3) of a 16-row slab
  const int gq = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  const int tcol = 16 * (gq & 1) + 4 * tp;
  const int trow = 8 * (gq >> 1) + tq;
  const int gbase = trow * LDGH + tcol;
  // slab ks covers brick rows 16ks..16ks+15 = (h = ks>>1, w = 2(ks&1) + (row>>3), d = row&7)
  const int hbase = ((gq >> 1) * HALO_D + tq) * LDH + tcol;

  uint4 hreg[6], greg[2];
  auto load_brick = [&](int brick) {
    int t = brick;
    const int bd = t % nbd; t /= nbd;
    const int bw = t % nbw; t /= nbw;
    const int bh = t % nbh;
    const int b = t / nbh;
    const int h0 = bh * 4, w0 = bw * 4, d0 = bd * 8;
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      const int idx = tid + p * 256;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (idx < HALO_VOX * VPV) {
        const int hv = idx / VPV, part = idx - hv * VPV;
        const int hd = hv % HALO_D, hw = (hv / HALO_D) % HALO_W, hh = hv / (HALO_D * HALO_W);
        const int h = h0 - 1 + hh, w = w0 - 1 + hw, d = d0 - 1 + hd;
        const int c = chunk * a.CC + part * 8;
        if ((unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W && (unsigned)d < (unsigned)a.D && c < a.C) {
          const long long vox = (((long long)b * a.H + h) * a.W + w) * a.D + d;
          const uint16_t* src = c < a.c0 ? reinterpret_cast<const uint16_t*>(a.x0) + vox * a.lda0 + c
                                         : reinterpret_cast<const uint16_t*>(a.x1) + vox * a.lda1 + (c - a.c0);
          v = *reinterpret_cast<const uint4*>(src);
        }
      }
      hreg[p] = v;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int idx = tid + p * 256;
      const int row = idx >> 2, n = n_blk + (idx & 3) * 8;
      const int h = h0 + (row >> 5), w = w0 + ((row >> 3) & 3), d = d0 + (row & 7);
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (h < a.H && w < a.W && d < a.D && n < a.N) {
        const long long vox = (((long long)b * a.H + h) * a.W + w) * a.D + d;
        v = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(a.grad) + vox * a.ldg + n);
      }
      greg[p] = v;
    }
  };
  auto store_brick = [&]() {
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      const int idx = tid + p * 256;
      if (idx < HALO_VOX * VPV) {
        const int hv = idx / VPV, part = idx - hv * VPV;
        *reinterpret_cast<uint4*>(&halo[hv * LDH + part * 8]) = hreg[p];
      }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int idx = tid + p * 256;
      *reinterpret_cast<uint4*>(&Gs[(idx >> 2) * LDGH + (idx & 3) * 8]) = greg[p];
    }
  };

  const bool do_bias = chunk == 0 && tid < 32;
  float bsum = 0.f;
  if (brick_lo < brick_hi) load_brick(brick_lo);
  for (int brick = brick_lo; brick < brick_hi; ++brick) {
    __syncthreads();
    store_brick();
    __syncthreads();
    if (brick + 1 < brick_hi) load_brick(brick + 1);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      union { struct { hs16x4 l, h; } s; bf16x8 v; } ua;
      const uint16_t* pg = &Gs[ks * 16 * LDGH + gbase];
      ua.s.l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_hs16x4*)pg);
      ua.s.h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_hs16x4*)(pg + 4 * LDGH));
      const int slab = ((ks >> 1) * HALO_W * HALO_D + (ks & 1) * 2 * HALO_D) * LDH + hbase;
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        if (wave + 4 * i < 27) {
          union { struct { hs16x4 l, h; } s; bf16x8 v; } ub;
          const uint16_t* px = &halo[slab + tapoff[i]];
          ub.s.l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_hs16x4*)px);
          ub.s.h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_hs16x4*)(px + 4 * LDH));
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, acc[i], 0, 0, 0);
        }
      }
    }
    if (do_bias) {
#pragma unroll 8
      for (int r = 0; r < 128; ++r) bsum += bf16_to_f32(Gs[r * LDGH + tid]);
    }
  }

  float* pz = a.part + (long long)blockIdx.z * a.npad * a.kpad;
  if (do_bias && n_blk + tid < a.N) a.bpart[(long long)blockIdx.z * a.npad + n_blk + tid] = bsum;
  const int c = chunk * a.CC + li;
  if (li >= a.CC || c >= a.C) return;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int tap = wave + 4 * i;
    if (tap >= 27) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n_blk + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (n < a.N) pz[(long long)n * a.kpad + tap * a.C + c] = acc[i][r];
    }
  }
}

static int whalo_blocks() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("LTU_WHALO_BLOCKS"); v = (e && atoi(e) > 0) ? atoi(e) : 512; }
  return v;
}

static bool whalo_shape_ok(int C, int N) { return N % 8 == 0 && N <= 64 && (C % 16 == 0 || C == 8); }

long long conv_wgrad_halo_ws_floats(int N, int K) {
  if (K % 27) return 0;
  const int C = K / 27;
  if (!whalo_shape_ok(C, N)) return 0;
  const int CC = C % 32 == 0 ? 32 : 16;
  long long ns = whalo_blocks() / ((long long)cdiv(C, CC) * cdiv(N, 32));
  if (ns < 1) ns = 1;
  return ns * N * ((long long)K + 1);
}

// fills part/bpart/npad/kpad/splits; returns LTU_OK after launching, or 1 when the shape is not handled
int launch_conv_wgrad_halo_bf16(WHaloArgs a, int* nsplit_out, hipStream_t st) {
  if (!whalo_shape_ok(a.C, a.N) || a.c0 % 8 || a.lda0 % 8 || a.lda1 % 8 || a.ldg % 8) return 1;
  if (a.H < 2 || a.W < 2 || a.D < 4) return 1;
  a.CC = a.C % 32 == 0 ? 32 : 16;
  const long long bricks = (long long)a.B * ((a.H + 3) / 4) * ((a.W + 3) / 4) * ((a.D + 7) / 8);
  if (bricks >= (1LL << 31)) return 1;
  const int nchunk = cdiv(a.C, a.CC), ntile = cdiv(a.N, 32);
  long long ns = whalo_blocks() / ((long long)nchunk * ntile);
  if (ns < 1) ns = 1;
  if (ns > bricks) ns = bricks;
  a.bricks = (int)bricks;
  a.bricks_per_split = (int)((bricks + ns - 1) / ns);
  const int nsplit = (int)((bricks + a.bricks_per_split - 1) / a.bricks_per_split);
  a.npad = a.N;
  a.kpad = 27 * a.C;
  a.bpart = a.part + (long long)nsplit * a.npad * a.kpad;
  *nsplit_out = nsplit;
  hipLaunchKernelGGL(conv3_wgrad_halo_bf16_kernel, dim3(nchunk, ntile, nsplit), dim3(256), 0, st, a);
  return ltu_check_launch();
}
