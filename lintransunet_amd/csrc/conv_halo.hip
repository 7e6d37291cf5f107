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
#include <type_traits>

#include "gemm_desc.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define HALO_H 6
#define HALO_W 6
#define HALO_D 10
#define HALO_VOX (HALO_H * HALO_W * HALO_D)
#define LDH 40            // LDS row stride in bf16 elements (32 channels + 8 pad = 80 bytes)

template <int WN, int TN, int TS>
constexpr int halo_smem_bytes() {
  constexpr int BN = WN * TN * 32;
  constexpr int be = HALO_VOX * 32 + 2 * TS * BN * LDH, se = 128 * (BN + 8);
  return 2 * (be > se ? be : se);
}
template <int WM, int WN, int TM, int TN, int TS, int CC>
__global__ void __launch_bounds__(256) conv3_halo_bf16_kernel(const HaloArgs a) {
  static_assert(WM * TM == 4 && WM * WN == 4, "128-row brick on 4 waves");
  constexpr int BN = WN * TN * 32, NSTAGE = 27 / TS;
  constexpr int VPV = CC / 8, KS = CC / 16;               // 16-byte parts per voxel / weight row of one chunk; k-steps per tap
  constexpr int PD = CC == 16 ? 12 : HALO_D, VMASK = VPV - 1;
  constexpr int HALO_ELEMS = HALO_VOX * 32, B_ELEMS = 2 * TS * BN * LDH, LDC = BN + 8, STAGE_ELEMS = 128 * LDC;
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];        // halo_smem_bytes<...>(): the deep-stage variants exceed 64 KB
  // halo image: voxel (hh, hw, hd) at row (hh * HALO_W + hw) * PD + hd, CC elements per row, the 16-byte parts XOR-ed with the
  // low bits of hw (PD = 12 for 32-byte rows): conflict-free ds_read_b128 fragments for every tap (tools/lds_conflicts.py; the
  // padded [360][40] image this kernel started with was a 3-way conflict)
  uint16_t* halo = smem;
  uint16_t* Bs = smem + HALO_ELEMS;        // [2][TS][BN][LDH]
  constexpr int NH = (HALO_VOX * VPV + 255) / 256;       // halo vectors per thread and chunk
  constexpr int LBV = (TS * BN * VPV + 255) / 256;       // weight vectors per thread and stage

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
  const int nchunk = (a.C + CC - 1) / CC;

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
    arow[i] = (((r >> 5) * HALO_W + ((r >> 3) & 3)) * PD + (r & 7)) * CC;
  }

  // Global loads are UNCONDITIONAL (clamped address; padding is selected to zero when the registers go to LDS): with a test
  // around each load hipcc gives it a basic block of its own and drains vmcnt at the next one, so the halo prefetch and the
  // weight prefetch of a stage were complete round trips one after the other instead of loads in flight behind the MFMAs.
  uint4 hreg[NH];
  bool hin[NH];
  auto load_halo = [&](int chunk) {
#pragma unroll
    for (int p = 0; p < NH; ++p) {
      const int idx = tid + p * 256;
      const int hv = idx / VPV, part = idx - hv * VPV;
      const int hd = hv % HALO_D, hw = (hv / HALO_D) % HALO_W, hh = hv / (HALO_D * HALO_W);
      const int h = h0 - 1 + hh, w = w0 - 1 + hw, d = d0 - 1 + hd;
      const int c = chunk * CC + part * 8;
      const bool in = idx < HALO_VOX * VPV && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W && (unsigned)d < (unsigned)a.D && c < a.C;
      const long long vox = in ? (((long long)b * a.H + h) * a.W + w) * a.D + d : 0;
      const int cc = in ? c : 0;
      const uint16_t* src = cc < a.c0 ? reinterpret_cast<const uint16_t*>(a.x0) + vox * a.lda0 + cc
                                      : reinterpret_cast<const uint16_t*>(a.x1) + vox * a.lda1 + (cc - a.c0);
      hreg[p] = *reinterpret_cast<const uint4*>(src);
      hin[p] = in;
    }
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int p = 0; p < NH; ++p) {
      const int idx = tid + p * 256;
      if (idx < HALO_VOX * VPV) {
        const int hv = idx / VPV, part = idx - hv * VPV;
        const int hd = hv % HALO_D, hw = (hv / HALO_D) % HALO_W, hh = hv / (HALO_D * HALO_W);
        *reinterpret_cast<uint4*>(&halo[((hh * HALO_W + hw) * PD + hd) * CC + ((part ^ (hw & VMASK)) << 3)]) =
            hin[p] ? hreg[p] : make_uint4(0u, 0u, 0u, 0u);
      }
    }
  };
  uint4 breg[LBV];
  bool bok[LBV];
  auto load_b = [&](int chunk, int stage) {
#pragma unroll
    for (int p = 0; p < LBV; ++p) {
      const int idx = tid + p * 256;
      const int part = idx % VPV, nl = (idx / VPV) % BN, t = idx / (VPV * BN);
      const int n = n_blk + nl, tap = stage * TS + t, c = chunk * CC + part * 8;
      const bool ok = idx < TS * BN * VPV && n < a.N && c < a.C;
      const long long off = ok ? ((long long)n * 27 + tap) * a.C + c : 0;
      breg[p] = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(a.w) + off);
      bok[p] = ok;
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int p = 0; p < LBV; ++p) {
      const int idx = tid + p * 256;
      if (idx < TS * BN * VPV) {
        const int part = idx % VPV, nl = (idx / VPV) % BN, t = idx / (VPV * BN);
        *reinterpret_cast<uint4*>(&Bs[((buf * TS + t) * BN + nl) * LDH + part * 8]) = bok[p] ? breg[p] : make_uint4(0u, 0u, 0u, 0u);
      }
    }
  };

  // fragments of one tap: KS k-steps x (TM halo rows + TN weight rows); the fragments of tap t + 1 are requested before the
  // MFMAs of tap t are issued (pinned with sched_barrier: the k loop used to be a run-time loop of two MFMAs behind their own
  // LDS round trip)
  struct Frags { bf16x8 av[KS][TM], bv[KS][TN]; };
  auto load_frags = [&](int s, int t, int buf, Frags& f) {
    const int tap = s * TS + t;
    int th = tap / 9, tw = (tap / 3) % 3, td = tap % 3;
    if (a.flip) { th = 2 - th; tw = 2 - tw; td = 2 - td; }
    const int tapoff = ((th * HALO_W + tw) * PD + td) * CC;
    const int sw = ((li >> 3) + tw) & VMASK;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int ca = ((ks * 2 + lh) ^ sw) << 3;
#pragma unroll
      for (int i = 0; i < TM; ++i) f.av[ks][i] = *reinterpret_cast<const bf16x8*>(&halo[arow[i] + tapoff + ca]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        f.bv[ks][j] = *reinterpret_cast<const bf16x8*>(&Bs[((buf * TS + t) * BN + (wn * TN + j) * 32 + li) * LDH + ks * 16 + lh * 8]);
    }
  };
  auto mma = [&](const Frags& f) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.av[ks][i], f.bv[ks][j], acc[i][j], 0, 0, 0);
  };

  const int c_lo = a.part != nullptr ? (int)blockIdx.z * a.cps : 0;
  const int c_hi = a.part != nullptr ? (c_lo + a.cps < nchunk ? c_lo + a.cps : nchunk) : nchunk;
  load_halo(c_lo);
  load_b(c_lo, 0);
  for (int chunk = c_lo; chunk < c_hi; ++chunk) {
    __syncthreads();                       // previous chunk fully consumed (halo and both weight buffers)
    store_halo();
    store_b(0);
    __syncthreads();
    if (chunk + 1 < c_hi) load_halo(chunk + 1);       // in flight during the 27 taps below
    for (int s = 0; s < NSTAGE; ++s) {
      const int buf = s & 1;
      if (s + 1 < NSTAGE) load_b(chunk, s + 1);
      else if (chunk + 1 < c_hi) load_b(chunk + 1, 0);  // the next chunk's first stage travels with its halo
      Frags fa, fb;
      load_frags(s, 0, buf, fa);
#pragma unroll
      for (int t = 0; t < TS; t += 2) {
        if (t + 1 < TS) load_frags(s, t + 1, buf, fb);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < TS) load_frags(s, t + 2, buf, fa);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < TS) mma(fb);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (s + 1 < NSTAGE) store_b(buf ^ 1);
      __syncthreads();
    }
  }

  if (a.part != nullptr) {                  // split over chunks: fp32 partial tile, folded by conv_halo_fold_kernel
    float* pz = a.part + (long long)blockIdx.z * ((long long)a.B * a.H * a.W * a.D) * a.N;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n_blk + (wn * TN + j) * 32 + li;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ml = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int h = h0 + (ml >> 5), w = w0 + ((ml >> 3) & 3), d = d0 + (ml & 7);
          if (n < a.N && h < a.H && w < a.W && d < a.D)
            pz[((((long long)b * a.H + h) * a.W + w) * a.D + d) * a.N + n] = acc[i][j][r];
        }
    }
    return;
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

// Few-channel variant (C <= 32, N <= 32: the two finest U-Net levels, 0.25-1 M voxels).  There the work per brick is 27-54
// MFMAs per wave and the kernel above is bound by workgroup turnover (8192 short-lived workgroups, each exposing one halo
// round trip and re-streaming the same 27 weight tiles).  Here the whole weight tensor [27][32][CC] stays in LDS, a workgroup
// is persistent over bricks and the next brick's halo is in flight (registers) during the current MFMAs and epilogue.
// Unpadded LDS rows (2 workgroups per CU), bank-spread by XOR (see the kernel).
template <int CC>
__global__ void __launch_bounds__(256) conv3_halo_ws_bf16_kernel(const HaloArgs a, int bricks) {
  constexpr int VPV = CC / 8, KS = CC / 16, LDC = 40;
  constexpr int NH = (HALO_VOX * VPV + 255) / 256, NWV = (27 * 32 * VPV + 255) / 256;
  // LDS images (bank analysis: tools/lds_conflicts.py).  Halo voxel (hh, hw, hd) at row (hh * HALO_W + hw) * PD + hd with its
  // 16-byte parts XOR-ed by hw; weight row (tap, n) with its parts XOR-ed by n >> 2 (64-byte rows) or n >> 3 (32-byte rows).
  // The images first written here (plain 32-byte voxels, 64-byte voxels XOR-ed by one bit of the row index) put the 16 lanes
  // a ds_read_b128 serves together on the same 16-byte columns four times over for the halo and twice for the weights.
  constexpr int PD = CC == 16 ? 12 : HALO_D;
  constexpr int HROWS = HALO_H * HALO_W * PD;
  __shared__ __attribute__((aligned(16))) uint16_t halo[HROWS * CC];        // also the 128 x LDC output staging
  __shared__ __attribute__((aligned(16))) uint16_t Wl[27 * 32 * CC];
  static_assert(HROWS * CC >= 128 * LDC, "staging fits");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int nbh = (a.H + 3) / 4, nbw = (a.W + 3) / 4, nbd = (a.D + 7) / 8;

  // weights once: row (tap, n) <- w[n][tap (mirrored for the data gradient)][0..C).  All loads first (unconditional, from a clamped
  // address; padding is selected to zero afterwards), then the LDS stores: with a test around each load hipcc waits for every
  // one where it stands - 14 dependent L2 round trips at the head of every workgroup.
  {
    uint4 wv[NWV];
#pragma unroll
    for (int p = 0; p < NWV; ++p) {
      const int idx = min(tid + p * 256, 27 * 32 * VPV - 1);
      const int part = idx % VPV, row = idx / VPV;
      const int n = min(row & 31, a.N - 1), tap = row >> 5;
      const int c = min(part * 8, a.C - 8);
      wv[p] = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(a.w) + ((long long)n * 27 + tap) * a.C + c);
    }
#pragma unroll
    for (int p = 0; p < NWV; ++p) {
      const int idx = tid + p * 256;
      if (idx < 27 * 32 * VPV) {
        const int part = idx % VPV, row = idx / VPV;
        const int n = row & 31;
        const uint4 v = (n < a.N && part * 8 < a.C) ? wv[p] : make_uint4(0u, 0u, 0u, 0u);
        const int slot = part ^ (CC == 32 ? ((row >> 2) & 3) : ((row >> 3) & 1));
        *reinterpret_cast<uint4*>(&Wl[row * CC + slot * 8]) = v;
      }
    }
  }

  // Index arithmetic is hoisted out of the brick loop: a thread always stages the same halo pieces and stores the same output
  // pieces relative to the brick origin, and the brick coordinates advance incrementally (a first version re-derived all of
  // it per brick: ~900 scalar / vector ALU instructions around 27 MFMAs).
  struct BrickPos { int b, bh, bw, bd; };
  auto decompose = [&](int brick) {
    BrickPos q;
    int t = brick;
    q.bd = t % nbd; t /= nbd;
    q.bw = t % nbw; t /= nbw;
    q.bh = t % nbh;
    q.b = t / nbh;
    return q;
  };
  // brick order: a contiguous run per workgroup, adjacent runs on one XCD (see conv3_halo_wr_bf16_kernel)
  const bool xcd_order = (gridDim.x & 7) == 0 && !a.no_xcd_order;
  const int slot = xcd_order ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int per = (bricks + (int)gridDim.x - 1) / (int)gridDim.x;
  const int brick_first = xcd_order ? slot * per : (int)blockIdx.x;
  const int brick_step = xcd_order ? 1 : (int)gridDim.x;
  const int brick_end = xcd_order ? min(bricks, brick_first + per) : bricks;
  const BrickPos stepd = decompose(brick_step);
  auto advance = [&](BrickPos& q) {
    q.bd += stepd.bd; if (q.bd >= nbd) { q.bd -= nbd; ++q.bw; }
    q.bw += stepd.bw; if (q.bw >= nbw) { q.bw -= nbw; ++q.bh; }
    q.bh += stepd.bh; if (q.bh >= nbh) { q.bh -= nbh; ++q.b; }
    q.b += stepd.b;
  };
  int p_hh[NH], p_hw[NH], p_hd[NH], p_ld[NH];
  long long p_rel[NH];
  const uint16_t* p_src[NH];
  bool p_ok[NH];
#pragma unroll
  for (int p = 0; p < NH; ++p) {
    const int idx = tid + p * 256;
    const int hv = idx / VPV, part = idx - hv * VPV;
    p_hd[p] = hv % HALO_D; p_hw[p] = (hv / HALO_D) % HALO_W; p_hh[p] = hv / (HALO_D * HALO_W);
    const int c = part * 8;
    p_ok[p] = idx < HALO_VOX * VPV && c < a.C;
    p_src[p] = c < a.c0 ? reinterpret_cast<const uint16_t*>(a.x0) + c : reinterpret_cast<const uint16_t*>(a.x1) + (c - a.c0);
    p_ld[p] = c < a.c0 ? a.lda0 : a.lda1;
    p_rel[p] = ((long long)(p_hh[p] - 1) * a.W + (p_hw[p] - 1)) * a.D + (p_hd[p] - 1);
  }
  uint4 hreg[NH];
  auto load_halo = [&](const BrickPos& q) {
    const int h0 = q.bh * 4, w0 = q.bw * 4, d0 = q.bd * 8;
    const long long vox0 = (((long long)q.b * a.H + h0) * a.W + w0) * a.D + d0;
#pragma unroll
    for (int p = 0; p < NH; ++p) {
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      const int h = h0 - 1 + p_hh[p], w = w0 - 1 + p_hw[p], d = d0 - 1 + p_hd[p];
      if (p_ok[p] && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W && (unsigned)d < (unsigned)a.D)
        v = *reinterpret_cast<const uint4*>(p_src[p] + (vox0 + p_rel[p]) * p_ld[p]);
      hreg[p] = v;
    }
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int p = 0; p < NH; ++p) {
      const int idx = tid + p * 256;
      if (idx < HALO_VOX * VPV) {
        const int part = idx % VPV;
        const int row = (p_hh[p] * HALO_W + p_hw[p]) * PD + p_hd[p];
        const int slot = part ^ (p_hw[p] & (VPV - 1));
        *reinterpret_cast<uint4*>(&halo[row * CC + slot * 8]) = hreg[p];
      }
    }
  };

  const int hv0 = (wave * HALO_W + (li >> 3)) * PD + (li & 7);          // halo row of this lane's voxel at tap (0,0,0)
  const int wsw = CC == 32 ? ((li >> 2) & 3) : ((li >> 3) & 1);
  float4 bv4[4];                           // bias of this lane's 4 x 4 consecutive output channels
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int n = 8 * rr + 4 * lh;
    bv4[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias != nullptr) {               // uniform test; the 4 loads inside are unconditional (clamped index, selected afterwards)
      const float b0 = a.bias[min(n + 0, a.N - 1)], b1 = a.bias[min(n + 1, a.N - 1)], b2 = a.bias[min(n + 2, a.N - 1)],
                  b3 = a.bias[min(n + 3, a.N - 1)];
      bv4[rr] = make_float4(n + 0 < a.N ? b0 : 0.f, n + 1 < a.N ? b1 : 0.f, n + 2 < a.N ? b2 : 0.f, n + 3 < a.N ? b3 : 0.f);
    }
  }
  uint16_t* Cs = halo;

  int brick = brick_first;
  BrickPos cur = decompose(min(brick, bricks - 1)), nxt = cur;
  if (brick < brick_end) load_halo(cur);
  // output pieces of this thread: rows (tid >> 3) + 32 it, 4 channels from (tid & 7) * 4
  const int o_w = tid >> 6, o_d = (tid >> 3) & 7, o_n = (tid & 7) * 4;
  uint16_t* const o_base = o_n < a.n0 ? reinterpret_cast<uint16_t*>(a.o0) + o_n : reinterpret_cast<uint16_t*>(a.o1) + (o_n - a.n0);
  const int o_ld = o_n < a.n0 ? a.ldo0 : a.ldo1;
  // The output stores of a brick are issued AFTER the next brick's halo registers have gone to LDS (a wait for the halo loads
  // placed behind freshly issued stores would also wait for their acknowledgement: vmcnt is one in-order counter).  Deferring
  // them by a whole brick was tried as well: 59 -> 55 us at C = 16 but 78 -> 92 us at C = 32; not kept.
  __syncthreads();                         // the weights are in place
  if (brick < brick_end) store_halo();
  __syncthreads();
  for (; brick < brick_end; brick += brick_step) {
    advance(nxt);
    const bool more = brick + brick_step < brick_end;
    if (more) load_halo(nxt);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // taps in 9 groups of 3 (one d-run); all fragments of a group are fetched before its MFMAs are issued (pinned with
    // sched_barrier: left alone, hipcc sinks each ds_read next to its MFMA, which then waits an LDS round trip).
    // Transposed product D[n][voxel]: a lane then owns 4 consecutive channels of one voxel (8-byte staging writes).
    auto load_group = [&](int gidx, bf16x8 (&av)[3][KS], bf16x8 (&bv)[3][KS]) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int tap = gidx * 3 + q;
        const int ts = a.flip ? 26 - tap : tap;      // data gradient: tap t reads the mirrored halo offset
        const int th = ts / 9, tw = (ts / 3) % 3, td = ts % 3;
        const int hv = hv0 + (th * HALO_W + tw) * PD + td;
        const int sw = ((li >> 3) + tw) & (VPV - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int ca = ((ks * 2 + lh) ^ sw) << 3;
          const int cb = ((ks * 2 + lh) ^ wsw) << 3;
          av[q][ks] = *reinterpret_cast<const bf16x8*>(&halo[hv * CC + ca]);
          bv[q][ks] = *reinterpret_cast<const bf16x8*>(&Wl[(tap * 32 + li) * CC + cb]);
        }
      }
    };
    auto mma_group = [&](const bf16x8 (&av)[3][KS], const bf16x8 (&bv)[3][KS]) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[q][ks], av[q][ks], acc, 0, 0, 0);
    };
    // one fragment set per group (the SIMD's second wave covers the read latency; a second set would cost that occupancy)
#pragma unroll
    for (int gi = 0; gi < 9; ++gi) {
      bf16x8 avA[3][KS], bvA[3][KS];
      load_group(gi, avA, bvA);
      __builtin_amdgcn_sched_barrier(0);
      mma_group(avA, bvA);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                       // every wave is done with the halo: it becomes the output staging
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      uint2 pk;
      pk.x = pack_bf16x2(acc[4 * rr + 0] + bv4[rr].x, acc[4 * rr + 1] + bv4[rr].y);
      pk.y = pack_bf16x2(acc[4 * rr + 2] + bv4[rr].z, acc[4 * rr + 3] + bv4[rr].w);
      *reinterpret_cast<uint2*>(&Cs[(wave * 32 + li) * LDC + 8 * rr + 4 * lh]) = pk;
    }
    __syncthreads();
    uint2 ov[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) ov[it] = *reinterpret_cast<const uint2*>(&Cs[((tid >> 3) + 32 * it) * LDC + o_n]);
    __syncthreads();                       // the staging has been read: the region takes the next halo
    if (more) store_halo();
    {
      const int h0 = cur.bh * 4, w = cur.bw * 4 + o_w, d = cur.bd * 8 + o_d;
      const long long vox0 = (((long long)cur.b * a.H + h0) * a.W + w) * a.D + d;
      const bool ok = o_n < a.N && w < a.W && d < a.D;
#pragma unroll
      for (int it = 0; it < 4; ++it) {     // row ml = (tid >> 3) + 32 it is brick voxel (h = it, w = o_w, d = o_d)
        if (!ok || h0 + it >= a.H) continue;
        *reinterpret_cast<uint2*>(o_base + (vox0 + (long long)it * a.W * a.D) * o_ld) = ov[it];
      }
    }
    __syncthreads();                       // the next halo is in place
    cur = nxt;
  }
}

// The same with the weights in REGISTERS (27 x CC/16 operand fragments per wave: 108 / 216 VGPRs): one LDS read per MFMA is
// left.  In-kernel clock stamps of the first version of this loop (4 barriers per brick, output staging shared by the four waves,
// addresses re-derived per brick with 64-bit per-lane multiplies) showed 1 270 of a brick's 4 100 cycles in the 27 MFMAs, 1 500 in
// ISSUING three loads and four stores (address arithmetic), the rest in barriers and staging.  This version:
//   * per-thread pointers are loop invariants; a brick adds one scalar offset (two with a concatenated input / split output);
//   * the halo is double-buffered in LDS and the output staging is private to a wave (a wave's 32 voxels are one h plane of the
//     brick), so ONE barrier per brick is left;
//   * the halo of brick i + 2 is requested at the end of brick i, in front of that brick's output stores (vmcnt is one in-order
//     counter: a wait for loads placed behind stores also waits for their acknowledgement), and goes to LDS at the end of brick
//     i + 1; the loads are unconditional (clamped address, selected when written to LDS: a test around a load gives it a basic
//     block and a wait of its own);
//   * LDS image of the halo: voxel (hh, hw, hd) at row (hh * HALO_W + hw) * PD + hd, its 16-byte parts XOR-ed with the low bits
//     of hw.  ds_read_b128 serves 16 lanes at a time ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32) from 64
//     banks: a lane's voxel is (w = li >> 3, d = li & 7) + tap, so a group touches four w rows; with the plain [360][CC] image
//     those land on the same 16-byte columns 4 times over (tools/lds_conflicts.py) and every fragment read took four LDS
//     passes.  64-byte voxels: XOR with hw & 3; 32-byte voxels: d pitch 12 and XOR with hw & 1.
template <int CC>
__global__ void __launch_bounds__(256) conv3_halo_wr_bf16_kernel(const HaloArgs a, int bricks) {
  constexpr int VPV = CC / 8, KS = CC / 16, LDC = 40;
  constexpr int NH = (HALO_VOX * VPV + 255) / 256;
  constexpr int PD = CC == 16 ? 12 : HALO_D;
  constexpr int HROWS = HALO_H * HALO_W * PD, HELEMS = HROWS * CC, SELEMS = 32 * LDC;
  __shared__ __attribute__((aligned(16))) uint16_t smem[2 * HELEMS + 4 * SELEMS];     // halo[2] | staging[4 waves]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int nbh = (a.H + 3) / 4, nbw = (a.W + 3) / 4, nbd = (a.D + 7) / 8;

  // weights once, straight into operand registers: lane (n = li, lh) holds channels ks*16 + lh*8 .. +7 of tap t (mirrored taps for
  // the data gradient are handled by the halo offset, as above)
  bf16x8 wreg[27][KS];
  {
    const int n = min(li, a.N - 1);
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int c = min(ks * 16 + lh * 8, a.C - 8);
        const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(a.w) + ((long long)n * 27 + t) * a.C + c);
        const bool ok = li < a.N && ks * 16 + lh * 8 < a.C;
        wreg[t][ks] = __builtin_bit_cast(bf16x8, ok ? v : make_uint4(0u, 0u, 0u, 0u));
      }
  }

  struct BrickPos { int b, bh, bw, bd; };
  auto decompose = [&](int brick) {
    BrickPos q;
    int t = brick;
    q.bd = t % nbd; t /= nbd;
    q.bw = t % nbw; t /= nbw;
    q.bh = t % nbh;
    q.b = t / nbh;
    return q;
  };
  // Brick order: workgroup -> a CONTIGUOUS run of bricks (consecutive bricks along d share a halo face), and the runs of the 64
  // workgroups that land on one XCD (linear id % 8 under round-robin dispatch) are adjacent, so the in-plane halo overlap of
  // neighbouring rows is served by that XCD's L2.  Strided (brick = id + k * grid) every brick fetched its whole 2.8x halo through
  // the fabric: no two bricks in flight on an XCD were neighbours.
  const bool xcd_order = (gridDim.x & 7) == 0 && !a.no_xcd_order;
  const int slot = xcd_order ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int per = (bricks + (int)gridDim.x - 1) / (int)gridDim.x;
  const int brick_first = xcd_order ? slot * per : (int)blockIdx.x;
  const int brick_step = xcd_order ? 1 : (int)gridDim.x;
  const int brick_end = xcd_order ? min(bricks, brick_first + per) : bricks;
  const BrickPos stepd = decompose(brick_step);
  auto advance = [&](BrickPos& q) {
    q.bd += stepd.bd; if (q.bd >= nbd) { q.bd -= nbd; ++q.bw; }
    q.bw += stepd.bw; if (q.bw >= nbw) { q.bw -= nbw; ++q.bh; }
    q.bh += stepd.bh; if (q.bh >= nbh) { q.bh -= nbh; ++q.b; }
    q.b += stepd.b;
  };
  auto origin = [&](const BrickPos& q) { return (((long long)q.b * a.H + q.bh * 4) * a.W + q.bw * 4) * a.D + q.bd * 8; };

  // ---- halo pieces of this thread (loop invariants): source pointer at the brick origin's offset, LDS slot, halo coordinates
  const char* h_ptr[NH];
  int h_lds[NH], h_pos[NH];                // h_pos: hh | hw << 8 | hd << 16 | source 1 << 24 | exists << 25
#pragma unroll
  for (int p = 0; p < NH; ++p) {
    const int idx = tid + p * 256;
    const int hv = idx / VPV, part = idx - hv * VPV;
    const int hd = hv % HALO_D, hw = (hv / HALO_D) % HALO_W, hh = hv / (HALO_D * HALO_W);
    const int c = part * 8;
    const bool ok = idx < HALO_VOX * VPV && c < a.C;
    const bool s1 = ok && c >= a.c0;
    const uint16_t* src = s1 ? reinterpret_cast<const uint16_t*>(a.x1) + (c - a.c0) : reinterpret_cast<const uint16_t*>(a.x0) + (ok ? c : 0);
    const long long rel = ((long long)(hh - 1) * a.W + (hw - 1)) * a.D + (hd - 1);
    h_ptr[p] = reinterpret_cast<const char*>(src + rel * (s1 ? a.lda1 : a.lda0));
    h_lds[p] = ((hh * HALO_W + hw) * PD + hd) * CC + ((part ^ (hw & (VPV - 1))) << 3);
    h_pos[p] = hh | (hw << 8) | (hd << 16) | ((s1 ? 1 : 0) << 24) | ((ok ? 1 : 0) << 25);
  }
  uint4 hreg[NH];
  bool hin[NH];                            // piece inside the volume (decided at request time, applied when it goes to LDS)
  auto load_halo = [&](const BrickPos& q) {
    const long long vox0 = origin(q);
    const long long sb0 = vox0 * a.lda0 * 2, sb1 = vox0 * a.lda1 * 2;       // scalar byte offsets of the brick origin
    const int h0 = q.bh * 4 - 1, w0 = q.bw * 4 - 1, d0 = q.bd * 8 - 1;
#pragma unroll
    for (int p = 0; p < NH; ++p) {
      const int h = h0 + (h_pos[p] & 255), w = w0 + ((h_pos[p] >> 8) & 255), d = d0 + ((h_pos[p] >> 16) & 255);
      const bool in = (h_pos[p] >> 25) != 0 && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W && (unsigned)d < (unsigned)a.D;
      const char* ad = h_ptr[p] + (((h_pos[p] >> 24) & 1) ? sb1 : sb0);
      ad = in ? ad : reinterpret_cast<const char*>(a.x0);
      hreg[p] = *reinterpret_cast<const uint4*>(ad);
      hin[p] = in;
    }
  };
  auto store_halo = [&](uint16_t* dst) {
#pragma unroll
    for (int p = 0; p < NH; ++p)
      if (tid + p * 256 < HALO_VOX * VPV)
        *reinterpret_cast<uint4*>(&dst[h_lds[p]]) = hin[p] ? hreg[p] : make_uint4(0u, 0u, 0u, 0u);
  };

  // ---- output pieces of this thread: the wave's 32 voxels (h = wave) x N/4 pieces of 4 channels, dealt linearly to the lanes
  const int npr = a.N >> 2;                // pieces per voxel (N <= 32: at most 8)
  char* o_ptr[4];
  int o_lds[4], o_pos[4];                  // o_pos: w | d << 8 | destination 1 << 16 | exists << 17
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int q = lane + 64 * it;
    const int row = q / npr, pc = q - row * npr;
    const bool ok = row < 32;
    const int on = pc * 4, rw = (row >> 3) & 3, rd = row & 7;
    const bool d1 = on >= a.n0;
    uint16_t* base = d1 ? reinterpret_cast<uint16_t*>(a.o1) + (on - a.n0) : reinterpret_cast<uint16_t*>(a.o0) + on;
    const long long rel = ((long long)wave * a.W + rw) * a.D + rd;
    o_ptr[it] = reinterpret_cast<char*>(base + rel * (d1 ? a.ldo1 : a.ldo0));
    o_lds[it] = (row & 31) * LDC + on;
    o_pos[it] = rw | (rd << 8) | ((d1 ? 1 : 0) << 16) | ((ok ? 1 : 0) << 17);
  }

  const int hv0 = (wave * HALO_W + (li >> 3)) * PD + (li & 7);          // halo row of this lane's voxel at tap (0,0,0)
  float4 bv4[4];                           // bias of this lane's 4 x 4 consecutive output channels
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int n = 8 * rr + 4 * lh;
    bv4[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias != nullptr) {               // uniform test; the 4 loads inside are unconditional (clamped index, selected afterwards)
      const float b0 = a.bias[min(n + 0, a.N - 1)], b1 = a.bias[min(n + 1, a.N - 1)], b2 = a.bias[min(n + 2, a.N - 1)],
                  b3 = a.bias[min(n + 3, a.N - 1)];
      bv4[rr] = make_float4(n + 0 < a.N ? b0 : 0.f, n + 1 < a.N ? b1 : 0.f, n + 2 < a.N ? b2 : 0.f, n + 3 < a.N ? b3 : 0.f);
    }
  }
  uint16_t* const Cs = smem + 2 * HELEMS + wave * SELEMS;

  int brick = brick_first;
  if (brick >= brick_end) return;
  BrickPos cur = decompose(brick), nxt = cur;
  load_halo(cur);
  store_halo(smem);
  advance(nxt);
  if (brick + brick_step < brick_end) load_halo(nxt);
  __syncthreads();
  int buf = 0;
  for (; brick < brick_end; brick += brick_step) {
    const bool more = brick + brick_step < brick_end;
    const uint16_t* const halo = smem + buf * HELEMS;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // taps in 9 groups of 3 (one d-run); all fragments of a group are fetched before its MFMAs are issued (pinned with
    // sched_barrier: left alone, hipcc sinks each ds_read next to its MFMA, which then waits an LDS round trip).
    // Transposed product D[n][voxel]: a lane then owns 4 consecutive channels of one voxel (8-byte staging writes).
    auto load_group = [&](int gidx, bf16x8 (&av)[3][KS]) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int tap = gidx * 3 + q;
        const int ts = a.flip ? 26 - tap : tap;      // data gradient: tap t reads the mirrored halo offset
        const int th = ts / 9, tw = (ts / 3) % 3, td = ts % 3;
        const int hv = hv0 + (th * HALO_W + tw) * PD + td;
        const int sw = ((li >> 3) + tw) & (VPV - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int ca = ((ks * 2 + lh) ^ sw) << 3;
          av[q][ks] = *reinterpret_cast<const bf16x8*>(&halo[hv * CC + ca]);
        }
      }
    };
    auto mma_group = [&](int gidx, const bf16x8 (&av)[3][KS]) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[gidx * 3 + q][ks], av[q][ks], acc, 0, 0, 0);
    };
    // one fragment set per group (a second set, the reads of group g+1 in flight during the MFMAs of group g, was slower:
    // 48 -> 79 us at C = 16)
#pragma unroll
    for (int gi = 0; gi < 9; ++gi) {
      bf16x8 avA[3][KS];
      load_group(gi, avA);
      __builtin_amdgcn_sched_barrier(0);
      mma_group(gi, avA);
      __builtin_amdgcn_sched_barrier(0);
    }
    // wave-private staging: the LDS executes one wave's instructions in order, so no barrier between these writes and reads
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      uint2 pk;
      pk.x = pack_bf16x2(acc[4 * rr + 0] + bv4[rr].x, acc[4 * rr + 1] + bv4[rr].y);
      pk.y = pack_bf16x2(acc[4 * rr + 2] + bv4[rr].z, acc[4 * rr + 3] + bv4[rr].w);
      *reinterpret_cast<uint2*>(&Cs[li * LDC + 8 * rr + 4 * lh]) = pk;
    }
    uint2 ov[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) ov[it] = *reinterpret_cast<const uint2*>(&Cs[o_lds[it]]);
    const BrickPos done = cur;
    cur = nxt;
    if (more) {
      store_halo(smem + (buf ^ 1) * HELEMS);
      advance(nxt);
      if (brick + 2 * brick_step < brick_end) load_halo(nxt);
    }
    {
      const long long vox0 = origin(done);
      const long long so0 = vox0 * a.ldo0 * 2, so1 = vox0 * a.ldo1 * 2;
      const int w0 = done.bw * 4, d0 = done.bd * 8;
      const bool hok = done.bh * 4 + wave < a.H;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const bool ok = hok && (o_pos[it] >> 17) != 0 && w0 + (o_pos[it] & 255) < a.W && d0 + ((o_pos[it] >> 8) & 255) < a.D;
        if (ok) *reinterpret_cast<uint2*>(o_ptr[it] + (((o_pos[it] >> 16) & 1) ? so1 : so0)) = ov[it];
      }
    }
    __syncthreads();                       // the next halo is in place; everybody has left the current one
    buf ^= 1;
  }
}

// returns LTU_OK after launching, or 1 when the shape is not handled here (the caller falls back to the implicit GEMM)
// out[v][n] = bf16(sum_z part[z][v][n] + bias[n]); 4 columns per thread (N, n0 are multiples of 4)
__global__ void __launch_bounds__(256) conv_halo_fold_kernel(const HaloArgs a, long long M) {
  const int nq = a.N / 4;
  const long long total = M * nq;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const long long v = t / nq;
    const int n = (int)(t - v * nq) * 4;
    float4 acc = a.bias != nullptr ? make_float4(a.bias[n], a.bias[n + 1], a.bias[n + 2], a.bias[n + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < a.ksplit; ++z) {
      const float4 q = *reinterpret_cast<const float4*>(a.part + ((long long)z * M + v) * a.N + n);
      acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w;
    }
    uint2 pk;
    pk.x = pack_bf16x2(acc.x, acc.y);
    pk.y = pack_bf16x2(acc.z, acc.w);
    uint16_t* dst = n < a.n0 ? reinterpret_cast<uint16_t*>(a.o0) + v * a.ldo0 + n
                             : reinterpret_cast<uint16_t*>(a.o1) + v * a.ldo1 + (n - a.n0);
    *reinterpret_cast<uint2*>(dst) = pk;
  }
}

// split geometry: only for grids that leave most of the chip idle
static int halo_split(long long bricks, int N, int C, int CC, int* cps) {
  if (N <= 32 && C <= 32) return 1;                      // weight-stationary kernel
  const int ntn = (N > 64 && bricks * cdiv(N, 128) >= 256) ? cdiv(N, 128) : (N > 32 ? cdiv(N, 64) : 1);
  const long long blocks = bricks * ntn;
  const int nchunk = (C + CC - 1) / CC;
  if (blocks >= ltu_knob_pos("LTU_HALO_SPLIT_BELOW", 200) || nchunk < 2) return 1;
  int want = (int)((512 + blocks - 1) / blocks);
  if (want > 8) want = 8;
  if (want > nchunk) want = nchunk;
  if (want < 2) return 1;
  *cps = (nchunk + want - 1) / want;
  return (nchunk + *cps - 1) / *cps;
}
static int halo_cc(int C, int c0) { return (C % 32 == 0 && c0 % 32 == 0) ? 32 : 16; }
long long conv_halo_ws_floats(int B, int H, int W, int D, int C, int N) {
  const long long bricks = (long long)B * ((H + 3) / 4) * ((W + 3) / 4) * ((D + 7) / 8);
  int cps = 0;
  if (conv_ring_splits(B, H, W, D, C, N)) return 8LL * B * H * W * D * N;
  const int ks = halo_split(bricks, N, C, halo_cc(C, C), &cps);
  // c0 may lower CC to 16 (more chunks, never more than 8 splits): bound by 8
  return (ks > 1 || halo_split(bricks, N, C, 16, &cps) > 1) ? 8LL * B * H * W * D * N : 0;
}

int launch_conv_halo_bf16(HaloArgs a, hipStream_t st) {
  if (a.C % 8 || a.c0 % 8 || a.lda0 % 8 || a.lda1 % 8 || a.N % 4 || a.n0 % 4 || a.ldo0 % 4 || a.ldo1 % 4) return 1;
  if (a.H < 2 || a.W < 2 || a.D < 4) return 1;
  a.CC = (a.C % 32 == 0 && a.c0 % 32 == 0) ? 32 : 16;
  if (a.C % 16 != 0 && a.C != 8) return 1;
  if (a.c0 % a.CC != 0 && a.c0 != a.C) return 1;
  const long long bricks = (long long)a.B * ((a.H + 3) / 4) * ((a.W + 3) / 4) * ((a.D + 7) / 8);
  if (bricks >= (1LL << 31)) return 1;
#ifdef LTU_EXPERIMENTS
  a.no_xcd_order = ltu_knob("LTU_HALO_NO_XCD", 0);
#else
  a.no_xcd_order = 0;
#endif
  if (a.C == 32 && a.N <= 32 && !ltu_knob("LTU_NO_FC_RING", 0)) {       // second generation (conv_fc_ring.hip)
    const int hr = launch_conv_fc_ring_bf16(a, st);
    if (hr != 1) return hr;
  }
  if (a.C == 16 && a.N <= 32 && !ltu_knob("LTU_NO_C16_RING", 0)) {      // second generation at 16 channels (conv_c16_ring.hip)
    const int hr = launch_conv_c16_ring_bf16(a, st);
    if (hr != 1) return hr;
  }
  if (a.N <= 32 && a.C <= 32 && !ltu_knob("LTU_NO_HALO_WS", 0)) {      // few channels: weights stationary, persistent over bricks
    int wsb = -1;
    wsb = ltu_knob_pos("LTU_HALO_WS_BLOCKS", 512);
    const unsigned nblk = (unsigned)(bricks < wsb ? bricks : wsb);
    // weights in registers (one LDS operand per MFMA): 59 -> 49 us at C = 16 (236 VGPRs, two workgroups per CU); at C = 32 the
    // 216 weight registers leave one wave per SIMD and the gain is lost (80 vs 78 us), so the LDS-weights kernel stays there
    if (ltu_knob("LTU_HALO_WR", a.C <= 16 ? 1 : 0)) {
      const int wrb = ltu_knob_pos("LTU_HALO_WR_BLOCKS", a.C > 16 ? 256 : 512);
      const unsigned nb = (unsigned)(bricks < wrb ? bricks : wrb);
      if (a.C > 16) hipLaunchKernelGGL((conv3_halo_wr_bf16_kernel<32>), dim3(nb), dim3(256), 0, st, a, (int)bricks);
      else hipLaunchKernelGGL((conv3_halo_wr_bf16_kernel<16>), dim3(nb), dim3(256), 0, st, a, (int)bricks);
      return ltu_check_launch();
    }
    if (a.C > 16) hipLaunchKernelGGL((conv3_halo_ws_bf16_kernel<32>), dim3(nblk), dim3(256), 0, st, a, (int)bricks);
    else hipLaunchKernelGGL((conv3_halo_ws_bf16_kernel<16>), dim3(nblk), dim3(256), 0, st, a, (int)bricks);
    return ltu_check_launch();
  }
  if ((a.C >= 64 || a.N > 32) && !ltu_knob("LTU_NO_CONV_RING", 0)) {     // second generation for grids that fill the machine (conv_ring.hip)
    a.ksplit = 1;
    const int hr = launch_conv_ring_bf16(a, st);
    if (hr == LTU_OK && a.ksplit > 1) {      // small grid: the channel chunks were split over workgroups
      const long long M = (long long)a.B * a.H * a.W * a.D;
      long long blocks = (M * (a.N / 4) + 255) / 256;
      if (blocks > 2048) blocks = 2048;
      hipLaunchKernelGGL(conv_halo_fold_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, M);
      return ltu_check_launch();
    }
    if (hr != 1) return hr;
  }
  int cps = 0;
  a.ksplit = a.part != nullptr && !ltu_knob("LTU_NO_HALO_SPLIT", 0) ? halo_split(bricks, a.N, a.C, a.CC, &cps) : 1;
  a.cps = cps;
  if (a.ksplit < 2) a.part = nullptr;
  else if ((long long)a.ksplit * a.B * a.H * a.W * a.D * a.N > a.part_floats) return LTU_E_ARG;      // the workspace is shorter than this split needs
  const unsigned gz = (unsigned)a.ksplit;
  // TS = taps per weight stage.  A stage's weights are requested one stage ahead and every stage ends in a barrier: with 3 taps
  // (12 MFMAs per wave) a stage is shorter than the L2 round trip of the next one's weights.  9 taps per stage (3 for the 128-column
  // tile; 115 / 84 KB of LDS, dynamic) cover it - worth 11-14 % where a workgroup walks at least four channel chunks
  // (16x16x64, 128 -> 64: 32.3 -> 28.7 us; 64+64 -> 64: 28.6 -> 24.8); with fewer chunks the larger first stage is exposed instead
  // (8x8x64, 256 -> 128 split over the chunks: 30.3 -> 36.3; data gradient of 32+32 -> 32: 50 -> 76), so those keep the short stages.
  const int chunks_per_wg = a.ksplit > 1 ? a.cps : (a.C + a.CC - 1) / a.CC;
  const bool deep = ltu_knob("LTU_HALO_DEEP", chunks_per_wg >= 4 ? 1 : 0) != 0;
#define HALO_LAUNCH1(WM, WN, TM, TN, TS, CCV)                                                                        \
  do {                                                                                                               \
    auto kern = &conv3_halo_bf16_kernel<WM, WN, TM, TN, TS, CCV>;                                                    \
    constexpr int bytes = halo_smem_bytes<WN, TN, TS>();                                                            \
    static LtuDevOnce once;                                                                                          \
    if (once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); \
    hipLaunchKernelGGL(kern, grid, dim3(256), bytes, st, a);                                                         \
  } while (0)
#define HALO_LAUNCH(WM, WN, TM, TN, TS)                     \
  do {                                                      \
    if (a.CC == 32) HALO_LAUNCH1(WM, WN, TM, TN, TS, 32);   \
    else HALO_LAUNCH1(WM, WN, TM, TN, TS, 16);              \
  } while (0)
  if (a.N > 64 && bricks * cdiv(a.N, 128) >= 256) {
    dim3 grid((unsigned)bricks, cdiv(a.N, 128), gz);
    if (deep) HALO_LAUNCH(2, 2, 2, 2, 3);
    else HALO_LAUNCH(2, 2, 2, 2, 1);
  } else if (a.N > 32) {                    // also wide outputs on small grids: 64-column tiles double the workgroup count
    dim3 grid((unsigned)bricks, cdiv(a.N, 64), gz);
    if (deep) HALO_LAUNCH(4, 1, 1, 2, 9);
    else HALO_LAUNCH(4, 1, 1, 2, 3);
  } else {
    dim3 grid((unsigned)bricks, 1, gz);
    if (deep) HALO_LAUNCH(4, 1, 1, 1, 9);
    else HALO_LAUNCH(4, 1, 1, 1, 3);
  }
#undef HALO_LAUNCH
#undef HALO_LAUNCH1
  if (a.part != nullptr) {
    const long long M = (long long)a.B * a.H * a.W * a.D;
    long long blocks = (M * (a.N / 4) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(conv_halo_fold_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, M);
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
// Row pitch of both LDS images: 32 elements, unpadded.  A ds_read_b64_tr_b16 serves the two 32-lane halves separately; a half
// reads 4 consecutive rows x 64 bytes, which tile the 64 banks exactly at a 64-byte pitch (the 80-byte pitch of the forward
// kernel's first image made rows 0 and 3 overlap: a 2-way conflict on every operand read; tools/lds_conflicts.py).
#define LDGH 32
#define LDWH 32

// PACK (16-channel chunks): a column tile holds TWO taps x 16 channels - lanes with columns 16..31 read the halo at the second
// tap's offset - so a wave owns 4 tap pairs instead of 7 taps: 32 MFMAs per brick instead of 56, none of them on padding columns.
template <bool PACK>
__global__ void __launch_bounds__(256, PACK ? 3 : 2) conv3_wgrad_halo_bf16_kernel(const WHaloArgs a) {
  constexpr int NA = PACK ? 4 : 7;         // accumulators per wave; the last one of wave 3 is spare and takes the bias sums
  __shared__ __attribute__((aligned(16))) uint16_t halo[HALO_VOX * LDWH];
  __shared__ __attribute__((aligned(16))) uint16_t Gs[128 * LDGH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int chunk = blockIdx.x, n_blk = blockIdx.y * 32;
  const int nbh = (a.H + 3) / 4, nbw = (a.W + 3) / 4, nbd = (a.D + 7) / 8;
  const int VPV = a.CC / 8;
  const int brick_lo = blockIdx.z * a.bricks_per_split;
  int brick_hi = brick_lo + a.bricks_per_split;
  if (brick_hi > a.bricks) brick_hi = a.bricks;

  f32x16 acc[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  int tapoff[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    // PACK: pair wave + 4 i = taps 2 (wave + 4 i) (columns 0..15) and + 1 (columns 16..31; tap 27 does not exist: clamped, dropped)
    const int tap = PACK ? min(2 * (wave + 4 * i) + ((lane >> 4) & 1), 26) : min(wave + 4 * i, 26);
    tapoff[i] = (((tap / 9) * HALO_W + (tap / 3) % 3) * HALO_D + tap % 3) * LDWH;
  }
  // transposing-read lane geometry (see wgrad_tn_bf16_kernel): lane -> (row trow (+4), columns tcol..tcol+3) of a 16-row slab
  const int gq = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  const int tcol = 16 * (gq & 1) + 4 * tp;
  const int trow = 8 * (gq >> 1) + tq;
  const int gbase = trow * LDGH + tcol;
  // slab ks covers brick rows 16ks..16ks+15 = (h = ks>>1, w = 2(ks&1) + (row>>3), d = row&7)
  const int hbase = ((gq >> 1) * HALO_D + tq) * LDWH + (PACK ? 4 * tp : tcol);

  // Global loads are UNCONDITIONAL (clamped address; padding is selected to zero when the registers go to LDS): with a test around
  // each load hipcc gives it a basic block of its own and waits for the earlier ones there (the ISA of the first version shows
  // vmcnt(0) between the pieces of one brick), so the prefetch of the next brick was a chain of round trips in front of the MFMAs
  // instead of eight loads in flight behind them.
  uint4 hreg[6], greg[2];
  unsigned hin = 0, gin = 0;
  auto load_brick = [&](int brick) {
    int t = brick;
    const int bd = t % nbd; t /= nbd;
    const int bw = t % nbw; t /= nbw;
    const int bh = t % nbh;
    const int b = t / nbh;
    const int h0 = bh * 4, w0 = bw * 4, d0 = bd * 8;
    hin = 0; gin = 0;
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      const int idx = tid + p * 256;
      const int hv = idx / VPV, part = idx - hv * VPV;
      const int hd = hv % HALO_D, hw = (hv / HALO_D) % HALO_W, hh = hv / (HALO_D * HALO_W);
      const int h = h0 - 1 + hh, w = w0 - 1 + hw, d = d0 - 1 + hd;
      const int c = chunk * a.CC + part * 8;
      const bool in = idx < HALO_VOX * VPV && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W && (unsigned)d < (unsigned)a.D && c < a.C;
      const long long vox = in ? (((long long)b * a.H + h) * a.W + w) * a.D + d : 0;
      const int cc = in ? c : 0;
      const uint16_t* src = cc < a.c0 ? reinterpret_cast<const uint16_t*>(a.x0) + vox * a.lda0 + cc
                                      : reinterpret_cast<const uint16_t*>(a.x1) + vox * a.lda1 + (cc - a.c0);
      hreg[p] = *reinterpret_cast<const uint4*>(src);
      hin |= in ? 1u << p : 0u;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int idx = tid + p * 256;
      const int row = idx >> 2, n = n_blk + (idx & 3) * 8;
      const int h = h0 + (row >> 5), w = w0 + ((row >> 3) & 3), d = d0 + (row & 7);
      const bool in = h < a.H && w < a.W && d < a.D && n < a.N;
      const long long vox = in ? (((long long)b * a.H + h) * a.W + w) * a.D + d : 0;
      const int nn = in ? n : 0;
      const uint16_t* gp = (a.grad1 != nullptr && nn >= a.gn0) ? reinterpret_cast<const uint16_t*>(a.grad1) + vox * a.ldg1 + (nn - a.gn0)
                                                                : reinterpret_cast<const uint16_t*>(a.grad) + vox * a.ldg + nn;
      greg[p] = *reinterpret_cast<const uint4*>(gp);
      gin |= in ? 1u << p : 0u;
    }
  };
  auto store_brick = [&]() {
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      const int idx = tid + p * 256;
      if (idx < HALO_VOX * VPV) {
        const int hv = idx / VPV, part = idx - hv * VPV;
        *reinterpret_cast<uint4*>(&halo[hv * LDWH + part * 8]) = (hin >> p) & 1u ? hreg[p] : make_uint4(0u, 0u, 0u, 0u);
      }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int idx = tid + p * 256;
      *reinterpret_cast<uint4*>(&Gs[(idx >> 2) * LDGH + (idx & 3) * 8]) = (gin >> p) & 1u ? greg[p] : make_uint4(0u, 0u, 0u, 0u);
    }
  };

  // bias gradient = column sums of G: one more MFMA per 16-voxel slab against a tile of ones, by one wave of the first channel chunk
  // (the G fragment is in registers anyway).  It used to be 128 dependent LDS reads per brick by 32 lanes of that wave - as long
  // as the wave's 56 MFMAs, with the other three waves waiting at the barrier.
  // Wave 3 owns only 6 taps (3, 7, .., 23): its seventh accumulator is free and takes the sums - no extra registers (a separate
  // accumulator in wave 0 cost the fourth workgroup per CU: 60 -> 84 us).
  const bool do_bias = chunk == 0 && wave == 3;
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
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
      const int slab = ((ks >> 1) * HALO_W * HALO_D + (ks & 1) * 2 * HALO_D) * LDWH + hbase;
      // no test around the MFMAs (the wave index comes from threadIdx: a divergent branch per MFMA for the compiler, which also
      // stops it from requesting the operand reads ahead): a slot without a tap reads a clamped tap and is dropped in the epilogue;
      // wave 3's spare slot takes the tile of ones (bias sums) instead
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        union { struct { hs16x4 l, h; } s; bf16x8 v; } ub;
        const uint16_t* px = &halo[slab + tapoff[i]];
        ub.s.l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_hs16x4*)px);
        ub.s.h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_hs16x4*)(px + 4 * LDWH));
        if (i == NA - 1) ub.v = do_bias ? ones : ub.v;
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, acc[i], 0, 0, 0);
      }
    }
  }

  float* pz = a.part + (long long)blockIdx.z * a.npad * a.kpad;
  if (do_bias && li == 0) {                // every column of accb holds the sums: lanes 0 and 32 own the 2 x 16 rows
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n_blk + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (n < a.N) a.bpart[(long long)blockIdx.z * a.npad + n] = acc[NA - 1][r];
    }
  }
  const int c = PACK ? chunk * 16 + (li & 15) : chunk * a.CC + li;
  if ((!PACK && li >= a.CC) || c >= a.C) return;
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int tap = PACK ? 2 * (wave + 4 * i) + (li >> 4) : wave + 4 * i;
    if (tap >= 27 || wave + 4 * i >= (PACK ? 14 : 27)) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n_blk + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (n < a.N) pz[(long long)n * a.kpad + tap * a.C + c] = acc[i][r];
    }
  }
}

// workgroups of the halo weight gradient: 2 per CU; the packed 16-channel variant runs 3 per CU (168 registers)
static int whalo_blocks(int CC) {
  int v = -1;
  v = ltu_knob_pos("LTU_WHALO_BLOCKS", CC == 16 && !ltu_knob("LTU_WHALO_NO_PACK", 0) ? 768 : 512);
  return v;
}

static bool whalo_shape_ok(int C, int N) { return N % 8 == 0 && N <= 256 && (C % 16 == 0 || C == 8); }

long long conv_wgrad_halo_ws_floats(int N, int K) {
  if (K % 27) return 0;
  const int C = K / 27;
  if (!whalo_shape_ok(C, N)) return 0;
  const int CC = C % 32 == 0 ? 32 : 16;
  long long ns = whalo_blocks(CC) / ((long long)cdiv(C, CC) * cdiv(N, 32));
  if (ns < 1) ns = 1;
  return ns * N * ((long long)K + 1);
}

// fills part/bpart/npad/kpad/splits; returns LTU_OK after launching, or 1 when the shape is not handled
int launch_conv_wgrad_halo_bf16(WHaloArgs a, int* nsplit_out, hipStream_t st) {
  if (!whalo_shape_ok(a.C, a.N) || a.c0 % 8 || a.lda0 % 8 || a.lda1 % 8 || a.ldg % 8) return 1;
  if (a.grad1 != nullptr && (a.gn0 % 8 || a.ldg1 % 8)) return 1;
  if (a.H < 2 || a.W < 2 || a.D < 4) return 1;
  if (a.C % 32 == 0 && ltu_knob("LTU_WHALO_RING", 1)) {        // second generation (wgrad_halo_ring.hip)
    const int rc = launch_conv_wgrad_halo_ring_bf16(a, nsplit_out, st);
    if (rc != 1) return rc;
  }
  a.CC = a.C % 32 == 0 ? 32 : 16;
  const long long bricks = (long long)a.B * ((a.H + 3) / 4) * ((a.W + 3) / 4) * ((a.D + 7) / 8);
  if (bricks >= (1LL << 31)) return 1;
  const int nchunk = cdiv(a.C, a.CC), ntile = cdiv(a.N, 32);
  long long ns = whalo_blocks(a.CC) / ((long long)nchunk * ntile);
  if (ns < 1) ns = 1;
  if (ns > bricks) ns = bricks;
  a.bricks = (int)bricks;
  a.bricks_per_split = (int)((bricks + ns - 1) / ns);
  const int nsplit = (int)((bricks + a.bricks_per_split - 1) / a.bricks_per_split);
  a.npad = a.N;
  a.kpad = 27 * a.C;
  if ((long long)nsplit * a.npad * ((long long)a.kpad + 1) > a.part_floats) return LTU_E_ARG;
  a.bpart = a.part + (long long)nsplit * a.npad * a.kpad;
  *nsplit_out = nsplit;
  if (a.CC == 16 && !ltu_knob("LTU_WHALO_NO_PACK", 0))
    hipLaunchKernelGGL(conv3_wgrad_halo_bf16_kernel<true>, dim3(nchunk, ntile, nsplit), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(conv3_wgrad_halo_bf16_kernel<false>, dim3(nchunk, ntile, nsplit), dim3(256), 0, st, a);
  return ltu_check_launch();
}

// ---------------------------------------------------------------------------------------------------------------------
// "Class" convolutions: every output voxel o = m q + p of a finer grid (m = 1 or 2 per axis, p < m its parity class) is a
// small stencil over the COARSE grid around q.  Two layers of the network have this shape:
//   * nearest-x2 upsampling + 3x3x3 conv (the ROI token un-embedding): 8 classes x 8 taps (sub-pixel form, upconv.hip);
//   * the data gradient of a stride-2 (or 2,2,1) 3x3x3 conv: class p of an axis sees tap 1 (p = 0) or taps 0 and 2 (p = 1).
// As 8 separate implicit GEMMs the coarse tensor is gathered once per (class, tap): 27-64 times.  Here a workgroup owns a
// 4x4x8 brick of coarse voxels, stages its halo once per 32-channel chunk and produces ALL classes from it: the stream of
// (class, offset, weight tile) entries only moves an LDS offset and selects an accumulator.
template <int NC, int TN>
__global__ void __launch_bounds__(256) conv_class_halo_bf16_kernel(const ClassHaloArgs a) {
  constexpr int BN = 32 * TN, TS = 4;
  constexpr int HALO_ELEMS = HALO_VOX * 32, B_ELEMS = 2 * TS * BN * LDH, LDC = BN + 8, STAGE_ELEMS = 128 * LDC;
  constexpr int SMEM_ELEMS = (HALO_ELEMS + B_ELEMS) > STAGE_ELEMS ? (HALO_ELEMS + B_ELEMS) : STAGE_ELEMS;
  __shared__ __attribute__((aligned(16))) uint16_t smem[SMEM_ELEMS];
  uint16_t* halo = smem;                   // [HALO_VOX][32], 16-byte parts XOR-ed with hw & 3 (see conv3_halo_bf16_kernel)
  uint16_t* Bs = smem + HALO_ELEMS;        // [2][TS][BN][LDH]
  constexpr int LBV = TS * BN * 4 / 256;   // weight vectors per thread and stage

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int nbh = (a.H + 3) / 4, nbw = (a.W + 3) / 4, nbd = (a.D + 7) / 8;
  int bid = blockIdx.x;
  const int bd = bid % nbd; bid /= nbd;
  const int bw = bid % nbw; bid /= nbw;
  const int bh = bid % nbh;
  const int b = bid / nbh;
  const int h0 = bh * 4, w0 = bw * 4, d0 = bd * 8;
  const int n_blk = blockIdx.y * BN;
  const int nchunk = a.C / 32;
  const int nstage = (a.nent + TS - 1) / TS;

  f32x16 acc[NC][TN];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][j][r] = 0.f;

  // wave w owns brick rows 32w..32w+31 = the h-plane w of the brick; row r -> (w, (r >> 3) & 3, r & 7)
  const int arow = ((wave * HALO_W + (li >> 3)) * HALO_D + (li & 7)) * 32;

  uint4 hreg[6];
  auto load_halo = [&](int chunk) {
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      const int idx = tid + p * 256;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (idx < HALO_VOX * 4) {
        const int hv = idx >> 2, part = idx & 3;
        const int hd = hv % HALO_D, hw = (hv / HALO_D) % HALO_W, hh = hv / (HALO_D * HALO_W);
        const int h = h0 - 1 + hh, w = w0 - 1 + hw, d = d0 - 1 + hd;
        if ((unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W && (unsigned)d < (unsigned)a.D)
          v = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(a.x) +
                                              ((((long long)b * a.H + h) * a.W + w) * a.D + d) * a.lda + chunk * 32 + part * 8);
      }
      hreg[p] = v;
    }
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      const int idx = tid + p * 256;
      if (idx < HALO_VOX * 4) {
        const int hv = idx >> 2, hw = (hv / HALO_D) % HALO_W;
        *reinterpret_cast<uint4*>(&halo[hv * 32 + (((idx & 3) ^ (hw & 3)) << 3)]) = hreg[p];
      }
    }
  };
  uint4 breg[LBV];
  auto load_b = [&](int chunk, int stage) {
#pragma unroll
    for (int p = 0; p < LBV; ++p) {
      const int idx = tid + p * 256;
      const int part = idx & 3, nl = (idx >> 2) % BN, t = idx / (4 * BN);
      const int n = n_blk + nl, e = stage * TS + t;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (n < a.N && e < a.nent)
        v = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(a.w) + a.ent[e].wbase + (long long)n * a.wrow + chunk * 32 + part * 8);
      breg[p] = v;
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int p = 0; p < LBV; ++p) {
      const int idx = tid + p * 256;
      const int part = idx & 3, nl = (idx >> 2) % BN, t = idx / (4 * BN);
      *reinterpret_cast<uint4*>(&Bs[((buf * TS + t) * BN + nl) * LDH + part * 8]) = breg[p];
    }
  };

  load_halo(0);
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    __syncthreads();                       // previous chunk fully consumed (halo and both weight buffers)
    store_halo();
    load_b(chunk, 0);
    store_b(0);
    __syncthreads();
    if (chunk + 1 < nchunk) load_halo(chunk + 1);
    for (int s = 0; s < nstage; ++s) {
      const int buf = s & 1;
      if (s + 1 < nstage) load_b(chunk, s + 1);
#pragma unroll
      for (int t = 0; t < TS; ++t) {
        const int e = s * TS + t;
        if (e < a.nent) {
          const ClsEntry en = a.ent[e];
          const int tapoff = (((en.dh + 1) * HALO_W + (en.dw + 1)) * HALO_D + (en.dd + 1)) * 32;
          const int sw = ((li >> 3) + en.dw + 1) & 3;
          bf16x8 av[2], bv[2][TN];
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            av[ks] = *reinterpret_cast<const bf16x8*>(&halo[arow + tapoff + (((ks * 2 + lh) ^ sw) << 3)]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
              bv[ks][j] = *reinterpret_cast<const bf16x8*>(&Bs[((buf * TS + t) * BN + j * 32 + li) * LDH + ks * 16 + lh * 8]);
          }
#define CLS_BODY(c)                                                                                         \
  case c:                                                                                                   \
    if (c < NC) {                                                                                           \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                      \
      _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                        \
        acc[c < NC ? c : 0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ks], bv[ks][j], acc[c < NC ? c : 0][j], 0, 0, 0); \
    }                                                                                                       \
    break;
          switch (en.cls) {
            CLS_BODY(0) CLS_BODY(1) CLS_BODY(2) CLS_BODY(3) CLS_BODY(4) CLS_BODY(5) CLS_BODY(6) CLS_BODY(7)
            default: break;
          }
#undef CLS_BODY
        }
      }
      if (s + 1 < nstage) store_b(buf ^ 1);
      __syncthreads();
    }
  }

  // epilogue, class by class: bias, convert, stage the 128 x BN tile in LDS, 8-byte stores to the fine grid
  uint16_t* Cs = smem;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    if (c >= a.ncls) break;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nl = j * 32 + li;
      const int n = n_blk + nl;
      const float bvv = (a.bias != nullptr && n < a.N) ? a.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ml = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        Cs[ml * LDC + nl] = f32_to_bf16(acc[c][j][r] + bvv);
      }
    }
    __syncthreads();
    constexpr int CPR = BN / 4;
    const int ph = a.cls_p[c][0], pw = a.cls_p[c][1], pd = a.cls_p[c][2];
    for (int idx = tid; idx < 128 * CPR; idx += 256) {
      const int ml = idx / CPR, nl = (idx % CPR) * 4;
      const int n = n_blk + nl;
      const int qh = h0 + (ml >> 5), qw = w0 + ((ml >> 3) & 3), qd = d0 + (ml & 7);
      const int h = qh * a.mh + ph, w = qw * a.mw + pw, d = qd * a.md + pd;
      if (n >= a.N || qh >= a.H || qw >= a.W || qd >= a.D || h >= a.Hh || w >= a.Wh || d >= a.Dh) continue;
      const long long vox = (((long long)b * a.Hh + h) * a.Wh + w) * a.Dh + d;
      const uint2 v = *reinterpret_cast<const uint2*>(&Cs[ml * LDC + nl]);
      uint16_t* dst = n < a.n0 ? reinterpret_cast<uint16_t*>(a.o0) + vox * a.ldo0 + n
                               : reinterpret_cast<uint16_t*>(a.o1) + vox * a.ldo1 + (n - a.n0);
      *reinterpret_cast<uint2*>(dst) = v;
    }
    __syncthreads();
  }
}

// ---- the same computation with everything asynchronous (LDS-DMA) -------------------------------------------------------
// The register-staged version above keeps 8 accumulator tiles per wave, which leaves one workgroup per CU: nothing hides the
// L2 round trip of each weight stage.  Here a workgroup owns a 4x8x8 brick (256 coarse voxels, 4 waves x 64 rows), the weight
// tiles stream through a 4-deep ring of 16 KB stages and the halo of the next channel chunk lands in a second buffer, all by
// global_load_lds with hand-counted vmcnt (see gemm_ring.hip for the idiom).  64-byte LDS rows, chunk c of row r at slot
// c ^ ((r >> 2) & 1) (conflict-free ds_read_b128 per 8-lane group).  Entries arrive sorted by offset so that an A fragment
// is fetched once per distinct offset (27 instead of 64 for the un-embedding); each B fragment feeds two row tiles.
__device__ __attribute__((aligned(64))) uint32_t ltu_zero_line[32];     // source of out-of-volume / padding rows

__device__ __forceinline__ void cglds16(const uint16_t* src, uint16_t* lds_wave_base) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
template <int N>
__device__ __forceinline__ void cring_sync() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

#define CR_HVOX 600          // 6 x 10 x 10 halo voxels
#define CR_HROWS 640         // padded to 40 LDS-DMA pieces of 16 rows
template <int I, int N, class F>
__device__ __forceinline__ void cr_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    cr_static_for<I + 1, N>(f);
  }
}
// The weight stream is organised class by class: stage (chunk, class c, sub-stage sc) holds up to TS entries of class c, so
// the accumulator a stage updates is known at compile time (a run-time class index makes hipcc shuffle all 256 accumulator
// registers around every entry).  SPCLS sub-stages per class; entries e in [cls_begin[c], cls_begin[c + 1]).
template <int NC, int TN, int SPCLS, bool FULL>
__global__ void __launch_bounds__(256) conv_class_ring_bf16_kernel(const ClassHaloArgs a) {
  static_assert(NC * TN == 8, "8 accumulator tile pairs per wave");
  constexpr int BN = 32 * TN, TS = 8 / TN, R = 4, SPC = NC * SPCLS;
  constexpr int HBUF = CR_HROWS * 32, WSTAGE = TS * BN * 32;       // elements (bf16)
  extern __shared__ __attribute__((aligned(1024))) uint16_t smem[];
  uint16_t* halo = smem;                   // [2][640][32]
  uint16_t* ring = smem + 2 * HBUF;        // [R][TS][BN][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int nbh = (a.H + 3) / 4, nbw = (a.W + 7) / 8, nbd = (a.D + 7) / 8;
  int bid = blockIdx.x;
  const int bd = bid % nbd; bid /= nbd;
  const int bw = bid % nbw; bid /= nbw;
  const int bh = bid % nbh;
  const int b = bid / nbh;
  const int h0 = bh * 4, w0 = bw * 8, d0 = bd * 8;
  const int n_blk = blockIdx.y * BN;
  const int nchunk = a.C / 32;
  const int total = nchunk * SPC;
  const uint16_t* zsrc = reinterpret_cast<const uint16_t*>(ltu_zero_line) + (lane & 3) * 8;
  // The entry table is held one entry per lane in two VGPRs and read with v_readlane (wave-uniform index): indexing the
  // kernel-argument copy at run time would be a vector-memory load inside the asynchronous span (hipcc drains vmcnt for it),
  // an LDS copy costs a round trip per use.
  int ent_doff, ent_wbase;                 // (halo voxel offset) * 4 + (dw + 1);  weight tile base or -1
  {
    const ClsEntry en = a.ent[lane < a.nent ? lane : 0];
    ent_doff = ((en.dh * 10 + en.dw) * 10 + en.dd) * 4 + (en.dw + 1);
    ent_wbase = lane < a.nent ? en.wbase : -1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the table is in registers before anything asynchronous starts
  // LDS-DMA lane geometry: piece = 16 rows x 64 B, lane -> (row lane >> 2, slot lane & 3).  64-byte rows: a quarter-wave of a
  // ds_read_b128 covers 16 rows = four times each 64-byte residue of the 256-byte bank row, so the slot needs two row-dependent
  // bits.  A ds_read_b128 lane group is lanes {0-3, 12-15, 20-27} (and its three siblings): for the weight tile those are rows
  // whose (row >> 2) & 3 are all different -> slot = chunk ^ ((row >> 2) & 3); for the halo they are four d-runs on four
  // consecutive w positions -> slot = chunk ^ (w & 3).
  const int prow = lane >> 2;
  const int wchunk = (lane & 3) ^ ((lane >> 4) & 3);
  // per-lane constant part of a weight source address: piece s of a stage holds rows (wave*4 + s)*16 + prow = t*BN + nl with a
  // wave-uniform t
  long long woff[4];
  int wt[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int row = (wave * 4 + s) * 16 + prow;
    wt[s] = row / BN;
    const int n = n_blk + row % BN;
    woff[s] = n < a.N ? (long long)n * a.wrow + wchunk * 8 : -1;
  }

  auto issue_halo = [&](int chunk) {
    uint16_t* hb = halo + (chunk & 1) * HBUF;
#pragma unroll
    for (int s = 0; s < 10; ++s) {
      const int piece = wave * 10 + s;
      const int hv = piece * 16 + prow;
      const int hd = hv % 10, hw = (hv / 10) % 10, hh = hv / 100;
      const int h = h0 - 1 + hh, w = w0 - 1 + hw, d = d0 - 1 + hd;
      const int lc = (lane & 3) ^ (hw & 3);
      const uint16_t* src = zsrc;
      if (hv < CR_HVOX && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W && (unsigned)d < (unsigned)a.D)
        src = reinterpret_cast<const uint16_t*>(a.x) + ((((long long)b * a.H + h) * a.W + w) * a.D + d) * a.lda + chunk * 32 + lc * 8;
      cglds16(src, hb + piece * 512);
    }
  };
  auto issue_w = [&](int g) {
    const int chunk = g / SPC, st = g - chunk * SPC;
    const int c = st / SPCLS, sc = st - c * SPCLS;
    int e0 = a.cls_begin[0], e1 = a.cls_begin[1];              // class c of a run-time stage index: uniform selects, no memory
#pragma unroll
    for (int q = 1; q < NC; ++q) {
      e0 = c == q ? a.cls_begin[q] : e0;
      e1 = c == q ? a.cls_begin[q + 1] : e1;
    }
    e0 += sc * TS;
    uint16_t* wb = ring + (g % R) * WSTAGE + wave * 4 * 512;
    const uint16_t* wsrc = reinterpret_cast<const uint16_t*>(a.w) + chunk * 32;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int e = e0 + __builtin_amdgcn_readfirstlane(wt[s]);
      const int wbase = e < e1 ? __builtin_amdgcn_readlane(ent_wbase, e & 63) : -1;
      const uint16_t* src = (wbase >= 0 && woff[s] >= 0) ? wsrc + wbase + woff[s] : zsrc;
      cglds16(src, wb + s * 512);
    }
  };

  f32x16 acc[NC][2][TN];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][i][j][r] = 0.f;

  // wave w owns brick rows 64w.. = h-plane w; tile i covers w positions 4i..4i+3; halo voxel of (row, offset 0,0,0 -> +1,+1,+1)
  int hv0[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) hv0[i] = ((wave + 1) * 10 + (i * 4 + (li >> 3) + 1)) * 10 + (li & 7) + 1;
  const int hwl = li >> 3;                                    // this lane's w inside its tile (halo w = 4 i + hwl + 1 + dw)
  const int wsw = (li >> 2) & 3;

  issue_halo(0);
  for (int g = 0; g < R - 1 && g < total; ++g) issue_w(g);
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    const uint16_t* hb = halo + (chunk & 1) * HBUF;
    cr_static_for<0, SPC>([&](auto ST) {
      constexpr int st = decltype(ST)::value, c = st / SPCLS, sc = st % SPCLS;
      const int g = chunk * SPC + st;
      if (g + 2 < total) {
        // LDS-DMA issued after W(g): W(g+1), W(g+2) and the halos issued at iterations g-3..g-1 (those with st == 0)
        constexpr bool near0 = st >= 1 && st <= 3;
        if (near0 && chunk + 1 < nchunk) cring_sync<18>(); else cring_sync<8>();
      } else {
        cring_sync<0>();
      }
      if (g + R - 1 < total) issue_w(g + R - 1);
      if (st == 0 && chunk + 1 < nchunk) issue_halo(chunk + 1);
      const uint16_t* wb = ring + (g % R) * WSTAGE;
      const int e0 = a.cls_begin[c] + sc * TS;
      // fragments of entry t+1 are fetched before the MFMAs of entry t are issued (pinned with sched_barrier: left alone,
      // hipcc sinks every ds_read next to its MFMA and each MFMA then waits a full LDS round trip)
      auto load_frags = [&](int t, int dv, bf16x8 (&af)[2][2], bf16x8 (&wf)[TN][2]) {
        const int doff = dv >> 2, sw = (hwl + (dv & 3)) & 3;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int hv = hv0[i] + doff;
          const uint16_t* p = hb + hv * 32;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) af[i][ks] = *reinterpret_cast<const bf16x8*>(p + (((ks * 2 + lh) ^ sw) << 3));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
            wf[j][ks] = *reinterpret_cast<const bf16x8*>(wb + (t * BN + j * 32 + li) * 32 + (((ks * 2 + lh) ^ wsw) << 3));
      };
      // transposed product D[n][voxel]: a lane then owns 4 consecutive n of one voxel (8-byte staging writes in the epilogue)
      auto mma = [&](const bf16x8 (&af)[2][2], const bf16x8 (&wf)[TN][2]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[c][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j][ks], af[i][ks], acc[c][i][j], 0, 0, 0);
      };
      if constexpr (FULL) {                // every stage holds TS entries: straight-line, software-pipelined
        int dv[TS];
#pragma unroll
        for (int t = 0; t < TS; ++t) dv[t] = __builtin_amdgcn_readlane(ent_doff, (e0 + t) & 63);
        bf16x8 afA[2][2], wfA[TN][2], afB[2][2], wfB[TN][2];
        load_frags(0, dv[0], afA, wfA);
#pragma unroll
        for (int t = 0; t < TS; t += 2) {
          if (t + 1 < TS) load_frags(t + 1, dv[t + 1], afB, wfB);
          __builtin_amdgcn_sched_barrier(0);
          mma(afA, wfA);
          __builtin_amdgcn_sched_barrier(0);
          if (t + 2 < TS) load_frags(t + 2, dv[t + 2], afA, wfA);
          __builtin_amdgcn_sched_barrier(0);
          if (t + 1 < TS) mma(afB, wfB);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        const int cnt = a.cls_begin[c + 1] - e0;
#pragma unroll 1
        for (int t = 0; t < cnt && t < TS; ++t) {
          bf16x8 af[2][2], wf[TN][2];
          load_frags(t, __builtin_amdgcn_readlane(ent_doff, (e0 + t) & 63), af, wf);
          mma(af, wf);
        }
      }
    });
  }

  // epilogue, class by class (everything asynchronous has landed: the last iterations waited vmcnt(0))
  constexpr int LDC = BN + 8;
  uint16_t* Cs = smem;                     // [256][LDC]
  float4 bv4[TN][4];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int n = n_blk + j * 32 + 8 * rr + 4 * lh;
      bv4[j][rr] = (a.bias != nullptr && n < a.N) ? *reinterpret_cast<const float4*>(a.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    if (c >= a.ncls) break;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          uint2 pk;
          pk.x = pack_bf16x2(acc[c][i][j][4 * rr + 0] + bv4[j][rr].x, acc[c][i][j][4 * rr + 1] + bv4[j][rr].y);
          pk.y = pack_bf16x2(acc[c][i][j][4 * rr + 2] + bv4[j][rr].z, acc[c][i][j][4 * rr + 3] + bv4[j][rr].w);
          *reinterpret_cast<uint2*>(&Cs[(wave * 64 + i * 32 + li) * LDC + j * 32 + 8 * rr + 4 * lh]) = pk;
        }
    __syncthreads();
    constexpr int CPR = BN / 4;
    const int ph = a.cls_p[c][0], pw = a.cls_p[c][1], pd = a.cls_p[c][2];
    for (int idx = tid; idx < 256 * CPR; idx += 256) {
      const int ml = idx / CPR, nl = (idx % CPR) * 4;
      const int n = n_blk + nl;
      const int qh = h0 + (ml >> 6), qw = w0 + ((ml >> 3) & 7), qd = d0 + (ml & 7);
      const int h = qh * a.mh + ph, w = qw * a.mw + pw, d = qd * a.md + pd;
      if (n >= a.N || qh >= a.H || qw >= a.W || qd >= a.D || h >= a.Hh || w >= a.Wh || d >= a.Dh) continue;
      const long long vox = (((long long)b * a.Hh + h) * a.Wh + w) * a.Dh + d;
      const uint2 v = *reinterpret_cast<const uint2*>(&Cs[ml * LDC + nl]);
      uint16_t* dst = n < a.n0 ? reinterpret_cast<uint16_t*>(a.o0) + vox * a.ldo0 + n
                               : reinterpret_cast<uint16_t*>(a.o1) + vox * a.ldo1 + (n - a.n0);
      *reinterpret_cast<uint2*>(dst) = v;
    }
  }
}

// LTU_OK after launching, or 1 when the shape is not handled (the caller keeps its implicit-GEMM path)
int launch_conv_class_halo_bf16(const ClassHaloArgs& a, hipStream_t st) {
  if (a.C % 32 || a.lda % 8 || a.N % 4 || a.n0 % 4 || a.ldo0 % 4 || a.ldo1 % 4 || a.wrow % 8) return 1;
  if (a.H < 2 || a.W < 2 || a.D < 2 || a.nent > 64 || a.ncls > 8) return 1;
  for (int e = 0; e < a.nent; ++e)
    if (a.ent[e].wbase % 8) return 1;
  const long long bricks = (long long)a.B * ((a.H + 3) / 4) * ((a.W + 3) / 4) * ((a.D + 7) / 8);
  if (bricks >= (1LL << 31)) return 1;
  if (!ltu_knob("LTU_NO_CLASS_RING", 0) && a.ncls >= 3 && a.W >= 4) {
    // entries grouped by class; the ring kernel holds at most 8 (8 classes) / 12 (4 classes) entries per class
    ClassHaloArgs r = a;
    int ne = 0;
    bool fits = true;
    for (int c = 0; c < a.ncls; ++c) {
      r.cls_begin[c] = ne;
      for (int i = 0; i < a.nent; ++i)
        if (a.ent[i].cls == c) r.ent[ne++] = a.ent[i];
      if (ne - r.cls_begin[c] > (a.ncls > 4 ? 8 : 12)) fits = false;
    }
    for (int c = a.ncls; c <= 8; ++c) r.cls_begin[c] = ne;
    if (fits) {
      const long long rb = (long long)a.B * ((a.H + 3) / 4) * ((a.W + 7) / 8) * ((a.D + 7) / 8);
      constexpr int smem_bytes = 2 * CR_HROWS * 64 + 4 * 16384;
      static LtuDevOnce attr_once;
      if (attr_once.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_class_ring_bf16_kernel<8, 1, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_class_ring_bf16_kernel<8, 1, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_class_ring_bf16_kernel<4, 2, 3, false>), hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
      }
      bool full = a.ncls == 8;
      for (int c = 0; c < a.ncls; ++c) full = full && (r.cls_begin[c + 1] - r.cls_begin[c] == 8);
      const dim3 g8((unsigned)rb, cdiv(a.N, 32)), g4((unsigned)rb, cdiv(a.N, 64));
      if (a.ncls > 4 && full) hipLaunchKernelGGL((conv_class_ring_bf16_kernel<8, 1, 1, true>), g8, dim3(256), smem_bytes, st, r);
      else if (a.ncls > 4) hipLaunchKernelGGL((conv_class_ring_bf16_kernel<8, 1, 1, false>), g8, dim3(256), smem_bytes, st, r);
      else hipLaunchKernelGGL((conv_class_ring_bf16_kernel<4, 2, 3, false>), g4, dim3(256), smem_bytes, st, r);
      return ltu_check_launch();
    }
  }
  // 8 classes keep 8 accumulator tiles per wave: 32 columns per workgroup (and more workgroups for the small grids)
  const bool wide = a.N > 32 && a.ncls <= 4;
  dim3 grid((unsigned)bricks, cdiv(a.N, wide ? 64 : 32));
  if (a.ncls <= 4) {
    if (wide) hipLaunchKernelGGL((conv_class_halo_bf16_kernel<4, 2>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_class_halo_bf16_kernel<4, 1>), grid, dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL((conv_class_halo_bf16_kernel<8, 1>), grid, dim3(256), 0, st, a);
  }
  return ltu_check_launch();
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the sub-pixel un-embedding (nearest x2 + 3x3x3 conv):  dWeff[class][slot] = sum_q G_class(q)^T X(q + off),
// folded onto the 27 real taps.  A workgroup owns (32-channel chunk of Ci, 32-column tile of Co, a range of coarse 4x4x8
// bricks); per brick it stages the X halo once and the 8 class tiles of the fine-grid gradient.  Each of the 4 waves
// keeps ALL 64 (class, slot) products for a 16 x 16 sub-tile (v_mfma_f32_16x16x32_bf16, 4 accumulator registers each),
// so the fold 64 -> 27 is a register add at the end and the partial sums leave in the [split][Co][27 Ci] layout that
// wgrad_reduce_kernel already folds into the PyTorch gradient.  An X fragment is read once per distinct offset (27), a
// gradient fragment once per class (8); next brick's tiles are prefetched into registers during the 256 MFMAs.
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__global__ void __launch_bounds__(256) upconv_wgrad_class_bf16_kernel(const UpWgradArgs a) {
  constexpr int LDX = 40, LDG = 40;
  __shared__ __attribute__((aligned(16))) uint16_t Xs[HALO_VOX * LDX];         // 28.8 KB
  __shared__ __attribute__((aligned(16))) uint16_t Gs[8 * 128 * LDG];          // 80 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wc = wave & 1;                 // 16-column sub-tile of Co / of the Ci chunk
  const int chunk = blockIdx.x, n_blk = blockIdx.y * 32;
  const int nbh = (a.H + 3) / 4, nbw = (a.W + 3) / 4, nbd = (a.D + 7) / 8;
  const int brick_lo = blockIdx.z * a.bricks_per_split;
  int brick_hi = brick_lo + a.bricks_per_split;
  if (brick_hi > a.bricks) brick_hi = a.bricks;

  f32x4_t acc[64];
#pragma unroll
  for (int e = 0; e < 64; ++e) acc[e] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;

  uint4 xreg[6], greg[16];
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
      if (idx < HALO_VOX * 4) {
        const int hv = idx >> 2, part = idx & 3;
        const int hd = hv % HALO_D, hw = (hv / HALO_D) % HALO_W, hh = hv / (HALO_D * HALO_W);
        const int h = h0 - 1 + hh, w = w0 - 1 + hw, d = d0 - 1 + hd;
        if ((unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W && (unsigned)d < (unsigned)a.D)
          v = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(a.x) +
                                              ((((long long)b * a.H + h) * a.W + w) * a.D + d) * a.Ci + chunk * 32 + part * 8);
      }
      xreg[p] = v;
    }
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const int idx = tid + p * 256;                       // [class][row][4 parts]
      const int part = idx & 3, row = (idx >> 2) & 127, cls = idx >> 9;
      const int qh = h0 + (row >> 5), qw = w0 + ((row >> 3) & 3), qd = d0 + (row & 7);
      const int n = n_blk + part * 8;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (qh < a.H && qw < a.W && qd < a.D && n < a.Co) {
        const int h = 2 * qh + (cls >> 2), w = 2 * qw + ((cls >> 1) & 1), d = 2 * qd + (cls & 1);
        v = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(a.grad) +
                                            ((((long long)b * 2 * a.H + h) * 2 * a.W + w) * 2 * a.D + d) * a.Co + n);
      }
      greg[p] = v;
    }
  };
  auto store_brick = [&]() {
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      const int idx = tid + p * 256;
      if (idx < HALO_VOX * 4) *reinterpret_cast<uint4*>(&Xs[(idx >> 2) * LDX + (idx & 3) * 8]) = xreg[p];
    }
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const int idx = tid + p * 256;
      *reinterpret_cast<uint4*>(&Gs[(idx >> 2) * LDG + (idx & 3) * 8]) = greg[p];
    }
  };

  // transposing-read geometry for 16x16x32: 16-lane group gq supplies k rows 8 gq .. 8 gq + 7 of the 32-row slab;
  // lane 4q + p of the group addresses row q (and q + 4), columns 4p..4p+3
  const int gq = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  const int growoff = (8 * gq + tq) * LDG + wn * 16 + 4 * tp;
  // slab ks covers brick rows 32 ks.. = h-plane ks; row r of it -> (w = r >> 3, d = r & 7); this lane: r = 8 gq + tq (+4)
  const int xrowoff = ((gq + 1) * HALO_D + tq + 1) * LDX + wc * 16 + 4 * tp;     // +1: offset (0,0,0) sits at halo (1,1,1)
  const bool do_bias = chunk == 0 && wc == 0;

  if (brick_lo < brick_hi) load_brick(brick_lo);
  for (int brick = brick_lo; brick < brick_hi; ++brick) {
    __syncthreads();
    store_brick();
    __syncthreads();
    if (brick + 1 < brick_hi) load_brick(brick + 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 ga[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        union { struct { hs16x4 l, h; } s; bf16x8 v; } u;
        const uint16_t* pg = &Gs[(c * 128 + ks * 32) * LDG + growoff];
        u.s.l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_hs16x4*)pg);
        u.s.h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_hs16x4*)(pg + 4 * LDG));
        ga[c] = u.v;
        if (do_bias) {
#pragma unroll
          for (int e = 0; e < 8; ++e) bsum += (float)u.v[e];
        }
      }
#pragma unroll
      for (int oh = -1; oh <= 1; ++oh)
#pragma unroll
        for (int ow = -1; ow <= 1; ++ow)
#pragma unroll
          for (int od = -1; od <= 1; ++od) {
            union { struct { hs16x4 l, h; } s; bf16x8 v; } ub;
            const uint16_t* px = &Xs[(((ks + 1 + oh) * HALO_W + ow) * HALO_D + od) * LDX + xrowoff];
            ub.s.l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_hs16x4*)px);
            ub.s.h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_hs16x4*)(px + 4 * LDX));
#pragma unroll
            for (int c = 0; c < 8; ++c) {
              const int ph = c >> 2, pw = (c >> 1) & 1, pd = c & 1;
              const int sh = oh + 1 - ph, sw = ow + 1 - pw, sd = od + 1 - pd;      // the slot of class c that reads this offset
              if (sh >= 0 && sh <= 1 && sw >= 0 && sw <= 1 && sd >= 0 && sd <= 1) {
                const int e = c * 8 + (sh * 2 + sw) * 2 + sd;
                acc[e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga[c], ub.v, acc[e], 0, 0, 0);
              }
            }
          }
    }
  }

  // fold the 64 (class, slot) products onto the 27 taps and store this split's partial sums
  float* pz = a.part + (long long)blockIdx.z * a.Co * a.kpad;
  const int ci = chunk * 32 + wc * 16 + (lane & 15);
#pragma unroll
  for (int th = 0; th < 3; ++th)
#pragma unroll
    for (int tw = 0; tw < 3; ++tw)
#pragma unroll
      for (int td = 0; td < 3; ++td) {
        f32x4_t s = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int ph = c >> 2, pw = (c >> 1) & 1, pd = c & 1;
          const int sh = ph == 0 ? (th == 0 ? 0 : 1) : (th == 2 ? 1 : 0);
          const int sw = pw == 0 ? (tw == 0 ? 0 : 1) : (tw == 2 ? 1 : 0);
          const int sd = pd == 0 ? (td == 0 ? 0 : 1) : (td == 2 ? 1 : 0);
          s += acc[c * 8 + (sh * 2 + sw) * 2 + sd];
        }
        const int tap = (th * 3 + tw) * 3 + td;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n_blk + wn * 16 + 4 * gq + r;
          if (n < a.Co) pz[(long long)n * a.kpad + tap * a.Ci + ci] = s[r];
        }
      }
  if (do_bias) {
    bsum = xrow_combine<LtuAdd>(bsum);
    bsum = xhalf_combine<LtuAdd>(bsum);
    const int n = n_blk + wn * 16 + (lane & 15);
    if (lane < 16 && n < a.Co) a.bpart[(long long)blockIdx.z * a.Co + n] = bsum;
  }
}

static int upw_blocks(int blocks) { return blocks > 0 ? blocks : ltu_knob_pos("LTU_UPW_BLOCKS", 256); }
static bool upw_shape_ok(int Ci, int Co, int H, int W, int D) { return Ci % 32 == 0 && Co % 8 == 0 && H >= 2 && W >= 2 && D >= 2; }
long long upconv_wgrad_class_ws_floats(int Ci, int Co, int blocks) {
  if (Ci % 32 || Co % 8) return 0;
  long long ns = upw_blocks(blocks) / ((long long)(Ci / 32) * cdiv(Co, 32));
  if (ns < 1) ns = 1;
  return ns * Co * (27LL * Ci + 1);
}
// fills the split geometry, launches; returns 1 when the shape is not handled
int launch_upconv_wgrad_class_bf16(UpWgradArgs a, int* nsplit_out, hipStream_t st) {
  if (!upw_shape_ok(a.Ci, a.Co, a.H, a.W, a.D)) return 1;
  const long long bricks = (long long)a.B * ((a.H + 3) / 4) * ((a.W + 3) / 4) * ((a.D + 7) / 8);
  if (bricks >= (1LL << 31)) return 1;
  const int nchunk = a.Ci / 32, ntile = cdiv(a.Co, 32);
  long long ns = upw_blocks(a.blocks) / ((long long)nchunk * ntile);
  if (ns < 1) ns = 1;
  if (ns > bricks) ns = bricks;
  a.bricks = (int)bricks;
  a.bricks_per_split = (int)((bricks + ns - 1) / ns);
  const int nsplit = (int)((bricks + a.bricks_per_split - 1) / a.bricks_per_split);
  a.kpad = 27 * a.Ci;
  if ((long long)nsplit * a.Co * ((long long)a.kpad + 1) > a.part_floats) return LTU_E_ARG;
  a.bpart = a.part + (long long)nsplit * a.Co * a.kpad;
  *nsplit_out = nsplit;
  if (ltu_knob("LTU_UPW_RING", 1)) {        // second generation (upconv_wgrad_ring.hip)
    const int rc = launch_upconv_wgrad_ring_bf16(a, nchunk, ntile, nsplit, st);
    if (rc != 1) return rc;
  }
  hipLaunchKernelGGL(upconv_wgrad_class_bf16_kernel, dim3(nchunk, ntile, nsplit), dim3(256), 0, st, a);
  return ltu_check_launch();
}
