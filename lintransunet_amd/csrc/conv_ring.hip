// Stride-1 3x3x3 convolution, forward and data gradient, for the levels with C >= 64 input channels (or N > 32 outputs) and enough
// bricks to fill the machine - model/Unet_3Dblock.py:325-341, 540-557 at 32x32x128 (64 -> 32 + 32 pair, 32 + 32 -> 32, their data
// gradients) and 16x16x64.  Second generation of conv3_halo_bf16_kernel in the style of upconv_ring.hip / sdgrad_ring.hip:
//   * a workgroup owns a 4x8x8 brick (256 voxels: every weight fragment feeds two row tiles) and BN = 64 or 128 output columns (TN = 2 / 4
//     column tiles: every activation fragment feeds TN MFMAs); accumulators 2 x TN tiles in AGPRs;
//   * per 32-channel chunk the 6x10x10 halo arrives by LDS-DMA into one of two buffers (per-lane pointers: the chunk's concat source,
//     zero line outside the volume), the 27 weight tiles as 9 stages of 3 taps (one d-run) through a 3-deep ring;
//   * tap offsets are compile-time (FLIP = data gradient: mirrored taps): a tap is 4 + 2 TN ds_read_b128 with immediate offsets and
//     4 TN MFMAs;
//   * epilogue: the 256 x BN tile staged once, 16-byte stores into the two destinations (conv pairs / gradient concat).
#include "gemm_desc.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define CR_HBUF (640 * 64)                // 6 x 10 x 10 halo rows padded to 40 pieces
#define CR_RING (2 * CR_HBUF)

__device__ __attribute__((aligned(64))) uint32_t ltu_zero_cr[512];

__device__ __forceinline__ void cr_glds16(const void* src, uint32_t lds_byte_addr) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_byte_addr);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
template <int N>
__device__ __forceinline__ void cr_sync() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}
template <int I, int N, class F>
__device__ __forceinline__ void cr_sfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    cr_sfor<I + 1, N>(f);
  }
}

// HB = halo buffers.  2: the next chunk's halo lands while this chunk is multiplied (one workgroup per CU at 116-152 KB of LDS).
// 1 (TN = 2 only: 64 accumulator registers): 76 KB of LDS and <= 256 registers = TWO workgroups per CU - the halo of a chunk is then
// requested when the previous chunk has been consumed and its latency is covered by the other workgroup's MFMAs, as are the
// LDS-DMA issue slots and barriers that one wave per SIMD has nothing to hide behind.
// NW = waves per workgroup = h-planes of the brick.  8 (HB = 1, TN = 2): an 8x8x8 brick of 512 voxels with a 10x10x10 halo - the
// weight tiles (73 % of the LDS-DMA traffic of the 4-wave kernel: every workgroup streams all of them) are fetched once per 512
// instead of 256 voxels and the halo factor drops from 2.34 to 1.95; one workgroup of 100 KB per CU, two waves per SIMD as before.
// The weight pieces of a stage (12) are issued by waves 0-3 only (3 each: a uniform count for the counted waits).
// SPLIT (small grids, the deep levels: a few dozen bricks, thousands of input channels x taps): workgroup z walks the channel chunks
// [z cps, (z + 1) cps) only and stores its fp32 tile to part[z][voxel][N]; conv_halo_fold_kernel adds the splits and the bias.
template <int TN, bool FLIP, int HB, int NW, bool SPLIT = false>
__global__ void __launch_bounds__(NW * 64, (HB == 1 && NW == 4) ? 2 : 1) conv3_ring_bf16_kernel(const HaloArgs a) {
  constexpr int BN = 32 * TN, WTAP = BN * 64, WSTAGE = 3 * WTAP, PW = 3 * BN / 64;      // bytes per tap tile / stage; weight pieces per (issuing) wave and stage
  constexpr int HROWS = (NW + 2) * 100, HPW = ((HROWS + 15) / 16 + NW - 1) / NW;        // halo rows; halo pieces per wave
  constexpr int HBUFB = NW * HPW * 1024;                                                // bytes per halo buffer
  constexpr int RING = HB * HBUFB;
  constexpr int NTHR = NW * 64;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int nbh = (a.H + NW - 1) / NW, nbw = (a.W + 7) / 8, nbd = (a.D + 7) / 8;
  int bid = blockIdx.x;
  const int bd = bid % nbd; bid /= nbd;
  const int bw = bid % nbw; bid /= nbw;
  const int bh = bid % nbh;
  const int b = bid / nbh;
  const int h0 = bh * NW, w0 = bw * 8, d0 = bd * 8;
  const int n_blk = blockIdx.y * BN;
  const int c_lo = SPLIT ? blockIdx.z * a.cps : 0;                       // first chunk of this workgroup; `chunk` below counts from it
  const int nchunk = SPLIT ? min(a.C / 32 - c_lo, a.cps) : a.C / 32;
  const int total = nchunk * 9;
  const bool wissue = NW == 4 || wave < 4;   // this wave issues weight pieces
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const char* zsrc = reinterpret_cast<const char*>(ltu_zero_cr) + (lane & 3) * 16;
  const int prow = lane >> 2;

  // ---- halo pieces: 10 per wave; voxel offset of the row and whether it lies inside the volume (the brick is fixed per workgroup) ----
  long long hvox[HPW];                     // voxel index of the piece's row, or -1 (outside the volume / padding row)
  int hslot[HPW];                          // channel quarter this lane fetches: (lane & 3) ^ (hw & 3)
#pragma unroll
  for (int s = 0; s < HPW; ++s) {
    const int hv = (wave * HPW + s) * 16 + prow;
    const int hd = hv % 10, hw = (hv / 10) % 10, hh = hv / 100;
    const int h = h0 - 1 + hh, w = w0 - 1 + hw, d = d0 - 1 + hd;
    const bool in = hv < HROWS && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W && (unsigned)d < (unsigned)a.D;
    hvox[s] = in ? (((long long)b * a.H + h) * a.W + w) * a.D + d : -1;
    hslot[s] = ((lane & 3) ^ (hw & 3)) * 8;
  }
  auto issue_halo = [&](int chunk) {
    const int c = (c_lo + chunk) * 32;
    const bool s1 = c >= a.c0;             // the chunk's concat source (c0 is a multiple of 32 or equals C)
    const char* base = s1 ? reinterpret_cast<const char*>(a.x1) + (c - a.c0) * 2 : reinterpret_cast<const char*>(a.x0) + c * 2;
    const int ld2 = (s1 ? a.lda1 : a.lda0) * 2;
    const uint32_t hb = lds0 + (HB == 2 ? (chunk & 1) * HBUFB : 0) + wave * HPW * 1024;
#pragma unroll
    for (int s = 0; s < HPW; ++s) {
      const char* src = hvox[s] >= 0 ? base + hvox[s] * ld2 + hslot[s] * 2 : zsrc;
      cr_glds16(src, hb + s * 1024);
    }
  };
  // ---- weight pieces of a stage (3 taps x BN rows): PW per wave -------------------------------------------------------------------
  int woff[PW];                            // element offset inside W [N][27][C] relative to the stage's first tap and the chunk, or -1
#pragma unroll
  for (int s = 0; s < PW; ++s) {
    const int r = ((wave & 3) * PW + s) * 16 + prow, t3 = r / BN, n = n_blk + r % BN;
    const int wchunk = (lane & 3) ^ ((lane >> 4) & 3);
    woff[s] = n < a.N ? (n * 27 + t3) * a.C + wchunk * 8 : -1;
  }
  auto issue_w = [&](int g) {               // stage g = chunk * 9 + st: taps 3 st .. 3 st + 2
    const int chunk = g / 9, st = g - chunk * 9;
    const uint16_t* wsrc = reinterpret_cast<const uint16_t*>(a.w) + st * 3 * a.C + (c_lo + chunk) * 32;
    const uint32_t wb = lds0 + RING + (st % 3) * WSTAGE + (wave & 3) * PW * 1024;
    if (!wissue) return;
#pragma unroll
    for (int s = 0; s < PW; ++s) cr_glds16(woff[s] >= 0 ? reinterpret_cast<const char*>(wsrc + woff[s]) : zsrc, wb + s * 1024);
  };

  // ---- fragment read addresses (see upconv_ring.hip) --------------------------------------------------------------------------------
  const int hwl = li >> 3;
  int baseA[2][2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int hv0 = ((wave + 1) * 10 + (i * 4 + hwl + 1)) * 10 + (li & 7) + 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int dwi = 0; dwi < 3; ++dwi) baseA[i][ks][dwi] = (hv0 - 111) * 64 + (((ks * 2 + lh) ^ ((hwl + dwi) & 3)) << 4);
  }
  int baseW[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) baseW[ks] = RING + li * 64 + (((ks * 2 + lh) ^ ((li >> 2) & 3)) << 4);

  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  issue_halo(0);
  issue_w(0);
  if (total > 1) issue_w(1);
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    const int hoff = HB == 2 ? (chunk & 1) * HBUFB : 0;
    int bA[2][2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int dwi = 0; dwi < 3; ++dwi) bA[i][ks][dwi] = baseA[i][ks][dwi] + hoff;
    cr_sfor<0, 9>([&](auto ST) {
      constexpr int st = decltype(ST)::value;
      const int g = chunk * 9 + st;
      // LDS-DMA issued after W(g): W(g+1) (PW pieces) and - at stages 1 and 2 - the next chunk's halo (10 pieces, issued in stage 0
      // behind W(g+2))
      if constexpr (HB == 2) {
        if (g + 1 < total) {
          if ((st == 1 || st == 2) && chunk + 1 < nchunk) cr_sync<PW + HPW>(); else cr_sync<PW>();
        } else {
          cr_sync<0>();
        }
        if (g + 2 < total) issue_w(g + 2);
        if (st == 0 && chunk + 1 < nchunk) issue_halo(chunk + 1);
      } else {
        // one halo buffer: the chunk's halo was requested behind W(g+1) at the end of the previous chunk (the youngest request: it has
        // landed only when nothing is outstanding)
        if (st == 0 || g + 1 >= total || !wissue) cr_sync<0>(); else cr_sync<PW>();      // (waves that issue no weight pieces have nothing else in flight)
        if (g + 2 < total) issue_w(g + 2);
      }
      auto load_frags = [&](auto TT, bf16x8 (&af)[2][2], bf16x8 (&wf)[TN][2]) {
        constexpr int td = decltype(TT)::value, th = st / 3, tw = st % 3;
        constexpr int dh = FLIP ? 1 - th : th - 1, dw = FLIP ? 1 - tw : tw - 1, dd = FLIP ? 1 - td : td - 1;
        constexpr int immA = (((dh * 10 + dw) * 10 + dd) + 111) * 64;
        constexpr int immW = (st % 3) * WSTAGE + td * WTAP;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) af[i][ks] = *reinterpret_cast<const bf16x8*>(smem + bA[i][ks][dw + 1] + immA);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) wf[j][ks] = *reinterpret_cast<const bf16x8*>(smem + baseW[ks] + immW + j * 2048);
      };
      auto mma = [&](const bf16x8 (&af)[2][2], const bf16x8 (&wf)[TN][2]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j][ks], af[i][ks], acc[i][j], 0, 0, 0);
      };
      bf16x8 afA[2][2], wfA[TN][2], afB[2][2], wfB[TN][2];
      load_frags(std::integral_constant<int, 0>{}, afA, wfA);
      load_frags(std::integral_constant<int, 1>{}, afB, wfB);
      __builtin_amdgcn_sched_barrier(0);
      mma(afA, wfA);
      __builtin_amdgcn_sched_barrier(0);
      load_frags(std::integral_constant<int, 2>{}, afA, wfA);
      __builtin_amdgcn_sched_barrier(0);
      mma(afB, wfB);
      __builtin_amdgcn_sched_barrier(0);
      mma(afA, wfA);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (HB == 1 && st == 8) {
        if (chunk + 1 < nchunk) {
          __syncthreads();                   // every wave has read its last fragments of this chunk's halo
          issue_halo(chunk + 1);
        }
      }
    });
  }

  if constexpr (SPLIT) {
    // fp32 tile of this channel range: lane (li, lh) owns voxel row wave * 64 + i * 32 + li and, per column tile, 4 x 4 columns
    const long long M = (long long)a.B * a.H * a.W * a.D;
    float* pz = a.part + (long long)blockIdx.z * M * a.N;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = wave * 64 + i * 32 + li;
      const int qh = h0 + (row >> 6), qw = w0 + ((row >> 3) & 7), qd = d0 + (row & 7);
      if (qh >= a.H || qw >= a.W || qd >= a.D) continue;
      float* pr = pz + ((((long long)b * a.H + qh) * a.W + qw) * a.D + qd) * a.N;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int n = n_blk + j * 32 + 8 * rr + 4 * lh;
          if (n < a.N) *reinterpret_cast<float4*>(pr + n) = make_float4(acc[i][j][4 * rr], acc[i][j][4 * rr + 1], acc[i][j][4 * rr + 2], acc[i][j][4 * rr + 3]);
        }
    }
    return;
  }
  // ---- epilogue: [256 voxels][BN n] staged (rows of 2 BN bytes, 16-byte parts XOR-ed with the row), bias added; 16 bytes per lane ----
  constexpr int PARTS = BN / 8, RB = 2 * BN;
  float4 bv[TN][4];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int n = n_blk + j * 32 + 8 * rr + 4 * lh;
      bv[j][rr] = (a.bias != nullptr && n < a.N) ? *reinterpret_cast<const float4*>(a.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wave * 64 + i * 32 + li;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        uint2 pk;
        pk.x = pack_bf16x2(acc[i][j][4 * rr + 0] + bv[j][rr].x, acc[i][j][4 * rr + 1] + bv[j][rr].y);
        pk.y = pack_bf16x2(acc[i][j][4 * rr + 2] + bv[j][rr].z, acc[i][j][4 * rr + 3] + bv[j][rr].w);
        *reinterpret_cast<uint2*>(smem + row * RB + (((j * 4 + rr) ^ (row & (PARTS - 1))) << 4) + lh * 8) = pk;
      }
  }
  __syncthreads();
#pragma unroll 4
  for (int it = 0; it < PARTS * 256 / NTHR * (NW / 4); ++it) {       // NW * 64 rows x PARTS pieces over NTHR threads
    const int idx = it * NTHR + tid, part = idx & (PARTS - 1), row = idx / PARTS;
    const int qh = h0 + (row >> 6), qw = w0 + ((row >> 3) & 7), qd = d0 + (row & 7);
    const int n = n_blk + part * 8;
    const uint4 v = *reinterpret_cast<const uint4*>(smem + row * RB + ((part ^ (row & (PARTS - 1))) << 4));
    if (qh < a.H && qw < a.W && qd < a.D && n < a.N) {
      const long long vox = (((long long)b * a.H + qh) * a.W + qw) * a.D + qd;
      uint16_t* dst = n < a.n0 ? reinterpret_cast<uint16_t*>(a.o0) + vox * a.ldo0 + n
                               : reinterpret_cast<uint16_t*>(a.o1) + vox * a.ldo1 + (n - a.n0);
      *reinterpret_cast<uint4*>(dst) = v;
    }
  }
}

// split geometry of the small grids (shared with the workspace query): number of channel ranges (1 = no split) and chunks per range
static int conv_ring_split(long long bricks, int C, int N, int* cps) {
  const int nchunk = C / 32;
  const long long tiles = bricks * ((N + 63) / 64);
  // up to 64 tiles (8x8x64 and deeper at B = 2): 256 -> 128 at 8x8x64 31.4 -> 24.7 us forward, 8x8x32 21.9 -> 20.1 / 25.1 -> 23.8 (data gradient),
  // 4x4x16 15.5 -> 14.5.  With 128 tiles (16x16x64, 128 -> 64; the data gradient of 256 -> 128 at 8x8x64) two ranges of fp32 partial tiles
  // cost more than the first generation's nine-tap stages save: 28.6 -> 31.1, 26.9 -> 29.7 - those stay there (profiles/r05_conv_ring_split.txt)
  if (tiles > ltu_knob_pos("LTU_CONV_RING_SPLIT_TILES", 64) || ltu_knob("LTU_NO_CONV_RING_SPLIT", 0)) return 1;
  int want = (int)((ltu_knob_pos("LTU_CONV_RING_SPLIT_SLOTS", 256) + tiles - 1) / tiles);
  if (want > 8) want = 8;
  if (want > nchunk) want = nchunk;
  if (want < 2) return 1;
  *cps = (nchunk + want - 1) / want;
  return (nchunk + *cps - 1) / *cps;
}
bool conv_ring_splits(int B, int H, int W, int D, int C, int N) {
  if (C % 32 || C > 512 || C < 64 || N % 8 || N <= 32 || H < 2 || W < 4 || D < 4 || ltu_knob("LTU_NO_CONV_RING", 0)) return false;
  int cps = 0;
  return conv_ring_split((long long)B * ((H + 3) / 4) * ((W + 7) / 8) * ((D + 7) / 8), C, N, &cps) > 1;
}

// LTU_OK after launching, or 1 when the shape is not handled here (the caller keeps the first-generation kernels)
int launch_conv_ring_bf16(HaloArgs& a, hipStream_t st) {
  if (a.C % 32 || a.C > 512 || a.N % 8 || a.n0 % 8 || a.lda0 % 8 || a.lda1 % 8 || a.ldo0 % 8 || a.ldo1 % 8) return 1;
  if (a.c0 % 32 != 0 && a.c0 != a.C) return 1;
  if (a.H < 2 || a.W < 4 || a.D < 4 || (long long)a.N * 27 * a.C >= (1LL << 31)) return 1;
  if (a.N <= 32) return 1;                // half of a 64-column tile would be padding: measured slower than the first generation
  const long long bricks = (long long)a.B * ((a.H + 3) / 4) * ((a.W + 7) / 8) * ((a.D + 7) / 8);
  if (bricks >= (1LL << 31)) return 1;
  // 128-column tiles where that still gives the machine enough workgroups, 64-column tiles otherwise; small grids stay with the
  // first-generation kernels, which split the input channels over workgroups
  const bool wide = a.N > 64 && bricks * ((a.N + 127) / 128) >= 200;
  const int bn = wide ? 128 : 64;
  const long long bricks8 = (long long)a.B * ((a.H + 7) / 8) * ((a.W + 7) / 8) * ((a.D + 7) / 8);
  // 8x8x8 bricks (eight waves) where they still fill the machine and H does not waste half a brick
  const bool big = !wide && a.H % 8 == 0 && bricks8 * ((a.N + 63) / 64) >= ltu_knob_pos("LTU_CONV_RING_BIG_MIN", 256) && !ltu_knob("LTU_CONV_RING_NO_BIG", 0);
  const long long nwg = (big ? bricks8 : bricks) * ((a.N + bn - 1) / bn);
  static LtuDevOnce attr_once;
  constexpr int smem2 = CR_HBUF + 3 * 3 * 64 * 64, smem4 = 2 * CR_HBUF + 3 * 3 * 128 * 64, smem8 = 64 * 1024 + 3 * 3 * 64 * 64;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_ring_bf16_kernel<2, false, 1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, smem2);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_ring_bf16_kernel<2, true, 1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, smem2);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_ring_bf16_kernel<4, false, 2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, smem4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_ring_bf16_kernel<4, true, 2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, smem4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_ring_bf16_kernel<2, false, 1, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, smem8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_ring_bf16_kernel<2, true, 1, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, smem8);
  }
  if (nwg < ltu_knob_pos("LTU_CONV_RING_MIN_WG", 200)) {
    // small grid: 64-column tiles and the channel chunks split over up to 8 workgroups (two workgroups per CU: 512 slots)
    int cps = 0;
    const int ks = conv_ring_split(bricks, a.C, a.N, &cps);
    if (a.part == nullptr || ks < 2) return 1;
    a.cps = cps;
    a.ksplit = ks;
    if ((long long)a.ksplit * a.B * a.H * a.W * a.D * a.N > a.part_floats) { a.ksplit = 1; return LTU_E_ARG; }
    static LtuDevOnce split_once;
    if (split_once.first()) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_ring_bf16_kernel<2, false, 1, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem2);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_ring_bf16_kernel<2, true, 1, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem2);
    }
    const dim3 sgrid((unsigned)bricks, (unsigned)((a.N + 63) / 64), (unsigned)a.ksplit);
    if (a.flip) hipLaunchKernelGGL((conv3_ring_bf16_kernel<2, true, 1, 4, true>), sgrid, dim3(256), smem2, st, a);
    else hipLaunchKernelGGL((conv3_ring_bf16_kernel<2, false, 1, 4, true>), sgrid, dim3(256), smem2, st, a);
    return ltu_check_launch();           // the caller folds part[0 .. ksplit) (conv_halo_fold_kernel)
  }
  const dim3 grid((unsigned)(big ? bricks8 : bricks), (unsigned)((a.N + bn - 1) / bn));
  if (wide) {
    if (a.flip) hipLaunchKernelGGL((conv3_ring_bf16_kernel<4, true, 2, 4>), grid, dim3(256), smem4, st, a);
    else hipLaunchKernelGGL((conv3_ring_bf16_kernel<4, false, 2, 4>), grid, dim3(256), smem4, st, a);
  } else if (big) {
    if (a.flip) hipLaunchKernelGGL((conv3_ring_bf16_kernel<2, true, 1, 8>), grid, dim3(512), smem8, st, a);
    else hipLaunchKernelGGL((conv3_ring_bf16_kernel<2, false, 1, 8>), grid, dim3(512), smem8, st, a);
  } else {
    if (a.flip) hipLaunchKernelGGL((conv3_ring_bf16_kernel<2, true, 1, 4>), grid, dim3(256), smem2, st, a);
    else hipLaunchKernelGGL((conv3_ring_bf16_kernel<2, false, 1, 4>), grid, dim3(256), smem2, st, a);
  }
  return ltu_check_launch();
}
