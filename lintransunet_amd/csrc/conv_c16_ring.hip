// Stride-1 3x3x3 convolution (forward and data gradient) at C = 16 input channels and N <= 32 outputs - the first U-Net level
// (model/Unet_3Dblock.py:312-316, 325-341 at 64x64x128: 16 -> 16, 16 -> 16 + 16 as the data gradient of the decoder's concat conv,
// 16 -> head).  Second generation of conv3_halo_wr_bf16_kernel<16>, built like conv_fc_ring.hip:
//   * persistent workgroups, TWO per CU (64 KB of LDS each), walk contiguous runs of 4x8x8 bricks (256 voxels);
//   * the 27 weight fragments live in REGISTERS (one 32 x 16 tile per tap: 108 VGPRs), so the LDS holds nothing but two halo
//     buffers and the output staging, and an MFMA needs one LDS operand;
//   * the 6x10x10 halo of the NEXT brick arrives by LDS-DMA (per-lane source pointers, zero line outside the volume) while the
//     current brick is multiplied: one barrier per brick; the staged outputs of a brick are stored one trip later, in front of
//     the next halo request, so the counted wait at the top of a trip sees halo pieces only;
//   * 32-byte voxel rows at a d pitch of 12 rows, the two 16-byte halves XOR-ed with the halo w coordinate's low bit: conflict-free
//     for ds_read_b128 (tools/lds_conflicts.py); tap offsets are compile-time immediates (FLIP = data gradient: mirrored taps).
#include "gemm_desc.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define C16_PITCH 12
#define C16_HROWS 768                     // 6 x 10 x 12 rows = 720, padded to 24 LDS-DMA pieces of 32 rows
#define C16_HBUF (C16_HROWS * 32)
#define C16_STAGE (2 * C16_HBUF)          // 4 wave-private output tiles of 64 voxels x 64 B
#define C16_BIAS (C16_STAGE + 4 * 4096)   // 32 floats
#define C16_SMEM (C16_BIAS + 128)
#define C16_CENTER ((1 * 10 + 1) * C16_PITCH + 1)      // row of halo voxel (1, 1, 1): the largest negative tap offset

__device__ __attribute__((aligned(64))) uint32_t ltu_zero_c16[16];      // source of out-of-volume halo rows / padding rows

__device__ __forceinline__ void c16_glds16(const void* src, uint32_t lds_byte_addr) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_byte_addr);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
template <int I, int N, class F>
__device__ __forceinline__ void c16_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    c16_static_for<I + 1, N>(f);
  }
}

template <bool FLIP>
__global__ void __launch_bounds__(256, 2) conv3_c16_ring_bf16_kernel(const HaloArgs a, int bricks) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int nbh = (a.H + 3) / 4, nbw = (a.W + 7) / 8, nbd = (a.D + 7) / 8;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const char* zsrc = reinterpret_cast<const char*>(ltu_zero_c16) + (lane & 1) * 16;

  // this workgroup's run of bricks
  const int per = (bricks + (int)gridDim.x - 1) / (int)gridDim.x;
  const int b_first = (int)blockIdx.x * per;
  const int b_end = min(bricks, b_first + per);
  if (b_first >= b_end) return;

  // ---- weights -> registers: tap t, lane (n = li, k half = lh): 8 consecutive input channels of row n of tile t --------------------
  bf16x8 wreg[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (li < a.N) v = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(a.w) + ((long long)li * 27 + t) * 16 + lh * 8);
    wreg[t] = __builtin_bit_cast(bf16x8, v);
  }
  // ---- halo pieces of this lane: 6 per wave, 32 rows each; row -> (hh, hw, hd) with hd < 12 (10, 11: padding rows); the 16-byte slot
  // (lane & 1) holds channel half (lane & 1) ^ (hw & 1), which comes from x0 (channels < c0) or x1 -----------------------------------
  const int prow = lane >> 1;
  const char* hbase[6];                    // source pointer of the piece for a brick at the volume origin (may point in front of the tensor)
  int hpos[6];                             // hh | hw << 8 | hd << 16 | valid << 24 | second source << 25
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    const int row = (wave * 6 + s) * 32 + prow;
    const int hd = row % C16_PITCH, hw = (row / C16_PITCH) % 10, hh = row / (10 * C16_PITCH);
    const int c = ((lane & 1) ^ (hw & 1)) * 8;
    const bool s1 = c >= a.c0;
    const long long rel = ((long long)(hh - 1) * a.W + (hw - 1)) * a.D + (hd - 1);
    hbase[s] = s1 ? reinterpret_cast<const char*>(a.x1) + ((c - a.c0) + rel * a.lda1) * 2 : reinterpret_cast<const char*>(a.x0) + (c + rel * a.lda0) * 2;
    hpos[s] = hh | (hw << 8) | (hd << 16) | ((row < 6 * 10 * C16_PITCH && hd < 10 ? 1 : 0) << 24) | ((s1 ? 1 : 0) << 25);
  }
  // brick coordinates advance incrementally along the run (the divisions of a decomposition per brick - twice: request and store -
  // were a seventh of a trip, in-kernel clock stamps)
  struct Coord { int b, h0, w0, d0; };
  auto decompose = [&](int brick) {
    int t = brick;
    Coord c;
    const int bd = t % nbd; t /= nbd;
    const int bw = t % nbw; t /= nbw;
    const int bh = t % nbh;
    c.b = t / nbh; c.h0 = bh * 4; c.w0 = bw * 8; c.d0 = bd * 8;
    return c;
  };
  auto advance = [&](Coord c) {
    c.d0 += 8;
    if (c.d0 >= nbd * 8) {
      c.d0 = 0; c.w0 += 8;
      if (c.w0 >= nbw * 8) {
        c.w0 = 0; c.h0 += 4;
        if (c.h0 >= nbh * 4) { c.h0 = 0; c.b += 1; }
      }
    }
    return c;
  };
  auto issue_halo = [&](const Coord& cc, int buf) {
    const int b = cc.b, h0 = cc.h0, w0 = cc.w0, d0 = cc.d0;
    const long long vox0 = (((long long)b * a.H + h0) * a.W + w0) * a.D + d0;
    const long long off0 = vox0 * a.lda0 * 2, off1 = vox0 * a.lda1 * 2;          // wave-uniform byte offsets of the brick origin
    const uint32_t hb = lds0 + buf * C16_HBUF + wave * 6 * 1024;
    if (h0 >= 1 && h0 + 5 <= a.H && w0 >= 1 && w0 + 9 <= a.W && d0 >= 1 && d0 + 9 <= a.D) {      // (uniform) the whole halo is inside the volume
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        const char* src = ((hpos[s] >> 24) & 1) ? hbase[s] + (((hpos[s] >> 25) & 1) ? off1 : off0) : zsrc;
        c16_glds16(src, hb + s * 1024);
      }
      return;
    }
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      const int h = h0 - 1 + (hpos[s] & 255), w = w0 - 1 + ((hpos[s] >> 8) & 255), d = d0 - 1 + ((hpos[s] >> 16) & 255);
      const bool in = ((hpos[s] >> 24) & 1) != 0 && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W && (unsigned)d < (unsigned)a.D;
      const char* src = in ? hbase[s] + (((hpos[s] >> 25) & 1) ? off1 : off0) : zsrc;
      c16_glds16(src, hb + s * 1024);
    }
  };

  // ---- fragment read addresses: tile i (w 0-3 / 4-7 of h-plane `wave`), lane (li: w = li >> 3, d = li & 7; lh: channel half);
  // tap (dh, dw, dd) adds the immediate ((dh * 10 + dw) * 12 + dd + CENTER) * 32; the slot depends on the parity of hw = 4 i + (li >> 3) + dw + 1
  const int hwl = li >> 3;
  int baseA[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row0 = ((wave + 1) * 10 + (i * 4 + hwl + 1)) * C16_PITCH + (li & 7) + 1;
#pragma unroll
    for (int dwi = 0; dwi < 3; ++dwi) baseA[i][dwi] = (row0 - C16_CENTER) * 32 + ((lh ^ ((hwl + dwi) & 1)) << 4);
  }

  // bias -> LDS (read back in the epilogue: 16 registers fewer than holding it)
  if (tid < 32) reinterpret_cast<float*>(smem + C16_BIAS)[tid] = (a.bias != nullptr && tid < a.N) ? a.bias[tid] : 0.f;
  // ---- output pieces of this lane: the wave's 64 voxels (h-plane `wave`) x 4 parts of 8 channels: 4 pieces per lane ----------------
  char* obase[4];                           // destination of the piece for a brick at the volume origin
  int opos[4];                              // w | d << 8 | valid << 16 | second destination << 17
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int idx = it * 64 + lane, vox = idx >> 2, part = idx & 3;
    const int n = part * 8;
    const bool d1 = n >= a.n0;
    const long long rel = (long long)(vox >> 3) * a.D + (vox & 7);
    obase[it] = d1 ? reinterpret_cast<char*>(a.o1) + ((n - a.n0) + rel * a.ldo1) * 2 : reinterpret_cast<char*>(a.o0) + (n + rel * a.ldo0) * 2;
    opos[it] = (vox >> 3) | ((vox & 7) << 8) | ((n < a.N ? 1 : 0) << 16) | ((d1 ? 1 : 0) << 17);
  }
  char* const stage = smem + C16_STAGE + wave * 4096;

  auto store_brick = [&](const Coord& cc) {
    const int b = cc.b, h0 = cc.h0, w0 = cc.w0, d0 = cc.d0;
    const long long vox0 = ((((long long)b * a.H + h0 + wave) * a.W + w0) * a.D + d0);
    const long long off0 = vox0 * a.ldo0 * 2, off1 = vox0 * a.ldo1 * 2;
    const bool hok = h0 + wave < a.H;
    const bool whole = w0 + 8 <= a.W && d0 + 8 <= a.D;            // (uniform) no ragged edge in w or d
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = it * 64 + lane, vox = idx >> 2, part = idx & 3;
      const uint4 v = *reinterpret_cast<const uint4*>(stage + vox * 64 + ((part ^ ((vox >> 1) & 3)) << 4));
      const int ow = opos[it] & 255, od = (opos[it] >> 8) & 255;
      if (hok && ((opos[it] >> 16) & 1) != 0 && (whole || (w0 + ow < a.W && d0 + od < a.D)))
        *reinterpret_cast<uint4*>(obase[it] + (((opos[it] >> 17) & 1) ? off1 : off0)) = v;
    }
  };
  Coord cur = decompose(b_first), prev = cur, next = advance(cur);
  issue_halo(cur, 0);
  int buf = 0;
  for (int brick = b_first; brick < b_end; ++brick) {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // this brick's halo has landed, everybody has left the other buffer
    if (brick > b_first) store_brick(prev);
    if (brick + 1 < b_end) issue_halo(next, buf ^ 1);
    prev = cur; cur = next; next = advance(next);
    const int hoff = buf * C16_HBUF;
    int bA[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int dwi = 0; dwi < 3; ++dwi) bA[i][dwi] = baseA[i][dwi] + hoff;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    auto load_frags = [&](auto TT, bf16x8 (&af)[2]) {
      constexpr int t = decltype(TT)::value;
      constexpr int th = t / 9, tw = (t / 3) % 3, td = t % 3;
      constexpr int dh = FLIP ? 1 - th : th - 1, dw = FLIP ? 1 - tw : tw - 1, dd = FLIP ? 1 - td : td - 1;
      constexpr int immA = (((dh * 10 + dw) * C16_PITCH + dd) + C16_CENTER) * 32;
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8*>(smem + bA[i][dw + 1] + immA);
    };
    bf16x8 afA[2], afB[2];
    load_frags(std::integral_constant<int, 0>{}, afA);
    c16_static_for<0, 14>([&](auto TP) {
      constexpr int t = decltype(TP)::value * 2;
      if constexpr (t + 1 < 27) load_frags(std::integral_constant<int, t + 1>{}, afB);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[t], afA[i], acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (t + 2 < 27) load_frags(std::integral_constant<int, t + 2>{}, afA);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (t + 1 < 27) {
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[t + 1], afB[i], acc[i], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    // wave-private staging (the LDS executes one wave's instructions in order: no barrier between these writes and the reads of
    // store_brick in the next trip)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const float4 bv = *reinterpret_cast<const float4*>(smem + C16_BIAS + (8 * rr + 4 * lh) * 4);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int vox = i * 32 + li;
        uint2 pk;
        pk.x = pack_bf16x2(acc[i][4 * rr + 0] + bv.x, acc[i][4 * rr + 1] + bv.y);
        pk.y = pack_bf16x2(acc[i][4 * rr + 2] + bv.z, acc[i][4 * rr + 3] + bv.w);
        *reinterpret_cast<uint2*>(stage + vox * 64 + ((rr ^ ((vox >> 1) & 3)) << 4) + lh * 8) = pk;
      }
    }
    buf ^= 1;
  }
  store_brick(prev);
}

// LTU_OK after launching, or 1 when the shape is not handled here (the caller keeps the first-generation kernels)
int launch_conv_c16_ring_bf16(const HaloArgs& a, hipStream_t st) {
  if (a.C != 16 || a.N > 32 || a.N % 8 || a.n0 % 8 || a.c0 % 8 || a.lda0 % 8 || a.lda1 % 8 || a.ldo0 % 8 || a.ldo1 % 8) return 1;
  if (a.H < 2 || a.W < 4 || a.D < 4 || a.part != nullptr) return 1;
  const long long bricks = (long long)a.B * ((a.H + 3) / 4) * ((a.W + 7) / 8) * ((a.D + 7) / 8);
  if (bricks >= (1LL << 31) || bricks < 128) return 1;        // tiny grids: the generic path splits channels over workgroups
  static LtuDevOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_c16_ring_bf16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, C16_SMEM);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_c16_ring_bf16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, C16_SMEM);
  }
  const int want = ltu_knob_pos("LTU_C16_RING_BLOCKS", 512);
  const int nblk = (int)(bricks < want ? bricks : want);
  if (a.flip) hipLaunchKernelGGL(conv3_c16_ring_bf16_kernel<true>, dim3(nblk), dim3(256), C16_SMEM, st, a, (int)bricks);
  else hipLaunchKernelGGL(conv3_c16_ring_bf16_kernel<false>, dim3(nblk), dim3(256), C16_SMEM, st, a, (int)bricks);
  return ltu_check_launch();
}
