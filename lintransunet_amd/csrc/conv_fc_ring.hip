// Stride-1 3x3x3 convolution (forward and data gradient) of the few-channel U-Net levels, C = 32 input channels (possibly a virtual
// concat of two sources) and N <= 32 outputs (possibly split over two destinations: conv pairs) - model/Unet_3Dblock.py:325-341,
// 540-557 at 32x32x128 (32 -> 32) and 64x64x128 (16 + 16 -> 16, 32 -> 16 + head).  Second generation of conv3_halo_ws_bf16_kernel,
// built like upconv_ring.hip / sdgrad_ring.hip:
//   * persistent workgroups (one per CU) walk contiguous runs of 4x8x8 bricks (256 voxels: a weight fragment feeds two row tiles);
//     the 27 weight tiles (54 KB) are loaded into LDS once per workgroup;
//   * the 6x10x10 halo of the NEXT brick arrives by LDS-DMA (per-lane source pointers: the two concat sources, zero line outside
//     the volume) while the current brick is multiplied: one barrier per brick, waited with a counted vmcnt that leaves the
//     previous brick's stores in flight;
//   * tap offsets are compile-time (FLIP = data gradient: mirrored taps): a tap is 4 + 2 ds_read_b128 with immediate offsets and
//     4 MFMAs; outputs leave through a wave-private 4 KB staging tile as 16-byte stores.
// Measured against the first generation at the shapes of the step: see profiles/r04_microbench.txt (bench_conv.py).
#include "gemm_desc.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define FC_HROWS 640                      // 6 x 10 x 10 halo voxels padded to 40 LDS-DMA pieces of 16 rows
#define FC_HBUF (FC_HROWS * 64)
#define FC_W (2 * FC_HBUF)                // 27 weight tiles of 32 n x 64 B
#define FC_STAGE (FC_W + 27 * 2048)       // 4 wave-private output tiles of 64 voxels x 64 B
#define FC_SMEM (FC_STAGE + 4 * 4096)

__device__ __attribute__((aligned(64))) uint32_t ltu_zero_fc[32];      // source of out-of-volume halo rows / padding weight rows

__device__ __forceinline__ void fc_glds16(const void* src, uint32_t lds_byte_addr) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_byte_addr);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
template <int I, int N, class F>
__device__ __forceinline__ void fc_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    fc_static_for<I + 1, N>(f);
  }
}

template <bool FLIP>
__global__ void __launch_bounds__(256) conv3_fc_ring_bf16_kernel(const HaloArgs a, int bricks) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int nbh = (a.H + 3) / 4, nbw = (a.W + 7) / 8, nbd = (a.D + 7) / 8;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const char* zsrc = reinterpret_cast<const char*>(ltu_zero_fc) + (lane & 3) * 16;

  // this workgroup's run of bricks
  const int per = (bricks + (int)gridDim.x - 1) / (int)gridDim.x;
  const int b_first = (int)blockIdx.x * per;
  const int b_end = min(bricks, b_first + per);
  if (b_first >= b_end) return;

  // ---- weights -> LDS once (27 tiles x 2 pieces of 16 rows; slot = channel quarter ^ ((row >> 2) & 3)) ----------------------------
  const int prow = lane >> 2;
  {
    const int wchunk = (lane & 3) ^ ((lane >> 4) & 3);
    for (int p = wave; p < 54; p += 4) {
      const int t = p >> 1, n = (p & 1) * 16 + prow;
      const char* src = n < a.N ? reinterpret_cast<const char*>(a.w) + (((long long)n * 27 + t) * 32 + wchunk * 8) * 2 : zsrc;
      fc_glds16(src, lds0 + FC_W + p * 1024);
    }
  }
  // ---- halo pieces of this lane: 10 per wave; row hv -> (hh, hw, hd); 16-byte slot lane & 3 holds channel quarter (lane & 3) ^ (hw & 3),
  // which comes from x0 (channels < c0) or x1 ---------------------------------------------------------------------------------------
  const char* hbase[10];                   // source pointer of the piece at the volume origin, minus nothing: + voxel offset * lda * 2
  int hrel[10], hpos[10], hld[10];         // voxel offset relative to the brick origin; hh | hw << 8 | hd << 16 | valid << 24; row pitch (bytes)
#pragma unroll
  for (int s = 0; s < 10; ++s) {
    const int hv = (wave * 10 + s) * 16 + prow;
    const int hd = hv % 10, hw = (hv / 10) % 10, hh = hv / 100;
    const int c = (((lane & 3) ^ (hw & 3))) * 8;
    const bool s1 = c >= a.c0;
    hbase[s] = s1 ? reinterpret_cast<const char*>(a.x1) + (c - a.c0) * 2 : reinterpret_cast<const char*>(a.x0) + c * 2;
    hld[s] = (s1 ? a.lda1 : a.lda0) * 2;
    hrel[s] = ((hh - 1) * a.W + (hw - 1)) * a.D + (hd - 1);
    hpos[s] = hh | (hw << 8) | (hd << 16) | ((hv < 600 ? 1 : 0) << 24);
  }
  auto decompose = [&](int brick, int& b, int& h0, int& w0, int& d0) {
    int t = brick;
    const int bd = t % nbd; t /= nbd;
    const int bw = t % nbw; t /= nbw;
    const int bh = t % nbh;
    b = t / nbh; h0 = bh * 4; w0 = bw * 8; d0 = bd * 8;
  };
  auto issue_halo = [&](int brick, int buf) {
    int b, h0, w0, d0;
    decompose(brick, b, h0, w0, d0);
    const long long vox0 = (((long long)b * a.H + h0) * a.W + w0) * a.D + d0;
    const uint32_t hb = lds0 + buf * FC_HBUF + wave * 10 * 1024;
#pragma unroll
    for (int s = 0; s < 10; ++s) {
      const int h = h0 - 1 + (hpos[s] & 255), w = w0 - 1 + ((hpos[s] >> 8) & 255), d = d0 - 1 + ((hpos[s] >> 16) & 255);
      const bool in = (hpos[s] >> 24) != 0 && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W && (unsigned)d < (unsigned)a.D;
      const char* src = in ? hbase[s] + (vox0 + hrel[s]) * hld[s] : zsrc;
      fc_glds16(src, hb + s * 1024);
    }
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
  for (int ks = 0; ks < 2; ++ks) baseW[ks] = FC_W + li * 64 + (((ks * 2 + lh) ^ ((li >> 2) & 3)) << 4);

  float4 bv4[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int n = 8 * rr + 4 * lh;
    bv4[rr] = (a.bias != nullptr && n < a.N) ? *reinterpret_cast<const float4*>(a.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // ---- output pieces of this lane: the wave's 64 voxels (h-plane `wave`) x 4 parts of 8 channels: 4 pieces per lane ----------------
  char* obase[4];
  int opos[4], old_[4];                     // w | d << 8 | valid << 16;  row pitch (bytes)
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int idx = it * 64 + lane, vox = idx >> 2, part = idx & 3;
    const int n = part * 8;
    const bool d1 = n >= a.n0;
    obase[it] = d1 ? reinterpret_cast<char*>(a.o1) + (n - a.n0) * 2 : reinterpret_cast<char*>(a.o0) + n * 2;
    old_[it] = (d1 ? a.ldo1 : a.ldo0) * 2;
    opos[it] = (vox >> 3) | ((vox & 7) << 8) | ((n < a.N ? 1 : 0) << 16);
  }
  char* const stage = smem + FC_STAGE + wave * 4096;

  // the staged tile of a brick is stored one iteration later, in front of the next halo request: at the top of a loop trip the only
  // vector-memory operations in flight are then the halo pieces (issued after the previous brick's stores, which are a brick old
  // and long acknowledged), so the counted wait is exact whatever the store predicates were
  auto store_brick = [&](int brick) {
    int b, h0, w0, d0;
    decompose(brick, b, h0, w0, d0);
    const long long vox0 = ((((long long)b * a.H + h0 + wave) * a.W + w0) * a.D + d0);
    const bool hok = h0 + wave < a.H;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = it * 64 + lane, vox = idx >> 2, part = idx & 3;
      const uint4 v = *reinterpret_cast<const uint4*>(stage + vox * 64 + ((part ^ ((vox >> 1) & 3)) << 4));
      const int ow = opos[it] & 255, od = (opos[it] >> 8) & 255;
      if (hok && (opos[it] >> 16) != 0 && w0 + ow < a.W && d0 + od < a.D)
        *reinterpret_cast<uint4*>(obase[it] + (vox0 + (long long)ow * a.D + od) * old_[it]) = v;
    }
  };
  issue_halo(b_first, 0);
  int buf = 0;
  for (int brick = b_first; brick < b_end; ++brick) {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // this brick's halo (the first time: and the weights) has landed
    if (brick > b_first) store_brick(brick - 1);
    if (brick + 1 < b_end) issue_halo(brick + 1, buf ^ 1);
    const int hoff = buf * FC_HBUF;
    int bA[2][2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int dwi = 0; dwi < 3; ++dwi) bA[i][ks][dwi] = baseA[i][ks][dwi] + hoff;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    auto load_frags = [&](auto TT, bf16x8 (&af)[2][2], bf16x8 (&wf)[2]) {
      constexpr int t = decltype(TT)::value;
      constexpr int th = t / 9, tw = (t / 3) % 3, td = t % 3;
      constexpr int dh = FLIP ? 1 - th : th - 1, dw = FLIP ? 1 - tw : tw - 1, dd = FLIP ? 1 - td : td - 1;
      constexpr int immA = (((dh * 10 + dw) * 10 + dd) + 111) * 64;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) af[i][ks] = *reinterpret_cast<const bf16x8*>(smem + bA[i][ks][dw + 1] + immA);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) wf[ks] = *reinterpret_cast<const bf16x8*>(smem + baseW[ks] + t * 2048);
    };
    auto mma = [&](const bf16x8 (&af)[2][2], const bf16x8 (&wf)[2]) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], af[i][ks], acc[i], 0, 0, 0);
    };
    bf16x8 afA[2][2], wfA[2], afB[2][2], wfB[2];
    load_frags(std::integral_constant<int, 0>{}, afA, wfA);
    fc_static_for<0, 14>([&](auto TP) {
      constexpr int t = decltype(TP)::value * 2;
      if constexpr (t + 1 < 27) load_frags(std::integral_constant<int, t + 1>{}, afB, wfB);
      __builtin_amdgcn_sched_barrier(0);
      mma(afA, wfA);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (t + 2 < 27) load_frags(std::integral_constant<int, t + 2>{}, afA, wfA);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (t + 1 < 27) mma(afB, wfB);
      __builtin_amdgcn_sched_barrier(0);
    });
    // wave-private staging (the LDS executes one wave's instructions in order: no barrier between these writes and the reads of
    // store_brick in the next trip)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int vox = i * 32 + li;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        uint2 pk;
        pk.x = pack_bf16x2(acc[i][4 * rr + 0] + bv4[rr].x, acc[i][4 * rr + 1] + bv4[rr].y);
        pk.y = pack_bf16x2(acc[i][4 * rr + 2] + bv4[rr].z, acc[i][4 * rr + 3] + bv4[rr].w);
        *reinterpret_cast<uint2*>(stage + vox * 64 + ((rr ^ ((vox >> 1) & 3)) << 4) + lh * 8) = pk;
      }
    }
    buf ^= 1;
  }
  store_brick(b_end - 1);
}

// LTU_OK after launching, or 1 when the shape is not handled here (the caller keeps the first-generation kernels)
int launch_conv_fc_ring_bf16(const HaloArgs& a, hipStream_t st) {
  if (a.C != 32 || a.N > 32 || a.N % 8 || a.n0 % 8 || a.c0 % 8 || a.lda0 % 8 || a.lda1 % 8 || a.ldo0 % 8 || a.ldo1 % 8) return 1;
  if (a.H < 2 || a.W < 4 || a.D < 4 || a.part != nullptr) return 1;
  const long long bricks = (long long)a.B * ((a.H + 3) / 4) * ((a.W + 7) / 8) * ((a.D + 7) / 8);
  if (bricks >= (1LL << 31) || bricks < 128) return 1;        // tiny grids: the generic path splits channels over workgroups
  static LtuDevOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_fc_ring_bf16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, FC_SMEM);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_fc_ring_bf16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, FC_SMEM);
  }
  const int nblk = (int)(bricks < 256 ? bricks : 256);
  if (a.flip) hipLaunchKernelGGL(conv3_fc_ring_bf16_kernel<true>, dim3(nblk), dim3(256), FC_SMEM, st, a, (int)bricks);
  else hipLaunchKernelGGL(conv3_fc_ring_bf16_kernel<false>, dim3(nblk), dim3(256), FC_SMEM, st, a, (int)bricks);
  return ltu_check_launch();
}
