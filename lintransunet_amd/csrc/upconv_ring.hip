// Sub-pixel un-embedding forward (nearest x2 + 3x3x3 conv, model/Unet_3Dblock.py:419-432) - second generation.
//
// Same decomposition as conv_class_ring_bf16_kernel (conv_halo.hip): a workgroup owns a 4x8x8 brick of COARSE voxels (4 waves x
// 64 rows), stages the 6x10x10 halo of one 32-channel chunk by LDS-DMA (double-buffered), streams the 64 (class, slot) weight
// tiles of the chunk through a 4-deep ring of 16 KB stages (one stage = the 8 slots of one class) and keeps all 8 class
// accumulators (8 x 2 tiles of 32 n x 32 voxels) in AGPRs.  What changed (round 4): the first generation was INSTRUCTION-ISSUE
// bound, not MFMA-, LDS- or memory-bound - ablation showed the empty loop skeleton (no DMA, no LDS reads, no MFMAs, no epilogue)
// taking 45-50 of the kernel's 117-140 us: per (class, slot) entry it spent ~12 VALU instructions on LDS addresses (run-time
// offsets through v_readlane), per stage ~150 on weight source addresses, per chunk ~250 on halo coordinates, and the epilogue
// ran 8 x (two barriers + 8-byte stores with per-store voxel arithmetic).  Here
//   * the (class, slot) -> (dh, dw, dd) pattern of the un-embedding is a compile-time function, so every fragment read is
//     `ds_read_b128 v, vbase offset:imm`: 12 halo base addresses (tile x k-half x dw swizzle) and 2 weight base addresses are
//     computed once; an entry is 6 LDS reads + 4 MFMAs and nothing else;
//   * LDS-DMA source addresses are per-lane constants + one scalar per stage / chunk (computed once in the prologue);
//   * the epilogue stages all 8 classes at once (the operand buffers are dead by then) and stores 16 bytes per lane with
//     per-thread voxel offsets computed once: 2 barriers instead of 16.
// Weights: wsub_f [8 classes][Co][8 slots][Ci] (weight-prep kind 5).  Output y [B][2H][2W][2D][Co].
#include "gemm_desc.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define UR_HROWS 640                      // 6 x 10 x 10 halo voxels padded to 40 LDS-DMA pieces of 16 rows
#define UR_HBUF (UR_HROWS * 64)           // bytes per halo buffer (64-byte rows: 32 channels)
#define UR_RING (2 * UR_HBUF)             // byte offset of the weight ring
#define UR_WSTAGE 16384                   // 8 slots x 32 n x 64 bytes
#define UR_SMEM (UR_RING + 4 * UR_WSTAGE)

__device__ __attribute__((aligned(64))) uint32_t ltu_zero_wide[512];      // 2 KB of zeros: source of out-of-volume halo rows (any chunk)

__host__ __device__ constexpr int ur_off(int p, int a) { return p == 0 ? (a == 0 ? -1 : 0) : (a == 0 ? 0 : 1); }

__device__ __forceinline__ void ur_glds16(const uint16_t* src, uint32_t lds_byte_addr) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_byte_addr);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
template <int N>
__device__ __forceinline__ void ur_sync() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}
template <int I, int N, class F>
__device__ __forceinline__ void ur_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    ur_static_for<I + 1, N>(f);
  }
}

struct UpRingArgs {
  const uint16_t* x;
  const uint16_t* w;
  const float* bias;
  uint16_t* y;
  int B, H, W, D, Ci, Co;
};

__global__ void __launch_bounds__(256) upconv_ring_bf16_kernel(const UpRingArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int nbh = (a.H + 3) / 4, nbw = (a.W + 7) / 8, nbd = (a.D + 7) / 8;
  int bid = blockIdx.x;
  const int bd = bid % nbd; bid /= nbd;
  const int bw = bid % nbw; bid /= nbw;
  const int bh = bid % nbh;
  const int b = bid / nbh;
  const int h0 = bh * 4, w0 = bw * 8, d0 = bd * 8;
  const int n_blk = blockIdx.y * 32;
  const int nchunk = a.Ci / 32;
  const int total = nchunk * 8;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;

  // ---- per-lane constants of the LDS-DMA pieces (piece = 16 rows x 64 B; lane -> row lane >> 2, 16-byte slot lane & 3) ----------
  const int prow = lane >> 2;
  const uint16_t* hsrc[10];               // halo piece s of this wave, chunk 0 (out-of-volume rows: the zero line)
#pragma unroll
  for (int s = 0; s < 10; ++s) {
    const int hv = (wave * 10 + s) * 16 + prow;
    const int hd = hv % 10, hw = (hv / 10) % 10, hh = hv / 100;
    const int h = h0 - 1 + hh, w = w0 - 1 + hw, d = d0 - 1 + hd;
    const int lc = (lane & 3) ^ (hw & 3);                    // slot = channel quarter ^ (halo w & 3): conflict-free fragment reads
    const bool in = hv < 600 && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W && (unsigned)d < (unsigned)a.D;
    hsrc[s] = in ? a.x + ((((long long)b * a.H + h) * a.W + w) * a.D + d) * a.Ci + lc * 8
                 : reinterpret_cast<const uint16_t*>(ltu_zero_wide) + (lane & 3) * 8;
  }
  int woff[4];                            // weight piece s of a stage: rows (wave * 4 + s) * 16 + prow = slot t * 32 + n
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int row = (wave * 4 + s) * 16 + prow;
    const int t = row >> 5, n = n_blk + (row & 31);
    const int wchunk = (lane & 3) ^ ((lane >> 4) & 3);      // slot = channel quarter ^ ((row >> 2) & 3)
    woff[s] = (n * 8 + t) * a.Ci + wchunk * 8;
  }
  const int wcls = a.Co * 8 * a.Ci;       // elements per class in wsub_f

  auto issue_halo = [&](int chunk) {
    const uint32_t hb = lds0 + (chunk & 1) * UR_HBUF + wave * 10 * 1024;
#pragma unroll
    for (int s = 0; s < 10; ++s) ur_glds16(hsrc[s] + chunk * 32, hb + s * 1024);
  };
  auto issue_w = [&](int g) {             // stage g = (chunk g >> 3, class g & 7)
    const uint16_t* wsrc = a.w + (long long)(g & 7) * wcls + (g >> 3) * 32;
    const uint32_t wb = lds0 + UR_RING + (g & 3) * UR_WSTAGE + wave * 4 * 1024;
#pragma unroll
    for (int s = 0; s < 4; ++s) ur_glds16(wsrc + woff[s], wb + s * 1024);
  };

  // ---- fragment read addresses ---------------------------------------------------------------------------------------------------
  // wave w owns h-plane w of the brick (64 rows); tile i covers w positions 4i..4i+3; lane li -> (w 4i + (li >> 3), d li & 7).
  // halo voxel of (row, offset) = hv0[i] + ((dh * 10 + dw) * 10 + dd); its 16-byte slot for k-half ks is (2 ks + lh) ^ ((hwl + dw + 1) & 3)
  const int hwl = li >> 3;
  int baseA[2][2][3];                     // [tile][ks][dw + 1]: byte address inside a halo buffer, biased by -111 rows
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
  for (int ks = 0; ks < 2; ++ks) baseW[ks] = UR_RING + li * 64 + (((ks * 2 + lh) ^ ((li >> 2) & 3)) << 4);

  f32x16 acc[8][2];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][i][r] = 0.f;

  issue_halo(0);
  for (int g = 0; g < 3 && g < total; ++g) issue_w(g);
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    const int hoff = (chunk & 1) * UR_HBUF;
    int bA[2][2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int dwi = 0; dwi < 3; ++dwi) bA[i][ks][dwi] = baseA[i][ks][dwi] + hoff;
    ur_static_for<0, 8>([&](auto ST) {
      constexpr int c = decltype(ST)::value;           // stage = class c; ring slot c & 3 (8 stages per chunk = 2 x the ring depth)
      const int g = chunk * 8 + c;
      if (g + 2 < total) {
        // LDS-DMA pieces issued after W(g): W(g+1), W(g+2) (4 each) and the halo of the next chunk (10) when it was issued
        // at stages 0 (iterations g-3 .. g-1 with class 0)
        constexpr bool near0 = c >= 1 && c <= 3;
        if (near0 && chunk + 1 < nchunk) ur_sync<18>(); else ur_sync<8>();
      } else {
        ur_sync<0>();
      }
      if (g + 3 < total) issue_w(g + 3);
      if (c == 0 && chunk + 1 < nchunk) issue_halo(chunk + 1);
      constexpr int ph = c >> 2, pw = (c >> 1) & 1, pd = c & 1;
      auto load_frags = [&](auto TT, bf16x8 (&af)[2][2], bf16x8 (&wf)[2]) {
        constexpr int t = decltype(TT)::value;
        constexpr int dh = ur_off(ph, t >> 2), dw = ur_off(pw, (t >> 1) & 1), dd = ur_off(pd, t & 1);
        constexpr int immA = (((dh * 10 + dw) * 10 + dd) + 111) * 64;
        constexpr int immW = (c & 3) * UR_WSTAGE + t * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) af[i][ks] = *reinterpret_cast<const bf16x8*>(smem + bA[i][ks][dw + 1] + immA);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wf[ks] = *reinterpret_cast<const bf16x8*>(smem + baseW[ks] + immW);
      };
      // transposed product D[n][voxel]: a lane owns 4 consecutive n of one voxel
      auto mma = [&](const bf16x8 (&af)[2][2], const bf16x8 (&wf)[2]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[c][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], af[i][ks], acc[c][i], 0, 0, 0);
      };
      bf16x8 afA[2][2], wfA[2], afB[2][2], wfB[2];
      load_frags(std::integral_constant<int, 0>{}, afA, wfA);
      ur_static_for<0, 4>([&](auto TP) {
        constexpr int t = decltype(TP)::value * 2;
        load_frags(std::integral_constant<int, t + 1>{}, afB, wfB);
        __builtin_amdgcn_sched_barrier(0);
        mma(afA, wfA);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (t + 2 < 8) load_frags(std::integral_constant<int, t + 2>{}, afA, wfA);
        __builtin_amdgcn_sched_barrier(0);
        mma(afB, wfB);
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  }

  // ---- epilogue: all 8 classes staged at once ([class][voxel 256][64 B], 16-byte parts XOR-ed with (voxel >> 1) & 3), then
  // 16 bytes per lane to the fine grid -------------------------------------------------------------------------------------------
  float4 bv4[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int n = n_blk + 8 * rr + 4 * lh;
    bv4[rr] = a.bias != nullptr ? *reinterpret_cast<const float4*>(a.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();                         // every wave is done with the operand buffers (all DMA landed: the last stages waited vmcnt(0))
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int vox = wave * 64 + i * 32 + li;
      char* row = smem + (c * 256 + vox) * 64;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        uint2 pk;
        pk.x = pack_bf16x2(acc[c][i][4 * rr + 0] + bv4[rr].x, acc[c][i][4 * rr + 1] + bv4[rr].y);
        pk.y = pack_bf16x2(acc[c][i][4 * rr + 2] + bv4[rr].z, acc[c][i][4 * rr + 3] + bv4[rr].w);
        // n = 8 rr + 4 lh .. + 3  ->  16-byte part rr, half lh
        *reinterpret_cast<uint2*>(row + ((rr ^ ((vox >> 1) & 3)) << 4) + lh * 8) = pk;
      }
    }
  __syncthreads();
  // thread -> (16-byte part tid & 3, voxel (tid >> 2) + 64 k for k = 0..3); the class is uniform per iteration
  const int part = tid & 3;
  const int Hh = 2 * a.H, Wh = 2 * a.W, Dh = 2 * a.D;
  long long vbase[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int ml = (tid >> 2) + 64 * k;
    const int qh = h0 + (ml >> 6), qw = w0 + ((ml >> 3) & 7), qd = d0 + (ml & 7);
    const bool ok = qh < a.H && qw < a.W && qd < a.D;
    vbase[k] = ok ? ((((long long)b * Hh + 2 * qh) * Wh + 2 * qw) * Dh + 2 * qd) * a.Co + n_blk + part * 8 : -1;
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const long long coff = ((long long)((c >> 2) * Wh + ((c >> 1) & 1)) * Dh + (c & 1)) * a.Co;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ml = (tid >> 2) + 64 * k;
      const uint4 v = *reinterpret_cast<const uint4*>(smem + (c * 256 + ml) * 64 + ((part ^ ((ml >> 1) & 3)) << 4));
      if (vbase[k] >= 0) *reinterpret_cast<uint4*>(a.y + vbase[k] + coff) = v;
    }
  }
}

// LTU_OK after launching, or 1 when the shape is not handled (the caller keeps the generic class kernel)
int launch_upconv_ring_bf16(const void* x, const void* wsub_f, const float* bias, void* y, int B, int H, int W, int D, int Ci, int Co,
                            hipStream_t st) {
  if (Ci % 32 || Co % 32 || Ci > 512 || H < 2 || W < 4 || D < 2) return 1;
  const long long rb = (long long)B * ((H + 3) / 4) * ((W + 7) / 8) * ((D + 7) / 8);
  if (rb >= (1LL << 31) || (long long)8 * Co * 8 * Ci >= (1LL << 31)) return 1;
  static LtuDevOnce attr_once;
  if (attr_once.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&upconv_ring_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, UR_SMEM);
  UpRingArgs a;
  a.x = (const uint16_t*)x; a.w = (const uint16_t*)wsub_f; a.bias = bias; a.y = (uint16_t*)y;
  a.B = B; a.H = H; a.W = W; a.D = D; a.Ci = Ci; a.Co = Co;
  hipLaunchKernelGGL(upconv_ring_bf16_kernel, dim3((unsigned)rb, Co / 32), dim3(256), UR_SMEM, st, a);
  return ltu_check_launch();
}
