// Data gradient of the sub-pixel un-embedding (nearest x2 + 3x3x3 conv, model/Unet_3Dblock.py:419-432):
//     dx[v][n] = sum over the 8 parity classes p and 8 slots s of  g[2 (v - off(p, s)) + p][:] . Weff[p][s][:][n]
// It ran as a 64-tap gather implicit GEMM (igemm_nt, 100-150 us per launch: the largest launches of the step).  Here it is the adjoint
// twin of upconv_ring.hip: class p of the FINE gradient grid is a coarse-shaped sub-grid g_p[j] = g[2 j + p]; for a 4x8x8 brick of
// coarse voxels and a 32-channel chunk the 5x9x9 halo of ONE class (fine voxels 2 (q0 + h) - p: per-lane LDS-DMA pointers with a
// stride of two voxels) is staged, its 8 slots are multiplied - slot (a_h, a_w, a_d) reads halo voxel v + 1 - a whatever the class -
// and the next class's halo lands meanwhile (double-buffered).  A workgroup keeps 256 voxels x 128 output channels (2 x 4 tiles) in
// AGPRs; the weight tiles [128 n][32 k] of two slots form a 16 KB stage of a 4-deep LDS-DMA ring.  Every fragment read has an
// immediate offset; LDS-DMA per MFMA is the same as in the forward kernel (0.18 pieces).
// Operands: g [B][2H][2W][2D][Co], wsub_d [Ci][64][Co] (weight-prep kind 6), dx [B][H][W][D][Ci]; Ci % 128 == 0, Co % 32 == 0.
#include "gemm_desc.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define UG_HP 7                           // LDS-DMA pieces per wave and halo: 5 x 9 x 9 = 405 rows -> 26 pieces -> 28 (padding rows: zero line)
#define UG_HBUF (4 * UG_HP * 1024)
#define UG_RING (2 * UG_HBUF)
#define UG_WSTAGE 16384                   // 2 slots x 128 n x 64 B
#define UG_SMEM (UG_RING + 4 * UG_WSTAGE)

__device__ __attribute__((aligned(64))) uint32_t ltu_zero_ug[512];      // 2 KB of zeros (any chunk offset)

__device__ __forceinline__ void ug_glds16(const void* src, uint32_t lds_byte_addr) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_byte_addr);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
template <int N>
__device__ __forceinline__ void ug_sync() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}
template <int I, int N, class F>
__device__ __forceinline__ void ug_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    ug_static_for<I + 1, N>(f);
  }
}

struct UpDgradArgs {
  const uint16_t* g;
  const uint16_t* w;
  uint16_t* dx;
  int B, H, W, D, Ci, Co;
};

template <int TN>      // column tiles of 32 output channels per workgroup: 4 (128 columns) or 2 (64: small grids get twice the workgroups)
__global__ void __launch_bounds__(256) updgrad_ring_bf16_kernel(const UpDgradArgs a) {
  constexpr int BN = 32 * TN, WSLOT = BN * 64, WST = 2 * WSLOT;        // rows per slot tile, bytes per slot tile, bytes per stage
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
  const int n_blk = blockIdx.y * BN;
  const int nchunk = a.Co / 32;
  const int total = nchunk * 32;            // stages: (chunk, class p, slot pair q)
  const int Hf = 2 * a.H, Wf = 2 * a.W, Df = 2 * a.D;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const char* zsrc = reinterpret_cast<const char*>(ltu_zero_ug) + (lane & 3) * 16;
  const int prow = lane >> 2;

  // ---- halo pieces: row hv -> (hh, hw, hd) of the 5x9x9 halo; class p reads fine voxel 2 (q0 + h) - p ------------------------------
  long long hoffs[UG_HP];                  // element offset of the class-(0,0,0) source (fine voxel 2 (q0 + h)), or -1 (padding row)
  int hfc[UG_HP];                          // fine coordinates of that voxel: fh | fw << 10 | fd << 20
#pragma unroll
  for (int s = 0; s < UG_HP; ++s) {
    const int hv = (wave * UG_HP + s) * 16 + prow;
    const int hd = hv % 9, hw = (hv / 9) % 9, hh = hv / 81;
    const int fh = 2 * (h0 + hh), fw = 2 * (w0 + hw), fd = 2 * (d0 + hd);
    const int lc = (lane & 3) ^ (hw & 3);
    hoffs[s] = hv < 405 ? ((((long long)b * Hf + fh) * Wf + fw) * Df + fd) * a.Co + lc * 8 : -1;
    hfc[s] = fh | (fw << 10) | (fd << 20);
  }
  auto issue_halo = [&](int chunk, auto PP) {
    constexpr int p = decltype(PP)::value, ph = p >> 2, pw = (p >> 1) & 1, pd = p & 1;
    const uint32_t hb = lds0 + (p & 1) * UG_HBUF + wave * UG_HP * 1024;
    const long long delta = (((long long)ph * Wf + pw) * Df + pd) * a.Co - chunk * 32;     // class p: one voxel back per odd axis
#pragma unroll
    for (int s = 0; s < UG_HP; ++s) {
      const int fh = (hfc[s] & 1023) - ph, fw = ((hfc[s] >> 10) & 1023) - pw, fd = (hfc[s] >> 20) - pd;
      const bool in = hoffs[s] >= 0 && (unsigned)fh < (unsigned)Hf && (unsigned)fw < (unsigned)Wf && (unsigned)fd < (unsigned)Df;
      const char* src = in ? reinterpret_cast<const char*>(a.g + (hoffs[s] - delta)) : zsrc;
      ug_glds16(src, hb + s * 1024);
    }
  };
  // ---- weight pieces of a stage: 16 pieces = rows [slot 2][n 128] ------------------------------------------------------------------
  int woff[TN];                            // stage = [slot 2][n BN] rows = BN / 8 pieces, TN per wave
#pragma unroll
  for (int s = 0; s < TN; ++s) {
    const int pp = wave * TN + s, sl2 = pp / (BN / 16), n = n_blk + (pp % (BN / 16)) * 16 + prow;
    const int wchunk = (lane & 3) ^ ((lane >> 4) & 3);
    woff[s] = (n * 64 + sl2) * a.Co + wchunk * 8;
  }
  auto issue_w = [&](int g) {               // stage g = ((chunk * 8 + p) * 4 + q): slots 2q, 2q + 1 of class p
    const int chunk = g >> 5, e0 = (g & 31) * 2;
    const uint16_t* wsrc = a.w + e0 * a.Co + chunk * 32;
    const uint32_t wb = lds0 + UG_RING + (g & 3) * UG_WSTAGE + wave * TN * 1024;
#pragma unroll
    for (int s = 0; s < TN; ++s) ug_glds16(wsrc + woff[s], wb + s * 1024);
  };

  // ---- fragment read addresses: wave w = h-plane w; tile i = w positions 4i..4i+3; lane li -> (w 4i + (li >> 3), d li & 7) ----------
  const int hwl = li >> 3;
  int baseA[2][2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int hv0 = (wave * 9 + (i * 4 + hwl)) * 9 + (li & 7);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int dw = 0; dw < 2; ++dw) baseA[i][ks][dw] = hv0 * 64 + (((ks * 2 + lh) ^ ((hwl + dw) & 3)) << 4);
  }
  int baseW[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) baseW[ks] = UG_RING + li * 64 + (((ks * 2 + lh) ^ ((li >> 2) & 3)) << 4);

  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  issue_halo(0, std::integral_constant<int, 0>{});
  for (int g = 0; g < 3 && g < total; ++g) issue_w(g);
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    ug_static_for<0, 32>([&](auto ST) {
      constexpr int st = decltype(ST)::value, p = st >> 2, q = st & 3;
      const int g = chunk * 32 + st;
      const bool halo_next = p < 7 || chunk + 1 < nchunk;        // a halo is requested during this class (for the next class)
      if (g + 2 < total) {
        // LDS-DMA issued after W(g): W(g+1), W(g+2) (TN pieces each) and - at q = 1, 2, 3 - the next class's halo (7 pieces)
        if (q != 0 && halo_next) ug_sync<2 * TN + UG_HP>(); else ug_sync<2 * TN>();
      } else {
        ug_sync<0>();
      }
      if (g + 3 < total) issue_w(g + 3);
      if constexpr (q == 0) {
        if (halo_next) {
          if constexpr (p < 7) issue_halo(chunk, std::integral_constant<int, p + 1>{});
          else issue_halo(chunk + 1, std::integral_constant<int, 0>{});
        }
      }
      auto load_frags = [&](auto TT, bf16x8 (&af)[2][2], bf16x8 (&wf)[TN][2]) {
        constexpr int t = decltype(TT)::value, sl = 2 * q + t;              // slot (a_h, a_w, a_d): reads halo voxel v + 1 - a
        constexpr int dh = 1 - (sl >> 2), dw = 1 - ((sl >> 1) & 1), dd = 1 - (sl & 1);
        constexpr int immA = ((dh * 9 + dw) * 9 + dd) * 64 + (p & 1) * UG_HBUF;
        constexpr int immW = q * UG_WSTAGE + t * WSLOT;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) af[i][ks] = *reinterpret_cast<const bf16x8*>(smem + baseA[i][ks][dw] + immA);
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
      mma(afB, wfB);
      __builtin_amdgcn_sched_barrier(0);
    });
  }

  // ---- epilogue: [256 voxels][BN n] staged (rows of 2 BN bytes, 16-byte parts XOR-ed with the row), then 16 bytes per lane ----------
  constexpr int PARTS = BN / 8, RB = 2 * BN;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wave * 64 + i * 32 + li;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        uint2 pk;
        pk.x = pack_bf16x2(acc[i][j][4 * rr + 0], acc[i][j][4 * rr + 1]);
        pk.y = pack_bf16x2(acc[i][j][4 * rr + 2], acc[i][j][4 * rr + 3]);
        *reinterpret_cast<uint2*>(smem + row * RB + (((j * 4 + rr) ^ (row & (PARTS - 1))) << 4) + lh * 8) = pk;
      }
  }
  __syncthreads();
#pragma unroll 4
  for (int it = 0; it < PARTS; ++it) {
    const int idx = it * 256 + tid, part = idx & (PARTS - 1), row = idx / PARTS;
    const int qh = h0 + (row >> 6), qw = w0 + ((row >> 3) & 7), qd = d0 + (row & 7);
    const uint4 v = *reinterpret_cast<const uint4*>(smem + row * RB + ((part ^ (row & (PARTS - 1))) << 4));
    if (qh < a.H && qw < a.W && qd < a.D)
      *reinterpret_cast<uint4*>(a.dx + ((((long long)b * a.H + qh) * a.W + qw) * a.D + qd) * a.Ci + n_blk + part * 8) = v;
  }
}

// LTU_OK after launching, or 1 when the shape is not handled (the caller keeps the gather implicit GEMM)
int launch_updgrad_ring_bf16(const void* grad, const void* wsub_d, void* dx, int B, int H, int W, int D, int Ci, int Co, hipStream_t st) {
  if (Ci % 128 || Co % 32 || Co > 512 || H < 2 || W < 4 || D < 2 || 2 * H >= 1024 || 2 * W >= 1024 || 2 * D >= 1024) return 1;
  const long long rb = (long long)B * ((H + 3) / 4) * ((W + 7) / 8) * ((D + 7) / 8);
  if (rb >= (1LL << 31) || (long long)Ci * 64 * Co >= (1LL << 31)) return 1;
  static LtuDevOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&updgrad_ring_bf16_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, UG_SMEM);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&updgrad_ring_bf16_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, UG_SMEM);
  }
  UpDgradArgs a;
  a.g = (const uint16_t*)grad; a.w = (const uint16_t*)wsub_d; a.dx = (uint16_t*)dx;
  a.B = B; a.H = H; a.W = W; a.D = D; a.Ci = Ci; a.Co = Co;
  // small grids: 64-column tiles give twice the workgroups (the halo is staged once more, the per-workgroup chain halves)
  if (rb * (Ci / 128) < ltu_knob_pos("LTU_UPDGRAD_WIDE_MIN", 160))
    hipLaunchKernelGGL(updgrad_ring_bf16_kernel<2>, dim3((unsigned)rb, Ci / 64), dim3(256), UG_SMEM, st, a);
  else
    hipLaunchKernelGGL(updgrad_ring_bf16_kernel<4>, dim3((unsigned)rb, Ci / 128), dim3(256), UG_SMEM, st, a);
  return ltu_check_launch();
}
