// Data gradient of the stride-(2,2,2) / (2,2,1) 3x3x3 convolutions (the ROI token embedding model/Unet_3Dblock.py:373-385 and the
// encoder's down-sampling convs :325-341) as a class convolution with a COMPILE-TIME entry table - the second generation of
// conv_class_ring_bf16_kernel for this op (see upconv_ring.hip for what was wrong with the first: instruction issue, not MFMA).
//
//   dx[m q + p][n] = sum over the entries (dh, dw, dd, tap) of parity class p of  g[q + (dh, dw, dd)][:] . Wd[n][tap][:]
//   stride 2 axis: p = 0 -> (off 0, tap 1);  p = 1 -> (off +1, tap 0), (off 0, tap 2);   stride 1 axis: (off 1 - t, tap t), t = 0..2
// 27 entries in 8 (SD = 2) or 4 (SD = 1) classes.  A workgroup owns a 4x8x8 brick of COARSE voxels (the gradient's grid), stages
// the 5 x 9 x (9 | 10) halo of one 32-channel chunk by LDS-DMA (double-buffered) and the 27 weight tiles of the chunk as 3 stages
// of 9 through a 3-deep ring; all class accumulators (NC x 2 tiles of 32 n x 32 voxels) live in AGPRs.  An entry is 6
// ds_read_b128 with immediate offsets + 4 MFMAs; every address that depends on the lane is computed once in the prologue.
// Operands: g [B][Ho][Wo][Do][Co] (bf16), wd [CiP][27][Co] (weight-prep kind 3), dx [B][Hl][Wl][Dl][N] with N = Ci.
#include "gemm_desc.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ __attribute__((aligned(64))) uint32_t ltu_zero_sd[512];      // 2 KB of zeros: source of out-of-volume halo rows and padding tiles (any chunk)

struct SdEnt { int c, dh, dw, dd, tap; };
// entry e (0..26) of the class table, classes in (ph, pw, pd) order, entries in (h, w, d) nesting - the order ltu_conv3d_dgrad used
// for the run-time table of the first generation
template <int SD>
__host__ __device__ constexpr SdEnt sd_entry(int e) {
  int idx = 0, cls = 0;
  for (int ph = 0; ph < 2; ++ph)
    for (int pw = 0; pw < 2; ++pw)
      for (int pd = 0; pd < SD; ++pd) {
        const int nh = ph ? 2 : 1, nw = pw ? 2 : 1, nd = SD == 2 ? (pd ? 2 : 1) : 3;
        for (int i = 0; i < nh; ++i)
          for (int j = 0; j < nw; ++j)
            for (int k = 0; k < nd; ++k) {
              if (idx == e) {
                const int oh = ph ? (i == 0 ? 1 : 0) : 0, th = ph ? (i == 0 ? 0 : 2) : 1;
                const int ow = pw ? (j == 0 ? 1 : 0) : 0, tw = pw ? (j == 0 ? 0 : 2) : 1;
                const int od = SD == 2 ? (pd ? (k == 0 ? 1 : 0) : 0) : 1 - k;
                const int td = SD == 2 ? (pd ? (k == 0 ? 0 : 2) : 1) : k;
                return SdEnt{cls, oh, ow, od, (th * 3 + tw) * 3 + td};
              }
              ++idx;
            }
        ++cls;
      }
  return SdEnt{0, 0, 0, 0, -1};
}

__device__ __forceinline__ void sg_glds16(const uint16_t* src, uint32_t lds_byte_addr) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_byte_addr);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
template <int N>
__device__ __forceinline__ void sg_sync() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}
template <int I, int N, class F>
__device__ __forceinline__ void sg_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sg_static_for<I + 1, N>(f);
  }
}

struct SdgradArgs {
  const uint16_t* g;
  const uint16_t* w;
  uint16_t* dx;
  int B, Ho, Wo, Do, Co;       // coarse grid, gradient channels (the reduction)
  int Hl, Wl, Dl, N;           // fine grid, input channels
  int tap[28];                 // tap of entry e (sd_entry<SD>(e).tap; 27 = padding slot)
};

template <int SD>
__global__ void __launch_bounds__(256) sdgrad_ring_bf16_kernel(const SdgradArgs a) {
  constexpr int NC = SD == 2 ? 8 : 4;
  constexpr int HH = 5, HW = 9, HD = SD == 2 ? 9 : 10;         // halo of the 4x8x8 brick: offsets 0..+1 (stride-2 axes), -1..+1 (stride 1)
  constexpr int NR = HH * HW * HD, HP = ((NR + 15) / 16 + 3) / 4;   // halo rows; LDS-DMA pieces per wave
  constexpr int HBUF = 4 * HP * 1024;                           // bytes per halo buffer
  constexpr int WSTAGE = 20 * 1024;                             // 9 tiles of 2 KB + 1 padding tile: 20 pieces, 5 per wave
  constexpr int RING = 2 * HBUF;
  constexpr int OD0 = SD == 2 ? 0 : 1;                          // halo d origin = d0 - OD0
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int nbh = (a.Ho + 3) / 4, nbw = (a.Wo + 7) / 8, nbd = (a.Do + 7) / 8;
  int bid = blockIdx.x;
  const int bd = bid % nbd; bid /= nbd;
  const int bw = bid % nbw; bid /= nbw;
  const int bh = bid % nbh;
  const int b = bid / nbh;
  const int h0 = bh * 4, w0 = bw * 8, d0 = bd * 8;
  const int n_blk = blockIdx.y * 32;
  const int nchunk = a.Co / 32;
  const int total = nchunk * 3;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const uint16_t* zsrc = reinterpret_cast<const uint16_t*>(ltu_zero_sd) + (lane & 3) * 8;

  // ---- LDS-DMA sources: per-lane constants (piece = 16 rows x 64 B; lane -> row lane >> 2, 16-byte slot lane & 3) ---------------
  const int prow = lane >> 2;
  const uint16_t* hsrc[HP];
#pragma unroll
  for (int s = 0; s < HP; ++s) {
    const int hv = (wave * HP + s) * 16 + prow;
    const int hd = hv % HD, hw = (hv / HD) % HW, hh = hv / (HD * HW);
    const int h = h0 + hh, w = w0 + hw, d = d0 - OD0 + hd;
    const int lc = (lane & 3) ^ (hw & 3);
    const bool in = hv < NR && (unsigned)h < (unsigned)a.Ho && (unsigned)w < (unsigned)a.Wo && (unsigned)d < (unsigned)a.Do;
    hsrc[s] = in ? a.g + ((((long long)b * a.Ho + h) * a.Wo + w) * a.Do + d) * a.Co + lc * 8 : zsrc;
  }
  const uint16_t* wsrc[3][5];              // weight piece s of stage st: rows (wave * 5 + s) * 16 + prow = slot t * 32 + n
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int row = (wave * 5 + s) * 16 + prow;
    const int t = row >> 5, n = n_blk + (row & 31);
    const int wchunk = (lane & 3) ^ ((lane >> 4) & 3);
#pragma unroll
    for (int st = 0; st < 3; ++st) {
      const int tap = t < 9 ? a.tap[st * 9 + t] : 27;
      wsrc[st][s] = (tap < 27 && n < a.N) ? a.w + ((long long)n * 27 + tap) * a.Co + wchunk * 8 : zsrc;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the tap table has been read before anything asynchronous starts

  auto issue_halo = [&](int chunk) {
    const uint32_t hb = lds0 + (chunk & 1) * HBUF + wave * HP * 1024;
#pragma unroll
    for (int s = 0; s < HP; ++s) sg_glds16(hsrc[s] + chunk * 32, hb + s * 1024);
  };
  auto issue_w = [&](int chunk, auto ST) {
    constexpr int st = decltype(ST)::value;
    const uint32_t wb = lds0 + RING + st * WSTAGE + wave * 5 * 1024;
#pragma unroll
    for (int s = 0; s < 5; ++s) sg_glds16(wsrc[st][s] + chunk * 32, wb + s * 1024);
  };

  // ---- fragment read addresses: wave w = h-plane w; tile i = w positions 4i..4i+3; lane li -> (w 4i + (li >> 3), d li & 7) ----------
  const int hwl = li >> 3;
  int baseA[2][2][2];                      // [tile][ks][dw]: byte address inside a halo buffer
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int hv0 = ((wave * HW) + (i * 4 + hwl)) * HD + (li & 7);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int dw = 0; dw < 2; ++dw) baseA[i][ks][dw] = hv0 * 64 + (((ks * 2 + lh) ^ ((hwl + dw) & 3)) << 4);
  }
  int baseW[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) baseW[ks] = RING + li * 64 + (((ks * 2 + lh) ^ ((li >> 2) & 3)) << 4);

  f32x16 acc[NC][2];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][i][r] = 0.f;

  issue_halo(0);
  issue_w(0, std::integral_constant<int, 0>{});
  issue_w(0, std::integral_constant<int, 1>{});
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    const int hoff = (chunk & 1) * HBUF;
    int bA[2][2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int dw = 0; dw < 2; ++dw) bA[i][ks][dw] = baseA[i][ks][dw] + hoff;
    sg_static_for<0, 3>([&](auto ST) {
      constexpr int st = decltype(ST)::value;
      const int g = chunk * 3 + st;
      // LDS-DMA issued after W(g): W(g+1) (5 pieces) and, at stages 1 and 2, the halo of the next chunk (HP pieces; it is issued in
      // stage 0 behind W(g+2))
      if (g + 1 < total) {
        if (st != 0 && chunk + 1 < nchunk) sg_sync<5 + HP>(); else sg_sync<5>();
      } else {
        sg_sync<0>();
      }
      if (g + 2 < total) {
        if constexpr (st == 0) issue_w(chunk, std::integral_constant<int, 2>{});
        else issue_w(chunk + 1, std::integral_constant<int, st - 1>{});
      }
      if (st == 0 && chunk + 1 < nchunk) issue_halo(chunk + 1);
      constexpr int NE = 9;                 // entries of this stage: st * 9 .. st * 9 + 8 (27 = 3 x 9: no padding entry is ever used)
      auto load_frags = [&](auto TT, bf16x8 (&af)[2][2], bf16x8 (&wf)[2]) {
        constexpr int t = decltype(TT)::value;
        constexpr SdEnt en = sd_entry<SD>(st * 9 + t);
        constexpr int immA = ((en.dh * HW + en.dw) * HD + en.dd + OD0) * 64;
        constexpr int immW = st * WSTAGE + t * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) af[i][ks] = *reinterpret_cast<const bf16x8*>(smem + bA[i][ks][en.dw] + immA);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wf[ks] = *reinterpret_cast<const bf16x8*>(smem + baseW[ks] + immW);
      };
      auto mma = [&](auto TT, const bf16x8 (&af)[2][2], const bf16x8 (&wf)[2]) {
        constexpr int c = sd_entry<SD>(st * 9 + decltype(TT)::value).c;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[c][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], af[i][ks], acc[c][i], 0, 0, 0);
      };
      bf16x8 afA[2][2], wfA[2], afB[2][2], wfB[2];
      load_frags(std::integral_constant<int, 0>{}, afA, wfA);
      sg_static_for<0, (NE + 1) / 2>([&](auto TP) {
        constexpr int t = decltype(TP)::value * 2;
        if constexpr (t + 1 < NE) load_frags(std::integral_constant<int, t + 1>{}, afB, wfB);
        __builtin_amdgcn_sched_barrier(0);
        mma(std::integral_constant<int, t>{}, afA, wfA);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (t + 2 < NE) load_frags(std::integral_constant<int, t + 2>{}, afA, wfA);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (t + 1 < NE) mma(std::integral_constant<int, t + 1>{}, afB, wfB);
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  }

  // ---- epilogue: all classes staged at once ([class][voxel 256][64 B], 16-byte parts XOR-ed with (voxel >> 1) & 3), then 16 bytes
  // per lane to the fine grid ----------------------------------------------------------------------------------------------------
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int vox = wave * 64 + i * 32 + li;
      char* row = smem + (c * 256 + vox) * 64;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        uint2 pk;
        pk.x = pack_bf16x2(acc[c][i][4 * rr + 0], acc[c][i][4 * rr + 1]);
        pk.y = pack_bf16x2(acc[c][i][4 * rr + 2], acc[c][i][4 * rr + 3]);
        *reinterpret_cast<uint2*>(row + ((rr ^ ((vox >> 1) & 3)) << 4) + lh * 8) = pk;
      }
    }
  __syncthreads();
  const int part = tid & 3;
  const bool pok = n_blk + part * 8 < a.N;            // N is a multiple of 8
  long long vbase[4];
  int qq[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int ml = (tid >> 2) + 64 * k;
    const int qh = h0 + (ml >> 6), qw = w0 + ((ml >> 3) & 7), qd = d0 + (ml & 7);
    const bool ok = pok && qh < a.Ho && qw < a.Wo && qd < a.Do;
    vbase[k] = ((((long long)b * a.Hl + 2 * qh) * a.Wl + 2 * qw) * a.Dl + SD * qd) * a.N + n_blk + part * 8;
    qq[k] = ok ? (qh << 20) | (qw << 10) | qd : -1;
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int ph = SD == 2 ? c >> 2 : c >> 1, pw = SD == 2 ? (c >> 1) & 1 : c & 1, pd = SD == 2 ? c & 1 : 0;
    const long long coff = ((long long)(ph * a.Wl + pw) * a.Dl + pd) * a.N;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ml = (tid >> 2) + 64 * k;
      const uint4 v = *reinterpret_cast<const uint4*>(smem + (c * 256 + ml) * 64 + ((part ^ ((ml >> 1) & 3)) << 4));
      const int qh = qq[k] >> 20, qw = (qq[k] >> 10) & 1023, qd = qq[k] & 1023;
      if (qq[k] >= 0 && 2 * qh + ph < a.Hl && 2 * qw + pw < a.Wl && SD * qd + pd < a.Dl)
        *reinterpret_cast<uint4*>(a.dx + vbase[k] + coff) = v;
    }
  }
}

template <int SD>
static int launch_sdgrad(const SdgradArgs& a0, hipStream_t st) {
  SdgradArgs a = a0;
  for (int e = 0; e < 27; ++e) a.tap[e] = sd_entry<SD>(e).tap;
  a.tap[27] = 27;
  constexpr int HD = SD == 2 ? 9 : 10;
  constexpr int HP = ((5 * 9 * HD + 15) / 16 + 3) / 4;
  constexpr int opnd = 2 * 4 * HP * 1024 + 3 * 20 * 1024, epi = (SD == 2 ? 8 : 4) * 256 * 64;
  constexpr int smem_bytes = opnd > epi ? opnd : epi;          // the epilogue stages every class tile over the (dead) operand buffers
  static LtuDevOnce attr_once;
  if (attr_once.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sdgrad_ring_bf16_kernel<SD>), hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
  const long long rb = (long long)a.B * ((a.Ho + 3) / 4) * ((a.Wo + 7) / 8) * ((a.Do + 7) / 8);
  if (rb >= (1LL << 31)) return 1;
  hipLaunchKernelGGL(sdgrad_ring_bf16_kernel<SD>, dim3((unsigned)rb, (a.N + 31) / 32), dim3(256), smem_bytes, st, a);
  return ltu_check_launch();
}

// LTU_OK after launching, or 1 when the shape is not handled (the caller keeps the generic class kernel)
int launch_sdgrad_ring_bf16(const void* grad, const void* wd, void* dx, int B, int Hl, int Wl, int Dl, int N, int Co, int sd, hipStream_t st) {
  if (Co % 32 || Co > 512 || N % 8 || (sd != 1 && sd != 2)) return 1;
  SdgradArgs a;
  a.g = (const uint16_t*)grad; a.w = (const uint16_t*)wd; a.dx = (uint16_t*)dx;
  a.B = B; a.Hl = Hl; a.Wl = Wl; a.Dl = Dl; a.N = N; a.Co = Co;
  a.Ho = (Hl - 1) / 2 + 1; a.Wo = (Wl - 1) / 2 + 1; a.Do = (Dl - 1) / sd + 1;
  if (a.Ho < 1 || a.Wo < 1 || a.Do < 1 || a.Ho >= 1024 || a.Wo >= 1024 || a.Do >= 1024) return 1;
  if ((long long)N * 27 * Co >= (1LL << 31)) return 1;
  return sd == 2 ? launch_sdgrad<2>(a, st) : launch_sdgrad<1>(a, st);
}
