// Weight gradient of the sub-pixel un-embedding (nearest x2 + 3x3x3 conv: UpEmbedBlock, model/Unet_3Dblock.py:388-432): second generation of
// upconv_wgrad_class_bf16_kernel (conv_halo.hip).  Same arithmetic - dWeff[class][slot] = sum_q G_class(q)^T X(q + off), every wave keeps
// all 64 (class, slot) products of its 16 x 16 sub-tile (v_mfma_f32_16x16x32_bf16) and folds them onto the 27 taps in registers - but the
// first generation spent 8.3 us per 4x4x8 brick on 1.7 us of MFMA work (un-embedding of ROI bridge 1: 250 us on 128 workgroups):
//   * its operands went global -> registers -> LDS between two barriers with one wave per SIMD: nothing ran beside that;
//     here a unit (HALF a brick: 2 h-planes, 64 coarse voxels; 15 KB of X halo + 32 KB of gradient tiles) arrives by LDS-DMA into a ring
//     of three 48 KB buffers, two units in flight behind the one being multiplied, one barrier per unit;
//   * its transposing reads met 2-way bank conflicts (the two 16-lane groups of a half-wave read rows 8 apart / one w-step apart: the
//     same banks at any row pitch that LDS-DMA can write): here the 16-byte parts of a row are stored with their 32-byte halves swapped
//     on every other 8-row block (gradient) / every other w (halo) - the lanes fetch the swapped part, the readers address it.
#include "gemm_desc.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float uw_f32x4;
typedef __attribute__((ext_vector_type(4))) short uw16x4;
typedef __attribute__((address_space(3))) uw16x4 lds_uw16x4;

#define UW_XROWS 240                       // halo of a unit: 4 x 6 x 10 coarse voxels
#define UW_GOFF (UW_XROWS * 32)            // element offset of the gradient tiles [8 classes][64 rows][32] inside a unit buffer
#define UW_BUF (48 * 512)                  // elements per unit buffer: 15 + 32 pieces of 16 rows x 32 + one spare piece = 48 KB
#define UW_NB 3
#define UW_PIECES 12                       // LDS-DMA pieces per wave and unit

__device__ __attribute__((aligned(64))) uint32_t ltu_zero_uw[16];

__device__ __forceinline__ void uw_glds16(const void* src, uint32_t lds_byte_addr) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_byte_addr);
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
typedef __attribute__((ext_vector_type(4))) int uw_i32x4;
// in-place accumulation on AGPRs: the builtin lets the register allocator rename the accumulator per MFMA and copy 256 registers back
// at the loop's back edge (993 v_accvgpr moves per 128 MFMAs in the first version of this kernel, 1 179 per 256 in the first generation)
__device__ __forceinline__ void uw_mfma(uw_f32x4& c, const uw_i32x4& a, const uw_i32x4& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
template <int I, int N, class F>
__device__ __forceinline__ void uw_sfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    uw_sfor<I + 1, N>(f);
  }
}
template <int N>
__device__ __forceinline__ void uw_sync() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

__global__ void __launch_bounds__(256, 1) upconv_wgrad_ring_bf16_kernel(const UpWgradArgs a) {
  extern __shared__ __attribute__((aligned(1024))) uint16_t smem[];      // [UW_NB][UW_BUF]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wc = wave & 1;                 // 16-column sub-tile of Co / of the Ci chunk
  const int chunk = blockIdx.x, n_blk = blockIdx.y * 32;
  const int nbh = (a.H + 3) / 4, nbw = (a.W + 3) / 4, nbd = (a.D + 7) / 8;
  const int unit_lo = 2 * blockIdx.z * a.bricks_per_split;
  int unit_hi = unit_lo + 2 * a.bricks_per_split;
  if (unit_hi > 2 * a.bricks) unit_hi = 2 * a.bricks;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;

  // ---- LDS-DMA pieces of a unit: 48 of 1 KB (16 rows of 64 bytes); wave w issues pieces w, w + 4, ..: 15 halo pieces, 32 gradient pieces
  // (class c = 4 of them), one spare.  Lane -> (row = lane >> 2, 16-byte slot = lane & 3); the slot holds part (slot ^ swap) of the row
  const int prow = lane >> 2, slot = lane & 3;
  int rel[UW_PIECES], pk[UW_PIECES];       // voxel offset relative to the unit's origin (coarse for X, fine for G); steps for the bounds test
#pragma unroll
  for (int s = 0; s < UW_PIECES; ++s) {
    const int p = wave + 4 * s;
    int e0 = 0, e1 = 0, e2 = 0, r = 0, ok = 0;
    if (p < 15) {                          // halo row hv = (hh, hw, hd): coarse voxel (h0 - 1 + hh, w0 - 1 + hw, d0 - 1 + hd)
      const int hv = p * 16 + prow;
      e2 = hv % 10, e1 = (hv / 10) % 6, e0 = hv / 60;
      const int part = slot ^ ((e1 & 1) << 1);
      r = (((e0 - 1) * a.W + (e1 - 1)) * a.D + (e2 - 1)) * a.Ci + part * 8;
      ok = 1;
    } else if (p < 47) {                   // gradient row (class, qh, qw, qd): fine voxel 2 q + class parity
      const int cls = (p - 15) >> 2, row = ((p - 15) & 3) * 16 + prow;
      const int qh = row >> 5, qw = (row >> 3) & 3, qd = row & 7;
      const int part = slot ^ (((row >> 3) & 1) << 1);
      e0 = qh + 1, e1 = qw + 1, e2 = qd + 1;
      r = (((2 * qh + (cls >> 2)) * 2 * a.W + 2 * qw + ((cls >> 1) & 1)) * 2 * a.D + 2 * qd + (cls & 1)) * a.Co + part * 8;
      ok = n_blk + part * 8 < a.Co;
    }
    rel[s] = r;
    pk[s] = ok ? (e0 << 16 | e1 << 8 | e2) : -1;
  }
  const uint16_t* xsrc = reinterpret_cast<const uint16_t*>(a.x) + chunk * 32;
  const uint16_t* gsrc = reinterpret_cast<const uint16_t*>(a.grad) + n_blk;
  const char* zsrc = reinterpret_cast<const char*>(ltu_zero_uw) + slot * 16;
  // origin of a unit (wave-uniform, scalar registers) and the issue of ONE of this wave's 12 pieces: the pieces of unit i + 2 are spread
  // over the MFMA groups of unit i (their address arithmetic runs in the shadow of the matrix pipe instead of in front of it)
  struct UnitOrg { int hq, wq, dq; long long xvox0, gvox0; uint32_t bb; };
  auto unit_org = [&](int unit, int buf) {
    UnitOrg u;
    int t = unit >> 1;
    const int bd = t % nbd; t /= nbd;
    const int bw = t % nbw; t /= nbw;
    const int bh = t % nbh;
    const int b = t / nbh;
    u.hq = bh * 4 + (unit & 1) * 2, u.wq = bw * 4, u.dq = bd * 8;          // origin of the unit (coarse)
    u.xvox0 = ((((long long)b * a.H + u.hq) * a.W + u.wq) * a.D + u.dq) * a.Ci;
    u.gvox0 = ((((long long)b * 2 * a.H + 2 * u.hq) * 2 * a.W + 2 * u.wq) * 2 * a.D + 2 * u.dq) * a.Co;
    u.bb = lds0 + buf * (UW_BUF * 2);
    return u;
  };
  auto issue_piece = [&](const UnitOrg& u, auto S) {
    constexpr int s = decltype(S)::value;
    const int p = wave + 4 * s;
    const bool isx = p < 15;
    const int h = u.hq - 1 + (pk[s] >> 16), w = u.wq - 1 + ((pk[s] >> 8) & 255), d = u.dq - 1 + (pk[s] & 255);
    const bool in = pk[s] >= 0 && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W && (unsigned)d < (unsigned)a.D;
    const char* ptr = reinterpret_cast<const char*>((isx ? xsrc : gsrc) + (isx ? u.xvox0 : u.gvox0) + rel[s]);
    uw_glds16(in ? ptr : zsrc, u.bb + p * 1024);
  };
  auto issue_unit = [&](int unit, int buf) {
    const UnitOrg u = unit_org(unit, buf);
    uw_sfor<0, UW_PIECES>([&](auto S) { issue_piece(u, S); });
  };

  uw_f32x4 acc[64];
#pragma unroll
  for (int e = 0; e < 64; ++e) acc[e] = uw_f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;

  // transposing-read geometry for 16x16x32 (see the first generation): 16-lane group gq supplies k rows 8 gq .. 8 gq + 7 of a 32-row slab,
  // lane 4 q + p of the group row q (and q + 4), columns 4 p .. 4 p + 3 of the wave's 16-column half - the half sits in the OTHER 32 bytes
  // of the row where the row's 8-block (gradient) / w index (halo) is odd
  const int gq = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  const int growoff = UW_GOFF + (8 * gq + tq) * 32 + (wn ^ (gq & 1)) * 16 + 4 * tp;
  int xrowoff[2];                          // by the parity of ow + 1
#pragma unroll
  for (int par = 0; par < 2; ++par) xrowoff[par] = (gq * 10 + tq + 1) * 32 + (wc ^ ((gq + par) & 1)) * 16 + 4 * tp;
  const bool do_bias = chunk == 0 && wc == 0;

  const int nu = unit_hi - unit_lo;
  if (nu > 0) issue_unit(unit_lo, 0);
  if (nu > 1) issue_unit(unit_lo + 1, 1);
  int buf = 0;
  for (int i = 0; i < nu; ++i) {
    if (i + 1 < nu) uw_sync<UW_PIECES>(); else uw_sync<0>();
    const bool more = i + 2 < nu;
    const UnitOrg nxt = unit_org(more ? unit_lo + i + 2 : unit_lo, buf >= 1 ? buf - 1 : UW_NB - 1);
    const uint16_t* base = smem + buf * UW_BUF;
    buf = buf + 1 == UW_NB ? 0 : buf + 1;
    uw_sfor<0, 2>([&](auto KS) {
      constexpr int ks = decltype(KS)::value;
      uw_i32x4 ga[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        union { struct { uw16x4 l, h; } s; uw_i32x4 i; } u;
        const uint16_t* pg = base + (c * 64 + ks * 32) * 32 + growoff;
        u.s.l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_uw16x4*)pg);
        u.s.h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_uw16x4*)(pg + 4 * 32));
        ga[c] = u.i;
      }
      if (do_bias) {                       // column sums of the gradient: v_dot2c against ones (one instruction per two elements)
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(bsum) : "v"(ga[c][e]), "v"(0x3f803f80));
      }
      // the 27 halo offsets, their fragments requested two offsets ahead of the MFMAs that consume them (one wave per SIMD: nothing else
      // covers the LDS latency); sched_barriers keep the compiler from collapsing the pipeline back into read - wait - multiply
      uw_i32x4 ub[3];
      auto load_x = [&](auto O) {
        constexpr int o = decltype(O)::value, oh = o / 9 - 1, ow = (o / 3) % 3 - 1, od = o % 3 - 1;
        union { struct { uw16x4 l, h; } s; uw_i32x4 i; } u;
        // halo row ((ks + 1 + oh), (gq + 1 + ow), (tq + 1 + od)): w parity = (gq + ow + 1) & 1
        const uint16_t* px = base + (((ks + 1 + oh) * 6 + (1 + ow)) * 10 + od) * 32 + xrowoff[(ow + 1) & 1];
        u.s.l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_uw16x4*)px);
        u.s.h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_uw16x4*)(px + 4 * 32));
        ub[o % 3] = u.i;
      };
      load_x(std::integral_constant<int, 0>{});
      load_x(std::integral_constant<int, 1>{});
      uw_sfor<0, 27>([&](auto O) {
        constexpr int o = decltype(O)::value, oh = o / 9 - 1, ow = (o / 3) % 3 - 1, od = o % 3 - 1;
        if constexpr (o + 2 < 27) load_x(std::integral_constant<int, o + 2>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int ph = c >> 2, pw = (c >> 1) & 1, pd = c & 1;
          const int sh = oh + 1 - ph, sw = ow + 1 - pw, sd = od + 1 - pd;      // the slot of class c that reads this offset
          if (sh >= 0 && sh <= 1 && sw >= 0 && sw <= 1 && sd >= 0 && sd <= 1) uw_mfma(acc[c * 8 + (sh * 2 + sw) * 2 + sd], ga[c], ub[o % 3]);
        }
        // one piece of unit i + 2 behind every fourth MFMA group (6 per h-plane)
        if constexpr (o % 4 == 1 && o / 4 < 6) {
          if (more) issue_piece(nxt, std::integral_constant<int, ks * 6 + o / 4>{});
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  }
  // the accumulators were written by MFMAs the compiler sees as opaque instructions: no wait states are inserted in front of their first
  // reader (XDL write -> VALU read needs up to 18)
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

  // fold the 64 (class, slot) products onto the 27 taps and store this split's partial sums (layout of the first generation)
  float* pz = a.part + (long long)blockIdx.z * a.Co * a.kpad;
  const int ci = chunk * 32 + wc * 16 + (lane & 15);
#pragma unroll
  for (int th = 0; th < 3; ++th)
#pragma unroll
    for (int tw = 0; tw < 3; ++tw)
#pragma unroll
      for (int td = 0; td < 3; ++td) {
        uw_f32x4 s = uw_f32x4{0.f, 0.f, 0.f, 0.f};
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

// called by launch_upconv_wgrad_class_bf16 with the split geometry filled in (bricks, bricks_per_split, kpad, bpart): LTU_OK / hipError,
// or 1 = shape not handled here
int launch_upconv_wgrad_ring_bf16(const UpWgradArgs& a, int nchunk, int ntile, int nsplit, hipStream_t st) {
  if (a.Ci % 32 || a.Co % 8 || a.H < 2 || a.W < 2 || a.D < 2) return 1;
  if (((uintptr_t)a.x | (uintptr_t)a.grad) & 15) return 1;
  // relative offsets are 32-bit elements: a unit's halo / fine tile stays far below that; the 64-bit part is the unit's origin
  if ((long long)6 * a.W * a.D * a.Ci >= (1LL << 31) || (long long)16 * a.W * a.D * a.Co >= (1LL << 31)) return 1;
  constexpr int smem_bytes = UW_NB * UW_BUF * 2;
  static LtuDevOnce attr_once;
  if (attr_once.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&upconv_wgrad_ring_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
  hipLaunchKernelGGL(upconv_wgrad_ring_bf16_kernel, dim3(nchunk, ntile, nsplit), dim3(256), smem_bytes, st, a);
  return ltu_check_launch();
}
