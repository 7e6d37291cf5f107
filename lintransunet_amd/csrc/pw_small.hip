// Small pointwise projections  y[M][N] (+)= x[M][K] W^T + b  with K, N <= 128 (bf16): the 1x1x1 convolutions of the attention gates
// (model/Unet_3Dblock.py:194-221: W_x C -> C, W_g 2C -> C at C = 16 ... 128) and their data gradients.
//
// These are streaming ops (K = N = 16 at 64x64x128: 67 MB for 0.5 GFLOP) that ran through the tap-table implicit GEMM at 30-40 us for
// 8 us of HBM time (VALU half busy on gather arithmetic that a dense [M][K] operand does not need).  Here a wave owns 32-row tiles:
// the MFMA B fragment of a tile (lane (li, lh): 8 consecutive k of row li) IS a 16-byte piece of the row, so it is loaded straight
// from global memory in operand layout - no LDS, no barrier; the weights (<= 2 column tiles x 8 k-steps) stay in registers; the
// transposed product D[n][row] leaves a lane quads of 4 consecutive output channels of one row; one v_permlane32_swap per quad pair
// turns them into 16-byte stores (the 32 rows of a tile are one contiguous span of the output).  The next tile's fragments are
// requested before the current tile's MFMAs.  Measured (tools/bench_gate_proj.py): 1 M rows 16 -> 16: 35.8 -> 13.1 us, 32 -> 16:
// 40.4 -> 20.8, 16 -> 32: 43.1 -> 25.6; 262 144 rows 32 -> 32: 12.3 -> 7.5, 64 -> 32: 15.0 -> 10.1, 32 -> 64: 16.5 -> 11.3.
#include "gemm_desc.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

struct PwArgs {
  const uint16_t* x;
  const uint16_t* w;      // [N][K] bf16
  const float* bias;      // nullable
  uint16_t* y;
  long long M;
  int N, accumulate;
};

template <int KS, int NT>      // K = 16 KS, N <= 32 NT
__global__ void __launch_bounds__(256) pw_small_bf16_kernel(const PwArgs a) {
  constexpr int K = 16 * KS;
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwave = (long long)gridDim.x * 4;
  const long long ntile = (a.M + 31) / 32;
  bf16x8 wf[NT][KS];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int n = j * 32 + li;
      const uint4 v = *reinterpret_cast<const uint4*>(a.w + (long long)(n < a.N ? n : 0) * K + ks * 16 + lh * 8);
      wf[j][ks] = __builtin_bit_cast(bf16x8, n < a.N ? v : make_uint4(0u, 0u, 0u, 0u));
    }
  float4 bv[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int n = j * 32 + 8 * rr + 4 * lh;
      bv[j][rr] = (a.bias != nullptr && n < a.N) ? *reinterpret_cast<const float4*>(a.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  auto fetch = [&](long long tile, uint4 (&xv)[KS]) {
    long long row = tile * 32 + li;
    if (row >= a.M) row = a.M - 1;                       // clamped: the rows past the end are not stored
    const uint16_t* p = a.x + row * K + lh * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xv[ks] = *reinterpret_cast<const uint4*>(p + ks * 16);
  };
  uint4 xc[KS], xn[KS];
  long long tile = wave;
  if (tile >= ntile) return;
  fetch(tile, xc);
  for (; tile < ntile; tile += nwave) {
    const bool more = tile + nwave < ntile;
    if (more) fetch(tile + nwave, xn);
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j][ks], __builtin_bit_cast(bf16x8, xc[ks]), acc[j], 0, 0, 0);
    }
    // Epilogue: a lane holds quads n = 8 rr + 4 lh .. + 3 of row li.  The two halves of the wave exchange one quad per pair of rr
    // (v_permlane32_swap: upper lanes of the first operand <-> lower lanes of the second), after which lane (li, 1) owns channels
    // 16 q .. + 7 and lane (li, 0) channels 16 q + 8 .. + 15 of its row: 16-byte stores instead of 8-byte ones
    const long long row = tile * 32 + li;
    uint16_t* yr = a.y + (row < a.M ? row : 0) * a.N;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int r0 = 2 * q, r1 = 2 * q + 1;
        float p0x = __uint_as_float(pack_bf16x2(acc[j][4 * r0] + bv[j][r0].x, acc[j][4 * r0 + 1] + bv[j][r0].y));
        float p0y = __uint_as_float(pack_bf16x2(acc[j][4 * r0 + 2] + bv[j][r0].z, acc[j][4 * r0 + 3] + bv[j][r0].w));
        float p1x = __uint_as_float(pack_bf16x2(acc[j][4 * r1] + bv[j][r1].x, acc[j][4 * r1 + 1] + bv[j][r1].y));
        float p1y = __uint_as_float(pack_bf16x2(acc[j][4 * r1 + 2] + bv[j][r1].z, acc[j][4 * r1 + 3] + bv[j][r1].w));
        permlane32_swap(p1x, p0x);           // p1* : lower lanes keep n 8-11, upper lanes receive n 0-3;  p0*: lower lanes receive n 12-15, upper keep n 4-7
        permlane32_swap(p1y, p0y);
        const int n = j * 32 + 16 * q + (lh ? 0 : 8);
        if (row < a.M && n < a.N) {          // N is a multiple of 8 here (checked by the launcher)
          uint4 v = make_uint4(__float_as_uint(p1x), __float_as_uint(p1y), __float_as_uint(p0x), __float_as_uint(p0y));
          if (a.accumulate) {
            const uint4 o = *reinterpret_cast<const uint4*>(yr + n);
            const uint32_t nv[4] = {v.x, v.y, v.z, v.w}, ov[4] = {o.x, o.y, o.z, o.w};
            uint32_t rv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
              rv[e] = pack_bf16x2(__uint_as_float(nv[e] << 16) + __uint_as_float(ov[e] << 16),
                                  __uint_as_float(nv[e] & 0xffff0000u) + __uint_as_float(ov[e] & 0xffff0000u));
            v = make_uint4(rv[0], rv[1], rv[2], rv[3]);
          }
          *reinterpret_cast<uint4*>(yr + n) = v;
        }
      }
    if (more) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xc[ks] = xn[ks];
    }
  }
}

// LTU_OK after launching, or 1 when the shape is not handled (the caller keeps its implicit GEMM)
int launch_pw_small_bf16(const void* x, int lda, const void* w, const float* bias, void* y, int ldy, long long M, int N, int K, int accumulate,
                         hipStream_t st) {
  if (lda != K || ldy != N || N % 8 || N > 64 || M < 65536) return 1;      // below ~64 K rows the launch is latency-bound either way
  if (K != 16 && K != 32 && K != 64 && K != 128) return 1;
  PwArgs a;
  a.x = (const uint16_t*)x; a.w = (const uint16_t*)w; a.bias = bias; a.y = (uint16_t*)y; a.M = M; a.N = N; a.accumulate = accumulate;
  const long long ntile = (M + 31) / 32;
  long long blocks = (ntile + 15) / 16;            // ~4 tiles per wave
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  const dim3 grid((unsigned)blocks), blk(256);
  const int nt = N > 32 ? 2 : 1;
#define PW_LAUNCH(KS_, NT_) hipLaunchKernelGGL((pw_small_bf16_kernel<KS_, NT_>), grid, blk, 0, st, a)
  if (nt == 1) {
    if (K == 16) PW_LAUNCH(1, 1); else if (K == 32) PW_LAUNCH(2, 1); else if (K == 64) PW_LAUNCH(4, 1); else PW_LAUNCH(8, 1);
  } else {
    if (K == 16) PW_LAUNCH(1, 2); else if (K == 32) PW_LAUNCH(2, 2); else if (K == 64) PW_LAUNCH(4, 2); else PW_LAUNCH(8, 2);
  }
#undef PW_LAUNCH
  return ltu_check_launch();
}
