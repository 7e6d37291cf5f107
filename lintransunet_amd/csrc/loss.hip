// Deep-supervision losses of one decoder level for gfx950, forward + analytic backward.
//
// Reference: loss/criterions.py 696-735 (class-mass-weighted CE on probabilities), 35-70 (per-class
// Dice), 416-442 (balanced Dice); loss/multi_criterions.py 594-615, 58-110 (one-hot variants: identical
// arithmetic once the one-hot target is "label == class"); label pyramid utils/utils_3D_embed_full.py:64,73-76.
//
// Every loss of a level is a function of four per-(sample, class) sums over the S voxels
//     P = sum p_c     T = sum t_c     I = sum p_c t_c     E = sum t_c (1-p_c) log(max(p_c, 1e-6))
// so the forward is one streaming reduction + a tiny finalize, and the backward is one streaming pass
//     dL/dp[s,c] = alpha[b,c] + t_c * (beta[b,c] + gamma[b,c] * f'(p)),   f(p) = (1-p) log(max(p,1e-6))
// with (alpha, beta, gamma) produced by the finalize kernel.
#include "common.h"

#define LOSS_MAXC 4

// Four voxels per thread and trip (S % 4 == 0): one 4-byte label load and C 16-byte probability loads, two trips in flight.  One
// voxel at a time (a 1-byte and C 4-byte loads per trip, 16 dependent trips per thread) the level-0 pass ran at 2.2 TB/s.
template <int C>
__global__ void __launch_bounds__(256) loss_sums_v4_kernel(const float* __restrict__ p, const uint8_t* __restrict__ label, float* __restrict__ sums,
                                                           long long S, int rows_per_block) {
  __shared__ float red[4][LOSS_MAXC * 4];
  const int b = blockIdx.y;
  float acc[C][4];
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[c][k] = 0.f;
  const long long s0 = (long long)blockIdx.x * rows_per_block;
  long long s1 = s0 + rows_per_block;
  if (s1 > S) s1 = S;
  auto fetch = [&](long long s, uint32_t& labs, float (&f)[4 * C]) {
    labs = *reinterpret_cast<const uint32_t*>(label + (long long)b * S + s);
    const float* pv = p + ((long long)b * S + s) * C;
#pragma unroll
    for (int q = 0; q < C; ++q) {
      const float4 t = *reinterpret_cast<const float4*>(pv + 4 * q);
      f[4 * q] = t.x; f[4 * q + 1] = t.y; f[4 * q + 2] = t.z; f[4 * q + 3] = t.w;
    }
  };
  auto add = [&](uint32_t labs, const float (&f)[4 * C]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int lab = (int)((labs >> (8 * j)) & 255u);
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float pc = f[j * C + c];
        const float t = lab == c ? 1.f : 0.f;
        acc[c][0] += pc;
        acc[c][1] += t;
        acc[c][2] += pc * t;
        acc[c][3] += t * (1.f - pc) * logf(fmaxf(pc, 1e-6f));
      }
    }
  };
  long long s = s0 + (long long)threadIdx.x * 4;
  for (; s + 1024 < s1; s += 2048) {
    uint32_t l0, l1;
    float f0[4 * C], f1[4 * C];
    fetch(s, l0, f0);
    fetch(s + 1024, l1, f1);
    add(l0, f0);
    add(l1, f1);
  }
  if (s < s1) {
    uint32_t l0;
    float f0[4 * C];
    fetch(s, l0, f0);
    add(l0, f0);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float v = wave_sum(acc[c][k]);
      if (lane == 0) red[wave][c * 4 + k] = v;
    }
  __syncthreads();
  if (threadIdx.x < C * 4) {
    float v = 0.f;
    for (int w = 0; w < 4; ++w) v += red[w][threadIdx.x];
    sums[((long long)(1 + blockIdx.x) * gridDim.y + b) * C * 4 + threadIdx.x] = v;
  }
}

// p f32 [B][S][C], label u8 [B][S]; sums [B][C][4] += {P,T,I,E}
__global__ void loss_sums_kernel(const float* __restrict__ p, const uint8_t* __restrict__ label, float* __restrict__ sums,
                                 long long S, int C, int rows_per_block) {
  __shared__ float red[4][LOSS_MAXC * 4];
  const int b = blockIdx.y;
  float acc[LOSS_MAXC][4];
#pragma unroll
  for (int c = 0; c < LOSS_MAXC; ++c)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[c][k] = 0.f;
  const long long s0 = (long long)blockIdx.x * rows_per_block;
  long long s1 = s0 + rows_per_block;
  if (s1 > S) s1 = S;
  for (long long s = s0 + threadIdx.x; s < s1; s += blockDim.x) {
    const int lab = label[(long long)b * S + s];
    const float* pv = p + ((long long)b * S + s) * C;
#pragma unroll
    for (int c = 0; c < LOSS_MAXC; ++c) {
      if (c < C) {
        const float pc = pv[c];
        const float t = lab == c ? 1.f : 0.f;
        acc[c][0] += pc;
        acc[c][1] += t;
        acc[c][2] += pc * t;
        acc[c][3] += t * (1.f - pc) * logf(fmaxf(pc, 1e-6f));
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < LOSS_MAXC; ++c)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float v = wave_sum(acc[c][k]);
      if (lane == 0) red[wave][c * 4 + k] = v;
    }
  __syncthreads();
  if (threadIdx.x < C * 4) {
    float v = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) v += red[w][threadIdx.x];
    // per-block partial [block][b][C*4] behind the final sums (no fp32 atomics: their order, and with it the last bit of every
    // loss coefficient, changed from run to run - bf16 rounding downstream turns that bit into 1e-2 of some gradients)
    sums[((long long)(1 + blockIdx.x) * gridDim.y + b) * C * 4 + threadIdx.x] = v;
  }
}

// weights: w_ce, w_bal, w_dice[c] (standard per-class Dice on class c).  values out: [0]=total, [1]=ce, [2]=bal,
// [3+c]=dice_c, [7]=foreground-union dice, [8]=total again.  coef [B][C][3] = alpha, beta, gamma (already multiplied by the loss weights).
struct LossCfg {
  float w_ce, w_bal, w_dice[LOSS_MAXC];
  float w_fg;          // Dice of the foreground union, p' = 1 - p_0 against t' = 1 - t_0 (loss/multi_criterions.py:30-56, DiceClassLoss0)
};

// scale_dev (nullable): a device-resident factor applied to all loss weights of this level (the per-epoch deep-supervision weight
// divided by the number of accumulated micro-steps): a captured HIP graph then follows weight changes without being re-captured.
__global__ void loss_finalize_kernel(float* __restrict__ sums, int nblk, float* __restrict__ values, float* __restrict__ coef, int B,
                                     long long S, int C, LossCfg cfg, const float* __restrict__ scale_dev) {
  {
    // fold the per-block partials in a fixed order: output o = tid % nout is shared by the 256 / nout thread groups (each sums
    // every ngrp-th block, 8 loads in flight), which meet in LDS
    __shared__ float fold[256];
    const int nout = B * C * 4;                          // <= 256
    const int ngrp = 256 / nout;
    const int o = threadIdx.x % nout, grp = threadIdx.x / nout;
    float a0 = 0.f, a1 = 0.f;
    if (grp < ngrp) {
      const float* pp = sums + nout + o;
      int z = grp;
      for (; z + 7 * ngrp < nblk; z += 8 * ngrp) {
        const float v0 = pp[(long long)z * nout], v1 = pp[(long long)(z + ngrp) * nout], v2 = pp[(long long)(z + 2 * ngrp) * nout],
                    v3 = pp[(long long)(z + 3 * ngrp) * nout], v4 = pp[(long long)(z + 4 * ngrp) * nout],
                    v5 = pp[(long long)(z + 5 * ngrp) * nout], v6 = pp[(long long)(z + 6 * ngrp) * nout],
                    v7 = pp[(long long)(z + 7 * ngrp) * nout];
        a0 += (v0 + v1) + (v2 + v3); a1 += (v4 + v5) + (v6 + v7);
      }
      for (; z < nblk; z += ngrp) a0 += pp[(long long)z * nout];
    }
    fold[threadIdx.x] = a0 + a1;
    __syncthreads();
    if ((int)threadIdx.x < nout) {
      float t = 0.f;
      for (int g = 0; g < ngrp; ++g) t += fold[g * nout + threadIdx.x];
      sums[threadIdx.x] = t;
    }
    __syncthreads();
  }
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (scale_dev != nullptr) {
    const float sc = scale_dev[0];
    cfg.w_ce *= sc; cfg.w_bal *= sc; cfg.w_fg *= sc;
    for (int c = 0; c < LOSS_MAXC; ++c) cfg.w_dice[c] *= sc;
  }
  float fg = 0.f;
  float ce = 0.f, bal = 0.f, dice[LOSS_MAXC] = {0.f, 0.f, 0.f, 0.f};
  const float Z = (float)B * (float)S * (float)C;
  for (int b = 0; b < B; ++b) {
    const float* sb = sums + (long long)b * C * 4;
    float Ttot = 0.f;
    for (int c = 0; c < C; ++c) Ttot += sb[c * 4 + 1];
    // balanced dice pieces
    float num = 0.f, den = 0.f, wc[LOSS_MAXC];
    for (int c = 0; c < C; ++c) {
      const float t = sb[c * 4 + 1] + 1e-5f;
      wc[c] = 1.f / (t * t);
      num += sb[c * 4 + 2] * wc[c];
      den += (sb[c * 4 + 0] + sb[c * 4 + 1]) * wc[c];
    }
    const float Nb = 2.f * num + 1e-5f, Db = den + 1e-5f;
    bal += Nb / Db;
    for (int c = 0; c < C; ++c) {
      const float P = sb[c * 4 + 0], T = sb[c * 4 + 1], I = sb[c * 4 + 2], E = sb[c * 4 + 3];
      const float w = (Ttot - (P + 1e-5f)) / Ttot;
      ce += -w * E;
      const float N = 2.f * I + 1e-9f, Dd = P + T + 1e-9f;
      dice[c] += N / Dd;
      float alpha = 0.f, beta = 0.f, gamma = 0.f;
      // CE: L = -(1/Z) sum w E  ->  dL/dp = (1/Z) (E/Ttot) - (1/Z) w t f'(p)
      alpha += cfg.w_ce * E / (Z * Ttot);
      gamma += -cfg.w_ce * w / Z;
      // Dice_c: L = 1 - (1/B) N/D -> dL/dp_c = (1/B) N/D^2 - (1/B) 2 t / D
      alpha += cfg.w_dice[c] * N / ((float)B * Dd * Dd);
      beta += -cfg.w_dice[c] * 2.f / ((float)B * Dd);
      // balanced Dice
      alpha += cfg.w_bal * Nb * wc[c] / ((float)B * Db * Db);
      beta += -cfg.w_bal * 2.f * wc[c] / ((float)B * Db);
      if (c == 0) {
        // foreground union: P' = S - P, T' = S - T, I' = S - P - T + I;  L = 1 - (1/B) N'/D', N' = 2 I' + eps, D' = P' + T' + eps
        //   dL/dp_0 = (1/B) (2 (1 - t_0) / D' - N' / D'^2)
        const float Sf = (float)S;
        const float Nf = 2.f * (Sf - P - T + I) + 1e-9f, Df = (Sf - P) + (Sf - T) + 1e-9f;
        fg += Nf / Df;
        alpha += cfg.w_fg * (2.f / Df - Nf / (Df * Df)) / (float)B;
        beta += -cfg.w_fg * 2.f / ((float)B * Df);
      }
      float* o = coef + ((long long)b * C + c) * 3;
      o[0] = alpha; o[1] = beta; o[2] = gamma;
    }
  }
  ce /= Z;
  bal = 1.f - bal / (float)B;
  float total = cfg.w_ce * ce + cfg.w_bal * bal;
  values[1] = ce;
  values[2] = bal;
  for (int c = 0; c < C; ++c) {
    const float dv = 1.f - dice[c] / (float)B;
    values[3 + c] = dv;
    total += cfg.w_dice[c] * dv;
  }
  const float fgv = 1.f - fg / (float)B;
  values[7] = fgv;
  total += cfg.w_fg * fgv;
  values[0] = total;
  values[8] = total;      // a second copy: the autograd wrapper exposes [8] as the differentiable scalar and [0..7] as the report
}

// four voxels per thread (S % 4 == 0): 16-byte loads and stores, the probabilities read unconditionally
template <int C>
__global__ void __launch_bounds__(256) loss_bwd_v4_kernel(const float* __restrict__ p, const uint8_t* __restrict__ label, const float* __restrict__ coef,
                                                          const float* __restrict__ gscale, float* __restrict__ dp, long long S) {
  const int b = blockIdx.y;
  const float gs = gscale[0];
  float k0[C], k1[C], k2[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float* k = coef + ((long long)b * C + c) * 3;
    k0[c] = k[0]; k1[c] = k[1]; k2[c] = k[2];
  }
  for (long long s = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; s < S; s += (long long)gridDim.x * 1024) {
    const long long i = (long long)b * S + s;
    const uint32_t labs = *reinterpret_cast<const uint32_t*>(label + i);
    float f[4 * C], o[4 * C];
#pragma unroll
    for (int q = 0; q < C; ++q) {
      const float4 t = *reinterpret_cast<const float4*>(p + i * C + 4 * q);
      f[4 * q] = t.x; f[4 * q + 1] = t.y; f[4 * q + 2] = t.z; f[4 * q + 3] = t.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int lab = (int)((labs >> (8 * j)) & 255u);
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float pc = f[j * C + c];
        const float fp = -logf(fmaxf(pc, 1e-6f)) + (pc > 1e-6f ? (1.f - pc) / pc : 0.f);
        o[j * C + c] = gs * (lab == c ? k0[c] + (k1[c] + k2[c] * fp) : k0[c]);
      }
    }
#pragma unroll
    for (int q = 0; q < C; ++q) *reinterpret_cast<float4*>(dp + i * C + 4 * q) = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
  }
}

// dp[s,c] = gscale * (alpha + t (beta + gamma f'(p)))
__global__ void loss_bwd_kernel(const float* __restrict__ p, const uint8_t* __restrict__ label, const float* __restrict__ coef,
                                const float* __restrict__ gscale, float* __restrict__ dp, int B, long long S, int C) {
  const long long n = (long long)B * S;
  const float gs = gscale[0];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / S);
    const int lab = label[i];
    for (int c = 0; c < C; ++c) {
      const float* k = coef + ((long long)b * C + c) * 3;
      float g = k[0];
      if (lab == c) {
        const float pc = p[i * C + c];
        const float fp = -logf(fmaxf(pc, 1e-6f)) + (pc > 1e-6f ? (1.f - pc) / pc : 0.f);
        g += k[1] + k[2] * fp;
      }
      dp[i * C + c] = gs * g;
    }
  }
}

static long long loss_rows(int B, long long S) {
  long long want = 1024 / (B > 0 ? B : 1);
  if (want < 1) want = 1;
  long long rows = (S + want - 1) / want;
  if (rows < 256) rows = 256;
  return (rows + 3) / 4 * 4;
}
extern "C" long long ltu_loss_ws_floats(int B, long long S, int C) { return (1 + cdiv(S, loss_rows(B, S))) * (long long)B * C * 4; }

extern "C" int ltu_loss_fwd(const float* p, const uint8_t* label, float* sums, long long sums_floats, float* values, float* coef, int B, long long S,
                            int C,
                            float w_ce, float w_bal, const float* w_dice, const float* scale_dev, ltu_stream_t s) {
  if (C < 1 || C > LOSS_MAXC || B * C * 4 > 256) return LTU_E_SHAPE;
  const long long rows = loss_rows(B, S);
  const int nblk = (int)cdiv(S, rows);
  if ((1 + (long long)nblk) * B * C * 4 > sums_floats) return LTU_E_ARG;          // the scratch is shorter than this geometry needs
  LossCfg cfg;
  cfg.w_ce = w_ce; cfg.w_bal = w_bal;
  for (int c = 0; c < LOSS_MAXC; ++c) cfg.w_dice[c] = (c < C && w_dice) ? w_dice[c] : 0.f;
  cfg.w_fg = w_dice ? w_dice[LOSS_MAXC] : 0.f;
  const bool v4 = S % 4 == 0 && !ltu_knob("LTU_LOSS_SCALAR", 0);
  if (v4 && C == 2) hipLaunchKernelGGL(loss_sums_v4_kernel<2>, dim3(nblk, B), dim3(256), 0, (hipStream_t)s, p, label, sums, S, (int)rows);
  else if (v4 && C == 3) hipLaunchKernelGGL(loss_sums_v4_kernel<3>, dim3(nblk, B), dim3(256), 0, (hipStream_t)s, p, label, sums, S, (int)rows);
  else if (v4 && C == 4) hipLaunchKernelGGL(loss_sums_v4_kernel<4>, dim3(nblk, B), dim3(256), 0, (hipStream_t)s, p, label, sums, S, (int)rows);
  else hipLaunchKernelGGL(loss_sums_kernel, dim3(nblk, B), dim3(256), 0, (hipStream_t)s, p, label, sums, S, C, (int)rows);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, sums, nblk, values, coef, B, S, C, cfg, scale_dev);
  return ltu_check_launch();
}

extern "C" int ltu_loss_bwd(const float* p, const uint8_t* label, const float* coef, const float* gscale, float* dp, int B,
                            long long S, int C, ltu_stream_t s) {
  const long long n = (long long)B * S;
  long long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  const bool v4 = S % 4 == 0 && C >= 2 && !ltu_knob("LTU_LOSS_SCALAR", 0);
  if (v4) {
    long long bx = (S / 4 + 255) / 256;
    const long long cap = 4096 / (B > 0 ? B : 1) > 1 ? 4096 / (B > 0 ? B : 1) : 1;
    if (bx > cap) bx = cap;
    const dim3 grid((unsigned)bx, B);
    if (C == 2) hipLaunchKernelGGL(loss_bwd_v4_kernel<2>, grid, dim3(256), 0, (hipStream_t)s, p, label, coef, gscale, dp, S);
    else if (C == 3) hipLaunchKernelGGL(loss_bwd_v4_kernel<3>, grid, dim3(256), 0, (hipStream_t)s, p, label, coef, gscale, dp, S);
    else hipLaunchKernelGGL(loss_bwd_v4_kernel<4>, grid, dim3(256), 0, (hipStream_t)s, p, label, coef, gscale, dp, S);
    return ltu_check_launch();
  }
  hipLaunchKernelGGL(loss_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, p, label, coef, gscale, dp, B, S, C);
  return ltu_check_launch();
}

// label pyramid: u8 [B][H][W][D] -> max over (2,2,kd) windows, kd in {1,2}
__global__ void label_maxpool_kernel(const uint8_t* __restrict__ x, uint8_t* __restrict__ y, int B, int H, int W, int D, int kd) {
  const int Ho = H / 2, Wo = W / 2, Do = D / kd;
  const long long n = (long long)B * Ho * Wo * Do;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int d = (int)(i % Do);
    long long t = i / Do;
    const int w = (int)(t % Wo); t /= Wo;
    const int h = (int)(t % Ho);
    const int b = (int)(t / Ho);
    uint8_t m = 0;
    for (int a = 0; a < 2; ++a)
      for (int e = 0; e < 2; ++e)
        for (int f = 0; f < kd; ++f) {
          const uint8_t v = x[(((long long)b * H + 2 * h + a) * W + 2 * w + e) * D + d * kd + f];
          m = v > m ? v : m;
        }
    y[i] = m;
  }
}
extern "C" int ltu_label_maxpool(const uint8_t* x, uint8_t* y, int B, int H, int W, int D, int kd, ltu_stream_t s) {
  if ((kd != 1 && kd != 2) || H % 2 || W % 2 || D % kd) return LTU_E_SHAPE;
  const long long n = (long long)B * (H / 2) * (W / 2) * (D / kd);
  long long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(label_maxpool_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, x, y, B, H, W, D, kd);
  return ltu_check_launch();
}
