// Descriptors shared by the fp32-MFMA (gemm.hip) and bf16-MFMA (gemm_bf16.hip) implicit-GEMM kernels.
#pragma once
#include "common.h"

struct Tap {
  int8_t dh, dw, dd, wt;
};

struct IGemmArgs {
  const void* a0;
  const void* a1;
  const void* w[3];    // weight blocks in the activation dtype (fp32 or bf16), [N/nseg][wrow] each
  const float* bias[3];
  void* o0;
  void* o1;
  long long M;
  int N, K;          // K = ntaps * C
  int C, c0;         // channels per tap; channels served by a0 (rest by a1)
  int lda0, lda1;    // voxel strides of a0 / a1 (elements)
  int nseg;          // weight row segments (N/nseg rows each)
  int wrow;          // weight row length = wtaps * C
  int nb, rh, rw, rd;
  int sh, sw, sd, ups;
  int mh, mw, md;
  int ntaps;
  Tap tap[27];
  int out_identity;
  int omh, omw, omd, ooh, oow, ood, oh, ow, od;
  int n0;            // columns [0,n0) go to o0 (row stride ldo0), the rest to o1 (ldo1)
  int ldo0, ldo1;
  int accum;         // != 0: add to the existing output instead of overwriting
  int dbg;           // experiments only: 1 = skip output stores, 2 = skip matrix work, 4 = skip A loads
};

struct RowCoord {
  int b, h, w, d;
};

__device__ __forceinline__ RowCoord split_row(const IGemmArgs& g, long long m) {
  RowCoord r;
  r.d = (int)(m % g.rd);
  long long t = m / g.rd;
  r.w = (int)(t % g.rw);
  t /= g.rw;
  r.h = (int)(t % g.rh);
  r.b = (int)(t / g.rh);
  return r;
}

struct WGradArgs {
  IGemmArgs g;        // gather description (a0/a1, taps, row grid); N,K as above
  const void* grad;   // G [M][ldg]
  int ldg;
  float* dw;          // [N][wrow]
  float* db;          // may be null
  int rows_per_split; // multiple of the tile's row count
  int t_co, t_ci;     // != 0: dw is a PyTorch conv weight gradient [t_co][t_ci][27] (padded rows/channels are dropped)
};


// bf16-MFMA launchers (gemm_bf16.hip)
int launch_nt_bf16(const IGemmArgs& g, hipStream_t st);
int launch_tn_bf16(WGradArgs& wa, hipStream_t st);
