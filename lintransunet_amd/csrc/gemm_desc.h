// Descriptors shared by the fp32-MFMA (gemm.hip) and bf16-MFMA (gemm_bf16.hip) implicit-GEMM kernels.
#pragma once
#include <stdlib.h>
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
  Tap tap[64];
  int out_identity;
  int omh, omw, omd, ooh, oow, ood, oh, ow, od;
  int n0;            // columns [0,n0) go to o0 (row stride ldo0), the rest to o1 (ldo1)
  int ldo0, ldo1;
  int accum;         // != 0: add to the existing output instead of overwriting
  int dbg;           // experiments only: 1 = skip output stores, 2 = skip matrix work, 4 = skip A loads
  // split over K for small grids with long K loops (bf16 NT kernel): workgroup z handles K tiles [z*kt_per_split, ...) and
  // stores its fp32 tile to part[z][M][N]; igemm_fold_kernel adds the splits and the bias.  part == nullptr: no split.
  float* part;
  long long part_floats;   // capacity of `part` (floats): a geometry that needs more is refused (LTU_E_ARG), never written past
  int ksplit, kt_per_split;
  // fused GELU + dropout epilogue of the weight-stationary projection kernel (transformer FFN: trans_block.py:208): the
  // pre-activation goes to o0 as usual and h = dropout(gelu(bf16(u))) to gelu_out [M][N]; nullptr: plain projection
  void* gelu_out;
  float drop_p;
  unsigned long long drop_seed;
  const unsigned long long* drop_step;
};
long long igemm_nt_ws_floats(long long M, int N, int K);

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
  // two-stage mode (part != null): every split stores its fp32 tile to part[split][npad][kpad] (+ bias partials to
  // bpart[split][npad]) with plain stores; wgrad_reduce_kernel sums the splits into the gradient.  No atomics.
  float* part;
  long long part_floats;   // capacity of `part` (floats)
  float* bpart;
  int npad, kpad;
  // second PyTorch-layout conv gradient (conv pairs): rows [n0_2, n0_2 + t_co2) of the tile go to dw2 [t_co2][t_ci][27] / db2
  float* dw2;
  float* db2;
  int n0_2, t_co2;
  float* dwseg[3];    // gradient blocks of N/nseg_w rows each (fused q,k,v projections)
  float* dbseg[3];
  int nseg_w;
};

// split geometry shared by the launcher and the workspace-size query
struct TnGeom {
  int bn, bk, nn, nk, rows, nsplit;
  long long ws_floats;
};
static inline TnGeom tn_geometry(long long M, int N, int K, int brows) {
  TnGeom t;
  t.bk = 128;
  t.bn = N > 64 ? 128 : (N > 32 ? 64 : 32);
  t.nk = (K + t.bk - 1) / t.bk;
  t.nn = (N + t.bn - 1) / t.bn;
  int tn_want = -1, tn_min = -1;
  tn_want = ltu_knob_pos("LTU_TN_WANT", 1024);     // swept: 1024 / 128 (tools/sweep_tn.sh)
  tn_min = ltu_knob_pos("LTU_TN_MINROWS", 128);
  long long want = tn_want / ((long long)t.nk * t.nn);
  if (want < 1) want = 1;
  long long rows = (M + want - 1) / want;
  if (rows < tn_min) rows = tn_min;
  rows = (rows + brows - 1) / brows * brows;
  t.rows = (int)rows;
  t.nsplit = (int)((M + rows - 1) / rows);
  t.ws_floats = (long long)t.nsplit * t.nn * t.bn * ((long long)t.nk * t.bk + 1);
  return t;
}


// bf16-MFMA launchers (gemm_bf16.hip)
int launch_nt_bf16(const IGemmArgs& g, hipStream_t st);
int launch_tn_bf16(WGradArgs& wa, hipStream_t st);

// LDS-halo stride-1 3x3x3 convolution (conv_halo.hip)
struct HaloArgs {
  const void* x0;
  const void* x1;
  const void* w;        // [N][27][C] in bf16
  const float* bias;
  void* o0;
  void* o1;
  int B, H, W, D;
  int C, c0, lda0, lda1;
  int N, n0, ldo0, ldo1;
  int flip;             // 1: data gradient (tap t reads halo offset 2 - t)
  int CC;               // channel chunk: 16 or 32 (chosen by the launcher)
  // split over channel chunks for small grids (the deep U-Net levels: a few dozen bricks, thousands of input channels x taps):
  // workgroup z accumulates chunks [z*cps, (z+1)*cps) and stores its fp32 tile to part[z][voxel][N]; conv_halo_fold_kernel
  // adds the splits and the bias.  part == nullptr: one workgroup per tile walks all chunks.
  float* part;
  long long part_floats;   // capacity of `part` (floats)
  int ksplit, cps;
  int no_xcd_order;     // 1: bricks dealt to the workgroups of the persistent kernels strided by the grid (LTU_HALO_NO_XCD: the old order)
};
long long conv_halo_ws_floats(int B, int H, int W, int D, int C, int N);
int launch_conv_halo_bf16(HaloArgs a, hipStream_t st);   // LTU_OK / hipError, or 1 = shape not handled (fall back)
bool conv_ring_splits(int B, int H, int W, int D, int C, int N);   // the ring kernel would split the channel chunks: `part` needed
int launch_conv_ring_bf16(HaloArgs& a, hipStream_t st);            // C >= 64 or N > 32 (conv_ring.hip): LTU_OK / hipError, or 1; small grids: sets ksplit / cps (> 1: the caller folds `part`)
int launch_conv_fc_ring_bf16(const HaloArgs& a, hipStream_t st);   // C = 32, N <= 32 (conv_fc_ring.hip): LTU_OK / hipError, or 1 = not handled
int launch_conv_c16_ring_bf16(const HaloArgs& a, hipStream_t st);  // C = 16, N <= 32 (conv_c16_ring.hip): LTU_OK / hipError, or 1 = not handled

// LDS-halo weight gradient of the stride-1 3x3x3 convs (conv_halo.hip), two-stage through part / wgrad_reduce_kernel
struct WHaloArgs {
  const void* x0;
  const void* x1;
  const void* grad;     // [voxels][ldg]
  const void* grad1;    // optional second gradient source: columns [gn0, N) come from grad1 [voxels][ldg1] (conv pairs)
  int gn0, ldg1;
  int B, H, W, D;
  int C, c0, lda0, lda1;
  int N, ldg;
  int CC;               // channel chunk per workgroup: 16 or 32
  int bricks, bricks_per_split;
  float* part;          // [nsplit][npad][kpad]
  long long part_floats;   // capacity of `part` (floats)
  float* bpart;         // [nsplit][npad]
  int npad, kpad;
};
long long conv_wgrad_halo_ws_floats(int N, int K);
int launch_conv_wgrad_halo_bf16(WHaloArgs a, int* nsplit_out, hipStream_t st);   // LTU_OK / hipError, or 1 = not handled
int launch_conv_wgrad_halo_ring_bf16(WHaloArgs a, int* nsplit_out, hipStream_t st);   // C % 32 == 0 (wgrad_halo_ring.hip): LTU_OK / hipError / LTU_E_ARG, or 1
int launch_wgrad_reduce(const WGradArgs& wa, int nsplit, hipStream_t st);

// LDS-DMA ring kernels for the dense projections (gemm_ring.hip)
bool tn_ring_shape_ok(long long M, int N, int K);
long long tn_ring_ws_floats(long long M, int N, int K);
int launch_tn_ring_bf16(WGradArgs& wa, hipStream_t st, int* nsplit_out = nullptr);   // LTU_OK / hipError, or 1 = shape not handled
int launch_nt_ring_bf16(const IGemmArgs& g, hipStream_t st);
// several dense weight gradients in one launch + one fold launch; 0 floats / 1 = group not handled
long long tn_ring_group_ws_floats(const ltu_wgrad_job* jobs, int njobs, int blocks);
int launch_tn_ring_group_bf16(const ltu_wgrad_job* jobs, int njobs, int blocks, float* ws, long long ws_floats, hipStream_t st);      // LTU_OK / hipError, or 1 = shape not handled (also: gelu_out set and no ring)

// class convolutions on an LDS halo brick (conv_halo.hip): fine voxel o = m q + p gets sum_e [cls_e == class(p)] x[q + d_e] . W_e^T
struct ClsEntry {
  int8_t dh, dw, dd, cls;
  int wbase;            // element offset of the entry's weight tile: W_e[n][k] = w[wbase + n * wrow + k]
};
struct ClassHaloArgs {
  const void* x;        // coarse source [B][H][W][D][lda]
  const void* w;
  const float* bias;
  void* o0;
  void* o1;             // fine outputs: columns [0,n0) -> o0, the rest -> o1
  int B, H, W, D;
  int C, lda;
  int N, n0, ldo0, ldo1;
  int Hh, Wh, Dh;       // fine grid
  int mh, mw, md;       // 1 or 2 per axis
  int wrow;
  int nent, ncls;
  int cls_begin[9];     // filled by the launcher of the ring kernel: entries of class c are [cls_begin[c], cls_begin[c+1])
  int8_t cls_p[8][4];   // parity (ph, pw, pd) of each class
  ClsEntry ent[64];
};
int launch_conv_class_halo_bf16(const ClassHaloArgs& a, hipStream_t st);
// y [M][N] (+)= x [M][K] W^T + b for K, N <= 128 / 64 (pw_small.hip): LTU_OK / hipError, or 1 = shape not handled
int launch_pw_small_bf16(const void* x, int lda, const void* w, const float* bias, void* y, int ldy, long long M, int N, int K, int accumulate,
                         hipStream_t st);
// data gradient of a stride-(2,2,sd) conv (sdgrad_ring.hip): LTU_OK / hipError, or 1 = shape not handled
int launch_sdgrad_ring_bf16(const void* grad, const void* wd, void* dx, int B, int Hl, int Wl, int Dl, int N, int Co, int sd, hipStream_t st);
// data gradient of the sub-pixel un-embedding (updgrad_ring.hip): LTU_OK / hipError, or 1 = shape not handled
int launch_updgrad_ring_bf16(const void* grad, const void* wsub_d, void* dx, int B, int H, int W, int D, int Ci, int Co, hipStream_t st);
// sub-pixel un-embedding forward, second generation (upconv_ring.hip): LTU_OK / hipError, or 1 = shape not handled
int launch_upconv_ring_bf16(const void* x, const void* wsub_f, const float* bias, void* y, int B, int H, int W, int D, int Ci, int Co,
                            hipStream_t st);   // LTU_OK / hipError, or 1 = shape not handled

// weight gradient of the sub-pixel un-embedding from LDS halo bricks (conv_halo.hip), two-stage through wgrad_reduce_kernel
struct UpWgradArgs {
  const void* x;        // coarse [B][H][W][D][Ci]
  const void* grad;     // fine   [B][2H][2W][2D][Co]
  int B, H, W, D, Ci, Co;
  int bricks, bricks_per_split;
  float* part;          // [nsplit][Co][27 Ci]
  long long part_floats;   // capacity of `part` (floats)
  float* bpart;         // [nsplit][Co]
  int kpad;
  int blocks;           // workgroup budget of the launch (<= 0: LTU_UPW_BLOCKS / 256)
};
long long upconv_wgrad_class_ws_floats(int Ci, int Co, int blocks);
int launch_upconv_wgrad_class_bf16(UpWgradArgs a, int* nsplit_out, hipStream_t st);   // LTU_OK / hipError, or 1 = not handled
int launch_upconv_wgrad_ring_bf16(const UpWgradArgs& a, int nchunk, int ntile, int nsplit, hipStream_t st);   // same work, LDS-DMA ring (upconv_wgrad_ring.hip)
