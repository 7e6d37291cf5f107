// nearest-x2 upsampling followed by a 3x3x3 conv (the ROI token un-embedding, model/Unet_3Dblock.py:419-432)
// as a SUB-PIXEL convolution: per output parity class p = o & 1 (per axis) the 3 taps collapse onto 2 source voxels,
//     p = 0:  { t=0 } -> x[q-1],  { t=1, t=2 } -> x[q]            p = 1:  { t=0, t=1 } -> x[q],  { t=2 } -> x[q+1]
// so each of the 8 classes is a 2x2x2-tap conv on the LOW-resolution grid with pre-summed weights: 64 instead of 216
// multiply-adds per low-res voxel and channel pair (3.4x fewer flops and gather bytes), forward, data gradient (one
// 64-tap launch, no pooling pass) and weight gradient (8 launches + a fold of the 8x8 effective taps onto the 27 real ones).
// Zero padding commutes: a tap outside the upsampled volume is exactly a source voxel outside the low-res volume.
// Everything runs on the tap-table implicit GEMM kernels of gemm.hip / gemm_bf16.hip.
#include <stdlib.h>

#include "gemm_desc.h"

extern "C" long long ltu_wgrad_ws_floats(long long M, int N, int K);
int launch_nt_f32(const IGemmArgs& g, hipStream_t st);
int launch_tn_f32(WGradArgs& wa, hipStream_t st);

static inline int sub_off(int p, int a) { return p == 0 ? (a == 0 ? -1 : 0) : (a == 0 ? 0 : 1); }

static int run_nt(const IGemmArgs& g, int dtype, hipStream_t st) {
  if (dtype == LTU_BF16) return launch_nt_bf16(g, st);
  if (dtype == LTU_F32) return launch_nt_f32(g, st);
  return LTU_E_DTYPE;
}

extern "C" int ltu_upconv_fwd(const void* x, const void* wsub_f, const float* bias, void* y, int B, int H, int W, int D, int Ci,
                              int Co, int dtype, ltu_stream_t s) {
  if (Ci % 4 || Co % 4) return LTU_E_SHAPE;
  const size_t esz = dtype == LTU_BF16 ? 2 : 4;
  if (dtype == LTU_BF16 && !ltu_knob("LTU_NO_UPRING", 0)) {          // all 8 classes from one LDS halo brick, compile-time tap pattern
    const int hr = launch_upconv_ring_bf16(x, wsub_f, bias, y, B, H, W, D, Ci, Co, (hipStream_t)s);
    if (hr != 1) return hr;
  }
  if (dtype == LTU_BF16 && !ltu_knob("LTU_NO_CLASS_HALO", 0)) {      // the generic class kernel (run-time entry table)
    ClassHaloArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.w = wsub_f; a.bias = bias; a.o0 = y; a.o1 = y;
    a.B = B; a.H = H; a.W = W; a.D = D; a.C = Ci; a.lda = Ci;
    a.N = Co; a.n0 = Co; a.ldo0 = Co; a.ldo1 = Co;
    a.Hh = 2 * H; a.Wh = 2 * W; a.Dh = 2 * D; a.mh = a.mw = a.md = 2;
    a.wrow = 8 * Ci; a.ncls = 8; a.nent = 64;
    for (int cls = 0; cls < 8; ++cls) {
      const int p[3] = {cls >> 2, (cls >> 1) & 1, cls & 1};
      a.cls_p[cls][0] = (int8_t)p[0]; a.cls_p[cls][1] = (int8_t)p[1]; a.cls_p[cls][2] = (int8_t)p[2];
      for (int sl = 0; sl < 8; ++sl)
        a.ent[cls * 8 + sl] = ClsEntry{(int8_t)sub_off(p[0], sl >> 2), (int8_t)sub_off(p[1], (sl >> 1) & 1), (int8_t)sub_off(p[2], sl & 1),
                                       (int8_t)cls, (cls * Co * 8 + sl) * Ci};
    }
    const int hr = launch_conv_class_halo_bf16(a, (hipStream_t)s);
    if (hr != 1) return hr;
  }
  for (int cls = 0; cls < 8; ++cls) {
    const int p[3] = {cls >> 2, (cls >> 1) & 1, cls & 1};
    IGemmArgs g;
    memset(&g, 0, sizeof(g));
    g.nb = B; g.rh = H; g.rw = W; g.rd = D;
    g.M = (long long)B * H * W * D;
    g.N = Co; g.C = Ci; g.c0 = Ci; g.K = 8 * Ci;
    g.lda0 = Ci; g.lda1 = Ci;
    g.nseg = 1; g.wrow = 8 * Ci;
    g.sh = H; g.sw = W; g.sd = D; g.ups = 0;
    g.mh = g.mw = g.md = 1;
    g.ntaps = 8;
    for (int sl = 0; sl < 8; ++sl)
      g.tap[sl] = Tap{(int8_t)sub_off(p[0], sl >> 2), (int8_t)sub_off(p[1], (sl >> 1) & 1), (int8_t)sub_off(p[2], sl & 1), (int8_t)sl};
    g.a0 = x; g.a1 = x;
    g.w[0] = reinterpret_cast<const char*>(wsub_f) + (size_t)cls * Co * 8 * Ci * esz;
    g.bias[0] = bias;
    g.out_identity = 0;
    g.omh = g.omw = g.omd = 2;
    g.ooh = p[0]; g.oow = p[1]; g.ood = p[2];
    g.oh = 2 * H; g.ow = 2 * W; g.od = 2 * D;
    g.n0 = Co; g.o0 = y; g.o1 = y; g.ldo0 = Co; g.ldo1 = Co;
    const int rc = run_nt(g, dtype, (hipStream_t)s);
    if (rc) return rc;
  }
  return LTU_OK;
}

// dx[v] = sum over (class, slot) of g[2v + (p - 2 off)] . Weff[class][slot]^T      (wsub_d [Ci][64][Co])
extern "C" int ltu_upconv_dgrad(const void* grad, const void* wsub_d, void* dx, int B, int H, int W, int D, int Ci, int Co,
                                float* ws, long long ws_floats, int dtype, ltu_stream_t s) {
  if (Ci % 4 || Co % 4) return LTU_E_SHAPE;
  if (dtype == LTU_BF16 && !ltu_knob("LTU_NO_UPDGRAD_RING", 0)) {      // class-planar halo kernel (updgrad_ring.hip)
    const int hr = launch_updgrad_ring_bf16(grad, wsub_d, dx, B, H, W, D, Ci, Co, (hipStream_t)s);
    if (hr != 1) return hr;
  }
  IGemmArgs g;
  memset(&g, 0, sizeof(g));
  g.nb = B; g.rh = H; g.rw = W; g.rd = D;
  g.M = (long long)B * H * W * D;
  g.N = Ci; g.C = Co; g.c0 = Co; g.K = 64 * Co;
  g.lda0 = Co; g.lda1 = Co;
  g.nseg = 1; g.wrow = 64 * Co;
  g.sh = 2 * H; g.sw = 2 * W; g.sd = 2 * D; g.ups = 0;
  g.mh = g.mw = g.md = 2;
  g.ntaps = 64;
  for (int cls = 0; cls < 8; ++cls)
    for (int sl = 0; sl < 8; ++sl) {
      const int p[3] = {cls >> 2, (cls >> 1) & 1, cls & 1};
      const int a[3] = {sl >> 2, (sl >> 1) & 1, sl & 1};
      g.tap[cls * 8 + sl] = Tap{(int8_t)(p[0] - 2 * sub_off(p[0], a[0])), (int8_t)(p[1] - 2 * sub_off(p[1], a[1])),
                                (int8_t)(p[2] - 2 * sub_off(p[2], a[2])), (int8_t)(cls * 8 + sl)};
    }
  g.a0 = grad; g.a1 = grad;
  g.w[0] = wsub_d;
  g.out_identity = 1;
  g.n0 = Ci; g.o0 = dx; g.o1 = dx; g.ldo0 = Ci; g.ldo1 = Ci;
  g.part = dtype == LTU_BF16 ? ws : nullptr; g.part_floats = ws_floats;     // small grids split the 64*Co-long K loop (ltu_igemm_ws_floats(B*H*W*D, Ci, 64*Co))
  return run_nt(g, dtype, (hipStream_t)s);
}

// dW[co][ci][t] += sum over the 8 classes of dweff[class][co][slot(class, t)][ci]
__global__ void upconv_fold_kernel(const float* __restrict__ dweff, float* __restrict__ dw, int Co, int Ci, int CoP, int CiP) {
  const long long n = (long long)Co * Ci * 27;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i % 27);
    const int ci = (int)((i / 27) % Ci);
    const int co = (int)(i / (27LL * Ci));
    const int tt[3] = {t / 9, (t / 3) % 3, t % 3};
    float acc = 0.f;
    for (int cls = 0; cls < 8; ++cls) {
      const int p[3] = {cls >> 2, (cls >> 1) & 1, cls & 1};
      int sl = 0;
      for (int a = 0; a < 3; ++a) {
        const int slot = p[a] == 0 ? (tt[a] == 0 ? 0 : 1) : (tt[a] == 2 ? 1 : 0);
        sl = sl * 2 + slot;
      }
      acc += dweff[(((long long)cls * CoP + co) * 8 + sl) * CiP + ci];
    }
    dw[i] += acc;
  }
}

// dweff [8][Co][8][Ci] must be zero-filled by the caller; dw_torch [co_real][ci_real][27] and db are accumulated (+=)
extern "C" int ltu_upconv_wgrad(const void* grad, const void* x, float* dweff, float* db, float* dw_torch, int co_real,
                                int ci_real, float* ws, long long ws_floats, int blocks, int B, int H, int W, int D, int Ci, int Co, int dtype,
                                ltu_stream_t s) {
  if (Ci % 4 || Co % 4) return LTU_E_SHAPE;
  if (dtype == LTU_BF16 && ws != nullptr && !ltu_knob("LTU_NO_CLASS_HALO", 0)) {
    // all classes and taps from LDS halo bricks, folded in registers; the reduce kernel writes the PyTorch layout
    UpWgradArgs u;
    memset(&u, 0, sizeof(u));
    u.x = x; u.grad = grad; u.B = B; u.H = H; u.W = W; u.D = D; u.Ci = Ci; u.Co = Co; u.part = ws; u.part_floats = ws_floats; u.blocks = blocks;
    int nsplit = 0;
    const int hr = launch_upconv_wgrad_class_bf16(u, &nsplit, (hipStream_t)s);
    if (hr == LTU_OK) {
      WGradArgs wa;
      memset(&wa, 0, sizeof(wa));
      IGemmArgs& g = wa.g;
      g.N = Co; g.C = Ci; g.c0 = Ci; g.K = 27 * Ci; g.wrow = 27 * Ci; g.ntaps = 27;
      for (int t = 0; t < 27; ++t) g.tap[t] = Tap{0, 0, 0, (int8_t)t};
      wa.dw = dw_torch; wa.db = db; wa.t_co = co_real; wa.t_ci = ci_real; wa.nseg_w = 1;
      wa.part = ws; wa.part_floats = ws_floats; wa.npad = Co; wa.kpad = 27 * Ci;
      wa.bpart = ws + (long long)nsplit * Co * wa.kpad;
      return launch_wgrad_reduce(wa, nsplit, (hipStream_t)s);
    }
    if (hr != 1) return hr;
  }
  for (int cls = 0; cls < 8; ++cls) {
    const int p[3] = {cls >> 2, (cls >> 1) & 1, cls & 1};
    WGradArgs wa;
    memset(&wa, 0, sizeof(wa));
    IGemmArgs& g = wa.g;
    g.nb = B; g.rh = H; g.rw = W; g.rd = D;
    g.M = (long long)B * H * W * D;
    g.N = Co; g.C = Ci; g.c0 = Ci; g.K = 8 * Ci;
    g.lda0 = Ci; g.lda1 = Ci;
    g.nseg = 1; g.wrow = 8 * Ci;
    g.sh = H; g.sw = W; g.sd = D;
    g.mh = g.mw = g.md = 1;
    g.ntaps = 8;
    for (int sl = 0; sl < 8; ++sl)
      g.tap[sl] = Tap{(int8_t)sub_off(p[0], sl >> 2), (int8_t)sub_off(p[1], (sl >> 1) & 1), (int8_t)sub_off(p[2], sl & 1), (int8_t)sl};
    g.a0 = x; g.a1 = x;
    // gradient rows of this class sit at the upsampled voxels 2q + p
    g.out_identity = 0;
    g.omh = g.omw = g.omd = 2;
    g.ooh = p[0]; g.oow = p[1]; g.ood = p[2];
    g.oh = 2 * H; g.ow = 2 * W; g.od = 2 * D;
    g.n0 = Co;
    wa.grad = grad; wa.ldg = Co;
    wa.dw = dweff + (size_t)cls * Co * 8 * Ci;
    wa.db = db;
    wa.nseg_w = 1;
    int rc;
    if (dtype == LTU_BF16) { wa.part = ws; wa.part_floats = ws_floats; rc = launch_tn_bf16(wa, (hipStream_t)s); }
    else if (dtype == LTU_F32) rc = launch_tn_f32(wa, (hipStream_t)s);
    else return LTU_E_DTYPE;
    if (rc) return rc;
  }
  const long long n = (long long)co_real * ci_real * 27;
  long long fblocks = (n + 255) / 256;
  if (fblocks > 4096) fblocks = 4096;
  hipLaunchKernelGGL(upconv_fold_kernel, dim3((unsigned)fblocks), dim3(256), 0, (hipStream_t)s, dweff, dw_torch, co_real, ci_real, Co, Ci);
  return ltu_check_launch();
}

// workspace (floats) ltu_upconv_wgrad needs for M coarse voxels
extern "C" long long ltu_upconv_wgrad_ws_floats(long long M, int Co, int Ci, int blocks) {
  const long long a = ltu_wgrad_ws_floats(M, Co, 8 * Ci), b = upconv_wgrad_class_ws_floats(Ci, Co, blocks);
  return a > b ? a : b;
}
