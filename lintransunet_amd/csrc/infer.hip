// Sliding-window whole-volume inference: restates inference_embed_attn.py:92-185 and the constant-weight
// sliding_window_inference of monai 0.7.0 (monai/inferers/utils.py) on device.  Windows are gathered from the volume, the
// model's eval forward yields one-hot arg-max windows (trans_3DUnet.py:199-202), votes and the hit count are accumulated in
// HBM, the quotient is the blended prediction; the evaluation metrics of the driver (criterions.py DiceClassLoss 35-70,
// Recall 280-311, Precision 348-379, LocalizationLoss 179-241) are computed from per-row sums of the thresholded prediction.
#include "common.h"

static unsigned sgrid(long long n, int per_block = 256) {
  long long blocks = (n + per_block - 1) / per_block;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}
#define GRID_STRIDE(i, n) \
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)

// win [n][h][w][d] (one channel) <- vol [B][H][W][D] at padded coordinates (start - pad); zero outside the volume.
// desc [n][4] = (b, h0, w0, d0) in the padded image.
__global__ void window_gather_kernel(const float* __restrict__ vol, float* __restrict__ win, const int* __restrict__ desc, int n,
                                     int H, int W, int D, int h, int w, int d, int ph, int pw, int pd) {
  const long long per = (long long)h * w * d, total = per * n;
  GRID_STRIDE(i, total) {
    const int k = (int)(i / per);
    long long r = i - (long long)k * per;
    const int z = (int)(r % d); r /= d;
    const int y = (int)(r % w);
    const int x = (int)(r / w);
    const int b = desc[4 * k], sh = desc[4 * k + 1] + x - ph, sw = desc[4 * k + 2] + y - pw, sd = desc[4 * k + 3] + z - pd;
    float v = 0.f;
    if ((unsigned)sh < (unsigned)H && (unsigned)sw < (unsigned)W && (unsigned)sd < (unsigned)D)
      v = vol[(((long long)b * H + sh) * W + sw) * D + sd];
    win[i] = v;
  }
}

// votes [B][C][Hp][Wp][Dp] += seg (channels-last [n][h][w][d][C]); count [B][Hp][Wp][Dp] += 1.  Windows of one batch may
// overlap each other, hence atomics (at most a handful of writers per address).
__global__ void vote_accumulate_kernel(const float* __restrict__ seg, float* __restrict__ votes, float* __restrict__ count,
                                       const int* __restrict__ desc, int n, int Hp, int Wp, int Dp, int h, int w, int d, int C) {
  const long long per = (long long)h * w * d, total = per * n;
  const long long vol = (long long)Hp * Wp * Dp;
  GRID_STRIDE(i, total) {
    const int k = (int)(i / per);
    long long r = i - (long long)k * per;
    const int z = (int)(r % d); r /= d;
    const int y = (int)(r % w);
    const int x = (int)(r / w);
    const int b = desc[4 * k];
    const long long pos = ((long long)(desc[4 * k + 1] + x) * Wp + (desc[4 * k + 2] + y)) * Dp + (desc[4 * k + 3] + z);
    for (int c = 0; c < C; ++c) {
      const float v = seg[i * C + c];
      if (v != 0.f) atomicAdd(votes + ((long long)b * C + c) * vol + pos, v);
    }
    atomicAdd(count + (long long)b * vol + pos, 1.f);
  }
}

// out [B][C][H][W][D] = votes / count on the un-padded region
__global__ void vote_finalize_kernel(const float* __restrict__ votes, const float* __restrict__ count, float* __restrict__ out,
                                     int B, int C, int H, int W, int D, int Hp, int Wp, int Dp, int ph, int pw, int pd) {
  const long long total = (long long)B * C * H * W * D;
  GRID_STRIDE(i, total) {
    long long r = i;
    const int z = (int)(r % D); r /= D;
    const int y = (int)(r % W); r /= W;
    const int x = (int)(r % H); r /= H;
    const int c = (int)(r % C);
    const int b = (int)(r / C);
    const long long pos = ((long long)(x + ph) * Wp + (y + pw)) * Dp + (z + pd);
    const long long vol = (long long)Hp * Wp * Dp;
    out[i] = votes[((long long)b * C + c) * vol + pos] / count[(long long)b * vol + pos];
  }
}

// rows [B][3][H]: per h the sums over (W, D) of p, t and p*t, with p = [pred[b][ci] >= thr] (thr < 0: p = pred[b][ci] itself,
// the un-thresholded probabilities the evaluation losses of train3D.py:143 see) and t = target (0/1)
__global__ void __launch_bounds__(256) seg_row_sums_kernel(const float* __restrict__ pred, const uint8_t* __restrict__ target,
                                                           float* __restrict__ rows, int C, int ci, int H, long long WD, float thr) {
  __shared__ float red[3][256];
  const int hh = blockIdx.x, b = blockIdx.y;
  const float* p = pred + (((long long)b * C + ci) * H + hh) * WD;
  const uint8_t* t = target + ((long long)b * H + hh) * WD;
  float sp = 0.f, st = 0.f, spt = 0.f;
  for (long long i = threadIdx.x; i < WD; i += 256) {
    const float pv = thr < 0.f ? p[i] : (p[i] >= thr ? 1.f : 0.f), tv = (float)t[i];
    sp += pv; st += tv; spt += pv * tv;
  }
  red[0][threadIdx.x] = sp; red[1][threadIdx.x] = st; red[2][threadIdx.x] = spt;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      red[0][threadIdx.x] += red[0][threadIdx.x + o];
      red[1][threadIdx.x] += red[1][threadIdx.x + o];
      red[2][threadIdx.x] += red[2][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x < 3) rows[((long long)b * 3 + threadIdx.x) * H + hh] = red[threadIdx.x][0];
}

// values[0..3] = DiceClassLoss, Recall, Precision, LocalizationLoss (means over the batch).  One workgroup.
// LocalizationLoss as the reference computes it: its three "axes" all reduce to the H profile (criterions.py:203-212 flattens
// the un-transposed tensor for i != 0), so the average over axes is that one term.
__global__ void __launch_bounds__(256) seg_metrics_kernel(const float* __restrict__ rows, float* __restrict__ values, int B, int H) {
  __shared__ float acc[4];
  __shared__ float sh[2];
  if (threadIdx.x < 4) acc[threadIdx.x] = 0.f;
  __syncthreads();
  for (int b = 0; b < B; ++b) {
    const float* rp = rows + (long long)b * 3 * H;
    const float* rt = rp + H;
    const float* rx = rp + 2 * H;
    if (threadIdx.x == 0) {
      double sp = 0, st = 0, sx = 0, qp = 0, qt = 0;
      for (int i = 0; i < H; ++i) {
        sp += rp[i]; st += rt[i]; sx += rx[i];
        qp += 1.0 / (1.0 + exp(-((double)rp[i] - 10.0)));
        qt += 1.0 / (1.0 + exp(-((double)rt[i] - 10.0)));
      }
      acc[0] += (float)((2.0 * sx + 1e-9) / (sp + st + 1e-9));
      acc[1] += (float)((sx + 1e-5) / (st + 1e-5));
      acc[2] += (float)((sx + 1e-5) / (sp + 1e-5));
      sh[0] = (float)qp; sh[1] = (float)qt;
      double cp = 0, ct = 0, dsum = 0;
      for (int i = 0; i < H; ++i) {
        cp += 1.0 / (1.0 + exp(-((double)rp[i] - 10.0)));
        ct += 1.0 / (1.0 + exp(-((double)rt[i] - 10.0)));
        dsum += fabs(cp / (qp + 1e-6) - ct / (qt + 1e-6));
      }
      acc[3] += (float)(8.0 * dsum / H);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    values[0] = 1.f - acc[0] / B;
    values[1] = acc[1] / B;
    values[2] = acc[2] / B;
    values[3] = acc[3] / B;
  }
}

extern "C" int ltu_window_gather(const float* vol, float* win, const int* desc, int n, int H, int W, int D, int h, int w, int d,
                                 int ph, int pw, int pd, ltu_stream_t s) {
  if (n <= 0) return LTU_OK;
  const long long total = (long long)n * h * w * d;
  hipLaunchKernelGGL(window_gather_kernel, dim3(sgrid(total)), dim3(256), 0, (hipStream_t)s, vol, win, desc, n, H, W, D, h, w, d, ph, pw, pd);
  return ltu_check_launch();
}
extern "C" int ltu_vote_accumulate(const float* seg, float* votes, float* count, const int* desc, int n, int Hp, int Wp, int Dp,
                                   int h, int w, int d, int C, ltu_stream_t s) {
  if (n <= 0) return LTU_OK;
  if (h > Hp || w > Wp || d > Dp || C < 1) return LTU_E_SHAPE;
  const long long total = (long long)n * h * w * d;
  hipLaunchKernelGGL(vote_accumulate_kernel, dim3(sgrid(total)), dim3(256), 0, (hipStream_t)s, seg, votes, count, desc, n, Hp, Wp, Dp, h, w, d, C);
  return ltu_check_launch();
}
extern "C" int ltu_vote_finalize(const float* votes, const float* count, float* out, int B, int C, int H, int W, int D, int Hp,
                                 int Wp, int Dp, int ph, int pw, int pd, ltu_stream_t s) {
  const long long total = (long long)B * C * H * W * D;
  if (total <= 0) return LTU_OK;
  hipLaunchKernelGGL(vote_finalize_kernel, dim3(sgrid(total)), dim3(256), 0, (hipStream_t)s, votes, count, out, B, C, H, W, D, Hp, Wp, Dp, ph, pw, pd);
  return ltu_check_launch();
}
extern "C" int ltu_seg_metrics(const float* pred, const uint8_t* target, float* rows, float* values, int B, int C, int ci, int H,
                               long long WD, float threshold, ltu_stream_t s) {
  if (B <= 0 || H <= 0 || ci < 0 || ci >= C) return LTU_E_SHAPE;
  hipLaunchKernelGGL(seg_row_sums_kernel, dim3(H, B), dim3(256), 0, (hipStream_t)s, pred, target, rows, C, ci, H, WD, threshold);
  hipLaunchKernelGGL(seg_metrics_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, rows, values, B, H);
  return ltu_check_launch();
}

// ------------------------------------------------------------------------------------------------ largest connected component
// Post-processing of inference_multi_classes.py:104,148-151: monai KeepLargestConnectedComponent(applied_labels=[1, 2],
// independent=False, connectivity=3) on the rounded one-hot prediction -- the union of the foreground channels is labelled
// with 26-connectivity, every foreground voxel outside the largest component is cleared, channel 0 = 1 - sum of the others.
// Labels are (smallest voxel index of the component) + 1, found by min-propagation over the 3x3x3 neighbourhood with
// pointer jumping; ties between equally large components go to the smaller label = first in raster order, as
// skimage.measure.label + bincount/argmax decide them.
__global__ void cc_init_kernel(const float* __restrict__ pred, int* __restrict__ labels, int C, long long S) {
  GRID_STRIDE(i, S) {
    float fg = 0.f;
    for (int c = 1; c < C; ++c) fg += pred[(long long)c * S + i];
    labels[i] = fg > 0.f ? (int)i + 1 : 0;
  }
}
__global__ void cc_sweep_kernel(int* __restrict__ labels, int* __restrict__ changed, int H, int W, int D) {
  const long long S = (long long)H * W * D;
  GRID_STRIDE(i, S) {
    int l = labels[i];
    if (l == 0) continue;
    const int z = (int)(i % D), y = (int)((i / D) % W), x = (int)(i / ((long long)D * W));
    int m = l;
    for (int dx = -1; dx <= 1; ++dx)
      for (int dy = -1; dy <= 1; ++dy)
        for (int dz = -1; dz <= 1; ++dz) {
          const int xx = x + dx, yy = y + dy, zz = z + dz;
          if ((unsigned)xx >= (unsigned)H || (unsigned)yy >= (unsigned)W || (unsigned)zz >= (unsigned)D) continue;
          const int n = labels[((long long)xx * W + yy) * D + zz];
          if (n != 0 && n < m) m = n;
        }
    const int j = labels[m - 1];                 // pointer jumping: the representative's own label (never larger)
    if (j != 0 && j < m) m = j;
    if (m < l) { atomicMin(labels + i, m); *changed = 1; }
  }
}
__global__ void cc_count_kernel(const int* __restrict__ labels, int* __restrict__ counts, long long S) {
  GRID_STRIDE(i, S) {
    const int l = labels[i];
    if (l != 0) atomicAdd(counts + l, 1);
  }
}
__global__ void cc_best_kernel(const int* __restrict__ counts, unsigned long long* __restrict__ best, long long S) {
  GRID_STRIDE(i, S) {
    const int c = counts[i + 1];
    if (c > 0) atomicMax(best, ((unsigned long long)(unsigned)c << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)(i + 1)));
  }
}
__global__ void cc_apply_kernel(float* __restrict__ pred, const int* __restrict__ labels, const unsigned long long* __restrict__ best,
                                int C, long long S) {
  const int keep = (int)(0xFFFFFFFFu - (unsigned)(*best & 0xFFFFFFFFull));
  const bool any = (*best >> 32) != 0;
  GRID_STRIDE(i, S) {
    const int l = labels[i];
    float rest = 0.f;
    for (int c = 1; c < C; ++c) {
      float v = pred[(long long)c * S + i];
      if (l != 0 && any && l != keep) { v = 0.f; pred[(long long)c * S + i] = 0.f; }
      rest += v;
    }
    pred[i] = 1.f - rest;
  }
}

/* pred f32 [C][H][W][D] (rounded one-hot, modified in place); labels int32 [S], counts int32 [S+1] and best u64[1] scratch,
 * changed int32[1].  step 0: initialise labels; step 1: one propagation sweep (sets *changed when a label moved);
 * step 2: count, pick the largest component, clear the rest and rewrite channel 0. */
extern "C" int ltu_keep_largest_component(float* pred, int* labels, int* counts, unsigned long long* best, int* changed, int C, int H,
                                          int W, int D, int step, ltu_stream_t s) {
  const long long S = (long long)H * W * D;
  if (S <= 0 || S >= (1LL << 31) - 1 || C < 2) return LTU_E_SHAPE;
  hipStream_t st = (hipStream_t)s;
  if (step == 0) {
    hipLaunchKernelGGL(cc_init_kernel, dim3(sgrid(S)), dim3(256), 0, st, pred, labels, C, S);
  } else if (step == 1) {
    hipLaunchKernelGGL(cc_sweep_kernel, dim3(sgrid(S)), dim3(256), 0, st, labels, changed, H, W, D);
  } else if (step == 2) {
    hipLaunchKernelGGL(cc_count_kernel, dim3(sgrid(S)), dim3(256), 0, st, labels, counts, S);
    hipLaunchKernelGGL(cc_best_kernel, dim3(sgrid(S)), dim3(256), 0, st, counts, best, S);
    hipLaunchKernelGGL(cc_apply_kernel, dim3(sgrid(S)), dim3(256), 0, st, pred, labels, best, C, S);
  } else {
    return LTU_E_ARG;
  }
  return ltu_check_launch();
}
