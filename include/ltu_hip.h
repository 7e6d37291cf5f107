/* C-ABI of libltu_hip.so — the MI355X (gfx950) kernels behind the LinTransUNet hot path.
 *
 * The reference (freshman97/LinTransUNet) is pure PyTorch and has no FFI of its own; each entry
 * point below replaces the ATen op(s) reached from the cited reference lines (paths relative to
 * the reference root).  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - Activations are channels-last: a reference tensor [B,C,H,W,D] lives in HBM as [B,H,W,D,C]
 *     (C fastest), so a voxel is a token and 1x1x1 convs / Linear layers are the same GEMM.
 *   - `dtype` is the storage type of activations (LTU_F32 / LTU_BF16); weights, statistics,
 *     probabilities and gradients of weights are fp32; all accumulation is fp32.
 *   - Ownership: the caller allocates every buffer (outputs, workspaces, saved statistics).  The
 *     library never allocates, frees, retains pointers or synchronises; all work is enqueued on
 *     `stream`.  Entry points keep no state between calls and may be called from several threads
 *     (one process per GPU is the intended use).  The only process-global data are (a) the tuning
 *     knob table of ltu_config_set (mutex-protected; knobs are looked up per call: override, then
 *     the environment variable of the same name, then the default) and (b) per-device "dynamic
 *     LDS limit raised" latches for four kernels (atomic bit masks).
 *   - Dropout: (p, seed) select a counter-based hash mask (a murmur3 finalizer + a linear expansion to 64 bits per
 *     4-element group, csrc/common.h); the backward entry points regenerate
 *     the mask from the same (p, seed) instead of reading a stored one.  Independence: the four keep decisions of a group are
 *     PAIRWISE independent only (every pair of the four 16-bit words is a full-rank GF(2) image of the 32-bit hash; the third and
 *     fourth word are deterministic functions of the first two), not 4-wise independent like nn.Dropout's Bernoulli draws; groups
 *     are independent of each other.  tests/test_gpu_ops.py::test_dropout_group_pattern_histogram holds the joint 16-pattern
 *     histogram of a group against Bernoulli^4.  p = 0 disables.  `step` (nullable) is a
 *     device-resident counter mixed into the seed at run time, so a captured HIP graph draws fresh masks per replay.
 *   - Workspaces: every entry point that takes a caller-owned workspace / scratch buffer also takes its CAPACITY (the argument
 *     right behind the pointer, in floats unless it says elements) and returns LTU_E_ARG WITHOUT launching anything when the
 *     geometry it is about to launch needs more.  The *_ws_floats() queries tell what a call needs; launch geometry may depend on
 *     tuning knobs (ltu_config_set / environment), so a query and a launch made under different knob values can disagree - the
 *     capacity check turns that into an error code instead of a write past the end.  The two entry points whose width the caller
 *     chooses per launch (ltu_linear_wgrad_group, ltu_upconv_wgrad) take that width as an argument of both the query and the launch.
 *   - Return value: 0 = LTU_OK, negative = LTU_E_* argument error, positive = hipError_t.
 */
#ifndef LTU_HIP_H
#define LTU_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ltu_stream_t; /* hipStream_t */

enum { LTU_F32 = 0, LTU_BF16 = 1 };
enum { LTU_OK = 0, LTU_E_DTYPE = -1, LTU_E_SHAPE = -2, LTU_E_ALIGN = -3, LTU_E_ARG = -4,
       LTU_E_COMM = -100 /* RCCL not loaded / not loadable; LTU_E_COMM - r = RCCL returned ncclResult_t r > 0 */ };
enum { LTU_ACT_NONE = 0, LTU_ACT_LRELU = 1 };

int ltu_version(void);
/* bit 0: built with -DLTU_EXPERIMENTS (kernel variants that lost their measurements and ltu_selftest_last_arriver are compiled in) */
int ltu_build_flags(void);
/* Tuning / ablation knobs (grid sizes, split counts, kernel-variant switches; names = the LTU_* environment variables
 * listed in DESIGN.md).  Sets (clear = 0) or removes (clear != 0) a process-wide override that takes precedence over
 * the environment.  Results never depend on a knob beyond fp32 summation order.  Used by the tests to force code
 * paths (e.g. several tiles per split in the linear-attention reductions at small N). */
int ltu_config_set(const char* name, int value, int clear);
/* Self-test of the cross-lane reductions every norm / softmax / loss kernel builds on (DPP + v_permlane16/32_swap, common.h):
 * sum[i] / mx[i] = sum / max of x over the aligned group of G lanes (G = 2 .. 64, a power of two) that holds element i; n is a
 * multiple of 64.  Exists because the permlane-swap builtins of hipcc 7.2 miscompile (tests/test_gpu_ops.py keeps the inline-asm
 * replacement honest on the hardware). */
int ltu_selftest_group_reduce(const float* x, float* sum, float* mx, int n, int G, ltu_stream_t s);
#ifdef LTU_EXPERIMENTS      /* exported by an experiments build only (make EXPERIMENTS=1); ltu_build_flags() & 1 tells */
/* Self-test and price of an in-launch "last arriver" fold against the two-stage form the step uses (per-workgroup partial rows + a
 * second small launch): nwg workgroups with uneven load (workgroup i sums 1 + (7 i mod skew) chunks of rows_per_chunk rows of x
 * [rows][n], n <= 256) publish one partial row each; mode 0 folds them with a second launch, mode 1 inside the launch (write-through
 * partials, agent-scope ticket, one acquire by the last arriver: the guide's hand-off recipe R1).  Both fold in the same fixed
 * order, so `out` [n] must be bit-identical.  part: nwg * n floats (pre-read by every workgroup: an L1-warm consumer), counter: one
 * zeroed word (left zero by the last arriver), sink: nwg floats (never written for finite data).  tests/test_gpu_ops.py runs 10^4
 * launches of it in one process; profiles/r03_last_arriver.txt has the timings of both forms. */
int ltu_selftest_last_arriver(const float* x, float* part, float* out, unsigned* counter, float* sink, int nwg, int n,
                              int rows_per_chunk, int skew, int mode, ltu_stream_t s);
#endif

/* ---- window embedding: model/Unet_3Dblock.py:123-136 ------------------------------------------
 * x f32 [B,1,H,W,D] (reference layout) -> y T [B,H/2,W/2,D,8]; channel kh*2+kw, channels 4..7 = 0
 * (padding so that the stem conv gathers whole vectors). */
int ltu_window_embed(const float* x, void* y, int dtype, int B, int H, int W, int D, ltu_stream_t s);

/* ---- weight repacking (weights are tiny; done per step) ----------------------------------------
 * GEMM weight operands are consumed in the activation dtype: these produce them from the fp32 masters.
 * conv weight [Co,Ci,3,3,3] -> wf [CoP][27][CiP] (forward / weight-gradient operand) and/or
 * wd [CiP][27][CoP] (data-gradient operand), zero padded, stored as out_dtype; either output may be NULL. */
int ltu_pack_conv_weight(const float* w, void* wf, void* wd, int Co, int Ci, int CoP, int CiP, int out_dtype, ltu_stream_t s);
/* gradient back to the PyTorch layout: dwf [CoP][27][CiP] -> dw [Co,Ci,3,3,3] */
int ltu_unpack_conv_wgrad(const float* dwf, float* dw, int Co, int Ci, int CiP, ltu_stream_t s);
/* out[c*ldo + col_off + r] = in[r*C + c], stored as out_dtype (Linear weight for the data-gradient GEMM) */
int ltu_transpose_f32(const float* in, void* out, int R, int C, int ldo, int col_off, int out_dtype, ltu_stream_t s);
/* out[i] = (out_dtype) in[i] */
int ltu_cast_f32(const float* in, void* out, long long n, int out_dtype, ltu_stream_t s);
/* All of the above for a whole model in ONE launch.  table: n device-resident 40-byte records
 * { const float* src; void* dst; int kind, R, C, p0, p1, pad; } with kind 0 = cast (R*C elements),
 * 1 = transpose ([R][C] -> dst[c*p0 + p1 + r]), 2 = pack wf ([R=Co][C=Ci][27] -> [p0=CoP][27][p1=CiP]),
 * 3 = pack wd (-> [p1=CiP][27][p0=CoP]), 4 = fp32 copy of R*C elements (padded biases), 5 / 6 = the sub-pixel
 * operands of ltu_upconv_* ([8][CoP][8][CiP] and [CiP][64][CoP]), 7 = pack wd of one member of a fused conv group: columns
 * [off, off+cnt) of dst [p1=CiP][27][p0=stride], pad = off << 16 | cnt (its wf rows are a kind-2 record at a row offset),
 * 8 / 9 = MFMA fragment order of a dense weight [R][C] for the row-block chain kernels (ltu_layer_tail_*): element
 * ((ct * KS + ks) * 64 + lane) * 8 + j = W[ct*32 + lane%32][ks*16 + 8*(lane/32) + j] (kind 8, KS = C/16: outputs = rows) or
 * W[ks*16 + 8*(lane/32) + j][ct*32 + lane%32] (kind 9, KS = R/16: outputs = columns); R*C destination elements. */
int ltu_weight_prep(const void* table, int n, int out_dtype, ltu_stream_t s);
/* The same with the work dealt out in chunks of LTU_WPREP_CHUNK destination elements: chunks = nchunks device-resident
 * { int record; int first_element / LTU_WPREP_CHUNK } pairs (a model has hundreds of records of very different sizes); the grid
 * walks the list, one chunk per workgroup and trip (four for lists of >= 6 000 chunks), at most LTU_WPREP_BLOCKS workgroups wide
 * when that knob is set.  Destination element counts per kind: 0/1/4: R*C; 2/3: p0*27*p1; 5/6: 64*p0*p1; 7: cnt*27*p1. */
#define LTU_WPREP_CHUNK 4096
int ltu_weight_prep_chunks(const void* table, const int* chunks, int nchunks, int out_dtype, ltu_stream_t s);

/* ---- dense projections: nn.Linear (model/trans_block.py:144,156,166,187,189) and 1x1x1 convs
 *      (model/Unet_3Dblock.py:200-215).  y[M,N] (+)= a[M,K] . w[N,K]^T + bias;  w in the activation dtype, bias fp32.
 * Up to three weight blocks of N/nw rows each may be given (q,k,v fused: nw = 3); bias entries may
 * be NULL.  accumulate != 0 adds to y. */
int ltu_linear_fwd(const void* a, int lda, const void* const* w, int nw, const float* const* bias, void* y, int ldy,
                   int M, int N, int K, int accumulate, int dtype, ltu_stream_t s);
/* FFN front half: u [M,N] = a . w^T + bias and h = dropout(gelu(u)) (model/trans_block.py:203-208).  bf16 shapes of the
 * weight-stationary projection kernel get both from one launch (GELU + dropout in its epilogue); otherwise the projection and
 * ltu_gelu_dropout_fwd run back to back with identical results.  ltu_gelu_dropout_bwd(dh, u, ...) is the matching backward. */
int ltu_linear_gelu_fwd(const void* a, int lda, const void* w, const float* bias, void* u, void* h, int M, int N, int K, float p,
                        uint64_t seed, const uint64_t* step, int dtype, ltu_stream_t s);
/* ---- post-attention half of a transformer layer as ONE launch (model/trans_block.py:203-211), bf16 storage, d = 128 | 256 ----
 *   z1 = x + drop(a Wo^T + bo); t1 = LN1(z1); u = t1 W1^T + b1; h = drop(gelu(u)); z2 = t1 + drop(h W2^T + b2); y = LN2(z2)
 * a workgroup carries 32 token rows through the whole chain in LDS (the five separate launches - projection, LayerNorm,
 * projection + GELU, projection, LayerNorm - re-read every intermediate and are latency-bound on the small token levels).
 * wo / w1 / w2: bf16 weights in MFMA fragment order (ltu_weight_prep kind 8 of the [out][in] fp32 masters).  All intermediate
 * tensors the backward pass needs are written (z1, t1, u, h, z2: bf16; stat1 / stat2 [M][2] = mean, rstd).  Rounding points and
 * dropout masks are those of the op-by-op path (ltu_linear_fwd, ltu_layernorm_fwd, ltu_linear_gelu_fwd).
 * u_mode 0: `u` receives the FFN pre-activation (what ltu_gelu_dropout_bwd expects); u_mode 1: it receives
 * dropout_mask * gelu'(u) instead, the factor ltu_layer_tail_bwd (same u_mode) multiplies dh by - the forward pass has the
 * Gaussian terms at hand, and the backward chain then needs no exp / rcp / mask hash for this stage.
 * qkv != NULL fuses phase B of the linear attention in front of the chain (trans_block.py:41-67 after the context has been
 * formed): the block's q rows are read from qkv [M][3d], `ctx` [B*H][32][32] is the merged context of ltu_linattn_ctx,
 * ntok = tokens per sample (a multiple of 32); `a` is then an OUTPUT ([M][d], the attention output the out-projection weight
 * gradient needs) and qstat [M][H][2] receives the row statistics ltu_linattn_bwd expects.  qkv == NULL: `a` is the input.
 * With qkv_next != NULL the q | k | v projection of the NEXT layer (model/trans_block.py:156-158, which follows :210 of this layer
 * with nothing in between) is formed from the block's y rows before they leave LDS: qkv_next [M][3d] = y cat(Wq, Wk, Wv)^T + cat(b);
 * wq_next = the three [d][d] weights of that layer in fragment order, back to back (ltu_weight_prep kind 8 each), bq0..2 their biases. */
int ltu_layer_tail_fwd(const void* a, const void* x, const void* wo, const void* w1, const void* w2, const float* bo,
                       const float* b1, const float* b2, const float* g1, const float* be1, const float* g2, const float* be2,
                       void* z1, void* t1, void* u, void* h, void* z2, void* y, float* stat1, float* stat2, long long M, int d,
                       float eps, float p, uint64_t seed1, uint64_t seedg, uint64_t seed2, const uint64_t* step, int u_mode,
                       const void* qkv, const float* ctx, float* qstat, int ntok, const void* wq_next, const float* bq0,
                       const float* bq1, const float* bq2, void* qkv_next, int dtype, ltu_stream_t s);

/* The backward of the same chain as ONE launch: LayerNorm 2 backward, data gradients through linear2 / GELU + dropout / linear1,
 * LayerNorm 1 backward (on dt1 + dz2), data gradient through the out projection.  dy2 (nullable): second gradient of y, summed on
 * load.  w2t / w1t / wot: fragment-ordered TRANSPOSED weights (ltu_weight_prep kind 9 of the fp32 masters).  Outputs: da, dz1
 * (gradient of the residual input x) and dr2 / du / dr1 = the G operands of the three weight gradients (ltu_linear_wgrad_group
 * with X = h / t1 / a).  `u` / u_mode: as written by ltu_layer_tail_fwd with the same u_mode.  lnws2 / lnws1: ltu_layer_tail_blocks(M) x 2d floats each = per-workgroup (gamma, beta) column sums,
 * interleaved, folded by ltu_reduce_batch (mode 1). */
long long ltu_layer_tail_blocks(long long M);
int ltu_layer_tail_bwd(const void* dy, const void* dy2, const void* z2, const void* z1, const void* u, const float* stat2,
                       const float* stat1, const float* g2, const float* g1, const void* w2t, const void* w1t, const void* wot,
                       void* dr2, void* du, void* dr1, void* dz1, void* da, float* lnws2, float* lnws1, long long lnws_floats /* each */,
                       long long M, int d, float p,
                       uint64_t seed1, uint64_t seedg, uint64_t seed2, const uint64_t* step, int u_mode, int dtype, ltu_stream_t s);

/* ---- deferred second stage of the two-stage reductions ------------------------------------------
 * ltu_linear_wgrad / ltu_layernorm_bwd can leave the folding of their per-split partial sums to the caller: pass a job
 * record, collect a few, and fold them with ONE launch (ltu_reduce_batch) before the gradients are read.  A record whose
 * `part` comes back NULL needs nothing (the call reduced in place).  The workspace handed to the producing call must stay
 * untouched until the batch has run. */
typedef struct ltu_reduce_job {
  const float* part;   /* mode 0: [nsplit][n][k] tiles followed by [nsplit][n] bias partials; mode 1: [nsplit][n] */
  int nsplit, n, k, nseg;
  float* out[3];       /* mode 0: weight-gradient blocks [n/nseg][k] (+=);  mode 1: out[0][i/2] (i even), out[1][i/2] (i odd) */
  float* outb[3];      /* mode 0: bias-gradient blocks (nullable) */
  int mode;
} ltu_reduce_job;
int ltu_reduce_batch(const ltu_reduce_job* jobs, int njobs, ltu_stream_t s);   /* jobs: host array */

/* dw_i[N/nw,K] += g[:, block i]^T . a[M,K];  db_i += column sums of g (fp32 gradients, accumulated; db may be NULL).
 * ws: optional workspace of ltu_wgrad_ws_floats(M,N,K) floats - with it (bf16) the row-split partial tiles are stored and
 * summed by a second kernel (no atomics, one launch for all nw blocks); without it fp32 atomics are used. */
long long ltu_wgrad_ws_floats(long long M, int N, int K);
int ltu_linear_wgrad(const void* g, int ldg, const void* a, int lda, float* const* dw, float* const* db, int nw, int M, int N,
                     int K, float* ws, long long ws_floats, ltu_reduce_job* defer, int dtype, ltu_stream_t s);
/* Several of the above in ONE launch + ONE fold: the weight / bias gradients of the four projections of a transformer layer
 * (model/trans_block.py:144,156,166,187,189: q,k,v as one job with nw = 3, out, linear1, linear2).  Together they offer 8-32
 * output tiles, so a few row splits per tile fill the chip and the fp32 partial tiles shrink 4x against four separate calls.
 * The host side may also hand in the jobs of SEVERAL layers of one transformer at once (small levels, where a layer's group is
 * launch-latency-bound): all jobs of a group share one split count, so they should have the same M.
 * jobs: host array (<= LTU_WGRAD_GROUP_MAX); blocks: the workgroup budget of the launch (<= 0: the library default, 256 = one per
 * CU; train.GraphedStep passes 128 for launches that run on its side stream beside the main chain).  The geometry (tile types,
 * per-job split counts, workspace layout) is a function of the jobs' shapes and `blocks` only; the size query and the launch take
 * the same `blocks`, and the launch is told the capacity of ws (ws_floats) and returns LTU_E_ARG without launching when its
 * geometry needs more.  ltu_linear_wgrad_group_ws_floats() == 0: this group is not handled - use ltu_linear_wgrad per job (handled:
 * bf16 storage, N and K multiples of 128, M a multiple of 32 and >= 1024). */
#define LTU_WGRAD_GROUP_MAX 32
typedef struct ltu_wgrad_job {
  const void* grad;    /* g [M][ldg] */
  const void* a;       /* a [M][lda] */
  float* dw[3];        /* nw gradient blocks [N/nw][K] (+=) */
  float* db[3];        /* nw bias-gradient blocks (nullable) */
  int ldg, lda, nw, M, N, K;
} ltu_wgrad_job;
long long ltu_linear_wgrad_group_ws_floats(const ltu_wgrad_job* jobs, int njobs, int blocks);
int ltu_linear_wgrad_group(const ltu_wgrad_job* jobs, int njobs, int blocks, float* ws, long long ws_floats, int dtype, ltu_stream_t s);

/* Workspace (floats) that ltu_upconv_wgrad needs for M = B*H*W*D coarse voxels (bf16 path; sub-pixel un-embedding of
 * model/Unet_3Dblock.py:419-432) at a workgroup budget of `blocks` (<= 0: the library default; the launch takes the same value). */
long long ltu_upconv_wgrad_ws_floats(long long M, int Co, int Ci, int blocks);

/* ---- 3x3x3 convolution, padding 1: model/Unet_3Dblock.py:310,314,375,421,523,528,588,1328,1353 --
 * x0 [B,Hi,Wi,Di,C0] (+ optional x1 [..,C1]: the channel concat of Unet_3Dblock.py:553 without
 * materialising it), wf [Co][27][C0+C1], stride (sh,sw,sd) in {1,2}; ups != 0: the conv reads the
 * nearest-neighbour x2 upsampling of x0 (nn.Upsample of Unet_3Dblock.py:421), (Hi,Wi,Di) are then
 * the physical dims.  y [B,Ho,Wo,Do,Co]. */
int ltu_conv3d_fwd(const void* x0, const void* x1, const void* wf, const float* bias, void* y, int B, int Hi, int Wi,
                   int Di, int C0, int C1, int Co, int sh, int sw, int sd, int ups, float* ws, long long ws_floats, int dtype,
                   ltu_stream_t s);
/* ws of ltu_conv3d_fwd / ltu_conv3d_dgrad (optional, bf16 stride-1 convs only): floats of workspace that let a conv over a
 * small grid (the deep U-Net levels) split its input channels over several workgroups per tile; 0 when the shape does not
 * split.  (B,H,W,D) = the conv's output grid, C = input channels of the call (Co for the data gradient), Co = its outputs. */
long long ltu_conv3d_ws_floats(int B, int H, int W, int D, int C, int Co);
/* the same for the gather implicit GEMMs (strided convs: M output voxels, N = Co, K = 27*C; ltu_upconv_dgrad: M = B*H*W*D,
 * N = Ci, K = 64*Co): floats of workspace that let a small grid split its K loop, 0 when it does not split */
long long ltu_igemm_ws_floats(long long M, int N, int K);
/* Two stride-1 convs over the same input in one pass (a decoder level's conv1 and its mask head read the same upsampled tensor:
 * model/Unet_3Dblock.py:1353 + 1380): wf [N0+N1][27][C], bias [N0+N1]; columns [0,N0) -> y0 [..,N0], the rest -> y1 [..,N1].
 * Data gradient of the pair: g0 [..,N0] and g1 [..,N1] read as one virtual concat against wd [C][27][N0+N1] -> dx [..,C]
 * (no separate add of two input gradients).  ws as for ltu_conv3d_fwd with Co = N0+N1 (C = N0+N1, Co = C for the gradient). */
int ltu_conv3d_pair_fwd(const void* x, const void* wf, const float* bias, void* y0, void* y1, int B, int H, int W, int D, int C,
                        int N0, int N1, float* ws, long long ws_floats, int dtype, ltu_stream_t s);
int ltu_conv3d_pair_dgrad(const void* g0, const void* g1, const void* wd, void* dx, int B, int H, int W, int D, int C, int N0,
                          int N1, float* ws, long long ws_floats, int dtype, ltu_stream_t s);
/* weight gradients of the pair (+=) straight into the two PyTorch-layout gradients dwa [co_a][ci][3][3][3], dwb [co_b][ci][3][3][3]
 * and dba / dbb; ws: ltu_wgrad_ws_floats(B*H*W*D, N0+N1, 27*C) floats (one pass over x for both) or NULL. */
int ltu_conv3d_pair_wgrad(const void* g0, const void* g1, const void* x, float* dwa, float* dba, float* dwb, float* dbb, int B,
                          int H, int W, int D, int C, int N0, int N1, int co_a, int co_b, int ci, float* ws, long long ws_floats,
                          int dtype, ltu_stream_t s);
/* data gradient: g [B,Ho,Wo,Do,Co], wd [C0+C1][27][Co] -> dx0 [B,Hl,Wl,Dl,C0] (+ dx1 [..,C1]); (Hl,Wl,Dl)
 * are the LOGICAL input dims (= 2x physical when the forward used ups: pool with ltu_sumpool2). */
int ltu_conv3d_dgrad(const void* g, const void* wd, void* dx0, void* dx1, int B, int Hl, int Wl, int Dl, int C0,
                     int C1, int Co, int sh, int sw, int sd, float* ws, long long ws_floats, int dtype, ltu_stream_t s);
/* weight gradient (+=, the caller zero-fills): torch_co == 0: into the packed layout dwf [Co][27][C0+C1];
 * torch_co != 0: straight into a PyTorch-layout gradient [torch_co][torch_ci][3][3][3] (padded rows/channels dropped).
 * db[Co] += column sums of g.
 * ws: optional workspace of ltu_wgrad_ws_floats(B*Ho*Wo*Do, Co, 27*(C0+C1)) floats (two-stage reduction, see above). */
int ltu_conv3d_wgrad(const void* g, const void* x0, const void* x1, float* dwf, float* db, int B, int Hi, int Wi,
                     int Di, int C0, int C1, int Co, int sh, int sw, int sd, int ups, int torch_co, int torch_ci, float* ws,
                     long long ws_floats, int dtype, ltu_stream_t s);
/* ---- nearest x2 upsampling + 3x3x3 conv as a sub-pixel conv: model/Unet_3Dblock.py:419-432 (UpEmbedBlock) ------------
 * Same result as ltu_conv3d_* with ups = 1 at 64/216 of the multiply-adds: the 8 output parity classes are 2x2x2-tap
 * convs on the low-res grid with pre-summed weights (ltu_weight_prep kinds 5 / 6).  x [B,H,W,D,Ci] -> y [B,2H,2W,2D,Co]. */
int ltu_upconv_fwd(const void* x, const void* wsub_f, const float* bias, void* y, int B, int H, int W, int D, int Ci, int Co,
                   int dtype, ltu_stream_t s);
int ltu_upconv_dgrad(const void* g, const void* wsub_d, void* dx, int B, int H, int W, int D, int Ci, int Co, float* ws,
                     long long ws_floats, int dtype, ltu_stream_t s);
/* dweff: zero-filled scratch [8][Co][8][Ci] fp32; dw_torch [co_real][ci_real][3][3][3] += and db[Co] += ;
 * ws: optional ltu_upconv_wgrad_ws_floats(B*H*W*D, Co, Ci, blocks) floats (bf16 two-stage reduction); blocks: workgroup budget of the
 * launch (<= 0: the library default), the value the size query was given */
int ltu_upconv_wgrad(const void* g, const void* x, float* dweff, float* db, float* dw_torch, int co_real, int ci_real, float* ws,
                     long long ws_floats, int blocks, int B, int H, int W, int D, int Ci, int Co, int dtype, ltu_stream_t s);
/* y[b,h,w,d,c] = sum of the 2x2x2 children of x [B,2H,2W,2D,C] (adjoint of nearest x2 upsampling) */
int ltu_sumpool2(const void* x, void* y, int B, int H, int W, int D, int C, int dtype, ltu_stream_t s);

/* ---- linear attention core: model/trans_block.py:41-67 -----------------------------------------
 * qkv [B*N][3d] (q | k | v; head h = columns h*32..h*32+31 of each third) -> out [B*N][d].
 * Saved for backward: ctx [B*H][32][32], colstats [B*H][64] (column max | column sum of exp),
 * qstat [B*N][H][2] (row max, 1/(rowsum*sqrt(32))).  part_ws: ltu_linattn_ws_floats(B, N, d) floats (= B * (ns + ns/16 + 2) * H * 1088
 * for the split count ns the launches pick; the backward needs B * ns * H * 1024 of them); ltu_linattn_splits(B, N) is an upper
 * bound of ns over d. */
int ltu_linattn_splits(int B, int N);
long long ltu_linattn_ws_floats(int B, int N, int d);
int ltu_linattn_fwd(const void* qkv, void* out, float* ctx, float* colstats, float* qstat, float* part_ws, long long ws_floats, int B,
                    int N, int d, int dtype, ltu_stream_t s);
/* phase A of ltu_linattn_fwd alone: ctx [B*H][32][32] and colstats [B*H][64] (same workspace); phase B then runs inside
 * ltu_layer_tail_fwd (its qkv / ctx / qstat arguments) */
int ltu_linattn_ctx(const void* qkv, float* ctx, float* colstats, float* part_ws, long long ws_floats, int B, int N, int d, int dtype,
                    ltu_stream_t s);
/* dqkv [B*N][3d] from dout [B*N][d]; dctx [B*H][32][32] is a scratch output; tvec [B*H][32] is reserved (may be NULL: the term it
 * held is formed inside the per-token kernel since round 2) */
int ltu_linattn_bwd(const void* qkv, const void* dout, const float* ctx, const float* colstats, const float* qstat,
                    void* dqkv, float* dctx, float* tvec, float* part_ws, long long ws_floats, int B, int N, int d, int dtype,
                    ltu_stream_t s);

/* ---- InstanceNorm3d (+LeakyReLU, residual, dropout): model/Unet_3Dblock.py:312-339,526-556,593 ----
 * x [B][S][C].  sums [B][C][3] = {shift, sum(x-shift), sum((x-shift)^2)} (zero-filled by the caller).
 * apply: y = dropout(act((x-mean)*rstd)) + res (res may be NULL). */
/* `ws` (nullable) is a scratch buffer of at least ltu_norm_ws_floats() floats: with it the per-block partial sums are folded by a
 * second small kernel instead of fp32 atomics (same results up to summation order).  One buffer can serve every call on a stream. */
long long ltu_norm_ws_floats(void);
int ltu_instnorm_stats(const void* x, float* sums, float* ws, long long ws_floats, int B, long long S, int C, int dtype, ltu_stream_t s);
int ltu_instnorm_apply(const void* x, const float* sums, const void* res, void* y, int B, long long S, int C, int act,
                       float slope, float p, uint64_t seed, const uint64_t* step, int dtype, ltu_stream_t s);
/* stats + apply in ONE call (what the model's forward uses): when the shape qualifies (bf16 or fp32, C a power of two, a few KB
 * of partial sums per sample) the statistics kernel leaves its per-chunk partials in `ws` and every workgroup of the apply kernel
 * folds them itself - no fold launch in between; workgroup 0 of each sample publishes sums[b][c][1..2] for the backward pass.
 * Otherwise exactly ltu_instnorm_stats followed by ltu_instnorm_apply. */
int ltu_instnorm_fwd(const void* x, float* sums, float* ws, long long ws_floats, const void* res, void* y, int B, long long S, int C, int act,
                     float slope, float p, uint64_t seed, const uint64_t* step, int dtype, ltu_stream_t s);
/* dx from dy (+ dy2 + dy3, nullable: the gradients of further consumers of y, summed on load instead of by a stand-alone add pass:
 * the skip tensors of the U-Net and the transformer inputs have two or three consumers); bsums [B][C][2] zero-filled scratch */
int ltu_instnorm_bwd(const void* dy, const void* dy2, const void* dy3, const void* x, const float* sums, float* bsums, float* ws,
                     long long ws_floats, void* dx, int B, long long S, int C, int act, float slope, float p, uint64_t seed, const uint64_t* step,
                     int dtype, ltu_stream_t s);

/* ---- residual LayerNorm: model/trans_block.py:205-206,209-210 -----------------------------------
 * y = LN(x + dropout(r)) * gamma + beta over rows of d in {32,64,128,256}; r is OVERWRITTEN with the
 * pre-norm sum z; stat [M][2] = {mean, rstd}. */
int ltu_layernorm_fwd(const void* x, void* r, const float* gamma, const float* beta, void* y, float* stat, long long M,
                      int d, float eps, float p, uint64_t seed, const uint64_t* step, int dtype, ltu_stream_t s);
/* dz (gradient of x) and dr = dz*dropmask (dr may alias dz when p = 0); dgamma/dbeta += (zero-filled) */
/* dy2 (nullable): a second upstream gradient, summed with dy on load (the layer output feeds both the next projection and the next
 * residual; folding the sum here saves autograd's separate add pass);
 * ws (nullable: atomics): at least 2 * d * ceil(M / rows) floats of partial (gamma, beta) rows with rows >= ceil(M / 1024), i.e. 2048 * d
 * floats always suffice (ltu_norm_ws_floats() too) */
int ltu_layernorm_bwd(const void* dy, const void* dy2, const void* z, const float* stat, const float* gamma, void* dz, void* dr,
                      float* dgamma, float* dbeta, float* ws, long long ws_floats, ltu_reduce_job* defer, long long M, int d, float p, uint64_t seed,
                      const uint64_t* step, int dtype, ltu_stream_t s);

/* ---- GELU(erf) + dropout: model/trans_block.py:208 ---------------------------------------------- */
int ltu_gelu_dropout_fwd(const void* u, void* h, long long n, float p, uint64_t seed, const uint64_t* step, int dtype, ltu_stream_t s);
int ltu_gelu_dropout_bwd(const void* dh, const void* u, void* du, long long n, float p, uint64_t seed, const uint64_t* step, int dtype,
                         ltu_stream_t s);

/* ---- class-probability heads ---------------------------------------------------------------------
 * mask head softmax (model/Unet_3Dblock.py:1380-1381): logits T [M][CP] (first C valid) -> p f32 [M][C] */
int ltu_head_softmax_fwd(const void* z, float* p, long long M, int C, int CP, int dtype, ltu_stream_t s);
int ltu_head_softmax_bwd(const float* dp, const float* p, void* dz, long long M, int C, int CP, int dtype, ltu_stream_t s);
/* final head (model/Unet_3Dblock.py:1392-1394): z [B,h,w,D,4C] -> window un-embedding + softmax -> p f32 [B,2h,2w,D,C] */
int ltu_final_softmax_fwd(const void* z, float* p, int B, int h, int w, int D, int C, int CP, int dtype, ltu_stream_t s);   /* z rows of CP >= 4C channels (padded conv output) */
int ltu_final_softmax_bwd(const float* dp, const float* p, void* dz, int B, int h, int w, int D, int C, int CP, int dtype,
                          ltu_stream_t s);
/* eval branch (model/trans_3DUnet.py:199-202): one-hot of the arg-max class, p/o f32 [M][C] */
int ltu_onehot_argmax(const float* p, float* o, long long M, int C, ltu_stream_t s);

/* ---- attention gate: model/Unet_3Dblock.py:217-221 + 1385 ---------------------------------------
 * u1 = Wx.skip, u2 = Wg.up ([B][S][C], from ltu_linear_fwd), sums1/sums2 their InstanceNorm sums.
 * a = sigmoid(psi . relu(IN(u1)+IN(u2)) + b) -> a_out f32 [B*S]; out = skip * a. */
int ltu_gate_fwd(const void* u1, const void* u2, const float* sums1, const float* sums2, const float* psi_w,
                 const float* psi_b, const void* skip, float* a_out, void* out, int B, long long S, int C, int dtype,
                 ltu_stream_t s);
/* -> dskip (direct path), du1, du2, dpsi_w[C] +=, dpsi_b[1] +=; ds_ws f32 [B*S], bs1/bs2 [B][C][2] zero-filled scratch */
int ltu_gate_bwd(const void* dout, const void* u1, const void* u2, const float* sums1, const float* sums2,
                 const float* psi_w, const void* skip, const float* a_in, void* dskip, float* ds_ws, float* dpsi_w,
                 float* dpsi_b, float* bs1, float* bs2, float* ws, long long ws_floats, void* du1, void* du2, int B, long long S, int C,
                 int dtype, ltu_stream_t s);   /* ws: ltu_norm_ws_floats() floats (two-stage reduction) or NULL (atomics) */

/* ---- positional depthwise conv: model/trans_block.py:86-96 on the grid of Unet_3Dblock.py:267-270 ----
 * y = chan_dropout(x + dwconv3x3x3(x) + bias), x [B,H,W,D,C]; w [C,1,3,3,3] with the reference's kernel
 * axes (D,H,W); nn.Dropout3d draws one keep/drop per (sample, channel). */
int ltu_dwconv_fwd(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int D, int C, float p,
                   uint64_t seed, const uint64_t* step, int dtype, ltu_stream_t s);
/* dx, and dw [C][27] +=, db [C] += (zero-filled by the caller); dy2 (nullable): gradient of a second consumer, summed on load.
 * ws: ltu_dwconv_bwd_ws_floats(...) floats for the per-workgroup partial sums of dw / db, folded in a fixed order; NULL falls
 * back to float atomics (the two gradients then differ in the last bits from run to run).  dx NULL: only dw / db are computed;
 * dw NULL: only dx (the two halves are independent launches: the weight gradient can be issued later, off the data-gradient chain). */
long long ltu_dwconv_bwd_ws_floats(int B, int H, int W, int D, int C, int dtype);
int ltu_dwconv_bwd(const void* dy, const void* dy2, const void* x, const float* w, void* dx, float* dw, float* db, float* ws,
                   long long ws_floats, int B, int H, int W, int D, int C, float p, uint64_t seed, const uint64_t* step, int dtype, ltu_stream_t s);

/* ---- dynamic ROI: model/Unet_3Dblock.py:821-873, 37-49 (box), 51-82 (index maps), 985-1117 (warps) ----
 * prob f32 [B,H,W,D,C]: foreground = (1 - prob[...,0]) >= thr.  Writes box [B][6] = (x0,y0,0,x1,y1,D-1)
 * and the sampling plans (sizes from ltu_roi_plan_size) used by ltu_roi_resample.  hist: n_hist ints of zero-filled scratch
 * (foreground histograms; the library issues no memsets of its own, see csrc/roi.hip). */
int ltu_roi_plan_size(int B, int H, int W, int roi_size, long long* n_int, long long* n_float, long long* n_hist);
int ltu_roi_plan(const float* prob, int B, int H, int W, int D, int C, int roi_size, float thr, float* box, int* plan_i,
                 float* plan_f, int* hist, ltu_stream_t s);
/* which = 0: image [B,H,W,D,C] -> ROI grid [B,eh,ew,D,C] (roi_alignment2); which = 1: ROI grid -> image
 * (post_processing2).  adjoint != 0 applies the transposed operator to a gradient (in has the shape of the
 * forward output, out the shape of the forward input). */
int ltu_roi_resample(const void* in, void* out, int* plan_i, float* plan_f, int which, int adjoint, int B, int H, int W,
                     int D, int C, int roi_size, int dtype, ltu_stream_t s);

/* ---- trilinear x(2,2,sd) upsampling, align_corners=True: model/Unet_3Dblock.py:1341-1345,1375-1378 ----
 * forward: in [B,H,W,D,C] -> out [B,2H,2W,sd*D,C]; adjoint != 0: in = gradient of the output, in2 (nullable, adjoint only) =
 * gradient of a second consumer of the output, summed on load. */
int ltu_trilinear_up(const void* in, const void* in2, void* out, int adjoint, int B, int H, int W, int D, int C, int sd, int dtype,
                     ltu_stream_t s);
/* The adjoint in separable form (one 1-D transposed interpolation per upsampled axis: at most 5 candidates per output instead of
 * ~64 gathers).  ws: ltu_trilinear_adjoint_ws_elems(...) elements of the storage type; intermediates are rounded to it. */
long long ltu_trilinear_adjoint_ws_elems(int B, int H, int W, int D, int C, int sd);
int ltu_trilinear_adjoint(const void* dy, const void* dy2, void* dx, void* ws, long long ws_elems, int B, int H, int W, int D, int C, int sd,
                          int dtype, ltu_stream_t s);

/* ---- deep-supervision losses of one level: loss/criterions.py:35-70,416-442,696-735;
 *      loss/multi_criterions.py:58-110,594-615 ------------------------------------------------------
 * p f32 [B][S][C] probabilities, label u8 [B][S].  total = w_ce*CE + w_bal*BalancedDice + sum_c w_dice[c]*Dice_c
 * + w_dice[4]*Dice of the foreground union (1 - p_0 vs label != 0: multi_criterions.py:30-56, DiceClassLoss0); w_dice: 5 host floats.
 * values (9 floats): [0] = total, [1] = CE, [2] = balanced Dice, [3+c] = Dice_c, [7] = union Dice, [8] = total again;
 * sums: scratch of ltu_loss_ws_floats(B, S, C) floats (no initialisation needed: per-block partial sums behind the final
 * [B][C][4] sums, folded in a fixed order - no floating-point atomics, the loss and its gradient are reproducible bit for bit);
 * coef [B][C][3] feeds ltu_loss_bwd: dp = gscale[0] * dTotal/dp.  scale_dev (nullable): device-resident factor on all three
 * weights, read at run time (the per-epoch level weight of train3D.py:122-137 divided by the accumulation count of
 * utils/utils_3D_embed_full.py:85, so a captured graph follows both without re-capture). */
long long ltu_loss_ws_floats(int B, long long S, int C);
int ltu_loss_fwd(const float* p, const uint8_t* label, float* sums, long long sums_floats, float* values, float* coef, int B, long long S, int C,
                 float w_ce, float w_bal, const float* w_dice, const float* scale_dev, ltu_stream_t s);
int ltu_loss_bwd(const float* p, const uint8_t* label, const float* coef, const float* gscale, float* dp, int B,
                 long long S, int C, ltu_stream_t s);
/* label pyramid (utils/utils_3D_embed_full.py:64,73-76): u8 [B,H,W,D] -> max over (2,2,kd) windows */
int ltu_label_maxpool(const uint8_t* x, uint8_t* y, int B, int H, int W, int D, int kd, ltu_stream_t s);

/* ---- sliding-window whole-volume inference (inference_embed_attn.py:92-185; monai 0.7.0 sliding_window_inference,
 * constant blending) ------------------------------------------------------------------------------------------------
 * desc int32 [n][4] = (sample, h0, w0, d0) of each window in the PADDED image (a dimension smaller than the window is padded
 * symmetrically with zeros: pad_lo = (roi - dim) / 2).  vol f32 [B][H][W][D] (one channel); win f32 [n][h][w][d];
 * seg f32 [n][h][w][d][C] = the model's eval output (channels-last one-hot); votes f32 [B][C][Hp][Wp][Dp] and
 * count f32 [B][Hp][Wp][Dp] zero-filled accumulators; out f32 [B][C][H][W][D] = votes / count without the padding. */
int ltu_window_gather(const float* vol, float* win, const int* desc, int n, int H, int W, int D, int h, int w, int d, int ph, int pw,
                      int pd, ltu_stream_t s);
int ltu_vote_accumulate(const float* seg, float* votes, float* count, const int* desc, int n, int Hp, int Wp, int Dp, int h, int w,
                        int d, int C, ltu_stream_t s);
int ltu_vote_finalize(const float* votes, const float* count, float* out, int B, int C, int H, int W, int D, int Hp, int Wp, int Dp,
                      int ph, int pw, int pd, ltu_stream_t s);
/* evaluation metrics of the driver on p = [pred[b][ci] >= threshold] against target u8 [B][H][W][D] (0/1):
 * values[0..3] = DiceClassLoss (criterions.py:35-70), Recall (280-311), Precision (348-379), LocalizationLoss (179-241),
 * means over the batch; rows f32 [B][3][H] scratch (per-h sums of p, t, p*t over W*D = WD elements). */
int ltu_seg_metrics(const float* pred, const uint8_t* target, float* rows, float* values, int B, int C, int ci, int H, long long WD,
                    float threshold, ltu_stream_t s);

/* post-processing of inference_multi_classes.py:104,148-151 = monai KeepLargestConnectedComponent(applied_labels = all foreground
 * channels, independent=False, connectivity=3): pred f32 [C][H][W][D] (rounded one-hot) is modified in place.  Scratch:
 * labels int32 [S], counts int32 [S+1] (zero), best u64 [1] (zero), changed int32 [1];  step 0 = initialise, step 1 = one label
 * propagation sweep (sets *changed), repeat until it stays 0, step 2 = keep the largest 26-connected component of the foreground
 * union, clear the rest, channel 0 = 1 - sum of the other channels. */
int ltu_keep_largest_component(float* pred, int* labels, int* counts, unsigned long long* best, int* changed, int C, int H, int W,
                               int D, int step, ltu_stream_t s);

/* ---- optimizer (train3D.py:193: torch.optim.AdamW(lr=1e-4)) -----------------------------------------------------
 * One AdamW step on flat, 16-byte aligned fp32 buffers (a gradient bucket and the parameters / moments laid out the same way):
 * decoupled weight decay, bias correction with `step` (>= 1), gradient multiplied by grad_scale on load. */
int ltu_adamw(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
              float weight_decay, long long step, float grad_scale, ltu_stream_t s);

/* ---- data-parallel gradient exchange (replaces the reduce half of nn.DataParallel, train3D.py:119) -------------------------------
 * Direct RCCL calls: one communicator per process (= per GPU), created from a 128-byte unique id that rank 0 generates and the
 * host side distributes (lintransunet_amd/comm.py: over the gloo control group).  librccl is not linked: ltu_comm_load dlopens
 * the copy the process already holds (PyTorch's torch/lib/librccl.so, bound to the HIP runtime that owns the caller's streams).
 * The all-reduce only ENQUEUES RCCL's kernel on `s`: no thread, no event polling, no synchronisation - inside a stream capture it
 * becomes a graph node (train.GraphedStep captures every bucket's all-reduce as a side branch of the step graph).
 * `comm` is an opaque handle owned by the caller; the only process-global state are the resolved function pointers.
 *   ltu_comm_allreduce_avg: buf[i] <- mean over ranks of buf[i] (fp32, in place, ncclAvg)
 *   ltu_comm_broadcast:     nbytes of buf from rank `root` to all (initial parameter sync) */
int ltu_comm_load(const char* librccl_path);
int ltu_comm_unique_id(void* id128);
int ltu_comm_init(void** comm, const void* id128, int world, int rank);
int ltu_comm_allreduce_avg(void* comm, float* buf, long long n, ltu_stream_t s);
int ltu_comm_broadcast(void* comm, void* buf, long long nbytes, int root, ltu_stream_t s);
int ltu_comm_destroy(void* comm);

/* ---- data side (dataset/CT_pancreas_ids.py:143-173): raw scan f32 [D][H][W] -> img f32 [H][W][D] = (clamp(raw, lo, hi) - mean) / std,
 * raw label u8 [D][H][W] -> lab u8 [H][W][D] (either pair may be NULL); reference constants lo -91, hi 250, mean 86.9, std 39.4 */
int ltu_ct_preprocess(const float* raw, float* img, const uint8_t* rawlab, uint8_t* lab, int D, int H, int W, float lo, float hi,
                      float mean, float std, ltu_stream_t s);
/* patches out [n][h][w][d] cut from vol [H][W][D] (elem_bytes 4 = f32, 1 = u8) at desc int32 [n][5] = (h0, w0, d0, flip_h, flip_w):
 * the crop of monai RandCropByPosNegLabeld at host-chosen centres followed by RandFlipd over the first two spatial axes */
int ltu_crop_flip(const void* vol, void* out, const int* desc, int n, int H, int W, int D, int h, int w, int d, int elem_bytes,
                  ltu_stream_t s);

/* ---- augmentations of the training dataset (dataset/CT_pancreas_ids.py:112-134; monai 0.7.0 RandRotated, RandAdjustContrastd,
 * RandZoomd) on batches of patches [n][H][W][D] f32 (image and label alike: the reference interpolates both and casts the
 * label back to uint8 at the end).  Random draws stay on the host; a sample that skips a transform gets the identity matrix /
 * zoom 1 / gamma <= 0, all of which reproduce the input exactly.
 * affine: out[k][p] = trilinear(in[k], M_k (p,1)), M_k = mats[k] (3x4 row-major, voxel coordinates), border padding.
 * zoom:   interpolate to zsize[k] = floor(size * zoom_k) per axis (int32 [n][3], computed by the caller in double precision as
 *         torch does; trilinear, align_corners) + centred edge pad / crop back to [H][W][D].
 * contrast: ((x - min)/(max - min + 1e-7))^gamma[k] * (max - min) + min over each patch; minmax_ws: 2 n ints. */
int ltu_affine_sample(const float* in, float* out, const float* mats, int n, int H, int W, int D, ltu_stream_t s);
int ltu_zoom_sample(const float* in, float* out, const int* zsize, int n, int H, int W, int D, ltu_stream_t s);
int ltu_adjust_contrast(const float* in, float* out, const float* gamma, int* minmax_ws, int n, long long per, ltu_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* LTU_HIP_H */
