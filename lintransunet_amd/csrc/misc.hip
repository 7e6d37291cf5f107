// Library version + the batched weight-operand preparation (one launch per step for the whole model).
#include <math.h>

#include "common.h"

#include <mutex>
#include <stdlib.h>

extern "C" int ltu_version(void) { return 4; }
extern "C" int ltu_build_flags(void) {
#ifdef LTU_EXPERIMENTS
  return 1;
#else
  return 0;
#endif
}

// ---- self-test of the cross-lane reductions (common.h) ----------------------------------------------------------------
template <int G>
__global__ void selftest_reduce_kernel(const float* __restrict__ x, float* __restrict__ sum, float* __restrict__ mx, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const float v = i < n ? x[i] : 0.f;          // n is a multiple of 64: whole waves only
  const float s = group_reduce<G, LtuAdd>(v), m = group_reduce<G, LtuMax>(v);
  if (i < n) { sum[i] = s; mx[i] = m; }
}
extern "C" int ltu_selftest_group_reduce(const float* x, float* sum, float* mx, int n, int G, ltu_stream_t s) {
  if (n <= 0 || n % 64) return LTU_E_SHAPE;
  const dim3 grid((unsigned)(n / 64)), block(64);
  hipStream_t st = (hipStream_t)s;
  switch (G) {
    case 2: hipLaunchKernelGGL(selftest_reduce_kernel<2>, grid, block, 0, st, x, sum, mx, n); break;
    case 4: hipLaunchKernelGGL(selftest_reduce_kernel<4>, grid, block, 0, st, x, sum, mx, n); break;
    case 8: hipLaunchKernelGGL(selftest_reduce_kernel<8>, grid, block, 0, st, x, sum, mx, n); break;
    case 16: hipLaunchKernelGGL(selftest_reduce_kernel<16>, grid, block, 0, st, x, sum, mx, n); break;
    case 32: hipLaunchKernelGGL(selftest_reduce_kernel<32>, grid, block, 0, st, x, sum, mx, n); break;
    case 64: hipLaunchKernelGGL(selftest_reduce_kernel<64>, grid, block, 0, st, x, sum, mx, n); break;
    default: return LTU_E_ARG;
  }
  return ltu_check_launch();
}

#ifdef LTU_EXPERIMENTS      // the in-launch last-arriver fold (DESIGN.md finding 15): slower than the second launch; experiments build only
// ---- last-arriver fold: self-test and price of the in-launch second stage (finding 15) -----------------------
// The step folds ~100 sets of per-workgroup partial sums with a second small launch each.  The alternative keeps the fold inside
// the producing launch: every workgroup publishes its partial row, takes a ticket, and the workgroup whose ticket is the last
// one folds all rows.  The hand-off follows the guide's recipe R1 (cdna_hip_programming.md, Guideline 16): payload stored
// write-through (sc1), every storing wave drains its stores (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane adds to the
// agent-scope counter; the last arriver issues ONE agent-scope acquire (its L1 may hold stale copies of the partial lines),
// waits for the invalidate, meets its other waves at a barrier, then reads with plain loads in a FIXED order (bit-reproducible).
// The self-test makes the consumer's L1 warm on purpose (every workgroup pre-reads the partial lines of the previous launch
// before it works) and the load uneven (workgroup i sums 1 + (7 i mod skew) row chunks).  mode 0: two-stage reference (this
// launch writes partials, ltu_selftest_fold sums them); mode 1: last-arriver fold inside the launch.
typedef __attribute__((address_space(1))) unsigned int gu32_t;
__global__ void __launch_bounds__(256) selftest_partials_kernel(const float* __restrict__ x, float* __restrict__ part, float* __restrict__ out,
                                                                unsigned* counter, float* __restrict__ sink, int n, int rows_per_chunk,
                                                                int skew, int mode) {
  __shared__ int s_last;
  const int tid = threadIdx.x, wg = blockIdx.x, nwg = gridDim.x;
  // L1-warm consumer: plain loads of the partial rows as the PREVIOUS launch left them (every workgroup: any of them may be last)
  float warm = 0.f;
  for (int i = tid; i < nwg * n; i += 256 * 8) warm += part[i];
  // uneven work: chunks [c0, c0 + cnt) of x, column t summed by thread t
  const int cnt = 1 + (7 * wg) % skew;
  long long c0 = 0;
  for (int i = 0; i < wg; ++i) c0 += 1 + (7 * i) % skew;
  float acc = 0.f;
  if (tid < n)
    for (long long r = c0 * rows_per_chunk; r < (c0 + cnt) * rows_per_chunk; ++r) acc += x[r * n + tid];
  if (warm == 12345.678f) sink[wg] = warm;                   // keeps the pre-reads alive, never true for the test data
  if (mode == 0) {
    if (tid < n) part[(long long)wg * n + tid] = acc;
    return;
  }
  if (tid < n) __hip_atomic_store(part + (long long)wg * n + tid, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // sc1
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every storing wave drains
  __syncthreads();
  if (tid == 0) {
    const unsigned ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = ticket == (unsigned)nwg - 1;
    if (s_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // drops this CU's stale L1 lines
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // holds the barrier below until the invalidate has completed
    }
  }
  __syncthreads();
  if (!s_last) return;
  if (tid < n) {
    float v = 0.f;
    for (int i = 0; i < nwg; ++i) v += part[(long long)i * n + tid];      // fixed order: the same bits as the two-stage fold
    out[tid] = v;
  }
  if (tid == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // ready for the next launch
}
__global__ void __launch_bounds__(256) selftest_fold_kernel(const float* __restrict__ part, float* __restrict__ out, int nwg, int n) {
  const int tid = threadIdx.x;
  if (tid >= n) return;
  float v = 0.f;
  for (int i = 0; i < nwg; ++i) v += part[(long long)i * n + tid];
  out[tid] = v;
}
// x: [sum_i (1 + 7 i mod skew)] * rows_per_chunk rows of n <= 256 floats; part: nwg * n floats; counter: one zeroed word (mode 1)
extern "C" int ltu_selftest_last_arriver(const float* x, float* part, float* out, unsigned* counter, float* sink, int nwg, int n,
                                         int rows_per_chunk, int skew, int mode, ltu_stream_t s) {
  if (nwg <= 0 || n <= 0 || n > 256 || rows_per_chunk <= 0 || skew <= 0 || (mode != 0 && mode != 1)) return LTU_E_ARG;
  hipStream_t st = (hipStream_t)s;
  hipLaunchKernelGGL(selftest_partials_kernel, dim3(nwg), dim3(256), 0, st, x, part, out, counter, sink, n, rows_per_chunk, skew, mode);
  if (mode == 0) hipLaunchKernelGGL(selftest_fold_kernel, dim3(1), dim3(256), 0, st, part, out, nwg, n);
  return ltu_check_launch();
}
#endif

// ---- knob overrides (ltu_config_set): a small table under a mutex; see common.h ------------------------------------------
namespace {
struct KnobOverride { char name[40]; int value; };
std::mutex g_knob_mu;
KnobOverride g_knob[32];
int g_nknob = 0;
}  // namespace

extern "C" int ltu_config_set(const char* name, int value, int clear) {
  if (name == nullptr || strlen(name) >= sizeof(g_knob[0].name)) return LTU_E_ARG;
  std::lock_guard<std::mutex> lock(g_knob_mu);
  for (int i = 0; i < g_nknob; ++i)
    if (strcmp(g_knob[i].name, name) == 0) {
      if (clear) { g_knob[i] = g_knob[g_nknob - 1]; --g_nknob; }
      else g_knob[i].value = value;
      return LTU_OK;
    }
  if (clear) return LTU_OK;
  if (g_nknob == 32) return LTU_E_ARG;
  strcpy(g_knob[g_nknob].name, name);
  g_knob[g_nknob++].value = value;
  return LTU_OK;
}

int ltu_knob(const char* name, int dflt) {
  {
    std::lock_guard<std::mutex> lock(g_knob_mu);
    for (int i = 0; i < g_nknob; ++i)
      if (strcmp(g_knob[i].name, name) == 0) return g_knob[i].value;
  }
  const char* e = getenv(name);
  return (e != nullptr && *e != 0) ? atoi(e) : dflt;
}

// One descriptor per prepared operand.  kinds:
//   0  cast        dst[i] = src[i]                                  R*C elements
//   1  transpose   dst[c*p0 + p1 + r] = src[r*C + c]                src [R][C]  (p0 = ld of dst, p1 = column offset)
//   2  pack_f      conv [R=Co][C=Ci][27] -> dst [p0=CoP][27][p1=CiP] zero padded
//   3  pack_d      conv [R=Co][C=Ci][27] -> dst [p1=CiP][27][p0=CoP] zero padded
//   4  copy_f32    dst[i] = src[i] as fp32 whatever the operand dtype (padded biases)      R*C elements
//   5  upconv_f    conv [R=Co][C=Ci][27] -> sub-pixel forward operand  dst [8 classes][p0=CoP][8 slots][p1=CiP]
//   6  upconv_d    conv [R=Co][C=Ci][27] -> sub-pixel data-gradient operand  dst [p1=CiP][64][p0=CoP]   (see upconv.hip)
struct WPrep {
  const float* src;
  void* dst;
  int kind, R, C, p0, p1, pad;
};

// Destination elements [t0, t1) of record d, `stride` apart per thread.  All index arithmetic is 32-bit (a record has < 2^31
// destination elements): the 64-bit divisions of a first version cost more than the memory traffic.
template <typename TW>
__device__ __forceinline__ void weight_prep_range(const WPrep& d, unsigned t0, unsigned t1, unsigned stride) {
  TW* dst = reinterpret_cast<TW*>(d.dst);
  if (d.kind == 0) {
    const unsigned n = (unsigned)d.R * (unsigned)d.C;
    if (t1 > n) t1 = n;
    for (unsigned i = t0; i < t1; i += stride) st1<TW>(dst + i, d.src[i]);
  } else if (d.kind == 4) {
    const unsigned n = (unsigned)d.R * (unsigned)d.C;
    if (t1 > n) t1 = n;
    for (unsigned i = t0; i < t1; i += stride) reinterpret_cast<float*>(d.dst)[i] = d.src[i];
  } else if (d.kind == 1) {
    const unsigned n = (unsigned)d.R * (unsigned)d.C, R = (unsigned)d.R, C = (unsigned)d.C;
    if (t1 > n) t1 = n;
    // i enumerates the DESTINATION (coalesced writes; sources are L2-resident); (r, c) advance incrementally
    unsigned r = t0 % R, c = t0 / R;
    const unsigned dr = stride % R, dc = stride / R;
    for (unsigned i = t0; i < t1; i += stride) {
      st1<TW>(dst + (size_t)c * d.p0 + d.p1 + r, d.src[(size_t)r * C + c]);
      r += dr; c += dc;
      if (r >= R) { r -= R; ++c; }
    }
  } else if (d.kind == 5 || d.kind == 6) {
    const unsigned CoP = (unsigned)d.p0, CiP = (unsigned)d.p1;
    const unsigned n = 64u * CoP * CiP;
    if (t1 > n) t1 = n;
    for (unsigned i = t0; i < t1; i += stride) {
      unsigned cls, sl, co, ci;
      if (d.kind == 5) { ci = i % CiP; const unsigned q = i / CiP; sl = q & 7; co = (q >> 3) % CoP; cls = (q >> 3) / CoP; }
      else { co = i % CoP; const unsigned q = i / CoP; const unsigned cs = q & 63; ci = q >> 6; cls = cs >> 3; sl = cs & 7; }
      float v = 0.f;
      if (co < (unsigned)d.R && ci < (unsigned)d.C) {
        // per axis: class p, slot a -> taps {0} | {1,2} (p = 0) or {0,1} | {2} (p = 1)
        int lo[3], hi[3];
        for (int a = 0; a < 3; ++a) {
          const int p = (cls >> (2 - a)) & 1, sa = (sl >> (2 - a)) & 1;
          if (p == 0) { lo[a] = sa == 0 ? 0 : 1; hi[a] = sa == 0 ? 0 : 2; }
          else { lo[a] = sa == 0 ? 0 : 2; hi[a] = sa == 0 ? 1 : 2; }
        }
        const float* w = d.src + ((size_t)co * d.C + ci) * 27;
        for (int th = lo[0]; th <= hi[0]; ++th)
          for (int tw = lo[1]; tw <= hi[1]; ++tw)
            for (int td = lo[2]; td <= hi[2]; ++td) v += w[(th * 3 + tw) * 3 + td];
      }
      st1<TW>(dst + i, v);
    }
  } else if (d.kind == 8 || d.kind == 9) {
    // MFMA fragment order of a dense weight for the row-block chain kernels (tlayer.hip): the 16 bytes lane l of a wave feeds to
    // v_mfma_f32_32x32x16_bf16 for output tile ct (32 outputs) and reduction step ks (16 terms) sit at ((ct * KS + ks) * 64 + l) * 8,
    // so a wave's operand is one coalesced 1 KiB load.  kind 8: outputs = rows of src [R][C] (forward: y = x W^T);
    // kind 9: outputs = columns of src (data gradient: dx = g W), reduction over its rows.
    const unsigned n = (unsigned)d.R * (unsigned)d.C;
    if (t1 > n) t1 = n;
    const unsigned KS = (d.kind == 8 ? (unsigned)d.C : (unsigned)d.R) >> 4;
    for (unsigned i = t0; i < t1; i += stride) {
      const unsigned j = i & 7u, l = (i >> 3) & 63u, t = i >> 9;
      const unsigned ks = t % KS, ct = t / KS;
      const unsigned o = ct * 32u + (l & 31u), r = ks * 16u + 8u * (l >> 5) + j;
      const float v = d.kind == 8 ? d.src[(size_t)o * d.C + r] : d.src[(size_t)r * d.C + o];
      st1<TW>(dst + i, v);
    }
  } else if (d.kind == 7) {
    // pack wd of one member of a fused conv group: columns [off, off+cnt) of dst [CiP][27][stride]; pad = off << 16 | cnt
    const unsigned stride_c = (unsigned)d.p0, CiP = (unsigned)d.p1, off = (unsigned)d.pad >> 16, cnt = (unsigned)d.pad & 0xffffu;
    const unsigned n = cnt * 27u * CiP;
    if (t1 > n) t1 = n;
    for (unsigned i = t0; i < t1; i += stride) {
      const unsigned co = i % cnt, q = i / cnt, t = q % 27u, ci = q / 27u;
      const float v = (co < (unsigned)d.R && ci < (unsigned)d.C) ? d.src[((size_t)co * d.C + ci) * 27 + t] : 0.f;
      st1<TW>(dst + ((size_t)ci * 27 + t) * stride_c + off + co, v);
    }
  } else {
    const unsigned CoP = (unsigned)d.p0, CiP = (unsigned)d.p1;
    const unsigned n = CoP * 27u * CiP;
    if (t1 > n) t1 = n;
    // destination digits (fast .. slow): kind 2 (ci, t, co), kind 3 (co, t, ci); advanced incrementally
    const unsigned R0 = d.kind == 2 ? CiP : CoP;
    unsigned x0 = t0 % R0, q = t0 / R0, x1 = q % 27u, x2 = q / 27u;
    const unsigned s0 = stride % R0, sq = stride / R0, s1 = sq % 27u, s2 = sq / 27u;
    for (unsigned i = t0; i < t1; i += stride) {
      const unsigned co = d.kind == 2 ? x2 : x0, ci = d.kind == 2 ? x0 : x2;
      const float v = (co < (unsigned)d.R && ci < (unsigned)d.C) ? d.src[((size_t)co * d.C + ci) * 27 + x1] : 0.f;
      st1<TW>(dst + i, v);
      x0 += s0; x1 += s1; x2 += s2;
      if (x0 >= R0) { x0 -= R0; ++x1; }
      if (x1 >= 27u) { x1 -= 27u; ++x2; }
    }
  }
}

// grid (256, n): every record spread over 256 workgroups (fine for a handful of records)
template <typename TW>
__global__ void weight_prep_kernel(const WPrep* __restrict__ table) {
  const WPrep d = table[blockIdx.y];
  weight_prep_range<TW>(d, blockIdx.x * blockDim.x + threadIdx.x, 0x7fffffffu, gridDim.x * blockDim.x);
}

// grid (nchunks): chunk c = {record, first destination element / LTU_WPREP_CHUNK}.  A model has hundreds of records of
// very different sizes; a fixed grid per record costs more in workgroup dispatch than in memory traffic.
// 8 consecutive DESTINATION elements per thread for the dense-weight kinds (0 copy, 1 transpose, 8 / 9 fragment order): the eight
// source values are requested together and leave as ONE 16-byte (bf16) store.  The element-at-a-time loop (a 4-byte load and a
// 2-byte store per iteration) ran the once-per-step preparation of 84 MB of masters at ~1.3 TB/s.
__device__ __forceinline__ bool wp8_ok(const WPrep& d, int tw_size) {
  if (tw_size != 2 || !(d.kind == 0 || d.kind == 1 || d.kind == 8 || d.kind == 9)) return false;
  const unsigned n = (unsigned)d.R * (unsigned)d.C, R = (unsigned)d.R;
  if (n % 8 || (reinterpret_cast<uintptr_t>(d.dst) & 15)) return false;
  if (d.kind == 1 && (R % 8 || ((d.p0 | d.p1) % 8))) return false;
  return true;
}
// the eight source values of destination elements i .. i + 7 of record d (i a multiple of 8); returns the destination offset of f[0]
__device__ __forceinline__ size_t wp8_load(const WPrep& d, unsigned i, float (&f)[8]) {
  const unsigned R = (unsigned)d.R, C = (unsigned)d.C;
  const bool src16 = (reinterpret_cast<uintptr_t>(d.src) & 15) == 0;      // masters inside a flat gradient bucket start anywhere
  size_t o = i;
  if (d.kind == 0) {
    if (src16) {
      const float4 a = *reinterpret_cast<const float4*>(d.src + i), b = *reinterpret_cast<const float4*>(d.src + i + 4);
      f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = d.src[i + j];
    }
  } else if (d.kind == 1) {                            // dst[c * p0 + p1 + r] = src[r * C + c], i = c * R + r, 8 consecutive r
    const unsigned c = i / R, r = i - c * R;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = d.src[(size_t)(r + j) * C + c];
    o = (size_t)c * d.p0 + d.p1 + r;
  } else {                                             // fragment order: i = ((ct * KS + ks) * 64 + l) * 8 + j
    const unsigned KS = (d.kind == 8 ? C : R) >> 4;
    const unsigned l = (i >> 3) & 63u, t = i >> 9;
    const unsigned ks = t % KS, ct = t / KS;
    const unsigned oo = ct * 32u + (l & 31u), r = ks * 16u + 8u * (l >> 5);
    if (d.kind == 8 && src16 && C % 4 == 0) {
      const float4 a = *reinterpret_cast<const float4*>(d.src + (size_t)oo * C + r), b = *reinterpret_cast<const float4*>(d.src + (size_t)oo * C + r + 4);
      f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    } else if (d.kind == 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = d.src[(size_t)oo * C + r + j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = d.src[(size_t)(r + j) * C + oo];
    }
  }
  return o;
}

// A workgroup takes WP_NC chunks per trip and requests all their source values (2 x WP_NC vectors of eight per thread) before
// its first store.  WP_NC = 4 for the long lists only: the ROI bridges' operands (9 000 chunks) are refreshed on a side stream beside
// the bottleneck transformer, whose 5-25 us kernels were held back by 110 us behind one-chunk workgroups and by 75 us behind
// four-chunk ones (the refresh itself 130 -> 152 us, off the chain); a short list (the encoder's 430 chunks, on the chain) would
// no longer fill the machine: 14 -> 33 us.  gridDim.x may be smaller than the number of trips (LTU_WPREP_BLOCKS; narrowing the side
// launches to 32 / 64 / 128 workgroups made them the critical path: +1.0 / +0.25 / +0.1 ms per step).  The step time itself is
// the same with either trip size (interleaved A/B, +-0.03 ms).
template <typename TW, int WP_NC>
__global__ void __launch_bounds__(256) weight_prep_chunk_kernel(const WPrep* __restrict__ table, const int2* __restrict__ chunks, int nchunks) {
  constexpr int PER = LTU_WPREP_CHUNK / 8 / 256;            // vectors of eight per thread and chunk
  static_assert(PER * 256 * 8 == LTU_WPREP_CHUNK, "chunk size");
  for (int c0 = blockIdx.x * WP_NC; c0 < nchunks; c0 += gridDim.x * WP_NC) {
    float f[WP_NC][PER][8];
    uint16_t* dp[WP_NC][PER];
#pragma unroll
    for (int k = 0; k < WP_NC; ++k) {
#pragma unroll
      for (int u = 0; u < PER; ++u) dp[k][u] = nullptr;
      if (c0 + k >= nchunks) continue;
      const int2 c = chunks[c0 + k];
      const WPrep d = table[c.x];
      const unsigned t0 = (unsigned)c.y * LTU_WPREP_CHUNK;
      if (!wp8_ok(d, (int)sizeof(TW))) {
        weight_prep_range<TW>(d, t0 + threadIdx.x, t0 + LTU_WPREP_CHUNK, blockDim.x);
        continue;
      }
      const unsigned n = (unsigned)d.R * (unsigned)d.C;
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const unsigned i = t0 + (threadIdx.x + u * 256u) * 8u;
        if (i < n) dp[k][u] = reinterpret_cast<uint16_t*>(d.dst) + wp8_load(d, i, f[k][u]);
      }
    }
#pragma unroll
    for (int k = 0; k < WP_NC; ++k)
#pragma unroll
      for (int u = 0; u < PER; ++u)
        if (dp[k][u] != nullptr)
          *reinterpret_cast<uint4*>(dp[k][u]) = make_uint4(pack_bf16x2(f[k][u][0], f[k][u][1]), pack_bf16x2(f[k][u][2], f[k][u][3]),
                                                           pack_bf16x2(f[k][u][4], f[k][u][5]), pack_bf16x2(f[k][u][6], f[k][u][7]));
  }
}

extern "C" int ltu_weight_prep(const void* table, int n, int out_dtype, ltu_stream_t s) {
  if (n <= 0) return LTU_OK;
  LTU_DISPATCH_T(out_dtype, {
    hipLaunchKernelGGL((weight_prep_kernel<T>), dim3(256, n), dim3(256), 0, (hipStream_t)s, (const WPrep*)table);
  });
  return ltu_check_launch();
}

extern "C" int ltu_weight_prep_chunks(const void* table, const int* chunks, int nchunks, int out_dtype, ltu_stream_t s) {
  if (nchunks <= 0) return LTU_OK;
  LTU_DISPATCH_T(out_dtype, {
    const int nc = nchunks >= ltu_knob_pos("LTU_WPREP_NC4_MIN", 6000) ? 4 : 1;
    const int trips = (nchunks + nc - 1) / nc;
    int blocks = ltu_knob_pos("LTU_WPREP_BLOCKS", trips);
    if (blocks > trips) blocks = trips;
    if (nc == 4)
      hipLaunchKernelGGL((weight_prep_chunk_kernel<T, 4>), dim3(blocks), dim3(256), 0, (hipStream_t)s, (const WPrep*)table,
                         (const int2*)chunks, nchunks);
    else
      hipLaunchKernelGGL((weight_prep_chunk_kernel<T, 1>), dim3(blocks), dim3(256), 0, (hipStream_t)s, (const WPrep*)table,
                         (const int2*)chunks, nchunks);
  });
  return ltu_check_launch();
}

// ------------------------------------------------------------------------------------------------ optimizer
// AdamW step of train3D.py:193 (torch.optim.AdamW, decoupled weight decay, bias-corrected moments) on flat fp32 buffers:
//   p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
// `gscale` multiplies the gradient on load (1 / accumulation count or loss-scale inverse).  One launch per bucket.
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             long long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2, float gscale) {
  const long long nv = n >> 2;
  const float step_size = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2), decay = 1.f - lr * wd;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
    float4 pv = reinterpret_cast<float4*>(p)[i], mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gg = f4at(gv, k) * gscale;
      const float mm = b1 * f4at(mv, k) + (1.f - b1) * gg;
      const float v2 = b2 * f4at(vv, k) + (1.f - b2) * gg * gg;
      f4at(mv, k) = mm; f4at(vv, k) = v2;
      f4at(pv, k) = f4at(pv, k) * decay - step_size * mm / (sqrtf(v2) * inv_sqrt_bc2 + eps);
    }
    reinterpret_cast<float4*>(p)[i] = pv; reinterpret_cast<float4*>(m)[i] = mv; reinterpret_cast<float4*>(v)[i] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {        // tail
    const long long i = (nv << 2) + threadIdx.x;
    const float gg = g[i] * gscale;
    const float mm = b1 * m[i] + (1.f - b1) * gg;
    const float v2 = b2 * v[i] + (1.f - b2) * gg * gg;
    m[i] = mm; v[i] = v2;
    p[i] = p[i] * decay - step_size * mm / (sqrtf(v2) * inv_sqrt_bc2 + eps);
  }
}

extern "C" int ltu_adamw(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                         float weight_decay, long long step, float grad_scale, ltu_stream_t s) {
  if (n <= 0) return LTU_OK;
  if (step < 1) return LTU_E_ARG;
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return LTU_E_ARG;
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  long long blocks = ((n >> 2) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, p, g, m, v, n, lr, beta1, beta2, eps,
                     weight_decay, bc1, bc2, grad_scale);
  return ltu_check_launch();
}
