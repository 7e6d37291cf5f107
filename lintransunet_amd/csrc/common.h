// Shared device helpers for the gfx950 kernels of the LinTransUNet hot path.
// Activations are stored as T in {float, bf16}; all arithmetic is fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/ltu_hip.h"

#define LTU_WAVE 64

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct bf16_t {
  uint16_t bits;
};

__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

// round-to-nearest-even, NaN stays NaN: the gfx950 hardware conversion (v_cvt_pk_bf16_f32)
typedef __bf16 ltu_bf16x2 __attribute__((ext_vector_type(2)));
typedef float ltu_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint16_t f32_to_bf16(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const ltu_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, ltu_bf16x2));
}

// ---- 4-wide vector access (the unit every pointwise kernel works in) ---------------------------
template <typename T>
struct Vec4;

template <>
struct Vec4<float> {
  static __device__ __forceinline__ float4 load(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ void store(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
};

template <>
struct Vec4<bf16_t> {
  static __device__ __forceinline__ float4 load(const bf16_t* p) {
    uint2 r = *reinterpret_cast<const uint2*>(p);
    float4 v;
    v.x = __uint_as_float(r.x << 16);
    v.y = __uint_as_float(r.x & 0xffff0000u);
    v.z = __uint_as_float(r.y << 16);
    v.w = __uint_as_float(r.y & 0xffff0000u);
    return v;
  }
  static __device__ __forceinline__ void store(bf16_t* p, float4 v) {
    uint2 r;
    r.x = pack_bf16x2(v.x, v.y);
    r.y = pack_bf16x2(v.z, v.w);
    *reinterpret_cast<uint2*>(p) = r;
  }
};

template <typename T>
__device__ __forceinline__ float ld1(const T* p);
template <>
__device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ld1<bf16_t>(const bf16_t* p) { return bf16_to_f32(p->bits); }

template <typename T>
__device__ __forceinline__ void st1(T* p, float v);
template <>
__device__ __forceinline__ void st1<float>(float* p, float v) { *p = v; }
template <>
__device__ __forceinline__ void st1<bf16_t>(bf16_t* p, float v) { p->bits = f32_to_bf16(v); }

__device__ __forceinline__ float& f4at(float4& v, int i) { return reinterpret_cast<float*>(&v)[i]; }
__device__ __forceinline__ float f4at(const float4& v, int i) { return reinterpret_cast<const float*>(&v)[i]; }

// ---- wave / block reductions (64-wide wavefronts) ----------------------------------------------
// All-lanes ("butterfly") reductions without the LDS crossbar: four DPP steps give every lane of a 16-lane row the row's value
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), v_permlane16_swap / v_permlane32_swap (gfx950) exchange rows and wave
// halves - with both operands a copy of v, a + b (or max(a, b)) is the combined value in EVERY lane, no select needed.
// `__shfl_xor` compiles to ds_bpermute_b32: an LDS round trip (~100 cycles) per step, and the steps of a reduction are
// dependent: two LayerNorm reductions over 64 lanes cost ~1 300 cycles per row pass and made up a third of the chain kernels.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
struct LtuAdd { static __device__ __forceinline__ float op(float a, float b) { return a + b; } };
struct LtuMax { static __device__ __forceinline__ float op(float a, float b) { return fmaxf(a, b); } };
// combine with the other half of the wave (lane ^ 32) / the neighbouring 16-lane row (lane ^ 16)
// (inline asm: the builtins __builtin_amdgcn_permlane16_swap / 32_swap of hipcc 7.2 return the FIRST result for both elements of
// their pair - `v_permlane16_swap v1, v2` followed by two uses of v1 -; the s_nop pairs cover the VALU -> swap -> VALU wait
// states the compiler would otherwise insert)
__device__ __forceinline__ void permlane16_swap(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void permlane32_swap(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
template <typename OP>
__device__ __forceinline__ float xhalf_combine(float v) {
  float a = v, b = v;
  permlane32_swap(a, b);
  return OP::op(a, b);
}
template <typename OP>
__device__ __forceinline__ float xrow_combine(float v) {
  float a = v, b = v;
  permlane16_swap(a, b);
  return OP::op(a, b);
}
// reduce over aligned groups of G lanes (G power of two <= 64); every lane of the group receives the result
template <int G, typename OP>
__device__ __forceinline__ float group_reduce(float v) {
  if (G >= 2) v = OP::op(v, dpp_mov<0xB1>(v));       // quad_perm [1,0,3,2]
  if (G >= 4) v = OP::op(v, dpp_mov<0x4E>(v));       // quad_perm [2,3,0,1]
  if (G >= 8) v = OP::op(v, dpp_mov<0x141>(v));      // row_half_mirror
  if (G >= 16) v = OP::op(v, dpp_mov<0x140>(v));     // row_mirror
  if (G >= 32) v = xrow_combine<OP>(v);
  if (G >= 64) v = xhalf_combine<OP>(v);
  return v;
}
template <int G>
__device__ __forceinline__ float group_sum(float v) { return group_reduce<G, LtuAdd>(v); }
__device__ __forceinline__ float wave_sum(float v) { return group_reduce<64, LtuAdd>(v); }
__device__ __forceinline__ float wave_max(float v) { return group_reduce<64, LtuMax>(v); }

// ---- counter-based dropout RNG -----------------------------------------------------------------
// Stateless: the keep bits of the 4-element group g4 are a hash of (seed, step counter, g4), so the backward kernels
// regenerate the forward mask instead of storing it.  One murmur3 finalizer + a linear expansion give 64 bits per group, 16 per
// element (keep iff bits >= p * 65536).  A first version ran Philox4x32-7 per group: its 28 quarter-rate 32-bit multiplies made
// every dropout site VALU-bound (GELU + dropout ran at 37 % of the HBM rate); this mixer needs 4.
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu;
  h ^= h >> 13; h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}

struct DropCfg {       // p == 0 disables
  float p;
  float scale;         // 1/(1-p)
  uint32_t thresh;     // keep iff 16 random bits >= thresh, thresh = p * 2^16
  uint32_t ka;         // per-site key
};

// `step` (nullable) points at a device-resident step counter mixed into the seed, so that a HIP-graph replay
// of a captured step (whose scalar arguments are frozen) still draws fresh masks.
__device__ __forceinline__ DropCfg make_drop(float p, uint64_t seed, const uint64_t* step) {
  DropCfg d;
  if (step != nullptr && p > 0.f) seed += *step * 0x9E3779B97F4A7C15ull;
  d.p = p;
  d.scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
  d.thresh = p > 0.f ? (uint32_t)fminf(p * 65536.f + 0.5f, 65535.f) : 0u;
  const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
  d.ka = fmix32(lo ^ 0x243F6A88u) + fmix32(hi ^ 0x85A308D3u) * 0x9E3779B1u;
  return d;
}

// multiplies the 4 values of group `g4` (a 64-bit element index / 4) by their keep-mask * scale
__device__ __forceinline__ float4 drop4(const DropCfg& d, uint64_t g4, float4 v) {
  if (d.p <= 0.f) return v;
  const uint32_t c1 = (uint32_t)(g4 >> 32);
  const uint32_t h1 = fmix32(((uint32_t)g4 ^ d.ka) + ((c1 << 16) | (c1 >> 16)));
  // second word: a linear expansion of the first (rotations chosen so that every pair of the four 16-bit words is a full-rank
  // map of h1, i.e. any two elements of a group are independent; tools checked keep rates, pair / lag correlations and the
  // conditional rates P(z | x, y), P(w | x, y, z) against a second murmur finalizer: indistinguishable at 4 M groups).  The second
  // finalizer cost two more quarter-rate multiplies per group: the masks were 0.24 ms of the training step.
  // LIMIT: pairwise independent only - z and w are deterministic functions of (x, y), so the four decisions of a group are not
  // 4-wise independent as nn.Dropout's are (different groups are).  The joint 16-pattern histogram of a group is held against
  // Bernoulli^4 by tests/test_gpu_ops.py::test_dropout_group_pattern_histogram, so a change of the rotations cannot silently bias
  // the per-group keep counts.
  const uint32_t h2 = h1 ^ ((h1 << 11) | (h1 >> 21)) ^ ((h1 << 19) | (h1 >> 13));
  v.x = (h1 & 0xFFFFu) >= d.thresh ? v.x * d.scale : 0.f;
  v.y = (h1 >> 16) >= d.thresh ? v.y * d.scale : 0.f;
  v.z = (h2 & 0xFFFFu) >= d.thresh ? v.z * d.scale : 0.f;
  v.w = (h2 >> 16) >= d.thresh ? v.w * d.scale : 0.f;
  return v;
}
// keep-mask*scale factors only (used by backward kernels)
__device__ __forceinline__ float4 dropmask4(const DropCfg& d, uint64_t g4) {
  return drop4(d, g4, make_float4(1.f, 1.f, 1.f, 1.f));
}

// ---- misc ---------------------------------------------------------------------------------------
// GELU(x) = x Phi(x) with Phi through the Abramowitz-Stegun 7.1.26 form of erfc (|error| < 1.5e-7 on erf): one v_rcp, one
// v_exp and 5 fmas, branch-free; libm's erff is ~70 instructions with divergent ranges and made the GELU kernels VALU-bound.
// The negative side uses erfc directly (no 1 - erf cancellation); exp(-x^2/2) is shared with the density in the derivative.
__device__ __forceinline__ float gelu_cdf(float x, float& e) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));      // raw v_rcp_f32 (1 ulp); __frcp_rn expands to a full division
  e = __expf(-z * z);
  const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
  const float half_erfc = 0.5f * poly * e;
  return x >= 0.f ? 1.f - half_erfc : half_erfc;
}
// two elements at a time: the multiplies and fmas become v_pk_mul_f32 / v_pk_fma_f32 (full rate on gfx950: half the issue slots of
// the scalar form; v_rcp / v_exp stay per element).  Same operations in the same order as gelu_cdf: bit-identical results.
typedef float ltu_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ltu_f2 gelu_cdf2(ltu_f2 x, ltu_f2& e) {
  const ltu_f2 z = __builtin_elementwise_abs(x) * 0.70710678118654752440f;
  const ltu_f2 a = 0.3275911f * z + 1.f;
  ltu_f2 t;
  t.x = __builtin_amdgcn_rcpf(a.x); t.y = __builtin_amdgcn_rcpf(a.y);
  const ltu_f2 mz = -z * z;
  e.x = __expf(mz.x); e.y = __expf(mz.y);
  const ltu_f2 poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
  const ltu_f2 half_erfc = 0.5f * poly * e;
  ltu_f2 r;
  r.x = x.x >= 0.f ? 1.f - half_erfc.x : half_erfc.x;
  r.y = x.y >= 0.f ? 1.f - half_erfc.y : half_erfc.y;
  return r;
}
__device__ __forceinline__ float gelu_erf(float x) {
  float e;
  return x * gelu_cdf(x, e);
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  float e;
  const float cdf = gelu_cdf(x, e);
  return cdf + x * 0.39894228040143267794f * e;
}

// ---- host-side configuration ---------------------------------------------------------------------
// Tuning / ablation knobs.  Resolution order: ltu_config_set() override, then the environment variable of the same name, then
// the built-in default.  Nothing is cached per call site, so the library holds no hidden per-process state besides the
// override table (mutex-protected, misc.hip); geometry therefore follows a changed knob from the next launch on.
int ltu_knob(const char* name, int dflt);
static inline int ltu_knob_pos(const char* name, int dflt) {
  const int v = ltu_knob(name, dflt);
  return v > 0 ? v : dflt;
}
// "first call on this device": kernel attributes (dynamic LDS limits) are per device, so latches are a device bit mask
#ifdef __cplusplus
#include <atomic>
struct LtuDevOnce {
  std::atomic<unsigned long long> mask{0};
  bool first() {
    int d = 0;
    (void)hipGetDevice(&d);
    const unsigned long long bit = 1ull << (d & 63);
    return !(mask.fetch_or(bit) & bit);
  }
};
#endif

static inline int ltu_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? LTU_OK : (int)e;
}

static inline unsigned cdiv(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

// dtype dispatch for host wrappers
#define LTU_DISPATCH_T(dtype, ...)                           \
  do {                                                       \
    if ((dtype) == LTU_F32) {                                \
      typedef float T;                                       \
      __VA_ARGS__                                            \
    } else if ((dtype) == LTU_BF16) {                        \
      typedef bf16_t T;                                      \
      __VA_ARGS__                                            \
    } else {                                                 \
      return LTU_E_DTYPE;                                    \
    }                                                        \
  } while (0)
