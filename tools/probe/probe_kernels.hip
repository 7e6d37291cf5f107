// Probe kernels for tools/bench_starve.py: a workgroup that holds `lds` bytes of LDS and spins for `ticks` of the 100 MHz realtime counter,
// and a short kernel of many small workgroups.  Built by tools/probe/build.sh into tools/probe/libprobe.so (not part of the product).
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" __global__ void __launch_bounds__(256) spin_kernel(uint64_t ticks, float* out) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (out != nullptr && threadIdx.x == 0 && blockIdx.x == 0) out[0] = lds[1];
}

extern "C" __global__ void __launch_bounds__(256) short_kernel(float* x, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] += 1.f;
}

// a workgroup that holds `lds` bytes of LDS and streams `bytes_per_wg` from global memory with 16-byte loads (8 in flight per lane)
extern "C" __global__ void __launch_bounds__(256) stream_kernel(const uint4* __restrict__ src, long long vec_per_wg, long long nvec, float* out) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = 0.f;
  __syncthreads();
  const long long base = ((long long)blockIdx.x * vec_per_wg) % nvec;
  uint4 acc = make_uint4(0u, 0u, 0u, 0u);
  for (long long i = threadIdx.x; i + 7 * 256 < vec_per_wg; i += 8 * 256) {
    uint4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = src[(base + i + k * 256) % nvec];
#pragma unroll
    for (int k = 0; k < 8; ++k) { acc.x ^= v[k].x; acc.y ^= v[k].y; acc.z ^= v[k].z; acc.w ^= v[k].w; }
  }
  if (out != nullptr && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[1] = lds[1];
}

extern "C" int probe_stream(int wgs, int lds, const void* src, long long vec_per_wg, long long nvec, float* out, void* stream) {
  (void)hipFuncSetAttribute((const void*)stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(stream_kernel, dim3(wgs), dim3(256), (size_t)lds, (hipStream_t)stream, (const uint4*)src, vec_per_wg, nvec, out);
  return (int)hipGetLastError();
}

extern "C" int probe_spin(int wgs, int lds, unsigned long long ticks, float* out, void* stream) {
  (void)hipFuncSetAttribute((const void*)spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(256), (size_t)lds, (hipStream_t)stream, (uint64_t)ticks, out);
  return (int)hipGetLastError();
}

extern "C" int probe_short(float* x, int n, void* stream) {
  hipLaunchKernelGGL(short_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, n);
  return (int)hipGetLastError();
}
