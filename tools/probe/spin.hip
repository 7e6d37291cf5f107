// probe kernels for tools/probe_boundary.py: a kernel of `nwg` workgroups that holds its CUs for `us` microseconds (optionally
// streaming `buf` meanwhile), to stand in for the narrow weight-gradient kernels of the side stream
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void __launch_bounds__(256) spin_kernel(long long ticks, float* buf, long long n) {
  extern __shared__ float lds[];
  const long long t0 = wall_clock64();
  float acc = 0.f;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  while (wall_clock64() - t0 < ticks) {
    if (buf != nullptr) {
      for (int k = 0; k < 16; ++k) { acc += buf[i]; i += (long long)gridDim.x * 256; if (i >= n) i -= n; }
    }
  }
  if (acc == 12345.f) lds[threadIdx.x] = acc, buf[0] = lds[0];
}
extern "C" int probe_spin(int nwg, int lds_bytes, double us, float* buf, long long n, void* stream) {
  static bool once = false;
  if (!once) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; }
  hipLaunchKernelGGL(spin_kernel, dim3(nwg), dim3(256), lds_bytes, (hipStream_t)stream, (long long)(us * 100.0), buf, n);      // wall_clock64: 100 MHz
  return (int)hipGetLastError();
}
