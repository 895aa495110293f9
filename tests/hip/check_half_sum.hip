// Development check of the cross-half primitives used by the split active-set kernel (hdsm_wave_gi.h):
// v_permlane32_swap on gfx950. Build: hipcc --offload-arch=gfx950 -O2 -o /tmp/check_half_sum check_half_sum.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ double half_sum64(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__device__ double half_lo64(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]);
}
__device__ double half_hi64(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(b[1], a[1]);
}
__global__ void k(double* out) {
  const int lane = threadIdx.x;
  const double v = 1000.0 * (lane >> 5) + (lane & 31) + 0.25;
  out[lane] = half_sum64(v);
  out[64 + lane] = half_lo64(v);
  out[128 + lane] = half_hi64(v);
}
int main() {
  double* d;
  double h[192];
  hipMalloc(&d, sizeof h);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane) {
    const int i = lane & 31;
    const double lo = i + 0.25, hi = 1000.0 + i + 0.25;
    if (h[lane] != lo + hi || h[64 + lane] != lo || h[128 + lane] != hi) ++bad;
  }
  printf("half-sum check: %s (lane 5: sum %.2f lo %.2f hi %.2f; lane 37: sum %.2f lo %.2f hi %.2f)\n", bad ? "FAIL" : "ok",
         h[5], h[69], h[133], h[37], h[101], h[165]);
  return bad != 0;
}
