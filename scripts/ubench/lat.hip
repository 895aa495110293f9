// Latency micro-benchmarks of the primitives the active-set iteration is built from, for ONE wavefront working alone on a CU
// (the situation of the iterating wave) — cycles by s_memtime. usage: ./lat [busy]  (busy = 1: a second wave on the CU hammers LDS)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define N_T 24
__device__ __forceinline__ long long now() { return (long long)__builtin_amdgcn_s_memtime(); }
// clock read that is ordered AFTER the completion of the VALU value x (v_readfirstlane waits for it) and whose result is waited for
__device__ __forceinline__ long long now_after(double& x) {
  long long t;
  int lo = __double2loint(x);
  asm volatile("v_readfirstlane_b32 s20, %1\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(lo) : : "s20", "memory");
  x = __hiloint2double(__double2hiint(x), lo);
  return t;
}
template <int CTRL>
__device__ __forceinline__ double dpp64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bcast64(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double half_sum64(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ double wave_max64(double v) {
  v = fmax(v, dpp64<0x121>(v)); v = fmax(v, dpp64<0x122>(v)); v = fmax(v, dpp64<0x124>(v)); v = fmax(v, dpp64<0x128>(v));
  return fmax(fmax(bcast64(v, 0), bcast64(v, 16)), fmax(bcast64(v, 32), bcast64(v, 48)));
}
__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__global__ __launch_bounds__(128) void k(long long* out, double* sink, int busy, int reps) {
  __shared__ double lds[4096];
  __shared__ int idx[1024];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 1.0 + 1e-3 * i;
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) idx[i] = (i * 37 + 11) & 1023;
  __syncthreads();
  if (w == 1) {  // optional noise: a second wave doing LDS traffic for the whole duration
    if (busy) {
      double acc = 0;
      for (int r = 0; r < reps * 400; ++r) { acc += lds[(lane * 5 + r) & 4095]; lds[2048 + ((lane + r) & 1023)] = acc; }
      sink[64 + lane] = acc;
    }
    return;
  }
  long long t[N_T];
  for (int i = 0; i < N_T; ++i) t[i] = 0;
  double v = 1.0 + lane, acc = 0;
  for (int r = 0; r < reps; ++r) {
    long long t0, t1;
    // 0: empty timing pair
    t0 = now_after(v); t1 = now_after(v); t[0] += t1 - t0;
    // 1: ONE dependent LDS read (address from a register -> value)
    { int a = lane; t0 = now_after(v); double x = lds[a]; asm volatile("" : "+v"(x)); t1 = now_after(x); t[1] += t1 - t0; acc += x; }
    // 2: chain of 4 dependent LDS reads (index -> index -> index -> value)
    { int a = lane; t0 = now_after(v); a = idx[a]; a = idx[a]; a = idx[a]; double x = lds[a]; asm volatile("" : "+v"(x)); t1 = now_after(x); t[2] += t1 - t0; acc += x; }
    // 3: 20 independent ds_read_b64 + sum
    { t0 = now_after(v); double x = 0; 
#pragma unroll
      for (int j = 0; j < 20; ++j) x += lds[lane + 64 * j]; asm volatile("" : "+v"(x)); t1 = now_after(x); t[3] += t1 - t0; acc += x; }
    // 4: 8 independent ds_read_b128 + sum
    { t0 = now_after(v); double x = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) { const double2 q = *reinterpret_cast<const double2*>(&lds[2 * lane + 128 * j]); x += q.x + q.y; } asm volatile("" : "+v"(x)); t1 = now_after(x); t[4] += t1 - t0; acc += x; }
    // 5: LDS write then read by another lane (write, wsync, read)
    { t0 = now_after(v); lds[lane] = v; wsync(); double x = lds[63 - lane]; asm volatile("" : "+v"(x)); t1 = now_after(x); t[5] += t1 - t0; acc += x; }
    // 6: 16 ds_write_b64 + wsync
    { t0 = now_after(v);
#pragma unroll
      for (int j = 0; j < 16; ++j) lds[lane + 64 * j] = v + j; wsync(); t1 = now_after(v); t[6] += t1 - t0; }
    // 7: one dpp64 dependent chain of 8
    { double x = v; t0 = now_after(x);
#pragma unroll
      for (int j = 0; j < 8; ++j) x += dpp64<0x140>(x); asm volatile("" : "+v"(x)); t1 = now_after(x); t[7] += t1 - t0; acc += x; }
    // 8: reduce-scatter butterfly on 16 values (15 adds, 30 dpp) + permlane16
    { double p[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) p[j] = v * (j + 1);
      asm volatile("" : "+v"(p[0]), "+v"(p[15]));
      t0 = now_after(p[0]);
#pragma unroll
      for (int s = 0; s < 8; ++s) p[s] += dpp64<0x140>(p[15 - s]);
#pragma unroll
      for (int s = 0; s < 4; ++s) p[s] += dpp64<0x141>(p[7 - s]);
#pragma unroll
      for (int s = 0; s < 2; ++s) p[s] += dpp64<0x4E>(p[s ^ 2]);
      p[0] += dpp64<0xB1>(p[1]);
      double x = p[0]; asm volatile("" : "+v"(x)); t1 = now_after(x); t[8] += t1 - t0; acc += x; }
    // 9: all-gather butterfly (30 dpp) + 16 FMA dot
    { double g[16]; g[0] = v; t0 = now_after(g[0]);
      g[1] = dpp64<0xB1>(g[0]);
#pragma unroll
      for (int s = 0; s < 2; ++s) g[s ^ 2] = dpp64<0x4E>(g[s]);
#pragma unroll
      for (int s = 0; s < 4; ++s) g[7 - s] = dpp64<0x141>(g[s]);
#pragma unroll
      for (int s = 0; s < 8; ++s) g[15 - s] = dpp64<0x140>(g[s]);
      double x0 = 0, x1 = 0;
#pragma unroll
      for (int j = 0; j < 16; j += 2) x0 += g[j] * (v + j), x1 += g[j + 1] * (v - j);
      double x = x0 + x1; asm volatile("" : "+v"(x)); t1 = now_after(x); t[9] += t1 - t0; acc += x; }
    // 10: half_sum64 (permlane32 swap x2 + add)
    { double x = v; t0 = now_after(x); x = half_sum64(x); asm volatile("" : "+v"(x)); t1 = now_after(x); t[10] += t1 - t0; acc += x; }
    // 11: wave_max64 + ballot + readlane (argmax)
    { double x = v; int id = lane; t0 = now_after(x); const double m = wave_max64(x); const unsigned long long mask = __ballot(x == m); int best = __builtin_amdgcn_readlane(id, __ffsll((long long)mask) - 1); asm volatile("" : "+s"(best)); t1 = now_after(x); t[11] += t1 - t0; acc += m + best; }
    // 12: fp64 division
    { double x = v; t0 = now_after(x); x = (v + 3.0) / x; asm volatile("" : "+v"(x)); t1 = now_after(x); t[12] += t1 - t0; acc += x; }
    // 13: fp64 sqrt
    { double x = v; t0 = now_after(x); x = sqrt(x); asm volatile("" : "+v"(x)); t1 = now_after(x); t[13] += t1 - t0; acc += x; }
    // 14: 16 dependent fp64 FMA
    { double x = v; t0 = now_after(x);
#pragma unroll
      for (int j = 0; j < 16; ++j) x = fma(x, 1.0000001, 0.5); asm volatile("" : "+v"(x)); t1 = now_after(x); t[14] += t1 - t0; acc += x; }
    // 15: 16 independent fp64 FMA (4 chains of 4)
    { double a0 = v, a1 = v + 1, a2 = v + 2, a3 = v + 3; t0 = now_after(a0);
#pragma unroll
      for (int j = 0; j < 4; ++j) { a0 = fma(a0, 1.0000001, 0.5); a1 = fma(a1, 1.0000001, 0.5); a2 = fma(a2, 1.0000001, 0.5); a3 = fma(a3, 1.0000001, 0.5); }
      double x = a0 + a1 + a2 + a3; asm volatile("" : "+v"(x)); t1 = now_after(x); t[15] += t1 - t0; acc += x; }
    // 16: bcast64 (2 readlane) then a VALU use
    { double x = v; t0 = now_after(x); double y = bcast64(x, 17) * x; asm volatile("" : "+v"(y)); t1 = now_after(y); t[16] += t1 - t0; acc += y; }
    // 17: uniform LDS read (all lanes same address) then dependent uniform read
    { t0 = now_after(v); int a = idx[r & 1023]; double x = lds[a]; asm volatile("" : "+v"(x)); t1 = now_after(x); t[17] += t1 - t0; acc += x; }
    // 18: global load (L2-resident)
    { t0 = now_after(v); double x = sink[128 + lane]; asm volatile("" : "+v"(x)); t1 = now_after(x); t[18] += t1 - t0; acc += x; }
    // 19: wave_sum via dpp + readlanes
    { double x = v; t0 = now_after(x); x += dpp64<0x121>(x); x += dpp64<0x122>(x); x += dpp64<0x124>(x); x += dpp64<0x128>(x); double y = (bcast64(x, 0) + bcast64(x, 16)) + (bcast64(x, 32) + bcast64(x, 48)); asm volatile("" : "+v"(y)); t1 = now_after(y); t[19] += t1 - t0; acc += y; }
    v += 1e-9 * acc;
  }
  {  // 20: 4096 dependent FMAs: ticks vs the wall clock of the launch give the tick rate
    double x = v; long long t0 = now_after(x);
    for (int j = 0; j < 4096; ++j) x = fma(x, 1.0000001, 0.5);
    long long t1 = now_after(x); t[20] = (t1 - t0) * reps; acc += x;
  }
  if (lane == 0) for (int i = 0; i < N_T; ++i) out[blockIdx.x * N_T + i] = t[i];
  sink[lane] = acc + v;
}
int main(int argc, char** argv) {
  const int busy = argc > 1 ? atoi(argv[1]) : 0, reps = 200, blocks = argc > 2 ? atoi(argv[2]) : 1;
  long long* d_out; double* d_sink;
  hipMalloc(&d_out, blocks * N_T * 8); hipMalloc(&d_sink, 4096 * 8); hipMemset(d_sink, 0, 4096 * 8);
  for (int it = 0; it < 2; ++it) { hipLaunchKernelGGL(k, dim3(blocks), dim3(128), 0, 0, d_out, d_sink, busy, reps); hipDeviceSynchronize(); }
  { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(128), 0, 0, d_out, d_sink, busy, reps); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); printf("kernel wall %.3f ms\n", ms); }
  std::vector<long long> h(blocks * N_T); hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
  const char* nm[N_T] = {"empty pair", "1 dependent ds_read_b64", "chain of 4 dependent LDS reads", "20 indep ds_read_b64 + sum", "8 indep ds_read_b128 + sum", "LDS write, wsync, read", "16 ds_write_b64 + wsync", "8 dependent dpp64+add", "reduce-scatter 16 (15 add, 30 dpp)", "all-gather (30 dpp) + 16 fma dot", "half_sum64", "wave_max64+ballot+readlane", "fp64 div", "fp64 sqrt", "16 dependent fma", "16 fma in 4 chains", "bcast64 + mul", "2 dependent uniform LDS reads", "global load (L2)", "wave_sum64", "4096 dependent fma", "", "", ""};
  printf("busy=%d blocks=%d: s_memtime ticks per call (100 MHz constant clock? or shader clock), mean over %d reps, block 0\n", busy, blocks, reps);
  for (int i = 0; i < 21; ++i) printf("%2d %-40s %8.1f (minus empty: %8.1f)\n", i, nm[i], (double)h[i] / reps, (double)(h[i] - h[0]) / reps);
  return 0;
}
