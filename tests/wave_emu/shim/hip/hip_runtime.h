// TEST INFRASTRUCTURE — a stand-in for <hip/hip_runtime.h> that lets g++ compile the DEVICE source of the solver kernel
// (multi_agent_pkgs_amd/csrc/hdsm_core.h + hdsm_wave_gi.h + hdsm_wave_gib.h) and run it on the CPU:
// the threads of one workgroup (one wavefront of 64, or four: the product's two launch shapes) are fibers (ucontext) run by one
// host thread; every cross-lane operation (DPP moves, v_readlane, v_permlane32_swap, ballot, shuffles, wsync) is a rendezvous of
// the lanes of ONE wavefront on its exchange buffer, __syncthreads() a rendezvous of the whole workgroup. See wave_emu.cpp.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <array>
#ifdef WEMU_DEBUG
#include <execinfo.h>
#endif

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__

namespace wemu {
struct Dim3 {
  unsigned x, y, z;
};
constexpr int W = 64;        // lanes of a wavefront
constexpr int MAXT = 256;    // threads of a workgroup (1 or 4 wavefronts)
struct Wave {                // rendezvous state of one wavefront
  int nlive = 0, arrived = 0;
  unsigned long generation = 0;
  int tag = -1;              // kind of the operation the current rendezvous belongs to (all lanes must agree)
  int parity = 0;            // exchange buffer in use
  int xi[2][W];
  double xd[2][W];
  int lane_tag[W];
};
struct Runtime {
  int cur = 0;               // thread whose fiber is running
  int nthreads = W;
  Wave wave[MAXT / W];
  int b_nlive = 0, b_arrived = 0;   // workgroup barrier
  unsigned long b_generation = 0;
  Dim3 tid[MAXT];
  Dim3 block{W, 1, 1}, grid{1, 1, 1}, bidx{0, 0, 0};
  long long clock = 0;
  long ops = 0;
  long by_kind[12] = {0};    // lockstep points by kind (1 barrier, 2 readlane, 3 ballot, 4/5 shuffle, 6 DPP, 7/8 permlane32_swap, 9 wsync)
#ifdef WEMU_DEBUG
  void* bt[MAXT][12];
  int nbt[MAXT];
#endif
};
Runtime& rt();
void yield();                  // back to the scheduler
[[noreturn]] void fail(const char* what);
inline int lane() { return rt().cur & (W - 1); }
inline Wave& my_wave() { return rt().wave[rt().cur / W]; }

// all live lanes of the calling lane's WAVEFRONT meet here; returns the buffer index that was written before the meeting
inline int rendezvous(int tag) {
  Runtime& r = rt();
  Wave& w = my_wave();
  w.lane_tag[lane()] = tag;
#ifdef WEMU_DEBUG
  r.nbt[r.cur] = backtrace(r.bt[r.cur], 12);
#endif
  if (w.arrived == 0) w.tag = tag;
  else if (w.tag != tag) {
    static char msg[200];
    unsigned long long at_first = 0;
    for (int l = 0; l < W; ++l) at_first |= (unsigned long long)(w.lane_tag[l] == w.tag && l != lane()) << l;
#ifdef WEMU_DEBUG
    fprintf(stderr, "---- arriving thread %d:\n", r.cur);
    backtrace_symbols_fd(r.bt[r.cur], r.nbt[r.cur], 2);
    for (int l = W - 1; l >= 0; --l)
      if ((at_first >> l) & 1) {
        fprintf(stderr, "---- waiting lane %d:\n", l);
        backtrace_symbols_fd(r.bt[(r.cur / W) * W + l], r.nbt[(r.cur / W) * W + l], 2);
        break;
      }
#endif
    snprintf(msg, sizeof msg, "lanes of wavefront %d reached DIFFERENT cross-lane operations (kinds %d and %d, %d lanes waiting, mask %llx): divergent use",
             r.cur / W, w.tag, tag, w.arrived, at_first);
    fail(msg);
  }
  const int p = w.parity;
  const unsigned long gen = w.generation;
  if (++w.arrived == w.nlive) {
    w.arrived = 0, ++w.generation, w.parity ^= 1, ++r.ops, ++r.by_kind[tag < 12 ? tag : 0];
  } else {
    while (w.generation == gen) yield();
  }
  return p;
}
// __syncthreads(): all live threads of the workgroup
inline void block_barrier() {
  Runtime& r = rt();
  const unsigned long gen = r.b_generation;
  if (++r.b_arrived == r.b_nlive) {
    r.b_arrived = 0, ++r.b_generation, ++r.ops, ++r.by_kind[1];
  } else {
    while (r.b_generation == gen) yield();
  }
}
}  // namespace wemu

#define threadIdx (wemu::rt().tid[wemu::rt().cur])
#define blockDim (wemu::rt().block)
#define gridDim (wemu::rt().grid)
#define blockIdx (wemu::rt().bidx)

struct alignas(16) double4 {
  double x, y, z, w;
};

// ---- scalar helpers ------------------------------------------------------------------------------------------------
inline int __double2loint(double v) {
  uint64_t u;
  memcpy(&u, &v, 8);
  return (int)(uint32_t)u;
}
inline int __double2hiint(double v) {
  uint64_t u;
  memcpy(&u, &v, 8);
  return (int)(uint32_t)(u >> 32);
}
inline double __hiloint2double(int hi, int lo) {
  const uint64_t u = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
  double v;
  memcpy(&v, &u, 8);
  return v;
}
inline double __longlong_as_double(long long x) {
  double v;
  memcpy(&v, &x, 8);
  return v;
}
inline long long __double_as_longlong(double x) {
  long long v;
  memcpy(&v, &x, 8);
  return v;
}
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline long long clock64() { return ++wemu::rt().clock; }
inline long long wall_clock64() { return ++wemu::rt().clock; }
inline int atomicAdd(int* p, int v) {
  const int old = *p;
  *p = old + v;
  return old;
}
inline int atomicCAS(int* p, int cmp, int val) {
  const int old = *p;
  if (old == cmp) *p = val;
  return old;
}
inline int atomicExch(int* p, int v) {
  const int old = *p;
  *p = v;
  return old;
}
inline unsigned long long atomicMin(unsigned long long* p, unsigned long long v) {
  const unsigned long long old = *p;
  if (v < old) *p = v;
  return old;
}
#define __ATOMIC_RELAXED_SHIM 0
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __hip_atomic_load(ptr, order, scope) (*(volatile decltype(ptr))(ptr))
#define __hip_atomic_store(ptr, val, order, scope) (*(volatile decltype(ptr))(ptr) = (val))
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __threadfence() ((void)0)
inline unsigned atomicOr(unsigned* p, unsigned v) {
  const unsigned old = *p;
  *p = old | v;
  return old;
}
inline unsigned atomicAnd(unsigned* p, unsigned v) {
  const unsigned old = *p;
  *p = old & v;
  return old;
}
inline unsigned atomicAdd(unsigned* p, unsigned v) {
  const unsigned old = *p;
  *p = old + v;
  return old;
}

// ---- cross-lane operations -----------------------------------------------------------------------------------------
inline void __syncthreads() { wemu::block_barrier(); }
inline int wemu_readlane_i(int v, int src) {
  wemu::Wave& w = wemu::my_wave();
  w.xi[w.parity][wemu::lane()] = v;
  const int p = wemu::rendezvous(2);
  return wemu::my_wave().xi[p][src & 63];
}
#define __builtin_amdgcn_readlane(v, l) wemu_readlane_i((v), (l))
#define __builtin_amdgcn_readfirstlane(v) wemu_readlane_i((v), 0)
inline unsigned long long __ballot(int pred) {
  wemu::Wave& w = wemu::my_wave();
  w.xi[w.parity][wemu::lane()] = pred ? 1 : 0;
  const int p = wemu::rendezvous(3);
  unsigned long long m = 0;
  for (int l = 0; l < wemu::W; ++l) m |= (unsigned long long)(wemu::my_wave().xi[p][l] & 1) << l;
  return m;
}
inline int __shfl_xor(int v, int mask, int width = 64) {
  wemu::Wave& w = wemu::my_wave();
  const int me = wemu::lane();
  w.xi[w.parity][me] = v;
  const int p = wemu::rendezvous(4);
  const int src = me ^ mask;
  return (src / width == me / width) ? wemu::my_wave().xi[p][src] : v;
}
inline int __shfl(int v, int src, int width = 64) {
  wemu::Wave& w = wemu::my_wave();
  const int me = wemu::lane();
  w.xi[w.parity][me] = v;
  const int p = wemu::rendezvous(4);
  return wemu::my_wave().xi[p][(me & ~(width - 1)) | (src & (width - 1))];
}
inline int __shfl_up(int v, int delta, int width = 64) {
  wemu::Wave& w = wemu::my_wave();
  const int me = wemu::lane();
  w.xi[w.parity][me] = v;
  const int p = wemu::rendezvous(4);
  return (me & (width - 1)) >= delta ? wemu::my_wave().xi[p][me - delta] : v;
}
inline double __shfl_xor(double v, int mask, int width = 64) {
  wemu::Wave& w = wemu::my_wave();
  const int me = wemu::lane();
  w.xd[w.parity][me] = v;
  const int p = wemu::rendezvous(5);
  const int src = me ^ mask;
  return (src / width == me / width) ? wemu::my_wave().xd[p][src] : v;
}
// DPP moves within 16-lane rows: quad_perm (0x00..0xff: lane 4g + k reads lane 4g + sel_k, two bits per k), row_shl:n (0x100 + n)
// reads lane + n, row_shr:n (0x110 + n) lane - n, row_ror:n (0x120 + n) rotates, row_mirror (0x140) reads lane 15 - i,
// row_half_mirror (0x141) lane 7 - i within each half row; a lane without a source keeps `old` (bound_ctrl: the callers pass
// old = 0 where they want zero fill)
inline int wemu_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  wemu::Wave& w = wemu::my_wave();
  const int me = wemu::lane();
  w.xi[w.parity][me] = src;
  const int p = wemu::rendezvous(6);
  const int row = me & ~15, i = me & 15, n = ctrl & 15;
  int from = -1;
  if (ctrl >= 0 && ctrl <= 0xff) from = (i & ~3) | ((ctrl >> (2 * (i & 3))) & 3);
  else if (ctrl == 0x140) from = 15 - i;
  else if (ctrl == 0x141) from = (i & 8) | (7 - (i & 7));
  else if ((ctrl & 0x1f0) == 0x100) from = (i + n < 16) ? i + n : -1;
  else if ((ctrl & 0x1f0) == 0x110) from = (i - n >= 0) ? i - n : -1;
  else if ((ctrl & 0x1f0) == 0x120) from = (i - n) & 15;
  else wemu::fail("DPP control not modelled");
  (void)row_mask, (void)bank_mask;
  return from >= 0 ? wemu::my_wave().xi[p][row + from] : (bound_ctrl ? 0 : old);
}
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) wemu_dpp((old), (src), (ctrl), (rm), (bm), (bc))
#define __builtin_amdgcn_mov_dpp(src, ctrl, rm, bm, bc) wemu_dpp(0, (src), (ctrl), (rm), (bm), (bc))  // (no `old`: undefined where no lane is read)
// v_permlane32_swap vdst, vsrc: lanes 32..63 of vdst <-> lanes 0..31 of vsrc; returns {new vdst, new vsrc}
inline std::array<int, 2> wemu_permlane32_swap(int vdst, int vsrc, bool, bool) {
  const int me = wemu::lane();
  wemu::my_wave().xi[wemu::my_wave().parity][me] = vdst;
  const int p1 = wemu::rendezvous(7);
  const int other_dst = wemu::my_wave().xi[p1][me ^ 32];
  wemu::my_wave().xi[wemu::my_wave().parity][me] = vsrc;
  const int p2 = wemu::rendezvous(8);
  const int other_src = wemu::my_wave().xi[p2][me ^ 32];
  return me < 32 ? std::array<int, 2>{vdst, other_dst} : std::array<int, 2>{other_src, vsrc};
}
#define __builtin_amdgcn_permlane32_swap(a, b, c, d) wemu_permlane32_swap((a), (b), (c), (d))
// v_permlane16_swap vdst, vsrc: the ODD 16-lane rows of vdst <-> the EVEN rows of vsrc (rows 1 <-> 0 and 3 <-> 2)
inline std::array<int, 2> wemu_permlane16_swap(int vdst, int vsrc, bool, bool) {
  const int me = wemu::lane();
  wemu::my_wave().xi[wemu::my_wave().parity][me] = vdst;
  const int p1 = wemu::rendezvous(10);
  const int other_dst = wemu::my_wave().xi[p1][me ^ 16];
  wemu::my_wave().xi[wemu::my_wave().parity][me] = vsrc;
  const int p2 = wemu::rendezvous(11);
  const int other_src = wemu::my_wave().xi[p2][me ^ 16];
  return (me & 16) == 0 ? std::array<int, 2>{vdst, other_dst} : std::array<int, 2>{other_src, vsrc};
}
#define __builtin_amdgcn_permlane16_swap(a, b, c, d) wemu_permlane16_swap((a), (b), (c), (d))
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_wave_barrier() ((void)wemu::rendezvous(9))  // wsync(): on the device the lanes are in lockstep anyway
#define __builtin_amdgcn_rsq(x) (1.0 / sqrt((double)(x)))
#define __builtin_amdgcn_rcp(x) (1.0 / (double)(x))
