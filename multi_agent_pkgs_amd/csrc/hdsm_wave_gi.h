// hdsm_wave_gi.h — device-only (gfx950): what the dual active-set iteration of ONE instance needs around its factor algebra, for
// ONE 64-lane wavefront (plus the scanner wave): lane mapping and per-lane constants, the state evaluation from the impulse
// responses, the violation scans and the pick rule, normals and residuals of a row, the neighbour sweep, cold rows.
//
// The algebra itself — J = L^{-T} Q with J^T N = [R; 0] in registers in butterfly order, U = R^{-1} in LDS, Householder add /
// drop, warm start, the regular loop and the scanner protocol — is hdsm_wave_gib.h (WaveGIB<NV, ...> derives from WaveGI<NV, ...>),
// for both instantiations: NV = 32 (n <= 30: every row of J split over two lanes) and NV = 48 (one lane per row). Until round 4
// this file also held the one-lane-per-row algebra of NV = 48 (LDS transposition for d = J^T a, select chains on run-time
// register indices); it is gone.
// Dimension handling: the factors are padded to NV (identity beyond n = 3N), so every loop has a
// compile-time trip count and unrolls; a padded direction never receives a step (d_k = 0 there).
#pragma once
#include <hip/hip_runtime.h>

#include "hdsm_types.h"

namespace hdsm {

template <int CTRL>
__device__ __forceinline__ double dpp64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  // (every control used with dpp64 is a permutation inside the row: each lane receives a value, so there is no "old" value to keep —
  // with update_dpp(x, x, ...) the compiler copied x first: one v_mov_b32 per v_mov_b32_dpp, 170 of them per active-set operation)
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bcast64(double v, int lane) {  // lane must be wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max64(double v) {
  v = fmax(v, dpp64<0x121>(v));  // row_ror:1
  v = fmax(v, dpp64<0x122>(v));  // row_ror:2
  v = fmax(v, dpp64<0x124>(v));  // row_ror:4
  v = fmax(v, dpp64<0x128>(v));  // row_ror:8  -> every lane holds the max of its 16-lane row
  return fmax(fmax(bcast64(v, 0), bcast64(v, 16)), fmax(bcast64(v, 32), bcast64(v, 48)));
}
__device__ __forceinline__ double wave_sum64(double v) {  // every lane gets the sum over the wave
  v += dpp64<0x121>(v);
  v += dpp64<0x122>(v);
  v += dpp64<0x124>(v);
  v += dpp64<0x128>(v);
  return (bcast64(v, 0) + bcast64(v, 16)) + (bcast64(v, 32) + bcast64(v, 48));
}
// DPP move with zero fill for lanes shifted in from outside the 16-lane row
template <int CTRL>
__device__ __forceinline__ double dpp64z(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);  // (bound_ctrl: a lane that reads from outside its row gets 0)
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// suffix sum over the wave: out[i] = sum_{k >= i} v[k]   (row_shl scans + row totals through v_readlane)
__device__ __forceinline__ double wave_suffix_sum(double v, int lane) {
  v += dpp64z<0x101>(v);
  v += dpp64z<0x102>(v);
  v += dpp64z<0x104>(v);
  v += dpp64z<0x108>(v);
  const double t1 = bcast64(v, 16), t2 = bcast64(v, 32), t3 = bcast64(v, 48);
  const int row = lane >> 4;
  return v + (row == 0 ? (t1 + t2) + t3 : (row == 1 ? t2 + t3 : (row == 2 ? t3 : 0.0)));
}
// prefix sum over the wave: out[i] = sum_{k <= i} v[k]
__device__ __forceinline__ double wave_prefix_sum(double v, int lane) {
  v += dpp64z<0x111>(v);
  v += dpp64z<0x112>(v);
  v += dpp64z<0x114>(v);
  v += dpp64z<0x118>(v);
  const double t0 = bcast64(v, 15), t1 = bcast64(v, 31), t2 = bcast64(v, 47);
  const int row = lane >> 4;
  return v + (row == 3 ? (t0 + t1) + t2 : (row == 2 ? t0 + t1 : (row == 1 ? t0 : 0.0)));
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// v[lane & 31] + v[32 + (lane & 31)] in every lane (v_permlane32_swap: no LDS round trip). The split kernel keeps the
// two halves of a row of J in lanes L and L + 32; this is how their partial dot products meet.
__device__ __forceinline__ double half_sum64(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}

// Ordering point INSIDE the one wavefront that runs the active-set iteration. LDS operations of a wave execute in
// program order, so no s_barrier is needed (and none may be used: the other waves of the workgroup are parked at a
// real barrier while wave 0 iterates) — only the compiler must not move LDS accesses across it.
__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// 1 / x and 1 / sqrt(x) to double precision for a NORMAL, finite, non-zero x: hardware seed + two Newton steps, 5 / 7 instructions
// instead of the ~17 / ~25 of the IEEE expansions (v_div_scale / v_div_fmas / v_div_fixup ...). The active-set operation is
// bound by the number of VALU instructions its one wavefront issues, so the scalar reciprocals of the Householder updates use these.
__device__ __forceinline__ double rcp_nr(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = fma(fma(-x, y, 1.0), y, y);
  y = fma(fma(-x, y, 1.0), y, y);
  return y;
}
__device__ __forceinline__ double rsq_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
}

// A slot of the staging area for one row: hot rows fill it from the bottom (counter `ncand`), cold rows from the top (`ncold`), every
// thread of the workgroup at once. A writer takes its slot with an atomic increment of its own counter and THEN reads the other list's
// counter: if the two lists have met, whichever writer incremented second sees it and gives way, so no slot is ever written twice.
// "Then" must hold in the machine code: the increment ACQUIRES (nothing after it moves in front of it) and the other counter is read
// ATOMICALLY (never a value the compiler kept from before). Written as a plain `s.ncold` after a relaxed atomicAdd the order was only an
// accident of instruction scheduling — one register-allocation change later (round 6) the read of a loop-invariant-looking `s.ncold`
// sat in front of the loop, both lists wrote through each other as soon as the area filled up, and the 320-row kernel returned
// "infeasible" for feasible instances WITHOUT the overflow flag.
template <int CMAX>
__device__ __forceinline__ bool stage_slot(int32_t& ncand, int32_t& ncold, bool hot, int& slot) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (hot) {
    slot = __hip_atomic_fetch_add(&ncand, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    const int other = __hip_atomic_load(&ncold, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return slot >= 0 && slot < CMAX - other;
  }
  slot = CMAX - 1 - __hip_atomic_fetch_add(&ncold, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
  const int other = __hip_atomic_load(&ncand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return slot < CMAX && slot >= other && slot >= 0;
#else
  slot = hot ? atomicAdd(&ncand, 1) : CMAX - 1 - atomicAdd(&ncold, 1);
  return (hot ? slot < CMAX - ncold : slot >= ncand) && slot >= 0 && slot < CMAX;
#endif
}

// an integer the optimiser must treat as recomputed here (no effect on the value): stops it from hoisting what depends on it
__device__ __forceinline__ void keep_in_loop(int& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(v));
#else
  (void)v;
#endif
}

// the thread index as a value formed HERE: the thread-indexed LDS addresses of the all-thread loops (PAR_FOR) are then computed where
// they are used; hoisted out of the branch-and-bound loop they stayed alive across the active-set run and were spilled
__device__ __forceinline__ int tid_here() {
  int t = (int)HDSM_TX;
  keep_in_loop(t);
  return t;
}

// a value the optimiser must treat as used (and redefined) here: keeps its computation above this point
__device__ __forceinline__ void keep_here(double& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(v));
#else
  (void)v;
#endif
}

// ... and the same for a pointer
template <class T>
__device__ __forceinline__ T* keep_in_loop(T* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(p));
#endif
  return p;
}

struct alignas(16) D2 {
  double x, y;
};

// A pointer the kernel KNOWS to address global memory. The launch arguments are kept in LDS (Shm::args), and a pointer that comes
// out of LDS is a generic one: its loads are flat_load, which count on the LDS counter as well and may alias every LDS access near
// them — the compiler orders them accordingly (a restore of a snapshot was 48 chained round trips, hdsm_wave_gib.h).
#if defined(__HIP_DEVICE_COMPILE__)
template <class T>
using GPtr = __attribute__((address_space(1))) T*;
template <class T>
__device__ __forceinline__ GPtr<const T> gptr(const T* p) { return (GPtr<const T>)p; }
template <class T>
__device__ __forceinline__ GPtr<T> gptr(T* p) { return (GPtr<T>)p; }
#else
template <class T>
using GPtr = T*;
template <class T>
__device__ __host__ inline GPtr<const T> gptr(const T* p) { return p; }  // (the host pass of hipcc, and the CPU execution of the tests)
template <class T>
__device__ __host__ inline GPtr<T> gptr(T* p) { return p; }
#endif

#ifdef HDSM_PROFILE
// cycle counters accumulated in LDS by lane 0 (keeps them out of the SGPR file)
#define PROF_DECL if (HDSM_TX == 0) s.prof_last = clock64();
#define PROF(k) if (HDSM_TX == 0) { const long long now_ = clock64(); s.prof_acc[k] += now_ - s.prof_last; s.prof_last = now_; }
#elif defined(HDSM_ISA_MARKS)
// ISA study build (hipcc -S -DHDSM_ISA_MARKS): the phase boundaries show up as comments in the listing
#define PROF_DECL
#define PROF(k) asm volatile("; ISA_MARK " #k);
#else
#define PROF_DECL
#define PROF(k)
#endif
// -DHDSM_PROF_STAGE (with -DHDSM_PROFILE): slots 16..20 time the steps of the staging phase instead of the sweeps
// -DHDSM_PROF_OP (with -DHDSM_PROFILE): slots 8..23 time the inside of a regular active-set operation (OP_PROF) instead of the
// sweeps, the set-up and the warm start
// -DHDSM_PROF_NODE (with -DHDSM_PROFILE): slots 16..23 time the node machinery of the branch and bound (NODE_PROF in hdsm_core.h)
// instead of the warm start: 16 end of a run, 17 leaf test, 18 verification / incumbent, 19 snapshot, 20 level, 21 conflict, 22 next child
#if defined(HDSM_PROFILE) && defined(HDSM_PROF_NODE)
#define SW_PROF(k) PROF(k)
#define ST_PROF(k)
#define WS_PROF(k)
#define OP_PROF(k)
#define NODE_PROF(k) PROF(k)
#elif defined(HDSM_PROFILE) && defined(HDSM_PROF_OP)
#define SW_PROF(k)
#define ST_PROF(k)
#define WS_PROF(k)
#define OP_PROF(k) PROF(k)
#define SC_PROF_DECL if (HDSM_TX == 64) s.prof_last1 = clock64();
#define SC_PROF(k) if (HDSM_TX == 64) { const long long now_ = clock64(); s.prof_acc[k] += now_ - s.prof_last1; s.prof_last1 = now_; }
#elif defined(HDSM_PROFILE) && defined(HDSM_PROF_STAGE)
#define SW_PROF(k)
#define ST_PROF(k) PROF(k)
#define WS_PROF(k) PROF(k)
#define OP_PROF(k)
#else
#define SW_PROF(k) PROF(k)
#define ST_PROF(k)
#define WS_PROF(k) PROF(k)
#define OP_PROF(k)
#endif
#ifndef NODE_PROF
#define NODE_PROF(k)
#endif
#if !defined(SC_PROF) && defined(HDSM_ISA_MARKS)
#define SC_PROF_DECL
#define SC_PROF(k) asm volatile("; ISA_MARK SC " #k);
#endif
#ifndef SC_PROF
#define SC_PROF_DECL
#define SC_PROF(k)
#endif

template <int NV, int CMAX, bool SMALL = false>
struct WaveGI {
  using S = Shm<NV, CMAX, SMALL>;
  static constexpr int LDT = S::LDT;
  static constexpr int HT = NV / 3;  // horizon capacity of this instantiation
  // NV = 32 (n <= 30): every row of J is SPLIT over two lanes, lane L holds columns [16 h, 16 h + 16) of row L & 31,
  // h = L >> 5, so the unrolled per-lane loops are 16 long and all 64 lanes work; partial dot products meet through
  // half_sum64. NV = 48: one lane per row, 48 columns. Per-row scalars (x, z, r, multipliers) always live in lane =
  // row index (< NV); in the split kernel lanes 32.. carry copies.
  static constexpr bool SPLIT = (NV == 32);
  static constexpr int NC = SPLIT ? NV / 2 : NV;
  static __device__ __forceinline__ int row_of(int lane) { return SPLIT ? (lane & 31) : lane; }
  static __device__ __forceinline__ int col0_of(int lane) { return SPLIT ? (lane >> 5) * NC : 0; }
  static __device__ __forceinline__ bool row_ok(int lane) { return SPLIT || lane < NV; }
  static __device__ __forceinline__ double hsum(double v) {
    if constexpr (SPLIT) return half_sum64(v);
    else return v;
  }

  struct Regs {
    double Jr[NC];  // columns [col0, col0 + NC) of row row_of(lane) of J (NV = 32: in the slot order of hdsm_wave_gib.h)
    double xi;      // u[lane]
    double lam;     // NV = 32 (hdsm_wave_gib.h): multiplier and id of working-set position pos_of(lane) while a run is going on
    int act;
    // per-lane constants of the violation scan, set by init_lane(). The bounds themselves are read from LDS (Shm::bnd, "absent"
    // mapped to +-DINF): six doubles per lane held in registers across the whole instance were spilled to scratch and reloaded
    // INSIDE every scan.
    int sb_id[2];               // id base (step << 5 | comp << 3 | axis << 1) of the (up to two) state-bound items of this lane, -1 = none
    int ax, kk;                 // variable row_of(lane) = jerk of axis ax at step kk (runtime divisions done once)
    float wu, sb_w[2];          // pick-rule weights of this lane's input box and state-bound items
  };

  // The warm-start guess of an instance as wave 0 fetches it during the set-up: lane g holds entry g of the stored working set
  // (head = count | WARM_CERT in every lane) and, for a neighbour row, what it is rebuilt from — the neighbour's has_plan flag and
  // its position at the row's step (requested by warm_prefetch as soon as the entry is there). Two dependent round trips to global
  // memory that used to start when the warm start began (2.8 us per instance) now overlap the set-up.
  struct WarmPre {
    int32_t head, code;
    int32_t has;  // the row's neighbour has a plan (0 also for entries that are not neighbour rows)
    double ox, oy, oz;
  };
  static __device__ __forceinline__ void warm_prefetch(const Args& g, WarmPre& w, int lane, int self, int N) {
    w.has = 0;
    int nw = w.head & ~WARM_CERT;
    if (nw > NV) nw = NV;
    if (lane < nw && id_kind(w.code) == K_C) {
      const int p = id_payload(w.code);
      const int i = ((p >> 1) & 31) - 1, k = p >> 6;
      if (i >= 0 && k != self && k < g.n_rob) {
        const double* op = g.pos + ((int64_t)k * N + i) * 3;
        w.has = g.has_plan[k], w.ox = op[0], w.oy = op[1], w.oz = op[2];
      }
    }
  }

  // per-lane constants of the iteration in two steps: the loads (issued with the staging requests of the set-up, so that
  // their latency is not a round trip of its own) and, once those have been consumed, the registers
  struct LaneReq {
    double hrow;  // Consts::hrow1 of this lane's variable (used once, by the set-up)
    double wu, sb_w[2];
    int comp[2], axs[2], ii[2];
  };
  static __device__ __forceinline__ LaneReq init_lane_request(const Consts& c, int lane) {
    const int n = c.n;
    LaneReq q;
    q.wu = c.wu[lane < n ? lane : 0];
    q.hrow = c.hrow1[lane < n ? lane : 0];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int idx = lane + 64 * e, k = idx % 6;
      q.ii[e] = idx / 6 + 1, q.comp[e] = 1 + k / 3, q.axs[e] = k % 3;
      q.sb_w[e] = c.ws[q.axs[e]][q.comp[e]][q.ii[e] <= MAXH ? q.ii[e] : 0];
    }
    return q;
  }
  static __device__ __forceinline__ void init_lane(Regs& R, const Consts& c, int lane, const LaneReq& q) {
    const int N = c.N;
    const int n_sb = 6 * (N - 1);
    R.ax = row_of(lane) / N, R.kk = row_of(lane) % N;
    R.wu = (float)q.wu;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const bool on = lane + 64 * e < n_sb;
      R.sb_id[e] = on ? ((q.ii[e] << 5) | (q.comp[e] << 3) | (q.axs[e] << 1)) : -1;
      R.sb_w[e] = (float)q.sb_w[e];
    }
  }

  // trajectory from s.x: lane (ax, m-1) evaluates p, v, a of step m (zero-padded Toeplitz table gz in LDS)
  // PART: 0 = positions, velocities and accelerations; 1 = positions only; 2 = velocities and accelerations only
  template <int PART = 0>
  static __device__ __forceinline__ void states(S& s, const Regs& R, int lane, int N) {
    if constexpr (SPLIT) {  // both halves of the wave work: each sums half of the impulse-response taps
      constexpr bool P0 = PART != 2, VA = PART != 1;
      const int row = lane & 31, h = lane >> 5;
      const bool on = row < 3 * N;
      const int ax = on ? R.ax : 0, m = on ? R.kk + 1 : 1;
      double acc0 = 0, acc1 = 0, acc2 = 0;
      if (h == 0) {
        if constexpr (P0) acc0 = s.fr[ax][m][0];
        if constexpr (VA) acc1 = s.fr[ax][m][1], acc2 = s.fr[ax][m][2];
      }
      const double* xx = s.x + ax * N;
      const double* g0 = &s.gz[ax][0][MAXH + m - 1];
      const double* g1 = &s.gz[ax][1][MAXH + m - 1];
      const double* g2 = &s.gz[ax][2][MAXH + m - 1];
      constexpr int HH = HT / 2;
      static_assert(HT % 2 == 0, "the split kernel halves the horizon capacity");
#pragma unroll
      for (int k = 0; k < HH; ++k) {
        const int kk = h * HH + k;
        const double xk = xx[kk];
        if constexpr (P0) acc0 += g0[-kk] * xk;
        if constexpr (VA) acc1 += g1[-kk] * xk, acc2 += g2[-kk] * xk;
      }
      if constexpr (P0) acc0 = half_sum64(acc0);
      if constexpr (VA) acc1 = half_sum64(acc1), acc2 = half_sum64(acc2);
      if (on && h == 0) {
        if constexpr (P0) s.st[m][ax] = acc0;
        if constexpr (VA) s.st[m][3 + ax] = acc1, s.st[m][6 + ax] = acc2;
      }
      wsync();
      return;
    }
    if (lane < 3 * N) {
      constexpr bool P0 = PART != 2, VA = PART != 1;
      const int ax = R.ax, m = R.kk + 1;
      double acc0 = 0, acc1 = 0, acc2 = 0;
      if constexpr (P0) acc0 = s.fr[ax][m][0];
      if constexpr (VA) acc1 = s.fr[ax][m][1], acc2 = s.fr[ax][m][2];
      const double* xx = s.x + ax * N;
      const double* g0 = &s.gz[ax][0][MAXH + m - 1];
      const double* g1 = &s.gz[ax][1][MAXH + m - 1];
      const double* g2 = &s.gz[ax][2][MAXH + m - 1];
#pragma unroll
      for (int k = 0; k < HT; ++k) {
        const double xk = xx[k];
        if constexpr (P0) acc0 += g0[-k] * xk;
        if constexpr (VA) acc1 += g1[-k] * xk, acc2 += g2[-k] * xk;
      }
      if constexpr (P0) s.st[m][ax] = acc0;
      if constexpr (VA) s.st[m][3 + ax] = acc1, s.st[m][6 + ax] = acc2;
    }
    wsync();
  }

  // The rule that picks the next row (Consts::pick_rule): every violated row (violation above `tol`) competes with the key
  // violation * weight, weight = 1 / sqrt(a^T Z a) of its normal (1 with pick_rule 0). (key, v, id) = the running best of a lane.
  struct Pick {
    double key, v;
    int id;
  };
  static __device__ __forceinline__ float plane_weight(const S& s, double nx, double ny, double nz, int m) {
    return mk_mw(s.kap, nx, ny, nz, m).w;
  }

  // violation of staged rows [lo, hi) -> running pick; four rows per trip, loads issued before first use
  // UN rows per lane and trip (64 UN rows per trip of the wavefront), loads issued before first use
  template <int UN>
  static __device__ __forceinline__ void scan_rows_n(const S& s, int lo, int hi, int lane, double tol, bool norm, Pick& pk, int stride) {
    for (int base = lo; base < hi; base += stride) {
      int idx[UN];
      MW mw[UN];
      D2 r01[UN], r23[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        idx[u] = base + 64 * u + lane;
        const int ii = idx[u] < hi ? idx[u] : lo;
        mw[u] = s.cand_mw[ii];
        r01[u] = *reinterpret_cast<const D2*>(&s.cand[ii][0]);
        r23[u] = *reinterpret_cast<const D2*>(&s.cand[ii][2]);
      }
      double px[UN], py[UN], pz[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const double* pm = s.st[mw[u].m];
        px[u] = pm[0], py[u] = pm[1], pz[u] = pm[2];
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const double vv = r01[u].x * px[u] + r01[u].y * py[u] + r23[u].x * pz[u] - r23[u].y;
        const double key = norm ? (double)((float)vv * mw[u].w) : vv;
        if (idx[u] < hi && vv > tol && key > pk.key) pk.key = key, pk.v = vv, pk.id = mk_kc(idx[u], mw[u].m);
      }
    }
  }
  // violation of staged rows [lo, hi) -> running pick. The operation is bound by the instructions its wavefront issues, and the
  // usual staging area holds 90-180 rows: one, two or four rows per lane, whatever covers [lo, hi) in one trip
  static __device__ __forceinline__ void scan_rows(const S& s, int lo, int hi, int lane, double tol, bool norm, Pick& pk,
                                                   int stride = 256) {
    const int cnt = hi - lo;
    if (stride == 256 && cnt <= 64) scan_rows_n<1>(s, lo, hi, lane, tol, norm, pk, 64);
    else if (stride == 256 && cnt <= 128) scan_rows_n<2>(s, lo, hi, lane, tol, norm, pk, 128);
    else scan_rows_n<4>(s, lo, hi, lane, tol, norm, pk, stride);
  }

  // Rows of the polyhedra assigned on the current branch: lane t < rows tests row t at p_i, lane rows + t at p_{i+1}, one
  // assigned step per trip. The assignment and the row counts are fetched ONCE (lane i holds assign[i], lane j the row count
  // of polyhedron j) and handed round with v_readlane, so a trip is one LDS round trip — row and point together — instead
  // of the three in a chain (assign[i] -> sp_rows[j] -> row) the step-by-step loop paid in every node of a tree.
  static __device__ __forceinline__ void scan_assigned(const S& s, int lane, int N, double tol, bool norm, Pick& pk, int pinned) {
    const int aj = lane < N ? s.assign[lane] : -1;
    const int nr = lane < S::PM ? s.sp_rows[lane] : 0;
    unsigned long long am = __ballot(aj >= 0);
    // One assigned step = at most 2 x 32 rows (both end points of the segment): one item per lane. Deep in a tree a dozen steps are
    // assigned and their rows were a chain of as many LDS round trips per pick; four steps go through a trip now, loads first.
    constexpr int UA = 4;
    while (am != 0ull) {
      int st[UA], id0[UA];
      bool on[UA];
      D2 a01[UA], a23[UA];
      double px[UA], py[UA], pz[UA];
#pragma unroll
      for (int u = 0; u < UA; ++u) {
        on[u] = false, st[u] = 1, id0[u] = 0;
        const double* row = s.sp[0][0];
        if (am != 0ull) {
          const int i = __ffsll((long long)am) - 1;
          am &= am - 1ull;
          const int j = __builtin_amdgcn_readlane(aj, i), rows = __builtin_amdgcn_readlane(nr, j);
          const int e = lane >= rows ? 1 : 0, r = lane - e * rows;
          on[u] = lane < 2 * rows && i + e > pinned;  // (rows on input-independent points only gate the choice: leaf_check)
          st[u] = on[u] ? i + e : 1, id0[u] = mk_id(K_P, (i << 7) | (e << 6) | r);
          row = s.sp[j][on[u] ? r : 0];
        }
        a01[u] = *reinterpret_cast<const D2*>(row), a23[u] = *reinterpret_cast<const D2*>(row + 2);
        const double* pm = s.st[st[u]];
        px[u] = pm[0], py[u] = pm[1], pz[u] = pm[2];
      }
#pragma unroll
      for (int u = 0; u < UA; ++u) {
        const double vv = a01[u].x * px[u] + a01[u].y * py[u] + a23[u].x * pz[u] - a23[u].y;
        if (on[u] && vv > tol) {
          const double key = norm ? (double)((float)vv * plane_weight(s, a01[u].x, a01[u].y, a23[u].x, st[u])) : vv;
          if (key > pk.key) pk.key = key, pk.v = vv, pk.id = id0[u];
        }
      }
    }
  }

  // the row that enters next among: input / state boxes, rows of assigned polyhedra, HOT staged rows (-1: nothing is violated)
  // MODE 0: all row families; 1: all but the state boxes; 2: the state boxes only (hdsm_wave_gib.h looks at those — and
  // evaluates the velocities and accelerations they bound — only when nothing else is violated)
  // SHARE: a long staging area is shared with the helper waves (two workgroup barriers; helper_loop below)
  template <int MODE = 0, bool SHARE = true>
  static __device__ __forceinline__ void select(S& s, const Consts& c, const Regs& R, int lane, double tol, int N,
                                                double& vbest, int& ibest, double* kbest = nullptr) {
    const bool norm = c.pick_rule != 0;
    Pick pk{0.0, 0.0, -1};
    auto offer = [&](double vv, float w, int id) {
      if (vv > tol) {
        const double key = norm ? (double)((float)vv * w) : vv;  // (single precision: a double copy of w, hoisted out of the loop, went to scratch)
        if (key > pk.key) pk.key = key, pk.v = vv, pk.id = id;
      }
    };
    // (a box: at most one side is violated, and with the same weight the larger violation wins anyway — ONE candidate per box)
    if (MODE != 2 && lane < c.n) {  // box of this lane's input (lane = variable; its axis is R.ax)
      const double vu = R.xi - s.bnd[3 + R.ax], vl = s.bnd[R.ax] - R.xi;
      offer(vl > vu ? vl : vu, R.wu, mk_id(K_U, (lane << 1) | (vl > vu ? 1 : 0)));
    }
#pragma unroll
    for (int e = 0; e < (MODE != 1 ? 2 : 0); ++e) {
      int sid = R.sb_id[e];
      keep_in_loop(sid);  // (the three LDS addresses below: hoisted out of the iteration loop they were spilled to scratch)
      if (sid >= 0) {
        const int ca = ((sid >> 3) & 3) * 3 + ((sid >> 1) & 3);  // 3 comp + axis
        const double sv = s.st[sid >> 5][ca];
        const double vu = sv - s.bnd[15 + ca], vl = s.bnd[6 + ca] - sv;
        offer(vl > vu ? vl : vu, R.sb_w[e], mk_id(K_S, sid | (vl > vu ? 1 : 0)));
      }
    }
    OP_PROF(8)
    if (MODE != 2 && (uni(s.level) | uni(s.forced)) != 0) scan_assigned(s, lane, N, tol, norm, pk, c.pinned_steps);  // rows of the polyhedra assigned on the current branch
    const int nc = MODE != 2 ? uni(s.ncand) : 0;
    const bool mw = SHARE && blockDim.x > 64 && nc > 256;  // worth waking the helper waves (two barriers)
    if (mw) {
      if (lane == 0) s.cmd = 1, s.part_tol = tol, s.part_norm = norm ? 1 : 0;
      __syncthreads();                             // helpers start on their share: rows [256 w, ...) stride 256 * waves
      scan_rows(s, 0, nc, lane, tol, norm, pk, 4 * (int)blockDim.x);
    } else {
      scan_rows(s, 0, nc, lane, tol, norm, pk);
    }
    OP_PROF(9)
    double m = wave_max64(pk.key);
    double mv = 0.0;
    int best = -1;
    if (m > 0.0) {
      const int src = __ffsll((long long)__ballot(pk.key == m && pk.id >= 0)) - 1;
      best = __builtin_amdgcn_readlane(pk.id, src);
      mv = bcast64(pk.v, src);
    }
    if (mw) {
      __syncthreads();                             // partial results of waves 1..3 are in LDS
#pragma unroll
      for (int w = 1; w < (int)blockDim.x >> 6; ++w) {
        const double pkey = s.part_key[w];
        if (pkey > m) m = pkey, mv = s.part_v[w], best = s.part_id[w];
      }
    }
    vbest = mv;
    ibest = (m > 0.0) ? best : -1;
    if (kbest != nullptr) *kbest = m;  // the winner's key: violation / sqrt(a^T Z a) with the normalised rule
  }

  // No hot row is violated: check the COLD staged rows (kept at the top of the staging area, scanned only
  // here) and promote the violated ones into the hot list. Returns the number promoted.
  static __device__ __forceinline__ int promote_cold(S& s, int lane, double tol) {
    const int ncold = uni(s.ncold);
    if (ncold == 0) return 0;
    const int before = uni(s.ncand);
    for (int idx = CMAX - ncold + lane; idx < CMAX; idx += 64) {
      const double* row = s.cand[idx];
      const double* pm = s.st[s.cand_mw[idx].m];
      const double vv = row[0] * pm[0] + row[1] * pm[1] + row[2] * pm[2] - row[3];
      if (vv > tol) {
        const int slot = atomicAdd(&s.ncand, 1);
        if (slot < CMAX - ncold) {
          s.cand[slot][0] = row[0], s.cand[slot][1] = row[1], s.cand[slot][2] = row[2], s.cand[slot][3] = row[3];
          s.cand_mw[slot] = s.cand_mw[idx];
          s.cand[idx][3] = DINF;  // neutralise the cold copy: it can never be violated again
        } else {
          s.overflow = 1;
        }
      }
    }
    wsync();
    if (lane == 0 && s.ncand > CMAX - ncold) s.ncand = CMAX - ncold;
    wsync();
    return uni(s.ncand) - before;
  }

  // entry `lane` of the dense normal a of constraint id (a . u <= rhs form)
  // (`var` = row_of(lane); its axis and step come from the registers)
  static __device__ __forceinline__ double normal_entry(const S& s, const Regs& R, int id, int var, int N, int n) {
    const int kind = id_kind(id), p = id_payload(id);
    if (var >= n) return 0.0;
    const int lane = var, ax = R.ax, kk = R.kk;
    if (kind == K_U) return (lane == (p >> 1)) ? ((p & 1) ? -1.0 : 1.0) : 0.0;
    if (kind == K_S) {
      const int cax = (p >> 1) & 3, comp = (p >> 3) & 3, m = p >> 5;
      const double sg = (p & 1) ? -1.0 : 1.0;
      return (ax == cax && kk < m) ? sg * s.gz[ax][comp][MAXH + m - 1 - kk] : 0.0;
    }
    if (kind == K_E) {
      const int cax = p % 3, comp = 1 + p / 3;
      return (ax == cax) ? s.gz[ax][comp][MAXH + N - 1 - kk] : 0.0;
    }
    const double* row;
    int m;
    if (kind == K_P) {
      const int r = p & 63, e = (p >> 6) & 1, i = p >> 7;
      row = s.sp[s.assign[i]][r];
      m = i + e;
    } else {
      row = s.cand[kc_slot(p)];
      m = kc_m(p);
    }
    // (the lane's axis as a value formed HERE: the byte offsets 8 ax / 24 MAXH ax that the two reads below add were hoisted in front of
    // the branch-and-bound loop as loop invariants and, in the two-per-CU kernel, spilled — its only scratch access)
    int axl = ax;
    keep_in_loop(axl);
    return (kk < m) ? row[axl] * s.gz[axl][0][MAXH + m - 1 - kk] : 0.0;
  }

  static __device__ __forceinline__ double resid(const S& s, const Consts& c, int id, int N) {
    const int kind = id_kind(id), p = id_payload(id);
    if (kind == K_U) {
      const int var = p >> 1, ax = var / N;
      return (p & 1) ? (s.bnd[ax] - s.x[var]) : (s.x[var] - s.bnd[3 + ax]);
    }
    if (kind == K_S) {
      const int sg = p & 1, ax = (p >> 1) & 3, comp = (p >> 3) & 3, i = p >> 5;
      const double v = s.st[i][3 * comp + ax];
      return sg ? (s.bnd[6 + 3 * comp + ax] - v) : (v - s.bnd[15 + 3 * comp + ax]);
    }
    if (kind == K_E) return s.st[N][3 * (1 + p / 3) + p % 3];
    const double* row;
    const double* pm;
    if (kind == K_P) {
      const int r = p & 63, e = (p >> 6) & 1, i = p >> 7;
      row = s.sp[s.assign[i]][r];
      pm = s.st[i + e];
    } else {
      row = s.cand[kc_slot(p)];
      pm = s.st[kc_m(p)];
    }
    return row[0] * pm[0] + row[1] * pm[1] + row[2] * pm[2] - row[3];
  }

  // ---- neighbour sweep, device version -----------------------------------------------------------------------
  // 1 / sqrt(x) to double precision: v_rsq_f64 seed + two Newton steps (no v_sqrt / v_rcp expansion per plane)
  static __device__ __forceinline__ double rsqrt_nr(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    return y;
  }

  // Same planes as tasc_plane_eval (AC:1100-1205), organised for the workgroup: thread <-> (neighbour, step) pair, so
  // the dependent f64 chain of one plane (two reciprocal square roots) is walked once per thread and not N times.
  // For large swarms (a.bounds) the neighbours first pass a sphere test and only the survivors form pairs.
  // SOLO: ONE wavefront sweeps (wave 1 of a two-wave workgroup, while wave 0 installs the warm start: hdsm_core.h) — `lane` is the lane
  // of that wavefront, the barriers are wave-local. `pts` / `pstride`: the trajectory points the slacks are evaluated at (Shm::st with
  // stride 9, or the reference points of the displacement test, Shm::sw_ref, stride 3).
  template <bool SOLO = false>
  static __device__ __forceinline__ void sweep_planes(S& s, const Consts& c, const Args& a, int self, double thresh,
                                                      bool check_fixed, int lane, const double* pts = nullptr, int pstride = 9) {
    const int N = c.N, n_rob = a.n_rob, nt = SOLO ? 64 : (int)blockDim.x;  // lane = thread of the WORKGROUP (all waves sweep) unless SOLO
    if (pts == nullptr) pts = &s.st[0][0];
    const GPtr<const double> g_bounds = gptr(a.bounds), g_pos = gptr(a.pos);
    const GPtr<const uint8_t> g_has = gptr(a.has_plan);
    auto bar = [&]() {
      if constexpr (SOLO) wsync();
      else __syncthreads();
    };
    PROF_DECL
    const double radius = c.radius, k2m1 = c.k2m1, pert = c.pert, tol = c.tol, hot_tau = c.hot_tau;
    const int pinned = c.pinned_steps;
    const bool pre = a.bounds != nullptr;
    // Rigorous cull. With p = c + delta:  slack(p) = n_f.(q - p) = |d|/2 - back - n_f.delta  and n_f.n = 1,
    // back <= s_max = max(r, h), |n_f| <= sqrt(1 + (3 pert)^2)  ==>  slack >= |d|/2 - s_max - |n_f| |delta|.
    // A neighbour with |d| >= 2 (max(thresh, tol) + s_max + |n_f| delta_max) can be neither staged nor violated.
    // Sphere prefilter: |o_i - c_i| >= |C_k - C_self| - rho_k - rho_self for every step i, so a neighbour whose
    // sphere is further than cull + rho_k + rho_self from ours is skipped without touching its plan.
    if (lane < 64) {  // wave 0: delta_max^2 over the 2N (step, endpoint) pairs and the own sphere, lane-parallel
      double d2 = 0.0, r2 = 0.0;
      const double mx = 0.5 * (s.cprev[0][0] + s.cprev[N - 1][0]), my = 0.5 * (s.cprev[0][1] + s.cprev[N - 1][1]),
                   mz = 0.5 * (s.cprev[0][2] + s.cprev[N - 1][2]);
      if (lane < 2 * N) {
        const int i = lane >> 1, m = i + (lane & 1);
        const double* pm0 = pts + m * pstride;
        const double ux = pm0[0] - s.cprev[i][0], uy = pm0[1] - s.cprev[i][1], uz = pm0[2] - s.cprev[i][2];
        d2 = ux * ux + uy * uy + uz * uz;
        const double wx = s.cprev[i][0] - mx, wy = s.cprev[i][1] - my, wz = s.cprev[i][2] - mz;
        r2 = wx * wx + wy * wy + wz * wz;
      }
      d2 = wave_max64(d2), r2 = wave_max64(r2);
      if (lane == 0) {
        const double smax = radius * fmax(1.0, rsqrt_nr(1.0 + k2m1));  // max(r, h): h = r / sqrt(1 + k2m1)
        const double nfmax = sqrt(1.0 + 9.0 * pert * pert);
        s.sw[0] = 2.0 * (fmax(thresh, tol) + smax + nfmax * sqrt(d2)) * (1.0 + 1e-9);
        s.sw[1] = mx, s.sw[2] = my, s.sw[3] = mz, s.sw[4] = sqrt(r2) * (1.0 + 1e-9);
        s.nlist = 0;
      }
    }
    bar();
    const double cull = s.sw[0], cull2 = cull * cull;
    SW_PROF(8)
    const int chunk = pre ? S::LC : n_rob;
    for (int base = 0; base < n_rob; base += chunk) {
      const int end = (base + chunk < n_rob) ? base + chunk : n_rob;
      int cnt = end - base;
      if (pre) {
        const double sx = s.sw[1], sy = s.sw[2], sz = s.sw[3], reach0 = cull + s.sw[4];
        constexpr int FB = 4;  // sphere records in flight per thread
        for (int k0 = base + lane; k0 < end; k0 += FB * nt) {
          double4 bk[FB];
#pragma unroll
          for (int f = 0; f < FB; ++f) {
            const int k = k0 + f * nt;
            bk[f] = *(GPtr<const double4>)(g_bounds + 4 * (int64_t)(k < end ? k : base));
          }
#pragma unroll
          for (int f = 0; f < FB; ++f) {
            const int k = k0 + f * nt;
            const double ux = bk[f].x - sx, uy = bk[f].y - sy, uz = bk[f].z - sz, reach = reach0 + bk[f].w;
            if (k < end && k != self && bk[f].w >= 0 && ux * ux + uy * uy + uz * uz < reach * reach)
              s.list[atomicAdd(&s.nlist, 1)] = k;
          }
        }
        bar();
        cnt = s.nlist;
        SW_PROF(9)
      }
      const int total = cnt * N;
      if (lane == 0) s.st_sph += pre ? (end - base) : 0, s.st_pairs += total;
      for (int idx0 = 0; idx0 < total; idx0 += nt) {
        const int idx = idx0 + lane;
        const bool in = idx < total;
        const int j = in ? idx / N : 0, i = in ? idx - j * N : 0;
        const int k = in ? (pre ? s.list[j] : base + j) : 0;  // idle threads read agent 0's record (always there)
        // packed positions [n_rob][N][3]: consecutive threads of a neighbour read consecutive 24-B records
        const GPtr<const double> op = g_pos + ((int64_t)k * N + i) * 3;
        const double ox = op[0], oy = op[1], oz = op[2];
        const bool on = in && k != self && (pre || g_has[k]);
        SW_PROF(10)
        const double cx = s.cprev[i][0], cy = s.cprev[i][1], cz = s.cprev[i][2];
        const double dx = ox - cx, dy = oy - cy, dz = oz - cz;
        const double n2 = dx * dx + dy * dy + dz * dz;
        if (on && n2 > 0 && n2 < cull2) {  // else absent / coincident (row 0.p <= 0) / provably slack
          const double inv = rsqrt_nr(n2), nrm = n2 * inv;
          const double hx = dx * inv, hy = dy * inv, hz = dz * inv;
          const double sd = radius * rsqrt_nr(1.0 + k2m1 * hz * hz);  // ellipsoid support distance
          const double back = 0.5 * fmin(2.0 * sd, nrm);
          const double qx = 0.5 * (cx + ox) - back * hx, qy = 0.5 * (cy + oy) - back * hy,
                       qz = 0.5 * (cz + oz) - back * hz;
          const double fx = hx + pert * (hy - hz) - pert * hz, fy = hy - pert * hx, fz = hz + 2.0 * pert * hx;
          const double rhs = fx * qx + fy * qy + fz * qz;
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int m = i + e;
            const double* pm = pts + m * pstride;
            const double v = fx * pm[0] + fy * pm[1] + fz * pm[2] - rhs;
            if (m <= pinned) {  // a constant row (hdsm_core.h): within ftol_fixed it holds, beyond it nothing can satisfy it
              if (check_fixed && v > c.ftol_fixed) s.fixed_bad = 1;
              continue;
            }
            if (v > tol) s.nviol = 1;
            if (-v < thresh) {
              const bool hot = -v < hot_tau;
              int slot;
              const bool fits = stage_slot<CMAX>(s.ncand, s.ncold, hot, slot);
              if (fits) {
                s.cand[slot][0] = fx, s.cand[slot][1] = fy, s.cand[slot][2] = fz, s.cand[slot][3] = rhs;
                s.cand_mw[slot] = mk_mw(s.kap, fx, fy, fz, m);
                s.cand_src[slot] = (k << 6) | (i << 1) | e;
              } else {
                s.overflow = 1;
              }
            }
          }
        }
        SW_PROF(11)
      }
      if (pre && end < n_rob) {  // next chunk reuses the list
        bar();
        if (lane == 0) s.nlist = 0;
        bar();
      }
    }
    bar();
    if (lane == 0 && s.ncand + s.ncold > CMAX) {  // rows that found no slot advanced the counters past the capacity: clamp
      s.overflow = 1;                             // (no slot is written twice, so everything below the clamped counts is complete)
      s.wanted_raw = s.ncand + s.ncold;
      const int cold = s.ncold < CMAX ? s.ncold : CMAX;
      if (s.ncand > CMAX - cold) s.ncand = CMAX - cold;
      s.ncold = cold;
    }
    bar();
    SW_PROF(12)
  }

  // ---- warm start ---------------------------------------------------------------------------------------
  // (implementation: hdsm_wave_gib.h, for both register layouts)
  // The optimal working set of the previous replan of this instance (a.warm, portable ids), moved one step
  // towards the present, seeds the dual method: its rows are put into the factorisation WITHOUT taking steps,
  // the minimiser x_W on them and its multipliers follow in closed form
  //      t = U^T v,   lambda = U t,   x_W = x0 + J1 t,   f_W = f(x0) + |t|^2 / 2        (v = violations at x0)
  // and entries with a negative multiplier are dropped until (x_W, W) is a valid S-pair. The regular loop then
  // continues from there; the result is the same optimum, reached in fewer iterations.
  // size of a snapshot of the solver state (hdsm_wave_gib.h writes it: J slots, U rows, multipliers, ids, x, f, q)
  static constexpr int SNAP_DOUBLES = (2 * NV + 3) * NV + 2;
};

}  // namespace hdsm
