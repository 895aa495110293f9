// hdsm_wave_gi.h — device-only (gfx950): the dual active-set iteration of ONE instance executed by ONE
// 64-lane wavefront with the factorisation held in REGISTERS.
//
// Same mathematics as Solver::gi_run in hdsm_core.h (Goldfarb-Idnani, J = L^{-T} Q with J^T N = [R; 0]),
// different data layout, chosen for CDNA4:
//   * the rows of J live in statically indexed registers: with n <= 30 (NV = 32) every row is split over TWO lanes
//     (lane L: columns [16h, 16h+16) of row L & 31, h = L >> 5; partial sums meet through v_permlane32_swap), with
//     larger n (NV = 48) lane i owns row i; U = R^{-1} is in LDS by rows; the multiplier / id of the working-set
//     entry at position k live in lane k;
//   * r = U d1 is a lane-local dot product: no back-substitution chain;
//   * d = J^T a is a transposition through LDS (T[j][i] = J[i][j] a_i, conflict-free strides), or a single
//     row broadcast when the incoming row is an input bound (a = +-e_k, the common case in bang-bang plans);
//   * "add" is ONE Householder reflection of the free columns q..NV-1 of J that maps d2 onto rho e_q: a
//     rank-1 update J2 -= (J2 v) beta v^T with v = d2 - rho e_q. J2 v = z - rho J[:,q] reuses the primal
//     direction z that the step needs anyway, so the update costs one FMA per free column, one sqrt and
//     one division per ITERATION (not per lane) and no coefficient exchange; U gets the column (-r/rho; 1/rho).
//     R = U^{-1} stays a general invertible matrix (it need not be triangular for the method);
//   * "drop l": any orthogonal G with G u_l^T = sigma e_{q-1} (u_l = row l of U) gives
//     U' = (E^T U G^T)[:, :q-1]; G is again one Householder reflection: rank-1 updates of the rows of U (LDS)
//     and of the columns 0..q-1 of J (registers), then the rows > l of U are written one slot up;
//   * cross-lane reductions use DPP row rotations + v_readlane, not LDS trees.
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

// an integer the optimiser must treat as recomputed here (no effect on the value): stops it from hoisting what depends on it
__device__ __forceinline__ void keep_in_loop(int& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(v));
#else
  (void)v;
#endif
}

// threadIdx.x as a value formed HERE: the thread-indexed LDS addresses of the all-thread loops (PAR_FOR) are then computed where
// they are used; hoisted out of the branch-and-bound loop they stayed alive across the active-set run and were spilled
__device__ __forceinline__ int tid_here() {
  int t = (int)threadIdx.x;
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

#ifdef HDSM_PROFILE
// cycle counters accumulated in LDS by lane 0 (keeps them out of the SGPR file)
#define PROF_DECL if (threadIdx.x == 0) s.prof_last = clock64();
#define PROF(k) if (threadIdx.x == 0) { const long long now_ = clock64(); s.prof_acc[k] += now_ - s.prof_last; s.prof_last = now_; }
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
#if defined(HDSM_PROFILE) && defined(HDSM_PROF_OP)
#define SW_PROF(k)
#define ST_PROF(k)
#define WS_PROF(k)
#define OP_PROF(k) PROF(k)
#define SC_PROF_DECL if (threadIdx.x == 64) s.prof_last1 = clock64();
#define SC_PROF(k) if (threadIdx.x == 64) { const long long now_ = clock64(); s.prof_acc[k] += now_ - s.prof_last1; s.prof_last1 = now_; }
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

  // register array access with a dynamic index (select chain); an index outside [0, NC) reads 0 / writes nothing
  static __device__ __forceinline__ double reg_get(const double (&a)[NC], int q) {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < NC; ++k) v = (k == q) ? a[k] : v;
    return v;
  }
  static __device__ __forceinline__ void reg_set(double (&a)[NC], int q, double v) {
#pragma unroll
    for (int k = 0; k < NC; ++k) a[k] = (k == q) ? v : a[k];
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
    while (am != 0ull) {
      const int i = __ffsll((long long)am) - 1;
      am &= am - 1ull;
      const int j = __builtin_amdgcn_readlane(aj, i), rows = __builtin_amdgcn_readlane(nr, j);
      for (int t = lane; t < 2 * rows; t += 64) {
        const int e = t >= rows ? 1 : 0, r = t - e * rows;
        if (i + e <= pinned) continue;  // (rows on input-independent points only gate the choice: leaf_check)
        const double* row = s.sp[j][r];
        const double* pm = s.st[i + e];
        const double vv = row[0] * pm[0] + row[1] * pm[1] + row[2] * pm[2] - row[3];
        if (vv > tol) {
          const double key = norm ? (double)((float)vv * plane_weight(s, row[0], row[1], row[2], i + e)) : vv;
          if (key > pk.key) pk.key = key, pk.v = vv, pk.id = mk_id(K_P, (i << 7) | (e << 6) | r);
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
    if (MODE != 2 && uni(s.level) > 0) scan_assigned(s, lane, N, tol, norm, pk, c.pinned_steps);  // rows of the polyhedra assigned on the current branch
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

  // The other waves of the workgroup while wave 0 iterates: wait for a command at the workgroup barrier, scan a share of the staged rows.
  static __device__ __forceinline__ void helper_loop(S& s, const Consts&, Regs&) {
    const int lane = (int)threadIdx.x & 63, w = (int)threadIdx.x >> 6;
    for (;;) {
      __syncthreads();
      if (uni(s.cmd) == 0) return;
      Pick pk{0.0, 0.0, -1};
      scan_rows(s, 256 * w, uni(s.ncand), lane, s.part_tol, s.part_norm != 0, pk, 4 * (int)blockDim.x);
      const double m = wave_max64(pk.key);
      int best = -1;
      double mv = 0.0;
      if (m > 0.0) {
        const int src = __ffsll((long long)__ballot(pk.key == m && pk.id >= 0)) - 1;
        best = __builtin_amdgcn_readlane(pk.id, src);
        mv = bcast64(pk.v, src);
      }
      if (lane == 0) s.part_key[w] = m, s.part_v[w] = mv, s.part_id[w] = best;
      __syncthreads();
    }
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
    return (kk < m) ? row[ax] * s.gz[ax][0][MAXH + m - 1 - kk] : 0.0;
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

  // d = J^T (-a) -> s.dvec (all of d) and s.dvz (d with the entries of the working-set columns, j < q, zeroed);
  // returns d_lane for lane < NV, 0 beyond. `ai` is entry row_of(lane) of the normal.
  static __device__ __forceinline__ double compute_d(S& s, const Regs& R, int id, double ai, int q, int lane) {
    const int row = row_of(lane), c0 = col0_of(lane);
    double dj = 0.0;
    if (id_kind(id) == K_U) {  // a = sg e_k: d = -sg * (row k of J)
      const int var = id_payload(id) >> 1;
      const double msg = (id_payload(id) & 1) ? 1.0 : -1.0;
      if (row_ok(lane) && row == var) {
#pragma unroll
        for (int j = 0; j < NC; j += 2) *reinterpret_cast<D2*>(&s.dvec[c0 + j]) = D2{msg * R.Jr[j], msg * R.Jr[j + 1]};
      }
      wsync();
      if (lane < NV) dj = s.dvec[lane], s.dvz[lane] = (lane >= q) ? dj : 0.0;
      wsync();
      return dj;
    }
    if (row_ok(lane)) {
#pragma unroll
      for (int j = 0; j < NC; ++j) s.T[(c0 + j) * LDT + row] = R.Jr[j] * ai;
    }
    wsync();
    {
      double a0 = 0, a1 = 0;
      if (row_ok(lane)) {  // entries [c0, c0 + NC) of row `row` of T
        const D2* tr = reinterpret_cast<const D2*>(&s.T[row * LDT + c0]);
#pragma unroll
        for (int i = 0; i < NC / 2; ++i) {
          const D2 t = tr[i];
          a0 += t.x;
          a1 += t.y;
        }
      }
      dj = -hsum(a0 + a1);
    }
    if (lane < NV) s.dvec[lane] = dj, s.dvz[lane] = (lane >= q) ? dj : 0.0;
    else dj = 0.0;
    wsync();
    return dj;
  }

  // d = J^T(-a), ||d||^2, ||d2||^2, d_q, z_i = (J2 d2)_i, r_i = (U d1)_i for the incoming constraint. dv = this lane's
  // columns of d (split kernel: with the working-set part zeroed — what the Householder update multiplies).
  static __device__ __forceinline__ void direction(S& s, const Regs& R, int id, double ai, int q, int lane, double (&dv)[NC],
                                                   double& dd, double& zz, double& dq, double& zi, double& ri) {
    const double dj = compute_d(s, R, id, ai, q, lane);
    const double sufj = wave_suffix_sum(dj * dj, lane);  // sum_{k >= lane} d_k^2
    dd = bcast64(sufj, 0);
    zz = (q < NV) ? bcast64(sufj, q) : 0.0;
    dq = (q < NV) ? bcast64(dj, q) : 0.0;
    const int row = row_of(lane), c0 = col0_of(lane);
    double r0 = 0, r1 = 0, z0 = 0, z1 = 0;
    if constexpr (SPLIT) {
      const D2* urow = reinterpret_cast<const D2*>(&s.U[row * LDT + c0]);
#pragma unroll
      for (int k = 0; k < NC; k += 2) {
        const D2 dk = *reinterpret_cast<const D2*>(&s.dvec[c0 + k]);
        const D2 zk = *reinterpret_cast<const D2*>(&s.dvz[c0 + k]);
        const D2 uk = urow[k / 2];
        dv[k] = zk.x, dv[k + 1] = zk.y;
        r0 += uk.x * dk.x, r1 += uk.y * dk.y;                  // r = U d1: U has zero columns >= q, no mask needed
        z0 += R.Jr[k] * zk.x, z1 += R.Jr[k + 1] * zk.y;        // z = J2 d2: dvz is zero on the other columns
      }
    } else {  // one lane per row: q is wave-uniform against the column index, the free columns are picked by predicate
#pragma unroll
      for (int k = 0; k < NC; k += 2) {
        const D2 dk = *reinterpret_cast<const D2*>(&s.dvec[k]);
        dv[k] = dk.x, dv[k + 1] = dk.y;
      }
      if (lane < NV) {
        const D2* urow = reinterpret_cast<const D2*>(&s.U[lane * LDT]);
#pragma unroll
        for (int k = 0; k < NC; k += 2) {
          const D2 uk = urow[k / 2];
          r0 += uk.x * dv[k];
          r1 += uk.y * dv[k + 1];
        }
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          if (k >= q) {
            if (k & 1) z1 += R.Jr[k] * dv[k];
            else z0 += R.Jr[k] * dv[k];
          }
        }
      }
    }
    ri = hsum(r0 + r1);
    zi = hsum(z0 + z1);
  }

  // working set += id at position q: ONE Householder reflection of the free columns, d2 -> rho e_q
  static __device__ __forceinline__ void householder_add(S& s, Regs& R, int id, double lam_p, int q, int lane,
                                                         const double (&dv)[NC], double zz, double dq, double zi, double ri) {
    const double rho = (dq > 0 ? -1.0 : 1.0) * sqrt(zz);
    const double beta = 1.0 / (rho * (rho - dq));
    const int c0 = col0_of(lane);
    const double jq = hsum(row_ok(lane) ? reg_get(R.Jr, q - c0) : 0.0);  // J[row][q]: held by one of the halves
    if (row_ok(lane)) {
      const double coef = (zi - rho * jq) * beta;  // (J2 v) beta, v = d2 - rho e_q
#pragma unroll
      for (int k = 0; k < NC; ++k)
        if (SPLIT || k >= q) R.Jr[k] -= coef * dv[k];  // (split kernel: dv is already zero on the working-set columns)
      reg_set(R.Jr, q - c0, jq - coef * (dq - rho));
    }
    if (lane < NV) s.U[lane * LDT + q] = (lane < q) ? -ri / rho : ((lane == q) ? 1.0 / rho : 0.0);
    if (lane == q) {
      s.lam[q] = lam_p;
      s.act[q] = id;
    }
    wsync();
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
  static __device__ __forceinline__ void sweep_planes(S& s, const Consts& c, const Args& a, int self, double thresh,
                                                      bool check_fixed, int lane) {
    const int N = c.N, n_rob = a.n_rob, nt = (int)blockDim.x;  // lane = thread of the WORKGROUP here (all waves sweep)
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
        const double ux = s.st[m][0] - s.cprev[i][0], uy = s.st[m][1] - s.cprev[i][1], uz = s.st[m][2] - s.cprev[i][2];
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
    __syncthreads();
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
            bk[f] = *reinterpret_cast<const double4*>(a.bounds + 4 * (int64_t)(k < end ? k : base));
          }
#pragma unroll
          for (int f = 0; f < FB; ++f) {
            const int k = k0 + f * nt;
            const double ux = bk[f].x - sx, uy = bk[f].y - sy, uz = bk[f].z - sz, reach = reach0 + bk[f].w;
            if (k < end && k != self && bk[f].w >= 0 && ux * ux + uy * uy + uz * uz < reach * reach)
              s.list[atomicAdd(&s.nlist, 1)] = k;
          }
        }
        __syncthreads();
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
        const double* op = a.pos + ((int64_t)k * N + i) * 3;
        const double ox = op[0], oy = op[1], oz = op[2];
        const bool on = in && k != self && (pre || a.has_plan[k]);
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
            const double* pm = s.st[m];
            const double v = fx * pm[0] + fy * pm[1] + fz * pm[2] - rhs;
            if (m <= pinned) {  // a constant row (hdsm_core.h): within ftol_fixed it holds, beyond it nothing can satisfy it
              if (check_fixed && v > c.ftol_fixed) s.fixed_bad = 1;
              continue;
            }
            if (v > tol) s.nviol = 1;
            if (-v < thresh) {
              const bool hot = -v < hot_tau;
              const int slot = hot ? atomicAdd(&s.ncand, 1) : CMAX - 1 - atomicAdd(&s.ncold, 1);
              const bool fits = hot ? slot < CMAX - s.ncold : slot >= s.ncand;
              if (fits && slot >= 0 && slot < CMAX) {
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
        __syncthreads();
        if (lane == 0) s.nlist = 0;
        __syncthreads();
      }
    }
    __syncthreads();
    if (lane == 0 && s.ncand + s.ncold > CMAX) {  // rows that found no slot advanced the counters past the capacity: clamp
      s.overflow = 1;                             // (no slot is written twice, so everything below the clamped counts is complete)
      s.wanted_raw = s.ncand + s.ncold;
      const int cold = s.ncold < CMAX ? s.ncold : CMAX;
      if (s.ncand > CMAX - cold) s.ncand = CMAX - cold;
      s.ncold = cold;
    }
    __syncthreads();
    SW_PROF(12)
  }

  // ---- warm start ---------------------------------------------------------------------------------------
  // The optimal working set of the previous replan of this instance (a.warm, portable ids), moved one step
  // towards the present, seeds the dual method: its rows are put into the factorisation WITHOUT taking steps,
  // the minimiser x_W on them and its multipliers follow in closed form
  //      t = U^T v,   lambda = U t,   x_W = x0 + J1 t,   f_W = f(x0) + |t|^2 / 2        (v = violations at x0)
  // and entries with a negative multiplier are dropped until (x_W, W) is a valid S-pair. The regular loop then
  // continues from there; the result is the same optimum, reached in fewer iterations.
  static __device__ __forceinline__ void warm_start(S& s, const Consts& c, const Args& a, Regs& R, int inst, int self,
                                                    int& iters, const WarmPre& wpre) {
    const int lane = (int)threadIdx.x;
    const int N = c.N, n = c.n;
    int nw = uni(wpre.head) & ~WARM_CERT;
    if (nw <= 0) return;
    if (nw > NV) nw = NV;
    PROF_DECL
    // Lane g prepares entry g of the guess — translation of the id to this replan's indices and, for a neighbour row,
    // the global reads and the plane itself — so the sequential loop below only broadcasts (v_readlane) what it needs.
    int pre = -1, my_m = 0, my_src = 0;  // pre: >= 0 a ready id, -2 a neighbour row held in my_row, -1 nothing usable
    double my_row[4] = {0.0, 0.0, 0.0, 0.0};
    if (lane < nw) {
      const int code = wpre.code;
      const int kind = id_kind(code), p = id_payload(code);
      if (kind == K_U) {
        const int var = p >> 1;
        if (var % N >= 1) pre = mk_id(K_U, ((var - 1) << 1) | (p & 1));
      } else if (kind == K_S) {
        const int i = p >> 5;
        if (i - 1 >= 1) pre = mk_id(K_S, ((i - 1) << 5) | (p & 31));
      } else if (kind == K_C && a.l1_rows == nullptr) {
        const int e = p & 1, i = ((p >> 1) & 31) - 1, k = p >> 6;
        if (i >= 0 && i + e > c.pinned_steps && k != self && k < a.n_rob && wpre.has) {
          const double op[3] = {wpre.ox, wpre.oy, wpre.oz};
          if (tasc_plane_eval(c, s.cprev[i], op, my_row)) pre = -2, my_m = i + e, my_src = (k << 6) | (i << 1) | e;
        }
      }
    }
    int q = uni(s.q);
    WS_PROF(16)
    for (int g = 0; g < nw && q < n; ++g) {
      int id = __builtin_amdgcn_readlane(pre, g);
      if (id == -2) {
        id = -1;
        const int slot = uni(s.ncand);
        if (slot < CMAX - uni(s.ncold)) {
          const double r0 = bcast64(my_row[0], g), r1 = bcast64(my_row[1], g), r2 = bcast64(my_row[2], g),
                       r3 = bcast64(my_row[3], g);
          const int m = __builtin_amdgcn_readlane(my_m, g), src = __builtin_amdgcn_readlane(my_src, g);
          if (lane == 0) {
            s.cand[slot][0] = r0, s.cand[slot][1] = r1, s.cand[slot][2] = r2, s.cand[slot][3] = r3;
            s.cand_mw[slot] = mk_mw(s.kap, r0, r1, r2, m);
            s.cand_src[slot] = src;
            s.ncand = slot + 1;
          }
          wsync();
          id = mk_kc(slot, m);
        }
      }
      if (id < 0) continue;
      WS_PROF(17)
      const double ai = normal_entry(s, R, id, row_of(lane), N, n);
      WS_PROF(18)
      double dv[NC], dd, zz, dq, zi, ri;
      direction(s, R, id, ai, q, lane, dv, dd, zz, dq, zi, ri);
      WS_PROF(19)
      ++iters;
      if (!(zz > 1e-8 * dd)) continue;  // (nearly) dependent on what is already in: leave it out
      householder_add(s, R, id, 0.0, q, lane, dv, zz, dq, zi, ri);
      WS_PROF(20)
      ++q;
    }
    if (q == uni(s.q)) return;  // nothing usable
    // violations of the working-set rows at the unconstrained minimiser x0
    if (lane < NV) s.x[lane] = s.x0[lane];
    wsync();
    states(s, R, lane, N);
    for (;;) {
      const double vk = (lane < q) ? resid(s, c, s.act[lane], N) : 0.0;
      if (lane < NV) s.dvec[lane] = vk;
      wsync();
      const int row = row_of(lane), c0 = col0_of(lane);
      double tj;  // t = U^T v: column `row` of U, this lane's share of the rows (rows >= q of U are zero)
      {
        double p0 = 0, p1 = 0;
        if (row_ok(lane)) {
#pragma unroll
          for (int k = 0; k < NC; k += 2) {
            const D2 vk2 = *reinterpret_cast<const D2*>(&s.dvec[c0 + k]);
            p0 += s.U[(c0 + k) * LDT + row] * vk2.x, p1 += s.U[(c0 + k + 1) * LDT + row] * vk2.y;
          }
        }
        tj = hsum(p0 + p1);
      }
      wsync();
      if (lane < NV) s.dvec[lane] = tj;
      else tj = 0.0;
      wsync();
      double lk, xw = 0.0;
      {
        double l0 = 0, l1 = 0, x0 = 0, x1 = 0;
        if (row_ok(lane)) {
          const D2* urow = reinterpret_cast<const D2*>(&s.U[row * LDT + c0]);
#pragma unroll
          for (int k = 0; k < NC; k += 2) {
            const D2 tk = *reinterpret_cast<const D2*>(&s.dvec[c0 + k]);
            const D2 uk = urow[k / 2];
            l0 += uk.x * tk.x, l1 += uk.y * tk.y;               // lambda = U t
            x0 += R.Jr[k] * tk.x, x1 += R.Jr[k + 1] * tk.y;     // J1 t  (t is zero beyond q)
          }
        }
        lk = hsum(l0 + l1);
        const double xs = hsum(x0 + x1);
        if (lane < NV) xw = s.x0[lane] + xs;
      }
      // most negative multiplier among the inequalities
      const bool ineq = lane < q && id_kind(s.act[lane]) != K_E;
      const double worst = -wave_max64(ineq ? -lk : -DINF);
      if (!(worst < -1e-12) || q <= 6) {
        const double tt = wave_suffix_sum(tj * tj, lane);
        if (lane < NV) {
          s.lam[lane] = (lane < q) ? lk : 0.0;
          if (lane < n) R.xi = xw, s.x[lane] = xw;
        }
        if (lane == 0) s.f = s.fx0 + 0.5 * tt, s.q = q;  // (lane 0's suffix sum is the whole sum)
        wsync();
        WS_PROF(21)
#ifdef HDSM_DEBUG
        states(s, R, lane, N);
        if (blockIdx.x == 2 && lane < q)
          printf("WS lane %d q %d nw %d act %x lam %.4e resid@xW %.3e f %.6e fx0 %.6e\n", lane, q, nw, s.act[lane], lk,
                 resid(s, c, s.act[lane], N), s.f, s.fx0);
#endif
        return;
      }
      const int l = uni(__ffsll((long long)__ballot(ineq && lk == worst)) - 1);
      WS_PROF(21)
      drop(s, R, l, q, lane);
      WS_PROF(22)
      --q;
      ++iters;
    }
  }

  // working set -= entry at position l (one Householder reflection, see the header comment)
  static __device__ __forceinline__ void drop(S& s, Regs& R, int l, int q, int lane) {
    const int t = q - 1;
    const int row = row_of(lane), c0 = col0_of(lane);
    const bool on = row_ok(lane);
    double uv[NC];  // this lane's columns of row l of U
    {
      const D2* rl = reinterpret_cast<const D2*>(&s.U[l * LDT + c0]);
#pragma unroll
      for (int j = 0; j < NC; j += 2) {
        const D2 v2 = rl[j / 2];
        uv[j] = v2.x, uv[j + 1] = v2.y;
      }
    }
    const double ut = s.U[l * LDT + t];
    double s0 = 0, s1 = 0;
#pragma unroll
    for (int j = 0; j < NC; j += 2) s0 += uv[j] * uv[j], s1 += uv[j + 1] * uv[j + 1];
    const double sigma = (ut > 0 ? -1.0 : 1.0) * sqrt(hsum(s0 + s1));
    const double beta = 1.0 / (sigma * (sigma - ut));  // 2 / (v^T v), v = u_l - sigma e_t
    double lam_next = 0.0;
    int act_next = -1;
    if (lane >= l && lane < t) lam_next = s.lam[lane + 1], act_next = s.act[lane + 1];
    if (on) {  // (always true in the split kernel, so the cross-half sums below are executed by every lane)
      // own row of U (LDS) and of J (registers): x -= (x . v) beta v
      double ur[NC];
      const D2* ro = reinterpret_cast<const D2*>(&s.U[row * LDT + c0]);
#pragma unroll
      for (int j = 0; j < NC; j += 2) {
        const D2 v2 = ro[j / 2];
        ur[j] = v2.x, ur[j + 1] = v2.y;
      }
      const double urt = s.U[row * LDT + t];
      const double jt = hsum(reg_get(R.Jr, t - c0));
      double wu0 = 0, wu1 = 0, wj0 = 0, wj1 = 0;
#pragma unroll
      for (int j = 0; j < NC; j += 2) {
        wu0 += ur[j] * uv[j], wu1 += ur[j + 1] * uv[j + 1];
        wj0 += R.Jr[j] * uv[j], wj1 += R.Jr[j + 1] * uv[j + 1];  // uv is zero beyond column t
      }
      const double cu = (hsum(wu0 + wu1) - sigma * urt) * beta, cj = (hsum(wj0 + wj1) - sigma * jt) * beta;
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        ur[j] -= cu * uv[j];
        R.Jr[j] -= cj * uv[j];
      }
      reg_set(R.Jr, t - c0, jt - cj * (ut - sigma));  // the freed direction stays in J as a free column
      wsync();                                        // every lane has read its row before anybody rewrites a slot
      // rows above l stay, rows l+1..q-1 move up one slot, row l (the dropped entry) disappears
      if (row != l && row < q) {
        D2* dst = reinterpret_cast<D2*>(&s.U[(row > l ? row - 1 : row) * LDT + c0]);
#pragma unroll
        for (int j = 0; j < NC; j += 2) dst[j / 2] = D2{ur[j], ur[j + 1]};
      }
    } else {
      wsync();
    }
    if (lane >= l && lane < t) s.lam[lane] = lam_next, s.act[lane] = act_next;
    wsync();
    // structural zeros: column t of every row belongs to the freed direction, slot t is empty again
    if (lane < NV) s.U[lane * LDT + t] = 0.0;
    if (on && row == t) {
      D2* dst = reinterpret_cast<D2*>(&s.U[row * LDT + c0]);
#pragma unroll
      for (int j = 0; j < NC; j += 2) dst[j / 2] = D2{0.0, 0.0};
    }
    wsync();
  }

  // Continues from the current (dual feasible) state until no row of the current node is violated.
  static __device__ __forceinline__ int run(S& s, const Consts& c, Regs& R, double f_cut, int& iters) {
    const int lane = (int)threadIdx.x;
    const int n = c.n, N = c.N, max_iters = c.max_iters;
    const double tol = c.tol;
    const long long time_ticks = c.time_ticks;  // 0 = no wall-clock budget (the default)
    double f = s.f;
    int q = uni(s.q), neq = uni(s.neq_done);
    int rc = GI_OK;
    PROF_DECL
    for (;;) {
      int ln = lane;  // (lane masks and addresses of the state evaluation and the scan: formed per operation, see hdsm_wave_gib.h)
      keep_in_loop(ln);
      int ip;
      double vip;
      if (neq < 6) {
        states(s, R, ln, N);
        PROF(0)
        ip = mk_id(K_E, neq);
        vip = resid(s, c, ip, N);
      } else {  // (state boxes last, as in hdsm_wave_gib.h)
        states<1>(s, R, ln, N);
        PROF(0)
        select<1>(s, c, R, ln, tol, N, vip, ip);
        ip = uni(ip);
        if (ip < 0) {
          states<2>(s, R, ln, N);
          select<2>(s, c, R, ln, tol, N, vip, ip);
          ip = uni(ip);
        }
        if (ip < 0) {
          if (promote_cold(s, lane, tol) > 0) continue;
          PROF(1)
          break;
        }
        PROF(1)
      }
      const bool is_eq = id_kind(ip) == K_E;
      const double ai = normal_entry(s, R, ip, row_of(lane), N, n);
      double lam_p = 0;
      bool stop = false;
      for (;;) {
        if (iters >= max_iters) {
          rc = GI_ITERLIM;
          stop = true;
          break;
        }
        if (time_ticks > 0 && (long long)wall_clock64() - s.t_start > time_ticks) {
          rc = GI_TIMELIM;
          stop = true;
          break;
        }
        ++iters;
        PROF(2)
        double dv[NC], dd, zz, dq, zi, ri;
        direction(s, R, ip, ai, q, lane, dv, dd, zz, dq, zi, ri);
        PROF(3)
        const bool dependent = !(zz > 1e-20 * dd) || q >= NV;
        double t1 = DINF;
        int l = -1;
        if (!is_eq) {  // ratio test over the active inequalities (position k lives in lane k)
          const bool okk = lane < q && id_kind(s.act[lane]) != K_E && ri > 0;
          const double ratio = okk ? s.lam[lane] / ri : DINF;
          const double m = -wave_max64(-ratio);
          if (m < DINF) {
            t1 = m;
            l = uni(__ffsll((long long)__ballot(okk && ratio == m)) - 1);
          }
        }
        PROF(4)
        if (dependent && l < 0) {
          rc = GI_INFEASIBLE;
          if (lane == 0) s.inf_id = ip;  // the row that cannot be satisfied together with the current working set
          stop = true;
          break;
        }
        if (dependent) {  // dual step only; constraint l leaves
          if (lane < q) s.lam[lane] -= t1 * ri;
          lam_p += t1;
          wsync();
          drop(s, R, l, q, lane);
          --q;
          PROF(7)
          continue;
        }
        const double t2 = vip / zz;
        const bool full = is_eq || t2 <= t1;
        const double t = full ? t2 : t1;
        if (lane < n) {
          R.xi += t * zi;
          s.x[lane] = R.xi;
        }
        if (lane < q) s.lam[lane] -= t * ri;
        f += t * zz * (0.5 * t + lam_p);
        lam_p += t;
        PROF(5)
        if (full) {
          // ---- add at position q: Householder on the free columns, d2 -> rho e_q
          householder_add(s, R, ip, lam_p, q, lane, dv, zz, dq, zi, ri);
          PROF(6)
          ++q;
          if (is_eq) ++neq;
          break;
        }
        wsync();
        drop(s, R, l, q, lane);
        PROF(7)
        --q;
        states(s, R, lane, N);
        vip = resid(s, c, ip, N);
        if (f >= f_cut) {
          rc = GI_CUTOFF;
          if (lane == 0) s.inf_id = ip;  // (gi_run turns a cut on the box bound into a proof of infeasibility: the row on its way in)
          stop = true;
          break;
        }
      }
      if (stop) break;
      if (f >= f_cut) {
        rc = GI_CUTOFF;
        if (lane == 0) s.inf_id = ip;
        break;
      }
    }
    wsync();
    if (lane == 0) s.f = f, s.q = q, s.neq_done = neq, s.cmd = 0;
    if (blockDim.x > 64) __syncthreads();  // releases the helper waves (they leave on cmd == 0)
    else wsync();
    return rc;
  }

  // snapshots: J rows from registers, U rows / multipliers / ids / x from LDS; layout [row j][lane]
  static constexpr int SNAP_DOUBLES = (2 * NV + 3) * NV + 2;
  static __device__ __forceinline__ void snapshot(S& s, Regs& R, double* buf, bool save, int lane) {
    keep_in_loop(lane);  // (the per-lane offsets of a snapshot are formed when one is taken, not kept alive across the active-set run)
    const int row = row_of(lane), c0 = col0_of(lane);
    if (row_ok(lane)) {
      if (save) {
#pragma unroll
        for (int j = 0; j < NC; ++j)
          buf[(c0 + j) * NV + row] = R.Jr[j], buf[(NV + c0 + j) * NV + row] = s.U[row * LDT + c0 + j];
      } else {
#pragma unroll
        for (int j = 0; j < NC; ++j)
          R.Jr[j] = buf[(c0 + j) * NV + row], s.U[row * LDT + c0 + j] = buf[(NV + c0 + j) * NV + row];
      }
    }
    if (lane < NV) {
      if (save) {
        buf[2 * NV * NV + lane] = R.xi;
        buf[(2 * NV + 1) * NV + lane] = s.lam[lane];
        buf[(2 * NV + 2) * NV + lane] = (double)s.act[lane];
      } else {
        R.xi = buf[2 * NV * NV + lane];
        s.lam[lane] = buf[(2 * NV + 1) * NV + lane];
        s.act[lane] = (int)buf[(2 * NV + 2) * NV + lane];
        s.x[lane] = R.xi;
      }
    }
    if (lane == 0) {
      if (save) {
        buf[(2 * NV + 3) * NV] = s.f;
        buf[(2 * NV + 3) * NV + 1] = (double)s.q;
      } else {
        s.f = buf[(2 * NV + 3) * NV];
        s.q = (int)buf[(2 * NV + 3) * NV + 1];
      }
    }
    wsync();
  }
};

}  // namespace hdsm
