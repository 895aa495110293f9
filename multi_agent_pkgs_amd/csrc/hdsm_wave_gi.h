// hdsm_wave_gi.h — device-only (gfx950): the dual active-set iteration of ONE instance executed by ONE
// 64-lane wavefront with the factorisation held in REGISTERS.
//
// Same mathematics as Solver::gi_run in hdsm_core.h (Goldfarb-Idnani, J = L^{-T} Q with J^T N = [R; 0]),
// different data layout, chosen for CDNA4:
//   * lane i owns row i of J (NV doubles, statically indexed registers) and row i of U = R^{-1};
//     the multiplier / id of the working-set entry at position k live in lane k;
//   * r = U d1 is a lane-local dot product: no back-substitution chain;
//   * d = J^T a is a transposition through LDS (T[j][i] = J[i][j] a_i, conflict-free strides), or a single
//     row broadcast when the incoming row is an input bound (a = +-e_k, the common case in bang-bang plans);
//   * "add": the Givens sweep on columns q..NV-1 of J is lane-local; its coefficients come from suffix sums
//     of d^2 (each lane computes its own pair, one LDS exchange) — no sequential sqrt chain;
//     U gets the new column (-r/rho ; 1/rho);
//   * "drop l": U' = E^T U G^T where G rotates row l of U onto the last axis: coefficients from PREFIX sums
//     of that row, sweep lane-local on U and J, then the rows >= l of U move up one lane;
//   * cross-lane reductions use DPP row rotations + v_readlane, not LDS trees.
// Dimension handling: the factors are padded to NV (identity beyond n = 3N), so every loop has a
// compile-time trip count and unrolls; a padded direction never receives a step (d_k = 0 there).
#pragma once
#ifndef HDSM_EMU
#include <hip/hip_runtime.h>

#include "hdsm_types.h"

namespace hdsm {

template <int CTRL>
__device__ __forceinline__ double dpp64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
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
// DPP move with zero fill for lanes shifted in from outside the 16-lane row
template <int CTRL>
__device__ __forceinline__ double dpp64z(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
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

#ifdef HDSM_WSYNC_STRONG
__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
}
#else
__device__ __forceinline__ void wsync() { __syncthreads(); }  // single-wave workgroup: lowers to a waitcnt
#endif

struct alignas(16) D2 {
  double x, y;
};

#ifdef HDSM_PROFILE
// cycle counters accumulated in LDS by lane 0 (keeps them out of the SGPR file)
#define PROF_DECL if (threadIdx.x == 0) s.prof_last = clock64();
#define PROF(k) if (threadIdx.x == 0) { const long long now_ = clock64(); s.prof_acc[k] += now_ - s.prof_last; s.prof_last = now_; }
#else
#define PROF_DECL
#define PROF(k)
#endif

template <int NV, int CMAX>
struct WaveGI {
  using S = Shm<NV, CMAX>;
  static constexpr int LDT = S::LDT;
  static constexpr int HT = NV / 3;  // horizon capacity of this instantiation

  struct Regs {
    double Jr[NV];  // row `lane` of J
    double xi;      // u[lane]
    // per-lane constants of the violation scan (bounds with "absent" mapped to +-DINF), set by init_lane()
    double ub_own, lb_own;      // box of this lane's input
    double sb_ub[2], sb_lb[2];  // boxes of the (up to two) state-bound items scanned by this lane
    int sb_off[2], sb_id[2];    // their offset in st[][] (as a flat index) and id base; -1 = no item
  };

  static __device__ __forceinline__ void init_lane(Regs& R, const Consts& c, int lane) {
    const int N = c.N, n = c.n;
    R.ub_own = DINF, R.lb_own = -DINF;
    if (lane < n) {
      const int ax = lane / N;
      if (fabs(c.ubu[ax]) < ABSENT) R.ub_own = c.ubu[ax];
      if (fabs(c.lbu[ax]) < ABSENT) R.lb_own = c.lbu[ax];
    }
    const int n_sb = 6 * (N - 1);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int idx = lane + 64 * e;
      R.sb_off[e] = -1, R.sb_id[e] = 0, R.sb_ub[e] = DINF, R.sb_lb[e] = -DINF;
      if (idx < n_sb) {
        const int i = idx / 6 + 1, k = idx % 6, comp = 1 + k / 3, ax = k % 3;
        R.sb_off[e] = i * 9 + 3 * comp + ax;
        R.sb_id[e] = (i << 5) | (comp << 3) | (ax << 1);
        if (fabs(c.ubs[comp][ax]) < ABSENT) R.sb_ub[e] = c.ubs[comp][ax];
        if (fabs(c.lbs[comp][ax]) < ABSENT) R.sb_lb[e] = c.lbs[comp][ax];
      }
    }
  }

  // trajectory from s.x: lane (ax, m-1) evaluates p, v, a of step m (zero-padded Toeplitz table gz in LDS)
  static __device__ __forceinline__ void states(S& s, const Consts& c, int lane) {
    const int N = c.N;
    if (lane < 3 * N) {
      const int ax = lane / N, m = lane % N + 1;
      double acc0 = s.fr[ax][m][0], acc1 = s.fr[ax][m][1], acc2 = s.fr[ax][m][2];
      const double* xx = s.x + ax * N;
      const double* g0 = &s.gz[ax][0][MAXH + m - 1];
      const double* g1 = &s.gz[ax][1][MAXH + m - 1];
      const double* g2 = &s.gz[ax][2][MAXH + m - 1];
#pragma unroll
      for (int k = 0; k < HT; ++k) {
        const double xk = xx[k];
        acc0 += g0[-k] * xk;
        acc1 += g1[-k] * xk;
        acc2 += g2[-k] * xk;
      }
      s.st[m][ax] = acc0;
      s.st[m][3 + ax] = acc1;
      s.st[m][6 + ax] = acc2;
    }
    wsync();
  }

  // most violated row of the current node -> (v, id), id < 0 if none exceeds tol
  static __device__ __forceinline__ void select(S& s, const Consts& c, const Regs& R, int lane, double tol, int N,
                                                double& vbest, int& ibest) {
    double v = tol;
    int id = -1;
    {  // this lane's input bound
      const double vu = R.xi - R.ub_own, vl = R.lb_own - R.xi;
      if (vu > v) v = vu, id = mk_id(K_U, lane << 1);
      if (vl > v) v = vl, id = mk_id(K_U, (lane << 1) | 1);
    }
    const double* stf = &s.st[0][0];
#pragma unroll
    for (int e = 0; e < 2; ++e) {  // velocity / acceleration boxes on x_1 .. x_{N-1}
      if (R.sb_off[e] >= 0) {
        const double sv = stf[R.sb_off[e]];
        const double vu = sv - R.sb_ub[e], vl = R.sb_lb[e] - sv;
        if (vu > v) v = vu, id = mk_id(K_S, R.sb_id[e]);
        if (vl > v) v = vl, id = mk_id(K_S, R.sb_id[e] | 1);
      }
    }
    if (uni(s.level) > 0) {  // rows of the polyhedra assigned on the current branch
      const int RS = c.RS, n_sp = N * 2 * RS;
      for (int idx = lane; idx < n_sp; idx += 64) {
        const int i = idx / (2 * RS), rem = idx % (2 * RS), e = rem / RS, r = rem % RS;
        const int j = s.assign[i];
        if (j < 0 || r >= s.sp_rows[j] || i + e == 0) continue;
        const double* row = s.sp[j][r];
        const double* pm = s.st[i + e];
        const double vv = row[0] * pm[0] + row[1] * pm[1] + row[2] * pm[2] - row[3];
        if (vv > v) v = vv, id = mk_id(K_P, (i << 7) | (e << 6) | r);
      }
    }
    // staged neighbour rows, four per trip with all loads issued before the first use
    const int nc = uni(s.ncand);
    for (int base = 0; base < nc; base += 256) {
      int idx[4], mm[4];
      D2 r01[4], r23[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        idx[u] = base + 64 * u + lane;
        const int ii = idx[u] < nc ? idx[u] : 0;
        mm[u] = s.cand_m[ii];
        r01[u] = *reinterpret_cast<const D2*>(&s.cand[ii][0]);
        r23[u] = *reinterpret_cast<const D2*>(&s.cand[ii][2]);
      }
      double px[4], py[4], pz[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double* pm = s.st[mm[u]];
        px[u] = pm[0], py[u] = pm[1], pz[u] = pm[2];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double vv = r01[u].x * px[u] + r01[u].y * py[u] + r23[u].x * pz[u] - r23[u].y;
        if (idx[u] < nc && vv > v) v = vv, id = mk_id(K_C, idx[u]);
      }
    }
    const double m = wave_max64(v);
    vbest = m;
    ibest = -1;
    if (m > tol) {
      const unsigned long long mask = __ballot(v == m && id >= 0);
      const int src = __ffsll((long long)mask) - 1;
      ibest = __builtin_amdgcn_readlane(id, src);
    }
  }

  // entry `lane` of the dense normal a of constraint id (a . u <= rhs form)
  static __device__ __forceinline__ double normal_entry(const S& s, const Consts& c, int id, int lane) {
    const int N = c.N, n = c.n, kind = id_kind(id), p = id_payload(id);
    if (lane >= n) return 0.0;
    const int ax = lane / N, kk = lane % N;
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
      row = s.cand[p];
      m = s.cand_m[p];
    }
    return (kk < m) ? row[ax] * s.gz[ax][0][MAXH + m - 1 - kk] : 0.0;
  }

  static __device__ __forceinline__ double resid(const S& s, const Consts& c, int id) {
    const int N = c.N, kind = id_kind(id), p = id_payload(id);
    if (kind == K_U) {
      const int var = p >> 1, ax = var / N;
      return (p & 1) ? (c.lbu[ax] - s.x[var]) : (s.x[var] - c.ubu[ax]);
    }
    if (kind == K_S) {
      const int sg = p & 1, ax = (p >> 1) & 3, comp = (p >> 3) & 3, i = p >> 5;
      const double v = s.st[i][3 * comp + ax];
      return sg ? (c.lbs[comp][ax] - v) : (v - c.ubs[comp][ax]);
    }
    if (kind == K_E) return s.st[N][3 * (1 + p / 3) + p % 3];
    const double* row;
    const double* pm;
    if (kind == K_P) {
      const int r = p & 63, e = (p >> 6) & 1, i = p >> 7;
      row = s.sp[s.assign[i]][r];
      pm = s.st[i + e];
    } else {
      row = s.cand[p];
      pm = s.st[s.cand_m[p]];
    }
    return row[0] * pm[0] + row[1] * pm[1] + row[2] * pm[2] - row[3];
  }

  // d = J^T (-a) -> s.dvec[0..NV) in LDS; returns this lane's own entry d_lane (0 beyond NV)
  static __device__ __forceinline__ double compute_d(S& s, const Regs& R, int id, double ai, int lane) {
    if (id_kind(id) == K_U) {  // a = sg e_k: d = -sg * (row k of J)
      const int var = id_payload(id) >> 1;
      const double msg = (id_payload(id) & 1) ? 1.0 : -1.0;
      if (lane == var) {
#pragma unroll
        for (int j = 0; j < NV; j += 2) *reinterpret_cast<D2*>(&s.dvec[j]) = D2{msg * R.Jr[j], msg * R.Jr[j + 1]};
      }
      wsync();
      return (lane < NV) ? s.dvec[lane] : 0.0;
    }
    if (lane < NV) {
#pragma unroll
      for (int j = 0; j < NV; ++j) s.T[j * LDT + lane] = R.Jr[j] * ai;
    }
    wsync();
    double dj = 0.0;
    if (lane < NV) {
      double a0 = 0, a1 = 0;
      const D2* row = reinterpret_cast<const D2*>(&s.T[lane * LDT]);
#pragma unroll
      for (int i = 0; i < NV / 2; ++i) {
        const D2 t = row[i];
        a0 += t.x;
        a1 += t.y;
      }
      dj = -(a0 + a1);
      s.dvec[lane] = dj;
    }
    return dj;  // the caller synchronises before anybody reads dvec
  }

  // working set += id at position q (full step taken)
  static __device__ __forceinline__ void add(S& s, Regs& R, int id, double lam_p, int q, int lane, double dj, double sufj,
                                             double zz, double ri) {
    {  // own Givens pair (column j = lane, j > q): zero d_j into d_{j-1}; identity for j <= q
      double cc = 1.0, ss = 0.0;
      if (lane > q && lane < NV) {
        const double dm1 = s.dvec[lane - 1];
        const double h = sqrt(sufj + dm1 * dm1);
        if (h > 0) {
          cc = dm1 / h;
          ss = ((lane == NV - 1) ? dj : sqrt(sufj)) / h;
        }
      }
      if (lane < NV) *reinterpret_cast<D2*>(&s.cs[2 * lane]) = D2{cc, ss};
    }
    const double rho = (q == NV - 1) ? s.dvec[NV - 1] : sqrt(zz);
    // new column q of U = R^{-1}: (-r / rho ; 1 / rho ; 0)
    if (lane < NV) s.U[lane * LDT + q] = (lane < q) ? -ri / rho : ((lane == q) ? 1.0 / rho : 0.0);
    if (lane == q) {
      s.lam[q] = lam_p;
      s.act[q] = id;
    }
    wsync();
    if (lane < NV) {
      // branch-free sweep from the last column down: the identity pairs (1, 0) for j <= q make the
      // recurrence copy those columns back onto themselves and leave the rotated tail in column q
      double carry = R.Jr[NV - 1];
#pragma unroll
      for (int j = NV - 1; j >= 1; --j) {
        const D2 g = *reinterpret_cast<const D2*>(&s.cs[2 * j]);
        const double t1 = R.Jr[j - 1];
        R.Jr[j] = g.x * carry - g.y * t1;
        carry = g.x * t1 + g.y * carry;
      }
      R.Jr[0] = carry;
    }
  }

  // working set -= entry at position l: U' = E^T U G^T with G rotating row l of U onto the last axis
  static __device__ __forceinline__ void drop(S& s, Regs& R, int l, int q, int lane) {
    {  // own pair for column j = lane in [l, q-2]: (a_j, u_{j+1}) -> (0, sigma_{j+1}); identity elsewhere
      const double ul = (lane >= l && lane < q) ? s.U[l * LDT + lane] : 0.0;  // row l of U, own entry
      const double pre = wave_prefix_sum(ul * ul, lane);                        // sum_{k=l..lane} u_k^2
      double cd = 1.0, sd = 0.0;
      if (lane >= l && lane <= q - 2) {
        const double aj = (lane == l) ? ul : sqrt(pre);
        const double bj = s.U[l * LDT + lane + 1];
        const double sg = sqrt(pre + bj * bj);
        cd = bj / sg;
        sd = aj / sg;
      }
      if (lane < NV) *reinterpret_cast<D2*>(&s.cs[2 * lane]) = D2{cd, sd};
    }
    double lam_next = 0.0;
    int act_next = -1;
    if (lane >= l && lane < q - 1) lam_next = s.lam[lane + 1], act_next = s.act[lane + 1];
    double ur[NV];
    if (lane < NV) {
      const D2* row = reinterpret_cast<const D2*>(&s.U[lane * LDT]);
#pragma unroll
      for (int j = 0; j < NV; j += 2) {
        const D2 t = row[j / 2];
        ur[j] = t.x;
        ur[j + 1] = t.y;
      }
    }
    wsync();
    if (lane < NV) {
      // branch-free forward sweep on this lane's row of U and of J: identity pairs outside [l, q-2]
      double cu = ur[0], cj = R.Jr[0];
#pragma unroll
      for (int j = 0; j < NV - 1; ++j) {
        const D2 g = *reinterpret_cast<const D2*>(&s.cs[2 * j]);
        const double tu = ur[j + 1], tj = R.Jr[j + 1];
        ur[j] = g.x * cu - g.y * tu;
        cu = g.y * cu + g.x * tu;
        R.Jr[j] = g.x * cj - g.y * tj;
        cj = g.y * cj + g.x * tj;
      }
      ur[NV - 1] = cu;
      R.Jr[NV - 1] = cj;
      // rows above l stay, rows l+1..q-1 move up one slot, row l (the dropped entry) disappears
      if (lane != l && lane < q) {
        D2* dst = reinterpret_cast<D2*>(&s.U[(lane > l ? lane - 1 : lane) * LDT]);
#pragma unroll
        for (int j = 0; j < NV; j += 2) dst[j / 2] = D2{ur[j], ur[j + 1]};
      }
    }
    if (lane >= l && lane < q - 1) s.lam[lane] = lam_next, s.act[lane] = act_next;
    wsync();
    // structural zeros: slot q-1 is free again, column q-1 of every row belongs to the freed direction
    if (lane < NV) {
      s.U[lane * LDT + (q - 1)] = 0.0;
      if (lane == q - 1) {
        D2* dst = reinterpret_cast<D2*>(&s.U[lane * LDT]);
#pragma unroll
        for (int j = 0; j < NV; j += 2) dst[j / 2] = D2{0.0, 0.0};
      }
    }
    wsync();
  }

  // Continues from the current (dual feasible) state until no row of the current node is violated.
  static __device__ __forceinline__ int run(S& s, const Consts& c, Regs& R, double f_cut, int& iters) {
    const int lane = (int)threadIdx.x;
    const int n = c.n, N = c.N, max_iters = c.max_iters;
    const double tol = c.tol;
    double f = s.f;
    int q = uni(s.q), neq = uni(s.neq_done);
    int rc = GI_OK;
    PROF_DECL
    for (;;) {
      states(s, c, lane);
      PROF(0)
      int ip;
      double vip;
      if (neq < 6) {
        ip = mk_id(K_E, neq);
        vip = resid(s, c, ip);
      } else {
        select(s, c, R, lane, tol, N, vip, ip);
        ip = uni(ip);
        PROF(1)
        if (ip < 0) break;
      }
      const bool is_eq = id_kind(ip) == K_E;
      const double ai = normal_entry(s, c, ip, lane);
      double lam_p = 0;
      bool stop = false;
      for (;;) {
        if (iters >= max_iters) {
          rc = GI_ITERLIM;
          stop = true;
          break;
        }
        ++iters;
        PROF(2)
        const double dj = compute_d(s, R, ip, ai, lane);
        PROF(3)
        const double sufj = wave_suffix_sum(dj * dj, lane);  // sum_{k >= lane} d_k^2
        const double dd = bcast64(sufj, 0);
        const double zz = (q < NV) ? bcast64(sufj, q) : 0.0;
        if (lane < NV) s.dz[lane] = (lane >= q) ? dj : 0.0;  // d2 padded with zeros: no predicate in the dot products
        wsync();
        double zi = 0, ri = 0;
        if (lane < NV) {
          const D2* urow = reinterpret_cast<const D2*>(&s.U[lane * LDT]);
          double z0 = 0, z1 = 0, r0 = 0, r1 = 0;  // independent accumulators: no 60-deep dependent FMA chain
#pragma unroll
          for (int k = 0; k < NV; k += 2) {
            const D2 dzk = *reinterpret_cast<const D2*>(&s.dz[k]);
            const D2 dk = *reinterpret_cast<const D2*>(&s.dvec[k]);
            const D2 uk = urow[k / 2];
            z0 += R.Jr[k] * dzk.x;
            z1 += R.Jr[k + 1] * dzk.y;
            r0 += uk.x * dk.x;  // U is upper triangular with zero columns >= q: r = U d1 needs no mask
            r1 += uk.y * dk.y;
          }
          zi = z0 + z1;
          ri = r0 + r1;
        }
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
          add(s, R, ip, lam_p, q, lane, dj, sufj, zz, ri);
          PROF(6)
          ++q;
          if (is_eq) ++neq;
          break;
        }
        wsync();
        drop(s, R, l, q, lane);
        PROF(7)
        --q;
        states(s, c, lane);
        vip = resid(s, c, ip);
        if (f >= f_cut) {
          rc = GI_CUTOFF;
          stop = true;
          break;
        }
      }
      if (stop) break;
      if (f >= f_cut) {
        rc = GI_CUTOFF;
        break;
      }
    }
    wsync();
    if (lane == 0) s.f = f, s.q = q, s.neq_done = neq;
    wsync();
    return rc;
  }

  // snapshots: J rows from registers, U rows / multipliers / ids / x from LDS; layout [row j][lane]
  static constexpr int SNAP_DOUBLES = (2 * NV + 3) * NV + 2;
  static __device__ __forceinline__ void snapshot(S& s, Regs& R, double* buf, bool save, int lane) {
    if (lane < NV) {
      if (save) {
#pragma unroll
        for (int j = 0; j < NV; ++j) buf[j * NV + lane] = R.Jr[j], buf[(NV + j) * NV + lane] = s.U[lane * LDT + j];
        buf[2 * NV * NV + lane] = R.xi;
        buf[(2 * NV + 1) * NV + lane] = s.lam[lane];
        buf[(2 * NV + 2) * NV + lane] = (double)s.act[lane];
      } else {
#pragma unroll
        for (int j = 0; j < NV; ++j) R.Jr[j] = buf[j * NV + lane], s.U[lane * LDT + j] = buf[(NV + j) * NV + lane];
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
#endif  // !HDSM_EMU
