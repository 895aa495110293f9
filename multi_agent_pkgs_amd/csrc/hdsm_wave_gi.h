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

template <int NV, int CMAX>
struct WaveGI {
  using S = Shm<NV, CMAX>;
  static constexpr int LDT = S::LDT;
  static constexpr int HT = NV / 3;  // horizon capacity of this instantiation

  struct Regs {
    double Jr[NV];
    double Ur[NV];
    double xi, lami;
    int acti;
  };

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
  static __device__ __forceinline__ void select(S& s, const Consts& c, int lane, double xi, double& vbest, int& ibest) {
    const int N = c.N, n = c.n, RS = c.RS;
    double v = c.tol;
    int id = -1;
    if (lane < n) {
      const int ax = lane / N;
      const double vu = (fabs(c.ubu[ax]) < ABSENT) ? xi - c.ubu[ax] : -DINF;
      const double vl = (fabs(c.lbu[ax]) < ABSENT) ? c.lbu[ax] - xi : -DINF;
      if (vu > v) v = vu, id = mk_id(K_U, lane << 1);
      if (vl > v) v = vl, id = mk_id(K_U, (lane << 1) | 1);
    }
    const int n_sb = 6 * (N - 1);
    for (int idx = lane; idx < n_sb; idx += 64) {
      const int i = idx / 6 + 1, k = idx % 6, comp = 1 + k / 3, ax = k % 3;
      const double sv = s.st[i][3 * comp + ax];
      const double vu = (fabs(c.ubs[comp][ax]) < ABSENT) ? sv - c.ubs[comp][ax] : -DINF;
      const double vl = (fabs(c.lbs[comp][ax]) < ABSENT) ? c.lbs[comp][ax] - sv : -DINF;
      const int base = (i << 5) | (comp << 3) | (ax << 1);
      if (vu > v) v = vu, id = mk_id(K_S, base);
      if (vl > v) v = vl, id = mk_id(K_S, base | 1);
    }
    if (s.level > 0) {  // rows of the polyhedra assigned on the current branch
      const int n_sp = N * 2 * RS;
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
    const int nc = s.ncand;
    for (int idx = lane; idx < nc; idx += 64) {
      const double* row = s.cand[idx];
      const double* pm = s.st[s.cand_m[idx]];
      const double vv = row[0] * pm[0] + row[1] * pm[1] + row[2] * pm[2] - row[3];
      if (vv > v) v = vv, id = mk_id(K_C, idx);
    }
    const double m = wave_max64(v);
    vbest = m;
    ibest = -1;
    if (m > c.tol) {
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

  // d = J^T (-a) -> s.dvec (LDS, read back by every lane as broadcast b128 loads)
  static __device__ __forceinline__ void compute_d(S& s, const Regs& R, int id, double ai, int lane) {
    if (id_kind(id) == K_U) {  // a = sg e_k: d = -sg * (row k of J)
      const int var = id_payload(id) >> 1;
      const double msg = (id_payload(id) & 1) ? 1.0 : -1.0;
      if (lane == var) {
#pragma unroll
        for (int j = 0; j < NV; j += 2) *reinterpret_cast<D2*>(&s.dvec[j]) = D2{msg * R.Jr[j], msg * R.Jr[j + 1]};
      }
      wsync();
    } else {
      if (lane < NV) {
#pragma unroll
        for (int j = 0; j < NV; ++j) s.T[j * LDT + lane] = R.Jr[j] * ai;
      }
      wsync();
      if (lane < NV) {
        double a0 = 0, a1 = 0;
        const D2* row = reinterpret_cast<const D2*>(&s.T[lane * LDT]);
#pragma unroll
        for (int i = 0; i < NV / 2; ++i) {
          const D2 t = row[i];
          a0 += t.x;
          a1 += t.y;
        }
        s.dvec[lane] = -(a0 + a1);
      }
      wsync();
    }
  }

  // working set += id (full step taken). dv = current d, zz = sum_{k>=q} d_k^2, ri = r of this lane.
  static __device__ __forceinline__ void add(S& s, Regs& R, int id, double lam_p, int q, int lane, double sufj, double zz,
                                             double ri) {
    // own Givens pair (column j = lane, j > q): zero d_j into d_{j-1}
    {
      double cc = 1.0, ss = 0.0;
      if (lane > q && lane < NV) {
        const double dj = s.dvec[lane], dm1 = s.dvec[lane - 1];
        const double h = sqrt(sufj + dm1 * dm1);
        if (h > 0) {
          cc = dm1 / h;
          ss = ((lane == NV - 1) ? dj : sqrt(sufj)) / h;
        }
      }
      if (lane < NV) *reinterpret_cast<D2*>(&s.cs[2 * lane]) = D2{cc, ss};
    }
    const double rho = (q == NV - 1) ? s.dvec[NV - 1] : sqrt(zz);
    wsync();
    if (lane < NV) {
      // branch-free sweep from the last column down: pairs for j <= q are the identity (1, 0), which makes
      // the recurrence copy every column back onto itself and leaves the rotated tail in column q
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
    // new column q of U = R^{-1}: (-r / rho ; 1 / rho)
    const double ucol = (lane < q) ? -ri / rho : ((lane == q) ? 1.0 / rho : 0.0);
#pragma unroll
    for (int j = 0; j < NV; ++j) R.Ur[j] = (j == q) ? ucol : R.Ur[j];
    if (lane == q) {
      R.lami = lam_p;
      R.acti = id;
    }
  }

  // working set -= entry at position l
  static __device__ __forceinline__ void drop(S& s, Regs& R, int l, int q, int lane) {
    if (lane == l) {
#pragma unroll
      for (int j = 0; j < NV; j += 2) *reinterpret_cast<D2*>(&s.dvec[j]) = D2{R.Ur[j], R.Ur[j + 1]};
    }
    wsync();
    {  // own rotation pair for column j = lane in [l, q-2]: (a_j, u_{j+1}) -> (0, sigma_{j+1})
      double cd = 1.0, sd = 0.0;
      if (lane >= l && lane <= q - 2) {
        double pre = 0;  // sum_{k=l..lane} u_k^2
#pragma unroll
        for (int k = 0; k < NV; ++k) {
          const double uk = s.dvec[k];
          if (k >= l && k <= lane) pre += uk * uk;
        }
        const double aj = (lane == l) ? s.dvec[l] : sqrt(pre);
        const double bj = s.dvec[lane + 1];
        const double sg = sqrt(pre + bj * bj);
        cd = bj / sg;
        sd = aj / sg;
      }
      if (lane < NV) *reinterpret_cast<D2*>(&s.cs[2 * lane]) = D2{cd, sd};
    }
    wsync();
    if (lane < NV) {
      // branch-free forward sweep: identity pairs outside [l, q-2] copy the columns through unchanged; the
      // freed direction ends up in column q-1 (kept in J as a free column, stale in U until the next add)
      double cu = R.Ur[0], cj = R.Jr[0];
#pragma unroll
      for (int j = 0; j < NV - 1; ++j) {
        const D2 g = *reinterpret_cast<const D2*>(&s.cs[2 * j]);
        const double tu = R.Ur[j + 1], tj = R.Jr[j + 1];
        R.Ur[j] = g.x * cu - g.y * tu;
        cu = g.y * cu + g.x * tu;
        R.Jr[j] = g.x * cj - g.y * tj;
        cj = g.y * cj + g.x * tj;
      }
      R.Ur[NV - 1] = cu;
      R.Jr[NV - 1] = cj;
    }
    // rows l+1 .. q-1 of U (and their multipliers / ids) move up one lane
    if (lane > l && lane < q) {
#pragma unroll
      for (int j = 0; j < NV; j += 2) *reinterpret_cast<D2*>(&s.T[(lane - 1) * LDT + j]) = D2{R.Ur[j], R.Ur[j + 1]};
      s.w[lane - 1] = R.lami;
      s.red_i[lane - 1] = R.acti;
    }
    wsync();
    if (lane >= l && lane < q - 1) {
#pragma unroll
      for (int j = 0; j < NV; j += 2) {
        const D2 t = *reinterpret_cast<const D2*>(&s.T[lane * LDT + j]);
        R.Ur[j] = t.x;
        R.Ur[j + 1] = t.y;
      }
      R.lami = s.w[lane];
      R.acti = s.red_i[lane];
    } else if (lane == q - 1) {
#pragma unroll
      for (int j = 0; j < NV; ++j) R.Ur[j] = 0.0;
      R.lami = 0.0;
      R.acti = -1;
    }
    wsync();
  }

  // Continues from the current (dual feasible) state until no row of the current node is violated.
  static __device__ __forceinline__ int run(S& s, const Consts& c, Regs& R, double f_cut, int& iters) {
    const int lane = (int)threadIdx.x;
    const int n = c.n;
    double f = s.f;
    int q = s.q, neq = s.neq_done;
    int rc = GI_OK;
    for (;;) {
      states(s, c, lane);
      int ip;
      double vip;
      if (neq < 6) {
        ip = mk_id(K_E, neq);
        vip = resid(s, c, ip);
      } else {
        select(s, c, lane, R.xi, vip, ip);
        if (ip < 0) break;
      }
      const bool is_eq = id_kind(ip) == K_E;
      const double ai = normal_entry(s, c, ip, lane);
      double lam_p = 0;
      bool stop = false;
      for (;;) {
        if (iters >= c.max_iters) {
          rc = GI_ITERLIM;
          stop = true;
          break;
        }
        ++iters;
        compute_d(s, R, ip, ai, lane);
        double dd = 0, zz = 0, zi = 0, ri = 0, sufj = 0;
#pragma unroll
        for (int k2 = 0; k2 < NV; k2 += 2) {
          const D2 dk = *reinterpret_cast<const D2*>(&s.dvec[k2]);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int k = k2 + h;
            const double dval = h ? dk.y : dk.x;
            const double d2 = dval * dval;
            dd += d2;
            if (k >= q) {
              zz += d2;
              zi += R.Jr[k] * dval;
            } else if (k >= lane) {
              ri += R.Ur[k] * dval;
            }
            if (k >= lane) sufj += d2;
          }
        }
        const bool dependent = !(zz > 1e-20 * dd) || q >= NV;
#ifdef HDSM_DEBUG
        if (blockIdx.x == 2 && lane < 4)
          printf("it %d lane %d ip %x q %d vip %.6e dd %.6e zz %.6e zi %.6e ri %.6e ai %.6e d[lane] %.6e J0 %.4e J1 %.4e\n", iters, lane, ip, q,
                 vip, dd, zz, zi, ri, ai, s.dvec[lane], R.Jr[0], R.Jr[1]);
#endif
        double t1 = DINF;
        int l = -1;
        if (!is_eq) {  // ratio test over the active inequalities (position k lives in lane k)
          const bool okk = lane < q && id_kind(R.acti) != K_E && ri > 0;
          const double ratio = okk ? R.lami / ri : DINF;
          const double m = -wave_max64(-ratio);
          if (m < DINF) {
            t1 = m;
            l = __ffsll((long long)__ballot(okk && ratio == m)) - 1;
          }
        }
        if (dependent && l < 0) {
          rc = GI_INFEASIBLE;
          stop = true;
          break;
        }
        if (dependent) {  // dual step only; constraint l leaves
          if (lane < q) R.lami -= t1 * ri;
          lam_p += t1;
          drop(s, R, l, q, lane);
          --q;
          continue;
        }
        const double t2 = vip / zz;
        const bool full = is_eq || t2 <= t1;
        const double t = full ? t2 : t1;
        if (lane < n) {
          R.xi += t * zi;
          s.x[lane] = R.xi;
        }
        if (lane < q) R.lami -= t * ri;
        f += t * zz * (0.5 * t + lam_p);
        lam_p += t;
        if (full) {
          add(s, R, ip, lam_p, q, lane, sufj, zz, ri);
          ++q;
          if (is_eq) ++neq;
          break;
        }
        drop(s, R, l, q, lane);
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

  // snapshots of the register state, layout [row j][lane] (coalesced), then x, lam, act, (f, q)
  static constexpr int SNAP_DOUBLES = (2 * NV + 3) * NV + 2;
  static __device__ __forceinline__ void snapshot(S& s, Regs& R, double* buf, bool save, int lane) {
    if (lane < NV) {
      if (save) {
#pragma unroll
        for (int j = 0; j < NV; ++j) buf[j * NV + lane] = R.Jr[j], buf[(NV + j) * NV + lane] = R.Ur[j];
        buf[2 * NV * NV + lane] = R.xi;
        buf[(2 * NV + 1) * NV + lane] = R.lami;
        buf[(2 * NV + 2) * NV + lane] = (double)R.acti;
      } else {
#pragma unroll
        for (int j = 0; j < NV; ++j) R.Jr[j] = buf[j * NV + lane], R.Ur[j] = buf[(NV + j) * NV + lane];
        R.xi = buf[2 * NV * NV + lane];
        R.lami = buf[(2 * NV + 1) * NV + lane];
        R.acti = (int)buf[(2 * NV + 2) * NV + lane];
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
