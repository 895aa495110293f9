// hdsm_consts.cpp — see hdsm_consts.h. Pure host code, no device dependency.
#include "hdsm_consts.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace hdsm {
namespace {

struct M3 {
  double m[3][3];
};
M3 eye() {
  M3 r{};
  for (int i = 0; i < 3; ++i) r.m[i][i] = 1;
  return r;
}
M3 mul(const M3& a, const M3& b) {
  M3 r{};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      for (int k = 0; k < 3; ++k) r.m[i][j] += a.m[i][k] * b.m[k][j];
  return r;
}
M3 axpy(const M3& a, double s, const M3& b) {  // a + s b
  M3 r = a;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i][j] += s * b.m[i][j];
  return r;
}

int fail(const char** err, const char* msg) {
  if (err) *err = msg;
  return HDSM_ERR_BAD_ARG;
}

}  // namespace

int build_consts(const hdsm_params* prm, Consts* c, const char** err) {
  if (!prm || !c) return fail(err, "null params");
  const int N = prm->n_hor, P = prm->poly_hor, RS = prm->max_rows_static;
  if (N < 2 || N > MAXH) return fail(err, "n_hor must be in [2, HDSM_MAX_HOR]");
  if (P < 1 || P > MAXP) return fail(err, "poly_hor must be in [1, HDSM_MAX_POLY]");
  if (RS < 1 || RS > MAXRS) return fail(err, "max_rows_static must be in [1, HDSM_MAX_ROWS_STATIC]");
  if (!(prm->dt > 0)) return fail(err, "dt must be positive");
  if (!(prm->r_u > 0)) return fail(err, "r_u must be positive (strict convexity, AC:2098)");
  if (!(prm->drone_radius > 0) || !(prm->drone_z_offset > 0)) return fail(err, "drone_radius/drone_z_offset must be positive");
  for (int k = 0; k < 3; ++k)
    if (std::fabs(prm->x_lb[k]) < ABSENT || std::fabs(prm->x_ub[k]) < ABSENT)
      return fail(err, "position bounds must be +-HDSM_INF (AC:2179-2182 leaves positions free)");
  for (int k = 0; k < 6; ++k)
    if (prm->r_x[k] < 0 || prm->r_n[k] < 0) return fail(err, "negative tracking weight");

  std::memset(c, 0, sizeof *c);
  const int n = 3 * N;
  c->N = N, c->n = n, c->P = P, c->RS = RS;
  c->max_nodes = prm->max_nodes > 0 ? prm->max_nodes : 2000;
  c->max_iters = prm->max_qp_iters > 0 ? prm->max_qp_iters : 100000;
  c->tol = prm->solver_tol > 0 ? prm->solver_tol : 1e-9;
  c->ftol_fixed = prm->feas_tol_fixed > 0 ? prm->feas_tol_fixed : 1e-6;
  // Execution knobs: hdsm_params first, then the HDSM_* environment variables (for scripts); values out of range are ignored.
  auto env_int = [](const char* name, int lo, int hi, int32_t* out) {
    if (const char* e = std::getenv(name)) {
      char* end = nullptr;
      const long v = std::strtol(e, &end, 10);
      if (end != e && *end == 0 && v >= lo && v <= hi) *out = (int32_t)v;
    }
  };
  auto env_real = [](const char* name, double lo, double hi, double* out) {
    if (const char* e = std::getenv(name)) {
      char* end = nullptr;
      const double v = std::strtod(e, &end);
      if (end != e && *end == 0 && v >= lo && v <= hi) *out = v;
    }
  };
  if (prm->presweep < 0 || prm->presweep > 2) return fail(err, "presweep must be 0 (automatic), 1 (never) or 2 (always)");
  if (prm->branch_rule < 0 || prm->branch_rule > 1) return fail(err, "branch_rule must be 0 (most infeasible) or 1 (first in time)");
  if (prm->stage_radius < 0 || prm->time_limit_s < 0) return fail(err, "stage_radius / time_limit_s must not be negative");
  if (!(prm->mip_gap >= 0) || prm->mip_gap >= 1) return fail(err, "mip_gap must be in [0, 1)");
  c->cand_tau = prm->stage_radius > 0 ? prm->stage_radius : 0.6;  // [m] rows whose slack at the staging point is below this are staged
  env_real("HDSM_CAND_TAU", 0.01, 10.0, &c->cand_tau);
  // step to branch on: 1 = most infeasible segment (measured 4-6x shorter rounds where the search is deep), 0 = first in time
  c->branch_rule = prm->branch_rule == 0 ? 1 : 0;
  env_int("HDSM_BRANCH_RULE", 0, 1, &c->branch_rule);
  // pre-sweep: 0 never, 1 always, 2 automatic (always for swarms below the prefilter size; prefiltered swarms only when the
  // warm start already holds neighbour rows)
  c->presweep = prm->presweep == 0 ? 2 : (prm->presweep == 1 ? 0 : 1);
  env_int("HDSM_PRESWEEP", 0, 2, &c->presweep);
  c->hot_tau = 1e30;  // [m] staged rows closer than this are scanned every iteration ("hot"); the rest only at
                      // convergence. Measured on MI355X: any finite radius costs more iterations than it saves.
  env_real("HDSM_HOT_TAU", 1e-3, 1e30, &c->hot_tau);
  c->pick_rule = 1;
  env_int("HDSM_PICK_RULE", 0, 1, &c->pick_rule);
  c->box_cut = 1;
  env_int("HDSM_BOX_CUT", 0, 1, &c->box_cut);
  c->scanner = 1;
  env_int("HDSM_SCANNER", 0, 2, &c->scanner);
  c->child_bound = 1, c->overlap_sweep = 1;
  env_int("HDSM_OVERLAP_SWEEP", 0, 1, &c->overlap_sweep);
  env_int("HDSM_CHILD_BOUND", 0, 1, &c->child_bound);
  c->dominance = 1;
  env_int("HDSM_DOMINANCE", 0, 1, &c->dominance);
  c->mip_gap = prm->mip_gap;
  c->leaf_mfma = 1;
  env_int("HDSM_LEAF_MFMA", 0, 1, &c->leaf_mfma);
  c->time_ticks = 0;  // set by hdsm_create from time_limit_s and the device's clock rate
  c->r_u = prm->r_u;
  for (int k = 0; k < 6; ++k) c->wx[k] = prm->r_x[k], c->wn[k] = prm->r_n[k];
  for (int ax = 0; ax < 3; ++ax) {
    c->lbu[ax] = prm->u_lb[ax], c->ubu[ax] = prm->u_ub[ax];
    for (int comp = 1; comp < 3; ++comp) {
      c->lbs[comp][ax] = prm->x_lb[3 * comp + ax];
      c->ubs[comp][ax] = prm->x_ub[3 * comp + ax];
    }
  }
  c->radius = prm->drone_radius;
  const double kk = prm->drone_radius / prm->drone_z_offset;
  c->k2m1 = kk * kk - 1.0;
  c->pert = prm->plane_perturb;

  // ---- discrete dynamics per axis. Continuous model (ModelODE, AC:2155-2167): d/dt (p,v,a) = Ac x + Bc u.
  for (int ax = 0; ax < 3; ++ax) {
    M3 Ac{};
    Ac.m[0][1] = 1, Ac.m[1][1] = -prm->drag[ax], Ac.m[1][2] = 1;
    const double dt = prm->dt;
    M3 Ad, Bm;  // x+ = Ad x + (Bm Bc) u
    if (!prm->rk4) {  // forward Euler, AC:2140-2151
      Ad = axpy(eye(), dt, Ac);
      Bm = eye();
      for (auto& row : Bm.m)
        for (double& v : row) v *= dt;
    } else {  // classical RK4 on a linear system with the input held: truncated exponential series
      const M3 A1 = Ac, A2 = mul(Ac, Ac), A3 = mul(A2, Ac), A4 = mul(A3, Ac);
      Ad = axpy(axpy(axpy(axpy(eye(), dt, A1), dt * dt / 2, A2), dt * dt * dt / 6, A3), dt * dt * dt * dt / 24, A4);
      M3 I = eye();
      for (auto& row : I.m)
        for (double& v : row) v *= dt;
      Bm = axpy(axpy(axpy(I, dt * dt / 2, A1), dt * dt * dt / 6, A2), dt * dt * dt * dt / 24, A3);
    }
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) c->Ad[ax][i][j] = Ad.m[i][j];
      c->Bd[ax][i] = Bm.m[i][2];  // Bc = e_3
    }
    M3 pw = eye();
    double v[3] = {c->Bd[ax][0], c->Bd[ax][1], c->Bd[ax][2]};
    for (int i = 0; i <= N; ++i) {
      for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc < 3; ++cc) c->phi[ax][i][r][cc] = pw.m[r][cc];
      pw = mul(Ad, pw);
    }
    for (int lag = 0; lag < N; ++lag) {
      for (int s = 0; s < 3; ++s) c->g[ax][s][lag] = v[s];
      double vn[3];
      for (int r = 0; r < 3; ++r) vn[r] = Ad.m[r][0] * v[0] + Ad.m[r][1] * v[1] + Ad.m[r][2] * v[2];
      for (int r = 0; r < 3; ++r) v[r] = vn[r];
    }
  }
  // p_m = free response + sum_{lag < m} g[ax][0][lag] u: the leading steps whose position no input reaches
  c->pinned_steps = 0;
  for (int m = 1; m <= N; ++m) {
    bool zero = true;
    for (int ax = 0; ax < 3; ++ax) zero = zero && c->g[ax][0][m - 1] == 0.0;
    if (!zero) break;
    c->pinned_steps = m;
  }
  {  // development switch for A/B runs: HDSM_PINNED_STEPS=0 turns the gridlock test of the sweeps off (never raises the count)
    int32_t cap = c->pinned_steps;
    env_int("HDSM_PINNED_STEPS", 0, c->pinned_steps, &cap);
    c->pinned_steps = cap;
  }

  // ---- Hessian of the tracking objective (AC:870-883, AC:2098) in u, its Cholesky factor and inverse
  std::vector<double> H(n * n, 0.0), L(n * n, 0.0);
  for (int ax = 0; ax < 3; ++ax)
    for (int k = 0; k < N; ++k)
      for (int l = 0; l < N; ++l) {
        double h = (k == l) ? 2 * prm->r_u : 0.0;
        for (int i = (k > l ? k : l) + 1; i <= N; ++i) {
          const double* w = (i == N) ? prm->r_n : prm->r_x;
          for (int comp = 0; comp < 2; ++comp)
            h += 2 * w[3 * comp + ax] * c->g[ax][comp][i - 1 - k] * c->g[ax][comp][i - 1 - l];
        }
        H[(ax * N + k) * n + ax * N + l] = h;
      }
  for (int j = 0; j < MAXNV; ++j) {  // row sums of |H| (bound of the objective over the input box), rounded up
    double r = 0;
    for (int k = 0; j < n && k < n; ++k) r += std::fabs(H[j * n + k]);
    c->hrow1[j] = r * (1.0 + 1e-12);
  }
  for (int j = 0; j < n; ++j) {
    double d = H[j * n + j];
    for (int k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k];
    if (!(d > 0)) return fail(err, "Hessian not positive definite");
    d = std::sqrt(d);
    L[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = H[i * n + j];
      for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = s / d;
    }
  }
  for (int j = 0; j < n; ++j)  // J0 = L^{-T}
    for (int i = n - 1; i >= 0; --i) {
      double s = (i == j) ? 1.0 : 0.0;
      for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * c->J0[k * n + j];
      c->J0[i * n + j] = s / L[i * n + i];
    }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double s = 0;
      for (int k = 0; k < n; ++k) s += c->J0[i * n + k] * c->J0[j * n + k];
      c->Hinv[i * n + j] = s;
    }
  // ---- state after the six terminal equalities: row e = (ax = e % 3, comp = 1 + e / 3), normal in u-space
  //      a_e[ax*N + k] = g[ax][comp][N-1-k]; added one after the other with Householder reflections (same update
  //      the device uses), in plain dense fp64.
  {
    std::vector<double> J(c->J0, c->J0 + (size_t)n * n), Rm(36, 0.0), E(6 * n, 0.0);
    for (int e = 0; e < 6; ++e) {
      const int ax = e % 3, comp = 1 + e / 3;
      for (int k = 0; k < N; ++k) E[(size_t)e * n + ax * N + k] = c->g[ax][comp][N - 1 - k];
    }
    for (int q = 0; q < 6; ++q) {
      std::vector<double> d(n, 0.0);
      for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) d[j] -= J[(size_t)i * n + j] * E[(size_t)q * n + i];  // d = J^T np, np = -a (device convention)
      double zz = 0;
      for (int j = q; j < n; ++j) zz += d[j] * d[j];
      if (!(zz > 0)) return fail(err, "terminal equalities are linearly dependent");
      const double rho = (d[q] > 0 ? -1.0 : 1.0) * std::sqrt(zz);
      const double beta = 1.0 / (rho * (rho - d[q]));
      for (int i = 0; i < n; ++i) {  // J2 -= (J2 v) beta v^T, v = d2 - rho e_q
        double w = -rho * J[(size_t)i * n + q];
        for (int j = q; j < n; ++j) w += J[(size_t)i * n + j] * d[j];
        w *= beta;
        for (int j = q; j < n; ++j) J[(size_t)i * n + j] -= w * (d[j] - (j == q ? rho : 0.0));
      }
      for (int i = 0; i < q; ++i) Rm[i * 6 + q] = d[i];
      Rm[q * 6 + q] = rho;
    }
    std::memcpy(c->Jeq, J.data(), sizeof(double) * n * n);
    std::memcpy(c->Req, Rm.data(), sizeof(double) * 36);
    // Ueq = Req^{-1} (upper triangular back substitution per column)
    for (int col = 0; col < 6; ++col)
      for (int i = 5; i >= 0; --i) {
        double sacc = (i == col) ? 1.0 : 0.0;
        for (int k = i + 1; k < 6; ++k) sacc -= Rm[i * 6 + k] * c->Ueq[k * 6 + col];
        c->Ueq[i * 6 + col] = sacc / Rm[i * 6 + i];
      }
    // Seq = (E H^{-1} E^T)^{-1} = Ueq Ueq^T ;  Meq = H^{-1} E^T Seq
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) {
        double sacc = 0;
        for (int k = 0; k < 6; ++k) sacc += c->Ueq[i * 6 + k] * c->Ueq[j * 6 + k];
        c->Seq[i * 6 + j] = sacc;
      }
    std::vector<double> HE((size_t)n * 6, 0.0);  // H^{-1} E^T
    for (int i = 0; i < n; ++i)
      for (int e = 0; e < 6; ++e)
        for (int k = 0; k < n; ++k) HE[(size_t)i * 6 + e] += c->Hinv[(size_t)i * n + k] * E[(size_t)e * n + k];
    for (int i = 0; i < n; ++i)
      for (int e = 0; e < 6; ++e) {
        double sacc = 0;
        for (int k = 0; k < 6; ++k) sacc += HE[(size_t)i * 6 + k] * c->Seq[k * 6 + e];
        c->Meq[(size_t)i * 6 + e] = sacc;
      }
    for (int j = 0; j < MAXNV; ++j)
      for (int lane = 0; lane < 64; ++lane) {
        const bool split = n <= SPLIT_N_MAX;
        if (split && j >= 16) continue;
        // split kernel (hdsm_wave_gib.h): slot j of lane L holds column ((j ^ L) & 15) + 16 (L >> 5) of row L & 31
        // NV = 48: one lane per row, slot 16 b + jj holds column 16 b + ((jj ^ L) & 15)
        const int row = split ? (lane & 31) : lane, col = split ? ((j ^ lane) & 15) + 16 * (lane >> 5) : (j & ~15) + ((j ^ lane) & 15);
        c->JeqP[(size_t)j * 64 + lane] = (row < n && col < n) ? c->Jeq[(size_t)row * n + col] : (row == col ? 1.0 : 0.0);
      }
    {  // pick-rule weights: Z = J2 J2^T (the inverse Hessian on the null space of the terminal equalities), a^T Z a per row family
      std::vector<double> Z((size_t)n * n, 0.0);
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
          double sacc = 0;
          for (int k = 6; k < n; ++k) sacc += c->Jeq[(size_t)i * n + k] * c->Jeq[(size_t)j * n + k];
          Z[(size_t)i * n + j] = sacc;
        }
      auto quad = [&](int ax, int comp, int m) {  // a = impulse response of component comp of axis ax at step m
        double sacc = 0;
        for (int k = 0; k < m; ++k)
          for (int l = 0; l < m; ++l) sacc += c->g[ax][comp][m - 1 - k] * Z[(size_t)(ax * N + k) * n + ax * N + l] * c->g[ax][comp][m - 1 - l];
        return sacc;
      };
      for (int k = 0; k < MAXNV; ++k) c->wu[k] = (k < n && Z[(size_t)k * n + k] > 1e-300) ? 1.0 / std::sqrt(Z[(size_t)k * n + k]) : 1.0;
      for (int ax = 0; ax < 3; ++ax)
        for (int comp = 0; comp < 3; ++comp)
          for (int m = 0; m <= MAXH; ++m) {
            const double dd = (m >= 1 && m <= N) ? quad(ax, comp, m) : 0.0;
            c->ws[ax][comp][m] = dd > 1e-300 ? 1.0 / std::sqrt(dd) : 1.0;
          }
      for (int m = 0; m <= MAXH; ++m)
        for (int ax = 0; ax < 3; ++ax) c->kap[m][ax] = (m >= 1 && m <= N) ? quad(ax, 0, m) : 0.0;
    }
    // Set-up map: free response -> gradient at u = 0 -> x0 = -H^{-1} grad -> equality residual -> x_eq, nu, applied
    // to the unit vectors of v = (state_curr, traj_ref). The tracking cost pairs x_i with ref_{i-1} (AC:870-883).
    const int nv = 9 + 6 * N;
    std::vector<double> v(nv), fr((size_t)3 * (N + 1) * 3), grad(n), x0(n), res(6);
    for (int col = 0; col < nv; ++col) {
      std::fill(v.begin(), v.end(), 0.0);
      v[col] = 1.0;
      const double* st0 = v.data();
      const double* ref = v.data() + 9;  // [N][6]
      for (int ax = 0; ax < 3; ++ax)
        for (int i = 0; i <= N; ++i)
          for (int comp = 0; comp < 3; ++comp) {
            double t = 0;
            for (int cc = 0; cc < 3; ++cc) t += c->phi[ax][i][comp][cc] * st0[3 * cc + ax];
            fr[((size_t)ax * (N + 1) + i) * 3 + comp] = t;
          }
      for (int k = 0; k < n; ++k) {
        const int ax = k / N, kk = k % N;
        double gsum = 0;
        for (int i = kk + 1; i <= N; ++i) {
          const double* w = (i == N) ? c->wn : c->wx;
          for (int comp = 0; comp < 2; ++comp)
            gsum += 2.0 * w[3 * comp + ax] * c->g[ax][comp][i - 1 - kk] *
                    (fr[((size_t)ax * (N + 1) + i) * 3 + comp] - ref[(i - 1) * 6 + 3 * comp + ax]);
        }
        grad[k] = gsum;
      }
      for (int k = 0; k < n; ++k) {
        double t = 0;
        const int j0 = (k / N) * N;
        for (int j = j0; j < j0 + N; ++j) t -= c->Hinv[(size_t)k * n + j] * grad[j];
        x0[k] = t;
      }
      for (int e = 0; e < 6; ++e) {
        const int ax = e % 3, comp = 1 + e / 3;
        double t = fr[((size_t)ax * (N + 1) + N) * 3 + comp];
        for (int k = 0; k < N; ++k) t += c->g[ax][comp][N - 1 - k] * x0[ax * N + k];
        res[e] = t;
      }
      double* out = c->KT + (size_t)col * KROWS;
      for (int k = 0; k < n; ++k) {
        double t = x0[k];
        for (int e = 0; e < 6; ++e) t -= c->Meq[(size_t)k * 6 + e] * res[e];
        out[k] = t, out[n + k] = x0[k], out[2 * n + k] = grad[k];
      }
      for (int e = 0; e < 6; ++e) {
        double nu = 0;
        for (int k = 0; k < 6; ++k) nu += c->Seq[e * 6 + k] * res[k];
        out[3 * n + e] = res[e], out[3 * n + 6 + e] = nu;
      }
    }
    const int nk = 3 * n + 12;
    double off_axis = 0.0, on_axis = 0.0;
    for (int row = 0; row < nk; ++row) {
      const int ax = (row < 3 * n) ? (row % n) / N : (row - 3 * n) % 3;
      c->kax[row] = ax;
      for (int u = 0; u < 3 + 2 * N; ++u) c->KTC[(size_t)u * KROWS + row] = c->KT[(size_t)(ax + 3 * u) * KROWS + row];
      for (int col = 0; col < nv; ++col) {
        const double v = std::fabs(c->KT[(size_t)col * KROWS + row]);
        if (col % 3 == ax) on_axis = std::fmax(on_axis, v);
        else off_axis = std::fmax(off_axis, v);
      }
    }
    if (!(off_axis <= 1e-9 * on_axis)) {  // cannot happen for per-axis dynamics and weights; guards the compact form
      if (err) *err = "internal: set-up map is not axis-separable";
      return HDSM_ERR_BAD_ARG;
    }
  }
  return HDSM_OK;
}

}  // namespace hdsm
