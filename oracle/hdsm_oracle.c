/*
 * hdsm_oracle.c — CPU oracle for the HDSM hot path.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * See hdsm_oracle.h for the rules of use and the "PARITY UNPINNED" statement.
 *
 * A restatement, in plain C99 + libm, of
 *   - Agent::CreateGurobiModel / ModelODE            AC:2071-2167   (variables, bounds, dynamics)
 *   - Agent::SolveOptimizationProblem                AC:858-1023    (objective, corridor rows, read-back)
 *   - Agent::GenerateTimeAwareSafeCorridor           AC:1086-1234   (separating planes)
 * of lis-epfl/multi_agent_pkgs (AC = multi_agent_planner/src/agent_class.cpp), with Gurobi's MIQP
 * branch-and-bound replaced by an exact dense dual active-set QP (Goldfarb & Idnani, Math. Prog. 27,
 * 1983 — restated from the paper) under a depth-first branch-and-bound over the one-hot polyhedron choice.
 *
 * Deliberately simple and dense: cold-started QPs, no factor reuse between nodes. Correctness first.
 */
#include "hdsm_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define MAXH HDSM_MAX_HOR
#define ON (3 * HDSM_MAX_HOR)
#define ABSENT 1e20

/* ============================================================================================ dynamics */

/* ModelODE, AC:2155-2167, one axis: d/dt (p, v, a) = (v, a - D v, u). */
static void ode_axis(const hdsm_params* prm, int ax, const double x[3], double u, double f[3]) {
  f[0] = x[1];
  f[1] = x[2] - prm->drag[ax] * x[1];
  f[2] = u;
}

void orc_step_axis(const hdsm_params* prm, int ax, const double x[3], double u, double xn[3]) {
  const double dt = prm->dt;
  double k1[3], k2[3], k3[3], k4[3], tmp[3];
  ode_axis(prm, ax, x, u, k1);
  if (!prm->rk4) { /* AC:2140-2151: x_{i+1} = x_i + dt * k1 */
    for (int s = 0; s < 3; s++) xn[s] = x[s] + dt * k1[s];
    return;
  }
  /* AC:2123-2139 */
  for (int s = 0; s < 3; s++) tmp[s] = x[s] + (dt / 2) * k1[s];
  ode_axis(prm, ax, tmp, u, k2);
  for (int s = 0; s < 3; s++) tmp[s] = x[s] + (dt / 2) * k2[s];
  ode_axis(prm, ax, tmp, u, k3);
  for (int s = 0; s < 3; s++) tmp[s] = x[s] + dt * k3[s];
  ode_axis(prm, ax, tmp, u, k4);
  for (int s = 0; s < 3; s++) xn[s] = x[s] + dt * ((k1[s] + 2 * k2[s] + 2 * k3[s] + k4[s]) / 6);
}

void orc_rollout(const hdsm_params* prm, const double state_curr[9], const double* ctrl, double* traj) {
  const int N = prm->n_hor;
  memcpy(traj, state_curr, 9 * sizeof(double));
  for (int i = 0; i < N; i++)
    for (int ax = 0; ax < 3; ax++) {
      double x[3] = {traj[9 * i + ax], traj[9 * i + 3 + ax], traj[9 * i + 6 + ax]}, xn[3];
      orc_step_axis(prm, ax, x, ctrl[3 * i + ax], xn);
      for (int s = 0; s < 3; s++) traj[9 * (i + 1) + 3 * s + ax] = xn[s];
    }
}

double orc_objective(const hdsm_params* prm, const double* traj, const double* ctrl,
                     const double* traj_ref) {
  const int N = prm->n_hor;
  double J = 0;
  for (int i = 0; i < N; i++) /* AC:2098 */
    for (int k = 0; k < 3; k++) J += prm->r_u * ctrl[3 * i + k] * ctrl[3 * i + k];
  for (int i = 1; i <= N; i++) { /* AC:871-883: x_i tracks ref row i-1, first 6 components only */
    const double* w = (i == N) ? prm->r_n : prm->r_x;
    for (int k = 0; k < 6; k++) {
      double e = traj[9 * i + k] - traj_ref[6 * (i - 1) + k];
      J += w[k] * e * e;
    }
  }
  return J;
}

/* Condensed maps obtained by SIMULATION of the literal step (superposition of unit responses):
 *   state s of axis ax at step i  =  sum_c Phi[ax][i][s][c] x0[c]  +  sum_{k<i} Gam[ax][i][k][s] u_k      */
typedef struct {
  int N, n;
  double Phi[3][MAXH + 1][3][3];
  double Gam[3][MAXH + 1][MAXH][3];
  double H[ON * ON];  /* Hessian of J in u (instance independent)                                          */
  double J0[ON * ON]; /* L^{-T} where H = L L^T                                                            */
  double Hinv[ON * ON];
} shared_t;

static int chol_lower(int n, const double* A, double* L) {
  memset(L, 0, sizeof(double) * n * n);
  for (int j = 0; j < n; j++) {
    double d = A[j * n + j];
    for (int k = 0; k < j; k++) d -= L[j * n + k] * L[j * n + k];
    if (!(d > 0)) return -1;
    d = sqrt(d);
    L[j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = A[i * n + j];
      for (int k = 0; k < j; k++) s -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = s / d;
    }
  }
  return 0;
}

static int build_shared(const hdsm_params* prm, shared_t* sh) {
  const int N = prm->n_hor, n = 3 * N;
  if (N < 1 || N > MAXH) return -1;
  sh->N = N;
  sh->n = n;
  memset(sh->Phi, 0, sizeof sh->Phi);
  memset(sh->Gam, 0, sizeof sh->Gam);
  for (int ax = 0; ax < 3; ax++) {
    for (int c = 0; c < 3; c++) { /* unit initial state, zero input */
      double x[3] = {0, 0, 0}, xn[3];
      x[c] = 1;
      for (int s = 0; s < 3; s++) sh->Phi[ax][0][s][c] = x[s];
      for (int i = 0; i < N; i++) {
        orc_step_axis(prm, ax, x, 0.0, xn);
        memcpy(x, xn, sizeof x);
        for (int s = 0; s < 3; s++) sh->Phi[ax][i + 1][s][c] = x[s];
      }
    }
    for (int k = 0; k < N; k++) { /* zero initial state, unit input at step k */
      double x[3] = {0, 0, 0}, xn[3];
      for (int i = 0; i < N; i++) {
        orc_step_axis(prm, ax, x, i == k ? 1.0 : 0.0, xn);
        memcpy(x, xn, sizeof x);
        for (int s = 0; s < 3; s++) sh->Gam[ax][i + 1][k][s] = x[s];
      }
    }
  }
  /* H = d2J/du2: 2 r_u I + sum_i 2 w (dstate/du)(dstate/du)^T over tracked components p (0..2), v (3..5) */
  memset(sh->H, 0, sizeof(double) * n * n);
  for (int j = 0; j < n; j++) sh->H[j * n + j] = 2 * prm->r_u;
  for (int i = 1; i <= N; i++) {
    const double* w = (i == N) ? prm->r_n : prm->r_x;
    for (int ax = 0; ax < 3; ax++)
      for (int s = 0; s < 2; s++) {
        double wk = w[3 * s + ax];
        if (wk == 0) continue;
        for (int k = 0; k < i; k++)
          for (int l = 0; l < i; l++)
            sh->H[(ax * N + k) * n + ax * N + l] += 2 * wk * sh->Gam[ax][i][k][s] * sh->Gam[ax][i][l][s];
      }
  }
  double* L = (double*)malloc(sizeof(double) * n * n);
  if (chol_lower(n, sh->H, L)) {
    free(L);
    return -1;
  }
  /* J0 = L^{-T}: column j solves L^T y = e_j */
  memset(sh->J0, 0, sizeof(double) * n * n);
  for (int j = 0; j < n; j++) {
    for (int i = n - 1; i >= 0; i--) {
      double s = (i == j) ? 1.0 : 0.0;
      for (int k = i + 1; k < n; k++) s -= L[k * n + i] * sh->J0[k * n + j];
      sh->J0[i * n + j] = s / L[i * n + i];
    }
  }
  /* Hinv = J0 J0^T */
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) {
      double s = 0;
      for (int k = 0; k < n; k++) s += sh->J0[i * n + k] * sh->J0[j * n + k];
      sh->Hinv[i * n + j] = s;
    }
  free(L);
  return 0;
}

/* ====================================================================================== TASC planes */

void orc_tasc_plane(const hdsm_params* prm, const double c[3], const double o[3], double out[4]) {
  /* AC:1151-1152 plane_normal = pos_other - pos_curr; Eigen normalized() leaves a zero vector as is. */
  double d[3] = {o[0] - c[0], o[1] - c[1], o[2] - c[2]};
  double nrm2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  double nrm = sqrt(nrm2);
  double nh[3] = {d[0], d[1], d[2]};
  if (nrm2 > 0)
    for (int k = 0; k < 3; k++) nh[k] = d[k] / nrm;
  double mid[3] = {(c[0] + o[0]) / 2, (c[1] + o[1]) / 2, (c[2] + o[2]) / 2}; /* AC:1155 */
  /* AC:1158-1164 ellipsoid support distance */
  double angle_x_axis = M_PI_2 - fabs(acos(nh[2]));
  double t_val = atan(prm->drone_radius / prm->drone_z_offset * tan(angle_x_axis));
  double x_val = prm->drone_radius * cos(t_val);
  double y_val = prm->drone_z_offset * sin(t_val);
  double safety_dist = hypot(x_val, y_val);
  /* AC:1167-1169 */
  double back = fmin(2 * safety_dist, nrm) / 2;
  double q[3] = {mid[0] - back * nh[0], mid[1] - back * nh[1], mid[2] - back * nh[2]};
  /* AC:1173-1177: right = n x (0,0,1) + n x (0,1,0); up_final = n x (0,1,0) */
  double c1[3] = {nh[1], -nh[0], 0.0};    /* n x (0,0,1) */
  double c2[3] = {-nh[2], 0.0, nh[0]};    /* n x (0,1,0) */
  double var_tmp = prm->plane_perturb, pert = 0.0; /* AC:1180, AC:1197 */
  for (int k = 0; k < 3; k++) out[k] = (var_tmp + pert) * (c1[k] + c2[k]) + var_tmp * c2[k] + nh[k];
  out[3] = out[0] * q[0] + out[1] * q[1] + out[2] * q[2]; /* AddHyperplane AC:1227-1228 */
}

void orc_tasc_planes(const hdsm_params* prm, int n_rob, int agent_id, const double state_curr[9],
                     const double* plans_all, const uint8_t* has_plan, double* planes, uint8_t* valid) {
  const int N = prm->n_hor;
  const int own_has = (agent_id >= 0 && agent_id < n_rob) ? has_plan[agent_id] : 0;
  for (int i = 0; i < N; i++) {
    double c[3];
    if (own_has) { /* AC:1103-1107 traj_curr_[i+1] */
      const double* st = plans_all + ((size_t)agent_id * (N + 1) + (i + 1)) * 9;
      c[0] = st[0], c[1] = st[1], c[2] = st[2];
    } else { /* AC:1109 state_ini_ (== state_curr_ while no plan exists) */
      c[0] = state_curr[0], c[1] = state_curr[1], c[2] = state_curr[2];
    }
    for (int j = 0; j < n_rob; j++) {
      double* out = planes + ((size_t)i * n_rob + j) * 4;
      out[0] = out[1] = out[2] = out[3] = 0;
      valid[(size_t)i * n_rob + j] = 0;
      if (j == agent_id || !has_plan[j]) continue; /* AC:617 (no self subscription), AC:1134 */
      const double* so = plans_all + ((size_t)j * (N + 1) + (i + 1)) * 9; /* AC:1147-1149 */
      orc_tasc_plane(prm, c, so, out);
      valid[(size_t)i * n_rob + j] = 1;
    }
  }
}

/* ==================================================================================== structured QP */

enum { K_UBOX = 0, K_SBOX = 1, K_PLANE = 2, K_EQ = 3 };
typedef struct {
  int kind, step, ax, comp;
  double sgn, nrm[3], rhs; /* sgn * value <= rhs  |  nrm . p_step <= rhs  |  value == rhs */
} con_t;

typedef struct {
  const hdsm_params* prm;
  const shared_t* sh;
  double fr[3][MAXH + 1][3]; /* free response                                                             */
  double g[ON], f0, x0[ON];  /* J(u) = 1/2 u'Hu + g'u + f0 ; x0 = -H^{-1} g                               */
} inst_t;

static void build_inst(const hdsm_params* prm, const shared_t* sh, const double* state,
                       const double* ref, inst_t* in) {
  const int N = sh->N, n = sh->n;
  in->prm = prm;
  in->sh = sh;
  for (int ax = 0; ax < 3; ax++)
    for (int i = 0; i <= N; i++)
      for (int s = 0; s < 3; s++) {
        double v = 0;
        for (int c = 0; c < 3; c++) v += sh->Phi[ax][i][s][c] * state[3 * c + ax];
        in->fr[ax][i][s] = v;
      }
  memset(in->g, 0, sizeof in->g);
  in->f0 = 0;
  for (int i = 1; i <= N; i++) {
    const double* w = (i == N) ? prm->r_n : prm->r_x;
    for (int ax = 0; ax < 3; ax++)
      for (int s = 0; s < 2; s++) {
        double wk = w[3 * s + ax];
        double e = in->fr[ax][i][s] - ref[6 * (i - 1) + 3 * s + ax];
        in->f0 += wk * e * e;
        for (int k = 0; k < i; k++) in->g[ax * N + k] += 2 * wk * sh->Gam[ax][i][k][s] * e;
      }
  }
  for (int i = 0; i < n; i++) {
    double s = 0;
    for (int j = 0; j < n; j++) s -= sh->Hinv[i * n + j] * in->g[j];
    in->x0[i] = s;
  }
}

static void states_from_u(const inst_t* in, const double* u, double st[3][MAXH + 1][3]) {
  const shared_t* sh = in->sh;
  const int N = sh->N;
  for (int ax = 0; ax < 3; ax++)
    for (int i = 0; i <= N; i++)
      for (int s = 0; s < 3; s++) {
        double v = in->fr[ax][i][s];
        for (int k = 0; k < i; k++) v += sh->Gam[ax][i][k][s] * u[ax * N + k];
        st[ax][i][s] = v;
      }
}

/* value - rhs (positive = violated for inequalities) */
static double con_resid(const con_t* c, const inst_t* in, const double* u, double st[3][MAXH + 1][3]) {
  const int N = in->sh->N;
  switch (c->kind) {
    case K_UBOX: return c->sgn * u[c->ax * N + c->step] - c->rhs;
    case K_SBOX:
    case K_EQ: return c->sgn * st[c->ax][c->step][c->comp] - c->rhs;
    default:
      return c->nrm[0] * st[0][c->step][0] + c->nrm[1] * st[1][c->step][0] +
             c->nrm[2] * st[2][c->step][0] - c->rhs;
  }
}

static void con_normal(const con_t* c, const inst_t* in, double* a) {
  const shared_t* sh = in->sh;
  const int N = sh->N, n = sh->n;
  memset(a, 0, sizeof(double) * n);
  switch (c->kind) {
    case K_UBOX: a[c->ax * N + c->step] = c->sgn; break;
    case K_SBOX:
    case K_EQ:
      for (int k = 0; k < c->step; k++) a[c->ax * N + k] = c->sgn * sh->Gam[c->ax][c->step][k][c->comp];
      break;
    default:
      for (int ax = 0; ax < 3; ax++)
        for (int k = 0; k < c->step; k++) a[ax * N + k] = c->nrm[ax] * sh->Gam[ax][c->step][k][0];
  }
}

/* ---- Goldfarb-Idnani dual active set ------------------------------------------------------------------ */
typedef struct {
  int n, q;
  double J[ON * ON], R[ON * ON];
  double x[ON], lam[ON], f;
  int act[ON];
  int iters;
} gi_t;

static void gi_add_col(gi_t* s, double* d) {
  const int n = s->n, q = s->q;
  for (int j = n - 1; j > q; j--) { /* rotate d[j] into d[j-1], same rotation on the columns of J */
    double h = hypot(d[j - 1], d[j]);
    if (h == 0) continue;
    double c = d[j - 1] / h, sn = d[j] / h;
    d[j - 1] = h;
    d[j] = 0;
    for (int k = 0; k < n; k++) {
      double t1 = s->J[k * n + j - 1], t2 = s->J[k * n + j];
      s->J[k * n + j - 1] = c * t1 + sn * t2;
      s->J[k * n + j] = -sn * t1 + c * t2;
    }
  }
  for (int i = 0; i <= q; i++) s->R[i * n + q] = d[i];
  s->q = q + 1;
}

static void gi_drop(gi_t* s, int l) {
  const int n = s->n;
  int q = s->q;
  for (int j = l; j < q - 1; j++) {
    for (int i = 0; i <= j + 1; i++) s->R[i * n + j] = s->R[i * n + j + 1];
    s->act[j] = s->act[j + 1];
    s->lam[j] = s->lam[j + 1];
  }
  q--;
  s->q = q;
  for (int j = l; j < q; j++) { /* remove the sub-diagonal R[j+1][j] */
    double h = hypot(s->R[j * n + j], s->R[(j + 1) * n + j]);
    if (h == 0) continue;
    double c = s->R[j * n + j] / h, sn = s->R[(j + 1) * n + j] / h;
    for (int k = j; k < q; k++) {
      double t1 = s->R[j * n + k], t2 = s->R[(j + 1) * n + k];
      s->R[j * n + k] = c * t1 + sn * t2;
      s->R[(j + 1) * n + k] = -sn * t1 + c * t2;
    }
    for (int k = 0; k < n; k++) {
      double t1 = s->J[k * n + j], t2 = s->J[k * n + j + 1];
      s->J[k * n + j] = c * t1 + sn * t2;
      s->J[k * n + j + 1] = -sn * t1 + c * t2;
    }
  }
}

enum { GI_OK = 0, GI_INFEASIBLE = 1, GI_CUTOFF = 2, GI_ITERLIM = 3 };

/* Solve min J(u) s.t. cons[0..m). Equalities must come first in `cons`. Stops early with GI_CUTOFF as soon
 * as the (monotonically increasing) dual objective reaches f_cut.                                         */
static int gi_solve(gi_t* s, const inst_t* in, const con_t* cons, int m, double tol, double f_cut,
                    int iter_budget) {
  const shared_t* sh = in->sh;
  const int n = sh->n;
  double a[ON], d[ON], z[ON], r[ON], st[3][MAXH + 1][3];
  s->n = n;
  s->q = 0;
  memcpy(s->J, sh->J0, sizeof(double) * n * n);
  memset(s->R, 0, sizeof(double) * n * n);
  memcpy(s->x, in->x0, sizeof(double) * n);
  s->f = in->f0;
  for (int i = 0; i < n; i++) s->f += 0.5 * in->g[i] * in->x0[i];
  s->iters = 0;

  for (;;) {
    /* pick the constraint to add: pending equalities in order, then the most violated inequality */
    int ip = -1;
    double vmax = tol, v_ip = 0;
    states_from_u(in, s->x, st);
    for (int c = 0; c < m && ip < 0; c++)
      if (cons[c].kind == K_EQ) {
        int active = 0;
        for (int k = 0; k < s->q; k++) active |= (s->act[k] == c);
        if (!active) {
          ip = c;
          v_ip = con_resid(&cons[c], in, s->x, st);
        }
      }
    if (ip < 0) {
      for (int c = 0; c < m; c++) {
        if (cons[c].kind == K_EQ) continue;
        double v = con_resid(&cons[c], in, s->x, st);
        if (v > vmax) {
          vmax = v;
          ip = c;
          v_ip = v;
        }
      }
      if (ip < 0) return GI_OK;
    }
    const int is_eq = cons[ip].kind == K_EQ;
    con_normal(&cons[ip], in, a);
    double lam_p = 0; /* multiplier of the incoming constraint */

    for (;;) { /* step loop for constraint ip (GI step 2) */
      if (++s->iters > iter_budget) return GI_ITERLIM;
      const int q = s->q;
      /* d = J^T np, np = -a */
      for (int j = 0; j < n; j++) {
        double t = 0;
        for (int i = 0; i < n; i++) t -= s->J[i * n + j] * a[i];
        d[j] = t;
      }
      double zz = 0, dd = 0;
      for (int j = 0; j < n; j++) dd += d[j] * d[j];
      for (int j = q; j < n; j++) zz += d[j] * d[j];
      for (int i = 0; i < n; i++) {
        double t = 0;
        for (int j = q; j < n; j++) t += s->J[i * n + j] * d[j];
        z[i] = t;
      }
      for (int i = q - 1; i >= 0; i--) { /* r = R^{-1} d[0..q) */
        double t = d[i];
        for (int j = i + 1; j < q; j++) t -= s->R[i * n + j] * r[j];
        r[i] = t / s->R[i * n + i];
      }
      const int dependent = !(zz > 1e-20 * dd) || q >= n;
      /* largest dual step keeping the multipliers of active INEQUALITIES non-negative */
      double t1 = INFINITY;
      int l = -1;
      if (!is_eq) /* equalities are added first, while only equalities are active */
        for (int k = 0; k < q; k++) {
          if (cons[s->act[k]].kind == K_EQ) continue;
          if (r[k] > 0) {
            double t = s->lam[k] / r[k];
            if (t < t1) {
              t1 = t;
              l = k;
            }
          }
        }
      if (dependent && l < 0) return GI_INFEASIBLE; /* no primal step and no dual step: infeasible */
      if (dependent) { /* dual step only, then drop constraint l and retry */
        for (int k = 0; k < q; k++) s->lam[k] -= t1 * r[k];
        lam_p += t1;
        gi_drop(s, l);
        continue;
      }
      const double t2 = v_ip / zz; /* full primal step (either sign for an equality) */
      const int full = is_eq || t2 <= t1;
      const double t = full ? t2 : t1;
      for (int i = 0; i < n; i++) s->x[i] += t * z[i];
      s->f += t * zz * (0.5 * t + lam_p);
      for (int k = 0; k < q; k++) s->lam[k] -= t * r[k];
      lam_p += t;
      if (full) { /* the constraint becomes active */
        gi_add_col(s, d);
        s->act[s->q - 1] = ip;
        s->lam[s->q - 1] = lam_p;
        break;
      }
      /* partial step: constraint l leaves, ip is still violated */
      gi_drop(s, l);
      states_from_u(in, s->x, st);
      v_ip = con_resid(&cons[ip], in, s->x, st);
      if (s->f >= f_cut) return GI_CUTOFF;
    }
    if (s->f >= f_cut) return GI_CUTOFF;
  }
}

/* =============================================================================== branch and bound */

typedef struct {
  const hdsm_params* prm;
  const inst_t* in;
  const orc_corridor* cor;
  con_t* cons;
  int ncons, cap;
  double tol, ftol_fixed;
  int pinned; /* p_0 .. p_pinned do not depend on the inputs (p_0 always; with jerk inputs and the Euler step also p_1, p_2): a row
               * there is a constant, judged like a row on p_0 — within feas_tol_fixed it holds, beyond it nothing can satisfy it */
  int have_inc, limit_hit;
  double inc_f, inc_u[ON], second;
  int inc_assign[MAXH], assign[MAXH];
  int nodes, qp_solves, qp_iters, max_nodes, iter_budget;
  double cut0; /* verification mode: an upper bound on the optimum known beforehand (INFINITY = none) */
  gi_t gi;
} bnb_t;

/* objective value at which a node is cut off: the incumbent (minus the exactness margin) or the caller's bound */
static double bnb_cut(const bnb_t* b) {
  double c = b->have_inc ? b->inc_f - 1e-9 * fmax(1.0, fabs(b->inc_f)) : INFINITY;
  return c < b->cut0 ? c : b->cut0;
}

static void push_con(bnb_t* b, const con_t* c) {
  if (b->ncons == b->cap) {
    b->cap = b->cap ? 2 * b->cap : 1024;
    b->cons = (con_t*)realloc(b->cons, sizeof(con_t) * b->cap);
  }
  b->cons[b->ncons++] = *c;
}

/* number of leading steps whose POSITION no input reaches (Gam[ax][m][k][0] == 0 for every k < m and every axis) */
static int pinned_steps(const shared_t* sh) {
  int pinned = 0;
  for (int m = 1; m <= sh->N; m++) {
    int zero = 1;
    for (int ax = 0; ax < 3; ax++)
      for (int k = 0; k < m; k++) zero = zero && sh->Gam[ax][m][k][0] == 0.0;
    if (!zero) break;
    pinned = m;
  }
  return pinned;
}

/* Constraints that hold whatever the assignment: terminal equalities (AC:2078-2081), input box
 * (AC:2185-2186), v/a boxes on x_1..x_{N-1} (AC:2084, AC:2179-2184), common planes on p_1..p_N.
 * Returns 0 if a FIXED row (on the pinned p_0) is violated beyond ftol_fixed -> infeasible. */
static int build_base(bnb_t* b) {
  const hdsm_params* prm = b->prm;
  const int N = prm->n_hor;
  con_t c;
  memset(&c, 0, sizeof c);
  for (int ax = 0; ax < 3; ax++)
    for (int comp = 1; comp <= 2; comp++) {
      c.kind = K_EQ, c.step = N, c.ax = ax, c.comp = comp, c.sgn = 1, c.rhs = 0;
      push_con(b, &c);
    }
  for (int k = 0; k < N; k++)
    for (int ax = 0; ax < 3; ax++) {
      c.kind = K_UBOX, c.step = k, c.ax = ax, c.comp = 0;
      if (fabs(prm->u_ub[ax]) < ABSENT) c.sgn = 1, c.rhs = prm->u_ub[ax], push_con(b, &c);
      if (fabs(prm->u_lb[ax]) < ABSENT) c.sgn = -1, c.rhs = -prm->u_lb[ax], push_con(b, &c);
    }
  for (int i = 1; i < N; i++)
    for (int ax = 0; ax < 3; ax++)
      for (int comp = 1; comp <= 2; comp++) {
        double ub = prm->x_ub[3 * comp + ax], lb = prm->x_lb[3 * comp + ax];
        c.kind = K_SBOX, c.step = i, c.ax = ax, c.comp = comp;
        if (fabs(ub) < ABSENT) c.sgn = 1, c.rhs = ub, push_con(b, &c);
        if (fabs(lb) < ABSENT) c.sgn = -1, c.rhs = -lb, push_con(b, &c);
      }
  for (int i = 0; i < N; i++)
    for (int r = 0; r < b->cor->ncommon[i]; r++) {
      const double* row = b->cor->common[i] + 4 * r;
      for (int e = 0; e < 2; e++) {
        int mstep = i + e;
        if (mstep <= b->pinned) { /* pinned point: constant row */
          double v = row[0] * b->in->fr[0][mstep][0] + row[1] * b->in->fr[1][mstep][0] +
                     row[2] * b->in->fr[2][mstep][0] - row[3];
          if (v > b->ftol_fixed) return 0;
          continue;
        }
        c.kind = K_PLANE, c.step = mstep, c.ax = 0, c.comp = 0, c.sgn = 1;
        c.nrm[0] = row[0], c.nrm[1] = row[1], c.nrm[2] = row[2], c.rhs = row[3];
        push_con(b, &c);
      }
    }
  return 1;
}

/* rows of polyhedron (i, j) on p_i and p_{i+1}; returns 0 if a fixed row (p_0) is violated */
static int push_poly(bnb_t* b, int i, int j) {
  const orc_corridor* cor = b->cor;
  con_t c;
  memset(&c, 0, sizeof c);
  for (int r = 0; r < cor->nrows[i][j]; r++) {
    const double* A = cor->A[i][j] + 3 * r;
    double rhs = cor->b[i][j][r];
    for (int e = 0; e < 2; e++) {
      int mstep = i + e;
      if (mstep <= b->pinned) {
        double v = A[0] * b->in->fr[0][mstep][0] + A[1] * b->in->fr[1][mstep][0] + A[2] * b->in->fr[2][mstep][0] - rhs;
        if (v > b->ftol_fixed) return 0;
        continue;
      }
      c.kind = K_PLANE, c.step = mstep, c.sgn = 1;
      c.nrm[0] = A[0], c.nrm[1] = A[1], c.nrm[2] = A[2], c.rhs = rhs;
      push_con(b, &c);
    }
  }
  return 1;
}

/* max over rows of polyhedron (i,j) of (A p - b) at p_i, p_{i+1}; rows on the pinned p_0 use ftol_fixed:
 * returns INFINITY if p_0 lies outside. */
static double poly_violation(const bnb_t* b, int i, int j, double st[3][MAXH + 1][3]) {
  const orc_corridor* cor = b->cor;
  double vmax = -INFINITY;
  for (int r = 0; r < cor->nrows[i][j]; r++) {
    const double* A = cor->A[i][j] + 3 * r;
    double rhs = cor->b[i][j][r];
    for (int e = 0; e < 2; e++) {
      int mstep = i + e;
      double v = A[0] * st[0][mstep][0] + A[1] * st[1][mstep][0] + A[2] * st[2][mstep][0] - rhs;
      if (mstep <= b->pinned) {
        if (v > b->ftol_fixed) return INFINITY;
        continue;
      }
      if (v > vmax) vmax = v;
    }
  }
  return vmax;
}

static int solve_node(bnb_t* b, double* u, double* f) {
  b->qp_solves++;
  int rc = gi_solve(&b->gi, b->in, b->cons, b->ncons, b->tol, bnb_cut(b), b->iter_budget - b->qp_iters);
  b->qp_iters += b->gi.iters;
  if (rc == GI_ITERLIM) b->limit_hit = 1;
  if (rc == GI_CUTOFF && b->gi.f < b->second) b->second = b->gi.f;
  if (rc != GI_OK) return rc;
  memcpy(u, b->gi.x, sizeof(double) * b->gi.n);
  *f = b->gi.f;
  return GI_OK;
}

static void bnb_node(bnb_t* b, int depth, const double* u, double f) {
  const int N = b->prm->n_hor;
  if (b->limit_hit) return;
  if (++b->nodes > b->max_nodes) {
    b->limit_hit = 1;
    return;
  }
  if (f >= bnb_cut(b)) {
    if (f < b->second) b->second = f;
    return;
  }
  if (depth == N) {
    if (b->have_inc && b->inc_f < b->second) b->second = b->inc_f;
    b->have_inc = 1;
    b->inc_f = f;
    memcpy(b->inc_u, u, sizeof(double) * 3 * N);
    memcpy(b->inc_assign, b->assign, sizeof(int) * N);
    return;
  }
  double st[3][MAXH + 1][3];
  states_from_u(b->in, u, st);
  const int m = b->cor->m[depth];
  double key[HDSM_MAX_POLY];
  int order[HDSM_MAX_POLY];
  for (int j = 0; j < m; j++) {
    double v = poly_violation(b, depth, j, st);
    key[j] = (v <= b->tol) ? 0.0 : v; /* containing polyhedra tie at 0 -> lowest index first */
    order[j] = j;
  }
  for (int x = 1; x < m; x++) /* stable insertion sort by key */
    for (int y = x; y > 0 && key[order[y]] < key[order[y - 1]]; y--) {
      int t = order[y];
      order[y] = order[y - 1];
      order[y - 1] = t;
    }
  for (int o = 0; o < m; o++) {
    const int j = order[o];
    if (key[j] == INFINITY) continue; /* pinned point outside polyhedron j */
    if (b->limit_hit) return;
    if (f >= bnb_cut(b)) return;
    b->assign[depth] = j;
    const int mark = b->ncons;
    if (!push_poly(b, depth, j)) {
      b->ncons = mark;
      continue;
    }
    if (key[j] == 0.0) {
      /* the node's minimiser already satisfies polyhedron j: it is the child's minimiser too */
      bnb_node(b, depth + 1, u, f);
    } else {
      double uc[ON], fc;
      int rc = solve_node(b, uc, &fc);
      if (rc == GI_OK) bnb_node(b, depth + 1, uc, fc);
    }
    b->ncons = mark;
  }
}

/* Second search order (orc_replan_ex, search = 1): a node branches only on a step whose segment lies in NO polyhedron at
 * the node's own minimiser — the most infeasible such step, children by ascending violation — and a node whose segments
 * are all contained is a leaf whatever its depth (unassigned steps take the lowest-index containing polyhedron). Exactness
 * does not depend on the order; this one proves optimality on trees whose step-ordered enumeration exceeds any budget
 * (many steps with two or three near-equivalent polyhedra). b->assign[i] = -1 marks an unassigned step. */
static void bnb_lazy(bnb_t* b, const double* u, double f) {
  const int N = b->prm->n_hor;
  if (b->limit_hit) return;
  if (++b->nodes > b->max_nodes) {
    b->limit_hit = 1;
    return;
  }
  if (f >= bnb_cut(b)) {
    if (f < b->second) b->second = f;
    return;
  }
  double st[3][MAXH + 1][3];
  states_from_u(b->in, u, st);
  int pick = -1, contain[MAXH];
  double worst = -INFINITY, keys[MAXH][HDSM_MAX_POLY];
  for (int i = N - 1; i >= 0; i--) {
    contain[i] = b->assign[i];
    if (b->assign[i] >= 0) continue;
    double best = INFINITY;
    for (int j = 0; j < b->cor->m[i]; j++) {
      keys[i][j] = poly_violation(b, i, j, st);
      if (keys[i][j] < best) best = keys[i][j];
    }
    for (int j = 0; j < b->cor->m[i]; j++)
      if (keys[i][j] <= b->tol) {
        contain[i] = j;
        break;
      }
    if (contain[i] < 0 && best >= worst) worst = best, pick = i; /* ties: the earlier step */
  }
  if (pick < 0) {
    if (b->have_inc && b->inc_f < b->second) b->second = b->inc_f;
    b->have_inc = 1;
    b->inc_f = f;
    memcpy(b->inc_u, u, sizeof(double) * 3 * N);
    memcpy(b->inc_assign, contain, sizeof(int) * N);
    return;
  }
  const int m = b->cor->m[pick];
  int order[HDSM_MAX_POLY];
  for (int j = 0; j < m; j++) order[j] = j;
  for (int x = 1; x < m; x++)
    for (int y = x; y > 0 && keys[pick][order[y]] < keys[pick][order[y - 1]]; y--) {
      int t = order[y];
      order[y] = order[y - 1];
      order[y - 1] = t;
    }
  for (int o = 0; o < m; o++) {
    const int j = order[o];
    if (keys[pick][j] == INFINITY) continue;
    if (b->limit_hit) break;
    if (f >= bnb_cut(b)) break;
    b->assign[pick] = j;
    const int mark = b->ncons;
    if (push_poly(b, pick, j)) {
      double uc[ON], fc;
      if (solve_node(b, uc, &fc) == GI_OK) bnb_lazy(b, uc, fc);
    }
    b->ncons = mark;
  }
  b->assign[pick] = -1;
}

static void finish(const hdsm_params* prm, const double* state, const double* ref, const double* u,
                   double* traj, double* ctrl, double* obj) {
  const int N = prm->n_hor;
  for (int i = 0; i < N; i++)
    for (int ax = 0; ax < 3; ax++) ctrl[3 * i + ax] = u[ax * N + i];
  orc_rollout(prm, state, ctrl, traj);       /* literal recursion, independent of the condensed maps */
  *obj = orc_objective(prm, traj, ctrl, ref); /* literal objective */
}

static int miqp_shared(const hdsm_params* prm, const shared_t* sh, const double* state, const double* ref,
                       const orc_corridor* cor, double* traj, double* ctrl, uint8_t* used,
                       orc_result* res, double cut0, int search) {
  const int N = prm->n_hor;
  inst_t in;
  build_inst(prm, sh, state, ref, &in);
  bnb_t* b = (bnb_t*)calloc(1, sizeof(bnb_t));
  b->prm = prm, b->in = &in, b->cor = cor;
  b->tol = prm->solver_tol > 0 ? prm->solver_tol : 1e-9;
  b->ftol_fixed = prm->feas_tol_fixed > 0 ? prm->feas_tol_fixed : 1e-6;
  b->pinned = pinned_steps(sh);
  b->max_nodes = prm->max_nodes > 0 ? prm->max_nodes : 100000;
  b->iter_budget = prm->max_qp_iters > 0 ? prm->max_qp_iters : 10000000;
  b->inc_f = INFINITY, b->second = INFINITY, b->cut0 = cut0;
  memset(res, 0, sizeof *res);
  res->status = HDSM_NO_SOLUTION;
  res->runner_up = HDSM_INF;
  int ok = 1;
  for (int i = 0; i < N; i++) ok &= cor->m[i] > 0;
  if (ok && build_base(b)) {
    double u[ON], f;
    if (search == 1)
      for (int i = 0; i < N; i++) b->assign[i] = -1;
    if (solve_node(b, u, &f) == GI_OK) {
      if (search == 1) bnb_lazy(b, u, f);
      else bnb_node(b, 0, u, f);
    }
  }
  if (b->have_inc) {
    res->status = b->limit_hit ? HDSM_LIMIT : HDSM_OPTIMAL;
    finish(prm, state, ref, b->inc_u, traj, ctrl, &res->obj);
    memset(used, 0, prm->poly_hor);
    for (int i = 0; i < N; i++) { /* AC:979-985 */
      res->assign[i] = b->inc_assign[i];
      if (b->inc_assign[i] < prm->poly_hor) used[b->inc_assign[i]] = 1;
    }
    if (b->second < INFINITY) res->runner_up = b->second;
  }
  res->nodes = b->nodes, res->qp_solves = b->qp_solves, res->qp_iters = b->qp_iters;
  free(b->cons);
  free(b);
  return 0;
}

int orc_miqp(const hdsm_params* prm, const double state[9], const double* ref, const orc_corridor* cor,
             double* traj, double* ctrl, uint8_t* used, orc_result* res) {
  shared_t* sh = (shared_t*)malloc(sizeof(shared_t));
  if (build_shared(prm, sh)) {
    free(sh);
    return -1;
  }
  int rc = miqp_shared(prm, sh, state, ref, cor, traj, ctrl, used, res, INFINITY, 0);
  free(sh);
  return rc;
}

static int qp_fixed_shared(const hdsm_params* prm, const shared_t* sh, const double* state,
                           const double* ref, const orc_corridor* cor, const int32_t* assign, double* u,
                           double* f, int* iters) {
  const int N = prm->n_hor;
  inst_t in;
  build_inst(prm, sh, state, ref, &in);
  bnb_t* b = (bnb_t*)calloc(1, sizeof(bnb_t));
  b->prm = prm, b->in = &in, b->cor = cor;
  b->tol = prm->solver_tol > 0 ? prm->solver_tol : 1e-9;
  b->ftol_fixed = prm->feas_tol_fixed > 0 ? prm->feas_tol_fixed : 1e-6;
  b->pinned = pinned_steps(sh);
  b->iter_budget = 10000000;
  int ok = build_base(b);
  for (int i = 0; i < N && ok; i++)
    if (assign[i] >= 0) ok = push_poly(b, i, assign[i]);
  int rc = GI_INFEASIBLE;
  if (ok) {
    rc = gi_solve(&b->gi, &in, b->cons, b->ncons, b->tol, INFINITY, b->iter_budget);
    if (rc == GI_OK) {
      memcpy(u, b->gi.x, sizeof(double) * 3 * N);
      *f = b->gi.f;
    }
    *iters = b->gi.iters;
  }
  free(b->cons);
  free(b);
  return rc;
}

int orc_qp_fixed(const hdsm_params* prm, const double state[9], const double* ref,
                 const orc_corridor* cor, const int32_t* assign, double* traj, double* ctrl,
                 orc_result* res) {
  shared_t* sh = (shared_t*)malloc(sizeof(shared_t));
  if (build_shared(prm, sh)) {
    free(sh);
    return -1;
  }
  double u[ON], f = 0;
  int iters = 0;
  memset(res, 0, sizeof *res);
  res->runner_up = HDSM_INF;
  int rc = qp_fixed_shared(prm, sh, state, ref, cor, assign, u, &f, &iters);
  res->status = rc == GI_OK ? HDSM_OPTIMAL : HDSM_NO_SOLUTION;
  res->qp_solves = 1, res->qp_iters = iters, res->nodes = 1;
  if (rc == GI_OK) {
    finish(prm, state, ref, u, traj, ctrl, &res->obj);
    for (int i = 0; i < prm->n_hor; i++) res->assign[i] = assign[i];
  }
  free(sh);
  return 0;
}

int orc_miqp_enum(const hdsm_params* prm, const double state[9], const double* ref,
                  const orc_corridor* cor, double* traj, double* ctrl, uint8_t* used, orc_result* res) {
  const int N = prm->n_hor;
  shared_t* sh = (shared_t*)malloc(sizeof(shared_t));
  if (build_shared(prm, sh)) {
    free(sh);
    return -1;
  }
  int32_t assign[MAXH] = {0}, best[MAXH];
  double best_f = INFINITY, second = INFINITY, best_u[ON];
  memset(res, 0, sizeof *res);
  res->status = HDSM_NO_SOLUTION;
  res->runner_up = HDSM_INF;
  int done = 0;
  for (int i = 0; i < N; i++)
    if (cor->m[i] <= 0) done = 1;
  while (!done) {
    double u[ON], f;
    int iters = 0;
    int rc = qp_fixed_shared(prm, sh, state, ref, cor, assign, u, &f, &iters);
    res->qp_solves++, res->qp_iters += iters, res->nodes++;
    if (rc == GI_OK) {
      if (f < best_f) {
        if (best_f < second) second = best_f;
        best_f = f;
        memcpy(best_u, u, sizeof(double) * 3 * N);
        memcpy(best, assign, sizeof(int32_t) * N);
      } else if (f < second) {
        second = f;
      }
    }
    int i = N - 1; /* odometer */
    while (i >= 0 && ++assign[i] == cor->m[i]) assign[i--] = 0;
    if (i < 0) done = 1;
  }
  if (best_f < INFINITY) {
    res->status = HDSM_OPTIMAL;
    finish(prm, state, ref, best_u, traj, ctrl, &res->obj);
    memset(used, 0, prm->poly_hor);
    for (int i = 0; i < N; i++) {
      res->assign[i] = best[i];
      used[best[i]] = 1;
    }
    if (second < INFINITY) res->runner_up = second;
  }
  free(sh);
  return 0;
}

/* ================================================================================== batch drivers */

typedef struct {
  const hdsm_params* prm;
  const shared_t* sh;
  int level, n_inst, n_rob, r_max;
  const int32_t *agent_id, *n_poly, *n_rows;
  const double *state, *ref, *A, *b, *plans;
  const uint8_t* has_plan;
  double *traj, *ctrl, *obj;
  uint8_t* used;
  int32_t *status, *nodes, *qp_iters;
  const double* obj_hint; /* orc_replan_ex: per-instance upper bound on the optimum (NaN / >= HDSM_INF = none) */
  int search;             /* orc_replan_ex: 0 steps in order (the default), 1 most infeasible uncontained step (bnb_lazy) */
  int next;
  pthread_mutex_t mtx;
} batch_t;

static void run_instance(batch_t* B, int k) {
  const hdsm_params* prm = B->prm;
  const int N = prm->n_hor, P = prm->poly_hor;
  orc_corridor cor;
  memset(&cor, 0, sizeof cor);
  double* planes = NULL;
  double* common = NULL;
  uint8_t* valid = NULL;
  if (B->level == 2) {
    const int RS = prm->max_rows_static;
    int np = B->n_poly[k] < P ? B->n_poly[k] : P; /* AC:913 */
    planes = (double*)malloc(sizeof(double) * 4 * N * B->n_rob);
    common = (double*)malloc(sizeof(double) * 4 * N * B->n_rob);
    valid = (uint8_t*)malloc((size_t)N * B->n_rob);
    orc_tasc_planes(prm, B->n_rob, B->agent_id[k], B->state + 9 * k, B->plans, B->has_plan, planes, valid);
    for (int i = 0; i < N; i++) {
      cor.m[i] = np;
      for (int j = 0; j < np; j++) {
        cor.nrows[i][j] = B->n_rows[(size_t)k * P + j];
        cor.A[i][j] = B->A + (((size_t)k * P + j) * RS) * 3;
        cor.b[i][j] = B->b + ((size_t)k * P + j) * RS;
      }
      double* dst = common + (size_t)4 * i * B->n_rob;
      int cnt = 0;
      for (int j = 0; j < B->n_rob; j++)
        if (valid[(size_t)i * B->n_rob + j]) memcpy(dst + 4 * cnt++, planes + ((size_t)i * B->n_rob + j) * 4, 32);
      cor.ncommon[i] = cnt;
      cor.common[i] = dst;
    }
  } else {
    for (int i = 0; i < N; i++) {
      int np = B->n_poly[(size_t)k * N + i];
      cor.m[i] = np < P ? np : P;
      for (int j = 0; j < cor.m[i]; j++) {
        size_t pj = ((size_t)k * N + i) * P + j;
        cor.nrows[i][j] = B->n_rows[pj];
        cor.A[i][j] = B->A + pj * B->r_max * 3;
        cor.b[i][j] = B->b + pj * B->r_max;
      }
    }
  }
  orc_result res;
  double traj[(MAXH + 1) * 9], ctrl[MAXH * 3];
  uint8_t used[HDSM_MAX_POLY];
  double cut0 = INFINITY;
  if (B->obj_hint && B->obj_hint[k] == B->obj_hint[k] && B->obj_hint[k] < HDSM_INF)
    cut0 = B->obj_hint[k] + 1e-6 * fmax(1.0, fabs(B->obj_hint[k])); /* the hinted optimum itself stays below the cut */
  miqp_shared(prm, B->sh, B->state + 9 * k, B->ref + (size_t)6 * N * k, &cor, traj, ctrl, used, &res, cut0, B->search);
  B->status[k] = res.status;
  if (B->nodes) B->nodes[k] = res.nodes;
  if (B->qp_iters) B->qp_iters[k] = res.qp_iters;
  if (res.status != HDSM_NO_SOLUTION) { /* outputs untouched on failure, like include/hdsm.h says */
    memcpy(B->traj + (size_t)9 * (N + 1) * k, traj, sizeof(double) * 9 * (N + 1));
    memcpy(B->ctrl + (size_t)3 * N * k, ctrl, sizeof(double) * 3 * N);
    memcpy(B->used + (size_t)P * k, used, P);
    B->obj[k] = res.obj;
  }
  free(planes);
  free(common);
  free(valid);
}

static void* worker(void* arg) {
  batch_t* B = (batch_t*)arg;
  for (;;) {
    pthread_mutex_lock(&B->mtx);
    int k = B->next++;
    pthread_mutex_unlock(&B->mtx);
    if (k >= B->n_inst) return NULL;
    run_instance(B, k);
  }
}

static int run_batch(batch_t* B, int n_threads) {
  shared_t* sh = (shared_t*)malloc(sizeof(shared_t));
  if (build_shared(B->prm, sh)) {
    free(sh);
    return -1;
  }
  B->sh = sh;
  B->next = 0;
  pthread_mutex_init(&B->mtx, NULL);
  if (n_threads <= 1) {
    worker(B);
  } else {
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * n_threads);
    for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, worker, B);
    for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    free(th);
  }
  pthread_mutex_destroy(&B->mtx);
  free(sh);
  return 0;
}

int orc_replan(const hdsm_params* prm, int32_t n_inst, int32_t n_rob, const int32_t* agent_id,
               const double* state_curr, const double* traj_ref, const int32_t* n_poly,
               const int32_t* n_rows_static, const double* A_static, const double* b_static,
               const double* plans_all, const uint8_t* has_plan, double* traj_out, double* ctrl_out,
               uint8_t* poly_used, int32_t* status, double* obj, int32_t* nodes, int32_t* qp_iters,
               int32_t n_threads) {
  batch_t B;
  memset(&B, 0, sizeof B);
  B.prm = prm, B.level = 2, B.n_inst = n_inst, B.n_rob = n_rob;
  B.agent_id = agent_id, B.n_poly = n_poly, B.n_rows = n_rows_static;
  B.state = state_curr, B.ref = traj_ref, B.A = A_static, B.b = b_static, B.plans = plans_all;
  B.has_plan = has_plan, B.traj = traj_out, B.ctrl = ctrl_out, B.obj = obj, B.used = poly_used;
  B.status = status, B.nodes = nodes, B.qp_iters = qp_iters;
  return run_batch(&B, n_threads);
}

/* Verification mode of orc_replan. search: 0 = steps in order, 1 = bnb_lazy. obj_hint (may be NULL): obj_hint[k] is an objective value CLAIMED for instance k (e.g. by the device): the
 * search cuts off every node whose bound exceeds it by more than 1e-6 relative. The claim is never trusted: if it is
 * right the search ends on the same optimum (status OPTIMAL, outputs comparable as usual); if it is too low nothing is
 * found (NO_SOLUTION); if it is too high a better point is returned. What the hint buys is pruning from the first node
 * on, so that trees the step-ordered search cannot finish within its node budget are finished. */
int orc_replan_ex(const hdsm_params* prm, int32_t n_inst, int32_t n_rob, const int32_t* agent_id,
                      const double* state_curr, const double* traj_ref, const int32_t* n_poly,
                      const int32_t* n_rows_static, const double* A_static, const double* b_static,
                      const double* plans_all, const uint8_t* has_plan, const double* obj_hint, int32_t search,
                      double* traj_out, double* ctrl_out, uint8_t* poly_used, int32_t* status, double* obj,
                      int32_t* nodes, int32_t* qp_iters, int32_t n_threads) {
  batch_t B;
  memset(&B, 0, sizeof B);
  B.prm = prm, B.level = 2, B.n_inst = n_inst, B.n_rob = n_rob;
  B.agent_id = agent_id, B.n_poly = n_poly, B.n_rows = n_rows_static;
  B.state = state_curr, B.ref = traj_ref, B.A = A_static, B.b = b_static, B.plans = plans_all;
  B.has_plan = has_plan, B.traj = traj_out, B.ctrl = ctrl_out, B.obj = obj, B.used = poly_used;
  B.status = status, B.nodes = nodes, B.qp_iters = qp_iters, B.obj_hint = obj_hint, B.search = search;
  return run_batch(&B, n_threads);
}

int orc_solve(const hdsm_params* prm, int32_t n_inst, int32_t r_max, const double* state_curr,
              const double* traj_ref, const int32_t* n_poly, const int32_t* n_rows, const double* A,
              const double* b, double* traj_out, double* ctrl_out, uint8_t* poly_used, int32_t* status,
              double* obj, int32_t n_threads) {
  batch_t B;
  memset(&B, 0, sizeof B);
  B.prm = prm, B.level = 1, B.n_inst = n_inst, B.r_max = r_max;
  B.n_poly = n_poly, B.n_rows = n_rows, B.state = state_curr, B.ref = traj_ref, B.A = A, B.b = b;
  B.traj = traj_out, B.ctrl = ctrl_out, B.obj = obj, B.used = poly_used, B.status = status;
  return run_batch(&B, n_threads);
}

/* ===================================================================== f1: reference trajectory (restatement) */

/* GetVelocityLimit, AC:1805-1817 */
static double velocity_limit(const hdsm_ref_config* c, double occ_val, double dist_start) {
  if (occ_val < 0) occ_val = 0;
  if (occ_val > 100) occ_val = 100;
  double alpha = (1 - pow(occ_val / 100, c->sens_pot) * (1 / exp(c->sens_dist * dist_start)));
  return c->path_vel_min + (c->path_vel_max - c->path_vel_min) * alpha;
}

int orc_reference(const hdsm_params* prm, const hdsm_ref_config* cfg, int32_t n_inst, int32_t n_rob,
                  const int32_t* agent_id, const double* path, const int32_t* n_path, int32_t pmax,
                  const double* vel_cap, const double* plans_all, const uint8_t* has_plan, double* ref_full,
                  double* ref, double* path_vel) {
  const int N = prm->n_hor;
  for (int k = 0; k < n_inst; k++) {
    const int self = agent_id[k];
    const double* pth = path + (size_t)k * pmax * 3;
    const int np = n_path[k];
    double* out = ref_full + (size_t)k * (N + 1) * 6;
    /* ComputePathVelocity: start from the cap (voxel part, host) and apply the neighbour term AC:1769-1801 */
    double pv = vel_cap ? vel_cap[k] : cfg->path_vel_max;
    if (pv > cfg->path_vel_max) pv = cfg->path_vel_max;
    const int own = self >= 0 && self < n_rob && has_plan[self];
    if (own && np >= 2) { /* SamplePath only calls ComputePathVelocity for paths with >= 2 points (AC:1595-1612) */
      for (int i = 0; i <= N; i++) { /* traj_curr_.size() = N + 1 */
        const double* s0 = plans_all + ((size_t)self * (N + 1) + i) * 9;
        double occ = 100 * pow(cfg->sens_other_agents, i);
        for (int j = 0; j < n_rob; j++) {
          if (j == self || !has_plan[j]) continue;
          const double* so = plans_all + ((size_t)j * (N + 1) + i) * 9;
          double d = sqrt((s0[0] - so[0]) * (s0[0] - so[0]) + (s0[1] - so[1]) * (s0[1] - so[1]) +
                          (s0[2] - so[2]) * (s0[2] - so[2]));
          double v = velocity_limit(cfg, occ, d);
          if (v < pv) pv = v;
        }
      }
    }
    if (np < 2) pv = 0; /* path_vel_ is not updated in that branch; report 0 */
    path_vel[k] = pv;
    /* SamplePath, AC:1591-1663 */
    double pts[HDSM_MAX_HOR + 1][3];
    int cnt = 0;
    if (np < 2) {
      for (int i = 0; i < N; i++) memcpy(pts[cnt++], pth, 24); /* N copies of the single point */
    } else {
      const double samp_dist = pv * prm->dt;
      int path_idx = 1, ref_idx = 0;
      double cur[3] = {pth[0], pth[1], pth[2]};
      memcpy(pts[cnt++], pth, 24);
      double limit = samp_dist;
      while (ref_idx < N) {
        const double* nx = pth + 3 * path_idx;
        double df[3] = {nx[0] - cur[0], nx[1] - cur[1], nx[2] - cur[2]};
        double dist_next = sqrt(df[0] * df[0] + df[1] * df[1] + df[2] * df[2]);
        if (dist_next > limit) {
          for (int c = 0; c < 3; c++) cur[c] = cur[c] + limit * df[c] / dist_next;
          memcpy(pts[cnt++], cur, 24);
          ref_idx++;
          limit = fmax(0.0, samp_dist - cfg->path_vel_dec * prm->dt);
        } else {
          memcpy(cur, nx, 24);
          path_idx++;
          if (path_idx == np) {
            for (int i = ref_idx; i < N; i++) memcpy(pts[cnt++], pth + 3 * (np - 1), 24);
            break;
          }
          limit -= dist_next;
        }
      }
    }
    /* velocity reference, AC:1527-1547 (points backwards along the path; the last row copies the previous one) */
    double v[3] = {0, 0, 0};
    for (int i = 0; i < cnt; i++) {
      if (i + 1 < cnt) {
        double dd[3] = {pts[i][0] - pts[i + 1][0], pts[i][1] - pts[i + 1][1], pts[i][2] - pts[i + 1][2]};
        double dist = sqrt(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]);
        for (int c = 0; c < 3; c++) v[c] = dist > 1e-2 ? pv * dd[c] / dist : 0.0;
      }
      for (int c = 0; c < 3; c++) out[6 * i + c] = pts[i][c], out[6 * i + 3 + c] = (cnt > 1) ? v[c] : 0.0;
    }
    for (int i = cnt; i <= N; i++) /* single-point case yields N rows: pad the (N+1)-th with the last one */
      for (int c = 0; c < 6; c++) out[6 * i + c] = out[6 * (cnt - 1) + c];
    if (ref) memcpy(ref + (size_t)k * N * 6, out, sizeof(double) * N * 6);
  }
  return 0;
}

/* ---- next row f4: map pre-processing (mapping_util/src/map_builder.cpp:209-216, "MB"; voxel_grid_util/src/
 * voxel_grid.cpp, "VG"). LITERAL restatement: the same scatter loops in the same order. -1 unknown, 0 free,
 * 100 occupied, 1..99 potential (voxel_grid.hpp:12-14). grid [nz][ny][nx], x fastest (VG:68-76).              */
typedef struct { int dx, dy, dz; int8_t val; } orc_mask_cell;

/* VG:192-226 CreateMask */
static int orc_create_mask(double mask_dist, double power, double res, orc_mask_cell* mask, int cap) {
  const double h_max = 100.0;
  const int rn = (int)ceil(mask_dist / res);
  int cnt = 0;
  if (mask_dist > 0)
    for (int n0 = -rn; n0 <= rn; ++n0)
      for (int n1 = -rn; n1 <= rn; ++n1)
        for (int n2 = -rn; n2 <= rn; ++n2) {
          double dist = hypot(hypot((double)n0, (double)n1), (double)n2);
          dist = fabs(dist - 1);
          if (dist * res >= mask_dist) continue;
          const double h = h_max * pow(1 - hypot(hypot((double)n0, (double)n1), (double)n2) / (rn + 1), power);
          if (h > 1e-3) {
            if (cnt < cap) mask[cnt].dx = n0, mask[cnt].dy = n1, mask[cnt].dz = n2, mask[cnt].val = (int8_t)h;
            ++cnt;
          }
        }
  return cnt;
}

int orc_map_preprocess(const hdsm_map_config* cfg, int32_t n_grids, const int32_t dim[3], const int8_t* in, int8_t* out) {
  const int nx = dim[0], ny = dim[1], nz = dim[2];
  const size_t vox = (size_t)nx * ny * nz;
  const double res = cfg->voxel_size;
  const int cap = 40000;
  orc_mask_cell* m1 = (orc_mask_cell*)malloc(sizeof(orc_mask_cell) * cap);
  orc_mask_cell* m2 = (orc_mask_cell*)malloc(sizeof(orc_mask_cell) * cap);
  int32_t* occ = (int32_t*)malloc(sizeof(int32_t) * vox);
  const int c1 = orc_create_mask(cfg->inflation_dist, 1.0, res, m1, cap);                          /* VG:255-256 */
  const int c2 = orc_create_mask(cfg->potential_dist, (double)cfg->potential_pow, res, m2, cap);   /* VG:281-282 */
  if (!m1 || !m2 || !occ || c1 > cap || c2 > cap) return -1;
#define IDX(x, y, z) ((size_t)(x) + (size_t)(y) * nx + (size_t)(z) * nx * ny)
#define INSIDE(x, y, z) ((x) >= 0 && (y) >= 0 && (z) >= 0 && (x) < nx && (y) < ny && (z) < nz)
  for (int g = 0; g < n_grids; ++g) {
    const int8_t* vg = in + (size_t)g * vox;
    int8_t* fin = out + (size_t)g * vox;
    /* MB:331-362 SetUncertainToUnknown: reads vg, writes a copy */
    memcpy(fin, vg, vox);
    const int cube = (int)ceil(cfg->inflation_dist / res);
    for (int i = cube; i < nx - cube; ++i)
      for (int j = cube; j < ny - cube; ++j)
        for (int k = cube; k < nz - cube; ++k)
          if (vg[IDX(i, j, k)] == -1)
            for (int a = -cube; a <= cube; ++a)
              for (int b = -cube; b <= cube; ++b)
                for (int c = -cube; c <= cube; ++c)
                  if (INSIDE(i + a, j + b, k + c) && vg[IDX(i + a, j + b, k + c)] != 100)  /* !IsOccupied */
                    fin[IDX(i + a, j + b, k + c)] = -1;
    /* VG:252-278 InflateObstacles: the occupied voxels are collected first, then the mask is stamped around each */
    int n_occ = 0;
    for (int x = 0; x < nx; ++x)
      for (int y = 0; y < ny; ++y)
        for (int z = 0; z < nz; ++z)
          if (fin[IDX(x, y, z)] == 100) occ[n_occ++] = (int32_t)IDX(x, y, z);
    for (int q = 0; q < n_occ; ++q) {
      const int z = occ[q] / (nx * ny), y = (occ[q] - z * nx * ny) / nx, x = occ[q] - z * nx * ny - y * nx;
      for (int t = 0; t < c1; ++t) {
        const int X = x + m1[t].dx, Y = y + m1[t].dy, Z = z + m1[t].dz;
        if (INSIDE(X, Y, Z)) fin[IDX(X, Y, Z)] = 100;
      }
    }
    /* VG:280-297 CreatePotentialField: in scan order, on the grid as it evolves */
    for (int x = 0; x < nx; ++x)
      for (int y = 0; y < ny; ++y)
        for (int z = 0; z < nz; ++z)
          if (fin[IDX(x, y, z)] == 100)
            for (int t = 0; t < c2; ++t) {
              const int X = x + m2[t].dx, Y = y + m2[t].dy, Z = z + m2[t].dz;
              if (INSIDE(X, Y, Z) && fin[IDX(X, Y, Z)] != -1) {
                const int8_t cur = fin[IDX(X, Y, Z)];
                fin[IDX(X, Y, Z)] = cur > m2[t].val ? cur : m2[t].val;
              }
            }
  }
#undef IDX
#undef INSIDE
  free(m1), free(m2), free(occ);
  return 0;
}

/* =====================================================================================================================
 * Row f2: Agent::GenerateSafeCorridor, agent_class.cpp:1236-1447 — see hdsm_oracle.h. Written from the reference text.
 * ===================================================================================================================== */
typedef struct {
  int rows;
  const double* A; /* [rows][3] */
  const double* b;
  double seed[3];
} sc_poly;

/* LinearConstraint::inside, polyhedron.h:130-137: d = A pt - b; any d(i) > 0 -> outside */
static int sc_inside(const sc_poly* p, const double pt[3]) {
  for (int i = 0; i < p->rows; i++) {
    double d = (p->A[3 * i] * pt[0] + p->A[3 * i + 1] * pt[1] + p->A[3 * i + 2] * pt[2]) - p->b[i];
    if (d > 0) return 0;
  }
  return 1;
}

/* VoxelGrid::IsOccupied(Vector3i): GetVoxelInt == ENV_BUILDER_OCC, and GetVoxelInt outside the grid is -1 (voxel_grid.cpp:110-117, 150-155) */
static int sc_is_occupied(const int8_t* data, const int32_t dim[3], int x, int y, int z) {
  if (!(x < dim[0] && y < dim[1] && z < dim[2] && x >= 0 && y >= 0 && z >= 0)) return 0;
  return data[x + y * dim[0] + z * dim[0] * dim[1]] == 100;
}

int orc_safe_corridor(int32_t poly_hor, int32_t n_it_decomp, int32_t use_cvx_new_cfg, int32_t rows_max, int32_t n_prev,
                      const int32_t* prev_rows, const double* prev_A, const double* prev_b, const double* prev_seed,
                      const uint8_t* poly_used, int32_t n_traj, const double* traj_pts, int32_t n_path, const double* path,
                      const int8_t* grid, const int32_t dim[3], const double origin[3], double voxel_size, orc_decomp_fn decomp,
                      void* ctx, int32_t* n_out, int32_t* out_rows, double* out_A, double* out_b, double* out_seed) {
  /* poly_const_vec_new / poly_seeds_new: entries point into prev_* (kept) or out_* (new); written out at the end */
  sc_poly* fresh = (sc_poly*)calloc((size_t)(n_prev + poly_hor + 1), sizeof(sc_poly));
  int n_new = 0;
  sc_poly last;
  memset(&last, 0, sizeof last);

  /* AC:1253-1267: if the trajectory fits inside the last polyhedron, keep that polyhedron */
  if (n_prev > 0) {
    int i = n_prev - 1;
    last.rows = prev_rows[i], last.A = prev_A + (size_t)i * rows_max * 3, last.b = prev_b + (size_t)i * rows_max;
    memcpy(last.seed, prev_seed + 3 * i, sizeof last.seed);
    int used = 1;
    for (int j = 0; j < n_traj; j++) {
      if (!sc_inside(&last, traj_pts + 3 * j)) {
        used = 0;
        break;
      }
    }
    if (used) fresh[n_new++] = last;
  }
  /* AC:1273-1282: else keep the polyhedra that were used in the previous optimisation */
  if (n_prev > 0 && n_new == 0) {
    for (int i = 0; i < n_prev; i++) {
      if (poly_used[i]) {
        sc_poly p;
        p.rows = prev_rows[i], p.A = prev_A + (size_t)i * rows_max * 3, p.b = prev_b + (size_t)i * rows_max;
        memcpy(p.seed, prev_seed + 3 * i, sizeof p.seed);
        fresh[n_new++] = p;
      }
    }
  }

  /* AC:1292-1296: the voxel grid copy, unknown -> occupied (VoxelGrid::OccupyUnknown, voxel_grid.cpp:234-240) */
  const size_t nvox = (size_t)dim[0] * dim[1] * dim[2];
  int8_t* vg = (int8_t*)malloc(nvox);
  memcpy(vg, grid, nvox);
  for (size_t i = 0; i < nvox; i++)
    if (vg[i] == -1) vg[i] = 100;
  int8_t* grid_data = (int8_t*)malloc(nvox);

  /* storage of the NEW polyhedra (they are referenced by `fresh` while the walk goes on) */
  double* new_A = (double*)calloc((size_t)(poly_hor + 1) * rows_max * 3, sizeof(double));
  double* new_b = (double*)calloc((size_t)(poly_hor + 1) * rows_max, sizeof(double));
  double* rows = (double*)calloc((size_t)rows_max * 4, sizeof(double));
  int n_made = 0, rc = 0;

  /* AC:1298-1313 */
  int n_poly = n_new;
  int path_idx = 1;
  double curr_pt[3] = {path[0], path[1], path[2]};
  while (n_poly < poly_hor) {
    /* AC:1314-1335 */
    double next_pt[3] = {path[3 * path_idx], path[3 * path_idx + 1], path[3 * path_idx + 2]};
    double diff[3] = {next_pt[0] - curr_pt[0], next_pt[1] - curr_pt[1], next_pt[2] - curr_pt[2]};
    double dist_next = sqrt(diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2]); /* Eigen norm(): sqrt of the sum in order */
    double samp_dist = voxel_size / 10;
    if (dist_next > samp_dist) {
      /* curr_pt + samp_dist * diff / dist_next: (scalar * vector) / scalar, component-wise */
      for (int k = 0; k < 3; k++) curr_pt[k] = curr_pt[k] + (samp_dist * diff[k]) / dist_next;
    } else {
      for (int k = 0; k < 3; k++) curr_pt[k] = next_pt[k];
      path_idx = path_idx + 1;
      if (path_idx == n_path) break; /* we reached the final point */
    }
    /* AC:1337-1350: inside at least one polyhedron -> next sample */
    int inside_at_least_one_poly = 0;
    for (int i = 0; i < n_new; i++) {
      if (sc_inside(&fresh[i], curr_pt)) {
        inside_at_least_one_poly = 1;
        break;
      }
    }
    if (inside_at_least_one_poly) continue;
    /* AC:1352-1359: the previous sample is the seed */
    double seed_pt[3] = {curr_pt[0], curr_pt[1], curr_pt[2]};
    if (dist_next > 0) {
      double back = samp_dist < dist_next ? samp_dist : dist_next; /* std::min(samp_dist, dist_next) */
      for (int k = 0; k < 3; k++) seed_pt[k] = curr_pt[k] - (back * diff[k]) / dist_next;
    }
    int32_t seed[3];
    for (int k = 0; k < 3; k++) seed[k] = (int)((seed_pt[k] - origin[k]) / voxel_size);
    /* AC:1361-1379: a seed that one of the kept / new polyhedra already has */
    double seed_world[3];
    for (int k = 0; k < 3; k++) seed_world[k] = seed[k] * voxel_size + voxel_size / 2 + origin[k];
    int is_previous_seed = 0;
    for (int i = 0; i < n_new; i++) {
      if (seed_world[0] == fresh[i].seed[0] && seed_world[1] == fresh[i].seed[1] && seed_world[2] == fresh[i].seed[2]) {
        is_previous_seed = 1;
        break;
      }
    }
    if (is_previous_seed) continue;
    /* AC:1381-1395: the seed constrained from a direction -> the shape-aware variant */
    int use_cvx_new = use_cvx_new_cfg;
    if ((sc_is_occupied(vg, dim, seed[0] - 1, seed[1], seed[2]) && sc_is_occupied(vg, dim, seed[0] + 1, seed[1], seed[2])) ||
        (sc_is_occupied(vg, dim, seed[0], seed[1] - 1, seed[2]) && sc_is_occupied(vg, dim, seed[0], seed[1] + 1, seed[2])) ||
        (sc_is_occupied(vg, dim, seed[0], seed[1], seed[2] - 1) && sc_is_occupied(vg, dim, seed[0], seed[1], seed[2] + 1))) {
      use_cvx_new = 1;
    }
    /* AC:1403-1420: the decomposition works on a copy of the grid data */
    memcpy(grid_data, vg, nvox);
    int32_t n_rows_new = 0;
    rc = decomp(ctx, seed, grid_data, dim, n_it_decomp, voxel_size, -(n_poly + 1), origin, use_cvx_new, rows, rows_max, &n_rows_new);
    if (rc != 0) break;
    n_poly = n_poly + 1;
    /* AC:1425-1435: rows (normal, point . normal) -> LinearConstraint3D */
    double* A = new_A + (size_t)n_made * rows_max * 3;
    double* bb = new_b + (size_t)n_made * rows_max;
    for (int i = 0; i < n_rows_new; i++) {
      A[3 * i] = rows[4 * i], A[3 * i + 1] = rows[4 * i + 1], A[3 * i + 2] = rows[4 * i + 2];
      bb[i] = rows[4 * i + 3];
    }
    sc_poly p;
    p.rows = n_rows_new, p.A = A, p.b = bb;
    memcpy(p.seed, seed_world, sizeof p.seed);
    fresh[n_new++] = p;
    n_made++;
  }
  /* AC:1438-1441: save polyhedra and seeds */
  *n_out = n_new;
  for (int i = 0; i < n_new; i++) {
    out_rows[i] = fresh[i].rows;
    memcpy(out_A + (size_t)i * rows_max * 3, fresh[i].A, sizeof(double) * 3 * (size_t)fresh[i].rows);
    memcpy(out_b + (size_t)i * rows_max, fresh[i].b, sizeof(double) * (size_t)fresh[i].rows);
    memcpy(out_seed + 3 * i, fresh[i].seed, sizeof(double) * 3);
  }
  free(fresh), free(vg), free(grid_data), free(new_A), free(new_b), free(rows);
  return rc;
}
