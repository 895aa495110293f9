/* hdsm_cpu_port.c — the PRODUCT's algorithm as plain C on one host core per instance: bench.py's `cpu_baseline_warm` leg.
 *
 * BENCH INFRASTRUCTURE, like the oracle next to it: nothing under multi_agent_pkgs_amd/ links or loads this file, libhdsm.so has no
 * CPU path. It answers one question of the bench line — what does a host core do with the SAME algorithm the kernel runs — because
 * the oracle (hdsm_oracle.c: every one of the N (n_rob - 1) planes through libm, a cold dense active set that scans all of them
 * in every iteration) is a deliberately naive restatement of the reference and flatters the GPU by two orders of magnitude.
 *
 * What is taken over from the kernel (csrc/hdsm_core.h, hdsm_wave_gib.h):
 *   - planes in closed form (s = r / sqrt(1 + ((r/h)^2 - 1) n_z^2)), never materialised: a sweep over the published plans stages the
 *     rows whose slack at the current iterate is below a radius; bounding spheres of the plans skip far neighbours, a rigorous
 *     cull (slack >= |d| / 2 - s_max - |n_f| delta_max) skips far pairs; after convergence a verification sweep stages the violated
 *     rows and the dual method continues;
 *   - Goldfarb-Idnani dual active set, the row that enters next picked in the metric of the problem (violation / sqrt(a^T Z a));
 *   - warm start: the optimal working set of the instance's previous replan, moved one step towards the present, is put into the
 *     factorisation without taking steps; minimiser and multipliers on it in closed form (t = R^-T v, lambda = R^-1 t); entries
 *     with negative multipliers leave; the regular loop continues;
 *   - rows on input-independent positions are constants judged with feas_tol_fixed.
 * What is NOT taken over: the branch and bound. An instance whose relaxation leaves a segment in no polyhedron goes to the
 * oracle's search (counted in `fallbacks`); on the bench's circle rounds there is none.
 * The factorisation is the oracle's textbook one (dense J, triangular R, Givens updates): this is a scalar port, not a tuned CPU
 * solver — `cores` in the bench line says how many of them ran. */
#include <time.h>

#include "hdsm_oracle.c"

#define CP_MAXC 2048 /* constraints of one instance: 6 + boxes + staged plane rows */

typedef struct {
  gi_t g;
  int m;
  con_t cons[CP_MAXC];
  double w[CP_MAXC];    /* pick-rule weight 1 / sqrt(a^T Z a) */
  int32_t pid[CP_MAXC]; /* portable id of the row (0 = none): the next replan's guess is made of these */
} cp_state;

/* portable ids: kind in bits 28.., payload: input box (var << 1 | lower), state box (step << 5 | comp << 3 | ax << 1 | lower),
 * neighbour plane (neighbour << 6 | step << 1 | end point) */
enum { CP_U = 1, CP_S = 2, CP_C = 3 };
static int32_t cp_id(int kind, int payload) { return (kind << 28) | payload; }

typedef struct {
  double Z[ON * ON]; /* H^-1 projected on the null space of the terminal equalities */
  int pinned;
} cp_shared;

static void cp_build_shared(const hdsm_params* prm, const shared_t* sh, cp_shared* cs) {
  const int N = sh->N, n = sh->n;
  double E[6][ON], HE[6][ON], M[6][6], Mi[6][6];
  memset(E, 0, sizeof E);
  for (int ax = 0; ax < 3; ax++)
    for (int comp = 1; comp <= 2; comp++)
      for (int k = 0; k < N; k++) E[2 * ax + comp - 1][ax * N + k] = sh->Gam[ax][N][k][comp];
  for (int e = 0; e < 6; e++)
    for (int i = 0; i < n; i++) {
      double s = 0;
      for (int j = 0; j < n; j++) s += sh->Hinv[i * n + j] * E[e][j];
      HE[e][i] = s;
    }
  for (int a = 0; a < 6; a++)
    for (int b = 0; b < 6; b++) {
      double s = 0;
      for (int i = 0; i < n; i++) s += E[a][i] * HE[b][i];
      M[a][b] = s;
    }
  /* 6 x 6 inverse by Gauss-Jordan */
  double aug[6][12];
  for (int a = 0; a < 6; a++)
    for (int b = 0; b < 12; b++) aug[a][b] = b < 6 ? M[a][b] : (b - 6 == a ? 1.0 : 0.0);
  for (int c = 0; c < 6; c++) {
    int p = c;
    for (int r = c + 1; r < 6; r++)
      if (fabs(aug[r][c]) > fabs(aug[p][c])) p = r;
    for (int b = 0; b < 12; b++) {
      double t = aug[c][b];
      aug[c][b] = aug[p][b], aug[p][b] = t;
    }
    double d = aug[c][c];
    for (int b = 0; b < 12; b++) aug[c][b] /= d;
    for (int r = 0; r < 6; r++)
      if (r != c) {
        double f = aug[r][c];
        for (int b = 0; b < 12; b++) aug[r][b] -= f * aug[c][b];
      }
  }
  for (int a = 0; a < 6; a++)
    for (int b = 0; b < 6; b++) Mi[a][b] = aug[a][6 + b];
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) {
      double s = sh->Hinv[i * n + j];
      for (int a = 0; a < 6; a++)
        for (int b = 0; b < 6; b++) s -= HE[a][i] * Mi[a][b] * HE[b][j];
      cs->Z[i * n + j] = s;
    }
  cs->pinned = pinned_steps(sh);
  (void)prm;
}

static void cp_push(cp_state* s, const inst_t* in, const cp_shared* cs, const con_t* c, int32_t pid) {
  if (s->m >= CP_MAXC) return;
  const int n = in->sh->n;
  double a[ON];
  con_normal(c, in, a);
  double q = 0;
  for (int i = 0; i < n; i++) {
    if (a[i] == 0) continue;
    double t = 0;
    for (int j = 0; j < n; j++) t += cs->Z[i * n + j] * a[j];
    q += a[i] * t;
  }
  s->cons[s->m] = *c, s->w[s->m] = q > 1e-60 ? 1.0 / sqrt(q) : 1e30, s->pid[s->m] = pid;
  s->m++;
}

/* the regular dual active-set loop on the rows staged so far, continuing from the current (dual feasible) state */
static int cp_run(cp_state* s, const inst_t* in, double tol, int iter_budget) {
  gi_t* g = &s->g;
  const int n = in->sh->n;
  double a[ON], d[ON], z[ON], r[ON], st[3][MAXH + 1][3];
  for (;;) {
    int ip = -1;
    double kmax = 0, v_ip = 0;
    states_from_u(in, g->x, st);
    for (int c = 0; c < s->m; c++) {
      if (s->cons[c].kind == K_EQ) continue;
      double v = con_resid(&s->cons[c], in, g->x, st);
      if (v > tol && v * s->w[c] > kmax) kmax = v * s->w[c], ip = c, v_ip = v;
    }
    if (ip < 0) return GI_OK;
    con_normal(&s->cons[ip], in, a);
    double lam_p = 0;
    for (;;) {
      if (++g->iters > iter_budget) return GI_ITERLIM;
      const int q = g->q;
      for (int j = 0; j < n; j++) {
        double t = 0;
        for (int i = 0; i < n; i++) t -= g->J[i * n + j] * a[i];
        d[j] = t;
      }
      double zz = 0, dd = 0;
      for (int j = 0; j < n; j++) dd += d[j] * d[j];
      for (int j = q; j < n; j++) zz += d[j] * d[j];
      for (int i = 0; i < n; i++) {
        double t = 0;
        for (int j = q; j < n; j++) t += g->J[i * n + j] * d[j];
        z[i] = t;
      }
      for (int i = q - 1; i >= 0; i--) {
        double t = d[i];
        for (int j = i + 1; j < q; j++) t -= g->R[i * n + j] * r[j];
        r[i] = t / g->R[i * n + i];
      }
      const int dependent = !(zz > 1e-20 * dd) || q >= n;
      double t1 = INFINITY;
      int l = -1;
      for (int k = 0; k < q; k++) {
        if (s->cons[g->act[k]].kind == K_EQ) continue;
        if (r[k] > 0) {
          double t = g->lam[k] / r[k];
          if (t < t1) t1 = t, l = k;
        }
      }
      if (dependent && l < 0) return GI_INFEASIBLE;
      if (dependent) {
        for (int k = 0; k < q; k++) g->lam[k] -= t1 * r[k];
        lam_p += t1;
        gi_drop(g, l);
        continue;
      }
      const double t2 = v_ip / zz;
      const int full = t2 <= t1;
      const double t = full ? t2 : t1;
      for (int i = 0; i < n; i++) g->x[i] += t * z[i];
      g->f += t * zz * (0.5 * t + lam_p);
      for (int k = 0; k < q; k++) g->lam[k] -= t * r[k];
      lam_p += t;
      if (full) {
        gi_add_col(g, d);
        g->act[g->q - 1] = ip, g->lam[g->q - 1] = lam_p;
        break;
      }
      gi_drop(g, l);
      v_ip -= t * zz; /* along z the violation of the entering row falls at the rate ||d2||^2 */
    }
  }
}

/* puts row c into the factorisation without taking a step; 0 if it is (nearly) dependent on what is in */
static int cp_install(cp_state* s, const inst_t* in, int c) {
  gi_t* g = &s->g;
  const int n = in->sh->n, q = g->q;
  if (q >= n) return 0;
  double a[ON], d[ON];
  con_normal(&s->cons[c], in, a);
  double zz = 0, dd = 0;
  for (int j = 0; j < n; j++) {
    double t = 0;
    for (int i = 0; i < n; i++) t -= g->J[i * n + j] * a[i];
    d[j] = t, dd += t * t;
    if (j >= q) zz += t * t;
  }
  if (!(zz > 1e-8 * dd)) return 0;
  gi_add_col(g, d);
  g->act[g->q - 1] = c, g->lam[g->q - 1] = 0;
  g->iters++;
  return 1;
}

/* minimiser and multipliers on the working set, in closed form from the unconstrained minimiser x0 (see the header); entries with
 * negative multipliers leave until the pair is a valid starting point of the dual method */
static void cp_s_pair(cp_state* s, const inst_t* in) {
  gi_t* g = &s->g;
  const int n = in->sh->n;
  double st[3][MAXH + 1][3], v[ON], t[ON], lam[ON];
  states_from_u(in, in->x0, st);
  double fx0 = in->f0;
  for (int i = 0; i < n; i++) fx0 += 0.5 * in->g[i] * in->x0[i];
  for (;;) {
    const int q = g->q;
    for (int k = 0; k < q; k++) v[k] = con_resid(&s->cons[g->act[k]], in, in->x0, st);
    for (int k = 0; k < q; k++) { /* R^T t = v (R upper triangular) */
      double acc = v[k];
      for (int j = 0; j < k; j++) acc -= g->R[j * n + k] * t[j];
      t[k] = acc / g->R[k * n + k];
    }
    for (int k = q - 1; k >= 0; k--) { /* R lambda = t */
      double acc = t[k];
      for (int j = k + 1; j < q; j++) acc -= g->R[k * n + j] * lam[j];
      lam[k] = acc / g->R[k * n + k];
    }
    int worst = -1;
    double lw = -1e-12;
    for (int k = 0; k < q; k++)
      if (s->cons[g->act[k]].kind != K_EQ && lam[k] < lw) lw = lam[k], worst = k;
    if (worst < 0) {
      double tt = 0;
      for (int i = 0; i < n; i++) {
        double acc = in->x0[i];
        for (int k = 0; k < q; k++) acc += g->J[i * n + k] * t[k];
        g->x[i] = acc;
      }
      for (int k = 0; k < q; k++) tt += t[k] * t[k], g->lam[k] = lam[k];
      g->f = fx0 + 0.5 * tt;
      return;
    }
    gi_drop(g, worst);
    g->iters++;
  }
}

typedef struct {
  const hdsm_params* prm;
  const shared_t* sh;
  const cp_shared* cs;
  int n_inst, n_rob;
  const int32_t *agent_id, *n_poly, *n_rows;
  const double *state, *ref, *A, *b, *plans;
  const uint8_t* has_plan;
  const double* sph; /* [n_rob][4] bounding spheres of steps 1..N of the published plans (radius < 0: no plan) */
  int32_t* warm;     /* [n_inst][ON + 1] count + portable ids, in / out */
  double *traj, *ctrl, *obj;
  uint8_t* used;
  int32_t *status, *iters, *fallbacks;
  int next;
  pthread_mutex_t mtx;
} cp_batch;

/* one separating plane in closed form: row (n_f, n_f . q) from own point c and neighbour point o */
static int cp_plane(const hdsm_params* prm, const double c[3], const double o[3], double out[4]) {
  const double dx = o[0] - c[0], dy = o[1] - c[1], dz = o[2] - c[2];
  const double n2 = dx * dx + dy * dy + dz * dz;
  if (!(n2 > 0)) return 0;
  const double nrm = sqrt(n2), inv = 1.0 / nrm;
  const double hx = dx * inv, hy = dy * inv, hz = dz * inv;
  const double k2m1 = (prm->drone_radius / prm->drone_z_offset) * (prm->drone_radius / prm->drone_z_offset) - 1.0;
  const double sd = prm->drone_radius / sqrt(1.0 + k2m1 * hz * hz);
  const double back = 0.5 * fmin(2.0 * sd, nrm);
  const double qx = 0.5 * (c[0] + o[0]) - back * hx, qy = 0.5 * (c[1] + o[1]) - back * hy, qz = 0.5 * (c[2] + o[2]) - back * hz;
  const double p = prm->plane_perturb;
  out[0] = hx + p * (hy - hz) - p * hz, out[1] = hy - p * hx, out[2] = hz + p * hx + p * hx;
  out[3] = out[0] * qx + out[1] * qy + out[2] * qz;
  return 1;
}

static void cp_instance(cp_batch* B, int k) {
  const hdsm_params* prm = B->prm;
  const shared_t* sh = B->sh;
  const int N = prm->n_hor, P = prm->poly_hor, RS = prm->max_rows_static, n = sh->n, n_rob = B->n_rob;
  const int self = B->agent_id[k], pinned = B->cs->pinned;
  const double tol = prm->solver_tol > 0 ? prm->solver_tol : 1e-9;
  const double ftol = prm->feas_tol_fixed > 0 ? prm->feas_tol_fixed : 1e-6;
  const double* state = B->state + 9 * k;
  inst_t in;
  build_inst(prm, sh, state, B->ref + (size_t)6 * N * k, &in);
  cp_state* s = (cp_state*)malloc(sizeof(cp_state));
  s->m = 0;
  gi_t* g = &s->g;
  g->n = n, g->q = 0, g->iters = 0;
  memcpy(g->J, sh->J0, sizeof(double) * n * n);
  memset(g->R, 0, sizeof(double) * n * n);
  /* rows that are always there: terminal equalities, input boxes, state boxes */
  con_t c;
  memset(&c, 0, sizeof c);
  for (int ax = 0; ax < 3; ax++)
    for (int comp = 1; comp <= 2; comp++) c.kind = K_EQ, c.step = N, c.ax = ax, c.comp = comp, c.sgn = 1, c.rhs = 0, cp_push(s, &in, B->cs, &c, 0);
  for (int st_ = 0; st_ < N; st_++)
    for (int ax = 0; ax < 3; ax++) {
      c.kind = K_UBOX, c.step = st_, c.ax = ax, c.comp = 0;
      if (fabs(prm->u_ub[ax]) < ABSENT) c.sgn = 1, c.rhs = prm->u_ub[ax], cp_push(s, &in, B->cs, &c, cp_id(CP_U, (ax * N + st_) << 1));
      if (fabs(prm->u_lb[ax]) < ABSENT) c.sgn = -1, c.rhs = -prm->u_lb[ax], cp_push(s, &in, B->cs, &c, cp_id(CP_U, ((ax * N + st_) << 1) | 1));
    }
  for (int i = 1; i < N; i++)
    for (int ax = 0; ax < 3; ax++)
      for (int comp = 1; comp <= 2; comp++) {
        double ub = prm->x_ub[3 * comp + ax], lb = prm->x_lb[3 * comp + ax];
        c.kind = K_SBOX, c.step = i, c.ax = ax, c.comp = comp;
        if (fabs(ub) < ABSENT) c.sgn = 1, c.rhs = ub, cp_push(s, &in, B->cs, &c, cp_id(CP_S, (i << 5) | (comp << 3) | (ax << 1)));
        if (fabs(lb) < ABSENT) c.sgn = -1, c.rhs = -lb, cp_push(s, &in, B->cs, &c, cp_id(CP_S, (i << 5) | (comp << 3) | (ax << 1) | 1));
      }
  const int n_base = s->m;
  for (int e = 0; e < 6; e++) cp_install(s, &in, e); /* the six terminal equalities are in every working set */
  /* the own previous plan (the planes are built around it), its bounding sphere */
  const int own_has = (self >= 0 && self < n_rob) ? B->has_plan[self] : 0;
  double cprev[MAXH][3];
  for (int i = 0; i < N; i++)
    for (int ax = 0; ax < 3; ax++) cprev[i][ax] = own_has ? B->plans[((size_t)self * (N + 1) + i + 1) * 9 + ax] : state[ax];
  double mid[3], rho_self = 0;
  for (int ax = 0; ax < 3; ax++) mid[ax] = 0.5 * (cprev[0][ax] + cprev[N - 1][ax]);
  for (int i = 0; i < N; i++) {
    double d2 = 0;
    for (int ax = 0; ax < 3; ax++) d2 += (cprev[i][ax] - mid[ax]) * (cprev[i][ax] - mid[ax]);
    if (d2 > rho_self) rho_self = d2;
  }
  rho_self = sqrt(rho_self) * (1 + 1e-9);
  uint8_t* staged = (uint8_t*)calloc((size_t)n_rob * N, 1); /* bit e: row (k, i, e) is staged */
  const double rr = prm->drone_radius, hh = prm->drone_z_offset;
  const double smax = rr > hh ? rr : hh, nfmax = sqrt(1.0 + 9.0 * prm->plane_perturb * prm->plane_perturb);
  int fixed_bad = 0, status = HDSM_NO_SOLUTION, fallback = 0;
  /* one sweep over the published plans at the current iterate: stage rows with slack below thresh; returns rows staged that are violated */
  double stt[3][MAXH + 1][3];
#define CP_SWEEP(thresh, check_fixed, n_viol)                                                                                          \
  do {                                                                                                                                 \
    states_from_u(&in, g->x, stt);                                                                                                     \
    double dmax = 0;                                                                                                                   \
    for (int i = 0; i < N; i++)                                                                                                        \
      for (int e = 0; e < 2; e++) {                                                                                                    \
        double d2 = 0;                                                                                                                 \
        for (int ax = 0; ax < 3; ax++) d2 += (stt[ax][i + e][0] - cprev[i][ax]) * (stt[ax][i + e][0] - cprev[i][ax]);                  \
        if (d2 > dmax) dmax = d2;                                                                                                      \
      }                                                                                                                                \
    const double cull = 2.0 * (fmax(thresh, tol) + smax + nfmax * sqrt(dmax)) * (1 + 1e-9);                                            \
    for (int j = 0; j < n_rob; j++) {                                                                                                  \
      const double* sj = B->sph + 4 * (size_t)j;                                                                                       \
      if (j == self || sj[3] < 0) continue;                                                                                            \
      const double ux = sj[0] - mid[0], uy = sj[1] - mid[1], uz = sj[2] - mid[2], reach = cull + rho_self + sj[3];                     \
      if (!(ux * ux + uy * uy + uz * uz < reach * reach)) continue;                                                                    \
      for (int i = 0; i < N; i++) {                                                                                                    \
        const double* o = B->plans + ((size_t)j * (N + 1) + i + 1) * 9;                                                                \
        const double dx = o[0] - cprev[i][0], dy = o[1] - cprev[i][1], dz = o[2] - cprev[i][2];                                        \
        if (!(dx * dx + dy * dy + dz * dz < cull * cull)) continue;                                                                    \
        double row[4];                                                                                                                 \
        if (!cp_plane(prm, cprev[i], o, row)) continue;                                                                                \
        for (int e = 0; e < 2; e++) {                                                                                                  \
          const int mstep = i + e;                                                                                                     \
          const double v = row[0] * stt[0][mstep][0] + row[1] * stt[1][mstep][0] + row[2] * stt[2][mstep][0] - row[3];                 \
          if (mstep <= pinned) {                                                                                                       \
            if ((check_fixed) && v > ftol) fixed_bad = 1;                                                                              \
            continue;                                                                                                                  \
          }                                                                                                                            \
          if (-v < (thresh) && !(staged[(size_t)j * N + i] & (1 << e))) {                                                              \
            staged[(size_t)j * N + i] |= (uint8_t)(1 << e);                                                                            \
            con_t pc;                                                                                                                  \
            memset(&pc, 0, sizeof pc);                                                                                                 \
            pc.kind = K_PLANE, pc.step = mstep, pc.sgn = 1, pc.nrm[0] = row[0], pc.nrm[1] = row[1], pc.nrm[2] = row[2], pc.rhs = row[3]; \
            cp_push(s, &in, B->cs, &pc, cp_id(CP_C, (j << 6) | (i << 1) | e));                                                         \
            if (v > tol) (n_viol)++;                                                                                                   \
          }                                                                                                                            \
        }                                                                                                                              \
      }                                                                                                                                \
    }                                                                                                                                  \
  } while (0)

  /* ---- warm start: the previous working set, one step on (rows that fall off the horizon or onto a constant position are left out) */
  int32_t* wp = B->warm ? B->warm + (size_t)k * (ON + 1) : NULL;
  memcpy(g->x, in.x0, sizeof(double) * n);
  g->f = in.f0;
  for (int i = 0; i < n; i++) g->f += 0.5 * in.g[i] * in.x0[i];
  int warm_rows = 0;
  if (wp && wp[0] > 0) {
    for (int w = 0; w < wp[0] && w < ON; w++) {
      const int kind = (wp[1 + w] >> 28) & 7, p = wp[1 + w] & 0x0fffffff;
      int idx = -1;
      if (kind == CP_U) {
        const int var = p >> 1;
        if (var % N >= 1) {
          const int32_t want = cp_id(CP_U, ((var - 1) << 1) | (p & 1));
          for (int cc = 6; cc < n_base; cc++)
            if (s->pid[cc] == want) idx = cc;
        }
      } else if (kind == CP_S) {
        const int i = p >> 5;
        if (i - 1 >= 1) {
          const int32_t want = cp_id(CP_S, ((i - 1) << 5) | (p & 31));
          for (int cc = 6; cc < n_base; cc++)
            if (s->pid[cc] == want) idx = cc;
        }
      } else if (kind == CP_C) {
        const int e = p & 1, i = ((p >> 1) & 31) - 1, j = p >> 6;
        if (i >= 0 && i + e > pinned && j != self && j < n_rob && B->has_plan[j] && !(staged[(size_t)j * N + i] & (1 << e))) {
          double row[4];
          if (cp_plane(prm, cprev[i], B->plans + ((size_t)j * (N + 1) + i + 1) * 9, row)) {
            staged[(size_t)j * N + i] |= (uint8_t)(1 << e);
            con_t pc;
            memset(&pc, 0, sizeof pc);
            pc.kind = K_PLANE, pc.step = i + e, pc.sgn = 1, pc.nrm[0] = row[0], pc.nrm[1] = row[1], pc.nrm[2] = row[2], pc.rhs = row[3];
            idx = s->m;
            cp_push(s, &in, B->cs, &pc, cp_id(CP_C, (j << 6) | (i << 1) | e));
          }
        }
      }
      if (idx >= 0) warm_rows += cp_install(s, &in, idx);
    }
  }
  cp_s_pair(s, &in); /* (with the equalities alone: the minimiser subject to v_N = a_N = 0) */
  (void)warm_rows;
  /* ---- stage around the starting point, iterate, verify */
  int sweeps = 0, rc = GI_OK;
  {
    int nv = 0;
    CP_SWEEP(0.6, 1, nv);
    sweeps++;
  }
  if (!fixed_bad) {
    for (;;) {
      rc = cp_run(s, &in, tol, 100000);
      if (rc != GI_OK) break;
      int nv = 0;
      CP_SWEEP(-tol, 0, nv);
      sweeps++;
      if (nv == 0) break;
    }
    if (rc == GI_OK) { /* every segment in a polyhedron? (lowest index containing one, rows on constant positions gate the choice) */
      orc_corridor cor;
      memset(&cor, 0, sizeof cor);
      bnb_t bb;
      memset(&bb, 0, sizeof bb);
      const int np = B->n_poly[k] < P ? B->n_poly[k] : P;
      for (int i = 0; i < N; i++) {
        cor.m[i] = np;
        for (int j = 0; j < np; j++) {
          cor.nrows[i][j] = B->n_rows[(size_t)k * P + j];
          cor.A[i][j] = B->A + (((size_t)k * P + j) * RS) * 3;
          cor.b[i][j] = B->b + ((size_t)k * P + j) * RS;
        }
      }
      bb.prm = prm, bb.in = &in, bb.cor = &cor, bb.tol = tol, bb.ftol_fixed = ftol, bb.pinned = pinned;
      states_from_u(&in, g->x, stt);
      int all_in = np > 0;
      uint8_t used[HDSM_MAX_POLY];
      memset(used, 0, sizeof used);
      for (int i = 0; i < N && all_in; i++) {
        int found = -1;
        for (int j = 0; j < np && found < 0; j++)
          if (poly_violation(&bb, i, j, stt) <= tol) found = j;
        if (found < 0) all_in = 0;
        else used[found] = 1;
      }
      if (all_in) {
        status = HDSM_OPTIMAL;
        finish(prm, state, B->ref + (size_t)6 * N * k, g->x, B->traj + (size_t)9 * (N + 1) * k, B->ctrl + (size_t)3 * N * k, &B->obj[k]);
        memcpy(B->used + (size_t)P * k, used, P);
      } else {
        fallback = 1;
      }
    }
  }
  if (wp) { /* the next replan's guess */
    int cnt = 0;
    if (status == HDSM_OPTIMAL)
      for (int q = 0; q < g->q; q++)
        if (s->pid[g->act[q]] != 0) wp[1 + cnt++] = s->pid[g->act[q]];
    wp[0] = cnt;
  }
  if (B->iters) B->iters[k] = g->iters;
  free(staged);
  free(s);
  if (fallback) { /* a tree is needed: the oracle's search (cold, every plane) */
    batch_t O;
    memset(&O, 0, sizeof O);
    O.prm = prm, O.sh = sh, O.level = 2, O.n_inst = B->n_inst, O.n_rob = n_rob, O.agent_id = B->agent_id, O.n_poly = B->n_poly, O.n_rows = B->n_rows;
    O.state = B->state, O.ref = B->ref, O.A = B->A, O.b = B->b, O.plans = B->plans, O.has_plan = B->has_plan;
    O.traj = B->traj, O.ctrl = B->ctrl, O.obj = B->obj, O.used = B->used, O.status = B->status, O.search = 1;
    run_instance(&O, k);
    if (B->fallbacks) __sync_fetch_and_add(B->fallbacks, 1);
    return;
  }
  B->status[k] = status;
#undef CP_SWEEP
}

static void* cp_worker(void* arg) {
  cp_batch* B = (cp_batch*)arg;
  for (;;) {
    pthread_mutex_lock(&B->mtx);
    int k = B->next++;
    pthread_mutex_unlock(&B->mtx);
    if (k >= B->n_inst) return NULL;
    cp_instance(B, k);
  }
}

static void cp_spheres(const hdsm_params* prm, int n_rob, const double* plans_all, const uint8_t* has_plan, double* sph) {
  const int N = prm->n_hor;
  for (int j = 0; j < n_rob; j++) { /* what the kernel's pre-pass computes: bounding sphere of steps 1..N of every plan */
    double* o = sph + 4 * (size_t)j;
    o[0] = o[1] = o[2] = 0, o[3] = -1;
    if (!has_plan[j]) continue;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int i = 1; i <= N; i++)
      for (int ax = 0; ax < 3; ax++) {
        const double v = plans_all[((size_t)j * (N + 1) + i) * 9 + ax];
        if (v < lo[ax]) lo[ax] = v;
        if (v > hi[ax]) hi[ax] = v;
      }
    double r2 = 0;
    for (int ax = 0; ax < 3; ax++) o[ax] = 0.5 * (lo[ax] + hi[ax]);
    for (int i = 1; i <= N; i++) {
      double d2 = 0;
      for (int ax = 0; ax < 3; ax++) {
        const double u = plans_all[((size_t)j * (N + 1) + i) * 9 + ax] - o[ax];
        d2 += u * u;
      }
      if (d2 > r2) r2 = d2;
    }
    o[3] = sqrt(r2) * (1 + 1e-9);
  }
}

/* ---- a recorded flight replayed on the host cores: R rounds of the same n_inst agents, every thread owns a block of agents and
 * walks the rounds in order (so every replan is warm-started from the same agent's previous one, like on the device); the first
 * n_warm rounds only build the warm-start stores, the rest is timed (all threads meet at a barrier before the clock starts).
 * Arrays are the per-round arrays of hdsm_replan stacked along a leading round axis. */
typedef struct {
  cp_batch* rounds;
  int n_rounds, n_warm, first, count, tid, n_threads;
  pthread_barrier_t* bar;
  struct timespec* t0;
} cp_replay_arg;

static void* cp_replay_worker(void* p) {
  cp_replay_arg* a = (cp_replay_arg*)p;
  for (int r = 0; r < a->n_rounds; r++) {
    if (r == a->n_warm) {
      pthread_barrier_wait(a->bar);
      if (a->tid == 0) clock_gettime(CLOCK_MONOTONIC, a->t0);
      pthread_barrier_wait(a->bar);
    }
    for (int k = a->first; k < a->first + a->count; k++) cp_instance(&a->rounds[r], k);
  }
  return NULL;
}

int cpu_port_replay(const hdsm_params* prm, int32_t n_rounds, int32_t n_warm, int32_t n_inst, int32_t n_rob, const int32_t* agent_id,
                    const double* state_curr, const double* traj_ref, const int32_t* n_poly, const int32_t* n_rows_static,
                    const double* A_static, const double* b_static, const double* plans_all, const uint8_t* has_plan, double* traj_out,
                    double* ctrl_out, uint8_t* poly_used, int32_t* status, double* obj, int32_t* iters, int32_t* fallbacks,
                    int32_t n_threads, double* seconds_timed) {
  shared_t* sh = (shared_t*)malloc(sizeof(shared_t));
  cp_shared* cs = (cp_shared*)malloc(sizeof(cp_shared));
  if (build_shared(prm, sh) || n_threads < 1 || n_warm < 0 || n_warm >= n_rounds) {
    free(sh), free(cs);
    return -1;
  }
  cp_build_shared(prm, sh, cs);
  const size_t N = prm->n_hor, P = prm->poly_hor, RS = prm->max_rows_static, I = n_inst, G = n_rob;
  double* sph = (double*)malloc(sizeof(double) * 4 * G * n_rounds);
  int32_t* warm = (int32_t*)calloc(I * (ON + 1), sizeof(int32_t));
  cp_batch* rounds = (cp_batch*)calloc(n_rounds, sizeof(cp_batch));
  for (int r = 0; r < n_rounds; r++) {
    cp_batch* B = &rounds[r];
    cp_spheres(prm, n_rob, plans_all + (size_t)r * G * (N + 1) * 9, has_plan + (size_t)r * G, sph + 4 * G * r);
    B->prm = prm, B->sh = sh, B->cs = cs, B->n_inst = n_inst, B->n_rob = n_rob;
    B->agent_id = agent_id + (size_t)r * I, B->n_poly = n_poly + (size_t)r * I, B->n_rows = n_rows_static + (size_t)r * I * P;
    B->state = state_curr + (size_t)r * I * 9, B->ref = traj_ref + (size_t)r * I * N * 6;
    B->A = A_static + (size_t)r * I * P * RS * 3, B->b = b_static + (size_t)r * I * P * RS;
    B->plans = plans_all + (size_t)r * G * (N + 1) * 9, B->has_plan = has_plan + (size_t)r * G, B->sph = sph + 4 * G * r, B->warm = warm;
    B->traj = traj_out + (size_t)r * I * (N + 1) * 9, B->ctrl = ctrl_out + (size_t)r * I * N * 3, B->obj = obj + (size_t)r * I;
    B->used = poly_used + (size_t)r * I * P, B->status = status + (size_t)r * I, B->iters = iters ? iters + (size_t)r * I : NULL, B->fallbacks = fallbacks;
  }
  if (n_threads > n_inst) n_threads = n_inst;
  pthread_barrier_t bar;
  pthread_barrier_init(&bar, NULL, n_threads);
  struct timespec t0, t1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * n_threads);
  cp_replay_arg* args = (cp_replay_arg*)calloc(n_threads, sizeof(cp_replay_arg));
  for (int t = 0; t < n_threads; t++) {
    const int lo = (int)((size_t)n_inst * t / n_threads), hi = (int)((size_t)n_inst * (t + 1) / n_threads);
    args[t].rounds = rounds, args[t].n_rounds = n_rounds, args[t].n_warm = n_warm, args[t].first = lo, args[t].count = hi - lo;
    args[t].tid = t, args[t].n_threads = n_threads, args[t].bar = &bar, args[t].t0 = &t0;
    pthread_create(&th[t], NULL, cp_replay_worker, &args[t]);
  }
  for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (seconds_timed) *seconds_timed = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  pthread_barrier_destroy(&bar);
  free(th), free(args), free(rounds), free(warm), free(sph), free(sh), free(cs);
  return 0;
}

/* Level 2 with the layouts of include/hdsm.h. warm: [n_inst][ON + 1] int32 (zeros: cold), carried from one call to the next like the
 * handle's warm-start store; iters (may be NULL): active-set operations per instance; fallbacks (may be NULL): instances that went to
 * the oracle's branch and bound. */
int cpu_port_replan(const hdsm_params* prm, int32_t n_inst, int32_t n_rob, const int32_t* agent_id, const double* state_curr,
                    const double* traj_ref, const int32_t* n_poly, const int32_t* n_rows_static, const double* A_static,
                    const double* b_static, const double* plans_all, const uint8_t* has_plan, int32_t* warm, double* traj_out,
                    double* ctrl_out, uint8_t* poly_used, int32_t* status, double* obj, int32_t* iters, int32_t* fallbacks,
                    int32_t n_threads) {
  shared_t* sh = (shared_t*)malloc(sizeof(shared_t));
  cp_shared* cs = (cp_shared*)malloc(sizeof(cp_shared));
  if (build_shared(prm, sh)) {
    free(sh), free(cs);
    return -1;
  }
  cp_build_shared(prm, sh, cs);
  double* sph = (double*)malloc(sizeof(double) * 4 * (size_t)n_rob);
  cp_spheres(prm, n_rob, plans_all, has_plan, sph);
  cp_batch B;
  memset(&B, 0, sizeof B);
  B.prm = prm, B.sh = sh, B.cs = cs, B.n_inst = n_inst, B.n_rob = n_rob, B.agent_id = agent_id, B.n_poly = n_poly, B.n_rows = n_rows_static;
  B.state = state_curr, B.ref = traj_ref, B.A = A_static, B.b = b_static, B.plans = plans_all, B.has_plan = has_plan, B.sph = sph, B.warm = warm;
  B.traj = traj_out, B.ctrl = ctrl_out, B.obj = obj, B.used = poly_used, B.status = status, B.iters = iters, B.fallbacks = fallbacks;
  pthread_mutex_init(&B.mtx, NULL);
  if (n_threads <= 1) {
    cp_worker(&B);
  } else {
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * n_threads);
    for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, cp_worker, &B);
    for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    free(th);
  }
  pthread_mutex_destroy(&B.mtx);
  free(sph), free(sh), free(cs);
  return 0;
}
