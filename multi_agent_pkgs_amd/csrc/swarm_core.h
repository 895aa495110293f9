// swarm_core.h — the part of multi_agent_planner::Agent that sits directly around the solve, as plain data + functions that
// compile for the host mirror (swarm_host.cpp) AND for the device-resident loop (swarm_kernels.hip): one source, same
// arithmetic (no floating-point contraction), so the two loops can be compared state by state.
// AC = multi_agent_planner/src/agent_class.cpp of lis-epfl/multi_agent_pkgs.
//
//   corridor_step      GenerateSafeCorridor (AC:1236-1447): keep-last / keep-used polyhedra, walk along the path in steps of
//                      voxel / 10, seed a new polyhedron where the walk leaves the kept ones (free-space closed form, or the
//                      voxel decomposition of corridor_core.h on a window of the world grid)
//   reference_polyline the polyline SamplePath walks this round (AC:1459-1496)
//   check_increment    CheckReferenceTrajIncrement + GetPathProgress (AC:569-585, path_tools.cpp:419-479)
//   commit_one         read-back bookkeeping, the shift-by-one fallback (AC:1000-1019), state advance (AC:233-238)
#pragma once
#include <float.h>
#include <math.h>
#include <stdint.h>

#include "../../include/hdsm_swarm.h"
#include "corridor_core.h"
#include "corridor_wave.h"
#include "hdsm_types.h"

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace hdsm_sw {

using hdsm_cd::Cell;
using hdsm_cd::WindowGrid;
using hdsm_cd::Work;

constexpr int PATH_PTS = 48;  // points of a global path (router output: <= ~30)

struct V3 {
  double v[3];
  CD_HD double& operator[](int i) { return v[i]; }
  CD_HD const double& operator[](int i) const { return v[i]; }
};
CD_HD V3 sub(const V3& a, const V3& b) { return {{a[0] - b[0], a[1] - b[1], a[2] - b[2]}}; }
CD_HD V3 axpy(const V3& a, double s, const V3& b) { return {{a[0] + s * b[0], a[1] + s * b[1], a[2] + s * b[2]}}; }
// a + s * d / len in the evaluation order of the reference's Eigen expressions (`curr_pt + samp_dist * diff / dist_next`,
// AC:1320, 1353, 1635; path_tools.cpp:450): (s * d_k) / len per component. The walks accumulate hundreds of these steps and
// truncate the result to a voxel index or compare it with a strict '<', so the order of the roundings is kept.
CD_HD V3 step_along(const V3& a, double s, const V3& d, double len) {
  return {{a[0] + (s * d[0]) / len, a[1] + (s * d[1]) / len, a[2] + (s * d[2]) / len}};
}
// The walk's shortcut (corridor_step): `n_safe` samples ahead of `curr` on the way to `next` are known to lie inside the kept
// polyhedron the current sample is in, so they need neither a test nor — being equally spaced on a straight line — a step each:
// the walk moves k samp along the segment in ONE step, k = the samples that fit before the segment's last one (the regular
// loop handles the end of a segment itself). The position differs from k single steps by their accumulated rounding (~k ulp);
// the shortcut's margin (three samples and kWalkTol of slack in every row) is orders of magnitude above that.
CD_HD void walk_jump(V3& curr, const V3& next, double samp, int n_safe) {
  const V3 df = {{next[0] - curr[0], next[1] - curr[1], next[2] - curr[2]}};
  const double dn = sqrt((df[0] * df[0] + df[1] * df[1]) + df[2] * df[2]);
  const double fit = (dn - samp) / samp;  // steps after which more than one sample of the segment is still ahead
  int k = fit > 1e6 ? 1000000 : (fit > 0 ? (int)fit : 0);
  if (k > n_safe) k = n_safe;
  if (k > 0) curr = step_along(curr, k * samp, df, dn);
}
CD_HD double dot(const V3& a, const V3& b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
CD_HD double norm(const V3& a) { return sqrt(dot(a, a)); }

constexpr double kWalkTol = 1e-6;  // slack every sample skipped by the walk's shortcut keeps in every row (corridor_step)

struct Poly {  // LinearConstraint3D: rows A x <= b
  int32_t rows;
  int32_t pad;
  double A[HDSM_MAX_ROWS_STATIC][3];
  double b[HDSM_MAX_ROWS_STATIC];
  V3 seed;  // poly_seeds_ entry (voxel centre in world coordinates)
};
// LinearConstraint::inside (decomp_geometry/polyhedron.h:130-137): rejected when A x - b > 0
CD_HD bool inside(const Poly& p, const V3& x) {
  for (int r = 0; r < p.rows; ++r)
    if (((p.A[r][0] * x[0] + p.A[r][1] * x[1]) + p.A[r][2] * x[2]) - p.b[r] > 0) return false;
  return true;
}

struct AgentS {
  int32_t id, n_path;
  V3 start, goal;
  V3 path[PATH_PTS];            // path_curr_: global path start -> goal (straight unless routed / set)
  double state_curr[9];
  int32_t has_traj, n_ref;      // traj_curr_ empty before the first solve; traj_ref_curr_ rows
  double traj_curr[hdsm::MAXH + 1][9];
  double ctrl_curr[hdsm::MAXH][3];
  double traj_ref[hdsm::MAXH + 1][6];
  int32_t n_poly, increment;    // poly_const_vec_.size(), increment_traj_ref_
  Poly polys[hdsm::MAXP];
  uint8_t poly_used[hdsm::MAXP];
  int32_t external_ref, n_fail, corridor_rc, pad;
  double path_vel;
};

struct Cfg {                    // what the functions below read of hdsm_params / hdsm_swarm_config / the world
  int32_t N, P, RS, step_plan, n_it_decomp, use_cvx_new, has_world;
  int32_t fast_walk;            // device corridor walk: skip the tests of samples provably inside (HDSM_FAST_WALK, default 1)
  double voxel_size, grid_range[3], grid_z_min, thresh_dist;
  const int8_t* world;          // [wz][wy][wx] or null
  int32_t wdim[3], pad2;
  double worigin[3];
};

// IsOnSegment, AC:1864-1884: |pa| + |pb| == |ab| within 1e-6 and (p - a).(p - b) <= 0
CD_HD bool on_segment(const V3& p, const V3& a, const V3& b) {
  const double d1 = norm(sub(p, a)), d2 = norm(sub(p, b)), d12 = norm(sub(a, b));
  if (fabs(d1 + d2 - d12) < 1e-6) return dot(sub(p, a), sub(p, b)) <= 0;
  return false;
}

// the part of the global path that is still ahead of `p`, a point ON the path (AC:1480-1495): out = [p, waypoints after the
// segment that contains p ...]; returns the number of points (<= PATH_PTS + 1)
CD_HD int path_ahead(const AgentS& ag, const V3& p, V3* out) {
  int start_idx = 0;
  for (int i = 0; i + 1 < ag.n_path; ++i)
    if (on_segment(p, ag.path[i], ag.path[i + 1])) {
      start_idx = i + 1;
      break;
    }
  int n = 0;
  out[n++] = p;
  for (int i = start_idx; i < ag.n_path; ++i) out[n++] = ag.path[i];
  return n;
}

// The polyline SamplePath walks this round (AC:1459-1496): starting point from the previous reference, then the rest of
// the global path.
CD_HD int reference_polyline(const AgentS& ag, V3* out) {
  V3 starting;
  if (ag.n_ref > 0) {
    const double* r = ag.increment ? ag.traj_ref[1] : ag.traj_ref[0];
    starting = {{r[0], r[1], r[2]}};
  } else {
    starting = ag.path[0];
  }
  return path_ahead(ag, starting, out);
}

// Free-space polyhedron of GetPolyOcta3D (convex_decomp.cpp:5-376) in closed form: every face advances
// one voxel layer per visit, round robin, n_it/6 visits each; growth stays inside local voxels 1..dim-2 and above the ground
// (voxels below world z = grid_z_min are unknown -> occupied).
CD_HD void free_space_poly(const Cfg& c, const V3& grid_origin, const int seed[3], Poly* out) {
  const double vs = c.voxel_size;
  const int layers = c.n_it_decomp / 6;
  double lo[3], hi[3];
  for (int ax = 0; ax < 3; ++ax) {
    const int dim = (int)floor(c.grid_range[ax] / vs);
    int lo_lim = 1, hi_lim = dim - 2;
    if (ax == 2) {
      const int first_free = (int)ceil((c.grid_z_min - grid_origin[2]) / vs - 1e-9);
      if (first_free > lo_lim) lo_lim = first_free;
    }
    int a = seed[ax] - layers, b = seed[ax] + layers;
    if (a < lo_lim) a = lo_lim;
    if (b > hi_lim) b = hi_lim;
    lo[ax] = a * vs + grid_origin[ax];
    hi[ax] = (b + 1) * vs + grid_origin[ax];
  }
  // face order of convex_decomp.cpp:361-373: -y, +x, +y, -x, +z, -z ; rows n.x <= n.p
  const double n[6][3] = {{0, -1, 0}, {1, 0, 0}, {0, 1, 0}, {-1, 0, 0}, {0, 0, 1}, {0, 0, -1}};
  const double rhs[6] = {-lo[1], hi[0], hi[1], -lo[0], hi[2], -lo[2]};
  out->rows = 6;
  for (int r = 0; r < 6; ++r) {
    for (int k = 0; k < 3; ++k) out->A[r][k] = n[r][k];
    out->b[r] = rhs[r];
  }
}

// Polyhedron around a seed on an occupied world: the agent's local voxel grid (what env_builder hands to the planner,
// environment_builder.cpp:58-67) is a WINDOW of the world grid (corridor_core.h WindowGrid: ground below, unknown = occupied,
// outside the world = free). A seed pinched between two occupied voxels along an axis gets the shape-aware variant
// (AC:1385-1397), every other seed the original one. `bits` = WindowGrid::WORDS words of scratch.

// the agent's local grid as a window of the world, overlay centred on `seed`
CD_HD WindowGrid make_window(const Cfg& c, const V3& grid_origin, const int seed[3], uint32_t* bits) {
  const double vs = c.voxel_size;
  int dim[3], off[3];
  for (int ax = 0; ax < 3; ++ax) {
    dim[ax] = (int)floor(c.grid_range[ax] / vs);
    off[ax] = (int)lround((grid_origin[ax] - c.worigin[ax]) / vs);  // local voxel 0 in world voxels
  }
  return WindowGrid{c.world, c.wdim[0], c.wdim[1], c.wdim[2], off[0], off[1], off[2], dim[0], dim[1], dim[2],
                    (int)ceil((c.grid_z_min - grid_origin[2]) / vs - 1e-9), -1, Cell{seed[0], seed[1], seed[2]}, bits};
}
CD_HD bool seed_in_grid(const Cfg& c, const int seed[3]) {
  for (int ax = 0; ax < 3; ++ax)
    if (seed[ax] < 0 || seed[ax] >= (int)floor(c.grid_range[ax] / c.voxel_size)) return false;
  return true;
}

// the rows of one decomposition into a polyhedron of the corridor
CD_HD int poly_from_rows(int rc, const double* rows, int n, Poly* out) {
  if (rc != hdsm_cd::CD_OK) return HDSM_ERR_CAPACITY;
  out->rows = n;
  for (int r = 0; r < n; ++r) {
    for (int k = 0; k < 3; ++k) out->A[r][k] = rows[4 * r + k];
    out->b[r] = rows[4 * r + 3];
  }
  return HDSM_OK;
}

CD_HD int world_poly(const Cfg& c, const V3& grid_origin, const int seed[3], Work* wk, uint32_t* bits, Poly* out) {
  out->rows = 0;
  if (!seed_in_grid(c, seed)) return HDSM_ERR_BAD_ARG;
  for (int w = 0; w < WindowGrid::WORDS; ++w) bits[w] = 0u;
  const Cell sc{seed[0], seed[1], seed[2]};
  WindowGrid g = make_window(c, grid_origin, seed, bits);
  const bool pinched = hdsm_cd::seed_is_pinched(g, sc);  // AC:1385-1395
  const double org[3] = {grid_origin[0], grid_origin[1], grid_origin[2]};
  double rows[HDSM_MAX_ROWS_STATIC * 4];
  int n = 0;
  const int cap = c.RS < HDSM_MAX_ROWS_STATIC ? c.RS : HDSM_MAX_ROWS_STATIC;
  const int rc = hdsm_cd::decompose_core(g, *wk, (pinched || c.use_cvx_new) ? 1 : 0, sc, c.n_it_decomp, c.voxel_size, -1, org, rows, cap, &n);
  return poly_from_rows(rc, rows, n, out);
}
#if defined(__HIPCC__) || defined(CD_EMU_COOP)
// the same by the whole wavefront (corridor_wave.h): same arguments and the same result in every lane
__device__ inline int world_poly_wave(const Cfg& c, const V3& grid_origin, const int seed[3], const hdsm_cd::WaveLds& lds, Poly* out, int lane) {
  if (!seed_in_grid(c, seed)) {
    out->rows = 0;
    return HDSM_ERR_BAD_ARG;
  }
  const WindowGrid g = make_window(c, grid_origin, seed, nullptr);
  const double org[3] = {grid_origin[0], grid_origin[1], grid_origin[2]};
  static_assert(HDSM_MAX_ROWS_STATIC <= hdsm_cd::WAVE_ROWS, "rows of a polyhedron: room in LDS");
  int n = 0;
  const int cap = c.RS < HDSM_MAX_ROWS_STATIC ? c.RS : HDSM_MAX_ROWS_STATIC;
  const int rc = hdsm_cd::wave_decompose(g, lds, c.use_cvx_new ? 1 : -1, c.n_it_decomp, c.voxel_size, org, lds.rows, cap, &n, lane);
  __syncthreads();
  if (rc != hdsm_cd::CD_OK) {
    out->rows = 0;
    return HDSM_ERR_CAPACITY;
  }
  if (lane == 0) out->rows = n;  // (every lane holds the same rows: the copy into the agent's state is shared out)
  for (int t = lane; t < 4 * n; t += 64) {
    const int r = t >> 2, k = t & 3;
    if (k < 3) out->A[r][k] = lds.rows[t];
    else out->b[r] = lds.rows[t];
  }
  return HDSM_OK;
}
#endif

// GenerateSafeCorridor, AC:1236-1447. `wk`, `bits`: scratch of the voxel decomposition (unused in free space).
CD_HD void corridor_step(const Cfg& c, AgentS& ag, Work* wk, uint32_t* bits) {
  ag.corridor_rc = 0;
  const int P = c.P, N = c.N;
  Poly* fresh = ag.polys;  // the kept polyhedra are compacted in place (kept indices only move down)
  int n_poly = 0;
  bool kept_last = false;
  if (ag.n_poly > 0) {  // AC:1253-1267: the whole previous plan inside the LAST polyhedron -> keep only it
    bool all_in = true;
    if (ag.has_traj)
      for (int j = 0; j <= N; ++j)
        if (!inside(ag.polys[ag.n_poly - 1], V3{{ag.traj_curr[j][0], ag.traj_curr[j][1], ag.traj_curr[j][2]}})) {
          all_in = false;
          break;
        }
    if (all_in) {
      if (ag.n_poly - 1 != 0) fresh[0] = ag.polys[ag.n_poly - 1];
      n_poly = 1, kept_last = true;
    }
  }
  if (ag.n_poly > 0 && !kept_last)  // AC:1273-1282: keep the polyhedra used by the last solve
    for (int i = 0; i < P && i < ag.n_poly; ++i)
      if (ag.poly_used[i]) {
        if (n_poly != i) fresh[n_poly] = ag.polys[i];
        ++n_poly;
      }

  // path: current position pushed in front of the global path (AC:1286-1290). The path thread re-plans from the kept
  // reference points (AC:328-350), so path_curr_ starts at the last reference start; the rest of the polyline follows.
  V3 path[PATH_PTS + 2];
  const V3 path_head = ag.n_ref == 0 ? ag.path[0] : V3{{ag.traj_ref[0][0], ag.traj_ref[0][1], ag.traj_ref[0][2]}};
  path[0] = {{ag.state_curr[0], ag.state_curr[1], ag.state_curr[2]}};
  const int n_path = 1 + path_ahead(ag, path_head, path + 1);
  // local voxel grid origin (env_builder GenerateVoxelGridMSG, environment_builder.cpp:58-67)
  const double vs = c.voxel_size;
  V3 origin;
  for (int ax = 0; ax < 3; ++ax) origin[ax] = floor((ag.state_curr[ax] - c.grid_range[ax] / 2) / vs) * vs;

  int path_idx = 1;
  V3 curr = path[0];
  const double samp = vs / 10;  // AC:1316
  while (n_poly < P) {
    const V3 next = path[path_idx];
    const V3 diff = sub(next, curr);
    const double dist_next = norm(diff);
    if (dist_next > samp) {
      curr = step_along(curr, samp, diff, dist_next);
    } else {
      curr = next;
      if (++path_idx == n_path) break;
    }
    int j_in = -1;
    for (int i = 0; i < n_poly; ++i)
      if (inside(fresh[i], curr)) {
        j_in = i;
        break;
      }
    if (j_in >= 0) {
      if (c.fast_walk && dist_next > samp) {  // (scalar twin of the device walk's shortcut, see swarm_kernels.hip; for tests)
        double t_exit = DBL_MAX;
        const Poly& pj = fresh[j_in];
        for (int r = 0; r < pj.rows; ++r) {
          const double rate = ((pj.A[r][0] * diff[0] + pj.A[r][1] * diff[1]) + pj.A[r][2] * diff[2]) / dist_next;
          const double slack = pj.b[r] - ((pj.A[r][0] * curr[0] + pj.A[r][1] * curr[1]) + pj.A[r][2] * curr[2]);
          if (!(slack > kWalkTol)) t_exit = 0;
          else if (rate > 0) t_exit = fmin(t_exit, (slack - kWalkTol) / rate);
        }
        const double cap = t_exit / samp - 3.0;
        int n_safe = cap > 1e6 ? 1000000 : (cap > 0 ? (int)cap : 0);
        if (!c.has_world) {
          // free space: the skipped samples are not even generated one by one (walk_jump). Only where the polyhedra are the
          // large boxes of an empty grid: next to obstacles a routed path slides along faces, the outcome of the first
          // TESTED sample after the shortcut can hang on the last bit of the position, and the sample-by-sample form below
          // keeps that bit what the reference's loop produces.
          if (n_safe > 0) walk_jump(curr, next, samp, n_safe);
          n_safe = 0;
        }
        for (; n_safe > 0; --n_safe) {
          const V3 df = sub(next, curr);
          const double dn = norm(df);
          if (!(dn > samp)) break;  // the end of the segment: the regular loop takes over
          curr = step_along(curr, samp, df, dn);
        }
      }
      continue;
    }
    V3 seed_pt = curr;  // AC:1351-1354: step back to the previous sample
    if (dist_next > 0) seed_pt = step_along(curr, -fmin(samp, dist_next), diff, dist_next);
    int seed[3];
    V3 seed_world;
    for (int ax = 0; ax < 3; ++ax) {
      seed[ax] = (int)((seed_pt[ax] - origin[ax]) / vs);  // AC:1357-1359 (truncation)
      seed_world[ax] = (seed[ax] * vs + vs / 2) + origin[ax];
    }
    bool previous_seed = false;  // AC:1361-1379
    for (int i = 0; i < n_poly; ++i)
      if (fresh[i].seed[0] == seed_world[0] && fresh[i].seed[1] == seed_world[1] && fresh[i].seed[2] == seed_world[2]) {
        previous_seed = true;
        break;
      }
    if (previous_seed) continue;
    Poly& np = fresh[n_poly];
    if (c.has_world) {
      // a seed outside the local grid or a polyhedron with more rows than the solver takes: stop generating for this agent
      // this round (it keeps the polyhedra it has) and report through hdsm_swarm_corridor_errors
      const int rc = world_poly(c, origin, seed, wk, bits, &np);
      if (rc != HDSM_OK) {
        ag.corridor_rc = rc;
        break;
      }
    } else {
      free_space_poly(c, origin, seed, &np);
    }
    np.seed = seed_world;
    ++n_poly;
  }
  ag.n_poly = n_poly;
}

// ---- the map-dependent half of the reference trajectory (row f1): ComputePathVelocity's voxel term + KeepOnlyFreeReference ----
// The planner's own voxel grid (voxel_grid_, AH:439) as a window of the world: raw values, -1 = unknown (below the ground,
// unknown in the world) — NOT turned into occupied here (only GenerateSafeCorridor does that, AC:1307); outside the local grid
// GetVoxelInt returns -1 (voxel_grid.cpp:110-117); outside the world = free.
struct RawWindow {
  const Cfg* c;
  int off[3], dim[3], ground_k;
  CD_HD bool inside(int i, int j, int k) const { return i >= 0 && j >= 0 && k >= 0 && i < dim[0] && j < dim[1] && k < dim[2]; }
  CD_HD int value(int i, int j, int k) const {
    if (!inside(i, j, k)) return -1;
    if (k < ground_k) return -1;
    const int gi = i + off[0], gj = j + off[1], gk = k + off[2];
    if (gi < 0 || gj < 0 || gk < 0 || gi >= c->wdim[0] || gj >= c->wdim[1] || gk >= c->wdim[2]) return 0;
    return c->world[(size_t)gi + (size_t)gj * c->wdim[0] + (size_t)gk * c->wdim[0] * c->wdim[1]];
  }
};
CD_HD RawWindow raw_window(const Cfg& c, const V3& grid_origin) {
  RawWindow w;
  w.c = &c;
  for (int ax = 0; ax < 3; ++ax) {
    w.dim[ax] = (int)floor(c.grid_range[ax] / c.voxel_size);
    w.off[ax] = (int)lround((grid_origin[ax] - c.worigin[ax]) / c.voxel_size);
  }
  w.ground_k = (int)ceil((c.grid_z_min - grid_origin[2]) / c.voxel_size - 1e-9);
  return w;
}
CD_HD V3 local_grid_origin(const Cfg& c, const AgentS& ag) {  // env_builder GenerateVoxelGridMSG, environment_builder.cpp:58-67
  V3 o;
  for (int ax = 0; ax < 3; ++ax) o[ax] = floor((ag.state_curr[ax] - c.grid_range[ax] / 2) / c.voxel_size) * c.voxel_size;
  return o;
}

// GetVelocityLimit, AC:1805-1817
CD_HD double velocity_limit(const hdsm_ref_config& rc, double occ, double dist) {
  if (occ < 0) occ = 0;
  if (occ > 100) occ = 100;
  const double alpha = 1 - pow(occ / 100, rc.sens_pot) * (1 / exp(rc.sens_dist * dist));
  return rc.path_vel_min + (rc.path_vel_max - rc.path_vel_min) * alpha;
}

CD_HD double rc_mod1(double v) { return fmod(fmod(v, 1.0) + 1.0, 1.0); }
CD_HD double rc_intbound(double s, double ds) {  // raycast.cpp:11-20: smallest positive t with s + t ds integer
  if (ds < 0) return rc_intbound(-s, -ds);
  return (1 - rc_mod1(s)) / ds;
}
CD_HD int rc_signum(int x) { return x == 0 ? 0 : (x < 0 ? -1 : 1); }

// voxel_grid_util::Raycast (raycast.cpp:22-183, Amanatides-Woo with the reference's modifications) from `start` to `end`, both
// in LOCAL VOXEL units. visit(pt) is called for every point the reference appends to its output (the real intersection points
// with the voxel boundaries), in order. Returns true when the ray hit an occupied voxel; `hit` = the collision point.
template <class F>
CD_HD bool raycast(const RawWindow& g, const V3& start, const V3& end, double max_dist, V3* hit, F visit) {
  int x = (int)floor(start[0]), y = (int)floor(start[1]), z = (int)floor(start[2]);
  const int ex = (int)floor(end[0]), ey = (int)floor(end[1]), ez = (int)floor(end[2]);
  const double max2 = max_dist * max_dist;
  const double dx = end[0] - start[0], dy = end[1] - start[1], dz = end[2] - start[2];
  const int sx = rc_signum(ex - x), sy = rc_signum(ey - y), sz = rc_signum(ez - z);
  double tmx = rc_intbound(start[0], dx), tmy = rc_intbound(start[1], dy), tmz = rc_intbound(start[2], dz);
  const double tdx = (double)sx / dx, tdy = (double)sy / dy, tdz = (double)sz / dz;
  if (sx == 0 && sy == 0 && sz == 0) {  // raycast.cpp:96-100: same voxel, no occupancy test
    visit(end);
    visit(start);
    return false;
  }
  double tmax = 0;
  int count = 0;
  for (;;) {
    const double tt = tmax < 1.0 ? tmax : 1.0;
    const V3 real = {{start[0] + tt * dx, start[1] + tt * dy, start[2] + tt * dz}};
    if (g.inside(x, y, z)) {
      if (g.value(x, y, z) == 100 && tmax <= 1) {
        *hit = real;
        visit(real);
        return true;
      }
      visit(real);
      const double ux = x - start[0], uy = y - start[1], uz = z - start[2];
      if ((ux * ux + uy * uy) + uz * uz > max2) break;
      if (++count > 1500) break;  // (the reference throws here)
    }
    if (tmax >= 1) break;
    if ((tmx < tmy && sx != 0) || sy == 0) {
      if ((tmx < tmz && sx != 0) || sz == 0) tmax = tmx, x += sx, tmx += tdx;
      else tmax = tmz, z += sz, tmz += tdz;
    } else {
      if ((tmy < tmz && sy != 0) || sz == 0) tmax = tmy, y += sy, tmy += tdy;
      else tmax = tmz, z += sz, tmz += tdz;
    }
  }
  return false;
}

// The voxel / potential-field term of Agent::ComputePathVelocity (AC:1709-1766) for the polyline `pts` (world coordinates;
// pts[0] = the sampling start): the minimum of GetVelocityLimit over the voxels the path crosses inside the agent's local grid.
// Quirks kept: the distance of a visited voxel is measured between path_start in WORLD metres and the visited point in LOCAL
// voxel units, times the voxel size (AC:1739); after a collision the distance is in voxel units (AC:1755) and the walk stops.
// One segment of that walk: the smallest limit it contributes (path_vel_max if none) and whether it ended in a collision (the
// reference stops at the first segment that does). The segments do not depend on each other: the device gives one to each lane.
CD_HD double voxel_velocity_cap_segment(const Cfg& c, const hdsm_ref_config& rc, const RawWindow& g, const V3& grid_origin, const V3& path_start,
                                        const V3& a, const V3& b, bool* collided_out) {
  double path_vel = rc.path_vel_max;
  const double vs = c.voxel_size;
  auto local = [&](const V3& p) { return V3{{(p[0] - grid_origin[0]) / vs, (p[1] - grid_origin[1]) / vs, (p[2] - grid_origin[2]) / vs}}; };
  auto consider = [&](const V3& pt) {
    double val = (double)g.value((int)pt[0], (int)pt[1], (int)pt[2]);  // GetVoxelInt(Vector3d): truncation
    if (val == -1) val = 100;
    const double v = velocity_limit(rc, val, norm(sub(path_start, pt)) * vs);
    if (v < path_vel) path_vel = v;
  };
  const V3 start = local(a), end = local(b);
  V3 hit = {{-1, -1, -1}};
  // a segment is scanned twice in the reference too: IsLineClear decides, then the visited points are weighed
  const bool collided = raycast(g, start, end, norm(sub(start, end)), &hit, [](const V3&) {});
  if (!collided) {
    raycast(g, start, end, norm(sub(start, end)), &hit, consider);
    consider(start);
  } else {
    const double val = (double)(int8_t)g.value((int)hit[0], (int)hit[1], (int)hit[2]);
    const double v = velocity_limit(rc, val, norm(sub(start, hit)));
    if (v < path_vel) path_vel = v;
  }
  *collided_out = collided;
  return path_vel;
}
CD_HD double voxel_velocity_cap(const Cfg& c, const hdsm_ref_config& rc, const V3& grid_origin, const V3* pts, int n) {
  double path_vel = rc.path_vel_max;
  if (!c.has_world || n < 1) return path_vel;
  const RawWindow g = raw_window(c, grid_origin);
  for (int i = 0; i + 1 < n; ++i) {
    bool collided = false;
    const double v = voxel_velocity_cap_segment(c, rc, g, grid_origin, pts[0], pts[i], pts[i + 1], &collided);
    if (v < path_vel) path_vel = v;
    if (collided) break;
  }
  return path_vel;
}

// Agent::KeepOnlyFreeReference (AC:1665-1693) on the n reference points `ref` (rows of 6: position, velocity) followed by the
// velocity references of AC:1527-1547 recomputed on the result: from the first point that lies in an unknown or occupied voxel
// of the local grid (outside the grid = unknown) the last free point is repeated.
CD_HD void keep_only_free(const Cfg& c, const V3& grid_origin, double path_vel, double (*ref)[6], int n) {
  if (!c.has_world || n < 2) return;
  const RawWindow g = raw_window(c, grid_origin);
  const double vs = c.voxel_size;
  int stop = n;
  for (int i = 1; i < n; ++i) {
    const int v = g.value((int)((ref[i][0] - grid_origin[0]) / vs), (int)((ref[i][1] - grid_origin[1]) / vs),
                          (int)((ref[i][2] - grid_origin[2]) / vs));
    if (v == -1 || v == 100) {
      stop = i;
      break;
    }
  }
  if (stop == n) return;
  for (int i = stop; i < n; ++i)
    for (int k = 0; k < 3; ++k) ref[i][k] = ref[stop - 1][k];
  double v[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i) {
    if (i + 1 < n) {
      const V3 d = {{ref[i][0] - ref[i + 1][0], ref[i][1] - ref[i + 1][1], ref[i][2] - ref[i + 1][2]}};
      const double dist = norm(d);
      for (int k = 0; k < 3; ++k) v[k] = dist > 1e-2 ? path_vel * d[k] / dist : 0.0;
    }
    for (int k = 0; k < 3; ++k) ref[i][3 + k] = v[k];
  }
}

// GetPathProgress (path_finding_util/src/path_tools.cpp:419-479) + CheckReferenceTrajIncrement (AC:569-585)
CD_HD void check_increment(const Cfg& c, AgentS& ag) {
  ag.increment = 0;
  if (!ag.has_traj || ag.n_ref < 2) return;
  const V3 pt = {{ag.traj_curr[1][0], ag.traj_curr[1][1], ag.traj_curr[1][2]}};
  auto ref_pt = [&](int i) { return V3{{ag.traj_ref[i][0], ag.traj_ref[i][1], ag.traj_ref[i][2]}}; };
  V3 curr = ref_pt(0);
  double dist_min = norm(sub(pt, curr)), progress = 0, progress_final = 0, proj_dist = dist_min;
  int idx = 1;
  V3 target = ref_pt(1);  // (kept in registers: it changes N times, the loop runs ~100 times per metre of reference)
  const double samp = 0.01;
  while (idx < ag.n_ref) {
    const V3 diff = sub(target, curr);
    const double dist_next = norm(diff);
    if (dist_next > samp) {
      curr = step_along(curr, samp, diff, dist_next);
      progress += samp;
    } else {
      curr = target;
      ++idx;
      if (idx < ag.n_ref) target = ref_pt(idx);
      progress += dist_next;
    }
    const double d = norm(sub(pt, curr));
    if (d < dist_min) {
      dist_min = d;
      proj_dist = d;
      progress_final = progress;
    }
  }
  if (progress_final > 0 && proj_dist < c.thresh_dist) ag.increment = 1;
}

// Read-back of one agent's solver outputs (AC:960-987) or the shift-by-one fallback (AC:1000-1019)
CD_HD void commit_copy(const Cfg& c, AgentS& ag, const double* traj /*[N+1][9]*/, const double* ctrl /*[N][3]*/, const uint8_t* used /*[P]*/,
                       int status) {
  const int N = c.N, P = c.P;
  if (status != HDSM_NO_SOLUTION) {  // AC:960-987
    for (int i = 0; i <= N; ++i)
      for (int k = 0; k < 9; ++k) ag.traj_curr[i][k] = traj[i * 9 + k];
    for (int i = 0; i < N; ++i)
      for (int k = 0; k < 3; ++k) ag.ctrl_curr[i][k] = ctrl[i * 3 + k];
    for (int j = 0; j < P; ++j) ag.poly_used[j] = used[j];
    ag.has_traj = 1;
  } else {  // AC:1000-1019: drop the first state/control of the previous plan, duplicate the last
    ++ag.n_fail;
    if (ag.has_traj) {
      for (int i = 0; i < N; ++i)
        for (int k = 0; k < 9; ++k) ag.traj_curr[i][k] = ag.traj_curr[i + 1][k];
      for (int i = 0; i + 1 < N; ++i)
        for (int k = 0; k < 3; ++k) ag.ctrl_curr[i][k] = ag.ctrl_curr[i + 1][k];
    }
  }
}

// One segment of the walk of check_increment, on its own: the walker always lands exactly ON a reference point before it
// starts the next segment, so the samples of segment `seg` (from ref[seg] towards ref[seg + 1], steps of 0.01 m, then the end
// point) — and their distances to `pt` — do not depend on the other segments. Returns the smallest distance of the segment.
CD_HD double increment_segment_min(const AgentS& ag, int seg, const V3& pt) {
  V3 curr = {{ag.traj_ref[seg][0], ag.traj_ref[seg][1], ag.traj_ref[seg][2]}};
  const V3 target = {{ag.traj_ref[seg + 1][0], ag.traj_ref[seg + 1][1], ag.traj_ref[seg + 1][2]}};
  const double samp = 0.01;
  double best = DBL_MAX;
  for (;;) {
    const V3 diff = sub(target, curr);
    const double dist_next = norm(diff);
    const bool last = !(dist_next > samp);
    curr = last ? target : step_along(curr, samp, diff, dist_next);
    const double d = norm(sub(pt, curr));
    if (d < best) best = d;
    if (last) break;
  }
  return best;
}
// The same minimum in closed form, to about 1e-11 m: the samples of a segment lie on the straight line from ref[seg] to
// ref[seg + 1] at multiples of 0.01 m (the literal walk accumulates a rounding of a few ulp of the position per step — with
// ~100 steps per segment and positions of hundreds of metres that is ~1e-11 m), then the end point; the distance to `pt` along
// a line is convex, so the closest sample is one of the two next to the foot of the perpendicular, or the end point. Whether the
// count of full steps is m or m + 1 when |segment| / 0.01 is within rounding of an integer does not matter: the extra sample
// then coincides with the end point to the same accuracy. The caller uses this value only when the decision of
// increment_from_minima does not depend on 1e-9 m, and walks the segment literally otherwise (k_commit).
CD_HD double increment_segment_min_closed_form(const AgentS& ag, int seg, const V3& pt) {
  const V3 s0 = {{ag.traj_ref[seg][0], ag.traj_ref[seg][1], ag.traj_ref[seg][2]}};
  const V3 target = {{ag.traj_ref[seg + 1][0], ag.traj_ref[seg + 1][1], ag.traj_ref[seg + 1][2]}};
  const double samp = 0.01;
  const V3 diff = sub(target, s0);
  const double len = norm(diff);
  double best = norm(sub(pt, target));
  if (!(len > samp)) return best;  // (a single sample: the end point)
  const V3 dir = {{diff[0] / len, diff[1] / len, diff[2] / len}};
  const double foot = dot(sub(pt, s0), dir) / samp;
  const double m = ceil(len / samp) - 1.0;  // full steps before the end point (>= 1 here)
  double k0 = floor(foot);
  k0 = k0 < 1.0 ? 1.0 : (k0 > m ? m : k0);
  const double k1 = k0 + 1.0 > m ? m : k0 + 1.0;
  for (int u = 0; u < 2; ++u) {
    const double t = (u == 0 ? k0 : k1) * samp;
    const V3 q = {{s0[0] + t * dir[0], s0[1] + t * dir[1], s0[2] + t * dir[2]}};
    const double d = norm(sub(pt, q));
    if (d < best) best = d;
  }
  return best;
}
// check_increment from the per-segment minima: the reference advances iff some sample after the starting point is STRICTLY
// closer to p_1 than the starting point (then progress_final > 0) and the closest sample is within thresh_dist.
CD_HD int increment_from_minima(const Cfg& c, double d_start, double d_min_rest) {
  return (d_min_rest < d_start && d_min_rest < c.thresh_dist) ? 1 : 0;
}

// Consumes one agent's solver outputs (AC:960-1019, 182, 233-238); returns 1 if the agent has a plan to publish.
CD_HD int commit_one(const Cfg& c, AgentS& ag, const double* traj /*[N+1][9]*/, const double* ctrl /*[N][3]*/, const uint8_t* used /*[P]*/,
                     int status) {
  commit_copy(c, ag, traj, ctrl, used, status);
  if (ag.has_traj) {
    check_increment(c, ag);  // AC:182
    for (int k = 0; k < 9; ++k) ag.state_curr[k] = ag.traj_curr[c.step_plan][k];  // AC:233-238
  }
  return ag.has_traj;
}

// the solver inputs of one agent (layouts of include/hdsm.h)
CD_HD void fill_inputs(const Cfg& c, const AgentS& ag, int32_t* agent_id, double* state_curr, double* traj_ref, int32_t* n_poly,
                       int32_t* n_rows_static, double* A_static, double* b_static) {
  const int N = c.N, P = c.P, RS = c.RS;
  *agent_id = ag.id;
  for (int k = 0; k < 9; ++k) state_curr[k] = ag.state_curr[k];
  for (int i = 0; i < N; ++i)
    for (int k = 0; k < 6; ++k) traj_ref[i * 6 + k] = ag.traj_ref[i][k];
  *n_poly = ag.n_poly;
  for (int j = 0; j < P; ++j) {
    const bool have = j < ag.n_poly;
    n_rows_static[j] = have ? ag.polys[j].rows : 0;
    for (int r = 0; r < RS; ++r) {
      const bool hr = have && r < ag.polys[j].rows;
      for (int k = 0; k < 3; ++k) A_static[(j * RS + r) * 3 + k] = hr ? ag.polys[j].A[r][k] : 0.0;
      b_static[j * RS + r] = hr ? ag.polys[j].b[r] : 0.0;
    }
  }
}

}  // namespace hdsm_sw
