// swarm_host.cpp — host-side planner state around the solve (see include/hdsm_swarm.h).
// AC = multi_agent_planner/src/agent_class.cpp of lis-epfl/multi_agent_pkgs. Pure host C++.
#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <queue>
#include <thread>
#include <unordered_map>
#include <vector>

#include <chrono>
#include <ctime>

#include "../../include/hdsm_stats.h"
#include "../../include/hdsm_swarm.h"
#include "swarm_core.h"

namespace {

using hdsm_sw::AgentS;
using hdsm_sw::axpy;
using hdsm_sw::dot;
using hdsm_sw::norm;
using hdsm_sw::on_segment;
using hdsm_sw::Poly;
using hdsm_sw::sub;
using hdsm_sw::V3;

// host-only companions of an agent's plain state (hdsm_sw::AgentS)
struct AgentX {
  void* stats = nullptr;         // hdsm_stats record of this agent (comp_time_*_, state_hist_; f3)
  double sc_ms = 0, ref_ms = 0;  // CPU time of this round's corridor / reference generation
  double yaw = 0;                // yaw_ (AC:16, 1025-1051)
};

struct Swarm {
  hdsm_params prm;
  hdsm_swarm_config cfg;
  int n_rob = 0, first_id = 0, n_local = 0;
  std::vector<AgentS> agents;
  std::vector<AgentX> extra;
  std::unique_ptr<hdsm_cd::Work> work{new hdsm_cd::Work};  // scratch of the voxel decomposition
  std::vector<uint32_t> bits = std::vector<uint32_t>(hdsm_cd::WindowGrid::WORDS);
  hdsm_sw::Cfg core_cfg() const {
    hdsm_sw::Cfg c{};
    c.N = prm.n_hor, c.P = prm.poly_hor, c.RS = prm.max_rows_static, c.step_plan = cfg.step_plan;
    c.n_it_decomp = cfg.n_it_decomp, c.use_cvx_new = cfg.use_cvx_new, c.has_world = has_world ? 1 : 0;
    c.voxel_size = cfg.voxel_size, c.grid_z_min = cfg.grid_z_min, c.thresh_dist = cfg.thresh_dist;
    for (int k = 0; k < 3; ++k) c.grid_range[k] = cfg.grid_range[k], c.wdim[k] = wdim[k], c.worigin[k] = worigin[k];
    c.world = has_world ? world.data() : nullptr;
    c.fast_walk = 1;  // the same walk as the device loop (shortcut past samples provably inside a kept polyhedron); 0 = every sample generated and tested
    if (const char* e = std::getenv("HDSM_FAST_WALK_HOST"))  // test hook (scalar twin of the device shortcut): "0" or "1", anything else is ignored
      if ((e[0] == '0' || e[0] == '1') && e[1] == 0) c.fast_walk = e[0] == '1';
    return c;
  }
  // optional occupancy of the world (hdsm_swarm_set_world): voxels of cfg.voxel_size, >= 100 occupied
  bool has_world = false;
  std::vector<int8_t> world;
  int wdim[3] = {0, 0, 0};
  double worigin[3] = {0, 0, 0};
  // timing of the round in flight (f3): the duration of the fused launch as reported by the caller, wall clock at prepare
  double solve_ms = 0;
  std::chrono::steady_clock::time_point t_round{};
  long long round_idx = 0;
  bool corridor_done = false;  // hdsm_swarm_prepare_corridor already ran this round
  ~Swarm() {
    for (AgentX& a : extra) hdsm_stats_destroy(a.stats);
  }
};

inline double cpu_ms_since(clock_t t0) { return (double)(clock() - t0) / CLOCKS_PER_SEC * 1e3; }

// GetVelocityLimit, AC:1805-1817
double velocity_limit(const hdsm_swarm_config& c, double occ, double dist) {
  if (occ < 0) occ = 0;
  if (occ > 100) occ = 100;
  const double alpha = 1 - std::pow(occ / 100, c.sens_pot) * (1 / std::exp(c.sens_dist * dist));
  return c.path_vel_min + (c.path_vel_max - c.path_vel_min) * alpha;
}

// ComputePathVelocity, AC:1695-1803: empty world -> only the neighbour term (AC:1769-1801) is active
hdsm_ref_config ref_config(const Swarm& sw) {
  return {sw.cfg.path_vel_min, sw.cfg.path_vel_max, sw.cfg.sens_dist, sw.cfg.sens_pot, sw.cfg.sens_other_agents, sw.cfg.path_vel_dec};
}

double compute_path_velocity(const Swarm& sw, const AgentS& ag, const std::vector<V3>& path, const double* plans_all,
                             const uint8_t* has_plan) {
  const int N = sw.prm.n_hor;
  // voxel / potential-field term (AC:1709-1766) on the agent's local grid, then the neighbour term (AC:1769-1801)
  const hdsm_sw::Cfg cc = sw.core_cfg();
  double path_vel = hdsm_sw::voxel_velocity_cap(cc, ref_config(sw), hdsm_sw::local_grid_origin(cc, ag), path.data(), (int)path.size());
  for (int i = 0; i < (ag.has_traj ? N + 1 : 0); ++i) {
    const V3 start = {{ag.traj_curr[i][0], ag.traj_curr[i][1], ag.traj_curr[i][2]}};
    const double occ = 100 * std::pow(sw.cfg.sens_other_agents, (double)i);
    for (int j = 0; j < sw.n_rob; ++j) {
      if (j == ag.id || !has_plan[j]) continue;
      const double* st = plans_all + ((size_t)j * (N + 1) + i) * 9;
      const double d = norm(sub(start, V3{{st[0], st[1], st[2]}}));
      const double v = velocity_limit(sw.cfg, occ, d);
      if (v < path_vel) path_vel = v;
    }
  }
  return path_vel;
}

// SamplePath, AC:1591-1663
std::vector<V3> sample_path(const Swarm& sw, AgentS& ag, const std::vector<V3>& path, const double* plans_all,
                            const uint8_t* has_plan) {
  const int N = sw.prm.n_hor;
  std::vector<V3> ref;
  if (path.size() < 2) {
    for (int i = 0; i < N; ++i) ref.push_back(path[0]);
    return ref;
  }
  ag.path_vel = compute_path_velocity(sw, ag, path, plans_all, has_plan);
  const double samp_dist = ag.path_vel * sw.prm.dt;
  size_t path_idx = 1;
  int ref_idx = 0;
  V3 curr = path[0];
  ref.push_back(path.front());
  double limit = samp_dist;
  while (ref_idx < N) {
    const V3 diff = sub(path[path_idx], curr);
    const double dist_next = norm(diff);
    if (dist_next > limit) {
      curr = hdsm_sw::step_along(curr, limit, diff, dist_next);
      ref.push_back(curr);
      ++ref_idx;
      limit = std::fmax(0.0, samp_dist - sw.cfg.path_vel_dec * sw.prm.dt);
    } else {
      curr = path[path_idx];
      if (++path_idx == path.size()) {
        for (int i = ref_idx; i < N; ++i) ref.push_back(path.back());
        return ref;
      }
      limit -= dist_next;
    }
  }
  return ref;
}

std::vector<V3> reference_polyline(const AgentS& ag) {
  V3 buf[hdsm_sw::PATH_PTS + 1];
  const int n = hdsm_sw::reference_polyline(ag, buf);
  return std::vector<V3>(buf, buf + n);
}

// GenerateReferenceTrajectory, AC:1449-1553
void generate_reference(const Swarm& sw, AgentS& ag, const double* plans_all, const uint8_t* has_plan) {
  const std::vector<V3> path_samp = reference_polyline(ag);  // AC:1459-1496
  std::vector<V3> pts = sample_path(sw, ag, path_samp, plans_all, has_plan);
  // velocity reference, AC:1527-1547 (points backwards along the path; reproduced as is)
  ag.n_ref = (int)pts.size();
  double v[3] = {0, 0, 0};
  for (size_t i = 0; i < pts.size(); ++i) {
    if (i + 1 < pts.size()) {
      const V3 d = sub(pts[i], pts[i + 1]);
      const double dist = norm(d);
      for (int k = 0; k < 3; ++k) v[k] = dist > 1e-2 ? ag.path_vel * d[k] / dist : 0.0;
    }
    for (int k = 0; k < 3; ++k) ag.traj_ref[i][k] = pts[i][k], ag.traj_ref[i][3 + k] = v[k];
  }
  const hdsm_sw::Cfg cc = sw.core_cfg();
  hdsm_sw::keep_only_free(cc, hdsm_sw::local_grid_origin(cc, ag), ag.path_vel, ag.traj_ref, ag.n_ref);  // AC:1512-1514
}

// ---- a minimal router on the world grid (see hdsm_swarm_route in hdsm_swarm.h) -------------------------------------
struct Router {
  const Swarm& sw;
  std::vector<uint8_t> cls;  // 0 free, 1 within two voxels of an obstacle, 2 occupied (or below the ground)
  int nx, ny, nz;
  explicit Router(const Swarm& s) : sw(s), nx(s.wdim[0]), ny(s.wdim[1]), nz(s.wdim[2]) {
    const size_t tot = (size_t)nx * ny * nz;
    cls.assign(tot, 0);
    const double vs = sw.cfg.voxel_size;
    const int k_ground = (int)std::ceil((sw.cfg.grid_z_min - sw.worigin[2]) / vs - 1e-9);
    for (int k = 0; k < nz; ++k)
      for (int j = 0; j < ny; ++j)
        for (int i = 0; i < nx; ++i) {
          const int8_t v = sw.world[idx(i, j, k)];
          if (v >= 100 || v < 0 || k < k_ground) cls[idx(i, j, k)] = 2;
        }
    // "near" band: separable box dilation of the occupied set by two voxels
    std::vector<uint8_t> a(tot), b(tot);
    for (size_t t = 0; t < tot; ++t) a[t] = cls[t] == 2;
    auto pass = [&](const std::vector<uint8_t>& in, std::vector<uint8_t>& out, int ax) {
      const int n[3] = {nx, ny, nz};
      const size_t st[3] = {1, (size_t)nx, (size_t)nx * ny};
      for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
          for (int i = 0; i < nx; ++i) {
            const int c[3] = {i, j, k};
            uint8_t v = 0;
            for (int d = -2; d <= 2 && !v; ++d) {
              const int q = c[ax] + d;
              if (q >= 0 && q < n[ax]) v = in[idx(i, j, k) + (size_t)((long)d * (long)st[ax])];
            }
            out[idx(i, j, k)] = v;
          }
    };
    pass(a, b, 0), pass(b, a, 1), pass(a, b, 2);
    for (size_t t = 0; t < tot; ++t)
      if (b[t] && cls[t] == 0) cls[t] = 1;
  }
  size_t idx(int i, int j, int k) const { return (size_t)i + (size_t)j * nx + (size_t)k * nx * ny; }
  bool in(int i, int j, int k) const { return i >= 0 && j >= 0 && k >= 0 && i < nx && j < ny && k < nz; }
  bool blocked(int i, int j, int k) const { return !in(i, j, k) || cls[idx(i, j, k)] == 2; }
  void voxel_of(const V3& p, int v[3]) const {
    for (int ax = 0; ax < 3; ++ax) v[ax] = (int)std::floor((p[ax] - sw.worigin[ax]) / sw.cfg.voxel_size);
  }
  V3 centre(int i, int j, int k) const {
    const double vs = sw.cfg.voxel_size;
    return {sw.worigin[0] + (i + 0.5) * vs, sw.worigin[1] + (j + 0.5) * vs, sw.worigin[2] + (k + 0.5) * vs};
  }
  // every voxel touched by the segment (sampled at a quarter voxel) is free — with `strict`, also outside the band next to
  // obstacles; points outside the world count as blocked
  bool line_clear(const V3& a, const V3& b, bool strict = false) const {
    const double len = norm(sub(b, a)), step = sw.cfg.voxel_size / 4;
    const int n = (int)std::ceil(len / step);
    for (int t = 0; t <= n; ++t) {
      const V3 p = axpy(a, n ? (double)t / n : 0.0, sub(b, a));
      int v[3];
      voxel_of(p, v);
      if (blocked(v[0], v[1], v[2])) return false;
      if (strict && cls[idx(v[0], v[1], v[2])] != 0) return false;
    }
    return true;
  }
  // nearest free voxel to v (breadth-first over a small cube), false if none within 6 voxels
  bool nearest_free(int v[3]) const {
    if (!blocked(v[0], v[1], v[2])) return true;
    for (int r = 1; r <= 6; ++r) {
      double best = 1e300;
      int bv[3] = {0, 0, 0};
      for (int dk = -r; dk <= r; ++dk)
        for (int dj = -r; dj <= r; ++dj)
          for (int di = -r; di <= r; ++di) {
            if (std::max(std::abs(di), std::max(std::abs(dj), std::abs(dk))) != r) continue;
            if (blocked(v[0] + di, v[1] + dj, v[2] + dk)) continue;
            const double d2 = di * di + dj * dj + dk * dk;
            if (d2 < best) best = d2, bv[0] = v[0] + di, bv[1] = v[1] + dj, bv[2] = v[2] + dk;
          }
      if (best < 1e300) {
        v[0] = bv[0], v[1] = bv[1], v[2] = bv[2];
        return true;
      }
    }
    return false;
  }
  // weighted A* (26-connected, Euclidean heuristic x 2: greedy enough to run straight through a forest and to flood only
  // the face of a wall until it finds a gap; cells of the near band cost 2x), searched inside a box around start and goal
  // (dense arrays, one workspace per thread), then greedy shortening. The whole world is searched if the box has no route.
  struct Work {
    std::vector<float> g;
    std::vector<int32_t> parent;
    std::vector<uint8_t> state;
  };
  bool route(const V3& start, const V3& goal, std::vector<V3>* out, Work& wk) const {
    int s[3], g[3];
    voxel_of(start, s), voxel_of(goal, g);
    if (!in(s[0], s[1], s[2]) || !in(g[0], g[1], g[2])) return false;
    if (!nearest_free(s) || !nearest_free(g)) return false;
    // the planner only ever sees a local grid of grid_range[2] metres of height around the agent: the route stays within
    // half of that above and below the start / goal altitudes (it does not climb over a forest)
    const int band = (int)std::floor(0.5 * sw.cfg.grid_range[2] / sw.cfg.voxel_size);
    const int n[3] = {nx, ny, nz};
    std::vector<int32_t> chain;
    int lo[3], hi[3], ext[3];
    bool found = false;
    for (int attempt = 0; attempt < 2 && !found; ++attempt) {
      const int margin[3] = {attempt ? nx : 12, attempt ? ny : 40, band};
      size_t cells = 1;
      for (int ax = 0; ax < 3; ++ax) {
        lo[ax] = std::max(0, std::min(s[ax], g[ax]) - margin[ax]);
        hi[ax] = std::min(n[ax] - 1, std::max(s[ax], g[ax]) + margin[ax]);
        ext[ax] = hi[ax] - lo[ax] + 1;
        cells *= (size_t)ext[ax];
      }
      if (cells > (size_t)400000000) return false;
      wk.g.assign(cells, 0.0f), wk.parent.assign(cells, -1), wk.state.assign(cells, 0);
      auto lid = [&](int i, int j, int k) { return (int32_t)((i - lo[0]) + ext[0] * ((j - lo[1]) + ext[1] * (k - lo[2]))); };
      auto h = [&](int i, int j, int k) {
        const double a = i - g[0], b = j - g[1], c = k - g[2];
        return 2.0 * std::sqrt(a * a + b * b + c * c);
      };
      struct Node {
        float f;
        int32_t id;
        bool operator<(const Node& o) const { return f > o.f; }
      };
      std::priority_queue<Node> open;
      const int32_t sid = lid(s[0], s[1], s[2]), gid = lid(g[0], g[1], g[2]);
      wk.state[sid] = 1, wk.parent[sid] = sid;
      open.push({(float)h(s[0], s[1], s[2]), sid});
      while (!open.empty()) {
        const Node cur = open.top();
        open.pop();
        if (wk.state[cur.id] == 2) continue;
        wk.state[cur.id] = 2;
        if (cur.id == gid) {
          found = true;
          break;
        }
        const int ci = lo[0] + cur.id % ext[0], cj = lo[1] + (cur.id / ext[0]) % ext[1], ck = lo[2] + cur.id / (ext[0] * ext[1]);
        const float gc = wk.g[cur.id];
        for (int dk = -1; dk <= 1; ++dk)
          for (int dj = -1; dj <= 1; ++dj)
            for (int di = -1; di <= 1; ++di) {
              if (!di && !dj && !dk) continue;
              const int ni = ci + di, nj = cj + dj, nk = ck + dk;
              if (ni < lo[0] || ni > hi[0] || nj < lo[1] || nj > hi[1] || nk < lo[2] || nk > hi[2]) continue;
              if (blocked(ni, nj, nk)) continue;
              // no squeezing diagonally between two blocked voxels
              if (di && dj && (blocked(ci + di, cj, ck) || blocked(ci, cj + dj, ck))) continue;
              if (di && dk && (blocked(ci + di, cj, ck) || blocked(ci, cj, ck + dk))) continue;
              if (dj && dk && (blocked(ci, cj + dj, ck) || blocked(ci, cj, ck + dk))) continue;
              const int32_t nid = lid(ni, nj, nk);
              if (wk.state[nid] == 2) continue;
              const float w = std::sqrt((float)(di * di + dj * dj + dk * dk)) * (cls[idx(ni, nj, nk)] == 1 ? 2.0f : 1.0f);
              const float ng = gc + w;
              if (wk.state[nid] == 0 || ng < wk.g[nid]) {
                wk.state[nid] = 1, wk.g[nid] = ng, wk.parent[nid] = cur.id;
                open.push({ng + (float)h(ni, nj, nk), nid});
              }
            }
      }
      if (found) {
        for (int32_t id = gid;; id = wk.parent[id]) {
          chain.push_back(id);
          if (id == sid) break;
        }
      }
    }
    if (!found) return false;
    std::vector<V3> raw;
    for (auto it = chain.rbegin(); it != chain.rend(); ++it)
      raw.push_back(centre(lo[0] + *it % ext[0], lo[1] + (*it / ext[0]) % ext[1], lo[2] + *it / (ext[0] * ext[1])));
    // the true end points replace the voxel centres when they can be reached in a straight line
    if (line_clear(start, raw.front())) raw.insert(raw.begin(), start);
    if (line_clear(raw.back(), goal)) raw.push_back(goal);
    out->clear();
    out->push_back(raw.front());
    size_t i = 0;
    while (i + 1 < raw.size()) {
      size_t j = raw.size() - 1;
      // shortcuts keep the clearance the search paid for: they may not enter the band next to obstacles (where the raw path
      // itself runs through that band — a narrow passage — its cells are kept one by one)
      while (j > i + 1 && !line_clear(raw[i], raw[j], true)) --j;
      if (j == i + 1) {  // inside the band: plain line of sight, over a short stretch only
        j = std::min(raw.size() - 1, i + 20);
        while (j > i + 1 && !line_clear(raw[i], raw[j])) --j;
      }
      out->push_back(raw[j]);
      i = j;
    }
    return out->size() >= 2;
  }
};

}  // namespace

extern "C" {

void hdsm_swarm_default_config(hdsm_swarm_config* c) {
  if (!c) return;
  c->path_vel_min = 4.5, c->path_vel_max = 9.0, c->sens_dist = 0.05, c->sens_pot = 0.18;
  c->sens_other_agents = 1.0, c->path_vel_dec = 0.0, c->thresh_dist = 1.0, c->voxel_size = 0.3;
  c->grid_range[0] = 20.0, c->grid_range[1] = 20.0, c->grid_range[2] = 6.0, c->grid_z_min = 0.0;
  c->n_it_decomp = 42, c->step_plan = 1, c->use_cvx_new = 0, c->reserved0 = 0;
}

int hdsm_swarm_create(const hdsm_params* prm, const hdsm_swarm_config* cfg, int32_t n_rob, int32_t first_id,
                      int32_t n_local, const double* starts, const double* goals, void** swarm) {
  if (!prm || !cfg || !starts || !goals || !swarm) return HDSM_ERR_BAD_ARG;
  if (n_rob < 1 || n_local < 0 || first_id < 0 || first_id + n_local > n_rob) return HDSM_ERR_BAD_ARG;
  if (prm->n_hor < 2 || prm->n_hor > HDSM_MAX_HOR || prm->poly_hor < 1 || prm->poly_hor > HDSM_MAX_POLY)
    return HDSM_ERR_BAD_ARG;
  if (prm->max_rows_static < 6 || cfg->step_plan < 1 || cfg->step_plan > prm->n_hor) return HDSM_ERR_BAD_ARG;
  Swarm* sw = new (std::nothrow) Swarm;
  if (!sw) return HDSM_ERR_DEVICE;
  sw->prm = *prm, sw->cfg = *cfg, sw->n_rob = n_rob, sw->first_id = first_id, sw->n_local = n_local;
  sw->agents.assign(n_local, AgentS{});
  sw->extra.assign(n_local, AgentX{});
  for (int k = 0; k < n_local; ++k) {
    AgentS& a = sw->agents[k];
    a.id = first_id + k;
    for (int ax = 0; ax < 3; ++ax) a.start[ax] = starts[3 * k + ax], a.goal[ax] = goals[3 * k + ax];
    a.n_path = 2, a.path[0] = a.start, a.path[1] = a.goal;
    sw->extra[k].stats = hdsm_stats_create(a.id, n_rob);
    for (int ax = 0; ax < 3; ++ax) a.state_curr[ax] = a.start[ax];
  }
  *swarm = sw;
  return HDSM_OK;
}

void hdsm_swarm_destroy(void* swarm) { delete static_cast<Swarm*>(swarm); }

int hdsm_swarm_prepare(void* swarm, const double* plans_all, const uint8_t* has_plan, int32_t* agent_id,
                       double* state_curr, double* traj_ref, int32_t* n_poly, int32_t* n_rows_static,
                       double* A_static, double* b_static) {
  Swarm* sw = static_cast<Swarm*>(swarm);
  if (!sw || !plans_all || !has_plan || !agent_id || !state_curr || !traj_ref || !n_poly || !n_rows_static ||
      !A_static || !b_static)
    return HDSM_ERR_BAD_ARG;
  const int N = sw->prm.n_hor, P = sw->prm.poly_hor, RS = sw->prm.max_rows_static;
  sw->t_round = std::chrono::steady_clock::now();
  sw->solve_ms = 0;
  const hdsm_sw::Cfg cc = sw->core_cfg();
  for (int k = 0; k < sw->n_local; ++k) {
    AgentS& ag = sw->agents[k];
    clock_t t0 = clock();
    if (!sw->corridor_done) {
      hdsm_sw::corridor_step(cc, ag, sw->work.get(), sw->bits.data());  // AC:165
      sw->extra[k].sc_ms = cpu_ms_since(t0);                               // comp_time_sc_, AC:1446
    }
    t0 = clock();
    if (!ag.external_ref) generate_reference(*sw, ag, plans_all, has_plan);  // AC:171 (or done on the device, f1)
    sw->extra[k].ref_ms = cpu_ms_since(t0);
    ag.external_ref = 0;
    hdsm_sw::fill_inputs(cc, ag, agent_id + k, state_curr + 9 * (size_t)k, traj_ref + (size_t)k * N * 6, n_poly + k,
                         n_rows_static + (size_t)k * P, A_static + (size_t)k * P * RS * 3, b_static + (size_t)k * P * RS);
  }
  sw->corridor_done = false;
  return HDSM_OK;
}

int hdsm_swarm_commit(void* swarm, const double* traj_out, const double* ctrl_out, const uint8_t* poly_used,
                      const int32_t* status, double* plans_local, uint8_t* has_plan_local) {
  Swarm* sw = static_cast<Swarm*>(swarm);
  if (!sw || !traj_out || !ctrl_out || !poly_used || !status || !plans_local || !has_plan_local)
    return HDSM_ERR_BAD_ARG;
  const int N = sw->prm.n_hor, P = sw->prm.poly_hor;
  const hdsm_sw::Cfg cc = sw->core_cfg();
  for (int k = 0; k < sw->n_local; ++k) {
    AgentS& ag = sw->agents[k];
    const int have_plan = hdsm_sw::commit_one(cc, ag, traj_out + (size_t)k * (N + 1) * 9, ctrl_out + (size_t)k * N * 3,
                                              poly_used + (size_t)k * P, status[k]);
    has_plan_local[k] = have_plan ? 1 : 0;
    for (int i = 0; i <= N; ++i)
      for (int c = 0; c < 9; ++c)
        plans_local[((size_t)k * (N + 1) + i) * 9 + c] = have_plan ? ag.traj_curr[i][c] : 0.0;
  }
  // f3: the records Agent::TrajPlanningIteration keeps (AC:193-245). The separating planes are generated inside the fused
  // launch: its duration (hdsm_swarm_record_solve_ms) is booked as comp_time_opt_, comp_time_tasc_ is 0.
  const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - sw->t_round).count();
  const double stamp = (double)(sw->round_idx + 1) * sw->prm.dt * sw->cfg.step_plan;
  for (int k = 0; k < sw->n_local; ++k) {
    const AgentX& x = sw->extra[k];
    hdsm_stats_add(x.stats, HDSM_STAT_SC, x.sc_ms);
    hdsm_stats_add(x.stats, HDSM_STAT_TASC, 0.0);
    hdsm_stats_add(x.stats, HDSM_STAT_OPT, sw->solve_ms);
    hdsm_stats_add(x.stats, HDSM_STAT_TOT, x.sc_ms + x.ref_ms + sw->solve_ms);
    hdsm_stats_add(x.stats, HDSM_STAT_TOT_WALL, wall_ms);
    hdsm_stats_add_state(x.stats, stamp, sw->agents[k].state_curr, 9);
  }
  ++sw->round_idx;
  return HDSM_OK;
}

int hdsm_swarm_record_solve_ms(void* swarm, double milliseconds) {
  Swarm* sw = static_cast<Swarm*>(swarm);
  if (!sw || !(milliseconds >= 0)) return HDSM_ERR_BAD_ARG;
  sw->solve_ms = milliseconds;
  return HDSM_OK;
}

int hdsm_swarm_shutdown(void* swarm, int32_t local_index, const char* dir, int32_t save_stats, char* report, int32_t report_cap) {
  Swarm* sw = static_cast<Swarm*>(swarm);
  if (!sw || local_index < 0 || local_index >= sw->n_local) return HDSM_ERR_BAD_ARG;
  return hdsm_stats_shutdown(sw->extra[local_index].stats, dir, save_stats, report, report_cap);
}

int hdsm_swarm_reference_inputs_n(void* swarm, int32_t pmax, double* path, int32_t* n_path) {
  Swarm* sw = static_cast<Swarm*>(swarm);
  if (!sw || !path || !n_path || pmax < 2) return HDSM_ERR_BAD_ARG;
  const double need = sw->prm.n_hor * sw->cfg.path_vel_max * sw->prm.dt;  // SamplePath never walks further than this
  for (int k = 0; k < sw->n_local; ++k) {
    const std::vector<V3> pl = reference_polyline(sw->agents[k]);
    int n = (int)pl.size();
    if (n > pmax) {
      double len = 0;
      for (int i = 0; i + 1 < pmax; ++i) len += norm(sub(pl[i + 1], pl[i]));
      if (len <= need) return HDSM_ERR_CAPACITY;
      n = pmax;
    }
    n_path[k] = n;
    for (int i = 0; i < pmax; ++i)
      for (int c = 0; c < 3; ++c) path[((size_t)k * pmax + i) * 3 + c] = i < n ? pl[i][c] : pl[n - 1][c];
  }
  return HDSM_OK;
}

int hdsm_swarm_reference_inputs(void* swarm, double* path, int32_t* n_path) {
  return hdsm_swarm_reference_inputs_n(swarm, 3, path, n_path);
}

int hdsm_swarm_vel_cap(void* swarm, double* vel_cap) {
  Swarm* sw = static_cast<Swarm*>(swarm);
  if (!sw || !vel_cap) return HDSM_ERR_BAD_ARG;
  const hdsm_sw::Cfg cc = sw->core_cfg();
  const hdsm_ref_config rc = ref_config(*sw);
  for (int k = 0; k < sw->n_local; ++k) {
    const AgentS& ag = sw->agents[k];
    V3 pl[hdsm_sw::PATH_PTS + 1];
    const int n = hdsm_sw::reference_polyline(ag, pl);
    vel_cap[k] = hdsm_sw::voxel_velocity_cap(cc, rc, hdsm_sw::local_grid_origin(cc, ag), pl, n);
  }
  return HDSM_OK;
}

int hdsm_swarm_set_reference(void* swarm, const double* ref_full, const double* path_vel) {
  Swarm* sw = static_cast<Swarm*>(swarm);
  if (!sw || !ref_full || !path_vel) return HDSM_ERR_BAD_ARG;
  const int N = sw->prm.n_hor;
  const hdsm_sw::Cfg cc = sw->core_cfg();
  for (int k = 0; k < sw->n_local; ++k) {
    AgentS& ag = sw->agents[k];
    ag.n_ref = N + 1;
    for (int i = 0; i <= N; ++i)
      for (int c = 0; c < 6; ++c) ag.traj_ref[i][c] = ref_full[((size_t)k * (N + 1) + i) * 6 + c];
    ag.path_vel = path_vel[k];
    ag.external_ref = 1;
    hdsm_sw::keep_only_free(cc, hdsm_sw::local_grid_origin(cc, ag), ag.path_vel, ag.traj_ref, ag.n_ref);  // AC:1512-1514
  }
  return HDSM_OK;
}

int hdsm_swarm_set_world(void* swarm, const int8_t* occupancy, const int32_t dim[3], const double origin[3]) {
  Swarm* sw = static_cast<Swarm*>(swarm);
  if (!sw || !dim || !origin) return HDSM_ERR_BAD_ARG;
  if (!occupancy) {
    sw->has_world = false;
    sw->world.clear();
    return HDSM_OK;
  }
  if (dim[0] < 1 || dim[1] < 1 || dim[2] < 1) return HDSM_ERR_BAD_ARG;
  for (int ax = 0; ax < 3; ++ax) {  // local grids must register with the world grid
    const double q = origin[ax] / sw->cfg.voxel_size;
    if (std::fabs(q - std::round(q)) > 1e-9) return HDSM_ERR_BAD_ARG;
    sw->wdim[ax] = dim[ax], sw->worigin[ax] = origin[ax];
  }
  sw->world.assign(occupancy, occupancy + (size_t)dim[0] * dim[1] * dim[2]);
  sw->has_world = true;
  return HDSM_OK;
}

int hdsm_swarm_set_paths(void* swarm, const double* paths, const int32_t* n_path, int32_t pmax) {
  Swarm* sw = static_cast<Swarm*>(swarm);
  if (!sw || !paths || !n_path || pmax < 2) return HDSM_ERR_BAD_ARG;
  for (int k = 0; k < sw->n_local; ++k)
    if (n_path[k] < 2 || n_path[k] > pmax || n_path[k] > hdsm_sw::PATH_PTS) return HDSM_ERR_BAD_ARG;
  for (int k = 0; k < sw->n_local; ++k) {
    AgentS& ag = sw->agents[k];
    ag.n_path = n_path[k];
    for (int i = 0; i < n_path[k]; ++i) {
      const double* p = paths + ((size_t)k * pmax + i) * 3;
      ag.path[i] = V3{{p[0], p[1], p[2]}};
    }
  }
  return HDSM_OK;
}

int hdsm_swarm_get_paths(void* swarm, int32_t pmax, double* paths, int32_t* n_path) {
  Swarm* sw = static_cast<Swarm*>(swarm);
  if (!sw || !paths || !n_path || pmax < 2) return HDSM_ERR_BAD_ARG;
  int rc = HDSM_OK;
  for (int k = 0; k < sw->n_local; ++k) {
    const AgentS& ag = sw->agents[k];
    n_path[k] = ag.n_path;
    if (ag.n_path > pmax) rc = HDSM_ERR_CAPACITY;
    for (int i = 0; i < pmax; ++i) {
      const V3& p = ag.path[i < ag.n_path ? i : ag.n_path - 1];
      for (int c = 0; c < 3; ++c) paths[((size_t)k * pmax + i) * 3 + c] = p[c];
    }
  }
  return rc;
}

int hdsm_swarm_route(void* swarm, int32_t* n_failed) {
  Swarm* sw = static_cast<Swarm*>(swarm);
  if (!sw || !sw->has_world) return HDSM_ERR_BAD_ARG;
  const Router router(*sw);
  std::atomic<int> failed{0}, next{0};
  auto worker = [&]() {
    Router::Work wk;
    for (int k = next.fetch_add(1); k < sw->n_local; k = next.fetch_add(1)) {
      AgentS& ag = sw->agents[k];
      std::vector<V3> path;
      bool ok = router.route(ag.start, ag.goal, &path, wk);
      if (ok) {
        // the reference sampling starts ON the path and corridor seeds are taken along it: the first point is the start
        if (norm(sub(path.front(), ag.start)) > 0) path.insert(path.begin(), ag.start);
        if (norm(sub(path.back(), ag.goal)) > 0) path.push_back(ag.goal);
        ok = (int)path.size() <= hdsm_sw::PATH_PTS;
      }
      if (ok) {
        ag.n_path = (int)path.size();
        for (int i = 0; i < ag.n_path; ++i) ag.path[i] = path[i];
      } else {
        ag.n_path = 2, ag.path[0] = ag.start, ag.path[1] = ag.goal;
        failed.fetch_add(1);
      }
    }
  };
  unsigned nt = std::thread::hardware_concurrency();
  nt = nt < 1 ? 1 : (nt > 64 ? 64 : nt);
  if ((int)nt > sw->n_local) nt = (unsigned)(sw->n_local > 0 ? sw->n_local : 1);
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < nt; ++t) pool.emplace_back(worker);
  worker();
  for (auto& th : pool) th.join();
  if (n_failed) *n_failed = failed.load();
  return HDSM_OK;
}

int hdsm_swarm_corridor_errors(void* swarm, int32_t* codes) {
  Swarm* sw = static_cast<Swarm*>(swarm);
  if (!sw) return HDSM_ERR_BAD_ARG;
  int n = 0;
  for (int k = 0; k < sw->n_local; ++k) {
    if (codes) codes[k] = sw->agents[k].corridor_rc;
    n += sw->agents[k].corridor_rc != 0;
  }
  return n;
}

// ---- hooks of the device-resident loop (swarm_kernels.hip): the plain agent states and the configuration of a shard ----
int hdsm_swarm_export_state(void* swarm, void* agents_out, int32_t* n_local, int32_t* n_rob, int32_t* first_id, hdsm_params* prm,
                            hdsm_swarm_config* cfg, const int8_t** world, int32_t wdim[3], double worigin[3]) {
  Swarm* sw = static_cast<Swarm*>(swarm);
  if (!sw) return HDSM_ERR_BAD_ARG;
  if (agents_out && sw->n_local) std::memcpy(agents_out, sw->agents.data(), sizeof(AgentS) * (size_t)sw->n_local);
  if (n_local) *n_local = sw->n_local;
  if (n_rob) *n_rob = sw->n_rob;
  if (first_id) *first_id = sw->first_id;
  if (prm) *prm = sw->prm;
  if (cfg) *cfg = sw->cfg;
  if (world) *world = sw->has_world ? sw->world.data() : nullptr;
  for (int k = 0; k < 3; ++k) {
    if (wdim) wdim[k] = sw->wdim[k];
    if (worigin) worigin[k] = sw->worigin[k];
  }
  return HDSM_OK;
}

int hdsm_swarm_import_state(void* swarm, const void* agents_in, int32_t n_local) {
  Swarm* sw = static_cast<Swarm*>(swarm);
  if (!sw || !agents_in || n_local != sw->n_local) return HDSM_ERR_BAD_ARG;
  std::memcpy(sw->agents.data(), agents_in, sizeof(AgentS) * (size_t)n_local);
  return HDSM_OK;
}

// GenerateSafeCorridor alone (AC:165), for callers that generate the reference elsewhere (row f1 on the device) and want the
// reference's own order: corridor from the PREVIOUS reference, then the new reference. The following hdsm_swarm_prepare of the
// same round does not repeat it.
int hdsm_swarm_prepare_corridor(void* swarm) {
  Swarm* sw = static_cast<Swarm*>(swarm);
  if (!sw) return HDSM_ERR_BAD_ARG;
  const hdsm_sw::Cfg cc = sw->core_cfg();
  for (int k = 0; k < sw->n_local; ++k) {
    const clock_t t0 = clock();
    hdsm_sw::corridor_step(cc, sw->agents[k], sw->work.get(), sw->bits.data());
    sw->extra[k].sc_ms = cpu_ms_since(t0);
  }
  sw->corridor_done = true;
  return HDSM_OK;
}

int hdsm_swarm_state(void* swarm, double* pos, double* dist_goal, int32_t* n_fail) {
  Swarm* sw = static_cast<Swarm*>(swarm);
  if (!sw) return HDSM_ERR_BAD_ARG;
  for (int k = 0; k < sw->n_local; ++k) {
    const AgentS& ag = sw->agents[k];
    const V3 p = {{ag.state_curr[0], ag.state_curr[1], ag.state_curr[2]}};
    if (pos)
      for (int c = 0; c < 3; ++c) pos[3 * k + c] = p[c];
    if (dist_goal) dist_goal[k] = norm(sub(p, ag.goal));
    if (n_fail) n_fail[k] = ag.n_fail;
  }
  return HDSM_OK;
}

int hdsm_swarm_yaw(void* swarm, int32_t yaw_idx, double k_p_yaw, double* yaw_out) {
  Swarm* sw = static_cast<Swarm*>(swarm);
  if (!sw || yaw_idx < 0 || yaw_idx > sw->prm.n_hor) return HDSM_ERR_BAD_ARG;
  const double pi = 3.14159265358979323846;
  for (int k = 0; k < sw->n_local; ++k) {
    const AgentS& ag = sw->agents[k];
    double& yaw = sw->extra[k].yaw;
    if (ag.n_ref > yaw_idx) {  // AC:1028-1030 (traj_ref_curr_ exists from the first reference on)
      double v[3] = {ag.traj_ref[yaw_idx][0] - ag.state_curr[0], ag.traj_ref[yaw_idx][1] - ag.state_curr[1], ag.traj_ref[yaw_idx][2] - ag.state_curr[2]};
      if (v[0] * v[0] + v[1] * v[1] + v[2] * v[2] > 0.1) {  // AC:1033
        const double yaw_ref = std::atan2(v[1], v[0]);        // AC:1035-1040 (the projection on the x-y plane drops v[2])
        double error_ang = yaw_ref - yaw;
        if (error_ang > pi) error_ang = error_ang - 2 * pi;  // AC:1043-1047
        else if (error_ang < -pi) error_ang = error_ang + 2 * pi;
        yaw = yaw + k_p_yaw * error_ang * sw->prm.dt;         // AC:1048
      }
    }
    if (yaw_out) yaw_out[k] = yaw;
  }
  return HDSM_OK;
}

// Test hook (tests/test_host.py; not in include/): the minimum distance of the samples of the increment check (AC:569-585) to `pt`,
// per segment, from the literal 1-cm walk and from the closed form the device's k_commit tries first. ref [n_ref][3].
extern "C" int hdsm_internal_increment_minima(const double* ref, int32_t n_ref, const double* pt, double* literal, double* closed_form) {
  if (!ref || !pt || n_ref < 2 || n_ref > hdsm::MAXH + 1) return HDSM_ERR_BAD_ARG;
  static thread_local AgentS ag;
  ag.n_ref = n_ref;
  for (int i = 0; i < n_ref; ++i)
    for (int k = 0; k < 3; ++k) ag.traj_ref[i][k] = ref[3 * i + k];
  const hdsm_sw::V3 p = {{pt[0], pt[1], pt[2]}};
  for (int seg = 0; seg + 1 < n_ref; ++seg) {
    if (literal) literal[seg] = hdsm_sw::increment_segment_min(ag, seg, p);
    if (closed_form) closed_form[seg] = hdsm_sw::increment_segment_min_closed_form(ag, seg, p);
  }
  return HDSM_OK;
}

int hdsm_swarm_view(void* swarm, int32_t k, double* traj_curr, int32_t* n_traj, double* traj_ref, int32_t* n_ref, double* path, int32_t pmax,
                    int32_t* n_path, int32_t* n_poly, int32_t* poly_rows, double* poly_A, double* poly_b, double* poly_seeds, double pos[3]) {
  Swarm* sw = static_cast<Swarm*>(swarm);
  if (!sw || k < 0 || k >= sw->n_local) return HDSM_ERR_BAD_ARG;
  const AgentS& ag = sw->agents[k];
  const int N = sw->prm.n_hor, P = sw->prm.poly_hor, RS = sw->prm.max_rows_static;
  const int nt = ag.has_traj ? N + 1 : 0, nr = ag.n_ref < N + 1 ? ag.n_ref : N + 1;
  if (n_traj) *n_traj = nt;
  if (traj_curr)
    for (int i = 0; i < nt; ++i)
      for (int c = 0; c < 3; ++c) traj_curr[3 * i + c] = ag.traj_curr[i][c];
  if (n_ref) *n_ref = nr;
  if (traj_ref)
    for (int i = 0; i < nr; ++i)
      for (int c = 0; c < 3; ++c) traj_ref[3 * i + c] = ag.traj_ref[i][c];
  const int np = ag.n_path < pmax ? ag.n_path : pmax;
  if (n_path) *n_path = np;
  if (path)
    for (int i = 0; i < np; ++i)
      for (int c = 0; c < 3; ++c) path[3 * i + c] = ag.path[i][c];
  const int npl = ag.n_poly < P ? ag.n_poly : P;
  if (n_poly) *n_poly = npl;
  for (int j = 0; j < npl; ++j) {
    const hdsm_sw::Poly& pl = ag.polys[j];
    const int rows = pl.rows < RS ? pl.rows : RS;
    if (poly_rows) poly_rows[j] = rows;
    for (int r = 0; r < rows; ++r) {
      if (poly_A)
        for (int c = 0; c < 3; ++c) poly_A[((size_t)j * RS + r) * 3 + c] = pl.A[r][c];
      if (poly_b) poly_b[(size_t)j * RS + r] = pl.b[r];
    }
    if (poly_seeds)
      for (int c = 0; c < 3; ++c) poly_seeds[3 * j + c] = pl.seed[c];
  }
  if (pos)
    for (int c = 0; c < 3; ++c) pos[c] = ag.state_curr[c];
  return HDSM_OK;
}

}  // extern "C"
