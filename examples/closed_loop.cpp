// closed_loop.cpp — the C ABI driven from C++ (the reference's language), no Python involved: a swarm of agents on the
// circular exchange of multi_agent_planner_circle.launch.py replans in lock-step, host planner state from
// include/hdsm_swarm.h (the part of Agent::TrajPlanningIteration around the solve), solve on the GPU through hdsm_replan.
// usage: closed_loop [n_agents = 16] [rounds = 120]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/hdsm.h"
#include "../include/hdsm_swarm.h"

int main(int argc, char** argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 16, rounds = argc > 2 ? std::atoi(argv[2]) : 120;
  const int N = 10;
  hdsm_params prm;
  hdsm_default_params(&prm, N);  // agent_agile_config.yaml values
  prm.max_rows_static = 18;
  hdsm_swarm_config cfg;
  hdsm_swarm_default_config(&cfg);
  const int P = prm.poly_hor, RS = prm.max_rows_static;

  const double pi = std::acos(-1.0), R = std::fmax(22.0, n / (2 * pi));
  std::vector<double> starts(3 * n), goals(3 * n);
  for (int k = 0; k < n; ++k) {
    starts[3 * k] = 18.0 + R * std::cos(2 * pi * k / n), starts[3 * k + 1] = 15.0 + R * std::sin(2 * pi * k / n), starts[3 * k + 2] = 1.5;
  }
  for (int k = 0; k < n; ++k)
    for (int c = 0; c < 3; ++c) goals[3 * k + c] = starts[3 * ((k + n / 2) % n) + c];

  void *solver = nullptr, *swarm = nullptr;
  if (hdsm_create(&prm, n, n, 0, &solver) != HDSM_OK) {
    std::fprintf(stderr, "hdsm_create: %s\n", hdsm_last_error());
    return 2;
  }
  if (hdsm_swarm_create(&prm, &cfg, n, 0, n, starts.data(), goals.data(), &swarm) != HDSM_OK) return 2;

  std::vector<int32_t> id(n), n_poly(n), n_rows(n * P), status(n);
  std::vector<double> state(9 * n), ref(6 * N * n), A((size_t)n * P * RS * 3), b((size_t)n * P * RS);
  std::vector<double> plans((size_t)n * (N + 1) * 9, 0.0), traj((size_t)n * (N + 1) * 9), ctrl((size_t)n * N * 3), obj(n);
  std::vector<uint8_t> has(n, 0), used(n * P), has_new(n);
  std::vector<double> plans_new(plans.size()), pos(3 * n), dist(n);
  std::vector<int32_t> nfail(n);

  double solve_ms = 0, min_sep = 1e9;
  int failures = 0;
  for (int r = 0; r < rounds; ++r) {
    if (hdsm_swarm_prepare(swarm, plans.data(), has.data(), id.data(), state.data(), ref.data(), n_poly.data(), n_rows.data(),
                           A.data(), b.data()) != HDSM_OK)
      return 3;
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = hdsm_replan(solver, n, n, id.data(), state.data(), ref.data(), n_poly.data(), n_rows.data(), A.data(), b.data(),
                               plans.data(), has.data(), traj.data(), ctrl.data(), used.data(), status.data(), obj.data());
    solve_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rc != HDSM_OK) {
      std::fprintf(stderr, "hdsm_replan: %s\n", hdsm_last_error());
      return 3;
    }
    for (int k = 0; k < n; ++k) failures += status[k] == HDSM_NO_SOLUTION;
    if (hdsm_swarm_commit(swarm, traj.data(), ctrl.data(), used.data(), status.data(), plans_new.data(), has_new.data()) != HDSM_OK)
      return 3;
    plans.swap(plans_new), has.swap(has_new);
    hdsm_swarm_state(swarm, pos.data(), dist.data(), nfail.data());
    for (int i = 0; i < n; ++i)
      for (int j = i + 1; j < n; ++j) {
        const double dx = pos[3 * i] - pos[3 * j], dy = pos[3 * i + 1] - pos[3 * j + 1], dz = pos[3 * i + 2] - pos[3 * j + 2];
        min_sep = std::fmin(min_sep, std::sqrt(dx * dx + dy * dy + dz * dz));
      }
  }
  double far = 0;
  for (int k = 0; k < n; ++k) far = std::fmax(far, dist[k]);
  std::printf("closed_loop: %d agents, %d rounds, %.3f ms per replan round through hdsm_replan (host buffers), "
              "instances without solution %d, closest approach %.3f m, farthest agent %.2f m from its goal\n",
              n, rounds, solve_ms / rounds, failures, min_sep, far);
  hdsm_swarm_destroy(swarm);
  hdsm_destroy(solver);
  return 0;
}
