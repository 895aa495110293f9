// sharded_loop.cpp — a C++ host (the reference's language) running ONE RANK of the sharded closed loop through the C ABI:
// agents [first, first + per) of the circular exchange live on this rank; every round
//     host:   hdsm_swarm_prepare (corridor)                                   AC:165
//     device: hdsm_reference_device -> hdsm_replan_device                     AC:168-174 (f1 + the hot path), one stream
//     host:   hdsm_swarm_commit (fallback, increment check, state advance)    AC:182, 233-238, 1000-1019
//     device: hdsm_publish_device -> hdsm_exchange_device                     ONE RCCL all-gather instead of AC:610-677
// The all-gathered plans and flags stay on the device for the next round's reference + replan; they are copied back only
// because the HOST corridor/commit code of this example wants them (row f2/f3 territory).
// usage: sharded_loop <rank> <world> <id_file> [n_agents = 64] [rounds = 60]
//   rank 0 writes the RCCL unique id to <id_file>, the other ranks wait for it (any launcher works: mpirun, a shell loop).
//   HIP_VISIBLE_DEVICES selects the GPU of a rank; with one GPU, world = 1.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <thread>
#include <vector>

#include "../include/hdsm.h"
#include "../include/hdsm_swarm.h"

#define CK(x)                                                                     \
  do {                                                                            \
    if ((x) != 0) {                                                               \
      std::fprintf(stderr, "%s failed: %s\n", #x, hdsm_last_error());             \
      return 3;                                                                   \
    }                                                                             \
  } while (0)
#define HK(x)                                                                     \
  do {                                                                            \
    if ((x) != hipSuccess) {                                                      \
      std::fprintf(stderr, "%s failed\n", #x);                                    \
      return 4;                                                                   \
    }                                                                             \
  } while (0)

template <class T>
T* dalloc(size_t n) {
  void* p = nullptr;
  if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) std::exit(5);
  (void)hipMemset(p, 0, (n ? n : 1) * sizeof(T));
  return static_cast<T*>(p);
}

int main(int argc, char** argv) {
  if (argc < 4) return std::fprintf(stderr, "usage: sharded_loop <rank> <world> <id_file> [n_agents] [rounds]\n"), 1;
  const int rank = std::atoi(argv[1]), world = std::atoi(argv[2]);
  const char* id_file = argv[3];
  const int n = argc > 4 ? std::atoi(argv[4]) : 64, rounds = argc > 5 ? std::atoi(argv[5]) : 60;
  const int N = 10, per = (n + world - 1) / world, first = rank * per, n_local = std::max(0, std::min(per, n - first));
  hdsm_params prm;
  hdsm_default_params(&prm, N);
  prm.max_rows_static = 18;
  hdsm_swarm_config cfg;
  hdsm_swarm_default_config(&cfg);
  hdsm_ref_config rcfg = {cfg.path_vel_min, cfg.path_vel_max, cfg.sens_dist, cfg.sens_pot, cfg.sens_other_agents, cfg.path_vel_dec};
  const int P = prm.poly_hor, RS = prm.max_rows_static, REC = (N + 1) * 9;

  const double pi = std::acos(-1.0), R = std::fmax(22.0, n / (2 * pi));
  std::vector<double> starts(3 * n), goals(3 * n);
  for (int k = 0; k < n; ++k)
    starts[3 * k] = 18.0 + R * std::cos(2 * pi * k / n), starts[3 * k + 1] = 15.0 + R * std::sin(2 * pi * k / n), starts[3 * k + 2] = 1.5;
  for (int k = 0; k < n; ++k)
    for (int c = 0; c < 3; ++c) goals[3 * k + c] = starts[3 * ((k + n / 2) % n) + c];

  void *solver = nullptr, *swarm = nullptr, *comm = nullptr;
  CK(hdsm_create(&prm, per, world * per, 0, &solver));
  CK(hdsm_swarm_create(&prm, &cfg, n, first, n_local, starts.data() + 3 * first, goals.data() + 3 * first, &swarm));
  uint8_t uid[HDSM_COMM_ID_BYTES];
  if (rank == 0) {
    CK(hdsm_comm_unique_id(uid));
    std::ofstream(std::string(id_file) + ".tmp", std::ios::binary).write(reinterpret_cast<char*>(uid), sizeof uid);
    std::rename((std::string(id_file) + ".tmp").c_str(), id_file);
  } else {
    for (int t = 0; t < 600; ++t) {
      std::ifstream f(id_file, std::ios::binary);
      if (f.read(reinterpret_cast<char*>(uid), sizeof uid)) break;
      std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
  }
  CK(hdsm_comm_create(solver, uid, rank, world, &comm));
  int32_t crank = -1, cworld = -1;
  CK(hdsm_comm_info(comm, &crank, &cworld));
  if (crank != rank || cworld != world) return std::fprintf(stderr, "communicator reports rank %d of %d\n", crank, cworld), 6;

  hipStream_t st;
  HK(hipStreamCreate(&st));
  const size_t L = (size_t)per, G = (size_t)world * per;
  // host mirrors
  std::vector<int32_t> id(L), n_poly(L), n_rows(L * P), status(L), n_path(L);
  std::vector<double> state(9 * L), ref(6 * N * L), A(L * P * RS * 3), b(L * P * RS), path(L * 3 * 3), ref_full(L * (N + 1) * 6), pv(L);
  std::vector<double> plans(G * REC, 0.0), traj(L * REC), ctrl(L * N * 3), plans_local(L * REC);
  std::vector<uint8_t> has(G, 0), used(L * P), has_local(L);
  // device buffers
  int32_t *d_id = dalloc<int32_t>(L), *d_npoly = dalloc<int32_t>(L), *d_nrows = dalloc<int32_t>(L * P), *d_status = dalloc<int32_t>(L),
          *d_npath = dalloc<int32_t>(L);
  double *d_state = dalloc<double>(9 * L), *d_ref = dalloc<double>(6 * N * L), *d_A = dalloc<double>(L * P * RS * 3),
         *d_b = dalloc<double>(L * P * RS), *d_path = dalloc<double>(L * 9), *d_full = dalloc<double>(L * (N + 1) * 6),
         *d_pv = dalloc<double>(L), *d_traj = dalloc<double>(L * REC), *d_ctrl = dalloc<double>(L * N * 3), *d_obj = dalloc<double>(L),
         *d_plans = dalloc<double>(G * REC), *d_local = dalloc<double>(L * REC), *d_commit = dalloc<double>(L * REC);
  uint8_t *d_has = dalloc<uint8_t>(G), *d_used = dalloc<uint8_t>(L * P), *d_has_local = dalloc<uint8_t>(L);

  int failures = 0;
  double checksum = 0;
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < rounds; ++r) {
    // f1 on the device: the polyline comes from the host state, the speed term reads the all-gathered plans in HBM
    CK(hdsm_swarm_reference_inputs(swarm, path.data(), n_path.data()));
    HK(hipMemcpyAsync(d_path, path.data(), L * 9 * 8, hipMemcpyHostToDevice, st));
    HK(hipMemcpyAsync(d_npath, n_path.data(), L * 4, hipMemcpyHostToDevice, st));
    for (int k = 0; k < n_local; ++k) id[k] = first + k;
    HK(hipMemcpyAsync(d_id, id.data(), L * 4, hipMemcpyHostToDevice, st));
    if (n_local) CK(hdsm_reference_device(solver, &rcfg, n_local, world * per, d_id, d_path, d_npath, 3, nullptr, d_plans, d_has, d_full, nullptr, d_pv, st));
    HK(hipMemcpyAsync(ref_full.data(), d_full, L * (N + 1) * 6 * 8, hipMemcpyDeviceToHost, st));
    HK(hipMemcpyAsync(pv.data(), d_pv, L * 8, hipMemcpyDeviceToHost, st));
    HK(hipStreamSynchronize(st));
    CK(hdsm_swarm_set_reference(swarm, ref_full.data(), pv.data()));
    CK(hdsm_swarm_prepare(swarm, plans.data(), has.data(), id.data(), state.data(), ref.data(), n_poly.data(), n_rows.data(), A.data(), b.data()));
    HK(hipMemcpyAsync(d_state, state.data(), L * 9 * 8, hipMemcpyHostToDevice, st));
    HK(hipMemcpyAsync(d_ref, ref.data(), L * 6 * N * 8, hipMemcpyHostToDevice, st));
    HK(hipMemcpyAsync(d_npoly, n_poly.data(), L * 4, hipMemcpyHostToDevice, st));
    HK(hipMemcpyAsync(d_nrows, n_rows.data(), L * P * 4, hipMemcpyHostToDevice, st));
    HK(hipMemcpyAsync(d_A, A.data(), L * P * RS * 3 * 8, hipMemcpyHostToDevice, st));
    HK(hipMemcpyAsync(d_b, b.data(), L * P * RS * 8, hipMemcpyHostToDevice, st));
    if (n_local) CK(hdsm_replan_device(solver, n_local, world * per, d_id, d_state, d_ref, d_npoly, d_nrows, d_A, d_b, d_plans, d_has, d_traj, d_ctrl,
                                       d_used, d_status, d_obj, st));
    HK(hipMemcpyAsync(traj.data(), d_traj, L * REC * 8, hipMemcpyDeviceToHost, st));
    HK(hipMemcpyAsync(ctrl.data(), d_ctrl, L * N * 3 * 8, hipMemcpyDeviceToHost, st));
    HK(hipMemcpyAsync(used.data(), d_used, L * P, hipMemcpyDeviceToHost, st));
    HK(hipMemcpyAsync(status.data(), d_status, L * 4, hipMemcpyDeviceToHost, st));
    HK(hipStreamSynchronize(st));
    for (int k = 0; k < n_local; ++k) failures += status[k] == HDSM_NO_SOLUTION;
    CK(hdsm_swarm_commit(swarm, traj.data(), ctrl.data(), used.data(), status.data(), plans_local.data(), has_local.data()));
    // publish + ONE all-gather, on the device
    HK(hipMemcpyAsync(d_commit, plans_local.data(), L * REC * 8, hipMemcpyHostToDevice, st));
    HK(hipMemcpyAsync(d_has_local, has_local.data(), L, hipMemcpyHostToDevice, st));
    CK(hdsm_publish_device(solver, per, n_local, d_commit, d_has_local, d_local, st));
    CK(hdsm_exchange_device(comm, per, d_local, d_plans, d_has, st));
    HK(hipMemcpyAsync(plans.data(), d_plans, G * REC * 8, hipMemcpyDeviceToHost, st));
    HK(hipMemcpyAsync(has.data(), d_has, G, hipMemcpyDeviceToHost, st));
    HK(hipStreamSynchronize(st));
    for (size_t k = 0; k < G; ++k)
      if (!has[k]) plans[k * REC] = 0.0;  // the sentinel is for the wire only
  }
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  int with_plan = 0;
  for (size_t k = 0; k < G; ++k) {
    with_plan += has[k];
    if (has[k])
      for (int e = 0; e < 3; ++e) checksum += plans[k * REC + 9 + e];
  }
  std::printf("sharded_loop rank %d/%d: agents [%d, %d) of %d, %d rounds, %.3f ms per round, instances without solution %d, "
              "agents with a plan %d, checksum of all first positions %.9f\n",
              rank, world, first, first + n_local, n, rounds, ms / rounds, failures, with_plan, checksum);
  hdsm_comm_destroy(comm);
  hdsm_swarm_destroy(swarm);
  hdsm_destroy(solver);
  return with_plan == n ? 0 : 7;
}
