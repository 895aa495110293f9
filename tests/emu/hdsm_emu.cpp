// hdsm_emu.cpp — CPU build of the kernel LOGIC (multi_agent_pkgs_amd/csrc/hdsm_core.h with HDSM_EMU).
//
// TEST INFRASTRUCTURE ONLY. It lets the CPU test-suite (-m "not gpu") exercise the exact statements the
// HIP kernel executes — active-set updates, lazy row staging, branch-and-bound state machine — against
// the oracle without a GPU. It is built into tests/emu/libhdsm_emu.so, is never linked into libhdsm.so and
// is not reachable from the product API (which fails with HDSM_ERR_NO_DEVICE when there is no GPU).
#define HDSM_EMU 1
#include "../../multi_agent_pkgs_amd/csrc/hdsm_consts.h"
#include "../../multi_agent_pkgs_amd/csrc/hdsm_core.h"
#include "../../multi_agent_pkgs_amd/csrc/hdsm_level1.h"

#include <cstring>
#include <memory>
#include <vector>

namespace {
template <int NV, int CMAX>
void run_all(const hdsm::Consts& c, hdsm::Args& a) {
  using Sol = hdsm::Solver<NV, CMAX>;
  a.scratch_stride = (int64_t)Sol::SNAP_STRIDE * hdsm::MAXH;
  std::vector<double> scratch((size_t)a.scratch_stride);
  auto shm = std::make_unique<typename Sol::S>();
  for (int k = 0; k < a.n_inst; ++k) {
    std::memset(shm.get(), 0, sizeof(typename Sol::S));
    hdsm::Args ak = a;
    ak.scratch = scratch.data() - (int64_t)k * a.scratch_stride;  // solve_instance adds inst*stride
    Sol::solve_instance(*shm, c, ak, k);
  }
}
}  // namespace

extern "C" int emu_replan(const hdsm_params* prm, int32_t n_inst, int32_t n_rob, const int32_t* agent_id,
                          const double* state_curr, const double* traj_ref, const int32_t* n_poly,
                          const int32_t* n_rows_static, const double* A_static, const double* b_static,
                          const double* plans_all, const uint8_t* has_plan, double* traj_out,
                          double* ctrl_out, uint8_t* poly_used, int32_t* status, double* obj,
                          int32_t* qp_iters, int32_t* nodes, int32_t* sweeps, int32_t* cand, int32_t cmax) {
  auto c = std::make_unique<hdsm::Consts>();
  const char* err = nullptr;
  int rc = hdsm::build_consts(prm, c.get(), &err);
  if (rc) return rc;
  hdsm::Args a{};
  a.n_inst = n_inst, a.n_rob = n_rob, a.agent_id = agent_id, a.state = state_curr, a.ref = traj_ref;
  a.n_poly = n_poly, a.n_rows = n_rows_static, a.A = A_static, a.b = b_static, a.plans = plans_all;
  a.has_plan = has_plan, a.traj = traj_out, a.ctrl = ctrl_out, a.used = poly_used, a.status = status;
  a.obj = obj, a.st_iters = qp_iters, a.st_nodes = nodes, a.st_sweeps = sweeps, a.st_cand = cand;
  const bool small = c->n <= 30;
  if (cmax > 0 && cmax <= 16) {  // tiny staging capacity: exercises the overflow path in tests
    if (small) run_all<30, 16>(*c, a); else run_all<48, 16>(*c, a);
  } else {
    if (small) run_all<30, 1536>(*c, a); else run_all<48, 1024>(*c, a);
  }
  return 0;
}

// level 1 (fully formed per-step polyhedra) through the same host-side split the product uses
extern "C" int emu_solve(const hdsm_params* prm, int32_t n_inst, int32_t r_max, const double* state_curr,
                         const double* traj_ref, const int32_t* n_poly, const int32_t* n_rows, const double* A,
                         const double* b, double* traj_out, double* ctrl_out, uint8_t* poly_used, int32_t* status,
                         double* obj) {
  auto c = std::make_unique<hdsm::Consts>();
  const char* err = nullptr;
  int rc = hdsm::build_consts(prm, c.get(), &err);
  if (rc) return rc;
  hdsm::Level1Split sp;
  rc = hdsm::level1_split(*prm, n_inst, r_max, n_poly, n_rows, A, b, &sp, &err);
  if (rc) return rc;
  std::vector<int32_t> ids(n_inst, -1);
  uint8_t zero = 0;
  double dummy_plans[16] = {0};
  hdsm::Args a{};
  a.n_inst = n_inst, a.n_rob = 0, a.agent_id = ids.data(), a.state = state_curr, a.ref = traj_ref;
  a.n_poly = sp.n_poly.data(), a.n_rows = sp.n_rows_static.data(), a.A = sp.A_static.data(), a.b = sp.b_static.data();
  a.plans = dummy_plans, a.has_plan = &zero, a.traj = traj_out, a.ctrl = ctrl_out, a.used = poly_used;
  a.status = status, a.obj = obj, a.l1_rows = sp.common.data(), a.l1_nrows = sp.n_common.data(), a.l1_rmax = sp.rc_max;
  if (c->n <= 30) run_all<30, 1536>(*c, a); else run_all<48, 1024>(*c, a);
  return 0;
}
