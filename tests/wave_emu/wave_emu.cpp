// wave_emu.cpp — runs the DEVICE source of the solver kernel on the CPU (TEST INFRASTRUCTURE ONLY).
//
// The code that actually runs on the GPU — the register-resident, wave-level iteration of hdsm_wave_gi.h / hdsm_wave_gib.h with
// its DPP butterflies and scans, lane-split rows, Householder add / drop, warm start, certificates, conflict learning, sweeps on
// the packed positions — is compiled by g++ against tests/wave_emu/shim/hip/hip_runtime.h: one workgroup = one wavefront = 64 fibers in
// lockstep — one wavefront (the product's 64-thread launch) or four (its default: the helper waves share the sweeps, the set-up,
// the leaf test and the staged-row scans) — one instance after the other. Never linked into libhdsm.so, not a fallback.
#include <ucontext.h>

#include <memory>
#include <vector>

#include "../../multi_agent_pkgs_amd/csrc/hdsm_consts.h"
#include "../../multi_agent_pkgs_amd/csrc/hdsm_core.h"
#include "../../multi_agent_pkgs_amd/csrc/hdsm_level1.h"

namespace wemu {
namespace {
Runtime g_rt;
ucontext_t g_main;
ucontext_t g_ctx[MAXT];
bool g_done[MAXT];
char g_error[256];
bool g_failed = false;
constexpr size_t STACK = 384 * 1024;
std::vector<char> g_stacks;
void (*g_body)(void*) = nullptr;
void* g_arg = nullptr;

void trampoline() {
  g_body(g_arg);
  Runtime& r = g_rt;
  Wave& w = r.wave[r.cur / W];
  g_done[r.cur] = true;
  --w.nlive, --r.b_nlive;
  if (w.nlive > 0 && w.arrived == w.nlive) fail("a lane left the kernel while the others of its wavefront wait at a cross-lane operation");
  if (r.b_nlive > 0 && r.b_arrived == r.b_nlive) fail("a thread left the kernel while the others wait at __syncthreads()");
  swapcontext(&g_ctx[r.cur], &g_main);
}
}  // namespace

Runtime& rt() { return g_rt; }
void yield() { swapcontext(&g_ctx[g_rt.cur], &g_main); }
void fail(const char* what) {
  snprintf(g_error, sizeof g_error, "%s (thread %d, lockstep point %ld)", what, g_rt.cur, g_rt.ops);
  g_failed = true;
  for (;;) swapcontext(&g_ctx[g_rt.cur], &g_main);  // never resumes: the scheduler stops on g_failed
}

// one workgroup of `nthreads` (64 or 256) threads executing body(arg); false + message on a lockstep violation
bool run_block(void (*body)(void*), void* arg, int block_index, int nthreads) {
  Runtime& r = g_rt;
  r = Runtime{};
  r.nthreads = nthreads;
  r.block = {(unsigned)nthreads, 1, 1};
  r.bidx = {(unsigned)block_index, 0, 0};
  r.b_nlive = nthreads;
  for (int w = 0; w < nthreads / W; ++w) r.wave[w].nlive = W;
  g_body = body, g_arg = arg, g_failed = false;
  if (g_stacks.size() != STACK * MAXT) g_stacks.assign(STACK * MAXT, 0);
  for (int l = 0; l < nthreads; ++l) {
    r.tid[l] = {(unsigned)l, 0, 0};
    g_done[l] = false;
    getcontext(&g_ctx[l]);
    g_ctx[l].uc_stack.ss_sp = g_stacks.data() + STACK * l;
    g_ctx[l].uc_stack.ss_size = STACK;
    g_ctx[l].uc_link = &g_main;
    makecontext(&g_ctx[l], trampoline, 0);
  }
  long idle_rounds = 0;
  while (r.b_nlive > 0 && !g_failed) {
    const long ops = r.ops;
    const int live = r.b_nlive;
    // (WEMU_ORDER=reverse: the wavefronts of the workgroup take their turns in the opposite order — what runs between two barriers must
    // not depend on which wave gets there first; a result that changes with the order is a data race between wavefronts)
    static const bool reverse = getenv("WEMU_ORDER") != nullptr && getenv("WEMU_ORDER")[0] == 'r';
    for (int l0 = 0; l0 < nthreads && !g_failed; ++l0) {
      const int nw = nthreads / W, l = reverse ? ((nw - 1 - l0 / W) * W + l0 % W) : l0;
      if (g_done[l]) continue;
      r.cur = l;
      swapcontext(&g_main, &g_ctx[l]);
    }
    if (r.ops == ops && r.b_nlive == live) {
      if (++idle_rounds > 4) {
        snprintf(g_error, sizeof g_error, "deadlock: %d of %d threads wait at __syncthreads(), wavefront 0 has %d of %d lanes at a cross-lane operation",
                 r.b_arrived, r.b_nlive, r.wave[0].arrived, r.wave[0].nlive);
        g_failed = true;
      }
    } else {
      idle_rounds = 0;
    }
  }
  return !g_failed;
}
const char* last_error() { return g_error; }
}  // namespace wemu

namespace {
template <int NV, int CMAX, bool SMALL = false>
struct Job {
  typename hdsm::Solver<NV, CMAX, SMALL>::S* s;
  const hdsm::Consts* c;
  hdsm::Args a;
  int inst, out, sub;
  int* wg_slot = nullptr;  // pass 2: the scratch slot the (one) emulated workgroup holds, kept across items
};
template <int NV, int CMAX, bool SMALL = false>
void body(void* p) {
  auto* j = static_cast<Job<NV, CMAX, SMALL>*>(p);
  int own_slot = 0;
  hdsm::Solver<NV, CMAX, SMALL>::solve_instance(*j->s, *j->c, j->a, j->inst, j->out, j->sub, -1, j->wg_slot ? *j->wg_slot : own_slot);
}
template <int NV, int CMAX, bool SMALL = false>
int run_all(const hdsm::Consts& c, hdsm::Args& a, int nthreads) {
  using Sol = hdsm::Solver<NV, CMAX, SMALL>;
  a.scratch_stride = (int64_t)Sol::SNAP_STRIDE * hdsm::MAXH;
  std::vector<double> scratch((size_t)a.scratch_stride);
  auto shm = std::make_unique<typename Sol::S>();
  if (a.warm_out == nullptr) a.warm_out = a.warm;
  // split_budget > 0: the three steps of hdsm_api.hip's split launch, one workgroup after the other — pass 1 with the node
  // budget (an instance that exceeds it writes a hand-over record and queues one item per open child of its open levels), then
  // every queued item (own outputs, the instance's shared incumbent word and node pool, the snapshots of pass 1 read from its
  // scratch), then the merge (hdsm::split_merge, the body of k_split_merge)
  const int rec_cap = a.n_inst * 64 < 256 ? 256 : a.n_inst * 64, rows_cap = CMAX, items_cap = rec_cap * 8;
  std::vector<int32_t> split_info, sub_status, sub_stats, sub_warm, rec_count(8 + rec_cap, 0), items, rec_src, node_pool;
  std::vector<unsigned long long> inc_bits;
  std::vector<double> sub_traj, sub_ctrl, sub_obj, rec_cand, all_scratch;
  std::vector<long long> rec_mw;
  std::vector<hdsm::SplitRec> recs;
  std::vector<uint8_t> sub_used;
  double* pass1_scratch = scratch.data();
  if (a.split_budget > 0) {
    split_info.assign((size_t)2 * a.n_inst, 0);
    a.split_info = split_info.data();
    recs.resize(rec_cap), rec_cand.assign((size_t)rec_cap * rows_cap * 4, 0.0), rec_mw.assign((size_t)rec_cap * rows_cap, 0), rec_src.assign((size_t)rec_cap * rows_cap, 0);
    items.assign(items_cap, 0), inc_bits.assign(a.n_inst, 0x7ff0000000000000ull), node_pool.assign(a.n_inst, 0);
    a.recs = recs.data(), a.rec_cand = rec_cand.data(), a.rec_mw = rec_mw.data(), a.rec_src = rec_src.data(), a.rec_count = rec_count.data(), a.items = items.data();
    a.rec_cap = rec_cap, a.rows_cap = rows_cap, a.items_cap = items_cap, a.inc_bits = inc_bits.data(), a.node_pool = node_pool.data();
    sub_status.assign((size_t)items_cap, hdsm::ST_NO_SOLUTION), a.item_status = sub_status.data();
    const int left_nodes = c.max_nodes - a.split_budget > 64 ? c.max_nodes - a.split_budget : 64;
    a.nodes_pool0 = left_nodes, a.node_cap = left_nodes / 128 > 0 ? left_nodes / 128 : 1;
    all_scratch.assign((size_t)a.scratch_stride * a.n_inst, 0.0);  // (pass 2 reads the snapshots of pass 1: every instance keeps its own)
    pass1_scratch = all_scratch.data();
  }
  for (int k = 0; k < a.n_inst; ++k) {
    memset(static_cast<void*>(shm.get()), 0, sizeof(typename Sol::S));
    Job<NV, CMAX, SMALL> job{shm.get(), &c, a, k, k, -1};
    job.a.scratch = a.split_budget > 0 ? pass1_scratch : scratch.data() - (int64_t)k * a.scratch_stride;  // solve_instance adds out * stride
    if (!wemu::run_block(body<NV, CMAX, SMALL>, &job, k, nthreads)) return -100;
    if (getenv("WEMU_OPS")) fprintf(stderr, "instance %d: %ld lockstep points (barrier %ld, readlane %ld, ballot %ld, dpp %ld, permlane %ld, wsync %ld), %d active-set operations\n", k, wemu::rt().ops, wemu::rt().by_kind[1], wemu::rt().by_kind[2], wemu::rt().by_kind[3], wemu::rt().by_kind[6], wemu::rt().by_kind[7] + wemu::rt().by_kind[8], wemu::rt().by_kind[9], a.st_iters ? a.st_iters[k] : -1);
  }
  if (a.split_budget > 0) {
    const size_t G = (size_t)items_cap, N = (size_t)c.N;
    hdsm::Args b = a;
    sub_stats.assign(8 * G, 0), sub_warm.assign((hdsm::MAXNV + 2) * G, 0);
    sub_traj.assign(G * (N + 1) * 9, 0.0), sub_ctrl.assign(G * N * 3, 0.0), sub_obj.assign(G, 0.0), sub_used.assign(G * c.P, 0);
    b.item_mode = 1, b.order = nullptr;
    // an item whose subtree outgrows the budget hands over again (hdsm_api.hip: HDSM_ITEM_BUDGET, default 32; here the budget of pass 1,
    // so that the tests reach it): its scratch stays with its record, the emulated workgroup moves to a fresh slot of the pool
    b.split_budget = getenv("WEMU_ITEM_BUDGET") ? atoi(getenv("WEMU_ITEM_BUDGET")) : a.split_budget;
    b.split_min = getenv("WEMU_ITEM_MIN") ? atoi(getenv("WEMU_ITEM_MIN")) : 2;  // (one workgroup after the other: the queue is empty whenever the last queued item runs)
    const int pool_cap = 1 + rec_cap;
    std::vector<double> pool((size_t)a.scratch_stride * pool_cap, 0.0);
    b.scratch = pool.data(), b.pool_cap = pool_cap;
    b.traj = sub_traj.data(), b.ctrl = sub_ctrl.data(), b.used = sub_used.data(), b.status = sub_status.data(), b.obj = sub_obj.data();
    b.warm_out = sub_warm.data();
    b.st_iters = sub_stats.data(), b.st_nodes = sub_stats.data() + G, b.st_sweeps = sub_stats.data() + 2 * G, b.st_cand = sub_stats.data() + 3 * G;
    b.st_sph = nullptr, b.st_pairs = nullptr, b.st_flags = reinterpret_cast<uint32_t*>(sub_stats.data() + 6 * G), b.st_key = nullptr;
    if (getenv("WEMU_SPLIT_TRACE")) fprintf(stderr, "split: %d records, %d items queued by pass 1\n", rec_count[0], rec_count[1]);
    int next_free = 0;
    for (int g = 0; g < (rec_count[1] < items_cap ? rec_count[1] : items_cap); ++g) {  // (the queue grows while items hand over again)
      const int item = items[g], inst = recs[item >> 8].inst;
      memset(static_cast<void*>(shm.get()), 0, sizeof(typename Sol::S));
      int wg_slot = next_free;  // (run_block's slot table: a slot left to a record stays busy)
      Job<NV, CMAX, SMALL> job{shm.get(), &c, b, inst, g, item, &wg_slot};
      if (!wemu::run_block(body<NV, CMAX, SMALL>, &job, 0, nthreads)) return -100;
      if (wg_slot < 0) ++next_free;
    }
    if (getenv("WEMU_SPLIT_TRACE")) fprintf(stderr, "split: %d records, %d items in the end\n", rec_count[0], rec_count[1]);
    for (int inst = 0; inst < a.n_inst; ++inst) hdsm::split_merge(c.N, c.P, a, b, inst, 0, 1);
  }
  return 0;
}
}  // namespace

extern "C" const char* wave_last_error(void) { return wemu::last_error(); }

// Level-2 replan through the device source. `warm` = the handle's warm-start store, [(MAXNV + 2) * n_inst] int32, in/out (zeros:
// cold; pass the same array again to continue like consecutive launches on one handle); null = warm start off.
// `bounds_min`: swarms of at least this many agents get the sphere prefilter records, as hdsm_api.hip's launch() does.
// `cmax` in 1..16: a build of the kernel with room for only 16 staged rows (staging-overflow tests); 256: the small LDS layout of the
// four-per-CU kernel; 0 = the product's one-per-CU sizes.
extern "C" int wave_replan(const hdsm_params* prm, int32_t n_inst, int32_t n_rob, const int32_t* agent_id, const double* state_curr,
                           const double* traj_ref, const int32_t* n_poly, const int32_t* n_rows_static, const double* A_static,
                           const double* b_static, const double* plans_all, const uint8_t* has_plan, double* traj_out, double* ctrl_out,
                           uint8_t* poly_used, int32_t* status, double* obj, int32_t* qp_iters, int32_t* nodes, int32_t* sweeps,
                           int32_t* cand, uint32_t* flags, int32_t* warm, int32_t bounds_min, int32_t threads, int32_t cmax,
                           int32_t split_budget) {
  if (threads != 64 && threads != 128 && threads != 256) return -1;
  auto c = std::make_unique<hdsm::Consts>();
  const char* err = nullptr;
  int rc = hdsm::build_consts(prm, c.get(), &err);
  if (rc) return rc;
  const int N = c->N;
  // what k_plan_prepass hands to the kernel: packed positions of steps 1..N, and (large swarms) the sphere records
  std::vector<double> pos((size_t)n_rob * N * 3, 0.0), bounds((size_t)n_rob * 4, 0.0);
  for (int k = 0; k < n_rob; ++k) {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int i = 0; i < N; ++i)
      for (int ax = 0; ax < 3; ++ax) {
        const double v = has_plan[k] ? plans_all[((size_t)k * (N + 1) + 1 + i) * 9 + ax] : 0.0;
        pos[((size_t)k * N + i) * 3 + ax] = v;
        lo[ax] = v < lo[ax] ? v : lo[ax], hi[ax] = v > hi[ax] ? v : hi[ax];
      }
    double* b = &bounds[(size_t)k * 4];
    b[3] = -1.0;
    if (has_plan[k]) {
      double r2 = 0;
      for (int ax = 0; ax < 3; ++ax) b[ax] = 0.5 * (lo[ax] + hi[ax]);
      for (int i = 0; i < N; ++i) {
        double d2 = 0;
        for (int ax = 0; ax < 3; ++ax) {
          const double u = pos[((size_t)k * N + i) * 3 + ax] - b[ax];
          d2 += u * u;
        }
        r2 = d2 > r2 ? d2 : r2;
      }
      b[3] = sqrt(r2) * (1.0 + 1e-9);
      if (!(b[3] >= 0.0) || !(b[3] < 1e299)) b[3] = 1e300;
    }
  }
  hdsm::Args a{};
  a.n_inst = n_inst, a.n_rob = n_rob, a.agent_id = agent_id, a.state = state_curr, a.ref = traj_ref;
  a.n_poly = n_poly, a.n_rows = n_rows_static, a.A = A_static, a.b = b_static, a.plans = plans_all;
  a.has_plan = has_plan, a.traj = traj_out, a.ctrl = ctrl_out, a.used = poly_used, a.status = status;
  a.obj = obj, a.st_iters = qp_iters, a.st_nodes = nodes, a.st_sweeps = sweeps, a.st_cand = cand, a.st_flags = flags;
  a.pos = pos.data();
  a.bounds = (n_rob >= bounds_min) ? bounds.data() : nullptr;
  a.warm = (prm->warm_start && warm) ? warm : nullptr;
  a.split_budget = split_budget;
  if (cmax > 0 && cmax <= 16)  // tiny staging capacity: exercises the overflow path in tests
    return c->n <= hdsm::SPLIT_N_MAX ? run_all<32, 16>(*c, a, threads) : run_all<48, 16>(*c, a, threads);
  if (cmax == 256 && c->n <= hdsm::SPLIT_N_MAX && c->P <= 4 && c->RS <= 20)  // the four-workgroups-per-CU shape: small LDS layout, 256 staged rows
    return run_all<32, 256, true>(*c, a, threads);
  if (c->n <= hdsm::SPLIT_N_MAX) return run_all<32, 1536>(*c, a, threads);
  return run_all<48, 1024>(*c, a, threads);
}

// Level 1 (fully formed per-step polyhedra, hdsm_solve) through the product's host-side split and the device source
extern "C" int wave_solve(const hdsm_params* prm, int32_t n_inst, int32_t r_max, const double* state_curr, const double* traj_ref,
                          const int32_t* n_poly, const int32_t* n_rows, const double* A, const double* b, double* traj_out,
                          double* ctrl_out, uint8_t* poly_used, int32_t* status, double* obj, int32_t threads, int32_t* qp_iters,
                          int32_t* nodes) {
  if (threads != 64 && threads != 128 && threads != 256) return -1;
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
  a.st_iters = qp_iters, a.st_nodes = nodes;  // (may be null)
  if (c->n <= hdsm::SPLIT_N_MAX) return run_all<32, 1536>(*c, a, threads);
  return run_all<48, 1024>(*c, a, threads);
}
