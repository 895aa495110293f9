// hdsm_types.h — plain structs shared by the host code and the device code (no HIP dependency).
#pragma once
#include <stdint.h>

namespace hdsm {

constexpr int MAXH = 16;   // HDSM_MAX_HOR
constexpr int MAXP = 8;    // HDSM_MAX_POLY
constexpr int MAXRS = 32;  // HDSM_MAX_ROWS_STATIC
constexpr int MAXNV = 3 * MAXH;
constexpr int MAXT = 256;  // largest workgroup
constexpr int KCOLS = 9 + 6 * MAXH;     // inputs of the set-up map: state_curr, traj_ref
constexpr int KROWS = 3 * MAXNV + 12;   // outputs: x_eq, x0, gradient, equality residuals and multipliers
constexpr int SPLIT_N_MAX = 30;         // n <= 30 runs on the split kernel (NV = 32), larger n on NV = 48
constexpr int KCH = 3 + 2 * MAXH;       // inputs one output of the set-up map really depends on (its own axis)
constexpr double ABSENT = 1e20;
constexpr double DINF = 1e300;

// ---------------------------------------------------------------------------------------------------------
// Config-level constants, built once on the host by hdsm_build_consts() (the counterpart of
// Agent::CreateGurobiModel, AC:2071-2153) and kept in device memory.
struct Consts {
  int32_t N, n, P, RS;
  int32_t max_nodes, max_iters;
  int32_t presweep;  // stage the neighbour rows around the starting point BEFORE the first active-set run: 0 never,
                     // 1 always, 2 (default): always for swarms below the prefilter size; prefiltered swarms only when
                     // the warm start already holds neighbour rows. Measured with the final kernel: -8 % on the bench
                     // line, -11 % at 1024 agents late in the flight, -7 % at H=15 (it also spares B&B nodes)
  int32_t leaf_mfma;    // leaf test through v_mfma_f64_16x16x4_f64 (HDSM_LEAF_MFMA, see hdsm_core.h leaf_check)
  int32_t branch_rule;  // step to branch on: 0 first uncontained segment in time, 1 (default) the most infeasible one
                        // (HDSM_BRANCH_RULE). Either is exact; 1 bisects the "where to switch polyhedron" choice
                        // instead of enumerating it: 509 -> 29 nodes on a gridlocked 128-agent ring
  int32_t pick_rule;    // row that enters the working set next: 0 the most violated one, 1 (default) the most violated one in the
                        // metric of the problem, violation / sqrt(a^T Z a) with Z = H^-1 projected on the terminal equalities
                        // (HDSM_PICK_RULE): a jerk bound (tens of m/s^3) and a separating plane (centimetres) become comparable
  int32_t box_cut;      // 1 (default): a dual objective above the largest objective any point of the input box can have ends an
                        // active-set run as infeasible (HDSM_BOX_CUT); 0: only the formal proof (dependent row, no multiplier to give way)
  int32_t scanner;      // 1 (default): n <= 30, workgroups of two or more wavefronts — wave 1 evaluates the trajectory and picks the next
                        // row while wave 0 applies the Householder update of the operation before (HDSM_SCANNER; 0: wave 0 does both;
                        // 2, development: as 1 but every pick of the scanner is confirmed by its exact evaluation)
  int32_t child_bound;  // 1 (default): a child of a branch-and-bound node whose lower bound f + v^2 / (2 a^T Z a) — v the violation of a row
                        // of its polyhedron at the node's minimiser — reaches the incumbent is not opened (HDSM_CHILD_BOUND=0: off). Exact.
  int32_t pad_child_bound;
  double tol, ftol_fixed, cand_tau, hot_tau;
  double mip_gap;  // relative gap at which a node is cut off against the incumbent (0 = exact)
  long long time_ticks;  // hdsm_params.time_limit_s in ticks of the device's constant-rate clock (0 = no time limit)
  double r_u, wx[6], wn[6];
  double lbu[3], ubu[3];       // input box (absent if |.| >= ABSENT)
  double lbs[3][3], ubs[3][3]; // state box [comp][ax], comp 1 = v, 2 = a
  double radius, k2m1, pert;   // drone_radius, (r/h)^2 - 1, plane perturbation
  double Ad[3][3][3], Bd[3][3];  // one-step maps per axis: x+ = Ad x + Bd u   (Euler or RK4, AC:2115-2152)
  double g[3][3][MAXH];          // impulse responses: g[ax][s][lag] = (Ad^lag Bd)[s]
  int32_t pinned_steps;          // positions p_1 .. p_pinned do not depend on the inputs at all (the impulse response of the
                                 // position is zero for that many lags: 2 with jerk inputs and the Euler model)
  int32_t pad_pinned;
  double phi[3][MAXH + 1][3][3]; // Ad^i
  double Hinv[MAXNV * MAXNV];    // inverse Hessian, dense n x n, row-major with stride n
  double J0[MAXNV * MAXNV];      // L^{-T}, H = L L^T
  // State of the active-set method AFTER the six terminal equalities v_N = a_N = 0 (AC:2078-2081) have been
  // added (they are in every working set and their normals are config constants):
  double Jeq[MAXNV * MAXNV];     // J with J^T E^T = [Req; 0], stride n
  double Req[36], Ueq[36];       // Req upper triangular 6x6, Ueq = Req^{-1}
  double Meq[MAXNV * 6];         // x_eq = x0 - Meq * resid,  resid_e = (E x0 - e)_e;  Meq = H^{-1}E^T (E H^{-1} E^T)^{-1}
  double Seq[36];                // (E H^{-1} E^T)^{-1}: multipliers nu = Seq * resid, f_eq = f(x0) + 1/2 resid' Seq resid
  // Everything an instance needs before its first iteration is LINEAR in v = (state_curr[9], traj_ref[N][6]):
  //   rows [0,n) x_eq, [n,2n) x0, [2n,3n) gradient at u = 0, [3n,3n+6) resid, [3n+6,3n+12) nu.
  // KT[j * KROWS + row] is the coefficient of v[j] (input-major, so a wavefront reads consecutive rows coalesced).
  double KT[KCOLS * KROWS];
  // The axes are decoupled (dynamics, cost and terminal equalities are per axis), so a row that belongs to axis ax has
  // non-zeros only at state_curr[3 cc + ax] and traj_ref[i][3 comp + ax] — flat positions ax + 3 u, u < 3 + 2N. The
  // device set-up uses this compact form: KTC[u * KROWS + row] = KT[(kax[row] + 3 u) * KROWS + row].
  double KTC[KCH * KROWS];
  int32_t kax[KROWS];
  // Jeq (identity beyond n) in the register layout of the kernel that serves this n (hdsm_wave_gi.h):
  //   n <= 30 (split kernel, NV = 32): JeqP[s * 64 + lane] = Jeq[lane & 31][((s ^ lane) & 15) + 16 (lane >> 5)], s < 16
  //                                    (the butterfly slot order of hdsm_wave_gib.h)
  //   n  > 30 (NV = 48):               JeqP[j * 64 + lane] = Jeq[lane][j], j < 48
  double JeqP[MAXNV * 64];
  // weights of the pick rule, 1 / sqrt(a^T Z a) of the constant rows and the per-axis factors of the position rows:
  double hrow1[MAXNV];            // sum_k |H_jk|, rounded up (the bound of the objective over the input box, Shm::f_box)
  double wu[MAXNV];               // input box of variable k
  double ws[3][3][MAXH + 1];      // state box [ax][comp][step]
  double kap[MAXH + 1][4];        // a row n . p_m <= b has a^T Z a = sum_ax n_ax^2 kap[m][ax]
};

// Per-launch arguments (device pointers), layouts of include/hdsm.h.
struct Args {
  int32_t n_inst, n_rob;
  const int32_t* agent_id;
  const double* state;
  const double* ref;
  const int32_t* n_poly;
  const int32_t* n_rows;
  const double* A;
  const double* b;
  const double* plans;
  const uint8_t* has_plan;
  double* traj;
  double* ctrl;
  uint8_t* used;
  int32_t* status;
  double* obj;
  int32_t* st_iters;
  int32_t* st_nodes;
  int32_t* st_sweeps;
  int32_t* st_cand;
  double* scratch;        // n_inst * scratch_stride doubles
  int64_t scratch_stride;
  // level 1 (hdsm_solve): explicit rows common to every polyhedron of a step, instead of the plans buffer
  const double* l1_rows;   // [n_inst][N][l1_rmax][4] or null
  const int32_t* l1_nrows; // [n_inst][N]
  int32_t l1_rmax;
  int32_t* warm;          // [n_inst][MAXNV + 2]: count + portable ids of the previous optimal working set (in/out), or null
  long long* prof;        // HDSM_PROFILE builds: 16 cycle counters per instance (else null)
  // [n_rob][4] = (centre, radius) of a sphere around steps 1..N of every published plan, radius < 0 = no plan;
  // written by k_plan_bounds before the launch for large swarms, null = sweeps test every neighbour step by step
  const double* bounds;
  // [n_rob][N][3]: positions of steps 1..N of every published plan, packed (24 B per (agent, step) instead of a 72-B
  // stride through the full states); written by the same pre-pass for every launch of level 2
  const double* pos;
  // launch order: workgroup w solves instance order[2 w] (most expensive first, judged by the previous launch) of agent
  // order[2 w + 1], or null = w
  const int32_t* order;
  int32_t* st_sph;    // sphere records read by the sweeps of this instance
  int32_t* st_pairs;  // (neighbour, step) positions loaded by the sweeps of this instance
  uint32_t* st_flags; // HDSM_FLAG_* bits
  // ---- subtree splitting (hdsm_api.hip, launch_split): a launch in three kernels for batches with deep branch-and-bound trees
  int32_t split_budget;   // pass 1: an instance whose tree is not finished after this many nodes stops WITHOUT outputs and
                          // records the step its root branched on in split_info (0 = ordinary launch)
  int32_t sub_k;          // pass 2: block b continues instance b / sub_k inside the subtree "root step = polyhedron b % sub_k";
                          // instances that were not handed over leave at once (0 = ordinary launch)
  int32_t* split_info;    // [n_inst][2]: handed over (0 / 1), root branching step
  unsigned long long* inc_bits;  // [n_inst]: best objective any sub-block has found so far (bits of a non-negative double,
                                 // +inf at the start): the sub-blocks of an instance prune against each other's incumbents
  int32_t node_cap;       // pass 2: share of the instance's node budget (what pass 1 left of Consts::max_nodes) every sub-block starts with
  int32_t* node_pool;     // pass 2: [n_inst] nodes handed back by sub-blocks that finished below their share; a sub-block that has used
                          // its share draws from here in chunks — the budget stays the instance's wherever in the tree the work is,
                          // and a tree that overruns it stops all its sub-blocks at about the same time
  int32_t* split_steps;   // pass 2, further split levels: [n_inst][split_ss] branching step agreed for the node behind every prefix of split
                          // digits (-1: nobody has reached it): the sub-blocks that share a prefix all solve its node, and the FIRST one to
                          // get there decides the step for all of them — their own minimisers differ by rounding, and near a tie they would
                          // otherwise branch on different steps and leave children that nobody searches
  int32_t split_ss;       // entries per instance in split_steps: poly_hor + poly_hor^2 (levels 1 and 2)
  int32_t* sub_slots;     // pass 2: pool of snapshot-scratch slots: [1] = capacity, [2 + i] = slot i taken (0 / 1)
  int32_t* tree_flag;     // host-visible word: set to 1 by an instance whose tree reached tree_mark nodes (ordinary launch) or, in a
                          // split launch, by the merge for a handed-over instance with a deep tree (TREE_MARK nodes over all its
                          // sub-blocks) — the handle keeps the NEXT launches in the split form while it sees it
  int32_t tree_mark;      // ordinary launches: node count from which an instance raises tree_flag (0 = never)
  int32_t rescue;         // 1: this launch only re-solves the instances whose last answer carries HDSM_FLAG_STAGING_OVERFLOW (st_flags), with the
                          // large staging area of the one-per-CU kernel; the others leave at once
  int32_t* ovf_flag;      // host-visible word an instance raises when it ends on a staging overflow (the handle then adds the rescue pass)
  int32_t* warm_out;      // where the NEXT replan's guess is written: warm itself, or the per-sub-block copy of pass 2
  int32_t* st_key;    // launch-order key for the NEXT launch: duration of this instance in 0.64-us units + 9 per active row (<= 254), 255 = no solution
};

}  // namespace hdsm
