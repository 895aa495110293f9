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
  int32_t overlap_sweep;  // 1 (default): two-wave workgroups run the first staging sweep on wave 1 while wave 0 installs the warm start (HDSM_OVERLAP_SWEEP=0: one after the other)
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
  int32_t dominance;             // 1 (default): the first time an instance has to branch, polyhedra CONTAINED in another polyhedron of the
                                 // instance are taken out of the choice (hdsm_core.h, dominated_mask): whatever lies in the smaller one lies in
                                 // the larger one, so offering both only multiplies the tree (HDSM_DOMINANCE=0: off). Exact.
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

constexpr int NOGOODS = 48;        // conflicts kept per instance (branch and bound)
constexpr int ITEMS_PER_REC = 64;   // items an instance can queue when it hands its search over (open children of <= MAXH levels)

// What an instance writes when it hands its search over to pass 2 of a split launch (Args::recs; see Args).
struct SplitRec {
  int32_t inst, level, ncand, ncold, n_nogood, first_item, n_items, nodes_done, sweeps_done, truncated;
  int32_t next;            // the next record of the same instance (-1: none): an item of pass 2 that hands over again chains its record in
  int32_t sp_dom;          // dominated polyhedra of the instance (Shm::sp_dom: bit j; the search that wrote the record had computed it)
  const double* snap;      // snapshots of the open levels (level l at snap + l * SNAP_STRIDE): the scratch the search was using
  double sw_tau;
  int32_t br_step[MAXH], br_cnt[MAXH], br_pos[MAXH], assign[MAXH];
  int32_t br_order[MAXH][MAXP], br_pk[MAXH][MAXP];
  double br_f[MAXH], br_lb[MAXH][MAXP], br_pv[MAXH][MAXP];
  double sw_ref[MAXH + 1][3];
  unsigned long long nogood[NOGOODS];
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
  // [n_inst][KROWS] the set-up map applied to every instance of the launch (rows [0, 3n + 12): x_eq, x0, gradient, residuals and
  // multipliers of the terminal equalities), written by the pre-pass kernel; null: every instance applies Consts::KTC itself
  const double* setup;
  // launch order: workgroup w solves instance order[2 w] (most expensive first, judged by the previous launch) of agent
  // order[2 w + 1], or null = w
  const int32_t* order;
  int32_t* st_sph;    // sphere records read by the sweeps of this instance
  int32_t* st_pairs;  // (neighbour, step) positions loaded by the sweeps of this instance
  uint32_t* st_flags; // HDSM_FLAG_* bits
  // ---- subtree splitting (hdsm_api.hip, launch()): a launch in three kernels for batches with deep branch-and-bound trees.
  // Pass 1 = the ordinary solve with a node budget: an instance whose tree is not finished by then stops and HANDS ITS SEARCH OVER — a
  // record of its open levels (branching steps, child orders, bounds, first picks), its staged neighbour rows and the assignments of
  // its current path goes to global memory (SplitRec + row arrays; the snapshots of the open levels are in its scratch already), and
  // every child of an open level that has not been explored yet becomes an ITEM of pass 2. Pass 2 = persistent workgroups that draw
  // items from one queue: set-up of the instance, the record, the snapshot of the item's level, and the search continues inside the
  // item's subtree exactly where pass 1 would have continued it — no warm start, no sweep, no node solved twice. Pass 3 = the merge.
  int32_t split_budget;   // pass 1: nodes after which an instance hands its search over (0 = ordinary launch); pass 2: ... an item hands over again
  int32_t split_min;      // pass 2: an item that has opened at least this many nodes hands over again as soon as the queue is EMPTY
                          // (workgroups are waiting for items): large subtrees are cut up while there is nobody to search them
  int32_t poll_sleep;     // pass 2: a workgroup that waits for items looks at the queue every poll_sleep x ~3.4 us (s_sleep 127)
  int32_t item_mode;      // pass 2: 1 = the workgroups draw items (instances come from the records; blockIdx is only a slot number)
  int32_t* split_info;    // [n_inst][2]: bit 0 handed over, bit 1 pass 1 left an incumbent in the instance's outputs; the record's slot
  unsigned long long* inc_bits;  // [n_inst]: best objective any item has found so far (bits of a non-negative double,
                                 // +inf at the start): the items of an instance prune against each other's incumbents
  int32_t node_cap;       // share of the instance's node budget (what pass 1 left of Consts::max_nodes) every item starts with
  int32_t nodes_pool0;    // ... and what pass 1 puts into the instance's pool when it hands over
  int32_t* node_pool;     // [n_inst] nodes handed back by items that finished below their share; an item that has used
                          // its share draws from here in chunks — the budget stays the instance's wherever in the tree the work is,
                          // and a tree that overruns it stops all its items at about the same time
  SplitRec* recs;         // [rec_cap] hand-over records
  double* rec_cand;       // [rec_cap][rows_cap][4] staged rows of a record: hot rows first, then the cold ones
  long long* rec_mw;      // [rec_cap][rows_cap] their (step, pick weight) pairs (MW of hdsm_core.h, 8 bytes)
  int32_t* rec_src;       // [rec_cap][rows_cap] their origins (Shm::cand_src)
  int32_t* rec_count;     // [0] records taken, [1] items queued (published), [2] tickets taken by the workgroups of pass 2 (ticket k = item k),
                          // [4] items completed, [5] places of the queue reserved, [6] abort word: a waiter of pass 2 or a publisher ran out of
                          // patience — nobody waits any longer, unpublished / undrawn items stay pending (zeroed before pass 1);
                          // [8 + r] the instance of record r (what SplitRec::inst says, packed for the merge's search)
  int32_t pool_cap;       // pass 2: snapshot-scratch slots (Args::scratch), taken by the workgroup of an item for its lifetime
  int32_t* slot_busy;     // [pool_cap] 0 / 1; an item that hands over again leaves its slot (busy) to its record
  int32_t* item_total;    // host-visible word: the merge leaves the number of items this launch queued
  int32_t* items;         // [items_cap]: (record << 8) | (level << 3) | child position
  int32_t* item_status;   // [items_cap] the status array of pass 2 (a queued item starts as pending)
  int32_t rec_cap, rows_cap, items_cap;
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
