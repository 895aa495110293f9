/*
 * hdsm.h — C ABI of the MI355X-native batched trajectory-optimisation solver.
 *
 * This is the drop-in boundary for ONE hot path of lis-epfl/multi_agent_pkgs: the per-agent
 * receding-horizon MIQP that `multi_agent_planner::Agent` builds and hands to Gurobi every 100 ms.
 * Citations are relative to the reference repo; AC = multi_agent_planner/src/agent_class.cpp,
 * AH = multi_agent_planner/include/multi_agent_planner/agent_class.hpp.
 *
 * The reference has no plugin/FFI interface for this path: the seam is the set of Gurobi C++ calls inside
 * the private method Agent::SolveOptimizationProblem() (AC:858-1023), which communicates only through
 * member variables. Each entry point below names the reference code it replaces. INTEGRATION.md shows the
 * few lines a maintainer of the reference adds to agent_class.cpp to call it.
 *
 * Conventions
 *   - plain C, plain pointers and sizes; all reals are IEEE double (decimal_t = double in the reference,
 *     decomp_basis/data_type.h:50; ROS float64; Gurobi doubles);
 *   - every function returns 0 on success and a negative hdsm_error on API misuse / device error; nothing
 *     throws (the reference swallows every Gurobi exception too, AC:988-995);
 *   - per-instance solver outcome is reported in `status[]` (hdsm_status), never through the return code;
 *   - one handle per calling thread; a handle is not thread-safe (the reference calls the solver from one
 *     dedicated thread per agent, AC:119);
 *   - the library needs a gfx950 device: there is NO CPU fallback. hdsm_create() fails with
 *     HDSM_ERR_NO_DEVICE when no HIP device is present.
 *
 * Array layouts (row-major, C order). N = n_hor, P = poly_hor, RS = max_rows_static:
 *   state_curr  [n_inst][9]          (px,py,pz, vx,vy,vz, ax,ay,az)           state_curr_   AH:446
 *   traj_ref    [n_inst][N][6]       rows 0..N-1 of traj_ref_curr_ (p, v)       AC:864-883
 *   traj_out    [n_inst][N+1][9]     traj_curr_                                 AH:459, AC:962-971
 *   ctrl_out    [n_inst][N][3]       control_curr_                              AH:461, AC:973-977
 *   poly_used   [n_inst][P]  uint8   poly_used_idx_                             AH:483, AC:979-985
 *   plans_all   [n_rob][N+1][9]      the all-gathered traj_full of every agent  Trajectory.msg, AC:645-677
 *   has_plan    [n_rob]      uint8   0 = no message received from that agent yet (AC:1134)
 */
#ifndef HDSM_H
#define HDSM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HDSM_MAX_HOR 16         /* largest supported n_hor (reference ships 9, BASELINE uses 10 and 15)   */
#define HDSM_MAX_POLY 8         /* largest supported poly_hor (reference ships 3 or 4)                    */
#define HDSM_MAX_ROWS_STATIC 32 /* largest supported rows per static polyhedron (GetPolyOcta3D emits <=18)*/
#define HDSM_INF 1e100          /* GRB_INFINITY: bounds with |value| >= 1e20 are treated as absent        */

typedef enum hdsm_error {
  HDSM_OK = 0,
  HDSM_ERR_BAD_ARG = -1,    /* null pointer, size out of range, unsupported parameter combination        */
  HDSM_ERR_NO_DEVICE = -2,  /* no gfx950 HIP device / device index out of range                           */
  HDSM_ERR_DEVICE = -3,     /* HIP runtime error (allocation, launch, copy); see hdsm_last_error()       */
  HDSM_ERR_CAPACITY = -4,   /* n_inst > max_instances or n_rob > handle capacity                         */
  HDSM_ERR_COMM = -5        /* RCCL error in hdsm_comm_* / hdsm_exchange_device; see hdsm_last_error()    */
} hdsm_error;

/* Per-instance outcome. The reference never inspects Gurobi's status (AC:959-995): a time-limit WITH an
 * incumbent is silently accepted as success, everything else surfaces as an exception -> fallback.       */
typedef enum hdsm_status {
  HDSM_OPTIMAL = 0,    /* proven optimal assignment + KKT point (residuals <= solver tolerance)            */
  HDSM_LIMIT = 1,      /* node/iteration limit hit, incumbent returned (Gurobi TimeLimit with incumbent)   */
  HDSM_NO_SOLUTION = 2 /* infeasible, limit without incumbent, or degenerate input (NaN plane): the caller
                          applies the shift-by-one fallback of AC:1000-1019; outputs are left untouched    */
} hdsm_status;

/* Mirrors the Agent members the solve reads (AH:297-357, AH:426-432); filled from the ROS parameters
 * exactly as InitializePlannerParameters does (AC:2169-2188).                                             */
typedef struct hdsm_params {
  int32_t n_hor;            /* N, n_hor_                                       AH:297                      */
  int32_t poly_hor;         /* P, poly_hor_                                    AH:373                      */
  int32_t rk4;              /* rk4_: 0 forward Euler, 1 classical RK4          AC:2115-2152                */
  int32_t max_rows_static;  /* RS: row capacity of one static polyhedron in A_static (<= HDSM_MAX_ROWS_STATIC) */
  double dt;                /* dt_                                                                          */
  double drag[3];           /* drag_coeff_                                     AC:2157-2165                */
  double r_u;               /* r_u_  jerk weight                               AC:2098                     */
  double r_x[9];            /* r_x_  running state weights (only [0..5] used)  AC:871-883                  */
  double r_n[9];            /* r_n_  terminal state weights (only [0..5] used)                             */
  double x_lb[9], x_ub[9];  /* x_lb_/x_ub_ (positions +-HDSM_INF)              AC:2179-2184                */
  double u_lb[3], u_ub[3];  /* u_lb_/u_ub_                                     AC:2185-2186                */
  double drone_radius;      /* drone_radius_                                   AH:342                      */
  double drone_z_offset;    /* drone_z_offset_                                 AH:344                      */
  double plane_perturb;     /* var_tmp = 0.1 hard-coded in the reference       AC:1180                     */
  /* Deterministic stand-ins for Gurobi's wall-clock TimeLimit = 0.08 s (AC:952): a time-limited solve is
   * not reproducible, so limits are expressed in work units. 0 = library default.                         */
  int32_t max_nodes;        /* branch-and-bound node budget per instance                                   */
  int32_t max_qp_iters;     /* active-set iteration budget per instance (summed over nodes)                */
  double feas_tol_fixed;    /* tolerance for rows on the pinned point p_0 (Gurobi FeasibilityTol 1e-6)     */
  double solver_tol;        /* primal feasibility tolerance of the exact active-set solver (default 1e-9)  */
  /* The reference keeps ONE GRBModel per agent across replans (AC:32), so Gurobi restarts from the previous
   * solution. 1 = do the same: the optimal working set of the previous hdsm_replan / hdsm_replan_device call is
   * kept per instance INDEX on the handle and seeds the next solve (the answer does not depend on it, only
   * the work). The caller must then keep the agent <-> instance index mapping stable between calls, or call
   * hdsm_reset_warm_start() when it changes.                                                              */
  int32_t warm_start;
  /* Execution knobs (0 = library default everywhere). They shape the work, never the answer; the HDSM_* environment
   * variables of the same names remain as overrides for scripts (validated, out-of-range values are ignored).   */
  int32_t threads_per_instance; /* 256 (default) or 64 threads per agent-replan                      HDSM_THREADS      */
  int32_t prefilter_min_agents; /* swarms of at least this many agents get the sphere prefilter
                                   (default 256; negative = never)                                  HDSM_BOUNDS_MIN   */
  int32_t duo_min_instances;    /* batches of at least this many instances run several workgroups per CU: two from this
                                   count on (default: compute units + 1; negative = never), and for n_hor <= 10 three
                                   128-thread ones from 2 x compute units + 1 on (HDSM_TRI_MIN)     HDSM_DUO_MIN      */
  int32_t presweep;             /* neighbour rows staged before the first active-set run: 0 automatic,
                                   1 never, 2 always                                                HDSM_PRESWEEP     */
  int32_t branch_rule;          /* branch on: 0 the most infeasible segment (default), 1 the first in time HDSM_BRANCH_RULE */
  int32_t launch_order;         /* batches of at least this many instances are launched most-expensive-first, judged by
                                   the previous launch (needs warm_start; default 2 x compute units + 1; negative = never)
                                                                                                    HDSM_ORDER_MIN    */
  double stage_radius;          /* [m] slack below which a neighbour row is staged (default 0.6)     HDSM_CAND_TAU     */
  /* (Environment only, for A/B scripts. HDSM_SPLIT 0 / 1 / 2 = never / always / automatically (default) run a launch whose
   * predecessors met a deep branch-and-bound tree (32 nodes) as three kernels — a budgeted solve in which an instance that
   * exceeds the budget hands its search over (a record of its open levels; one queue item per unexplored child); persistent
   * workgroups that draw the items and continue inside their subtrees, handing over again when a subtree grows large
   * (HDSM_ITEM_BUDGET nodes, default 32; HDSM_ITEM_MIN while workgroups wait for items); the merge. HDSM_SPLIT_BUDGET = the
   * budget of the first kernel in nodes (1 or more; default by batch size: 2 for batches of at most 2 x compute units instances,
   * 8 beyond — a hand-over costs about one node since round 5). HDSM_ITEM_MIN has the same by-batch default (2 / 8).
   * HDSM_SPLIT_RECORDS: hand-over records per launch (default 8 x max_instances, clamped to 256 .. 2048); HDSM_POLL_SLEEP: s_sleep(127)
   * periods between two looks of a waiting workgroup at the item queue (default 2). max_nodes stays
   * the budget of an INSTANCE: its items draw from one pool. HDSM_DOMINANCE 1 (default) / 0: the first time an instance has
   * to branch, polyhedra that lie inside another polyhedron of the instance are taken out of the choice (one left: its rows
   * are assigned to every uncontained step at once) / every polyhedron is offered. HDSM_CHILD_BOUND 1 (default) / 0: children of a branching node
   * get a lower bound and their first entering row from the node's leaf test / neither. HDSM_SETUP_MFMA 1 (default) / 0: the
   * set-up map of all instances of a launch is one product on the matrix cores in the pre-pass kernel / every instance applies
   * it itself. HDSM_OVERLAP_SWEEP 1 (default) / 0: two-wave workgroups run the first staging sweep during the warm-start
   * install / after it.
   * HDSM_PICK_RULE 1 (default) / 0: the row that enters the working
   * set next is the most violated one in the metric of the problem (violation / sqrt(a' Z a)) / the most violated one.
   * HDSM_BOX_CUT 1 (default) / 0: a dual objective above the largest objective any point of the input box can have ends an
   * active-set run as infeasible / only the formal proof does.
   * HDSM_QUAD_MIN: batches of at least this many instances with n_hor <= 10 run FOUR 128-thread workgroups per CU (small
   * LDS layout; default 3 x compute units + 1, 0 = never). HDSM_SCANNER 1 (default) / 0: in workgroups of more than one
   * wavefront the second one evaluates the trajectory and picks the row that enters next while the first applies the update
   * of the operation before / the iterating wavefront does both.
   * None of these changes an answer that is HDSM_OPTIMAL.)                                                       */
  /* Gurobi's TimeLimit (0.08 s, AC:952) as an OPTIONAL wall-clock budget per instance, measured on the device's
   * constant-rate clock from the start of the instance's workgroup: when it is spent the branch-and-bound stops and
   * returns the incumbent (HDSM_LIMIT) or HDSM_NO_SOLUTION — what Gurobi does, and just as irreproducible. 0 = none
   * (the deterministic work budgets above are the default and what every parity test uses).                      */
  double time_limit_s;
  /* Gurobi's MIPGap (default 1e-4, never set by the reference: SURVEY section 7-4): a branch-and-bound node is cut off when its
   * bound is within mip_gap * |incumbent| of the incumbent, so the returned assignment may be worse than the best one by that
   * much — what Gurobi accepts. 0 (default) = prove optimality exactly (nodes are cut off only at 1e-9 relative).            */
  double mip_gap;
} hdsm_params;

/* Fills `p` with the agile configuration shipped by the reference
 * (multi_agent_planner/config/agent_agile_config.yaml) with n_hor overridden by the caller.               */
void hdsm_default_params(hdsm_params* p, int32_t n_hor);

/* Replaces: GRBEnv/GRBModel construction + Agent::CreateGurobiModel() (AC:5, AC:32, AC:2063-2153), done
 * once per process in the reference. Builds the condensed dynamics, the Hessian factor and every
 * config-level constant on the host, uploads them, allocates per-instance device scratch.
 * `n_rob_max` bounds the n_rob later passed to hdsm_replan / hdsm_replan_device. `device` = HIP device ordinal.               */
int hdsm_create(const hdsm_params* params, int32_t max_instances, int32_t n_rob_max, int32_t device,
                void** handle);
void hdsm_destroy(void* handle);

/* Level 2 — fused stand-in for Agent::GenerateTimeAwareSafeCorridor() (AC:1086-1215, AddHyperplane
 * AC:1217-1234) followed by Agent::SolveOptimizationProblem() (AC:858-1023): the caller passes the static
 * polyhedra (poly_const_vec_, AH:469) and the all-gathered plans of every agent; the separating planes
 * are generated on the device and never materialised.
 *
 *   agent_id      [n_inst]  id_ of each instance: its own previous plan is plans_all[agent_id] when
 *                 has_plan[agent_id] != 0 (traj_curr_), else state_curr (state_ini_, AC:1103-1110); it
 *                 never builds a plane against itself (AC:617).
 *   n_poly        [n_inst]  poly_const_vec_.size(); only the first min(P, n_poly) are used (AC:913).
 *   n_rows_static [n_inst][P]
 *   A_static      [n_inst][P][RS][3], b_static [n_inst][P][RS]   rows A x <= b (polyhedron.h:98-147)
 *
 * Host-pointer form: copies inputs to the device, runs, copies results back (PCIe-inclusive).             */
int hdsm_replan(void* handle, int32_t n_inst, int32_t n_rob, const int32_t* agent_id,
                const double* state_curr, const double* traj_ref, const int32_t* n_poly,
                const int32_t* n_rows_static, const double* A_static, const double* b_static,
                const double* plans_all, const uint8_t* has_plan, double* traj_out, double* ctrl_out,
                uint8_t* poly_used, int32_t* status, double* obj);

/* Same contract, every pointer is a DEVICE pointer, the launch is asynchronous on `hip_stream`
 * (a hipStream_t passed as void*; NULL = the default stream). Nothing is copied or synchronised: this is
 * the form the batched harness and bench.py time, with inputs resident in HBM. `traj_out` may alias the
 * caller's shard of the NEXT round's plans buffer (it must not alias `plans_all` of this call).
 * Streams: a handle owns per-instance device state (branch-and-bound snapshots, warm-start sets, the prefilter
 * records). Launches on ONE stream are ordered by the stream; when a call arrives on a different stream than the
 * previous launch of the same handle, the library makes the new stream wait for that launch (event), so successive
 * calls never overlap on the handle's state whatever streams they use. The host-pointer entry points use the handle's
 * own stream. Every entry point selects the handle's device itself.                                          */
int hdsm_replan_device(void* handle, int32_t n_inst, int32_t n_rob, const int32_t* agent_id,
                       const double* state_curr, const double* traj_ref, const int32_t* n_poly,
                       const int32_t* n_rows_static, const double* A_static, const double* b_static,
                       const double* plans_all, const uint8_t* has_plan, double* traj_out,
                       double* ctrl_out, uint8_t* poly_used, int32_t* status, double* obj,
                       void* hip_stream);

/* Optional, for the host-pointer entry points: page-locks a LONG-LIVED array of the caller (the binding's input and output
 * buffers, allocated once and reused every round — the reference's Agent keeps its own members the same way) so that
 * hdsm_replan / hdsm_solve / hdsm_reference move it by DMA without the driver's staging copy. When ALL FIVE output arrays of an
 * hdsm_replan call lie in registered memory the results are written straight into them by the device, and only for instances
 * that have a solution (the "left untouched" rule is kept; no staging download, no host-side copy). Arrays that were never
 * registered keep working as before. Unregister an array before freeing it. Not needed for device pointers.
 * The library records every range registered HERE and takes the direct paths only for arrays that lie inside one of them with all
 * the bytes the call touches (n_inst / n_rob items): an array that runs past its registered range, or memory page-locked by other
 * means, goes through the copy path like pageable memory — never a device fault.                                             */
int hdsm_host_register(void* ptr, size_t bytes);
int hdsm_host_unregister(void* ptr);

/* Level 1 — stand-in for the Gurobi part alone (AC:870-1019): the caller passes the fully formed
 * per-step polyhedra poly_const_final_vec_[N][<=P] (AH:471), i.e. static rows followed by the neighbour
 * planes AddHyperplane appended. Accepted input = what GenerateTimeAwareSafeCorridor produces: every step has the
 * SAME number of polyhedra, polyhedron (i, j) = the static rows of polyhedron j (identical for every step, at most
 * max_rows_static of them) followed by rows that are identical for every j of that step (AddHyperplane appends each
 * plane to all polyhedra of the step, AC:1217-1234). Anything else is rejected with HDSM_ERR_BAD_ARG: the solver
 * treats the common suffix as ordinary rows and only the static head as the disjunction.
 *
 *   n_poly  [n_inst][N]          poly_const_final_vec_[i].size()
 *   n_rows  [n_inst][N][P]       rows of polyhedron (i, j)
 *   A       [n_inst][N][P][r_max][3],  b [n_inst][N][P][r_max]
 * Host pointers; PCIe-inclusive.                                                                          */
int hdsm_solve(void* handle, int32_t n_inst, int32_t r_max, const double* state_curr,
               const double* traj_ref, const int32_t* n_poly, const int32_t* n_rows, const double* A,
               const double* b, double* traj_out, double* ctrl_out, uint8_t* poly_used, int32_t* status,
               double* obj);

/* ---- next row f1: reference-trajectory generation (Agent::GenerateReferenceTrajectory AC:1449-1553) ----------
 * Mirrors the ROS parameters the reference reads for it (AC:2190-2248).                                    */
typedef struct hdsm_ref_config {
  double path_vel_min, path_vel_max; /* path_vel_min_/max_  (agile config: 4.5 / 9.0)                        */
  double sens_dist, sens_pot;        /* GetVelocityLimit sensitivities (0.05 / 0.18)                         */
  double sens_other_agents;          /* default 1.0 (AC:2213)                                                */
  double path_vel_dec;               /* deceleration of the sampling distance (0.0)                          */
} hdsm_ref_config;

/* For every instance: path_vel = min(vel_cap, neighbour term of Agent::ComputePathVelocity AC:1769-1801 with
 * GetVelocityLimit AC:1805-1817) over all steps of the agent's own previous plan (plans_all[agent_id], skipped
 * while it has none) and all other agents that have a plan; then Agent::SamplePath (AC:1591-1663) along the
 * polyline path[n_path] (first point = sampling start) and the velocity references of AC:1527-1547.
 *   path     [n_inst][pmax][3], n_path [n_inst] (>= 1)
 *   vel_cap  [n_inst] or NULL: the voxel/potential-field part of ComputePathVelocity (AC:1709-1766), which stays
 *            on the host with the map; NULL = path_vel_max (obstacle-free world)
 *   ref_full [n_inst][N+1][6]  traj_ref_curr_ (all N+1 rows: the increment check AC:569-585 needs them)
 *   ref      [n_inst][N][6]    rows 0..N-1, the layout hdsm_replan* reads (may be NULL)
 *   path_vel [n_inst]          path_vel_ (a single-point path, n_path = 1, writes 0: the reference leaves path_vel_ at its
 *                              previous value there, AC:1598-1611, and never uses it)
 * Host pointers / device pointers + stream, like hdsm_replan / hdsm_replan_device. The host form checks
 * 1 <= n_path[k] <= pmax (HDSM_ERR_BAD_ARG); the device form trusts its caller.                             */
int hdsm_reference(void* handle, const hdsm_ref_config* cfg, int32_t n_inst, int32_t n_rob,
                   const int32_t* agent_id, const double* path, const int32_t* n_path, int32_t pmax,
                   const double* vel_cap, const double* plans_all, const uint8_t* has_plan,
                   double* ref_full, double* ref, double* path_vel);
int hdsm_reference_device(void* handle, const hdsm_ref_config* cfg, int32_t n_inst, int32_t n_rob,
                          const int32_t* agent_id, const double* path, const int32_t* n_path, int32_t pmax,
                          const double* vel_cap, const double* plans_all, const uint8_t* has_plan,
                          double* ref_full, double* ref, double* path_vel, void* hip_stream);

/* ---- next row f4: map pre-processing ---------------------------------------------------------------------
 * The three stencil passes mapping_util's MapBuilder runs on every voxel grid before the planner sees it
 * (map_builder.cpp:209-216): SetUncertainToUnknown (map_builder.cpp:331-362), VoxelGrid::InflateObstacles and
 * VoxelGrid::CreatePotentialField (voxel_grid_util/src/voxel_grid.cpp:252-297, masks from CreateMask :192-226).
 * Grids are int8 [nz][ny][nx], x fastest: -1 unknown, 0 free, 100 occupied; the output adds 1..99 = potential.
 * A batch of n_grids grids of the same dimensions (the local grids of n agents) is processed at once.          */
typedef struct hdsm_map_config {
  double voxel_size;      /* 0.3                                  */
  double inflation_dist;  /* 0.3  (map_builder_default_config.yaml:8) */
  double potential_dist;  /* 1.5                                  */
  int32_t potential_pow;  /* 4    (CreatePotentialField takes an int) */
  int32_t reserved0;
} hdsm_map_config;
/* Host pointers: copies in, runs, copies out, synchronises.                                                 */
int hdsm_map_preprocess(int32_t device, const hdsm_map_config* cfg, int32_t n_grids, const int32_t dim[3],
                        const int8_t* grids_in, int8_t* grids_out);
/* Device pointers on `hip_stream`; scratch = 2 * n_grids * nx*ny*nz bytes of device memory.                  */
int hdsm_map_preprocess_device(int32_t device, const hdsm_map_config* cfg, int32_t n_grids, const int32_t dim[3],
                               const int8_t* grids_in, int8_t* grids_out, void* scratch, void* hip_stream);
const char* hdsm_map_last_error(void);

/* Stand-alone plane generator = Agent::GenerateTimeAwareSafeCorridor's inner maths (AC:1100-1205) for one
 * batch: planes[n_inst][N][n_rob][4] = (n_f.x, n_f.y, n_f.z, n_f . q), rows of absent/self agents are
 * filled with zeros. Host pointers. Used by tests and by callers that want the level-1 input.             */
int hdsm_tasc_planes(void* handle, int32_t n_inst, int32_t n_rob, const int32_t* agent_id,
                     const double* state_curr, const double* plans_all, const uint8_t* has_plan,
                     double* planes);

/* Diagnostics of the last hdsm_replan / hdsm_replan_device / hdsm_solve call on this handle (host arrays, may be NULL):
 *   qp_iters[n_inst]  active-set iterations, nodes[n_inst]  branch-and-bound nodes,
 *   sweeps[n_inst]    full passes over the neighbour buffer, cand[n_inst] neighbour rows ever staged.
 * Synchronises the handle's stream.                                                                       */
int hdsm_last_stats(void* handle, int32_t n_inst, int32_t* qp_iters, int32_t* nodes, int32_t* sweeps,
                    int32_t* cand);

/* More diagnostics of the last launch (host arrays, may be NULL):
 *   sphere_records[n_inst]  32-byte prefilter records read over all sweeps (0 when the prefilter is off),
 *   pairs[n_inst]           (neighbour, step) positions loaded by the sweeps, 24 bytes each,
 *   flags[n_inst]           HDSM_FLAG_* bits: why an instance stopped short of a proof.                          */
#define HDSM_FLAG_NODE_LIMIT 1u       /* max_nodes spent                                                        */
#define HDSM_FLAG_ITER_LIMIT 2u       /* max_qp_iters spent                                                     */
#define HDSM_FLAG_TIME_LIMIT 4u       /* time_limit_s spent                                                     */
#define HDSM_FLAG_STAGING_OVERFLOW 8u /* more violated neighbour rows than staging slots: not a search budget. The kernels that
                                       * share a compute unit (large batches) have fewer slots than the one-per-CU kernel;
                                       * hdsm_replan solves an instance that overflowed them again at once with the large
                                       * area, hdsm_replan_device reports the flag on the first launch and adds that rescue
                                       * pass to the launches that follow (the handle keeps it on for 256 launches)        */
int hdsm_last_sweep_stats(void* handle, int32_t n_inst, int32_t* sphere_records, int32_t* pairs, uint32_t* flags);

/* Duration of the SOLVER KERNEL of the last hdsm_replan[_device] / hdsm_solve on this handle, by HIP events recorded on the
 * launch stream right before and right after it (i.e. after the pre-pass; measurement aid: bench.py's roofline, which must
 * agree with the kernel's duration in a rocprofv3 trace). Off by default; hdsm_last_kernel_ms synchronises on the second event. */
int hdsm_set_kernel_timing(void* handle, int32_t on);
int hdsm_last_kernel_ms(void* handle, float* ms);

/* ---- multi-GPU: the per-round exchange of the published plans -------------------------------------------------
 * Replaces the DDS all-to-all of the reference (publisher AC:46-48 / 645-677, n_rob - 1 subscriptions AC:610-627,
 * callback AC:629-643) by ONE RCCL all-gather per replan round. Agents are sharded in contiguous id blocks of `per`
 * agents per rank (the last block may be padded). A rank publishes plans_local[per][N+1][9]; an agent WITHOUT a
 * plan (has_plan = 0: nothing solved yet, AC:1134, or padding) is published as a record whose first entry is NaN,
 * so the flag travels inside the same message — hdsm_publish_device writes that form from (traj, has_plan).
 * hdsm_exchange_device all-gathers the shards into plans_all[world * per][N+1][9] (rank r's block at r * per) and
 * derives has_plan_all[world * per] from the sentinel; both feed the next hdsm_replan_device / hdsm_reference_device
 * directly. Everything is asynchronous on `hip_stream`; no host round trip.                                       */
#define HDSM_COMM_ID_BYTES 128
int hdsm_comm_unique_id(uint8_t id[HDSM_COMM_ID_BYTES]);      /* rank 0 creates it (ncclGetUniqueId), the launcher
                                                                 hands it to every rank                         */
int hdsm_comm_create(void* handle, const uint8_t id[HDSM_COMM_ID_BYTES], int32_t rank, int32_t world, void** comm);
int hdsm_comm_info(void* comm, int32_t* rank, int32_t* world);
void hdsm_comm_destroy(void* comm);
int hdsm_publish_device(void* handle, int32_t per, int32_t n_local, const double* traj, const uint8_t* has_plan_local,
                        double* plans_local, void* hip_stream);
int hdsm_exchange_device(void* comm, int32_t per, const double* plans_local, double* plans_all, uint8_t* has_plan_all,
                         void* hip_stream);

/* Forget the working sets kept for warm_start (e.g. after re-assigning agents to instance indices).      */
int hdsm_reset_warm_start(void* handle);

/* Text of the last error on this thread (HIP error string or argument check that failed).                 */
const char* hdsm_last_error(void);

/* Library/ABI version: (major << 16) | minor. 1.1: hdsm_params grew the execution knobs and time_limit_s;
 * 1.2: + hdsm_poly_octa3d_batch_wave / hdsm_poly_octa3d_device_wave (hdsm_swarm.h), hdsm_set_kernel_timing /
 * hdsm_last_kernel_ms; 1.3: + hdsm_host_register / hdsm_host_unregister, hdsm_swarm_yaw / hdsm_swarm_view
 * (hdsm_swarm.h); nothing removed or changed.                                                                */
int32_t hdsm_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HDSM_H */
