/*
 * hdsm_oracle.h — CPU oracle for the HDSM hot path.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / reported baseline. The product (multi_agent_pkgs_amd/) never links or calls it.
 *
 * PARITY UNPINNED: the reference (lis-epfl/multi_agent_pkgs) holds no tests, golden vectors or fixtures
 * for multi_agent_planner, and its solve runs inside Gurobi 10.0.x (closed source, licence-gated, absent
 * here; README.md:34, multi_agent_planner/CMakeLists.txt:46). agent_class.cpp cannot be compiled in this
 * image (needs rclcpp, PCL, Eigen, Gurobi). This oracle is therefore a restatement of the mathematics in
 * agent_class.cpp, pinned only by (i) solver-independent KKT certificates, (ii) exhaustive enumeration on
 * small instances, (iii) scipy cross-checks and analytic known answers committed under tests/golden/.
 *
 * AC = multi_agent_planner/src/agent_class.cpp of the reference.
 */
#ifndef HDSM_ORACLE_H
#define HDSM_ORACLE_H

#include "../include/hdsm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- dynamics (AC:2115-2167) ------------------------------------------------------------------------ */
/* One integration step of one axis: x = (p, v, a), jerk u, drag D. Literal Euler / RK4 of ModelODE.      */
void orc_step_axis(const hdsm_params* prm, int ax, const double x[3], double u, double x_next[3]);
/* Roll a full trajectory out of (state_curr, controls): traj[N+1][9], literal recursion.                 */
void orc_rollout(const hdsm_params* prm, const double state_curr[9], const double* ctrl /*[N][3]*/,
                 double* traj /*[N+1][9]*/);
/* Literal objective of AC:870-883 + AC:2098 evaluated on a trajectory.                                   */
double orc_objective(const hdsm_params* prm, const double* traj, const double* ctrl,
                     const double* traj_ref /*[N][6]*/);

/* ---- time-aware safe corridor planes (AC:1100-1205, AddHyperplane AC:1217-1234) ---------------------- */
/* One plane between own position c and other position o: out = (n_f[3], n_f . q). Literal libm chain.    */
void orc_tasc_plane(const hdsm_params* prm, const double c[3], const double o[3], double out[4]);
/* All planes of one agent: planes[N][n_rob][4]; valid[N][n_rob] = 1 where a row exists.                  */
void orc_tasc_planes(const hdsm_params* prm, int n_rob, int agent_id, const double state_curr[9],
                     const double* plans_all /*[n_rob][N+1][9]*/, const uint8_t* has_plan,
                     double* planes, uint8_t* valid);

/* ---- MIQP (AC:858-1023) ------------------------------------------------------------------------------ */
typedef struct orc_corridor {
  int32_t m[HDSM_MAX_HOR];                              /* min(P, polyhedra available at step i)          */
  int32_t nrows[HDSM_MAX_HOR][HDSM_MAX_POLY];
  const double* A[HDSM_MAX_HOR][HDSM_MAX_POLY];         /* [rows][3]                                      */
  const double* b[HDSM_MAX_HOR][HDSM_MAX_POLY];         /* [rows]                                         */
  int32_t ncommon[HDSM_MAX_HOR];                        /* rows that hold for p_i and p_{i+1} whatever j  */
  const double* common[HDSM_MAX_HOR];                   /* [ncommon][4] = (n, rhs)                        */
} orc_corridor;

typedef struct orc_result {
  int32_t status;            /* hdsm_status                                                               */
  int32_t nodes, qp_solves, qp_iters;
  int32_t assign[HDSM_MAX_HOR];
  double obj;                /* full objective J incl. constants (what Gurobi's ObjVal would be)          */
  double runner_up;          /* best objective among pruned/other leaves (HDSM_INF if none evaluated)     */
} orc_result;

/* Exact branch-and-bound over the one-hot polyhedron assignment, steps in order, exact dense dual
 * active-set QP at every node. traj[N+1][9], ctrl[N][3], used[P].                                        */
int orc_miqp(const hdsm_params* prm, const double state_curr[9], const double* traj_ref,
             const orc_corridor* cor, double* traj, double* ctrl, uint8_t* used, orc_result* res);
/* Second opinion: literally every assignment (prod m[i] QPs), no pruning. Small instances only.          */
int orc_miqp_enum(const hdsm_params* prm, const double state_curr[9], const double* traj_ref,
                  const orc_corridor* cor, double* traj, double* ctrl, uint8_t* used, orc_result* res);
/* One convex QP for a FIXED assignment (assign[i] in [0,m[i]) or -1 = step unconstrained).                */
int orc_qp_fixed(const hdsm_params* prm, const double state_curr[9], const double* traj_ref,
                 const orc_corridor* cor, const int32_t* assign, double* traj, double* ctrl,
                 orc_result* res);

/* ---- next row f1: reference trajectory (AC:1449-1553, 1591-1663, 1769-1817), same layouts as hdsm_reference -- */
int orc_reference(const hdsm_params* prm, const hdsm_ref_config* cfg, int32_t n_inst, int32_t n_rob,
                  const int32_t* agent_id, const double* path, const int32_t* n_path, int32_t pmax,
                  const double* vel_cap, const double* plans_all, const uint8_t* has_plan, double* ref_full,
                  double* ref, double* path_vel);

/* ---- next row f4: map pre-processing, literal scatter loops (map_builder.cpp:209-216, 331-362; voxel_grid.cpp:
 * 192-297), same layouts as hdsm_map_preprocess ------------------------------------------------------------- */
int orc_map_preprocess(const hdsm_map_config* cfg, int32_t n_grids, const int32_t dim[3], const int8_t* in,
                       int8_t* out);

/* ---- batch entry points with the SAME array layouts as include/hdsm.h -------------------------------- */
/* Level 2 (fused planes + solve): restates hdsm_replan. n_threads > 1 farms instances over pthreads.     */
int orc_replan(const hdsm_params* prm, int32_t n_inst, int32_t n_rob, const int32_t* agent_id,
               const double* state_curr, const double* traj_ref, const int32_t* n_poly,
               const int32_t* n_rows_static, const double* A_static, const double* b_static,
               const double* plans_all, const uint8_t* has_plan, double* traj_out, double* ctrl_out,
               uint8_t* poly_used, int32_t* status, double* obj, int32_t* nodes, int32_t* qp_iters,
               int32_t n_threads);
/* orc_replan with a choice of search order (0 = steps in order, what orc_replan does; 1 = branch on the most infeasible
 * uncontained step) and, optionally, a claimed objective per instance as an initial cut-off — verification of a claimed
 * optimum on trees the step-ordered search cannot finish within its budget; the claim is not trusted (hdsm_oracle.c).
 * obj_hint NULL, or obj_hint[k] NaN / >= HDSM_INF: no hint. */
int orc_replan_ex(const hdsm_params* prm, int32_t n_inst, int32_t n_rob, const int32_t* agent_id,
                      const double* state_curr, const double* traj_ref, const int32_t* n_poly,
                      const int32_t* n_rows_static, const double* A_static, const double* b_static,
                      const double* plans_all, const uint8_t* has_plan, const double* obj_hint, int32_t search,
                      double* traj_out, double* ctrl_out, uint8_t* poly_used, int32_t* status, double* obj,
                      int32_t* nodes, int32_t* qp_iters, int32_t n_threads);
/* Level 1 (fully formed per-step polyhedra, literal: every row is a choice row): restates hdsm_solve.    */
int orc_solve(const hdsm_params* prm, int32_t n_inst, int32_t r_max, const double* state_curr,
              const double* traj_ref, const int32_t* n_poly, const int32_t* n_rows, const double* A,
              const double* b, double* traj_out, double* ctrl_out, uint8_t* poly_used, int32_t* status,
              double* obj, int32_t n_threads);

/* ---- row f2: corridor maintenance, Agent::GenerateSafeCorridor (agent_class.cpp:1236-1447) ---------------------------------
 * A statement-by-statement restatement of the reference method on plain arrays: keep the last polyhedron if the whole previous
 * plan lies in it (AC:1253-1267, LinearConstraint::inside = "no row with A x - b > 0", decomp_geometry/polyhedron.h:130-137), else
 * keep the polyhedra the last solve used (AC:1273-1282); walk the path in steps of voxel_size / 10 (AC:1316-1335) until a sample
 * lies in none of the kept polyhedra (AC:1337-1350); step back one sample, truncate to a voxel (AC:1352-1359), skip it if it is
 * the seed of a kept polyhedron (AC:1361-1379); a seed pinched between two occupied voxels along an axis takes the shape-aware
 * decomposition (AC:1385-1395); the new polyhedron's rows are appended (AC:1403-1435). The voxel decomposition itself
 * (convex_decomp_lib::GetPolyOcta3D / GetPolyOcta3DNew) is NOT restated here: the caller supplies it (`decomp`; the tests hand in
 * the product's host functions hdsm_poly_octa3d[_new], which are pinned separately against recorded outputs of the reference's
 * convex_decomp.cpp). Everything else — control flow, arithmetic, evaluation order — follows the reference text, not the product.
 *
 *   prev_*      poly_const_vec_ / poly_seeds_ before the call: n_prev polyhedra of prev_rows[i] rows (stride rows_max)
 *   poly_used   poly_used_idx_[n_prev]
 *   traj_pts    positions of traj_curr_ (n_traj may be 0: before the first solve the reference's loop body never runs)
 *   path        path_curr with the current position already pushed in front (AC:1286-1290), n_path >= 2
 *   grid        the agent's voxel grid [dim[2]][dim[1]][dim[0]] (x fastest), raw: -1 unknown, 0 free, 100 occupied; origin, voxel_size
 *   decomp      (ctx, seed, grid copy with unknown already occupied — it may mark it, dim, n_it, voxel_size, mark, origin, use_new,
 *                rows[max_rows][4] = (n, n . p), max_rows, &n_rows) -> 0, or an error code that ends the call (returned)
 *   out_*       the new poly_const_vec_ / poly_seeds_: n_out polyhedra, same layout as prev_* (capacity poly_hor)                 */
typedef int (*orc_decomp_fn)(void* ctx, const int32_t seed[3], int8_t* grid, const int32_t dim[3], int32_t n_it, double voxel_size,
                             int32_t mark, const double origin[3], int32_t use_new, double* rows, int32_t max_rows, int32_t* n_rows);
int orc_safe_corridor(int32_t poly_hor, int32_t n_it_decomp, int32_t use_cvx_new, int32_t rows_max, int32_t n_prev,
                      const int32_t* prev_rows, const double* prev_A, const double* prev_b, const double* prev_seed,
                      const uint8_t* poly_used, int32_t n_traj, const double* traj_pts, int32_t n_path, const double* path,
                      const int8_t* grid, const int32_t dim[3], const double origin[3], double voxel_size, orc_decomp_fn decomp,
                      void* ctx, int32_t* n_out, int32_t* out_rows, double* out_A, double* out_b, double* out_seed);

#ifdef __cplusplus
}
#endif
#endif
