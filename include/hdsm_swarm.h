/*
 * hdsm_swarm.h — host-side planner state of a shard of agents, driving hdsm_replan() in closed loop.
 *
 * This is the part of multi_agent_planner::Agent that sits directly around the solve in
 * Agent::TrajPlanningIteration (AC:157-258), restated for a BATCH of agents and for the obstacle-free
 * environments of BASELINE configs 1, 2 and 4 (circle exchange, empty world):
 *
 *   hdsm_swarm_prepare()  = GenerateSafeCorridor (AC:1236-1447, with the free-space polyhedron of
 *                           convex_decomp.cpp:54-373 in closed form: SURVEY.md App. D.2)
 *                         + GenerateReferenceTrajectory (AC:1449-1553; SamplePath AC:1591-1663,
 *                           ComputePathVelocity's neighbour term AC:1769-1801, GetVelocityLimit AC:1805-1817)
 *                         -> the input arrays of hdsm_replan / hdsm_replan_device
 *   hdsm_swarm_commit()   = read-back bookkeeping (poly_used_idx_), the shift-by-one fallback on failure
 *                           (AC:1000-1019), CheckReferenceTrajIncrement (AC:569-585, GetPathProgress
 *                           path_tools.cpp:419-479), state advance (AC:233-238), and the record this shard
 *                           publishes (PublishTrajectoryFull AC:645-677) for the next all-gather.
 *
 * Simplifications, stated: the global path is the straight segment start->goal (what JPS+DMP returns in
 * an empty world up to voxel snapping); the voxel grid is all-free, so the potential
 * field term of ComputePathVelocity is inactive and KeepOnlyFreeReference is the identity.
 * Pure host code; arrays use the layouts of hdsm.h.
 */
#ifndef HDSM_SWARM_H
#define HDSM_SWARM_H

#include <stddef.h>

#include "hdsm.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hdsm_swarm_config {
  double path_vel_min, path_vel_max; /* agent_agile_config.yaml: 4.5 / 9.0                                */
  double sens_dist, sens_pot;        /* 0.05 / 0.18                                                       */
  double sens_other_agents;          /* default 1.0 (AC:2213)                                             */
  double path_vel_dec;               /* 0.0                                                               */
  double thresh_dist;                /* 1.0                                                               */
  double voxel_size;                 /* 0.3                                                               */
  double grid_range[3];              /* local grid extent, 20 x 20 x 6 m                                  */
  double grid_z_min;                 /* z of the local grid origin (ground), 0.0                          */
  int32_t n_it_decomp;               /* 42 -> 7 voxel layers per face                                     */
  int32_t step_plan;                 /* 1                                                                 */
  int32_t use_cvx_new;               /* use_cvx_new_ (AC:2222, shipped: false): always use the shape-aware decomposition;
                                        0 = only where the seed is pinched between occupied voxels (AC:1385-1395)      */
  int32_t reserved0;
} hdsm_swarm_config;

void hdsm_swarm_default_config(hdsm_swarm_config* cfg);

/* Agents [first_id, first_id + n_local) of a swarm of n_rob. starts/goals: [n_local][3] (state_ini, goal). */
int hdsm_swarm_create(const hdsm_params* prm, const hdsm_swarm_config* cfg, int32_t n_rob, int32_t first_id,
                      int32_t n_local, const double* starts, const double* goals, void** swarm);
void hdsm_swarm_destroy(void* swarm);

/* Fills the solver inputs of this round for the n_local agents from the all-gathered plans of last round. */
int hdsm_swarm_prepare(void* swarm, const double* plans_all, const uint8_t* has_plan, int32_t* agent_id,
                       double* state_curr, double* traj_ref, int32_t* n_poly, int32_t* n_rows_static,
                       double* A_static, double* b_static);

/* Consumes the solver outputs; writes the shard's published plans [n_local][N+1][9] and has_plan flags. */
int hdsm_swarm_commit(void* swarm, const double* traj_out, const double* ctrl_out, const uint8_t* poly_used,
                      const int32_t* status, double* plans_local, uint8_t* has_plan_local);

/* f1 on the device: (1) hdsm_swarm_reference_inputs() exports, for every local agent, the polyline
 * GenerateReferenceTrajectory would sample this round (AC:1459-1496: the starting point taken from the previous
 * reference, then the rest of the global path): path[n_local][3][3], n_path[n_local]; (2) the caller runs
 * hdsm_reference / hdsm_reference_device; (3) hdsm_swarm_set_reference() hands the result back — the next
 * hdsm_swarm_prepare() then uses it instead of generating the reference on the host.                        */
int hdsm_swarm_reference_inputs(void* swarm, double* path, int32_t* n_path);
/* vel_cap[n_local] for hdsm_reference*: the voxel / potential-field term of ComputePathVelocity (AC:1709-1766: raycast of the
 * polyline through the agent's local grid, GetVelocityLimit of every voxel crossed) on the world of hdsm_swarm_set_world;
 * path_vel_max in free space. hdsm_swarm_set_reference then applies KeepOnlyFreeReference (AC:1665-1693) to what comes back.
 * (The device-resident loop does both on the device: k_vel_cap, k_keep_free.) */
int hdsm_swarm_vel_cap(void* swarm, double* vel_cap);
int hdsm_swarm_set_reference(void* swarm, const double* ref_full, const double* path_vel);

/* Next row f2: an occupied world. occupancy [dim[2]][dim[1]][dim[0]] int8 (x fastest), voxels of cfg.voxel_size,
 * >= 100 occupied (already inflated by the drone radius — the map builder's job, f4); origin = world position of
 * voxel (0,0,0), a multiple of the voxel size so that the agents' local grids register with it. From then on
 * hdsm_swarm_prepare() cuts each agent's local grid (20 x 20 x 6 m around it) out of this world and builds the
 * corridor polyhedra with hdsm_poly_octa3d instead of the free-space closed form; with an empty world both give
 * the same polyhedra. NULL occupancy = back to free space. The global path stays the straight segment
 * start -> goal (f3, JPS + DMP, is not built): the caller is responsible for worlds in which that is collision-free. */
int hdsm_swarm_set_world(void* swarm, const int8_t* occupancy, const int32_t dim[3], const double origin[3]);

/* Next row f2, first piece — the convex voxel decomposition GenerateSafeCorridor calls for every seed
 * (convex_decomp_lib::GetPolyOcta3D, convex_decomp_util/src/convex_decomp.cpp:5-376): a cuboid of free voxels grown
 * from `seed` face by face (n_it face turns, order -y +x +y -x +z -z), chamfered with integer slopes where obstacles
 * cut an edge.
 *   grid  [dim[2]][dim[1]][dim[0]] int8, x fastest: < 100 free, >= 100 occupied (CVX_DCMP_OCC); voxels taken by the
 *         polyhedron are overwritten with `mark` (the reference's CONV: a negative value, one per polyhedron)
 *   rows  [max_rows][4] = (n, n . p): n . x <= n . p; chamfers first, then the six faces; at most 18 rows
 * Returns HDSM_ERR_CAPACITY (and the needed count in n_rows) if max_rows is too small.                        */
int hdsm_poly_octa3d(const int32_t seed[3], int8_t* grid, const int32_t dim[3], int32_t n_it, double res,
                     int32_t mark, const double origin[3], double* rows, int32_t max_rows, int32_t* n_rows);
/* The shape-aware variant, convex_decomp_lib::GetPolyOcta3DNew (convex_decomp.cpp:590-1160, helpers FindCorners :378-564 and
 * SideIsEmpty :577-588): same growth, but a chamfer only starts where an obstacle really lies behind it, a layer that
 * covers less than half of its allowance is skipped, and layers may reach the last voxel of the grid. GenerateSafeCorridor
 * switches to it when the seed is pinched between two occupied voxels along an axis (AC:1385-1395). Same arguments; voxels
 * with a positive value below 100 (potential field) are free for the growth but count as "not empty" for the chamfer test. */
int hdsm_poly_octa3d_new(const int32_t seed[3], int8_t* grid, const int32_t dim[3], int32_t n_it, double res,
                         int32_t mark, const double origin[3], double* rows, int32_t max_rows, int32_t* n_rows);

/* Row f2 on the device: a BATCH of decompositions on local grids that are windows into one world grid (csrc/corridor_kernels.hip;
 * one thread per seed, the same source as the two host functions above, rows bit-identical to theirs). For seed t:
 *   off[t][3]      local voxel (0,0,0) in world voxels; ldim = dimensions of every local grid
 *   ground_k[t]    local voxels with k < ground_k are unknown -> occupied (AC:1302, 1307); unknown (negative) world voxels are
 *                  occupied, voxels outside the world are free (what hdsm_swarm_set_world's host path does)
 *   seed[t][3]     local voxel; variant[t] 0 = GetPolyOcta3D, 1 = GetPolyOcta3DNew, -1 = decide like AC:1385-1395
 *   origin[t][3]   world position of local voxel (0,0,0)
 *   rows[t][max_rows][4], n_rows[t], rc[t] (hdsm_error per seed), cells[t] (voxels of the polyhedron; may be NULL)
 * hdsm_poly_octa3d_batch: host pointers (copies the world in, PCIe-inclusive). hdsm_poly_octa3d_device: device pointers,
 * asynchronous on hip_stream, `scratch` = hdsm_poly_octa3d_scratch_bytes(n) bytes of device memory. */
int hdsm_poly_octa3d_batch(int32_t device, int32_t n, const int8_t* world, const int32_t wdim[3], const int32_t ldim[3],
                           const int32_t* off, const int32_t* ground_k, const int32_t* seed, const int32_t* variant,
                           const double* origin, int32_t n_it, double res, double* rows, int32_t max_rows, int32_t* n_rows,
                           int32_t* rc, int32_t* cells);
int hdsm_poly_octa3d_device(int32_t device, int32_t n, const int8_t* world, const int32_t wdim[3], const int32_t ldim[3],
                            const int32_t* off, const int32_t* ground_k, const int32_t* seed, const int32_t* variant,
                            const double* origin, int32_t n_it, double res, double* rows, int32_t max_rows, int32_t* n_rows,
                            int32_t* rc, int32_t* cells, void* scratch, void* hip_stream);
size_t hdsm_poly_octa3d_scratch_bytes(int32_t n);
/* The same batch with ONE WAVEFRONT per seed (workspace in LDS, the 64 lanes run the decomposition cooperatively — the form the
 * device-resident loop uses for one agent's seeds): lower latency per seed, fewer seeds in flight, no scratch; same results. */
int hdsm_poly_octa3d_batch_wave(int32_t device, int32_t n, const int8_t* world, const int32_t wdim[3], const int32_t ldim[3],
                                const int32_t* off, const int32_t* ground_k, const int32_t* seed, const int32_t* variant,
                                const double* origin, int32_t n_it, double res, double* rows, int32_t max_rows, int32_t* n_rows,
                                int32_t* rc, int32_t* cells);
int hdsm_poly_octa3d_device_wave(int32_t device, int32_t n, const int8_t* world, const int32_t wdim[3], const int32_t ldim[3],
                                 const int32_t* off, const int32_t* ground_k, const int32_t* seed, const int32_t* variant,
                                 const double* origin, int32_t n_it, double res, double* rows, int32_t max_rows, int32_t* n_rows,
                                 int32_t* rc, int32_t* cells, void* hip_stream);
const char* hdsm_corridor_last_error(void);

/* Global paths (path_curr_ of the reference, produced there by the path thread: JPS + DMP + shortening, AC:261-567 — out of
 * scope as such). Default: the straight segment start -> goal. hdsm_swarm_set_paths installs caller-supplied polylines
 * [n_local][pmax][3] with n_path[k] >= 2 points each (first = start, last = goal). hdsm_swarm_route computes them on the
 * world given to hdsm_swarm_set_world: a minimal collision-free router (3-D A* over free voxels, 26-connected, extra cost next
 * to obstacles, then greedy line-of-sight shortening) — NOT the reference's JPS3D / distance-map planner, only something
 * that lets BASELINE's forest configurations fly. hdsm_swarm_get_paths reads the current paths back (n_path > pmax -> CAPACITY).
 * Both corridor generation (AC:1286-1290) and reference sampling (AC:1459-1496) then walk these polylines. */
int hdsm_swarm_set_paths(void* swarm, const double* paths, const int32_t* n_path, int32_t pmax);
int hdsm_swarm_route(void* swarm, int32_t* n_failed);
int hdsm_swarm_get_paths(void* swarm, int32_t pmax, double* paths, int32_t* n_path);
/* hdsm_swarm_reference_inputs for paths with more than two points: path[n_local][pmax][3]; a polyline longer than pmax is
 * cut after pmax points, which changes nothing as long as the kept part is longer than n_hor * path_vel_max * dt (checked:
 * otherwise HDSM_ERR_CAPACITY). */
int hdsm_swarm_reference_inputs_n(void* swarm, int32_t pmax, double* path, int32_t* n_path);
/* Number of local agents whose corridor generation failed in the last hdsm_swarm_prepare (seed outside the local grid, or a
 * polyhedron with more rows than max_rows_static); codes[n_local] (may be NULL) receives the hdsm_error per agent. Those
 * agents kept the polyhedra they had. */
int hdsm_swarm_corridor_errors(void* swarm, int32_t* codes);

/* GenerateSafeCorridor alone (AC:165). With the reference generated elsewhere (hdsm_reference*, row f1) the reference's own order
 * is: corridor from the PREVIOUS reference, then the new reference, then the solve — call this first, then
 * hdsm_swarm_reference_inputs* / hdsm_reference* / hdsm_swarm_set_reference, then hdsm_swarm_prepare (which then skips the corridor). */
int hdsm_swarm_prepare_corridor(void* swarm);

/* ---- the device-resident closed loop ------------------------------------------------------------------------------------------
 * The planner state of the shard moves into HBM (hdsm_dswarm_create copies it out of a host mirror that has been set up —
 * starts, goals, world, paths — and possibly flown for some rounds) and one replan round becomes a chain of launches on one
 * stream with no host round trip (csrc/swarm_kernels.hip):
 *   corridor (AC:1236-1447, row f2 on the device) -> reference (row f1) -> solver inputs -> hdsm_replan_device -> commit
 *   (AC:960-1019, 569-585, 233-238) -> published records -> ONE RCCL all-gather (hdsm_exchange_device; a copy on a single rank).
 * `solver` is an hdsm handle with max_instances >= the shard and n_rob_max >= world_size * ceil(n_rob / world_size); `comm` an
 * hdsm_comm (NULL when world_size == 1). hdsm_dswarm_round is asynchronous on hip_stream. hdsm_dswarm_download synchronises
 * and copies out what the caller asks for (any pointer may be NULL): the agent states back into the host mirror `swarm`
 * (so that every hdsm_swarm_* diagnostic works on them), the all-gathered plans [world_size * per][N+1][9] and flags, the
 * statuses of the last round, the number of instances without solution so far.
 * The world and the configuration are those of the host mirror at hdsm_dswarm_create and stay fixed for the dswarm's life (the
 * device corridor keeps each agent's last polyhedra and forms the rows of one that is asked for again from them instead of
 * growing it again — same rows, bit for bit; environment HDSM_POLY_CACHE=0 switches that off, for A/B runs). */
int hdsm_dswarm_create(void* swarm, void* solver, int32_t device, int32_t world_size, void** dswarm);
/* the all-gathered plans [n_rob][N+1][9] and flags [n_rob] the next round starts from (a swarm taken over in mid-flight) */
int hdsm_dswarm_upload_plans(void* dswarm, const double* plans_all, const uint8_t* has_plan);
int hdsm_dswarm_round(void* dswarm, void* comm, void* hip_stream);
int hdsm_dswarm_download(void* dswarm, void* swarm, double* plans_all, uint8_t* has_plan, int32_t* status, int32_t* failed_total);
void hdsm_dswarm_destroy(void* dswarm);
const char* hdsm_dswarm_last_error(void);
/* Where a round goes (a measurement aid; the reference books the same intervals per agent, AC:165-227 comp_time_sc_ / _tasc_ / _opt_):
 * with timing on, hdsm_dswarm_round records HIP events on its stream between its launches; hdsm_dswarm_last_phase_ms waits for the
 * last timed round and returns milliseconds of [0] k_corridor (GenerateSafeCorridor, AC:1236-1447), [1] k_vel_cap (ComputePathVelocity's
 * voxel term, AC:1709-1766), [2] hdsm_reference_device (AC:1449-1553), [3] k_keep_free (AC:1665-1693), [4] hdsm_replan_device
 * (AC:1086-1215 + 858-1023), [5] k_commit (AC:955-1019, 569-585, 233-238), [6] the exchange (AC:610-677; one rank: nothing).
 * Every record is a barrier packet in front of the next launch: a timed round is a few microseconds longer than a plain one. */
int hdsm_dswarm_set_phase_timing(void* dswarm, int32_t on);
int hdsm_dswarm_last_phase_ms(void* dswarm, float ms[7]);
/* The device corridor's polyhedron cache since hdsm_dswarm_create, summed over the shard: out[0] polyhedra asked for, out[1] formed
 * from a structure recorded in the same local grid, out[2] from one recorded in another grid at the same height (the interior
 * rule), out[3] 1 if the cache is on (a world is set and HDSM_POLY_CACHE is not 0). Synchronises the device. */
int hdsm_dswarm_cache_stats(void* dswarm, int64_t out[4]);

/* Next row f3 (ROS-free half): every local agent keeps the records of Agent::TrajPlanningIteration — comp_time_sc_ (CPU time of
 * its corridor generation), comp_time_opt_ (the duration of the fused launch, handed in with hdsm_swarm_record_solve_ms between
 * prepare and commit; comp_time_tasc_ = 0 because the planes are generated inside that launch), comp_time_tot_,
 * comp_time_tot_wall_ and state_hist_. hdsm_swarm_shutdown = Agent::OnShutdown (AC:2446-2466) for one local agent: the CSV
 * files of hdsm_stats.h in `dir` (when save_stats) and the printed report. */
int hdsm_swarm_record_solve_ms(void* swarm, double milliseconds);
int hdsm_swarm_shutdown(void* swarm, int32_t local_index, const char* dir, int32_t save_stats, char* report, int32_t report_cap);

/* Diagnostics: current positions [n_local][3], distance to goal [n_local], failures so far. */
int hdsm_swarm_state(void* swarm, double* pos, double* dist_goal, int32_t* n_fail);

/* Agent::ComputeYawAngle (AC:1025-1051) for every local agent: the yaw follows the direction from the current position to
 * reference point `yaw_idx` (projected on the x-y plane) with a P controller, yaw += k_p_yaw * error * dt, error wrapped into
 * (-pi, pi]; nothing moves while that point is closer than sqrt(0.1) m. Call it where the reference does (AC:177): after the
 * solve, BEFORE hdsm_swarm_commit advances the state. yaw_out [n_local] = Trajectory.msg:11 of the plans published this round. */
int hdsm_swarm_yaw(void* swarm, int32_t yaw_idx, double k_p_yaw, double* yaw_out);

/* What the reference's rviz publishers show of local agent `k` (AC:679-857): traj_curr_ positions [n_traj <= N + 1][3],
 * traj_ref_curr_ positions [n_ref <= N + 1][3], path_curr_ [n_path <= pmax][3], the corridor (poly_const_vec_ rows n . x <= b with
 * the capacity of hdsm_params: poly_A [poly_hor][max_rows_static][3], poly_b, poly_rows[poly_hor]; poly_seeds_ [poly_hor][3]),
 * the current position. Any output pointer may be NULL. */
int hdsm_swarm_view(void* swarm, int32_t k, double* traj_curr, int32_t* n_traj, double* traj_ref, int32_t* n_ref, double* path, int32_t pmax,
                    int32_t* n_path, int32_t* n_poly, int32_t* poly_rows, double* poly_A, double* poly_b, double* poly_seeds, double pos[3]);

#ifdef __cplusplus
}
#endif
#endif
