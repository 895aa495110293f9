// swarm_kernels.hip — the device-resident closed loop: the planner state of a shard of agents (hdsm_sw::AgentS) lives in HBM and
// one replan round is a chain of launches on ONE stream with no host round trip:
//   k_corridor   GenerateSafeCorridor (AC:1236-1447; row f2: voxel decomposition on a window of the world grid) + the polyline
//                of this round's reference (AC:1459-1496) + the solver inputs that do not depend on the reference (id, state,
//                corridor rows in the layouts of hdsm.h)                                          one wavefront per agent
//   k_reference  GenerateReferenceTrajectory's neighbour speed term + SamplePath (row f1)       hdsm_reference_device
//   k_replan     planes + MIQP (the hot path)                                                   hdsm_replan_device
//   k_commit     the new reference into the agent state, read-back, shift fallback, increment check, state advance, published
//                record                                                                          one wavefront per agent
//   exchange     ONE RCCL all-gather (hdsm_exchange_device), or a local copy on a single rank
// The per-agent functions are the SAME source as the host mirror (swarm_core.h), which is how the two loops are compared in
// tests/test_gpu_configs.py. AC = multi_agent_planner/src/agent_class.cpp of the reference.
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdlib>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/hdsm.h"
#include "../../include/hdsm_swarm.h"
#include "swarm_core.h"

// (csrc/hdsm_api.hip, internal: see hdsm_dswarm_round)
extern "C" int hdsm_internal_defer_done(void* handle, int on);
extern "C" int hdsm_internal_record_done(void* handle, void* hip_stream);

#ifdef CD_PROFILE
// development builds only: the phase counters of the corridor kernel's decompositions (read and cleared); [12] cycles of the whole
// corridor step, [13] of its decompositions, [14] agent-rounds, [15] decompositions. (The polyhedron cache's counters are NOT in here:
// hdsm_dswarm_cache_stats, every build.)
extern "C" int hdsm_swarm_corridor_profile(unsigned long long out[16]) {
  if (hipDeviceSynchronize() != hipSuccess) return HDSM_ERR_DEVICE;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(hdsm_cd::g_cd_prof), 16 * sizeof(unsigned long long)) != hipSuccess) return HDSM_ERR_DEVICE;
  unsigned long long zero[16] = {0};
  if (hipMemcpyToSymbol(HIP_SYMBOL(hdsm_cd::g_cd_prof), zero, sizeof zero) != hipSuccess) return HDSM_ERR_DEVICE;
  return HDSM_OK;
}
#endif

extern "C" int hdsm_swarm_export_state(void* swarm, void* agents_out, int32_t* n_local, int32_t* n_rob, int32_t* first_id,
                                       hdsm_params* prm, hdsm_swarm_config* cfg, const int8_t** world, int32_t wdim[3],
                                       double worigin[3]);
extern "C" int hdsm_swarm_import_state(void* swarm, const void* agents_in, int32_t n_local);

namespace {

using hdsm_sw::AgentS;
using hdsm_sw::Cfg;
using hdsm_sw::V3;

thread_local std::string g_err;
int fail(int code, const std::string& m) {
  g_err = m;
  return code;
}
#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) return fail(HDSM_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

constexpr int PTS = hdsm_sw::PATH_PTS + 1;  // points of a reference polyline handed to k_reference
// scratch of one agent's voxel decompositions, in LDS (dynamic shared memory of k_corridor): the workspace, the overlay bits and
// the 2-bit cache of the world under the overlay. The decomposition is one lane's chain of small dependent accesses — in global
// memory every one of them was a round trip (33 ms per round for 256 agents in the pillar forest).
constexpr size_t SLAB = hdsm_cd::WAVE_LDS_MAX;  // (the launch asks for what its n_it needs: wave_lds_bytes)
// k_corridor also has static __shared__ state (the walk's rows and flags, < 4 KB); together they must stay inside the 64 KB a
// kernel may use without hipFuncAttributeMaxDynamicSharedMemorySize — a growth of Work / the overlay fails HERE, not at launch
static_assert(SLAB + 4096 <= 64 * 1024, "k_corridor: dynamic + static LDS exceed the default 64 KB limit");

// The polyhedra an agent's last decompositions produced. The corridor keeps only the polyhedra the last plan used, so the ones
// further down the path are dropped and asked for again round after round with the same seed voxel — more than half of all
// decompositions of a forest flight. What a decomposition yields depends on the (constant) world and configuration, on the seed,
// and on the local grid only where the growth meets the grid's border or the ground plane moves; so an entry holds the
// polyhedron as INTEGERS relative to the seed (hdsm_cd::PolyStruct) with the seed's world voxel, the grid's offset in the world,
// the ground plane's world level and whether everything a decomposition can look at (the seed +- wave_map_radius) lay inside the
// grid's interior in x and y. A request for the same world voxel from a grid at the same height with that property too (or from
// the same grid) gets its rows
// formed from the entry — with the arithmetic a decomposition in ITS grid would use, origin included (rows_from_structure), so the
// host mirror, which grows the polyhedron again, gets the same bits.
constexpr int CACHE_POLYS = 6;
struct PolyCache {
  hdsm_cd::PolyStruct ps[CACHE_POLYS];
  int32_t seed_w[CACHE_POLYS][3], off[CACHE_POLYS][3], ground_w[CACHE_POLYS], interior[CACHE_POLYS];
  int32_t n, next;
  // what the cache did for this agent (hdsm_dswarm_cache_stats; written by lane 0 of the agent's own wavefront): polyhedra asked
  // for, found with the same grid, found through the interior rule (another grid at the same height)
  int32_t asked, hits_same_grid, hits_interior, pad_;
};

// GenerateSafeCorridor (AC:1236-1447), ONE WAVEFRONT PER AGENT. The walk along the path (steps of voxel / 10: hundreds of
// them per round) tests every sample against every row of the kept polyhedra; with one thread per agent each test was a chain of
// global loads (1.3 ms per round for 1024 agents). Here the rows live in the registers of the 64 lanes (two rows per lane,
// P * RS <= 128), a sample is tested against all of them at once and `inside` is a ballot; the walk state is computed redundantly
// by every lane (wave-uniform), so the arithmetic — and the result — is exactly that of hdsm_sw::corridor_step. New polyhedra:
// the closed form in free space (lane 0), the voxel decomposition on a window of the world grid by the whole wavefront
// (corridor_wave.h: the world under the overlay classified into bit maps in LDS, layers grown as bit planes).

// min over the 64 lanes of a double, in every lane: four DPP row rotations + four v_readlane (six __shfl_xor stages are twelve
// ds_bpermute round trips)
__device__ __forceinline__ double wave_min_f64(double v) {
  auto rot = [](double x, auto ctrl) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), decltype(ctrl)::value, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), decltype(ctrl)::value, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
  };
  v = fmin(v, rot(v, std::integral_constant<int, 0x121>{}));  // row_ror:1
  v = fmin(v, rot(v, std::integral_constant<int, 0x122>{}));  // row_ror:2
  v = fmin(v, rot(v, std::integral_constant<int, 0x124>{}));  // row_ror:4
  v = fmin(v, rot(v, std::integral_constant<int, 0x128>{}));  // row_ror:8
  auto lane_of = [](double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
  };
  return fmin(fmin(lane_of(v, 0), lane_of(v, 16)), fmin(lane_of(v, 32), lane_of(v, 48)));
}

// (inlined into the kernel: only there does the compiler know that the workspace is LDS — behind a call every access of the
// decomposition was a flat load)
__device__ __forceinline__ void corridor_step_wave(const Cfg& c, AgentS& ag, const hdsm_cd::WaveLds& lds, V3* path, int lane, PolyCache* pc) {
  using namespace hdsm_sw;
  const int P = c.P, N = c.N, RS = c.RS;
  __shared__ int sh_npath;
  // keep-last / keep-used (AC:1253-1282), by the whole wavefront: lane j tests point j of the current plan against the last
  // polyhedron (the verdict is a ballot — the serial loop's early exit does not change it) and a polyhedron that moves to another
  // slot is copied by all lanes (1 KB: one lane copying it was a chain of a hundred global accesses per round)
  static_assert(sizeof(Poly) % 8 == 0, "copied as 8-byte words");
  auto copy_poly = [&](Poly* dst, const Poly* src) {
    const unsigned long long* sw = reinterpret_cast<const unsigned long long*>(src);
    unsigned long long* dw = reinterpret_cast<unsigned long long*>(dst);
    for (int e = lane; e < (int)(sizeof(Poly) / 8); e += 64) dw[e] = sw[e];
  };
  int n_poly = 0;
  {
    const int np0 = ag.n_poly;
    bool kept_last = false;
    if (np0 > 0) {
      bool ok = true;
      if (ag.has_traj && lane <= N) ok = inside(ag.polys[np0 - 1], V3{{ag.traj_curr[lane][0], ag.traj_curr[lane][1], ag.traj_curr[lane][2]}});
      if (__ballot(!ok) == 0ull) {
        if (np0 - 1 != 0) copy_poly(&ag.polys[0], &ag.polys[np0 - 1]);
        n_poly = 1, kept_last = true;
      }
    }
    if (np0 > 0 && !kept_last)
      for (int i = 0; i < P && i < np0; ++i)
        if (ag.poly_used[i]) {
          if (n_poly != i) copy_poly(&ag.polys[n_poly], &ag.polys[i]);
          ++n_poly;
        }
  }
  if (lane == 0) {  // the path ahead (AC:1286-1290): a handful of operations
    ag.corridor_rc = 0;
    const V3 path_head = ag.n_ref == 0 ? ag.path[0] : V3{{ag.traj_ref[0][0], ag.traj_ref[0][1], ag.traj_ref[0][2]}};
    path[0] = {{ag.state_curr[0], ag.state_curr[1], ag.state_curr[2]}};
    sh_npath = 1 + path_ahead(ag, path_head, path + 1);
  }
  __syncthreads();
  const int n_path = sh_npath;
  const double vs = c.voxel_size;
  V3 origin;
  for (int ax = 0; ax < 3; ++ax) origin[ax] = floor((ag.state_curr[ax] - c.grid_range[ax] / 2) / vs) * vs;
  // rows lane and lane + 64 of the flattened [P][RS] table
  double ra[2][4];
  bool rv[2];
  int pj[2];  // the polyhedron the row belongs to (-1: no row)
  auto load_rows = [&]() {
    for (int h = 0; h < 2; ++h) {
      const int lin = lane + 64 * h, j = lin / RS, r = lin % RS;
      rv[h] = j < n_poly && r < ag.polys[j].rows;
      pj[h] = rv[h] ? j : -1;
      for (int q = 0; q < 3; ++q) ra[h][q] = rv[h] ? ag.polys[j].A[r][q] : 0.0;
      ra[h][3] = rv[h] ? ag.polys[j].b[r] : 0.0;
    }
  };
  load_rows();
  int path_idx = 1;
  V3 curr = path[0];
  V3 next = path[1];
  const double samp = vs / 10;  // AC:1316
  // (the shortcut below is pure bookkeeping — it never changes a sample — so it is not tried again where it has just found nothing
  // to skip: same polyhedron, same path segment, closer to the exit than its margin. Computing the exit distance for each of the
  // last samples before an exit was most of this kernel's time in free space.)
  int hold_j = -1, hold_seg = -1;
  while (n_poly < P) {
    const V3 diff = sub(next, curr);
    const double dist_next = norm(diff);
    if (dist_next > samp) {
      curr = step_along(curr, samp, diff, dist_next);
    } else {
      curr = next;
      if (++path_idx == n_path) break;
      next = path[path_idx];
    }
    // inside at least one kept polyhedron? (LinearConstraint::inside: no row with A x - b > 0)
    bool bad[2];
    for (int h = 0; h < 2; ++h) bad[h] = rv[h] && (((ra[h][0] * curr[0] + ra[h][1] * curr[1]) + ra[h][2] * curr[2]) - ra[h][3] > 0);
    int j_in = -1;
    for (int j = 0; j < n_poly; ++j)  // (one ballot per kept polyhedron: the first one without a violated row)
      if (__ballot((bad[0] && pj[0] == j) || (bad[1] && pj[1] == j)) == 0ull) {
        j_in = j;
        break;
      }
    if (j_in >= 0) {
      // The sample lies in polyhedron j_in. While the walk keeps its direction, every sample closer than the exit distance of the
      // ray from that polyhedron is inside it too and the reference loop would just `continue`: those samples are generated
      // (same statements, same rounding) without being tested. The exit distance is a min over the rows held by the lanes; three
      // samples of margin cover the rounding of the accumulated positions.
      if (c.fast_walk && dist_next > samp && !(j_in == hold_j && path_idx == hold_seg)) {
        double t_exit = DBL_MAX;
        for (int h = 0; h < 2; ++h)
          if (pj[h] == j_in) {
            const double rate = ((ra[h][0] * diff[0] + ra[h][1] * diff[1]) + ra[h][2] * diff[2]) / dist_next;
            const double slack = ra[h][3] - ((ra[h][0] * curr[0] + ra[h][1] * curr[1]) + ra[h][2] * curr[2]);
            // every skipped sample keeps a slack >= kWalkTol in every row, far above the rounding of A x - b: a path
            // that slides along a face (slack ~ 0, rate ~ 0) is left to the regular loop, whose outcome there depends
            // on the last bit exactly as the reference's does
            if (!(slack > kWalkTol)) t_exit = 0;
            else if (rate > 0) t_exit = fmin(t_exit, (slack - kWalkTol) / rate);
          }
        t_exit = wave_min_f64(t_exit);
        const double cap = t_exit / samp - 3.0;
        int n_safe = cap > 1e6 ? 1000000 : (cap > 0 ? (int)cap : 0);
        hold_j = n_safe == 0 ? j_in : -1, hold_seg = path_idx;
        if (!c.has_world) {
          // free space: the skipped samples are not even generated one by one (walk_jump). Only where the polyhedra are the
          // large boxes of an empty grid: next to obstacles a routed path slides along faces, the outcome of the first
          // TESTED sample after the shortcut can hang on the last bit of the position, and the sample-by-sample form below
          // keeps that bit what the reference's loop produces.
          if (n_safe > 0) walk_jump(curr, next, samp, n_safe);
          n_safe = 0;
        }
        for (; n_safe > 0; --n_safe) {
          const V3 df = sub(next, curr);
          const double dn = norm(df);
          if (!(dn > samp)) break;  // the end of the segment: the regular loop takes over
          curr = step_along(curr, samp, df, dn);
        }
      }
      continue;
    }
    V3 seed_pt = curr;  // AC:1351-1354: step back to the previous sample
    if (dist_next > 0) seed_pt = step_along(curr, -fmin(samp, dist_next), diff, dist_next);
    int seed[3];
    V3 seed_world;
    for (int ax = 0; ax < 3; ++ax) {
      seed[ax] = (int)((seed_pt[ax] - origin[ax]) / vs);
      seed_world[ax] = (seed[ax] * vs + vs / 2) + origin[ax];
    }
    bool previous_seed = false;  // AC:1361-1379
    for (int i = 0; i < n_poly; ++i)
      if (ag.polys[i].seed[0] == seed_world[0] && ag.polys[i].seed[1] == seed_world[1] && ag.polys[i].seed[2] == seed_world[2]) {
        previous_seed = true;
        break;
      }
    if (previous_seed) continue;
    int rc = HDSM_OK;
    if (c.has_world) {
      // (the local grid of this agent in world voxels, as make_window forms it)
      const hdsm_cd::WindowGrid wg = hdsm_sw::make_window(c, origin, seed, nullptr);
      const int seed_w[3] = {seed[0] + wg.ox, seed[1] + wg.oy, seed[2] + wg.oz}, off_w[3] = {wg.ox, wg.oy, wg.oz}, ground_w = wg.ground_k + wg.oz;
      const int rad = hdsm_cd::wave_map_radius(c.n_it_decomp);
      // (in x and y: the local grid is 66 voxels wide there, 20 in z — in z the growth does meet the border, so a request must come
      // from a grid at the same height; an agent changes its z voxel rarely, its x / y voxel every other round)
      const bool interior = rad > 0 && hdsm_sw::seed_in_grid(c, seed) && seed[0] - rad >= 1 && seed[1] - rad >= 1 && seed[0] + rad <= wg.lnx - 2 &&
                            seed[1] + rad <= wg.lny - 2;
      int hit = -1;
      bool hit_same_grid = false;
      if (pc != nullptr)
        for (int k = 0; k < pc->n && hit < 0; ++k) {
          const bool same_voxel = pc->seed_w[k][0] == seed_w[0] && pc->seed_w[k][1] == seed_w[1] && pc->seed_w[k][2] == seed_w[2];
          const bool same_grid = pc->off[k][0] == off_w[0] && pc->off[k][1] == off_w[1] && pc->off[k][2] == off_w[2];
          if (same_voxel && (same_grid || (interior && pc->interior[k] != 0 && pc->off[k][2] == off_w[2] && pc->ground_w[k] == ground_w))) hit = k, hit_same_grid = same_grid;
        }
      if (lane == 0 && pc != nullptr) {
        ++pc->asked;
        if (hit >= 0) ++(hit_same_grid ? pc->hits_same_grid : pc->hits_interior);
      }
      if (hit >= 0) {  // grown before from this voxel: the rows from the integers, in this grid's arithmetic
        const double org[3] = {origin[0], origin[1], origin[2]};
        const int cap = c.RS < HDSM_MAX_ROWS_STATIC ? c.RS : HDSM_MAX_ROWS_STATIC;
        const int n = hdsm_cd::rows_from_structure(pc->ps[hit], hdsm_cd::Cell{seed[0], seed[1], seed[2]}, c.voxel_size, org, lds.rows, cap);
        __syncthreads();
        if (n > cap) {
          rc = HDSM_ERR_CAPACITY, ag.polys[n_poly].rows = 0, ag.corridor_rc = rc;
        } else {
          Poly* out = &ag.polys[n_poly];
          if (lane == 0) out->rows = n;
          for (int t = lane; t < 4 * n; t += 64) {
            const int r = t >> 2, q = t & 3;
            if (q < 3) out->A[r][q] = lds.rows[t];
            else out->b[r] = lds.rows[t];
          }
          out->seed = seed_world;
        }
      } else {  // the whole wavefront, cooperatively: same arguments, same result in every lane
#if defined(CD_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
        const unsigned long long tp0 = __builtin_readcyclecounter();
#endif
        rc = world_poly_wave(c, origin, seed, lds, &ag.polys[n_poly], lane);
#if defined(CD_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
        if (lane == 0) atomicAdd(&hdsm_cd::g_cd_prof[13], __builtin_readcyclecounter() - tp0);
#endif
        if (rc != HDSM_OK) {
          ag.corridor_rc = rc;
        } else {
          ag.polys[n_poly].seed = seed_world;
          if (pc != nullptr && rad > 0) {  // (rad = 0: the plain form ran; it keeps no chamfer sources)
            const int slot = pc->next;
            hdsm_cd::wave_poly_structure(lds, hdsm_cd::Cell{seed[0], seed[1], seed[2]}, &pc->ps[slot], lane);
            if (lane < 3) pc->seed_w[slot][lane] = seed_w[lane], pc->off[slot][lane] = off_w[lane];
            __syncthreads();
            if (lane == 0) {
              pc->ground_w[slot] = ground_w, pc->interior[slot] = interior ? 1 : 0;
              pc->next = (slot + 1) % CACHE_POLYS, pc->n = pc->n < CACHE_POLYS ? pc->n + 1 : CACHE_POLYS;
            }
          }
        }
      }
    } else if (lane == 0) {
      free_space_poly(c, origin, seed, &ag.polys[n_poly]);
      ag.polys[n_poly].seed = seed_world;
    }
    __syncthreads();
    if (rc != HDSM_OK) break;
    ++n_poly;
    load_rows();
    hold_j = -1;
  }
  if (lane == 0) ag.n_poly = n_poly;
}

__device__ __attribute__((noinline)) void corridor_step_plain(const Cfg& c, AgentS& ag, hdsm_cd::Work* wk, uint32_t* bits) {
  hdsm_sw::corridor_step(c, ag, wk, bits);
}

// (the solver's inputs that do not depend on the reference — id, state, the corridor just built — leave with this kernel: a kernel
// of their own cost 5 us per round in the live loop)
__global__ __launch_bounds__(64) void k_corridor(Cfg c, int n, AgentS* agents, double* path, int32_t* n_path, int32_t* agent_id,
                                                 double* state_curr, int32_t* n_poly, int32_t* n_rows, double* A, double* b, PolyCache* cache) {
  __shared__ V3 path_s[hdsm_sw::PATH_PTS + 2];
  __shared__ V3 poly_s[PTS];
  __shared__ int np_s;
  const int k = (int)blockIdx.x, lane = (int)threadIdx.x;
  if (k >= n) return;
  AgentS& ag = agents[k];
  extern __shared__ __attribute__((aligned(16))) unsigned char slab[];  // SLAB bytes when there is a world, else none
  const hdsm_cd::WaveLds lds(slab);
  if (c.P * c.RS <= 128) {
#if defined(CD_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long tp0 = __builtin_readcyclecounter();
#endif
    corridor_step_wave(c, ag, lds, path_s, lane, cache != nullptr ? cache + k : nullptr);
#if defined(CD_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    if (lane == 0) atomicAdd(&hdsm_cd::g_cd_prof[12], __builtin_readcyclecounter() - tp0), atomicAdd(&hdsm_cd::g_cd_prof[14], 1ull);
#endif
  } else if (lane == 0) {
    const Cfg c_call = c;  // (a copy for the call: the kernel's own stays in registers)
    corridor_step_plain(c_call, ag, lds.wk, lds.bits);  // more rows than two per lane: the plain per-agent code
  }
  __syncthreads();
  if (lane == 0) np_s = hdsm_sw::reference_polyline(ag, poly_s);
  __syncthreads();
  const int np = np_s;
  if (lane == 0) n_path[k] = np;
  for (int i = lane; i < PTS * 3; i += 64) path[(size_t)k * PTS * 3 + i] = poly_s[(i / 3) < np ? i / 3 : np - 1][i % 3];
  const int P = c.P, RS = c.RS, npo = ag.n_poly;
  if (lane < 9) state_curr[9 * (size_t)k + lane] = ag.state_curr[lane];
  if (lane == 0) agent_id[k] = ag.id, n_poly[k] = npo;
  if (lane < P) n_rows[(size_t)k * P + lane] = lane < npo ? ag.polys[lane].rows : 0;
  for (int t = lane; t < P * RS; t += 64) {
    const int j = t / RS, r = t % RS;
    const bool hr = j < npo && r < ag.polys[j].rows;
    for (int q = 0; q < 3; ++q) A[((size_t)k * P * RS + t) * 3 + q] = hr ? ag.polys[j].A[r][q] : 0.0;
    b[(size_t)k * P * RS + t] = hr ? ag.polys[j].b[r] : 0.0;
  }
}

// the map-dependent half of the reference (row f1 remainder): ComputePathVelocity's voxel term before k_reference ...
__global__ __launch_bounds__(64) void k_vel_cap(Cfg c, hdsm_ref_config rc, int n, const AgentS* agents, const double* path, const int32_t* n_path,
                                                double* vel_cap) {
  // one wavefront per agent, one path segment per lane (two ray casts each): the reference stops at the first segment that collides,
  // i.e. the cap is the minimum over the segments up to and including that one
  const int k = (int)blockIdx.x, lane = (int)threadIdx.x;
  if (k >= n) return;
  const int np = n_path[k];
  double cap = rc.path_vel_max;
  if (c.has_world && np >= 1) {
    const V3 origin = hdsm_sw::local_grid_origin(c, agents[k]);
    const hdsm_sw::RawWindow g = hdsm_sw::raw_window(c, origin);
    auto pt = [&](int i) { return V3{{path[((size_t)k * PTS + i) * 3], path[((size_t)k * PTS + i) * 3 + 1], path[((size_t)k * PTS + i) * 3 + 2]}}; };
    const V3 p0 = pt(0);
    for (int base = 0; base + 1 < np; base += 64) {  // (PTS <= 64 segments in practice: one trip)
      const int i = base + lane;
      bool collided = false;
      double v = rc.path_vel_max;
      if (i + 1 < np) v = hdsm_sw::voxel_velocity_cap_segment(c, rc, g, origin, p0, pt(i), pt(i + 1), &collided);
      const unsigned long long hit = __ballot(collided);
      const int first = hit != 0ull ? __ffsll((long long)hit) - 1 : 64;
      if (lane > first) v = rc.path_vel_max;
      for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off));
      cap = fmin(cap, v);
      if (hit != 0ull) break;
    }
  }
  if (lane == 0) vel_cap[k] = cap;
}

// ... and KeepOnlyFreeReference (AC:1665-1693) after it, on the rows k_reference wrote
__global__ __launch_bounds__(64) void k_keep_free(Cfg c, int n, const AgentS* agents, double* ref_full, double* ref, const double* path_vel) {
  const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (k >= n) return;
  const int N = c.N;
  double rows[hdsm::MAXH + 1][6];
  for (int i = 0; i <= N; ++i)
    for (int q = 0; q < 6; ++q) rows[i][q] = ref_full[((size_t)k * (N + 1) + i) * 6 + q];
  hdsm_sw::keep_only_free(c, hdsm_sw::local_grid_origin(c, agents[k]), path_vel[k], rows, N + 1);
  for (int i = 0; i <= N; ++i)
    for (int q = 0; q < 6; ++q) {
      ref_full[((size_t)k * (N + 1) + i) * 6 + q] = rows[i][q];
      if (i < N) ref[((size_t)k * N + i) * 6 + q] = rows[i][q];
    }
}

__global__ __launch_bounds__(64) void k_commit(Cfg c, int n, int per, AgentS* agents, const double* traj, const double* ctrl,
                                               const uint8_t* used, const int32_t* status, double* plans_local, int32_t* fails,
                                               uint8_t* has_direct, const double* ref_full, const double* path_vel) {
  // one wavefront per published record. Lane 0 does the read-back / fallback; the increment check — a walk of ~100 samples per
  // metre of reference — is split by reference segment over the lanes (the samples of a segment do not depend on the others,
  // hdsm_sw::increment_segment_min), the minima meet in a wave reduction; then all lanes write the record.
  const int k = (int)blockIdx.x, lane = (int)threadIdx.x;
  if (k >= per) return;
  const int N = c.N, rec = (N + 1) * 9;
  double* out = plans_local + (size_t)k * rec;
  __shared__ int have_s;
  if (lane == 0) have_s = 0;
  __syncthreads();
  if (k < n) {
    AgentS& ag = agents[k];
    // the reference of this round becomes traj_ref_curr_ (the increment check below and the next round's corridor read it)
    for (int t = lane; t < (N + 1) * 6; t += 64) ag.traj_ref[t / 6][t % 6] = ref_full[(size_t)k * (N + 1) * 6 + t];
    if (lane == 0) ag.n_ref = N + 1, ag.path_vel = path_vel[k];
    {  // hdsm_sw::commit_copy with the copies spread over the lanes (one lane doing them was a chain of ~130 dependent
       // global accesses, 35 of the kernel's 40 us): flat element e of traj_curr[][9] / ctrl_curr[][3]
      const int st = status[k];
      double* tc = &ag.traj_curr[0][0];
      double* cc = &ag.ctrl_curr[0][0];
      if (st != HDSM_NO_SOLUTION) {  // AC:960-987
        const double* tin = traj + (size_t)k * rec;
        const double* cin = ctrl + (size_t)k * N * 3;
        for (int e = lane; e < rec; e += 64) tc[e] = tin[e];
        for (int e = lane; e < N * 3; e += 64) cc[e] = cin[e];
        if (lane < c.P) ag.poly_used[lane] = used[(size_t)k * c.P + lane];
        if (lane == 0) ag.has_traj = 1, have_s = 1;
      } else {  // AC:1000-1019: every lane reads what it moves before anybody writes
        constexpr int PER = (hdsm::MAXH * 9 + 63) / 64;
        const bool shift = ag.has_traj != 0;
        double tv[PER], cv[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
          const int e = lane + 64 * u;
          tv[u] = (shift && e < N * 9) ? tc[e + 9] : 0.0;
          cv[u] = (shift && e < (N - 1) * 3) ? cc[e + 3] : 0.0;
        }
        __syncthreads();
        if (shift) {
#pragma unroll
          for (int u = 0; u < PER; ++u) {
            const int e = lane + 64 * u;
            if (e < N * 9) tc[e] = tv[u];
            if (e < (N - 1) * 3) cc[e] = cv[u];
          }
        }
        if (lane == 0) {
          ++ag.n_fail;
          atomicAdd(fails, 1);
          have_s = shift ? 1 : 0;
        }
      }
    }
    __syncthreads();
    if (have_s) {
      int inc = 0;
      if (ag.n_ref >= 2) {
        const V3 pt = {{ag.traj_curr[1][0], ag.traj_curr[1][1], ag.traj_curr[1][2]}};
        const double d0 = hdsm_sw::norm(hdsm_sw::sub(pt, V3{{ag.traj_ref[0][0], ag.traj_ref[0][1], ag.traj_ref[0][2]}}));
        // The minimum over the samples in closed form first (three candidate samples per segment instead of ~90 generated one
        // after the other, two square roots and three divisions in a chain each: half of this kernel's time). It differs from
        // the literal walk's minimum by ~1e-11 m at most; the decision compares the minimum with d0 and thresh_dist, so unless one
        // of those comparisons is closer than 1e-9 m the literal walk would decide the same — and when one is, it is walked.
        double best = DBL_MAX;
        if (lane < ag.n_ref - 1) best = hdsm_sw::increment_segment_min_closed_form(ag, lane, pt);
        for (int off = 32; off > 0; off >>= 1) best = fmin(best, __shfl_xor(best, off));
        // (the literal walk's accumulated rounding is ~100 steps x ulp(position): the guard grows with the magnitude of the
        // coordinates, 1e-9 m up to a few hundred metres, so that worlds of 1e5 m keep the same safety factor)
        const double mag = fmax(fmax(fabs(pt.v[0]), fabs(pt.v[1])), fabs(pt.v[2]));
        const double guard = 1e-9 * fmax(1.0, mag * 0.01);
        if (!(fabs(best - d0) > guard && fabs(best - c.thresh_dist) > guard)) {
          best = DBL_MAX;
          if (lane < ag.n_ref - 1) best = hdsm_sw::increment_segment_min(ag, lane, pt);
          for (int off = 32; off > 0; off >>= 1) best = fmin(best, __shfl_xor(best, off));
        }
        inc = hdsm_sw::increment_from_minima(c, d0, best);
      }
      if (lane == 0) ag.increment = inc;
      if (lane < 9) ag.state_curr[lane] = ag.traj_curr[c.step_plan][lane];  // AC:233-238
      for (int e = lane; e < rec; e += 64) out[e] = ag.traj_curr[e / 9][e % 9];
    }
  }
  if (!have_s) {  // no plan yet (or padding): the record carries the sentinel instead of a flag (hdsm_exchange_device)
    for (int e = lane; e < rec; e += 64) out[e] = e == 0 ? __longlong_as_double(0x7ff8000000000000LL) : 0.0;
  }
  // a single rank publishes straight into the plans buffer of the next round (no copy, no flag kernel): the flag too
  if (has_direct != nullptr && lane == 0) has_direct[k] = have_s ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_flags(int rec, int n, const double* plans, uint8_t* has) {
  const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (k >= n) return;
  const double v = plans[(size_t)k * rec];
  has[k] = (v == v) ? 1 : 0;
}

struct DSwarm {
  void* solver = nullptr;
  int device = 0, n_local = 0, n_rob = 0, first = 0, per = 0, world = 1;
  hdsm_params prm{};
  hdsm_swarm_config cfg{};
  hdsm_ref_config rcfg{};
  Cfg c{};
  AgentS* d_agents = nullptr;
  PolyCache* d_cache = nullptr;  // [n_local], worlds only (HDSM_POLY_CACHE=0: none)
  int8_t* d_world = nullptr;
  double *d_cap = nullptr, *d_path = nullptr, *d_ref_full = nullptr, *d_ref = nullptr, *d_pv = nullptr, *d_state = nullptr, *d_A = nullptr, *d_b = nullptr,
         *d_traj = nullptr, *d_ctrl = nullptr, *d_obj = nullptr, *d_local = nullptr, *d_plans = nullptr;
  int32_t *d_npath = nullptr, *d_id = nullptr, *d_npoly = nullptr, *d_nrows = nullptr, *d_status = nullptr, *d_fails = nullptr;
  uint8_t *d_used = nullptr, *d_has = nullptr;
  long long rounds = 0;
  // hdsm_dswarm_set_phase_timing: HIP events between the launches of a round (development / bench aid: every record is a barrier
  // packet in front of the next kernel, so a timed round is a few us longer than a plain one — ms_per_round is never taken from it)
  bool phase_timing = false, phase_valid = false;
  hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

template <class T>
hipError_t dalloc(T** p, size_t count) {
  hipError_t e = hipMalloc(reinterpret_cast<void**>(p), (count ? count : 1) * sizeof(T));
  if (e == hipSuccess) e = hipMemset(*p, 0, (count ? count : 1) * sizeof(T));
  return e;
}

void free_all(DSwarm* d) {
  void* ptrs[] = {d->d_cap, d->d_agents, d->d_cache, d->d_world, d->d_path, d->d_ref_full, d->d_ref, d->d_pv, d->d_state, d->d_A, d->d_b, d->d_traj,
                  d->d_ctrl, d->d_obj, d->d_local, d->d_plans, d->d_npath, d->d_id, d->d_npoly, d->d_nrows, d->d_status, d->d_fails,
                  d->d_used, d->d_has};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  for (hipEvent_t e : d->ev)
    if (e) (void)hipEventDestroy(e);
}

}  // namespace

extern "C" {

const char* hdsm_dswarm_last_error(void) { return g_err.c_str(); }

// Where a round of the device-resident loop goes, measured with HIP events on the round's own stream: the records sit between the
// launches of hdsm_dswarm_round — [0] k_corridor, [1] k_vel_cap, [2] hdsm_reference_device (pack + reference), [3] k_keep_free,
// [4] hdsm_replan_device (pre-pass, solver kernels, merge), [5] k_commit, [6] the exchange. hdsm_dswarm_last_phase_ms synchronises
// with the last timed round and returns the seven durations in milliseconds (a phase the round did not launch: ~0).
int hdsm_dswarm_set_phase_timing(void* dswarm, int32_t on) {
  DSwarm* d = static_cast<DSwarm*>(dswarm);
  if (!d) return fail(HDSM_ERR_BAD_ARG, "null dswarm");
  HIP_TRY(hipSetDevice(d->device));
  if (on)
    for (hipEvent_t& e : d->ev)
      if (!e) HIP_TRY(hipEventCreate(&e));
  d->phase_timing = on != 0, d->phase_valid = false;
  return HDSM_OK;
}

// The polyhedron cache of the device corridor (k_corridor, PolyCache), summed over the shard's agents since hdsm_dswarm_create:
// out[0] polyhedra asked for, out[1] formed from a cached structure recorded in the same local grid, out[2] formed from one recorded in
// ANOTHER grid at the same height (the interior rule), out[3] = 1 if the cache is on (HDSM_POLY_CACHE, a world is set), else 0.
int hdsm_dswarm_cache_stats(void* dswarm, int64_t out[4]) {
  DSwarm* d = static_cast<DSwarm*>(dswarm);
  if (!d || !out) return fail(HDSM_ERR_BAD_ARG, "null argument");
  out[0] = out[1] = out[2] = out[3] = 0;
  if (!d->d_cache || d->n_local == 0) return HDSM_OK;
  HIP_TRY(hipSetDevice(d->device));
  HIP_TRY(hipDeviceSynchronize());
  out[3] = 1;
  // (only the four counters of every entry travel: one strided 2-D copy)
  std::vector<int32_t> cnt((size_t)d->n_local * 4);
  HIP_TRY(hipMemcpy2D(cnt.data(), 16, reinterpret_cast<const char*>(d->d_cache) + offsetof(PolyCache, asked), sizeof(PolyCache), 16, (size_t)d->n_local,
                      hipMemcpyDeviceToHost));
  for (int k = 0; k < d->n_local; ++k) out[0] += cnt[4 * (size_t)k], out[1] += cnt[4 * (size_t)k + 1], out[2] += cnt[4 * (size_t)k + 2];
  return HDSM_OK;
}

int hdsm_dswarm_last_phase_ms(void* dswarm, float ms[7]) {
  DSwarm* d = static_cast<DSwarm*>(dswarm);
  if (!d || !ms) return fail(HDSM_ERR_BAD_ARG, "null argument");
  if (!d->phase_valid) return fail(HDSM_ERR_BAD_ARG, "no round has run with hdsm_dswarm_set_phase_timing on");
  HIP_TRY(hipSetDevice(d->device));
  HIP_TRY(hipEventSynchronize(d->ev[7]));
  for (int k = 0; k < 7; ++k) HIP_TRY(hipEventElapsedTime(&ms[k], d->ev[k], d->ev[k + 1]));
  return HDSM_OK;
}

int hdsm_dswarm_create(void* swarm, void* solver, int32_t device, int32_t world_size, void** dswarm) {
  if (!swarm || !solver || !dswarm || world_size < 1) return fail(HDSM_ERR_BAD_ARG, "null argument");
  *dswarm = nullptr;
  DSwarm* d = new (std::nothrow) DSwarm;
  if (!d) return fail(HDSM_ERR_DEVICE, "out of host memory");
  d->solver = solver, d->device = device, d->world = world_size;
  const int8_t* hworld = nullptr;
  int32_t wdim[3] = {0, 0, 0};
  double worigin[3] = {0, 0, 0};
  int rc = hdsm_swarm_export_state(swarm, nullptr, &d->n_local, &d->n_rob, &d->first, &d->prm, &d->cfg, &hworld, wdim, worigin);
  if (rc) {
    delete d;
    return fail(rc, "hdsm_swarm_export_state");
  }
  d->per = (d->n_rob + world_size - 1) / world_size;
  if (d->n_local > d->per) {
    delete d;
    return fail(HDSM_ERR_BAD_ARG, "the shard is larger than ceil(n_rob / world_size)");
  }
  // Layout contract of the device loop (hdsm_swarm.h): the plans buffer holds agent a's record at slot a — the all-gather puts
  // rank r's block at r * per, a single rank copies its block to slot 0 — and the solver finds an agent's OWN plan (and skips
  // its own record in the sweeps) by agent id. So the shard must be the block of a rank of the ceil(n_rob / world_size) split.
  // (An EMPTY trailing shard — n_rob = 5 on 4 ranks leaves rank 3 with first_id = n_rob and no agent, what swarm.shard_range
  // produces — is a valid block; its rank comes from the communicator, not from first_id / per.)
  const bool empty_tail = d->n_local == 0 && d->first == d->n_rob;
  if (d->per > 0 && !empty_tail &&
      (d->first % d->per != 0 || (d->n_local < d->per && d->first + d->n_local != d->n_rob) || (world_size == 1 && d->first != 0))) {
    delete d;
    return fail(HDSM_ERR_BAD_ARG, "the shard is not a block of the ceil(n_rob / world_size) split: first_id must be rank * per, a short "
                                  "shard must be the last one (and first_id 0 with world_size 1)");
  }
  d->rcfg = {d->cfg.path_vel_min, d->cfg.path_vel_max, d->cfg.sens_dist, d->cfg.sens_pot, d->cfg.sens_other_agents, d->cfg.path_vel_dec};
  Cfg& c = d->c;
  c.N = d->prm.n_hor, c.P = d->prm.poly_hor, c.RS = d->prm.max_rows_static, c.step_plan = d->cfg.step_plan;
  c.n_it_decomp = d->cfg.n_it_decomp, c.use_cvx_new = d->cfg.use_cvx_new, c.has_world = hworld ? 1 : 0;
  c.voxel_size = d->cfg.voxel_size, c.grid_z_min = d->cfg.grid_z_min, c.thresh_dist = d->cfg.thresh_dist;
  c.fast_walk = 1;
  if (const char* e = std::getenv("HDSM_FAST_WALK")) {  // development switch (A/B of the walk shortcut): "0" or "1", anything else is ignored
    if ((e[0] == '0' || e[0] == '1') && e[1] == 0) c.fast_walk = e[0] == '1';
  }
  for (int k = 0; k < 3; ++k) c.grid_range[k] = d->cfg.grid_range[k], c.wdim[k] = wdim[k], c.worigin[k] = worigin[k];
  if (hipSetDevice(device) != hipSuccess) {
    delete d;
    return fail(HDSM_ERR_NO_DEVICE, "hipSetDevice failed");
  }
  const size_t n = (size_t)d->n_local, L = (size_t)d->per, G = (size_t)d->per * world_size, N = (size_t)c.N, P = (size_t)c.P, RS = (size_t)c.RS,
               REC = (N + 1) * 9;
  hipError_t e = hipSuccess;
  auto ok = [&](hipError_t r) {
    if (e == hipSuccess) e = r;
  };
  ok(dalloc(&d->d_agents, n));
  if (hworld) {
    ok(dalloc(&d->d_world, (size_t)wdim[0] * wdim[1] * wdim[2]));
    bool cache_on = true;  // development switch (A/B): HDSM_POLY_CACHE=0 grows every polyhedron again
    if (const char* pcenv = std::getenv("HDSM_POLY_CACHE")) cache_on = !(pcenv[0] == '0' && pcenv[1] == 0);
    if (cache_on) ok(dalloc(&d->d_cache, n));
  }
  ok(dalloc(&d->d_path, n * PTS * 3)), ok(dalloc(&d->d_npath, n)), ok(dalloc(&d->d_ref_full, n * (N + 1) * 6)), ok(dalloc(&d->d_ref, n * N * 6));
  ok(dalloc(&d->d_cap, n)), ok(dalloc(&d->d_pv, n)), ok(dalloc(&d->d_id, n)), ok(dalloc(&d->d_state, n * 9)), ok(dalloc(&d->d_npoly, n)), ok(dalloc(&d->d_nrows, n * P));
  ok(dalloc(&d->d_A, n * P * RS * 3)), ok(dalloc(&d->d_b, n * P * RS)), ok(dalloc(&d->d_traj, L * REC)), ok(dalloc(&d->d_ctrl, L * N * 3));
  ok(dalloc(&d->d_obj, L)), ok(dalloc(&d->d_used, L * P)), ok(dalloc(&d->d_status, L)), ok(dalloc(&d->d_local, L * REC));
  ok(dalloc(&d->d_plans, G * REC)), ok(dalloc(&d->d_has, G)), ok(dalloc(&d->d_fails, 1));
  if (e == hipSuccess && n) {
    AgentS* tmp = static_cast<AgentS*>(std::malloc(n * sizeof(AgentS)));
    if (!tmp) e = hipErrorOutOfMemory;
    else {
      rc = hdsm_swarm_export_state(swarm, tmp, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
      if (rc == HDSM_OK) e = hipMemcpy(d->d_agents, tmp, n * sizeof(AgentS), hipMemcpyHostToDevice);
      std::free(tmp);
    }
  }
  if (e == hipSuccess && n) {  // agent ids of the shard (contiguous block)
    std::string ids(n * 4, '\0');
    for (size_t k = 0; k < n; ++k) reinterpret_cast<int32_t*>(&ids[0])[k] = d->first + (int)k;
    e = hipMemcpy(d->d_id, ids.data(), n * 4, hipMemcpyHostToDevice);
  }
  if (e == hipSuccess && hworld) e = hipMemcpy(d->d_world, hworld, (size_t)wdim[0] * wdim[1] * wdim[2], hipMemcpyHostToDevice);
  c.world = d->d_world;
  if (e != hipSuccess || rc) {
    free_all(d);
    delete d;
    return fail(rc ? rc : HDSM_ERR_DEVICE, std::string("hdsm_dswarm_create: ") + (rc ? "export failed" : hipGetErrorString(e)));
  }
  *dswarm = d;
  return HDSM_OK;
}

void hdsm_dswarm_destroy(void* dswarm) {
  DSwarm* d = static_cast<DSwarm*>(dswarm);
  if (!d) return;
  (void)hipSetDevice(d->device);
  (void)hipDeviceSynchronize();
  free_all(d);
  delete d;
}

int hdsm_dswarm_round(void* dswarm, void* comm, void* hip_stream) {
  DSwarm* d = static_cast<DSwarm*>(dswarm);
  if (!d) return fail(HDSM_ERR_BAD_ARG, "null dswarm");
  if (d->world > 1 && !comm) return fail(HDSM_ERR_BAD_ARG, "a sharded swarm needs a communicator");
  if (comm) {  // the communicator must be the one this shard was cut for: its rank's block starts at first
    int32_t crank = -1, cworld = -1;
    if (hdsm_comm_info(comm, &crank, &cworld) != HDSM_OK) return fail(HDSM_ERR_COMM, std::string("hdsm_comm_info: ") + hdsm_last_error());
    const bool empty_tail = d->n_local == 0 && d->first == d->n_rob;  // (any rank behind the last agent)
    if (cworld != d->world || (d->per > 0 && (empty_tail ? (int64_t)crank * d->per < d->n_rob : crank != d->first / d->per)))
      return fail(HDSM_ERR_BAD_ARG, "communicator rank / size do not match the shard (first_id / per, world_size)");
  }
  HIP_TRY(hipSetDevice(d->device));
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const int n = d->n_local, G = d->per * d->world, rec = (d->c.N + 1) * 9;
  const unsigned gb = (unsigned)((n + 63) / 64);
  // The solver handle's "done" event (what a later call on another stream waits for) is recorded once, when the round has been
  // issued, instead of after each of its entry points: every record is a barrier packet in front of the next kernel (5-6 us each).
  struct DoneOnce {
    void* solver;
    hipStream_t st;
    DoneOnce(void* s, hipStream_t q) : solver(s), st(q) { (void)hdsm_internal_defer_done(solver, 1); }
    ~DoneOnce() {
      (void)hdsm_internal_defer_done(solver, 0);
      (void)hdsm_internal_record_done(solver, st);
    }
  } done_once(d->solver, st);
  const bool timing = d->phase_timing;
#define PHASE_MARK(k) \
  do {                \
    if (timing) HIP_TRY(hipEventRecord(d->ev[k], st)); \
  } while (0)
  PHASE_MARK(0);
  if (n > 0) {
    hipLaunchKernelGGL(k_corridor, dim3((unsigned)n), dim3(64), d->c.has_world ? hdsm_cd::wave_lds_bytes(hdsm_cd::wave_map_radius(d->c.n_it_decomp)) : 0, st, d->c, n, d->d_agents, d->d_path, d->d_npath, d->d_id, d->d_state,
                       d->d_npoly, d->d_nrows, d->d_A, d->d_b, d->d_cache);
    HIP_TRY(hipGetLastError());
    PHASE_MARK(1);
    if (d->c.has_world) {
      hipLaunchKernelGGL(k_vel_cap, dim3((unsigned)n), dim3(64), 0, st, d->c, d->rcfg, n, d->d_agents, d->d_path, d->d_npath, d->d_cap);
      HIP_TRY(hipGetLastError());
    }
    PHASE_MARK(2);
    int rc = hdsm_reference_device(d->solver, &d->rcfg, n, G, d->d_id, d->d_path, d->d_npath, PTS, d->c.has_world ? d->d_cap : nullptr,
                                   d->d_plans, d->d_has, d->d_ref_full, d->d_ref, d->d_pv, st);
    if (rc) return fail(rc, std::string("hdsm_reference_device: ") + hdsm_last_error());
    PHASE_MARK(3);
    if (d->c.has_world) {
      hipLaunchKernelGGL(k_keep_free, dim3(gb), dim3(64), 0, st, d->c, n, d->d_agents, d->d_ref_full, d->d_ref, d->d_pv);
      HIP_TRY(hipGetLastError());
    }
    PHASE_MARK(4);
    rc = hdsm_replan_device(d->solver, n, G, d->d_id, d->d_state, d->d_ref, d->d_npoly, d->d_nrows, d->d_A, d->d_b, d->d_plans, d->d_has,
                            d->d_traj, d->d_ctrl, d->d_used, d->d_status, d->d_obj, st);
    if (rc) return fail(rc, std::string("hdsm_replan_device: ") + hdsm_last_error());
  } else {
    for (int k = 1; k <= 4; ++k) PHASE_MARK(k);
  }
  PHASE_MARK(5);
  // (one rank: the solve of this round is behind us on the stream, so the records go straight into the plans buffer — slot k =
  // agent k — and the flags with them; several ranks: into the send buffer of the ONE all-gather)
  const bool direct = d->world == 1;
  hipLaunchKernelGGL(k_commit, dim3((unsigned)(d->per > 0 ? d->per : 1)), dim3(64), 0, st, d->c, n, d->per, d->d_agents, d->d_traj, d->d_ctrl, d->d_used, d->d_status,
                     direct ? d->d_plans : d->d_local, d->d_fails, direct ? d->d_has : nullptr, d->d_ref_full, d->d_pv);
  HIP_TRY(hipGetLastError());
  PHASE_MARK(6);
  if (!direct) {
    const int rc = hdsm_exchange_device(comm, d->per, d->d_local, d->d_plans, d->d_has, st);
    if (rc) return fail(rc, std::string("hdsm_exchange_device: ") + hdsm_last_error());
  }
  PHASE_MARK(7);
#undef PHASE_MARK
  if (timing) d->phase_valid = true;
  ++d->rounds;
  return HDSM_OK;
}

int hdsm_dswarm_upload_plans(void* dswarm, const double* plans_all, const uint8_t* has_plan) {
  DSwarm* d = static_cast<DSwarm*>(dswarm);
  if (!d || !plans_all || !has_plan) return fail(HDSM_ERR_BAD_ARG, "null argument");
  HIP_TRY(hipSetDevice(d->device));
  HIP_TRY(hipDeviceSynchronize());
  const size_t G = (size_t)d->per * d->world, rec = (size_t)(d->c.N + 1) * 9, have = (size_t)d->n_rob < G ? (size_t)d->n_rob : G;
  HIP_TRY(hipMemset(d->d_has, 0, G));
  HIP_TRY(hipMemcpy(d->d_plans, plans_all, have * rec * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d->d_has, has_plan, have, hipMemcpyHostToDevice));
  return HDSM_OK;
}

int hdsm_dswarm_download(void* dswarm, void* swarm, double* plans_all, uint8_t* has_plan, int32_t* status, int32_t* failed_total) {
  DSwarm* d = static_cast<DSwarm*>(dswarm);
  if (!d) return fail(HDSM_ERR_BAD_ARG, "null dswarm");
  HIP_TRY(hipSetDevice(d->device));
  HIP_TRY(hipDeviceSynchronize());
  const size_t n = (size_t)d->n_local, G = (size_t)d->per * d->world, rec = (size_t)(d->c.N + 1) * 9;
  if (swarm && n) {
    AgentS* tmp = static_cast<AgentS*>(std::malloc(n * sizeof(AgentS)));
    if (!tmp) return fail(HDSM_ERR_DEVICE, "out of host memory");
    hipError_t e = hipMemcpy(tmp, d->d_agents, n * sizeof(AgentS), hipMemcpyDeviceToHost);
    const int rc = e == hipSuccess ? hdsm_swarm_import_state(swarm, tmp, d->n_local) : HDSM_ERR_DEVICE;
    std::free(tmp);
    if (rc) return fail(rc, "state download failed");
  }
  if (plans_all) {
    HIP_TRY(hipMemcpy(plans_all, d->d_plans, G * rec * 8, hipMemcpyDeviceToHost));
    for (size_t k = 0; k < G; ++k)
      if (plans_all[k * rec] != plans_all[k * rec]) plans_all[k * rec] = 0.0;  // the sentinel is for the wire only
  }
  if (has_plan) HIP_TRY(hipMemcpy(has_plan, d->d_has, G, hipMemcpyDeviceToHost));
  if (status && n) HIP_TRY(hipMemcpy(status, d->d_status, n * 4, hipMemcpyDeviceToHost));
  if (failed_total) HIP_TRY(hipMemcpy(failed_total, d->d_fails, 4, hipMemcpyDeviceToHost));
  return HDSM_OK;
}

}  // extern "C"
