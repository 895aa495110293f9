// hdsm_core.h — per-instance MIQP solver: ONE agent-replan per workgroup, all solver state in LDS.
//
// Replaces, for one agent and one replan, Agent::GenerateTimeAwareSafeCorridor (AC:1086-1215) +
// Agent::SolveOptimizationProblem (AC:858-1023) of lis-epfl/multi_agent_pkgs
// (AC = multi_agent_planner/src/agent_class.cpp). Design notes: DESIGN.md.
//
//   * decision vector u = jerk inputs, condensed (states eliminated), n = 3 N, index ax*N + k;
//   * exact dual active-set QP (Goldfarb-Idnani): the iteration runs on one wavefront with the factor in registers
//     (n <= 30: hdsm_wave_gib.h, larger n: hdsm_wave_gi.h);
//   * everything an instance needs before its first iteration is one linear map of (state, reference), precomputed
//     by hdsm_create (Consts::KT);
//   * the n_rob-1 neighbour planes per step are NEVER materialised: a sweep over the all-gathered plans
//     buffer generates each plane on the fly (trig-free closed form of the ellipsoid support distance)
//     and stages only the rows close to the current iterate in LDS; after convergence a verification sweep
//     re-checks every row, stages the violated ones and the dual method simply continues (it stays dual
//     feasible when rows are added), so the result is exact; a verification sweep is skipped when a displacement
//     bound proves that no unstaged row can be violated;
//   * the one-hot polyhedron choice is handled by a lazy depth-first branch-and-bound: a node branches
//     only on a step whose segment lies in no polyhedron; children continue the parent's factorisation
//     (snapshots of the solver state live in a per-instance global scratch, one per depth).
//
// Device code only (hipcc, gfx950): PAR_FOR distributes a loop over the threads of the workgroup, SYNC() is __syncthreads().
// The CPU test-suite runs THIS source as lockstep fibers (tests/wave_emu/); there is no second formulation of the algorithm.
#pragma once
#include <math.h>
#include <stdint.h>

#include <hip/hip_runtime.h>
#define HD __device__ __forceinline__
#define HDN __device__ __noinline__
#define PAR_FOR(i, cnt) for (int i = tid_here(); i < (cnt); i += (int)blockDim.x)  // (tid_here: hdsm_wave_gi.h)
#define HDSM_UNROLL _Pragma("unroll")
#define SYNC() __syncthreads()
#define IS_T0 (HDSM_TX == 0)
namespace hdsm {
__device__ __forceinline__ int atomic_inc_i32(int* p) { return atomicAdd(p, 1); }
// The thread index as a value formed WHERE IT IS READ (HDSM_TX replaces threadIdx.x in the solver): whatever depends on it is then
// computed below that point. The persistent workgroups of pass 2 (hdsm_api.hip, run_items) run the solver inside a loop over items; with
// the plain built-in the optimiser computed every lane mask and lane address once, before the loop, and kept them alive — spilled,
// 700 bytes per lane — across all items and their active-set runs.
__device__ __forceinline__ unsigned tx() {
  unsigned t = threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(t));
#endif
  return t;
}
}  // namespace hdsm
#define HDSM_TX (hdsm::tx())
#if defined(__HIP_DEVICE_COMPILE__)
#define HDSM_KERNARG_WORD __attribute__((address_space(4))) long long
#else
#define HDSM_KERNARG_WORD long long
#endif

#include "hdsm_types.h"

namespace hdsm {

enum { GI_OK = 0, GI_INFEASIBLE = 1, GI_CUTOFF = 2, GI_ITERLIM = 3, GI_DONE = 4, GI_TIMELIM = 5 };
constexpr int NODE_CHUNK = 4;       // nodes a sub-block of a split launch takes from its instance's pool at a time
constexpr int TREE_MARK = 32;       // nodes from which a tree counts as deep (the split form of a launch pays from about there)
constexpr int WARM_CERT = 1 << 30;  // bit of the stored working-set size: the set is an infeasibility certificate
enum { FLAG_NODE_LIMIT = 1, FLAG_ITER_LIMIT = 2, FLAG_TIME_LIMIT = 4, FLAG_STAGING_OVERFLOW = 8 };  // HDSM_FLAG_* of hdsm.h
enum { ST_OPTIMAL = 0, ST_LIMIT = 1, ST_NO_SOLUTION = 2, ST_PENDING = 3 };  // (ST_PENDING: internal, a queued item of a split launch)

// constraint ids: kind in bits 28..30
enum { K_U = 0, K_S = 1, K_P = 2, K_C = 3, K_E = 4 };
HD int mk_id(int kind, int payload) { return (kind << 28) | payload; }
HD int id_kind(int id) { return (id >> 28) & 7; }
HD int id_payload(int id) { return id & 0x0fffffff; }
// staged neighbour rows (K_C): the payload carries the slot in cand[] AND the trajectory point the row acts on, so that the
// normal of an entering row needs one LDS round trip (row and impulse-response entry together), not two in a chain
HD int mk_kc(int slot, int m) { return mk_id(K_C, (m << 12) | slot); }
HD int kc_slot(int payload) { return payload & 0xfff; }
HD int kc_m(int payload) { return payload >> 12; }

constexpr int LISTCAP = 768;   // neighbours per chunk of a sphere-prefiltered sweep

// ---------------------------------------------------------------------------------------------------------
// All solver state of one instance. NV = capacity for n = 3N; CMAX = staged neighbour rows.
// step m of a staged row and the weight 1 / sqrt(a^T Z a) of its normal (Consts::pick_rule; single precision: it only orders picks)
struct alignas(8) MW {
  int32_t m;
  float w;
};
static_assert(sizeof(MW) == 8, "hand-over records copy MW as one 8-byte word");
HD MW mk_mw(const double (*kap)[4], double nx, double ny, double nz, int m) {
  const double q = nx * nx * kap[m][0] + ny * ny * kap[m][1] + nz * nz * kap[m][2];
  // (a row on a pinned position cannot be moved at all: if it is violated it goes first and ends the instance)
  return MW{m, q > 1e-60 ? (float)(1.0 / sqrt(q)) : 1e30f};
}

// SMALL: the four-workgroups-per-CU shape (40 KB of LDS per instance): capacity for 4 polyhedra of 20 rows (the agile configuration
// of the reference: poly_hor 4, 18 rows) and 512 neighbours per chunk of a prefiltered sweep instead of the limits of hdsm.h
template <int NV, int CMAX, bool SMALL = false>
struct Shm {
  static_assert(CMAX <= 4096, "the slot of a staged row must fit the 12 bits kc_slot() reads");
  static constexpr int PM = SMALL ? 4 : MAXP, RSM = SMALL ? 20 : MAXRS, LC = SMALL ? 512 : LISTCAP;
  static constexpr int LDT = (NV <= 32) ? 34 : NV + 2;  // even (16-B rows) and conflict-free for b128 row reads
  alignas(16) double T[2];  // (was the transposition buffer of the one-lane-per-row code for d = J^T a: 19 KB at NV = 48, now staging rows)
  alignas(16) double U[NV * LDT];       // U = R^{-1}, row k = working-set position k (upper triangular, zero-padded)
  alignas(16) double dvec[NV + 2];      // broadcast vector (d, or a row of U)
  alignas(16) double dvz[NV + 2];       // d with the working-set columns (j < q) zeroed
  double gz[3][3][2 * MAXH];            // zero-padded impulse responses: gz[ax][s][MAXH + lag]
  double x0[NV];                        // unconstrained minimiser (base point of the warm start)
  double fx0;                           // J(x0)
  int32_t cand_src[CMAX];               // origin of a staged row: (neighbour << 6) | (step << 1) | endpoint, -1 = explicit
  int32_t inc_act[NV], inc_nact;        // working set of the incumbent (portable ids) -> next replan's guess
  int32_t st_sph, st_pairs;             // sweep counters: sphere records read, (neighbour, step) positions loaded
  long long t_start;                    // constant-rate clock at the start of the instance (time_limit_s)
#ifdef HDSM_PROFILE
  long long prof_acc[24];  // 0..7 iteration phases, 8..15 sweeps / set-up, 16..23 inside the warm start
  long long prof_last;
  long long prof_last1;  // ... of the scanner wave (thread 64)
#endif
  double x[NV], lam[NV], w[NV], grad[NV];
  double inc_x[NV];
  double g[3][3][MAXH];
  double fr[3][MAXH + 1][3];
  double ref[MAXH][6];
  double st[MAXH + 1][9];
  double cprev[MAXH][3];
  alignas(16) double sp[PM][RSM][4];  // rows of the instance's polyhedra (A, b): read as two 16-byte halves by scan_assigned
  double keys[MAXH][PM];
  // child bound (leaf_check -> select_child): klb[i][j] = a lower bound of what assigning polyhedron j to step i adds to the
  // node's objective; br_lb[L][pos] = the resulting lower bound of the child at position pos of level L
  static constexpr int HTS = NV / 3;  // steps of this instantiation (10 for NV = 32, 16 for NV = 48)
  // Per open level of the branch and bound, per child position (leaf_check -> level open -> select_child):
  //   br_lb  lower bound of the child's objective: a child whose bound reaches the incumbent is never opened;
  //   br_pk / br_pv  the child's FIRST entering row — the row of its polyhedron with the largest v^2 / (a^T Z a) at the node's minimiser,
  //          code = (end point << 6) | row, -1: none — and its violation v. The child starts at that minimiser (a snapshot restores it
  //          exactly), where every row of the node holds: what enters first is known without evaluating anything.
  // (the leaf test leaves its per-(step, polyhedron) values in red_v, which nothing else uses between two runs: lb_at / pv_at / pk_at)
  double br_lb[HTS][PM], br_pv[HTS][PM];
  int32_t br_pk[HTS][PM];
  // (item = step * n_poly + polyhedron < 64: the one-wavefront leaf test; the other forms of the test leave leaf_lb = 0 and no values)
  HD double& lb_at(int item) { return red_v[item]; }  // what assigning the polyhedron to the step adds at least
  HD double& pv_at(int item) { return red_v[64 + item]; }
  HD int32_t& pk_at(int item) { return reinterpret_cast<int32_t*>(&red_v[128])[item]; }
  static_assert(128 + 32 <= MAXT, "the leaf test's per-(step, polyhedron) values fit the reduction scratch");
  int32_t leaf_lb;  // the last leaf test left child bounds and first picks (1) or not (0)
  int32_t first_id;  // the run that is about to start enters this row first (ids as in act[]; -1: pick as usual)
  double first_v;
  double lb_top1, lb_top2;  // largest / second largest over the uncontained steps of min_j klb[i][j] (the node bound), and the step
  int32_t lb_step;          // of the largest (-1: none)
  alignas(16) double cand[CMAX][4];      // staged neighbour rows (n_f, rhs): read as two 16-byte halves by the scans
  double red_v[MAXT];
  double br_f[MAXH];
  double state0[9];
  double f, inc_f, f0;
  MW cand_mw[CMAX];      // step of each staged row + its pick-rule weight (one 8-byte read in the scan)
  int32_t act[NV];
  int32_t sp_rows[PM];
  int32_t assign[MAXH], contain[MAXH], inc_assign[MAXH];
  int32_t br_step[MAXH], br_pos[MAXH], br_cnt[MAXH], br_order[MAXH][PM];
  int32_t q, neq_done, ncand, n_poly, level, have_inc, fixed_bad, overflow;
  // conflicts learned by the branch and bound: a set of (step, polyhedron) assignments, as a bit mask (bit 4 i + j), that
  // makes the QP infeasible together with the rows common to every node — no node containing it needs to be opened
  int32_t inf_id;  // row whose addition proved the last node infeasible (ids as in act[])
  unsigned long long nogood[NOGOODS];
  int32_t n_nogood, ng_skipped, ng_global;  // ng_global: a node proved the instance infeasible whatever the assignment
  int32_t lb_skipped;                       // children never opened because their lower bound reached the incumbent
  int32_t ncold;  // rows staged but not scanned every iteration (top of cand[])
  double f_box;        // upper bound of the objective over the input box (DINF: no box / Consts::box_cut off)
  double inc_shared;   // split launches: best objective found by ANY sub-block of this instance (DINF: none / ordinary launch)
  int32_t wanted_raw;  // after a sweep that overflowed: ncand + ncold as counted past the capacity (before the clamp)
  int32_t nviol;  // rows found violated (> tol) by the last sweep
  int32_t warm_head;  // first word of the instance's warm-start record (count | WARM_CERT), fetched by the set-up
  int32_t warm_ncand; // staged rows after the warm start has placed the rows of its guess (before a concurrent sweep adds its own)
  int32_t leaf_pick;  // result of the one-wavefront leaf test
  // Dominated polyhedra (bit j): polyhedron j lies inside another polyhedron of the instance, so it is never offered as a choice and
  // never counts as the container of a segment (its dominator does). Computed ONCE, by the first leaf test that finds an uncontained
  // step (dom_done) — an instance that never has to branch never pays for it; an item of pass 2 takes it from its record.
  int32_t sp_dom, dom_done;
  int32_t forced;  // 1: steps were assigned outside the tree (one polyhedron left, leaf_check): assigned rows exist although level == 0
  double* snap;       // this workgroup's snapshot scratch (global memory; kept here, not in a register pair across the active-set run)
  int32_t node_res;   // pass 2 of a split launch: nodes drawn from the instance's pool and not yet opened
  int32_t rc, iters_sh;  // device build: results of wave 0's active-set run, shared with the other waves
  int32_t cmd;           // command word for the helper waves (0 = leave, 1 = scan staged rows)
  int32_t nlist;         // sweeps with a.bounds: neighbours of the current chunk that survive the sphere test
  int32_t list[LC];
  double sw[5];          // sweep scalars: cull radius, own sphere (centre, radius)
  double sw_ref[MAXH + 1][3];  // positions at the last STAGING sweep and its radius (0 = none): every row not staged then
  double sw_tau;               // had slack >= sw_tau there, so it cannot be violated while |p - sw_ref| |n_f| <= sw_tau
  double sw_d2;                // max_m |st[m] - sw_ref[m]|^2 (scratch of the displacement test)
  double bnd[24];        // lbu[3], ubu[3], lbs[3][3], ubs[3][3], absent = -+DINF (read by select() and resid() inside the iteration)
  // set-up: v = (state_curr, traj_ref) flat, the input of the map KT, zero-padded. It lives in red_v[16 ...): the set-up itself
  // uses red_v[0..5] (residuals of the terminal equalities), the leaf test comes later
  static_assert(MAXT >= 16 + KCOLS + 8, "the reduction scratch doubles as the set-up's input vector");
  HD double* vin() { return &red_v[16]; }
  double part_v[4];      // per-wave partial results of the staged-row scan: violation, key and id of the wave's pick
  double part_key[4];
  double part_tol;       // ... and what wave 0 tells the helpers: the tolerance and the pick rule of this scan
  int32_t part_id[4];
  int32_t part_norm;
  double kap[MAXH + 1][4];  // Consts::kap (pick-rule factors of the position rows)
  Args args;  // launch arguments, copied once so that the kernarg SGPRs are dead after the prologue
};

// One separating plane (AC:1100-1205) from own position cp and neighbour position op: out = (n_f, n_f . q).
HD bool tasc_plane_eval(const Consts& c, const double* cp, const double* op, double* out) {
  const double dx = op[0] - cp[0], dy = op[1] - cp[1], dz = op[2] - cp[2];
  const double n2 = dx * dx + dy * dy + dz * dz;
  if (!(n2 > 0)) return false;  // Eigen normalized() keeps a zero vector: the row is 0.p <= 0
  const double nrm = sqrt(n2), inv = 1.0 / nrm;
  const double hx = dx * inv, hy = dy * inv, hz = dz * inv;
  // ellipsoid support distance: hypot(r cos t, h sin t), t = atan((r/h) tan(pi/2 - |acos hz|))
  //   == r / sqrt(1 + ((r/h)^2 - 1) hz^2)
  const double sd = c.radius / sqrt(1.0 + c.k2m1 * hz * hz);
  const double back = 0.5 * fmin(2.0 * sd, nrm);
  const double qx = 0.5 * (cp[0] + op[0]) - back * hx;
  const double qy = 0.5 * (cp[1] + op[1]) - back * hy;
  const double qz = 0.5 * (cp[2] + op[2]) - back * hz;
  // n x (0,0,1) = (hy,-hx,0);  n x (0,1,0) = (-hz,0,hx);  n_f = n + pert*(c1+c2) + pert*c2
  const double fx = hx + c.pert * (hy - hz) - c.pert * hz;
  const double fy = hy - c.pert * hx;
  const double fz = hz + c.pert * hx + c.pert * hx;
  out[0] = fx, out[1] = fy, out[2] = fz;
  out[3] = fx * qx + fy * qy + fz * qz;
  return true;
}


}  // namespace hdsm
#include "hdsm_wave_gi.h"
#include "hdsm_wave_gib.h"
namespace hdsm {

template <int NV, int CMAX, bool SMALL = false>
struct Solver {
  using S = Shm<NV, CMAX, SMALL>;

  // ---- neighbour sweep: planes on the fly (AC:1100-1205), stage rows with slack < thresh --------------------
  static HD bool tasc_plane(const Consts& c, const double* cp, const double* op, double* out) {
    return tasc_plane_eval(c, cp, op, out);
  }

  // Rows on input-independent positions. p_0 is pinned by the current state, and the positions of the first `pinned_steps` steps
  // do not depend on the inputs either (with jerk inputs and the Euler model p_1 and p_2 are fixed by the current state): a row
  // there is a CONSTANT. It is judged with the tolerance a solver applies to a row it cannot do anything about (Consts::ftol_fixed =
  // Gurobi's FeasibilityTol, 1e-6 — in the squeeze the separating planes pass exactly THROUGH the own previous position, slack 0 +-
  // rounding is the normal case): within it the row holds and is not a constraint, beyond it no trajectory satisfies it — a common
  // row ends the instance (infeasible whatever the polyhedra: the gridlock test of the first sweep, known after one sweep instead
  // of the dozens of active-set operations the dual method needs to run into the contradiction; two thirds of the infeasible
  // instances of the bench rounds are of this kind: 114 of 178 in round 170, 222 of 222 in round 175), a row of a polyhedron makes
  // that polyhedron inadmissible for the step.

  // After a sweep (one thread): the counters were advanced past the capacity by rows that found no slot. No slot is ever
  // written twice (a writer checks the other list's counter AFTER its own atomic increment), so the slots below the clamped
  // counts hold complete rows; the scans must never look beyond them.
  static HD void clamp_staged(S& s) {
    if (s.ncand + s.ncold > CMAX) {
      s.overflow = 1;
      s.wanted_raw = s.ncand + s.ncold;
      const int cold = s.ncold < CMAX ? s.ncold : CMAX;
      if (s.ncand > CMAX - cold) s.ncand = CMAX - cold;
      s.ncold = cold;
    }
  }

  static HD void sweep(S& s, const Consts& c, const Args& a, int inst, int self, double thresh, bool check_fixed) {
    const int N = c.N;
    const bool explicit_rows = a.l1_rows != nullptr;  // level 1: rows given by the caller
    if (!explicit_rows) {  // device build: one thread per (neighbour, step) pair, sphere prefilter (hdsm_wave_gi.h)
      WaveGI<NV, CMAX, SMALL>::sweep_planes(s, c, a, self, thresh, check_fixed, tid_here());
      return;
    }
    const int total = explicit_rows ? N * a.l1_rmax : a.n_rob * N;
    PAR_FOR(idx, total) {
      double row[4];
      int i;
      if (explicit_rows) {
        i = idx / a.l1_rmax;
        const int r = idx % a.l1_rmax;
        if (r >= a.l1_nrows[(int64_t)inst * N + i]) continue;
        const double* src = a.l1_rows + (((int64_t)inst * N + i) * a.l1_rmax + r) * 4;
        row[0] = src[0], row[1] = src[1], row[2] = src[2], row[3] = src[3];
      } else {
        const int k = idx / N;
        i = idx % N;
        if (k == self || !a.has_plan[k]) continue;
        const double* op = a.plans + ((int64_t)k * (N + 1) + (i + 1)) * 9;
        if (!tasc_plane(c, s.cprev[i], op, row)) continue;
      }
      for (int e = 0; e < 2; ++e) {
        const int m = i + e;
        const double* pm = s.st[m];
        const double v = row[0] * pm[0] + row[1] * pm[1] + row[2] * pm[2] - row[3];
        if (m <= c.pinned_steps) {  // a constant row
          if (check_fixed && v > c.ftol_fixed) s.fixed_bad = 1;
          continue;
        }
        if (v > c.tol) s.nviol = 1;  // benign race: every writer stores the same value
        if (-v < thresh) {
          // violated (or almost) -> hot list, scanned every iteration; merely close -> cold list at the top of
          // the staging area, scanned only when the hot rows are all satisfied
          const bool hot = -v < c.hot_tau;
          int slot;
          const bool fits = stage_slot<CMAX>(s.ncand, s.ncold, hot, slot);  // (hdsm_wave_gi.h: increment, THEN the other list's counter)
          if (fits) {
            s.cand[slot][0] = row[0], s.cand[slot][1] = row[1], s.cand[slot][2] = row[2];
            s.cand[slot][3] = row[3];
            s.cand_mw[slot] = mk_mw(s.kap, row[0], row[1], row[2], m);
            s.cand_src[slot] = explicit_rows ? -1 : (((idx / N) << 6) | (i << 1) | e);
          } else {
            s.overflow = 1;
          }
        }
      }
    }
    SYNC();
    if (IS_T0) {
      clamp_staged(s);
    }
    SYNC();
  }

  // ---- dominated polyhedra (one wavefront, `lane` = its lane; every lane returns the mask) ---------------------------------
  // P_j is CONTAINED in P_k when every row (a, b) of P_k is implied by a row (a2, b2) of P_j with the same direction:
  // a = t a2, t > 0 and t b2 <= b — then a . p = t a2 . p <= t b2 <= b wherever P_j's row holds. (A sufficient test — no vertex
  // enumeration — that finds what actually occurs: the corridor of an agent that is held up keeps seeding polyhedra from
  // neighbouring voxels, which grow into the SAME chamfers and faces with one or two faces shifted; BASELINE cfg 5's instances
  // that ended on the node budget had two or three such copies among their four polyhedra and enumerated 3^12 equivalent
  // assignments: 2366 -> 13 nodes, 2353 -> 19, 2368 -> 9 on the dumped cases, same optimum.) A trajectory that is feasible with the
  // smaller polyhedron at a step is feasible with the larger one there, so the MIQP optimum does not need the smaller one:
  // polyhedron j is dominated if it is contained in some k and (k is not contained in j, or k < j — of two copies the first stays).
  // Directions are compared exactly up to 1e-12 (rows of the voxel decomposition carry small integers), right-hand sides with
  // 1e-10 max(1, |b|): a tenth of the solver's own feasibility tolerance.
  static HD unsigned dominated_mask(const S& s, int np, int lane) {
    // item = (ordered pair (j, k), row r of k), 32 items per pair: two pairs per trip of the wavefront
    unsigned long long sub = 0ull;  // bit j * np + k: P_j is contained in P_k
    const int pairs = np * np;
    for (int p0 = 0; p0 < pairs; p0 += 2) {
      const int p = p0 + (lane >> 5), r = lane & 31;
      const int j = p < pairs ? p / np : 0, k = p < pairs ? p - j * np : 0;
      const int rows_k = s.sp_rows[k], rows_j = s.sp_rows[j];
      bool implied = true;
      if (p < pairs && j != k && r < rows_k) {
        const double a0 = s.sp[k][r][0], a1 = s.sp[k][r][1], a2 = s.sp[k][r][2], b = s.sp[k][r][3];
        const double amax = fmax(fabs(a0), fmax(fabs(a1), fabs(a2)));
        implied = amax == 0.0 && b >= 0.0;  // (a zero row 0 . p <= b holds everywhere)
        const double bb = b + 1e-10 * fmax(1.0, fabs(b));
        for (int q = 0; q < rows_j && !implied; ++q) {
          const double c0 = s.sp[j][q][0], c1 = s.sp[j][q][1], c2 = s.sp[j][q][2], d = s.sp[j][q][3];
          const double cmax = fmax(fabs(c0), fmax(fabs(c1), fabs(c2)));
          const double x0 = a1 * c2 - a2 * c1, x1 = a2 * c0 - a0 * c2, x2 = a0 * c1 - a1 * c0;
          const double dot = a0 * c0 + a1 * c1 + a2 * c2, cc = c0 * c0 + c1 * c1 + c2 * c2;
          const bool parallel = fmax(fabs(x0), fmax(fabs(x1), fabs(x2))) <= 1e-12 * amax * cmax && dot > 0.0;
          implied = parallel && dot * d <= bb * cc;  // t = dot / cc:  t d <= b (+ tolerance)
        }
      }
      if (p >= pairs || j == k) implied = false;
      const unsigned long long ok = __ballot(implied || (p < pairs && j != k && r >= rows_k));  // (rows beyond the polyhedron's count: nothing to imply)
      for (int h = 0; h < 2; ++h) {
        const int ph = p0 + h;
        if (ph < pairs && ph / np != ph % np && (unsigned)(ok >> (32 * h)) == 0xffffffffu && s.sp_rows[ph % np] > 0 && s.sp_rows[ph / np] > 0) sub |= 1ull << ph;
      }
    }
    unsigned dom = 0u;
    for (int j = 0; j < np; ++j)
      for (int k = 0; k < np; ++k)
        if (k != j && ((sub >> (j * np + k)) & 1ull) && (!((sub >> (k * np + j)) & 1ull) || k < j)) dom |= 1u << j;
    if (dom == (1u << np) - 1u) dom = 0u;  // (cannot happen with a consistent containment relation: never leave an instance without a choice)
    return dom;
  }

  // ---- leaf test: which unassigned steps lie in no polyhedron -----------------------------------------------
  // keys[i][j] = max row violation of polyhedron j on (p_i, p_{i+1}); DINF if an input-independent end point is outside.
  // Returns the step to branch on, -1: every step lies in a polyhedron (a leaf), -2: steps were ASSIGNED here (one polyhedron is left
  // after the dominated ones are gone: every uncontained step must take it) — the node's run continues with their rows.
  static HD int leaf_check(S& s, const Consts& c) {
    const int N = c.N, np = s.n_poly;
#ifdef HDSM_LEAF_MFMA
    // Opt-in build (-DHDSM_LEAF_MFMA): measured on MI355X it is performance-neutral on every workload tried but costs the
    // two-workgroups-per-CU kernel spilled VGPRs (scratch 24 -> 68 B/lane, HBM writes 3.9 -> 10.4 MB per launch), so the
    // default build keeps the scalar leaf test. Counters of both builds: profiles/r02_mfma_ab.json.
    if (c.leaf_mfma && N <= 15 && blockDim.x == 256) {
      // The slack of every static row at every trajectory point is ONE matrix product in homogeneous coordinates:
      //   D[r][m] = (a_r, -b_r) . (p_m, 1),   rows r of a polyhedron (tiles of 16), points m = 0..N (<= 16 columns), K = 4
      // — exactly the shape of v_mfma_f64_16x16x4_f64 (A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
      // D: col = lane & 15, row = (lane >> 4) + 4 reg). Wave w takes polyhedra w and w + 4; the row maxima per point are
      // reduced in the lane (4 registers x 2 tiles) and across the four 16-lane groups, and land in red_v[j * 16 + m].
      using v4d = double __attribute__((ext_vector_type(4)));
      const int lane = (int)HDSM_TX & 63, w = (int)HDSM_TX >> 6;
      const int col = lane & 15, kk = lane >> 4;
      const double bop = (kk < 3) ? ((col <= N) ? s.st[col][kk] : 0.0) : 1.0;
      for (int j = w; j < np; j += 4) {
        const int rows = s.sp_rows[j];
        double pmax = -DINF;
        for (int t = 0; 16 * t < rows; ++t) {
          const int r = 16 * t + col;
          const double aop = (r < rows) ? ((kk < 3) ? s.sp[j][r][kk] : -s.sp[j][r][3]) : ((kk < 3) ? 0.0 : -1e300);
          v4d acc = {0.0, 0.0, 0.0, 0.0};
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc, 0, 0, 0);
          pmax = fmax(pmax, fmax(fmax(acc[0], acc[1]), fmax(acc[2], acc[3])));
        }
        pmax = fmax(pmax, __shfl_xor(pmax, 16));
        pmax = fmax(pmax, __shfl_xor(pmax, 32));
        if (lane < 16) s.red_v[j * 16 + lane] = pmax;
      }
      SYNC();
      PAR_FOR(idx, N * np) {
        const int i = idx / np, j = idx % np;
        double vmax = -DINF;
        if (s.assign[i] < 0) {
          const double v0 = s.red_v[j * 16 + i], v1 = s.red_v[j * 16 + i + 1];
          // rows on input-independent points only gate the choice
          const bool g0 = i <= c.pinned_steps, g1 = i + 1 <= c.pinned_steps;
          if ((g0 && v0 > c.ftol_fixed) || (g1 && v1 > c.ftol_fixed)) vmax = DINF;
          else vmax = g1 ? -DINF : (g0 ? v1 : (v0 > v1 ? v0 : v1));
        }
        s.keys[i][j] = vmax;
      }
      SYNC();
    } else
#endif
    if (N * np <= 64) {
      // The usual shape (N np <= 64): ONE wavefront, lane = (step i, polyhedron j), both end points of the segment against every
      // row of j, then the containing polyhedron per step from a ballot and the step to branch on from one wave reduction — no
      // workgroup barrier inside (the version below spends four, and two runtime divisions per item: 2.7 us per leaf test on
      // the bench rounds against 0.4 us for this one). The other wavefronts wait at the barrier that publishes the result.
      if (HDSM_TX < 64) {
        const int lane = tid_here();
        const bool on = lane < N * np;
        // (N np <= 64: the quotient by a small runtime np is exact in single precision)
        const int i = on ? (int)(((float)lane + 0.5f) * (1.0f / (float)np)) : 0, j = on ? lane - i * np : 0;
        const int ai = s.assign[i];
        double vmax = -DINF;
        unsigned dom = (unsigned)uni(s.sp_dom);
        if (on && ai < 0 && ((dom >> j) & 1u)) {
          vmax = DINF;  // a dominated polyhedron: neither a container nor a choice
        } else if (on && ai < 0) {
          const int rows = s.sp_rows[j];
          const double* p0 = s.st[i];
          const double* p1 = s.st[i + 1];
          const double ax_ = p0[0], ay_ = p0[1], az_ = p0[2], bx_ = p1[0], by_ = p1[1], bz_ = p1[2];
          double v0max = -DINF, v1max = -DINF;
          // rows on input-independent points only gate the choice
          const bool g0 = i <= c.pinned_steps, g1 = i + 1 <= c.pinned_steps;
          for (int r = 0; r < rows; ++r) {
            const double* row = s.sp[j][r];
            const double r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3];
            const double v0 = r0 * ax_ + r1 * ay_ + r2 * az_ - r3, v1 = r0 * bx_ + r1 * by_ + r2 * bz_ - r3;
            v0max = v0 > v0max ? v0 : v0max, v1max = v1 > v1max ? v1 : v1max;
          }
          if ((g0 && v0max > c.ftol_fixed) || (g1 && v1max > c.ftol_fixed)) vmax = DINF;
          else vmax = g1 ? -DINF : (g0 ? v1max : (v0max > v1max ? v0max : v1max));
        }
        // lane i < N: its step
        const int my_a = lane < N ? s.assign[lane] : 0;
        int cont = 0;
        double best = -DINF;  // smallest violation among the polyhedra of an uncontained step (-DINF: contained / no step)
        unsigned long long open = 0ull;
        for (int pass = 0;; ++pass) {
          if (on) s.keys[i][j] = vmax;
          const unsigned long long inside = __ballot(on && vmax <= c.tol);
          const unsigned fits = lane < N ? (unsigned)((inside >> (lane * np)) & ((1ull << np) - 1ull)) : 1u;
          cont = (lane < N) ? (my_a >= 0 ? my_a : (fits != 0u ? __builtin_ffs((int)fits) - 1 : -1)) : 0;
          if (lane < N) s.contain[lane] = cont;
          wsync();
          best = -DINF;
          if (lane < N && cont < 0) {
            best = DINF;
            for (int jj = 0; jj < np; ++jj) best = s.keys[lane][jj] < best ? s.keys[lane][jj] : best;
          }
          open = __ballot(lane < N && cont < 0);
          // the first time this instance would have to branch: which polyhedra are contained in another one (dominated_mask)
          if (pass > 0 || open == 0ull || np < 2 || c.dominance == 0 || uni(s.dom_done) != 0) break;
          dom = dominated_mask(s, np, lane);
          wsync();
          if (lane == 0) s.dom_done = 1, s.sp_dom = (int32_t)dom;
          if (dom == 0u) break;
          if (on && ai < 0 && ((dom >> j) & 1u)) vmax = DINF;  // (what lies in a dominated polyhedron lies in its dominator: `open` stays as it is)
          wsync();
        }
        // ONE polyhedron left: every uncontained step has to take it — assigned here, all at once, without a level and without a node
        // (the assignment holds at every node of the instance; it is made at the root, where the mask is computed, or found again by an item)
        const unsigned alive = ~dom & ((1u << np) - 1u);
        bool forced = false;  // (wave-uniform)
        // (also when the instance came with one polyhedron: step-by-step branching with one child per level was a node per step)
        if (open != 0ull && c.dominance != 0 && (alive & (alive - 1u)) == 0u) {
          const int j0 = __builtin_ffs((int)alive) - 1;
          // (a step whose pinned end point lies outside j0 has no admissible polyhedron at all: the ordinary path below reports it — no child)
          const bool admissible = __ballot(lane < N && cont < 0 && !(s.keys[lane][j0] < DINF)) == 0ull;
          if (admissible) {
            if (lane < N && cont < 0) s.assign[lane] = j0;
            if (lane == 0) s.leaf_pick = -2, s.leaf_lb = 0, s.first_id = -1, s.forced = 1;
            wsync();
            open = 0ull, forced = true;
          }
        }
        int pick = -1;
        if (open != 0ull) {
          // Child bound — a second walk over the rows, only at a node that WILL branch (in open space no instance ever gets here).
          // The node's minimiser x_p is optimal for the rows of the node; a child must also satisfy row r of the polyhedron it
          // assigns, violated by v at x_p: every point that does is at least v / sqrt(a^T Z a) away from x_p in the metric of the
          // problem, so the child's optimum is >= f + v^2 / (2 a^T Z a) (Z: inverse Hessian on the null space of the terminal
          // equalities; a^T Z a = sum_ax n_ax^2 kap[m][ax] for a row on point m — the quantity behind the pick rule). The best such
          // ratio over the rows and both end points is kept by cross-multiplication (one division per lane); the row that attains it
          // is the child's first entering row.
          if (on && ai < 0) {
            const int rows = s.sp_rows[j];
            const double* p0 = s.st[i];
            const double* p1 = s.st[i + 1];
            const double ax_ = p0[0], ay_ = p0[1], az_ = p0[2], bx_ = p1[0], by_ = p1[1], bz_ = p1[2];
            const bool g0 = i <= c.pinned_steps, g1 = i + 1 <= c.pinned_steps;
            double bv2 = 0.0, bq = 1.0, bv = 0.0;
            int bcode = -1;
            const double k00 = s.kap[i][0], k01 = s.kap[i][1], k02 = s.kap[i][2];
            const double k10 = s.kap[i + 1][0], k11 = s.kap[i + 1][1], k12 = s.kap[i + 1][2];
            for (int r = 0; r < rows; ++r) {
              const double* row = s.sp[j][r];
              const double r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3];
              const double v0 = r0 * ax_ + r1 * ay_ + r2 * az_ - r3, v1 = r0 * bx_ + r1 * by_ + r2 * bz_ - r3;
              const double s0 = r0 * r0, s1 = r1 * r1, s2 = r2 * r2;
              const double q0 = s0 * k00 + s1 * k01 + s2 * k02, q1 = s0 * k10 + s1 * k11 + s2 * k12;
              if (!g0 && v0 > c.tol && q0 > 1e-60 && v0 * v0 * bq > bv2 * q0) bv2 = v0 * v0, bq = q0, bv = v0, bcode = r;
              if (!g1 && v1 > c.tol && q1 > 1e-60 && v1 * v1 * bq > bv2 * q1) bv2 = v1 * v1, bq = q1, bv = v1, bcode = 64 | r;
            }
            s.pk_at(lane) = c.child_bound != 0 ? bcode : -1, s.pv_at(lane) = bv;
            s.lb_at(lane) = c.child_bound != 0 ? 0.5 * (bv2 / bq) * (1.0 - 1e-9) : 0.0;  // (rounded towards "no bound")
          }
          wsync();
          if (c.branch_rule == 0) {
            pick = __ffsll((long long)open) - 1;  // the first in time
          } else {
            const double worst = wave_max64(best);  // the most infeasible one; ties: the earlier step
            pick = __ffsll((long long)__ballot(lane < N && cont < 0 && best == worst)) - 1;
          }
        }
        // Node bound: EVERY uncontained step must still take a polyhedron, so the cheapest admissible choice of any one of them
        // bounds the whole subtree: f + max_i min_j klb[i][j]. Kept as the largest and the second largest of those minima (and
        // whose step the largest is), so that a child of step i gets max(its own klb, the best of the OTHER steps).
        double mine = 0.0;
        if (open != 0ull && lane < N && cont < 0) {
          mine = DINF;
          for (int jj = 0; jj < np; ++jj)
            if (s.keys[lane][jj] < DINF) mine = s.lb_at(lane * np + jj) < mine ? s.lb_at(lane * np + jj) : mine;
          if (!(mine < DINF)) mine = 0.0;  // (no admissible polyhedron: the branching finds no child anyway)
        }
        double top1 = 0.0, top2 = 0.0;
        int l1 = -1;
        if (open != 0ull) {  // (wave-uniform)
          top1 = wave_max64(mine);
          l1 = top1 > 0.0 ? __ffsll((long long)__ballot(mine == top1)) - 1 : -1;
          top2 = wave_max64(lane == l1 ? 0.0 : mine);
        }
        if (lane == 0 && !forced) s.leaf_pick = pick, s.lb_top1 = top1, s.lb_top2 = top2, s.lb_step = l1, s.leaf_lb = 1;
      }
      SYNC();
      return s.leaf_pick;
    }
    {
      // One item = (step i, polyhedron j, end point e, half of the rows): up to 4 N np items of a handful of rows each — the
      // loop over all rows at both points used to sit in N np threads (2.2 us per leaf test). Partial maxima go through
      // red_v (MAXT entries: with more than 4 polyhedra the rows are not halved).
      const int K = (N * np * 4 <= MAXT) ? 4 : 2, halves = K / 2;
      PAR_FOR(idx, N * np * K) {
        const int t = idx % K, e = t & 1, rc = t >> 1, ij = idx / K, i = ij / np, j = ij % np;
        double vmax = -DINF;
        if (s.assign[i] < 0) {
          const int rows = s.sp_rows[j], per = (rows + halves - 1) / halves, r0 = rc * per, r1 = r0 + per < rows ? r0 + per : rows;
          const double* pm = s.st[i + e];
          const double px = pm[0], py = pm[1], pz = pm[2];
          for (int r = r0; r < r1; ++r) {
            const double* row = s.sp[j][r];
            const double v = row[0] * px + row[1] * py + row[2] * pz - row[3];
            if (i + e <= c.pinned_steps) {
              if (v > c.ftol_fixed) vmax = DINF;  // rows on input-independent points only gate the choice
            } else if (v > vmax) {
              vmax = v;
            }
          }
        }
        s.red_v[idx] = vmax;
      }
      SYNC();
      PAR_FOR(ij, N * np) {
        double vmax = s.red_v[K * ij];
        for (int t = 1; t < K; ++t) vmax = s.red_v[K * ij + t] > vmax ? s.red_v[K * ij + t] : vmax;
        s.keys[ij / np][ij % np] = vmax;
      }
    }
    SYNC();
    PAR_FOR(i, N) {  // one thread per step: lowest-index polyhedron containing the segment
      int cont = -1;
      if (s.assign[i] >= 0) {
        cont = s.assign[i];
      } else {
        for (int j = 0; j < np; ++j)
          if (s.keys[i][j] <= c.tol) {
            cont = j;
            break;
          }
      }
      s.contain[i] = cont;
    }
    SYNC();
    // (no node / child bound on this path — more than 64 (step, polyhedron) pairs: leaf_lb = 0 says so, and whoever reads lb_top1 /
    // lb_top2 / lb_step looks at it first. Writing zeros into the two doubles here instead cost the shared-CU kernels their only scratch
    // access: the 16-byte zero was hoisted in front of the tree loop as a loop-invariant constant and SPILLED, 20 B/lane)
    if (IS_T0) s.leaf_lb = 0;
    // the step to branch on among those whose segment lies in no polyhedron: the first in time (rule 0), or the MOST
    // infeasible one — largest violation of its best polyhedron (rule 1). Any choice is exact; it only shapes the tree.
    int pick = -1;
    double worst = -DINF;
    for (int i = N - 1; i >= 0; --i) {
      if (s.contain[i] >= 0) continue;
      if (c.branch_rule == 0) {
        pick = i;
        continue;
      }
      double best = DINF;
      for (int j = 0; j < np; ++j) best = s.keys[i][j] < best ? s.keys[i][j] : best;
      if (best >= worst) worst = best, pick = i;  // ties: the earlier step (the scan runs backwards)
    }
    return pick;
  }

  // ---- snapshots of the solver state (global scratch), one per branching depth ----------------------------------
  // rows of J in registers, columns in butterfly order (hdsm_wave_gib.h): n <= 30 split over two lanes, larger n one lane per row
  using W = WaveGIB<NV, CMAX, SMALL>;
  using GIState = typename W::Regs;
  static constexpr int SNAP_STRIDE = W::SNAP_DOUBLES + 2;  // doubles per level
  // The factorisation lives in the registers of wave 0; the other waves of the workgroup (they take part in the
  // sweeps, the set-up and the leaf test) wait at the barrier and pick the outcome up from LDS.
  static HD void snapshot_io(S& s, const Consts& c, GIState& R, double* buf, bool save) {
    // (the per-lane addresses of a snapshot are formed here, when one is taken: hoisted out of the branch-and-bound loop they
    // were kept alive across the whole active-set run — 37 dwords per lane spilled to scratch by EVERY instance, tree or not)
    buf = keep_in_loop(buf);
    // (workgroups of two or more wavefronts: wave 1 moves U — LDS <-> global — while wave 0 moves J from / to its registers)
    if (blockDim.x > 64) {
      if (HDSM_TX < 64) W::template snapshot<false>(s, R, buf, save, tid_here());
      else if (HDSM_TX < 128) W::snapshot_u(s, buf, save, tid_here() & 63);
    } else if (HDSM_TX < 64) {
      W::template snapshot<true>(s, R, buf, save, tid_here());
    }
    SYNC();
  }
  // No point of the input box has an objective above f_box (set-up), and the dual method's f only grows: once it passes f_box the
  // working set together with the row on its way in cannot be satisfied inside the box — infeasible, proven without driving the
  // multipliers to the formal dependency (the longest proofs of the bench rounds spend their last five operations there). The
  // run sees one cut-off, min(incumbent bound, f_box); which of the two it was is sorted out here.
  static HD int gi_run(S& s, const Consts& c, GIState& R, double f_cut, int& iters) {
    const double f_box = s.f_box, f_eff = f_cut < f_box ? f_cut : f_box;
    if (s.f >= f_eff) {  // (a warm start can arrive above the bound: its guess was a certificate)
      const int rc0 = s.f >= f_cut ? GI_CUTOFF : GI_INFEASIBLE;
      SYNC();
      if (IS_T0 && rc0 == GI_INFEASIBLE) s.inf_id = s.act[s.q - 1];
      SYNC();
      return rc0;
    }
    if (HDSM_TX < 64) {
      const int rc = W::run(s, c, R, f_eff, iters);
      if (HDSM_TX == 0) s.rc = (rc == GI_CUTOFF && !(s.f >= f_cut)) ? GI_INFEASIBLE : rc, s.iters_sh = iters;
    } else {
      W::helper_loop(s, c, R);  // n <= 30: wave 1 evaluates and picks while wave 0 updates; larger n: waves 1..3 share long scans
    }
    SYNC();
    iters = s.iters_sh;
    return s.rc;
  }

  static HD double cutoff(const S& s, const Consts& c) {
    // the incumbent of this workgroup, or — pass 2 of a split launch — the best one of all sub-blocks of the instance
    // (s.inc_shared: refreshed from global memory by thread 0 at every node, see select_child)
    const double inc = s.have_inc ? (s.inc_f < s.inc_shared ? s.inc_f : s.inc_shared) : s.inc_shared;
    if (!(inc < DINF)) return DINF;
    const double exact = 1e-9 * fmax(1.0, fabs(inc)), gap = c.mip_gap * fabs(inc);  // hdsm_params.mip_gap (Gurobi MIPGap)
    return inc - (gap > exact ? gap : exact);
  }

  // Moves to the next unexplored child of the deepest open level: restores the parent's solver state and
  // assigns the child's polyhedron. Returns false when the tree is exhausted (or the node budget is).
  static HD bool select_child(S& s, const Consts& c, GIState& R, int& nodes, bool& limit, int inst, bool& handed_over, int& rec_slot) {
    if (s.args.item_mode) {  // pass 2 of a split launch: what the other items of this instance have found
      SYNC();
      if (IS_T0) {
        const unsigned long long bits = __hip_atomic_load(&s.args.inc_bits[inst], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s.inc_shared = __longlong_as_double((long long)bits);
      }
      SYNC();
    }
    for (;;) {
      const int level = s.level;
      if (level == 0) return false;
      const int L = level - 1;
      const int pos = s.br_pos[L];
      const double cut = cutoff(s, c);
      if (pos < s.br_cnt[L] && !(s.br_f[L] >= cut)) {
        if (s.br_lb[L][pos] >= cut) {  // the child's lower bound (leaf_check) reaches the incumbent: neither opened nor counted
          SYNC();
          if (IS_T0) s.br_pos[L] = pos + 1, ++s.lb_skipped;
          SYNC();
          continue;
        }
        const int n_ng = uni(s.n_nogood);
        if (n_ng > 0) {  // a child whose assignments contain a learned conflict is infeasible: not opened, not counted
          // (lane-parallel in every wavefront: the assignments of the child as a mask from four ballots — bit 16 j + i = polyhedron j
          // at step i — and one learnt conflict per lane; one thread walking both lists was 4 k cycles of LDS round trips per node)
          const int ln = tid_here() & 63, bs = uni(s.br_step[L]), bj = uni(s.br_order[L][pos]);
          const int aj = ln < c.N ? (ln == bs ? bj : s.assign[ln]) : -1;
          unsigned long long cur = 0ull;
          for (int j = 0; j < 4; ++j) cur |= (__ballot(aj == j) & 0xffffull) << (16 * j);
          const unsigned long long mine = ln < n_ng ? s.nogood[ln] : ~0ull;
          const bool blocked = __ballot((mine & ~cur) == 0ull) != 0ull;
          if (blocked) {
            SYNC();
            if (IS_T0) s.br_pos[L] = pos + 1, ++s.ng_skipped;
            SYNC();
            continue;
          }
        }
        if (s.args.split_budget > 0 && (nodes >= s.args.split_budget || (s.args.item_mode && nodes >= s.args.split_min && (nodes & 1) == 0))) {
          // a split launch: hand the search over — pass 1 after its node budget; an item of pass 2 after its own, or, from
          // split_min nodes on, as soon as the queue is empty (workgroups are waiting for items: looked at every other node)
          // (if there is a record slot left and the staged rows fit the staging area of pass 2 — otherwise the search goes on here)
          SYNC();
          if (IS_T0) {
            int slot = -1;
            bool want = nodes >= s.args.split_budget;
            if (!want) {
              int q = __hip_atomic_load(&s.args.rec_count[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              q = q < s.args.items_cap ? q : s.args.items_cap;
              want = __hip_atomic_load(&s.args.rec_count[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= q;  // (tickets beyond the queue: workgroups wait)
            }
            if (want) {
              if (s.ncand + s.ncold <= s.args.rows_cap) {
                slot = atomicAdd(&s.args.rec_count[0], 1);
                if (slot >= s.args.rec_cap) slot = -1;
              }
              if (slot < 0) s.args.split_budget = 0;  // (no record left, or rows that do not fit: this search stays here to its end)
            }
            s.iters_sh = slot;
          }
          SYNC();
          const int slot = uni(s.iters_sh);
          SYNC();
          if (slot >= 0) {
            rec_slot = slot;
            handed_over = true;
            return false;
          }
        }
        if (s.args.item_mode) {  // pass 2 of a split launch: own share first, then what finished items handed back
          SYNC();
          if (IS_T0 && s.node_res == 0) {  // (compare-and-swap: the pool never goes negative, nothing handed back later is lost)
            int cur = __hip_atomic_load(&s.args.node_pool[inst], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), got = 0;
            while (cur > 0) {
              const int take = cur < NODE_CHUNK ? cur : NODE_CHUNK;
              const int old = atomicCAS(&s.args.node_pool[inst], cur, cur - take);
              if (old == cur) {
                got = take;
                break;
              }
              cur = old;
            }
            s.node_res = got;
          }
          SYNC();
          if (s.node_res == 0) {
            limit = true;
            return false;
          }
          SYNC();
          if (IS_T0) --s.node_res;
        } else if (nodes >= c.max_nodes) {
          limit = true;
          return false;
        }
        ++nodes;
        const int j = s.br_order[L][pos];
        SYNC();
        NODE_PROF(22)
        if (pos > 0) snapshot_io(s, c, R, s.snap + (int64_t)L * SNAP_STRIDE, false);
        NODE_PROF(23)
        if (IS_T0) {
          s.br_pos[L] = pos + 1;
          s.assign[s.br_step[L]] = j;
          const int code = s.br_pk[L][pos];
          s.first_id = code >= 0 ? mk_id(K_P, (s.br_step[L] << 7) | code) : -1, s.first_v = s.br_pv[L][pos];
        }
        SYNC();
        return true;
      }
      SYNC();
      if (IS_T0) {
        s.assign[s.br_step[L]] = -1;
        s.level = L;
      }
      SYNC();
    }
  }

  // ---- one instance, start to finish ------------------------------------------------------------------------
  // `inst`: the instance (inputs, warm-start guess); `out`: where its outputs and statistics live — `inst` itself, or, in pass 2 of a
  // split launch, the item's index in the queue; `item`: -1 = an ordinary solve, else (record << 8) | (level << 3) | child position:
  // continue the handed-over search of the record's instance inside the subtree of that child (Args, SplitRec).
  // `self_in`: the agent id of the instance if the caller already has it (launch order pairs), -1 = agent_id[inst]
  // `args_words`: the launch arguments once more, as the 8-byte words of the kernel-argument segment (device kernels; null: copy `a_in`).
  // The copy of the arguments in LDS is made from THEM, one word per thread: as a struct assignment from the by-value kernel parameter
  // it went through a private copy of the whole struct (256 bytes of scratch per lane) as soon as the kernel did anything with
  // ordering semantics — the atomics of the item queue — before it.
  static HD void solve_instance(S& s, const Consts& c, const Args& a_in, int inst, int out, int item, int self_in, int& wg_slot,
                                const HDSM_KERNARG_WORD* args_words = nullptr) {
#ifdef HDSM_POISON_LDS
    // test builds (tests/wave_emu, scripts/gpu_poison.sh): LDS is not cleared between workgroups — whatever is read before it is
    // written shows up as NaN / garbage here instead of depending on the kernel that ran on the CU before
    for (int i = (int)HDSM_TX; i < (int)(sizeof(S) / 8); i += (int)blockDim.x) reinterpret_cast<unsigned long long*>(&s)[i] = 0xFFF8DEADBEEF0BADull;
    SYNC();
#endif
    if (args_words != nullptr) {
      static_assert(sizeof(Args) % 8 == 0 && sizeof(Args) / 8 <= 64, "one word of the launch arguments per thread of the smallest workgroup");
      const int t = (int)HDSM_TX;
      if (t < (int)(sizeof(Args) / 8)) reinterpret_cast<long long*>(&s.args)[t] = args_words[t];
    } else if (IS_T0) {
      s.args = a_in;
    }
    SYNC();
    const Args& a = s.args;
    const int N = c.N, n = c.n, P = c.P, RS = c.RS;
    const int self = self_in >= 0 ? self_in : a.agent_id[inst];
    // snapshot scratch: the instance's own, or — pass 2, persistent workgroups — the slot this workgroup holds at the moment (`wg_slot`)
    double* snap = a.scratch + (int64_t)(item >= 0 ? wg_slot : out) * a.scratch_stride;
    if (IS_T0) s.snap = snap;  // (read back at the few places a snapshot is taken or restored; published by the set-up's barriers)

    const long long tl_begin_ = (long long)wall_clock64();  // constant-rate clock (100 MHz), common to all CUs
#ifdef HDSM_TIMELINE
    long long tl_su1_ = 0, tl_su2_ = 0, tl_tail1_ = 0, tl_tail2_ = 0;
#endif
#ifdef HDSM_PROFILE
    const long long t_begin_ = clock64();
    long long t_sweep_ = 0, t_leaf_ = 0;
    if (HDSM_TX < 24) s.prof_acc[HDSM_TX] = 0;
    SYNC();
    PROF_DECL
#ifdef HDSM_PROF_OP
#define SU_PROF(k)
#else
#define SU_PROF(k) PROF(k)
#endif
#else
#define SU_PROF(k)
#endif
    // ---- stage the instance in LDS and apply the set-up map. Every global read of the set-up is REQUESTED before
    // the first one is consumed (memory latency is paid once, or twice for the own plan, whose address needs
    // agent_id[inst]); the generic loops of the CPU build above do the same thing one array at a time.
    GIState R;
    typename W::WarmPre wpre{0, 0, 0, 0.0, 0.0, 0.0};
    const int nk = 3 * n + 12, nvt = 9 + 6 * N;
    double fw0, fw1;  // weights of the (at most two) tracking residuals this lane of wave 0 squares for the constant term
    double hrow_own;  // Consts::hrow1 of this lane's variable (the box bound of the objective, below)
    const int np = uni(min_i(a.n_poly[inst], P));  // (wave-uniform: a scalar register)
    {
      const int tid = (int)HDSM_TX, nt = (int)blockDim.x;
      constexpr int KH = 3 + 2 * (NV / 3);  // inputs one output of the set-up map depends on (compact form, KTC)
      // one output of the set-up map per thread: its coefficients
      double kv[KH];
      const bool k_on = tid < nk;
      const int krow = k_on ? tid : 0, kbase = c.kax[krow];
      // (Args::setup: the outputs of the map for every instance of the launch, formed by the pre-pass kernel as one dense product on
      // the matrix cores — then a thread asks for ONE number here instead of its 23 coefficients)
      const bool pre_map = a_in.setup != nullptr;
      double pre_v = 0.0;
      if (pre_map) {
        if (k_on) pre_v = a_in.setup[(int64_t)inst * KROWS + krow];
      } else if (tid < ((nk + 63) & ~63)) {  // whole wavefronts only
        HDSM_UNROLL
        for (int u = 0; u < KH; ++u) kv[u] = c.KTC[u * KROWS + krow];
      }
      // wave 0 takes Jeq (identity beyond n) in the lane layout of its kernel (JeqP, see hdsm_wave_gi.h). With n <= 30
      // the registers hold this next to the coefficients; the larger kernel asks for it once those are consumed.
      if constexpr (NV <= 32) {
        if (tid < 64) {
          HDSM_UNROLL
          for (int j = 0; j < W::NC; ++j) R.Jr[j] = c.JeqP[j * 64 + tid];
        }
      }
      // Everything else is staged in "virtual thread" slots: slot vt takes item vt of every array. Round 0 (vt = tid)
      // is split into requests and LDS writes so that the two reads that need agent_id[inst] (the own plan) go out
      // while all the others are already in flight; the kernel arguments are used straight from the SGPRs here.
      const Args& g = a_in;
      struct Req {
        double v_in, v_g, a0, a1, a2, a3, p0, p1, p2, q0, q1, q2, v_u, v_b, v_k;
        int nr, sj, sr, fi, fcomp, fax;
      };
      auto request = [&](int vt) {
        Req r;
        // (config constants first: their addresses do not need `inst`, whose own load — the launch-order entry — is still in
        // flight when a workgroup starts; the reads that do need it follow, so nothing that could go out waits behind them)
        r.v_g = (vt < 9 * MAXH) ? (&c.g[0][0][0])[vt] : 0.0;
        r.fi = (vt < 9 * (N + 1)) ? vt / 9 : 0, r.fcomp = (vt % 9) / 3, r.fax = vt % 3;
        r.p0 = c.phi[r.fax][r.fi][r.fcomp][0], r.p1 = c.phi[r.fax][r.fi][r.fcomp][1], r.p2 = c.phi[r.fax][r.fi][r.fcomp][2];
        const int ui = vt / S::LDT, uj = vt % S::LDT;
        r.v_u = (ui < 6 && uj < 6) ? c.Ueq[ui * 6 + uj] : 0.0;
        r.v_b = (vt < 3) ? c.lbu[vt] : (vt < 6) ? c.ubu[vt - 3] : (vt < 15) ? (&c.lbs[0][0])[vt - 6]
                                                                : (&c.ubs[0][0])[vt < 24 ? vt - 15 : 0];
        r.v_k = (&c.kap[0][0])[vt < 4 * (MAXH + 1) ? vt : 0];
        r.v_in = (vt < nvt) ? (vt < 9 ? g.state[(int64_t)inst * 9 + vt] : g.ref[(int64_t)inst * 6 * N + (vt - 9)]) : 0.0;
        r.sj = (vt < P * RS) ? vt / RS : 0, r.sr = (vt < P * RS) ? vt % RS : 0;
        const double* Ar = g.A + (((int64_t)inst * P + r.sj) * RS + r.sr) * 3;
        r.a0 = Ar[0], r.a1 = Ar[1], r.a2 = Ar[2], r.a3 = g.b[((int64_t)inst * P + r.sj) * RS + r.sr];
        r.nr = g.n_rows[(int64_t)inst * P + (vt < P ? vt : 0)];
        r.q0 = g.state[(int64_t)inst * 9 + r.fax], r.q1 = g.state[(int64_t)inst * 9 + 3 + r.fax],
        r.q2 = g.state[(int64_t)inst * 9 + 6 + r.fax];
        return r;
      };
      auto commit = [&](int vt, const Req& r) {
        if (vt < KCOLS + 8) s.vin()[vt] = r.v_in;  // zero beyond 9 + 6N: padded coefficients (N < NV / 3) meet finite numbers
        if (vt < nvt) {
          if (vt < 9) s.state0[vt] = r.v_in;
          else (&s.ref[0][0])[vt - 9] = r.v_in;
        }
        if (vt < 9 * MAXH) {
          (&s.g[0][0][0])[vt] = r.v_g;
          const int ax = vt / (3 * MAXH), comp = (vt / MAXH) % 3, lag = vt % MAXH;
          s.gz[ax][comp][lag] = 0.0;
          s.gz[ax][comp][MAXH + lag] = (lag < N) ? r.v_g : 0.0;
        }
        if (vt < P * RS) s.sp[r.sj][r.sr][0] = r.a0, s.sp[r.sj][r.sr][1] = r.a1, s.sp[r.sj][r.sr][2] = r.a2, s.sp[r.sj][r.sr][3] = r.a3;
        if (vt < P) s.sp_rows[vt] = (vt < np) ? min_i(r.nr, RS) : 0;
        if (vt < 9 * (N + 1)) {
          s.fr[r.fax][r.fi][r.fcomp] = r.p0 * r.q0 + r.p1 * r.q1 + r.p2 * r.q2;
          if (r.fi == 0) s.st[0][3 * r.fcomp + r.fax] = (r.fcomp == 0) ? r.q0 : (r.fcomp == 1 ? r.q1 : r.q2);
        }
        if (vt < 6 * S::LDT) s.U[vt] = r.v_u;  // rows 0..5 of U = Ueq, zero-padded
        if (vt < NV) {
          if (vt >= 6) s.lam[vt] = 0.0;
          s.act[vt] = (vt < 6) ? mk_id(K_E, vt) : -1;
          if (vt >= n) s.x[vt] = 0.0;
        }
        if (vt < MAXH) s.assign[vt] = -1;
        if (vt < 24) {  // lbu[3], ubu[3], lbs[3][3], ubs[3][3]; an absent bound can never be violated
          const bool lower = vt < 3 || (vt >= 6 && vt < 15);
          s.bnd[vt] = fabs(r.v_b) < ABSENT ? r.v_b : (lower ? -DINF : DINF);
        }
        if (vt < 4 * (MAXH + 1)) (&s.kap[0][0])[vt] = r.v_k;
      };
      ST_PROF(8)
      // the guess of the warm start (wave 0, lane = entry): requested with everything else, not when the warm start begins
      if (g.warm != nullptr && tid < 64) {
        const int32_t* wp = g.warm + (int64_t)inst * (MAXNV + 2);
        wpre.head = wp[0], wpre.code = tid < MAXNV ? wp[1 + tid] : 0;
      }
      const typename W::LaneReq lane_req = W::init_lane_request(c, tid & 63);  // (every wave: wave 1 scans with its own copy)
      const int fi0 = tid / 6 + 1, fk0 = tid % 6, fi1 = (tid + 64) / 6 + 1, fk1 = (tid + 64) % 6;
      const double wn0 = c.wn[fk0], wx0 = c.wx[fk0], wn1 = c.wn[fk1], wx1 = c.wx[fk1];
      // The arrays that are longer than a workgroup of 128 threads are config constants only (impulse responses: 9 MAXH entries,
      // the rows of Ueq: 6 LDT): their later items are requested NOW, with everything else, instead of in a second round that
      // starts when the first one has been written to LDS (a second exposed round trip to global memory per instance).
      // everything else (the instance's arrays, the short constants) fits one round of 128 threads
      const int items_inst = max_i(max_i(P * RS, 9 * (N + 1)), max_i(KCOLS + 8, max_i(NV, 4 * (MAXH + 1))));
      constexpr int KC = ((6 * S::LDT > 9 * MAXH ? 6 * S::LDT : 9 * MAXH) + 63) / 64 - 1;  // enough for a workgroup of one wave
      double cg_late[KC], cu_late[KC];
      HDSM_UNROLL
      for (int u = 0; u < KC; ++u) {
        const int vt = tid + nt * (u + 1), ui = vt / S::LDT, uj = vt % S::LDT;
        cg_late[u] = (vt < 9 * MAXH) ? (&c.g[0][0][0])[vt] : 0.0;
        cu_late[u] = (vt < 6 * S::LDT && ui < 6 && uj < 6) ? c.Ueq[ui * 6 + uj] : 0.0;
      }
      const Req r0 = request(tid);

      ST_PROF(9)
      // the own plan (or, before the first plan exists, the current position at every step)
      const bool self_ok = self >= 0 && self < g.n_rob;
      const int sidx = self_ok ? self : 0;
      const uint8_t has_own = g.has_plan[sidx];
      for (int k = tid; k < 3 * N; k += nt) {
        const int ci = k / 3, cax = k % 3;
        const double from_plan = g.plans[((int64_t)sidx * (N + 1) + (ci + 1)) * 9 + cax];
        const double from_state = g.state[(int64_t)inst * 9 + cax];
        s.cprev[ci][cax] = (self_ok && has_own) ? from_plan : from_state;
      }
      ST_PROF(10)
      commit(tid, r0);
      // ... and, now that the guess is here, the neighbour positions its plane rows are rebuilt from (the set-up's remaining
      // steps hide this second round trip)
      if (g.warm != nullptr && g.l1_rows == nullptr && tid < 64) W::warm_prefetch(g, wpre, tid, self, N);
      ST_PROF(11)
      // (per-lane constants of the iteration: their loads use what they fetch at once, so they come after the staging requests)
      W::init_lane(R, c, tid & 63, lane_req);
      hrow_own = lane_req.hrow;
      fw0 = (tid < 64) ? ((fi0 == N) ? wn0 : wx0) : 0.0;
      fw1 = (tid < 64) ? ((fi1 == N) ? wn1 : wx1) : 0.0;
      HDSM_UNROLL
      for (int u = 0; u < KC; ++u) {  // the late items of the constant arrays (requested above)
        const int vt = tid + nt * (u + 1);
        if (vt < 9 * MAXH) {
          (&s.g[0][0][0])[vt] = cg_late[u];
          const int ax = vt / (3 * MAXH), comp = (vt / MAXH) % 3, lag = vt % MAXH;
          s.gz[ax][comp][lag] = 0.0;
          s.gz[ax][comp][MAXH + lag] = (lag < N) ? cg_late[u] : 0.0;
        }
        if (vt < 6 * S::LDT) s.U[vt] = cu_late[u];
      }
      static_assert(6 * S::LDT <= 64 * (KC + 1) && 9 * MAXH <= 64 * (KC + 1), "the late constant items cover the smallest workgroup");
      for (int vt = tid + nt; vt < items_inst; vt += nt) commit(vt, request(vt));  // (64-thread workgroups only)
      for (int k = 6 * S::LDT + tid; k < NV * S::LDT; k += nt) s.U[k] = 0.0;
      if (tid == 0) {
        s.n_poly = np, s.q = 6, s.neq_done = 6, s.ncand = 0, s.level = 0, s.have_inc = 0;
        s.fixed_bad = 0, s.overflow = 0, s.inc_f = DINF, s.ncold = 0, s.rc = 0, s.iters_sh = 0;
        s.warm_head = wpre.head;
        s.sp_dom = 0, s.dom_done = 0, s.forced = 0;
        s.st_sph = 0, s.st_pairs = 0, s.n_nogood = 0, s.ng_skipped = 0, s.ng_global = 0, s.lb_skipped = 0, s.first_id = -1, s.inc_shared = DINF, s.node_res = 0;  // (pass 2: every node after the item's own is drawn from the instance's pool)
        s.t_start = c.time_ticks > 0 ? (long long)wall_clock64() : 0;
      }
      ST_PROF(12)
#ifdef HDSM_TIMELINE
      tl_su1_ = (long long)wall_clock64();  // staging done (before the barrier)
#endif
      SYNC();
      SU_PROF(13)
      // the set-up map: x_eq (minimiser subject to v_N = a_N = 0), x0 (unconstrained minimiser), the gradient at
      // u = 0, the residual of the six terminal equalities at x0 and their multipliers
      const double* vin = s.vin();
      auto store = [&](int row, double acc) {
        if (row < n) s.x[row] = acc;
        else if (row < 2 * n) s.w[row - n] = acc;
        else if (row < 3 * n) s.grad[row - 2 * n] = acc;
        else if (row < 3 * n + 6) s.red_v[row - 3 * n] = acc;
        else s.lam[row - 3 * n - 6] = acc;
      };
      if (pre_map) {
        if (k_on) store(tid, pre_v);
        for (int row = tid + nt; row < nk; row += nt) store(row, a_in.setup[(int64_t)inst * KROWS + row]);
      } else {
        if (tid < ((nk + 63) & ~63)) {
          double acc = 0;
          HDSM_UNROLL
          for (int u = 0; u < KH; ++u) acc += kv[u] * vin[kbase + 3 * u];
          if (k_on) store(tid, acc);
        }
        for (int row = tid + nt; row < nk; row += nt) {  // workgroups with fewer threads than outputs
          const int base = c.kax[row];
          double acc = 0;
          for (int u = 0; u < 3 + 2 * N; ++u) acc += c.KTC[u * KROWS + row] * vin[base + 3 * u];
          store(row, acc);
        }
      }
      if constexpr (NV > 32) {
        if (tid < 64) {
          HDSM_UNROLL
          for (int j = 0; j < W::NC; ++j) R.Jr[j] = c.JeqP[j * 64 + tid];
        }
      }
    }
    SU_PROF(14)
#ifdef HDSM_TIMELINE
    tl_su2_ = (long long)wall_clock64();  // set-up map applied
#endif
    SYNC();
    // constant term f0, J(x0) = f0 + grad.x0 / 2 and J(x_eq) = J(x0) + resid.nu / 2
    R.xi = ((int)HDSM_TX < NV) ? s.x[HDSM_TX] : 0.0;
    PAR_FOR(k, NV) s.x0[k] = (k < n) ? s.w[k] : 0.0;
    if (HDSM_TX < 64) {
      const int lane = (int)HDSM_TX;
      double t0 = 0.0;
      if (lane < 6 * N) {
        const int i = lane / 6 + 1, k = lane % 6;
        const double e = s.fr[k % 3][i][k / 3] - s.ref[i - 1][k];
        t0 = fw0 * e * e;
      }
      if (lane + 64 < 6 * N) {
        const int i = (lane + 64) / 6 + 1, k = (lane + 64) % 6;
        const double e = s.fr[k % 3][i][k / 3] - s.ref[i - 1][k];
        t0 += fw1 * e * e;
      }
      t0 = wave_sum64(t0);
      const double t1 = wave_sum64(lane < n ? s.grad[lane] * s.w[lane] : 0.0);
      const double t2 = wave_sum64(lane < 6 ? s.red_v[lane] * s.lam[lane] : 0.0);
      // the largest objective inside the input box, bounded from above: J(u) = J(x0) + (u - x0)^T H (u - x0) / 2 and
      // w^T H w <= sum_j (sum_k |H_jk|) w_j^2, |w_j| <= the distance from x0_j to the farther end of its interval (no box: no bound)
      double wm = 0.0;
      if (lane < n) wm = fmax(fabs(s.bnd[3 + lane / N] - s.w[lane]), fabs(s.bnd[lane / N] - s.w[lane]));
      const double t3 = wave_sum64(lane < n ? hrow_own * wm * wm : 0.0);
      if (lane == 0) {
        s.f_box = c.box_cut != 0 ? (t0 + 0.5 * t1 + 0.5 * t3) * (1.0 + 1e-9) + 1e-9 : DINF;
        s.f0 = t0;
        s.fx0 = t0 + 0.5 * t1;
        s.f = s.fx0 + 0.5 * t2;
        s.inc_nact = 0;
      }
    }
    SYNC();
    SU_PROF(15)

#ifdef HDSM_PROFILE
    const long long t_setup_ = clock64() - t_begin_;
#endif
#ifdef HDSM_TIMELINE
    const long long tl_setup_ = (long long)wall_clock64();  // (10-ns ticks; phases of the instance for scripts/timeline_fit.py)
    long long tl_warm_ = 0, tl_sweep_ = 0, tl_run_ = 0, tl_leaf_ = 0;
    int tl_warm_it_ = 0, tl_runs_ = 0;
#define TL_T0 const long long tl_t0_ = (long long)wall_clock64();
#define TL_ADD(acc) acc += (long long)wall_clock64() - tl_t0_;
#else
#define TL_T0
#define TL_ADD(acc)
#endif
    // ---- branch and bound (gi_run and sweep have exactly one call site each: they are inlined)
    int iters = 0, nodes = 1, sweeps = 0;
    // The guess is an infeasibility certificate (see the hand-over below), or the last replan ended on the gridlock test of
    // its sweep (a constant row violated): the neighbourhood was gridlocked. Such an instance sweeps FIRST, at the cold starting point: if
    // the gridlock persists the test ends it right there; the certificate's own minimiser is a far-away point, so nothing
    // is pre-staged around it afterwards.
    bool warm_cert = false;
    if (a.warm != nullptr && np > 0) warm_cert = (s.warm_head & WARM_CERT) != 0;  // (fetched by the set-up; published by its barriers)
#if defined(HDSM_PROFILE)
    long long t_warm_ = 0;
    int it_warm_ = 0;
#endif
    bool limit = false, handed_over = false;
    int rec_slot = -1;  // pass 1 of a split launch: the record this instance writes when it hands its search over
    unsigned flags = 0;
    bool run = np > 0;
    if (IS_T0) s.sw_tau = 0.0;
    // pass 2 of a split launch: this workgroup continues the search of a handed-over instance inside ONE open child of one of its
    // open levels. The set-up above has rebuilt what depends on the instance's inputs only; the record brings back what the
    // search had accumulated: staged rows in their slots (the working set of the snapshot names them), conflicts, the displacement
    // reference of the sweeps, the assignments of the path down to the item's level; the snapshot of that level is the solver state.
    bool item_run = false;
    if (item >= 0) {
      const SplitRec& rc = a.recs[item >> 8];
      const int L = (item >> 3) & 15, pos = item & 7;
      const int nh = rc.ncand, ncl = rc.ncold;
      const int64_t rbase = (int64_t)(item >> 8) * a.rows_cap;
      SYNC();
      PAR_FOR(k, nh + ncl) {  // hot rows keep their slots; the cold ones go back to the top of the staging area
        const int slot = k < nh ? k : CMAX - ncl + (k - nh);
        const double* src = a.rec_cand + (rbase + k) * 4;
        s.cand[slot][0] = src[0], s.cand[slot][1] = src[1], s.cand[slot][2] = src[2], s.cand[slot][3] = src[3];
        *reinterpret_cast<long long*>(&s.cand_mw[slot]) = a.rec_mw[rbase + k];  // (MW: 8 bytes, copied as they are)
        s.cand_src[slot] = a.rec_src[rbase + k];
      }
      PAR_FOR(k, 3 * (N + 1)) s.sw_ref[k / 3][k % 3] = rc.sw_ref[k / 3][k % 3];
      PAR_FOR(k, rc.n_nogood) s.nogood[k] = rc.nogood[k];
      PAR_FOR(l, L + 1) {
        s.br_step[l] = rc.br_step[l], s.br_cnt[l] = 0, s.br_pos[l] = 0, s.br_f[l] = rc.br_f[l];
        s.assign[rc.br_step[l]] = l < L ? rc.assign[rc.br_step[l]] : rc.br_order[L][pos];  // (rc.assign: the path at the hand-over)
      }
      if (IS_T0) {
        s.ncand = nh, s.ncold = ncl, s.n_nogood = rc.n_nogood, s.sw_tau = rc.sw_tau, s.level = L + 1;
        s.sp_dom = rc.sp_dom, s.dom_done = 1;  // (the search that handed over had branched: it had looked)
        const int code = rc.br_pk[L][pos];
        s.first_id = code >= 0 ? mk_id(K_P, (rc.br_step[L] << 7) | code) : -1, s.first_v = rc.br_pv[L][pos];
        const unsigned long long bits = __hip_atomic_load(&a.inc_bits[inst], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s.inc_shared = __longlong_as_double((long long)bits);
      }
      SYNC();
      item_run = run && nh + ncl <= CMAX && !(rc.br_lb[L][pos] >= cutoff(s, c));  // (the bound may have been reached since the hand-over)
      if (item_run) snapshot_io(s, c, R, const_cast<double*>(rc.snap) + (int64_t)L * SNAP_STRIDE, false);
      if (IS_T0) s.neq_done = 6;
      run = item_run;
      sweeps = rc.sweeps_done;  // (a staging sweep that pass 1 has made is not made again: the rows are here)
      nodes = item_run ? 1 : 0;
    }
    SYNC();
    // all neighbour rows near (first call) or violated at (later calls) the current point -> staging area; a staging
    // radius that overflows the LDS slots is tightened (it only decides what is pre-staged: exactness comes from the
    // verification sweeps)
    auto sweep_all = [&](int before, int before_cold) {
      double thresh = (sweeps == 0) ? c.cand_tau : -c.tol;
      if (sweeps > 0 && a.l1_rows == nullptr && s.sw_tau > 0) {
        // Verification: rows left unstaged by the staging sweep had slack >= sw_tau at sw_ref; a row's slack moves by
        // at most |n_f| |dp| with |n_f| <= sqrt(1 + (3 pert)^2) (the planes themselves are fixed during an instance).
        // If no trajectory point has moved further than that allows, nothing unstaged can be violated: no sweep.
        if (HDSM_TX < 64) {
          const int m = tid_here();
          double d2 = 0;
          if (m <= N) {
            const double ux = s.st[m][0] - s.sw_ref[m][0], uy = s.st[m][1] - s.sw_ref[m][1], uz = s.st[m][2] - s.sw_ref[m][2];
            d2 = ux * ux + uy * uy + uz * uz;
          }
          d2 = wave_max64(d2);
          if (m == 0) s.sw_d2 = d2;
        }
        SYNC();
        const double nf2 = 1.0 + 9.0 * c.pert * c.pert;
        if (nf2 * s.sw_d2 * (1.0 + 1e-9) <= s.sw_tau * s.sw_tau) {
          if (IS_T0) s.nviol = 0;
          SYNC();
          return;
        }
      }
      for (;;) {
        SYNC();
        if (IS_T0) s.nviol = 0;
        SYNC();
#ifdef HDSM_PROFILE
        const long long ts_ = clock64();
#endif
        TL_T0
        sweep(s, c, a, inst, self, thresh, sweeps == 0);
        TL_ADD(tl_sweep_)
#ifdef HDSM_PROFILE
        t_sweep_ += clock64() - ts_;
#endif
        ++sweeps;
        if (!s.overflow || thresh <= -c.tol) break;
        SYNC();
        const int wanted = s.wanted_raw - before - before_cold;  // rows that asked for a slot (counted past the capacity)
        SYNC();
        if (IS_T0) s.ncand = before, s.ncold = before_cold, s.overflow = 0;
        // far more rows than slots (a start point in the middle of a gridlock: thousands of rows): a quarter of the radius
        // would overflow again, and every retry is a full sweep — stage the violated rows only
        thresh = (thresh > 0.02 && wanted <= 4 * CMAX) ? 0.25 * thresh : -c.tol;
      }
      if (thresh > 0 && !s.overflow) {  // a staging sweep went through: remember where, and with what radius
        SYNC();
        PAR_FOR(k, 3 * (N + 1)) s.sw_ref[k / 3][k % 3] = s.st[k / 3][k % 3];
        if (IS_T0) s.sw_tau = thresh;
        SYNC();
      }
    };
    if (run && item < 0 && warm_cert && c.presweep != 0 && a.l1_rows == nullptr) {
      if (HDSM_TX < 64) W::states(s, R, tid_here(), N);
      SYNC();
      sweep_all(s.ncand, s.ncold);
      if (s.fixed_bad) run = false;  // still gridlocked: infeasible whatever the choice
    }
    bool swept_conc = false;
    // (n <= 30 only: a second copy of the sweep in the kernels of the larger factor costs them 170 bytes of spilled registers per lane)
    const bool overlap_ok = NV == 32 && c.overlap_sweep != 0 && blockDim.x == 128 && a.l1_rows == nullptr && !warm_cert && c.presweep != 0 && (uni(s.warm_head) & ~WARM_CERT) > 0;
    if (run && item < 0 && a.warm != nullptr) {
#ifdef HDSM_PROFILE
      const long long tw_ = clock64();
#endif
      TL_T0
      // Two-wave workgroups (the four-per-CU kernel of the bench line): the FIRST staging sweep runs on wave 1 WHILE wave 0 installs the
      // guess — the two used to follow each other, 13 + 7 us of the slowest instances, with one of the two waves idle in either. The sweep
      // cannot wait for the warm-start point, so it stages around the points every replan is expected to end near: the own previous
      // plan, one step on (the positions the planes themselves are built from); those points become the reference of the displacement
      // test, which decides at the end whether a verification sweep is needed, exactly as it does for a sweep at the warm-start point.
      const bool conc = overlap_ok;
      if (conc) {
        PAR_FOR(k, 3 * (N + 1)) {
          const int m = k / 3, ax = k % 3;
          s.sw_ref[m][ax] = m == 0 ? s.st[0][ax] : (m <= c.pinned_steps ? s.fr[ax][m][0] : s.cprev[m - 1][ax]);
        }
        if (IS_T0) s.nviol = 0;
        SYNC();
      }
      if (HDSM_TX < 64) {
        W::warm_start(s, c, a, R, inst, self, iters, wpre, conc);
        if (HDSM_TX == 0) s.iters_sh = iters;
      } else if (conc) {
        __syncthreads();  // (the guess's rows have their slots)
        swept_conc = c.presweep == 1 || a.bounds == nullptr || s.ncand > 0;  // (the rule of the sequential pre-sweep below)
        if constexpr (NV == 32) {
          if (swept_conc) WaveGI<NV, CMAX, SMALL>::template sweep_planes<true>(s, c, a, self, c.cand_tau, true, (int)HDSM_TX & 63, &s.sw_ref[0][0], 3);
        }
      }
      SYNC();
      iters = s.iters_sh;
      if (conc) {
        swept_conc = c.presweep == 1 || a.bounds == nullptr || s.warm_ncand > 0;  // (what wave 1 decided on: the count it saw)
        if (swept_conc) {
          if (s.overflow) {  // the staging area overflowed: as if no sweep had been made — the sequential path below shrinks the radius
            SYNC();
            if (IS_T0) s.ncand = s.warm_ncand, s.ncold = 0, s.overflow = 0, s.fixed_bad = 0;
            SYNC();
          } else {
            ++sweeps;
            if (IS_T0) s.sw_tau = c.cand_tau;
            if (s.fixed_bad) run = false;  // a common row is violated at a pinned point: infeasible whatever the choice
            SYNC();
          }
        }
      }
      TL_ADD(tl_warm_)
#ifdef HDSM_TIMELINE
      tl_warm_it_ = iters;
#endif
#ifdef HDSM_PROFILE
      t_warm_ = clock64() - tw_;
      it_warm_ = iters;
#endif
    }
    if (run && item < 0 && sweeps == 0 && (c.presweep == 1 || (c.presweep == 2 && (a.bounds == nullptr || s.ncand > 0)))) {
      // stage around the starting point (x_eq, or the warm-start point) before iterating. Automatic mode: always for
      // small swarms; for large (prefiltered) ones only when the warm start already holds neighbour rows, i.e. in a
      // dense neighbourhood (early in a flight the one sweep after the run is cheaper)
      if (HDSM_TX < 64) W::states(s, R, tid_here(), N);
      SYNC();
      sweep_all(s.ncand, s.ncold);
      if (s.fixed_bad) run = false;  // a common row is violated at the pinned point: infeasible whatever the choice
    }
    int last_rc = GI_OK;
    (void)last_rc;  // (read by the warm-start hand-over of the device build only)
    while (run) {
      int rc;
      {
        TL_T0
        rc = gi_run(s, c, R, cutoff(s, c), iters);
        TL_ADD(tl_run_)
#ifdef HDSM_TIMELINE
        ++tl_runs_;
#endif
      }
      last_rc = rc;
      NODE_PROF(16)
      if (rc == GI_ITERLIM || rc == GI_TIMELIM) {
        limit = true;
        flags |= rc == GI_ITERLIM ? FLAG_ITER_LIMIT : FLAG_TIME_LIMIT;
        break;
      }
      if (rc == GI_OK) {
#ifdef HDSM_PROFILE
        const long long tl_ = clock64();
#endif
        int bstep;
        {
          TL_T0
          bstep = leaf_check(s, c);
          TL_ADD(tl_leaf_)
        }
#ifdef HDSM_PROFILE
        t_leaf_ += clock64() - tl_;
#endif
        NODE_PROF(17)
        if (bstep == -2) continue;  // steps were assigned by the leaf test (one polyhedron left): the run goes on with their rows, same node
        if (bstep < 0) {
          // every step lies in a polyhedron: before accepting, re-check ALL neighbour rows
          const int before = s.ncand;
          const int before_cold = s.ncold;
          sweep_all(before, before_cold);
          if (s.fixed_bad) break;  // a common row is violated at the pinned point: infeasible whatever j
          // rows were staged AND at least one of them is violated: the dual method continues on this node.
          // (Rows staged merely because they are close do not move the iterate: the point is verified.)
          if (s.ncand > before && s.nviol) continue;
          if (s.overflow) {  // staging capacity exhausted, a violated row could not be staged
            limit = true;
            flags |= FLAG_STAGING_OVERFLOW;
            break;
          }
          PAR_FOR(k, n) s.inc_x[k] = s.x[k];
          PAR_FOR(i, N) s.inc_assign[i] = s.contain[i];
          PAR_FOR(k, NV) {
            int code = 0;
            if (k < s.q) {
              code = s.act[k];
              if (id_kind(code) == K_C) {
                const int src = s.cand_src[kc_slot(id_payload(code))];
                code = src >= 0 ? mk_id(K_C, src) : mk_id(K_E, 0);  // explicit rows carry no portable identity
              }
            }
            s.inc_act[k] = code;
          }
          if (IS_T0) s.inc_nact = s.q;
          if (IS_T0) {
            s.inc_f = s.f, s.have_inc = 1;
            // split launches: the other sub-blocks of this instance prune against it (objectives are >= 0: the bit patterns order)
            if (a.item_mode && s.f >= 0.0) atomicMin(&a.inc_bits[inst], (unsigned long long)__double_as_longlong(s.f));
          }
          SYNC();
        } else if (s.leaf_lb != 0 && s.f + s.lb_top1 >= cutoff(s, c)) {
          // the node bound reaches the incumbent: no leaf below this node can beat it — closed without a snapshot, without a level
        } else {  // open a new level on the first step that lies in no polyhedron
          const int L = s.level, bstep_own = bstep;
          NODE_PROF(18)
          snapshot_io(s, c, R, s.snap + (int64_t)L * SNAP_STRIDE, true);
          NODE_PROF(19)
          if (IS_T0) {
            const int bstep = bstep_own;
            int cnt = 0;
            for (int j = 0; j < np; ++j)
              if (s.keys[bstep][j] < DINF) s.br_order[L][cnt++] = j;
            for (int x1 = 1; x1 < cnt; ++x1)  // insertion sort, ascending violation, stable in j
              for (int y = x1; y > 0 && s.keys[bstep][s.br_order[L][y]] < s.keys[bstep][s.br_order[L][y - 1]]; --y) {
                const int t = s.br_order[L][y];
                s.br_order[L][y] = s.br_order[L][y - 1];
                s.br_order[L][y - 1] = t;
              }
            const double others = s.leaf_lb == 0 ? 0.0 : (bstep == s.lb_step ? s.lb_top2 : s.lb_top1);  // the node bound of the steps that stay uncontained
            for (int x1 = 0; x1 < cnt; ++x1) {
              const int item = bstep * np + s.br_order[L][x1];
              const double own = s.leaf_lb ? s.lb_at(item) : 0.0;
              s.br_lb[L][x1] = s.f + (own > others ? own : others);
              s.br_pk[L][x1] = s.leaf_lb ? s.pk_at(item) : -1, s.br_pv[L][x1] = s.leaf_lb ? s.pv_at(item) : 0.0;
            }
            s.br_cnt[L] = cnt, s.br_pos[L] = 0, s.br_step[L] = bstep, s.br_f[L] = s.f;
            s.level = L + 1;
          }
          SYNC();
        }
      }
      NODE_PROF(20)
      if (rc == GI_INFEASIBLE && c.P <= 4 && N <= 16) {
        // Conflict learning. The dual method stopped because row inf_id depends on the working set and no multiplier can
        // give way: working set + that row are infeasible TOGETHER. Apart from the rows of assigned polyhedra, everything
        // in there (terminal equalities, boxes, neighbour planes) holds at every node, so the assignments those rows come
        // from — usually one or two, decided high in the tree — are a conflict wherever they appear again. No assignment
        // involved at all: the instance is infeasible whatever the choice, the search ends.
        SYNC();
        if (IS_T0) {
          unsigned long long m = 0;
          for (int k = 0; k <= s.q && k < NV + 1; ++k) {
            const int code = (k < s.q) ? s.act[k] : s.inf_id;
            if (id_kind(code) == K_P) {
              const int i = id_payload(code) >> 7;
              if (s.assign[i] >= 0) m |= 1ull << (16 * s.assign[i] + i);  // (bit 16 j + i: polyhedron j at step i)
            }
          }
          if (m == 0ull) {
            if (!s.have_inc) s.level = 0, s.ng_global = 1;  // (with an incumbent in hand the proof can only be a numerical artefact: ignored)
          } else {
            bool known = false;
            for (int k = 0; k < s.n_nogood; ++k) known = known || (s.nogood[k] & ~m) == 0ull;
            if (!known && s.n_nogood < NOGOODS) s.nogood[s.n_nogood++] = m;
          }
        }
        SYNC();
      }
      // node closed (incumbent recorded / infeasible / cut off) or level opened: go to the next child
      const bool lim_before = limit;
      NODE_PROF(21)
      run = select_child(s, c, R, nodes, limit, inst, handed_over, rec_slot);
      NODE_PROF(22)
      if (limit && !lim_before) flags |= FLAG_NODE_LIMIT;
    }

#ifdef HDSM_TIMELINE
    const long long tl_loop_end_ = (long long)wall_clock64();
#endif
#ifdef HDSM_PROFILE
    if (IS_T0 && a.prof) {
      long long* pr = a.prof + (int64_t)inst * 32;
      for (int k = 0; k < 8; ++k) pr[k] = s.prof_acc[k], pr[16 + k] = s.prof_acc[8 + k], pr[24 + k] = s.prof_acc[16 + k];  // 16..23: inside the sweeps, 24..31: inside the warm start
      pr[8] = t_setup_, pr[9] = t_sweep_, pr[10] = t_leaf_, pr[11] = clock64() - t_begin_;
      pr[12] = iters, pr[13] = sweeps, pr[14] = it_warm_, pr[15] = t_warm_;
    }
#endif
    // ---- read-back (AC:955-987): controls, literal rollout of the dynamics, literal objective
    // (an instance handed over to pass 2 of a split launch leaves without outputs: the merge kernel writes them)
    // (... with one exception: an incumbent pass 1 has already found stays where the outputs go — bit 1 of split_info — so that
    // pass 2 prunes against it from its first node and the merge can fall back on it)
    if (handed_over) {
      // The record of the hand-over (Args, SplitRec): the open levels with their orders, bounds and first picks, the path's
      // assignments, the staged rows (hot ones first, then the cold ones), conflicts, the sweeps' displacement reference — and one
      // ITEM per child of an open level that is still to be explored and whose bound does not reach the incumbent (shallow levels,
      // the large subtrees, first). The snapshots of the open levels are in this instance's scratch already.
      SplitRec& rc = a.recs[rec_slot];
      const int lev = s.level, nh = s.ncand, ncl = s.ncold;
      const int64_t rbase = (int64_t)rec_slot * a.rows_cap;
      SYNC();
      PAR_FOR(k, nh + ncl) {
        const int slot = k < nh ? k : CMAX - ncl + (k - nh);
        double* dst = a.rec_cand + (rbase + k) * 4;
        dst[0] = s.cand[slot][0], dst[1] = s.cand[slot][1], dst[2] = s.cand[slot][2], dst[3] = s.cand[slot][3];
        a.rec_mw[rbase + k] = *reinterpret_cast<const long long*>(&s.cand_mw[slot]);
        a.rec_src[rbase + k] = s.cand_src[slot];
      }
      PAR_FOR(k, 3 * (N + 1)) rc.sw_ref[k / 3][k % 3] = s.sw_ref[k / 3][k % 3];
      PAR_FOR(k, s.n_nogood) rc.nogood[k] = s.nogood[k];
      PAR_FOR(k, MAXH) rc.assign[k] = s.assign[k];
      PAR_FOR(k, lev * S::PM) {
        const int l = k / S::PM, x1 = k % S::PM;
        if (x1 < MAXP) rc.br_order[l][x1] = s.br_order[l][x1], rc.br_pk[l][x1] = s.br_pk[l][x1], rc.br_lb[l][x1] = s.br_lb[l][x1], rc.br_pv[l][x1] = s.br_pv[l][x1];
      }
      PAR_FOR(l, lev) rc.br_step[l] = s.br_step[l], rc.br_cnt[l] = s.br_cnt[l], rc.br_pos[l] = s.br_pos[l], rc.br_f[l] = s.br_f[l];
      __threadfence();  // (pass 2 draws items while it runs: the record must be visible before its items are)
      SYNC();
      if (IS_T0) {
        a.rec_count[8 + rec_slot] = inst;  // (the merge's lanes look for an instance's records HERE: consecutive words, not one per 1.6-KB record)
        rc.inst = inst, rc.level = lev, rc.ncand = nh, rc.ncold = ncl, rc.n_nogood = s.n_nogood, rc.sw_tau = s.sw_tau;
        rc.nodes_done = nodes, rc.sweeps_done = sweeps, rc.snap = s.snap, rc.sp_dom = s.sp_dom;
        // (an item that hands over again: its record goes in front of the instance's chain, and the workgroup moves to a fresh scratch slot)
        rc.next = item >= 0 ? atomicExch(&a.split_info[2 * inst + 1], rec_slot) : -1;
        const double cut = cutoff(s, c);
        auto open_child = [&](int l, int x1, unsigned long long path) {  // still to be explored: bound below the incumbent, no learnt conflict
          if (s.br_lb[l][x1] >= cut) return false;
          const unsigned long long cur = path | (1ull << (16 * s.br_order[l][x1] + s.br_step[l]));
          bool blocked = false;
          if (c.P <= 4 && N <= 16)
            for (int k = 0; k < s.n_nogood; ++k) blocked = blocked || (s.nogood[k] & ~cur) == 0ull;
          return !blocked;
        };
        int cnt = 0;
        unsigned long long path = 0ull;
        for (int l = 0; l < lev; ++l) {
          for (int x1 = s.br_pos[l]; x1 < s.br_cnt[l]; ++x1) cnt += open_child(l, x1, path) ? 1 : 0;
          if (l + 1 < lev) path |= 1ull << (16 * s.assign[s.br_step[l]] + s.br_step[l]);
        }
        // the items' places in the queue are RESERVED first ([5]: the reserved end), written, and then PUBLISHED in order ([1]: the
        // end of what the workgroups of pass 2 may draw) — an item is never drawn before it is there
        const int first = atomicAdd(&a.rec_count[5], cnt);
        const int room = a.items_cap - first < 0 ? 0 : a.items_cap - first;
        rc.first_item = first, rc.n_items = cnt < room ? cnt : room, rc.truncated = cnt > room ? 1 : 0;  // (a queue that is full: the merge reports a limit)
        int w = 0;
        path = 0ull;
        for (int l = 0; l < lev; ++l) {
          for (int x1 = s.br_pos[l]; x1 < s.br_cnt[l]; ++x1)
            if (open_child(l, x1, path) && w < rc.n_items) {
              a.item_status[first + w] = ST_PENDING;  // (an item no workgroup gets to — a grid or a queue too short — is reported as a limit)
              a.items[first + w++] = (rec_slot << 8) | (l << 3) | x1;
            }
          if (l + 1 < lev) path |= 1ull << (16 * s.assign[s.br_step[l]] + s.br_step[l]);
        }
        // what pass 2 starts from: the incumbent found so far (objectives are >= 0: the bit patterns order) and the rest of the node budget
        if (item < 0) {
          a.inc_bits[inst] = s.have_inc && s.inc_f >= 0.0 ? (unsigned long long)__double_as_longlong(s.inc_f) : 0x7ff0000000000000ull;
          a.node_pool[inst] = a.nodes_pool0;
        }
        __threadfence();
        // (those before: a few stores away. Bounded all the same — a predecessor that never publishes, or a launch a waiter has
        // aborted (rec_count[6], hdsm_api.hip run_block), must not hang this workgroup: it raises the abort word and leaves its items
        // unpublished; they stay ST_PENDING and the merge reports the instance as LIMIT)
        int spins = 0;
        bool in_order = true;
        while (__hip_atomic_load(&a.rec_count[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != first) {
          if (++spins > (1 << 22) || __hip_atomic_load(&a.rec_count[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
            in_order = false;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        if (in_order) __hip_atomic_store(&a.rec_count[1], first + cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else atomicExch(&a.rec_count[6], 1);
      }
      if (item >= 0) wg_slot = -1;  // (pass 2: this workgroup's scratch, with the snapshots of the open levels, stays with the record)
    }
    if (IS_T0 && a_in.split_budget > 0 && item < 0)  // (an item of pass 2 that hands over again only chains its record in, above)
      a.split_info[2 * inst] = handed_over ? (1 | (s.have_inc ? 2 : 0)) : 0, a.split_info[2 * inst + 1] = handed_over ? rec_slot : -1;
    if (IS_T0 && a.tree_flag != nullptr && a.tree_mark > 0 && nodes >= a.tree_mark) *a.tree_flag = 1;  // (split launches: the merge raises it)
    const int status = s.have_inc ? (limit ? ST_LIMIT : ST_OPTIMAL) : ST_NO_SOLUTION;
    if (s.have_inc) {
      double* tr = a.traj + (int64_t)out * 9 * (N + 1);
      double* cu = a.ctrl + (int64_t)out * 3 * N;
      PAR_FOR(k, 3 * N) cu[k] = s.inc_x[(k % 3) * N + k / 3];
      PAR_FOR(ax, 3) {
        double xs[3] = {s.state0[ax], s.state0[3 + ax], s.state0[6 + ax]};
        for (int comp = 0; comp < 3; ++comp) s.st[0][3 * comp + ax] = xs[comp];
        for (int i = 0; i < N; ++i) {
          const double u = s.inc_x[ax * N + i];
          double xn[3];
          for (int r = 0; r < 3; ++r)
            xn[r] = c.Ad[ax][r][0] * xs[0] + c.Ad[ax][r][1] * xs[1] + c.Ad[ax][r][2] * xs[2] + c.Bd[ax][r] * u;
          for (int r = 0; r < 3; ++r) {
            xs[r] = xn[r];
            s.st[i + 1][3 * r + ax] = xn[r];
          }
        }
      }
      SYNC();
#ifdef HDSM_TIMELINE
      tl_tail1_ = (long long)wall_clock64();  // rollout done
#endif
      PAR_FOR(k, 9 * (N + 1)) tr[k] = s.st[k / 9][k % 9];
      if (HDSM_TX < 64) {  // literal objective (AC:870-883, AC:2098), one term per lane, summed across the wave
        const int lane = tid_here();
        double part = (lane < n) ? c.r_u * s.inc_x[lane] * s.inc_x[lane] : 0.0;
        for (int idx = lane; idx < 6 * N; idx += 64) {
          const int i = idx / 6 + 1, k = idx % 6;
          const double e = s.st[i][k] - s.ref[i - 1][k];
          part += ((i == N) ? c.wn[k] : c.wx[k]) * e * e;
        }
        part = wave_sum64(part);
        if (lane == 0) a.obj[out] = part;
      }
      if (IS_T0) {
        uint8_t* us = a.used + (int64_t)out * P;
        for (int j = 0; j < P; ++j) us[j] = 0;
        for (int i = 0; i < N; ++i)
          if (s.inc_assign[i] >= 0 && s.inc_assign[i] < P) us[s.inc_assign[i]] = 1;
      }
    }
#ifdef HDSM_TIMELINE
    tl_tail2_ = (long long)wall_clock64();  // outputs written
#endif
    if (a.warm != nullptr && (!handed_over || s.have_inc)) {  // next replan's guess (handed over: the merge may replace it)
      int32_t* wp = a.warm_out + (int64_t)out * (MAXNV + 2);
      // No solution because the ROOT relaxation is infeasible (the usual case in a gridlocked neighbourhood, and it
      // tends to persist for several rounds): hand over the certificate — the working set at the moment of the proof
      // and the row that could not join it. Seeded with it, the next replan finds the contradiction (or its absence)
      // after a few operations instead of rebuilding it from the unconstrained optimum.
      const bool certificate = !s.have_inc && !limit && (item < 0 ? (nodes == 1 || s.ng_global) : s.ng_global != 0) && last_rc == GI_INFEASIBLE && s.q < NV;
      if (certificate) {
        SYNC();
        PAR_FOR(k, NV) {
          const int code = (k < s.q) ? s.act[k] : (k == s.q ? s.inf_id : 0);
          int portable = code;
          if (k <= s.q && id_kind(code) == K_C) {
            const int src = s.cand_src[kc_slot(id_payload(code))];
            portable = src >= 0 ? mk_id(K_C, src) : mk_id(K_E, 0);
          }
          s.inc_act[k] = portable;
        }
        SYNC();
      }
      const int cnt = s.have_inc ? s.inc_nact : (certificate ? s.q + 1 : 0);
      PAR_FOR(k, NV) if (k < cnt) wp[1 + k] = s.inc_act[k];
      const bool gridlock = s.fixed_bad && !s.have_inc;  // ended on the pinned-position test of a sweep
      if (IS_T0) wp[0] = cnt | ((certificate || gridlock) ? WARM_CERT : 0);
    }
#ifdef HDSM_TIMELINE
    if (HDSM_TX == 64 && a.prof) {  // ... and where its second wavefront (the scanner) sat (scripts/timeline_simd.py)
      unsigned hw1;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw1));
      a.prof[(int64_t)inst * 32 + 30] = (long long)hw1;
    }
    if (IS_T0 && a.prof) {  // development aid (scripts/gpu_timeline.sh): when and where this instance ran
      long long* pr = a.prof + (int64_t)inst * 32;
      unsigned hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      pr[0] = tl_begin_, pr[1] = (long long)wall_clock64(), pr[2] = (long long)blockIdx.x, pr[3] = (long long)hw | ((long long)(xcc & 15u) << 32), pr[4] = iters;
      pr[5] = nodes, pr[6] = sweeps, pr[7] = s.ncand, pr[8] = (long long)flags, pr[9] = s.ncold, pr[10] = status;
      pr[11] = s.st_pairs, pr[12] = s.st_sph, pr[13] = s.q, pr[14] = s.n_nogood, pr[15] = s.ng_skipped;
      pr[16] = tl_setup_ - tl_begin_, pr[17] = tl_warm_, pr[18] = tl_warm_it_, pr[19] = tl_sweep_, pr[20] = tl_run_, pr[21] = tl_runs_, pr[22] = tl_leaf_, pr[23] = (long long)wall_clock64() - tl_loop_end_;
      pr[24] = tl_su1_ - tl_begin_, pr[25] = tl_su2_ - tl_su1_, pr[26] = tl_setup_ - tl_su2_;
      pr[27] = tl_tail1_ ? tl_tail1_ - tl_loop_end_ : 0, pr[28] = tl_tail1_ ? tl_tail2_ - tl_tail1_ : 0, pr[29] = (long long)wall_clock64() - tl_tail2_;
    }
#endif
    if (IS_T0) {
      a.status[out] = status;
      if (a.st_iters) a.st_iters[out] = iters;
      if (a.st_nodes) a.st_nodes[out] = nodes;
      // (the sweeps pass 1 had made come from the record again: carried in a register from the hand-over to here they were one more value to spill)
      if (a.st_sweeps) a.st_sweeps[out] = sweeps - (item >= 0 ? a.recs[item >> 8].sweeps_done : 0);
      if (a.st_cand) a.st_cand[out] = s.ncand + s.ncold;
      if (a.st_sph) a.st_sph[out] = s.st_sph;
      if (a.st_pairs) a.st_pairs[out] = s.st_pairs;
      if (a.st_flags) a.st_flags[out] = flags;
      if (a.st_key) {
        // what the next launch sorts by (hdsm_api.hip, launch_order_block): how long this instance took; an instance without
        // a solution goes first whatever it took — its next replan either ends on the certificate at once or is among the
        // longest of the launch, and starting a short one early costs nothing
        // (+ 9 units per row of the final working set: next to the duration, the size of the active set is what predicts the
        // next replan's length best on the crossing rounds — list-scheduling replay of recorded launches, profiles/README.md)
        // (n > 30: instances last hundreds of microseconds — in 0.64-us units every one of them beyond 163 us carried the SAME key, 254, and
        // the long ones of a cfg 5 launch started in arbitrary order: 19 of 20 launches were set by a late starter, span 1.37 ms against
        // 0.90 ms for the slowest instance. 2.56-us units there: 650 us before the key saturates, and still 30 levels for the 85-us instances of an open-space H = 15 round)
#ifdef HDSM_OLD_KEY
        constexpr int KEY_SHIFT = 6;
#else
        constexpr int KEY_SHIFT = NV > 32 ? 8 : 6;
#endif
        const long long ticks = (((long long)wall_clock64() - tl_begin_) >> KEY_SHIFT) + ((9 * s.q) >> (KEY_SHIFT - 6));
        a.st_key[out] = status == ST_NO_SOLUTION ? 255 : (int)(ticks < 0 ? 0 : (ticks > 254 ? 254 : ticks));
      }
      if (a.ovf_flag != nullptr && (flags & FLAG_STAGING_OVERFLOW)) *a.ovf_flag = 1;
      if (a.item_mode && s.node_res > 0) atomicAdd(&a.node_pool[inst], s.node_res);  // the unused part of the share: to the instance's pool
    }
    SYNC();
  }

  static HD int min_i(int x, int y) { return x < y ? x : y; }
  static HD int max_i(int x, int y) { return x > y ? x : y; }
};

// Split launches, last step (k_split_merge: one wavefront per instance; `lane` of `lanes`): the best answer of the items of an
// instance that pass 1 handed over becomes the instance's answer. `a` holds the instance-indexed arrays of the launch, `b` the
// arrays of pass 2 (indexed by the item's place in the queue; the record of the instance names its range).
HD void split_merge(int N, int P, const Args& a, const Args& b, int inst, int lane, int lanes) {
  if (inst >= a.n_inst || a.split_info[2 * inst] == 0) return;
  // every lane takes the items lane, lane + lanes, ... of each record (the reads are dependent global round trips: one lane walking
  // all of them took 30 - 200 us per launch); the lanes' partial results meet through shuffles (device) — one lane: nothing to meet
  int best = -1, lim = 0, iters = 0, nodes = 0, sweeps = 0, cand = 0, sph = 0, pairs = 0;
  unsigned flags = 0;
  const bool own = (a.split_info[2 * inst] & 2) != 0;  // pass 1 left an incumbent in the instance's own outputs
  double obj = DINF;
  // The records of this instance — the one of pass 1 and those of the items that handed over again — are a chain through
  // SplitRec::next: walking it is one dependent global round trip per record (0.1 - 0.2 ms per launch in the pillar forest, where an
  // instance leaves dozens of records). On the device the lanes look at ALL records of the launch instead (their `inst` words: independent
  // loads), list the matching ones in LDS, and only then go through them; the chain is walked when the list does not fit.
  constexpr int LIST = 256;
  int n_list = -1;  // -1: walk the chain
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ int32_t rec_list[LIST];
  if (lanes == 64) {
    const int taken = a.rec_count[0], R = taken < a.rec_cap ? taken : a.rec_cap;
    n_list = 0;
    for (int base = 0; base < R && n_list <= LIST; base += 64) {
      const int r = base + lane;
      const bool mine = r < R && a.rec_count[8 + r] == inst;
      const unsigned long long m = __ballot(mine);
      if (mine) {
        const int at = n_list + __popcll(m & ((1ull << lane) - 1ull));
        if (at < LIST) rec_list[at] = r;
      }
      n_list += __popcll(m);
    }
    __syncthreads();
    if (n_list > LIST) n_list = -1;
  }
#endif
  for (int q = 0, r = n_list < 0 ? a.split_info[2 * inst + 1] : -1; n_list < 0 ? r >= 0 : q < n_list; ++q) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (n_list >= 0) r = rec_list[q];
#endif
    const SplitRec& rc = a.recs[r];
    const int r_next = n_list < 0 ? rc.next : -1;
    if (rc.truncated) lim = 1, flags |= (unsigned)FLAG_NODE_LIMIT;
    for (int g = rc.first_item + lane; g < rc.first_item + rc.n_items; g += lanes) {
      const int st = b.status[g];
      if (st == ST_PENDING) {  // never started
        lim = 1, flags |= (unsigned)FLAG_NODE_LIMIT;
        continue;
      }
      iters += b.st_iters[g], nodes += b.st_nodes[g], sweeps += b.st_sweeps[g];
      if (b.st_sph) sph += b.st_sph[g], pairs += b.st_pairs[g];
      cand = b.st_cand[g] > cand ? b.st_cand[g] : cand;
      flags |= b.st_flags[g];
      lim |= st == ST_LIMIT || (b.st_flags[g] & (FLAG_NODE_LIMIT | FLAG_ITER_LIMIT | FLAG_TIME_LIMIT | FLAG_STAGING_OVERFLOW)) != 0;
      if (st != ST_NO_SOLUTION && (b.obj[g] < obj || (b.obj[g] == obj && best >= 0 && g < best))) obj = b.obj[g], best = g;
    }
    r = r_next;
  }
#if defined(__HIP_DEVICE_COMPILE__)
  if (lanes > 1) {
    for (int off = 32; off > 0; off >>= 1) {
      iters += __shfl_xor(iters, off), nodes += __shfl_xor(nodes, off), sweeps += __shfl_xor(sweeps, off), sph += __shfl_xor(sph, off), pairs += __shfl_xor(pairs, off);
      const int c2 = __shfl_xor(cand, off);
      cand = c2 > cand ? c2 : cand;
      flags |= (unsigned)__shfl_xor((int)flags, off), lim |= __shfl_xor(lim, off);
      const double o2 = __shfl_xor(obj, off);
      const int b2 = __shfl_xor(best, off);
      if (b2 >= 0 && (o2 < obj || (o2 == obj && (best < 0 || b2 < best)))) obj = o2, best = b2;  // (ties: the first item of the queue, in every lane)
    }
  }
#endif
  iters += a.st_iters[inst], nodes += a.st_nodes[inst], sweeps += a.st_sweeps[inst];
  cand = a.st_cand[inst] > cand ? a.st_cand[inst] : cand;
  if (a.st_sph) sph += a.st_sph[inst], pairs += a.st_pairs[inst];
  if (own && !(obj < a.obj[inst])) obj = a.obj[inst], best = -1;  // (pass 1's own incumbent is at least as good)
  if (!own && best < 0) obj = DINF;
  const int status = (best < 0 && !own) ? ST_NO_SOLUTION : (lim ? ST_LIMIT : ST_OPTIMAL);
  if (best >= 0) {
    for (int e = lane; e < (N + 1) * 9; e += lanes) a.traj[(int64_t)inst * (N + 1) * 9 + e] = b.traj[(int64_t)best * (N + 1) * 9 + e];
    for (int e = lane; e < N * 3; e += lanes) a.ctrl[(int64_t)inst * N * 3 + e] = b.ctrl[(int64_t)best * N * 3 + e];
    for (int e = lane; e < P; e += lanes) a.used[(int64_t)inst * P + e] = b.used[(int64_t)best * P + e];
  }
  if (a.warm != nullptr && !(own && best < 0)) {  // next replan's guess: the best item's working set (none: start cold; pass 1's own: in place)
    int32_t* wp = a.warm + (int64_t)inst * (MAXNV + 2);
    const int32_t* src = b.warm_out + (int64_t)(best >= 0 ? best : 0) * (MAXNV + 2);
    for (int e = lane; e < MAXNV + 2; e += lanes) wp[e] = best >= 0 ? src[e] : 0;
  }
  if (lane == 0) {
    if (a.tree_flag != nullptr && nodes >= TREE_MARK) *a.tree_flag = 1;  // still a deep tree: the next launches stay in the split form
    if (best >= 0) a.obj[inst] = obj;
    a.status[inst] = status;
    a.st_iters[inst] = iters, a.st_nodes[inst] = nodes, a.st_sweeps[inst] = sweeps, a.st_cand[inst] = cand;
    if (a.st_sph) a.st_sph[inst] = sph, a.st_pairs[inst] = pairs;
    a.st_flags[inst] = flags;
#ifdef HDSM_OLD_KEY
    if (a.st_key) a.st_key[inst] = status == ST_NO_SOLUTION ? 255 : 254;  // a deep tree: launch it first next time
#else
    // a deep tree: early next time — among the handed-over instances in the order of what pass 1 spent on them (its own key stays: what
    // the next pass 1 will spend is what the launch order has to predict; 254 for all of them left their order to chance)
    if (a.st_key) a.st_key[inst] = status == ST_NO_SOLUTION ? 255 : (a.st_key[inst] > 128 ? a.st_key[inst] : 128);
#endif
  }
}

}  // namespace hdsm
