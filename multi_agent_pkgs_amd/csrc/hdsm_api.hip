// hdsm_api.hip — gfx950 kernels and the extern "C" boundary declared in include/hdsm.h.
//
// One workgroup solves one agent-replan (hdsm_core.h); the launch is a plain 1-D grid of n_inst blocks.
// There is no CPU path in this library: without a HIP device hdsm_create() fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <mutex>

#include <cfloat>
#include <rccl/rccl.h>

#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/hdsm.h"
#include "hdsm_consts.h"
#include "hdsm_core.h"
#include "hdsm_level1.h"

namespace {

thread_local std::string g_err;

int set_err(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) {                                                                             \
      (void)hipGetLastError(); /* reported here: the next call must not trip over it again */            \
      return set_err(HDSM_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));               \
    }                                                                                                   \
  } while (0)

// staged neighbour rows (LDS). One workgroup per CU is resident anyway (the iteration wave needs > 256 registers,
// a 256-register budget spills 644 B/lane), so LDS capacity is spent on fewer staging-radius retries.
constexpr int CMAX30 = 1536;  // n <= 30
constexpr int CMAX48 = 1024;  // n <= 48

// What a workgroup works on. Ordinary launch: block b -> instance order[b] (or b). Pass 2 of a split launch (a.item_mode): the
// workgroups are PERSISTENT — each draws items from the queue pass 1 filled (Args::items: one open child of an open level of a
// handed-over instance) until the queue is empty; blockIdx only names the workgroup's snapshot scratch.
template <class Sol>
__device__ __forceinline__ void run_block(typename Sol::S* sp, const hdsm::Consts* cp, const hdsm::Args& a) {
  typename Sol::S& s = *sp;
  int inst, out, item = -1, self = -1, slot = 0;  // (ONE call site of the solver per kernel: it is a single inlined body of ~17 k instructions)
  if (a.item_mode) {
    // Pass 2 of a split launch: this workgroup takes ONE item from the queue (Args::items: an open child of an open level of a
    // handed-over instance) and a snapshot-scratch slot, and leaves. The queue can still GROW while items are running (an item
    // whose subtree turns out large hands over again), so a workgroup that finds it empty waits — on a CU that would be idle
    // anyway — until an item appears or no workgroup holds one any more (rec_count[4]). The grid is an upper bound of the items
    // of the launch (hdsm_api.hip, launch()); the workgroups beyond them find the queue empty and nothing running, and leave.
    // (the few launch arguments the loops below need, as plain values: `a` itself referenced inside a loop is kept as a private copy
    // of the whole struct — 256 bytes of scratch per lane in every kernel that shares this function)
    int32_t* const rcnt = a.rec_count;
    int32_t* const busy = a.slot_busy;
    const int icap = a.items_cap, pcap = a.pool_cap, psleep = a.poll_sleep;
    if (threadIdx.x == 0) {
      // ONE atomic per workgroup: a ticket (rcnt[2]). Ticket k is item k of the queue — when it has been published (rcnt[1] > k)
      // the workgroup takes it; until then it waits, READING only. It leaves without an item when every published item has been
      // completed (rcnt[4], raised by a workgroup after its item — and after whatever that item queued was published) and the
      // queue still ends at or before its ticket: nothing is running, so nothing can be queued any more. (Drawing with a
      // compare-and-swap on a shared counter was measured first: 512 workgroups retrying against each other cost a millisecond.)
      const int ticket = atomicAdd(&rcnt[2], 1);
      int got = -1;
      bool timed_out = ticket < icap;
      for (int spins = 0; spins < (1 << 19) && ticket < icap; ++spins) {  // (seconds: whatever holds the last items up)
        const int done = __hip_atomic_load(&rcnt[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int q = __hip_atomic_load(&rcnt[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        q = q < icap ? q : icap;
        if (ticket < q) {
          got = ticket, timed_out = false;
          break;
        }
        // (read BEFORE the queue's end: all of it was completed, and only a running item can extend it) — or the launch was
        // ABORTED (rcnt[6]): a waiter ran out of patience, so its ticket will never be drawn and `done` can never reach the end
        // of the queue again; every other waiter would spin out its own seconds one after the other
        if (done >= q || __hip_atomic_load(&rcnt[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
          timed_out = false;
          break;
        }
        // (hundreds of workgroups may be waiting, and every look is a device-scope read that no L2 can serve. Letting the few whose
        // tickets come next look without sleeping was measured in round 6: no difference on cfg 3 or cfg 5 — scripts/gpu_r6_poll_ab.sh)
        for (int w = 0; w < psleep; ++w) __builtin_amdgcn_s_sleep(127);
      }
      if (timed_out) atomicExch(&rcnt[6], 1);  // (items left in the queue stay ST_PENDING: the merge reports their instances as LIMIT)
      int sl = -1;
      if (got >= 0) {  // a scratch slot: at most gridDim-resident + records slots are ever busy (a slot left to a record stays busy)
        for (int probe = 0; probe < pcap && sl < 0; ++probe) {
          const int i = (int)(((unsigned)blockIdx.x * 7u + (unsigned)probe) % (unsigned)pcap);
          if (atomicCAS(&busy[i], 0, 1) == 0) sl = i;
        }
        if (sl < 0) atomicAdd(&rcnt[4], 1), got = -2;  // (cannot happen by the count above; the item stays pending: the merge reports a limit)
      }
      s.iters_sh = got, s.rc = sl;
    }
    __syncthreads();
    const int k = hdsm::uni(s.iters_sh);
    slot = hdsm::uni(s.rc);
    __syncthreads();
    if (k < 0) return;  // (uniform: the whole workgroup leaves)
    __threadfence();    // (the item, its record and the snapshots it names were written by another workgroup)
    item = __hip_atomic_load(&a.items[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), out = k, inst = a.recs[item >> 8].inst;
  } else {
    if (a.order) {  // (instance, its agent id): one load — the own plan is requested together with the other inputs of the instance
      const int2 os = reinterpret_cast<const int2*>(a.order)[blockIdx.x];
      inst = os.x, self = os.y;
    } else {
      inst = (int)blockIdx.x;
    }
    out = inst;
    if (a.rescue && !(a.st_flags[inst] & hdsm::FLAG_STAGING_OVERFLOW)) return;  // (uniform)
  }
  // (every solver kernel has the signature (const Consts*, Args): the arguments start at byte 8 of the kernel-argument segment)
#if defined(__HIP_DEVICE_COMPILE__)
  const HDSM_KERNARG_WORD* words = (const HDSM_KERNARG_WORD*)__builtin_amdgcn_kernarg_segment_ptr() + 1;
#else
  const HDSM_KERNARG_WORD* words = nullptr;
#endif
  Sol::solve_instance(s, *cp, a, inst, out, item, self, slot, words);
  if (item >= 0 && threadIdx.x == 0) {  // (the launch arguments from the copy in LDS: the kernel's own are dead after the solver's prologue)
    if (slot >= 0) atomicExch(&s.args.slot_busy[slot], 0);  // (slot < 0: the item handed over again and its scratch stays with its record)
    __threadfence();
    atomicAdd(&s.args.rec_count[4], 1);                     // this item is completed; what it queued has been published
  }
}

template <int NV, int CMAX, int NT>
__global__ __launch_bounds__(NT) void k_replan(const hdsm::Consts* __restrict__ cp, hdsm::Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using Sol = hdsm::Solver<NV, CMAX>;
  run_block<Sol>(reinterpret_cast<typename Sol::S*>(smem), cp, a);
}

// The same solver budgeted for TWO workgroups per CU (registers: 2 waves per SIMD; LDS: a staging area of CMAX_DUO
// rows makes the instance state fit twice into 160 KB). A single instance is bound by the latency of its one iterating
// wave, so when there are more instances than CUs a second resident workgroup nearly doubles the throughput. Used for
// n <= 30 only (the NV = 48 factor does not fit the halved register file).
constexpr int CMAX_DUO = 768;
template <int NV, int CMAX, int NT>
__global__ __launch_bounds__(NT, 2) void k_replan_duo(const hdsm::Consts* __restrict__ cp, hdsm::Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using Sol = hdsm::Solver<NV, CMAX>;
  run_block<Sol>(reinterpret_cast<typename Sol::S*>(smem), cp, a);
}

// Pre-pass of every level-2 launch, one thread per agent of the swarm:
//   pos[n_rob][N][3]   positions of steps 1..N of every published plan, packed: the sweeps of the replan kernel read 24 B
//                      per (neighbour, step) instead of striding through 72-B state records (zeros for agents without a plan);
//   bounds[n_rob][4]   (only for swarms of at least bounds_min agents) centre of the bounding box of those positions and the
//                      radius of the sphere around it that holds them (radius -1 = no plan): the sweeps use it to skip
//                      whole neighbours (hdsm_wave_gi.h, sweep_planes).
__device__ void launch_order_block(int n_inst, const int32_t* __restrict__ key_prev, const int32_t* __restrict__ agent_id, int32_t* __restrict__ order);

// The set-up map of ALL instances of a launch as ONE dense product on the matrix cores. Everything an instance needs before its first
// iteration — x_eq, x0, the gradient at u = 0, the residual of the terminal equalities and their multipliers — is linear in
// v = (state_curr, traj_ref) (Consts::KT, built once by hdsm_create from the Hessian factor): OUT[row][inst] = sum_j KT[j][row] v[inst][j],
// (3n + 12) x (9 + 6N) times (9 + 6N) x n_inst — the one contraction of the path with matrix-matrix shape and more than a handful of
// columns. One wavefront = one 16 x 16 tile of OUT through v_mfma_f64_16x16x4_f64 (A[i = lane & 15][k = lane >> 4] = KT[4 k0 + k][row0 + i],
// B[k = lane >> 4][j = lane & 15] = v[inst0 + j][4 k0 + k], D[row = (lane >> 4) + 4 r][col = lane & 15] in register r). The solver's
// set-up then reads one number per thread instead of 23 coefficients and a 23-term dot product (hdsm_core.h, Args::setup).
struct SetupMapArgs {
  const hdsm::Consts* c;
  const double* state;  // [n_inst][9]
  const double* ref;    // [n_inst][N][6]
  double* out;          // [n_inst][KROWS]
  int32_t n_inst, N, nk, row_tiles, first_block;  // first_block: the workgroup of the pre-pass kernel at which the tiles begin
};
__device__ void setup_map_tile(const SetupMapArgs& m, int tile) {
  using v4d = double __attribute__((ext_vector_type(4)));
  const int lane = (int)threadIdx.x & 63;
  const int rt = tile % m.row_tiles, it = tile / m.row_tiles;
  if (it * 16 >= m.n_inst) return;
  const int i = lane & 15, kk = lane >> 4;
  const int row = rt * 16 + i, inst = it * 16 + i, nvt = 9 + 6 * m.N;
  const bool row_on = row < m.nk, inst_on = inst < m.n_inst;
  const double* st = m.state + (int64_t)(inst_on ? inst : 0) * 9;
  const double* rf = m.ref + (int64_t)(inst_on ? inst : 0) * 6 * m.N;
  v4d acc = {0.0, 0.0, 0.0, 0.0};
  // (the operands of NINE k-steps are requested before the first product is formed: with one step per trip the tile was a chain of 18 - 25
  // dependent round trips to memory, and the tile workgroups — not the packing ones — set the duration of the pre-pass kernel: 6.9 -> 11 us
  // in round 5; two or three trips now)
  constexpr int CH = 9;
  for (int k0 = 0; k0 < nvt; k0 += 4 * CH) {
    double av[CH], bv[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int j = k0 + 4 * u + kk;
      av[u] = (row_on && j < nvt) ? m.c->KT[(int64_t)j * hdsm::KROWS + row] : 0.0;
      bv[u] = (inst_on && j < nvt) ? (j < 9 ? st[j] : rf[j - 9]) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < CH; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
  }
  const int col = lane & 15, oi = it * 16 + col;
  if (oi < m.n_inst) {
    for (int r = 0; r < 4; ++r) {
      const int orow = rt * 16 + (lane >> 4) + 4 * r;
      if (orow < m.nk) m.out[(int64_t)oi * hdsm::KROWS + orow] = acc[r];
    }
  }
}

__global__ __launch_bounds__(256) void k_plan_prepass(int N, int n_rob, const double* __restrict__ plans,
                                                      const uint8_t* __restrict__ has_plan, double* __restrict__ pos,
                                                      double* __restrict__ bounds, int n_order, const int32_t* __restrict__ key_prev,
                                                      const int32_t* __restrict__ agent_id, int32_t* __restrict__ order, SetupMapArgs sm) {
  if (sm.out != nullptr && (int)blockIdx.x >= sm.first_block) {  // the workgroups behind the pre-pass proper: four tiles of the set-up map each
    setup_map_tile(sm, ((int)blockIdx.x - sm.first_block) * 4 + ((int)threadIdx.x >> 6));
    return;
  }
  if (order != nullptr && (int)blockIdx.x == (n_rob + 15) / 16) {  // one extra workgroup: the launch order of the solve that follows
    launch_order_block(n_order, key_prev, agent_id, order);
    return;
  }
  if ((int)blockIdx.x >= (n_rob + 15) / 16) return;
  // 16 lanes per agent (N <= 16 = HDSM_MAX_HOR): lane i copies the position of step i + 1, the box / sphere reductions run
  // over the 16-lane group with DPP-able shuffles — every load of a plan is issued at once instead of N dependent ones
  const int tid = (int)threadIdx.x, i = tid & 15;
  const int k = (int)blockIdx.x * 16 + (tid >> 4);
  const bool live = k < n_rob;
  const bool has = live && has_plan[k];
  const bool on = has && i < N;
  double p[3] = {0.0, 0.0, 0.0};
  if (on) {
    const double* rec = plans + ((int64_t)k * (N + 1) + 1 + i) * 9;
    p[0] = rec[0], p[1] = rec[1], p[2] = rec[2];
  }
  if (live && i < N) {
    double* pk = pos + ((int64_t)k * N + i) * 3;
    pk[0] = p[0], pk[1] = p[1], pk[2] = p[2];
  }
  if (!bounds) return;
  double lo[3], hi[3];
  for (int ax = 0; ax < 3; ++ax) lo[ax] = on ? p[ax] : 1e300, hi[ax] = on ? p[ax] : -1e300;
  for (int off = 8; off > 0; off >>= 1)
    for (int ax = 0; ax < 3; ++ax) {
      lo[ax] = fmin(lo[ax], __shfl_xor(lo[ax], off, 16));
      hi[ax] = fmax(hi[ax], __shfl_xor(hi[ax], off, 16));
    }
  const double cx = 0.5 * (lo[0] + hi[0]), cy = 0.5 * (lo[1] + hi[1]), cz = 0.5 * (lo[2] + hi[2]);
  const double ux = p[0] - cx, uy = p[1] - cy, uz = p[2] - cz;
  double r2 = on ? ux * ux + uy * uy + uz * uz : 0.0;
  bool finite = !on || (r2 == r2);
  for (int off = 8; off > 0; off >>= 1) {
    r2 = fmax(r2, __shfl_xor(r2, off, 16));
    finite = finite && __shfl_xor((int)finite, off, 16);
  }
  if (live && i == 0) {
    double4 out = {0.0, 0.0, 0.0, -1.0};
    if (has) {
      out.x = cx, out.y = cy, out.z = cz;
      out.w = sqrt(r2) * (1.0 + 1e-9);
      if (!finite || !(out.w >= 0.0) || !(out.w < 1e299)) out.w = 1e300;  // non-finite plan: never culled, the step-by-step test decides
    }
    *reinterpret_cast<double4*>(bounds + 4 * (int64_t)k) = out;
  }
}

// Longest-processing-time-first launch order. A launch lasts as long as its slowest workgroup chain: with more instances than
// resident workgroups (2 per CU) an expensive instance that happens to start late sets the kernel time. Workgroups are
// dispatched in index order, so workgroup w takes instance order[w], the instances sorted by the key they left in the
// PREVIOUS launch on this handle (largest first; a counting sort on 256 values): the time the instance took, or the maximum if it
// found no solution (hdsm_core.h, st_key). The previous replan of the same agent is a good predictor (gridlocked neighbourhoods
// persist); the answer of an instance does not depend on the order. An entry is the pair (instance, its agent id): the workgroup
// then needs no second, dependent load (agent_id[instance]) before it can ask for the agent's own plan.
__device__ void launch_order_block(int n_inst, const int32_t* __restrict__ key_prev, const int32_t* __restrict__ agent_id, int32_t* __restrict__ order) {
  __shared__ int bucket[256];
  const int tid = (int)threadIdx.x, nt = (int)blockDim.x;
  for (int b = tid; b < 256; b += nt) bucket[b] = 0;
  __syncthreads();
  for (int k = tid; k < n_inst; k += nt) {
    const int it = key_prev[k];
    atomicAdd(&bucket[255 - (it < 0 ? 0 : (it > 255 ? 255 : it))], 1);
  }
  __syncthreads();
  // exclusive prefix over the 256 buckets, a bucket per thread (every launch of this block has 256 threads): scan inside the wavefront,
  // then the totals of the wavefronts before. (One thread walked the buckets before round 6, 256 dependent LDS round trips; the bench line
  // does not see the difference — the pre-pass launch is as long as its set-up map tiles.)
  {
    __shared__ int wave_total[4];
    const int c = bucket[tid & 255];
    int incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(incl, off, 64);
      if ((tid & 63) >= off) incl += v;
    }
    if ((tid & 63) == 63) wave_total[(tid >> 6) & 3] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < ((tid >> 6) & 3); ++w) base += wave_total[w];
    if (tid < 256) bucket[tid] = base + incl - c;
  }
  __syncthreads();
  for (int k = tid; k < n_inst; k += nt) {
    const int it = key_prev[k];
    const int slot = atomicAdd(&bucket[255 - (it < 0 ? 0 : (it > 255 ? 255 : it))], 1);
    order[2 * slot] = k, order[2 * slot + 1] = agent_id[k];
  }
}
__global__ __launch_bounds__(256) void k_launch_order(int n_inst, const int32_t* __restrict__ key_prev, const int32_t* __restrict__ agent_id, int32_t* __restrict__ order) {
  launch_order_block(n_inst, key_prev, agent_id, order);  // level 1 has no pre-pass to ride on
}

// hdsm_publish_device / hdsm_exchange_device: the has_plan flag travels inside the record (first entry NaN = no plan)
__global__ __launch_bounds__(256) void k_publish(int rec, int per, int n_local, const double* __restrict__ traj,
                                                 const uint8_t* __restrict__ has_local, double* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)per * rec) return;
  const int k = (int)(idx / rec), e = (int)(idx % rec);
  const bool has = k < n_local && has_local[k];
  double v = has ? traj[idx] : 0.0;
  if (!has && e == 0) v = __longlong_as_double(0x7ff8000000000000LL);
  out[idx] = v;
}
__global__ __launch_bounds__(256) void k_has_from_sentinel(int rec, int n, const double* __restrict__ plans,
                                                           uint8_t* __restrict__ has) {
  const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (k >= n) return;
  const double v = plans[(int64_t)k * rec];
  has[k] = (v == v) ? 1 : 0;
}

// planes[n_inst][N][n_rob][4] for tests / level-1 callers (AC:1100-1205)
__global__ __launch_bounds__(256) void k_tasc_planes(const hdsm::Consts* __restrict__ cp, int n_inst, int n_rob,
                                                     const int32_t* agent_id, const double* state,
                                                     const double* plans, const uint8_t* has_plan,
                                                     double* planes) {
  const hdsm::Consts& c = *cp;
  const int N = c.N;
  const int64_t total = (int64_t)n_inst * N * n_rob;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx % n_rob);
    const int i = (int)((idx / n_rob) % N);
    const int inst = (int)(idx / ((int64_t)n_rob * N));
    const int self = agent_id[inst];
    double row[4] = {0, 0, 0, 0};
    if (k != self && has_plan[k]) {
      double cp3[3];
      const bool own = self >= 0 && self < n_rob && has_plan[self];
      for (int ax = 0; ax < 3; ++ax)
        cp3[ax] = own ? plans[((int64_t)self * (N + 1) + (i + 1)) * 9 + ax] : state[(int64_t)inst * 9 + ax];
      double tmp[4];
      if (hdsm::Solver<32, 16>::tasc_plane(c, cp3, plans + ((int64_t)k * (N + 1) + (i + 1)) * 9, tmp))
        row[0] = tmp[0], row[1] = tmp[1], row[2] = tmp[2], row[3] = tmp[3];
    }
    double* out = planes + idx * 4;
    out[0] = row[0], out[1] = row[1], out[2] = row[2], out[3] = row[3];
  }
}

// ---- next row f1: reference trajectory (AC:1449-1553). One workgroup per agent: the neighbour term of
// ComputePathVelocity is a min-reduction over (step, neighbour) streamed from the all-gathered plans buffer
// (thread <-> neighbour, all N+1 positions of that neighbour read back to back), SamplePath is a short sequential
// walk done by one thread, the velocity references are elementwise.
struct RefArgs {
  int32_t n_inst, n_rob, pmax, N;
  double dt;
  hdsm_ref_config cfg;
  const int32_t* agent_id;
  const double* path;
  const int32_t* n_path;
  const double* vel_cap;
  const double* plans;
  const uint8_t* has_plan;
  double* ref_full;
  double* ref;
  double* path_vel;
  const double* rpos;  // [n_rob][N + 1][3] positions of steps 0..N, packed (k_ref_pack)
  const double* rsph;  // [n_rob][4] enclosing sphere of those positions (radius < 0: no plan)
  double wocc[hdsm::MAXH + 1];  // GetVelocityLimit's weight of step i (AC:1791-1795, 1805-1817): config only, evaluated by the host's libm
};

// Positions of steps 0..N of every published plan, packed, and their enclosing sphere: 16 lanes per agent (N + 1 <= 17: lane
// 15 also takes step 16). The velocity limit reads 24 B per (neighbour, step) from here instead of a 72-B stride of the
// records, and skips a neighbour whose sphere is further away than the closest one found.
// In the device-resident loop (one stream, same plans buffer for the reference and the solve of a round) this kernel also leaves
// what k_plan_prepass would compute a few microseconds later from the same records — positions of steps 1..N (`pos`), their
// bounding sphere (`bounds`, may be null), the launch order of the solve (one extra workgroup) — and the solve skips its pre-pass.
__global__ __launch_bounds__(256) void k_ref_pack(int N, int n_rob, const double* __restrict__ plans,
                                                   const uint8_t* __restrict__ has_plan, double* __restrict__ rpos,
                                                   double* __restrict__ rsph, double* __restrict__ pos, double* __restrict__ bounds,
                                                   int n_order, const int32_t* __restrict__ key_prev, const int32_t* __restrict__ agent_id,
                                                   int32_t* __restrict__ order) {
  if (order != nullptr && blockIdx.x == gridDim.x - 1) {
    launch_order_block(n_order, key_prev, agent_id, order);
    return;
  }
  const int tid = (int)threadIdx.x, i = tid & 15;
  const int k = (int)blockIdx.x * 16 + (tid >> 4);
  const bool live = k < n_rob;
  const bool has = live && has_plan[k];
  double p[2][3] = {{0, 0, 0}, {0, 0, 0}};
  bool on[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int st = i + 16 * u;
    on[u] = has && st <= N && (u == 0 || i == 0);
    if (on[u]) {
      const double* rec = plans + ((int64_t)k * (N + 1) + st) * 9;
      p[u][0] = rec[0], p[u][1] = rec[1], p[u][2] = rec[2];
    }
    if (live && st <= N && (u == 0 || i == 0)) {
      double* pk = rpos + ((int64_t)k * (N + 1) + st) * 3;
      pk[0] = p[u][0], pk[1] = p[u][1], pk[2] = p[u][2];
      if (pos != nullptr && st >= 1) {
        double* pq = pos + ((int64_t)k * N + st - 1) * 3;
        pq[0] = p[u][0], pq[1] = p[u][1], pq[2] = p[u][2];
      }
    }
  }
  if (bounds != nullptr) {  // the sphere of steps 1..N, exactly as k_plan_prepass forms it
    const bool in0 = on[0] && i >= 1;
    double lo[3], hi[3];
    for (int ax = 0; ax < 3; ++ax) {
      lo[ax] = in0 ? p[0][ax] : 1e300, hi[ax] = in0 ? p[0][ax] : -1e300;
      if (on[1]) lo[ax] = fmin(lo[ax], p[1][ax]), hi[ax] = fmax(hi[ax], p[1][ax]);
    }
    for (int off = 8; off > 0; off >>= 1)
      for (int ax = 0; ax < 3; ++ax) {
        lo[ax] = fmin(lo[ax], __shfl_xor(lo[ax], off, 16));
        hi[ax] = fmax(hi[ax], __shfl_xor(hi[ax], off, 16));
      }
    const double cx = 0.5 * (lo[0] + hi[0]), cy = 0.5 * (lo[1] + hi[1]), cz = 0.5 * (lo[2] + hi[2]);
    double r2 = 0.0;
    if (in0) {
      const double ux = p[0][0] - cx, uy = p[0][1] - cy, uz = p[0][2] - cz;
      r2 = ux * ux + uy * uy + uz * uz;
    }
    if (on[1]) {
      const double ux = p[1][0] - cx, uy = p[1][1] - cy, uz = p[1][2] - cz;
      r2 = fmax(r2, ux * ux + uy * uy + uz * uz);
    }
    bool finite = r2 == r2;
    for (int off = 8; off > 0; off >>= 1) {
      r2 = fmax(r2, __shfl_xor(r2, off, 16));
      finite = finite && __shfl_xor((int)finite, off, 16);
    }
    if (live && i == 0) {
      double4 out = {0.0, 0.0, 0.0, -1.0};
      if (has) {
        out.x = cx, out.y = cy, out.z = cz;
        out.w = sqrt(r2) * (1.0 + 1e-9);
        if (!finite || !(out.w >= 0.0) || !(out.w < 1e299)) out.w = 1e300;
      }
      *reinterpret_cast<double4*>(bounds + 4 * (int64_t)k) = out;
    }
  }
  double lo[3], hi[3];
  for (int ax = 0; ax < 3; ++ax) {
    lo[ax] = on[0] ? p[0][ax] : 1e300, hi[ax] = on[0] ? p[0][ax] : -1e300;
    if (on[1]) lo[ax] = fmin(lo[ax], p[1][ax]), hi[ax] = fmax(hi[ax], p[1][ax]);
  }
  for (int off = 8; off > 0; off >>= 1)
    for (int ax = 0; ax < 3; ++ax) {
      lo[ax] = fmin(lo[ax], __shfl_xor(lo[ax], off, 16));
      hi[ax] = fmax(hi[ax], __shfl_xor(hi[ax], off, 16));
    }
  const double cx = 0.5 * (lo[0] + hi[0]), cy = 0.5 * (lo[1] + hi[1]), cz = 0.5 * (lo[2] + hi[2]);
  double r2 = 0.0;
#pragma unroll
  for (int u = 0; u < 2; ++u)
    if (on[u]) {
      const double ux = p[u][0] - cx, uy = p[u][1] - cy, uz = p[u][2] - cz;
      r2 = fmax(r2, ux * ux + uy * uy + uz * uz);
    }
  bool finite = r2 == r2;
  for (int off = 8; off > 0; off >>= 1) {
    r2 = fmax(r2, __shfl_xor(r2, off, 16));
    finite = finite && __shfl_xor((int)finite, off, 16);
  }
  if (live && i == 0) {
    double4 out = {0.0, 0.0, 0.0, -1.0};
    if (has) {
      out.x = cx, out.y = cy, out.z = cz;
      out.w = sqrt(r2) * (1.0 + 1e-9);
      if (!finite || !(out.w >= 0.0) || !(out.w < 1e299)) out.w = 1e300;  // a non-finite plan is never skipped
    }
    *reinterpret_cast<double4*>(rsph + (int64_t)k * 4) = out;
  }
}

// wave reductions of k_reference: DPP row rotations + v_readlane (hdsm_wave_gi.h) — a __shfl_xor stage on a double is two
// ds_bpermute round trips, and the kernel reduces 14 values per instance
__device__ __forceinline__ double ref_wave_min(double v) { return -hdsm::wave_max64(-v); }

// NT threads per instance: 256 for small batches (more lanes on the one instance's neighbour scans), 64 — one wavefront, every
// instance of a 1024-agent round resident at once, the workgroup barriers of the reductions cost nothing — for large ones
// (46 -> ~20 us per 1024-agent round).
template <int NT>
__global__ __launch_bounds__(NT) void k_reference(RefArgs a) {
  __shared__ double own[hdsm::MAXH + 1][3];
  __shared__ double wocc[hdsm::MAXH + 1];
  __shared__ double red[NT];
  __shared__ double d2w[NT / 64][hdsm::MAXH + 1];
  __shared__ int idx[NT];
  constexpr int PATH_LDS = 64;
  __shared__ double spath[PATH_LDS * 3];
  __shared__ double pts[hdsm::MAXH + 1][3];
  __shared__ int cnt_s;
  constexpr int SURV_CAP = 2048;
  __shared__ int surv[SURV_CAP];
  __shared__ int surv_n;
  const int inst = blockIdx.x, tid = threadIdx.x, N = a.N;
  const int self = a.agent_id[inst];
  const int np = min(max(a.n_path[inst], 1), a.pmax);  // the host wrapper rejects counts outside [1, pmax]; device callers are clamped
  const bool own_has = self >= 0 && self < a.n_rob && a.has_plan[self];
  // the polyline goes through LDS: the sampling walk below is one thread's chain, and every global read in it was a
  // dependent round trip
  const double* pth_g = a.path + (int64_t)inst * a.pmax * 3;
  const bool path_fits = np <= PATH_LDS;
  if (path_fits)
    for (int e = tid; e < np * 3; e += NT) spath[e] = pth_g[e];
  const double* pth = path_fits ? spath : pth_g;
  if (tid <= N) {
    for (int c = 0; c < 3; ++c) own[tid][c] = own_has ? a.plans[((int64_t)self * (N + 1) + tid) * 9 + c] : 0.0;
    wocc[tid] = a.wocc[tid];
  }
  __syncthreads();
  double pv = a.vel_cap ? a.vel_cap[inst] : a.cfg.path_vel_max;
  if (pv > a.cfg.path_vel_max) pv = a.cfg.path_vel_max;
  // The limit of one (neighbour, step) pair, v = v_min + (v_max - v_min) (1 - w_i / exp(k d)), does not decrease with the
  // distance d (k >= 0, w_i >= 0, v_max >= v_min; every operation of the chain is monotone), so the minimum over the
  // neighbours is taken on the SQUARED distances — three subtractions and three multiply-adds per pair — and the square root,
  // the exponential and the division are evaluated once per step on the closest neighbour instead of once per pair.
  const bool monotone = a.cfg.sens_dist >= 0 && a.cfg.path_vel_max >= a.cfg.path_vel_min;
  if (own_has && np >= 2 && monotone) {
    // (1) the neighbour whose sphere is closest: its exact squared distances bound the minima from above
    const double4 ss = *reinterpret_cast<const double4*>(a.rsph + (int64_t)self * 4);
    double gbest = DBL_MAX;
    int jbest = -1;
    constexpr int UB = 8;  // sphere records in flight per thread: the scans are chains of L2 round trips otherwise
    for (int j0 = tid; j0 < a.n_rob; j0 += UB * NT) {
      double4 sj[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int j = j0 + u * NT;
        sj[u] = *reinterpret_cast<const double4*>(a.rsph + (int64_t)(j < a.n_rob ? j : self) * 4);
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int j = j0 + u * NT;
        if (j >= a.n_rob || j == self || sj[u].w < 0) continue;  // (w < 0: no plan)
        const double cx = sj[u].x - ss.x, cy = sj[u].y - ss.y, cz = sj[u].z - ss.z;
        const double g = sqrt(cx * cx + cy * cy + cz * cz) - sj[u].w - ss.w;  // every step of j is at least this far (g may be < 0)
        if (g < gbest || jbest < 0) gbest = g, jbest = j;
      }
    }
    {  // (which of several equally close spheres wins only moves the bound below)
      const double key = jbest >= 0 ? gbest : DBL_MAX;
      const double m = ref_wave_min(key);
      const unsigned long long who = __ballot(jbest >= 0 && key == m);
      const int src = who != 0ull ? __ffsll((long long)who) - 1 : 0;
      jbest = who != 0ull ? __builtin_amdgcn_readlane(jbest, src) : -1;
      gbest = m;
    }
    if constexpr (NT > 64) {
      if ((tid & 63) == 0) red[tid >> 6] = gbest, idx[tid >> 6] = jbest;
      __syncthreads();
      gbest = red[0], jbest = idx[0];
#pragma unroll
      for (int w = 1; w < NT / 64; ++w)
        if (idx[w] >= 0 && (jbest < 0 || red[w] < gbest)) gbest = red[w], jbest = idx[w];
      __syncthreads();
    }
    const int jstar = jbest;
    double umax = 0.0;
    {
      const int lane = tid & 63;
      double u = 0.0;
      if (lane <= N && jstar >= 0) {
        const double* rp = a.rpos + ((int64_t)jstar * (N + 1) + lane) * 3;
        const double dx = own[lane][0] - rp[0], dy = own[lane][1] - rp[1], dz = own[lane][2] - rp[2];
        u = dx * dx + dy * dy + dz * dz;
        u = (u == u) ? u : DBL_MAX;  // (a non-finite plan bounds nothing)
      }
      u = hdsm::wave_max64(u);
      umax = u;
    }
    // (2) minima of the squared distances over the neighbours that can still lower one of them
    double d2min[hdsm::MAXH + 1];
#pragma unroll
    for (int i = 0; i <= hdsm::MAXH; ++i) d2min[i] = DBL_MAX;
    // The neighbours that pass the sphere test are first LISTED (LDS) and then shared out evenly, one per thread and trip: taken
    // where they are found, a trip of the scan cost a full round trip to memory for the whole wavefront whenever ANY lane had a
    // survivor in it — 16 round trips per instance in a dense ring, half of this kernel's time.
    auto drain = [&]() {
      __syncthreads();
      const int cnt = surv_n < SURV_CAP ? surv_n : SURV_CAP;
      for (int k = tid; k < cnt; k += NT) {
        const double* rp = a.rpos + (int64_t)surv[k] * (N + 1) * 3;
#pragma unroll
        for (int i = 0; i <= hdsm::MAXH; ++i)
          if (i <= N) {
            const double dx = own[i][0] - rp[3 * i], dy = own[i][1] - rp[3 * i + 1], dz = own[i][2] - rp[3 * i + 2];
            d2min[i] = fmin(d2min[i], dx * dx + dy * dy + dz * dz);
          }
      }
      __syncthreads();
      if (tid == 0) surv_n = 0;
      __syncthreads();
    };
    if (tid == 0) surv_n = 0;
    __syncthreads();
    int listed = 0;  // (an upper bound of surv_n, the same in every thread)
    for (int j0 = tid; j0 - tid < a.n_rob; j0 += UB * NT) {
      if (listed + UB * NT > SURV_CAP) drain(), listed = 0;
      double4 sj[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int j = j0 + u * NT;
        sj[u] = *reinterpret_cast<const double4*>(a.rsph + (int64_t)(j < a.n_rob ? j : self) * 4);
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int j = j0 + u * NT;
        if (j >= a.n_rob || j == self || sj[u].w < 0) continue;
        const double cx = sj[u].x - ss.x, cy = sj[u].y - ss.y, cz = sj[u].z - ss.z;
        // (squared: all its steps are further than the closest neighbour's when the gap between the spheres is)
        const double c2 = cx * cx + cy * cy + cz * cz, reach = sj[u].w + ss.w + sqrt(umax) * (1.0 + 1e-9);
        if (c2 > reach * reach * (1.0 + 1e-9)) continue;
        surv[atomicAdd(&surv_n, 1)] = j;
      }
      listed += UB * NT;
    }
    drain();
#pragma unroll
    for (int i = 0; i <= hdsm::MAXH; ++i)
      if (i <= N) {
        double m = d2min[i];
        m = ref_wave_min(m);
        if ((tid & 63) == 0) d2w[tid >> 6][i] = m;
      }
    __syncthreads();
    if (tid <= N) {
      double m = d2w[0][tid];
#pragma unroll
      for (int w = 1; w < NT / 64; ++w) m = fmin(m, d2w[w][tid]);
      if (m < DBL_MAX) {
        const double d = sqrt(m);
        const double alpha = (1 - wocc[tid] * (1 / exp(a.cfg.sens_dist * d)));
        const double v = a.cfg.path_vel_min + (a.cfg.path_vel_max - a.cfg.path_vel_min) * alpha;
        if (v < pv) pv = v;
      }
    }
  } else if (own_has && np >= 2) {  // (a configuration whose limit is not monotone in the distance: every pair is evaluated)
    for (int j = tid; j < a.n_rob; j += NT) {
      if (j == self || !a.has_plan[j]) continue;
      const double* rec = a.plans + (int64_t)j * (N + 1) * 9;
      for (int i = 0; i <= N; ++i) {
        const double dx = own[i][0] - rec[9 * i], dy = own[i][1] - rec[9 * i + 1], dz = own[i][2] - rec[9 * i + 2];
        const double d = sqrt(dx * dx + dy * dy + dz * dz);
        const double alpha = (1 - wocc[i] * (1 / exp(a.cfg.sens_dist * d)));
        const double v = a.cfg.path_vel_min + (a.cfg.path_vel_max - a.cfg.path_vel_min) * alpha;
        if (v < pv) pv = v;
      }
    }
  }
  pv = ref_wave_min(pv);
  if constexpr (NT > 64) {
    if ((tid & 63) == 0) red[tid >> 6] = pv;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) pv = fmin(pv, red[w]);
  }
  pv = (np < 2) ? 0.0 : pv;
  if (tid == 0) {  // SamplePath, AC:1591-1663
    int cnt = 0;
    if (np < 2) {
      for (int i = 0; i < N; ++i, ++cnt)
        for (int c = 0; c < 3; ++c) pts[cnt][c] = pth[c];
    } else {
      const double samp = pv * a.dt;
      int path_idx = 1, ref_idx = 0;
      double cur[3] = {pth[0], pth[1], pth[2]};
      for (int c = 0; c < 3; ++c) pts[0][c] = cur[c];
      cnt = 1;
      double limit = samp;
      while (ref_idx < N) {
        const double* nx = pth + 3 * path_idx;
        const double d0 = nx[0] - cur[0], d1 = nx[1] - cur[1], d2 = nx[2] - cur[2];
        const double dist_next = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
        if (dist_next > limit) {
          cur[0] = cur[0] + limit * d0 / dist_next, cur[1] = cur[1] + limit * d1 / dist_next;
          cur[2] = cur[2] + limit * d2 / dist_next;
          for (int c = 0; c < 3; ++c) pts[cnt][c] = cur[c];
          ++cnt, ++ref_idx;
          limit = fmax(0.0, samp - a.cfg.path_vel_dec * a.dt);
        } else {
          cur[0] = nx[0], cur[1] = nx[1], cur[2] = nx[2];
          if (++path_idx == np) {
            for (int i = ref_idx; i < N; ++i, ++cnt)
              for (int c = 0; c < 3; ++c) pts[cnt][c] = pth[3 * (np - 1) + c];
            break;
          }
          limit -= dist_next;
        }
      }
    }
    cnt_s = cnt;
    a.path_vel[inst] = pv;
  }
  __syncthreads();
  const int cnt = cnt_s;
  if (tid <= N) {  // velocity reference AC:1527-1547: row i looks back from i+1; the last row copies the previous one
    const int i = tid < cnt ? tid : cnt - 1;
    const int ii = (i + 1 < cnt) ? i : (cnt >= 2 ? cnt - 2 : 0);  // pair used by row i
    double v[3] = {0, 0, 0};
    if (cnt > 1) {
      const double d0 = pts[ii][0] - pts[ii + 1][0], d1 = pts[ii][1] - pts[ii + 1][1], d2 = pts[ii][2] - pts[ii + 1][2];
      const double dist = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
      if (dist > 1e-2) v[0] = pv * d0 / dist, v[1] = pv * d1 / dist, v[2] = pv * d2 / dist;
    }
    double* out = a.ref_full + ((int64_t)inst * (N + 1) + tid) * 6;
    for (int c = 0; c < 3; ++c) out[c] = pts[i][c], out[3 + c] = v[c];
    if (a.ref && tid < N) {
      double* o2 = a.ref + ((int64_t)inst * N + tid) * 6;
      for (int c = 0; c < 6; ++c) o2[c] = out[c];
    }
  }
}

struct Handle {
  int device = 0;
  int max_inst = 0, n_rob_max = 0;
  int N = 0, P = 0, RS = 0, n = 0;
  int threads = 256, cus = 256;
  int bounds_min = 256;       // swarms of at least this many agents get the sphere prefilter (HDSM_BOUNDS_MIN)
  int duo_min = 0;            // batches of at least this many instances run two workgroups per CU (HDSM_DUO_MIN; set at create: CUs + 1)
  int tri_min = 0;            // ... and of at least this many three 128-thread workgroups per CU (HDSM_TRI_MIN; 2 x CUs + 1, 0 = never)
  int quad_min = 0;           // ... and of at least this many four per CU, small LDS layout (HDSM_QUAD_MIN; 3 x CUs + 1, 0 = never)
  // subtree splitting (launch_split): 0 never, 1 always, 2 automatic (when the previous launch saw a deep tree)
  int split_mode = 2, split_budget = 0, split_ttl = 0;  // split_budget 0: by batch size, see launch()
  int rec_cap = 2048, rows_cap = 0, items_cap = 0, sub_slots_n = 0, pool_cap = 0, item_budget = 32, item_min = 0, poll_sleep = 2;  // hand-over records, staged rows per record, queue length, persistent workgroups of pass 2
  int32_t* h_ovf_flag = nullptr;    // pinned host word: an instance ended on a staging overflow (Args::ovf_flag), and its device alias
  int32_t* d_ovf_flag = nullptr;
  int rescue_ttl = 0;               // launches left that carry the rescue pass
  bool last_small = false;          // the last launch used a kernel shape with a reduced staging area
  hdsm::Args last_args;             // ... and its arguments (the host-buffer path adds the rescue pass at once)
  int32_t* h_tree_flag = nullptr;   // pinned host word the kernels raise (Args::tree_flag), and its device alias
  int32_t* d_tree_flag = nullptr;
  hdsm::SplitRec* d_recs = nullptr;  // hand-over records of pass 1 and their staged rows (Args::recs, rec_cand, rec_mw, rec_src)
  double* d_rec_cand = nullptr;
  long long* d_rec_mw = nullptr;
  int32_t *d_rec_src = nullptr, *d_rec_count = nullptr, *d_items = nullptr;
  int32_t* d_slot_busy = nullptr;    // [pool_cap] snapshot-scratch slots of pass 2 taken (Args::slot_busy)
  int32_t *h_item_total = nullptr, *d_item_total = nullptr;  // pinned host word: items queued by the last split launch (the merge writes it)
  int32_t *d_split = nullptr, *d_sub_stats = nullptr, *d_sub_warm = nullptr, *d_sub_status = nullptr;
  unsigned long long* d_inc = nullptr;
  int32_t* d_node_pool = nullptr;  // [max_inst] nodes the sub-blocks of an instance may still open (Args::node_pool)
  double *d_sub_traj = nullptr, *d_sub_ctrl = nullptr, *d_sub_obj = nullptr, *d_sub_scratch = nullptr;
  uint8_t* d_sub_used = nullptr;
  bool sub_ready = false;
  void* h_out = nullptr;      // pinned staging of hdsm_replan's outputs (grow-only)
  size_t h_out_cap = 0;
  double* d_bounds = nullptr; // [n_rob_max][4]
  double* d_setup = nullptr;  // [max_inst][KROWS] the set-up map of every instance of a launch (Args::setup)
  int setup_mfma = 1;         // 0 (HDSM_SETUP_MFMA=0): every instance applies the map itself (the form of rounds 1-4)
  double* d_pos = nullptr;    // [n_rob_max][N][3] packed positions (pre-pass)
  double* d_rpos = nullptr;   // [n_rob_max][N + 1][3] packed positions of steps 0..N (k_ref_pack)
  double* d_rsph = nullptr;   // [n_rob_max][4] their spheres
  int32_t* d_order = nullptr; // [max_inst][2] launch order: (instance, its agent id) per workgroup (k_launch_order)
  int order_min = 0;          // batches of at least this many instances are launched most-expensive-first (0 = never)
  uint8_t* d_zero = nullptr;  // n_rob_max zero bytes (has_plan of level 1)
  hipEvent_t ev_done = nullptr;  // recorded after every launch: orders launches that arrive on different streams
  hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr;  // hdsm_set_kernel_timing: around the solver kernel alone (after the pre-pass)
  bool time_kernel = false, timed = false;
  bool launched = false;
  struct Buf {                // grow-only device scratch of the host-pointer entry points
    void* p = nullptr;
    size_t cap = 0;
  } b_planes, b_common, b_ncommon, b_path, b_cap, b_full, b_pv, b_np;
  hdsm_params prm{};
  hdsm::Consts* d_consts = nullptr;
  double* d_scratch = nullptr;
  int64_t scratch_stride = 0;
  int32_t* d_stats = nullptr;  // 8 * max_inst: iterations, nodes, sweeps, staged rows, sphere records, pairs, flags, launch-order key
  long long* d_prof = nullptr; // 24 * max_inst (HDSM_PROFILE builds)
  int32_t* d_warm = nullptr;   // (MAXNV + 2) * max_inst: previous optimal working sets (params.warm_start)
  // staging for the host-pointer entry points
  int32_t *d_agent = nullptr, *d_npoly = nullptr, *d_nrows = nullptr, *d_status = nullptr;
  double *d_state = nullptr, *d_ref = nullptr, *d_A = nullptr, *d_b = nullptr, *d_plans = nullptr;
  double *d_traj = nullptr, *d_ctrl = nullptr, *d_obj = nullptr;
  uint8_t *d_has = nullptr, *d_used = nullptr;
  hipStream_t stream = nullptr;
  hipStream_t last_stream = nullptr;
  const double* prepass_for = nullptr;  // device loop: the plans buffer k_ref_pack has just packed for the solve as well (launch() skips its pre-pass once)
  int prepass_n_rob = 0, prepass_n_inst = 0;
  bool prepass_ordered = false;
  bool defer_done = false;  // the device-resident loop records ev_done once per round (hdsm_internal_record_done), not once per call
};

template <int NV, int NT>
int launch_nv(Handle* h, const hdsm::Args& a, hipStream_t st, int blocks) {
  constexpr int CM = (NV <= 32) ? CMAX30 : CMAX48;
  using Sol = hdsm::Solver<NV, CM>;
  const size_t shm = sizeof(typename Sol::S);
  auto kern = k_replan<NV, CM, NT>;
  static thread_local int attr_dev[2] = {-1, -1};
  if (attr_dev[a.item_mode ? 1 : 0] != h->device) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)shm));
    attr_dev[a.item_mode ? 1 : 0] = h->device;
  }
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(NT), shm, st, h->d_consts, a);
  HIP_TRY(hipGetLastError());
  return HDSM_OK;
}

int launch_duo(Handle* h, const hdsm::Args& a, hipStream_t st, int blocks) {
  using Sol = hdsm::Solver<32, CMAX_DUO>;
#ifndef HDSM_PROFILE  // (the counters of the profile build live in LDS: that build runs at a lower occupancy)
  static_assert(sizeof(typename Sol::S) * 2 <= 160 * 1024, "two instances must fit the LDS of one CU");
#endif
  const size_t shm = sizeof(typename Sol::S);
  auto kern = k_replan_duo<32, CMAX_DUO, 256>;
  static thread_local int attr_dev[2] = {-1, -1};
  if (attr_dev[a.item_mode ? 1 : 0] != h->device) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)shm));
    attr_dev[a.item_mode ? 1 : 0] = h->device;
  }
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), shm, st, h->d_consts, a);
  HIP_TRY(hipGetLastError());
  return HDSM_OK;
}

// THREE workgroups of 128 threads (two wavefronts: the iterating one and one helper) per CU, for batches that outnumber the
// resident slots of the two-per-CU kernel. A launch lasts as long as its slowest workgroup CHAIN: with 1024 instances on 512
// slots an instance that was predicted cheap and turns out long starts late and sets the kernel time (measured: mean span
// 119 us against 107 us for the slowest instance, profiles/r03_launch_timeline.json); with 768 slots the late starters begin
// when the first instances without iterations leave (~18 us) and finish inside the slowest instance. The sweeps and the set-up
// run on half the threads (+2..3 us per instance), the staging area shrinks to 384 rows (three states in 160 KB).
constexpr int CMAX_TRI = 384;
template <int NV, int CMAX, int NT>
__global__ __launch_bounds__(NT, 2) void k_replan_tri(const hdsm::Consts* __restrict__ cp, hdsm::Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using Sol = hdsm::Solver<NV, CMAX>;
  run_block<Sol>(reinterpret_cast<typename Sol::S*>(smem), cp, a);
}

// FOUR workgroups of 128 threads per CU: every instance of a 1024-agent round is resident at once (1024 slots), so no instance
// starts late behind a short one — in the rounds of the bench window whose instances are all of similar length (five of twenty)
// the three-per-CU launch ended 15-25 us after its slowest instance, set by a late starter. Four instance states fit 160 KB with
// the small LDS layout (Shm<.., SMALL>: 4 polyhedra of <= 20 rows, 512-neighbour chunks) and a staging area of 256 rows (the
// bench rounds stage <= 190; an overflow is re-solved by the rescue pass like for the other shared-CU kernels).
constexpr int CMAX_QUAD = 256;
template <int NV, int CMAX, int NT>
__global__ __launch_bounds__(NT, 2) void k_replan_quad(const hdsm::Consts* __restrict__ cp, hdsm::Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using Sol = hdsm::Solver<NV, CMAX, true>;
  run_block<Sol>(reinterpret_cast<typename Sol::S*>(smem), cp, a);
}
int launch_quad(Handle* h, const hdsm::Args& a, hipStream_t st, int blocks) {
  using Sol = hdsm::Solver<32, CMAX_QUAD, true>;
#ifndef HDSM_PROFILE  // (the counters of the profile build live in LDS: that build runs at a lower occupancy)
  static_assert(sizeof(typename Sol::S) * 4 <= 160 * 1024, "four instances must fit the LDS of one CU");
#endif
  const size_t shm = sizeof(typename Sol::S);
  auto kern = k_replan_quad<32, CMAX_QUAD, 128>;
  static thread_local int attr_dev = -1;
  if (attr_dev != h->device) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    attr_dev = h->device;
  }
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(128), shm, st, h->d_consts, a);
  HIP_TRY(hipGetLastError());
  return HDSM_OK;
}

// n > 30 (H up to 16): the factor needs more than 256 registers per lane, so a wavefront must have a SIMD to itself — but a
// 128-thread workgroup has only two, and TWO such workgroups (four wavefronts, one per SIMD) fit a CU once the staging area is cut
// to 720 rows (2 x 80 KB of LDS; 736 until round 5 added the per-level child bounds; 320 rows until the butterfly layout freed the
// 19 KB transposition buffer of the old code). Batches larger than the CU count are throughput-bound at one instance per CU (cfg 5:
// 4096 instances, 16 per CU one after the other), and every instance is one latency-bound wavefront: the second one doubles the rate.
// (Until round 6 a second instantiation with 320 rows stayed selectable for the staging-overflow test, HDSM_DUO48_ROWS=320. When the
// last spilled registers of these kernels were removed, THAT instantiation — and only it — began to end feasible instances as
// "infeasible" after two operations, deterministically per build, while builds with a spill, with -O2, with -fwrapv or with device
// printf in the loop gave the oracle's answers; the CPU execution of the same source is right in either wave order. The cause was not
// found — scripts/gpu_r6_overflow_raw.py reproduces it on the sources of that commit — so the product keeps ONE instantiation per
// launch shape, each of which the whole -m gpu suite, the fuzz and the oracle check of the timed rounds run through, and the
// overflow test fills the 256 rows of the four-per-CU kernel instead.)
constexpr int CMAX_DUO48 = 720;
template <int NV, int CMAX, int NT>
__global__ __launch_bounds__(NT, 1) void k_replan_duo48(const hdsm::Consts* __restrict__ cp, hdsm::Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using Sol = hdsm::Solver<NV, CMAX>;
  run_block<Sol>(reinterpret_cast<typename Sol::S*>(smem), cp, a);
}
int launch_duo48(Handle* h, const hdsm::Args& a, hipStream_t st, int blocks) {
  using Sol = hdsm::Solver<48, CMAX_DUO48>;
#ifndef HDSM_PROFILE  // (the counters of the profile build live in LDS: that build runs at a lower occupancy)
  static_assert(sizeof(typename Sol::S) * 2 <= 160 * 1024, "two instances must fit the LDS of one CU");
#endif
  const size_t shm = sizeof(typename Sol::S);
  auto kern = k_replan_duo48<48, CMAX_DUO48, 128>;
  static thread_local int attr_dev[2] = {-1, -1};
  if (attr_dev[a.item_mode ? 1 : 0] != h->device) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    attr_dev[a.item_mode ? 1 : 0] = h->device;
  }
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(128), shm, st, h->d_consts, a);
  HIP_TRY(hipGetLastError());
  return HDSM_OK;
}

int launch_tri(Handle* h, const hdsm::Args& a, hipStream_t st, int blocks) {
  using Sol = hdsm::Solver<32, CMAX_TRI>;
#ifndef HDSM_PROFILE  // (the counters of the profile build live in LDS: that build runs at a lower occupancy)
  static_assert(sizeof(typename Sol::S) * 3 <= 160 * 1024, "three instances must fit the LDS of one CU");
#endif
  const size_t shm = sizeof(typename Sol::S);
  auto kern = k_replan_tri<32, CMAX_TRI, 128>;
  static thread_local int attr_dev = -1;
  if (attr_dev != h->device) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    attr_dev = h->device;
  }
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(128), shm, st, h->d_consts, a);
  HIP_TRY(hipGetLastError());
  return HDSM_OK;
}

// ---- subtree splitting: set-up of pass 2, merge, lazily allocated state -------------------------------------------------
// hdsm_replan with page-locked input arrays: ONE kernel reads them from mapped host memory (coalesced reads over PCIe) into the
// handle's device arrays instead of nine stream-ordered copies, and of the static polyhedra only the rows that exist
// (n_rows_static of max_rows_static; the rest of the device array is never read by the solver). Blocks 0 .. n_inst - 1 take one
// instance each, the blocks after them the plans of every agent.
struct FetchArgs {
  int32_t n_inst, n_rob, N, P, RS, plan_blocks;
  const int32_t *agent_id, *n_poly, *n_rows;
  const double *state, *ref, *A, *b, *plans;
  const uint8_t* has;
  int32_t *d_agent, *d_npoly, *d_nrows;
  double *d_state, *d_ref, *d_A, *d_b, *d_plans;
  uint8_t* d_has;
};
__global__ __launch_bounds__(256) void k_fetch(FetchArgs a) {
  const int tid = (int)threadIdx.x, blk = (int)blockIdx.x;
  if (blk >= a.n_inst) {
    const int64_t total = (int64_t)a.n_rob * (a.N + 1) * 9;
    for (int64_t e = (int64_t)(blk - a.n_inst) * 256 + tid; e < total; e += (int64_t)a.plan_blocks * 256) a.d_plans[e] = a.plans[e];
    for (int e = (blk - a.n_inst) * 256 + tid; e < a.n_rob; e += a.plan_blocks * 256) a.d_has[e] = a.has[e];
    return;
  }
  const int k = blk, P = a.P, RS = a.RS;
  __shared__ int32_t rows_s[HDSM_MAX_POLY];  // (read once over PCIe, not once per entry)
  if (tid == 0) a.d_agent[k] = a.agent_id[k];
  const int np = a.n_poly[k];
  if (tid == 1) a.d_npoly[k] = np;
  if (tid < 9) a.d_state[(int64_t)k * 9 + tid] = a.state[(int64_t)k * 9 + tid];
  if (tid < P) {
    const int32_t nr = a.n_rows[(int64_t)k * P + tid];
    rows_s[tid] = nr, a.d_nrows[(int64_t)k * P + tid] = nr;
  }
  for (int e = tid; e < 6 * a.N; e += 256) a.d_ref[(int64_t)k * 6 * a.N + e] = a.ref[(int64_t)k * 6 * a.N + e];
  __syncthreads();
  for (int e = tid; e < P * RS * 4; e += 256) {  // entry (polyhedron j, row r, component c): c < 3 -> A, c = 3 -> b
    const int c = e & 3, jr = e >> 2, j = jr / RS, r = jr % RS;
    if (j >= np || r >= rows_s[j]) continue;
    const int64_t row = ((int64_t)k * P + j) * RS + r;
    if (c < 3) a.d_A[row * 3 + c] = a.A[row * 3 + c];
    else a.d_b[row] = a.b[row];
  }
}

// hdsm_replan with page-locked output arrays (hdsm_host_register): the results go from HBM straight into the caller's arrays —
// mapped host memory, written over PCIe by the device — and only for instances that HAVE a solution ("outputs are left
// untouched" otherwise): no staging download, no host-side filter copy. One 64-lane group per instance.
__global__ __launch_bounds__(256) void k_deliver(int n_inst, int trj, int ctl, int P, const double* __restrict__ traj, const double* __restrict__ ctrl,
                                                 const double* __restrict__ obj, const int32_t* __restrict__ status, const uint8_t* __restrict__ used,
                                                 double* o_traj, double* o_ctrl, double* o_obj, int32_t* o_status, uint8_t* o_used) {
  const int k = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
  if (k >= n_inst) return;
  const int stt = status[k];
  if (lane == 0) o_status[k] = stt;
  if (stt == HDSM_NO_SOLUTION) return;
  for (int e = lane; e < trj; e += 64) o_traj[(int64_t)k * trj + e] = traj[(int64_t)k * trj + e];
  for (int e = lane; e < ctl; e += 64) o_ctrl[(int64_t)k * ctl + e] = ctrl[(int64_t)k * ctl + e];
  if (lane < P) o_used[(int64_t)k * P + lane] = used[(int64_t)k * P + lane];
  if (lane == 0) o_obj[k] = obj[k];
}

// One wavefront per instance that pass 1 handed over: the best answer of its items becomes the instance's answer. `a` holds
// the instance-indexed arrays of the launch, `b` the arrays of pass 2 (indexed by the item's place in the queue).
__global__ __launch_bounds__(64) void k_split_merge(int N, int P, hdsm::Args a, hdsm::Args b) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && b.item_total != nullptr) *b.item_total = a.rec_count[1];  // (sizes the grid of the next split launch)
  hdsm::split_merge(N, P, a, b, (int)blockIdx.x, (int)threadIdx.x, 64);
}

hipError_t ensure_sub(Handle* h);

// the one-per-CU kernel (largest staging area) over the batch of `a`; only instances flagged HDSM_FLAG_STAGING_OVERFLOW work
int launch_rescue(Handle* h, const hdsm::Args& a, hipStream_t st) {
  hdsm::Args r = a;
  r.rescue = 1, r.order = nullptr, r.split_budget = 0, r.item_mode = 0, r.split_info = nullptr, r.inc_bits = nullptr, r.node_pool = nullptr;
  r.tree_flag = nullptr, r.tree_mark = 0, r.warm_out = r.warm, r.ovf_flag = nullptr;
  if (h->n <= hdsm::SPLIT_N_MAX) return h->threads == 64 ? launch_nv<32, 64>(h, r, st, r.n_inst) : launch_nv<32, 256>(h, r, st, r.n_inst);
  return h->threads == 64 ? launch_nv<48, 64>(h, r, st, r.n_inst) : launch_nv<48, 256>(h, r, st, r.n_inst);
}

int launch(Handle* h, hdsm::Args a, hipStream_t st) {
  a.scratch = h->d_scratch;
  a.scratch_stride = h->scratch_stride;
  a.st_iters = h->d_stats;
  a.st_nodes = h->d_stats + h->max_inst;
  a.st_sweeps = h->d_stats + 2 * h->max_inst;
  a.st_cand = h->d_stats + 3 * h->max_inst;
  a.st_sph = h->d_stats + 4 * h->max_inst;
  a.st_pairs = h->d_stats + 5 * h->max_inst;
  a.st_flags = reinterpret_cast<uint32_t*>(h->d_stats + 6 * h->max_inst);
  a.st_key = h->d_stats + 7 * h->max_inst;
  a.prof = h->d_prof;
  a.warm = (h->prm.warm_start && a.l1_rows == nullptr) ? h->d_warm : nullptr;
  // the handle's device state (snapshots, warm-start sets, prefilter records) is shared by all launches: a launch that
  // arrives on another stream than the previous one waits for it
  if (h->launched && st != h->last_stream) HIP_TRY(hipStreamWaitEvent(st, h->ev_done, 0));
  h->last_stream = st;
  a.bounds = nullptr, a.pos = nullptr, a.order = nullptr;
  const bool ordered = a.warm != nullptr && h->order_min > 0 && a.n_inst >= h->order_min;
  if (ordered) a.order = h->d_order;
  const bool packed = h->defer_done && a.l1_rows == nullptr && h->prepass_for == a.plans && h->prepass_n_rob == a.n_rob &&
                      h->prepass_n_inst == a.n_inst && h->prepass_ordered == ordered;
  h->prepass_for = nullptr;  // (good for one solve: the plans change with the commit that follows it)
  // the set-up map of every instance as one product on the matrix cores (k_plan_prepass, setup_map_tile): rides on the pre-pass launch
  SetupMapArgs sm{};
  a.setup = nullptr;
  if (a.l1_rows == nullptr && h->setup_mfma && h->d_setup != nullptr) {
    sm.c = h->d_consts, sm.state = a.state, sm.ref = a.ref, sm.out = h->d_setup, sm.n_inst = a.n_inst, sm.N = h->N, sm.nk = 3 * h->n + 12;
    sm.row_tiles = (sm.nk + 15) / 16;
    a.setup = h->d_setup;
  }
  const int sm_blocks = sm.out != nullptr ? (sm.row_tiles * ((a.n_inst + 15) / 16) + 3) / 4 : 0;
  if (packed) {
    a.pos = h->d_pos;
    a.bounds = a.n_rob >= h->bounds_min ? h->d_bounds : nullptr;
    if (sm_blocks > 0) {  // (the device-resident loop packs the plans elsewhere: the tiles alone)
      sm.first_block = 0;
      hipLaunchKernelGGL(k_plan_prepass, dim3(sm_blocks), dim3(256), 0, st, h->N, 0, a.plans, a.has_plan, h->d_pos, nullptr, 0, a.st_key, a.agent_id, nullptr, sm);
      HIP_TRY(hipGetLastError());
    }
  } else if (a.l1_rows == nullptr) {
    const bool pre = a.n_rob >= h->bounds_min;
    sm.first_block = (a.n_rob + 15) / 16 + (ordered ? 1 : 0);
    hipLaunchKernelGGL(k_plan_prepass, dim3(sm.first_block + sm_blocks), dim3(256), 0, st, h->N, a.n_rob, a.plans, a.has_plan,
                       h->d_pos, pre ? h->d_bounds : nullptr, a.n_inst, a.st_key, a.agent_id, ordered ? h->d_order : nullptr, sm);
    HIP_TRY(hipGetLastError());
    a.pos = h->d_pos;
    a.bounds = pre ? h->d_bounds : nullptr;
  } else if (ordered) {
    hipLaunchKernelGGL(k_launch_order, dim3(1), dim3(256), 0, st, a.n_inst, a.st_key, a.agent_id, h->d_order);
    HIP_TRY(hipGetLastError());
  }
  // one workgroup per agent-replan. The active-set iteration runs on wave 0 (factorisation in its registers);
  // with 256 threads the other three waves of the CU share the sweeps, the set-up and the leaf test.
  int rc;
  if (h->time_kernel) HIP_TRY(hipEventRecord(h->ev_k0, st));
  bool small = false;  // a kernel shape with a reduced staging area was used (two or three workgroups per CU)
  auto solve = [&](const hdsm::Args& x, int blocks) -> int {  // the kernel shape that suits `blocks` workgroups
    hdsm::Args y = x;
    y.n_inst = x.n_inst;
    // (`small` follows the shape that is actually launched — every shared-CU kernel has a reduced staging area — not the
    // thresholds: with HDSM_DUO_MIN=0 and HDSM_QUAD_MIN / HDSM_TRI_MIN set by hand the two can disagree)
    if (h->n <= hdsm::SPLIT_N_MAX && h->threads == 256 && h->quad_min > 0 && blocks >= h->quad_min && h->P <= 4 && h->RS <= 20) {
      small = true;
      return launch_quad(h, y, st, blocks);
    }
    if (h->n <= hdsm::SPLIT_N_MAX && h->threads == 256 && h->tri_min > 0 && blocks >= h->tri_min) {
      small = true;
      return launch_tri(h, y, st, blocks);
    }
    if (h->n <= hdsm::SPLIT_N_MAX && h->threads == 256 && h->duo_min > 0 && blocks >= h->duo_min) {
      small = true;
      return launch_duo(h, y, st, blocks);
    }
    if (h->n <= hdsm::SPLIT_N_MAX) return h->threads == 64 ? launch_nv<32, 64>(h, y, st, blocks) : launch_nv<32, 256>(h, y, st, blocks);
    if (h->threads == 256 && h->duo_min > 0 && blocks >= h->duo_min) {
      small = true;
      return launch_duo48(h, y, st, blocks);
    }
    return h->threads == 64 ? launch_nv<48, 64>(h, y, st, blocks) : launch_nv<48, 256>(h, y, st, blocks);
  };
  a.warm_out = a.warm;
  a.tree_flag = h->d_tree_flag;
  a.ovf_flag = h->d_ovf_flag, a.rescue = 0;
  // nodes after which an instance is handed over: small batches leave most CUs idle, so sub-blocks are free; in a batch that
  // fills the GPU every handed-over instance costs poly_hor set-ups and sweeps on busy CUs, so only the deep trees go
  // (measured on MI355X with the hand-over records of round 5 — a hand-over costs about one node, no search is repeated: cfg 3, 256
  // instances, 2 / 4 / 8 / 16 nodes -> 0.65 / 0.70 / 0.84 / 0.93 ms per round; cfg 5, 4096 instances, 16 / 24 / 32 / 48 / 96 nodes ->
  // 8.1 / 8.4 / 8.9 / 9.4 / 11.3 ms)
  const bool roomy = a.n_inst <= 2 * h->cus;
  // (round 6, after the dominance rule made the trees small — cfg 5: 77 k -> 23 k nodes per round: 8 / 8 instead of 16 / 16 for full
  // batches, 4.18 -> 3.51 ms on rounds 8..13 and 7.24 -> 5.86 ms on rounds 30..35, scripts/gpu_r6_cfg5_sweep.sh; cfg 3 is flat in both knobs)
  const int budget = h->split_budget > 0 ? h->split_budget : (roomy ? 2 : 8);
  a.tree_mark = budget > hdsm::TREE_MARK ? budget : hdsm::TREE_MARK;
  // Subtree splitting. A launch lasts as long as its slowest instance, and in obstacle worlds that is one agent between pillars
  // whose branch and bound needs hundreds of nodes while the other workgroups have been idle for milliseconds. When the last
  // launch met such a tree (tree_flag), this one runs in three kernels: (1) the ordinary solve with a small node budget — an
  // instance that exceeds it stops without outputs and records the step its root branched on; (2) poly_hor workgroups per
  // handed-over instance, workgroup j searching the subtree "polyhedron j at that step" (a partition of the search space that
  // does not depend on numerical noise), pruning against the best objective any of them has found (one atomicMin per
  // incumbent); (3) a merge that keeps the best answer. Same answers as the one-kernel form: the search is exact either way.
  bool split = h->split_mode == 1;
  if (h->split_mode == 2 && h->h_tree_flag != nullptr) {
    // The word is raised by kernels that may still be running when the next launch is enqueued (callers that queue dozens of
    // rounds back to back see it dozens of launches late), so a sighting keeps the split form on for the next 256 launches:
    // deep trees persist over rounds, and a split launch in which nothing is handed over costs three near-empty kernels.
    if (*h->h_tree_flag != 0) *h->h_tree_flag = 0, h->split_ttl = 256;
    if (h->split_ttl > 0) split = true, --h->split_ttl;
  }
  if (split && !h->sub_ready) split = ensure_sub(h) == hipSuccess;
  if (!split) {
    rc = solve(a, a.n_inst);
  } else {
    a.split_budget = budget, a.split_info = h->d_split, a.tree_mark = 0;  // (a.tree_flag stays: k_split_merge raises it for trees that are still deep)
    a.recs = h->d_recs, a.rec_cand = h->d_rec_cand, a.rec_mw = h->d_rec_mw, a.rec_src = h->d_rec_src, a.rec_count = h->d_rec_count, a.items = h->d_items;
    a.rec_cap = h->rec_cap, a.rows_cap = h->rows_cap, a.items_cap = h->items_cap, a.inc_bits = nullptr, a.node_pool = nullptr, a.item_status = h->d_sub_status;
    // the node budget is the INSTANCE's: every item starts with a small share of what pass 1 left of it and hands the unused part
    // back to the instance's pool when it finishes; an item that has used its share draws from that pool (NODE_CHUNK at a time)
    const int total_nodes = h->prm.max_nodes > 0 ? h->prm.max_nodes : 2000;
    const int left_nodes = total_nodes - budget > 64 ? total_nodes - budget : 64;
    const int node_cap = left_nodes / 128 > 0 ? left_nodes / 128 : 1;
    a.nodes_pool0 = left_nodes, a.node_cap = node_cap;
    HIP_TRY(hipMemsetAsync(h->d_rec_count, 0, 8 * sizeof(int32_t), st));
    HIP_TRY(hipMemsetAsync(h->d_slot_busy, 0, (size_t)h->pool_cap * sizeof(int32_t), st));
    hdsm::Args a1 = a;
    a1.inc_bits = h->d_inc, a1.node_pool = h->d_node_pool;  // (pass 1 only WRITES them, at a hand-over: its own search runs on Consts::max_nodes)
    rc = solve(a1, a.n_inst);
    if (rc) return rc;
    hdsm::Args b = a;
    b.item_mode = 1, b.order = nullptr, b.inc_bits = h->d_inc, b.node_pool = h->d_node_pool, b.tree_flag = nullptr;
    b.traj = h->d_sub_traj, b.ctrl = h->d_sub_ctrl, b.used = h->d_sub_used, b.status = h->d_sub_status, b.obj = h->d_sub_obj;
    b.scratch = h->d_sub_scratch, b.warm_out = h->d_sub_warm, b.prof = nullptr, b.pool_cap = h->pool_cap, b.slot_busy = h->d_slot_busy, b.item_total = h->d_item_total;
    b.split_budget = h->item_budget;  // an item whose subtree outgrows this many nodes hands over again (0: never)
    // ... or, while workgroups are waiting for items, this many: 2 in a batch that leaves CUs idle (cfg 3: 0.84 ms against 1.14 without),
    // 16 in one that fills the GPU (every look at the queue is two device-scope reads per node: cfg 5 8.15 ms with 2 - 4, 7.5 with >= 8)
    b.split_min = h->item_min > 0 ? h->item_min : (roomy ? 2 : 8);
    b.poll_sleep = h->poll_sleep;
    int32_t* ss = h->d_sub_stats;
    const size_t GI = (size_t)h->items_cap;
    b.st_iters = ss, b.st_nodes = ss + GI, b.st_sweeps = ss + 2 * GI, b.st_cand = ss + 3 * GI, b.st_sph = ss + 4 * GI, b.st_pairs = ss + 5 * GI;
    b.st_flags = reinterpret_cast<uint32_t*>(ss + 6 * GI), b.st_key = ss + 7 * GI;
    // pass 2: persistent workgroups, as many as fit the GPU at once (two per CU where the shape allows it)
    const bool two = h->threads == 256 && h->duo_min > 0;
    small = small || two;
    // the grid: an upper bound of the items of this launch — four times what the last split launch queued (the merge leaves the
    // count in pinned host memory), at least 4096; a launch that outgrows it leaves items pending and their instances end as LIMIT
    int grid = 4096;
    if (h->h_item_total != nullptr && 4 * *h->h_item_total > grid) grid = 4 * *h->h_item_total;
    if (grid > h->items_cap) grid = h->items_cap;
    if (h->n <= hdsm::SPLIT_N_MAX) rc = two ? launch_duo(h, b, st, grid) : (h->threads == 64 ? launch_nv<32, 64>(h, b, st, grid) : launch_nv<32, 256>(h, b, st, grid));
    else if (two) rc = launch_duo48(h, b, st, grid);
    else rc = h->threads == 64 ? launch_nv<48, 64>(h, b, st, grid) : launch_nv<48, 256>(h, b, st, grid);
    if (rc) return rc;
    hipLaunchKernelGGL(k_split_merge, dim3(a.n_inst), dim3(64), 0, st, h->N, h->P, a, b);
    HIP_TRY(hipGetLastError());
#ifdef HDSM_SPLIT_TRACE  // development aid (scripts/gpu_split_trace.sh): how the nodes of the handed-over instances spread over their items
    {
      HIP_TRY(hipStreamSynchronize(st));
      int32_t cnt[4];
      HIP_TRY(hipMemcpy(cnt, h->d_rec_count, sizeof cnt, hipMemcpyDeviceToHost));
      const size_t q = (size_t)(cnt[1] < h->items_cap ? cnt[1] : h->items_cap);
      std::vector<int32_t> nd(q), it(q);
      if (q) {
        HIP_TRY(hipMemcpy(it.data(), ss, q * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(nd.data(), ss + GI, q * 4, hipMemcpyDeviceToHost));
      }
      long long nodes_total = 0, iters_total = 0;
      int working = 0;
      std::vector<int> tops(nd.begin(), nd.end());
      for (size_t k = 0; k < q; ++k) working += nd[k] > 0, nodes_total += nd[k], iters_total += it[k];
      std::sort(tops.begin(), tops.end(), std::greater<int>());
      std::fprintf(stderr, "HDSM_SPLIT_TRACE handed over %d of %d instances (records cap %d), %d items queued, %d worked, nodes %lld iters %lld; largest items:",
                   cnt[0], a.n_inst, h->rec_cap, cnt[1], working, nodes_total, iters_total);
      for (size_t k = 0; k < tops.size() && k < 12; ++k) std::fprintf(stderr, " %d", tops[k]);
      std::fprintf(stderr, "\n");
    }
#endif
  }
  if (rc) return rc;
  // Staging overflow. The kernels that share a CU have a fraction of the staging rows of the one-per-CU kernel (384 / 768 / 320
  // against 1536 / 1024); an instance in a very dense neighbourhood can fill them with VIOLATED rows alone and then ends with
  // HDSM_FLAG_STAGING_OVERFLOW. It raises a word in pinned host memory; while the handle has seen that word in the last 256
  // launches, every launch that used such a kernel is followed by a rescue pass — the one-per-CU kernel over the batch, in
  // which only the instances that carry the flag are solved again (the host-buffer entry point adds it at once, see hdsm_replan).
  h->last_small = small, h->last_args = a;
  if (small && h->h_ovf_flag != nullptr) {
    if (*h->h_ovf_flag != 0) *h->h_ovf_flag = 0, h->rescue_ttl = 256;
    if (h->rescue_ttl > 0) {
      --h->rescue_ttl;
      rc = launch_rescue(h, a, st);
      if (rc) return rc;
    }
  }
  if (h->time_kernel) {
    HIP_TRY(hipEventRecord(h->ev_k1, st));
    h->timed = true;
  }
  if (!h->defer_done) HIP_TRY(hipEventRecord(h->ev_done, st));
  h->launched = true;
  return HDSM_OK;
}

template <class T>
hipError_t dmalloc(T** p, size_t count) {
  return hipMalloc(reinterpret_cast<void**>(p), (count ? count : 1) * sizeof(T));
}

// grow-only scratch: reallocated (after draining the handle's stream) only when a call needs more than any before
hipError_t ensure(Handle* h, Handle::Buf& b, size_t bytes) {
  if (bytes <= b.cap) return hipSuccess;
  hipError_t e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) return e;
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr, b.cap = 0;
  e = hipMalloc(&b.p, bytes);
  if (e == hipSuccess) b.cap = bytes;
  return e;
}

int64_t scratch_stride_for(int n) {
  return n <= hdsm::SPLIT_N_MAX ? (int64_t)hdsm::Solver<32, CMAX30>::SNAP_STRIDE * hdsm::MAXH
                 : (int64_t)hdsm::Solver<48, CMAX48>::SNAP_STRIDE * hdsm::MAXH;
}

// state of the split launches, allocated on the first one: the hand-over records of pass 1 with their staged rows, the item queue,
// outputs / statistics / guesses per item, snapshot scratch for the persistent workgroups of pass 2
hipError_t ensure_sub(Handle* h) {
  const size_t N = (size_t)h->N;
  const bool two = h->threads == 256 && h->duo_min > 0;
  // (resident workgroups of pass 2 per CU: two for the 256-thread shared-CU shapes, ONE for everything else — the one-per-CU kernels,
  // with 64 threads too, are held to that by LDS: their instance state is more than half of a CU's 160 KB)
  static_assert(sizeof(hdsm::Solver<32, CMAX30>::S) > 80 * 1024 && sizeof(hdsm::Solver<48, CMAX48>::S) > 80 * 1024,
                "pool_cap counts ONE resident pass-2 workgroup per CU for the one-per-CU kernels");
  h->sub_slots_n = (two ? 2 : 1) * h->cus;
  h->rows_cap = h->n <= hdsm::SPLIT_N_MAX ? (two ? CMAX_DUO : CMAX30) : (two ? CMAX_DUO48 : CMAX48);
  // (rec_cap — records: one per instance that hands its search over + one per item that hands over again — set by hdsm_create)
  h->items_cap = h->rec_cap * 16 < 4096 ? 4096 : h->rec_cap * 16;
  // snapshot scratch of pass 2: one slot per workgroup that can be resident + one per record (an item that hands over again leaves
  // its slot to its record for the rest of the launch)
  h->pool_cap = h->sub_slots_n + h->rec_cap;
  const size_t G = (size_t)h->items_cap, R = (size_t)h->rec_cap;
  hipError_t e = hipSuccess;
  auto ok = [&](hipError_t r) {
    if (e == hipSuccess) e = r;
  };
  ok(dmalloc(&h->d_recs, R)), ok(dmalloc(&h->d_rec_cand, R * h->rows_cap * 4)), ok(dmalloc(&h->d_rec_mw, R * h->rows_cap)), ok(dmalloc(&h->d_rec_src, R * h->rows_cap));
  ok(dmalloc(&h->d_rec_count, 8 + R)), ok(dmalloc(&h->d_items, G)), ok(dmalloc(&h->d_slot_busy, (size_t)h->pool_cap));
  ok(dmalloc(&h->d_split, 2 * (size_t)h->max_inst)), ok(dmalloc(&h->d_inc, (size_t)h->max_inst)), ok(dmalloc(&h->d_node_pool, (size_t)h->max_inst));
  ok(dmalloc(&h->d_sub_stats, 8 * G)), ok(dmalloc(&h->d_sub_warm, (hdsm::MAXNV + 2) * G)), ok(dmalloc(&h->d_sub_status, G));
  ok(dmalloc(&h->d_sub_traj, G * (N + 1) * 9)), ok(dmalloc(&h->d_sub_ctrl, G * N * 3)), ok(dmalloc(&h->d_sub_obj, G)), ok(dmalloc(&h->d_sub_used, G * h->P));
  ok(dmalloc(&h->d_sub_scratch, (size_t)h->pool_cap * (size_t)h->scratch_stride));
  if (e == hipSuccess) e = hipMemset(h->d_split, 0, 2 * (size_t)h->max_inst * sizeof(int32_t));
  if (e == hipSuccess) e = hipMemset(h->d_rec_count, 0, 8 * sizeof(int32_t));
  h->sub_ready = e == hipSuccess;
  if (!h->sub_ready) {  // partial failure: give back what was allocated, clear the pending error — the handle goes on without split launches
    void** sub[] = {(void**)&h->d_recs, (void**)&h->d_rec_cand, (void**)&h->d_rec_mw, (void**)&h->d_rec_src, (void**)&h->d_rec_count, (void**)&h->d_items, (void**)&h->d_slot_busy,
                    (void**)&h->d_split, (void**)&h->d_inc, (void**)&h->d_node_pool, (void**)&h->d_sub_stats,
                    (void**)&h->d_sub_warm, (void**)&h->d_sub_status, (void**)&h->d_sub_traj, (void**)&h->d_sub_ctrl, (void**)&h->d_sub_obj,
                    (void**)&h->d_sub_used, (void**)&h->d_sub_scratch};
    for (void** p : sub) {
      if (*p) (void)hipFree(*p);
      *p = nullptr;
    }
    (void)hipGetLastError();
    h->split_mode = 0;
  }
  return e;
}

void free_all(Handle* h) {
  void* sub[] = {h->d_recs, h->d_rec_cand, h->d_rec_mw, h->d_rec_src, h->d_rec_count, h->d_items, h->d_slot_busy, h->d_split, h->d_inc, h->d_node_pool, h->d_sub_stats, h->d_sub_warm,
                 h->d_sub_status, h->d_sub_traj, h->d_sub_ctrl, h->d_sub_obj, h->d_sub_used, h->d_sub_scratch};
  for (void* p : sub)
    if (p) (void)hipFree(p);
  if (h->h_tree_flag) (void)hipHostFree(h->h_tree_flag);
  if (h->h_item_total) (void)hipHostFree(h->h_item_total);
  if (h->h_ovf_flag) (void)hipHostFree(h->h_ovf_flag);
  if (h->h_out) (void)hipHostFree(h->h_out);
  void* ptrs[] = {h->d_warm, h->d_prof, h->d_consts, h->d_scratch, h->d_stats, h->d_agent, h->d_npoly, h->d_nrows, h->d_status,
                  h->d_state,  h->d_ref,     h->d_A,     h->d_b,     h->d_plans, h->d_bounds, h->d_setup, h->d_traj,  h->d_ctrl,
                  h->d_obj,    h->d_has,     h->d_used,  h->d_pos,   h->d_rpos,  h->d_rsph,  h->d_zero,  h->d_order, h->b_planes.p, h->b_common.p,
                  h->b_ncommon.p, h->b_path.p, h->b_cap.p, h->b_full.p, h->b_pv.p, h->b_np.p};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (h->ev_done) (void)hipEventDestroy(h->ev_done);
  if (h->ev_k0) (void)hipEventDestroy(h->ev_k0);
  if (h->ev_k1) (void)hipEventDestroy(h->ev_k1);
  if (h->stream) (void)hipStreamDestroy(h->stream);
}

int check_common(Handle* h, int n_inst, int n_rob) {
  if (!h) return set_err(HDSM_ERR_BAD_ARG, "null handle");
  if (n_inst < 0 || n_rob < 0) return set_err(HDSM_ERR_BAD_ARG, "negative size");
  if (n_inst > h->max_inst) return set_err(HDSM_ERR_CAPACITY, "n_inst exceeds max_instances of the handle");
  if (n_rob > h->n_rob_max) return set_err(HDSM_ERR_CAPACITY, "n_rob exceeds n_rob_max of the handle");
  return HDSM_OK;
}

}  // namespace

extern "C" {

int32_t hdsm_version(void) { return (1 << 16) | 3; }  // 1.2: + hdsm_poly_octa3d_batch_wave / _device_wave, hdsm_set_kernel_timing / hdsm_last_kernel_ms; 1.3: + hdsm_host_register / _unregister

const char* hdsm_last_error(void) { return g_err.c_str(); }

void hdsm_default_params(hdsm_params* p, int32_t n_hor) {
  if (!p) return;
  std::memset(p, 0, sizeof *p);
  // multi_agent_planner/config/agent_agile_config.yaml -> InitializePlannerParameters (AC:2169-2188)
  p->n_hor = n_hor, p->poly_hor = 4, p->rk4 = 0, p->max_rows_static = 18;
  p->dt = 0.1, p->r_u = 0.01;
  const double w[9] = {100, 100, 100, 1, 1, 1, 0, 0, 0};
  for (int k = 0; k < 9; ++k) p->r_x[k] = p->r_n[k] = w[k];
  const double max_vel = 20.0, max_acc = 15.0, max_jerk = 60.0;
  for (int k = 0; k < 3; ++k) {
    p->x_lb[k] = -HDSM_INF, p->x_ub[k] = HDSM_INF;
    p->x_lb[3 + k] = -max_vel, p->x_ub[3 + k] = max_vel;
    p->x_lb[6 + k] = -max_acc, p->x_ub[6 + k] = max_acc;
    p->u_lb[k] = -max_jerk, p->u_ub[k] = max_jerk;
  }
  p->drone_radius = 0.25, p->drone_z_offset = 0.25, p->plane_perturb = 0.1;
  p->max_nodes = 0, p->max_qp_iters = 0, p->feas_tol_fixed = 1e-6, p->solver_tol = 1e-9;
  p->warm_start = 1;  // execution knobs and time_limit_s stay 0 = library defaults / no wall-clock limit
}

int hdsm_create(const hdsm_params* params, int32_t max_instances, int32_t n_rob_max, int32_t device,
                void** handle) {
  if (!params || !handle) return set_err(HDSM_ERR_BAD_ARG, "null argument");
  if (max_instances < 1 || n_rob_max < 1) return set_err(HDSM_ERR_BAD_ARG, "max_instances/n_rob_max must be >= 1");
  *handle = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return set_err(HDSM_ERR_NO_DEVICE, "no HIP device: this library has no CPU fallback");
  if (device < 0 || device >= ndev) return set_err(HDSM_ERR_NO_DEVICE, "device ordinal out of range");
  HIP_TRY(hipSetDevice(device));

  hdsm::Consts* hc = new (std::nothrow) hdsm::Consts;
  if (!hc) return set_err(HDSM_ERR_DEVICE, "out of host memory");
  const char* msg = nullptr;
  int rc = hdsm::build_consts(params, hc, &msg);
  if (rc) {
    delete hc;
    return set_err(rc, msg ? msg : "bad params");
  }
  Handle* h = new (std::nothrow) Handle;
  if (!h) {
    delete hc;
    return set_err(HDSM_ERR_DEVICE, "out of host memory");
  }
  h->device = device, h->max_inst = max_instances, h->n_rob_max = n_rob_max, h->prm = *params;
  h->N = hc->N, h->P = hc->P, h->RS = hc->RS, h->n = hc->n;
  // execution knobs: hdsm_params, then the environment (scripts); out-of-range values are ignored
  auto env_int = [](const char* name, long lo, long hi, int* out) {
    if (const char* e = std::getenv(name)) {
      char* end = nullptr;
      const long v = std::strtol(e, &end, 10);
      if (end != e && *end == 0 && v >= lo && v <= hi) *out = (int)v;
    }
  };
  if (params->threads_per_instance == 64 || params->threads_per_instance == 256) h->threads = params->threads_per_instance;
  {
    int t = h->threads;
    env_int("HDSM_THREADS", 64, 256, &t);
    if (t == 64 || t == 256) h->threads = t;
  }
  h->bounds_min = params->prefilter_min_agents > 0 ? params->prefilter_min_agents : (params->prefilter_min_agents < 0 ? INT_MAX : 256);
  env_int("HDSM_BOUNDS_MIN", 1, INT_MAX, &h->bounds_min);
  {
    hipDeviceProp_t prop;
    const int cus = (hipGetDeviceProperties(&prop, device) == hipSuccess) ? prop.multiProcessorCount : 256;
    h->cus = cus;
    h->duo_min = params->duo_min_instances > 0 ? params->duo_min_instances : (params->duo_min_instances < 0 ? 0 : cus + 1);
    env_int("HDSM_DUO_MIN", 0, INT_MAX, &h->duo_min);  // 0 = never
    h->tri_min = h->duo_min > 0 ? 2 * cus + 1 : 0;     // more instances than the two-per-CU kernel has resident slots
    env_int("HDSM_TRI_MIN", 0, INT_MAX, &h->tri_min);  // 0 = never
    h->quad_min = h->tri_min > 0 ? 3 * cus + 1 : 0;    // more instances than the three-per-CU kernel has resident slots
    env_int("HDSM_QUAD_MIN", 0, INT_MAX, &h->quad_min);  // 0 = never
    env_int("HDSM_SETUP_MFMA", 0, 1, &h->setup_mfma);
    env_int("HDSM_SPLIT", 0, 2, &h->split_mode);       // subtree splitting: 0 never, 1 always, 2 (default) when the last launch met a deep tree
    env_int("HDSM_SPLIT_BUDGET", 1, 100000, &h->split_budget);  // (unset: by batch size, see launch())
    env_int("HDSM_ITEM_BUDGET", 0, 100000, &h->item_budget);    // pass 2: nodes after which an item hands over again (32; 0 = never)
    env_int("HDSM_ITEM_MIN", 1, 100000, &h->item_min);          // ... or, while workgroups wait for items, this many (unset: by batch size)
    env_int("HDSM_POLL_SLEEP", 1, 1000, &h->poll_sleep);
    h->rec_cap = 8 * max_instances < 256 ? 256 : (8 * max_instances > 2048 ? 2048 : 8 * max_instances);
    env_int("HDSM_SPLIT_RECORDS", 1, 1 << 20, &h->rec_cap);
    // more instances than can be resident at once (two workgroups per CU): launch the expensive ones first
    h->order_min = params->launch_order == 0 ? 2 * cus + 1 : (params->launch_order < 0 ? 0 : params->launch_order);
    env_int("HDSM_ORDER_MIN", 0, INT_MAX, &h->order_min);  // 0 = never
  }
  if (params->time_limit_s > 0) {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) != hipSuccess || khz <= 0) khz = 100000;
    hc->time_ticks = (long long)(params->time_limit_s * 1e3 * (double)khz);
    if (hc->time_ticks < 1) hc->time_ticks = 1;
  }
  h->scratch_stride = scratch_stride_for(h->n);
  const size_t I = (size_t)max_instances, N = (size_t)h->N, P = (size_t)h->P, RS = (size_t)h->RS;
  hipError_t e = hipSuccess;
  auto ok = [&](hipError_t r) {
    if (e == hipSuccess) e = r;
  };
  ok(dmalloc(&h->d_consts, 1));
  ok(dmalloc(&h->d_scratch, I * (size_t)h->scratch_stride));
  ok(dmalloc(&h->d_stats, 8 * I));
  ok(dmalloc(&h->d_warm, (hdsm::MAXNV + 2) * I));
#if defined(HDSM_PROFILE) || defined(HDSM_TIMELINE)
  ok(dmalloc(&h->d_prof, 32 * I));
#endif
  ok(dmalloc(&h->d_agent, I));
  ok(dmalloc(&h->d_npoly, I));
  ok(dmalloc(&h->d_nrows, I * P));
  ok(dmalloc(&h->d_status, I));
  ok(dmalloc(&h->d_state, I * 9));
  ok(dmalloc(&h->d_ref, I * N * 6));
  ok(dmalloc(&h->d_A, I * P * RS * 3));
  ok(dmalloc(&h->d_b, I * P * RS));
  ok(dmalloc(&h->d_plans, (size_t)n_rob_max * (N + 1) * 9));
  ok(dmalloc(&h->d_bounds, (size_t)n_rob_max * 4));
  ok(dmalloc(&h->d_setup, I * hdsm::KROWS));
  ok(dmalloc(&h->d_pos, (size_t)n_rob_max * N * 3));
  ok(dmalloc(&h->d_rpos, (size_t)n_rob_max * (N + 1) * 3));
  ok(dmalloc(&h->d_rsph, (size_t)n_rob_max * 4));
  ok(dmalloc(&h->d_zero, (size_t)n_rob_max));
  ok(dmalloc(&h->d_order, 2 * I));
  ok(dmalloc(&h->d_traj, I * (N + 1) * 9));
  ok(dmalloc(&h->d_ctrl, I * N * 3));
  ok(dmalloc(&h->d_obj, I));
  ok(dmalloc(&h->d_has, (size_t)n_rob_max));
  ok(dmalloc(&h->d_used, I * P));
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&h->h_ovf_flag), sizeof(int32_t), hipHostMallocMapped);
  if (e == hipSuccess) {
    *h->h_ovf_flag = 0;
    e = hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_ovf_flag), h->h_ovf_flag, 0);
  }
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&h->h_tree_flag), sizeof(int32_t), hipHostMallocMapped);
  if (e == hipSuccess) {
    *h->h_tree_flag = 0;
    e = hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_tree_flag), h->h_tree_flag, 0);
  }
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&h->h_item_total), sizeof(int32_t), hipHostMallocMapped);
  if (e == hipSuccess) {
    *h->h_item_total = 0;
    e = hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_item_total), h->h_item_total, 0);
  }
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMemcpy(h->d_consts, hc, sizeof *hc, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreate(&h->ev_k0);
  if (e == hipSuccess) e = hipEventCreate(&h->ev_k1);
  if (e == hipSuccess) e = hipMemset(h->d_stats, 0, 8 * I * sizeof(int32_t));
  if (e == hipSuccess) e = hipMemset(h->d_zero, 0, (size_t)n_rob_max);
  // (the fetch kernel of hdsm_replan uploads only the rows of the static polyhedra that exist: the rest stays finite)
  if (e == hipSuccess) e = hipMemset(h->d_A, 0, I * P * RS * 3 * sizeof(double));
  if (e == hipSuccess) e = hipMemset(h->d_b, 0, I * P * RS * sizeof(double));
  if (e == hipSuccess) e = hipMemset(h->d_plans, 0, (size_t)n_rob_max * (N + 1) * 9 * sizeof(double));
  if (e == hipSuccess) e = hipMemset(h->d_warm, 0, (hdsm::MAXNV + 2) * I * sizeof(int32_t));
  delete hc;
  if (e != hipSuccess) {
    free_all(h);
    delete h;
    return set_err(HDSM_ERR_DEVICE, std::string("hdsm_create: ") + hipGetErrorString(e));
  }
  *handle = h;
  return HDSM_OK;
}

void hdsm_destroy(void* handle) {
  Handle* h = static_cast<Handle*>(handle);
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  free_all(h);
  delete h;
}

int hdsm_replan_device(void* handle, int32_t n_inst, int32_t n_rob, const int32_t* agent_id,
                       const double* state_curr, const double* traj_ref, const int32_t* n_poly,
                       const int32_t* n_rows_static, const double* A_static, const double* b_static,
                       const double* plans_all, const uint8_t* has_plan, double* traj_out,
                       double* ctrl_out, uint8_t* poly_used, int32_t* status, double* obj,
                       void* hip_stream) {
  Handle* h = static_cast<Handle*>(handle);
  if (int rc = check_common(h, n_inst, n_rob)) return rc;
  if (n_inst == 0) return HDSM_OK;
  if (!agent_id || !state_curr || !traj_ref || !n_poly || !n_rows_static || !A_static || !b_static ||
      !plans_all || !has_plan || !traj_out || !ctrl_out || !poly_used || !status || !obj)
    return set_err(HDSM_ERR_BAD_ARG, "null array argument");
  if (traj_out == plans_all) return set_err(HDSM_ERR_BAD_ARG, "traj_out must not alias plans_all");
  HIP_TRY(hipSetDevice(h->device));
  hdsm::Args a{};
  a.n_inst = n_inst, a.n_rob = n_rob, a.agent_id = agent_id, a.state = state_curr, a.ref = traj_ref;
  a.n_poly = n_poly, a.n_rows = n_rows_static, a.A = A_static, a.b = b_static, a.plans = plans_all;
  a.has_plan = has_plan, a.traj = traj_out, a.ctrl = ctrl_out, a.used = poly_used, a.status = status;
  a.obj = obj;
  return launch(h, a, static_cast<hipStream_t>(hip_stream));
}

// Registered (page-locked, mapped) host memory. hdsm_host_register records every range it maps — base, length, device-side
// address — and hdsm_replan takes the kernel paths (k_fetch / k_deliver) only for arrays that lie INSIDE a recorded range with all
// the bytes the call will touch (an array that merely starts in one, or memory page-locked by somebody else, goes through the copy
// path like pageable memory). While nothing is registered the look-up is one load: no runtime query per array and call.
namespace {
struct HostRange {
  char* base;
  size_t bytes;
  char* dev;
};
std::mutex g_reg_mutex;
std::vector<HostRange> g_reg;
std::atomic<int> g_reg_count{0};

bool mapped_host_range(const void* p, size_t bytes, void** dev) {
  if (g_reg_count.load(std::memory_order_acquire) == 0) return false;
  std::lock_guard<std::mutex> lock(g_reg_mutex);
  const char* c = static_cast<const char*>(p);
  for (const HostRange& r : g_reg)
    if (c >= r.base && bytes <= r.bytes && (size_t)(c - r.base) <= r.bytes - bytes) {
      *dev = r.dev + (c - r.base);
      return true;
    }
  return false;
}
}  // namespace

int hdsm_host_register(void* ptr, size_t bytes) {
  if (!ptr || bytes == 0) return set_err(HDSM_ERR_BAD_ARG, "hdsm_host_register: null or empty range");
  hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return set_err(HDSM_ERR_DEVICE, std::string("hdsm_host_register: ") + hipGetErrorString(e));
  }
  void* dev = nullptr;
  e = hipHostGetDevicePointer(&dev, ptr, 0);
  if (e != hipSuccess || dev == nullptr) {
    (void)hipGetLastError();
    (void)hipHostUnregister(ptr);
    return set_err(HDSM_ERR_DEVICE, std::string("hdsm_host_register: no device address for the range: ") + hipGetErrorString(e));
  }
  std::lock_guard<std::mutex> lock(g_reg_mutex);
  g_reg.push_back(HostRange{static_cast<char*>(ptr), bytes, static_cast<char*>(dev)});
  g_reg_count.store((int)g_reg.size(), std::memory_order_release);
  return HDSM_OK;
}

int hdsm_host_unregister(void* ptr) {
  if (!ptr) return set_err(HDSM_ERR_BAD_ARG, "hdsm_host_unregister: null pointer");
  {
    std::lock_guard<std::mutex> lock(g_reg_mutex);
    for (size_t k = 0; k < g_reg.size(); ++k)
      if (g_reg[k].base == static_cast<char*>(ptr)) {
        g_reg.erase(g_reg.begin() + (long)k);
        break;
      }
    g_reg_count.store((int)g_reg.size(), std::memory_order_release);
  }
  const hipError_t e = hipHostUnregister(ptr);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return set_err(HDSM_ERR_DEVICE, std::string("hdsm_host_unregister: ") + hipGetErrorString(e));
  }
  return HDSM_OK;
}

int hdsm_replan(void* handle, int32_t n_inst, int32_t n_rob, const int32_t* agent_id,
                const double* state_curr, const double* traj_ref, const int32_t* n_poly,
                const int32_t* n_rows_static, const double* A_static, const double* b_static,
                const double* plans_all, const uint8_t* has_plan, double* traj_out, double* ctrl_out,
                uint8_t* poly_used, int32_t* status, double* obj) {
  Handle* h = static_cast<Handle*>(handle);
  if (int rc = check_common(h, n_inst, n_rob)) return rc;
  if (n_inst == 0) return HDSM_OK;
  if (!agent_id || !state_curr || !traj_ref || !n_poly || !n_rows_static || !A_static || !b_static ||
      !plans_all || !has_plan || !traj_out || !ctrl_out || !poly_used || !status || !obj)
    return set_err(HDSM_ERR_BAD_ARG, "null array argument");
  HIP_TRY(hipSetDevice(h->device));
  const size_t I = (size_t)n_inst, N = (size_t)h->N, P = (size_t)h->P, RS = (size_t)h->RS;
  hipStream_t st = h->stream;
  const auto H2D = hipMemcpyHostToDevice;
  const auto D2H = hipMemcpyDeviceToHost;
  if (h->launched && st != h->last_stream) HIP_TRY(hipStreamWaitEvent(st, h->ev_done, 0));
  bool in_mapped = true;
  {
    const void* hp[9] = {agent_id, state_curr, traj_ref, n_poly, n_rows_static, A_static, b_static, plans_all, has_plan};
    const size_t hb[9] = {I * 4, I * 9 * 8, I * N * 6 * 8, I * 4, I * P * 4, I * P * RS * 3 * 8, I * P * RS * 8, (size_t)n_rob * (N + 1) * 9 * 8, (size_t)n_rob};
    void* dp[9];
    for (int k = 0; k < 9 && in_mapped; ++k) in_mapped = mapped_host_range(hp[k], hb[k], &dp[k]);
    if (in_mapped) {  // page-locked arrays of the caller (hdsm_host_register): one fetch kernel
      FetchArgs f{};
      f.n_inst = n_inst, f.n_rob = n_rob, f.N = (int)N, f.P = (int)P, f.RS = (int)RS;
      const int64_t plan_items = (int64_t)n_rob * (int64_t)(N + 1) * 9;
      f.plan_blocks = (int)((plan_items + 1023) / 1024 < 1 ? 1 : ((plan_items + 1023) / 1024 > 1024 ? 1024 : (plan_items + 1023) / 1024));
      f.agent_id = static_cast<const int32_t*>(dp[0]), f.state = static_cast<const double*>(dp[1]), f.ref = static_cast<const double*>(dp[2]);
      f.n_poly = static_cast<const int32_t*>(dp[3]), f.n_rows = static_cast<const int32_t*>(dp[4]), f.A = static_cast<const double*>(dp[5]);
      f.b = static_cast<const double*>(dp[6]), f.plans = static_cast<const double*>(dp[7]), f.has = static_cast<const uint8_t*>(dp[8]);
      f.d_agent = h->d_agent, f.d_state = h->d_state, f.d_ref = h->d_ref, f.d_npoly = h->d_npoly, f.d_nrows = h->d_nrows;
      f.d_A = h->d_A, f.d_b = h->d_b, f.d_plans = h->d_plans, f.d_has = h->d_has;
      hipLaunchKernelGGL(k_fetch, dim3((unsigned)(n_inst + f.plan_blocks)), dim3(256), 0, st, f);
      HIP_TRY(hipGetLastError());
    }
  }
  if (!in_mapped) {
    HIP_TRY(hipMemcpyAsync(h->d_agent, agent_id, I * 4, H2D, st));
    HIP_TRY(hipMemcpyAsync(h->d_state, state_curr, I * 9 * 8, H2D, st));
    HIP_TRY(hipMemcpyAsync(h->d_ref, traj_ref, I * N * 6 * 8, H2D, st));
    HIP_TRY(hipMemcpyAsync(h->d_npoly, n_poly, I * 4, H2D, st));
    HIP_TRY(hipMemcpyAsync(h->d_nrows, n_rows_static, I * P * 4, H2D, st));
    HIP_TRY(hipMemcpyAsync(h->d_A, A_static, I * P * RS * 3 * 8, H2D, st));
    HIP_TRY(hipMemcpyAsync(h->d_b, b_static, I * P * RS * 8, H2D, st));
    HIP_TRY(hipMemcpyAsync(h->d_plans, plans_all, (size_t)n_rob * (N + 1) * 9 * 8, H2D, st));
    HIP_TRY(hipMemcpyAsync(h->d_has, has_plan, (size_t)n_rob, H2D, st));
  }
  int rc = hdsm_replan_device(handle, n_inst, n_rob, h->d_agent, h->d_state, h->d_ref, h->d_npoly, h->d_nrows,
                              h->d_A, h->d_b, h->d_plans, h->d_has, h->d_traj, h->d_ctrl, h->d_used,
                              h->d_status, h->d_obj, st);
  if (rc) return rc;
  if (h->last_small && h->h_ovf_flag != nullptr && h->rescue_ttl == 0) {  // (a launch that already carried the rescue pass needs no second one)
    HIP_TRY(hipStreamSynchronize(st));
    if (*h->h_ovf_flag != 0) {  // an instance ran out of staging rows in a shared-CU kernel: solve it again now, with the large area
      *h->h_ovf_flag = 0, h->rescue_ttl = 256;
      rc = launch_rescue(h, h->last_args, st);
      if (rc) return rc;
    }
  }
  // Outputs are "left untouched" for instances without a solution. The caller's arrays are not uploaded to seed the device
  // copies (a megabyte each way per 1024 agents): the results come back into a pinned staging block of the handle and only
  // the instances that HAVE a solution are copied into the caller's arrays.
  const size_t trj = (N + 1) * 9, ctl = N * 3;
  void* dp[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  {  // page-locked output arrays (hdsm_host_register, every array inside a registered range): delivered by the device, filtered there
    void* hp[5] = {traj_out, ctrl_out, obj, status, poly_used};
    const size_t hb[5] = {I * trj * 8, I * ctl * 8, I * 8, I * 4, I * P};
    bool mapped = true;
    for (int k = 0; k < 5 && mapped; ++k) mapped = mapped_host_range(hp[k], hb[k], &dp[k]);
    if (!mapped) dp[0] = nullptr;
  }
  if (dp[0] != nullptr) {
    hipLaunchKernelGGL(k_deliver, dim3((unsigned)((I + 3) / 4)), dim3(256), 0, st, n_inst, (int)trj, (int)ctl, (int)P, h->d_traj, h->d_ctrl, h->d_obj,
                       h->d_status, h->d_used, static_cast<double*>(dp[0]), static_cast<double*>(dp[1]), static_cast<double*>(dp[2]),
                       static_cast<int32_t*>(dp[3]), static_cast<uint8_t*>(dp[4]));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
  } else {
    const size_t need = I * (trj * 8 + ctl * 8 + 8 + 4 + P);
    if (need > h->h_out_cap) {
      if (h->h_out) (void)hipHostFree(h->h_out);
      h->h_out = nullptr, h->h_out_cap = 0;
      HIP_TRY(hipHostMalloc(&h->h_out, need, hipHostMallocDefault));
      h->h_out_cap = need;
    }
    double* o_traj = static_cast<double*>(h->h_out);
    double* o_ctrl = o_traj + I * trj;
    double* o_obj = o_ctrl + I * ctl;
    int32_t* o_status = reinterpret_cast<int32_t*>(o_obj + I);
    uint8_t* o_used = reinterpret_cast<uint8_t*>(o_status + I);
    HIP_TRY(hipMemcpyAsync(o_traj, h->d_traj, I * trj * 8, D2H, st));
    HIP_TRY(hipMemcpyAsync(o_ctrl, h->d_ctrl, I * ctl * 8, D2H, st));
    HIP_TRY(hipMemcpyAsync(o_obj, h->d_obj, I * 8, D2H, st));
    HIP_TRY(hipMemcpyAsync(o_status, h->d_status, I * 4, D2H, st));
    HIP_TRY(hipMemcpyAsync(o_used, h->d_used, I * P, D2H, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (size_t k = 0; k < I; ++k) {
      status[k] = o_status[k];
      if (o_status[k] == HDSM_NO_SOLUTION) continue;
      std::memcpy(traj_out + k * trj, o_traj + k * trj, trj * 8);
      std::memcpy(ctrl_out + k * ctl, o_ctrl + k * ctl, ctl * 8);
      std::memcpy(poly_used + k * P, o_used + k * P, P);
      obj[k] = o_obj[k];
    }
  }
  // (one exit for both delivery paths: anything added after the download applies to registered callers too)
  return HDSM_OK;
}

int hdsm_tasc_planes(void* handle, int32_t n_inst, int32_t n_rob, const int32_t* agent_id,
                     const double* state_curr, const double* plans_all, const uint8_t* has_plan,
                     double* planes) {
  Handle* h = static_cast<Handle*>(handle);
  if (int rc = check_common(h, n_inst, n_rob)) return rc;
  if (n_inst == 0 || n_rob == 0) return HDSM_OK;
  if (!agent_id || !state_curr || !plans_all || !has_plan || !planes)
    return set_err(HDSM_ERR_BAD_ARG, "null array argument");
  HIP_TRY(hipSetDevice(h->device));
  const size_t I = (size_t)n_inst, N = (size_t)h->N;
  const size_t total = I * N * (size_t)n_rob * 4;
  HIP_TRY(ensure(h, h->b_planes, total * sizeof(double)));
  double* d_planes = static_cast<double*>(h->b_planes.p);
  hipStream_t st = h->stream;
  if (h->launched && st != h->last_stream) HIP_TRY(hipStreamWaitEvent(st, h->ev_done, 0));  // staging buffers are shared
  hipError_t e = hipMemcpyAsync(h->d_agent, agent_id, I * 4, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(h->d_state, state_curr, I * 9 * 8, hipMemcpyHostToDevice, st);
  if (e == hipSuccess)
    e = hipMemcpyAsync(h->d_plans, plans_all, (size_t)n_rob * (N + 1) * 9 * 8, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(h->d_has, has_plan, (size_t)n_rob, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    const int64_t items = (int64_t)(total / 4);
    const int blocks = (int)((items + 255) / 256 > 4096 ? 4096 : (items + 255) / 256);
    hipLaunchKernelGGL(k_tasc_planes, dim3(blocks), dim3(256), 0, st, h->d_consts, n_inst, n_rob, h->d_agent,
                       h->d_state, h->d_plans, h->d_has, d_planes);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(planes, d_planes, total * 8, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return set_err(HDSM_ERR_DEVICE, std::string("hdsm_tasc_planes: ") + hipGetErrorString(e));
  return HDSM_OK;
}

int hdsm_reference_device(void* handle, const hdsm_ref_config* cfg, int32_t n_inst, int32_t n_rob,
                          const int32_t* agent_id, const double* path, const int32_t* n_path, int32_t pmax,
                          const double* vel_cap, const double* plans_all, const uint8_t* has_plan,
                          double* ref_full, double* ref, double* path_vel, void* hip_stream) {
  Handle* h = static_cast<Handle*>(handle);
  if (int rc = check_common(h, n_inst, n_rob)) return rc;
  if (n_inst == 0) return HDSM_OK;
  if (!cfg || !agent_id || !path || !n_path || !plans_all || !has_plan || !ref_full || !path_vel || pmax < 1)
    return set_err(HDSM_ERR_BAD_ARG, "null or empty argument");
  HIP_TRY(hipSetDevice(h->device));
  RefArgs a{};
  a.n_inst = n_inst, a.n_rob = n_rob, a.pmax = pmax, a.N = h->N, a.dt = h->prm.dt, a.cfg = *cfg;
  a.agent_id = agent_id, a.path = path, a.n_path = n_path, a.vel_cap = vel_cap, a.plans = plans_all;
  a.has_plan = has_plan, a.ref_full = ref_full, a.ref = ref, a.path_vel = path_vel;
  a.rpos = h->d_rpos, a.rsph = h->d_rsph;
  for (int i = 0; i <= hdsm::MAXH; ++i) {
    double occ = 100 * std::pow(cfg->sens_other_agents, (double)i);  // AC:1791-1795
    occ = occ < 0 ? 0 : (occ > 100 ? 100 : occ);
    a.wocc[i] = std::pow(occ / 100, cfg->sens_pot);  // GetVelocityLimit AC:1805-1817
  }
  // d_rpos / d_rsph are scratch of the HANDLE: a call that arrives on another stream than the previous launch waits for it
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  if (h->launched && st != h->last_stream) HIP_TRY(hipStreamWaitEvent(st, h->ev_done, 0));
  h->last_stream = st;
  {
    // device-resident loop (defer_done: one stream, the solve of this round follows on the same plans): the pre-pass rides along
    const bool with_pre = h->defer_done;
    const bool ordered = with_pre && h->prm.warm_start && h->order_min > 0 && n_inst >= h->order_min;
    const bool pre = n_rob >= h->bounds_min;
    hipLaunchKernelGGL(k_ref_pack, dim3((n_rob + 15) / 16 + (ordered ? 1 : 0)), dim3(256), 0, st, h->N, n_rob, plans_all, has_plan, h->d_rpos, h->d_rsph,
                       with_pre ? h->d_pos : nullptr, with_pre && pre ? h->d_bounds : nullptr, n_inst, h->d_stats + 7 * h->max_inst, agent_id,
                       ordered ? h->d_order : nullptr);
    h->prepass_for = with_pre ? plans_all : nullptr, h->prepass_n_rob = n_rob, h->prepass_n_inst = n_inst, h->prepass_ordered = ordered;
  }
  HIP_TRY(hipGetLastError());
  if (n_inst >= 256) hipLaunchKernelGGL(k_reference<64>, dim3(n_inst), dim3(64), 0, st, a);
  else hipLaunchKernelGGL(k_reference<256>, dim3(n_inst), dim3(256), 0, st, a);
  HIP_TRY(hipGetLastError());
  if (!h->defer_done) HIP_TRY(hipEventRecord(h->ev_done, st));
  h->launched = true;
  return HDSM_OK;
}

int hdsm_reference(void* handle, const hdsm_ref_config* cfg, int32_t n_inst, int32_t n_rob, const int32_t* agent_id,
                   const double* path, const int32_t* n_path, int32_t pmax, const double* vel_cap,
                   const double* plans_all, const uint8_t* has_plan, double* ref_full, double* ref, double* path_vel) {
  Handle* h = static_cast<Handle*>(handle);
  if (int rc = check_common(h, n_inst, n_rob)) return rc;
  if (n_inst == 0) return HDSM_OK;
  if (!cfg || !agent_id || !path || !n_path || !plans_all || !has_plan || !ref_full || !path_vel || pmax < 1)
    return set_err(HDSM_ERR_BAD_ARG, "null or empty argument");
  for (int32_t k = 0; k < n_inst; ++k)
    if (n_path[k] < 1 || n_path[k] > pmax) return set_err(HDSM_ERR_BAD_ARG, "n_path[k] must be in [1, pmax]");
  HIP_TRY(hipSetDevice(h->device));
  const size_t I = (size_t)n_inst, N = (size_t)h->N;
  hipStream_t st = h->stream;
  hipError_t e = ensure(h, h->b_path, I * pmax * 3 * sizeof(double));
  if (e == hipSuccess) e = ensure(h, h->b_cap, I * sizeof(double));
  if (e == hipSuccess) e = ensure(h, h->b_full, I * (N + 1) * 6 * sizeof(double));
  if (e == hipSuccess) e = ensure(h, h->b_pv, I * sizeof(double));
  if (e == hipSuccess) e = ensure(h, h->b_np, I * sizeof(int32_t));
  if (e != hipSuccess) return set_err(HDSM_ERR_DEVICE, std::string("hdsm_reference: ") + hipGetErrorString(e));
  double *d_path = static_cast<double*>(h->b_path.p), *d_cap = static_cast<double*>(h->b_cap.p),
         *d_full = static_cast<double*>(h->b_full.p), *d_pv = static_cast<double*>(h->b_pv.p);
  int32_t* d_np = static_cast<int32_t*>(h->b_np.p);
  if (h->launched && st != h->last_stream) HIP_TRY(hipStreamWaitEvent(st, h->ev_done, 0));
  auto cp = [&](void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    if (e == hipSuccess && bytes) e = hipMemcpyAsync(dst, src, bytes, kind, st);
  };
  cp(d_path, path, I * pmax * 3 * 8, hipMemcpyHostToDevice);
  cp(d_np, n_path, I * 4, hipMemcpyHostToDevice);
  if (vel_cap) cp(d_cap, vel_cap, I * 8, hipMemcpyHostToDevice);
  cp(h->d_agent, agent_id, I * 4, hipMemcpyHostToDevice);
  cp(h->d_plans, plans_all, (size_t)n_rob * (N + 1) * 9 * 8, hipMemcpyHostToDevice);
  cp(h->d_has, has_plan, (size_t)n_rob, hipMemcpyHostToDevice);
  int rc = HDSM_OK;
  if (e == hipSuccess)
    rc = hdsm_reference_device(handle, cfg, n_inst, n_rob, h->d_agent, d_path, d_np, pmax, vel_cap ? d_cap : nullptr,
                               h->d_plans, h->d_has, d_full, ref ? h->d_ref : nullptr, d_pv, st);
  cp(ref_full, d_full, I * (N + 1) * 6 * 8, hipMemcpyDeviceToHost);
  if (ref) cp(ref, h->d_ref, I * N * 6 * 8, hipMemcpyDeviceToHost);
  cp(path_vel, d_pv, I * 8, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (rc) return rc;
  if (e != hipSuccess) return set_err(HDSM_ERR_DEVICE, std::string("hdsm_reference: ") + hipGetErrorString(e));
  return HDSM_OK;
}

int hdsm_reset_warm_start(void* handle) {
  Handle* h = static_cast<Handle*>(handle);
  if (!h) return set_err(HDSM_ERR_BAD_ARG, "null handle");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemset(h->d_warm, 0, (size_t)(hdsm::MAXNV + 2) * h->max_inst * sizeof(int32_t)));
  return HDSM_OK;
}

// Internal (csrc/swarm_kernels.hip; not in include/): every device entry point records ev_done so that a later call on ANOTHER
// stream can wait for the handle's scratch. Each record is a barrier packet — 5-6 us of idle queue in front of the next kernel.
// The device-resident loop issues its whole round on one stream, so it defers the records and leaves one at the end of the round.
extern "C" int hdsm_internal_defer_done(void* handle, int on) {
  Handle* h = static_cast<Handle*>(handle);
  if (!h) return HDSM_ERR_BAD_ARG;
  h->defer_done = on != 0;
  return HDSM_OK;
}
extern "C" int hdsm_internal_record_done(void* handle, void* hip_stream) {
  Handle* h = static_cast<Handle*>(handle);
  if (!h) return HDSM_ERR_BAD_ARG;
  HIP_TRY(hipEventRecord(h->ev_done, static_cast<hipStream_t>(hip_stream)));
  return HDSM_OK;
}

int hdsm_set_kernel_timing(void* handle, int32_t on) {
  Handle* h = static_cast<Handle*>(handle);
  if (!h) return set_err(HDSM_ERR_BAD_ARG, "null handle");
  h->time_kernel = on != 0;
  if (!h->time_kernel) h->timed = false;
  return HDSM_OK;
}

int hdsm_last_kernel_ms(void* handle, float* ms) {
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !ms) return set_err(HDSM_ERR_BAD_ARG, "null argument");
  if (!h->timed) return set_err(HDSM_ERR_BAD_ARG, "no launch since hdsm_set_kernel_timing(handle, 1)");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipEventSynchronize(h->ev_k1));
  HIP_TRY(hipEventElapsedTime(ms, h->ev_k0, h->ev_k1));
  return HDSM_OK;
}

int hdsm_last_stats(void* handle, int32_t n_inst, int32_t* qp_iters, int32_t* nodes, int32_t* sweeps,
                    int32_t* cand) {
  Handle* h = static_cast<Handle*>(handle);
  if (int rc = check_common(h, n_inst, 0)) return rc;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->last_stream));
#ifdef HDSM_TIMELINE
  {  // development aid: the launch seen from the instances (100 MHz clock): span, the slowest instance, late starters
    std::vector<long long> pr((size_t)n_inst * 32);
    HIP_TRY(hipMemcpy(pr.data(), h->d_prof, pr.size() * sizeof(long long), hipMemcpyDeviceToHost));
    long long t0 = pr[0], t1 = pr[1];
    int worst = 0, last = 0;
    double busy = 0;
    for (int k = 0; k < n_inst; ++k) {
      const long long b = pr[(size_t)k * 32], e = pr[(size_t)k * 32 + 1];
      if (b < t0) t0 = b;
      if (e > t1) t1 = e, last = k;
      if (e - b > pr[(size_t)worst * 32 + 1] - pr[(size_t)worst * 32]) worst = k;
      busy += (double)(e - b);
    }
    int late = 0;
    for (int k = 0; k < n_inst; ++k) late += pr[(size_t)k * 32] - t0 > 200;  // started more than 2 us after the first
    if (const char* path = std::getenv("HDSM_TIMELINE_DUMP")) {  // raw rows for offline analysis, one block per launch
      if (FILE* f = std::fopen(path, "ab")) {
        const long long head[2] = {0x54494d454c494e33LL, n_inst};  // ("TIMELIN3": 32 entries per instance)
        std::fwrite(head, sizeof(long long), 2, f);
        for (int k = 0; k < n_inst; ++k) std::fwrite(&pr[(size_t)k * 32], sizeof(long long), 32, f);
        std::fclose(f);
      }
    }
    auto us = [](long long ticks) { return (double)ticks * 0.01; };
    const long long *w = &pr[(size_t)worst * 32], *l = &pr[(size_t)last * 32];
    std::fprintf(stderr,
                 "HDSM_TIMELINE span %.2f us | slowest inst %d: %.2f us, start +%.2f, block %lld, iters %lld nodes %lld sweeps %lld staged %lld+%lld flags %lld status %lld | last to finish inst %d: "
                 "start +%.2f dur %.2f block %lld iters %lld | %d of %d started > 2 us late | sum of instance times %.1f us (%.2f per span-slot of 512)\n",
                 us(t1 - t0), worst, us(w[1] - w[0]), us(w[0] - t0), w[2], w[4], w[5], w[6], w[7], w[9], w[8], w[10], last, us(l[0] - t0), us(l[1] - l[0]), l[2], l[4], late,
                 n_inst, us((long long)busy), busy / (double)(t1 - t0) / 512.0);
  }
#endif
#ifdef HDSM_PROFILE
  {  // development aid: phase cycle counters of the slowest instance and the batch mean
    std::vector<long long> pr((size_t)n_inst * 32);
    HIP_TRY(hipMemcpy(pr.data(), h->d_prof, pr.size() * sizeof(long long), hipMemcpyDeviceToHost));
    int worst = 0;
    double mean[32] = {0};
    for (int k = 0; k < n_inst; ++k) {
      if (pr[(size_t)k * 32 + 11] > pr[(size_t)worst * 32 + 11]) worst = k;
      for (int j = 0; j < 32; ++j) mean[j] += (double)pr[(size_t)k * 32 + j] / n_inst;
    }
    static const char* nm[32] = {"states", "select", "normal", "d", "sums", "upd", "add", "drop", "setup", "sweep",
                                 "leaf", "TOTAL", "iters", "sweeps", "warm_ops", "warm_cycles", "sw_cull", "sw_filter",
                                 "sw_load", "sw_body", "sw_tail", "su_stage", "su_grad", "su_fact",
                                 "w_prep", "w_fetch", "w_normal", "w_dir", "w_add", "w_pair", "w_drop", "w_7"};
#ifdef HDSM_PROF_OP  // record 16..31 = slots 8..23: the inside of a regular operation (OP_PROF in hdsm_wave_gi.h / hdsm_wave_gib.h)
    static const char* op_nm[16] = {"sel_boxes", "sel_rows", "w0_wait_done", "normal_entry", "d_reduce", "d_sums", "d_gather_z", "d_urow", "add_scalars", "add_gather",
                                    "add_update", "sc_states", "sc_rows", "sc_max", "sc_normal", "sc_wait_go"};
    for (int j = 0; j < 16; ++j) nm[16 + j] = op_nm[j];
#endif
    std::fprintf(stderr, "HDSM_PROFILE worst inst %d:", worst);
    for (int j = 0; j < 32; ++j) std::fprintf(stderr, " %s=%lld", nm[j], pr[(size_t)worst * 32 + j]);
    std::fprintf(stderr, "\nHDSM_PROFILE mean:");
    for (int j = 0; j < 32; ++j) std::fprintf(stderr, " %s=%.0f", nm[j], mean[j]);
    std::fprintf(stderr, "\n");
  }
#endif
  int32_t* dst[4] = {qp_iters, nodes, sweeps, cand};
  for (int k = 0; k < 4; ++k)
    if (dst[k])
      HIP_TRY(hipMemcpy(dst[k], h->d_stats + (size_t)k * h->max_inst, (size_t)n_inst * 4, hipMemcpyDeviceToHost));
  return HDSM_OK;
}

int hdsm_solve(void* handle, int32_t n_inst, int32_t r_max, const double* state_curr,
               const double* traj_ref, const int32_t* n_poly, const int32_t* n_rows, const double* A,
               const double* b, double* traj_out, double* ctrl_out, uint8_t* poly_used, int32_t* status,
               double* obj) {
  Handle* h = static_cast<Handle*>(handle);
  if (int rc = check_common(h, n_inst, 0)) return rc;
  if (n_inst == 0) return HDSM_OK;
  if (!state_curr || !traj_ref || !n_poly || !n_rows || !A || !b || !traj_out || !ctrl_out || !poly_used || !status || !obj)
    return set_err(HDSM_ERR_BAD_ARG, "null array argument");
  hdsm::Level1Split sp;
  const char* msg = nullptr;
  if (int rc = hdsm::level1_split(h->prm, n_inst, r_max, n_poly, n_rows, A, b, &sp, &msg))
    return set_err(rc, msg ? msg : "bad level-1 input");
  HIP_TRY(hipSetDevice(h->device));
  const size_t I = (size_t)n_inst, N = (size_t)h->N, P = (size_t)h->P, RS = (size_t)h->RS;
  hipStream_t st = h->stream;
  HIP_TRY(ensure(h, h->b_common, sp.common.size() * sizeof(double)));
  hipError_t e = ensure(h, h->b_ncommon, sp.n_common.size() * sizeof(int32_t));
  double* d_common = static_cast<double*>(h->b_common.p);
  int32_t* d_ncommon = static_cast<int32_t*>(h->b_ncommon.p);
  if (e == hipSuccess && h->launched && st != h->last_stream) e = hipStreamWaitEvent(st, h->ev_done, 0);
  const auto H2D = hipMemcpyHostToDevice;
  const auto D2H = hipMemcpyDeviceToHost;
  std::vector<int32_t> ids(I, -1);
  auto cp = [&](void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    if (e == hipSuccess && bytes) e = hipMemcpyAsync(dst, src, bytes, kind, st);
  };
  cp(d_common, sp.common.data(), sp.common.size() * 8, H2D);
  cp(d_ncommon, sp.n_common.data(), sp.n_common.size() * 4, H2D);
  cp(h->d_agent, ids.data(), I * 4, H2D);
  cp(h->d_state, state_curr, I * 9 * 8, H2D);
  cp(h->d_ref, traj_ref, I * N * 6 * 8, H2D);
  cp(h->d_npoly, sp.n_poly.data(), I * 4, H2D);
  cp(h->d_nrows, sp.n_rows_static.data(), I * P * 4, H2D);
  cp(h->d_A, sp.A_static.data(), I * P * RS * 3 * 8, H2D);
  cp(h->d_b, sp.b_static.data(), I * P * RS * 8, H2D);
  cp(h->d_traj, traj_out, I * (N + 1) * 9 * 8, H2D);
  cp(h->d_ctrl, ctrl_out, I * N * 3 * 8, H2D);
  cp(h->d_used, poly_used, I * P, H2D);
  cp(h->d_obj, obj, I * 8, H2D);
  int rc = HDSM_OK;
  if (e == hipSuccess) {
    hdsm::Args a{};
    a.n_inst = n_inst, a.n_rob = 0, a.agent_id = h->d_agent, a.state = h->d_state, a.ref = h->d_ref;
    a.n_poly = h->d_npoly, a.n_rows = h->d_nrows, a.A = h->d_A, a.b = h->d_b, a.plans = h->d_plans;
    a.has_plan = h->d_zero, a.traj = h->d_traj, a.ctrl = h->d_ctrl, a.used = h->d_used, a.status = h->d_status;
    a.obj = h->d_obj, a.l1_rows = d_common, a.l1_nrows = d_ncommon, a.l1_rmax = sp.rc_max;
    rc = launch(h, a, st);
  }
  cp(traj_out, h->d_traj, I * (N + 1) * 9 * 8, D2H);
  cp(ctrl_out, h->d_ctrl, I * N * 3 * 8, D2H);
  cp(poly_used, h->d_used, I * P, D2H);
  cp(status, h->d_status, I * 4, D2H);
  cp(obj, h->d_obj, I * 8, D2H);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (rc) return rc;
  if (e != hipSuccess) return set_err(HDSM_ERR_DEVICE, std::string("hdsm_solve: ") + hipGetErrorString(e));
  return HDSM_OK;
}

int hdsm_last_sweep_stats(void* handle, int32_t n_inst, int32_t* sphere_records, int32_t* pairs, uint32_t* flags) {
  Handle* h = static_cast<Handle*>(handle);
  if (int rc = check_common(h, n_inst, 0)) return rc;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->last_stream));
  void* dst[3] = {sphere_records, pairs, flags};
  for (int k = 0; k < 3; ++k)
    if (dst[k])
      HIP_TRY(hipMemcpy(dst[k], h->d_stats + (size_t)(4 + k) * h->max_inst, (size_t)n_inst * 4, hipMemcpyDeviceToHost));
  return HDSM_OK;
}

// ---- multi-GPU exchange (RCCL) -------------------------------------------------------------------------------------
struct Comm {
  ncclComm_t nccl = nullptr;
  int rank = 0, world = 1, device = 0, rec = 0;  // rec = doubles per published record, (N + 1) * 9
};

#define NCCL_TRY(expr)                                                                                 \
  do {                                                                                                  \
    ncclResult_t r_ = (expr);                                                                           \
    if (r_ != ncclSuccess) return set_err(HDSM_ERR_COMM, std::string(#expr) + ": " + ncclGetErrorString(r_)); \
  } while (0)

int hdsm_comm_unique_id(uint8_t id[HDSM_COMM_ID_BYTES]) {
  static_assert(sizeof(ncclUniqueId) <= HDSM_COMM_ID_BYTES, "ncclUniqueId does not fit HDSM_COMM_ID_BYTES");
  if (!id) return set_err(HDSM_ERR_BAD_ARG, "null id");
  ncclUniqueId u;
  NCCL_TRY(ncclGetUniqueId(&u));
  std::memset(id, 0, HDSM_COMM_ID_BYTES);
  std::memcpy(id, &u, sizeof u);
  return HDSM_OK;
}

int hdsm_comm_create(void* handle, const uint8_t id[HDSM_COMM_ID_BYTES], int32_t rank, int32_t world, void** comm) {
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !id || !comm || world < 1 || rank < 0 || rank >= world) return set_err(HDSM_ERR_BAD_ARG, "bad hdsm_comm_create argument");
  *comm = nullptr;
  HIP_TRY(hipSetDevice(h->device));
  Comm* c = new (std::nothrow) Comm;
  if (!c) return set_err(HDSM_ERR_DEVICE, "out of host memory");
  c->rank = rank, c->world = world, c->device = h->device, c->rec = (h->N + 1) * 9;
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  ncclResult_t r = ncclCommInitRank(&c->nccl, world, u, rank);
  if (r != ncclSuccess) {
    delete c;
    return set_err(HDSM_ERR_COMM, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
  }
  *comm = c;
  return HDSM_OK;
}

int hdsm_comm_info(void* comm, int32_t* rank, int32_t* world) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c) return set_err(HDSM_ERR_BAD_ARG, "null comm");
  int r = -1, w = -1;
  NCCL_TRY(ncclCommUserRank(c->nccl, &r));   // what RCCL itself says, not what the caller passed
  NCCL_TRY(ncclCommCount(c->nccl, &w));
  if (rank) *rank = r;
  if (world) *world = w;
  return HDSM_OK;
}

void hdsm_comm_destroy(void* comm) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->nccl) (void)ncclCommDestroy(c->nccl);
  delete c;
}

int hdsm_publish_device(void* handle, int32_t per, int32_t n_local, const double* traj, const uint8_t* has_plan_local,
                        double* plans_local, void* hip_stream) {
  Handle* h = static_cast<Handle*>(handle);
  if (!h || per < 0 || n_local < 0 || n_local > per) return set_err(HDSM_ERR_BAD_ARG, "bad hdsm_publish_device argument");
  if (per == 0) return HDSM_OK;
  if (!traj || !has_plan_local || !plans_local) return set_err(HDSM_ERR_BAD_ARG, "null array argument");
  HIP_TRY(hipSetDevice(h->device));
  const int rec = (h->N + 1) * 9;
  const int64_t tot = (int64_t)per * rec;
  hipLaunchKernelGGL(k_publish, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(hip_stream), rec, per,
                     n_local, traj, has_plan_local, plans_local);
  HIP_TRY(hipGetLastError());
  return HDSM_OK;
}

int hdsm_exchange_device(void* comm, int32_t per, const double* plans_local, double* plans_all, uint8_t* has_plan_all,
                         void* hip_stream) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c || per < 0) return set_err(HDSM_ERR_BAD_ARG, "bad hdsm_exchange_device argument");
  if (per == 0) return HDSM_OK;
  if (!plans_local || !plans_all || !has_plan_all) return set_err(HDSM_ERR_BAD_ARG, "null array argument");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  // ONE collective per replan round: rank r's shard lands at plans_all + r * per * rec (in place if it already is there)
  NCCL_TRY(ncclAllGather(plans_local, plans_all, (size_t)per * c->rec, ncclDouble, c->nccl, st));
  const int n = per * c->world;
  hipLaunchKernelGGL(k_has_from_sentinel, dim3((n + 255) / 256), dim3(256), 0, st, c->rec, n, plans_all, has_plan_all);
  HIP_TRY(hipGetLastError());
  return HDSM_OK;
}

}  // extern "C"
