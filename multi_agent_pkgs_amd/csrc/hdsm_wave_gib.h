// hdsm_wave_gib.h — device-only (gfx950): the dual active-set iteration for n <= 30 (NV = 32) with ALL cross-lane traffic of the
// factor J in registers: no LDS transposition, no dynamically indexed register access.
//
// Same mathematics as hdsm_wave_gi.h (Goldfarb-Idnani on J = L^{-T} Q, U = R^{-1}, Householder add / drop); what changes is where
// the numbers live, chosen so that the two matrix-vector products with J that sit on the dependency chain of every active-set
// operation — d = J^T a (a reduction over the ROWS of J, i.e. across lanes) and z = J2 d2 (a reduction over the columns, i.e.
// inside a lane) — need DPP moves only:
//
//   * rows: row i of J is split over lanes i (columns 0..15) and i + 32 (columns 16..31), as before ("row layout": per-row
//     scalars x_i, z_i, a_i live in lanes i and i + 32);
//   * columns inside a lane are kept in BUTTERFLY ORDER: slot s of lane L holds column  col(L, s) = ((s ^ L) & 15) + 16 (L >> 5).
//     With that order the sum over the 16 lanes of a DPP row of one value per column ("reduce-scatter") is four stages of
//     `slot[s] += dpp(slot[partner(s)])` with STATIC slot numbers — row_mirror, row_half_mirror, quad_perm xor 2, quad_perm xor 1 —
//     15 double additions and 30 v_mov_dpp, no select, and lane L ends up holding column (L & 15) + 16 (L >> 5) in slot 0;
//     v_permlane16_swap adds the second DPP row of the half. The reverse walk ("all-gather") hands every lane the 16 entries
//     of a column-distributed vector in exactly the order of its own J slots: 30 v_mov_dpp, no LDS round trip;
//   * "position layout": what belongs to working-set position / column k — d_k, the multiplier, the constraint id, r_k and
//     row k of U (LDS, natural column order) — lives in lanes P(k) = (k & 15) + 32 (k >> 4) and P(k) + 16; the lane with bit 4
//     clear takes columns 0..15 of the U row, its twin columns 16..31, and the two partial dot products meet through
//     v_permlane16_swap. Multipliers and ids stay in registers during a run (they were an LDS round trip per operation);
//   * the Householder reflections are applied as J -= (J v) beta v^T with the COMPLETE v (v = d2 - rho e_q, or u_l - sigma e_t
//     for a drop) gathered from its column-distributed form, so column q (or t) needs no special treatment and no register of
//     J is ever addressed by a run-time index (the old layout paid a 16-deep select chain per access).
//
// Everything that does not touch J's layout (state evaluation, violation scan, helper waves, neighbour sweep, residuals) is
// inherited from WaveGI<NV, CMAX>. NV = 48 (H > 10): the same source with one lane per row and three blocks of slots (below).
#pragma once
#include "hdsm_wave_gi.h"

namespace hdsm {

// v[L] + v[L ^ 16] in every lane (v_permlane16_swap: the two DPP rows of a half)
__device__ __forceinline__ double row16_sum64(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}

// v_permlane16_swap / v_permlane32_swap on doubles with two different operands: .v = the new first operand, .s = the new second one.
//   swap16(a, b): even 16-lane rows get {own a, a of the odd row next to them}, odd rows get {b of the even row, own b}
//   swap32(a, b): lanes 0..31 get {own a, a of lane + 32}, lanes 32..63 get {b of lane - 32, own b}
struct Pair64 {
  double v, s;
};
__device__ __forceinline__ Pair64 swap16(double a, double b) {
  const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
  return Pair64{__hiloint2double(hi[0], lo[0]), __hiloint2double(hi[1], lo[1])};
}
__device__ __forceinline__ Pair64 swap32(double a, double b) {
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
  return Pair64{__hiloint2double(hi[0], lo[0]), __hiloint2double(hi[1], lo[1])};
}

// NVT = 32: the layout described above. NVT = 48 (n <= 45, H <= 15; round 4): one lane per row (lanes 0..47; the last DPP row
// carries zeros), 48 slots = three 16-column blocks, each in butterfly order — slot 16 b + j of lane L holds column
// 16 b + ((j ^ L) & 15). The reduce-scatter runs per block inside the 16-lane DPP rows and the three DPP rows are then combined so
// that the total of column k lands in lane k ("position layout" = lane k for position k, one copy): blocks 0 and 1 with ONE
// v_permlane16_swap of (block 0, block 1) — the even row receives its neighbour's block-0 partial, the odd row its neighbour's
// block-1 partial — plus a v_permlane32_swap sum; block 2 with the two swap sums. The all-gather is the reverse: three seeds per
// lane from three swaps, then the butterfly walk per block. No LDS transposition, no select chain on a run-time register index
// (what the one-lane-per-row code of hdsm_wave_gi.h paid: 48-deep v_cndmask chains, 21 k cycles per operation).
template <int NVT, int CMAX, bool SMALL = false>
struct WaveGIB : WaveGI<NVT, CMAX, SMALL> {
  using Base = WaveGI<NVT, CMAX, SMALL>;
  using S = typename Base::S;
  using Regs = typename Base::Regs;
  static constexpr bool SPLIT = Base::SPLIT;
  static constexpr int NV = NVT, NC = Base::NC, NB = NC / 16, LDT = S::LDT;
  static_assert(NVT == 32 || NVT == 48, "two layouts");

  static __device__ __forceinline__ int pos_of(int lane) { return SPLIT ? (lane & 15) + 16 * (lane >> 5) : lane; }
  static __device__ __forceinline__ int lane_of_pos(int k) { return SPLIT ? (k & 15) + 32 * (k >> 4) : k; }
  static __device__ __forceinline__ bool first_copy(int lane) { return SPLIT ? (lane & 16) == 0 : lane < NV; }
  // a position that exists, for addresses (NVT = 48: lanes 48..63 hold no position; what they read is never used)
  static __device__ __forceinline__ int pos_addr(int lane) { return SPLIT ? pos_of(lane) : (lane < NV ? lane : NV - 1); }
  static __device__ __forceinline__ int row_addr(int lane) { return SPLIT ? (lane & 31) : (lane < NV ? lane : NV - 1); }
  // first column of this lane's share of a row of U, and the sum over the lanes that share a position
  static __device__ __forceinline__ int ucol0(int lane) { return SPLIT ? (lane & 16) : 0; }
  static __device__ __forceinline__ double psum(double v) {
    if constexpr (SPLIT) return row16_sum64(v);
    else return v;
  }

  // reduce-scatter of one 16-slot block inside the 16-lane DPP row: p[0] = this row's partial of column (L & 15) of the block
  static __device__ __forceinline__ void reduce_block(double* p) {
#pragma unroll
    for (int s = 0; s < 8; ++s) p[s] += dpp64<0x140>(p[15 - s]);  // row_mirror: partner L ^ 15
#pragma unroll
    for (int s = 0; s < 4; ++s) p[s] += dpp64<0x141>(p[7 - s]);   // row_half_mirror: partner L ^ 7
#pragma unroll
    for (int s = 0; s < 2; ++s) p[s] += dpp64<0x4E>(p[s ^ 2]);    // quad_perm [2,3,0,1]: partner L ^ 2
    p[0] += dpp64<0xB1>(p[1]);                                    // quad_perm [1,0,3,2]: partner L ^ 1
  }
  static __device__ __forceinline__ void gather_block(double* g) {  // g[0] = entry (L & 15) of the block -> g[j] = entry (j ^ L) & 15
    g[1] = dpp64<0xB1>(g[0]);
#pragma unroll
    for (int s = 0; s < 2; ++s) g[s ^ 2] = dpp64<0x4E>(g[s]);
#pragma unroll
    for (int s = 0; s < 4; ++s) g[7 - s] = dpp64<0x141>(g[s]);
#pragma unroll
    for (int s = 0; s < 8; ++s) g[15 - s] = dpp64<0x140>(g[s]);
  }
  // sum over the rows of J of slot-ordered per-column values: returns, in lane L, the total of column pos_of(L)
  static __device__ __forceinline__ double reduce_cols(double (&p)[NC], int lane = 0) {
    if constexpr (SPLIT) {
      reduce_block(p);
      return row16_sum64(p[0]);
    } else {
#pragma unroll
      for (int b = 0; b < NB; ++b) reduce_block(p + 16 * b);
      const Pair64 e = swap16(p[0], p[16]);              // DPP rows 0 / 2: block 0 of rows {0,1} / {2,3}; rows 1 / 3: block 1
      const double t01 = half_sum64(e.v + e.s);          // rows 0 and 2: all of block 0; rows 1 and 3: all of block 1
      const double t2 = half_sum64(row16_sum64(p[32]));  // every row: all of block 2
      return lane < NV ? ((lane & 32) ? t2 : t01) : 0.0;
    }
  }
  // v = entry pos_of(L) of a column-distributed vector -> g[s] = the entry of the column slot s of lane L holds
  static __device__ __forceinline__ void gather_cols(double v, double (&g)[NC]) {
    if constexpr (SPLIT) {
      g[0] = v;
      gather_block(g);
    } else {
      const Pair64 a = swap16(v, v);      // .v: the entry of DPP row 0 (rows 0, 1) / of row 2 (rows 2, 3); .s: of row 1 / of row 3
      const Pair64 b = swap32(a.v, a.v);  // .v: the entry of row 0 in every row; .s: the entry of row 2 in every row
      const Pair64 c = swap32(a.s, a.s);  // .v: the entry of row 1 in every row
      g[0] = b.v, g[16] = c.v, g[32] = b.s;
#pragma unroll
      for (int bb = 0; bb < NB; ++bb) gather_block(g + 16 * bb);
    }
  }
  // sum over the DISTINCT positions of a position-layout value (every position has two copies)
  static __device__ __forceinline__ double pos_sum(double v, int lane) { return wave_sum64(first_copy(lane) ? v : 0.0); }

  // multipliers / ids: LDS (between the phases of an instance, position k at index k) <-> registers (inside a run)
  static __device__ __forceinline__ void load_pos(const S& s, Regs& R, int lane) {
    R.lam = s.lam[pos_addr(lane)], R.act = first_copy(lane) || SPLIT ? s.act[pos_addr(lane)] : -1;
  }
  static __device__ __forceinline__ void store_pos(S& s, const Regs& R, int lane) {
    if (first_copy(lane)) s.lam[pos_of(lane)] = R.lam, s.act[pos_of(lane)] = R.act;
  }

  // dot product of this lane's half (columns 16 (lane bit 4) ..) of row k = pos_of(lane) of U with s.dvec, both halves summed.
  // (Reading only the pairs below q — columns >= q of U are zero — was tried: the data-dependent trip count costs more than the
  // reads it saves, 8.35 -> 7.85 M agent-replans/s on the bench line.)
  static __device__ __forceinline__ double u_row_dot(const S& s, int lane) {
    const int k = pos_addr(lane), c0 = ucol0(lane);
    const D2* urow = reinterpret_cast<const D2*>(&s.U[k * LDT + c0]);
    const D2* dv = reinterpret_cast<const D2*>(&s.dvec[c0]);
    double r0 = 0, r1 = 0;
#pragma unroll
    for (int j = 0; j < NC / 2; ++j) {
      const D2 u = urow[j], d = dv[j];
      r0 += u.x * d.x, r1 += u.y * d.y;
    }
    return psum(r0 + r1);
  }

  // ---- hand-over between wave 0 and the scanner (wave 1): workgroup barriers with a command word. (Words in LDS polled by the
  // other side — no barrier at all — were built and measured: slower, 11.1 against 11.55 M agent-replans/s on the bench line. What
  // costs is the round trip of the LDS write that carries the message, about 250 cycles either way, and a polling wave adds its
  // sleep granularity on top.)
  static constexpr int CMD_PICK = 2, CMD_PREP = 3;
  // scalars of the Householder reflection that maps d2 onto rho e_q (v = d2 - rho e_q, beta = 2 / v^T v)
  struct Refl {
    double rho, inv_rho, beta;
  };
  static __device__ __forceinline__ Refl reflection(double zz, double dq) {
    const double zs = zz > 0 ? zz : 1.0;  // (a dependent row: the scalars are not used)
    const double inv_rho_abs = rsq_nr(zs);
    Refl h;
    h.rho = (dq > 0 ? -1.0 : 1.0) * (zs * inv_rho_abs), h.inv_rho = (dq > 0 ? -1.0 : 1.0) * inv_rho_abs;
    h.beta = rcp_nr(h.rho * (h.rho - dq));
    return h;
  }

  // d = J^T(-a) (position layout), ||d||^2, ||d2||^2, d_q, z = J2 d2 (row layout), r = U d1 (position layout), and dz = d with the
  // working-set columns zeroed (position layout: the source of the Householder vector). `ai` = entry row_of(lane) of the normal.
  // WANT_Z = false (warm-start additions: no step is taken): z is not formed (one all-gather and one dot product less).
  // WANT_DD = false: ||d||^2 is not formed (a wave sum less; the regular loop then tests the dependency of the entering row against
  // a^T Z a, which the pick rule hands over with the row — see run()).
  // prep_msg != 0 (regular loop of a workgroup with a scanner): z goes to LDS (Shm::w) and the scanner is told to prepare the next
  // pick (CMD_PREP) as soon as z exists — it works while this wave forms r, runs the ratio test and chooses the step.
  template <bool WANT_Z = true, bool WANT_DD = true>
  static __device__ __forceinline__ void direction(S& s, const Regs& R, double ai, int q, int lane, double& dj, double& dz,
                                                   double& dd, double& zz, double& dq, double& zi, double& ri, int prep_msg = 0,
                                                   Refl* refl = nullptr) {
    double p[NC];
    const double na = -ai;
#pragma unroll
    for (int k = 0; k < NC; ++k) p[k] = R.Jr[k] * na;
    dj = reduce_cols(p, lane);
    if constexpr (WANT_Z) { OP_PROF(12) }
    const int pos = pos_of(lane);
    if (first_copy(lane)) s.dvec[pos] = dj;  // for r = U d (U has zero columns >= q: no mask needed)
    dz = (pos >= q) ? dj : 0.0;
    dd = WANT_DD ? pos_sum(dj * dj, lane) : 0.0;
    zz = pos_sum(dz * dz, lane);
    dq = (q < NV) ? bcast64(dj, lane_of_pos(q < NV ? q : 0)) : 0.0;
    zi = 0.0;
    // (two chains of dependent scalar operations: formed here they run in the shadow of the all-gather, not after the step)
    if (refl != nullptr) *refl = reflection(zz, dq);
    if constexpr (WANT_Z) {
      OP_PROF(13)
      double g[NC];
      gather_cols(dz, g);
      double z0 = 0, z1 = 0;
#pragma unroll
      for (int k = 0; k < NC; k += 2) z0 += R.Jr[k] * g[k], z1 += R.Jr[k + 1] * g[k + 1];
      zi = Base::hsum(z0 + z1);
      OP_PROF(14)
      if (prep_msg != 0) {  // message to the scanner: z is there
        if (lane < NV) s.w[lane] = zi;
        if (lane == 0) s.cmd = prep_msg;
        __syncthreads();  // B1
      }
    }
    wsync();
    ri = u_row_dot(s, lane);
    if constexpr (WANT_Z) { OP_PROF(15) }
  }

  // working set += id at position q: ONE Householder reflection of the free columns, d2 -> rho e_q, applied with the complete
  // vector v = d2 - rho e_q: (J2 v) comes from the same gathered v that the rank-1 update multiplies
  template <bool PROBE = false>  // (PROBE: the call of the regular loop, timed by the -DHDSM_PROF_OP build)
  static __device__ __forceinline__ void householder_add(S& s, Regs& R, int id, double lam_p, int q, int lane, double dz, double zz,
                                                         double dq, double ri, const Refl* refl = nullptr) {
    const Refl hh = refl != nullptr ? *refl : reflection(zz, dq);  // (zz > 0: the caller has tested it against ||d||^2)
    const double rho = hh.rho, inv_rho = hh.inv_rho, beta = hh.beta;
    const int pos = pos_of(lane);
    double g[NC];
    if constexpr (PROBE) { OP_PROF(16) }
    gather_cols(pos == q ? dz - rho : dz, g);
    if constexpr (PROBE) { OP_PROF(17) }
    double w0 = 0, w1 = 0;
#pragma unroll
    for (int k = 0; k < NC; k += 2) w0 += R.Jr[k] * g[k], w1 += R.Jr[k + 1] * g[k + 1];
    const double coef = Base::hsum(w0 + w1) * beta;
#pragma unroll
    for (int k = 0; k < NC; ++k) R.Jr[k] -= coef * g[k];
    if constexpr (PROBE) { OP_PROF(18) }
    if (first_copy(lane)) s.U[pos * LDT + q] = (pos < q) ? -ri * inv_rho : ((pos == q) ? inv_rho : 0.0);
    if (pos == q) R.lam = lam_p, R.act = id;
    wsync();
  }

  // working set -= entry at position l: one Householder reflection G with G u_l^T = sigma e_t (u_l = row l of U, t = q - 1).
  // The rows of U (LDS, natural order) are updated by the lane pair of their position; J through the gathered vector.
  static __device__ __forceinline__ void drop(S& s, Regs& R, int l, int q, int lane) {
    const int t = q - 1, pos = pos_of(lane), pa = pos_addr(lane), c0 = ucol0(lane);
    double uv[NC];  // this lane's columns of row l of U
    {
      const D2* rl = reinterpret_cast<const D2*>(&s.U[l * LDT + c0]);
#pragma unroll
      for (int j = 0; j < NC; j += 2) {
        const D2 v2 = rl[j / 2];
        uv[j] = v2.x, uv[j + 1] = v2.y;
      }
    }
    const double ut = s.U[l * LDT + t];
    const double ulp = s.U[l * LDT + pa];  // entry pos of u_l: the column-distributed source for J's update
    double s0 = 0, s1 = 0;
#pragma unroll
    for (int j = 0; j < NC; j += 2) s0 += uv[j] * uv[j], s1 += uv[j + 1] * uv[j + 1];
    const double ss = psum(s0 + s1);  // |u_l|^2 > 0: u_l is a row of the inverse of a regular triangular factor
    const double sigma = (ut > 0 ? -1.0 : 1.0) * (ss * rsq_nr(ss));
    const double beta = rcp_nr(sigma * (sigma - ut));  // 2 / (v^T v), v = u_l - sigma e_t
    // J (registers): x -= (x . v) beta v with the complete v
    {
      double g[NC];
      gather_cols(pos == t ? ulp - sigma : ulp, g);
      double w0 = 0, w1 = 0;
#pragma unroll
      for (int k = 0; k < NC; k += 2) w0 += R.Jr[k] * g[k], w1 += R.Jr[k + 1] * g[k + 1];
      const double cj = Base::hsum(w0 + w1) * beta;
#pragma unroll
      for (int k = 0; k < NC; ++k) R.Jr[k] -= cj * g[k];
    }
    // own row of U (row pos): x -= (x . v) beta v; column t belongs to the freed direction and is zeroed below
    double ur[NC];
    {
      const D2* ro = reinterpret_cast<const D2*>(&s.U[pa * LDT + c0]);
#pragma unroll
      for (int j = 0; j < NC; j += 2) {
        const D2 v2 = ro[j / 2];
        ur[j] = v2.x, ur[j + 1] = v2.y;
      }
    }
    const double urt = s.U[pa * LDT + t];
    double wu0 = 0, wu1 = 0;
#pragma unroll
    for (int j = 0; j < NC; j += 2) wu0 += ur[j] * uv[j], wu1 += ur[j + 1] * uv[j + 1];
    const double cu = (psum(wu0 + wu1) - sigma * urt) * beta;
#pragma unroll
    for (int j = 0; j < NC; ++j) ur[j] -= cu * uv[j];
    // multipliers / ids of the positions above l move down by one (through LDS: positions cross the DPP rows)
    if (first_copy(lane)) s.lam[pos] = R.lam, s.act[pos] = R.act;
    wsync();  // every lane has read its rows before anybody rewrites a slot
    if (pos != l && pos < q) {  // rows above l stay, rows l+1..q-1 move up one slot, row l (the dropped entry) disappears
      D2* dst = reinterpret_cast<D2*>(&s.U[(pos > l ? pos - 1 : pos) * LDT + c0]);
#pragma unroll
      for (int j = 0; j < NC; j += 2) dst[j / 2] = D2{ur[j], ur[j + 1]};
    }
    if (pos >= l && pos < t) R.lam = s.lam[pos + 1], R.act = s.act[pos + 1];
    if (pos == t) R.lam = 0.0, R.act = -1;
    wsync();
    // structural zeros: column t of every row belongs to the freed direction, slot t is empty again
    if (first_copy(lane)) s.U[pos * LDT + t] = 0.0;
    if (pos == t) {
      D2* dst = reinterpret_cast<D2*>(&s.U[pos * LDT + c0]);
#pragma unroll
      for (int j = 0; j < NC; j += 2) dst[j / 2] = D2{0.0, 0.0};
    }
    wsync();
  }

  // ---- warm start (described in hdsm_wave_gi.h): the guess is put into the factorisation without taking steps, then the S-pair in
  // closed form  t = U^T v, lambda = U t, x_W = x0 + J1 t, f_W = f(x0) + |t|^2 / 2
  using WarmPre = typename Base::WarmPre;
  using Base::warm_prefetch;
  // `conc`: wave 1 sweeps the neighbour rows while this wave installs the guess (hdsm_core.h): ONE workgroup barrier, on every path,
  // right after the rows of the guess have taken their staging slots — from there on this function no longer touches the counters
  // of the staging area, and the sweep may add its rows behind them.
  static __device__ __forceinline__ void warm_start(S& s, const Consts& c, const Args& a, Regs& R, int inst, int self, int& iters,
                                                    const WarmPre& wpre, bool conc = false) {
    const int lane = (int)HDSM_TX;
    const int N = c.N, n = c.n;
    int nw = uni(wpre.head) & ~WARM_CERT;
    if (nw <= 0) {
      if (conc) {
        if (lane == 0) s.warm_ncand = s.ncand;
        __syncthreads();
      }
      return;
    }
    if (nw > NV) nw = NV;
    PROF_DECL
    int pre = -1, my_m = 0, my_src = 0;  // pre: >= 0 a ready id, -2 a neighbour row held in my_row, -1 nothing usable
    double my_row[4] = {0.0, 0.0, 0.0, 0.0};
    if (lane < nw) {
      const int code = wpre.code;
      const int kind = id_kind(code), p = id_payload(code);
      if (kind == K_U) {
        const int var = p >> 1;
        if (var % N >= 1) pre = mk_id(K_U, ((var - 1) << 1) | (p & 1));
      } else if (kind == K_S) {
        const int i = p >> 5;
        if (i - 1 >= 1) pre = mk_id(K_S, ((i - 1) << 5) | (p & 31));
      } else if (kind == K_C && a.l1_rows == nullptr) {
        const int e = p & 1, i = ((p >> 1) & 31) - 1, k = p >> 6;
        if (i >= 0 && i + e > c.pinned_steps && k != self && k < a.n_rob && wpre.has) {
          const double op[3] = {wpre.ox, wpre.oy, wpre.oz};
          if (tasc_plane_eval(c, s.cprev[i], op, my_row)) pre = -2, my_m = i + e, my_src = (k << 6) | (i << 1) | e;
        }
      }
    }
    int q = uni(s.q);
    const int q_in = q;
    load_pos(s, R, lane);
    {  // the neighbour rows of the guess go into the staging area in ONE step: every lane that holds one takes the next free slot
      const unsigned long long want = __ballot(pre == -2);
      if (want != 0ull) {
        const int base = uni(s.ncand), room = CMAX - uni(s.ncold) - base;
        const int rank = __popcll(want & ((1ull << lane) - 1ull));
        if (pre == -2) {
          if (rank < room) {
            const int slot = base + rank;
            s.cand[slot][0] = my_row[0], s.cand[slot][1] = my_row[1], s.cand[slot][2] = my_row[2], s.cand[slot][3] = my_row[3];
            s.cand_mw[slot] = mk_mw(s.kap, my_row[0], my_row[1], my_row[2], my_m);
            s.cand_src[slot] = my_src;
            pre = mk_kc(slot, my_m);
          } else {
            pre = -1;
          }
        }
        const int cnt = __popcll(want);
        if (lane == 0) s.ncand = base + (cnt < room ? cnt : (room > 0 ? room : 0));
        wsync();
      }
    }
    if (conc) {
      if (lane == 0) s.warm_ncand = s.ncand;
      __syncthreads();
    }
    WS_PROF(16)
    for (int g = 0; g < nw && q < n; ++g) {
      const int id = __builtin_amdgcn_readlane(pre, g);
      if (id < 0) continue;
      WS_PROF(17)
      double ai;
      if (id_kind(id) == K_C) {  // the row is in lane g's registers: its normal needs one LDS read (the impulse response), not two in a chain
        const double nx = bcast64(my_row[0], g), ny = bcast64(my_row[1], g), nz = bcast64(my_row[2], g);
        const int m = kc_m(id_payload(id)), var = Base::row_of(lane);
        const double nax = R.ax == 0 ? nx : (R.ax == 1 ? ny : nz);
        ai = (var < n && R.kk < m) ? nax * s.gz[R.ax][0][MAXH + m - 1 - R.kk] : 0.0;
      } else {
        ai = Base::normal_entry(s, R, id, Base::row_of(lane), N, n);
      }
      WS_PROF(18)
      double dj, dz, dd, zz, dq, zi, ri;
      direction<false>(s, R, ai, q, lane, dj, dz, dd, zz, dq, zi, ri);
      WS_PROF(19)
      ++iters;
      if (!(zz > 1e-8 * dd)) continue;  // (nearly) dependent on what is already in: leave it out
      householder_add(s, R, id, 0.0, q, lane, dz, zz, dq, ri);
      WS_PROF(20)
      ++q;
    }
    if (q == q_in) return;  // nothing usable (multipliers / ids in LDS are untouched)
    // violations of the working-set rows at the unconstrained minimiser x0
    if (lane < NV) s.x[lane] = s.x0[lane];
    store_pos(s, R, lane);  // resid() of K_P rows and the hand-over read act[] from LDS
    wsync();
    Base::states(s, R, lane, N);
    const int pos = pos_of(lane), pa = pos_addr(lane), c0 = ucol0(lane);
    for (;;) {
      const double vk = (pos < q) ? Base::resid(s, c, R.act, N) : 0.0;
      if (first_copy(lane)) s.dvec[pos] = vk;
      wsync();
      double tj;  // t = U^T v: column pos of U, this lane's share of the rows (rows >= q of U are zero)
      {
        double p0 = 0, p1 = 0;
#pragma unroll
        for (int k = 0; k < NC; k += 2) {
          const D2 vk2 = *reinterpret_cast<const D2*>(&s.dvec[c0 + k]);
          p0 += s.U[(c0 + k) * LDT + pa] * vk2.x, p1 += s.U[(c0 + k + 1) * LDT + pa] * vk2.y;
        }
        tj = psum(p0 + p1);
      }
      wsync();
      if (first_copy(lane)) s.dvec[pos] = tj;
      wsync();
      const double lk = u_row_dot(s, lane);  // lambda = U t
      // most negative multiplier among the inequalities
      const bool ineq = pos < q && id_kind(R.act) != K_E;
      const double worst = -wave_max64(ineq ? -lk : -DINF);
      if (!(worst < -1e-12) || q <= 6) {
        double xw;  // x_W = x0 + J1 t (t is zero beyond q): only the pair that is kept needs it
        {
          double g[NC];
          gather_cols(tj, g);
          double x0 = 0, x1 = 0;
#pragma unroll
          for (int k = 0; k < NC; k += 2) x0 += R.Jr[k] * g[k], x1 += R.Jr[k + 1] * g[k + 1];
          xw = s.x0[row_addr(lane)] + Base::hsum(x0 + x1);
        }
        const double tt = pos_sum(tj * tj, lane);
        R.lam = (pos < q) ? lk : 0.0;
        if (lane < n) R.xi = xw, s.x[lane] = xw;
        else if (lane < 64 && Base::row_ok(lane) && Base::row_of(lane) < n) R.xi = xw;
        if (lane == 0) s.f = s.fx0 + 0.5 * tt, s.q = q;
        store_pos(s, R, lane);
        wsync();
        WS_PROF(21)
        return;
      }
      const int l = pos_of(uni(__ffsll((long long)__ballot(ineq && lk == worst)) - 1));
      WS_PROF(21)
      drop(s, R, l, q, lane);
      WS_PROF(22)
      --q;
      ++iters;
    }
  }

  // Continues from the current (dual feasible) state until no row of the current node is violated.
  static __device__ __forceinline__ int run(S& s, const Consts& c, Regs& R, double f_cut, int& iters) {
    const int lane = (int)HDSM_TX;
    const int n = c.n, N = c.N, max_iters = c.max_iters;
    const double tol = c.tol;
    const long long time_ticks = c.time_ticks;  // 0 = no wall-clock budget (the default)
    const bool norm_pick = c.pick_rule != 0;
    double f = s.f;
    int q = uni(s.q), neq = uni(s.neq_done);
    int rc = GI_OK;
    const int pos = pos_of(lane);
    load_pos(s, R, lane);
    // Workgroups of more than one wavefront: wave 1 is the SCANNER (helper_loop below). Evaluating the trajectory at the new
    // iterate and picking the row that enters next reads nothing the Householder update writes, so the scanner does both while
    // this wave applies the update of the operation before; the normal of the picked row comes back through LDS with the pick.
    const bool duo = blockDim.x > 64 && c.scanner != 0;
    bool pending = false;  // a pick is under way (requested with the last step): its answer is behind the next barrier
    // a child of a branch-and-bound node starts at the node's minimiser, where only rows of its new polyhedron are violated: the
    // leaf test of the node has already found the one to enter first (hdsm_core.h, select_child) — no evaluation, no scan
    int first_id = uni(s.first_id);
    if (lane == 0) s.first_id = -1;
    PROF_DECL
    for (;;) {
      // (the lane masks and addresses of the state evaluation and the scan are formed here, per operation: hoisted out of
      // the loop they sat in spilled SGPR pairs and came back through v_readlane, 18 per operation)
      int ln = lane;
      keep_in_loop(ln);
      int ip;
      double vip, kip = 0.0;  // kip: the pick's key = vip / sqrt(a^T Z a) (normalised rule)
      double ai;
      if (neq < 6) {
        Base::states(s, R, ln, N);
        PROF(0)
        ip = mk_id(K_E, neq);
        vip = Base::resid(s, c, ip, N);
        ai = Base::normal_entry(s, R, ip, Base::row_of(lane), N, n);
      } else if (first_id >= 0) {
        ip = first_id, vip = s.first_v, kip = 0.0;  // (no key: the dependency test of this operation forms ||d||^2 itself)
        first_id = -1;
        ai = Base::normal_entry(s, R, ip, Base::row_of(lane), N, n);
      } else if (duo) {
        if (!pending) {
          if (lane == 0) s.cmd = CMD_PICK;
          __syncthreads();  // go
        }
        OP_PROF(2)          // (what was left of the update: booked with the operation before)
        __syncthreads();    // done
        OP_PROF(10)         // waiting for the scanner
        pending = false;
        ip = uni(s.part_id[1]);
        if (ip == -2) {  // nothing is violated at the extrapolated point: the exact evaluation at s.x confirms it (or finds the next row)
          if (lane == 0) s.cmd = CMD_PICK;
          __syncthreads();  // go
          __syncthreads();  // done
          ip = uni(s.part_id[1]);
        }
        PROF(1)
        if (ip < 0) break;
        vip = s.part_v[1], kip = s.part_key[1];
        ai = s.dvz[row_addr(ln)];
      } else {
        // the state boxes (velocity / acceleration limits) are looked at — and the states they bound evaluated — only when no
        // input box and no plane is violated: a third of the instructions of evaluation + scan, spared in most operations
        Base::template states<1>(s, R, ln, N);
        PROF(0)
        Base::template select<1, false>(s, c, R, ln, tol, N, vip, ip, &kip);
        ip = uni(ip);
        if (ip < 0) {
          Base::template states<2>(s, R, ln, N);
          Base::template select<2, false>(s, c, R, ln, tol, N, vip, ip, &kip);
          ip = uni(ip);
        }
        if (ip < 0) {
          if (Base::promote_cold(s, lane, tol) > 0) continue;
          PROF(1)
          break;
        }
        PROF(1)
        ai = Base::normal_entry(s, R, ip, Base::row_of(lane), N, n);
        OP_PROF(11)
      }
      const bool is_eq = id_kind(ip) == K_E;
      // 1e-20 a^T Z a of the entering row, from the key of its pick (normalised rule: kip = vip / sqrt(a^T Z a)); < 0: not available
      double dep_thr = -1.0;
      if (norm_pick && kip > 0.0) {
        const double root = vip * rcp_nr(kip);
        dep_thr = 1e-20 * root * root;
      }
#ifdef HDSM_TRACE_GI
      if (lane == 0) {
        const int kd = id_kind(ip), pl = id_payload(ip);
        if (kd == K_C) printf("  [blk %d] pick C m=%d src(k=%d,i=%d,e=%d) v=%.3e q=%d f=%.6g\n", kc_m(pl), (int)blockIdx.x, s.cand_src[kc_slot(pl)] >> 6, (s.cand_src[kc_slot(pl)] >> 1) & 31, s.cand_src[kc_slot(pl)] & 1, vip, q, f);
        else printf("  [blk %d] pick kind=%d payload=%d v=%.3e q=%d f=%.6g\n", (int)blockIdx.x, kd, pl, vip, q, f);
      }
#endif
      double lam_p = 0;
      bool stop = false;
      for (;;) {
        if (iters >= max_iters) {
          rc = GI_ITERLIM;
          stop = true;
          break;
        }
        if (time_ticks > 0 && (long long)wall_clock64() - s.t_start > time_ticks) {
          rc = GI_TIMELIM;
          stop = true;
          break;
        }
        ++iters;
        PROF(2)
        double dj, dz, dd, zz, dq, zi, ri;
        // Dependency of the entering row on the working set: ||d_2||^2 against ||d||^2 = a^T H^-1 a. With the normalised pick rule the
        // row arrives with a^T Z a = (vip / kip)^2 at the moment of the pick (dep_thr), and ||d_2||^2 <= a^T Z a <= a^T H^-1 a: the test
        // against a^T Z a differs only where ||d_2||^2 is 1e-20 of either — far below rounding — and spares the wave sum of ||d||^2.
        bool dependent;
        const bool prep = duo && neq >= 6;
        Refl refl;
        direction<true, false>(s, R, ai, q, lane, dj, dz, dd, zz, dq, zi, ri, prep ? CMD_PREP : 0, &refl);  // (one call site: the body is inlined)
        // (after B1 the scanner prepares, then it waits at B2 for the step, or for the word that there is none)
        auto post_step = [&](double tv) {
          if (prep) {
            if (lane == 0) s.part_v[0] = tv;
            __syncthreads();  // B2
          }
        };
        auto no_step = [&]() { post_step(-1.0); };
        if (dep_thr >= 0.0) {
          dependent = !(zz > dep_thr) || q >= NV;
        } else {
          dd = pos_sum(dj * dj, lane);
          dependent = !(zz > 1e-20 * dd) || q >= NV;
        }
        PROF(3)
        double t1 = DINF;
        int l = -1;
        if (!is_eq) {  // ratio test over the active inequalities (position layout)
          const bool okk = pos < q && id_kind(R.act) != K_E && ri > 1e-250;  // (1e-250: the Newton reciprocal wants a normal number)
          const double ratio = okk ? R.lam * rcp_nr(okk ? ri : 1.0) : DINF;
          const double m = -wave_max64(-ratio);
          if (m < DINF) {
            t1 = m;
            l = pos_of(uni(__ffsll((long long)__ballot(okk && ratio == m)) - 1));
          }
        }
        PROF(4)
#ifdef HDSM_TRACE_GI
        if (lane == 0) printf("    [blk %d] step dep=%d t1=%.3e t2=%.3e l=%d zz=%.3e dd=%.3e dep_thr=%.3e kip=%.3e\n", (int)blockIdx.x, (int)dependent, t1, dependent ? 0.0 : vip / zz, l, zz, dd, dep_thr, kip);
#endif
        if (dependent && l < 0) {
          no_step();
          rc = GI_INFEASIBLE;
          if (lane == 0) s.inf_id = ip;  // the row that cannot be satisfied together with the current working set
          stop = true;
          break;
        }
        if (dependent) {  // dual step only; constraint l leaves
          no_step();
          if (pos < q) R.lam -= t1 * ri;
          lam_p += t1;
          drop(s, R, l, q, lane);
          --q;
          PROF(7)
          continue;
        }
        const double t2 = vip * rcp_nr(zz);  // (not dependent: zz > 1e-20 ||d||^2 > 0)
        const bool full = is_eq || t2 <= t1;
        const double t = full ? t2 : t1;
        R.xi += t * zi;  // (lanes beyond n carry zeros: z is zero on padded rows)
        if (pos < q) R.lam -= t * ri;
        f += t * zz * (0.5 * t + lam_p);
        lam_p += t;
        PROF(5)
        if (full) {
          if (prep) {  // the scanner picks the next row at x + t z (it holds x, z and everything it prepared from them)
            post_step(t);  // (t >= 0)
            pending = true;
          }
          if (lane < n) s.x[lane] = R.xi;  // (after B2: the scanner reads the OLD iterate between B1 and B2)
          householder_add<true>(s, R, ip, lam_p, q, lane, dz, zz, dq, ri, &refl);
          PROF(6)
          ++q;
          if (is_eq) ++neq;
          break;
        }
        no_step();
        if (lane < n) s.x[lane] = R.xi;
        drop(s, R, l, q, lane);
        PROF(7)
        --q;
        if (prep) {
          vip -= t * zz;  // along z the violation of the entering row falls at the rate a^T z = -||d2||^2 (what t2 = vip / zz uses)
        } else {
          Base::states(s, R, lane, N);
          vip = Base::resid(s, c, ip, N);
        }
        if (f >= f_cut) {
          rc = GI_CUTOFF;
          if (lane == 0) s.inf_id = ip;  // (gi_run turns a cut on the box bound into a proof of infeasibility: the row on its way in)
          stop = true;
          break;
        }
      }
      if (stop) break;
      if (f >= f_cut) {
        rc = GI_CUTOFF;
        if (lane == 0) s.inf_id = ip;
        break;
      }
    }
#ifdef HDSM_TRACE_GI
    if (lane == 0) printf("  [blk %d] run ends rc=%d iters=%d q=%d f=%.9g f_cut=%.9g f_box=%.9g\n", (int)blockIdx.x, rc, iters, q, f, f_cut, s.f_box);
#endif
    store_pos(s, R, lane);
    wsync();
    if (pending) __syncthreads();  // (a pick nobody needs any more: its "done")
    if (lane == 0) s.f = f, s.q = q, s.neq_done = neq, s.cmd = 0;
    if (blockDim.x > 64) __syncthreads();  // releases the other waves (they leave on cmd == 0)
    else wsync();
    return rc;
  }

  // The other waves of the workgroup while wave 0 iterates. Wave 1, the scanner, finds the row that enters next; the other waves
  // only keep the barrier count. Commands (Shm::cmd at a "go" barrier):
  //   CMD_PICK   exact: trajectory at s.x, then the pick (state boxes only when nothing else is violated, cold rows promoted when
  //              no hot row is) -> part_id[1] (-1: nothing is violated), part_v[1], part_key[1], and the dense normal of the pick,
  //              entry `var` in dvz[var] (dvz is otherwise unused by this layout); one "done" barrier;
  //   CMD_PREP   (B1) wave 0 has the primal direction z (Shm::w) but not yet the step. Phase 1: trajectory at the OLD iterate and
  //              its derivative along z, the violation v and its rate dv for the lane's staged rows and its input box — all of it
  //              while wave 0 forms r = U d and runs the ratio test. Then B2 with the step in part_v[0]:
  //     t >= 0     a full step t is taken and the row enters. Phase 2: v + t dv for every row -> the pick as above
  //                (-2 instead of -1: "nothing violated" is only ever believed from the exact evaluation); "done" barrier;
  //     t = -1     no step / a partial step (wave 0 follows the entering row by itself): nothing more to do.
  // `R` holds this wave's per-lane constants of the scan (init_lane with lane = thread & 63).
  //
  // The scanner's answer is on the critical path of every operation (wave 0 needs it after its Householder update), and what it
  // costs is LDS round trips in a chain. Whatever can be computed before the step length is known is computed then (phase 1 has
  // the time wave 0 spends on r = U d and the ratio test): after B2 a row costs one multiply-add, and only rows beyond the RC per
  // lane that phase 1 prepares, or the rows of assigned polyhedra, are evaluated from LDS at the new point.
  static constexpr int RC = 3;  // staged rows per lane prepared in phase 1 (64 RC rows; longer staging areas: the rest from LDS)
  static __device__ __forceinline__ void helper_loop(S& s, const Consts& c, Regs& R) {
    const int w = (int)HDSM_TX >> 6;
    if (w != 1) {
      for (;;) {
        __syncthreads();  // go / B1
        const int cmd = uni(s.cmd);
        if (cmd == 0) return;
        __syncthreads();  // done / B2
        if (cmd == CMD_PREP && uni(__double2hiint(s.part_v[0])) >= 0) __syncthreads();  // done
      }
    }
    const int N = c.N, n = c.n;
    const double tol = c.tol;
    const bool norm = c.pick_rule != 0;
    SC_PROF_DECL
    for (;;) {
      __syncthreads();  // go / B1
      SC_PROF(23)
      const int cmd = uni(s.cmd);
      if (cmd == 0) return;
      int lane = (int)HDSM_TX & 63;
      keep_in_loop(lane);
      // this lane's trajectory point: (axis, step m), half h of the impulse-response taps
      // (NVT = 32: the two lanes of a row take half of the taps each; NVT = 48: one lane per point, all taps)
      constexpr int HH = SPLIT ? Base::HT / 2 : Base::HT;
      const int row = SPLIT ? (lane & 31) : lane, h = SPLIT ? (lane >> 5) : 0;
      const bool on = row < 3 * N;
      const int ax = on ? R.ax : 0, m = on ? R.kk + 1 : 1;
      const double* gp = &s.gz[ax][0][MAXH + m - 1 - h * HH];  // taps gp[-k]
      const double* xx = s.x + ax * N + h * HH;
      const double* zx = s.w + ax * N + h * HH;
      double* dpv = s.red_v;  // derivative of the positions along z: entry 3 m + axis (red_v is free during a run)
      int ip = -1;
      double vip = 0.0, kip = 0.0;
      bool answered = false;
      if (cmd == CMD_PREP) {
        // ---- phase 1: everything that needs x and z but not the step
        const int nc = uni(s.ncand), level = uni(s.level) | uni(s.forced);  // (> 0: rows of assigned polyhedra exist)
        double xk[HH], zk[HH], gk[HH];
#pragma unroll
        for (int k = 0; k < HH; ++k) xk[k] = xx[k], zk[k] = zx[k], gk[k] = gp[-k];
        const double xi = lane < NV ? s.x[lane] : 0.0, zl = lane < NV ? s.w[lane] : 0.0;
        const double lbu = s.bnd[R.ax], ubu = s.bnd[3 + R.ax];
        int rm[RC];
        float rw[RC];
        double r0[RC], r1[RC], r2[RC], r3[RC];
#pragma unroll
        for (int u = 0; u < RC; ++u) {  // the lane's staged rows (a row that is not there is never violated)
          const int idx = 64 * u + lane;
          r0[u] = r1[u] = r2[u] = 0.0, r3[u] = DINF, rm[u] = 1, rw[u] = 0.0f;
          if (idx < nc) {
            const MW mw = s.cand_mw[idx];
            const D2 a01 = *reinterpret_cast<const D2*>(&s.cand[idx][0]), a23 = *reinterpret_cast<const D2*>(&s.cand[idx][2]);
            r0[u] = a01.x, r1[u] = a01.y, r2[u] = a23.x, r3[u] = a23.y;
            rm[u] = mw.m, rw[u] = mw.w;
          }
        }
        double accx = h == 0 ? s.fr[ax][m][0] : 0.0, accz = 0.0;
#pragma unroll
        for (int k = 0; k < HH; ++k) accx += gk[k] * xk[k], accz += gk[k] * zk[k];
        accx = Base::hsum(accx), accz = Base::hsum(accz);
        if (on && h == 0) s.st[m][ax] = accx, dpv[3 * m + ax] = accz;
        wsync();
        double v0[RC], dv[RC];
        {
          double px[RC], py[RC], pz[RC], qx[RC], qy[RC], qz[RC];
#pragma unroll
          for (int u = 0; u < RC; ++u) {
            const double* pm = s.st[rm[u]];
            const double* dm = dpv + 3 * rm[u];
            px[u] = pm[0], py[u] = pm[1], pz[u] = pm[2], qx[u] = dm[0], qy[u] = dm[1], qz[u] = dm[2];
          }
#pragma unroll
          for (int u = 0; u < RC; ++u) {
            v0[u] = r0[u] * px[u] + r1[u] * py[u] + r2[u] * pz[u] - r3[u];
            dv[u] = r0[u] * qx[u] + r1[u] * qy[u] + r2[u] * qz[u];
          }
        }
#pragma unroll
        for (int u = 0; u < RC; ++u) keep_here(v0[u]), keep_here(dv[u]);  // (formed BEFORE the barrier, not where they are used)
        SC_PROF(19)
        __syncthreads();  // B2
        SC_PROF(23)
        // ---- phase 2: the step (one word: t >= 0, or -1 = no step / a partial one that wave 0 follows by itself)
        const double t_in = s.part_v[0];
        const int t_hi = uni(__double2hiint(t_in));
        if (t_hi < 0) continue;
        const double t = __hiloint2double(t_hi, uni(__double2loint(t_in)));
        typename Base::Pick pk{0.0, 0.0, -1};
        auto offer = [&](double vv, float wgt, int id) {
          if (vv > tol) {
            const double key = norm ? (double)((float)vv * wgt) : vv;
            if (key > pk.key) pk.key = key, pk.v = vv, pk.id = id;
          }
        };
        if (lane < n) {  // box of this lane's input
          const double xn = xi + t * zl;
          const double vu = xn - ubu, vl = lbu - xn;
          offer(vl > vu ? vl : vu, R.wu, mk_id(K_U, (lane << 1) | (vl > vu ? 1 : 0)));
        }
#pragma unroll
        for (int u = 0; u < RC; ++u) offer(v0[u] + t * dv[u], rw[u], mk_kc(64 * u + lane, rm[u]));
        if (nc > 64 * RC || level > 0) {  // rows that are read from LDS need the new positions there
          if (on && h == 0) s.st[m][ax] = accx + t * accz;
          wsync();
          if (nc > 64 * RC) Base::scan_rows(s, 64 * RC, nc, lane, tol, norm, pk);
          if (level > 0) Base::scan_assigned(s, lane, N, tol, norm, pk, c.pinned_steps);  // rows of the polyhedra assigned on the current branch
        }
        SC_PROF(20)
        const double mk = wave_max64(pk.key);
        ip = -2, kip = mk;
        if (mk > 0.0) {
          const int src = __ffsll((long long)__ballot(pk.key == mk && pk.id >= 0)) - 1;
          ip = __builtin_amdgcn_readlane(pk.id, src);
          vip = bcast64(pk.v, src);
        }
        SC_PROF(21)
        if (c.scanner == 2) ip = -2;  // (development switch HDSM_SCANNER=2: every pick through the exact evaluation)
#ifdef HDSM_CHECK_EXTRAP
        {  // debugging aid: the same pick from the exact evaluation at the new iterate (wave 0 wrote it right after B2)
          for (int w8 = 0; w8 < 8; ++w8) __builtin_amdgcn_s_sleep(64);
          wsync();
          R.xi = lane < NV ? s.x[lane] : 0.0;
          int i2;
          double v2, k2 = 0.0;
          Base::template states<1>(s, R, lane, N);
          Base::template select<1, false>(s, c, R, lane, tol, N, v2, i2, &k2);
          i2 = uni(i2);
          const int ipx = ip == -2 ? -1 : ip;
          if (lane == 0 && (i2 != ipx || fabs(v2 - vip) > 1e-9 * (1.0 + fabs(v2))))
            printf("EXTRAP block %d: extrapolated pick %x v %.12e key %.6e | exact pick %x v %.12e key %.6e | t %.6e nc %d level %d\n", (int)blockIdx.x, ip, vip, kip, i2, v2, k2, t, nc, level);
        }
#endif
        answered = true;
      }
      if (!answered) {  // CMD_PICK: the exact evaluation at s.x (first pick of a run; confirmation that nothing is violated)
        R.xi = lane < NV ? s.x[lane] : 0.0;  // (the box of this lane's input)
        for (;;) {
          Base::template states<1>(s, R, lane, N);
          Base::template select<1, false>(s, c, R, lane, tol, N, vip, ip, &kip);
          ip = uni(ip);
          if (ip < 0) {  // nothing but the state boxes is left: velocities and accelerations too
            Base::template states<2>(s, R, lane, N);
            Base::template select<2, false>(s, c, R, lane, tol, N, vip, ip, &kip);
            ip = uni(ip);
          }
          if (ip >= 0 || Base::promote_cold(s, lane, tol) <= 0) break;
        }
      }
      if (ip >= 0 && lane < NV) s.dvz[lane] = Base::normal_entry(s, R, ip, lane, N, n);
      if (lane == 0) s.part_id[1] = ip, s.part_v[1] = vip, s.part_key[1] = kip;
      SC_PROF(22)
      __syncthreads();  // done
      SC_PROF(23)
    }
  }

  // snapshots: J slots from registers (layout [slot][lane]), U rows / multipliers / ids / x from LDS
  static constexpr int SNAP_DOUBLES = Base::SNAP_DOUBLES;
  // (the snapshot buffer is GLOBAL memory, and said to be: through a generic pointer — it comes out of LDS, Shm::snap — every load of
  // a restore was a flat_load that might alias the LDS store next to it, and the compiler waited for each one before the next:
  // 48 memory round trips in a chain, 13 k cycles per restore, a sixth of the time of a deep tree)
#if defined(__HIP_DEVICE_COMPILE__)
  using GBuf = __attribute__((address_space(1))) double*;
  static __device__ __forceinline__ GBuf as_global(double* p) { return (GBuf)p; }
#else
  using GBuf = double*;
  static __device__ __host__ GBuf as_global(double* p) { return p; }
#endif
  // the U part of a snapshot (LDS <-> global), any wavefront: `lane` = lane of that wavefront
  static __device__ __forceinline__ void snapshot_u(S& s, double* buf_generic, bool save, int lane) {
    const GBuf buf = as_global(buf_generic);
    keep_in_loop(lane);
    const int row = row_addr(lane), c0 = Base::col0_of(lane);
    if (Base::row_ok(lane)) {
      if (save) {
#pragma unroll
        for (int j = 0; j < NC; ++j) buf[(NV + c0 + j) * NV + row] = s.U[row * LDT + c0 + j];
      } else {
        double t[NC];  // (every load requested before the first one is stored)
#pragma unroll
        for (int j = 0; j < NC; ++j) t[j] = buf[(NV + c0 + j) * NV + row];
#pragma unroll
        for (int j = 0; j < NC; ++j) s.U[row * LDT + c0 + j] = t[j];
      }
    }
  }
  // WITH_U = false: the U part is taken by another wavefront at the same time (snapshot_u; hdsm_core.h, snapshot_io)
  template <bool WITH_U = true>
  static __device__ __forceinline__ void snapshot(S& s, Regs& R, double* buf_generic, bool save, int lane) {
    const GBuf buf = as_global(buf_generic);
    keep_in_loop(lane);  // (the per-lane offsets of a snapshot are formed when one is taken, not kept alive across the active-set run)
    const int row = row_addr(lane), c0 = Base::col0_of(lane);
    constexpr int JL = SPLIT ? 64 : NV;  // lanes that hold slots of J (NC JL = NV NV doubles)
    if (Base::row_ok(lane)) {
      if (save) {
#pragma unroll
        for (int j = 0; j < NC; ++j) {
          buf[j * JL + lane] = R.Jr[j];
          if constexpr (WITH_U) buf[(NV + c0 + j) * NV + row] = s.U[row * LDT + c0 + j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < NC; ++j) {
          R.Jr[j] = buf[j * JL + lane];
          if constexpr (WITH_U) s.U[row * LDT + c0 + j] = buf[(NV + c0 + j) * NV + row];
        }
      }
    }
    if (lane < NV) {
      if (save) {
        buf[2 * NV * NV + lane] = R.xi;
        buf[(2 * NV + 1) * NV + lane] = s.lam[lane];
        buf[(2 * NV + 2) * NV + lane] = (double)s.act[lane];
      } else {
        s.lam[lane] = buf[(2 * NV + 1) * NV + lane];
        s.act[lane] = (int)buf[(2 * NV + 2) * NV + lane];
        s.x[lane] = buf[2 * NV * NV + lane];
      }
    }
    if (!save && Base::row_ok(lane)) R.xi = buf[2 * NV * NV + row];  // both copies of a row
    if (lane == 0) {
      if (save) {
        buf[(2 * NV + 3) * NV] = s.f;
        buf[(2 * NV + 3) * NV + 1] = (double)s.q;
      } else {
        s.f = buf[(2 * NV + 3) * NV];
        s.q = (int)buf[(2 * NV + 3) * NV + 1];
      }
    }
    wsync();
  }
};

}  // namespace hdsm
