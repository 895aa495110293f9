// corridor_core.h — next row f2: the convex voxel decomposition the reference's corridor generator calls for every seed
// (GenerateSafeCorridor, agent_class.cpp:1236-1447 -> convex_decomp_lib::GetPolyOcta3D, convex_decomp_util/src/convex_decomp.cpp
// :5-376, and its shape-aware variant GetPolyOcta3DNew, :590-1160, with the helpers FindCorners :378-564 and SideIsEmpty
// :577-588; "CD" below). ONE source for the host entry points (corridor_host.cpp) and the device kernel (corridor_kernels.hip):
// fixed-capacity containers, no allocation, no floating-point contraction, so both builds produce the same rows bit for bit.
//
// What the algorithm does (restated; the tables below are DERIVED from the cube's geometry, only the numbering of
// faces and edges is taken over because it fixes the order of the output rows):
//   * a cuboid of voxels grows from the seed, one face per iteration, round robin over (-y, +x, +y, -x, +z, -z)
//     (CD:17-18, 54-55); a face advances by one voxel layer;
//   * the new layer is itself grown in the face's plane from a 2-D seed, side by side (+u, +v, -u, -v round robin),
//     over free voxels that sit on top of voxels already in the polyhedron; voxels next to the polyhedron but not on
//     top of it are carried along as "virtual" cells so that the sides keep their shape (CD:112-200);
//   * where a new layer comes out SHORTER than the previous one on some side, the edge shared with the neighbouring
//     face becomes a chamfer with an integer slope; a small state machine per edge (slope, steps taken on the current
//     stair, which of the two faces is the long direction, whether the slope is final) decides whether later layers
//     are still consistent with ONE plane through that edge — if not, the face stops growing (CD:209-283);
//   * a layer that reaches the full extent of the previous one on a side also extends the neighbouring face's
//     outermost layer (CD:291-301);
//   * the result: one half-space per chamfered edge (normal = slope * long-face normal + other-face normal) and one
//     per face (CD:322-373). Rows are n . x <= n . p (decomp_geometry/polyhedron.h:98-147).
// The shape-aware variant (`variant` = 1) adds, on top of that (CD:590-1160):
//   * a layer that covers less than half of the area it was allowed is skipped for this turn (CD:691-693, 816-826);
//   * a chamfer may only START where there really is an obstacle behind it: the voxels one step beyond the short
//     side (and, for a one-voxel step, beyond the neighbouring face's edge row) must not all be empty (CD:933-975),
//     and a trial growth of one more layer (FindCorners) must confirm the slope (CD:978-1066); during the first
//     round of six turns no chamfer starts at all (CD:868-870);
//   * layers may reach the last voxel of the grid (CD:705-707 tests < dim where the original tests < dim - 1), the
//     chamfer point is placed half a voxel further out (CD:836-848), and an over-long step on an own fixed chamfer
//     ends the edge scan without stopping the face (CD:875-878 lacks the original's valid_border = false).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CD_HD __host__ __device__ inline
#else
#define CD_HD inline
#endif

#if defined(__clang__)
#define CD_UNROLL _Pragma("unroll")  // loops over the four sides of a face: their small arrays must be registers on the device
#else
#define CD_UNROLL
#endif
#if defined(__clang__)
#pragma clang fp contract(off)  // the plane offsets must come out bit-identical on the host and on the device
#endif

namespace hdsm_cd {

constexpr int kOccupied = 100;  // CVX_DCMP_OCC (convex_decomp.hpp:11): values below it are free
constexpr int CELLS = 768;      // capacity of a face's outermost layer (a layer is at most (2 n_it / 6 + 1)^2 voxels plus the
                                // rows neighbouring faces hand over: 435 for n_it = 42, 700 for n_it = 54)
constexpr int RIM = 96;         // capacity of one side of a growing layer (deque with room at both ends: a side is at most 2 n_it / 6 + 1
                                // cells long and grows by as much at either end — 31 + 2 x 31 for n_it = 78, beyond which CELLS overflows first)
constexpr int RIM0 = 32;        // where an empty deque starts inside its buffer

enum { CD_OK = 0, CD_BAD_ARG = -1, CD_CAPACITY = -4, CD_WORK_OVERFLOW = -6 };

struct Cell {
  int x, y, z;
};
CD_HD bool same(Cell a, Cell b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
CD_HD Cell add(Cell a, Cell b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
CD_HD Cell sub(Cell a, Cell b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
CD_HD Cell neg(Cell a) { return {-a.x, -a.y, -a.z}; }
CD_HD int dot(Cell a, Cell b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// outward normals in the order of the output rows (CD:17-18)
CD_HD Cell normal_of(int f) {  // (-y, +x, +y, -x, +z, -z), computed: an indexed table would live in scratch memory on the device
  return Cell{(f == 1) - (f == 3), (f == 2) - (f == 0), (f == 4) - (f == 5)};
}
CD_HD int face_with_normal(Cell n) {
  for (int f = 0; f < 6; ++f)
    if (same(normal_of(f), n)) return f;
  return -1;
}
// the two faces meeting in edge e (numbering of CD:28-31: it fixes the order of the chamfer rows)
CD_HD int edge_face(int e, int which) {
  // {0,1} {0,4} {0,3} {0,5} {1,5} {1,4} {3,4} {3,5} {1,2} {2,4} {2,3} {2,5}, one nibble per edge (no table in scratch memory)
  const unsigned long long first = 0x222133110000ull, second = 0x534254455341ull;
  return (int)(((which ? second : first) >> (4 * e)) & 0xfull);
}

struct Frame {      // in-plane frame of a face and what lies across each of its four sides
  Cell side[4];     // +u, +v, -u, -v
  int face[4];      // neighbouring face across that side
  int edge[4];      // edge shared with that neighbour
  int back[4];      // index, among the NEIGHBOUR's sides, of the direction this face grows in
};

// u = normal of the next lateral face and v = +z for the four lateral faces; (-y, +x) / (-y, -x) for top / bottom
// (CD:33-41). Everything else follows from that.
CD_HD Cell frame_side(int f, int j) {  // side j (+u, +v, -u, -v) of face f
  Cell u, v;
  if (f < 4) u = normal_of((f + 1) % 4), v = Cell{0, 0, 1};
  else u = normal_of(0), v = (f == 4) ? normal_of(1) : normal_of(3);
  return j == 0 ? u : (j == 1 ? v : (j == 2 ? neg(u) : neg(v)));
}
CD_HD void build_frames(Frame fr[6]) {
  for (int f = 0; f < 6; ++f)
    for (int j = 0; j < 4; ++j) fr[f].side[j] = frame_side(f, j);
  for (int f = 0; f < 6; ++f)
    CD_UNROLL
    for (int j = 0; j < 4; ++j) {
      const int g = face_with_normal(fr[f].side[j]);
      fr[f].face[j] = g;
      fr[f].edge[j] = -1;
      for (int e = 0; e < 12; ++e)
        if ((edge_face(e, 0) == f && edge_face(e, 1) == g) || (edge_face(e, 0) == g && edge_face(e, 1) == f)) fr[f].edge[j] = e;
      fr[f].back[j] = -1;
      CD_UNROLL
      for (int k = 0; k < 4; ++k)
        if (same(fr[g].side[k], normal_of(f))) fr[f].back[j] = k;
    }
}

struct Edge {      // Corner3D (convex_decomp.hpp:21-36)
  double pos[3];
  int slope;       // 0 = square edge
  int dir;         // face along which the chamfer runs `slope` voxels per voxel of the other face; -1 = undecided
  int fixed;
  int steps;       // voxels taken on the current stair
};
CD_HD Edge fresh_edge() { return Edge{{0.0, 0.0, 0.0}, 0, -1, 0, 0}; }

// Where a chamfer's point came from (kept beside the edges, written when a layer that starts a chamfer is accepted): the voxel, the
// face that was growing, the neighbouring face and whether the shape-aware half voxel was added. With the anchors of the faces and
// the edges' slopes this is the polyhedron as INTEGERS — its rows can be formed again for another origin of the local grid,
// with the arithmetic a decomposition in that grid would use (PolyStruct, rows_from_structure).
struct EdgeSrc {
  Cell c;
  int f, nbf, extra;
};
// the point of a chamfer's plane, between the layer and the neighbouring face (CD:322-373 / 836-848; written as the reference writes
// it: products first, then left-to-right sums; no contraction)
CD_HD void edge_point(Cell c, Cell up, Cell nb, double res, bool half_more, double pos[3]) {
  const double extra = half_more ? res / 2 : 0.0;  // (CD:836-848 adds res/2 twice)
  const double h = res / 2;
  pos[0] = (((c.x * res - up.x * res / 2) + nb.x * res / 2) + h) + extra;
  pos[1] = (((c.y * res - up.y * res / 2) + nb.y * res / 2) + h) + extra;
  pos[2] = (((c.z * res - up.z * res / 2) + nb.z * res / 2) + h) + extra;
}
// the half-space of chamfered edge e: normal = slope * long-face normal + other-face normal
CD_HD void edge_row(int e, int slope, int dir, const double pos[3], const double origin[3], double r[4]) {
  const int fa = edge_face(e, 0), fb = edge_face(e, 1);
  const int lng = (dir == fa) ? fa : fb, oth = (dir == fa) ? fb : fa;
  const Cell nl = normal_of(lng), no = normal_of(oth);
  const double nrm[3] = {(double)(slope * nl.x + no.x), (double)(slope * nl.y + no.y), (double)(slope * nl.z + no.z)};
  const double p[3] = {pos[0] + origin[0], pos[1] + origin[1], pos[2] + origin[2]};
  r[0] = nrm[0], r[1] = nrm[1], r[2] = nrm[2];
  r[3] = (nrm[0] * p[0] + nrm[1] * p[1]) + nrm[2] * p[2];
}
// the half-space of face f through the outer side of voxel `anchor`
CD_HD void face_row(int f, Cell anchor, double res, const double origin[3], double r[4]) {
  const Cell nf = normal_of(f);
  const double nrm[3] = {(double)nf.x, (double)nf.y, (double)nf.z};
  const double h = res / 2;
  const double p[3] = {((anchor.x * res + nf.x * res / 2) + h) + origin[0], ((anchor.y * res + nf.y * res / 2) + h) + origin[1],
                       ((anchor.z * res + nf.z * res / 2) + h) + origin[2]};
  r[0] = nrm[0], r[1] = nrm[1], r[2] = nrm[2];
  r[3] = (nrm[0] * p[0] + nrm[1] * p[1]) + nrm[2] * p[2];
}
// A finished polyhedron as integers, voxels relative to the seed
struct PolyStruct {
  int32_t slope[12], dir[12], f[12], nbf[12], extra[12];
  Cell c[12];
  Cell anchor[6];
};
// rows[max_rows][4] of a polyhedron with this structure grown from `seed` (a voxel of the local grid whose voxel (0,0,0) lies at
// `origin`): what decompose_core writes for it. Returns the number of rows (may exceed max_rows: CD_CAPACITY for the caller).
CD_HD int rows_from_structure(const PolyStruct& ps, Cell seed, double res, const double origin[3], double* rows, int max_rows) {
  int n = 0;
  for (int e = 0; e < 12; ++e) {
    if (ps.slope[e] <= 0) continue;
    double pos[3];
    edge_point(add(seed, ps.c[e]), normal_of(ps.f[e]), normal_of(ps.nbf[e]), res, ps.extra[e] != 0, pos);
    if (n < max_rows) edge_row(e, ps.slope[e], ps.dir[e], pos, origin, rows + 4 * n);
    ++n;
  }
  for (int f = 0; f < 6; ++f) {
    if (n < max_rows) face_row(f, add(seed, ps.anchor[f]), res, origin, rows + 4 * n);
    ++n;
  }
  return n;
}

// cells are stored as offsets from the seed (a polyhedron never reaches further than n_it / 6 + 1 layers from it)
struct Packed {
  int8_t x, y, z, pad;  // (four bytes: one LDS access per cell)
};

struct CellList {   // std::vector<Vec3i> with a fixed capacity
  int n;
  Packed c[CELLS];
};
struct CellDeque {  // std::deque<Vec3i> with a fixed capacity and room at both ends
  int b, e;
  Packed c[RIM];
};

struct FaceState {  // Border3D (convex_decomp.hpp:39-43)
  CellList outer;   // outermost layer of the face
  int reach[4];     // extent of that layer along the face's four sides (dot products)
};

struct Layer {
  int found;
  CellList cells;       // border_real_tmp
  CellDeque rim_real[4];  // borders_2d_real
  Cell far[4];          // border_limit_tmp
  // cooperative mode: the layer's cells as rows of a bit plane of the overlay, normal axis and level index of that plane
  uint32_t plane[32];
  int plane_axis, plane_level;
};

// everything one decomposition needs; ~30 KB, provided by the caller (heap on the host, global scratch or LDS on the device)
struct Work {
  Frame fr[6];
  FaceState faces[6], face_t;  // face_t: the face under trial with its new layer as outermost layer (CD:978-1066)
  Layer L, L2;
  CellDeque edge_row;
  Edge edges[12], edges_t[12];
  EdgeSrc esrc[12];  // (see EdgeSrc)
  Cell anchor[6];  // a voxel of each face's outermost layer (gives the face plane)
#ifdef CD_PROFILE
  unsigned long long prof[16];  // cycles per phase of this decomposition (lane 0), added to g_cd_prof at its end: probes that hit
                                // global memory themselves made every wait for memory look expensive
#endif
  uint32_t seed_plane[32];  // cooperative mode: where a 2-D seed may lie, rows of a bit plane (one word per lane)
  Cell seed;
  int overflow;
  // the serial form's rim deques, LAST: the cooperative form does not use them, and its LDS layout puts the y-fast copy of the
  // overlay over them (corridor_wave.h, WaveLds)
  CellDeque rim[4], moved, moved_real;
};

// The local voxel grid of one agent as a window into a WORLD grid (what env_builder's GenerateVoxelGridMSG cuts out,
// environment_builder.cpp:58-67): local voxel (i, j, k) = world voxel (i, j, k) + off; voxels below local k = ground_k are
// unknown -> occupied (AC:1302, 1307), unknown (negative) world voxels are occupied, voxels outside the world are free.
// The polyhedron's voxels are kept in a bit overlay around the seed instead of being written into the (shared) world.
struct WindowGrid {
  const int8_t* world;
  int wnx, wny, wnz;
  int ox, oy, oz;      // off
  int lnx, lny, lnz;   // local dimensions
  int ground_k;
  int mark;
  Cell seed;
  uint32_t* bits;      // overlay over offsets [-OV, OV)^3 from the seed: word dy + OVW * dz, bit dx ("x-fast")
  // optional (cooperative mode of the device: built by the whole wavefront in LDS, build_world_maps): bit maps of the world under
  // the overlay, same indexing, for the offsets |d| <= map_r from the seed —
  //   maps       FREE, x-fast: inside the local grid and a value below kOccupied
  //   maps_pos   POS,  x-fast: outside the local grid or a value above 0 (what SideIsEmpty tests, CD:577-588)
  //   maps_t     FREE, y-fast: word dx + OVW * dz, bit dy
  // (each array holds the 2 map_r + 1 z-levels that are classified only; the pointers are moved back by the words of the levels
  // below them, so that the index is the overlay's — never dereferenced outside those levels)
  // and a y-fast copy of the overlay (bits_t). With them a whole plane of voxels is 32 words — one per lane (grow_layer_planes).
  const uint32_t *maps = nullptr, *maps_pos = nullptr, *maps_t = nullptr;
  uint32_t* bits_t = nullptr;
  int map_r = 0;
  static constexpr int OV = 16, OVW = 32, WORDS = OVW * OVW * OVW / 32, MAP_WORDS = 3 * WORDS;
  static constexpr bool kHasPlanes = true;    // maps / bits_t may be there
  CD_HD int nx() const { return lnx; }
  CD_HD int ny() const { return lny; }
  CD_HD int nz() const { return lnz; }
  CD_HD bool inside(Cell c) const { return c.x >= 0 && c.y >= 0 && c.z >= 0 && c.x < lnx && c.y < lny && c.z < lnz; }
  CD_HD int bit_index(Cell c) const {
    const int dx = c.x - seed.x + OV, dy = c.y - seed.y + OV, dz = c.z - seed.z + OV;
    if (dx < 0 || dy < 0 || dz < 0 || dx >= OVW || dy >= OVW || dz >= OVW) return -1;
    return dx + OVW * (dy + OVW * dz);
  }
  CD_HD bool marked(Cell c) const {
    const int b = bit_index(c);
    return b >= 0 && ((bits[b >> 5] >> (b & 31)) & 1u);
  }
  CD_HD int world_value(Cell c) const {  // what lies under the overlay
    if (c.z < ground_k) return kOccupied;
    const int gi = c.x + ox, gj = c.y + oy, gk = c.z + oz;
    if (gi < 0 || gj < 0 || gk < 0 || gi >= wnx || gj >= wny || gk >= wnz) return 0;
    const int v = world[(size_t)gi + (size_t)gj * wnx + (size_t)gk * wnx * wny];
    return v < 0 ? kOccupied : v;
  }
  CD_HD int value(Cell c) const {
    const int dx = c.x - seed.x + OV, dy = c.y - seed.y + OV, dz = c.z - seed.z + OV;
    if (dx >= 0 && dy >= 0 && dz >= 0 && dx < OVW && dy < OVW && dz < OVW) {
      const int w = dy + OVW * dz;
      if ((bits[w] >> dx) & 1u) return mark;
      if (maps && dx >= OV - map_r && dx <= OV + map_r && dy >= OV - map_r && dy <= OV + map_r && dz >= OV - map_r && dz <= OV + map_r) {
        // (the decomposition compares values with kOccupied, with 0 and with the mark only: the two bits are enough; callers test
        // inside() before they look at a value, FREE already holds it)
        if (!((maps[w] >> dx) & 1u)) return kOccupied;
        return (maps_pos[w] >> dx) & 1u;
      }
    }
    return world_value(c);
  }
  CD_HD void set(Cell c, int) {
    const int dx = c.x - seed.x + OV, dy = c.y - seed.y + OV, dz = c.z - seed.z + OV;
    if (dx < 0 || dy < 0 || dz < 0 || dx >= OVW || dy >= OVW || dz >= OVW) return;
    bits[dy + OVW * dz] |= 1u << dx;
    if (bits_t) bits_t[dx + OVW * dz] |= 1u << dy;
  }
  CD_HD void trial_set(Cell c, int m) { set(c, m); }
  CD_HD void trial_unset(Cell c) {
    const int dx = c.x - seed.x + OV, dy = c.y - seed.y + OV, dz = c.z - seed.z + OV;
    if (dx < 0 || dy < 0 || dz < 0 || dx >= OVW || dy >= OVW || dz >= OVW) return;
    bits[dy + OVW * dz] &= ~(1u << dx);
    if (bits_t) bits_t[dx + OVW * dz] &= ~(1u << dy);
  }
#if defined(__HIP_DEVICE_COMPILE__) || defined(CD_EMU_COOP)
  // cooperative mode: the lanes of a wavefront mark different voxels of a layer at once (decompose_core, mark_cells) —
  // one copy only: the x-fast one (bits) or the y-fast one (bits_t)
  __device__ void mark_atomic(Cell c, bool on, bool x_fast) const {
    const int dx = c.x - seed.x + OV, dy = c.y - seed.y + OV, dz = c.z - seed.z + OV;
    if (dx < 0 || dy < 0 || dz < 0 || dx >= OVW || dy >= OVW || dz >= OVW) return;
    uint32_t* w = x_fast ? &bits[dy + OVW * dz] : &bits_t[dx + OVW * dz];
    const uint32_t b = 1u << (x_fast ? dx : dy);
    if (on) atomicOr(w, b);
    else atomicAnd(w, ~b);
  }
#endif
  CD_HD int count() const {
    int n = 0;
    for (int w = 0; w < WORDS; ++w) {
      uint32_t v = bits[w];
      while (v) v &= v - 1, ++n;
    }
    return n;
  }
};

// Cooperative mode (device, corridor kernel of the swarm loop): ALL 64 lanes of a wavefront run the decomposition with the same
// data — the control flow is uniform and every lane stores the same values — and the loops over the cells of a layer / a rim
// (where one lane spent its time: a chain of LDS round trips per cell) are spread over the lanes, with ballots where the serial
// loop stops at the first hit or appends in order. The results are those of the serial code, statement for statement.
#if (defined(__HIP_DEVICE_COMPILE__) || defined(CD_EMU_COOP)) && !defined(CD_NO_COOP)
#define CD_HAS_COOP 1
#define CD_COOP(cx) ((cx).coop)
#define CD_SYNC() __syncthreads()
#else
#define CD_HAS_COOP 0
#define CD_COOP(cx) false
#define CD_SYNC()
#endif
// grow_layer and find_corners are real calls on the device. Inlined everywhere (three copies of grow_layer in one kernel),
// hipcc -O3 produced gfx950 code for the one-thread-per-seed kernel that faulted on some seeds (memory aperture violation;
// fine at -O2, with -fno-unroll-loops or -fno-vectorize, with either function out of line, and the same source is clean on the
// host under ASan / UBSan). The calls also keep the kernels a third of the size.
// The COOPERATIVE instantiations (template parameter COOP) are inlined into their kernel instead: only then does the compiler
// see that the workspace is LDS and the small arrays are registers — behind a call they were flat loads, and the allowance of a
// layer a scratch (global-memory) load per rim move.
#if defined(__HIP_DEVICE_COMPILE__)
#define CD_NOINLINE __attribute__((noinline))
#define CD_INLINE __attribute__((always_inline))
#else
#define CD_NOINLINE
#define CD_INLINE
#endif

// -DCD_PROFILE (development builds, scripts/gpu_corridor_profile.sh): cycles per phase of the cooperative decomposition,
// accumulated by lane 0 into a device array that hdsm_corridor_profile() reads
#if defined(CD_PROFILE) && defined(__HIPCC__)
static __device__ unsigned long long g_cd_prof[16];
#endif
#if defined(CD_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
#define CD_PROF_BEGIN() unsigned long long cd_t0_ = __builtin_readcyclecounter()
#define CD_PROF(i)                                                  \
  do {                                                              \
    const unsigned long long cd_t1_ = __builtin_readcyclecounter(); \
    if (cx.lane == 0) cx.wk->prof[i] += cd_t1_ - cd_t0_;            \
    cd_t0_ = cd_t1_;                                                \
  } while (0)
#else
#define CD_PROF_BEGIN() ((void)0)
#define CD_PROF(i) ((void)0)
#endif

struct Ctx {
  Work* wk;
  bool coop = false;  // device only: see above
  int lane = 0;
  Cell seed{0, 0, 0};  // = wk->seed, by value (every pack / unpack would re-read it from the workspace otherwise)
  CD_HD Packed pack(Cell c) const {
    const int dx = c.x - seed.x, dy = c.y - seed.y, dz = c.z - seed.z;
    if (dx < -127 || dx > 127 || dy < -127 || dy > 127 || dz < -127 || dz > 127) wk->overflow = 1;
    return Packed{(int8_t)dx, (int8_t)dy, (int8_t)dz, 0};
  }
  CD_HD Cell unpack(Packed p) const { return Cell{seed.x + p.x, seed.y + p.y, seed.z + p.z}; }
  CD_HD void push(CellList& l, Cell c) const {
    if (l.n < CELLS) l.c[l.n++] = pack(c);
    else wk->overflow = 1;
  }
  CD_HD Cell at(const CellList& l, int i) const { return unpack(l.c[i]); }
  CD_HD void clear(CellDeque& d) const { d.b = d.e = RIM0; }
  CD_HD void assign1(CellDeque& d, Cell c) const { d.b = RIM0, d.e = RIM0 + 1, d.c[RIM0] = pack(c); }
  CD_HD bool empty(const CellDeque& d) const { return d.b == d.e; }
  CD_HD int size(const CellDeque& d) const { return d.e - d.b; }
  CD_HD Cell get(const CellDeque& d, int i) const { return unpack(d.c[d.b + i]); }
  CD_HD Cell front(const CellDeque& d) const { return unpack(d.c[d.b]); }
  CD_HD Cell back(const CellDeque& d) const { return unpack(d.c[d.e - 1]); }
  CD_HD void push_back(CellDeque& d, Cell c) const {
    if (d.e < RIM) d.c[d.e++] = pack(c);
    else wk->overflow = 1;
  }
  CD_HD void push_front(CellDeque& d, Cell c) const {
    if (d.b > 0) d.c[--d.b] = pack(c);
    else wk->overflow = 1;
  }
  CD_HD void copy(CellDeque& dst, const CellDeque& src) const {
    const int b = src.b, e = src.e;
    dst.b = b, dst.e = e;
    if (CD_COOP(*this)) {
      for (int i = b + lane; i < e; i += 64) dst.c[i] = src.c[i];
      CD_SYNC();
      return;
    }
    for (int i = b; i < e; ++i) dst.c[i] = src.c[i];
  }
  CD_HD void copy(CellList& dst, const CellList& src) const {
    const int n = src.n;
    dst.n = n;
    if (CD_COOP(*this)) {
      for (int i = lane; i < n; i += 64) dst.c[i] = src.c[i];
      CD_SYNC();
      return;
    }
    for (int i = 0; i < n; ++i) dst.c[i] = src.c[i];
  }
  // l += the cells of d, in order (push() for each)
  CD_HD void append(CellList& l, const CellDeque& d) const {
    const int cnt = d.e - d.b;
    if (CD_COOP(*this)) {
      const int n0 = l.n;
      for (int q = lane; q < cnt; q += 64)
        if (n0 + q < CELLS) l.c[n0 + q] = d.c[d.b + q];
      if (n0 + cnt > CELLS) wk->overflow = 1;
      CD_SYNC();
      l.n = n0 + cnt > CELLS ? CELLS : n0 + cnt;
      CD_SYNC();
      return;
    }
    for (int q = 0; q < cnt; ++q) push(l, get(d, q));
  }
};

// how far the next layer of face f may extend on each side, given the chamfers already started (CD:71-91)
// (the edges' state is read where it lies; the working copies `trial` of the four edges are taken by the caller AFTER the layer
// has been grown — 40 words that would otherwise be live, in registers, across the whole in-plane growth)
// (`fv`: the frame of face f BY VALUE — the caller reads it once per turn, all twenty-four words together, instead of a chain of
// dependent LDS reads frame -> edge index -> edge at every use)
CD_HD void allowance(const Frame& fv, int f, const FaceState& fs, const Edge* edges, int allow[4]) {
  CD_UNROLL
  for (int j = 0; j < 4; ++j) {
    allow[j] = fs.reach[j];
    const Edge& e = edges[fv.edge[j]];
    if (e.slope > 0) {
      if (e.dir == f) allow[j] -= e.slope;                         // our layers retreat `slope` voxels each
      else if (e.fixed && e.steps >= e.slope) allow[j] -= 1;       // the other face's stair is complete: step in
    }
  }
}
CD_HD void edges_of_face(const Frame& fv, const Edge* edges, Edge trial[4]) {
  CD_UNROLL
  for (int j = 0; j < 4; ++j) trial[j] = edges[fv.edge[j]];
}

#if CD_HAS_COOP
// One layer on top of face F in the cooperative mode, WITHOUT the cell deques of grow_layer_serial below. What that function's
// loop maintains is a rectangle: a side only ever moves as a whole line, its cells are the rectangle's edge in counter-clockwise
// order (+u: along +v, +v: along -u, -u: along -v, -v: along +u), and a line may move iff it stays inside the allowance and no
// cell of it lies on top of the polyhedron without being free — a failure is final, the line only gets longer. So: lane r < 32
// holds row r of two bit planes of the overlay's 32 x 32 cross-section at the layer's level (REAL: on top of the polyhedron and
// free; BLK: on top of it and not free — three LDS words per lane, from the x-fast or y-fast copies so that the plane's rows
// are words), a move along the bit axis is a ballot of one bit column, a move along the row axis a v_readlane of one row, both
// masked to the rectangle; the layer's cells (the REAL cells of every line that moved, in the deque's order) are written by
// the lanes that hold them, ranked by a popcount. At the end the four rim_real deques and `far` are written out as the serial
// loop leaves them — every cell of rim_real[j] is a REAL cell of side j's final line and vice versa, far[j] is read only
// through its coordinate along side j — so the caller goes on unchanged. The 2-D seed (the first cell of the face's outer list
// whose neighbour above qualifies) is looked up in the same plane: free, inside the allowance, inside the grid's margin.
// F = the face, a template parameter: normals, axes and signs of the four sides are constants of each instantiation (with a
// run-time face every rim move dragged select chains over them along; one wavefront per CU issues an instruction every 4-5
// cycles, so the instruction count IS the time).
template <class G, int F>
CD_INLINE __device__ inline bool grow_layer_wave_f(const Ctx& cx, const G& g, const FaceState& fs, const int allow_in[4], int margin, Layer& L) {
  Work& wk = *cx.wk;
  constexpr int f = F;
  constexpr int OV = G::OV, OVW = G::OVW;
  const int allow[4] = {allow_in[0], allow_in[1], allow_in[2], allow_in[3]};
  const int lane = cx.lane;
  const Cell up = normal_of(f);
  const Cell sd[4] = {frame_side(f, 0), frame_side(f, 1), frame_side(f, 2), frame_side(f, 3)};
  const int aw = up.x ? 0 : (up.y ? 1 : 2);          // axis of the face's normal
  const int ab = aw == 0 ? 1 : 0, ar = aw == 2 ? 1 : 2;  // axis along the bits of a row / across the rows
  auto comp = [](Cell c, int a) { return a == 0 ? c.x : (a == 1 ? c.y : c.z); };
  const int su = up.x + up.y + up.z;
  const int sb = comp(cx.seed, ab), sr = comp(cx.seed, ar), sw = comp(cx.seed, aw);
  const int lo_ok = OV - g.map_r, hi_ok = OV + g.map_r;  // what build_world_maps classified
  bool bit_axis[4];
  int sg[4];
  CD_UNROLL
  for (int j = 0; j < 4; ++j) bit_axis[j] = comp(sd[j], ab) != 0, sg[j] = comp(sd[j], ab) + comp(sd[j], ar);
  auto range_mask = [](int lo, int hi) { return lo > hi ? 0u : (uint32_t)(((1ull << (hi + 1)) - 1ull) & ~((1ull << lo) - 1ull)); };
  const uint32_t* mk = aw == 0 ? g.bits_t : g.bits;
  const uint32_t* fr = aw == 0 ? g.maps_t : g.maps;
  uint32_t real = 0, blk = 0, free_t = 0;
  auto load_rows = [&](int lw) {  // the planes at level lw (and the level below it)
    real = blk = free_t = 0;
    if (lane < OVW && (aw == 2 || (lane >= lo_ok && lane <= hi_ok))) {  // (aw != 2: the lane is the row's z-level — only classified levels exist)
      const int lb = lw - su;
      const int wt = aw == 2 ? lane + OVW * lw : lw + OVW * lane, wb = aw == 2 ? lane + OVW * lb : lb + OVW * lane;
      const uint32_t below = mk[wb];
      free_t = mk[wt] | fr[wt];  // (a voxel of the polyhedron counts as free: its value is the mark)
      real = below & free_t, blk = below & ~free_t;
    }
  };
  CD_PROF_BEGIN();
  // ---- the 2-D seed
  const int n_outer = fs.outer.n;
  const Cell c0 = add(cx.at(fs.outer, 0), up);
  int lw = comp(c0, aw) - sw + OV;
  bool plane_ok = lw - su >= lo_ok && lw - su <= hi_ok && lw >= lo_ok && lw <= hi_ok;
  if (plane_ok) {
    load_rows(lw);
    // rows / bits a seed may lie in: voxels [1, dim - 1 - margin] (CD:94-116 / 700-723), the allowance, the classified box
    int lim_lo[2] = {1, 1}, lim_hi[2] = {(ab == 0 ? g.nx() : g.ny()) - margin - 1, (ar == 1 ? g.ny() : g.nz()) - margin - 1};
    CD_UNROLL
    for (int j = 0; j < 4; ++j) {
      const int a = bit_axis[j] ? 0 : 1;
      if (sg[j] > 0) lim_hi[a] = lim_hi[a] < allow[j] ? lim_hi[a] : allow[j];
      else lim_lo[a] = lim_lo[a] > -allow[j] ? lim_lo[a] : -allow[j];
    }
    const int xw = sw + lw - OV, nw = aw == 0 ? g.nx() : (aw == 1 ? g.ny() : g.nz());
    const bool level_in = xw >= 1 && xw < nw - margin;
    const int ib_lo = lim_lo[0] - sb + OV, ib_hi = lim_hi[0] - sb + OV, ir_lo = lim_lo[1] - sr + OV, ir_hi = lim_hi[1] - sr + OV;
    const uint32_t bits_ok = range_mask(ib_lo > lo_ok ? ib_lo : lo_ok, ib_hi < hi_ok ? ib_hi : hi_ok);
    const bool row_ok = level_in && lane >= ir_lo && lane <= ir_hi;
    if (lane < OVW) wk.seed_plane[lane] = row_ok ? (free_t & bits_ok) : 0u;
    CD_SYNC();
  }
  auto seed_ok = [&](Cell t) {  // the general form (grow_layer_serial's), for cells outside the plane that was loaded
    if (t.x < 1 || t.y < 1 || t.z < 1 || t.x >= g.nx() - margin || t.y >= g.ny() - margin || t.z >= g.nz() - margin) return false;
    if (g.value(t) >= kOccupied) return false;
    bool in = true;
    CD_UNROLL
    for (int k = 0; k < 4; ++k) in = in && dot(t, sd[k]) <= allow[k];
    return in;
  };
  bool found = false;
  Cell s2{0, 0, 0};
  for (int base = 0; base < n_outer && !found; base += 64) {
    const int q = base + lane;
    bool hit = false;
    if (q < n_outer) {
      const Cell t = add(cx.at(fs.outer, q), up);
      const int iw = comp(t, aw) - sw + OV, ib = comp(t, ab) - sb + OV, ir = comp(t, ar) - sr + OV;
      if (plane_ok && iw == lw && ib >= lo_ok && ib <= hi_ok && ir >= lo_ok && ir <= hi_ok) hit = (wk.seed_plane[ir] >> ib) & 1u;
      else hit = seed_ok(t);
    }
    const unsigned long long m = __ballot(hit);
    if (m) s2 = add(cx.at(fs.outer, base + __ffsll((long long)m) - 1), up), found = true;
  }
  CD_PROF(0);
  if (!found) return false;
  // ---- the growth
  int b0 = comp(s2, ab) - sb + OV, r0 = comp(s2, ar) - sr + OV, b1 = b0, r1 = r0;
  L.cells.c[0] = cx.pack(s2);
  {
    const int lw2 = comp(s2, aw) - sw + OV;
    if (lw2 - su < lo_ok || lw2 - su > hi_ok || lw2 < lo_ok || lw2 > hi_ok || b0 < lo_ok || b0 > hi_ok || r0 < lo_ok || r0 > hi_ok) {
      wk.overflow = 1;  // (beyond what build_world_maps classified)
      L.cells.n = 1;
      return true;
    }
    if (!plane_ok || lw2 != lw) lw = lw2, load_rows(lw);
  }
  int farc[4];
  CD_UNROLL
  for (int j = 0; j < 4; ++j) farc[j] = dot(s2, sd[j]);
  int n = 1;
  auto line_of = [&](bool along_bits, int at, uint32_t plane, int lo, int hi) {  // one line of a plane, indexed along the OTHER axis
    const uint32_t m = along_bits ? (uint32_t)__ballot(lane < OVW && ((plane >> at) & 1u)) : (uint32_t)__builtin_amdgcn_readlane((int)plane, at);
    return m & range_mask(lo, hi);
  };
  auto write_line = [&](Packed* dst, int cap, bool along_bits, int at, uint32_t m, bool ascending) {
    if (lane < OVW && ((m >> lane) & 1u)) {
      const int rank = ascending ? __builtin_popcount(m & ((1u << lane) - 1u)) : __builtin_popcountll((unsigned long long)m >> (lane + 1));
      int d[3];
      d[aw] = lw - OV, d[ab] = (along_bits ? at : lane) - OV, d[ar] = (along_bits ? lane : at) - OV;
      if (rank < cap) dst[rank] = Packed{(int8_t)d[0], (int8_t)d[1], (int8_t)d[2], 0};
      else wk.overflow = 1;
    }
  };
  CD_PROF(1);
  // The rim moves, a BATCH at a time instead of one after the other. While no line is blocked the round robin is a fixed schedule:
  // side s makes its t-th move in turn t, its line lies at bound_s +- t and spans what the two perpendicular sides have reached by
  // then (each min(turns so far, what the allowance leaves it)). So lane l = (turn l / 4 + 1, place l % 4 in the round robin)
  // forms ITS move's line in closed form, fetches the line's word of the two planes from the lane that holds it (rows as they are,
  // columns transposed once per layer), and tests it. The schedule is valid up to the first blocked move F (a ballot): the moves
  // before F are committed together — their cells go to the layer's list at the prefix sum of the lines' counts, in the deque's
  // order — the side of move F is dead, and the round robin continues behind it with a new batch. At most 4 sides die, a batch
  // holds 16 turns: 1 - 5 batches per layer instead of ~ 60 sequential moves of ~ 90 instructions.
  // `far` (read only through its coordinate along its side): side j's position the last time its line had cells on top of the
  // polyhedron when looked at after a move. While a side stays where it is its line only grows, so it is enough to look at every
  // position once, when the side leaves it (with the span of the move that leaves: the lane of that move does it) or at the end
  // of a batch.
#ifdef CD_CHECK_BATCH
  // (test builds of tests/wave_emu only: the moves one after the other, as rounds 5's first version made them, for comparison)
  int ck_n = 1, ck_b0 = b0, ck_b1 = b1, ck_r0 = r0, ck_r1 = r1, ck_farc[4];
  Packed ck_cells[CELLS];
  bool ck_mine[CELLS] = {false};
  {
    bool alive_[4] = {true, true, true, true}, nonempty[4] = {true, true, true, true};
    for (int j = 0; j < 4; ++j) ck_farc[j] = farc[j];
    while (alive_[0] || alive_[1] || alive_[2] || alive_[3]) {
      for (int s = 0; s < 4; ++s) {
        if (!alive_[s]) continue;
        const int nxt = (s + 1) & 3, prv = (s + 3) & 3;
        const bool ba = bit_axis[s];
        const int at = ba ? (sg[s] > 0 ? ck_b1 + 1 : ck_b0 - 1) : (sg[s] > 0 ? ck_r1 + 1 : ck_r0 - 1);
        const int lo = ba ? ck_r0 : ck_b0, hi = ba ? ck_r1 : ck_b1;
        bool ok = sg[s] * ((ba ? sb : sr) + at - OV) <= allow[s] && at >= lo_ok && at <= hi_ok;
        uint32_t lr = 0;
        if (ok) {
          ok = line_of(ba, at, blk, lo, hi) == 0u;
          lr = line_of(ba, at, real, lo, hi);
        }
        if (!ok) {
          alive_[s] = false;
          continue;
        }
        const bool asc = sg[nxt] > 0;
        if (lane < OVW && ((lr >> lane) & 1u)) {
          const int rank = asc ? __builtin_popcount(lr & ((1u << lane) - 1u)) : __builtin_popcountll((unsigned long long)lr >> (lane + 1));
          int d[3];
          d[aw] = lw - OV, d[ab] = (ba ? at : lane) - OV, d[ar] = (ba ? lane : at) - OV;
          if (ck_n + rank < CELLS) ck_cells[ck_n + rank] = Packed{(int8_t)d[0], (int8_t)d[1], (int8_t)d[2], 0}, ck_mine[ck_n + rank] = true;
        }
        ck_n += __builtin_popcount(lr);
        if (ba) (sg[s] > 0 ? ck_b1 : ck_b0) = at;
        else (sg[s] > 0 ? ck_r1 : ck_r0) = at;
        nonempty[s] = lr != 0u;
        if ((lr >> (asc ? lo : hi)) & 1u) nonempty[prv] = true;
        if ((lr >> (asc ? hi : lo)) & 1u) nonempty[nxt] = true;
        for (int j = 0; j < 4; ++j)
          if (nonempty[j]) ck_farc[j] = sg[j] * ((bit_axis[j] ? sb : sr) + (bit_axis[j] ? (sg[j] > 0 ? ck_b1 : ck_b0) : (sg[j] > 0 ? ck_r1 : ck_r0)) - OV);
      }
    }
  }
#endif
  uint32_t col_real = 0, col_blk = 0;  // lane b: column b of the planes (bits over the rows)
  int lim[4];                          // the last index side j may reach (allowance and the classified box)
  CD_UNROLL
  for (int j = 0; j < 4; ++j) {
    const int a = sg[j] * allow[j] - (bit_axis[j] ? sb : sr) + OV;  // sg (seed + index - OV) <= allow
    lim[j] = sg[j] > 0 ? (a < hi_ok ? a : hi_ok) : (a > lo_ok ? a : lo_ok);
    if (sg[j] > 0 ? a > hi_ok : a < lo_ok) wk.overflow = 1;  // (the allowance reaches beyond what build_world_maps classified)
  }
  {
    int c_lo = OVW, c_hi = -1;
    CD_UNROLL
    for (int j = 0; j < 4; ++j)
      if (bit_axis[j]) (sg[j] > 0 ? c_hi : c_lo) = lim[j];
    for (int b = c_lo < b0 ? c_lo : b0; b <= (c_hi > b0 ? c_hi : b0); ++b) {
      const uint32_t mr = (uint32_t)__ballot(lane < OVW && ((real >> b) & 1u)), mb = (uint32_t)__ballot(lane < OVW && ((blk >> b) & 1u));
      if (lane == b) col_real = mr, col_blk = mb;
    }
  }
  auto pick4 = [](int j, int a0, int a1, int a2, int a3) { return j == 0 ? a0 : (j == 1 ? a1 : (j == 2 ? a2 : a3)); };
  unsigned alive = 0xfu;
  int s_next = 0;
  for (;;) {
    int bnd[4], cap[4];
    CD_UNROLL
    for (int j = 0; j < 4; ++j) {
      bnd[j] = bit_axis[j] ? (sg[j] > 0 ? b1 : b0) : (sg[j] > 0 ? r1 : r0);
      const int left = sg[j] > 0 ? lim[j] - bnd[j] : bnd[j] - lim[j];
      cap[j] = ((alive >> j) & 1u) && left > 0 ? left : 0;
    }
    if ((cap[0] | cap[1] | cap[2] | cap[3]) == 0) break;
    const int t = (lane >> 2) + 1, pos = lane & 3, s = (s_next + pos) & 3, q1 = (s + 1) & 3, q3 = (s + 3) & 3;
    const bool ba = pick4(s, bit_axis[0], bit_axis[1], bit_axis[2], bit_axis[3]) != 0;
    const int sgs = pick4(s, sg[0], sg[1], sg[2], sg[3]);
#ifndef CD_BATCH_TURNS
#define CD_BATCH_TURNS 16  // turns per batch: the 64 lanes (test builds: fewer, so that layers continue over several batches)
#endif
    const bool active = t <= pick4(s, cap[0], cap[1], cap[2], cap[3]) && t <= CD_BATCH_TURNS;
    const int at = pick4(s, bnd[0], bnd[1], bnd[2], bnd[3]) + sgs * t;
    auto reached = [&](int q) {  // moves side q has made when this lane's move is made
      const int tt = ((q - s_next) & 3) < pos ? t : t - 1, cq = pick4(q, cap[0], cap[1], cap[2], cap[3]);
      return tt < cq ? tt : cq;
    };
    const int e1 = pick4(q1, bnd[0], bnd[1], bnd[2], bnd[3]) + pick4(q1, sg[0], sg[1], sg[2], sg[3]) * reached(q1);
    const int e3 = pick4(q3, bnd[0], bnd[1], bnd[2], bnd[3]) + pick4(q3, sg[0], sg[1], sg[2], sg[3]) * reached(q3);
    const int lo = e1 < e3 ? e1 : e3, hi = e1 < e3 ? e3 : e1;
    const uint32_t span = range_mask(lo, hi);
    const int src = active ? at : 0, src_prev = active ? at - sgs : 0;
    // (every lane takes part in every shuffle: both orientations are fetched, then one is kept)
    const uint32_t cr = (uint32_t)__shfl((int)col_real, src), rr = (uint32_t)__shfl((int)real, src);
    const uint32_t cb = (uint32_t)__shfl((int)col_blk, src), rb = (uint32_t)__shfl((int)blk, src);
    const uint32_t cp = (uint32_t)__shfl((int)col_real, src_prev), rp = (uint32_t)__shfl((int)real, src_prev);
    const uint32_t w_real = ba ? cr : rr, w_blk = ba ? cb : rb, w_prev = ba ? cp : rp;
    const uint32_t lr = w_real & span;
    const unsigned long long blocked = __ballot(active && (w_blk & span) != 0u);
    const int first_blocked = blocked ? __ffsll((long long)blocked) - 1 : 64;
    const bool com = active && lane < first_blocked;
    const int cnt = com ? __builtin_popcount(lr) : 0;
    int incl = cnt;  // inclusive prefix sum over the lanes
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d);
      if (lane >= d) incl += o;
    }
    const int total = __shfl(incl, 63);
    if (com) {  // the line's cells in the deque's order: along side s + 1
      const bool asc = pick4(q1, sg[0], sg[1], sg[2], sg[3]) > 0;
      uint32_t w = lr;
      int k = n + incl - cnt;
      while (w) {
        const int i = asc ? __builtin_ctz(w) : 31 - __builtin_clz(w);
        w &= ~(1u << i);
        int d[3];
        d[aw] = lw - OV, d[ab] = (ba ? at : i) - OV, d[ar] = (ba ? i : at) - OV;
        if (k < CELLS) L.cells.c[k] = Packed{(int8_t)d[0], (int8_t)d[1], (int8_t)d[2], 0};
        else wk.overflow = 1;
        ++k;
      }
    }
    n = n + total < CELLS ? n + total : CELLS;
    // positions left in this batch that had cells on top of the polyhedron when they were left; the sides' new positions
    int moved[4];
    CD_UNROLL
    for (int p = 0; p < 4; ++p) {
      const int j = (s_next + p) & 3;
      const unsigned long long mine = __ballot(com && pos == p), left_full = __ballot(com && pos == p && (w_prev & span) != 0u);
      const int mv = __popcll(mine);
      if (left_full) {
        const int turn = ((63 - __builtin_clzll(left_full)) >> 2) + 1;  // the latest of them: the position before that turn's move
        const int idx = pick4(j, bnd[0], bnd[1], bnd[2], bnd[3]) + pick4(j, sg[0], sg[1], sg[2], sg[3]) * (turn - 1);
        const int val = pick4(j, sg[0], sg[1], sg[2], sg[3]) * ((pick4(j, bit_axis[0], bit_axis[1], bit_axis[2], bit_axis[3]) ? sb : sr) + idx - OV);
        CD_UNROLL
        for (int jj = 0; jj < 4; ++jj)
          if (jj == j) farc[jj] = val;
      }
      CD_UNROLL
      for (int jj = 0; jj < 4; ++jj)
        if (jj == j) moved[jj] = mv;
    }
    CD_UNROLL
    for (int j = 0; j < 4; ++j) {
      if (bit_axis[j]) (sg[j] > 0 ? b1 : b0) += sg[j] * moved[j];
      else (sg[j] > 0 ? r1 : r0) += sg[j] * moved[j];
    }
    if (first_blocked < 64) {
      const int pf = first_blocked & 3;
      alive &= ~(1u << ((s_next + pf) & 3));
      s_next = (s_next + pf + 1) & 3;
    }
    // ... and the positions the sides are in now (lanes 0..3 look at sides 0..3)
    {
      const int j = lane & 3;
      const bool bj = pick4(j, bit_axis[0], bit_axis[1], bit_axis[2], bit_axis[3]) != 0;
      const int sj = pick4(j, sg[0], sg[1], sg[2], sg[3]);
      const int idx = bj ? (sj > 0 ? b1 : b0) : (sj > 0 ? r1 : r0);
      const uint32_t wc = (uint32_t)__shfl((int)col_real, idx), wr = (uint32_t)__shfl((int)real, idx);
      const uint32_t w = bj ? wc : wr;
      const bool full = (w & (bj ? range_mask(r0, r1) : range_mask(b0, b1))) != 0u;
      const unsigned long long now_full = __ballot(lane < 4 && full);
      CD_UNROLL
      for (int jj = 0; jj < 4; ++jj)
        if ((now_full >> jj) & 1ull) farc[jj] = sg[jj] * ((bit_axis[jj] ? sb : sr) + (bit_axis[jj] ? (sg[jj] > 0 ? b1 : b0) : (sg[jj] > 0 ? r1 : r0)) - OV);
    }
  }
#ifdef CD_CHECK_BATCH
  CD_SYNC();
  {
    bool bad = ck_n != n || ck_b0 != b0 || ck_b1 != b1 || ck_r0 != r0 || ck_r1 != r1;
    for (int j = 0; j < 4; ++j) bad = bad || ck_farc[j] != farc[j];
    // (the cells of the check list: every lane wrote the ones of its own rows / columns, and compares those)
    if (lane < OVW)
      for (int q = 1; q < ck_n && q < n && q < CELLS; ++q) {
        const Packed a = L.cells.c[q], b = ck_cells[q];
        const int own = lane - OV;  // this lane's row or column offset
        if ((b.x == own || b.y == own || b.z == own) && ck_mine[q] && (a.x != b.x || a.y != b.y || a.z != b.z)) bad = true;
      }
    bad = __ballot(bad) != 0ull;
    if (bad && lane == 0)
      fprintf(stderr, "BATCH MISMATCH F %d: n %d / %d, b [%d %d] / [%d %d], r [%d %d] / [%d %d], far %d %d %d %d / %d %d %d %d allow %d %d %d %d s2idx b %d r %d\n", F, n, ck_n, b0, b1, ck_b0,
              ck_b1, r0, r1, ck_r0, ck_r1, farc[0], farc[1], farc[2], farc[3], ck_farc[0], ck_farc[1], ck_farc[2], ck_farc[3], allow[0], allow[1], allow[2], allow[3],
              comp(s2, ab) - sb + OV, comp(s2, ar) - sr + OV);
    if (bad) wk.overflow = 1;  // (the decomposition then fails, and the test with it)
  }
#endif
  L.cells.n = n;
  CD_PROF(2);
  CD_UNROLL
  for (int j = 0; j < 4; ++j) {
    const bool ba = bit_axis[j];
    const int at = ba ? (sg[j] > 0 ? b1 : b0) : (sg[j] > 0 ? r1 : r0);
    const uint32_t m = line_of(ba, at, real, ba ? r0 : b0, ba ? r1 : b1);
    write_line(L.rim_real[j].c + RIM0, RIM - RIM0, ba, at, m, sg[(j + 1) & 3] > 0);
    L.rim_real[j].b = RIM0, L.rim_real[j].e = RIM0 + __builtin_popcount(m);
    L.far[j] = Cell{sd[j].x * farc[j], sd[j].y * farc[j], sd[j].z * farc[j]};
  }
  // the layer's cells as plane rows (= the REAL cells inside the rectangle), for mark_cells: the overlay copy whose words are
  // this plane's rows takes them with one OR per lane
  if (lane < OVW) L.plane[lane] = (lane >= r0 && lane <= r1) ? (real & range_mask(b0, b1)) : 0u;
  L.plane_axis = aw, L.plane_level = lw;
  CD_SYNC();
  CD_PROF(3);
  return true;
}
template <class G>
CD_INLINE __device__ inline bool grow_layer_wave(const Ctx& cx, const G& g, int f, const FaceState& fs, const int allow[4], int margin, Layer& L) {
  switch (f) {
    case 0: return grow_layer_wave_f<G, 0>(cx, g, fs, allow, margin, L);
    case 1: return grow_layer_wave_f<G, 1>(cx, g, fs, allow, margin, L);
    case 2: return grow_layer_wave_f<G, 2>(cx, g, fs, allow, margin, L);
    case 3: return grow_layer_wave_f<G, 3>(cx, g, fs, allow, margin, L);
    case 4: return grow_layer_wave_f<G, 4>(cx, g, fs, allow, margin, L);
    default: return grow_layer_wave_f<G, 5>(cx, g, fs, allow, margin, L);
  }
}
#endif

// One layer on top of face f: a free 2-D seed above the current outer layer, inside the allowance and inside voxels
// [1, dim - 1 - margin] (CD:94-116: margin 1; CD:700-723: margin 0), grown in its plane (CD:118-200).
// Grid G: nx(), ny(), nz(), inside(Cell), value(Cell) (the voxel, `mark` where the polyhedron already is), set(Cell, v),
// trial_set / trial_unset, and kHasPlanes (true: the cooperative mode's bit maps and mark_atomic may be there).
template <class G>
CD_NOINLINE CD_HD bool grow_layer_serial(const Ctx& cx, const G& g, int f, const FaceState& fs, const int allow[4], int mark, int margin, Layer& L) {
  Work& wk = *cx.wk;
  const Frame* fr = wk.fr;
  L.found = 0;
  L.cells.n = 0;
  const Cell up = normal_of(f);
  const Cell* sd = fr[f].side;
  Cell s2{0, 0, 0};
  auto seed_ok = [&](Cell t) {
    if (t.x < 1 || t.y < 1 || t.z < 1 || t.x >= g.nx() - margin || t.y >= g.ny() - margin || t.z >= g.nz() - margin) return false;
    if (g.value(t) >= kOccupied) return false;
    bool in = true;
    CD_UNROLL
    for (int k = 0; k < 4; ++k) in = in && dot(t, sd[k]) <= allow[k];
    return in;
  };
  bool found = false;
  CD_PROF_BEGIN();
  for (int q = 0; q < fs.outer.n; ++q) {
    const Cell t = add(cx.at(fs.outer, q), up);
    if (seed_ok(t)) {
      s2 = t, found = true;
      break;
    }
  }
  L.found = found ? 1 : 0;
  CD_PROF(0);
  if (!found) return false;
  // current outline of the layer per side: all cells (rim) / cells of the layer (rim_real)
  CD_UNROLL
  for (int j = 0; j < 4; ++j) cx.assign1(wk.rim[j], s2), cx.assign1(L.rim_real[j], s2), L.far[j] = s2;
  cx.push(L.cells, s2);
  bool alive[4] = {true, true, true, true};
  for (int k = 0; alive[0] || alive[1] || alive[2] || alive[3]; ++k) {
    if (wk.overflow) return true;
    const int s = k % 4, prev = (k + 3) % 4, next = (k + 1) % 4;
    cx.clear(wk.moved), cx.clear(wk.moved_real);
    bool ok = true;
    const int cnt = cx.size(wk.rim[s]);
    for (int q = 0; q < cnt; ++q) {
      const Cell t = add(cx.get(wk.rim[s], q), sd[s]);
      if (dot(t, sd[s]) > allow[s]) {
        ok = false;
        break;
      }
      const Cell below = sub(t, up);
      if (g.inside(below) && g.value(below) == mark) {  // on top of the polyhedron: must be free
        if (g.inside(t) && g.value(t) < kOccupied) {
          cx.push_back(wk.moved, t), cx.push_back(wk.moved_real, t);
        } else {
          ok = false;
          break;
        }
      } else {
        cx.push_back(wk.moved, t);  // beside the polyhedron: carried along, not part of the layer
      }
    }
    if (!ok) {
      alive[s] = false;  // (a side that failed is still tried again on later turns, as in CD:128-133)
      continue;
    }
    cx.copy(wk.rim[s], wk.moved);
    cx.append(L.cells, wk.moved_real);
    cx.copy(L.rim_real[s], wk.moved_real);
    cx.push_back(wk.rim[prev], cx.front(wk.moved));
    cx.push_front(wk.rim[next], cx.back(wk.moved));
    if (!cx.empty(wk.moved_real)) {
      if (same(cx.front(wk.moved), cx.front(wk.moved_real))) cx.push_back(L.rim_real[prev], cx.front(wk.moved));
      if (same(cx.back(wk.moved), cx.back(wk.moved_real))) cx.push_front(L.rim_real[next], cx.back(wk.moved));
    }
    CD_UNROLL
    for (int j = 0; j < 4; ++j)
      if (!cx.empty(L.rim_real[j])) L.far[j] = cx.front(L.rim_real[j]);
  }
  return true;
}

template <class G, bool COOP>
CD_INLINE CD_HD bool grow_layer(const Ctx& cx, const G& g, int f, const FaceState& fs, const int allow[4], int mark, int margin, Layer& L) {
#if CD_HAS_COOP
  if constexpr (COOP) return grow_layer_wave(cx, g, f, fs, allow, margin, L);
#endif
  return grow_layer_serial(cx, g, f, fs, allow, mark, margin, L);
}

CD_HD double dabs(double v) { return v < 0 ? -v : v; }
// |  |l0| - |l2|  | * |  |l1| - |l3|  |   (CD:691-693) — small integers: exact in double
CD_HD double span_area(const int l[4]) {
  return dabs(dabs((double)l[0]) - dabs((double)l[2])) * dabs(dabs((double)l[1]) - dabs((double)l[3]));
}

// extents of the grown layer as the reference reads them for its area test (CD:816-820): the front cell of every
// side. A side without layer cells has no front in the reference (it reads an empty deque there); the last known
// front (border_limit_tmp) stands in for it.
CD_HD void layer_extent(const Ctx& cx, const Frame& fv, const Layer& L, int ext[4]) {
  CD_UNROLL
  for (int j = 0; j < 4; ++j) ext[j] = dot(cx.empty(L.rim_real[j]) ? L.far[j] : cx.front(L.rim_real[j]), fv.side[j]);
}

// SideIsEmpty, CD:577-588 (GetVoxel: outside the grid = occupied; any positive value, potential field included, counts)
template <class G>
CD_HD bool side_is_empty(const Ctx& cx, const G& g, const CellDeque& cells, Cell step) {
  if (cx.empty(cells)) return false;
#if CD_HAS_COOP
  if (cx.coop) {
    const int cnt = cx.size(cells);
    bool any = false;
    for (int q = cx.lane; q < cnt; q += 64) {
      const Cell t = add(cx.get(cells, q), step);
      any = any || (g.inside(t) ? g.value(t) : kOccupied) > 0;
    }
    return __ballot(any) == 0;
  }
#endif
  for (int q = 0; q < cx.size(cells); ++q) {
    const Cell t = add(cx.get(cells, q), step);
    const int v = g.inside(t) ? g.value(t) : kOccupied;
    if (v > 0) return false;
  }
  return true;
}

// FindCorners, CD:378-564: a trial layer on face f from the given state; which square edges would become chamfers.
template <class G, bool COOP>
CD_INLINE CD_HD void find_corners_impl(const Ctx& cx, const G& g, int f, bool growing_f, const FaceState& face, const Edge* edges, int mark,
                                       bool& valid, Edge out[4]) {
  if (!growing_f) {
    valid = false;
    return;
  }
  Work& wk = *cx.wk;
  const Frame fv = wk.fr[f];
  int allow[4];
  allowance(fv, f, face, edges, allow);
  const double area = span_area(allow);
  Layer& L = wk.L2;
  // (whether a seed was found is the function's value, not L.found: in the cooperative mode a lane must not depend on when
  // another lane starts the next layer)
  const bool layer_found = grow_layer<G, COOP>(cx, g, f, face, allow, mark, 0, L);
  edges_of_face(fv, edges, out);
  if (!layer_found) return;
  int ext[4];
  layer_extent(cx, fv, L, ext);
  if (span_area(ext) < area / 2) valid = false;
  CD_UNROLL
  for (int j = 0; j < 4; ++j) {
    if (cx.empty(L.rim_real[j])) continue;
    const int gap = face.reach[j] - dot(cx.front(L.rim_real[j]), fv.side[j]);
    if (out[j].slope == 0 && gap > 0) {
      out[j].slope = out[j].steps = gap;
      if (gap > 1) out[j].dir = f;
    }
  }
}

// (out of line in the cooperative mode too: a trial layer is rare, and two more inlined copies of the six per-face layer
// functions would push the loop of decompose_core out of the instruction cache)
template <class G, bool COOP>
CD_NOINLINE CD_HD void find_corners(const Ctx& cx, const G& g, int f, bool growing_f, const FaceState& face, const Edge* edges, int mark, bool& valid,
                                    Edge out[4]) {
  find_corners_impl<G, COOP>(cx, g, f, growing_f, face, edges, mark, valid, out);
}

// rows[max_rows][4] = (n, n . p); returns CD_OK, CD_CAPACITY (n_rows = needed count) or CD_WORK_OVERFLOW (a fixed-capacity
// container of `wk` was too small for this grid: the caller falls back to a bigger workspace / reports it)
template <class G, bool COOP = false>
CD_HD int decompose_core(G& g, Work& wk, int variant, Cell seed, int n_it, double res, int mark, const double origin[3], double* rows,
                         int max_rows, int* n_rows, int lane = 0) {
  Ctx cx{&wk, COOP, lane, seed};
  // mark / unmark every cell of a list
  auto mark_cells = [&](const Layer& layer, int how) {  // 0 set, 1 trial_set, 2 trial_unset
    const CellList& l = layer.cells;
#if CD_HAS_COOP
    if constexpr (COOP) {
      // the copy of the overlay whose words are rows of the layer's plane takes the layer with one word per lane, the other
      // copy cell by cell
      const int aw = layer.plane_axis, lw = layer.plane_level;
      if (cx.lane < G::OVW) {
        uint32_t* copy = aw == 0 ? g.bits_t : g.bits;
        const int w = aw == 2 ? cx.lane + G::OVW * lw : lw + G::OVW * cx.lane;
        const uint32_t row = layer.plane[cx.lane];
        if (row) copy[w] = how == 2 ? (copy[w] & ~row) : (copy[w] | row);
      }
      for (int q = cx.lane; q < l.n; q += 64) g.mark_atomic(cx.at(l, q), how != 2, aw == 0);
      CD_SYNC();
      return;
    }
#endif
    for (int q = 0; q < l.n; ++q) {
      if (how == 0) g.set(cx.at(l, q), mark);
      else if (how == 1) g.trial_set(cx.at(l, q), mark);
      else g.trial_unset(cx.at(l, q));
    }
  };
  wk.overflow = 0;
  wk.seed = seed;
  build_frames(wk.fr);
  const Frame* fr = wk.fr;
  const bool aware = variant != 0;
  FaceState* faces = wk.faces;
  Edge* edges = wk.edges;
  Cell* anchor = wk.anchor;
  unsigned growing = 0x3fu;     // faces that still grow, one bit each (indexed by the turn's face: no local array)
  for (int e = 0; e < 12; ++e) edges[e] = fresh_edge();
  for (int f = 0; f < 6; ++f) {
    faces[f].outer.n = 1;  // (stores of values, no read-modify-write: in the cooperative mode every lane executes them)
    faces[f].outer.c[0] = cx.pack(seed);
    anchor[f] = seed;
    CD_UNROLL
    for (int j = 0; j < 4; ++j) faces[f].reach[j] = dot(seed, fr[f].side[j]);
  }
  g.set(seed, mark);

  CD_PROF_BEGIN();
  for (int it = 0; it < n_it; ++it) {
    if (wk.overflow) return CD_WORK_OVERFLOW;
    const int f = it % 6;
    if (!((growing >> f) & 1u)) continue;
    const Cell up = normal_of(f);
    const Frame fv = fr[f];  // (by value: see allowance)
    const Cell* sd = fv.side;

    int allow[4];
    Edge trial[4];
    allowance(fv, f, faces[f], edges, allow);
    Layer& L = wk.L;
    CD_PROF(4);
    const bool layer_found = grow_layer<G, COOP>(cx, g, f, faces[f], allow, mark, aware ? 0 : 1, L);
    if (wk.overflow) return CD_WORK_OVERFLOW;
    CD_PROF(11);  // (the layer itself: phases 0..3 lie inside)
    if (!layer_found) continue;
    edges_of_face(fv, edges, trial);

    bool soft = true;  // shape-aware variant: layer acceptable this turn
    if (aware) {
      int ext[4];
      layer_extent(cx, fv, L, ext);
      if (span_area(ext) < span_area(allow) / 2) soft = false;
    }

    // is the layer consistent with ONE plane through every edge? (CD:209-283 / CD:828-910)
    bool accept = true;
    int fresh[4] = {0, 0, 0, 0};  // corner_new_state: 1 = a one-voxel chamfer starts on this side, 2 = a longer one
    Cell fresh_c[4] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // ... from this voxel
    bool stop = false;  // (the reference's loop ends at !accept or at a break: written so that the loop unrolls and trial[] stays in registers)
    CD_UNROLL
    for (int j = 0; j < 4; ++j) {
      if (!accept || stop) continue;
      if (cx.empty(L.rim_real[j])) continue;
      Edge e = trial[j];
      const Cell c = cx.front(L.rim_real[j]);
      const int gap = faces[f].reach[j] - dot(c, sd[j]);  // voxels this layer falls short of the last one
      if (e.slope == 0) {
        if (gap > 0) {  // a chamfer starts here: a point of its plane, between this layer and the neighbouring face
          edge_point(c, up, normal_of(fv.face[j]), res, aware, e.pos);
          fresh_c[j] = c;
          e.slope = e.steps = gap;
          if (gap > 1) e.dir = f;
          fresh[j] = gap > 1 ? 2 : 1;
          if (aware && it < 6) soft = false;  // no chamfer during the first round of turns
        }
      } else if (e.fixed) {
        if (e.dir == f || e.dir == -1) {
          if (gap > e.slope) {
            if (aware) {  // CD:875-878: the scan of the edges ends here, this and the later sides keep their state
              stop = true;
              continue;
            }
            accept = false;
          }
        } else if (e.steps >= e.slope) {  // the other face has finished a stair: we may step in by one, once
          if (gap > 1) accept = false;
          else e.steps = 1;
        } else {                          // in the middle of a stair: no step allowed
          if (gap != 0) accept = false;
          else e.steps += 1;
        }
      } else if (e.dir == -1) {           // slope 1 so far, long direction still open
        if (gap == 0) e.dir = fv.face[j], e.steps += 1, e.slope += 1;
        else if (gap == 1) e.fixed = 1;
        else accept = false;
      } else if (e.dir == f) {            // first layer after our own multi-voxel retreat fixes the slope
        e.slope = gap, e.fixed = 1;
      } else {                            // the other face is the long direction and is still lengthening its stair
        if (gap == 0) e.slope += 1, e.steps += 1;
        else if (gap == 1) e.fixed = 1, e.steps = 1;
        else accept = false;
      }
      if (accept) trial[j] = e;
    }
    CD_PROF(5);
    if (!accept) {
      growing &= ~(1u << f);
      continue;
    }
    if (!soft) continue;

    if (aware) {  // CD:930-1066: does every chamfer that starts with this layer follow a real obstacle?
      bool expand = true;
      CD_UNROLL
      for (int j = 0; j < 4; ++j) {
        if (!fresh[j]) continue;
        const bool first = side_is_empty(cx, g, L.rim_real[j], up);
        const int nbf = fv.face[j], ci = fv.back[j];
        bool second = true;
        if (fresh[j] == 1) {
          int e_end = RIM0;  // the neighbouring face's cells along the shared edge
          for (int q = 0; q < faces[nbf].outer.n; ++q) {
            const Cell c = cx.at(faces[nbf].outer, q);
            if (dot(c, fr[nbf].side[ci]) == faces[nbf].reach[ci]) {
              if (e_end < RIM) wk.edge_row.c[e_end++] = cx.pack(c);
              else wk.overflow = 1;
            }
          }
          wk.edge_row.b = RIM0, wk.edge_row.e = e_end;
          if constexpr (COOP) CD_SYNC();
          second = side_is_empty(cx, g, wk.edge_row, normal_of(nbf));
        }
        expand = !(first && second);
        if (!expand) {
          growing &= ~(1u << f);
          break;
        }
      }
      CD_PROF(6);
      if (expand && (fresh[0] || fresh[1] || fresh[2] || fresh[3])) {
        // trial: put the layer in, grow one more on top of it, take it out again (its cells were free voxels)
        mark_cells(L, 1);
        FaceState& face_t = wk.face_t;  // (the other faces are what they are: the trial reads only the face it grows)
        cx.copy(face_t.outer, L.cells);
        CD_UNROLL
        for (int j = 0; j < 4; ++j) face_t.reach[j] = dot(L.far[j], sd[j]);
        Edge* edges_t = wk.edges_t;
        for (int k = 0; k < 12; ++k) edges_t[k] = edges[k];
        CD_UNROLL
        for (int j = 0; j < 4; ++j)
          if (!fresh[j]) edges_t[fv.edge[j]] = trial[j];
        bool valid = true;
        Edge fin[4];
        // (the out-of-line calls get copies of the context and the grid: the originals stay in registers instead of moving to
        // the stack for good)
        const Ctx cx_call = cx;
        const G g_call = g;
        find_corners<G, COOP>(cx_call, g_call, f, ((growing >> f) & 1u) != 0, face_t, edges_t, mark, valid, fin);
        mark_cells(L, 2);
        if (valid) {
          CD_UNROLL
          for (int j = 0; j < 4; ++j)
            if (fresh[j] == 2 && fin[j].slope < trial[j].slope) {
              expand = false;
              growing &= ~(1u << f);
              break;
            }
          if (expand) {
            // (this loop stays a loop — its body holds a whole trial layer — so nothing in it indexes a local array with j:
            // on the device such an array would live in scratch memory)
            auto pick = [](int j, int a0, int a1, int a2, int a3) { return j == 0 ? a0 : (j == 1 ? a1 : (j == 2 ? a2 : a3)); };
            for (int j = 0; j < 4; ++j) {
              if (pick(j, fresh[0], fresh[1], fresh[2], fresh[3]) != 1) continue;
              const int nbf = fv.face[j];
              bool v2 = true;
              Edge fin2[4];
              find_corners<G, COOP>(cx_call, g_call, nbf, ((growing >> nbf) & 1u) != 0, faces[nbf], edges, mark, v2, fin2);
              const int nb_slope = pick(fv.back[j], fin2[0].slope, fin2[1].slope, fin2[2].slope, fin2[3].slope);
              if (v2 && nb_slope == 0 && pick(j, fin[0].slope, fin[1].slope, fin[2].slope, fin[3].slope) == 0) {
                expand = false;
                break;
              }
            }
          }
        }
      }
      CD_PROF(7);
      if (wk.overflow) return CD_WORK_OVERFLOW;
      if (!expand) continue;
    }

    cx.copy(faces[f].outer, L.cells);
    CD_UNROLL
    for (int j = 0; j < 4; ++j) {
      faces[f].reach[j] = dot(L.far[j], sd[j]);
      edges[fv.edge[j]] = trial[j];
      if (fresh[j]) wk.esrc[fv.edge[j]] = EdgeSrc{fresh_c[j], f, fv.face[j], aware ? 1 : 0};
      // full-width side on a square edge: these voxels are now also the outermost layer of the neighbouring face
      if (trial[j].slope == 0 && !cx.empty(L.rim_real[j]) && faces[f].reach[j] == dot(cx.front(L.rim_real[j]), sd[j])) {
        const int nbf = fv.face[j];
        cx.append(faces[nbf].outer, L.rim_real[j]);
        const int reach_nb = faces[nbf].reach[fv.back[j]] + 1;
        if constexpr (COOP) CD_SYNC();
        faces[nbf].reach[fv.back[j]] = reach_nb;
      }
    }
    anchor[f] = cx.at(L.cells, 0);
    mark_cells(L, 0);
    CD_PROF(8);
  }
  if (wk.overflow) return CD_WORK_OVERFLOW;

  // half-spaces: chamfered edges first (edge numbering order), then the six faces (CD:322-373)
  int n = 0;
  for (int e = 0; e < 12; ++e) {
    if (edges[e].slope <= 0) continue;
    if (n < max_rows) edge_row(e, edges[e].slope, edges[e].dir, edges[e].pos, origin, rows + 4 * n);
    ++n;
  }
  for (int f = 0; f < 6; ++f) {
    if (n < max_rows) face_row(f, anchor[f], res, origin, rows + 4 * n);
    ++n;
  }
  *n_rows = n;
  CD_PROF(9);
  return n <= max_rows ? CD_OK : CD_CAPACITY;
}

}  // namespace hdsm_cd
