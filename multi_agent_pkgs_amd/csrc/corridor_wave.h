// corridor_wave.h — the cooperative (one wavefront per seed) form of the voxel decomposition: what a wavefront does around
// hdsm_cd::decompose_core. Device code, shared by k_poly_octa3d_wave (corridor_kernels.hip), k_corridor of the device-resident
// loop (swarm_kernels.hip) and the CPU execution of the device source in the test-suite (tests/wave_emu).
#pragma once
#include "corridor_core.h"

namespace hdsm_cd {

// AC:1385-1395: a seed pinched between two occupied voxels along an axis takes the shape-aware variant (VoxelGrid::IsOccupied:
// == 100, outside the grid: not occupied). Nothing is marked yet, so this is the world under the overlay.
CD_HD bool seed_is_pinched(const WindowGrid& g, Cell s) {
  auto occ = [&](int dx, int dy, int dz) {
    const Cell c{s.x + dx, s.y + dy, s.z + dz};
    return g.inside(c) && g.world_value(c) == kOccupied;
  };
  return (occ(-1, 0, 0) && occ(1, 0, 0)) || (occ(0, -1, 0) && occ(0, 1, 0)) || (occ(0, 0, -1) && occ(0, 0, 1));
}

#if defined(__HIPCC__) || defined(CD_EMU_COOP)
// LDS of one cooperative decomposition: the workspace with the y-fast copy of the overlay laid over its last members (the serial
// form's rim deques, which the cooperative form does not use), the rows of one polyhedron, the overlay, and the world maps for the
// 2 r + 1 z-levels a decomposition of n_it turns looks at.
constexpr int WAVE_ROWS = 32;  // room for the rows of one polyhedron (HDSM_MAX_ROWS_STATIC) (the swarm loop's corridor kernel collects them here)
constexpr size_t WAVE_BITS_T_AT = offsetof(Work, rim);
static_assert(WAVE_BITS_T_AT % 16 == 0 || WAVE_BITS_T_AT % 4 == 0, "the overlay copy starts on a word");
constexpr size_t WAVE_HEAD_BYTES = (((WAVE_BITS_T_AT + WindowGrid::WORDS * 4 > sizeof(Work) ? WAVE_BITS_T_AT + WindowGrid::WORDS * 4 : sizeof(Work)) + 15) / 16) * 16;
constexpr size_t wave_lds_bytes(int r) {  // r = wave_map_radius(n_it); 0: no maps (the plain form on lane 0)
  return WAVE_HEAD_BYTES + WAVE_ROWS * 4 * 8 + WindowGrid::WORDS * 4 + (size_t)(r > 0 ? 3 * (2 * r + 1) * WindowGrid::OVW * 4 : 0);
}
constexpr size_t WAVE_LDS_MAX = wave_lds_bytes(WindowGrid::OV - 1);
struct WaveLds {
  Work* wk;
  double* rows;
  uint32_t *bits, *bits_t, *maps;  // maps: 3 x (2 r + 1) x OVW words, in the order FREE x-fast, POS x-fast, FREE y-fast
  __device__ explicit WaveLds(unsigned char* lds)
      : wk(reinterpret_cast<Work*>(lds)), rows(reinterpret_cast<double*>(lds + WAVE_HEAD_BYTES)),
        bits(reinterpret_cast<uint32_t*>(lds + WAVE_HEAD_BYTES + WAVE_ROWS * 4 * 8)), bits_t(reinterpret_cast<uint32_t*>(lds + WAVE_BITS_T_AT)),
        maps(bits + WindowGrid::WORDS) {}
};

// how far from the seed a decomposition of n_it turns looks: n_it / 6 layers per face (rounded up), the layer on top of the last
// one, and SideIsEmpty one voxel beyond that; 0 = more than the overlay holds (no maps: the plain cooperative form runs)
__host__ __device__ inline int wave_map_radius(int n_it) {
  const int r = (n_it + 5) / 6 + 2;
  return r <= WindowGrid::OV - 1 ? r : 0;
}

// Clears the overlay and classifies the world under it (WindowGrid::maps) for the offsets |d| <= r from the seed. Lanes 0..31
// and 32..63 take two z-levels at once, a lane = one x: the byte loads of a row are contiguous, the x-fast words are the two
// halves of a ballot, the y-fast words accumulate in the lane while it walks along y.
__device__ inline void build_world_maps(const WindowGrid& g, const WaveLds& m, int r, int lane) {
  constexpr int OV = WindowGrid::OV, OVW = WindowGrid::OVW, WORDS = WindowGrid::WORDS;
  for (int w = lane; w < WORDS; w += 64) m.bits[w] = 0u;
  if (r <= 0) return;  // (no maps and no second copy of the overlay: where it would lie the plain form keeps its deques)
  for (int w = lane; w < WORDS; w += 64) m.bits_t[w] = 0u;
  const int levels = 2 * r + 1;
  uint32_t* const map_free = m.maps - OVW * (OV - r);  // indexed like the overlay, see WindowGrid
  uint32_t* const map_pos = map_free + OVW * levels;
  uint32_t* const map_free_t = map_pos + OVW * levels;
  const int lx = lane & 31, half = lane >> 5;
  const int cx = g.seed.x + lx - OV;
  const bool x_in = lx >= OV - r && lx <= OV + r;
  const size_t wsize = (size_t)g.wnx * g.wny * g.wnz;
  // the class of one voxel, 1 = free | 2 = positive, without a branch around the load (its address is clamped into the world): the
  // loads of a batch of rows are then in flight together instead of one global round trip per row
  auto load = [&](Cell c, bool wanted, int& raw) {
    const int gi = c.x + g.ox, gj = c.y + g.oy, gk = c.z + g.oz;
    const bool in_world = gi >= 0 && gj >= 0 && gk >= 0 && gi < g.wnx && gj < g.wny && gk < g.wnz;
    const size_t idx = (size_t)gi + (size_t)gj * g.wnx + (size_t)gk * g.wnx * g.wny;
    raw = g.world[(wanted && in_world && idx < wsize) ? idx : 0];
    return in_world;
  };
  auto classify = [&](Cell c, bool wanted, bool in_world, int raw) {
    if (!(wanted && g.inside(c))) return 2u;                       // not free, "positive" (GetVoxel outside the grid: occupied)
    const int v = c.z < g.ground_k ? kOccupied : (!in_world ? 0 : (raw < 0 ? kOccupied : raw));  // WindowGrid::world_value
    return (v < kOccupied ? 1u : 0u) | (v > 0 ? 2u : 0u);
  };
  constexpr int B = 10;  // rows per batch (19 rows for n_it = 42: two batches; every slot costs its instructions whether a row is wanted or not, and larger batches were slower)
  for (int dz0 = OV - r; dz0 <= OV + r; dz0 += 2) {
    const int dz = dz0 + half;
    const int cz = g.seed.z + dz - OV;
    const bool z_in = dz <= OV + r;
    uint32_t fy = 0;
    for (int dy0 = OV - r; dy0 <= OV + r; dy0 += B) {
      int raw[B];
      bool inw[B];
#pragma unroll
      for (int u = 0; u < B; ++u) inw[u] = load(Cell{cx, g.seed.y + dy0 + u - OV, cz}, x_in && z_in && dy0 + u <= OV + r, raw[u]);
#pragma unroll
      for (int u = 0; u < B; ++u) {
        const int dy = dy0 + u;
        const bool wanted = x_in && z_in && dy <= OV + r;
        const uint32_t cls = classify(Cell{cx, g.seed.y + dy - OV, cz}, wanted, inw[u], raw[u]);
        const unsigned long long bf = __ballot(cls & 1u), bp = __ballot(cls & 2u);
        if (dy <= OV + r) {
          if (lx == 0 && z_in) {
            map_free[dy + OVW * dz] = (uint32_t)(bf >> (32 * half));
            map_pos[dy + OVW * dz] = (uint32_t)(bp >> (32 * half));
          }
          fy |= (cls & 1u) << dy;
        }
      }
    }
    if (z_in) map_free_t[lx + OVW * dz] = fy;
  }
}

// One decomposition by the whole wavefront: `g` = the window without overlay / maps (filled in here), variant -1 = decide like
// AC:1385-1395. Every lane returns the same code and the same rows.
__device__ inline int wave_decompose(WindowGrid g, const WaveLds& m, int variant, int n_it, double res, const double origin[3], double* rows,
                                     int max_rows, int* n_rows, int lane) {
  const int r = wave_map_radius(n_it);
  g.bits = m.bits;
#if defined(CD_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (lane == 0)
    for (int i = 0; i < 16; ++i) m.wk->prof[i] = 0;
#endif
  build_world_maps(g, m, r, lane);
  __syncthreads();
#if defined(CD_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
  if (lane == 0) m.wk->prof[10] = __builtin_readcyclecounter() - t0, m.wk->prof[15] = 1;
#endif
  if (variant < 0) variant = seed_is_pinched(g, g.seed) ? 1 : 0;
  if (r <= 0) {
    // more turns than the overlay's planes hold (n_it > 78): the plain form, by lane 0 (rows: the caller's array must then be
    // memory every lane sees — global memory or LDS)
    if (lane == 0) {
      int n = 0;
      WindowGrid g_plain = g;  // (its address goes into out-of-line calls: a copy, so that `g` itself stays in registers)
      const int rc = decompose_core<WindowGrid, false>(g_plain, *m.wk, variant, g.seed, n_it, res, g.mark, origin, rows, max_rows, &n);
      m.wk->seed_plane[0] = (uint32_t)rc, m.wk->seed_plane[1] = (uint32_t)n;
    }
    __syncthreads();
    *n_rows = (int)m.wk->seed_plane[1];
    return (int)m.wk->seed_plane[0];
  }
  g.maps = m.maps - WindowGrid::OVW * (WindowGrid::OV - r), g.maps_pos = g.maps + WindowGrid::OVW * (2 * r + 1), g.maps_t = g.maps_pos + WindowGrid::OVW * (2 * r + 1);
  g.bits_t = m.bits_t, g.map_r = r;
  const int rc_wave = decompose_core<WindowGrid, true>(g, *m.wk, variant, g.seed, n_it, res, g.mark, origin, rows, max_rows, n_rows, lane);
#if defined(CD_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
  __syncthreads();
  if (lane < 16 && lane != 12 && lane != 13 && lane != 14) atomicAdd(&g_cd_prof[lane], m.wk->prof[lane]);
#endif
  return rc_wave;
}

// The structure of the polyhedron the last wave_decompose of this workgroup produced (PolyStruct: what its rows are made of, as
// integers relative to the seed), written by the lanes to `out` (any memory).
__device__ inline void wave_poly_structure(const WaveLds& m, Cell seed, PolyStruct* out, int lane) {
  const Work& wk = *m.wk;
  if (lane < 12) {
    const int e = lane, sl = wk.edges[e].slope;
    out->slope[e] = sl, out->dir[e] = wk.edges[e].dir;
    out->f[e] = sl > 0 ? wk.esrc[e].f : 0, out->nbf[e] = sl > 0 ? wk.esrc[e].nbf : 0, out->extra[e] = sl > 0 ? wk.esrc[e].extra : 0;
    out->c[e] = sl > 0 ? sub(wk.esrc[e].c, seed) : Cell{0, 0, 0};
  } else if (lane < 18) {
    out->anchor[lane - 12] = sub(wk.anchor[lane - 12], seed);
  }
}

#endif  // __HIPCC__

}  // namespace hdsm_cd
