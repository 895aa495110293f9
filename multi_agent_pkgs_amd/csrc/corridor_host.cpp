// corridor_host.cpp — next row f2, host entry points of the convex voxel decomposition (hdsm_poly_octa3d = GetPolyOcta3D,
// hdsm_poly_octa3d_new = GetPolyOcta3DNew; include/hdsm_swarm.h). The algorithm itself is in corridor_core.h, shared with the
// device kernel (corridor_kernels.hip).
#include <memory>
#include <new>
#include <vector>

#include "../../include/hdsm_swarm.h"
#include "corridor_core.h"

namespace {

using namespace hdsm_cd;

// a caller-owned int8 grid, marks written in place (the contract of hdsm_poly_octa3d)
struct ArrayGrid {
  [[maybe_unused]] static constexpr bool kHasPlanes = false;
  int8_t* data;
  int dx, dy, dz;
  std::vector<int8_t> saved;  // values under a trial layer (CD:989-1000), restored in the order they were taken
  size_t restore_at = 0;
  int nx() const { return dx; }
  int ny() const { return dy; }
  int nz() const { return dz; }
  bool inside(Cell c) const { return c.x >= 0 && c.y >= 0 && c.z >= 0 && c.x < dx && c.y < dy && c.z < dz; }
  int8_t& at(Cell c) const { return data[c.x + c.y * dx + c.z * dx * dy]; }
  int value(Cell c) const { return at(c); }
  void set(Cell c, int v) { at(c) = (int8_t)v; }
  void trial_set(Cell c, int v) {
    if (restore_at == saved.size()) saved.clear(), restore_at = 0;
    saved.push_back(at(c));
    at(c) = (int8_t)v;
  }
  void trial_unset(Cell c) { at(c) = saved[restore_at++]; }
};

int decompose(int variant, const int32_t seed_in[3], int8_t* grid, const int32_t dim[3], int32_t n_it, double res, int32_t mark,
              const double origin[3], double* rows, int32_t max_rows, int32_t* n_rows) {
  if (!seed_in || !grid || !dim || !origin || !rows || !n_rows || n_it < 0 || !(res > 0) || mark >= kOccupied)
    return HDSM_ERR_BAD_ARG;
  ArrayGrid g{grid, dim[0], dim[1], dim[2], {}, 0};
  const Cell seed{seed_in[0], seed_in[1], seed_in[2]};
  if (!g.inside(seed)) return HDSM_ERR_BAD_ARG;
  std::unique_ptr<Work> wk(new (std::nothrow) Work);
  if (!wk) return HDSM_ERR_DEVICE;
  int n = 0;
  const int rc = decompose_core(g, *wk, variant, seed, n_it, res, mark, origin, rows, max_rows, &n);
  *n_rows = n;
  if (rc == CD_WORK_OVERFLOW) return HDSM_ERR_CAPACITY;  // a grid / n_it beyond the fixed workspace (n_it <= 54 always fits)
  return rc == CD_OK ? HDSM_OK : HDSM_ERR_CAPACITY;
}

}  // namespace

extern "C" int hdsm_poly_octa3d(const int32_t seed[3], int8_t* grid, const int32_t dim[3], int32_t n_it, double res,
                                int32_t mark, const double origin[3], double* rows, int32_t max_rows, int32_t* n_rows) {
  return decompose(0, seed, grid, dim, n_it, res, mark, origin, rows, max_rows, n_rows);
}

extern "C" int hdsm_poly_octa3d_new(const int32_t seed[3], int8_t* grid, const int32_t dim[3], int32_t n_it, double res,
                                    int32_t mark, const double origin[3], double* rows, int32_t max_rows, int32_t* n_rows) {
  return decompose(1, seed, grid, dim, n_it, res, mark, origin, rows, max_rows, n_rows);
}
