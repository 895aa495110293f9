// corridor_emu.cpp — runs the DEVICE source of the cooperative voxel decomposition (corridor_wave.h: world maps in "LDS", layers
// grown as bit planes, one wavefront per seed) on the CPU, 64 fibers in lockstep (TEST INFRASTRUCTURE ONLY; see wave_emu.cpp).
// Same arguments as hdsm_poly_octa3d_batch_wave (include/hdsm_swarm.h), host pointers, no device.
#define CD_EMU_COOP 1
#include <hip/hip_runtime.h>  // the shim

#include <vector>

#include "../../include/hdsm.h"
#include "../../multi_agent_pkgs_amd/csrc/corridor_wave.h"

namespace wemu {
bool run_block(void (*body)(void*), void* arg, int block_index, int nthreads);
const char* last_error();
}  // namespace wemu

namespace {
using namespace hdsm_cd;
struct Job {
  const int8_t* world;
  const int32_t *wdim, *ldim, *off, *ground, *seed, *variant;
  const double* origin;
  int n_it, max_rows;
  double res;
  double* rows;
  int32_t *n_rows, *rc, *cells;
  unsigned char* lds;
  int t;
};
void body(void* p) {
  const Job& b = *static_cast<Job*>(p);
  const int t = b.t, lane = (int)threadIdx.x;
  const WaveLds m(b.lds);
  const Cell seed{b.seed[3 * t], b.seed[3 * t + 1], b.seed[3 * t + 2]};
  WindowGrid g{b.world, b.wdim[0], b.wdim[1], b.wdim[2], b.off[3 * t], b.off[3 * t + 1], b.off[3 * t + 2],
               b.ldim[0], b.ldim[1], b.ldim[2], b.ground[t], -1, seed, m.bits};
  int rc = HDSM_OK, n = 0;
  if (!g.inside(seed)) {
    rc = HDSM_ERR_BAD_ARG;
  } else {
    const double org[3] = {b.origin[3 * t], b.origin[3 * t + 1], b.origin[3 * t + 2]};
    double rows[64 * 4];  // (every lane its own copy, as on the device; lane 0's goes out)
    const int r = wave_decompose(g, m, b.variant[t], b.n_it, b.res, org, rows, b.max_rows < 64 ? b.max_rows : 64, &n, lane);
    rc = (r == CD_OK) ? HDSM_OK : HDSM_ERR_CAPACITY;
    if (lane == 0)
      for (int i = 0; i < 4 * (n < b.max_rows ? n : b.max_rows) && i < 256; ++i) b.rows[(size_t)t * b.max_rows * 4 + i] = rows[i];
  }
  __syncthreads();
  if (lane == 0) {
    b.n_rows[t] = n;
    b.rc[t] = rc;
    if (b.cells) b.cells[t] = (rc == HDSM_ERR_BAD_ARG) ? 0 : g.count();
  }
}
}  // namespace

extern "C" int wave_poly_octa3d_batch(int32_t n, const int8_t* world, const int32_t wdim[3], const int32_t ldim[3], const int32_t* off,
                                      const int32_t* ground_k, const int32_t* seed, const int32_t* variant, const double* origin,
                                      int32_t n_it, double res, double* rows, int32_t max_rows, int32_t* n_rows, int32_t* rc,
                                      int32_t* cells) {
  const size_t bytes = wave_lds_bytes(wave_map_radius(n_it)), guard = 4096;  // what the launch asks for, between two guard zones
  std::vector<unsigned char> mem(guard + bytes + guard);
  unsigned char* lds_base = mem.data() + guard;
  for (int t = 0; t < n; ++t) {
    for (auto& v : mem) v = 0xA5;  // uninitialised LDS
    Job j{world, wdim, ldim, off, ground_k, seed, variant, origin, n_it, max_rows, res, rows, n_rows, rc, cells, lds_base, t};
    if (!wemu::run_block(body, &j, t, 64)) return -100;
    for (size_t i = 0; i < guard; ++i)
      if (mem[i] != 0xA5 || mem[guard + bytes + i] != 0xA5) return -101;  // a store outside the LDS of the launch
  }
  return 0;
}
