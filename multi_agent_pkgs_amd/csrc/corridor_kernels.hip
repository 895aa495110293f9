// corridor_kernels.hip — next row f2 on the device: a BATCH of convex voxel decompositions (GetPolyOcta3D / GetPolyOcta3DNew,
// see corridor_core.h) on local grids that are windows into ONE world grid resident in HBM.
//
// The algorithm is an irregular integer flood with a sequential state machine per seed; what a swarm offers is thousands of
// independent seeds per replan round (one per agent and new polyhedron). One THREAD runs one decomposition, compiled from the
// same source as the host entry points, so the rows agree with them bit for bit; its ~45 KB of containers live in a global
// scratch slab, the polyhedron's voxels in a 4 KB bit overlay around the seed (the shared world grid is never written).
// Lanes of a wavefront diverge freely — the unit of parallelism is the seed, not the voxel. k_poly_octa3d_wave is the other
// form: one wavefront per seed, cooperative (what the swarm loop's corridor kernel does for one agent's seeds).
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/hdsm_swarm.h"
#include "corridor_core.h"
#include "corridor_wave.h"

namespace {

using namespace hdsm_cd;

thread_local std::string g_err;

struct Batch {
  const int8_t* world;
  int wdim[3];
  int ldim[3];
  int n, n_it, max_rows;
  double res;
  const int32_t* off;      // [n][3] local voxel 0 in world voxels
  const int32_t* ground;   // [n] first local k that is not below the ground
  const int32_t* seed;     // [n][3] local
  const int32_t* variant;  // [n] 0 = GetPolyOcta3D, 1 = GetPolyOcta3DNew, -1 = decide like AC:1385-1395 (pinched seed)
  const double* origin;    // [n][3] world position of local voxel (0,0,0)
  double* rows;            // [n][max_rows][4]
  int32_t* n_rows;         // [n]
  int32_t* rc;             // [n] hdsm_error
  int32_t* cells;          // [n] voxels of the polyhedron (may be null)
  unsigned char* scratch;  // [n] x SLAB
};

constexpr size_t SLAB = ((sizeof(Work) + 15) / 16) * 16 + WindowGrid::WORDS * 4;

__global__ __launch_bounds__(64) void k_poly_octa3d(Batch b) {
  const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (t >= b.n) return;
  unsigned char* slab = b.scratch + (size_t)t * SLAB;
  Work& wk = *reinterpret_cast<Work*>(slab);
  uint32_t* bits = reinterpret_cast<uint32_t*>(slab + ((sizeof(Work) + 15) / 16) * 16);
  for (int w = 0; w < WindowGrid::WORDS; ++w) bits[w] = 0u;
  const Cell seed{b.seed[3 * t], b.seed[3 * t + 1], b.seed[3 * t + 2]};
  WindowGrid g{b.world, b.wdim[0], b.wdim[1], b.wdim[2], b.off[3 * t], b.off[3 * t + 1], b.off[3 * t + 2],
               b.ldim[0], b.ldim[1], b.ldim[2], b.ground[t], -1, seed, bits};
  int rc = HDSM_OK, n = 0;
  if (!g.inside(seed)) {
    rc = HDSM_ERR_BAD_ARG;
  } else {
    int variant = b.variant[t];
    if (variant < 0) variant = seed_is_pinched(g, seed) ? 1 : 0;  // AC:1385-1395
    const double org[3] = {b.origin[3 * t], b.origin[3 * t + 1], b.origin[3 * t + 2]};
    const int r = decompose_core(g, wk, variant, seed, b.n_it, b.res, -1, org, b.rows + (size_t)t * b.max_rows * 4, b.max_rows, &n);
    rc = (r == CD_OK) ? HDSM_OK : HDSM_ERR_CAPACITY;
  }
  b.n_rows[t] = n;
  b.rc[t] = rc;
  if (b.cells) b.cells[t] = (rc == HDSM_ERR_BAD_ARG) ? 0 : g.count();
}

// The same batch with ONE WAVEFRONT per seed: workspace, overlay and bit maps of the world under the overlay in LDS, the
// decomposition run cooperatively by the 64 lanes (corridor_wave.h, corridor_core.h Ctx::coop) — the form the device-resident
// swarm loop uses inside its corridor kernel, where one agent's decompositions are a latency chain. Lower latency per seed,
// fewer seeds in flight; same rows, bit for bit (tests/test_gpu_configs.py).
static_assert(WAVE_LDS_MAX + 1024 <= 64 * 1024, "k_poly_octa3d_wave: LDS exceeds the default 64 KB limit of a launch");

__global__ __launch_bounds__(64) void k_poly_octa3d_wave(Batch b) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int t = (int)blockIdx.x, lane = (int)threadIdx.x;
  if (t >= b.n) return;
  const WaveLds m(lds);
  const Cell seed{b.seed[3 * t], b.seed[3 * t + 1], b.seed[3 * t + 2]};
  WindowGrid g{b.world, b.wdim[0], b.wdim[1], b.wdim[2], b.off[3 * t], b.off[3 * t + 1], b.off[3 * t + 2],
               b.ldim[0], b.ldim[1], b.ldim[2], b.ground[t], -1, seed, m.bits};
  int rc = HDSM_OK, n = 0;
  if (!g.inside(seed)) {
    rc = HDSM_ERR_BAD_ARG;
  } else {
    const double org[3] = {b.origin[3 * t], b.origin[3 * t + 1], b.origin[3 * t + 2]};
    const int r = wave_decompose(g, m, b.variant[t], b.n_it, b.res, org, b.rows + (size_t)t * b.max_rows * 4, b.max_rows, &n, lane);
    rc = (r == CD_OK) ? HDSM_OK : HDSM_ERR_CAPACITY;
  }
  __syncthreads();
  if (lane == 0) {
    b.n_rows[t] = n;
    b.rc[t] = rc;
    if (b.cells) b.cells[t] = (rc == HDSM_ERR_BAD_ARG) ? 0 : g.count();
  }
}

int fail(int code, const std::string& m) {
  g_err = m;
  return code;
}

}  // namespace

extern "C" {

const char* hdsm_corridor_last_error(void) { return g_err.c_str(); }

#ifdef CD_PROFILE
// development builds only: the phase counters of the cooperative decompositions launched from THIS file (read and cleared)
int hdsm_corridor_profile(unsigned long long out[16]) {
  if (hipDeviceSynchronize() != hipSuccess) return HDSM_ERR_DEVICE;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(hdsm_cd::g_cd_prof), 16 * sizeof(unsigned long long)) != hipSuccess) return HDSM_ERR_DEVICE;
  unsigned long long zero[16] = {0};
  if (hipMemcpyToSymbol(HIP_SYMBOL(hdsm_cd::g_cd_prof), zero, sizeof zero) != hipSuccess) return HDSM_ERR_DEVICE;
  return HDSM_OK;
}
#endif

size_t hdsm_poly_octa3d_scratch_bytes(int32_t n) { return (size_t)(n > 0 ? n : 0) * SLAB; }

static int launch_batch(bool wave, int32_t device, int32_t n, const int8_t* world, const int32_t wdim[3], const int32_t ldim[3],
                        const int32_t* off, const int32_t* ground_k, const int32_t* seed, const int32_t* variant,
                        const double* origin, int32_t n_it, double res, double* rows, int32_t max_rows, int32_t* n_rows,
                        int32_t* rc, int32_t* cells, void* scratch, void* hip_stream) {
  if (n < 0 || !wdim || !ldim || n_it < 0 || !(res > 0) || max_rows < 6) return fail(HDSM_ERR_BAD_ARG, "bad size argument");
  if (n == 0) return HDSM_OK;
  if (!world || !off || !ground_k || !seed || !variant || !origin || !rows || !n_rows || !rc || (!wave && !scratch))
    return fail(HDSM_ERR_BAD_ARG, "null array argument");
  if (hipSetDevice(device) != hipSuccess) return fail(HDSM_ERR_NO_DEVICE, "hipSetDevice failed");
  Batch b{};
  b.world = world;
  for (int k = 0; k < 3; ++k) b.wdim[k] = wdim[k], b.ldim[k] = ldim[k];
  b.n = n, b.n_it = n_it, b.max_rows = max_rows, b.res = res;
  b.off = off, b.ground = ground_k, b.seed = seed, b.variant = variant, b.origin = origin;
  b.rows = rows, b.n_rows = n_rows, b.rc = rc, b.cells = cells, b.scratch = static_cast<unsigned char*>(scratch);
  if (wave) hipLaunchKernelGGL(k_poly_octa3d_wave, dim3(n), dim3(64), wave_lds_bytes(wave_map_radius(n_it)), static_cast<hipStream_t>(hip_stream), b);
  else hipLaunchKernelGGL(k_poly_octa3d, dim3((n + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(hip_stream), b);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(HDSM_ERR_DEVICE, std::string("k_poly_octa3d: ") + hipGetErrorString(e));
  return HDSM_OK;
}

int hdsm_poly_octa3d_device(int32_t device, int32_t n, const int8_t* world, const int32_t wdim[3], const int32_t ldim[3],
                            const int32_t* off, const int32_t* ground_k, const int32_t* seed, const int32_t* variant,
                            const double* origin, int32_t n_it, double res, double* rows, int32_t max_rows, int32_t* n_rows,
                            int32_t* rc, int32_t* cells, void* scratch, void* hip_stream) {
  return launch_batch(false, device, n, world, wdim, ldim, off, ground_k, seed, variant, origin, n_it, res, rows, max_rows, n_rows, rc, cells,
                      scratch, hip_stream);
}

int hdsm_poly_octa3d_device_wave(int32_t device, int32_t n, const int8_t* world, const int32_t wdim[3], const int32_t ldim[3],
                                 const int32_t* off, const int32_t* ground_k, const int32_t* seed, const int32_t* variant,
                                 const double* origin, int32_t n_it, double res, double* rows, int32_t max_rows, int32_t* n_rows,
                                 int32_t* rc, int32_t* cells, void* hip_stream) {
  return launch_batch(true, device, n, world, wdim, ldim, off, ground_k, seed, variant, origin, n_it, res, rows, max_rows, n_rows, rc, cells,
                      nullptr, hip_stream);
}

static int batch_impl(bool wave, int32_t device, int32_t n, const int8_t* world, const int32_t wdim[3], const int32_t ldim[3],
                      const int32_t* off, const int32_t* ground_k, const int32_t* seed, const int32_t* variant,
                      const double* origin, int32_t n_it, double res, double* rows, int32_t max_rows, int32_t* n_rows,
                      int32_t* rc, int32_t* cells) {
  if (n < 0 || !wdim || !ldim) return fail(HDSM_ERR_BAD_ARG, "bad size argument");
  if (n == 0) return HDSM_OK;
  if (!world || !off || !ground_k || !seed || !variant || !origin || !rows || !n_rows || !rc) return fail(HDSM_ERR_BAD_ARG, "null array argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(HDSM_ERR_NO_DEVICE, "no such HIP device");
  if (hipSetDevice(device) != hipSuccess) return fail(HDSM_ERR_NO_DEVICE, "hipSetDevice failed");
  const size_t wtot = (size_t)wdim[0] * wdim[1] * wdim[2], N = (size_t)n;
  void *d_world = nullptr, *d_off = nullptr, *d_ground = nullptr, *d_seed = nullptr, *d_var = nullptr, *d_org = nullptr, *d_rows = nullptr,
       *d_nrows = nullptr, *d_rc = nullptr, *d_cells = nullptr, *d_scratch = nullptr;
  hipError_t e = hipSuccess;
  auto al = [&](void** p, size_t bytes) {
    if (e == hipSuccess) e = hipMalloc(p, bytes ? bytes : 1);
  };
  al(&d_world, wtot), al(&d_off, N * 12), al(&d_ground, N * 4), al(&d_seed, N * 12), al(&d_var, N * 4), al(&d_org, N * 24);
  al(&d_rows, N * max_rows * 32), al(&d_nrows, N * 4), al(&d_rc, N * 4), al(&d_cells, N * 4), al(&d_scratch, wave ? 16 : hdsm_poly_octa3d_scratch_bytes(n));
  auto up = [&](void* d, const void* h, size_t bytes) {
    if (e == hipSuccess) e = hipMemcpy(d, h, bytes, hipMemcpyHostToDevice);
  };
  up(d_world, world, wtot), up(d_off, off, N * 12), up(d_ground, ground_k, N * 4), up(d_seed, seed, N * 12), up(d_var, variant, N * 4);
  up(d_org, origin, N * 24);
  int rcall = HDSM_OK;
  if (e == hipSuccess)
    rcall = launch_batch(wave, device, n, (const int8_t*)d_world, wdim, ldim, (const int32_t*)d_off, (const int32_t*)d_ground,
                         (const int32_t*)d_seed, (const int32_t*)d_var, (const double*)d_org, n_it, res, (double*)d_rows, max_rows,
                         (int32_t*)d_nrows, (int32_t*)d_rc, (int32_t*)d_cells, d_scratch, nullptr);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  auto down = [&](void* h, const void* d, size_t bytes) {
    if (e == hipSuccess && h) e = hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost);
  };
  down(rows, d_rows, N * max_rows * 32), down(n_rows, d_nrows, N * 4), down(rc, d_rc, N * 4), down(cells, d_cells, N * 4);
  for (void* p : {d_world, d_off, d_ground, d_seed, d_var, d_org, d_rows, d_nrows, d_rc, d_cells, d_scratch})
    if (p) (void)hipFree(p);
  if (rcall) return rcall;
  if (e != hipSuccess) return fail(HDSM_ERR_DEVICE, std::string("hdsm_poly_octa3d_batch: ") + hipGetErrorString(e));
  return HDSM_OK;
}

int hdsm_poly_octa3d_batch(int32_t device, int32_t n, const int8_t* world, const int32_t wdim[3], const int32_t ldim[3],
                           const int32_t* off, const int32_t* ground_k, const int32_t* seed, const int32_t* variant,
                           const double* origin, int32_t n_it, double res, double* rows, int32_t max_rows, int32_t* n_rows,
                           int32_t* rc, int32_t* cells) {
  return batch_impl(false, device, n, world, wdim, ldim, off, ground_k, seed, variant, origin, n_it, res, rows, max_rows, n_rows, rc, cells);
}

int hdsm_poly_octa3d_batch_wave(int32_t device, int32_t n, const int8_t* world, const int32_t wdim[3], const int32_t ldim[3],
                                const int32_t* off, const int32_t* ground_k, const int32_t* seed, const int32_t* variant,
                                const double* origin, int32_t n_it, double res, double* rows, int32_t max_rows, int32_t* n_rows,
                                int32_t* rc, int32_t* cells) {
  return batch_impl(true, device, n, world, wdim, ldim, off, ground_k, seed, variant, origin, n_it, res, rows, max_rows, n_rows, rc, cells);
}

}  // extern "C"
