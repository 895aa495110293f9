// map_kernels.hip — next row f4: the map pre-processing stencils of mapping_util's MapBuilder (map_builder.cpp:209-216)
// on a batch of int8 voxel grids: SetUncertainToUnknown (MB:331-362), InflateObstacles and CreatePotentialField
// (voxel_grid.cpp:252-297, "VG"). See include/hdsm.h.
//
// The reference stamps a mask around every occupied voxel (scatter). Both masks are RADIAL — membership and value of
// an offset depend on its length only (CreateMask, VG:192-226) and the value decreases with the length — so
//   inflated(v)  <=>  the nearest occupied voxel, searched in the cube |d| <= rn per axis, is at a length the mask holds
//   potential(v)  =   mask value at the length of the nearest occupied voxel in that cube
// and the nearest length comes from an exact, windowed, separable squared distance transform: three 1-D min-plus
// passes (x, y, z) of 2 rn + 1 taps each instead of (2 rn + 1)^3 stamps — 33 taps for the shipped potential field
// instead of ~1200. Squared lengths are small integers (<= 3 rn^2 <= 255 for rn <= 9), kept in one byte; the mask
// becomes a table indexed by squared length, built on the host with the reference's own double arithmetic
// (hypot, pow, the int8 cast) and checked to be the same for every offset of equal squared length.
// Byte work, HBM/L2-bound: thread <-> voxel, x fastest, so every tap of every pass is a coalesced byte stream.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hdsm.h"

namespace {

constexpr int kRnMax = 9;        // 3 * 9^2 = 243 < 255
constexpr uint8_t kFar = 255;    // "no occupied voxel in the window"

struct MapTables {
  int rn0, rn1, rn2;             // cube of SetUncertainToUnknown, window of the inflation / potential masks
  uint8_t in1[256];              // squared length -> inflation mask holds it
  int8_t val2[256];              // squared length -> potential value (0 = not in the mask or no effect)
};

struct Dims {
  int nx, ny, nz;
  int vox;                       // voxels per grid (< 2^31); the grid of the batch is blockIdx.y
};

// Every thread owns FOUR consecutive voxels of one grid (x fastest): taps are fetched as one (unaligned) 32-bit load
// per row instead of four byte loads, and the index arithmetic is 32-bit and shared by the four.
constexpr int VPT = 4;

__device__ __forceinline__ uint32_t load4(const void* p) {  // 4 bytes at any alignment
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}
__device__ __forceinline__ void store4(void* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
__device__ __forceinline__ int byte_of(uint32_t w, int k) { return (int)((w >> (8 * k)) & 0xffu); }

// Packed arithmetic on the four voxels of a quad: two registers of 2 x u16 (v_pk_add_u16 / v_pk_min_u16), bytes moved
// in and out with v_perm_b32. A tap then costs 7 instructions for four voxels instead of ~40 scalar ones.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_pk(uint32_t w) { return __builtin_bit_cast(u16x2, w); }
__device__ __forceinline__ uint32_t as_u32(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ u16x2 widen_lo(uint32_t w) { return as_pk(__builtin_amdgcn_perm(0u, w, 0x0c010c00u)); }  // bytes 0, 1
__device__ __forceinline__ u16x2 widen_hi(uint32_t w) { return as_pk(__builtin_amdgcn_perm(0u, w, 0x0c030c02u)); }  // bytes 2, 3
__device__ __forceinline__ uint32_t narrow(u16x2 lo, u16x2 hi) {  // values already <= 255
  return __builtin_amdgcn_perm(as_u32(hi), as_u32(lo), 0x06040200u);
}
// per byte: 0x00 where the byte of w equals `key`, 0xff elsewhere
__device__ __forceinline__ uint32_t far_unless(uint32_t w, uint32_t key4) {
  const uint32_t x = w ^ key4;                                                 // zero byte <=> match
  const uint32_t nz = (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;   // bit 7 set <=> byte non-zero
  return (nz >> 7) * 255u;
}

struct Pos {  // coordinates of voxel p + k for k < VPT (a quad may run over the end of a row / slice)
  int x[VPT], y[VPT], z[VPT];
};
__device__ __forceinline__ Pos coords(const Dims& d, int p) {
  Pos c;
  const unsigned slice = (unsigned)d.nx * (unsigned)d.ny;
  int z = (int)((unsigned)p / slice);
  const unsigned r = (unsigned)p - (unsigned)z * slice;
  int y = (int)(r / (unsigned)d.nx), x = (int)(r - (unsigned)y * (unsigned)d.nx);
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    c.x[k] = x, c.y[k] = y, c.z[k] = z;
    if (++x == d.nx) {
      x = 0;
      if (++y == d.ny) y = 0, ++z;
    }
  }
  return c;
}

// One pass of a windowed, separable min-plus transform along AXIS (0 x, 1 y, 2 z), 2 rn + 1 taps.
//   SRC 0: taps read the byte field of the previous pass          cost = src + w(t)
//   SRC 1: taps read the int8 grid, occupied voxels (== 100)       cost = w(t)            (distance transforms)
//   SRC 2: taps read the int8 grid, unknown voxels (== -1) that lie at least rn voxels inside the grid (x pass only)
//   SQ   : w(t) = t^2 (Euclidean: squared distance to the nearest source voxel in the cube |d| <= rn), else w = 0
//          (Chebyshev: is there a source voxel in the cube)
//   LAST : 0 write the byte field; otherwise merge into the grid read at the voxel itself and write the grid:
//          1 inflation (VG:252-278): nearest occupied voxel at a length the mask holds -> occupied, unknown voxels too
//          2 potential field (VG:280-297): known voxels take max(own value, mask value at the nearest occupied voxel)
//          3 SetUncertainToUnknown (MB:331-362): not occupied and an inner unknown voxel in the cube -> unknown
template <int AXIS, int SRC, int LAST, bool SQ>
__global__ __launch_bounds__(256) void k_pass(Dims d, int rn, MapTables tb, const int8_t* __restrict__ grid,
                                              const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int8_t* __restrict__ out) {
  const size_t g0 = (size_t)blockIdx.y * d.vox;
  const int stride = (AXIS == 0) ? 1 : (AXIS == 1 ? d.nx : d.nx * d.ny);
  const int len = (AXIS == 0) ? d.nx : (AXIS == 1 ? d.ny : d.nz);
  const uint8_t* taps = (SRC == 0) ? src + g0 : reinterpret_cast<const uint8_t*>(grid) + g0;
  for (int p = (blockIdx.x * blockDim.x + threadIdx.x) * VPT; p < d.vox; p += gridDim.x * blockDim.x * VPT) {
    const int nq = (d.vox - p < VPT) ? d.vox - p : VPT;
    const Pos c = coords(d, p);
    // per voxel: the taps t in [lo, hi] stay inside the grid (and, SRC 2, land on an inner voxel of an inner row)
    int lo[VPT], hi[VPT], lo_all = -rn, hi_all = rn;
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
      const int pos = (AXIS == 0) ? c.x[k] : (AXIS == 1 ? c.y[k] : c.z[k]);
      int a = -pos, b = len - 1 - pos;
      if (SRC == 2) {
        a += rn, b -= rn;
        if (c.y[k] < rn || c.y[k] >= d.ny - rn || c.z[k] < rn || c.z[k] >= d.nz - rn) a = 1, b = 0;
      }
      lo[k] = a > -rn ? a : -rn, hi[k] = b < rn ? b : rn;
      if (k >= nq) lo[k] = 1, hi[k] = 0;
      lo_all = lo[k] > lo_all ? lo[k] : lo_all, hi_all = hi[k] < hi_all ? hi[k] : hi_all;
    }
    u16x2 blo = as_pk(0x00ff00ffu), bhi = as_pk(0x00ff00ffu);
    constexpr int TB = 11;  // taps whose loads are in flight together (the whole window for rn <= 5)
    for (int t0 = -rn; t0 <= rn; t0 += TB) {
      uint32_t w[TB];
#pragma unroll
      for (int u = 0; u < TB; ++u) {
        const int t = t0 + u;
        w[u] = 0xffffffffu;
        if (t > rn) continue;  // (wave-uniform: short windows issue only their own taps)
        // the four taps sit next to each other in memory; at the ends of the grid the load is moved inside and the
        // bytes are shifted back into place (what falls off is masked below: those taps are outside the grid)
        const int q = p + t * stride;
        int qc = q < 0 ? 0 : q;
        qc = (qc > d.vox - VPT) ? d.vox - VPT : qc;
        uint32_t v = load4(taps + qc);
        const int sh = (q - qc) * 8;
        v = (sh == 0) ? v : (sh > 0 ? (sh > 31 ? 0u : v >> sh) : (sh < -31 ? 0u : v << -sh));
        w[u] = v;
      }
#pragma unroll
      for (int u = 0; u < TB; ++u) {
        const int t = t0 + u;
        if (t > rn) continue;
        uint32_t wu = w[u];
        if (SRC == 1) wu = far_unless(wu, 0x64646464u);  // occupied (100) -> 0, anything else -> far
        if (SRC == 2) wu = far_unless(wu, 0xffffffffu);  // unknown (-1) -> 0
        const unsigned short tt = SQ ? (unsigned short)(t * t) : (unsigned short)0;
        const u16x2 add = {tt, tt};
        u16x2 vlo = widen_lo(wu) + add, vhi = widen_hi(wu) + add;  // far + t^2 >= 255 stays "far" after the clamp
        if (t < lo_all || t > hi_all) {  // some voxel of the quad has no such tap
          const uint32_t m0 = (t < lo[0] || t > hi[0]) ? 0x0000ffffu : 0u, m1 = (t < lo[1] || t > hi[1]) ? 0xffff0000u : 0u;
          const uint32_t m2 = (t < lo[2] || t > hi[2]) ? 0x0000ffffu : 0u, m3 = (t < lo[3] || t > hi[3]) ? 0xffff0000u : 0u;
          vlo = as_pk(as_u32(vlo) | m0 | m1), vhi = as_pk(as_u32(vhi) | m2 | m3);
        }
        blo = __builtin_elementwise_min(blo, vlo);
        bhi = __builtin_elementwise_min(bhi, vhi);
      }
    }
    const uint32_t best4 = narrow(blo, bhi);  // four squared distances (or 255)
    if (LAST == 0) {
      if (nq == VPT) store4(dst + g0 + p, best4);
      else
        for (int k = 0; k < nq; ++k) dst[g0 + p + k] = (uint8_t)byte_of(best4, k);
    } else {
      // merge into the grid read at the voxels themselves
      uint32_t g4 = 0;
      if (nq == VPT) g4 = load4(grid + g0 + p);
      else
        for (int k = 0; k < nq; ++k) g4 |= (uint32_t)(uint8_t)grid[g0 + p + k] << (8 * k);
      uint32_t r4 = 0;
#pragma unroll
      for (int k = 0; k < VPT; ++k) {
        const int8_t v = (int8_t)byte_of(g4, k);
        const int bk = byte_of(best4, k);
        int8_t r = v;
        if (bk != kFar) {
          if (LAST == 1) {
            if (tb.in1[bk]) r = 100;
          } else if (LAST == 2) {
            if (v != -1) {
              const int8_t pv = tb.val2[bk];
              if (pv > v) r = pv;
            }
          } else if (v != 100) {
            r = -1;
          }
        }
        r4 |= (uint32_t)(uint8_t)r << (8 * k);
      }
      if (nq == VPT) store4(out + g0 + p, r4);
      else
        for (int k = 0; k < nq; ++k) out[g0 + p + k] = (int8_t)byte_of(r4, k);
    }
  }
}

thread_local std::string g_map_err;

// CreateMask (VG:192-226) folded into tables indexed by squared length; fails if two offsets of equal squared length
// disagree (they do not for any setting tried; the check keeps the equivalence honest).
int build_tables(const hdsm_map_config& c, MapTables* tb) {
  std::memset(tb, 0, sizeof *tb);
  const double res = c.voxel_size;
  tb->rn0 = (int)std::ceil(c.inflation_dist / res);
  tb->rn1 = (int)std::ceil(c.inflation_dist / res);
  tb->rn2 = (int)std::ceil(c.potential_dist / res);
  if (tb->rn1 > kRnMax || tb->rn2 > kRnMax || tb->rn1 < 0 || tb->rn2 < 0) {
    g_map_err = "mask radius above 9 voxels";
    return HDSM_ERR_BAD_ARG;
  }
  for (int pass = 0; pass < 2; ++pass) {
    const double mask_dist = pass == 0 ? c.inflation_dist : c.potential_dist;
    const double power = pass == 0 ? 1.0 : (double)c.potential_pow;
    const int rn = pass == 0 ? tb->rn1 : tb->rn2;
    int seen[256];
    for (int k = 0; k < 256; ++k) seen[k] = -1000;
    if (!(mask_dist > 0)) continue;
    for (int n0 = -rn; n0 <= rn; ++n0)
      for (int n1 = -rn; n1 <= rn; ++n1)
        for (int n2 = -rn; n2 <= rn; ++n2) {
          double dist = std::hypot(std::hypot((double)n0, (double)n1), (double)n2);
          dist = std::fabs(dist - 1);
          int val = -999;  // not in the mask
          if (!(dist * res >= mask_dist)) {
            const double h = 100.0 * std::pow(1 - std::hypot(std::hypot((double)n0, (double)n1), (double)n2) / (rn + 1), power);
            if (h > 1e-3) val = (int)(int8_t)h;
          }
          const int s = n0 * n0 + n1 * n1 + n2 * n2;
          if (seen[s] == -1000) seen[s] = val;
          else if (seen[s] != val) {
            g_map_err = "mask is not a function of the squared offset length for these parameters";
            return HDSM_ERR_BAD_ARG;
          }
        }
    int last = 1000;
    for (int s = 0; s < 256; ++s) {
      if (seen[s] == -1000 || seen[s] == -999) continue;
      if (pass == 0) tb->in1[s] = 1;
      else tb->val2[s] = (int8_t)seen[s];
      if (s > 0) {  // the nearest-voxel argument needs a value that does not increase with the length
        if (seen[s] > last) {
          g_map_err = "mask value increases with the offset length";
          return HDSM_ERR_BAD_ARG;
        }
        last = seen[s];
      }
    }
    // the origin (length 0) is in the mask only if voxel_size < mask_dist; an occupied voxel keeps its value either way
    // membership must also be "closed downwards" from the largest length held (lengths >= 1)
    bool ended = false;
    for (int s = 1; s < 256; ++s) {
      if (seen[s] == -1000) continue;
      const bool in = seen[s] != -999;
      if (!in) ended = true;
      else if (ended) {
        g_map_err = "mask membership is not monotone in the offset length";
        return HDSM_ERR_BAD_ARG;
      }
    }
    if (pass == 0 && seen[0] == -999) tb->in1[0] = 1;  // length 0 = the voxel itself is occupied: stays occupied
    if (pass == 1 && seen[0] == -999) tb->val2[0] = 0;
  }
  return HDSM_OK;
}

int run_device(const hdsm_map_config* cfg, int32_t n_grids, const int32_t dim[3], const int8_t* d_in, int8_t* d_out,
               uint8_t* scratch, hipStream_t st) {
  MapTables tb;
  if (int rc = build_tables(*cfg, &tb)) return rc;
  const int64_t vox64 = (int64_t)dim[0] * dim[1] * dim[2];
  if (vox64 < VPT) {
    g_map_err = "grids of fewer than 4 voxels are not supported";
    return HDSM_ERR_BAD_ARG;
  }
  if (vox64 >= (int64_t)1 << 30 || n_grids > 65535) {
    g_map_err = "grid too large (>= 2^30 voxels) or more than 65535 grids in one batch";
    return HDSM_ERR_CAPACITY;
  }
  Dims d{dim[0], dim[1], dim[2], (int)vox64};
  const size_t total = (size_t)vox64 * n_grids;
  uint8_t* a = scratch;
  uint8_t* b = scratch + total;
  int8_t* stage0 = reinterpret_cast<int8_t*>(b);  // output of SetUncertainToUnknown, until the inflation is applied
  const int quads = (int)((vox64 + VPT - 1) / VPT);
  const dim3 grid((unsigned)std::min<int>((quads + 255) / 256, 4096), (unsigned)n_grids), block(256);
  const uint8_t* nou = nullptr;
  const int8_t* nog = nullptr;
  uint8_t* o8 = reinterpret_cast<uint8_t*>(d_out);
  int8_t* no8 = nullptr;
  uint8_t* nod = nullptr;
  // SetUncertainToUnknown: box dilation of the inner unknown voxels, merged into scratch b (= stage0)
  hipLaunchKernelGGL((k_pass<0, 2, 0, false>), grid, block, 0, st, d, tb.rn0, tb, d_in, nou, a, no8);
  hipLaunchKernelGGL((k_pass<1, 0, 0, false>), grid, block, 0, st, d, tb.rn0, tb, nog, a, o8, no8);
  hipLaunchKernelGGL((k_pass<2, 0, 3, false>), grid, block, 0, st, d, tb.rn0, tb, d_in, o8, nod, stage0);
  // inflation: squared distances to the occupied voxels (x, y passes), merged by the z pass into scratch a
  hipLaunchKernelGGL((k_pass<0, 1, 0, true>), grid, block, 0, st, d, tb.rn1, tb, stage0, nou, a, no8);
  hipLaunchKernelGGL((k_pass<1, 0, 0, true>), grid, block, 0, st, d, tb.rn1, tb, nog, a, o8, no8);
  int8_t* inflated = reinterpret_cast<int8_t*>(a);
  hipLaunchKernelGGL((k_pass<2, 0, 1, true>), grid, block, 0, st, d, tb.rn1, tb, stage0, o8, nod, inflated);
  // potential field on the inflated grid, merged by the z pass into the output
  hipLaunchKernelGGL((k_pass<0, 1, 0, true>), grid, block, 0, st, d, tb.rn2, tb, inflated, nou, o8, no8);
  hipLaunchKernelGGL((k_pass<1, 0, 0, true>), grid, block, 0, st, d, tb.rn2, tb, nog, o8, b, no8);
  hipLaunchKernelGGL((k_pass<2, 0, 2, true>), grid, block, 0, st, d, tb.rn2, tb, inflated, b, nod, d_out);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_map_err = hipGetErrorString(e);
    return HDSM_ERR_DEVICE;
  }
  return HDSM_OK;
}

bool bad_args(const hdsm_map_config* cfg, int32_t n_grids, const int32_t* dim, const void* in, const void* out) {
  return !cfg || !dim || !in || !out || n_grids < 0 || dim[0] < 1 || dim[1] < 1 || dim[2] < 1 || !(cfg->voxel_size > 0) ||
         cfg->inflation_dist < 0 || cfg->potential_dist < 0 || cfg->potential_pow < 0;
}

}  // namespace

extern "C" {

const char* hdsm_map_last_error(void) { return g_map_err.c_str(); }

int hdsm_map_preprocess_device(int32_t device, const hdsm_map_config* cfg, int32_t n_grids, const int32_t dim[3],
                               const int8_t* grids_in, int8_t* grids_out, void* scratch, void* hip_stream) {
  if (bad_args(cfg, n_grids, dim, grids_in, grids_out) || !scratch) return HDSM_ERR_BAD_ARG;
  if (n_grids == 0) return HDSM_OK;
  if (hipSetDevice(device) != hipSuccess) return HDSM_ERR_NO_DEVICE;
  return run_device(cfg, n_grids, dim, grids_in, grids_out, static_cast<uint8_t*>(scratch), static_cast<hipStream_t>(hip_stream));
}

int hdsm_map_preprocess(int32_t device, const hdsm_map_config* cfg, int32_t n_grids, const int32_t dim[3],
                        const int8_t* grids_in, int8_t* grids_out) {
  if (bad_args(cfg, n_grids, dim, grids_in, grids_out)) return HDSM_ERR_BAD_ARG;
  if (n_grids == 0) return HDSM_OK;
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= device || hipSetDevice(device) != hipSuccess) {
    g_map_err = "no HIP device";
    return HDSM_ERR_NO_DEVICE;
  }
  const size_t total = (size_t)dim[0] * dim[1] * dim[2] * n_grids;
  int8_t *d_in = nullptr, *d_out = nullptr;
  uint8_t* d_scr = nullptr;
  hipError_t e = hipMalloc(&d_in, total);
  if (e == hipSuccess) e = hipMalloc(&d_out, total);
  if (e == hipSuccess) e = hipMalloc(&d_scr, 2 * total);
  int rc = HDSM_OK;
  if (e == hipSuccess) e = hipMemcpy(d_in, grids_in, total, hipMemcpyHostToDevice);
  if (e == hipSuccess) rc = run_device(cfg, n_grids, dim, d_in, d_out, d_scr, nullptr);
  if (e == hipSuccess && rc == HDSM_OK) e = hipMemcpy(grids_out, d_out, total, hipMemcpyDeviceToHost);
  (void)hipFree(d_in), (void)hipFree(d_out), (void)hipFree(d_scr);
  if (e != hipSuccess) {
    g_map_err = hipGetErrorString(e);
    return HDSM_ERR_DEVICE;
  }
  return rc;
}

}  // extern "C"
