// hdsm_level1.cpp — see hdsm_level1.h. Pure host code.
#include "hdsm_level1.h"

#include <algorithm>
#include <cstring>

namespace hdsm {
namespace {
int fail(const char** err, const char* msg) {
  if (err) *err = msg;
  return HDSM_ERR_BAD_ARG;
}
}  // namespace

int level1_split(const hdsm_params& prm, int n_inst, int r_max, const int32_t* n_poly, const int32_t* n_rows,
                 const double* A, const double* b, Level1Split* out, const char** err) {
  const int N = prm.n_hor, P = prm.poly_hor, RS = prm.max_rows_static;
  if (r_max < 1) return fail(err, "r_max must be >= 1");
  auto rowA = [&](int k, int i, int j, int r) { return A + ((((size_t)k * N + i) * P + j) * r_max + r) * 3; };
  auto rowb = [&](int k, int i, int j, int r) { return b[(((size_t)k * N + i) * P + j) * r_max + r]; };
  auto same_row = [&](int k, int i1, int j1, int r1, int i2, int j2, int r2) {
    return std::memcmp(rowA(k, i1, j1, r1), rowA(k, i2, j2, r2), 3 * sizeof(double)) == 0 &&
           rowb(k, i1, j1, r1) == rowb(k, i2, j2, r2);
  };
  out->n_poly.assign(n_inst, 0);
  out->n_rows_static.assign((size_t)n_inst * P, 0);
  out->A_static.assign((size_t)n_inst * P * RS * 3, 0.0);
  out->b_static.assign((size_t)n_inst * P * RS, 0.0);
  out->n_common.assign((size_t)n_inst * N, 0);
  // pass 1: common suffix per (instance, step)
  int rc_max = 1;
  for (int k = 0; k < n_inst; ++k)
    for (int i = 0; i < N; ++i) {
      const int m = std::min(P, n_poly[(size_t)k * N + i]);  // AC:913
      if (m < 0) return fail(err, "negative n_poly");
      if (m != std::min(P, n_poly[(size_t)k * N])) return fail(err, "n_poly differs between steps (AC:1098 copies the same list)");
      int nc = 0;
      if (m > 0) {
        int rmin = r_max;
        for (int j = 0; j < m; ++j) {
          const int r = n_rows[((size_t)k * N + i) * P + j];
          if (r < 0 || r > r_max) return fail(err, "n_rows out of range");
          rmin = std::min(rmin, r);
        }
        for (; nc < rmin; ++nc) {  // grow the suffix while row (R_j - 1 - nc) is identical in every polyhedron
          bool all = true;
          const int r0 = n_rows[((size_t)k * N + i) * P] - 1 - nc;
          for (int j = 1; j < m && all; ++j) all = same_row(k, i, 0, r0, i, j, n_rows[((size_t)k * N + i) * P + j] - 1 - nc);
          if (!all) break;
        }
        if (m == 1) {  // a single polyhedron: every row is "common"; keep the head that matches step 0's static part
          nc = 0;     // (decided in pass 2 against step 0)
        }
      }
      out->n_common[(size_t)k * N + i] = nc;
    }
  // pass 2: static heads must be the same polyhedra at every step; for m == 1 the head is the longest prefix
  // shared by all steps
  for (int k = 0; k < n_inst; ++k) {
    const int m = std::min(P, n_poly[(size_t)k * N]);
    out->n_poly[k] = m;
    for (int j = 0; j < m; ++j) {
      int head = n_rows[((size_t)k * N) * P + j] - out->n_common[(size_t)k * N];
      if (m == 1) {
        head = n_rows[((size_t)k * N) * P];
        for (int i = 1; i < N; ++i) {
          int h = 0;
          const int ri = n_rows[((size_t)k * N + i) * P];
          while (h < head && h < ri && same_row(k, 0, 0, h, i, 0, h)) ++h;
          head = h;
        }
        head = std::min(head, RS);
        for (int i = 0; i < N; ++i) out->n_common[(size_t)k * N + i] = n_rows[((size_t)k * N + i) * P] - head;
      }
      if (head < 0 || head > RS) return fail(err, "static part of a polyhedron exceeds max_rows_static");
      for (int i = 0; i < N; ++i) {
        if (n_rows[((size_t)k * N + i) * P + j] - out->n_common[(size_t)k * N + i] != head)
          return fail(err, "static polyhedra differ between steps: not the shape AC:1098 produces");
        for (int r = 0; r < head; ++r)
          if (!same_row(k, 0, j, r, i, j, r)) return fail(err, "static polyhedra differ between steps: not the shape AC:1098 produces");
      }
      out->n_rows_static[(size_t)k * P + j] = head;
      for (int r = 0; r < head; ++r) {
        std::memcpy(&out->A_static[(((size_t)k * P + j) * RS + r) * 3], rowA(k, 0, j, r), 3 * sizeof(double));
        out->b_static[((size_t)k * P + j) * RS + r] = rowb(k, 0, j, r);
      }
    }
    for (int i = 0; i < N; ++i) rc_max = std::max(rc_max, out->n_common[(size_t)k * N + i]);
  }
  out->rc_max = rc_max;
  out->common.assign((size_t)n_inst * N * rc_max * 4, 0.0);
  for (int k = 0; k < n_inst; ++k) {
    if (out->n_poly[k] == 0) continue;
    for (int i = 0; i < N; ++i) {
      const int nc = out->n_common[(size_t)k * N + i];
      const int R0 = n_rows[((size_t)k * N + i) * P];
      for (int r = 0; r < nc; ++r) {
        double* dst = &out->common[(((size_t)k * N + i) * rc_max + r) * 4];
        std::memcpy(dst, rowA(k, i, 0, R0 - nc + r), 3 * sizeof(double));
        dst[3] = rowb(k, i, 0, R0 - nc + r);
      }
    }
  }
  return HDSM_OK;
}

}  // namespace hdsm
