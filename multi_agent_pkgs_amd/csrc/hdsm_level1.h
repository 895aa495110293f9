// hdsm_level1.h — host-side preparation of the level-1 input (hdsm_solve): fully formed per-step polyhedra
// poly_const_final_vec_[N][<=P] (AH:471) are split into
//   * the rows shared by EVERY polyhedron of a step (bitwise-identical trailing rows: exactly what
//     Agent::AddHyperplane appends to each polyhedron, AC:1217-1234) -> explicit "common" rows of that step;
//   * the remaining head of each polyhedron -> the static polyhedra of the level-2 kernel.
// The reference copies the same poly_const_vec_ into every step (AC:1098), so the heads must be identical
// across steps; input that is not of that shape is rejected with HDSM_ERR_BAD_ARG.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/hdsm.h"

namespace hdsm {
struct Level1Split {
  int rc_max = 0;                       // common-row capacity per step
  std::vector<int32_t> n_poly;          // [n_inst]
  std::vector<int32_t> n_rows_static;   // [n_inst][P]
  std::vector<double> A_static;         // [n_inst][P][RS][3]
  std::vector<double> b_static;         // [n_inst][P][RS]
  std::vector<int32_t> n_common;        // [n_inst][N]
  std::vector<double> common;           // [n_inst][N][rc_max][4]
};
int level1_split(const hdsm_params& prm, int n_inst, int r_max, const int32_t* n_poly, const int32_t* n_rows,
                 const double* A, const double* b, Level1Split* out, const char** err);
}  // namespace hdsm
