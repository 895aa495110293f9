// corridor_host.cpp — next row f2: the convex voxel decomposition the reference's corridor generator calls for every
// seed (GenerateSafeCorridor, agent_class.cpp:1236-1447 -> convex_decomp_lib::GetPolyOcta3D,
// convex_decomp_util/src/convex_decomp.cpp:5-376, and its shape-aware variant GetPolyOcta3DNew, :590-1160, with the
// helpers FindCorners :378-564 and SideIsEmpty :577-588; "CD" below). Host code, plain C ABI (include/hdsm_swarm.h).
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
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <vector>

#include "../../include/hdsm_swarm.h"

namespace {

struct Cell {
  int x, y, z;
  bool operator==(const Cell& o) const { return x == o.x && y == o.y && z == o.z; }
};
inline Cell operator+(Cell a, Cell b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Cell operator-(Cell a, Cell b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Cell neg(Cell a) { return {-a.x, -a.y, -a.z}; }
inline int dot(Cell a, Cell b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

constexpr int kOccupied = 100;  // CVX_DCMP_OCC (convex_decomp.hpp:11): values below it are free

// outward normals in the order of the output rows (CD:17-18)
const Cell kNormal[6] = {{0, -1, 0}, {1, 0, 0}, {0, 1, 0}, {-1, 0, 0}, {0, 0, 1}, {0, 0, -1}};
// the two faces meeting in edge e (numbering of CD:28-31: it fixes the order of the chamfer rows)
const int kEdgeFaces[12][2] = {{0, 1}, {0, 4}, {0, 3}, {0, 5}, {1, 5}, {1, 4}, {3, 4}, {3, 5}, {1, 2}, {2, 4}, {2, 3}, {2, 5}};

struct Frame {      // in-plane frame of a face and what lies across each of its four sides
  Cell side[4];     // +u, +v, -u, -v
  int face[4];      // neighbouring face across that side
  int edge[4];      // edge shared with that neighbour
  int back[4];      // index, among the NEIGHBOUR's sides, of the direction this face grows in
};

int face_with_normal(Cell n) {
  for (int f = 0; f < 6; ++f)
    if (kNormal[f] == n) return f;
  return -1;
}

// u = normal of the next lateral face and v = +z for the four lateral faces; (-y, +x) / (-y, -x) for top / bottom
// (CD:33-41). Everything else follows from that.
void build_frames(Frame fr[6]) {
  for (int f = 0; f < 6; ++f) {
    Cell u, v;
    if (f < 4) u = kNormal[(f + 1) % 4], v = Cell{0, 0, 1};
    else u = kNormal[0], v = (f == 4) ? kNormal[1] : kNormal[3];
    fr[f].side[0] = u, fr[f].side[1] = v, fr[f].side[2] = neg(u), fr[f].side[3] = neg(v);
  }
  for (int f = 0; f < 6; ++f)
    for (int j = 0; j < 4; ++j) {
      const int g = face_with_normal(fr[f].side[j]);
      fr[f].face[j] = g;
      fr[f].edge[j] = -1;
      for (int e = 0; e < 12; ++e)
        if ((kEdgeFaces[e][0] == f && kEdgeFaces[e][1] == g) || (kEdgeFaces[e][0] == g && kEdgeFaces[e][1] == f))
          fr[f].edge[j] = e;
      fr[f].back[j] = -1;
      for (int k = 0; k < 4; ++k)
        if (fr[g].side[k] == kNormal[f]) fr[f].back[j] = k;
    }
}

const Frame* frames() {
  static Frame fr[6];
  static const bool ready = (build_frames(fr), true);
  (void)ready;
  return fr;
}

struct Edge {      // Corner3D (convex_decomp.hpp:21-36)
  double pos[3] = {0, 0, 0};
  int slope = 0;   // 0 = square edge
  int dir = -1;    // face along which the chamfer runs `slope` voxels per voxel of the other face; -1 = undecided
  bool fixed = false;
  int steps = 0;   // voxels taken on the current stair
};

struct Grid {
  int8_t* data;
  int nx, ny, nz;
  bool inside(Cell c) const { return c.x >= 0 && c.y >= 0 && c.z >= 0 && c.x < nx && c.y < ny && c.z < nz; }
  int8_t& at(Cell c) const { return data[c.x + c.y * nx + c.z * nx * ny]; }
};

struct FaceState {             // Border3D (convex_decomp.hpp:39-43)
  std::vector<Cell> outer;     // outermost layer of the face
  int reach[4];                // extent of that layer along the face's four sides (dot products)
};

// how far the next layer of face f may extend on each side, given the chamfers already started (CD:71-91)
void allowance(const Frame* fr, int f, const FaceState& fs, const Edge* edges, int allow[4], Edge trial[4]) {
  for (int j = 0; j < 4; ++j) {
    allow[j] = fs.reach[j];
    const Edge& e = trial[j] = edges[fr[f].edge[j]];
    if (e.slope > 0) {
      if (e.dir == f) allow[j] -= e.slope;                         // our layers retreat `slope` voxels each
      else if (e.fixed && e.steps >= e.slope) allow[j] -= 1;       // the other face's stair is complete: step in
    }
  }
}

struct Layer {
  bool found = false;
  std::vector<Cell> cells;      // border_real_tmp
  std::deque<Cell> rim_real[4]; // borders_2d_real
  Cell far[4];                  // border_limit_tmp
};

// One layer on top of face f: a free 2-D seed above the current outer layer, inside the allowance and inside voxels
// [1, dim - 1 - margin] (CD:94-116: margin 1; CD:700-723: margin 0), grown in its plane (CD:118-200).
Layer grow_layer(const Grid& g, const Frame* fr, int f, const FaceState& fs, const int allow[4], int mark, int margin) {
  Layer L;
  const Cell up = kNormal[f];
  const Cell* sd = fr[f].side;
  Cell s2{0, 0, 0};
  for (const Cell& c : fs.outer) {
    const Cell t = c + up;
    if (t.x < 1 || t.y < 1 || t.z < 1 || t.x >= g.nx - margin || t.y >= g.ny - margin || t.z >= g.nz - margin) continue;
    if (g.at(t) >= kOccupied) continue;
    bool in = true;
    for (int k = 0; k < 4; ++k) in = in && dot(t, sd[k]) <= allow[k];
    if (in) {
      s2 = t, L.found = true;
      break;
    }
  }
  if (!L.found) return L;
  std::deque<Cell> rim[4];  // current outline of the layer per side: all cells (rim) / cells of the layer (rim_real)
  for (int j = 0; j < 4; ++j) rim[j].assign(1, s2), L.rim_real[j].assign(1, s2), L.far[j] = s2;
  L.cells.assign(1, s2);
  bool alive[4] = {true, true, true, true};
  for (int k = 0; alive[0] || alive[1] || alive[2] || alive[3]; ++k) {
    const int s = k % 4, prev = (k + 3) % 4, next = (k + 1) % 4;
    std::deque<Cell> moved, moved_real;
    bool ok = true;
    for (const Cell& c : rim[s]) {
      const Cell t = c + sd[s];
      if (dot(t, sd[s]) > allow[s]) {
        ok = false;
        break;
      }
      const Cell below = t - up;
      if (g.inside(below) && g.at(below) == (int8_t)mark) {  // on top of the polyhedron: must be free
        if (g.inside(t) && g.at(t) < kOccupied) {
          moved.push_back(t), moved_real.push_back(t);
        } else {
          ok = false;
          break;
        }
      } else {
        moved.push_back(t);  // beside the polyhedron: carried along, not part of the layer
      }
    }
    if (!ok) {
      alive[s] = false;  // (a side that failed is still tried again on later turns, as in CD:128-133)
      continue;
    }
    rim[s] = moved;
    L.cells.insert(L.cells.end(), moved_real.begin(), moved_real.end());
    L.rim_real[s] = moved_real;
    rim[prev].push_back(moved.front());
    rim[next].push_front(moved.back());
    if (!moved_real.empty()) {
      if (moved.front() == moved_real.front()) L.rim_real[prev].push_back(moved.front());
      if (moved.back() == moved_real.back()) L.rim_real[next].push_front(moved.back());
    }
    for (int j = 0; j < 4; ++j)
      if (!L.rim_real[j].empty()) L.far[j] = L.rim_real[j].front();
  }
  return L;
}

// |  |l0| - |l2|  | * |  |l1| - |l3|  |   (CD:691-693)
double span_area(const int l[4]) {
  return std::fabs(std::fabs((double)l[0]) - std::fabs((double)l[2])) * std::fabs(std::fabs((double)l[1]) - std::fabs((double)l[3]));
}

// extents of the grown layer as the reference reads them for its area test (CD:816-820): the front cell of every
// side. A side without layer cells has no front in the reference (it reads an empty deque there); the last known
// front (border_limit_tmp) stands in for it.
void layer_extent(const Frame* fr, int f, const Layer& L, int ext[4]) {
  for (int j = 0; j < 4; ++j) ext[j] = dot(L.rim_real[j].empty() ? L.far[j] : L.rim_real[j].front(), fr[f].side[j]);
}

// SideIsEmpty, CD:577-588 (GetVoxel: outside the grid = occupied; any positive value, potential field included, counts)
bool side_is_empty(const Grid& g, const std::deque<Cell>& cells, Cell step) {
  if (cells.empty()) return false;
  for (const Cell& c : cells) {
    const Cell t = c + step;
    const int v = g.inside(t) ? (int)g.at(t) : kOccupied;
    if (v > 0) return false;
  }
  return true;
}

// FindCorners, CD:378-564: a trial layer on face f from the given state; which square edges would become chamfers.
void find_corners(const Grid& g, const Frame* fr, int f, const bool growing[6], const FaceState faces[6], const Edge* edges,
                  int mark, bool& valid, Edge out[4]) {
  if (!growing[f]) {
    valid = false;
    return;
  }
  int allow[4];
  allowance(fr, f, faces[f], edges, allow, out);
  const double area = span_area(allow);
  const Layer L = grow_layer(g, fr, f, faces[f], allow, mark, 0);
  if (!L.found) return;
  int ext[4];
  layer_extent(fr, f, L, ext);
  if (span_area(ext) < area / 2) valid = false;
  for (int j = 0; j < 4; ++j) {
    if (L.rim_real[j].empty()) continue;
    const int gap = faces[f].reach[j] - dot(L.rim_real[j].front(), fr[f].side[j]);
    if (out[j].slope == 0 && gap > 0) {
      out[j].slope = out[j].steps = gap;
      if (gap > 1) out[j].dir = f;
    }
  }
}

int decompose(int variant, const int32_t seed_in[3], int8_t* grid, const int32_t dim[3], int32_t n_it, double res, int32_t mark,
              const double origin[3], double* rows, int32_t max_rows, int32_t* n_rows) {
  if (!seed_in || !grid || !dim || !origin || !rows || !n_rows || n_it < 0 || !(res > 0) || mark >= kOccupied)
    return HDSM_ERR_BAD_ARG;
  Grid g{grid, dim[0], dim[1], dim[2]};
  const Cell seed{seed_in[0], seed_in[1], seed_in[2]};
  if (!g.inside(seed)) return HDSM_ERR_BAD_ARG;
  const Frame* fr = frames();
  const bool aware = variant != 0;

  FaceState faces[6];
  Cell anchor[6];              // a voxel of the outermost layer (gives the face plane)
  Edge edges[12];
  bool growing[6];
  for (int f = 0; f < 6; ++f) {
    faces[f].outer.assign(1, seed);
    anchor[f] = seed;
    growing[f] = true;
    for (int j = 0; j < 4; ++j) faces[f].reach[j] = dot(seed, fr[f].side[j]);
  }
  g.at(seed) = (int8_t)mark;

  for (int it = 0; it < n_it; ++it) {
    const int f = it % 6;
    if (!growing[f]) continue;
    const Cell up = kNormal[f];
    const Cell* sd = fr[f].side;

    int allow[4];
    Edge trial[4];
    allowance(fr, f, faces[f], edges, allow, trial);
    const Layer L = grow_layer(g, fr, f, faces[f], allow, mark, aware ? 0 : 1);
    if (!L.found) continue;

    bool soft = true;  // shape-aware variant: layer acceptable this turn
    if (aware) {
      int ext[4];
      layer_extent(fr, f, L, ext);
      if (span_area(ext) < span_area(allow) / 2) soft = false;
    }

    // is the layer consistent with ONE plane through every edge? (CD:209-283 / CD:828-910)
    bool accept = true;
    int fresh[4] = {0, 0, 0, 0};  // corner_new_state: 1 = a one-voxel chamfer starts on this side, 2 = a longer one
    for (int j = 0; j < 4 && accept; ++j) {
      if (L.rim_real[j].empty()) continue;
      Edge e = trial[j];
      const int gap = faces[f].reach[j] - dot(L.rim_real[j].front(), sd[j]);  // voxels this layer falls short of the last one
      if (e.slope == 0) {
        if (gap > 0) {  // a chamfer starts here: a point of its plane, between this layer and the neighbouring face
          const Cell c = L.rim_real[j].front();
          const Cell nb = kNormal[fr[f].face[j]];
          const double extra = aware ? res / 2 : 0.0;  // (CD:836-848 adds res/2 twice)
          e.pos[0] = c.x * res - up.x * res / 2 + nb.x * res / 2 + res / 2 + extra;
          e.pos[1] = c.y * res - up.y * res / 2 + nb.y * res / 2 + res / 2 + extra;
          e.pos[2] = c.z * res - up.z * res / 2 + nb.z * res / 2 + res / 2 + extra;
          e.slope = e.steps = gap;
          if (gap > 1) e.dir = f;
          fresh[j] = gap > 1 ? 2 : 1;
          if (aware && it < 6) soft = false;  // no chamfer during the first round of turns
        }
      } else if (e.fixed) {
        if (e.dir == f || e.dir == -1) {
          if (gap > e.slope) {
            if (aware) break;  // CD:875-878: the scan of the edges ends here, this and the later sides keep their state
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
        if (gap == 0) e.dir = fr[f].face[j], e.steps += 1, e.slope += 1;
        else if (gap == 1) e.fixed = true;
        else accept = false;
      } else if (e.dir == f) {            // first layer after our own multi-voxel retreat fixes the slope
        e.slope = gap, e.fixed = true;
      } else {                            // the other face is the long direction and is still lengthening its stair
        if (gap == 0) e.slope += 1, e.steps += 1;
        else if (gap == 1) e.fixed = true, e.steps = 1;
        else accept = false;
      }
      if (accept) trial[j] = e;
    }
    if (!accept) {
      growing[f] = false;
      continue;
    }
    if (!soft) continue;

    if (aware) {  // CD:930-1066: does every chamfer that starts with this layer follow a real obstacle?
      bool expand = true;
      for (int j = 0; j < 4; ++j) {
        if (!fresh[j]) continue;
        const bool first = side_is_empty(g, L.rim_real[j], up);
        const int nbf = fr[f].face[j], ci = fr[f].back[j];
        bool second = true;
        if (fresh[j] == 1) {
          std::deque<Cell> edge_row;  // the neighbouring face's cells along the shared edge
          for (const Cell& c : faces[nbf].outer)
            if (dot(c, fr[nbf].side[ci]) == faces[nbf].reach[ci]) edge_row.push_back(c);
          second = side_is_empty(g, edge_row, kNormal[nbf]);
        }
        expand = !(first && second);
        if (!expand) {
          growing[f] = false;
          break;
        }
      }
      if (expand && (fresh[0] || fresh[1] || fresh[2] || fresh[3])) {
        // trial: put the layer in, grow one more on top of it, take it out again
        std::vector<int8_t> saved(L.cells.size());
        for (size_t k = 0; k < L.cells.size(); ++k) saved[k] = g.at(L.cells[k]), g.at(L.cells[k]) = (int8_t)mark;
        FaceState faces_t[6];
        for (int k = 0; k < 6; ++k) faces_t[k] = faces[k];
        faces_t[f].outer = L.cells;
        for (int j = 0; j < 4; ++j) faces_t[f].reach[j] = dot(L.far[j], sd[j]);
        Edge edges_t[12];
        for (int k = 0; k < 12; ++k) edges_t[k] = edges[k];
        for (int j = 0; j < 4; ++j)
          if (!fresh[j]) edges_t[fr[f].edge[j]] = trial[j];
        bool valid = true;
        Edge fin[4];
        find_corners(g, fr, f, growing, faces_t, edges_t, mark, valid, fin);
        for (size_t k = 0; k < L.cells.size(); ++k) g.at(L.cells[k]) = saved[k];
        if (valid) {
          for (int j = 0; j < 4; ++j)
            if (fresh[j] == 2 && fin[j].slope < trial[j].slope) {
              expand = false;
              growing[f] = false;
              break;
            }
          if (expand)
            for (int j = 0; j < 4; ++j) {
              if (fresh[j] != 1) continue;
              const int nbf = fr[f].face[j];
              bool v2 = true;
              Edge fin2[4];
              find_corners(g, fr, nbf, growing, faces_t, edges, mark, v2, fin2);
              if (v2 && fin2[fr[f].back[j]].slope == 0 && fin[j].slope == 0) {
                expand = false;
                break;
              }
            }
        }
      }
      if (!expand) continue;
    }

    faces[f].outer = L.cells;
    for (int j = 0; j < 4; ++j) {
      faces[f].reach[j] = dot(L.far[j], sd[j]);
      edges[fr[f].edge[j]] = trial[j];
      // full-width side on a square edge: these voxels are now also the outermost layer of the neighbouring face
      if (trial[j].slope == 0 && !L.rim_real[j].empty() && faces[f].reach[j] == dot(L.rim_real[j].front(), sd[j])) {
        const int nbf = fr[f].face[j];
        faces[nbf].outer.insert(faces[nbf].outer.end(), L.rim_real[j].begin(), L.rim_real[j].end());
        faces[nbf].reach[fr[f].back[j]] += 1;
      }
    }
    anchor[f] = L.cells.front();
    for (const Cell& c : L.cells) g.at(c) = (int8_t)mark;
  }

  // half-spaces: chamfered edges first (edge numbering order), then the six faces (CD:322-373)
  int n = 0;
  auto emit = [&](const double nrm[3], const double p[3]) {
    if (n < max_rows) {
      double* r = rows + 4 * (size_t)n;
      r[0] = nrm[0], r[1] = nrm[1], r[2] = nrm[2];
      r[3] = nrm[0] * p[0] + nrm[1] * p[1] + nrm[2] * p[2];
    }
    ++n;
  };
  for (int e = 0; e < 12; ++e) {
    if (edges[e].slope <= 0) continue;
    const int fa = kEdgeFaces[e][0], fb = kEdgeFaces[e][1];
    const int lng = (edges[e].dir == fa) ? fa : fb, oth = (edges[e].dir == fa) ? fb : fa;
    const double nrm[3] = {(double)(edges[e].slope * kNormal[lng].x + kNormal[oth].x),
                           (double)(edges[e].slope * kNormal[lng].y + kNormal[oth].y),
                           (double)(edges[e].slope * kNormal[lng].z + kNormal[oth].z)};
    const double p[3] = {edges[e].pos[0] + origin[0], edges[e].pos[1] + origin[1], edges[e].pos[2] + origin[2]};
    emit(nrm, p);
  }
  for (int f = 0; f < 6; ++f) {
    const double nrm[3] = {(double)kNormal[f].x, (double)kNormal[f].y, (double)kNormal[f].z};
    const double p[3] = {anchor[f].x * res + kNormal[f].x * res / 2 + res / 2 + origin[0],
                         anchor[f].y * res + kNormal[f].y * res / 2 + res / 2 + origin[1],
                         anchor[f].z * res + kNormal[f].z * res / 2 + res / 2 + origin[2]};
    emit(nrm, p);
  }
  *n_rows = n;
  return n <= max_rows ? HDSM_OK : HDSM_ERR_CAPACITY;
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
