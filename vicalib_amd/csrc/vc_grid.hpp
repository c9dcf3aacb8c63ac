// vc_grid.hpp -- host-side association of detected dots with the calibration target's grid.
//
// Stands in for calibu::TargetGridDot::FindTarget at its call site vicalib-task.cc:274-277 (input: the conics of one image, output:
// `ellipse_target_map`, the target-dot index of every conic or -1) and for calibu::MakePattern (vicalib-engine.cc:459-461).  Calibu's
// source is not in the reference tree, so this is an implementation of the *published idea* of its dot target -- a regular grid of
// dots of two sizes whose large / small pattern makes every sufficiently large window of the grid unique -- not a restatement of its
// code: "parity unpinned" (DESIGN 4.4).  What it must do, and what the tests hold it to (tests/test_grid_cpu.py): every detected dot
// of a rendered view gets its true grid index under perspective, with 10 % of the dots missing and a few false detections present.
//
//   1. lattice walk   From a seed dot with two pairs of opposite neighbours the grid coordinates spread breadth-first: a dot at
//                     integer position (i, j) with local lattice vectors (u, v) predicts its four neighbours at c +- u, c +- v (and,
//                     across a missing dot, at c +- 2u, c +- 2v); the detection nearest to a prediction, within a fraction of the
//                     step, takes the position, and inherits lattice vectors refreshed from the step actually taken (perspective
//                     changes them slowly from dot to dot).  The largest component over a few seeds wins.
//   2. dot sizes      area of the dot's image ellipse over the area of its local lattice cell |u x v| -- invariant under the view's
//                     local affine map -- splits into two clusters (1-D 2-means): large and small.
//   3. pattern match  the observed binary grid against the target's pattern under the four in-plane rotations and every offset that
//                     keeps it inside the target: the placement with the most agreeing dots wins, if it explains >= 85 % of them
//                     and beats the runner-up clearly.
// Front-end code: runs once per image on the CPU (a few hundred dots), never inside the solver loop.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <queue>
#include <vector>

namespace vc {

// The large / small pattern of a rows x cols target from a seed: 1 = large dot.  (calibu::MakePattern's generator is not in the
// reference tree; this one is splitmix64 on (seed, row, col), about one dot in three large -- a printed Calibu target needs its own
// pattern passed in, see apps/vicalib.cpp -grid_pattern_file.)
inline void grid_make_pattern(int rows, int cols, unsigned seed, int* out) {
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) {
      uint64_t z = ((uint64_t)seed << 32) ^ ((uint64_t)(unsigned)r << 16) ^ (uint64_t)(unsigned)c;
      z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; z ^= z >> 31;
      out[r * cols + c] = (z % 100) < 36 ? 1 : 0;
    }
}

// area of the ellipse x^T C x = 0 (C symmetric 3 x 3, any scale); <= 0 if C is not an ellipse
inline double grid_conic_area(const double* C) {
  const double a = C[0], b = C[1], c = C[4], d = C[2], e = C[5], f = C[8];
  const double det2 = a * c - b * b;
  if (!(det2 > 0.0)) return 0.0;
  const double det3 = a * (c * f - e * e) - b * (b * f - e * d) + d * (b * e - c * d);
  // centred form (x - x0)^T A (x - x0) = k with k = -det3 / det2; area = pi k / sqrt(det A)
  const double k = -det3 / det2;
  const double area = 3.14159265358979323846 * k / std::sqrt(det2);
  return area > 0.0 ? area : -area;
}

struct GridWalk { std::vector<int> gi, gj; std::vector<double> ux, uy, vx, vy; int count = 0; };

// step 1 from one seed: grid coordinates of every dot reached (gi = INT_MIN: not reached)
inline void grid_walk_from(const double* cen, int n, int seed, double ux, double uy, double vx, double vy, GridWalk* w) {
  const int kNone = -(1 << 30);
  w->gi.assign(n, kNone); w->gj.assign(n, kNone);
  w->ux.assign(n, 0.0); w->uy.assign(n, 0.0); w->vx.assign(n, 0.0); w->vy.assign(n, 0.0);
  w->count = 0;
  std::queue<int> q;
  w->gi[seed] = 0; w->gj[seed] = 0; w->ux[seed] = ux; w->uy[seed] = uy; w->vx[seed] = vx; w->vy[seed] = vy;
  q.push(seed); w->count = 1;
  // occupied positions: a position is given once (the first dot to claim it keeps it)
  std::vector<std::pair<long long, int>> taken;
  auto key = [](int i, int j) { return ((long long)i << 32) ^ (unsigned)j; };
  auto is_taken = [&](int i, int j) { const long long k = key(i, j); for (auto& t : taken) if (t.first == k) return true; return false; };
  taken.push_back({key(0, 0), seed});
  while (!q.empty()) {
    const int k = q.front(); q.pop();
    const double cx = cen[2 * k], cy = cen[2 * k + 1];
    for (int dir = 0; dir < 4; ++dir)
      for (int hop = 1; hop <= 2; ++hop) {
        const double sx = (dir == 0 ? w->ux[k] : dir == 1 ? -w->ux[k] : dir == 2 ? w->vx[k] : -w->vx[k]);
        const double sy = (dir == 0 ? w->uy[k] : dir == 1 ? -w->uy[k] : dir == 2 ? w->vy[k] : -w->vy[k]);
        const int ni = w->gi[k] + (dir == 0 ? hop : dir == 1 ? -hop : 0), nj = w->gj[k] + (dir == 2 ? hop : dir == 3 ? -hop : 0);
        if (is_taken(ni, nj)) continue;
        const double px = cx + hop * sx, py = cy + hop * sy, step = std::sqrt(sx * sx + sy * sy);
        const double tol = (hop == 1 ? 0.30 : 0.38) * step;
        int best = -1; double bd = tol * tol;
        for (int m = 0; m < n; ++m) {
          if (w->gi[m] != kNone) continue;
          const double dx = cen[2 * m] - px, dy = cen[2 * m + 1] - py, d2 = dx * dx + dy * dy;
          if (d2 < bd) { bd = d2; best = m; }
        }
        if (best < 0) continue;
        // the step actually taken refreshes the lattice vector along the walk; the other one is inherited
        const double ax = (cen[2 * best] - cx) / hop, ay = (cen[2 * best + 1] - cy) / hop;
        w->gi[best] = ni; w->gj[best] = nj;
        w->ux[best] = w->ux[k]; w->uy[best] = w->uy[k]; w->vx[best] = w->vx[k]; w->vy[best] = w->vy[k];
        if (dir == 0) { w->ux[best] = ax; w->uy[best] = ay; } else if (dir == 1) { w->ux[best] = -ax; w->uy[best] = -ay; }
        else if (dir == 2) { w->vx[best] = ax; w->vy[best] = ay; } else { w->vx[best] = -ax; w->vy[best] = -ay; }
        taken.push_back({key(ni, nj), best});
        q.push(best); ++w->count;
      }
  }
}

// Lattice vectors at a dot from its neighbourhood: the nearest neighbour and the one opposite to it give u; the nearest neighbour
// well off that line, with its opposite, gives v.  false: the dot has no such pairs (border dot, isolated dot).
inline bool grid_seed_vectors(const double* cen, int n, int k, double* u, double* v) {
  std::vector<std::pair<double, int>> nb;
  for (int m = 0; m < n; ++m) if (m != k) { const double dx = cen[2 * m] - cen[2 * k], dy = cen[2 * m + 1] - cen[2 * k + 1]; nb.push_back({dx * dx + dy * dy, m}); }
  std::sort(nb.begin(), nb.end());
  if (nb.size() < 4) return false;
  const int lim = (int)std::min<size_t>(nb.size(), 10);
  auto vec = [&](int m, double* o) { o[0] = cen[2 * m] - cen[2 * k]; o[1] = cen[2 * m + 1] - cen[2 * k + 1]; };
  auto opposite = [&](const double* d, double* o) {     // the neighbour nearest to -d, within 35 % of |d|
    const double len2 = d[0] * d[0] + d[1] * d[1];
    int best = -1; double bd = 0.35 * 0.35 * len2;
    for (int a = 0; a < lim; ++a) { double w[2]; vec(nb[a].second, w); const double ex = w[0] + d[0], ey = w[1] + d[1], d2 = ex * ex + ey * ey; if (d2 < bd) { bd = d2; best = nb[a].second; } }
    if (best < 0) return false;
    vec(best, o); return true;
  };
  double d1[2], o1[2];
  vec(nb[0].second, d1);
  if (!opposite(d1, o1)) return false;
  u[0] = 0.5 * (d1[0] - o1[0]); u[1] = 0.5 * (d1[1] - o1[1]);
  const double ul = std::sqrt(u[0] * u[0] + u[1] * u[1]);
  for (int a = 1; a < lim; ++a) {
    double d2[2], o2[2];
    vec(nb[a].second, d2);
    const double l2 = std::sqrt(d2[0] * d2[0] + d2[1] * d2[1]);
    const double cs = (d2[0] * u[0] + d2[1] * u[1]) / (l2 * ul);
    if (std::fabs(cs) > 0.6 || l2 > 1.8 * ul) continue;           // along u, or a diagonal / second-ring neighbour
    if (!opposite(d2, o2)) continue;
    v[0] = 0.5 * (d2[0] - o2[0]); v[1] = 0.5 * (d2[1] - o2[1]);
    return true;
  }
  return false;
}

// cen: 2 per dot (pixels); conics: 9 per dot (image ellipse, any scale) or nullptr with areas given; pattern: rows x cols, 1 = large.
// dot_index (n): row * cols + col of the target dot, or -1.  Returns the number of associated dots (0: no unambiguous placement).
inline int grid_find_target(const double* cen, const double* conics, const double* areas_in, int n, const int* pattern, int rows, int cols,
                            int* dot_index) {
  for (int k = 0; k < n; ++k) dot_index[k] = -1;
  if (n < 8 || rows < 2 || cols < 2) return 0;
  // ---- 1. lattice walk from a few seeds near the middle of the detections: keep the largest component
  double mx = 0, my = 0;
  for (int k = 0; k < n; ++k) { mx += cen[2 * k]; my += cen[2 * k + 1]; }
  mx /= n; my /= n;
  std::vector<std::pair<double, int>> order;
  for (int k = 0; k < n; ++k) { const double dx = cen[2 * k] - mx, dy = cen[2 * k + 1] - my; order.push_back({dx * dx + dy * dy, k}); }
  std::sort(order.begin(), order.end());
  GridWalk best, cur;
  int tried = 0;
  for (size_t a = 0; a < order.size() && tried < 6; ++a) {
    double u[2], v[2];
    if (!grid_seed_vectors(cen, n, order[a].second, u, v)) continue;
    // handedness: with x to the right and y down in the image, (columns, rows) of a target seen from its front turn the same way as
    // (x, y) -- the four rotations below then cover every placement, no reflection
    if (u[0] * v[1] - u[1] * v[0] < 0.0) { v[0] = -v[0]; v[1] = -v[1]; }
    ++tried;
    grid_walk_from(cen, n, order[a].second, u[0], u[1], v[0], v[1], &cur);
    if (cur.count > best.count) best = cur;
    if (best.count >= (n * 9) / 10) break;
  }
  if (best.count < 8) return 0;
  const int kNone = -(1 << 30);
  // ---- 2. large / small: ellipse area over lattice-cell area, two clusters
  std::vector<double> ratio(n, 0.0);
  std::vector<int> idx;
  for (int k = 0; k < n; ++k) {
    if (best.gi[k] == kNone) continue;
    const double cell = std::fabs(best.ux[k] * best.vy[k] - best.uy[k] * best.vx[k]);
    const double ar = areas_in ? areas_in[k] : grid_conic_area(conics + 9 * (size_t)k);
    if (!(cell > 0.0) || !(ar > 0.0)) { best.gi[k] = kNone; continue; }
    ratio[k] = ar / cell; idx.push_back(k);
  }
  if (idx.size() < 8) return 0;
  double lo = 1e300, hi = 0.0;
  for (int k : idx) { lo = std::min(lo, ratio[k]); hi = std::max(hi, ratio[k]); }
  double c0 = lo, c1 = hi;
  for (int it = 0; it < 20; ++it) {
    double s0 = 0, s1 = 0; int n0 = 0, n1 = 0;
    for (int k : idx) { if (std::fabs(ratio[k] - c0) <= std::fabs(ratio[k] - c1)) { s0 += ratio[k]; ++n0; } else { s1 += ratio[k]; ++n1; } }
    if (n0) c0 = s0 / n0;
    if (n1) c1 = s1 / n1;
  }
  const bool two_sizes = c1 > 1.25 * c0;            // (all dots of one size: nothing to match against; every placement ties)
  std::vector<int> big(n, 0);
  for (int k : idx) big[k] = two_sizes && std::fabs(ratio[k] - c1) < std::fabs(ratio[k] - c0) ? 1 : 0;
  // ---- 3. placement: rotation (0, 90, 180, 270 degrees) + offset with the most agreeing dots
  int imin = 1 << 30, imax = -(1 << 30), jmin = 1 << 30, jmax = -(1 << 30);
  for (int k : idx) { imin = std::min(imin, best.gi[k]); imax = std::max(imax, best.gi[k]); jmin = std::min(jmin, best.gj[k]); jmax = std::max(jmax, best.gj[k]); }
  int best_score = -1, second = -1, best_rot = 0, best_or = 0, best_oc = 0;
  for (int rot = 0; rot < 4; ++rot) {
    // observed (i, j) -> (row, col) before the offset: rotations of the integer lattice
    auto map = [&](int i, int j, int* r, int* c) { if (rot == 0) { *r = j; *c = i; } else if (rot == 1) { *r = i; *c = -j; } else if (rot == 2) { *r = -j; *c = -i; } else { *r = -i; *c = j; } };
    int rmin = 1 << 30, rmax = -(1 << 30), cmin = 1 << 30, cmax = -(1 << 30);
    for (int a = 0; a < 4; ++a) { int r, c; map(a & 1 ? imax : imin, a & 2 ? jmax : jmin, &r, &c); rmin = std::min(rmin, r); rmax = std::max(rmax, r); cmin = std::min(cmin, c); cmax = std::max(cmax, c); }
    if (rmax - rmin >= rows || cmax - cmin >= cols) continue;
    for (int orow = -rmin; orow + rmax < rows; ++orow)
      for (int ocol = -cmin; ocol + cmax < cols; ++ocol) {
        int score = 0;
        for (int k : idx) { int r, c; map(best.gi[k], best.gj[k], &r, &c); score += pattern[(r + orow) * cols + (c + ocol)] == big[k] ? 1 : 0; }
        if (score > best_score) { second = best_score; best_score = score; best_rot = rot; best_or = orow; best_oc = ocol; }
        else if (score > second) second = score;
      }
  }
  if (best_score < 0) return 0;
  const int m = (int)idx.size();
  // unambiguous: explains >= 85 % of the dots and beats every other placement by a margin (a target seen in full with a symmetric
  // pattern, or too few dots, fails here -- as FindTarget's `false` makes the reference skip the frame, vicalib-task.cc:278-281)
  if (best_score * 100 < 85 * m || (second >= 0 && best_score - second < std::max(3, m / 12))) return 0;
  int count = 0;
  for (int k : idx) {
    int r, c;
    const int i = best.gi[k], j = best.gj[k];
    if (best_rot == 0) { r = j; c = i; } else if (best_rot == 1) { r = i; c = -j; } else if (best_rot == 2) { r = -j; c = -i; } else { r = -i; c = j; }
    dot_index[k] = (r + best_or) * cols + (c + best_oc); ++count;
  }
  return count;
}

}  // namespace vc
