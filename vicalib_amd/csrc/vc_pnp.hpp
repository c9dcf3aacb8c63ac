// vc_pnp.hpp -- host-side pose initialisation of one (frame, camera) view of the planar calibration target.
//
// Stands in for calibu::PosePnPRansac at its call site vicalib-task.cc:323-325 (the reference then stores
// T_wk = T_cw^-1 * T_ck, vicalib-task.cc:344-348); its output only seeds the optimiser.  The reference passes
// robust_3pt_its = 0, robust_3pt_tol = 0 there, i.e. the plain (non-robust) branch of Calibu's routine (not vendored).
//   pnp_planar         plane-to-image homography (normalised DLT, after undoing the current distortion estimate) -> pose ->
//                      Levenberg-Marquardt refinement of the 6 pose parameters on the full camera model: the its = 0 case.
//   pnp_planar_ransac  the robust branch (its > 0): minimal 4-point homographies from a deterministic counter-based
//                      sampler, consensus by reprojection error <= tol pixels on the full model, refit on the consensus
//                      set -- mismatched dots (a wrong grid association) no longer drag the seed pose.
// Front-end code: runs once per view on the CPU before the solve, never inside the loop.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "vc_math.hpp"

namespace vc {

// Symmetric eigen-decomposition (cyclic Jacobi), n <= 9.  A is destroyed; V columns = eigenvectors.
inline void jacobi_eig(double* A, double* V, int n) {
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
    if (off < 1e-300) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        if (std::fabs(A[p * n + q]) < 1e-300) continue;
        const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * A[p * n + q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {
          const double akp = A[k * n + p], akq = A[k * n + q];
          A[k * n + p] = c * akp - s * akq; A[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          const double apk = A[p * n + k], aqk = A[q * n + k];
          A[p * n + k] = c * apk - s * aqk; A[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = V[k * n + p], vkq = V[k * n + q];
          V[k * n + p] = c * vkp - s * vkq; V[k * n + q] = s * vkp + c * vkq;
        }
      }
  }
}

// Pixel -> ideal pinhole coordinates (x/z, y/z) by inverting the radial model numerically on its profile
// r_d = f(r_u) (Newton with a numeric slope on the model's own Project; f is monotone on the useful range).
inline bool pnp_unproject(int model, const double* K, double u, double v, double* xy) {
  const double xd = (u - K[2]) / K[0], yd = (v - K[3]) / K[1];
  if (model == kLinear) { xy[0] = xd; xy[1] = yd; return true; }
  const double rd = std::sqrt(xd * xd + yd * yd);
  if (rd < 1e-12) { xy[0] = xd; xy[1] = yd; return true; }
  double Kn[10];
  const int nk = model_nk(model);
  for (int i = 0; i < nk; ++i) Kn[i] = K[i];
  Kn[0] = 1.0; Kn[1] = 1.0; Kn[2] = 0.0; Kn[3] = 0.0;
  ModelPre pre;
  model_precompute(model, Kn, &pre);
  auto profile = [&](double ru) { const double pc[3] = {ru, 0.0, 1.0}; double pix[2]; project_any<false>(model, pc, Kn, pre, pix, nullptr, nullptr); return pix[0]; };
  double ru = rd;
  for (int it = 0; it < 40; ++it) {
    const double f = profile(ru) - rd;
    if (std::fabs(f) < 1e-13 * (1.0 + rd)) break;
    const double h = 1e-6 * (1.0 + ru);
    double slope = (profile(ru + h) - profile(ru - h)) / (2.0 * h);
    if (!(slope > 1e-6)) slope = 1e-6;
    double nxt = ru - f / slope;
    if (!(nxt > 0.0)) nxt = 0.5 * ru;
    if (nxt > 1e3) return false;             // beyond the model's field of view
    ru = nxt;
  }
  if (!std::isfinite(ru)) return false;
  xy[0] = xd * ru / rd; xy[1] = yd * ru / rd;
  return true;
}

// T_cw from >= 4 corners of the plane z = const.  pw: n x 3 world, uv: n x 2 pixels.  Returns the RMS reprojection
// error of the refined pose in *rms (pixels).  T = [qx qy qz qw tx ty tz].
// use: optional mask of the correspondences that take part (DLT and refinement); refine = false stops after the homography pose.
inline bool pnp_planar_masked(int model, const double* K, int n, const double* pw, const double* uv, const char* use, bool refine,
                              double* T_cw, double* rms) {
  if (n < 4) return false;
  const double z0 = pw[2];
  for (int i = 0; i < n; ++i) if (std::fabs(pw[3 * i + 2] - z0) > 1e-9) return false;     // the grid is planar (vicalib-task.cc:355-356)
  // ---- normalised DLT:  s [x y 1]^T = H [X Y 1]^T ---------------------------------------------------------
  std::vector<double> xy(2 * (size_t)n); std::vector<char> ok((size_t)n, 0);
  int m = 0;
  double mX = 0, mY = 0, mx = 0, my = 0;
  for (int i = 0; i < n; ++i) {
    ok[i] = ((!use || use[i]) && pnp_unproject(model, K, uv[2 * i], uv[2 * i + 1], &xy[2 * (size_t)i])) ? 1 : 0;
    if (ok[i]) { ++m; mX += pw[3 * i]; mY += pw[3 * i + 1]; mx += xy[2 * i]; my += xy[2 * i + 1]; }
  }
  if (m < 4) return false;
  mX /= m; mY /= m; mx /= m; my /= m;
  double dW = 0, dI = 0;
  for (int i = 0; i < n; ++i) if (ok[i]) {
    dW += std::hypot(pw[3 * i] - mX, pw[3 * i + 1] - mY); dI += std::hypot(xy[2 * i] - mx, xy[2 * i + 1] - my);
  }
  if (dW <= 0 || dI <= 0) return false;
  const double sW = std::sqrt(2.0) * m / dW, sI = std::sqrt(2.0) * m / dI;
  double AtA[81]; std::memset(AtA, 0, sizeof(AtA));
  for (int i = 0; i < n; ++i) if (ok[i]) {
    const double X = (pw[3 * i] - mX) * sW, Y = (pw[3 * i + 1] - mY) * sW, x = (xy[2 * i] - mx) * sI, y = (xy[2 * i + 1] - my) * sI;
    const double r1[9] = {-X, -Y, -1, 0, 0, 0, x * X, x * Y, x}, r2[9] = {0, 0, 0, -X, -Y, -1, y * X, y * Y, y};
    for (int a = 0; a < 9; ++a) for (int b = 0; b < 9; ++b) AtA[a * 9 + b] += r1[a] * r1[b] + r2[a] * r2[b];
  }
  double V[81];
  jacobi_eig(AtA, V, 9);
  int best = 0;
  for (int k = 1; k < 9; ++k) if (AtA[k * 9 + k] < AtA[best * 9 + best]) best = k;
  double Hn[9];
  for (int k = 0; k < 9; ++k) Hn[k] = V[k * 9 + best];
  // de-normalise: H = Ti^-1 Hn Tw with Tw = [sW 0 -sW mX; 0 sW -sW mY; 0 0 1], Ti likewise
  double H[9];
  {
    const double Tw[9] = {sW, 0, -sW * mX, 0, sW, -sW * mY, 0, 0, 1}, TiInv[9] = {1 / sI, 0, mx, 0, 1 / sI, my, 0, 0, 1};
    double tmp[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { tmp[3 * i + j] = 0; for (int k = 0; k < 3; ++k) tmp[3 * i + j] += Hn[3 * i + k] * Tw[3 * k + j]; }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { H[3 * i + j] = 0; for (int k = 0; k < 3; ++k) H[3 * i + j] += TiInv[3 * i + k] * tmp[3 * k + j]; }
  }
  // ---- pose from H = lambda [r1 r2 t'] (plane z = z0: t = t' - r3 z0) --------------------------------------
  double r1[3] = {H[0], H[3], H[6]}, r2[3] = {H[1], H[4], H[7]}, t[3] = {H[2], H[5], H[8]};
  const double n1 = std::sqrt(r1[0] * r1[0] + r1[1] * r1[1] + r1[2] * r1[2]), n2 = std::sqrt(r2[0] * r2[0] + r2[1] * r2[1] + r2[2] * r2[2]);
  if (!(n1 > 0) || !(n2 > 0)) return false;
  double lam = 2.0 / (n1 + n2);
  if (t[2] * lam < 0) lam = -lam;                 // the target is in front of the camera
  for (int i = 0; i < 3; ++i) { r1[i] *= lam; r2[i] *= lam; t[i] *= lam; }
  double R[9];
  const double r3[3] = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};
  for (int i = 0; i < 3; ++i) { R[3 * i] = r1[i]; R[3 * i + 1] = r2[i]; R[3 * i + 2] = r3[i]; }
  for (int it = 0; it < 30; ++it) {               // nearest rotation: Newton polar iteration R <- (R + R^-T)/2
    const double c00 = R[4] * R[8] - R[5] * R[7], c01 = R[5] * R[6] - R[3] * R[8], c02 = R[3] * R[7] - R[4] * R[6];
    const double det = R[0] * c00 + R[1] * c01 + R[2] * c02;
    if (std::fabs(det) < 1e-12) return false;
    const double cof[9] = {c00, c01, c02,
                           R[2] * R[7] - R[1] * R[8], R[0] * R[8] - R[2] * R[6], R[1] * R[6] - R[0] * R[7],
                           R[1] * R[5] - R[2] * R[4], R[2] * R[3] - R[0] * R[5], R[0] * R[4] - R[1] * R[3]};
    double diff = 0;
    for (int k = 0; k < 9; ++k) { const double nr = 0.5 * (R[k] + cof[k] / det); diff += std::fabs(nr - R[k]); R[k] = nr; }
    if (diff < 1e-15) break;
  }
  for (int i = 0; i < 3; ++i) t[i] -= R[3 * i + 2] * z0;
  // rotation matrix -> quaternion (x y z w)
  double q[4];
  {
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0) { const double s = std::sqrt(tr + 1.0) * 2; q[3] = 0.25 * s; q[0] = (R[7] - R[5]) / s; q[1] = (R[2] - R[6]) / s; q[2] = (R[3] - R[1]) / s; }
    else if (R[0] > R[4] && R[0] > R[8]) { const double s = std::sqrt(1.0 + R[0] - R[4] - R[8]) * 2; q[3] = (R[7] - R[5]) / s; q[0] = 0.25 * s; q[1] = (R[1] + R[3]) / s; q[2] = (R[2] + R[6]) / s; }
    else if (R[4] > R[8]) { const double s = std::sqrt(1.0 + R[4] - R[0] - R[8]) * 2; q[3] = (R[2] - R[6]) / s; q[0] = (R[1] + R[3]) / s; q[1] = 0.25 * s; q[2] = (R[5] + R[7]) / s; }
    else { const double s = std::sqrt(1.0 + R[8] - R[0] - R[4]) * 2; q[3] = (R[3] - R[1]) / s; q[0] = (R[2] + R[6]) / s; q[1] = (R[5] + R[7]) / s; q[2] = 0.25 * s; }
    const double nq = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) q[i] /= nq;
  }
  double T[7] = {q[0], q[1], q[2], q[3], t[0], t[1], t[2]};
  if (!refine) { std::memcpy(T_cw, T, sizeof(T)); if (rms) *rms = 0.0; return std::isfinite(t[0] + t[1] + t[2]); }
  // ---- LM refinement of T_cw <- T_cw exp(delta) on the full model ---------------------------------------------
  ModelPre pre;
  model_precompute(model, K, &pre);
  int n_cost = 0;                                  // points the last cost_of call summed over (all used points, not only the m unprojectable ones)
  auto cost_of = [&](const double* Tc, double* Hm, double* g) {
    double Rc[9]; quat_to_R(Tc, Rc);
    double cost = 0.0;
    n_cost = 0;
    if (Hm) { std::memset(Hm, 0, 36 * sizeof(double)); std::memset(g, 0, 6 * sizeof(double)); }
    for (int i = 0; i < n; ++i) {
      if (use && !use[i]) continue;
      ++n_cost;
      const double* p = pw + 3 * i;
      double pc[3];
      for (int a = 0; a < 3; ++a) pc[a] = Rc[3 * a] * p[0] + Rc[3 * a + 1] * p[1] + Rc[3 * a + 2] * p[2] + Tc[4 + a];
      if (pc[2] <= 1e-9) { cost += 1e6; continue; }
      double pix[2], A[6], B[20];
      project_any<true>(model, pc, K, pre, pix, A, B);
      const double r[2] = {pix[0] - uv[2 * i], pix[1] - uv[2 * i + 1]};
      if (!std::isfinite(r[0]) || !std::isfinite(r[1])) { cost += 1e6; continue; }
      cost += r[0] * r[0] + r[1] * r[1];
      if (Hm) {
        // d pc / d upsilon = R,  d pc / d omega = -R [p]x
        double J[12];
        for (int a = 0; a < 2; ++a) {
          double AR[3];
          for (int b = 0; b < 3; ++b) AR[b] = A[3 * a] * Rc[b] + A[3 * a + 1] * Rc[3 + b] + A[3 * a + 2] * Rc[6 + b];
          J[6 * a] = AR[0]; J[6 * a + 1] = AR[1]; J[6 * a + 2] = AR[2];
          J[6 * a + 3] = -(AR[1] * p[2] - AR[2] * p[1]);
          J[6 * a + 4] = -(AR[2] * p[0] - AR[0] * p[2]);
          J[6 * a + 5] = -(AR[0] * p[1] - AR[1] * p[0]);
        }
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 6; ++b) { g[b] += J[6 * a + b] * r[a]; for (int c = 0; c < 6; ++c) Hm[6 * b + c] += J[6 * a + b] * J[6 * a + c]; }
      }
    }
    return cost;
  };
  double lambda = 1e-3, Hm[36], g[6];
  double cost = cost_of(T, Hm, g);
  for (int it = 0; it < 50; ++it) {
    double M[36], d[6];
    for (int k = 0; k < 36; ++k) M[k] = Hm[k];
    for (int k = 0; k < 6; ++k) { M[7 * k] += lambda * (Hm[7 * k] + 1e-12); d[k] = -g[k]; }
    if (!chol_small<6>(M)) { lambda *= 10; if (lambda > 1e12) break; continue; }
    fwd_solve<6>(M, d); bwd_solve<6>(M, d);
    double Tn[7];
    se3_plus(T, d, Tn);
    const double cn = cost_of(Tn, nullptr, nullptr);
    if (cn < cost) {
      const double rel = (cost - cn) / (cost + 1e-300);
      std::memcpy(T, Tn, sizeof(T)); cost = cost_of(T, Hm, g); lambda = std::fmax(lambda * 0.3, 1e-12);
      if (rel < 1e-12) break;
    } else { lambda *= 10; if (lambda > 1e12) break; }
  }
  std::memcpy(T_cw, T, sizeof(T));
  cost = cost_of(T, nullptr, nullptr);             // (sets n_cost for the accepted pose)
  if (rms) *rms = std::sqrt(cost / (n_cost > 0 ? n_cost : 1));
  return std::isfinite(cost);
}
inline bool pnp_planar(int model, const double* K, int n, const double* pw, const double* uv, double* T_cw, double* rms) {
  return pnp_planar_masked(model, K, n, pw, uv, nullptr, true, T_cw, rms);
}

// reprojection error (pixels) of every correspondence under T_cw; points behind the camera get a huge error
inline void pnp_errors(int model, const double* K, int n, const double* pw, const double* uv, const double* T, double* err) {
  ModelPre pre;
  model_precompute(model, K, &pre);
  double Rc[9]; quat_to_R(T, Rc);
  for (int i = 0; i < n; ++i) {
    const double* p = pw + 3 * i;
    double pc[3], pix[2];
    for (int a = 0; a < 3; ++a) pc[a] = Rc[3 * a] * p[0] + Rc[3 * a + 1] * p[1] + Rc[3 * a + 2] * p[2] + T[4 + a];
    if (pc[2] <= 1e-9) { err[i] = 1e30; continue; }
    project_any<false>(model, pc, K, pre, pix, nullptr, nullptr);
    const double e = std::hypot(pix[0] - uv[2 * i], pix[1] - uv[2 * i + 1]);
    err[i] = std::isfinite(e) ? e : 1e30;
  }
}
// Robust pose of one view: `its` minimal samples of 4 correspondences, consensus at `tol` pixels, refit on the consensus set.
// rms: RMS reprojection error over the inliers; n_inliers / inlier (n flags) optional.
inline bool pnp_planar_ransac(int model, const double* K, int n, const double* pw, const double* uv, int its, double tol,
                              double* T_cw, double* rms, int* n_inliers, char* inlier) {
  if (its <= 0 || n < 5) {
    const bool ok = pnp_planar(model, K, n, pw, uv, T_cw, rms);
    if (n_inliers) *n_inliers = ok ? n : 0;
    if (inlier) std::memset(inlier, ok ? 1 : 0, (size_t)std::max(n, 0));
    return ok;
  }
  std::vector<char> mask((size_t)n), best((size_t)n, 0);
  std::vector<double> err((size_t)n);
  int best_cnt = 0;
  double best_sum = 0.0;
  uint64_t state = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;          // deterministic: the same view gives the same pose
  auto next = [&]() { state += 0x9E3779B97F4A7C15ull; uint64_t z = state; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
  for (int it = 0; it < its; ++it) {
    int pick[4];
    for (int k = 0; k < 4; ++k) {
      bool again = true;
      while (again) { pick[k] = (int)(next() % (uint64_t)n); again = false; for (int j = 0; j < k; ++j) if (pick[j] == pick[k]) again = true; }
    }
    std::fill(mask.begin(), mask.end(), 0);
    for (int k = 0; k < 4; ++k) mask[pick[k]] = 1;
    double T[7];
    if (!pnp_planar_masked(model, K, n, pw, uv, mask.data(), false, T, nullptr)) continue;
    pnp_errors(model, K, n, pw, uv, T, err.data());
    int cnt = 0; double sum = 0.0;
    for (int i = 0; i < n; ++i) if (err[i] <= tol) { ++cnt; sum += err[i]; }
    if (cnt > best_cnt || (cnt == best_cnt && cnt > 0 && sum < best_sum)) {
      best_cnt = cnt; best_sum = sum;
      for (int i = 0; i < n; ++i) best[i] = err[i] <= tol;
    }
  }
  if (best_cnt < 4) return false;
  // refit on the consensus set, then let the refined pose re-vote once (a minimal sample's pose is rough)
  double T[7], r = 0.0;
  if (!pnp_planar_masked(model, K, n, pw, uv, best.data(), true, T, &r)) return false;
  pnp_errors(model, K, n, pw, uv, T, err.data());
  int cnt = 0;
  for (int i = 0; i < n; ++i) { mask[i] = err[i] <= tol; cnt += mask[i]; }
  if (cnt >= 4 && cnt != best_cnt) { double T2[7], r2; if (pnp_planar_masked(model, K, n, pw, uv, mask.data(), true, T2, &r2)) { std::memcpy(T, T2, sizeof(T)); r = r2; best = mask; best_cnt = cnt; } }
  std::memcpy(T_cw, T, sizeof(T));
  if (rms) *rms = r;
  if (n_inliers) *n_inliers = best_cnt;
  if (inlier) std::memcpy(inlier, best.data(), (size_t)n);
  return true;
}

}  // namespace vc
