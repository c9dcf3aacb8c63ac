// ORACLE -- TEST INFRASTRUCTURE ONLY (see vco_math.h header). PARITY UNPINNED.
//
// vco_solver.h: CPU restatement of ViCalibrator (vicalibrator.h:119-1086):
// problem container (AddCamera :332, AddFrame :355, AddImuMeasurements :370,
// AddObservation :385), SetupProblem's constancy rules and residual
// multiplicities (:548-679), the iteration callback (:690-721), per-camera
// RMSE (:958-971), RemoveOutliers (:859-916) and the SolveThread stage machine
// (:919-1040), on top of a restatement of the Ceres pieces it configures
// (:136-152): AutoDiffCostFunction (forward-mode duals), LossFunction
// correction, Jacobi scaling and the trust-region loop.  Ceres is un-vendored
// and unpinned (CMakeLists.txt:44-56); its algorithm is restated from the
// published trust_region_minimizer / levenberg_marquardt_strategy.
// BASELINE.json's north_star fixes the strategy to Levenberg-Marquardt.
#pragma once
#include <vector>
#include <cstring>
#include <cstdio>
#include <cmath>
#include <algorithm>
#include "vco_math.h"
#include "vco_models.h"
#include "vco_imu.h"
#include "vco_weights.h"

namespace vco {

struct Camera { int model; int nk; double K[10]; double T_ck[7]; int width, height; };
struct Frame { double T_wk[7]; double v[3]; double time; };
struct Obs { int frame, cam; double p_w[3]; double z[2]; int mult_delta; /* 0, or -1 after outlier removal */ };

struct IterRecord {
  int iteration; double cost; double cost_change; double gradient_max_norm; double gradient_norm;
  double step_norm; double relative_decrease; double radius; int accepted; int stage;
};

enum Termination { kConvergence = 0, kNoConvergence = 1, kUserSuccess = 2, kFailure = 3 };

struct Options {
  int max_iters = 200;               // FLAGS_max_iters, vicalib-engine.cc:94
  double function_tolerance = 1e-6;  // vicalibrator.h:149, vicalib-task.cc:233
  double gradient_tolerance = 1e-10; // Ceres default
  double parameter_tolerance = 1e-8; // Ceres default
  double initial_radius = 1e4, max_radius = 1e16, min_radius = 1e-32;
  double min_relative_decrease = 1e-3;
  double min_diag = 1e-6, max_diag = 1e32;
  bool jacobi_scaling = true;
  bool calibrate_imu = true;         // FLAGS_calibrate_imu, vicalib-engine.cc:35
  bool remove_outliers = false;      // vicalib-engine.cc:100
  double outlier_threshold = 2.0;    // vicalib-engine.cc:102
  int num_threads = 1;
  bool dense_check = false;          // solve with a dense Cholesky instead of the block elimination
  bool closed_form = false;          // bench.py's "best CPU" leg only (never a parity test): reprojection Jacobians in closed form
                                     // instead of forward duals -- the work shape of a hand-optimised CPU path, not of Ceres' AutoDiff
};

class Calibrator;
void reproj_block_closed_form(const Calibrator& c, const Obs& o, double* r, double* Jf, double* Jr, double* Jt, double* Jk);   // vco_fast.h

class Calibrator {
 public:
  std::vector<Camera> cams;
  std::vector<Frame> frames;
  std::vector<Obs> obs;               // sorted by (frame, cam) lazily before a solve
  ImuBuffer imu;
  double g_dir[2] = {0, 0}, time_offset = 0;   // ImuCalibrationT g_, time_offset_ (types.h:141,150)
  double biases[6] = {0, 0, 0, 0, 0, 0}, scale[6] = {1, 1, 1, 1, 1, 1};
  double gyro_sigma = 5.3088444e-5, accel_sigma = 0.001883649;  // types.h:34-35
  Options opt;
  // Clear() defaults, vicalibrator.h:232-249
  bool fix_intrinsics = false, is_bias_active = false, is_scale_active = false, is_inertial_active = false,
       is_visual_active = true, rotation_only = true, optimize_time_offset = true, is_finished = false,
       gravity_initialized = false, outliers_removed = false;
  int vis_mult = 0, imu_mult = 0;     // how many times SetupProblem re-added the blocks (:641-656)
  bool latest_copy_full = true;       // projection_residuals_ = a complete fresh copy (:551-552, :645) vs inliers only (:914)
  unsigned num_imu_residuals = 0;     // :607
  std::vector<double> imu_w_sqrt;     // (N-1) x 81, weight_sqrt_ of each IMU cost (:616, :796)
  std::vector<double> cam_rmse;
  double mse = 0;
  unsigned num_iterations = 0;
  std::vector<IterRecord> trace;
  int stage = 0;

  // ---- layout of the reduced (free) parameters --------------------------------
  struct Layout {
    int df = 6, D = 0;
    std::vector<int> rot, trans, kk;   // per camera offsets into the shared vector, -1 = constant
    int g = -1, b = -1, sf = -1, toff = -1;
    bool imu = false;
  } L;

  void build_layout() {
    L = Layout();
    L.imu = opt.calibrate_imu && is_inertial_active;
    L.df = (L.imu && !rotation_only) ? 9 : 6;
    const int C = (int)cams.size();
    L.rot.assign(C, -1); L.trans.assign(C, -1); L.kk.assign(C, -1);
    int o = 0;
    for (int c = 0; c < C; ++c) {
      bool rf = true, tf = true;
      if (c == 0) {            // vicalibrator.h:572-587
        if (!is_inertial_active) { rf = false; tf = false; }
        else { rf = true; tf = !rotation_only; }
      }
      if (rf) { L.rot[c] = o; o += 3; }
      if (tf) { L.trans[c] = o; o += 3; }
      if (!fix_intrinsics) { L.kk[c] = o; o += cams[c].nk; }
    }
    if (L.imu) {
      if (!rotation_only) { L.g = o; o += 2; }        // :657-660, :986
      if (is_bias_active) { L.b = o; o += 6; }         // :663-666, :990
      if (is_scale_active) { L.sf = o; o += 6; }       // :668-671, :994
      if (optimize_time_offset) { L.toff = o; o += 1; }// :673-676
    }
    L.D = o;
  }

  // ---- normal equations in block-arrow form -----------------------------------
  int N() const { return (int)frames.size(); }
  std::vector<double> A, Cc, W, Hss, gf, gs;   // A: N x 81, Cc: N x 81 (f,f+1), W: N x 9 x D
  void zero_normal() {
    const int n = N(), D = L.D;
    A.assign((size_t)n * 81, 0.0); Cc.assign((size_t)n * 81, 0.0); W.assign((size_t)n * 9 * D, 0.0);
    Hss.assign((size_t)D * D, 0.0); gf.assign((size_t)n * 9, 0.0); gs.assign(D, 0.0);
  }

  void sort_obs() {
    std::stable_sort(obs.begin(), obs.end(), [](const Obs& a, const Obs& b) {
      return a.frame != b.frame ? a.frame < b.frame : a.cam < b.cam; });
  }

  // One reprojection block: residual (2) and local Jacobian 2 x (6 + 3 + 3 + nk)
  // via forward duals over the 7+4+3+nk global parameters times the local
  // parameterisation Jacobians -- what AutoDiffCostFunction + LocalParamSe3/So3
  // hand to Ceres (vicalibrator.h:413-461, local-param-se3.h).
  template <int NK> void reproj_block(const Obs& o, double* r, double* Jf /*2x6*/, double* Jr /*2x3*/,
                                      double* Jt /*2x3*/, double* Jk /*2xNK*/) const {
    typedef Dual<14 + NK> D;
    const Camera& c = cams[o.cam];
    const Frame& f = frames[o.frame];
    D t_wk[7], r_ck[4], p_ck[3], K[NK], res[2];
    for (int i = 0; i < 7; ++i) t_wk[i] = D::Var(f.T_wk[i], i);
    for (int i = 0; i < 4; ++i) r_ck[i] = D::Var(c.T_ck[i], 7 + i);
    for (int i = 0; i < 3; ++i) p_ck[i] = D::Var(c.T_ck[4 + i], 11 + i);
    for (int i = 0; i < NK; ++i) K[i] = D::Var(c.K[i], 14 + i);
    reproj_residual<D>(c.model, t_wk, r_ck, p_ck, K, o.p_w, o.z, res);
    double P7[42], P4[12];
    local_jac_se3(f.T_wk, P7);
    local_jac_so3(c.T_ck, P4);
    for (int row = 0; row < 2; ++row) {
      r[row] = res[row].a;
      for (int j = 0; j < 6; ++j) { double s = 0; for (int i = 0; i < 7; ++i) s += res[row].v[i] * P7[i * 6 + j]; Jf[row * 6 + j] = s; }
      for (int j = 0; j < 3; ++j) { double s = 0; for (int i = 0; i < 4; ++i) s += res[row].v[7 + i] * P4[i * 3 + j]; Jr[row * 3 + j] = s; }
      for (int j = 0; j < 3; ++j) Jt[row * 3 + j] = res[row].v[11 + j];
      for (int j = 0; j < NK; ++j) Jk[row * NK + j] = res[row].v[14 + j];
    }
  }
  void reproj_block_any(const Obs& o, double* r, double* Jf, double* Jr, double* Jt, double* Jk) const {
#ifdef VCO_WITH_FAST
    if (opt.closed_form) { reproj_block_closed_form(*this, o, r, Jf, Jr, Jt, Jk); return; }
#endif
    switch (cams[o.cam].nk) {
      case 4: reproj_block<4>(o, r, Jf, Jr, Jt, Jk); break;
      case 5: reproj_block<5>(o, r, Jf, Jr, Jt, Jk); break;
      case 6: reproj_block<6>(o, r, Jf, Jr, Jt, Jk); break;
      case 7: reproj_block<7>(o, r, Jf, Jr, Jt, Jk); break;
      case 10: reproj_block<10>(o, r, Jf, Jr, Jt, Jk); break;
      default: reproj_block<8>(o, r, Jf, Jr, Jt, Jk); break;
    }
  }
  void reproj_value(const Obs& o, double* r) const {
    const Camera& c = cams[o.cam];
    reproj_residual<double>(c.model, frames[o.frame].T_wk, c.T_ck, c.T_ck + 4, c.K, o.p_w, o.z, r);
  }

  // IMU block j (frames j-1 -> j): residual 9 and local Jacobian 9 x (6,6,3,3,2,6,6,1)
  // AutoDiff sizes <9, 7,7,3,3,2,6,6,1> (vicalibrator.h:620-621).
  struct ImuJac { double r[9]; double J2[54], J1[54], Jv2[27], Jv1[27], Jg[18], Jb[54], Js[54], Jt[9]; };
  void imu_block(int j, ImuJac* out) const {
    typedef Dual<35> D;
    const Frame& f2 = frames[j]; const Frame& f1 = frames[j - 1];
    D tx2[7], tx1[7], v2[3], v1[3], g[2], b[6], sf[6], toff, res[9];
    for (int i = 0; i < 7; ++i) tx2[i] = D::Var(f2.T_wk[i], i);
    for (int i = 0; i < 7; ++i) tx1[i] = D::Var(f1.T_wk[i], 7 + i);
    for (int i = 0; i < 3; ++i) v2[i] = D::Var(f2.v[i], 14 + i);
    for (int i = 0; i < 3; ++i) v1[i] = D::Var(f1.v[i], 17 + i);
    for (int i = 0; i < 2; ++i) g[i] = D::Var(g_dir[i], 20 + i);
    for (int i = 0; i < 6; ++i) b[i] = D::Var(biases[i], 22 + i);
    for (int i = 0; i < 6; ++i) sf[i] = D::Var(scale[i], 28 + i);
    toff = D::Var(time_offset, 34);
    imu_residual<D>(imu, f1.time, f2.time, &imu_w_sqrt[(size_t)(j - 1) * 81], rotation_only, tx2, tx1, v2, v1, g, b, sf,
                    &toff, res);
    double P2[42], P1[42];
    local_jac_se3(f2.T_wk, P2);
    local_jac_se3(f1.T_wk, P1);
    for (int row = 0; row < 9; ++row) {
      out->r[row] = res[row].a;
      for (int c = 0; c < 6; ++c) {
        double s2 = 0, s1 = 0;
        for (int i = 0; i < 7; ++i) { s2 += res[row].v[i] * P2[i * 6 + c]; s1 += res[row].v[7 + i] * P1[i * 6 + c]; }
        out->J2[row * 6 + c] = s2; out->J1[row * 6 + c] = s1;
      }
      for (int c = 0; c < 3; ++c) { out->Jv2[row * 3 + c] = res[row].v[14 + c]; out->Jv1[row * 3 + c] = res[row].v[17 + c]; }
      for (int c = 0; c < 2; ++c) out->Jg[row * 2 + c] = res[row].v[20 + c];
      for (int c = 0; c < 6; ++c) { out->Jb[row * 6 + c] = res[row].v[22 + c]; out->Js[row * 6 + c] = res[row].v[28 + c]; }
      out->Jt[row] = res[row].v[34];
    }
  }
  void imu_value(int j, double* r) const {
    const Frame& f2 = frames[j]; const Frame& f1 = frames[j - 1];
    imu_residual<double>(imu, f1.time, f2.time, &imu_w_sqrt[(size_t)(j - 1) * 81], rotation_only, f2.T_wk, f1.T_wk, f2.v,
                         f1.v, g_dir, biases, scale, &time_offset, r);
  }

  double obs_mult(const Obs& o) const { return (double)(vis_mult + o.mult_delta); }

  // Cost only: 1/2 sum_blocks rho(|r|^2), each block counted with its multiplicity.
  double evaluate_cost() const {
    double cost = 0;
    const long n = (long)obs.size();
    if (is_visual_active) {
#pragma omp parallel for reduction(+ : cost) schedule(static) num_threads(opt.num_threads)
      for (long i = 0; i < n; ++i) {
        const double m = obs_mult(obs[i]);
        if (m <= 0) continue;
        double r[2], rho[3];
        reproj_value(obs[i], r);
        loss_soft_l1(0.5, r[0] * r[0] + r[1] * r[1], rho);
        cost += 0.5 * m * rho[0];
      }
    }
    if (L.imu) {
      double ci = 0;
#pragma omp parallel for reduction(+ : ci) schedule(static) num_threads(opt.num_threads)
      for (int j = 1; j < N(); ++j) {
        double r[9], rho[3], s = 0;
        imu_value(j, r);
        for (int i = 0; i < 9; ++i) s += r[i] * r[i];
        loss_cauchy(100.0, s, rho);
        ci += 0.5 * imu_mult * rho[0];
      }
      cost += ci;
    }
    return cost;
  }

  // Accumulate J^T J / J^T r of one block given its column groups.
  struct Group { int frame; int off; int n; const double* J; };  // frame = -1 -> shared vector
  void accumulate(int nres, const double* r, const Group* g, int ng, double w, std::vector<double>& Hs,
                  std::vector<double>& gsv) {
    const int D = L.D;
    for (int a = 0; a < ng; ++a) {
      if (g[a].off < 0) continue;
      for (int i = 0; i < g[a].n; ++i) {
        double gr = 0;
        for (int k = 0; k < nres; ++k) gr += g[a].J[k * g[a].n + i] * r[k];
        gr *= w;
        if (g[a].frame < 0) gsv[g[a].off + i] += gr; else gf[(size_t)g[a].frame * 9 + g[a].off + i] += gr;
      }
      for (int b = 0; b < ng; ++b) {
        if (g[b].off < 0) continue;
        const int fa = g[a].frame, fb = g[b].frame;
        // store: shared-shared (full), frame-shared (W), frame-frame same (A full), frame f -> f+1 (Cc)
        if (fa < 0 && fb >= 0) continue;
        if (fa >= 0 && fb >= 0 && !(fb == fa || fb == fa + 1)) continue;
        for (int i = 0; i < g[a].n; ++i)
          for (int j = 0; j < g[b].n; ++j) {
            double s = 0;
            for (int k = 0; k < nres; ++k) s += g[a].J[k * g[a].n + i] * g[b].J[k * g[b].n + j];
            s *= w;
            const int ia = g[a].off + i, ib = g[b].off + j;
            if (fa < 0) Hs[(size_t)ia * D + ib] += s;
            else if (fb < 0) W[((size_t)fa * 9 + ia) * D + ib] += s;
            else if (fb == fa) A[(size_t)fa * 81 + ia * 9 + ib] += s;
            else Cc[(size_t)fa * 81 + ia * 9 + ib] += s;
          }
      }
    }
  }

  // Full linearisation at the current state -> normal equations; returns cost.
  double linearize() {
    zero_normal();
    double cost = 0;
    const int D = L.D;
    if (is_visual_active) {
      // frames are independent in the visual terms: parallel over frame ranges
      std::vector<size_t> fstart(N() + 1, obs.size());
      {
        size_t i = 0;
        for (int f = 0; f <= N(); ++f) { while (i < obs.size() && obs[i].frame < f) ++i; fstart[f] = i; }
      }
#pragma omp parallel num_threads(opt.num_threads)
      {
        std::vector<double> Hs((size_t)D * D, 0.0), gsl(D, 0.0);
        double cl = 0;
#pragma omp for schedule(dynamic, 8)
        for (int f = 0; f < N(); ++f) {
          for (size_t i = fstart[f]; i < fstart[f + 1]; ++i) {
            const Obs& o = obs[i];
            const double m = obs_mult(o);
            if (m <= 0) continue;
            double r[2], Jf[12], Jr[6], Jt[6], Jk[20], rho[3];
            reproj_block_any(o, r, Jf, Jr, Jt, Jk);
            loss_soft_l1(0.5, r[0] * r[0] + r[1] * r[1], rho);
            cl += 0.5 * m * rho[0];
            // Corrector with rho'' <= 0: residual and Jacobian both scaled by sqrt(rho')
            const double w = m * rho[1];
            Group g[4] = {{o.frame, 0, 6, Jf}, {-1, L.rot[o.cam], 3, Jr}, {-1, L.trans[o.cam], 3, Jt},
                          {-1, L.kk[o.cam], cams[o.cam].nk, Jk}};
            accumulate(2, r, g, 4, w, Hs, gsl);
          }
        }
#pragma omp critical
        {
          for (size_t i = 0; i < Hs.size(); ++i) Hss[i] += Hs[i];
          for (int i = 0; i < D; ++i) gs[i] += gsl[i];
          cost += cl;
        }
      }
    }
    if (L.imu) {
      // the dual-number evaluation of the blocks is independent per block (threads); the accumulation into the shared
      // normal equations stays sequential, in block order
      std::vector<ImuJac> blocks((size_t)std::max(N(), 1));
#pragma omp parallel for schedule(dynamic, 4) num_threads(opt.num_threads)
      for (int j = 1; j < N(); ++j) imu_block(j, &blocks[j]);
      for (int j = 1; j < N(); ++j) {
        const ImuJac& B = blocks[j];
        double s = 0, rho[3];
        for (int i = 0; i < 9; ++i) s += B.r[i] * B.r[i];
        loss_cauchy(100.0, s, rho);
        cost += 0.5 * imu_mult * rho[0];
        const double w = imu_mult * rho[1];
        const int vo = (L.df == 9) ? 6 : -1;
        // order frame j-1 before frame j so that the (f, f+1) coupling lands in Cc[f]
        Group g[8] = {{j - 1, 0, 6, B.J1}, {j - 1, vo, 3, B.Jv1}, {j, 0, 6, B.J2}, {j, vo, 3, B.Jv2},
                      {-1, L.g, 2, B.Jg}, {-1, L.b, 6, B.Jb}, {-1, L.sf, 6, B.Js}, {-1, L.toff, 1, B.Jt}};
        accumulate(9, B.r, g, 8, w, Hss, gs);
      }
    }
    return cost;
  }

  // ---- linear algebra -----------------------------------------------------------
  static bool chol(double* M, int n, int ld) {   // in place lower Cholesky
    for (int j = 0; j < n; ++j) {
      double d = M[j * ld + j];
      for (int k = 0; k < j; ++k) d -= M[j * ld + k] * M[j * ld + k];
      if (!(d > 0)) return false;
      d = std::sqrt(d); M[j * ld + j] = d;
      for (int i = j + 1; i < n; ++i) {
        double s = M[i * ld + j];
        for (int k = 0; k < j; ++k) s -= M[i * ld + k] * M[j * ld + k];
        M[i * ld + j] = s / d;
      }
    }
    return true;
  }
  static void chol_solve(const double* Lm, int n, int ld, double* x, int nrhs, int ldx) {  // x: n x nrhs row-major
    for (int c = 0; c < nrhs; ++c) {
      for (int i = 0; i < n; ++i) { double s = x[i * ldx + c]; for (int k = 0; k < i; ++k) s -= Lm[i * ld + k] * x[k * ldx + c]; x[i * ldx + c] = s / Lm[i * ld + i]; }
      for (int i = n - 1; i >= 0; --i) { double s = x[i * ldx + c]; for (int k = i + 1; k < n; ++k) s -= Lm[k * ld + i] * x[k * ldx + c]; x[i * ldx + c] = s / Lm[i * ld + i]; }
    }
  }

  // Solve (H + Lambda) delta = -g with Lambda = diag(lam).  lam is indexed
  // [frame f: f*9 + i] then [shared: N*9 + s].  Block forward elimination along
  // the frame chain (exact; equals Ceres' sparse normal Cholesky up to roundoff).
  bool solve_blocks(const std::vector<double>& lam, std::vector<double>& dfv, std::vector<double>& dsv) {
    const int n = N(), D = L.D, df = L.df;
    std::vector<double> Ah((size_t)n * 81), Wh((size_t)n * 9 * D), gh((size_t)n * 9), S(Hss), gr(gs);
    std::vector<double> Cs(Cc);  // couplings
    for (int i = 0; i < D; ++i) S[(size_t)i * D + i] += lam[(size_t)n * 9 + i];
    for (int i = 0; i < D; ++i) for (int j = 0; j < i; ++j) S[(size_t)i * D + j] = S[(size_t)j * D + i];  // mirror upper
    // note: Hss accumulates both triangles already (accumulate adds (a,b) and (b,a)); mirror is harmless
    std::vector<double> X(9 * (9 + D + 1));
    for (int f = 0; f < n; ++f) {
      double* Af = &Ah[(size_t)f * 81];
      std::memcpy(Af, &A[(size_t)f * 81], 81 * sizeof(double));
      for (int i = 0; i < df; ++i) Af[i * 9 + i] += lam[(size_t)f * 9 + i];
      std::memcpy(&Wh[(size_t)f * 9 * D], &W[(size_t)f * 9 * D], (size_t)9 * D * sizeof(double));
      std::memcpy(&gh[(size_t)f * 9], &gf[(size_t)f * 9], 9 * sizeof(double));
      if (f > 0 && L.imu) {
        // eliminate coupling to f-1: X = Ahat_{f-1}^{-1} [C | What | ghat]
        const double* Lp = &Ah[(size_t)(f - 1) * 81];  // already factored
        const int nc = df + D + 1;
        for (int i = 0; i < df; ++i) {
          for (int j = 0; j < df; ++j) X[i * nc + j] = Cs[(size_t)(f - 1) * 81 + i * 9 + j];
          for (int j = 0; j < D; ++j) X[i * nc + df + j] = Wh[((size_t)(f - 1) * 9 + i) * D + j];
          X[i * nc + df + D] = gh[(size_t)(f - 1) * 9 + i];
        }
        chol_solve(Lp, df, 9, X.data(), nc, nc);
        const double* Cp = &Cs[(size_t)(f - 1) * 81];
        for (int i = 0; i < df; ++i) {
          for (int j = 0; j < df; ++j) { double s = 0; for (int k = 0; k < df; ++k) s += Cp[k * 9 + i] * X[k * nc + j]; Af[i * 9 + j] -= s; }
          for (int j = 0; j < D; ++j) { double s = 0; for (int k = 0; k < df; ++k) s += Cp[k * 9 + i] * X[k * nc + df + j]; Wh[((size_t)f * 9 + i) * D + j] -= s; }
          double s = 0; for (int k = 0; k < df; ++k) s += Cp[k * 9 + i] * X[k * nc + df + D];
          gh[(size_t)f * 9 + i] -= s;
        }
      }
      if (!chol(Af, df, 9)) return false;
      // S -= What^T Ahat^-1 What ; gr -= What^T Ahat^-1 ghat
      const int nc = D + 1;
      std::vector<double> Y((size_t)df * nc);
      for (int i = 0; i < df; ++i) { for (int j = 0; j < D; ++j) Y[(size_t)i * nc + j] = Wh[((size_t)f * 9 + i) * D + j]; Y[(size_t)i * nc + D] = gh[(size_t)f * 9 + i]; }
      chol_solve(Af, df, 9, Y.data(), nc, nc);
      for (int a = 0; a < D; ++a) {
        for (int b = 0; b < D; ++b) { double s = 0; for (int k = 0; k < df; ++k) s += Wh[((size_t)f * 9 + k) * D + a] * Y[(size_t)k * nc + b]; S[(size_t)a * D + b] -= s; }
        double s = 0; for (int k = 0; k < df; ++k) s += Wh[((size_t)f * 9 + k) * D + a] * Y[(size_t)k * nc + D];
        gr[a] -= s;
      }
    }
    dsv.assign(D, 0.0);
    if (D > 0) {
      if (!chol(S.data(), D, D)) return false;
      for (int i = 0; i < D; ++i) dsv[i] = -gr[i];
      chol_solve(S.data(), D, D, dsv.data(), 1, 1);
    }
    // back substitution: delta_f = -Ahat^-1 (ghat + What ds + C_f delta_{f+1})
    dfv.assign((size_t)n * 9, 0.0);
    for (int f = n - 1; f >= 0; --f) {
      double rhs[9];
      for (int i = 0; i < df; ++i) {
        double s = gh[(size_t)f * 9 + i];
        for (int j = 0; j < D; ++j) s += Wh[((size_t)f * 9 + i) * D + j] * dsv[j];
        if (f + 1 < n && L.imu) for (int j = 0; j < df; ++j) s += Cs[(size_t)f * 81 + i * 9 + j] * dfv[(size_t)(f + 1) * 9 + j];
        rhs[i] = -s;
      }
      chol_solve(&Ah[(size_t)f * 81], df, 9, rhs, 1, 1);
      for (int i = 0; i < df; ++i) dfv[(size_t)f * 9 + i] = rhs[i];
    }
    return true;
  }
  // Dense cross-check of solve_blocks (small problems only).
  bool solve_dense(const std::vector<double>& lam, std::vector<double>& dfv, std::vector<double>& dsv) {
    const int n = N(), D = L.D, df = L.df, T = n * df + D;
    std::vector<double> H((size_t)T * T, 0.0), rhs(T);
    for (int f = 0; f < n; ++f) {
      for (int i = 0; i < df; ++i) {
        for (int j = 0; j < df; ++j) H[(size_t)(f * df + i) * T + f * df + j] = A[(size_t)f * 81 + i * 9 + j];
        H[(size_t)(f * df + i) * T + f * df + i] += lam[(size_t)f * 9 + i];
        for (int j = 0; j < D; ++j) { const double v = W[((size_t)f * 9 + i) * D + j]; H[(size_t)(f * df + i) * T + n * df + j] = v; H[(size_t)(n * df + j) * T + f * df + i] = v; }
        if (f + 1 < n) for (int j = 0; j < df; ++j) { const double v = Cc[(size_t)f * 81 + i * 9 + j]; H[(size_t)(f * df + i) * T + (f + 1) * df + j] = v; H[(size_t)((f + 1) * df + j) * T + f * df + i] = v; }
        rhs[f * df + i] = -gf[(size_t)f * 9 + i];
      }
    }
    for (int i = 0; i < D; ++i) { for (int j = 0; j < D; ++j) H[(size_t)(n * df + i) * T + n * df + j] = Hss[(size_t)i * D + j]; H[(size_t)(n * df + i) * T + n * df + i] += lam[(size_t)n * 9 + i]; rhs[n * df + i] = -gs[i]; }
    if (!chol(H.data(), T, T)) return false;
    chol_solve(H.data(), T, T, rhs.data(), 1, 1);
    dfv.assign((size_t)n * 9, 0.0); dsv.assign(D, 0.0);
    for (int f = 0; f < n; ++f) for (int i = 0; i < df; ++i) dfv[(size_t)f * 9 + i] = rhs[f * df + i];
    for (int i = 0; i < D; ++i) dsv[i] = rhs[n * df + i];
    return true;
  }

  // ---- state handling -------------------------------------------------------------
  struct State { std::vector<Frame> frames; std::vector<Camera> cams; double g[2], b[6], sf[6], toff; };
  State snapshot() const { State s; s.frames = frames; s.cams = cams; std::memcpy(s.g, g_dir, 16); std::memcpy(s.b, biases, 48); std::memcpy(s.sf, scale, 48); s.toff = time_offset; return s; }
  void restore(const State& s) { frames = s.frames; cams = s.cams; std::memcpy(g_dir, s.g, 16); std::memcpy(biases, s.b, 48); std::memcpy(scale, s.sf, 48); time_offset = s.toff; }
  // x <- Plus(x, delta); returns squared ambient step norm.  Also returns |x|^2 of free blocks (before).
  void apply_step(const std::vector<double>& dfv, const std::vector<double>& dsv, double* step2, double* xnorm2) {
    double s2 = 0, x2 = 0;
    for (int f = 0; f < N(); ++f) {
      Frame& fr = frames[f];
      double o[7];
      plus_se3(fr.T_wk, &dfv[(size_t)f * 9], o);
      for (int i = 0; i < 7; ++i) { x2 += fr.T_wk[i] * fr.T_wk[i]; const double d = o[i] - fr.T_wk[i]; s2 += d * d; fr.T_wk[i] = o[i]; }
      if (L.df == 9) for (int i = 0; i < 3; ++i) { x2 += fr.v[i] * fr.v[i]; const double d = dfv[(size_t)f * 9 + 6 + i]; s2 += d * d; fr.v[i] += d; }
    }
    for (size_t c = 0; c < cams.size(); ++c) {
      Camera& cm = cams[c];
      if (L.rot[c] >= 0) { double o[4]; plus_so3(cm.T_ck, &dsv[L.rot[c]], o); for (int i = 0; i < 4; ++i) { x2 += cm.T_ck[i] * cm.T_ck[i]; const double d = o[i] - cm.T_ck[i]; s2 += d * d; cm.T_ck[i] = o[i]; } }
      if (L.trans[c] >= 0) for (int i = 0; i < 3; ++i) { x2 += cm.T_ck[4 + i] * cm.T_ck[4 + i]; const double d = dsv[L.trans[c] + i]; s2 += d * d; cm.T_ck[4 + i] += d; }
      if (L.kk[c] >= 0) for (int i = 0; i < cm.nk; ++i) { x2 += cm.K[i] * cm.K[i]; const double d = dsv[L.kk[c] + i]; s2 += d * d; cm.K[i] += d; }
    }
    if (L.g >= 0) for (int i = 0; i < 2; ++i) { x2 += g_dir[i] * g_dir[i]; const double d = dsv[L.g + i]; s2 += d * d; g_dir[i] += d; }
    if (L.b >= 0) for (int i = 0; i < 6; ++i) { x2 += biases[i] * biases[i]; const double d = dsv[L.b + i]; s2 += d * d; biases[i] += d; }
    if (L.sf >= 0) for (int i = 0; i < 6; ++i) { x2 += scale[i] * scale[i]; const double d = dsv[L.sf + i]; s2 += d * d; scale[i] += d; }
    if (L.toff >= 0) { x2 += time_offset * time_offset; const double d = dsv[L.toff]; s2 += d * d; time_offset += d; }
    *step2 = s2; *xnorm2 = x2;
  }

  // UpdateImuWeights, vicalibrator.h:723-799.
  void update_imu_weights() {
    if (!(is_inertial_active && !rotation_only)) return;
#pragma omp parallel for schedule(dynamic, 4) num_threads(opt.num_threads)
    for (int j = 1; j < N(); ++j) {
      std::vector<ImuMeas<double>> meas;
      imu.range(frames[j - 1].time, frames[j].time, time_offset, &meas);
      if (meas.empty()) continue;
      imu_weight_sqrt(meas, frames[j - 1].T_wk, frames[j - 1].v, frames[j].T_wk, biases, scale, g_dir, gyro_sigma, accel_sigma,
                      &imu_w_sqrt[(size_t)(j - 1) * 81]);
    }
  }

  // diag of H in the lam indexing
  void hdiag(std::vector<double>& d) const {
    const int n = N(), D = L.D;
    d.assign((size_t)n * 9 + D, 0.0);
    for (int f = 0; f < n; ++f) for (int i = 0; i < L.df; ++i) d[(size_t)f * 9 + i] = A[(size_t)f * 81 + i * 9 + i];
    for (int i = 0; i < D; ++i) d[(size_t)n * 9 + i] = Hss[(size_t)i * D + i];
  }
  void grad_norms(double* gmax, double* g2) const {
    double m = 0, s = 0;
    for (int f = 0; f < N(); ++f) for (int i = 0; i < L.df; ++i) { const double v = gf[(size_t)f * 9 + i]; m = std::max(m, std::fabs(v)); s += v * v; }
    for (int i = 0; i < L.D; ++i) { m = std::max(m, std::fabs(gs[i])); s += gs[i] * gs[i]; }
    *gmax = m; *g2 = std::sqrt(s);
  }

  // ---- the trust-region (Levenberg-Marquardt) loop ---------------------------------
  // Restates ceres::internal::TrustRegionMinimizer + LevenbergMarquardtStrategy
  // with the reference's options (vicalibrator.h:141-151) except the strategy
  // (north_star: LM).  Returns the termination type (ceres::Solve :956).
  Termination solve_once(double* final_cost, int* num_residual_scalars) {
    build_layout();
    const int n = N(), D = L.D;
    double cost = linearize();
    long nres = 0;
    for (const Obs& o : obs) nres += 2 * std::max(0, vis_mult + o.mult_delta);
    if (L.imu) nres += (long)9 * imu_mult * std::max(0, n - 1);
    *num_residual_scalars = (int)nres;
    std::vector<double> hd, scale2((size_t)n * 9 + D, 1.0), lam((size_t)n * 9 + D, 0.0), diag;
    hdiag(hd);
    if (opt.jacobi_scaling) for (size_t i = 0; i < hd.size(); ++i) { const double s = 1.0 / (1.0 + std::sqrt(hd[i])); scale2[i] = s * s; }
    double radius = opt.initial_radius, decrease_factor = 2.0;
    bool reuse_diagonal = false;
    double gmax, gnorm;
    grad_norms(&gmax, &gnorm);
    IterRecord it0 = {0, cost, 0, gmax, gnorm, 0, 0, radius, 1, stage};
    trace.push_back(it0);
    if (gmax <= opt.gradient_tolerance) { *final_cost = cost; return kConvergence; }
    int iter = 0, invalid = 0;
    IterRecord last = it0;
    double xnorm2_cached = -1;
    while (true) {
      if (!iteration_callback(last)) { *final_cost = cost; return kUserSuccess; }
      if (iter >= opt.max_iters) { *final_cost = cost; return kNoConvergence; }
      ++iter;
      IterRecord rec = {iter, cost, 0, gmax, gnorm, 0, 0, radius, 0, stage};
      // LevenbergMarquardtStrategy::ComputeStep
      if (!reuse_diagonal) {
        diag = hd;
        for (size_t i = 0; i < diag.size(); ++i) diag[i] = std::min(std::max(diag[i] * scale2[i], opt.min_diag), opt.max_diag);
      }
      for (size_t i = 0; i < lam.size(); ++i) lam[i] = diag[i] / (radius * scale2[i]);
      reuse_diagonal = true;
      std::vector<double> dfv, dsv;
      bool ok = opt.dense_check ? solve_dense(lam, dfv, dsv) : solve_blocks(lam, dfv, dsv);
      double model_change = 0;
      if (ok) {
        // model_cost_change = -(g.d + 1/2 d^T H d) = -1/2 g.d + 1/2 d^T Lambda d
        double gd = 0, dld = 0;
        for (int f = 0; f < n; ++f) for (int i = 0; i < L.df; ++i) { const double d = dfv[(size_t)f * 9 + i]; gd += gf[(size_t)f * 9 + i] * d; dld += lam[(size_t)f * 9 + i] * d * d; }
        for (int i = 0; i < D; ++i) { gd += gs[i] * dsv[i]; dld += lam[(size_t)n * 9 + i] * dsv[i] * dsv[i]; }
        model_change = -0.5 * gd + 0.5 * dld;
      }
      if (!ok || !(model_change > 0)) {
        if (++invalid >= 5) { *final_cost = cost; trace.push_back(rec); return kFailure; }
        radius *= 0.5; rec.radius = radius;
        trace.push_back(rec); last = rec;
        continue;
      }
      invalid = 0;
      const State saved = snapshot();
      double step2, xnorm2;
      apply_step(dfv, dsv, &step2, &xnorm2);
      (void)xnorm2_cached;
      const double new_cost = evaluate_cost();
      rec.step_norm = std::sqrt(step2);
      const double xnorm = std::sqrt(xnorm2);
      if (rec.step_norm <= opt.parameter_tolerance * (xnorm + opt.parameter_tolerance)) {
        restore(saved); trace.push_back(rec); *final_cost = cost; return kConvergence;
      }
      rec.cost_change = cost - new_cost;
      if (std::fabs(rec.cost_change) < opt.function_tolerance * cost) {
        restore(saved); trace.push_back(rec); *final_cost = cost; return kConvergence;
      }
      rec.relative_decrease = rec.cost_change / model_change;
      if (rec.relative_decrease > opt.min_relative_decrease) {
        rec.accepted = 1;
        cost = linearize();   // new Jacobian with the weights current at this time (callback runs after)
        hdiag(hd);
        grad_norms(&gmax, &gnorm);
        rec.cost = cost; rec.gradient_max_norm = gmax; rec.gradient_norm = gnorm;
        const double q = 2.0 * rec.relative_decrease - 1.0;
        radius = radius / std::max(1.0 / 3.0, 1.0 - q * q * q);
        radius = std::min(opt.max_radius, radius);
        decrease_factor = 2.0; reuse_diagonal = false;
        rec.radius = radius;
        trace.push_back(rec); last = rec;
        if (gmax <= opt.gradient_tolerance) { *final_cost = cost; return kConvergence; }
      } else {
        restore(saved);
        radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
        rec.radius = radius;
        trace.push_back(rec); last = rec;
        if (radius < opt.min_radius) { *final_cost = cost; return kConvergence; }
      }
    }
  }

  // operator()(IterationSummary), vicalibrator.h:690-721. false = terminate.
  bool iteration_callback(const IterRecord& s) {
    update_imu_weights();
    ++num_iterations;
    last_cost_for_mse = s.cost;
    if (s.gradient_norm > 0 && s.gradient_norm < 1e-9) return false;
    return true;
  }
  double last_cost_for_mse = 0;

  // Per-camera RMSE, vicalibrator.h:958-971: sqrt( (1/2 sum |r|^2) / n_blocks ),
  // evaluated without the loss on the latest copy of each camera's blocks.
  void compute_rmse() {
    const size_t C = cams.size();
    cam_rmse.assign(C, 0.0);
    std::vector<double> s(C, 0.0); std::vector<long> cnt(C, 0);
    for (const Obs& o : obs) {
      if (o.mult_delta < 0 && !latest_copy_full) continue;   // removed from the latest copy (:911-914)
      double r[2]; reproj_value(o, r);
      s[o.cam] += 0.5 * (r[0] * r[0] + r[1] * r[1]); cnt[o.cam]++;
    }
    for (size_t c = 0; c < C; ++c) cam_rmse[c] = std::sqrt(s[c] / (double)cnt[c]);
  }
  // RemoveOutliers, vicalibrator.h:859-916.
  void remove_outliers_pass() {
    for (Obs& o : obs) {
      if (o.mult_delta < 0) continue;
      double r[2]; reproj_value(o, r);
      const double err = std::sqrt(r[0] * r[0] + r[1] * r[1]);
      if (err > opt.outlier_threshold * cam_rmse[o.cam]) o.mult_delta = -1;
    }
  }
  // Gravity initialisation, vicalibrator.h:927-949.
  void init_gravity() {
    const int idx = N() / 2;
    size_t k; const double zero = 0.0;
    ImuMeas<double> m = imu.element(frames[idx].time, zero, &k);
    const double nrm = std::sqrt(m.a[0] * m.a[0] + m.a[1] * m.a[1] + m.a[2] * m.a[2]);
    double gb[3] = {m.a[0] / nrm, m.a[1] / nrm, m.a[2] / nrm}, gw[3];
    quat_rotate(frames[idx].T_wk, gb, gw);
    const double p = std::asin(gw[1]);
    const double q = std::asin(-gw[0] / std::cos(p));
    g_dir[0] = p; g_dir[1] = q;
    gravity_initialized = true;
  }

  // SolveThread, vicalibrator.h:919-1040 (single-threaded, blocking).
  int solve() {
    sort_obs();
    is_finished = false;
    int guard = 0;
    while (!is_finished && guard++ < 64) {
      // SetupProblem :548-679: re-adds every block -> multiplicities
      if (is_visual_active) { vis_mult += 1; latest_copy_full = true; }
      if (imu_w_sqrt.size() != (size_t)std::max(0, N() - 1) * 81) {
        imu_w_sqrt.assign((size_t)std::max(0, N() - 1) * 81, 0.0);
        for (int j = 0; j + 1 < N(); ++j) for (int i = 0; i < 9; ++i) imu_w_sqrt[(size_t)j * 81 + i * 10] = 500.0;  // :616
      }
      if (opt.calibrate_imu && is_inertial_active) imu_mult += 1;
      // outlier deltas apply to the latest copy only; a re-add creates a fresh full copy:
      // older copies keep their removed blocks removed -> fold into per-obs deltas (kept as is).
      if (is_inertial_active && !rotation_only && !gravity_initialized) init_gravity();
      bool stage_done = false;
      int inner_guard = 0;
      while (!stage_done && !is_finished && inner_guard++ < 64) {
        if (obs.empty()) { is_finished = true; break; }
        update_imu_weights();                      // :955
        double fc; int nr;
        const Termination t = solve_once(&fc, &nr); // :956
        compute_rmse();                            // :959-971
        mse = fc / std::max(1, nr);                // :975
        ++stage;
        if (t != kNoConvergence && opt.calibrate_imu) {
          if (!is_inertial_active) is_inertial_active = true;                 // :978-981
          else if (rotation_only) { rotation_only = false; is_bias_active = true; }  // :982-990
          else if (!is_scale_active) is_scale_active = true;                  // :991-994
          else if (opt.remove_outliers && !outliers_removed) { remove_outliers_pass(); outliers_removed = true; latest_copy_full = false; }
          else is_finished = true;
          stage_done = true;                        // break :1022 -> SetupProblem again
        } else if (t != kNoConvergence) {
          if (opt.remove_outliers && !outliers_removed) { remove_outliers_pass(); outliers_removed = true; latest_copy_full = false; }
          else is_finished = true;
        }
      }
    }
    return 0;
  }
};

}  // namespace vco
