// Test-only host build of vicalib_amd/csrc/vc_math.hpp (the arithmetic the HIP kernels call), so
// that the closed-form Jacobians and the tile -> frame/camera block algebra can be checked against
// the oracle on a machine without a GPU.  Emulates what one wavefront of k_reproj_jac and one thread
// of k_frame_prep / k_schur_final compute, serially.  Not part of the product library.
#include <cstring>
#include "../../vicalib_amd/csrc/vc_math.hpp"
#include <vector>
#include "../../vicalib_amd/csrc/vc_imu.hpp"
#include "seq_weights.hpp"
using namespace vc;

template <int MODEL>
static double gram(const TileXf& x, const double* K, int n, const double* pw, const double* uv, double mult, double* G) {
  double cost = 0;
  ModelPre pre; model_precompute(MODEL, K, &pre);
  for (int d = 0; d < n; ++d) {
    double r0[16], r1[16], rs[2];
    cost += corner_rows<MODEL>(x, K, pre, pw + 3 * d, uv[2 * d], uv[2 * d + 1], mult, r0, r1, rs);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) G[i * 16 + j] += r0[i] * r0[j] + r1[i] * r1[j];
    if (MODEL == kRational6) for (int i = 0; i < 16; ++i) G[kGGrad + i] += r0[i] * rs[0] + r1[i] * rs[1];     // the side vector (see kGGrad)
  }
  return cost;
}
template <int MODEL>
static double resid(const TileXf& x, const double* K, int n, const double* pw, const double* uv, double* r) {
  double cost = 0;
  ModelPre pre; model_precompute(MODEL, K, &pre);
  for (int d = 0; d < n; ++d) cost += corner_residual<MODEL>(x, K, pre, pw + 3 * d, uv[2 * d], uv[2 * d + 1], r + 2 * d);
  return cost;
}

extern "C" {
void hh_project(int model, const double* pc, const double* K, double* pix, double* A, double* B) { ModelPre pre; model_precompute(model, K, &pre); project_any<true>(model, pc, K, pre, pix, A, B); }
double hh_tile_gram(int model, const double* T_wk, const double* T_ck, const double* K, int n, const double* pw, const double* uv,
                    double mult, double* G) {
  TileXf x; make_tile_xf(T_wk, T_ck, &x);
  std::memset(G, 0, 272 * sizeof(double));       // the kernels' Gram record: 16 x 16 block + side vector
  switch (model) {
    case kFov: return gram<kFov>(x, K, n, pw, uv, mult, G);
    case kPoly2: return gram<kPoly2>(x, K, n, pw, uv, mult, G);
    case kPoly3: return gram<kPoly3>(x, K, n, pw, uv, mult, G);
    case kKb4: return gram<kKb4>(x, K, n, pw, uv, mult, G);
    case kRational6: return gram<kRational6>(x, K, n, pw, uv, mult, G);
    default: return gram<kLinear>(x, K, n, pw, uv, mult, G);
  }
}
double hh_tile_resid(int model, const double* T_wk, const double* T_ck, const double* K, int n, const double* pw, const double* uv, double* r) {
  TileXf x; make_tile_xf(T_wk, T_ck, &x);
  switch (model) {
    case kFov: return resid<kFov>(x, K, n, pw, uv, r);
    case kPoly2: return resid<kPoly2>(x, K, n, pw, uv, r);
    case kPoly3: return resid<kPoly3>(x, K, n, pw, uv, r);
    case kKb4: return resid<kKb4>(x, K, n, pw, uv, r);
    case kRational6: return resid<kRational6>(x, K, n, pw, uv, r);
    default: return resid<kLinear>(x, K, n, pw, uv, r);
  }
}
void hh_frame_blocks(const double* G, const double* q_ck, int nk, int flags, double* Hff, double* gf, double* W) {
  double R[9]; quat_to_R(q_ck, R);
  tile_to_frame_blocks(G, R, nk, flags, Hff, gf, W);
}
void hh_cam_block(const double* G, const double* q_ck, int nk, int flags, double* Hcc, double* gc) {
  double R[9]; quat_to_R(q_ck, R);
  cam_block_from_gsum(G, R, nk, flags, Hcc, gc);
}
void hh_se3_plus(const double* T, const double* d, double* o) { se3_plus(T, d, o); }
void hh_so3_plus(const double* q, const double* w, double* o) { so3_plus(q, w, o); }
// IMU block as the wavefront of k_imu_jac computes it: one derivative direction per "lane", then the
// local-parameterisation Jacobians.  J out: 9 x 33 row-major in the oracle's column order
// (T_j local 6, T_{j-1} local 6, v_j 3, v_{j-1} 3, g 2, b 6, sf 6, toff 1).
void hh_imu_block(int n, const double* t, const double* w, const double* a, double t_start, double t_end, const double* w_sqrt,
                  int rotation_only, const double* T2, const double* T1, const double* v2, const double* v1, const double* gdir,
                  const double* b, const double* sf, double toff, double* r, double* J) {
  ImuView buf = {t, w, a, n, imu_average_dt(t, n)};
  double Jg[35][9];
  for (int d = 0; d < 35; ++d) imu_block_direction(buf, t_start, t_end, w_sqrt, rotation_only, T2, T1, v2, v1, gdir, b, sf, toff, d, r, Jg[d]);
  double P2[42], P1[42];
  local_jac_se3(T2, P2); local_jac_se3(T1, P1);
  for (int row = 0; row < 9; ++row) {
    double* o = J + row * 33;
    for (int c = 0; c < 6; ++c) {
      double s2 = 0, s1 = 0;
      for (int i = 0; i < 7; ++i) { s2 += Jg[i][row] * P2[i * 6 + c]; s1 += Jg[7 + i][row] * P1[i * 6 + c]; }
      o[c] = s2; o[6 + c] = s1;
    }
    for (int c = 0; c < 21; ++c) o[12 + c] = Jg[14 + c][row];
  }
}
// The same block in the delta form the device runs (vc_imu.hpp: imu_block_delta_record = the block's interval deltas from the
// identity state -- one dual direction per gyro parameter / the time offset, accelerometer partials analytically --, composed in
// k_imu_block's bracketing, then imu_block_final_direction per local column).  Same output layout as hh_imu_block.
void hh_imu_block_deltas(int n, const double* t, const double* w, const double* a, double t_start, double t_end, const double* w_sqrt,
                         int rotation_only, const double* T2, const double* T1, const double* v2, const double* v1, const double* gdir,
                         const double* b, const double* sf, double toff, double* r, double* J) {
  ImuView buf = {t, w, a, n, imu_average_dt(t, n)};
  std::vector<double> blk(kBlockDeltaStride, 0.0);
  const int valid = imu_block_delta_record(buf, t_start, t_end, toff, b, sf, blk.data());
  double grav9[9];
  imu_gravity_record(gdir, grav9);
  for (int oc = 0; oc < 33; ++oc) {
    const int col = oc < 6 ? oc : oc < 12 ? oc + 3 : oc < 15 ? oc - 6 : oc;
    double dr[9];
    imu_block_final_direction(valid, blk.data(), w_sqrt, rotation_only, T2, T1, v2, v1, grav9, col, r, dr);
    for (int row = 0; row < 9; ++row) J[row * 33 + oc] = dr[row];
  }
}
void hh_imu_weight(int n, const double* t, const double* w, const double* a, double t_start, double t_end, double toff,
                   const double* T1, const double* v1, const double* T2, const double* b, const double* sf, const double* gdir,
                   double gs, double as, double* w_sqrt) {
  ImuView buf = {t, w, a, n, imu_average_dt(t, n)};
  imu_weight_sqrt(buf, t_start, t_end, toff, T1, v1, T2, b, sf, gdir, gs, as, w_sqrt);
}
// The same weight in the interval-parallel form k_imu_weights runs (vc_imu_weights.hpp: maps per interval at the prefix state,
// fold, Cholesky-form factor W = L^-T).  Returns 1 if W was written.
int hh_imu_weight_intervals(int n, const double* t, const double* w, const double* a, double t_start, double t_end, double toff,
                            const double* T1, const double* v1, const double* T2, const double* b, const double* sf, const double* gdir,
                            double gs, double as, double* W) {
  ImuView buf = {t, w, a, n, imu_average_dt(t, n)};
  return imu_weight_factor_intervals(buf, t_start, t_end, toff, T1, v1, T2, b, sf, gdir, gs, as, W);
}
void hh_dlog_dse3(const double* T, double* full, double* lean) { w_dlog_dse3(T, full); w_dlog_dse3_lean(T, lean); }
// the Jacobian of the device's SE3 logarithm (tse3_log, what k_imu_jac differentiates) by dual numbers, 6 x 7 row-major with the
// columns in dLog_dSE3's order [t(3), q(4)]; T stored [q, t]
void hh_se3_log_dual_jacobian(const double* T, double* J42) {
  for (int c = 0; c < 7; ++c) {
    const int kk = c < 3 ? 4 + c : c - 3;
    D1 X[7], d[6];
    for (int k = 0; k < 7; ++k) X[k] = mk(T[k], k == kk ? 1.0 : 0.0);
    tse3_log(X, d);
    for (int r = 0; r < 6; ++r) J42[r * 7 + c] = d[r].v;
  }
}
int hh_chol6(double* M) { return chol_small<6>(M) ? 1 : 0; }
// exp of a rotation vector with the left Jacobian's coefficients (vc_imu.hpp: so3_exp_jl, what the closed-form partials of the IMU
// block deltas use) and exp's derivative along `dir` under a dual number (tso3_exp<D1>): large angles go through halving + squaring
void hh_so3_exp_jl(const double* w, const double* dir, double* q, double* AB, double* dq) {
  so3_exp_jl(w, q, AB, AB + 1);
  D1 W[3], Q[4];
  for (int k = 0; k < 3; ++k) W[k] = mk(w[k], dir[k]);
  tso3_exp(W, Q);
  for (int k = 0; k < 4; ++k) dq[k] = Q[k].v;
}
}
