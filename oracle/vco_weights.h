// ORACLE -- TEST INFRASTRUCTURE ONLY (see vco_math.h header). PARITY UNPINNED.
//
// vco_weights.h: the covariance-weight path of UpdateImuWeights
// (vicalibrator.h:723-799): double-precision RK4 with hand-derived state and
// bias Jacobians and covariance propagation (ImuResidualT::IntegratePose
// types.h:330-378, GetPoseDerivative :380-425, IntegrateImu :427-595,
// IntegrateResidual :611-687) and the small Jacobian helpers it uses
// (vicalibrator-utils.h: dLog_dq :107-154, dqExp_dw :188-202, dq1q2_dq2 :215,
// dq1q2_dq1 :225, dqx_dq :235-254, dt1t2_dt1 :261-274, dLog_dSE3 :308-434).
// The approximations the reference makes are part of its result and are kept:
// dqExp_dw is a low-order series, dk/dx ignores the scale factors
// (types.h:417-422), and the theta<eps branch of dLog_dSE3 carries the
// reference's (wx*x)*(wy*y) term (vicalibrator-utils.h:372).
// State vectors here are [p(3), q(4: x,y,z,w), v(3)] (types.h:188-194).
#pragma once
#include <vector>
#include <cmath>
#include <cstring>
#include "vco_math.h"
#include "vco_imu.h"

namespace vco {

struct Mat {   // tiny dense row-major helper for the fixed-size algebra below
  int r, c; double d[100];
  Mat(int r_, int c_) : r(r_), c(c_) { for (int i = 0; i < r * c; ++i) d[i] = 0.0; }
  double& operator()(int i, int j) { return d[i * c + j]; }
  double operator()(int i, int j) const { return d[i * c + j]; }
  static Mat I(int n) { Mat m(n, n); for (int i = 0; i < n; ++i) m(i, i) = 1.0; return m; }
};
inline Mat mul(const Mat& a, const Mat& b) {
  Mat o(a.r, b.c);
  for (int i = 0; i < a.r; ++i) for (int j = 0; j < b.c; ++j) { double s = 0; for (int k = 0; k < a.c; ++k) s += a(i, k) * b(k, j); o(i, j) = s; }
  return o;
}
inline Mat tr(const Mat& a) { Mat o(a.c, a.r); for (int i = 0; i < a.r; ++i) for (int j = 0; j < a.c; ++j) o(j, i) = a(i, j); return o; }
inline Mat add(const Mat& a, const Mat& b, double sb = 1.0) { Mat o(a.r, a.c); for (int i = 0; i < a.r * a.c; ++i) o.d[i] = a.d[i] + sb * b.d[i]; return o; }
inline void set_block(Mat& m, int r0, int c0, const Mat& b) { for (int i = 0; i < b.r; ++i) for (int j = 0; j < b.c; ++j) m(r0 + i, c0 + j) = b(i, j); }

// vicalibrator-utils.h:215-220, d(q1*q2)/dq2 as a function of q1 ([x,y,z,w] order)
inline Mat dq1q2_dq2(const double* q1) {
  const double x = q1[0], y = q1[1], z = q1[2], w = q1[3];
  Mat m(4, 4);
  const double v[16] = {w, -z, y, x, z, w, -x, y, -y, x, w, z, -x, -y, -z, w};
  std::memcpy(m.d, v, sizeof(v));
  return m;
}
// vicalibrator-utils.h:225-230, d(q1*q2)/dq1 as a function of q2
inline Mat dq1q2_dq1(const double* q2) {
  const double x = q2[0], y = q2[1], z = q2[2], w = q2[3];
  Mat m(4, 4);
  const double v[16] = {w, z, -y, x, -z, w, x, y, y, -x, w, z, -x, -y, -z, w};
  std::memcpy(m.d, v, sizeof(v));
  return m;
}
// vicalibrator-utils.h:188-202 (series form of d exp(w) / dw, 4x3)
inline Mat dqexp_dw(const double* w) {
  const double t = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  const double s1 = t / 20 - 1;
  const double s2 = t * t / 48 - 0.5;
  const double s6 = t * t;
  Mat m(4, 3);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      m(i, j) = (i == j) ? (s1 * w[i] * w[i]) / 24 - s6 / 48 + 0.5 : (s1 * w[i] * w[j]) / 24;
  for (int j = 0; j < 3; ++j) m(3, j) = (s2 * w[j]) / 2;
  return m;
}
// vicalibrator-utils.h:235-254, d(q * vec)/dq, 3x4
inline Mat dqx_dq(const double* q, const double* vec) {
  const double x = vec[0], y = vec[1], z = vec[2];
  const double qx = q[0], qy = q[1], qz = q[2], qw = q[3];
  Mat m(3, 4);
  m(0, 0) = 2 * qy * y + 2 * qz * z;
  m(0, 1) = 2 * qx * y - 4 * qy * x + 2 * qw * z;
  m(0, 2) = 2 * qx * z - 2 * qw * y - 4 * qz * x;
  m(0, 3) = 2 * qy * z - 2 * qz * y;
  m(1, 0) = 2 * qy * x - 4 * qx * y - 2 * qw * z;
  m(1, 1) = 2 * qx * x + 2 * qz * z;
  m(1, 2) = 2 * qy * z + 2 * qw * x - 4 * qz * y;
  m(1, 3) = 2 * qz * x - 2 * qx * z;
  m(2, 0) = 2 * qz * x + 2 * qw * y - 4 * qx * z;
  m(2, 1) = 2 * qz * y - 2 * qw * x - 4 * qy * z;
  m(2, 2) = 2 * qy * y + 2 * qx * x;
  m(2, 3) = 2 * qx * y - 2 * qy * x;
  return m;
}
// vicalibrator-utils.h:107-154, d log(q) / dq, 3x4
inline Mat dlog_dq(const double* q) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double v[3] = {x, y, z};
  const double n2 = x * x + y * y + z * z;
  const double n = std::sqrt(n2);
  Mat m(3, 4);
  if (n < 1e-9) {   // kTestingEps, vicalibrator-utils.h:52
    const double s1 = 2 * n2, s2 = 1.0 / (w * w * w);
    const double s3 = (3 * s1) / (w * w * w * w) - 2 / (w * w), s4 = 2 / w;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) m(i, j) = (i == j) ? -4 * s2 * v[i] * v[i] + s4 - s1 * s2 : -4 * v[i] * v[j] * s2;
      m(i, 3) = v[i] * s3;
    }
  } else {
    const double s1 = n2;
    const double s2 = 1 / (s1 / (w * w) + 1);
    const double s3 = std::atan(std::sqrt(s1) / w);
    const double s4 = 1 / std::pow(s1, 1.5);
    const double s5 = 1 / s1, s6 = 1 / w;
    const double s7 = (2 * s3) / std::sqrt(s1);
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) {
        const double off = 2 * v[i] * v[j] * s2 * s5 * s6 - 2 * v[i] * v[j] * s3 * s4;
        m(i, j) = (i == j) ? s7 - 2 * v[i] * v[i] * s3 * s4 + 2 * v[i] * v[i] * s2 * s5 * s6 : off;
      }
      m(i, 3) = -(2 * v[i] * s2) / (w * w);
    }
  }
  return m;
}
// vicalibrator-utils.h:308-434: d log(T) / dT, 6x7 with T as [t(3), q(4)].
inline Mat dlog_dse3(const double* T /* [q, t] storage */) {
  const Mat dw_dq = dlog_dq(T);
  const double x = T[4], y = T[5], z = T[6];
  double w[3], theta;
  so3_log(T, w, &theta);
  const double wx = w[0], wy = w[1], wz = w[2];
  double O[9], O2[9];
  hat(w, O); mat3_mul(O, O, O2);
  const bool small = std::fabs(theta) < kSophusEps;
  const double c = small ? 1.0 / 12.0 : (1.0 - theta / (2.0 * std::tan(theta / 2.0))) / (theta * theta);
  Mat vinv(3, 3);
  for (int i = 0; i < 9; ++i) vinv.d[i] = -0.5 * O[i] + c * O2[i];
  vinv(0, 0) += 1; vinv(1, 1) += 1; vinv(2, 2) += 1;
  Mat dl(6, 7);
  set_block(dl, 0, 0, vinv);
  set_block(dl, 3, 3, dw_dq);
  Mat dw(3, 3);
  if (small) {
    const double d12 = 1. / 12, d6 = 1. / 6.;
    dw(0, 0) = d12 * (wy * y + wz * z);
    dw(0, 1) = d12 * wx * y - d6 * wy * x - 0.5 * z;
    dw(0, 2) = 0.5 * y - d6 * wz * x + d12 * wx * z;
    dw(1, 0) = 0.5 * z + d12 * wy * x - d6 * wx * y;
    dw(1, 1) = d12 * (wx * x + wz * z);
    dw(1, 2) = d12 * wy * z - d6 * wz * y - 0.5 * x;
    dw(2, 0) = d12 * wz * x - d6 * wx * z - 0.5 * y;
    dw(2, 1) = 0.5 * x + d12 * wz * y - d6 * wy * z;
    dw(2, 2) = d12 * ((wx * x) * (wy * y));   // as in the reference (:372)
  } else {
    const double s1 = wx * wx + wy * wy + wz * wz;
    const double rs1 = std::sqrt(s1);
    const double s2 = std::tan(rs1 / 2);
    const double s3 = rs1 / (2 * s2) - 1;
    const double s4 = wz / (2 * rs1 * s2) - (wz * (s2 * s2 + 1)) / (4 * s2 * s2);
    const double s5 = wy / (2 * rs1 * s2) - (wy * (s2 * s2 + 1)) / (4 * s2 * s2);
    const double s6 = wx / (2 * rs1 * s2) - (wx * (s2 * s2 + 1)) / (4 * s2 * s2);
    const double s7 = 1 / s1, s8 = 1 / (s1 * s1);
    const double s9 = wx * wx + wy * wy, s10 = wx * wx + wz * wz, s11 = wy * wy + wz * wz;
    const double s12 = 2 * s3 * s8 * wx * wy * wz;
    const double s13 = -2 * s3 * s8 * wy * wz * wz + s4 * s7 * wy * wz + s3 * s7 * wy;
    const double s14 = -2 * s3 * s8 * wx * wz * wz + s4 * s7 * wx * wz + s3 * s7 * wx;
    const double s15 = -2 * s3 * s8 * wz * wy * wy + s5 * s7 * wz * wy + s3 * s7 * wz;
    const double s16 = -2 * s3 * s8 * wz * wx * wx + s6 * s7 * wz * wx + s3 * s7 * wz;
    const double s17 = -2 * s3 * s8 * wx * wy * wy + s5 * s7 * wx * wy + s3 * s7 * wx;
    const double s18 = -2 * s3 * s8 * wy * wx * wx + s6 * s7 * wy * wx + s3 * s7 * wy;
    const double s19 = 2 * s3 * s7 * wy, s20 = 2 * s3 * s7 * wx;
    dw(0, 0) = x * (s6 * s7 * s11 - 2 * s3 * s8 * s11 * wx) - s18 * y - s16 * z;
    dw(0, 1) = x * (s19 + s5 * s7 * s11 - 2 * s3 * s8 * s11 * wy) - s17 * y - z * (s5 * s7 * wx * wz - 2 * s3 * s8 * wx * wy * wz + 0.5);
    dw(0, 2) = x * (s4 * s7 * s11 + 2 * s3 * s7 * wz - 2 * s3 * s8 * s11 * wz) - s14 * z + y * (s12 - s4 * s7 * wx * wy + 0.5);
    dw(1, 0) = y * (s20 + s6 * s7 * s10 - 2 * s3 * s8 * s10 * wx) - s18 * x + z * (s12 - s6 * s7 * wy * wz + 0.5);
    dw(1, 1) = y * (s5 * s7 * s10 - 2 * s3 * s8 * s10 * wy) - s17 * x - s15 * z;
    dw(1, 2) = y * (s4 * s7 * s10 + 2 * s3 * s7 * wz - 2 * s3 * s8 * s10 * wz) - s13 * z - x * (s4 * s7 * wx * wy - s12 + 0.5);
    dw(2, 0) = z * (s20 + s6 * s7 * s9 - 2 * s3 * s8 * s9 * wx) - s16 * x - y * (s6 * s7 * wy * wz - s12 + 0.5);
    dw(2, 1) = z * (s19 + s5 * s7 * s9 - 2 * s3 * s8 * s9 * wy) - s15 * y + x * (s12 - s5 * s7 * wx * wz + 0.5);
    dw(2, 2) = z * (s4 * s7 * s9 - 2 * s3 * s8 * s9 * wz) - s14 * x - s13 * y;
  }
  set_block(dl, 0, 3, mul(dw, dw_dq));
  return dl;
}
// vicalibrator-utils.h:261-274: d(T1*T2)/dT1, 7x7 in [t, q] ordering.
inline Mat dt1t2_dt1(const double* T1, const double* T2) {
  Mat m(7, 7);
  m(0, 0) = m(1, 1) = m(2, 2) = 1.0;
  set_block(m, 0, 3, dqx_dq(T1, T2 + 4));
  set_block(m, 3, 3, dq1q2_dq1(T2));
  return m;
}

struct WPose { double T[7]; double v[3]; };   // [q,t] storage + velocity

// ImuResidualT::IntegratePose, types.h:330-378
inline WPose w_integrate_pose(const WPose& p, const double* k, double dt, Mat* dy_dk, Mat* dy_dy) {
  double wdt[3] = {k[3] * dt, k[4] * dt, k[5] * dt}, rq[4];
  so3_exp(wdt, rq);
  WPose y = p;
  for (int i = 0; i < 3; ++i) y.T[4 + i] = p.T[4 + i] + k[i] * dt;
  quat_mul(rq, p.T, y.T);
  for (int i = 0; i < 3; ++i) y.v[i] = p.v[i] + k[6 + i] * dt;
  if (dy_dk) {
    *dy_dk = Mat(10, 9);
    for (int i = 0; i < 3; ++i) { (*dy_dk)(i, i) = dt; (*dy_dk)(7 + i, 6 + i) = dt; }
    Mat b = mul(dq1q2_dq1(p.T), dqexp_dw(wdt));
    for (int i = 0; i < 12; ++i) b.d[i] *= dt;
    set_block(*dy_dk, 3, 3, b);
  }
  if (dy_dy) {
    *dy_dy = Mat(10, 10);
    for (int i = 0; i < 3; ++i) { (*dy_dy)(i, i) = 1.0; (*dy_dy)(7 + i, 7 + i) = 1.0; }
    set_block(*dy_dy, 3, 3, dq1q2_dq2(rq));
  }
  return y;
}
// ImuResidualT::GetPoseDerivative, types.h:380-425
inline void w_pose_derivative(const WPose& p, const double* g_w, const ImuMeas<double>& z0, const ImuMeas<double>& z1,
                              const double* bg, const double* ba, const double* sf, double dt, double* k, Mat* dk_db,
                              Mat* dk_dx) {
  const double alpha = (z1.time - (z0.time + dt)) / (z1.time - z0.time);
  double zg[3], za[3], u[3], R[9], o[3];
  for (int i = 0; i < 3; ++i) { zg[i] = z0.w[i] * alpha + z1.w[i] * (1.0 - alpha); za[i] = z0.a[i] * alpha + z1.a[i] * (1.0 - alpha); }
  for (int i = 0; i < 3; ++i) k[i] = p.v[i];
  quat_to_matrix(p.T, R);
  for (int i = 0; i < 3; ++i) u[i] = zg[i] * sf[i] + bg[i];
  mat3_vec(R, u, o);
  for (int i = 0; i < 3; ++i) k[3 + i] = o[i];
  for (int i = 0; i < 3; ++i) u[i] = za[i] * sf[3 + i] + ba[i];
  quat_rotate(p.T, u, o);
  for (int i = 0; i < 3; ++i) k[6 + i] = o[i] - g_w[i];
  *dk_db = Mat(9, 6);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { (*dk_db)(3 + i, j) = R[3 * i + j]; (*dk_db)(6 + i, 3 + j) = R[3 * i + j]; }
  *dk_dx = Mat(9, 10);
  for (int i = 0; i < 3; ++i) (*dk_dx)(i, 7 + i) = 1.0;
  set_block(*dk_dx, 3, 3, add(dqx_dq(p.T, zg), dqx_dq(p.T, bg)));
  set_block(*dk_dx, 6, 3, add(dqx_dq(p.T, za), dqx_dq(p.T, ba)));
}
// ImuResidualT::IntegrateImu with Jacobians and covariance, types.h:427-595
inline WPose w_integrate_imu(const WPose& pose, const ImuMeas<double>& z0, const ImuMeas<double>& z1, const double* bg,
                             const double* ba, const double* sf, const double* g, Mat* c_prior, const Mat& cov_meas) {
  const double dt = z1.time - z0.time;
  if (dt == 0) return pose;
  Mat dy_db(10, 6), dy_dy0 = Mat::I(10), dk_db(9, 6), dk_dy(9, 10), dy_dk(10, 9), dy_dy(10, 10);
  double k1[9], k2[9], k3[9], k4[9], k[9];
  w_pose_derivative(pose, g, z0, z1, bg, ba, sf, 0, k1, &dk_db, &dk_dy);
  const Mat dk1_db = add(dk_db, mul(dk_dy, dy_db)), dk1_dy = mul(dk_dy, dy_dy0);
  const WPose y1 = w_integrate_pose(pose, k1, dt * 0.5, &dy_dk, &dy_dy);
  dy_db = mul(dy_dk, dk1_db); dy_dy0 = add(dy_dy, mul(dy_dk, dk1_dy));
  w_pose_derivative(y1, g, z0, z1, bg, ba, sf, dt / 2, k2, &dk_db, &dk_dy);
  const Mat dk2_db = add(dk_db, mul(dk_dy, dy_db)), dk2_dy = mul(dk_dy, dy_dy0);
  const WPose y2 = w_integrate_pose(pose, k2, dt * 0.5, &dy_dk, &dy_dy);
  dy_db = mul(dy_dk, dk2_db); dy_dy0 = add(dy_dy, mul(dy_dk, dk2_dy));
  w_pose_derivative(y2, g, z0, z1, bg, ba, sf, dt / 2, k3, &dk_db, &dk_dy);
  const Mat dk3_db = add(dk_db, mul(dk_dy, dy_db)), dk3_dy = mul(dk_dy, dy_dy0);
  const WPose y3 = w_integrate_pose(pose, k3, dt, &dy_dk, &dy_dy);
  dy_db = mul(dy_dk, dk3_db); dy_dy0 = add(dy_dy, mul(dy_dk, dk3_dy));
  w_pose_derivative(y3, g, z0, z1, bg, ba, sf, dt, k4, &dk_db, &dk_dy);
  const Mat dk4_db = add(dk_db, mul(dk_dy, dy_db)), dk4_dy = mul(dk_dy, dy_dy0);
  for (int i = 0; i < 9; ++i) k[i] = k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i];
  const Mat dkt_db = add(add(dk1_db, dk2_db, 2.0), add(dk4_db, dk3_db, 2.0));
  const Mat dkt_dy = add(add(dk1_dy, dk2_dy, 2.0), add(dk4_dy, dk3_dy, 2.0));
  const WPose res = w_integrate_pose(pose, k, dt / 6.0, &dy_dk, &dy_dy);
  dy_db = mul(dy_dk, dkt_db); dy_dy0 = add(dy_dy, mul(dy_dk, dkt_dy));
  // Sigma <- F Sigma F^T + G R G^T  (types.h:571-573)
  *c_prior = add(mul(mul(dy_dy0, *c_prior), tr(dy_dy0)), mul(mul(dy_db, cov_meas), tr(dy_db)));
  return res;
}

// 9x9 inverse by LU with partial pivoting (Eigen's .inverse() for this size).
inline bool inverse9(const double* M, double* out) {
  const int n = 9;
  double a[81], inv[81];
  std::memcpy(a, M, sizeof(a));
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) inv[i * n + j] = (i == j);
  for (int c = 0; c < n; ++c) {
    int p = c;
    for (int r = c + 1; r < n; ++r) if (std::fabs(a[r * n + c]) > std::fabs(a[p * n + c])) p = r;
    if (a[p * n + c] == 0.0) return false;
    if (p != c) for (int j = 0; j < n; ++j) { std::swap(a[p * n + j], a[c * n + j]); std::swap(inv[p * n + j], inv[c * n + j]); }
    const double d = 1.0 / a[c * n + c];
    for (int r = 0; r < n; ++r) {
      if (r == c) continue;
      const double f = a[r * n + c] * d;
      if (f == 0.0) continue;
      for (int j = 0; j < n; ++j) { a[r * n + j] -= f * a[c * n + j]; inv[r * n + j] -= f * inv[c * n + j]; }
    }
    for (int j = 0; j < n; ++j) { a[c * n + j] *= d; inv[c * n + j] *= d; }
  }
  std::memcpy(out, inv, sizeof(inv));
  return true;
}
// Principal square root of a symmetric PSD 9x9 (MatrixBase::sqrt(), vicalibrator.h:796):
// cyclic Jacobi eigen-decomposition of the symmetrised input.
inline void sqrt_spd9(const double* M, double* out) {
  const int n = 9;
  double a[81], v[81];
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { a[i * n + j] = 0.5 * (M[i * n + j] + M[j * n + i]); v[i * n + j] = (i == j); }
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0, dg = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { if (i != j) off += a[i * n + j] * a[i * n + j]; else dg += a[i * n + i] * a[i * n + i]; }
    if (off <= 1e-60 * dg || off == 0) break;
    for (int p = 0; p < n - 1; ++p) for (int q = p + 1; q < n; ++q) {
      if (a[p * n + q] == 0.0) continue;
      const double th = (a[q * n + q] - a[p * n + p]) / (2.0 * a[p * n + q]);
      const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0));
      const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
      for (int k = 0; k < n; ++k) { const double akp = a[k * n + p], akq = a[k * n + q]; a[k * n + p] = cs * akp - sn * akq; a[k * n + q] = sn * akp + cs * akq; }
      for (int k = 0; k < n; ++k) { const double apk = a[p * n + k], aqk = a[q * n + k]; a[p * n + k] = cs * apk - sn * aqk; a[q * n + k] = sn * apk + cs * aqk; }
      for (int k = 0; k < n; ++k) { const double vkp = v[k * n + p], vkq = v[k * n + q]; v[k * n + p] = cs * vkp - sn * vkq; v[k * n + q] = sn * vkp + cs * vkq; }
    }
  }
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
    double s = 0;
    for (int k = 0; k < n; ++k) s += v[i * n + k] * std::sqrt(std::fmax(a[k * n + k], 0.0)) * v[j * n + k];
    out[i * n + j] = s;
  }
}

// One IMU cost's weight_sqrt_ (vicalibrator.h:737-796). meas = GetRange output.
inline void imu_weight_sqrt(const std::vector<ImuMeas<double>>& meas, const double* T1, const double* v1, const double* T2,
                            const double* biases, const double* sf, const double* g_dir, double gyro_sigma,
                            double accel_sigma, double* w_sqrt /*81*/, double* cov_out = nullptr) {
  double gvec[3];
  gravity_vector(g_dir, gravity_magnitude(), gvec);
  Mat R(6, 6);
  for (int i = 0; i < 3; ++i) { R(i, i) = gyro_sigma * gyro_sigma; R(3 + i, 3 + i) = accel_sigma * accel_sigma; }
  Mat sigma(10, 10);
  WPose pose;
  std::memcpy(pose.T, T1, 56); std::memcpy(pose.v, v1, 24);
  for (size_t i = 1; i < meas.size(); ++i) pose = w_integrate_imu(pose, meas[i - 1], meas[i], biases, biases + 3, sf, gvec, &sigma, R);
  double t2w[7], rel[7];
  se3_inv(T2, t2w);
  se3_mul(pose.T, t2w, rel);
  const Mat J67 = mul(dlog_dse3(rel), dt1t2_dt1(pose.T, t2w));
  Mat J(9, 10);
  set_block(J, 0, 0, J67);
  J(6, 7) = J(7, 8) = J(8, 9) = 1.0;
  const Mat P = mul(mul(J, sigma), tr(J));
  // Where the reference is undefined (Eigen's inverse() / sqrt() of a matrix that is not positive definite -- seen only for
  // the one block that straddles the end of a truncated IMU stream, itself a read past the buffer end in the reference,
  // interpolation-buffer.h:195-199) the block keeps its weight.  Test: symmetric Cholesky of J Sigma J^T.
  {
    double Lc[81];
    for (int i = 0; i < 81; ++i) Lc[i] = P.d[i];
    for (int c = 0; c < 9; ++c) {
      double d = Lc[c * 9 + c];
      for (int k = 0; k < c; ++k) d -= Lc[c * 9 + k] * Lc[c * 9 + k];
      if (!(d > 0.0)) return;
      const double piv = std::sqrt(d);
      Lc[c * 9 + c] = piv;
      for (int i = c + 1; i < 9; ++i) {
        double a = Lc[i * 9 + c];
        for (int k = 0; k < c; ++k) a -= Lc[i * 9 + k] * Lc[c * 9 + k];
        Lc[i * 9 + c] = a / piv;
      }
    }
  }
  double cov[81];
  inverse9(P.d, cov);
  if (cov_out) std::memcpy(cov_out, cov, sizeof(cov));
  sqrt_spd9(cov, w_sqrt);
}

}  // namespace vco
