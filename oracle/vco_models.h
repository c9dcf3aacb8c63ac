// ORACLE -- TEST INFRASTRUCTURE ONLY (see vco_math.h header). PARITY UNPINNED.
//
// vco_models.h: camera projections (Calibu, un-vendored: restated from the
// published camera_models_crtp.h formulas), the per-corner
// reprojection functor (ceres-cost-functions.h:342-377), the SE3/SO3 local
// parameterisations (local-param-se3.h:10-166) and the robust losses
// (ceres::SoftLOneLoss / CauchyLoss, vicalibrator.h:127, :133).
#pragma once
#include "vco_math.h"

namespace vco {

// Model ids follow the -models strings of vicalib-engine.cc:203-253.
enum Model { kFov = 0, kPoly2 = 1, kPoly3 = 2, kKb4 = 3, kLinear = 4, kRational6 = 5 };
inline int model_num_params(int m) {
  switch (m) { case kFov: return 5; case kPoly2: return 6; case kPoly3: return 7;
    case kKb4: return 8; case kLinear: return 4; case kRational6: return 10; default: return -1; }
}

// calibu::FovCamera::Project (call site ceres-cost-functions.h:369).
template <class T> inline void project_fov(const T* ray, const T* k, T* pix) {
  const T x = ray[0] / ray[2], y = ray[1] / ray[2];
  const T rad = sqrt(x * x + y * y);
  const T w = k[4];
  T fac;
  if (w * w > 1e-5) {
    const T m = 2.0 * tan(w / 2.0);
    if (rad * rad < 1e-5) {
      fac = m / w;
    } else {
      fac = atan(rad * m) / (rad * w);
    }
  } else {
    fac = T(1.0);
  }
  pix[0] = fac * k[0] * x + k[2];
  pix[1] = fac * k[1] * y + k[3];
}
// calibu::Poly2Camera / Poly3Camera::Project.
template <class T> inline void project_poly(const T* ray, const T* k, int nk, T* pix) {
  const T x = ray[0] / ray[2], y = ray[1] / ray[2];
  const T rad = sqrt(x * x + y * y);
  const T r2 = rad * rad;
  const T r4 = r2 * r2;
  T fac = 1.0 + k[4] * r2 + k[5] * r4;
  if (nk == 3) fac = fac + k[6] * r4 * r2;
  pix[0] = fac * k[0] * x + k[2];
  pix[1] = fac * k[1] * y + k[3];
}
// calibu::KannalaBrandtCamera::Project.
template <class T> inline void project_kb4(const T* ray, const T* k, T* pix) {
  const T xy2 = ray[0] * ray[0] + ray[1] * ray[1];
  const T theta = atan2(sqrt(xy2), ray[2]);
  const T psi = atan2(ray[1], ray[0]);
  const T th2 = theta * theta;
  const T th3 = th2 * theta, th5 = th3 * th2, th7 = th5 * th2, th9 = th7 * th2;
  const T r = theta + k[4] * th3 + k[5] * th5 + k[6] * th7 + k[7] * th9;
  pix[0] = k[0] * r * cos(psi) + k[2];
  pix[1] = k[1] * r * sin(psi) + k[3];
}
// calibu::Rational6Camera::Project (camera_models_rational.h; SURVEY 9.1: the OpenCV rational radial factor without tangential
// terms -- the least certain of the reconstructed models): fac = (1 + k1 r^2 + k2 r^4 + k3 r^6) / (1 + k4 r^2 + k5 r^4 + k6 r^6).
template <class T> inline void project_rational6(const T* ray, const T* k, T* pix) {
  const T x = ray[0] / ray[2], y = ray[1] / ray[2];
  const T r2 = x * x + y * y;
  const T r4 = r2 * r2, r6 = r4 * r2;
  const T fac = (1.0 + k[4] * r2 + k[5] * r4 + k[6] * r6) / (1.0 + k[7] * r2 + k[8] * r4 + k[9] * r6);
  pix[0] = fac * k[0] * x + k[2];
  pix[1] = fac * k[1] * y + k[3];
}
// calibu::LinearCamera::Project.
template <class T> inline void project_linear(const T* ray, const T* k, T* pix) {
  pix[0] = k[0] * (ray[0] / ray[2]) + k[2];
  pix[1] = k[1] * (ray[1] / ray[2]) + k[3];
}
template <class T> inline void project(int model, const T* ray, const T* k, T* pix) {
  switch (model) {
    case kFov: project_fov(ray, k, pix); break;
    case kPoly2: project_poly(ray, k, 2, pix); break;
    case kPoly3: project_poly(ray, k, 3, pix); break;
    case kKb4: project_kb4(ray, k, pix); break;
    case kRational6: project_rational6(ray, k, pix); break;
    default: project_linear(ray, k, pix); break;
  }
}

// ImuReprojectionCostFunctor::operator() (ceres-cost-functions.h:350-373):
//   t_kw = t_wk^-1 ; p_c = t_ck * (t_kw * p_w) ; r = Project(p_c, K) - z.
template <class T>
inline void reproj_residual(int model, const T* t_wk /*7*/, const T* r_ck /*4*/, const T* p_ck /*3*/,
                            const T* cam, const double* p_w, const double* z, T* r) {
  T t_kw[7];
  se3_inv(t_wk, t_kw);
  T pw[3] = {T(p_w[0]), T(p_w[1]), T(p_w[2])};
  T pk[3], pc[3], t[3];
  se3_act(t_kw, pw, pk);
  quat_rotate(r_ck, pk, t);
  for (int i = 0; i < 3; ++i) pc[i] = t[i] + p_ck[i];
  T pix[2];
  project(model, pc, cam, pix);
  r[0] = pix[0] - z[0];
  r[1] = pix[1] - z[1];
}

// LocalParamSe3::ComputeJacobian (local-param-se3.h:28-91): 7x6 row-major
// d(T * exp(delta))/d(delta) at delta = 0, global [q(4), t(3)], local [v, w].
inline void local_jac_se3(const double* x, double* J /*7x6*/) {
  for (int i = 0; i < 42; ++i) J[i] = 0.0;
  const double q1 = x[0], q2 = x[1], q3 = x[2], q0 = x[3];
  J[0 * 6 + 3] = 0.5 * q0; J[0 * 6 + 4] = -0.5 * q3; J[0 * 6 + 5] = 0.5 * q2;
  J[1 * 6 + 3] = 0.5 * q3; J[1 * 6 + 4] = 0.5 * q0; J[1 * 6 + 5] = -0.5 * q1;
  J[2 * 6 + 3] = -0.5 * q2; J[2 * 6 + 4] = 0.5 * q1; J[2 * 6 + 5] = 0.5 * q0;
  J[3 * 6 + 3] = -0.5 * q1; J[3 * 6 + 4] = -0.5 * q2; J[3 * 6 + 5] = -0.5 * q3;
  J[4 * 6 + 0] = 1.0 - 2.0 * (q2 * q2 + q3 * q3); J[4 * 6 + 1] = 2.0 * (q1 * q2 - q0 * q3); J[4 * 6 + 2] = 2.0 * (q1 * q3 + q0 * q2);
  J[5 * 6 + 0] = 2.0 * (q1 * q2 + q0 * q3); J[5 * 6 + 1] = 1.0 - 2.0 * (q1 * q1 + q3 * q3); J[5 * 6 + 2] = 2.0 * (q2 * q3 - q0 * q1);
  J[6 * 6 + 0] = 2.0 * (q1 * q3 - q0 * q2); J[6 * 6 + 1] = 2.0 * (q2 * q3 + q0 * q1); J[6 * 6 + 2] = 1.0 - 2.0 * (q1 * q1 + q2 * q2);
}
// LocalParamSo3::ComputeJacobian (local-param-se3.h:121-157): 4x3 row-major.
inline void local_jac_so3(const double* x, double* J /*4x3*/) {
  const double q1 = x[0], q2 = x[1], q3 = x[2], q0 = x[3];
  J[0] = 0.5 * q0; J[1] = -0.5 * q3; J[2] = 0.5 * q2;
  J[3] = 0.5 * q3; J[4] = 0.5 * q0; J[5] = -0.5 * q1;
  J[6] = -0.5 * q2; J[7] = 0.5 * q1; J[8] = 0.5 * q0;
  J[9] = -0.5 * q1; J[10] = -0.5 * q2; J[11] = -0.5 * q3;
}
// LocalParamSe3::Plus / LocalParamSo3::Plus (local-param-se3.h:14-26, :107-119).
inline void plus_se3(const double* x, const double* d, double* o) {
  double e[7];
  se3_exp(d, e);
  se3_mul(x, e, o);
}
inline void plus_so3(const double* x, const double* d, double* o) {
  double e[4];
  so3_exp(d, e);
  so3_mul(x, e, o);
}

// ceres::SoftLOneLoss(a) / ceres::CauchyLoss(a): rho[0..2] = rho, rho', rho''.
inline void loss_soft_l1(double a, double s, double* rho) {
  const double b = a * a, c = 1.0 / b;
  const double sum = 1.0 + s * c, tmp = std::sqrt(sum);
  rho[0] = 2.0 * b * (tmp - 1.0);
  rho[1] = std::fmax(2.2250738585072014e-308, 1.0 / tmp);
  rho[2] = -(c * rho[1]) / (2.0 * sum);
}
inline void loss_cauchy(double a, double s, double* rho) {
  const double b = a * a, c = 1.0 / b;
  const double sum = 1.0 + s * c, inv = 1.0 / sum;
  rho[0] = b * std::log(sum);
  rho[1] = std::fmax(2.2250738585072014e-308, inv);
  rho[2] = -c * (inv * inv);
}

}  // namespace vco
