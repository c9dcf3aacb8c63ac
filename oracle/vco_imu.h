// ORACLE -- TEST INFRASTRUCTURE ONLY (see vco_math.h header). PARITY UNPINNED.
//
// vco_imu.h: CPU restatement of the inertial side of the hot path:
//   * InterpolationBufferT  (interpolation-buffer.h:50-227)
//   * GetGravityVector      (types.h:94-104), gravity() (types.h:40-42)
//   * IntegratePoseJet / GetPoseDerivativeJet / IntegrateImuJet /
//     IntegrateResidualJet  (ceres-cost-functions.h:38-227)
//   * SwitchedFullImuCostFunction::operator() (ceres-cost-functions.h:402-484)
// All templated on the scalar so the same code runs on double and Dual<35>.
#pragma once
#include <vector>
#include <cstddef>
#include "vco_math.h"

namespace vco {

inline double gravity_magnitude() { return 9.8007; }  // types.h:40-42

template <class T> struct ImuMeas { T w[3]; T a[3]; T time; };

// types.h:94-104
template <class T> inline void gravity_vector(const T* dir, T g, T* out) {
  const T sp = sin(dir[0]), cp = cos(dir[0]), sq = sin(dir[1]), cq = cos(dir[1]);
  out[0] = (cp * sq) * (-1.0 * g);
  out[1] = (-1.0 * sp) * (-1.0 * g);
  out[2] = (cp * cq) * (-1.0 * g);
}

// interpolation-buffer.h:50-227.  Stored samples are doubles; the time offset
// dt may be a Dual so its derivative flows through the two end interpolations
// and the shifted sample times, exactly as in the reference.
struct ImuBuffer {
  std::vector<ImuMeas<double>> e;
  double start_time = -1, end_time = -1, average_dt = 0;

  // AddElement :70-85 (caller guarantees time > end_time, vicalibrator.h:373)
  void add(const double* w, const double* a, double t) {
    const size_t n = e.size();
    double dt = 0;
    if (n > 0) dt = t - e.back().time;
    average_dt = (average_dt * n + dt) / (n + 1);
    ImuMeas<double> m;
    for (int i = 0; i < 3; ++i) { m.w[i] = w[i]; m.a[i] = a[i]; }
    m.time = t;
    e.push_back(m);
    end_time = t;
    start_time = e.front().time;
  }
  // HasElement :122-125
  template <class T> bool has(double time, const T& dt) const {
    return time >= start_time + scalar_of(dt) && time <= end_time + scalar_of(dt);
  }
  template <class T> static ImuMeas<T> shifted(const ImuMeas<double>& m, const T& dt) {
    ImuMeas<T> r;
    for (int i = 0; i < 3; ++i) { r.w[i] = T(m.w[i]); r.a[i] = T(m.a[i]); }
    r.time = T(m.time) + dt;
    return r;
  }
  // Interpolate :136-144 + InterpolateElements :147-156
  template <class T> ImuMeas<T> interp(size_t ia, size_t ib, const T& dt, double time) const {
    const T ta = T(e[ia].time) + dt, tb = T(e[ib].time) + dt, tout = T(time);
    const T f = (tout - ta) / (tb - ta);
    const T omf = T(1.0) - f;
    ImuMeas<T> r;
    for (int i = 0; i < 3; ++i) {
      r.w[i] = T(e[ia].w[i]) * omf + T(e[ib].w[i]) * f;
      r.a[i] = T(e[ia].a[i]) * omf + T(e[ib].a[i]) * f;
    }
    r.time = tout;
    return r;
  }
  // GetElement :160-204.  Index search on the scalar part of dt.
  template <class T> ImuMeas<T> element(double time, const T& dt, size_t* idx) const {
    const double off = scalar_of(dt);
    const size_t n = e.size();
    const double guess = (time - start_time + off) / average_dt;
    size_t g = guess > 0 ? static_cast<size_t>(guess) : 0;  // (negative -> 0: reference casts to size_t, UB)
    if (g > n - 1) g = n - 1;
    if (e[g].time + off > time) {
      if (g == 0) { *idx = 0; return shifted(e.front(), dt); }
      while ((g - 1) > 0 && e[g - 1].time + off > time) --g;
      *idx = g - 1;
      return interp(g - 1, g, dt, time);
    }
    if (g == n - 1) { *idx = g; return shifted(e.back(), dt); }
    while ((g + 1) < n && (e[g + 1].time + off) < time) ++g;
    if (g + 1 >= n) {  // reference reads elements_[n] here (:195-199, out of bounds); clamp instead
      *idx = n - 1;
      return shifted(e.back(), dt);
    }
    *idx = g;
    return interp(g, g + 1, dt, time);
  }
  // GetNext :100-117
  template <class T> bool next(double max_time, const T& dt, size_t* idx, ImuMeas<T>* out) const {
    if (*idx + 1 >= e.size()) { *out = element(max_time, dt, idx); return false; }
    if (T(e[*idx + 1].time) + dt > T(max_time)) { *out = element(max_time, dt, idx); return false; }
    *out = shifted(e[++*idx], dt);
    return true;
  }
  // GetRange :208-226
  template <class T> void range(double t0, double t1, const T& dt, std::vector<ImuMeas<T>>* out) const {
    out->clear();
    if (e.empty()) return;
    size_t idx;
    if (has(t0, dt)) {
      out->push_back(element(t0, dt, &idx));
      ImuMeas<T> m;
      while (next(t1, dt, &idx, &m)) out->push_back(m);
      out->push_back(m);
    }
  }
};

template <class T> struct ImuPose { T t_wp[7]; T v[3]; T w[3]; T time; };

// IntegratePoseJet, ceres-cost-functions.h:39-56 (left-multiplied, NOT renormalised).
template <class T> inline ImuPose<T> integrate_pose(const ImuPose<T>& p, const T* k, const T& dt) {
  T wdt[3] = {k[3] * dt, k[4] * dt, k[5] * dt};
  T rq[4];
  so3_exp(wdt, rq);
  ImuPose<T> y = p;
  for (int i = 0; i < 3; ++i) y.t_wp[4 + i] = p.t_wp[4 + i] + k[i] * dt;
  quat_mul(rq, p.t_wp, y.t_wp);
  for (int i = 0; i < 3; ++i) y.v[i] = p.v[i] + k[6 + i] * dt;
  return y;
}
// GetPoseDerivativeJet, ceres-cost-functions.h:80-105.
template <class T> inline void pose_derivative(const ImuPose<T>& p, const T* g_w, const ImuMeas<T>& z0,
                                               const ImuMeas<T>& z1, const T* bg, const T* ba, const T* sf,
                                               const T& dt, T* k) {
  const T alpha = (z1.time - (z0.time + dt)) / (z1.time - z0.time);
  const T oma = 1.0 - alpha;
  T zg[3], za[3], u[3], R[9], o[3];
  for (int i = 0; i < 3; ++i) {
    zg[i] = z0.w[i] * alpha + z1.w[i] * oma;
    za[i] = z0.a[i] * alpha + z1.a[i] * oma;
  }
  for (int i = 0; i < 3; ++i) k[i] = p.v[i];
  for (int i = 0; i < 3; ++i) u[i] = zg[i] * sf[i] + bg[i];
  quat_to_matrix(p.t_wp, R);           // so3().Adj() * (...)
  mat3_vec(R, u, o);
  for (int i = 0; i < 3; ++i) k[3 + i] = o[i];
  for (int i = 0; i < 3; ++i) u[i] = za[i] * sf[3 + i] + ba[i];
  quat_rotate(p.t_wp, u, o);           // so3() * (...)
  for (int i = 0; i < 3; ++i) k[6 + i] = o[i] - g_w[i];
}
// IntegrateImuJet, ceres-cost-functions.h:139-177 (classical RK4).
template <class T> inline ImuPose<T> integrate_imu(const ImuPose<T>& p, const ImuMeas<T>& z0, const ImuMeas<T>& z1,
                                                   const T* bg, const T* ba, const T* sf, const T* g) {
  if (z1.time == z0.time) return p;
  const T dt = z1.time - z0.time;
  T k1[9], k2[9], k3[9], k4[9], k[9];
  pose_derivative(p, g, z0, z1, bg, ba, sf, T(0.0), k1);
  const ImuPose<T> y1 = integrate_pose(p, k1, dt * 0.5);
  pose_derivative(y1, g, z0, z1, bg, ba, sf, dt / 2.0, k2);
  const ImuPose<T> y2 = integrate_pose(p, k2, dt * 0.5);
  pose_derivative(y2, g, z0, z1, bg, ba, sf, dt / 2.0, k3);
  const ImuPose<T> y3 = integrate_pose(p, k3, dt);
  pose_derivative(y3, g, z0, z1, bg, ba, sf, dt, k4);
  for (int i = 0; i < 9; ++i) k[i] = k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i];
  ImuPose<T> res = integrate_pose(p, k, dt / 6.0);
  for (int i = 0; i < 3; ++i) res.w[i] = k[3 + i];
  res.time = z1.time;
  return res;
}
// IntegrateResidualJet, ceres-cost-functions.h:200-227.
template <class T> inline ImuPose<T> integrate_residual(const ImuPose<T>& start, const std::vector<ImuMeas<T>>& meas,
                                                        const T* bg, const T* ba, const T* sf, const T* g) {
  ImuPose<T> pose = start;
  for (size_t i = 1; i < meas.size(); ++i) pose = integrate_imu(pose, meas[i - 1], meas[i], bg, ba, sf, g);
  return pose;
}

// SwitchedFullImuCostFunction::operator(), ceres-cost-functions.h:402-484.
// Parameter blocks in the order of vicalibrator.h:628-632.
// w_sqrt is 9x9 row-major; r <- (r^T W)^T (:476-477); rotation-only switch
// zeroes rows 0-2 and 6-8 (:479-482); empty range -> zero residual (:452-455).
template <class T>
inline void imu_residual(const ImuBuffer& buf, double t_start, double t_end, const double* w_sqrt,
                         bool rotation_only, const T* tx2, const T* tx1, const T* v2, const T* v1,
                         const T* g2, const T* b, const T* sf, const T* toff, T* r) {
  T g_dir[2] = {g2[0], g2[1]};
  T gvec[3];
  gravity_vector(g_dir, T(gravity_magnitude()), gvec);
  std::vector<ImuMeas<T>> meas;
  buf.range(t_start, t_end, *toff, &meas);
  if (meas.empty()) { for (int i = 0; i < 9; ++i) r[i] = T(0.0); return; }
  ImuPose<T> start;
  for (int i = 0; i < 7; ++i) start.t_wp[i] = tx1[i];
  for (int i = 0; i < 3; ++i) { start.v[i] = v1[i]; start.w[i] = T(0.0); }
  start.time = meas.front().time;
  const ImuPose<T> end = integrate_residual(start, meas, b, b + 3, sf, gvec);
  T inv2[7], rel[7], raw[9];
  se3_inv(tx2, inv2);
  se3_mul(end.t_wp, inv2, rel);
  se3_log(rel, raw);
  for (int i = 0; i < 3; ++i) raw[6 + i] = end.v[i] - v2[i];
  for (int j = 0; j < 9; ++j) {
    T s(0.0);
    for (int i = 0; i < 9; ++i) s = s + raw[i] * w_sqrt[i * 9 + j];
    r[j] = s;
  }
  if (rotation_only) {
    for (int i = 0; i < 3; ++i) { r[i] = T(0.0); r[6 + i] = T(0.0); }
  }
}

}  // namespace vco
