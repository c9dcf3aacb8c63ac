// ORACLE -- TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
//
// PARITY UNPINNED: the reference (arpg/vicalib) cannot be built here (Ceres,
// Calibu, Sophus, Eigen are absent and un-vendored, CMakeLists.txt:44-56) and
// its tests hold no golden vectors.  This file restates
// the third-party arithmetic the reference's functors call, from the published
// algorithms of those libraries, and is checked against
// mpmath known-answer vectors generated in tests/golden/.
//
// vco_math.h: forward-mode dual numbers (the role ceres::Jet plays under
// ceres::AutoDiffCostFunction, vicalibrator.h:413-453, :620-621) and the
// quaternion / SO3 / SE3 operations the reference takes from Sophus + Eigen
// (call sites: ceres-cost-functions.h:42-48, :98-102, :361-367, :468;
// local-param-se3.h:20-24, :113-117).
#pragma once
#include <cmath>

namespace vco {

// ---------------------------------------------------------------------------
// Dual<N>: value + N partial derivatives.  Comparisons look at the value only
// (same convention as ceres::Jet), so branches in templated code follow the
// scalar part exactly as they would in the reference under autodiff.
// ---------------------------------------------------------------------------
template <int N>
struct Dual {
  double a;
  double v[N];
  Dual() : a(0.0) { for (int i = 0; i < N; ++i) v[i] = 0.0; }
  Dual(double s) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0.0; }  // NOLINT
  static Dual Var(double s, int k) { Dual d(s); d.v[k] = 1.0; return d; }
};

template <int N> inline Dual<N> operator+(const Dual<N>& x, const Dual<N>& y) {
  Dual<N> r; r.a = x.a + y.a; for (int i = 0; i < N; ++i) r.v[i] = x.v[i] + y.v[i]; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& x, const Dual<N>& y) {
  Dual<N> r; r.a = x.a - y.a; for (int i = 0; i < N; ++i) r.v[i] = x.v[i] - y.v[i]; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& x) {
  Dual<N> r; r.a = -x.a; for (int i = 0; i < N; ++i) r.v[i] = -x.v[i]; return r; }
template <int N> inline Dual<N> operator*(const Dual<N>& x, const Dual<N>& y) {
  Dual<N> r; r.a = x.a * y.a; for (int i = 0; i < N; ++i) r.v[i] = x.a * y.v[i] + x.v[i] * y.a; return r; }
template <int N> inline Dual<N> operator/(const Dual<N>& x, const Dual<N>& y) {
  // same arrangement as ceres::Jet: (x.v - (x.a/y.a) y.v) / y.a
  Dual<N> r; const double inv = 1.0 / y.a; const double q = x.a * inv; r.a = q;
  for (int i = 0; i < N; ++i) r.v[i] = (x.v[i] - q * y.v[i]) * inv; return r; }
template <int N> inline Dual<N> operator+(const Dual<N>& x, double s) { Dual<N> r = x; r.a += s; return r; }
template <int N> inline Dual<N> operator+(double s, const Dual<N>& x) { Dual<N> r = x; r.a += s; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& x, double s) { Dual<N> r = x; r.a -= s; return r; }
template <int N> inline Dual<N> operator-(double s, const Dual<N>& x) { Dual<N> r = -x; r.a += s; return r; }
template <int N> inline Dual<N> operator*(const Dual<N>& x, double s) {
  Dual<N> r; r.a = x.a * s; for (int i = 0; i < N; ++i) r.v[i] = x.v[i] * s; return r; }
template <int N> inline Dual<N> operator*(double s, const Dual<N>& x) { return x * s; }
template <int N> inline Dual<N> operator/(const Dual<N>& x, double s) { return x * (1.0 / s); }
template <int N> inline Dual<N> operator/(double s, const Dual<N>& y) {
  Dual<N> r; const double inv = 1.0 / y.a; r.a = s * inv; const double c = -s * inv * inv;
  for (int i = 0; i < N; ++i) r.v[i] = c * y.v[i]; return r; }
template <int N> inline Dual<N>& operator+=(Dual<N>& x, const Dual<N>& y) { x = x + y; return x; }
template <int N> inline Dual<N>& operator-=(Dual<N>& x, const Dual<N>& y) { x = x - y; return x; }
template <int N> inline Dual<N>& operator*=(Dual<N>& x, const Dual<N>& y) { x = x * y; return x; }

#define VCO_CMP(op) \
  template <int N> inline bool operator op(const Dual<N>& x, const Dual<N>& y) { return x.a op y.a; } \
  template <int N> inline bool operator op(const Dual<N>& x, double y) { return x.a op y; } \
  template <int N> inline bool operator op(double x, const Dual<N>& y) { return x op y.a; }
VCO_CMP(<) VCO_CMP(>) VCO_CMP(<=) VCO_CMP(>=) VCO_CMP(==) VCO_CMP(!=)
#undef VCO_CMP

template <int N> inline Dual<N> chain(const Dual<N>& x, double f, double df) {
  Dual<N> r; r.a = f; for (int i = 0; i < N; ++i) r.v[i] = df * x.v[i]; return r; }
template <int N> inline Dual<N> sqrt(const Dual<N>& x) { const double s = std::sqrt(x.a); return chain(x, s, 1.0 / (2.0 * s)); }
template <int N> inline Dual<N> sin(const Dual<N>& x) { return chain(x, std::sin(x.a), std::cos(x.a)); }
template <int N> inline Dual<N> cos(const Dual<N>& x) { return chain(x, std::cos(x.a), -std::sin(x.a)); }
template <int N> inline Dual<N> tan(const Dual<N>& x) { const double t = std::tan(x.a); return chain(x, t, 1.0 + t * t); }
template <int N> inline Dual<N> atan(const Dual<N>& x) { return chain(x, std::atan(x.a), 1.0 / (1.0 + x.a * x.a)); }
template <int N> inline Dual<N> asin(const Dual<N>& x) { return chain(x, std::asin(x.a), 1.0 / std::sqrt(1.0 - x.a * x.a)); }
template <int N> inline Dual<N> log(const Dual<N>& x) { return chain(x, std::log(x.a), 1.0 / x.a); }
template <int N> inline Dual<N> atan2(const Dual<N>& g, const Dual<N>& f) {
  // d atan2(g,f) = (f dg - g df) / (f^2 + g^2)
  Dual<N> r; r.a = std::atan2(g.a, f.a); const double t = 1.0 / (f.a * f.a + g.a * g.a);
  for (int i = 0; i < N; ++i) r.v[i] = t * (f.a * g.v[i] - g.a * f.v[i]); return r; }
template <int N> inline Dual<N> abs(const Dual<N>& x) { return x.a < 0.0 ? -x : x; }

using std::sqrt; using std::sin; using std::cos; using std::tan; using std::atan; using std::atan2;
using std::asin; using std::abs; using std::log;

inline double scalar_of(double x) { return x; }
template <int N> inline double scalar_of(const Dual<N>& x) { return x.a; }

// Sophus' SophusConstants<double>::epsilon() (pre-1.0 API).
static const double kSophusEps = 1e-10;

// ---------------------------------------------------------------------------
// Quaternions are stored [x, y, z, w] (Eigen coeffs() order = Sophus SO3 data
// order, SURVEY 8a-a1 [CHECKED]); SE3 is [q(4), t(3)].
// ---------------------------------------------------------------------------
// Hamilton product, Eigen::Quaternion operator*.
template <class T> inline void quat_mul(const T* a, const T* b, T* o) {
  const T x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  const T y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  const T z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  const T w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
template <class T> inline void quat_conj(const T* a, T* o) { o[0] = -a[0]; o[1] = -a[1]; o[2] = -a[2]; o[3] = a[3]; }
template <class T> inline void quat_normalize(T* q) {
  const T n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; ++i) q[i] = q[i] / n;
}
// Eigen QuaternionBase::_transformVector: v + w*(2 u x v) + u x (2 u x v).
// This is what `SO3 * point` evaluates (ceres-cost-functions.h:101, :367).
template <class T> inline void quat_rotate(const T* q, const T* v, T* o) {
  const T ux = 2.0 * (q[1] * v[2] - q[2] * v[1]);
  const T uy = 2.0 * (q[2] * v[0] - q[0] * v[2]);
  const T uz = 2.0 * (q[0] * v[1] - q[1] * v[0]);
  const T rx = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
  const T ry = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
  const T rz = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
  o[0] = rx; o[1] = ry; o[2] = rz;
}
// Eigen toRotationMatrix (row-major 3x3) = SO3::matrix() = SO3::Adj()
// (ceres-cost-functions.h:98).
template <class T> inline void quat_to_matrix(const T* q, T* R) {
  const T tx = 2.0 * q[0], ty = 2.0 * q[1], tz = 2.0 * q[2];
  const T twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const T txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const T tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1.0 - (txx + tyy);
}
template <class T> inline void mat3_vec(const T* R, const T* v, T* o) {
  const T a = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
  const T b = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
  const T c = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  o[0] = a; o[1] = b; o[2] = c;
}

// Sophus SO3::exp (expAndTheta), small-angle branch below epsilon.
template <class T> inline void so3_exp(const T* w, T* q) {
  const T th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const T th = sqrt(th2);
  T imag, real;
  if (th < kSophusEps) {
    const T th4 = th2 * th2;
    imag = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4;
    real = 1.0 - (1.0 / 8.0) * th2 + (1.0 / 384.0) * th4;
  } else {
    const T half = 0.5 * th;
    imag = sin(half) / th;
    real = cos(half);
  }
  q[0] = imag * w[0]; q[1] = imag * w[1]; q[2] = imag * w[2]; q[3] = real;
}
// Sophus SO3::logAndTheta.
template <class T> inline void so3_log(const T* q, T* w, T* theta_out = nullptr) {
  const T n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
  const T n = sqrt(n2);
  const T qw = q[3];
  T c;
  if (n < kSophusEps) {
    c = 2.0 / qw - 2.0 * n2 / (qw * qw * qw);
  } else if (abs(qw) < kSophusEps) {
    c = (qw > 0.0) ? T(M_PI) / n : T(-M_PI) / n;
  } else {
    c = 2.0 * atan(n / qw) / n;
  }
  if (theta_out) *theta_out = c * n;
  w[0] = c * q[0]; w[1] = c * q[1]; w[2] = c * q[2];
}
// hat(w)^2 applied pieces: Omega = [w]x.
template <class T> inline void hat(const T* w, T* O) {
  O[0] = T(0.0); O[1] = -w[2]; O[2] = w[1];
  O[3] = w[2]; O[4] = T(0.0); O[5] = -w[0];
  O[6] = -w[1]; O[7] = w[0]; O[8] = T(0.0);
}
template <class T> inline void mat3_mul(const T* A, const T* B, T* C) {
  T t[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
    t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  for (int i = 0; i < 9; ++i) C[i] = t[i];
}
// Sophus SE3::exp([upsilon, omega]) -> [q, t], V = so3.matrix() below epsilon.
template <class T> inline void se3_exp(const T* d, T* X) {
  const T* w = d + 3;
  so3_exp(w, X);
  const T th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const T th = sqrt(th2);
  T V[9];
  if (th < kSophusEps) {
    quat_to_matrix(X, V);
  } else {
    T O[9], O2[9];
    hat(w, O); mat3_mul(O, O, O2);
    const T a = (1.0 - cos(th)) / th2;
    const T b = (th - sin(th)) / (th2 * th);
    for (int i = 0; i < 9; ++i) V[i] = a * O[i] + b * O2[i];
    V[0] = V[0] + 1.0; V[4] = V[4] + 1.0; V[8] = V[8] + 1.0;
  }
  mat3_vec(V, d, X + 4);
}
// Sophus SE3::log -> [upsilon, omega].
template <class T> inline void se3_log(const T* X, T* d) {
  T th;
  so3_log(X, d + 3, &th);
  T O[9], O2[9], Vi[9];
  hat(d + 3, O); mat3_mul(O, O, O2);
  T c;
  if (abs(th) < kSophusEps) {
    c = T(1.0 / 12.0);
  } else {
    c = (1.0 - th / (2.0 * tan(th / 2.0))) / (th * th);
  }
  for (int i = 0; i < 9; ++i) Vi[i] = -0.5 * O[i] + c * O2[i];
  Vi[0] = Vi[0] + 1.0; Vi[4] = Vi[4] + 1.0; Vi[8] = Vi[8] + 1.0;
  mat3_vec(Vi, X + 4, d);
}
// Sophus SO3 product renormalises (pre-1.0 operator*=), SE3 product uses it.
template <class T> inline void so3_mul(const T* a, const T* b, T* o) { quat_mul(a, b, o); quat_normalize(o); }
template <class T> inline void se3_mul(const T* A, const T* B, T* o) {
  T q[4], t[3];
  so3_mul(A, B, q);
  quat_rotate(A, B + 4, t);
  for (int i = 0; i < 4; ++i) o[i] = q[i];
  for (int i = 0; i < 3; ++i) o[4 + i] = A[4 + i] + t[i];
}
template <class T> inline void se3_inv(const T* A, T* o) {
  T q[4], nt[3], t[3];
  quat_conj(A, q);
  nt[0] = A[4] * -1.0; nt[1] = A[5] * -1.0; nt[2] = A[6] * -1.0;
  quat_rotate(q, nt, t);
  for (int i = 0; i < 4; ++i) o[i] = q[i];
  for (int i = 0; i < 3; ++i) o[4 + i] = t[i];
}
template <class T> inline void se3_act(const T* A, const T* p, T* o) {
  T t[3];
  quat_rotate(A, p, t);
  for (int i = 0; i < 3; ++i) o[i] = t[i] + A[4 + i];
}

}  // namespace vco
