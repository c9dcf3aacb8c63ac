// vc_imu.hpp -- inertial residual of the calibration hot path, for the HIP kernels.
//
// Device-side counterpart of SwitchedFullImuCostFunction::operator() (ceres-cost-functions.h:402-484):
// sample range under the camera<->IMU time offset (InterpolationBufferT::GetRange,
// interpolation-buffer.h:208-226), classical RK4 on (p, q, v) with linearly interpolated gyro/accel
// (IntegrateImuJet :139-177, GetPoseDerivativeJet :80-105, IntegratePoseJet :39-56: left-multiplied,
// NOT renormalised quaternion), residual [log(T_pred T_j^-1); v_pred - v_j] times weight_sqrt (:468-477)
// and the rotation-only switch (:479-482).
//
// The reference differentiates this with ceres::Jet<double,35>.  Here every lane of a wavefront carries
// ONE derivative direction: the code is templated on the scalar and instantiated with D1 = (value, one
// partial); lane d seeds global parameter d of the 35 (vicalibrator.h:620-632 block order).  The value
// part is identical in all lanes, so branches stay wave-uniform.
#pragma once
#include "vc_math.hpp"

namespace vc {

struct D1 { double a, v; };
VC_HD D1 mk(double a, double v = 0.0) { D1 r; r.a = a; r.v = v; return r; }
VC_HD D1 operator+(D1 x, D1 y) { return mk(x.a + y.a, x.v + y.v); }
VC_HD D1 operator-(D1 x, D1 y) { return mk(x.a - y.a, x.v - y.v); }
VC_HD D1 operator-(D1 x) { return mk(-x.a, -x.v); }
VC_HD D1 operator*(D1 x, D1 y) { return mk(x.a * y.a, x.a * y.v + x.v * y.a); }
// (denominators: reciprocal by v_rcp_f64 + Newton on the device -- fast_rcp, vc_math.hpp; a literal divisor keeps the plain quotient,
//  which the compiler folds)
VC_HD D1 operator/(D1 x, D1 y) { const double inv = fast_rcp(y.a), q = x.a * inv; return mk(q, (x.v - q * y.v) * inv); }
VC_HD D1 operator+(D1 x, double s) { return mk(x.a + s, x.v); }
VC_HD D1 operator+(double s, D1 x) { return mk(x.a + s, x.v); }
VC_HD D1 operator-(D1 x, double s) { return mk(x.a - s, x.v); }
VC_HD D1 operator-(double s, D1 x) { return mk(s - x.a, -x.v); }
VC_HD D1 operator*(D1 x, double s) { return mk(x.a * s, x.v * s); }
VC_HD D1 operator*(double s, D1 x) { return mk(x.a * s, x.v * s); }
VC_HD D1 operator/(D1 x, double s) { const double inv = 1.0 / s; return mk(x.a * inv, x.v * inv); }
VC_HD D1 operator/(double s, D1 y) { const double inv = fast_rcp(y.a); return mk(s * inv, -s * inv * inv * y.v); }
VC_HD bool operator<(D1 x, double y) { return x.a < y; }
VC_HD bool operator>(D1 x, double y) { return x.a > y; }
VC_HD D1 sqrt(D1 x) { double s, is; fast_sqrt_rsqrt(x.a, &s, &is); return mk(s, 0.5 * x.v * is); }
VC_HD D1 sin(D1 x) { return mk(::sin(x.a), ::cos(x.a) * x.v); }
VC_HD D1 cos(D1 x) { return mk(::cos(x.a), -::sin(x.a) * x.v); }
VC_HD D1 tan(D1 x) { const double t = ::tan(x.a); return mk(t, (1.0 + t * t) * x.v); }
VC_HD D1 atan(D1 x) { return mk(vc_atan(x.a), x.v * fast_rcp(1.0 + x.a * x.a)); }      // (branch-free arctangent, < 2 ulp: vc_math.hpp)
VC_HD D1 fabs(D1 x) { return x.a < 0.0 ? -x : x; }
VC_HD double val(double x) { return x; }
VC_HD double val(D1 x) { return x.a; }
template <class T> VC_HD T cst(double x);
template <> VC_HD double cst<double>(double x) { return x; }
template <> VC_HD D1 cst<D1>(double x) { return mk(x); }
using ::sqrt; using ::sin; using ::cos; using ::tan; using ::atan; using ::fabs;

// ---- scalar-generic quaternion / SE3 pieces (Sophus pre-1.0 + Eigen semantics) ----------
template <class T> VC_HD void tq_mul(const T* a, const T* b, T* o) {
  const T x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  const T y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  const T z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  const T w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
template <class T> VC_HD void tq_rotate(const T* q, const T* v, T* o) {   // Eigen _transformVector
  const T ux = 2.0 * (q[1] * v[2] - q[2] * v[1]);
  const T uy = 2.0 * (q[2] * v[0] - q[0] * v[2]);
  const T uz = 2.0 * (q[0] * v[1] - q[1] * v[0]);
  const T rx = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
  const T ry = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
  const T rz = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
  o[0] = rx; o[1] = ry; o[2] = rz;
}
template <class T> VC_HD void tq_matvec(const T* q, const T* v, T* o) {   // toRotationMatrix() * v  (SO3::Adj())
  const T tx = 2.0 * q[0], ty = 2.0 * q[1], tz = 2.0 * q[2];
  const T twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const T txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const T tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  const T a = (1.0 - (tyy + tzz)) * v[0] + (txy - twz) * v[1] + (txz + twy) * v[2];
  const T b = (txy + twz) * v[0] + (1.0 - (txx + tzz)) * v[1] + (tyz - twx) * v[2];
  const T c = (txz - twy) * v[0] + (tyz + twx) * v[1] + (1.0 - (txx + tyy)) * v[2];
  o[0] = a; o[1] = b; o[2] = c;
}
// SO3::exp as a unit quaternion [sin(th/2) w/th, cos(th/2)].  Within th/2 <= pi/4 -- the RK4 increments are
// gyro rate x a few milliseconds -- both factors are polynomials in z = th^2/4 (the fdlibm kernel forms, < 1 ulp; Sophus'
// own small-angle series is their leading part): no square root, no division, no sin / cos call, and under dual numbers
// the derivative comes from the same polynomial.  Larger angles (round 6; a sin / cos call before -- ~600 instructions inlined at every
// call site for a branch a gyro stream never takes): the vector is halved until it is in range and the quaternion squared back,
// on the two factors alone -- with q(u) = [i u, r]:  q(2u) = q(u)^2 = [(r i) 2u, r^2 - i^2 |u|^2].
template <class T> VC_HD void tso3_exp_factors(const T& th2, T* imag_out, T* real_out) {
  T t2 = th2;
  int halvings = 0;
  while (val(t2) > 2.4 && halvings < 60) { t2 = t2 * 0.25; ++halvings; }      // (not taken by any lane of a sane stream)
  const T z = 0.25 * t2;
  const T S = -1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 +
              z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
  const T Cc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
               z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
  T imag = 0.5 + 0.5 * (z * S);                  // sin(th/2) / th
  T real = (1.0 - 0.5 * z) + (z * z) * Cc;
  for (; halvings > 0; --halvings) {
    const T i2 = real * imag;
    real = real * real - (imag * imag) * t2;
    imag = i2;
    t2 = t2 * 4.0;
  }
  *imag_out = imag; *real_out = real;
}
template <class T> VC_HD void tso3_exp(const T* w, T* q) {
  const T th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  T imag, real;
  tso3_exp_factors(th2, &imag, &real);
  q[0] = imag * w[0]; q[1] = imag * w[1]; q[2] = imag * w[2]; q[3] = real;
}
// log of the SE3 element [q, t] -> [upsilon, omega]  (Sophus SE3::log / SO3::logAndTheta)
template <class T> VC_HD void tse3_log(const T* X, T* d) {
  const T n2 = X[0] * X[0] + X[1] * X[1] + X[2] * X[2];
  const T n = sqrt(n2);
  const T qw = X[3];
  T c;
  if (n < kSophusEps) c = 2.0 / qw - 2.0 * n2 / (qw * qw * qw);
  else if (fabs(qw) < kSophusEps) c = (qw > 0.0) ? 3.14159265358979323846 / n : -3.14159265358979323846 / n;
  else c = 2.0 * atan(n / qw) / n;
  const T th = c * n;
  const T w0 = c * X[0], w1 = c * X[1], w2 = c * X[2];
  d[3] = w0; d[4] = w1; d[5] = w2;
  // k = (1 - th / (2 tan(th/2))) / th^2.  On the regular branch th/2 = atan(n / qw), so tan(th/2) = n / qw: no tangent to evaluate
  // (the reference's tan() of that very angle returns the same number up to rounding); the |qw| < eps branch keeps the call
  T k;
  if (fabs(th) < kSophusEps) k = cst<T>(1.0 / 12.0);
  else if (!(n < kSophusEps) && !(fabs(qw) < kSophusEps)) k = (1.0 - (th * qw) / (2.0 * n)) / (th * th);
  else k = (1.0 - th / (2.0 * tan(th / 2.0))) / (th * th);
  // V^-1 t = t - 1/2 w x t + k w x (w x t)
  const T* t = X + 4;
  const T c0 = w1 * t[2] - w2 * t[1], c1 = w2 * t[0] - w0 * t[2], c2 = w0 * t[1] - w1 * t[0];
  const T e0 = w1 * c2 - w2 * c1, e1 = w2 * c0 - w0 * c2, e2 = w0 * c1 - w1 * c0;
  d[0] = t[0] - 0.5 * c0 + k * e0;
  d[1] = t[1] - 0.5 * c1 + k * e1;
  d[2] = t[2] - 0.5 * c2 + k * e2;
}

// ---- sample range (interpolation-buffer.h:100-226) ----------------------------------------------------
struct ImuView { const double* t; const double* w; const double* a; int n; double avg_dt; };   // w, a: n x 3; avg_dt: imu_average_dt
// InterpolationBufferT::average_dt_ as AddElement accumulates it (interpolation-buffer.h:70-85): the running mean of the sample
// spacings, the first sample counted with spacing 0.  GetElement's first guess of an index divides by it.
inline double imu_average_dt(const double* t, int n) {
  double avg = 0.0;
  for (int i = 0; i < n; ++i) { const double dt = i ? t[i] - t[i - 1] : 0.0; avg = (avg * (double)i + dt) / (double)(i + 1); }
  return avg;
}
template <class T> struct Meas { T w[3], a[3], time; };

// bracketing interval [i, i+1] of image-clock time `time` under offset `off` (samples shifted by +off): the largest i <= n - 2
// with t[i] + off <= time.  IMU streams are (nearly) uniformly sampled, so the search starts at the interpolated position and
// walks -- two or three dependent loads instead of the ~15 of a bisection over 20 000 samples, which every lane of every IMU
// kernel used to wait for twice; the result is the bisection's.
VC_HD int imu_bracket(const ImuView& b, double time, double off, double* t_at = nullptr) {
  const double t0 = b.t[0], span = b.t[b.n - 1] - t0;
  int g = (span > 0.0) ? (int)((((time - off) - t0) / span) * (double)(b.n - 1)) : 0;
  g = g > b.n - 2 ? b.n - 2 : g;
  g = g < 0 ? 0 : g;                                      // (never below 0; every caller -- imu_range, imu_range_lanes -- returns an empty range for fewer than two samples before it gets here: the reference runs its schedule through with no IMU samples at all, tests/test_robustness_gpu.py)
  // the guess and its neighbours in ONE round of loads (a dependent load is ~1 us on the device): with nearly uniform sampling
  // the answer is among them; the walks below only run for gappy streams.  All four comparisons are taken before any branch, so
  // that the loads cannot be deferred into the branches that use them.
  const int gm = g > 0 ? g - 1 : 0, gpp = g + 2 < b.n ? g + 2 : b.n - 1;
  const double tm = b.t[gm], tg = b.t[g], tp = b.t[g + 1], tpp = b.t[gpp];
  const bool dn = tg + off > time, dn_m = tm + off > time, up_p = tp + off <= time, up_pp = tpp + off <= time;
  const bool dn_stop = g == 0, dn_one = !dn_m || gm == 0;            // downwards: stays at 0 / one step
  const bool up_stop = g >= b.n - 2 || !up_p, up_one = g + 1 >= b.n - 2 || !up_pp;
  int is = dn ? (dn_stop ? 0 : gm) : (up_stop ? g : g + 1);
  double tis = dn ? (dn_stop ? tg : tm) : (up_stop ? tg : tp);
  const bool slow = dn ? !(dn_stop || dn_one) : !(up_stop || up_one);
  if (slow) {
    if (dn) { g = gm; while (g > 0 && b.t[g] + off > time) --g; }
    else { g = g + 2; while (g < b.n - 2 && b.t[g + 1] + off <= time) ++g; }
    is = g; tis = b.t[g];
  }
  if (t_at) *t_at = tis;
  return is;
}
template <class T> VC_HD void imu_interp(const ImuView& b, int i, T off, double time, Meas<T>* m) {
  const T ta = b.t[i] + off, tb = b.t[i + 1] + off;
  const T f = (cst<T>(time) - ta) / (tb - ta);
  const T omf = 1.0 - f;
  for (int k = 0; k < 3; ++k) {
    m->w[k] = b.w[3 * i + k] * omf + b.w[3 * (i + 1) + k] * f;
    m->a[k] = b.a[3 * i + k] * omf + b.a[3 * (i + 1) + k] * f;
  }
  m->time = cst<T>(time);
}
template <class T> VC_HD void imu_shift(const ImuView& b, int i, T off, Meas<T>* m) {
  for (int k = 0; k < 3; ++k) { m->w[k] = cst<T>(b.w[3 * i + k]); m->a[k] = cst<T>(b.a[3 * i + k]); }
  m->time = b.t[i] + off;
}
// Range description: first = element(t0), interior samples k0 .. k1 (inclusive, may be empty), last = element(t1).
struct ImuRange { int valid; int i0, first_end; int k0, k1; int i1, last_end; };   // *_end: 1 = clamp to an end sample
// GetElement :160-204 without its walk.  The reference guesses an index -- (time - start + offset) / average_dt, the offset with that
// sign -- and walks from there to the bracketing pair; where it ends up is the pair [i, i + 1] with t[i] + off <= time < t[i + 1] + off
// (i = imu_bracket), EXCEPT when `time` falls exactly on a shifted sample and the walk came from below: the forward walk stops one
// pair early, at [i - 1, i] with interpolation weight 1.  Same measurement values, but the interval that follows has length zero
// and is skipped (ceres-cost-functions.h:150-152), and with it the time offset's influence through that end -- frames that
// sit exactly on shifted sample times are what synthetic sequences produce.  The clamps to the first / last sample follow the
// reference as well.  last_le: the largest sample index with t + off <= time (-1: none) -- where GetNext :100-117 stops.
VC_HD void imu_element_index(const ImuView& b, double time, double off, int* idx, int* end_clamp, int* last_le) {
  const int n = b.n;
  const double q = (time - b.t[0] + off) / b.avg_dt;
  int g = (q > 0.0) ? (q < (double)(n - 1) ? (int)q : n - 1) : 0;         // (a negative quotient is cast to size_t in the reference: 0 here)
  const double t_guess = b.t[g];                   // (requested with the bracket's loads: nothing below depends on it before they return)
  const double t_first = b.t[0], t_last = b.t[n - 1];
  double t_is;
  const int ib = imu_bracket(b, time, off, &t_is);
  const bool below = t_first + off > time, above = t_last + off <= time;
  const int is = below ? -1 : (above ? n - 1 : ib);     // largest i with t[i] + off <= time
  if (above) t_is = t_last;
  *last_le = is;
  const bool back = t_guess + off > time, on_sample = t_is + off == time;     // (compared before the branches: see imu_bracket)
  if (back) {                                          // the walk goes backwards
    if (g == 0) { *idx = 0; *end_clamp = 1; return; }
    *idx = is > 0 ? is : 0; *end_clamp = 0;            // [is, is + 1]  ([0, 1], extrapolating, when even the first sample is later)
    if (*idx > n - 2) *idx = n - 2;
    return;
  }
  if (g == n - 1) { *idx = n - 1; *end_clamp = 1; return; }      // forwards from the last sample: clamp
  if (on_sample && g < is) { *idx = is - 1; *end_clamp = 0; return; }     // stopped one pair early, weight 1
  if (is == n - 1) { *idx = n - 1; *end_clamp = 1; return; }     // walked off the end (the reference reads past it: clamp)
  *idx = is; *end_clamp = 0;
}
VC_HD ImuRange imu_range(const ImuView& b, double t0, double t1, double off) {
  ImuRange r; r.valid = 0; r.k0 = 0; r.k1 = -1; r.i0 = r.i1 = 0; r.first_end = r.last_end = 0;
  if (b.n < 2) return r;
  if (!(t0 >= b.t[0] + off && t0 <= b.t[b.n - 1] + off)) return r;   // HasElement :122-125
  r.valid = 1;
  int le0, le1;
  imu_element_index(b, t0, off, &r.i0, &r.first_end, &le0);
  imu_element_index(b, t1, off, &r.i1, &r.last_end, &le1);
  // GetNext :100-117: interior samples are the stored samples idx + 1 .. while time + off <= t1 -- up to the last sample not after
  // t1, which the look-up of t1 has just found (no scan over the samples in between)
  r.k0 = r.i0 + 1;
  r.k1 = le1;
  if (r.k1 < r.i0) r.k1 = r.i0;
  return r;
}
template <class T> VC_HD void imu_range_get(const ImuView& b, const ImuRange& r, T off, double t0, double t1, int which, Meas<T>* m) {
  // which: 0 = first, 1 .. n_int = interior, n_int + 1 = last
  const int n_int = r.k1 - r.k0 + 1;
  if (which == 0) { if (r.first_end) imu_shift(b, r.i0, off, m); else imu_interp(b, r.i0, off, t0, m); }
  else if (which <= n_int) imu_shift(b, r.k0 + which - 1, off, m);
  else { if (r.last_end) imu_shift(b, r.i1, off, m); else imu_interp(b, r.i1, off, t1, m); }
}

// The same element without control flow around the loads (what the device kernels call: a branch per element kind costs a round of
// dependent loads per branch): one sample pair [idx, idx + 1] is read whatever the kind, an interpolated element blends it,
// a shifted sample takes the first of the pair.  Bit-identical to imu_range_get.
template <class T> VC_HD void imu_range_get_flat(const ImuView& b, const ImuRange& r, T off, double t0, double t1, int which, Meas<T>* m) {
  const int n_int = r.k1 - r.k0 + 1;
  const bool first = which == 0, last = which > n_int;
  const int idx = first ? r.i0 : (last ? r.i1 : r.k0 + which - 1);
  const bool interp = (first && !r.first_end) || (last && !r.last_end);
  const int idn = idx + 1 < b.n ? idx + 1 : b.n - 1;
  const double tq = first ? t0 : t1;
  const double ta_ = b.t[idx], tb_ = b.t[idn];
  double wa[3], wb[3], aa[3], ab[3];
  for (int k = 0; k < 3; ++k) { wa[k] = b.w[3 * idx + k]; wb[k] = b.w[3 * idn + k]; aa[k] = b.a[3 * idx + k]; ab[k] = b.a[3 * idn + k]; }
  const T ta = ta_ + off, tb = tb_ + off;
  const T den = interp ? tb - ta : cst<T>(1.0);
  const T f = (cst<T>(tq) - ta) / den;
  const T omf = 1.0 - f;
  for (int k = 0; k < 3; ++k) {
    const T wi = wa[k] * omf + wb[k] * f, ai = aa[k] * omf + ab[k] * f;
    m->w[k] = interp ? wi : cst<T>(wa[k]);
    m->a[k] = interp ? ai : cst<T>(aa[k]);
  }
  m->time = interp ? cst<T>(tq) : ta;
}

// ---- RK4 (ceres-cost-functions.h:39-177) --------------------------------------------------------------
template <class T> struct PoseV { T q[4], p[3], v[3]; };
template <class T, class M> VC_HD void imu_integrate_pose(const PoseV<T>& s, const T* k, M dt, PoseV<T>* y) {
  const T wdt[3] = {k[3] * dt, k[4] * dt, k[5] * dt};
  T rq[4];
  tso3_exp(wdt, rq);
  for (int i = 0; i < 3; ++i) { y->p[i] = s.p[i] + k[i] * dt; y->v[i] = s.v[i] + k[6 + i] * dt; }
  tq_mul(rq, s.q, y->q);
}
// The gyro / accelerometer model u = measurement * scale factor + bias (ceres-cost-functions.h:93-100) behind a small accessor.
// The measurements' scalar type M may be plainer than the state's T: along a bias / scale-factor direction the samples, their
// times and the interpolation weights carry no derivative at all (M = double under T = D1), only the time-offset direction moves them.
template <class T> struct ImuParamsPtr {
  const T* b; const T* sf;
  template <class X> VC_HD auto apply(int i, X x) const -> decltype(x * sf[i] + b[i]) { return x * sf[i] + b[i]; }
};
template <class T, class M, class P> VC_HD void imu_pose_derivative(const PoseV<T>& s, const T* g_w, const Meas<M>& z0, const Meas<M>& z1,
                                                                   const P& par, M dt, T* k) {
  const M alpha = (z1.time - (z0.time + dt)) / (z1.time - z0.time);
  const M oma = 1.0 - alpha;
  T u[3], o[3];
  for (int i = 0; i < 3; ++i) k[i] = s.v[i];
  for (int i = 0; i < 3; ++i) u[i] = par.apply(i, z0.w[i] * alpha + z1.w[i] * oma);
  tq_matvec(s.q, u, o);
  for (int i = 0; i < 3; ++i) k[3 + i] = o[i];
  for (int i = 0; i < 3; ++i) u[i] = par.apply(3 + i, z0.a[i] * alpha + z1.a[i] * oma);
  tq_rotate(s.q, u, o);
  for (int i = 0; i < 3; ++i) k[6 + i] = o[i] - g_w[i];
}
template <class T, class M, class P> VC_HD void imu_rk4_step_p(PoseV<T>* s, const Meas<M>& z0, const Meas<M>& z1, const P& par, const T* g_w) {
  if (val(z1.time) == val(z0.time)) return;       // :150-152
  const M dt = z1.time - z0.time;
  // k1 + 2 k2 + 2 k3 + k4 as a running sum in that order (bit-identical to summing at the end): one stage vector and the sum
  // are live instead of all four -- 27 fewer values per lane, 54 under dual numbers, which is what kept k_imu_jac in scratch
  T k[9], ks[9];
  PoseV<T> y;
  imu_pose_derivative(*s, g_w, z0, z1, par, cst<M>(0.0), k);
  for (int i = 0; i < 9; ++i) ks[i] = k[i];
  imu_integrate_pose(*s, k, dt * 0.5, &y);
  imu_pose_derivative(y, g_w, z0, z1, par, dt / 2.0, k);
  for (int i = 0; i < 9; ++i) ks[i] = ks[i] + 2.0 * k[i];
  imu_integrate_pose(*s, k, dt * 0.5, &y);
  imu_pose_derivative(y, g_w, z0, z1, par, dt / 2.0, k);
  for (int i = 0; i < 9; ++i) ks[i] = ks[i] + 2.0 * k[i];
  imu_integrate_pose(*s, k, dt, &y);
  imu_pose_derivative(y, g_w, z0, z1, par, dt, k);
  for (int i = 0; i < 9; ++i) k[i] = ks[i] + k[i];
  imu_integrate_pose(*s, k, dt / 6.0, &y);
  *s = y;
}
template <class T> VC_HD void imu_rk4_step(PoseV<T>* s, const Meas<T>& z0, const Meas<T>& z1, const T* b, const T* sf, const T* g_w) {
  const ImuParamsPtr<T> par = {b, sf};
  imu_rk4_step_p(s, z0, z1, par, g_w);
}
// types.h:94-104
template <class T> VC_HD void imu_gravity(const T* dir, T* out) {
  const double g = 9.8007;
  const T sp = sin(dir[0]), cp = cos(dir[0]), sq = sin(dir[1]), cq = cos(dir[1]);
  out[0] = (cp * sq) * (-g);
  out[1] = (-1.0 * sp) * (-g);
  out[2] = (cp * cq) * (-g);
}

// the gravity vector and its partials along the two direction angles, [g_w 3 | d/d dir0 3 | d/d dir1 3]: formed once per state (one thread
// of k_imu_block) instead of four sin / cos calls in every lane of every IMU block of k_imu_jac
VC_HD void imu_gravity_record(const double* gdir, double* rec9) {
  for (int c = 0; c < 2; ++c) {
    const D1 d[2] = {mk(gdir[0], c == 0 ? 1.0 : 0.0), mk(gdir[1], c == 1 ? 1.0 : 0.0)};
    D1 g[3];
    imu_gravity(d, g);
    for (int i = 0; i < 3; ++i) { rec9[i] = g[i].a; rec9[3 + 3 * c + i] = g[i].v; }
  }
}
// Tail of the residual: the predicted state against frame j -- [log(T_pred T_j^-1); v_pred - v_j] times weight_sqrt, then the
// rotation-only switch (ceres-cost-functions.h:468-482).
template <class T>
VC_HD void imu_residual_tail(const PoseV<T>& s, const double* w_sqrt, int rotation_only, const T* T2, const T* v2, T* r) {
  // rel = T_pred * T2^-1 (SE3 product: SO3 part renormalised), then log
  T qc[4] = {-T2[0], -T2[1], -T2[2], T2[3]};
  T nt[3] = {T2[4] * -1.0, T2[5] * -1.0, T2[6] * -1.0}, ti[3];
  tq_rotate(qc, nt, ti);
  T rel[7], tr[3];
  tq_mul(s.q, qc, rel);
  const T nrm = sqrt(rel[0] * rel[0] + rel[1] * rel[1] + rel[2] * rel[2] + rel[3] * rel[3]);
  for (int i = 0; i < 4; ++i) rel[i] = rel[i] / nrm;
  tq_rotate(s.q, ti, tr);
  for (int i = 0; i < 3; ++i) rel[4 + i] = s.p[i] + tr[i];
  T raw[9];
  tse3_log(rel, raw);
  for (int i = 0; i < 3; ++i) raw[6 + i] = s.v[i] - v2[i];
  for (int j = 0; j < 9; ++j) {
    T acc = cst<T>(0.0);
    for (int i = 0; i < 9; ++i) acc = acc + raw[i] * w_sqrt[i * 9 + j];
    r[j] = acc;
  }
  if (rotation_only) for (int i = 0; i < 3; ++i) { r[i] = cst<T>(0.0); r[6 + i] = cst<T>(0.0); }
}

// The residual. Parameters: T2 = frame j [q t], T1 = frame j-1, v2, v1, gdir(2), b(6), sf(6), toff.
// w_sqrt 9x9 row-major; r <- (r^T W)^T; rotation-only zeroes rows 0-2, 6-8. Empty range -> r = 0.
template <class T>
VC_HD void imu_residual(const ImuView& buf, double t_start, double t_end, const double* w_sqrt, int rotation_only,
                        const T* T2, const T* T1, const T* v2, const T* v1, const T* gdir, const T* b, const T* sf, T toff, T* r) {
  const ImuRange rg = imu_range(buf, t_start, t_end, val(toff));
  if (!rg.valid) { for (int i = 0; i < 9; ++i) r[i] = cst<T>(0.0); return; }
  T gw[3];
  imu_gravity(gdir, gw);
  PoseV<T> s;
  for (int i = 0; i < 4; ++i) s.q[i] = T1[i];
  for (int i = 0; i < 3; ++i) { s.p[i] = T1[4 + i]; s.v[i] = v1[i]; }
  const int n_meas = (rg.k1 - rg.k0 + 1) + 2;
  Meas<T> z0, z1;
  imu_range_get(buf, rg, toff, t_start, t_end, 0, &z0);
  for (int m = 1; m < n_meas; ++m) {
    imu_range_get(buf, rg, toff, t_start, t_end, m, &z1);
    imu_rk4_step(&s, z0, z1, b, sf, gw);
    z0 = z1;
  }
  imu_residual_tail(s, w_sqrt, rotation_only, T2, v2, r);
}

// LocalParamSe3::ComputeJacobian (local-param-se3.h:28-91): 7x6 row-major d(T exp(delta))/d(delta) at 0,
// global [q(4), t(3)], local [upsilon, omega].
VC_HD void local_jac_se3(const double* x, double* J) {
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

// One lane's share of the IMU block in the INTEGRATED form (the reference's own order of operations: RK4 along the block under a
// dual number): residual values r[9] and the partials of all 9 rows with respect to global parameter `dir` (0..34 in the block
// order of vicalibrator.h:628-632; dir < 0: values only).  The kernels run the delta form below; this one is what the delta form
// is checked against (tests/host_harness, tests/test_device_math_cpu.py) and what vc_get_integration_poses' host arithmetic shares.
VC_HD void imu_block_direction(const ImuView& buf, double t_start, double t_end, const double* w_sqrt, int rotation_only,
                               const double* T2, const double* T1, const double* v2, const double* v1, const double* gdir,
                               const double* b, const double* sf, double toff, int dir, double* r, double* dr) {
  D1 P[35];
  for (int i = 0; i < 7; ++i) { P[i] = mk(T2[i]); P[7 + i] = mk(T1[i]); }
  for (int i = 0; i < 3; ++i) { P[14 + i] = mk(v2[i]); P[17 + i] = mk(v1[i]); }
  P[20] = mk(gdir[0]); P[21] = mk(gdir[1]);
  for (int i = 0; i < 6; ++i) { P[22 + i] = mk(b[i]); P[28 + i] = mk(sf[i]); }
  P[34] = mk(toff);
  for (int i = 0; i < 35; ++i) P[i].v = (i == dir) ? 1.0 : 0.0;
  D1 res[9];
  imu_residual<D1>(buf, t_start, t_end, w_sqrt, rotation_only, P, P + 7, P + 14, P + 17, P + 20, P + 22, P + 28, P[34], res);
  for (int i = 0; i < 9; ++i) { r[i] = res[i].a; dr[i] = res[i].v; }
}

// ---- delta form of the block: what the device's Jacobian sweep runs --------------------------------------------------
// The RK4 map of one sample interval commutes with the state it starts from.  With (dq, dp, dv) the step taken from the
// identity state without gravity,
//     q+ = q dq,     p+ = p + v dt - g dt^2 / 2 + R(q) dp,     v+ = v - g dt + R(q) dv :
// the left-multiplied exponential of the world-frame rate is exp(R(q) u h) q = q exp(u h), Runge-Kutta schemes are
// affine-equivariant, and the contributions of v and g are polynomials of degree <= 2 in time, which RK4 integrates exactly.
// The deltas depend on the samples, biases and scale factors and -- the two partial intervals at the ends of a block -- on the
// time offset, not on any pose: they are formed for all intervals in parallel and composed by a scan (k_imu_block, in
// registers), and a block then costs one application of its delta to the start state (k_imu_jac) instead of a dependent RK4
// chain under dual numbers.  Equal to imu_residual up to
// rounding (tests/test_device_math_cpu.py compares the two and the oracle).
constexpr int kDeltaCols = 14;                    // value | d/db (6) | d/dsf (6) | d/dtoff

// Deltas accumulate without reference to a pose: appending the interval (dq, dp, dv; dt) to the block's running delta (Q, P, V; T),
//     P <- P + V dt + R(Q) dp,   V <- V + R(Q) dv,   Q <- Q dq,   T <- T + dt,
// and the block's end state from its start state (q, p, v) is  q Q,  p + v T - g T^2 / 2 + R(q) P,  v - g T + R(q) V.
template <class T> struct DeltaAcc { T q[4], p[3], v[3], t; };
template <class T> VC_HD void imu_delta_append(DeltaAcc<T>* A, const PoseV<T>& d, T dt) {
  T rp[3], rv[3], q[4];
  tq_rotate(A->q, d.p, rp);
  tq_rotate(A->q, d.v, rv);
  tq_mul(A->q, d.q, q);
  for (int i = 0; i < 3; ++i) {
    A->p[i] = (A->p[i] + A->v[i] * dt) + rp[i];
    A->v[i] = A->v[i] + rv[i];
  }
  for (int i = 0; i < 4; ++i) A->q[i] = q[i];
  A->t = A->t + dt;
}
// a <- a followed by x (appending is associative: the block's delta is the ordered product of its intervals' deltas, in any bracketing)
template <class T> VC_HD void imu_delta_then(DeltaAcc<T>* a, const DeltaAcc<T>& x) {
  PoseV<T> d;
  for (int k = 0; k < 4; ++k) d.q[k] = x.q[k];
  for (int k = 0; k < 3; ++k) { d.p[k] = x.p[k]; d.v[k] = x.v[k]; }
  imu_delta_append(a, d, x.t);
}
template <class T> VC_HD void imu_delta_identity(DeltaAcc<T>* a) {
  for (int k = 0; k < 3; ++k) { a->q[k] = cst<T>(0.0); a->p[k] = cst<T>(0.0); a->v[k] = cst<T>(0.0); }
  a->q[3] = cst<T>(1.0); a->t = cst<T>(0.0);
}
constexpr int kBlockDeltaStride = kDeltaCols * 11;     // one block: [column][Q 4, P 3, V 3, T]

// ---- the block's delta record as k_imu_block forms it ------------------------------------------------------------------------
// One interval: m = 1 .. n_int runs from range element m - 1 to element m (the first and the last are the partial intervals at the
// frame times, those in between stored sample pairs).  Its RK4 step from the identity state without gravity, written out
// (the stage states of imu_rk4_step_p from s = identity: the products with the identity quaternion are exact):
//     w1 = ug(t0)                                 q1 = exp(w1 h/2)
//     w2 = R(q1) ug(tm),  a2 = R(q1) ua(tm)       q2 = exp(w2 h/2)
//     w3 = R(q2) ug(tm),  a3 = R(q2) ua(tm)       q3 = exp(w3 h)
//     w4 = R(q3) ug(t1),  a4 = R(q3) ua(t1)
//     dq = exp((w1 + 2 w2 + 2 w3 + w4) h/6),  dv = (ua(t0) + 2 a2 + 2 a3 + a4) h/6,  dp = (2 ua(t0) h/2 + 2 a2 h/2 + a3 h) h/6
// with ug = gyro sample * scale factor + bias, ua likewise (ceres-cost-functions.h:93-100), the samples interpolated at t0, the
// midpoint and t1 (weights 1, 1/2, 0: the interpolation weight of :89 does not depend on anything that is optimised).
//
// PARTIALS IN CLOSED FORM (round 6; rounds 2-5 pushed a dual number through every operation above -- exp's polynomial, the
// quaternion -> matrix products, the sums: ~2.1 k fp64 instructions per interval and direction).  One direction of the parameter space
// -- gyro bias / scale factor `gsel` (0..2 / 3..5), or (gsel >= 6 with off.v = 1) the time offset, which moves the interpolated end
// samples and the interval's length -- enters through the tangents of the inputs (d ug(t0), d ug(t1), d ua(t0), d ua(t1), d h) and is
// carried as ROTATION TANGENTS, not as quaternion partials: a stage rotation moves as R -> exp(dth) R with
//     dth = Jl(phi) d phi,    Jl(phi) = I + A [phi]x + B [phi]x^2,   A = (1 - cos th) / th^2,  B = (th - sin th) / th^3
// (the left Jacobian of SO3; A and B fall out of exp's own two factors, so3_exp_jl), and a rotated vector as
//     d (R u) = dth x (R u) + R du .
// Per stage: one 3 x 3 matrix from the quaternion (shared by the gyro rate, the accelerometer input, both tangents and the
// accelerometer columns), two cross products, one Jl product.  The ACCELEROMETER parameters need no tangent at all: dq does not
// depend on them and dv, dp are linear in ua, so their partials are sums of the stage rotations' columns (ap, av).
// The same derivative as the dual numbers' up to rounding (tests/test_device_math_cpu.py: against the integrated dual form at
// 1e-10 and the oracle's Dual<35> at 1e-8).
VC_HD void so3_exp_jl(const double* w, double* q, double* A, double* B) {
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double imag, real;
  tso3_exp_factors(th2, &imag, &real);
  q[0] = imag * w[0]; q[1] = imag * w[1]; q[2] = imag * w[2]; q[3] = real;
  // 1 - cos th = 2 sin^2(th/2);  sin th = 2 sin(th/2) cos(th/2).  The difference 1 - 2 imag real cancels to ~1e-16 / th^2 relative,
  // against a term of relative size th^2 / 6 in Jl: 1e-16 absolute; below th^2 = 1e-6 the series' first two terms are exact to rounding
  const bool tiny = th2 < 1e-6;
  *A = 2.0 * imag * imag;
  *B = tiny ? 1.0 / 6.0 - th2 * (1.0 / 120.0) : (1.0 - 2.0 * imag * real) * fast_rcp(tiny ? 1.0 : th2);
}
VC_HD void cross3(const double* a, const double* b, double* o) {
  const double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
VC_HD void so3_jl_apply(const double* w, double A, double B, const double* d, double* o) {
  double c1[3], c2[3];
  cross3(w, d, c1);
  cross3(w, c1, c2);
  for (int i = 0; i < 3; ++i) o[i] = d[i] + A * c1[i] + B * c2[i];
}
VC_HD void mat3_vec(const double* R, const double* v, double* o) {
  const double a = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
  const double b = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
  const double c = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  o[0] = a; o[1] = b; o[2] = c;
}
// An interval's (or a run of intervals') delta with one parameter direction beside it: the values (Q, P, V, T), the direction's
// tangents -- rotation as exp(dth) Q, the rest plain partials -- and the partials along the group's accelerometer parameter.
struct IntervalDeltaT { DeltaAcc<double> d; double dth[3], dp[3], dv[3], dt; double ap[3], av[3]; };
constexpr int kDtDoubles = 27;                     // as doubles: values 11 | tangent 10 | ap 3 | av 3
VC_HD void imu_delta_t_identity(IntervalDeltaT* X) {
  imu_delta_identity(&X->d);
  for (int i = 0; i < 3; ++i) { X->dth[i] = 0.0; X->dp[i] = 0.0; X->dv[i] = 0.0; X->ap[i] = 0.0; X->av[i] = 0.0; }
  X->dt = 0.0;
}
VC_HD void imu_interval_delta_t(const ImuView& buf, const ImuRange& rg, D1 off, double t_start, double t_end, int m, int n_int,
                                const double* b, const double* sf, int gsel, IntervalDeltaT* X) {
  imu_delta_t_identity(X);
  const int kk = gsel < 3 ? gsel : (gsel < 6 ? gsel - 3 : 0);
  const bool seeded = gsel < 6, scale = gsel >= 3;
  // the model inputs at t0 and t1 with their tangents, formed as soon as the samples are there (the samples themselves are not kept).
  // The midpoint's inputs are the mean of the two (the model is affine in the sample).
  double u0[3], u1[3], a0[3], a1[3], du0[3], du1[3], da0[3], da1[3], h, dh;
  double c0, c1;                                   // component kk of the accelerometer samples at t0, t1 (scale factor) or 1 (bias)
  {
    Meas<D1> z0, z1;
    for (int k = 0; k < 3; ++k) { z0.w[k] = z0.a[k] = z1.w[k] = z1.a[k] = mk(0.0); }
    z0.time = z1.time = mk(0.0);
    if (m <= n_int) { imu_range_get_flat(buf, rg, off, t_start, t_end, m - 1, &z0); imu_range_get_flat(buf, rg, off, t_start, t_end, m, &z1); }
    if (z1.time.a == z0.time.a) return;            // a zero-length interval is skipped (:150-152), and so is m > n_int
    h = z1.time.a - z0.time.a; dh = z1.time.v - z0.time.v;
    for (int i = 0; i < 3; ++i) {
      const double s0 = (seeded && i == kk) ? (scale ? z0.w[i].a : 1.0) : 0.0, s1 = (seeded && i == kk) ? (scale ? z1.w[i].a : 1.0) : 0.0;
      u0[i] = z0.w[i].a * sf[i] + b[i]; du0[i] = z0.w[i].v * sf[i] + s0;
      u1[i] = z1.w[i].a * sf[i] + b[i]; du1[i] = z1.w[i].v * sf[i] + s1;
      a0[i] = z0.a[i].a * sf[3 + i] + b[3 + i]; da0[i] = z0.a[i].v * sf[3 + i];
      a1[i] = z1.a[i].a * sf[3 + i] + b[3 + i]; da1[i] = z1.a[i].v * sf[3 + i];
    }
    // (selected, not indexed: a run-time index would put the arrays into scratch memory on the device)
    c0 = scale ? (kk == 0 ? z0.a[0].a : kk == 1 ? z0.a[1].a : z0.a[2].a) : 1.0;
    c1 = scale ? (kk == 0 ? z1.a[0].a : kk == 1 ? z1.a[1].a : z1.a[2].a) : 1.0;
  }
  const double hh = h * 0.5, h6 = h / 6.0, dhh = dh * 0.5, dh6 = dh / 6.0;
  const double cm = 0.5 * c0 + 0.5 * c1;
  // running sums k1 + 2 k2 + 2 k3 + k4 in that order (they start as k1) with their tangents; sp = ua(t0) + a2 + a3 (dp = sp h^2 / 6);
  // cv, cp: the same sums for d ua / d (accelerometer parameter kk) = c(t) e_kk -- column kk of the stage rotations, values only
  double sw[3], sv[3], sp[3], dsw[3], dsv[3], dsp[3], cv[3], cp[3], phi[3], dphi[3];
  for (int i = 0; i < 3; ++i) {
    sw[i] = u0[i]; dsw[i] = du0[i]; sv[i] = a0[i]; dsv[i] = da0[i]; sp[i] = a0[i]; dsp[i] = da0[i];
    cv[i] = (i == kk) ? c0 : 0.0; cp[i] = cv[i];
    phi[i] = u0[i] * hh; dphi[i] = du0[i] * hh + u0[i] * dhh;
  }
  // stages 2, 3, 4 (statically unrolled: `stage` folds)
  for (int stage = 2; stage <= 4; ++stage) {
    double q[4], A, B, R[9], dth[3], ug[3], ua[3], dug[3], dua[3], wk[3], ak[3], dwk[3], dak[3], t1[3], t2[3];
    so3_exp_jl(phi, q, &A, &B);
    quat_to_R(q, R);
    so3_jl_apply(phi, A, B, dphi, dth);
    for (int i = 0; i < 3; ++i) {
      if (stage < 4) { ug[i] = u0[i] * 0.5 + u1[i] * 0.5; ua[i] = a0[i] * 0.5 + a1[i] * 0.5; dug[i] = du0[i] * 0.5 + du1[i] * 0.5; dua[i] = da0[i] * 0.5 + da1[i] * 0.5; }
      else { ug[i] = u1[i]; ua[i] = a1[i]; dug[i] = du1[i]; dua[i] = da1[i]; }
    }
    mat3_vec(R, ug, wk); mat3_vec(R, ua, ak);
    cross3(dth, wk, t1); mat3_vec(R, dug, t2);
    for (int i = 0; i < 3; ++i) dwk[i] = t1[i] + t2[i];
    cross3(dth, ak, t1); mat3_vec(R, dua, t2);
    for (int i = 0; i < 3; ++i) dak[i] = t1[i] + t2[i];
    const double r[3] = {kk == 0 ? R[0] : kk == 1 ? R[1] : R[2], kk == 0 ? R[3] : kk == 1 ? R[4] : R[5], kk == 0 ? R[6] : kk == 1 ? R[7] : R[8]};
    const double wgt = stage < 4 ? 2.0 : 1.0, cc = stage < 4 ? cm : c1;
    for (int i = 0; i < 3; ++i) {
      sw[i] = sw[i] + wgt * wk[i]; dsw[i] = dsw[i] + wgt * dwk[i];
      sv[i] = sv[i] + wgt * ak[i]; dsv[i] = dsv[i] + wgt * dak[i];
      cv[i] = cv[i] + wgt * (cc * r[i]);
      if (stage < 4) { sp[i] = sp[i] + ak[i]; dsp[i] = dsp[i] + dak[i]; cp[i] = cp[i] + cc * r[i]; }
      const double hs = stage == 2 ? hh : h, dhs = stage == 2 ? dhh : dh;
      if (stage < 4) { phi[i] = wk[i] * hs; dphi[i] = dwk[i] * hs + wk[i] * dhs; }
    }
  }
  {
    double A, B;
    for (int i = 0; i < 3; ++i) { phi[i] = sw[i] * h6; dphi[i] = dsw[i] * h6 + sw[i] * dh6; }
    so3_exp_jl(phi, X->d.q, &A, &B);
    so3_jl_apply(phi, A, B, dphi, X->dth);
    const double hh6 = h * h6, dhh6 = 2.0 * h * dh6;          // d (h^2 / 6) = h dh / 3
    for (int i = 0; i < 3; ++i) {
      X->d.v[i] = sv[i] * h6; X->dv[i] = dsv[i] * h6 + sv[i] * dh6;
      X->d.p[i] = sp[i] * hh6; X->dp[i] = dsp[i] * hh6 + sp[i] * dhh6;
      X->av[i] = cv[i] * h6; X->ap[i] = cp[i] * hh6;
    }
    X->d.t = h; X->dt = dh;
  }
}
// a <- a followed by x.  Values: P <- P + V t + R(Q) p, V <- V + R(Q) v, Q <- Q q, T <- T + t.  Tangents: with Q -> exp(dth_a) Q and
// q -> exp(dth_x) q the product moves as exp(dth_a) exp(R(Q) dth_x) Q q; a rotated vector as d (R u) = dth_a x (R u) + R du.
// The accelerometer partials (Q carries none): ap <- ap + av t + R(Q) ap_x, av <- av + R(Q) av_x.
VC_HD void imu_delta_t_then(IntervalDeltaT* A, const IntervalDeltaT& X) {
  double R[9], rp[3], rv[3], t1[3], t2[3], t3[3], q[4];
  quat_to_R(A->d.q, R);
  mat3_vec(R, X.d.p, rp); mat3_vec(R, X.d.v, rv);
  cross3(A->dth, rp, t1); mat3_vec(R, X.dp, t2);
  for (int i = 0; i < 3; ++i) A->dp[i] = ((A->dp[i] + A->dv[i] * X.d.t) + A->d.v[i] * X.dt) + (t1[i] + t2[i]);
  cross3(A->dth, rv, t1); mat3_vec(R, X.dv, t2);
  for (int i = 0; i < 3; ++i) A->dv[i] = A->dv[i] + (t1[i] + t2[i]);
  mat3_vec(R, X.dth, t3);
  for (int i = 0; i < 3; ++i) A->dth[i] = A->dth[i] + t3[i];
  A->dt = A->dt + X.dt;
  mat3_vec(R, X.ap, t1); mat3_vec(R, X.av, t2);
  for (int i = 0; i < 3; ++i) {
    A->ap[i] = (A->ap[i] + A->av[i] * X.d.t) + t1[i];
    A->av[i] = A->av[i] + t2[i];
  }
  for (int i = 0; i < 3; ++i) {
    A->d.p[i] = (A->d.p[i] + A->d.v[i] * X.d.t) + rp[i];
    A->d.v[i] = A->d.v[i] + rv[i];
  }
  quat_mul(A->d.q, X.d.q, q);
  for (int i = 0; i < 4; ++i) A->d.q[i] = q[i];
  A->d.t = A->d.t + X.d.t;
}
// Record columns (k_imu_jac's dcol): 0 values | 1..3 gyro bias | 4..6 accelerometer bias | 7..9 gyro scale | 10..12 accelerometer
// scale | 13 time offset.  Group g of the kernel (8 lanes) carries direction g (0..5: gyro parameter g, 6: time offset) and the
// accelerometer parameter g beside it.
VC_HD int imu_group_dual_col(int g) { return g < 3 ? 1 + g : (g < 6 ? 4 + g : 13); }
VC_HD int imu_group_accel_col(int g) { return g < 3 ? 4 + g : 7 + g; }
// what lane 7 of group g stores once the block's delta has been composed.  The record keeps quaternion partials (what k_imu_jac's
// dual numbers start from): d Q = 1/2 (dth, 0) Q.
VC_HD void imu_block_record_store(const IntervalDeltaT& X, int g, double* rec) {
  if (g > 6) return;
  double* cd = rec + imu_group_dual_col(g) * 11;
  const double hq[4] = {0.5 * X.dth[0], 0.5 * X.dth[1], 0.5 * X.dth[2], 0.0};
  double dq[4];
  quat_mul(hq, X.d.q, dq);
  for (int k = 0; k < 4; ++k) cd[k] = dq[k];
  for (int k = 0; k < 3; ++k) { cd[4 + k] = X.dp[k]; cd[7 + k] = X.dv[k]; }
  cd[10] = X.dt;
  if (g < 6) {
    double* ca = rec + imu_group_accel_col(g) * 11;
    for (int k = 0; k < 4; ++k) ca[k] = 0.0;
    for (int k = 0; k < 3; ++k) { ca[4 + k] = X.ap[k]; ca[7 + k] = X.av[k]; }
    ca[10] = 0.0;
  }
  if (g == 0) {
    for (int k = 0; k < 4; ++k) rec[k] = X.d.q[k];
    for (int k = 0; k < 3; ++k) { rec[4 + k] = X.d.p[k]; rec[7 + k] = X.d.v[k]; }
    rec[10] = X.d.t;
  }
}
// Host form of k_imu_block, for tests/host_harness: the same bracketing -- rounds of 16 intervals, lane l of a group takes
// intervals 2 l + 1, 2 l + 2 of the round and appends them, a Hillis-Steele inclusive scan over the group's 8 lanes, the rounds'
// totals appended in order.  Returns 0 for an empty sample range (the kernel flags it by T = -1 in the record).
inline int imu_block_delta_record(const ImuView& buf, double t_start, double t_end, double toff, const double* b, const double* sf,
                                  double* rec /* kBlockDeltaStride */) {
  const ImuRange rg = imu_range(buf, t_start, t_end, toff);
  if (!rg.valid) { rec[10] = -1.0; return 0; }
  const int n_int = (rg.k1 - rg.k0 + 1) + 1;
  for (int g = 0; g < 7; ++g) {
    const D1 offD = mk(toff, g == 6 ? 1.0 : 0.0);
    IntervalDeltaT carry;
    for (int base = 0; base < n_int; base += 16) {
      IntervalDeltaT X[8], Y[8];
      for (int l = 0; l < 8; ++l) {
        IntervalDeltaT second;
        imu_interval_delta_t(buf, rg, offD, t_start, t_end, base + 2 * l + 1, n_int, b, sf, g, &X[l]);
        imu_interval_delta_t(buf, rg, offD, t_start, t_end, base + 2 * l + 2, n_int, b, sf, g, &second);
        imu_delta_t_then(&X[l], second);
      }
      for (int n = 1; n < 8; n <<= 1) {
        for (int l = 0; l < 8; ++l) { Y[l] = X[l]; if (l >= n) { Y[l] = X[l - n]; imu_delta_t_then(&Y[l], X[l]); } }
        for (int l = 0; l < 8; ++l) X[l] = Y[l];
      }
      if (base == 0) carry = X[7]; else imu_delta_t_then(&carry, X[7]);
    }
    imu_block_record_store(carry, g, rec);
  }
  return 1;
}

// Local columns of the block (k_imu_jac's order): frame j pose 6 | its velocity 3 | frame j-1 pose 6 | its velocity 3 | gravity 2 |
// biases 6 | scale factors 6 | time offset.  One lane's share: residual values r[9] and their partials along local column `col`
// (col < 0: values only), from the block's delta record.  `valid`: the block's sample range is not empty.
VC_HD void imu_block_final_direction(int valid, const double* block_rec, const double* w_sqrt, int rotation_only, const double* T2,
                                     const double* T1, const double* v2, const double* v1, const double* grav9 /* imu_gravity_record */, int col,
                                     double* r, double* dr) {
  if (!valid) { for (int i = 0; i < 9; ++i) { r[i] = 0.0; dr[i] = 0.0; } return; }
  const int dcol = (col >= 20) ? col - 19 : 0;                 // biases 1..6, scale factors 7..12, time offset 13
  D1 Q[4], P[3], V[3];
  double bd[11];
  for (int k = 0; k < 11; ++k) { const double d = block_rec[dcol * 11 + k]; bd[k] = dcol ? d : 0.0; }      // (unconditional loads)
  for (int k = 0; k < 4; ++k) Q[k] = mk(block_rec[k], bd[k]);
  for (int k = 0; k < 3; ++k) { P[k] = mk(block_rec[4 + k], bd[4 + k]); V[k] = mk(block_rec[7 + k], bd[7 + k]); }
  const D1 Tt = mk(block_rec[10], bd[10]);
  // seeds: poses move along T exp(delta) (local_jac_se3), everything else is a plain coordinate
  D1 T2D[7], v2D[3], gw[3], q0[4], p0[3], v0[3];
  {
    double J[42];
    local_jac_se3(T2, J);
    for (int i = 0; i < 7; ++i) {
      double sel = 0.0;
      for (int c = 0; c < 6; ++c) sel = (col == c) ? J[i * 6 + c] : sel;
      T2D[i] = mk(T2[i], sel);
    }
    local_jac_se3(T1, J);
    for (int i = 0; i < 7; ++i) {
      double sel = 0.0;
      for (int c = 0; c < 6; ++c) sel = (col == 9 + c) ? J[i * 6 + c] : sel;
      if (i < 4) q0[i] = mk(T1[i], sel); else p0[i - 4] = mk(T1[i], sel);
    }
  }
  for (int k = 0; k < 3; ++k) { v2D[k] = mk(v2[k], col == 6 + k ? 1.0 : 0.0); v0[k] = mk(v1[k], col == 15 + k ? 1.0 : 0.0); }
  for (int k = 0; k < 3; ++k) gw[k] = mk(grav9[k], col == 18 ? grav9[3 + k] : (col == 19 ? grav9[6 + k] : 0.0));
  PoseV<D1> x;
  D1 rp[3], rv[3];
  tq_rotate(q0, P, rp);
  tq_rotate(q0, V, rv);
  tq_mul(q0, Q, x.q);
  const D1 ht2 = 0.5 * (Tt * Tt);
  for (int i = 0; i < 3; ++i) {
    x.p[i] = ((p0[i] + v0[i] * Tt) - gw[i] * ht2) + rp[i];
    x.v[i] = (v0[i] - gw[i] * Tt) + rv[i];
  }
  D1 res[9];
  imu_residual_tail(x, w_sqrt, rotation_only, T2D, v2D, res);
  for (int i = 0; i < 9; ++i) { r[i] = res[i].a; dr[i] = res[i].v; }
}

}  // namespace vc
