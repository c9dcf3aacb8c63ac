// vc_math.hpp -- closed-form arithmetic of the calibration hot path, written for
// the HIP kernels (vc_kernels.hip) and usable from host code.
//
// What the reference evaluates with ceres::Jet autodiff per corner
// (ImuReprojectionCostFunctor, ceres-cost-functions.h:350-373, instantiated at
// vicalibrator.h:412-453) is evaluated here in closed form:
//   p_c = R_ck (R_wk^T (p_w - t_wk)) + t_ck ,  r = Project(p_c, K) - z
//   dr/d[v_wk | w_wk | w_ck | t_ck | K] = [ -A R_ck | (A [q]x) R_ck | -(A [q]x) R_ck | A | B ]
// with A = dProject/dp_c (2x3), B = dProject/dK (2xnk), q = p_c - t_ck
// (manifold Jacobians w.r.t. T <- T exp(delta), local-param-se3.h:14-26, :107-119).
// Only the "unique" columns u = [A | A[q]x | B | r] are formed per corner; the
// constant-per-tile rotations are applied after the tile reduction.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define VC_HD __host__ __device__ __forceinline__
#else
#define VC_HD inline
#endif

namespace vc {

// model ids = order of the -models strings (vicalib-engine.cc:203-253)
enum Model { kFov = 0, kPoly2 = 1, kPoly3 = 2, kKb4 = 3, kLinear = 4, kRational6 = 5 };
VC_HD int model_nk(int m) { return m == kFov ? 5 : m == kPoly2 ? 6 : m == kPoly3 ? 7 : m == kKb4 ? 8 : m == kLinear ? 4 : m == kRational6 ? 10 : -1; }

constexpr int kPoseStride = 8;    // [qx qy qz qw tx ty tz pad] = 64 B per frame
constexpr int kCamStride = 24;    // [T_ck(7) pad | K(<=10) pad..]
constexpr int kCamK = 8;          // offset of K inside a camera record
constexpr int kUCols = 16;        // padded width of a unique-column row
// A tile's Gram record: the 16 x 16 block over u = [A (3) | A x q (3) | B (nk) | r] at 0, and 16 more doubles at kGGrad.  With
// nk <= 9 the residual column r sits inside the block (column 6 + nk).  rational6 has nk = 10: its 16 Jacobian columns fill the
// block and the products with r (J^T r, 16 values) live in the side vector at kGGrad.
constexpr int kGGrad = 256;
// In HBM the record is PACKED (round 4): the block's upper triangle row by row (136 doubles) + the side vector = kGPack doubles --
// the symmetric block was written and read in full before (2176 -> 1216 B per tile and buffer).  Readers expand it on load (LDS images and
// chunk partials keep the 16 x 16 layout above).
constexpr int kGPack = 152, kGPackGrad = 136;
VC_HD int g_pack_idx(int r, int c) { const int a = r <= c ? r : c, b = r <= c ? c : r; return a * 16 - (a * (a - 1)) / 2 + (b - a); }
VC_HD double gram_grad(const double* G, int row, int nk) { return nk < 10 ? G[row * kUCols + 6 + nk] : G[kGGrad + row]; }
constexpr int kSoftL1A = 0;       // loss ids
constexpr double kSophusEps = 1e-10;

// ---- quaternion / SO3 / SE3 (Sophus pre-1.0 + Eigen semantics) ----
VC_HD void quat_to_R(const double* q, double* R) {
  const double tx = 2.0 * q[0], ty = 2.0 * q[1], tz = 2.0 * q[2];
  const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1.0 - (txx + tyy);
}
VC_HD void quat_mul(const double* a, const double* b, double* o) {
  const double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  const double y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  const double z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  const double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
VC_HD void quat_rotate(const double* q, const double* v, double* o) {
  const double ux = 2.0 * (q[1] * v[2] - q[2] * v[1]);
  const double uy = 2.0 * (q[2] * v[0] - q[0] * v[2]);
  const double uz = 2.0 * (q[0] * v[1] - q[1] * v[0]);
  const double rx = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
  const double ry = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
  const double rz = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
  o[0] = rx; o[1] = ry; o[2] = rz;
}
// 1/sqrt(d) to full double precision: hardware estimate (v_rsq_f64) + two Newton steps on the device
// (replaces an IEEE sqrt + an IEEE divide, ~70 dependent instructions, on the factorisation's critical path)
VC_HD double fast_rsqrt(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rsq(d);
  y = y * (1.5 - 0.5 * d * y * y);
  y = y * (1.5 - 0.5 * d * y * y);
  return y;
#else
  return 1.0 / sqrt(d);
#endif
}
// 1/d to full double precision: hardware estimate (v_rcp_f64) + two Newton steps on the device -- 5 instructions instead of the
// ~11 of an IEEE divide (v_div_scale x2, v_rcp, 4 fma, v_div_fmas, v_div_fixup), no special-case handling: operands here are
// finite, normal and non-zero (depths, norms, polynomial denominators), or the result is discarded by a select.
VC_HD double fast_rcp(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rcp(d);
  y = y * (2.0 - d * y);
  y = y * (2.0 - d * y);
  return y;
#else
  return 1.0 / d;
#endif
}
// sqrt(d) and 1/sqrt(d) from one v_rsq_f64 chain (d > 0; d = 0 gives 0 and +inf)
VC_HD void fast_sqrt_rsqrt(double d, double* s, double* is) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double r = fast_rsqrt(d);
  *is = r;
  *s = d > 0.0 ? d * r : 0.0;
#else
  const double q = sqrt(d);
  *s = q; *is = 1.0 / q;
#endif
}
// exp of a rotation vector as a unit quaternion [sin(th/2) w/th, cos(th/2)].  For th/2 <= pi/4 both factors are
// polynomials in z = th^2/4 (the fdlibm kernel forms sin h = h + h^3 S(z), cos h = 1 - z/2 + z^2 C(z), < 1 ulp): no square
// root, no division, no library call and no special case at th -> 0 -- LM steps practically never leave this range, and
// the math library's sin / cos / IEEE divide are ~1k cycles of a single lane's time on the GPU.  Larger angles take the
// closed form.
VC_HD void so3_exp(const double* w, double* q) {
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double imag, real;
  if (th2 <= 2.4) {
    const double z = 0.25 * th2;
    const double S = -1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 +
                     z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
    const double Cc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                      z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
    imag = 0.5 + 0.5 * z * S;                     // sin(th/2) / th
    real = (1.0 - 0.5 * z) + z * z * Cc;
  } else {
    const double th = sqrt(th2), half = 0.5 * th;
    imag = sin(half) / th;
    real = cos(half);
  }
  q[0] = imag * w[0]; q[1] = imag * w[1]; q[2] = imag * w[2]; q[3] = real;
}
// R <- R * exp(w), renormalised (LocalParamSo3::Plus, local-param-se3.h:107-119)
VC_HD void so3_plus(const double* q, const double* w, double* o) {
  double e[4], r[4];
  so3_exp(w, e);
  quat_mul(q, e, r);
  const double in = fast_rsqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
  o[0] = r[0] * in; o[1] = r[1] * in; o[2] = r[2] * in; o[3] = r[3] * in;
}
// T <- T * exp([v, w]) (LocalParamSe3::Plus, local-param-se3.h:14-26).  V = I + a [w]x + b [w]x^2 with
// a = (1 - cos th) / th^2, b = (th - sin th) / th^3: their Taylor series in t = th^2 below t = 0.5 (truncation < 1e-16; the
// closed forms cancel there anyway), the closed forms above.
VC_HD void se3_plus(const double* T, const double* d, double* o) {
  const double* w = d + 3;
  const double t = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double a, b;
  if (t < 0.5) {
    a = 1.0 / 2 + t * (-1.0 / 24 + t * (1.0 / 720 + t * (-1.0 / 40320 + t * (1.0 / 3628800 + t * (-1.0 / 479001600 +
        t * (1.0 / 87178291200.0 + t * (-1.0 / 20922789888000.0)))))));
    b = 1.0 / 6 + t * (-1.0 / 120 + t * (1.0 / 5040 + t * (-1.0 / 362880 + t * (1.0 / 39916800 + t * (-1.0 / 6227020800.0 +
        t * (1.0 / 1307674368000.0 + t * (-1.0 / 355687428096000.0)))))));
  } else {
    const double th = sqrt(t);
    a = (1.0 - cos(th)) / t;
    b = (th - sin(th)) / (t * th);
  }
  const double wx = w[0], wy = w[1], wz = w[2];
  double V[9];
  V[0] = 1.0 + b * (-(wy * wy + wz * wz)); V[1] = -a * wz + b * (wx * wy); V[2] = a * wy + b * (wx * wz);
  V[3] = a * wz + b * (wx * wy); V[4] = 1.0 + b * (-(wx * wx + wz * wz)); V[5] = -a * wx + b * (wy * wz);
  V[6] = -a * wy + b * (wx * wz); V[7] = a * wx + b * (wy * wz); V[8] = 1.0 + b * (-(wx * wx + wy * wy));
  const double tv[3] = {V[0] * d[0] + V[1] * d[1] + V[2] * d[2], V[3] * d[0] + V[4] * d[1] + V[5] * d[2],
                        V[6] * d[0] + V[7] * d[1] + V[8] * d[2]};
  double rt[3];
  quat_rotate(T, tv, rt);
  so3_plus(T, w, o);
  o[4] = T[4] + rt[0]; o[5] = T[5] + rt[1]; o[6] = T[6] + rt[2];
}

// ---- robust losses (ceres::SoftLOneLoss(0.5) vicalibrator.h:127, CauchyLoss(100) :133) ----
VC_HD void loss_soft_l1(double s, double* rho, double* rho1) {
  const double b = 0.25, c = 4.0;
  double t, it;
  fast_sqrt_rsqrt(1.0 + s * c, &t, &it);
  *rho = 2.0 * b * (t - 1.0);
  *rho1 = it;
}
VC_HD void loss_cauchy100(double s, double* rho, double* rho1) {
  const double b = 1.0e4, c = 1.0e-4;
  const double sum = 1.0 + s * c;
  *rho = b * log(sum);
  *rho1 = 1.0 / sum;
}

// Branch-free arctangent (Cephes atan.c scheme: two-step argument reduction + degree 4/5 rational, < 2 ulp): the math
// library's atan carries a divergent branch, which on the GPU would cut the region that overlaps with the MFMAs.
VC_HD double vc_atan(double x) {
  const double ax = fabs(x);
  const bool big = ax > 2.41421356237309504880, mid = ax > 0.66;
  // one division with selected operands (two conditional divisions would be turned back into branches by the compiler)
  const double t = (big ? -1.0 : (mid ? ax - 1.0 : ax)) * fast_rcp(big ? ax : (mid ? ax + 1.0 : 1.0));
  const double y0 = big ? 1.57079632679489661923 : (mid ? 0.78539816339744830962 : 0.0);
  const double more = big ? 6.123233995736765886130e-17 : (mid ? 3.061616997868382943065e-17 : 0.0);
  const double z = t * t;
  const double pn = (((-8.750608600031904122785e-1 * z - 1.615753718733365076637e1) * z - 7.500855792314704667340e1) * z
                     - 1.228866684490136173410e2) * z - 6.485021904942025371773e1;
  const double qd = ((((z + 2.485846490142306297962e1) * z + 1.650270098316988542046e2) * z + 4.328810604912902668951e2) * z
                     + 4.853903996359136964868e2) * z + 1.945506571482613964425e2;
  const double r = y0 + (t * (z * pn * fast_rcp(qd)) + t + more);
  return x < 0.0 ? -r : r;
}
// atan(y / x) for y >= 0 without forming the quotient: the argument reduction of vc_atan works on the pair (the reduced argument is a
// ratio of sums of y and |x|), one reciprocal instead of a division followed by a second one
VC_HD double vc_atan_ratio_pos(double y, double x) {
  const double ax = fabs(x);
  const bool big = y > 2.41421356237309504880 * ax, mid = y > 0.66 * ax;
  const double t = (big ? -ax : (mid ? y - ax : y)) * fast_rcp(big ? y : (mid ? y + ax : ax));
  const double y0 = big ? 1.57079632679489661923 : (mid ? 0.78539816339744830962 : 0.0);
  const double more = big ? 6.123233995736765886130e-17 : (mid ? 3.061616997868382943065e-17 : 0.0);
  const double z = t * t;
  const double pn = (((-8.750608600031904122785e-1 * z - 1.615753718733365076637e1) * z - 7.500855792314704667340e1) * z
                     - 1.228866684490136173410e2) * z - 6.485021904942025371773e1;
  const double qd = ((((z + 2.485846490142306297962e1) * z + 1.650270098316988542046e2) * z + 4.328810604912902668951e2) * z
                     + 4.853903996359136964868e2) * z + 1.945506571482613964425e2;
  const double r = y0 + (t * (z * pn * fast_rcp(qd)) + t + more);
  return x < 0.0 ? -r : r;
}
// atan2(y, x) for y >= 0
VC_HD double vc_atan2_pos(double y, double x) {
  const double a = vc_atan_ratio_pos(y, x);
  return x > 0.0 ? a : (x < 0.0 ? a + 3.14159265358979323846 : 1.57079632679489661923);
}

// per-camera constants that do not depend on the corner (fov: m = 2 tan(w/2), dm = dm/dw)
struct ModelPre { double m, dm; };
VC_HD void model_precompute(int model, const double* K, ModelPre* p) {
  p->m = 0.0; p->dm = 0.0;
  if (model == kFov) { p->m = 2.0 * tan(0.5 * K[4]); p->dm = 1.0 + 0.25 * p->m * p->m; }
}

// ---- camera projections with closed-form Jacobians (Calibu formulas) ----
// A: 2x3 d pix / d p_c ; B: 2 x nk d pix / d K  (row-major). JAC=false skips A and B.
template <bool JAC>
VC_HD void project_radial(int model, const double* pc, const double* K, const ModelPre& pre, double* pix, double* A, double* B) {
  const double iz = fast_rcp(pc[2]);
  const double x = pc[0] * iz, y = pc[1] * iz;
  const double r2 = x * x + y * y;
  double fac = 1.0, h = 0.0;           // h = fac'(r) / r
  double dk0 = 0.0, dk1 = 0.0, dk2 = 0.0, dk3 = 0.0, dk4 = 0.0, dk5 = 0.0;
  if (model == kFov) {
    // Branch-free (selects, not jumps): on the GPU this arithmetic runs in the shadow of the previous pass's MFMAs and a
    // jump would cut the scheduling region.  The discarded alternatives may hold inf/nan; they are never blended in.
    const double w = K[4];
    const double m = pre.m, dm = pre.dm;
    const bool w_big = w * w > 1e-5, r_small = r2 < 1e-5;
    double r, ir_raw;
    fast_sqrt_rsqrt(r2, &r, &ir_raw);
    const double at = vc_atan(r * m);
    // every reciprocal is unconditional and on a safe operand; the alternatives differ only by cheap selects
    const double iw = fast_rcp(w_big ? w : 1.0), ir = r_small ? 1.0 : ir_raw, iden = fast_rcp(1.0 + r2 * m * m);
    const double fac_g = at * ir * iw, fac_s = m * iw;
    fac = w_big ? (r_small ? fac_s : fac_g) : 1.0;
    if (JAC) {
      const double h_g = (m * r * iden - at) * ir * ir * ir * iw;
      const double dk_g = dm * iw * iden - fac_g * iw, dk_s = dm * iw - m * iw * iw;
      h = (w_big && !r_small) ? h_g : 0.0;
      dk0 = w_big ? (r_small ? dk_s : dk_g) : 0.0;
    }
  } else if (model == kPoly2) {
    fac = 1.0 + r2 * (K[4] + r2 * K[5]);
    if (JAC) { h = 2.0 * K[4] + 4.0 * K[5] * r2; dk0 = r2; dk1 = r2 * r2; }
  } else if (model == kPoly3) {
    fac = 1.0 + r2 * (K[4] + r2 * (K[5] + r2 * K[6]));
    if (JAC) { h = 2.0 * K[4] + r2 * (4.0 * K[5] + 6.0 * K[6] * r2); dk0 = r2; dk1 = r2 * r2; dk2 = dk1 * r2; }
  } else if (model == kRational6) {
    // fac = N / Dn, N = 1 + k1 r^2 + k2 r^4 + k3 r^6, Dn = 1 + k4 r^2 + k5 r^4 + k6 r^6
    const double N = 1.0 + r2 * (K[4] + r2 * (K[5] + r2 * K[6])), Dn = 1.0 + r2 * (K[7] + r2 * (K[8] + r2 * K[9]));
    const double iD = fast_rcp(Dn);
    fac = N * iD;
    if (JAC) {
      const double dN = K[4] + r2 * (2.0 * K[5] + 3.0 * K[6] * r2), dD = K[7] + r2 * (2.0 * K[8] + 3.0 * K[9] * r2);
      h = 2.0 * (dN - fac * dD) * iD;                 // d fac / d r^2 = (N' Dn - N Dn') / Dn^2, times 2
      const double r4 = r2 * r2, r6 = r4 * r2, q = -fac * iD;
      dk0 = r2 * iD; dk1 = r4 * iD; dk2 = r6 * iD; dk3 = q * r2; dk4 = q * r4; dk5 = q * r6;
    }
  }
  const double fu = K[0], fv = K[1];
  pix[0] = fu * x * fac + K[2];
  pix[1] = fv * y * fac + K[3];
  if (JAC) {
    const double uxx = fu * (fac + x * x * h), uxy = fu * x * y * h;
    const double vxx = fv * x * y * h, vxy = fv * (fac + y * y * h);
    A[0] = uxx * iz; A[1] = uxy * iz; A[2] = -(uxx * x + uxy * y) * iz;
    A[3] = vxx * iz; A[4] = vxy * iz; A[5] = -(vxx * x + vxy * y) * iz;
    const int nk = model_nk(model);
    B[0] = x * fac; B[1] = 0.0; B[2] = 1.0; B[3] = 0.0;
    B[nk + 0] = 0.0; B[nk + 1] = y * fac; B[nk + 2] = 0.0; B[nk + 3] = 1.0;
    if (nk > 4) { B[4] = fu * x * dk0; B[nk + 4] = fv * y * dk0; }
    if (nk > 5) { B[5] = fu * x * dk1; B[nk + 5] = fv * y * dk1; }
    if (nk > 6) { B[6] = fu * x * dk2; B[nk + 6] = fv * y * dk2; }
    if (nk > 7) { B[7] = fu * x * dk3; B[nk + 7] = fv * y * dk3; B[8] = fu * x * dk4; B[nk + 8] = fv * y * dk4; B[9] = fu * x * dk5; B[nk + 9] = fv * y * dk5; }
  }
}
template <bool JAC>
VC_HD void project_kb4(const double* pc, const double* K, double* pix, double* A, double* B) {
  const double X = pc[0], Y = pc[1], Z = pc[2];
  const double rho2 = X * X + Y * Y;
  double rho, irho;
  fast_sqrt_rsqrt(rho2, &rho, &irho);
  const double th = vc_atan2_pos(rho, Z);
  const double t2 = th * th;
  const double poly = 1.0 + t2 * (K[4] + t2 * (K[5] + t2 * (K[6] + t2 * K[7])));
  const double Rr = th * poly;
  const bool off_axis = rho > 0.0;            // selects, not jumps (see project_radial)
  const double irho1 = off_axis ? irho : 1.0;
  const double c = X * irho1 + (off_axis ? 0.0 : 1.0), s = Y * irho1;   // on the axis X = Y = 0
  const double fu = K[0], fv = K[1];
  pix[0] = fu * Rr * c + K[2];
  pix[1] = fv * Rr * s + K[3];
  if (JAC) {
    const double dR = 1.0 + t2 * (3.0 * K[4] + t2 * (5.0 * K[5] + t2 * (7.0 * K[6] + t2 * 9.0 * K[7])));
    const double n2 = rho2 + Z * Z;
    const double in2 = fast_rcp(n2);
    const double thX = Z * c * in2, thY = Z * s * in2, thZ = -rho * in2;
    // R/rho -> 1/Z on the axis
    const double Rq = off_axis ? Rr * irho : fast_rcp(Z);
    const double cX = s * s * Rq, cY = -c * s * Rq, sX = cY, sY = c * c * Rq;   // R * dc/dX etc.
    A[0] = fu * (dR * thX * c + cX); A[1] = fu * (dR * thY * c + cY); A[2] = fu * dR * thZ * c;
    A[3] = fv * (dR * thX * s + sX); A[4] = fv * (dR * thY * s + sY); A[5] = fv * dR * thZ * s;
    const double t3 = t2 * th, t5 = t3 * t2, t7 = t5 * t2, t9 = t7 * t2;
    B[0] = Rr * c; B[1] = 0.0; B[2] = 1.0; B[3] = 0.0;
    B[4] = fu * c * t3; B[5] = fu * c * t5; B[6] = fu * c * t7; B[7] = fu * c * t9;
    B[8] = 0.0; B[9] = Rr * s; B[10] = 0.0; B[11] = 1.0;
    B[12] = fv * s * t3; B[13] = fv * s * t5; B[14] = fv * s * t7; B[15] = fv * s * t9;
  }
}
template <bool JAC>
VC_HD void project_any(int model, const double* pc, const double* K, const ModelPre& pre, double* pix, double* A, double* B) {
  if (model == kKb4) project_kb4<JAC>(pc, K, pix, A, B);
  else project_radial<JAC>(model, pc, K, pre, pix, A, B);
}

// Per-tile constants: p_c = Rcw p_w + tcw, with R_ck kept for the post-reduction transform.
struct TileXf { double Rcw[9]; double tcw[3]; double tck[3]; };
VC_HD void make_tile_xf(const double* T_wk, const double* T_ck, TileXf* x) {
  double Rwk[9], Rck[9];
  quat_to_R(T_wk, Rwk);
  quat_to_R(T_ck, Rck);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)   // Rcw = Rck * Rwk^T
      x->Rcw[3 * i + j] = Rck[3 * i] * Rwk[3 * j] + Rck[3 * i + 1] * Rwk[3 * j + 1] + Rck[3 * i + 2] * Rwk[3 * j + 2];
  for (int i = 0; i < 3; ++i) {
    x->tck[i] = T_ck[4 + i];
    x->tcw[i] = T_ck[4 + i] - (x->Rcw[3 * i] * T_wk[4] + x->Rcw[3 * i + 1] * T_wk[5] + x->Rcw[3 * i + 2] * T_wk[6]);
  }
}
VC_HD void tile_point(const TileXf& x, const double* pw, double* pc) {
  pc[0] = x.Rcw[0] * pw[0] + x.Rcw[1] * pw[1] + x.Rcw[2] * pw[2] + x.tcw[0];
  pc[1] = x.Rcw[3] * pw[0] + x.Rcw[4] * pw[1] + x.Rcw[5] * pw[2] + x.tcw[1];
  pc[2] = x.Rcw[6] * pw[0] + x.Rcw[7] * pw[1] + x.Rcw[8] * pw[2] + x.tcw[2];
}

// Robustified unique-column rows of one corner: row[i] = sqrt(w) [A_i | A_i x q | B_i | r_i | 0..],
// w = mult * rho'(|r|^2); returns mult * rho(|r|^2) (twice the block's cost).  rs (optional): sqrt(w) r, the residual as it is
// scaled in the rows -- rational6's 16 Jacobian columns leave no room for it in the row (see kGGrad).
template <int MODEL>
VC_HD double corner_rows(const TileXf& x, const double* K, const ModelPre& pre, const double* pw, double u, double v, double mult,
                         double* row0 /*16*/, double* row1 /*16*/, double* rs = nullptr /*2*/) {
  constexpr int model = MODEL;
  double pc[3], pix[2], A[6], B[20];
  tile_point(x, pw, pc);
  project_any<true>(model, pc, K, pre, pix, A, B);
  const double r0 = pix[0] - u, r1 = pix[1] - v;
  double rho, rho1;
  loss_soft_l1(r0 * r0 + r1 * r1, &rho, &rho1);
  double sw, isw_unused;
  fast_sqrt_rsqrt(mult * rho1, &sw, &isw_unused);
  const double q0 = pc[0] - x.tck[0], q1 = pc[1] - x.tck[1], q2 = pc[2] - x.tck[2];
  constexpr int nk = MODEL == kFov ? 5 : MODEL == kPoly2 ? 6 : MODEL == kPoly3 ? 7 : MODEL == kKb4 ? 8 : MODEL == kRational6 ? 10 : 4;
  for (int i = 7 + nk; i < kUCols; ++i) { row0[i] = 0.0; row1[i] = 0.0; }
  if (rs) { rs[0] = sw * r0; rs[1] = sw * r1; }
  for (int i = 0; i < 2; ++i) {
    double* row = i ? row1 : row0;
    const double* a = A + 3 * i;
    row[0] = sw * a[0]; row[1] = sw * a[1]; row[2] = sw * a[2];
    row[3] = sw * (a[1] * q2 - a[2] * q1);
    row[4] = sw * (a[2] * q0 - a[0] * q2);
    row[5] = sw * (a[0] * q1 - a[1] * q0);
    for (int k = 0; k < nk; ++k) row[6 + k] = sw * B[i * nk + k];
    if (6 + nk < kUCols) row[6 + nk] = sw * (i ? r1 : r0);
  }
  return mult * rho;
}
// Residual only: r[2]; returns rho(|r|^2).
template <int MODEL>
VC_HD double corner_residual(const TileXf& x, const double* K, const ModelPre& pre, const double* pw, double u, double v, double* r) {
  constexpr int model = MODEL;
  double pc[3], pix[2];
  tile_point(x, pw, pc);
  project_any<false>(model, pc, K, pre, pix, nullptr, nullptr);
  r[0] = pix[0] - u; r[1] = pix[1] - v;
  double rho, rho1;
  loss_soft_l1(r[0] * r[0] + r[1] * r[1], &rho, &rho1);
  return rho;
}

// ---- per-camera column maps -------------------------------------------------------
// Shared columns of camera c, in layout order [w_ck(3) if free][t_ck(3) if free][K(nk) if free].
constexpr int kCamRotFree = 1, kCamTransFree = 2, kCamKFree = 4;
VC_HD int cam_ncols(int flags, int nk) { return ((flags & kCamRotFree) ? 3 : 0) + ((flags & kCamTransFree) ? 3 : 0) + ((flags & kCamKFree) ? nk : 0); }

// From a tile's 16x16 Gram block G (over u = [A V B r]) build, for the frame:
//   Hff (6x6, adds), gf (6, adds) and -- when W != nullptr -- W (6 x ncols, row-major ld 16, overwrites).
// J_f = [A V] * diag(-R, R);  J_c = [ -V R | A | B ].
VC_HD void tile_to_frame_blocks(const double* G, const double* Rck, int nk, int flags, double* Hff /*36*/, double* gf /*6*/,
                                double* W /*6x16*/) {
  // T = Gaa * Qa  (6x6), Qa = diag(-R, R)
  double T[36];
  for (int i = 0; i < 6; ++i) {
    const double* g = G + i * kUCols;
    for (int j = 0; j < 3; ++j) {
      T[i * 6 + j] = -(g[0] * Rck[j] + g[1] * Rck[3 + j] + g[2] * Rck[6 + j]);
      T[i * 6 + 3 + j] = g[3] * Rck[j] + g[4] * Rck[3 + j] + g[5] * Rck[6 + j];
    }
  }
  if (Hff) {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 6; ++j) {
        Hff[i * 6 + j] += -(Rck[i] * T[0 * 6 + j] + Rck[3 + i] * T[1 * 6 + j] + Rck[6 + i] * T[2 * 6 + j]);
        Hff[(3 + i) * 6 + j] += Rck[i] * T[3 * 6 + j] + Rck[3 + i] * T[4 * 6 + j] + Rck[6 + i] * T[5 * 6 + j];
      }
    for (int i = 0; i < 3; ++i) {
      gf[i] += -(Rck[i] * gram_grad(G, 0, nk) + Rck[3 + i] * gram_grad(G, 1, nk) + Rck[6 + i] * gram_grad(G, 2, nk));
      gf[3 + i] += Rck[i] * gram_grad(G, 3, nk) + Rck[3 + i] * gram_grad(G, 4, nk) + Rck[6 + i] * gram_grad(G, 5, nk);
    }
  }
  if (W) {
    // W = Qa^T Gaa Ea | Qa^T GaB ;  (Qa^T Gaa)[r][p] = T[p][r] (Gaa symmetric)
    int col = 0;
    if (flags & kCamRotFree) {      // columns -V R : sum_p T[3+p][r] * (-R[p][j])
      for (int r = 0; r < 6; ++r)
        for (int j = 0; j < 3; ++j)
          W[r * kUCols + col + j] = -(T[3 * 6 + r] * Rck[j] + T[4 * 6 + r] * Rck[3 + j] + T[5 * 6 + r] * Rck[6 + j]);
      col += 3;
    }
    if (flags & kCamTransFree) {    // columns A : T[p][r], p = 0..2
      for (int r = 0; r < 6; ++r)
        for (int j = 0; j < 3; ++j) W[r * kUCols + col + j] = T[j * 6 + r];
      col += 3;
    }
    if (flags & kCamKFree) {        // Qa^T GaB
      for (int j = 0; j < nk; ++j) {
        const double b0 = G[0 * kUCols + 6 + j], b1 = G[1 * kUCols + 6 + j], b2 = G[2 * kUCols + 6 + j];
        const double b3 = G[3 * kUCols + 6 + j], b4 = G[4 * kUCols + 6 + j], b5 = G[5 * kUCols + 6 + j];
        for (int i = 0; i < 3; ++i) {
          W[i * kUCols + col + j] = -(Rck[i] * b0 + Rck[3 + i] * b1 + Rck[6 + i] * b2);
          W[(3 + i) * kUCols + col + j] = Rck[i] * b3 + Rck[3 + i] * b4 + Rck[6 + i] * b5;
        }
      }
      col += nk;
    }
    for (int r = 0; r < 6; ++r) for (int j = col; j < kUCols; ++j) W[r * kUCols + j] = 0.0;
  }
}
// Camera c's own block from the sum of its tiles' Gram blocks: Hcc (ncols x ncols, ld 16) and gc (ncols).
// P maps u -> columns: rot: rows V * (-R); trans: rows A * I; K: rows B * I.
VC_HD void cam_block_from_gsum(const double* G, const double* Rck, int nk, int flags, double* Hcc /*16x16*/, double* gc /*16*/) {
  double P[16 * 16];   // (6+nk) x ncols
  const int nu = 6 + nk;
  const int nc = cam_ncols(flags, nk);
  for (int i = 0; i < nu * 16; ++i) P[i] = 0.0;
  int col = 0;
  if (flags & kCamRotFree) { for (int p = 0; p < 3; ++p) for (int j = 0; j < 3; ++j) P[(3 + p) * 16 + col + j] = -Rck[3 * p + j]; col += 3; }
  if (flags & kCamTransFree) { for (int p = 0; p < 3; ++p) P[p * 16 + col + p] = 1.0; col += 3; }
  if (flags & kCamKFree) { for (int p = 0; p < nk; ++p) P[(6 + p) * 16 + col + p] = 1.0; col += nk; }
  for (int a = 0; a < nc; ++a) {
    // t = G[u,u] P[:,a]
    double t[16];
    for (int i = 0; i < nu; ++i) { double s = 0; for (int k = 0; k < nu; ++k) s += G[i * kUCols + k] * P[k * 16 + a]; t[i] = s; }
    for (int b = 0; b < nc; ++b) { double s = 0; for (int i = 0; i < nu; ++i) s += P[i * 16 + b] * t[i]; Hcc[b * 16 + a] = s; }
    double s = 0;
    for (int i = 0; i < nu; ++i) s += P[i * 16 + a] * gram_grad(G, i, nk);
    gc[a] = s;
  }
}

// N x N in-place lower Cholesky (row-major); returns false if not positive definite.
template <int N>
VC_HD bool chol_small(double* M, double* dinv = nullptr) {   // dinv[j] = 1 / L[j][j] (optional)
  for (int j = 0; j < N; ++j) {
    double d = M[j * N + j];
    for (int k = 0; k < j; ++k) d -= M[j * N + k] * M[j * N + k];
    if (!(d > 0.0)) return false;
    const double id = fast_rsqrt(d);
    M[j * N + j] = d * id;
    if (dinv) dinv[j] = id;
    for (int i = j + 1; i < N; ++i) {
      double s = M[i * N + j];
      for (int k = 0; k < j; ++k) s -= M[i * N + k] * M[j * N + k];
      M[i * N + j] = s * id;
    }
  }
  return true;
}
template <int N> VC_HD void fwd_solve(const double* L, double* x) {   // L y = x
  for (int i = 0; i < N; ++i) { double s = x[i]; for (int k = 0; k < i; ++k) s -= L[i * N + k] * x[k]; x[i] = s / L[i * N + i]; }
}
template <int N> VC_HD void bwd_solve(const double* L, double* x) {   // L^T y = x
  for (int i = N - 1; i >= 0; --i) { double s = x[i]; for (int k = i + 1; k < N; ++k) s -= L[k * N + i] * x[k]; x[i] = s / L[i * N + i]; }
}
// same with the reciprocal diagonal supplied (no divisions on the dependent chain)
template <int N> VC_HD void fwd_solve_inv(const double* L, const double* dinv, double* x) {
  for (int i = 0; i < N; ++i) { double s = x[i]; for (int k = 0; k < i; ++k) s -= L[i * N + k] * x[k]; x[i] = s * dinv[i]; }
}
template <int N> VC_HD void bwd_solve_inv(const double* L, const double* dinv, double* x) {
  for (int i = N - 1; i >= 0; --i) { double s = x[i]; for (int k = i + 1; k < N; ++k) s -= L[k * N + i] * x[k]; x[i] = s * dinv[i]; }
}

// Levenberg-Marquardt damping of one parameter (LevenbergMarquardtStrategy::ComputeStep with
// Jacobi scaling folded in): diag = clamp(h * scale2, 1e-6, 1e32), lambda = diag / (radius * scale2).
VC_HD double lm_clamped_diag(double h, double scale2) { return fmin(fmax(h * scale2, 1e-6), 1e32); }
VC_HD double jacobi_scale2(double h) { const double s = 1.0 / (1.0 + sqrt(h)); return s * s; }

}  // namespace vc
