// vc_imu_weights.hpp -- weight_sqrt_ of one IMU cost, for the HIP kernel k_imu_weights.
//
// Device-side counterpart of ViCalibrator::UpdateImuWeights (vicalibrator.h:723-799): propagate
// Sigma <- F Sigma F^T + G R G^T over the samples of the segment with the reference's hand-derived RK4
// Jacobians (ImuResidualT::IntegrateImu types.h:427-595, GetPoseDerivative :380-425, IntegratePose
// :330-378), project through d log(T_pred T_2^-1)/d(state) (dLog_dSE3 vicalibrator-utils.h:308-434,
// dt1t2_dt1 :261-274, dqx_dq :235-254, dq1q2_dq1/2 :215-230, dLog_dq :107-154, dqExp_dw :188-202),
// invert the 9x9 and take its principal square root (vicalibrator.h:783-796).
// The reference's approximations are part of its result and are kept: dqExp_dw is a low-order series,
// dk/dx ignores the scale factors (types.h:417-422).  State order [p(3), q(4: x y z w), v(3)].
#pragma once
#include "vc_imu.hpp"

namespace vc {

// C(m x n) = A(m x k) * B(k x n), row-major
VC_HD void mm(const double* A, const double* B, double* C, int m, int k, int n) {
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j) { double s = 0.0; for (int q = 0; q < k; ++q) s += A[i * k + q] * B[q * n + j]; C[i * n + j] = s; }
}
// C(m x n) = A(m x k) * B^T  (B is n x k)
VC_HD void mmt(const double* A, const double* B, double* C, int m, int k, int n) {
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j) { double s = 0.0; for (int q = 0; q < k; ++q) s += A[i * k + q] * B[j * k + q]; C[i * n + j] = s; }
}
VC_HD void w_dq1q2_dq2(const double* q1, double* m) {   // 4x4
  const double x = q1[0], y = q1[1], z = q1[2], w = q1[3];
  const double v[16] = {w, -z, y, x, z, w, -x, y, -y, x, w, z, -x, -y, -z, w};
  for (int i = 0; i < 16; ++i) m[i] = v[i];
}
VC_HD void w_dq1q2_dq1(const double* q2, double* m) {   // 4x4
  const double x = q2[0], y = q2[1], z = q2[2], w = q2[3];
  const double v[16] = {w, z, -y, x, -z, w, x, y, y, -x, w, z, -x, -y, -z, w};
  for (int i = 0; i < 16; ++i) m[i] = v[i];
}
VC_HD void w_dqexp_dw(const double* w, double* m) {     // 4x3
  const double t = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  const double s1 = t / 20 - 1, s2 = t * t / 48 - 0.5, s6 = t * t;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) m[i * 3 + j] = (i == j) ? (s1 * w[i] * w[i]) / 24 - s6 / 48 + 0.5 : (s1 * w[i] * w[j]) / 24;
  for (int j = 0; j < 3; ++j) m[9 + j] = (s2 * w[j]) / 2;
}
VC_HD void w_dqx_dq(const double* q, const double* vec, double* m) {   // 3x4
  const double x = vec[0], y = vec[1], z = vec[2], qx = q[0], qy = q[1], qz = q[2], qw = q[3];
  m[0] = 2 * qy * y + 2 * qz * z;            m[1] = 2 * qx * y - 4 * qy * x + 2 * qw * z;
  m[2] = 2 * qx * z - 2 * qw * y - 4 * qz * x; m[3] = 2 * qy * z - 2 * qz * y;
  m[4] = 2 * qy * x - 4 * qx * y - 2 * qw * z; m[5] = 2 * qx * x + 2 * qz * z;
  m[6] = 2 * qy * z + 2 * qw * x - 4 * qz * y; m[7] = 2 * qz * x - 2 * qx * z;
  m[8] = 2 * qz * x + 2 * qw * y - 4 * qx * z; m[9] = 2 * qz * y - 2 * qw * x - 4 * qy * z;
  m[10] = 2 * qy * y + 2 * qx * x;           m[11] = 2 * qx * y - 2 * qy * x;
}
// ---------------------------------------------------------------------------------------------------------------------
// Interval-parallel form of the same propagation (what k_imu_weights runs since round 3).
//
// The step maps of one sample interval, F = d(y_end)/d(y_start) (10 x 10) and G = d(y_end)/d(bias) (10 x 6), depend on the state
// the interval starts from only through its quaternion (GetPoseDerivative / IntegratePose never read p or v when they build
// dk_dx, dk_db, dy_dk, dy_dy: types.h:330-425) -- and that quaternion is q_start * Q with Q the product of the block's earlier
// interval deltas (vc_imu.hpp, delta form), which a prefix scan delivers without walking the block.  So every interval of a
// block forms its maps at once, each ONCE (not once per column-carrying lane), and the block is the short recurrence
// Sigma <- F Sigma F^T + G R G^T over them.  The reference's chain multiplies mostly zeros; with state order [p q v] it leaves
//        [ I   Fpq  fpv I ]         [ Gp_bg  Gp_ba ]        the four q columns and the three gyro-bias columns are the only ones
//    F = [ 0   Fqq    0   ]     G = [ Gq_bg    0   ]        that pass through the dense 4 x 4 quaternion blocks; the accelerometer
//        [ 0   Fvq    I   ]         [ Gv_bg  Gv_ba ]        bias columns are running sums of the stage rotations
// (p columns: identity; v columns: fpv = (dt / 6) * 6).  Same numbers as the sequential dense form (tests/host_harness/seq_weights.hpp: w_integrate_imu) up to rounding: every product the
// dense form takes with an exact 0 or 1 is dropped, nothing else is reordered inside a column.
// Interval record: f = [Fpq 3x4 | Fqq 4x4 | Fvq 3x4 | fpv] (41), g = [Gp_bg 3x3 | Gq_bg 4x3 | Gv_bg 3x3 | Gp_ba 3x3 | Gv_ba 3x3] (48),
// all row-major.  Symmetric 10 x 10 matrices (Sigma, G R G^T) travel as packed lower triangles: entry (a >= b) at a (a + 1) / 2 + b.
constexpr int kWMapF = 41, kWMapG = 48, kWMapQ = 55;
VC_HD int w_qidx(int a, int b) { return a >= b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a; }

// GetPoseDerivative (types.h:380-425) at the stage state qc: the rate k_w and the blocks of dk_dx / dk_db the columns meet
VC_HD void w_stage_derivative(const double* qc, const Meas<double>& z0, const Meas<double>& z1, const double* b, const double* sf,
                              double tau, double inv_dt, double* kw, double* R, double* Mw, double* Ma) {
  const double alpha = (z1.time - (z0.time + tau)) * inv_dt;
  double zg[3], za[3], u[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) { zg[i] = z0.w[i] * alpha + z1.w[i] * (1.0 - alpha); za[i] = z0.a[i] * alpha + z1.a[i] * (1.0 - alpha); }
  quat_to_R(qc, R);
#pragma unroll
  for (int i = 0; i < 3; ++i) u[i] = zg[i] * sf[i] + b[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) kw[i] = R[3 * i] * u[0] + R[3 * i + 1] * u[1] + R[3 * i + 2] * u[2];
  // (dqx_dq is linear in its vector argument: one evaluation at z + b stands for the reference's sum of two, types.h:417-422)
  { const double zb[3] = {zg[0] + b[0], zg[1] + b[1], zg[2] + b[2]}; w_dqx_dq(qc, zb, Mw); }
  { const double zb[3] = {za[0] + b[3], za[1] + b[4], za[2] + b[5]}; w_dqx_dq(qc, zb, Ma); }
}
// IntegratePose (types.h:330-378) from the interval's start quaternion with rate kw over h: the next stage state and the blocks
// of dy_dk (A E h on the quaternion rows) and dy_dy (D2 = dq1q2_dq2(rq), returned as rq)
VC_HD void w_stage_pose(const double* q_st, const double* kw, double h, double* rq, double* AE, double* q_next) {
  const double wdt[3] = {kw[0] * h, kw[1] * h, kw[2] * h};
  double A[16], E[12];
  so3_exp(wdt, rq);
  quat_mul(rq, q_st, q_next);
  w_dq1q2_dq1(q_st, A); w_dqexp_dw(wdt, E);
  mm(A, E, AE, 4, 4, 3);
}
// The maps of one interval from its start quaternion: the seven columns of [dy_dy0 | dy_db] that pass through the quaternion
// blocks (0..3: the quaternion columns of dy_dy0, seed e_j; 4..6: the gyro-bias columns of dy_db, seed 0, dk_db adds R's
// column to the rate rows) carried through the RK4 stages (types.h:427-595) together, the stage matrices formed once per
// stage; the accelerometer-bias columns are sums of the stage rotations.  Per column only what a later stage reads is kept:
// the quaternion rows Cq and the running k sums -- the velocity rows of a stage are h times the previous stage's k_a, so their
// contribution to the position rows' sum is added a stage ahead (weight wgt[s + 1] hh[s]), and the position rows themselves
// are never read before the end.  z1.time != z0.time (a zero-length interval is skipped by the caller, types.h:150-152).
VC_HD void w_interval_maps(const double* q_st, const Meas<double>& z0, const Meas<double>& z1, const double* b, const double* sf,
                           double* f, double* g) {
  constexpr int NC = 7;
  const double dt = z1.time - z0.time, inv_dt = 1.0 / dt;
  double Cq[NC][4], ktw[NC][3], kta[NC][3], ktp[NC][3], a_kta[9], a_ktp[9];
#pragma unroll
  for (int j = 0; j < NC; ++j) {
#pragma unroll
    for (int i = 0; i < 4; ++i) Cq[j][i] = (i == j) ? 1.0 : 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) { ktw[j][i] = 0.0; kta[j][i] = 0.0; ktp[j][i] = 0.0; }
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) { a_kta[i] = 0.0; a_ktp[i] = 0.0; }
  double ksum[3] = {0.0, 0.0, 0.0}, qc[4] = {q_st[0], q_st[1], q_st[2], q_st[3]};
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int stage = 0; stage < 5; ++stage) {
    // (stage constants by selection, not from a table: a dynamically indexed local array lives in scratch memory on the device)
    const double h = stage == 4 ? dt / 6.0 : (stage == 2 ? dt : dt * 0.5);
    double kw[3], Kw[NC][3];
    if (stage < 4) {
      double R[9], Mw[12], Ma[12];
      const double tau = stage == 0 ? 0.0 : (stage == 3 ? dt : dt / 2);
      w_stage_derivative(qc, z0, z1, b, sf, tau, inv_dt, kw, R, Mw, Ma);
      const double wg = (stage == 0 || stage == 3) ? 1.0 : 2.0;
      const double wp = stage == 3 ? 0.0 : (stage == 2 ? h : 2.0 * h);       // wgt[stage + 1] * hh[stage]
#pragma unroll
      for (int j = 0; j < NC; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          double a = 0.0, c = 0.0;
#pragma unroll
          for (int q = 0; q < 4; ++q) { a += Mw[i * 4 + q] * Cq[j][q]; c += Ma[i * 4 + q] * Cq[j][q]; }
          if (j >= 4) a += R[3 * i + (j - 4)];
          Kw[j][i] = a;
          ktw[j][i] += wg * a; kta[j][i] += wg * c; ktp[j][i] += wp * c;
        }
#pragma unroll
      for (int i = 0; i < 3; ++i) ksum[i] += wg * kw[i];
#pragma unroll
      for (int i = 0; i < 9; ++i) { a_kta[i] += wg * R[i]; a_ktp[i] += wp * R[i]; }
    } else {                                       // final combination: k1 + 2 k2 + 2 k3 + k4, h = dt / 6
#pragma unroll
      for (int j = 0; j < NC; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) Kw[j][i] = ktw[j][i];
#pragma unroll
      for (int i = 0; i < 3; ++i) kw[i] = ksum[i];
    }
    if (stage == 3) continue;                      // k4 only enters the sums
    double rq[4], AE[12], qn[4], D2[16];
    w_stage_pose(q_st, kw, h, rq, AE, qn);
    w_dq1q2_dq2(rq, D2);
#pragma unroll
    for (int j = 0; j < NC; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < 3; ++q) a += AE[i * 3 + q] * Kw[j][q];
        a *= h;
        if (j < 4) a += D2[i * 4 + j];
        Cq[j][i] = a;
      }
#pragma unroll
    for (int i = 0; i < 4; ++i) qc[i] = qn[i];
  }
  const double h6 = dt / 6.0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[i * 4 + j] = h6 * ktp[j][i]; f[28 + i * 4 + j] = h6 * kta[j][i]; }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) f[12 + i * 4 + j] = Cq[j][i];
  f[40] = h6 * 6.0;                                // the velocity columns' k_p sum is 1 + 2 + 2 + 1
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      g[i * 3 + j] = h6 * ktp[4 + j][i]; g[21 + i * 3 + j] = h6 * kta[4 + j][i];
      g[30 + i * 3 + j] = h6 * a_ktp[i * 3 + j]; g[39 + i * 3 + j] = h6 * a_kta[i * 3 + j];
    }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) g[9 + i * 3 + j] = Cq[4 + j][i];
}
// t = F s for one column s of a 10 x 10 matrix in [p q v] order
VC_HD void w_apply_F(const double* f, const double* s, double* t) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    double a = s[i], c = s[7 + i];
#pragma unroll
    for (int j = 0; j < 4; ++j) { a += f[i * 4 + j] * s[3 + j]; c += f[28 + i * 4 + j] * s[3 + j]; }
    t[i] = a + f[40] * s[7 + i]; t[7 + i] = c;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double a = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) a += f[12 + i * 4 + j] * s[3 + j];
    t[3 + i] = a;
  }
}
// Composition of two interval maps, o = Fb Fa (a first): the block form is closed under products.
VC_HD void w_map_compose(const double* fb, const double* fa, double* o) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) a += fb[12 + i * 4 + k] * fa[12 + k * 4 + j];
      o[12 + i * 4 + j] = a;
    }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double a = fa[i * 4 + j], c = fa[28 + i * 4 + j];
#pragma unroll
      for (int k = 0; k < 4; ++k) { a += fb[i * 4 + k] * fa[12 + k * 4 + j]; c += fb[28 + i * 4 + k] * fa[12 + k * 4 + j]; }
      o[i * 4 + j] = a + fb[40] * fa[28 + i * 4 + j];
      o[28 + i * 4 + j] = c;
    }
  o[40] = fa[40] + fb[40];
}
VC_HD void w_map_identity(double* f) {
#pragma unroll
  for (int e = 0; e < kWMapF; ++e) f[e] = 0.0;
  f[12] = f[17] = f[22] = f[27] = 1.0;
}
// out = F Q F^T for a symmetric Q, both as packed lower triangles (w_qidx).  Column by column: t = F Q[:, c], then
// out[i][j] += t[i] F[j][c] for i >= j with F's columns taken by structure (p: unit vector; q: dense; v: fpv e_p + e_v) --
// the sum over all columns is symmetric, so its lower triangle is all that is formed.
VC_HD void w_conj(const double* f, const double* Q, double* out) {
#pragma unroll
  for (int e = 0; e < kWMapQ; ++e) out[e] = 0.0;
#pragma unroll
  for (int c = 0; c < 10; ++c) {
    double s[10], t[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) s[i] = Q[w_qidx(i, c)];
    w_apply_F(f, s, t);
    if (c < 3) {
#pragma unroll
      for (int i = c; i < 10; ++i) out[w_qidx(i, c)] += t[i];
    } else if (c < 7) {
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        const double fj = j < 3 ? f[j * 4 + (c - 3)] : j < 7 ? f[12 + (j - 3) * 4 + (c - 3)] : f[28 + (j - 7) * 4 + (c - 3)];
#pragma unroll
        for (int i = j; i < 10; ++i) out[w_qidx(i, j)] += t[i] * fj;
      }
    } else {
#pragma unroll
      for (int i = c - 7; i < 10; ++i) out[w_qidx(i, c - 7)] += f[40] * t[i];
#pragma unroll
      for (int i = c; i < 10; ++i) out[w_qidx(i, c)] += t[i];
    }
  }
}
// One interval's share of the block's covariance, P G R G^T P^T (packed), with P the product of the maps after the interval:
// H = P G column by column -- the accelerometer-bias columns have no quaternion rows, so P only shifts them
// (p rows + fpv v rows) -- then sigma_g^2 H_bg H_bg^T + sigma_a^2 H_ba H_ba^T.  G R G^T itself is never formed.
VC_HD void w_noise_term(const double* P, const double* g, double sg2, double sa2, double* term) {
  double Hb[3][10], Ha[3][6];                      // Ha: rows p (0..2) and v (3..5)
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    double s[10];
#pragma unroll
    for (int i = 0; i < 3; ++i) { s[i] = g[i * 3 + j]; s[7 + i] = g[21 + i * 3 + j]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) s[3 + i] = g[9 + i * 3 + j];
    w_apply_F(P, s, Hb[j]);
#pragma unroll
    for (int i = 0; i < 3; ++i) { Ha[j][i] = g[30 + i * 3 + j] + P[40] * g[39 + i * 3 + j]; Ha[j][3 + i] = g[39 + i * 3 + j]; }
  }
#pragma unroll
  for (int a = 0; a < 10; ++a)
#pragma unroll
    for (int c = 0; c <= a; ++c) {
      double x = 0.0;
#pragma unroll
      for (int j = 0; j < 3; ++j) x += Hb[j][a] * Hb[j][c];
      x *= sg2;
      if ((a < 3 || a >= 7) && (c < 3 || c >= 7)) {
        const int ia = a < 3 ? a : a - 4, ic = c < 3 ? c : c - 4;
        double y = 0.0;
#pragma unroll
        for (int j = 0; j < 3; ++j) y += Ha[j][ia] * Ha[j][ic];
        x += sa2 * y;
      }
      term[a * (a + 1) / 2 + c] = x;
    }
}
// dLog_dSE3 (vicalibrator-utils.h:107-154, :308-434), arranged for the device (the only copy in the product; the formula as the
// reference writes it is kept with the tests, tests/host_harness/seq_weights.hpp: w_dlog_dse3): the arctangent of
// dLog_dq and of the SO3 logarithm is one value, tan(theta / 2) is |q_v| / |q_w| on the logarithm's regular branch (theta =
// 2 atan(|q_v| / q_w)), and the quotients that share a denominator share one reciprocal.  Same numbers up to rounding
// (tests/test_device_math_cpu.py compares the two).
VC_HD void w_dlog_dse3_lean(const double* T, double* dl) {
  const double qx = T[0], qy = T[1], qz = T[2], qw = T[3];
  const double v[3] = {qx, qy, qz};
  const double n2 = qx * qx + qy * qy + qz * qz, n = sqrt(n2);
  const double inv_w = 1.0 / qw, inv_w2 = inv_w * inv_w;
  const bool regular = !(n < kSophusEps) && !(fabs(qw) < kSophusEps);
  double at = 0.0, inv_n = 0.0;
  if (!(n < 1e-9) || regular) { inv_n = 1.0 / n; at = atan(n * inv_w); }
  double dw_dq[12];
  if (n < 1e-9) {
    const double s1 = 2 * n2, s2 = inv_w2 * inv_w, s3 = (3 * s1) * (inv_w2 * inv_w2) - 2 * inv_w2, s4 = 2 * inv_w;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) dw_dq[i * 4 + j] = (i == j) ? -4 * s2 * v[i] * v[i] + s4 - s1 * s2 : -4 * v[i] * v[j] * s2;
      dw_dq[i * 4 + 3] = v[i] * s3;
    }
  } else {
    const double s2 = 1 / (n2 * inv_w2 + 1), s3 = at, s5 = inv_n * inv_n, s4 = s5 * inv_n, s6 = inv_w, s7 = (2 * s3) * inv_n;
    const double s256 = s2 * s5 * s6, s34 = s3 * s4;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const double off = 2 * v[i] * v[j] * s256 - 2 * v[i] * v[j] * s34;
        dw_dq[i * 4 + j] = (i == j) ? s7 - 2 * v[i] * v[i] * s34 + 2 * v[i] * v[i] * s256 : off;
      }
      dw_dq[i * 4 + 3] = -(2 * v[i] * s2) * inv_w2;
    }
  }
  double c;
  if (n < kSophusEps) c = 2.0 * inv_w - 2.0 * n2 * (inv_w2 * inv_w);
  else if (fabs(qw) < kSophusEps) c = (qw > 0.0) ? 3.14159265358979323846 * inv_n : -3.14159265358979323846 * inv_n;
  else c = 2.0 * at * inv_n;
  const double theta = c * n;
  const double wx = c * qx, wy = c * qy, wz = c * qz, x = T[4], y = T[5], z = T[6];
  const bool small = fabs(theta) < kSophusEps;
  // tan(|theta| / 2): on the regular branch theta = 2 atan(n / qw)
  double s2t = 0.0;
  if (!small) s2t = regular ? n * fabs(inv_w) : tan(fabs(theta) / 2.0);
  const double sgn = theta < 0.0 ? -1.0 : 1.0;
  const double cc = small ? 1.0 / 12.0 : (1.0 - theta / (2.0 * (sgn * s2t))) / (theta * theta);
  const double O[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
  double O2[9];
  mm(O, O, O2, 3, 3, 3);
#pragma unroll
  for (int i = 0; i < 42; ++i) dl[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) dl[i * 7 + j] = (i == j ? 1.0 : 0.0) - 0.5 * O[3 * i + j] + cc * O2[3 * i + j];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) dl[(3 + i) * 7 + 3 + j] = dw_dq[i * 4 + j];
  double dw[9];
  if (small) {
    const double d12 = 1. / 12, d6 = 1. / 6.;
    dw[0] = d12 * (wy * y + wz * z); dw[1] = d12 * wx * y - d6 * wy * x - 0.5 * z; dw[2] = 0.5 * y - d6 * wz * x + d12 * wx * z;
    dw[3] = 0.5 * z + d12 * wy * x - d6 * wx * y; dw[4] = d12 * (wx * x + wz * z); dw[5] = d12 * wy * z - d6 * wz * y - 0.5 * x;
    dw[6] = d12 * wz * x - d6 * wx * z - 0.5 * y; dw[7] = 0.5 * x + d12 * wz * y - d6 * wy * z; dw[8] = d12 * ((wx * x) * (wy * y));
  } else {
    const double s1 = wx * wx + wy * wy + wz * wz, rs1 = sqrt(s1), s2 = s2t, inv_s2 = 1.0 / s2, inv_rs1 = 1.0 / rs1;
    const double s3 = rs1 * (0.5 * inv_s2) - 1;
    const double k1 = 0.5 * inv_rs1 * inv_s2, k2 = (s2 * s2 + 1) * (0.25 * inv_s2 * inv_s2);
    const double s4 = wz * k1 - wz * k2, s5 = wy * k1 - wy * k2, s6 = wx * k1 - wx * k2;
    const double s7 = inv_rs1 * inv_rs1, s8 = s7 * s7, s9 = wx * wx + wy * wy, s10 = wx * wx + wz * wz, s11 = wy * wy + wz * wz;
    const double s12 = 2 * s3 * s8 * wx * wy * wz;
    const double s13 = -2 * s3 * s8 * wy * wz * wz + s4 * s7 * wy * wz + s3 * s7 * wy;
    const double s14 = -2 * s3 * s8 * wx * wz * wz + s4 * s7 * wx * wz + s3 * s7 * wx;
    const double s15 = -2 * s3 * s8 * wz * wy * wy + s5 * s7 * wz * wy + s3 * s7 * wz;
    const double s16 = -2 * s3 * s8 * wz * wx * wx + s6 * s7 * wz * wx + s3 * s7 * wz;
    const double s17 = -2 * s3 * s8 * wx * wy * wy + s5 * s7 * wx * wy + s3 * s7 * wx;
    const double s18 = -2 * s3 * s8 * wy * wx * wx + s6 * s7 * wy * wx + s3 * s7 * wy;
    const double s19 = 2 * s3 * s7 * wy, s20 = 2 * s3 * s7 * wx;
    dw[0] = x * (s6 * s7 * s11 - 2 * s3 * s8 * s11 * wx) - s18 * y - s16 * z;
    dw[1] = x * (s19 + s5 * s7 * s11 - 2 * s3 * s8 * s11 * wy) - s17 * y - z * (s5 * s7 * wx * wz - 2 * s3 * s8 * wx * wy * wz + 0.5);
    dw[2] = x * (s4 * s7 * s11 + 2 * s3 * s7 * wz - 2 * s3 * s8 * s11 * wz) - s14 * z + y * (s12 - s4 * s7 * wx * wy + 0.5);
    dw[3] = y * (s20 + s6 * s7 * s10 - 2 * s3 * s8 * s10 * wx) - s18 * x + z * (s12 - s6 * s7 * wy * wz + 0.5);
    dw[4] = y * (s5 * s7 * s10 - 2 * s3 * s8 * s10 * wy) - s17 * x - s15 * z;
    dw[5] = y * (s4 * s7 * s10 + 2 * s3 * s7 * wz - 2 * s3 * s8 * s10 * wz) - s13 * z - x * (s4 * s7 * wx * wy - s12 + 0.5);
    dw[6] = z * (s20 + s6 * s7 * s9 - 2 * s3 * s8 * s9 * wx) - s16 * x - y * (s6 * s7 * wy * wz - s12 + 0.5);
    dw[7] = z * (s19 + s5 * s7 * s9 - 2 * s3 * s8 * s9 * wy) - s15 * y + x * (s12 - s5 * s7 * wx * wz + 0.5);
    dw[8] = z * (s4 * s7 * s9 - 2 * s3 * s8 * s9 * wz) - s14 * x - s13 * y;
  }
  double blk[12];
  mm(dw, dw_dq, blk, 3, 3, 4);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) dl[i * 7 + 3 + j] = blk[i * 4 + j];
}
// Row i (< 6) of J67 = dLog_dSE3(T_pred T2^-1) dt1t2_dt1(T_pred, T2^-1) (vicalibrator.h:762-775); dt1t2_dt1 = [I3, dqx_dq(q, t);
// 0, dq1q2_dq1] is sparse, so the product is taken row by row.  rel / t2w as w_projection_prepare builds them.
VC_HD void w_projection_prepare(const double* q_pred, const double* p_pred, const double* T2, double* rel, double* t2w) {
  const double qc[4] = {-T2[0], -T2[1], -T2[2], T2[3]}, nt[3] = {-T2[4], -T2[5], -T2[6]};
  double tr[3];
  quat_rotate(qc, nt, t2w + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) t2w[i] = qc[i];
  quat_mul(q_pred, qc, rel);
  const double nrm = sqrt(rel[0] * rel[0] + rel[1] * rel[1] + rel[2] * rel[2] + rel[3] * rel[3]);
#pragma unroll
  for (int i = 0; i < 4; ++i) rel[i] /= nrm;
  quat_rotate(q_pred, t2w + 4, tr);
#pragma unroll
  for (int i = 0; i < 3; ++i) rel[4 + i] = p_pred[i] + tr[i];
}
VC_HD void w_projection_row(const double* dl, const double* m34, const double* m44, int i, double* row /* 7 */) {
#pragma unroll
  for (int jj = 0; jj < 3; ++jj) row[jj] = dl[i * 7 + jj];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    double a = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) a += dl[i * 7 + k] * m34[k * 4 + jj];
#pragma unroll
    for (int k = 0; k < 4; ++k) a += dl[i * 7 + 3 + k] * m44[k * 4 + jj];
    row[3 + jj] = a;
  }
}

#if !defined(__HIP_DEVICE_COMPILE__)
// Host emulation of k_imu_weights, lane by lane in a loop (tests/host_harness): interval deltas from the identity, their
// prefix products, the maps at each interval's start quaternion; then, parallel in time as on the device,
//     Sigma_end = sum_k P_k Q_k P_k^T ,   P_k = F_n ... F_{k+1}  (suffix products of the later maps),
// which is the recurrence Sigma <- F Sigma F^T + Q unrolled from Sigma = 0; the projection and the factor the kernel stores --
// W = L^-T with J Sigma J^T = L L^T (W W^T = the reference's information matrix).  Returns 0 and leaves W untouched for an
// empty sample range (vicalibrator.h:731-733) or a projection that is not positive definite.
inline int imu_weight_factor_intervals(const ImuView& buf, double t_start, double t_end, double toff, const double* T1, const double* v1,
                                       const double* T2, const double* b, const double* sf, const double* gdir, double gyro_sigma,
                                       double accel_sigma, double* W) {
  const ImuRange rg = imu_range(buf, t_start, t_end, toff);
  if (!rg.valid) return 0;
  double gw[3];
  imu_gravity(gdir, gw);
  const int n_meas = (rg.k1 - rg.k0 + 1) + 2, n_int = n_meas - 1;
  const double sg2 = gyro_sigma * gyro_sigma, sa2 = accel_sigma * accel_sigma, g0[3] = {0.0, 0.0, 0.0};
  DeltaAcc<double> acc;
  for (int k = 0; k < 3; ++k) { acc.q[k] = 0.0; acc.p[k] = 0.0; acc.v[k] = 0.0; }
  acc.q[3] = 1.0; acc.t = 0.0;
  constexpr int kRec = kWMapF + kWMapG;
  double* maps = new double[(size_t)(n_int + 1) * kRec];
  for (int m = 1; m <= n_int; ++m) {
    double* fq = maps + (size_t)m * kRec;
    Meas<double> z0, z1;
    imu_range_get(buf, rg, toff, t_start, t_end, m - 1, &z0);
    imu_range_get(buf, rg, toff, t_start, t_end, m, &z1);
    if (z1.time == z0.time) {                                    // skipped (types.h:150-152): the identity map, no noise
      w_map_identity(fq);
      for (int e = 0; e < kWMapG; ++e) fq[kWMapF + e] = 0.0;
      continue;
    }
    double q_st[4];
    quat_mul(T1, acc.q, q_st);                                   // the state this interval starts from: q_start * (prefix)
    w_interval_maps(q_st, z0, z1, b, sf, fq, fq + kWMapF);
    PoseV<double> d;
    for (int k = 0; k < 3; ++k) { d.q[k] = 0.0; d.p[k] = 0.0; d.v[k] = 0.0; }
    d.q[3] = 1.0;
    imu_rk4_step(&d, z0, z1, b, sf, g0);
    imu_delta_append(&acc, d, z1.time - z0.time);
  }
  double Sp[kWMapQ], P[kWMapF];                                  // packed Sigma; suffix product of the maps after interval m
  for (int e = 0; e < kWMapQ; ++e) Sp[e] = 0.0;
  w_map_identity(P);
  for (int m = n_int; m >= 1; --m) {
    const double* fq = maps + (size_t)m * kRec;
    double term[kWMapQ], Pn[kWMapF];
    w_noise_term(P, fq + kWMapF, sg2, sa2, term);
    for (int e = 0; e < kWMapQ; ++e) Sp[e] += term[e];
    w_map_compose(P, fq, Pn);                                    // P_{m-1} = P_m F_m
    for (int e = 0; e < kWMapF; ++e) P[e] = Pn[e];
  }
  delete[] maps;
  double Sigma[100];
  for (int i = 0; i < 10; ++i) for (int c = 0; c < 10; ++c) Sigma[i * 10 + c] = Sp[w_qidx(i, c)];
  // end state from the block's delta (vc_imu.hpp)
  double q_end[4], p_end[3], rp[3];
  quat_mul(T1, acc.q, q_end);
  tq_rotate(T1, acc.p, rp);
  const double ht2 = 0.5 * (acc.t * acc.t);
  for (int i = 0; i < 3; ++i) p_end[i] = ((T1[4 + i] + v1[i] * acc.t) - gw[i] * ht2) + rp[i];
  double rel[7], t2w[7], dl[42], m34[12], m44[16], J[90];
  w_projection_prepare(q_end, p_end, T2, rel, t2w);
  w_dlog_dse3_lean(rel, dl);
  w_dqx_dq(q_end, t2w + 4, m34);
  w_dq1q2_dq1(t2w, m44);
  for (int i = 0; i < 90; ++i) J[i] = 0.0;
  for (int i = 0; i < 6; ++i) w_projection_row(dl, m34, m44, i, J + i * 10);
  J[6 * 10 + 7] = J[7 * 10 + 8] = J[8 * 10 + 9] = 1.0;
  double JS[90], Pm[81];
  mm(J, Sigma, JS, 9, 10, 10);
  mmt(JS, J, Pm, 9, 10, 9);
  double L[81], inv_d[9];
  for (int i = 0; i < 81; ++i) L[i] = Pm[i];
  for (int cc = 0; cc < 9; ++cc) {
    double dgn = L[cc * 9 + cc];
    for (int k = 0; k < cc; ++k) dgn -= L[cc * 9 + k] * L[cc * 9 + k];
    if (!(dgn > 0.0)) return 0;
    const double id = 1.0 / sqrt(dgn);
    L[cc * 9 + cc] = dgn * id; inv_d[cc] = id;
    for (int i = cc + 1; i < 9; ++i) {
      double a = L[i * 9 + cc];
      for (int k = 0; k < cc; ++k) a -= L[i * 9 + k] * L[cc * 9 + k];
      L[i * 9 + cc] = a * id;
    }
  }
  for (int c = 0; c < 9; ++c) {                                  // column c of X = L^-1; W[c][i] = X[i][c]
    double x[9];
    for (int i = 0; i < 9; ++i) {
      double a = (i == c) ? 1.0 : 0.0;
      for (int k = 0; k < i; ++k) a -= L[i * 9 + k] * x[k];
      x[i] = (i >= c) ? a * inv_d[i] : 0.0;
    }
    for (int i = 0; i < 9; ++i) W[c * 9 + i] = x[i];
  }
  return 1;
}
#endif

}  // namespace vc
