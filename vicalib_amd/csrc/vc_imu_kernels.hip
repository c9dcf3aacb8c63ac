// vc_imu_kernels.hip -- inertial stages of the LM engine (gfx950, wave64).
//
// With IMU blocks in the problem (SetupProblem, vicalibrator.h:607-637, :651-655) consecutive frames are
// coupled, so the per-frame elimination of the vision-only path becomes the factorisation of a
// block-tridiagonal chain (9 x 9 blocks: pose 6 + velocity 3) bordered by the shared parameters.
//
//  k_imu_jac       one wavefront per IMU block; lane d carries derivative direction d of the 35 global
//                  parameters through the RK4 preintegration (vc_imu.hpp) -- the lane-parallel form of
//                  ceres::Jet<double,35>; then local-parameterisation Jacobians, Cauchy(100) weight and
//                  the 33 x 33 weighted J^T J / J^T r of the block
//  k_imu_res       thread per block: residual cost at a state (trial-point evaluation)
//  k_imu_weights   thread per block: UpdateImuWeights (vicalibrator.h:723-799, vc_imu_weights.hpp)
//  k_chain_init    wavefront per frame: 9 x 9 diagonal block (visual tiles + two IMU blocks), coupling to the
//                  next frame, dense border row W (9 x D) and gradient; damping; per-chunk sums of the
//                  camera Gram blocks and of the IMU shared-parameter block
//  k_cr_elim / k_cr_update   one level of block cyclic reduction: odd frames are eliminated (L, P = L^-1 B_prev^T,
//                  Q = L^-1 B_self, Y = L^-1 [W | g]), even frames absorb their two eliminated neighbours
//  k_chain_gram    sum over all frames of [Y | z]^T [Y | z] on the matrix pipe (v_mfma_f64_16x16x4_f64)
//  k_cr_back       back-substitution, one level per launch;  k_frame_update  trial poses / velocities
#include <hip/hip_runtime.h>
#include <algorithm>
#include "vc_math.hpp"
#include "vc_imu.hpp"
#include "vc_imu_weights.hpp"
#include "vc_device.h"
#include "vc_kutil.hpp"

namespace vc {

__device__ __forceinline__ ImuView imu_view(const DevView& v) { ImuView b = {v.imu_t, v.imu_w, v.imu_a, v.n_imu}; return b; }

// ------------------------------------------------------------------------------------------ IMU Jacobian
constexpr int kImuJacLds = 35 * 9 + 33 * 9 + 16;
// Two IMU blocks per wavefront: 32 lanes carry the 32 derivative directions that need the dual propagation (the three
// directions of the later frame's velocity do not -- d r / d v2 = -W^T rows 6..8, written directly), so a block fits a
// half wave and the kernel, bound by per-lane latency at one wave per SIMD, needs half the waves.
__global__ __launch_bounds__(256) void k_imu_jac(DevView v, int wr) {
  __shared__ double sh[8 * kImuJacLds];
  const Ctrl* ct = v.ctrl;
  if (ct->done || !ct->need_lin) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int half = lane >> 5, l = lane & 31;
  const int n_blocks = v.n_frames - 1;
  const int s_raw = (blockIdx.x * 4 + wave) * 2 + half;       // block s couples frames s -> s+1
  const bool exists = s_raw < n_blocks;
  const int s = exists ? s_raw : n_blocks - 1;                 // a half past the end shadows the last block and stores nothing
  const int cur = ct->cur, j = s + 1;
  double* Jg = sh + (wave * 2 + half) * kImuJacLds;            // [35][9] global-parameter partials
  double* Jl = Jg + 35 * 9;                                    // [33][9] local columns: cur9 | prev9 | imu15
  const double* T2 = v.poses[cur] + (size_t)j * kPoseStride;
  const double* T1 = v.poses[cur] + (size_t)(j - 1) * kPoseStride;
  const double* v2 = v.vel[cur] + (size_t)j * 4;
  const double* v1 = v.vel[cur] + (size_t)(j - 1) * 4;
  const double* im = v.imus[cur];
  const double* wq = v.wsqrtb[wr] + (size_t)s * 81;
  double r[9], dr[9];
  const ImuView buf = imu_view(v);
  const int dir = (l < 14) ? l : l + 3;                        // global directions 0..13 and 17..34
  imu_block_direction(buf, v.frame_time[j - 1], v.frame_time[j], wq, v.rotation_only, T2, T1, v2, v1, im, im + 2,
                      im + 8, im[14], dir, r, dr);
#pragma unroll
  for (int k = 0; k < 9; ++k) Jg[dir * 9 + k] = dr[k];
  if (l < 3) {                                                 // later frame's velocity: r = W^T raw, raw[6 + l] = v_pred - v2
    const bool valid = imu_range(buf, v.frame_time[j - 1], v.frame_time[j], im[14]).valid;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const bool zeroed = v.rotation_only && (k < 3 || k >= 6);
      Jg[(14 + l) * 9 + k] = (valid && !zeroed) ? -wq[(6 + l) * 9 + k] : 0.0;
    }
  }
  wave_lds_sync();
  for (int col = l; col < 33; col += 32) {
    double cv[9];
    if (col < 6 || (col >= 9 && col < 15)) {
      const bool is_cur = col < 6;
      const int c = is_cur ? col : col - 9;
      double P[42];
      local_jac_se3(is_cur ? T2 : T1, P);
      const double* src = Jg + (is_cur ? 0 : 7) * 9;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        double a = 0.0;
#pragma unroll
        for (int i = 0; i < 7; ++i) a += src[i * 9 + k] * P[i * 6 + c];
        cv[k] = a;
      }
    } else {
      // cur velocity 6..8 <- global 14..16 ; prev velocity 15..17 <- global 17..19 ; imu 18..32 <- global 20..34
      const int gidx = (col < 9) ? 14 + (col - 6) : (col < 18) ? 17 + (col - 15) : 20 + (col - 18);
#pragma unroll
      for (int k = 0; k < 9; ++k) cv[k] = Jg[gidx * 9 + k];
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) Jl[col * 9 + k] = cv[k];
  }
  wave_lds_sync();
  double ss = 0.0;
#pragma unroll
  for (int k = 0; k < 9; ++k) ss += r[k] * r[k];
  double rho, rho1;
  loss_cauchy100(ss, &rho, &rho1);
  const double w = ct->imu_mult * rho1;
  if (!exists) return;
  double* H = v.segH + (size_t)s * (33 * 33);
  for (int e = l; e < 33 * 33; e += 32) {
    const int a = e / 33, bb = e % 33;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) acc += Jl[a * 9 + k] * Jl[bb * 9 + k];
    H[e] = w * acc;
  }
  for (int col = l; col < 33; col += 32) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) acc += Jl[col * 9 + k] * r[k];
    v.segg[(size_t)s * 33 + col] = w * acc;
  }
  if (l == 0) v.seg_cost[s] = ct->imu_mult * rho;
}

// residual cost of every IMU block at a state: sel 2 = accepted buffer, 3 = trial buffer
__global__ __launch_bounds__(64) void k_imu_res(DevView v, int sel, int wr) {
  const Ctrl* ct = v.ctrl;
  if (ct->done) return;
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s >= v.n_frames - 1) return;
  const int st = (sel == 3) ? 1 - ct->cur : ct->cur, j = s + 1;
  const double* im = v.imus[st];
  double r[9];
  imu_residual<double>(imu_view(v), v.frame_time[j - 1], v.frame_time[j], v.wsqrtb[wr] + (size_t)s * 81, v.rotation_only,
                       v.poses[st] + (size_t)j * kPoseStride, v.poses[st] + (size_t)(j - 1) * kPoseStride, v.vel[st] + (size_t)j * 4,
                       v.vel[st] + (size_t)(j - 1) * 4, im, im + 2, im + 8, im[14], r);
  double ss = 0.0;
  for (int k = 0; k < 9; ++k) ss += r[k] * r[k];
  double rho, rho1;
  loss_cauchy100(ss, &rho, &rho1);
  v.seg_trial[s] = ct->imu_mult * rho;
}

// UpdateImuWeights from the accepted state (vicalibrator.h:723-799): covariance propagation along the block's samples
// (w_step_cols below), projection through the residual's Jacobian, factorisation.
// The weight is stored as W = L^-T with  J Sigma J^T = L L^T  (Cholesky): W W^T = (J Sigma J^T)^-1 exactly as
// for the reference's symmetric square root (vicalibrator.h:783-796), and cost, gradient and Gauss-Newton
// Hessian of the block depend on W only through W W^T -- same optimisation, no 9x9 eigen-decomposition.
// Per-block LDS of the weight update: M / T are the 10 x 16 images used to transpose / broadcast the step's
// sensitivity matrices (two round trips per IMU step), the rest serves the final 9 x 10 projection.
struct WLds { double M[160], T[160], tmp[100], Sigma[100], J[90], P[81]; };

// One RK4 step of the covariance propagation (types.h:427-595) with lane = column: lane c (< 16) carries column c of
// [dy_dy0 (10) | dy_db (6)] through the four stages in registers.  The stage matrices of the reference's hand-derived
// chain (dk_dx 9 x 10, dk_db 9 x 6, dy_dk 10 x 9, dy_dy 10 x 10) are sparse with a handful of small dense blocks; every
// lane forms those blocks from the (replicated) state and applies them to its own column, so the whole step needs no
// communication until Sigma <- F Sigma F^T + G R G^T, which goes through two LDS images.  Sc: column c of Sigma (c < 10).
__device__ void w_step_cols(WLds& L, WState* st, const Meas<double>& z0, const Meas<double>& z1, const double* b, const double* sf,
                            const double* g, double sg2, double sa2, int c, double* Sc) {
  const double dt = z1.time - z0.time;
  if (dt == 0) return;
  const double tau[4] = {0.0, dt / 2, dt / 2, dt}, hh[4] = {dt * 0.5, dt * 0.5, dt, dt / 6.0}, wgt[4] = {1.0, 2.0, 2.0, 1.0};
  double Y0[10], Yc[10], kt[9], ksum[9];
#pragma unroll
  for (int i = 0; i < 10; ++i) { Y0[i] = (i == c) ? 1.0 : 0.0; Yc[i] = Y0[i]; }     // dy_dy0 = I, dy_db = 0 at the step start
#pragma unroll
  for (int i = 0; i < 9; ++i) { kt[i] = 0.0; ksum[i] = 0.0; }
  WState cur = *st, y = *st;
#pragma unroll 1
  for (int stage = 0; stage < 5; ++stage) {
    double kcol[9], kv[9];
    if (stage < 4) {
      // k = f(cur) (GetPoseDerivative, types.h:380-425) and this lane's column of dk/d[y0 | b]
      const double alpha = (z1.time - (z0.time + tau[stage])) / (z1.time - z0.time);
      double zg[3], za[3], u[3], o[3], R[9], m1[12], m2[12];
#pragma unroll
      for (int i = 0; i < 3; ++i) { zg[i] = z0.w[i] * alpha + z1.w[i] * (1.0 - alpha); za[i] = z0.a[i] * alpha + z1.a[i] * (1.0 - alpha); }
      quat_to_R(cur.q, R);
#pragma unroll
      for (int i = 0; i < 3; ++i) { kv[i] = cur.v[i]; u[i] = zg[i] * sf[i] + b[i]; }
#pragma unroll
      for (int i = 0; i < 3; ++i) kv[3 + i] = R[3 * i] * u[0] + R[3 * i + 1] * u[1] + R[3 * i + 2] * u[2];
#pragma unroll
      for (int i = 0; i < 3; ++i) u[i] = za[i] * sf[3 + i] + b[3 + i];
      quat_rotate(cur.q, u, o);
#pragma unroll
      for (int i = 0; i < 3; ++i) kv[6 + i] = o[i] - g[i];
      // dk_dx: rows 0-2 = d/dv, rows 3-5 = (dqx_dq(q, zg) + dqx_dq(q, bg)) on the quaternion, rows 6-8 likewise with za, ba
#pragma unroll
      for (int i = 0; i < 3; ++i) kcol[i] = Yc[7 + i];
      w_dqx_dq(cur.q, zg, m1); w_dqx_dq(cur.q, b, m2);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) a += (m1[i * 4 + q] + m2[i * 4 + q]) * Yc[3 + q];
        kcol[3 + i] = a;
      }
      w_dqx_dq(cur.q, za, m1); w_dqx_dq(cur.q, b + 3, m2);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) a += (m1[i * 4 + q] + m2[i * 4 + q]) * Yc[3 + q];
        kcol[6 + i] = a;
      }
      // dk_db: R in rows 3-5 for the gyro bias columns (10..12), in rows 6-8 for the accelerometer bias columns (13..15)
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          kcol[3 + i] += (c == 10 + q) ? R[3 * i + q] : 0.0;
          kcol[6 + i] += (c == 13 + q) ? R[3 * i + q] : 0.0;
        }
#pragma unroll
      for (int i = 0; i < 9; ++i) { kt[i] += wgt[stage] * kcol[i]; ksum[i] += wgt[stage] * kv[i]; }
    } else {
#pragma unroll
      for (int i = 0; i < 9; ++i) { kcol[i] = kt[i]; kv[i] = ksum[i]; }      // final combination: k1 + 2 k2 + 2 k3 + k4, h = dt / 6
    }
    if (stage == 3) continue;                     // k4 only enters the sum
    // y = IntegratePose(st, k, h) (types.h:330-378) and this lane's column of dy/d[y0 | b] = dy_dk kcol + dy_dy Y0
    const double h = hh[stage == 4 ? 3 : stage];
    const double wdt[3] = {kv[3] * h, kv[4] * h, kv[5] * h};
    double rq[4], A[16], E[12], AE[12], D2[16];
    so3_exp(wdt, rq);
#pragma unroll
    for (int i = 0; i < 3; ++i) { y.p[i] = st->p[i] + kv[i] * h; y.v[i] = st->v[i] + kv[6 + i] * h; }
    quat_mul(rq, st->q, y.q);
    w_dq1q2_dq1(st->q, A); w_dqexp_dw(wdt, E);
    mm(A, E, AE, 4, 4, 3);
    w_dq1q2_dq2(rq, D2);
#pragma unroll
    for (int i = 0; i < 3; ++i) { Yc[i] = h * kcol[i] + Y0[i]; Yc[7 + i] = h * kcol[6 + i] + Y0[7 + i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      double a = 0.0;
#pragma unroll
      for (int q = 0; q < 3; ++q) a += AE[i * 3 + q] * kcol[3 + q];
      a *= h;
#pragma unroll
      for (int q = 0; q < 4; ++q) a += D2[i * 4 + q] * Y0[3 + q];
      Yc[3 + i] = a;
    }
    cur = y;
  }
  // ---- Sigma <- F Sigma F^T + G R G^T:  F = columns 0..9, G = columns 10..15 of the lanes ------------------------------
#pragma unroll
  for (int i = 0; i < 10; ++i) L.M[i * 16 + c] = Yc[i];
  wave_lds_sync();
  double T[10], Frow[16];
#pragma unroll
  for (int i = 0; i < 10; ++i) T[i] = 0.0;
#pragma unroll
  for (int r = 0; r < 16; ++r) Frow[r] = (c < 10) ? L.M[c * 16 + r] : 0.0;          // row c of [F | G]
#pragma unroll
  for (int q = 0; q < 10; ++q) {
#pragma unroll
    for (int i = 0; i < 10; ++i) T[i] += L.M[i * 16 + q] * Sc[q];                  // column c of F Sigma (broadcast reads)
    __builtin_amdgcn_sched_barrier(0);          // ten loads in flight at a time, not a hundred (register pressure)
  }
#pragma unroll
  for (int i = 0; i < 10; ++i) L.T[i * 16 + c] = T[i];
  wave_lds_sync();
  double Sn[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) Sn[i] = 0.0;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
#pragma unroll
    for (int i = 0; i < 10; ++i) Sn[i] += L.T[i * 16 + r] * Frow[r];                // (F Sigma) F^T, column c
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const double rw = (q < 3 ? sg2 : sa2) * Frow[10 + q];
#pragma unroll
    for (int i = 0; i < 10; ++i) Sn[i] += L.M[i * 16 + 10 + q] * rw;                // G R G^T, column c
    __builtin_amdgcn_sched_barrier(0);
  }
  wave_lds_sync();
#pragma unroll
  for (int i = 0; i < 10; ++i) Sc[i] = (c < 10) ? Sn[i] : 0.0;
  *st = y;
}

// Four IMU blocks per wavefront: the propagation only uses 16 lanes (one per column of [dy_dy0 | dy_db]), and the kernel is
// bound by per-lane instruction latency at one wave per SIMD (it needs the whole register file), so packing four 16-lane
// groups into a wave quarters the number of waves.  Everything below is per group: its own LDS record, 16-lane loops,
// stores predicated on the group's state; no early return (the groups of a wave finish together).
__global__ __launch_bounds__(256) void k_imu_weights(DevView v, int wr) {
  extern __shared__ __attribute__((aligned(16))) double w_lds[];
  const Ctrl* ct = v.ctrl;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int grp = lane >> 4, c = lane & 15;      // c: column of [dy_dy0 | dy_db] this lane carries within its group
  const int n_blocks = v.n_frames - 1;
  const int s_raw = (blockIdx.x * 4 + wave) * 4 + grp;
  const bool exists = s_raw < n_blocks;
  const int s = exists ? s_raw : n_blocks - 1;   // groups past the end shadow the last block and store nothing
  WLds& L = *reinterpret_cast<WLds*>(w_lds + (size_t)(wave * 4 + grp) * (sizeof(WLds) / sizeof(double)));
  // the buffer being written starts as a copy of the current weights: blocks that keep their weight (no samples, singular
  // projection) and passes queued behind a finished solve leave a consistent buffer behind
  if (exists) for (int e = c; e < 81; e += 16) v.wsqrtb[1 - wr][(size_t)s * 81 + e] = v.wsqrtb[wr][(size_t)s * 81 + e];
  if (ct->done || !v.weights_on) return;        // wave-uniform
  const int st = ct->cur, j = s + 1;
  const double* im = v.imus[st];
  double T1[7], T2[7], b[6], sf[6], g2[2], gw[3];
#pragma unroll
  for (int i = 0; i < 7; ++i) { T1[i] = v.poses[st][(size_t)(j - 1) * kPoseStride + i]; T2[i] = v.poses[st][(size_t)j * kPoseStride + i]; }
#pragma unroll
  for (int i = 0; i < 6; ++i) { b[i] = im[2 + i]; sf[i] = im[8 + i]; }
  g2[0] = im[0]; g2[1] = im[1];
  const double toff = im[14];
  const double t_start = v.frame_time[j - 1], t_end = v.frame_time[j];
  const ImuView buf = imu_view(v);
  const ImuRange rg = imu_range(buf, t_start, t_end, toff);
  bool live = exists && rg.valid;                // an empty range keeps its current weight (vicalibrator.h:731-733)
  imu_gravity(g2, gw);
  WState sx;
#pragma unroll
  for (int i = 0; i < 4; ++i) sx.q[i] = T1[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) { sx.p[i] = T1[4 + i]; sx.v[i] = v.vel[st][(size_t)(j - 1) * 4 + i]; }
  const double sg2 = v.gyro_sigma * v.gyro_sigma, sa2 = v.accel_sigma * v.accel_sigma;
  const int n_meas = live ? (rg.k1 - rg.k0 + 1) + 2 : 0;
  int n_max = n_meas;                            // the wave runs to its longest group
#pragma unroll
  for (int o = 32; o >= 16; o >>= 1) n_max = max(n_max, __shfl_xor(n_max, o, 64));
  double Sc[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) Sc[i] = 0.0;
  Meas<double> z0, z1;
  if (live) imu_range_get(buf, rg, toff, t_start, t_end, 0, &z0);
  for (int m = 1; m < n_max; ++m) {
    if (m < n_meas) {
      imu_range_get(buf, rg, toff, t_start, t_end, m, &z1);
      w_step_cols(L, &sx, z0, z1, b, sf, gw, sg2, sa2, c, Sc);
      z0 = z1;
    }
  }
  if (c < 10) {
#pragma unroll
    for (int i = 0; i < 10; ++i) L.Sigma[i * 10 + c] = Sc[i];
  }
  // J = dLog_dSE3(T_pred T2^-1) dt1t2_dt1(T_pred, T2^-1), velocity identity appended (9 x 10)
  const double qc[4] = {-T2[0], -T2[1], -T2[2], T2[3]}, nt[3] = {-T2[4], -T2[5], -T2[6]};
  double t2w[7], rel[7], tr[3];
  quat_rotate(qc, nt, t2w + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) t2w[i] = qc[i];
  quat_mul(sx.q, qc, rel);
  const double nrm = sqrt(rel[0] * rel[0] + rel[1] * rel[1] + rel[2] * rel[2] + rel[3] * rel[3]);
#pragma unroll
  for (int i = 0; i < 4; ++i) rel[i] /= nrm;
  quat_rotate(sx.q, t2w + 4, tr);
#pragma unroll
  for (int i = 0; i < 3; ++i) rel[4 + i] = sx.p[i] + tr[i];
  for (int e = c; e < 90; e += 16) L.J[e] = 0.0;
  wave_lds_sync();
  {
    // J67 = dLog_dSE3 * dt1t2_dt1 with dt1t2_dt1 = [I3, dqx_dq(q, t); 0, dq1q2_dq1] (sparse): row i of J67 in lane i of the group
    double dl[42], m34[12], m44[16];
    w_dlog_dse3(rel, dl);
    w_dqx_dq(sx.q, t2w + 4, m34);
    w_dq1q2_dq1(t2w, m44);
    if (c < 6) {
      double row[7];
#pragma unroll
      for (int jj = 0; jj < 7; ++jj) row[jj] = 0.0;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        if (i == c) {
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
      }
#pragma unroll
      for (int jj = 0; jj < 7; ++jj) L.J[c * 10 + jj] = row[jj];
    }
    if (c == 0) L.J[6 * 10 + 7] = L.J[7 * 10 + 8] = L.J[8 * 10 + 9] = 1.0;
  }
  wave_lds_sync();
  for (int e = c; e < 90; e += 16) {             // tmp = J Sigma (9 x 10)
    const int i = e / 10, jj = e % 10;
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < 10; ++q) acc += L.J[i * 10 + q] * L.Sigma[q * 10 + jj];
    L.tmp[e] = acc;
  }
  wave_lds_sync();
  for (int e = c; e < 81; e += 16) {             // P = (J Sigma) J^T
    const int i = e / 9, jj = e % 9;
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < 10; ++q) acc += L.tmp[i * 10 + q] * L.J[jj * 10 + q];
    L.P[e] = acc;
  }
  wave_lds_sync();
  // Cholesky P = L L^T in place (lane 0 of the group; 9 columns), then X = L^-1 column per lane, W = X^T
  if (c == 0) {
    for (int cc = 0; cc < 9; ++cc) {
      double d = L.P[cc * 9 + cc];
      for (int k = 0; k < cc; ++k) d -= L.P[cc * 9 + k] * L.P[cc * 9 + k];
      const double id = (d > 0.0) ? fast_rsqrt(d) : 0.0;
      L.P[cc * 9 + cc] = d * id;
      L.tmp[cc] = id;
      for (int i = cc + 1; i < 9; ++i) {
        double a = L.P[i * 9 + cc];
        for (int k = 0; k < cc; ++k) a -= L.P[i * 9 + k] * L.P[cc * 9 + k];
        L.P[i * 9 + cc] = a * id;
      }
    }
  }
  wave_lds_sync();
  bool ok = true;
  for (int cc = 0; cc < 9; ++cc) ok = ok && (L.tmp[cc] > 0.0);
  if (live && !ok && c == 0) atomicAdd((unsigned long long*)&v.dbg[20], 1ull);     // singular projection: keeps the previous weight (counted)
  if (live && ok && c < 9) {                     // column c of X = L^-1
    double x[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) x[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      double a = (i == c) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < 9; ++k) if (k < i) a -= L.P[i * 9 + k] * x[k];
      x[i] = (i >= c) ? a * L.tmp[i] : 0.0;
    }
    double* w = v.wsqrtb[1 - wr] + (size_t)s * 81;       // W[a][b] = X[b][a]
#pragma unroll
    for (int i = 0; i < 9; ++i) w[c * 9 + i] = x[i];
  }
}

// ------------------------------------------------------------------------------------------ chain assembly
// One wavefront per frame, one workgroup per chunk (groups of 4 frames).
constexpr int kInitPad = 64;
__global__ __launch_bounds__(256) void k_chain_init(DevView v) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  const Ctrl* ct = v.ctrl;
  if (ct->done) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int C = v.n_cams, D = v.D, N = v.n_frames, ldw = v.ldw;
  double* Gw = sh + wave * (C * kGStride + kInitPad);
  double* Hs = Gw + C * kGStride;
  const int cur = ct->cur;
  const int init_scale = ct->init_scale, reuse = ct->reuse_diag;
  const double radius = ct->radius;
  const double* cams = v.cams[cur];
  const int chunk = blockIdx.x;
  const int f0 = chunk * v.chunk_frames, f1 = min(f0 + v.chunk_frames, N);
  double gsum[kMaxCams][4];
  double isum[4] = {0.0, 0.0, 0.0, 0.0};      // IMU shared block: Hii[a][b] at a*16+b (a,b<15), g_i[a] at a*16+15
#pragma unroll
  for (int c = 0; c < kMaxCams; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q) gsum[c][q] = 0.0;

  for (int fg = f0; fg < f1; fg += 4) {
    const int f = fg + wave;
    if (f >= f1) continue;
    const int t0 = v.frame_tile_off[f], nt = v.frame_tile_off[f + 1] - t0;
    double* Wf = v.cW + (size_t)f * 9 * ldw;
    // pinned frames (separator / ghost, see DevView): their rows go to sep_strip, the chain sees an isolated identity block
    const bool pin_f = (f == 0 && v.pin_first), pin_l = (f == N - 1 && v.pin_last), pin_self = pin_f || pin_l;
    const bool pin_prev = (f == 1 && v.pin_first), pin_next = (f + 1 == N - 1 && v.pin_last);
    double* Wt = pin_self ? v.sep_strip + (size_t)(pin_f ? 0 : 1) * 9 * ldw : Wf;
    const int sep_self = pin_f ? v.sep_col0 : v.sep_col1;
    for (int i = lane; i < 9 * ldw; i += 64) { Wf[i] = 0.0; if (pin_self) Wt[i] = 0.0; }
    if (lane < 42) Hs[lane] = 0.0;
    wave_lds_sync();
    if (nt > 0) {
      for (int m = 0; m < nt * 4; ++m) {
        const int t = m >> 2, q = m & 3;
        const double val = v.Gb[cur][(size_t)(t0 + t) * kGStride + q * 64 + lane];
        Gw[t * kGStride + q * 64 + lane] = val;
        const int c = v.tile_cam[t0 + t];
#pragma unroll
        for (int k = 0; k < kMaxCams; ++k)
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) gsum[k][qq] += (k == c && qq == q) ? val : 0.0;
      }
      wave_lds_sync();
      if (lane < 42) {
        double hval = 0.0;
        for (int t = 0; t < nt; ++t) {
          const int c = v.tile_cam[t0 + t];
          double Rm[9];
          quat_to_R(cams + (size_t)c * kCamStride, Rm);
          const double* g = Gw + t * kGStride;
          if (lane < 36) {
            const int i = lane / 6, j = lane % 6, a = i / 3, ii = i % 3, b = j / 3, jj = j % 3;
            double s = 0.0;
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
              for (int q = 0; q < 3; ++q) s += Rm[3 * p + ii] * g[(3 * a + p) * 16 + 3 * b + q] * Rm[3 * q + jj];
            hval += (a == b) ? s : -s;
          } else {
            const int i = lane - 36, a = i / 3, ii = i % 3;
            const int rc = 6 + model_nk(v.cd[c].model);
            double s = 0.0;
#pragma unroll
            for (int p = 0; p < 3; ++p) s += Rm[3 * p + ii] * g[(3 * a + p) * 16 + rc];
            hval += (a == 0) ? -s : s;
          }
        }
        Hs[lane] = hval;
      }
      // W columns of the cameras (unsolved): lane -> (tile, column)
      for (int idx = lane; idx < nt * 16; idx += 64) {
        const int t = idx >> 4, j = idx & 15;
        const int c = v.tile_cam[t0 + t];
        const int flags = v.cd[c].flags, nk = model_nk(v.cd[c].model);
        const int nrot = (flags & kCamRotFree) ? 3 : 0, ntr = (flags & kCamTransFree) ? 3 : 0;
        const int nc = nrot + ntr + ((flags & kCamKFree) ? nk : 0);
        if (j < nc) {
          double Rm[9];
          quat_to_R(cams + (size_t)c * kCamStride, Rm);
          const double* g = Gw + t * kGStride;
          double u[6];
          if (j < nrot) {
#pragma unroll
            for (int r = 0; r < 6; ++r) u[r] = -(g[r * 16 + 3] * Rm[j] + g[r * 16 + 4] * Rm[3 + j] + g[r * 16 + 5] * Rm[6 + j]);
          } else {
            const int jj = (j < nrot + ntr) ? j - nrot : 6 + (j - nrot - ntr);
#pragma unroll
            for (int r = 0; r < 6; ++r) u[r] = g[r * 16 + jj];
          }
          const int col = v.cd[c].col0 + j;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            Wt[i * ldw + col] = -(Rm[i] * u[0] + Rm[3 + i] * u[1] + Rm[6 + i] * u[2]);
            Wt[(3 + i) * ldw + col] = Rm[i] * u[3] + Rm[3 + i] * u[4] + Rm[6 + i] * u[5];
          }
        }
      }
    }
    wave_lds_sync();
    // IMU blocks: block f-1 has this frame as "cur" (rows/cols 0..8), block f as "prev" (9..17)
    const double* Hc = (f >= 1) ? v.segH + (size_t)(f - 1) * (33 * 33) : nullptr;
    const double* Hp = (f + 1 < N) ? v.segH + (size_t)f * (33 * 33) : nullptr;
    const double* gc = (f >= 1) ? v.segg + (size_t)(f - 1) * 33 : nullptr;
    const double* gp = (f + 1 < N) ? v.segg + (size_t)f * 33 : nullptr;
    double aval[2] = {0.0, 0.0};
    for (int q = 0; q < 2; ++q) {
      const int e = lane + 64 * q;
      if (e < 81) {
        const int i = e / 9, j = e % 9;
        double a = (i < 6 && j < 6) ? Hs[i * 6 + j] : 0.0;
        if (Hc) a += Hc[i * 33 + j];
        if (Hp) a += Hp[(9 + i) * 33 + 9 + j];
        aval[q] = a;
        // rows: this frame (prev of block f), cols: frame f+1 (cur); a pinned end cuts the chain and couples through
        // the separator's columns of the border instead
        v.cB[(size_t)f * 81 + e] = (Hp && !pin_self && !pin_next) ? Hp[(9 + i) * 33 + j] : 0.0;
        if (pin_prev && Hc) Wf[i * ldw + v.sep_col0 + j] = Hc[i * 33 + 9 + j];
        if (pin_next && Hp) Wf[i * ldw + v.sep_col1 + j] = Hp[(9 + i) * 33 + j];
      }
    }
    double gval = 0.0;
    if (lane < 9) {
      gval = (lane < 6) ? Hs[36 + lane] : 0.0;
      if (gc) gval += gc[lane];
      if (gp) gval += gp[9 + lane];
    }
    // border columns of the IMU shared parameters
    for (int idx = lane; idx < 9 * 15; idx += 64) {
      const int i = idx / 15, a = idx % 15;
      const int col = v.imu_param_col[a];
      if (col >= 0) {
        double w = 0.0;
        if (Hc) w += Hc[i * 33 + 18 + a];
        if (Hp) w += Hp[(9 + i) * 33 + 18 + a];
        Wt[i * ldw + col] = w;
      }
    }
    // IMU shared block of block f-1 (every block counted once, by its "cur" frame)
    if (Hc) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = lane + 64 * q, a = e >> 4, b = e & 15;
        if (a < 15) isum[q] += (b < 15) ? Hc[(18 + a) * 33 + 18 + b] : gc[18 + a];
      }
    }
    // damping of the 9 frame parameters (Jacobi scaling fixed per Solve, diagonal re-used after rejections)
    double lam = 0.0, hd = 0.0;
    // diagonal entry i lives in lane (i*10) % 64, slot (i*10) / 64
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int e = i * 10;
      const double d = readlane_f64(aval[e >> 6], e & 63);
      if (lane == i) hd = d;
    }
    if (lane < 9) {
      double sc2, dg;
      if (init_scale) { sc2 = jacobi_scale2(hd); v.cscale2[(size_t)f * 9 + lane] = sc2; } else sc2 = v.cscale2[(size_t)f * 9 + lane];
      if (!reuse) { dg = lm_clamped_diag(hd, sc2); v.cdiag[(size_t)f * 9 + lane] = dg; } else dg = v.cdiag[(size_t)f * 9 + lane];
      lam = pin_self ? 0.0 : dg / (radius * sc2);
      v.clam[(size_t)f * 9 + lane] = lam;
      v.cg[(size_t)f * 9 + lane] = pin_self ? 0.0 : gval;
      Wt[lane * ldw + D] = gval;                // right-hand side rides as column D
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = lane + 64 * q;
      if (e < 81) {
        const int i = e / 9, j = e % 9;
        double a = aval[q];
        const double li = __shfl(lam, i, 64);
        if (i == j) a += li;
        if (pin_self) { Wt[i * ldw + sep_self + j] = aval[q]; a = (i == j) ? 1.0 : 0.0; }
        v.cA[(size_t)f * 81 + e] = a;
      }
    }
  }
  // chunk sums of the camera Gram blocks and of the IMU shared block (4 wavefronts combined in fixed order)
  __syncthreads();
  const int slot = C * kGStride + kGStride;
#pragma unroll
  for (int c = 0; c < kMaxCams; ++c)
    if (c < C) {
#pragma unroll
      for (int q = 0; q < 4; ++q) sh[wave * slot + c * kGStride + q * 64 + lane] = gsum[c][q];
    }
#pragma unroll
  for (int q = 0; q < 4; ++q) sh[wave * slot + C * kGStride + q * 64 + lane] = isum[q];
  __syncthreads();
  double* part = v.part + (size_t)chunk * v.part_stride;
  for (int e = tid; e < slot; e += 256)
    part[D * D + D + e] = (sh[e] + sh[slot + e]) + (sh[2 * slot + e] + sh[3 * slot + e]);
}

// ------------------------------------------------------------------------------------------ cyclic reduction
// Level `s` (s = 2^l): frames e = s (mod 2s) are eliminated. s = 0: the last remaining frame 0.
__global__ __launch_bounds__(256) void k_cr_elim(DevView v, int s) {
  const Ctrl* ct = v.ctrl;
  if (ct->done) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int idx = blockIdx.x * 4 + wave;
  const int N = v.n_frames, D = v.D, ldw = v.ldw;
  const int e = (s == 0) ? 0 : s + 2 * s * idx;
  if (e >= N || (s == 0 && idx > 0)) return;
  double L[81];
  const double* A = v.cA + (size_t)e * 81;
#pragma unroll
  for (int i = 0; i < 81; ++i) L[i] = A[i];
  double dinv[9];
  if (!chol_small<9>(L, dinv)) {
    if (lane == 0) atomicAdd(&v.flags[4 + 2 * v.par], 1);
#pragma unroll
    for (int i = 0; i < 81; ++i) L[i] = (i % 10 == 0) ? 1.0 : 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) dinv[i] = 1.0;
  }
  if (lane < 45) {      // store the factor (lower triangle) over A
    int r = 0, acc = 0;
    while (acc + r + 1 <= lane) { acc += r + 1; ++r; }
    const int c = lane - acc;
    double val = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) val = (i == r && j == c) ? L[i * 9 + j] : val;
    v.cA[(size_t)e * 81 + r * 9 + c] = val;
  }
  const bool has_p = (s > 0) && (e - s >= 0), has_n = (s > 0) && (e + s < N);
  const double* Bp = v.cB + (size_t)(has_p ? e - s : 0) * 81;
  const double* Be = v.cB + (size_t)e * 81;
  double* Wf = v.cW + (size_t)e * 9 * ldw;
  // columns: 0..8 -> P (rhs = row c of B_prev), 9..17 -> Q (rhs = column of B_self), 18.. -> Y | z
  for (int c = lane; c < 18 + D + 1; c += 64) {
    double x[9];
    if (c < 9) {
#pragma unroll
      for (int k = 0; k < 9; ++k) x[k] = has_p ? Bp[c * 9 + k] : 0.0;
    } else if (c < 18) {
#pragma unroll
      for (int k = 0; k < 9; ++k) x[k] = has_n ? Be[k * 9 + (c - 9)] : 0.0;
    } else {
#pragma unroll
      for (int k = 0; k < 9; ++k) x[k] = Wf[k * ldw + (c - 18)];
    }
    fwd_solve_inv<9>(L, dinv, x);
    if (c < 9) {
#pragma unroll
      for (int k = 0; k < 9; ++k) v.cP[(size_t)e * 81 + k * 9 + c] = x[k];
    } else if (c < 18) {
#pragma unroll
      for (int k = 0; k < 9; ++k) v.cQ[(size_t)e * 81 + k * 9 + (c - 9)] = x[k];
    } else {
#pragma unroll
      for (int k = 0; k < 9; ++k) Wf[k * ldw + (c - 18)] = x[k];
    }
  }
}
// Survivors p = 0 (mod 2s) absorb their eliminated neighbours el = p - s (as its "next") and er = p + s (as its "prev").
__global__ __launch_bounds__(256) void k_cr_update(DevView v, int s) {
  const Ctrl* ct = v.ctrl;
  if (ct->done) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int idx = blockIdx.x * 4 + wave;
  const int N = v.n_frames, D = v.D, ldw = v.ldw;
  const int p = 2 * s * idx;
  if (p >= N) return;
  const int el = p - s, er = p + s;
  const bool has_l = el >= 0, has_r = er < N;
  const double* Pr = v.cP + (size_t)(has_r ? er : 0) * 81;
  const double* Qr = v.cQ + (size_t)(has_r ? er : 0) * 81;
  const double* Ql = v.cQ + (size_t)(has_l ? el : 0) * 81;
  const bool r_has_next = has_r && (er + s < N);
  for (int e = lane; e < 81; e += 64) {
    const int i = e / 9, j = e % 9;
    double a = v.cA[(size_t)p * 81 + e], b = 0.0;
    if (has_r) {
      double t = 0.0, u = 0.0;
#pragma unroll
      for (int k = 0; k < 9; ++k) { t += Pr[k * 9 + i] * Pr[k * 9 + j]; u += Pr[k * 9 + i] * Qr[k * 9 + j]; }
      a -= t;
      if (r_has_next) b = -u;
    }
    if (has_l) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 9; ++k) t += Ql[k * 9 + i] * Ql[k * 9 + j];
      a -= t;
    }
    v.cA[(size_t)p * 81 + e] = a;
    v.cB[(size_t)p * 81 + e] = b;
  }
  const double* Yr = v.cW + (size_t)(has_r ? er : 0) * 9 * ldw;
  const double* Yl = v.cW + (size_t)(has_l ? el : 0) * 9 * ldw;
  double* Wp = v.cW + (size_t)p * 9 * ldw;
  for (int e = lane; e < 9 * (D + 1); e += 64) {
    const int i = e / (D + 1), j = e % (D + 1);
    double w = Wp[i * ldw + j];
    if (has_r) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 9; ++k) t += Pr[k * 9 + i] * Yr[k * ldw + j];
      w -= t;
    }
    if (has_l) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 9; ++k) t += Ql[k * 9 + i] * Yl[k * ldw + j];
      w -= t;
    }
    Wp[i * ldw + j] = w;
  }
}
// delta_e = -L^-T (z + P delta_prev + Q delta_next + Y delta_s) for the frames eliminated at level s
__global__ __launch_bounds__(256) void k_cr_back(DevView v, int s) {
  const Ctrl* ct = v.ctrl;
  if (ct->done) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int idx = blockIdx.x * 4 + wave;
  const int N = v.n_frames, D = v.D, ldw = v.ldw;
  const int e = (s == 0) ? 0 : s + 2 * s * idx;
  if (e >= N || (s == 0 && idx > 0)) return;
  const double* Wf = v.cW + (size_t)e * 9 * ldw;
  double y[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) y[k] = 0.0;
  for (int j = lane; j < D; j += 64) {
    const double dj = v.delta_s[j];
#pragma unroll
    for (int k = 0; k < 9; ++k) y[k] += Wf[k * ldw + j] * dj;
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) y[k] = wave_allsum(y[k]) + Wf[k * ldw + D];
  if (s > 0) {
    if (e - s >= 0) {
      const double* P = v.cP + (size_t)e * 81;
      const double* dp = v.cdelta + (size_t)(e - s) * 9;
#pragma unroll
      for (int k = 0; k < 9; ++k) { double t = 0.0;
#pragma unroll
        for (int c = 0; c < 9; ++c) t += P[k * 9 + c] * dp[c];
        y[k] += t; }
    }
    if (e + s < N) {
      const double* Q = v.cQ + (size_t)e * 81;
      const double* dn = v.cdelta + (size_t)(e + s) * 9;
#pragma unroll
      for (int k = 0; k < 9; ++k) { double t = 0.0;
#pragma unroll
        for (int c = 0; c < 9; ++c) t += Q[k * 9 + c] * dn[c];
        y[k] += t; }
    }
  }
  double L[81];
  const double* A = v.cA + (size_t)e * 81;
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int j = 0; j < 9; ++j) L[i * 9 + j] = (j <= i) ? A[i * 9 + j] : 0.0;
  bwd_solve<9>(L, y);
  if (lane < 9) {
    double d = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) d = (k == lane) ? -y[k] : d;
    v.cdelta[(size_t)e * 9 + lane] = d;
  }
}

// sum over all frames of [Y | z]^T [Y | z]: part[chunk] = [ D x D | D ]  (same layout as the vision path)
constexpr int kMaxPairsPerWaveI = 9;
__global__ __launch_bounds__(256) void k_chain_gram(DevView v) {
  extern __shared__ __attribute__((aligned(16))) double R[];    // 36 x ld
  const Ctrl* ct = v.ctrl;
  if (ct->done) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int D = v.D, ld = v.ldw, N = v.n_frames;
  const int nT = (D + 1 + 15) / 16, nPairs = nT * (nT + 1) / 2;
  const int chunk = blockIdx.x;
  const int f0 = chunk * v.chunk_frames, f1 = min(f0 + v.chunk_frames, N);
  double* part = v.part + (size_t)chunk * v.part_stride;
  // column-tile pairs are processed in batches of 36 (9 accumulators per wavefront); more than 8 column tiles
  // (D + 1 > 128) re-reads the chunk's rows once per batch
  for (int pb = 0; pb < nPairs; pb += 4 * kMaxPairsPerWaveI) {
    const int pe = min(nPairs, pb + 4 * kMaxPairsPerWaveI);
    int Ib = 0, Jb = 0;
    for (int p = 0; p < pb; ++p) if (++Jb == nT) { ++Ib; Jb = Ib; }
    v4d acc[kMaxPairsPerWaveI];
#pragma unroll
    for (int i = 0; i < kMaxPairsPerWaveI; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int fg = f0; fg < f1; fg += 4) {
      const int nf = min(4, f1 - fg);
      for (int i = tid; i < 36 * ld; i += 256) {
        const int row = i / ld;
        R[i] = (row < nf * 9) ? v.cW[(size_t)fg * 9 * ld + i] : 0.0;
      }
      __syncthreads();
      int I = Ib, J = Jb, pi = 0;
      for (int p = pb; p < pe; ++p) {
        if ((p & 3) == wave) {
          const double* ra = R + (lane >> 4) * ld + I * 16 + (lane & 15);
          const double* rb = R + (lane >> 4) * ld + J * 16 + (lane & 15);
          v4d a4 = acc[0];
#pragma unroll
          for (int q = 0; q < kMaxPairsPerWaveI; ++q) a4 = (q == pi) ? acc[q] : a4;
#pragma unroll
          for (int ks = 0; ks < 9; ++ks) a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[ks * 4 * ld], rb[ks * 4 * ld], a4, 0, 0, 0);
#pragma unroll
          for (int q = 0; q < kMaxPairsPerWaveI; ++q) acc[q] = (q == pi) ? a4 : acc[q];
          ++pi;
        }
        if (++J == nT) { ++I; J = I; }
      }
      __syncthreads();
    }
    int I = Ib, J = Jb, pi = 0;
    for (int p = pb; p < pe; ++p) {
      if ((p & 3) == wave) {
        v4d a4 = acc[0];
#pragma unroll
        for (int q = 0; q < kMaxPairsPerWaveI; ++q) a4 = (q == pi) ? acc[q] : a4;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = I * 16 + (lane >> 4) + 4 * g, col = J * 16 + (lane & 15);
          if (row < D) { if (col < D) part[row * D + col] = a4[g]; else if (col == D) part[D * D + row] = a4[g]; }
        }
        ++pi;
      }
      if (++J == nT) { ++I; J = I; }
    }
  }
}

// trial poses / velocities of every frame and its step terms
__global__ __launch_bounds__(64) void k_frame_update(DevView v) {
  const Ctrl* ct = v.ctrl;
  if (ct->done) return;
  const int f = blockIdx.x * 64 + threadIdx.x;
  if (f >= v.n_frames) return;
  const int cur = ct->cur;
  // pinned frames step with the reduced system's solution; their gradient / damping terms are counted there, and only the
  // owner (not the rank that holds the ghost copy) counts the step and parameter norms
  const bool pin_f = (f == 0 && v.pin_first), pin_l = (f == v.n_frames - 1 && v.pin_last);
  const double* d = pin_f ? v.delta_s + v.sep_col0 : pin_l ? v.delta_s + v.sep_col1 : v.cdelta + (size_t)f * 9;
  const double* pin = v.poses[cur] + (size_t)f * kPoseStride;
  double* pout = v.poses[1 - cur] + (size_t)f * kPoseStride;
  double Tin[7], Tout[7], dd[6];
  for (int i = 0; i < 7; ++i) Tin[i] = pin[i];
  for (int i = 0; i < 6; ++i) dd[i] = d[i];
  se3_plus(Tin, dd, Tout);
  double gd = 0, dld = 0, step2 = 0, x2 = 0, g2 = 0, gmax = 0;
  for (int i = 0; i < 7; ++i) { pout[i] = Tout[i]; const double e = Tout[i] - Tin[i]; step2 += e * e; x2 += Tin[i] * Tin[i]; }
  pout[7] = 0.0;
  const double* vin = v.vel[cur] + (size_t)f * 4;
  double* vout = v.vel[1 - cur] + (size_t)f * 4;
  for (int i = 0; i < 3; ++i) { const double dv = d[6 + i]; vout[i] = vin[i] + dv; step2 += dv * dv; x2 += vin[i] * vin[i]; }
  vout[3] = 0.0;
  for (int i = 0; i < 9; ++i) {
    const double gi = v.cg[(size_t)f * 9 + i];
    gd += gi * d[i]; dld += v.clam[(size_t)f * 9 + i] * d[i] * d[i]; g2 += gi * gi; gmax = fmax(gmax, fabs(gi));
  }
  if (pin_f || pin_l) { gd = 0; dld = 0; g2 = 0; gmax = 0; }
  if (pin_l) { step2 = 0; x2 = 0; }
  double* o = v.fpart + (size_t)f * kNumScal;
  o[kScGd] = gd; o[kScDld] = dld; o[kScStep2] = step2; o[kScX2] = x2; o[kScG2] = g2; o[kScCost] = 0.0; o[kScGmax] = gmax; o[kScSq] = 0.0;
}

// ------------------------------------------------------------------------------------------ launchers
void launch_imu_jac(const DevView& v, int wr, hipStream_t s) {
  if (v.n_frames < 2) return;
  hipLaunchKernelGGL(k_imu_jac, dim3((v.n_frames - 1 + 7) / 8), dim3(256), 0, s, v, wr);      // 8 blocks per workgroup
}
void launch_imu_res(const DevView& v, int sel, int wr, hipStream_t s) {
  if (v.n_frames < 2) return;
  hipLaunchKernelGGL(k_imu_res, dim3((v.n_frames - 1 + 63) / 64), dim3(64), 0, s, v, sel, wr);
}
void launch_imu_weights(const DevView& v, int wr, hipStream_t s) {
  if (v.n_frames < 2) return;
  const size_t lds = 16 * sizeof(WLds);        // 16 blocks per workgroup (4 per wavefront)
  static bool granted = false;
  if (!granted) { (void)hipFuncSetAttribute((const void*)k_imu_weights, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); granted = true; }
  hipLaunchKernelGGL(k_imu_weights, dim3((v.n_frames - 1 + 15) / 16), dim3(256), lds, s, v, wr);
}
void launch_chain_solve_a(const DevView& v, hipStream_t s) {
  const int N = v.n_frames;
  {
    const size_t slot = (size_t)v.n_cams * kGStride + kGStride;
    const size_t lds = std::max((size_t)4 * (v.n_cams * kGStride + kInitPad), 4 * slot) * sizeof(double);
    static size_t granted = 0;
    if (lds > 65536 && lds > granted) { (void)hipFuncSetAttribute((const void*)k_chain_init, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); granted = lds; }
    hipLaunchKernelGGL(k_chain_init, dim3(v.n_chunks), dim3(256), lds, s, v);
  }
  for (int st = 1; st < N; st *= 2) {
    const int n_elim = (N - st + 2 * st - 1) / (2 * st), n_surv = (N + 2 * st - 1) / (2 * st);
    hipLaunchKernelGGL(k_cr_elim, dim3((n_elim + 3) / 4), dim3(256), 0, s, v, st);
    hipLaunchKernelGGL(k_cr_update, dim3((n_surv + 3) / 4), dim3(256), 0, s, v, st);
  }
  hipLaunchKernelGGL(k_cr_elim, dim3(1), dim3(256), 0, s, v, 0);
  {
    const size_t lds = (size_t)36 * v.ldw * sizeof(double);
    static size_t granted = 0;
    if (lds > 65536 && lds > granted) { (void)hipFuncSetAttribute((const void*)k_chain_gram, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); granted = lds; }
    hipLaunchKernelGGL(k_chain_gram, dim3(v.n_chunks), dim3(256), lds, s, v);
  }
}
void launch_chain_solve_b(const DevView& v, hipStream_t s) {
  const int N = v.n_frames;
  hipLaunchKernelGGL(k_cr_back, dim3(1), dim3(256), 0, s, v, 0);
  int top = 1;
  while (top * 2 < N) top *= 2;
  for (int st = top; st >= 1; st /= 2) {
    if (st >= N) continue;
    const int n_elim = (N - st + 2 * st - 1) / (2 * st);
    hipLaunchKernelGGL(k_cr_back, dim3((n_elim + 3) / 4), dim3(256), 0, s, v, st);
  }
  hipLaunchKernelGGL(k_frame_update, dim3((N + 63) / 64), dim3(64), 0, s, v);
}

}  // namespace vc
