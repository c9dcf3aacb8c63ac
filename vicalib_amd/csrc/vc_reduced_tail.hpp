// vc_reduced_tail.hpp -- what follows the reduced solve: the trial state of the shared parameters and their terms of the step's scalars.
// Included by vc_kernels.hip (k_reduced runs it itself: vision-only passes, the level-by-level back-substitution) and by
// vc_imu_kernels.hip (round 6: one extra workgroup of the back-substitution's launch runs it -- k_reduced, a single workgroup on the critical
// path, then ends with the step and the trial IMU parameters stored: DevView::tail_deferred).  Same arithmetic, same lane layout, same
// summation order in both places: the scalars are identical to the bit.
#pragma once
#include "vc_math.hpp"
#include "vc_device.h"
#include "vc_kutil.hpp"
#ifndef VC_STAMP
#define VC_STAMP(i) do { } while (0)
#endif
namespace vc {
// x: the solution delta_s (D); gs: g_s (D); lamv: the damping (D); s_cam: the accepted camera records (n_cams x kCamStride); cd: camera
// descriptors; ipc: DevView::imu_param_col (15) -- all of them LDS or global; pre_imu: accepted IMU parameter a in lane a < 16 of the
// wavefront whose thread forms the trial ones (wavefront 0 at D <= 64, else 1); red: 6 x 256 doubles of LDS (D > 64 only).
// 256 threads (a workgroup with more lets the first 256 in).
__device__ __forceinline__ void reduced_tail(const DevView& v, int cur, const double* x, const double* gs, const double* lamv, const double* s_cam, const CamDesc* cd,
                                             const int* ipc, double pre_imu, const double* x2_noobs, double* red, bool store_step, bool store_imu) {
  const int tid = threadIdx.x, D = v.D;
  double gd = 0, dld = 0, step2 = 0, x2 = 0, g2 = 0, gmax = 0;
  for (int i = tid; i < D; i += 256) {
    const double d = x[i], g = gs[i];
    if (store_step) v.delta_s[i] = d;
    gd += g * d; dld += lamv[i] * d * d; g2 += g * g; gmax = fmax(gmax, fabs(g));
  }
  VC_STAMP(16);
  // D <= 64: every term lives in wavefront 0 (the IMU parameters move to its lane 63) -- no staging through LDS, no barriers
  const bool one_wave = D <= 64;
  if (one_wave) { if (tid < 64) for (int i = tid; i < v.n_cams * kCamStride; i += 64) v.cams[1 - cur][i] = s_cam[i]; }
  else {
    for (int i = tid; i < v.n_cams * kCamStride; i += 256) v.cams[1 - cur][i] = s_cam[i];
    __syncthreads();
  }
  if (tid < v.n_cams) {
    const int c = tid;
    const double* cin = s_cam + (size_t)c * kCamStride;
    double* cout = v.cams[1 - cur] + (size_t)c * kCamStride;
    const int flags = cd[c].flags, nk = model_nk(cd[c].model);
    int cc = cd[c].col0;
    if (flags & kCamRotFree) {
      double q[4], w[3] = {x[cc], x[cc + 1], x[cc + 2]}, qi[4] = {cin[0], cin[1], cin[2], cin[3]};
      so3_plus(qi, w, q);
#pragma unroll
      for (int i = 0; i < 4; ++i) { const double e = q[i] - qi[i]; step2 += e * e; x2 += qi[i] * qi[i]; cout[i] = q[i]; }
      cc += 3;
    }
    if (flags & kCamTransFree) {
      for (int i = 0; i < 3; ++i) { const double d = x[cc + i], o = cin[4 + i]; step2 += d * d; x2 += o * o; cout[4 + i] = o + d; }
      cc += 3;
    }
    if (flags & kCamKFree) {
      for (int i = 0; i < nk; ++i) { const double d = x[cc + i], o = cin[kCamK + i]; step2 += d * d; x2 += o * o; cout[kCamK + i] = o + d; }
    }
  }
  VC_STAMP(17);
  double o[16];      // accepted IMU parameters: requested at kernel entry by lanes 0..15 of this thread's wavefront
  if (v.imu_on && (tid >> 6) == (one_wave ? 0 : 1)) {
#pragma unroll
    for (int a = 0; a < 16; ++a) o[a] = readlane_f64(pre_imu, a);
  }
  if (v.imu_on && tid == (one_wave ? 63 : 64)) {     // g(2) b(6) sf(6) toff(1): plain additive parameters
    double* iout = v.imus[1 - cur];
    // all loads first, then the arithmetic, then the stores: interleaved, every store would hold back the next element's loads
    // (the compiler cannot tell the two buffers apart) -- 15 dependent memory round trips in the tail of a critical-path kernel
    double dlt[15];
#pragma unroll
    for (int a = 0; a < 15; ++a) { const int col = ipc[a]; dlt[a] = (col >= 0) ? x[col >= 0 ? col : 0] : 0.0; }
#pragma unroll
    for (int a = 0; a < 15; ++a) {
      if (ipc[a] >= 0) { step2 += dlt[a] * dlt[a]; x2 += o[a] * o[a]; }
      if (store_imu) iout[a] = o[a] + dlt[a];
    }
    if (store_imu) iout[15] = o[15];
  }
  VC_STAMP(18);
  double t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;      // totals, valid in thread 0
  if (one_wave) {
    if (tid < 64) {
      const double in6[6] = {gd, dld, step2, x2, g2, 0.0};
      double out6[6];
      wave_sum6(in6, out6, tid);
      t0 = out6[0]; t1 = out6[1]; t2 = out6[2]; t3 = out6[3]; t4 = out6[4]; t5 = gmax;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) t5 = fmax(t5, __shfl_down(t5, o, 64));
    }
  } else {
    // only threads < max(D, 65) hold terms: stage them, one wavefront adds them in fixed order
    red[tid] = gd; red[256 + tid] = dld; red[512 + tid] = step2; red[768 + tid] = x2; red[1024 + tid] = g2; red[1280 + tid] = gmax;
    __syncthreads();
    if (tid < 64) {
      double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = tid + 64 * q;
        a0 += red[i]; a1 += red[256 + i]; a2 += red[512 + i]; a3 += red[768 + i]; a4 += red[1024 + i]; a5 = fmax(a5, red[1280 + i]);
      }
      t0 = wave_sum(a0); t1 = wave_sum(a1); t2 = wave_sum(a2); t3 = wave_sum(a3); t4 = wave_sum(a4); t5 = a5;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) t5 = fmax(t5, __shfl_down(t5, o, 64));
    }
  }
  VC_STAMP(6);
  if (tid == 0) {
    double* h = v.scal + kNumScal;
    h[kScGd] = t0; h[kScDld] = t1; h[kScStep2] = t2; h[kScX2] = t3; h[kScG2] = t4;
    h[kScCost] = 0.0; h[kScGmax] = t5; h[kScSq] = 0.0;
    if (v.merged) {
      // frames without observations take no part in k_trial: their parameter norm (chunk sums in the Schur partials) is
      // added here; then flag the record (a trial point is about to exist) and clear the failure flags of the next pass
      // (x2_noobs: that sum, left in LDS by phase A when it ran in this launch)
      double x2 = 0.0;
      if (x2_noobs) x2 = *x2_noobs;
      else x2 = v.part_total[v.part_stride - 2];
      h[kScX2] += x2;
      v.ctrl->needs_decision = 1;
      v.flags[4 + 2 * (1 - v.par)] = 0; v.flags[5 + 2 * (1 - v.par)] = 0;
    }
  }
}
}  // namespace vc
