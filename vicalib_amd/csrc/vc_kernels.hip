// vc_kernels.hip -- CDNA4 (gfx950, wave64) kernels of the batched Levenberg-Marquardt engine.
//
// Replaces, on the device, what the reference does per residual block inside ceres::Solve
// (vicalibrator.h:956): AutoDiffCostFunction::Evaluate of ImuReprojectionCostFunctor
// (ceres-cost-functions.h:350-373) + loss correction + block-sparse J^T J, the elimination of the
// per-frame pose blocks (Ceres' sparse Cholesky), the dense solve on the shared parameters, the
// manifold update (local-param-se3.h), the cost evaluation of the trial point and the trust-region
// bookkeeping (step quality, radius, termination tests, iteration callback vicalibrator.h:690-721).
//
//  k_reproj_jac      one wavefront per (frame,camera) tile: closed-form unique-column Jacobian rows per
//                    corner (lane = corner), rows staged in wave-private LDS, tile Gram block
//                    G = sum w u^T u accumulated on the matrix pipe (v_mfma_f64_16x16x4_f64)
//  k_frame_prep      one wavefront per frame: H_pp, g_p from the tile Gram blocks, damping, 6x6 Cholesky,
//                    Y = L^-1 W per tile (lanes = columns)
//  k_schur_reduce    per frame chunk: sum Y^T Y, sum Y^T z, per-camera sum of G   (LDS staged)
//  k_schur_final     fixed-order sum of the chunk partials + camera blocks -> packed reduced system
//  k_reduced_solve   one workgroup: damped Cholesky in LDS, delta_s, trial state of the shared parameters
//  k_trial           one wavefront per tile: back-substitution, T <- T exp(delta), trial residual sweep
//  k_final           fixed-order reduction of the step scalars + the accept/reject decision (device Ctrl)
#include <hip/hip_runtime.h>
#include "vc_math.hpp"
#include "vc_device.h"

namespace vc {

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int kDotStride = 34;   // doubles per corner in LDS: 2 rows x 16 + 2 pad (272 B: conflict-free b128 stores)

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ double wave_sum(double x) {      // result valid in lane 0
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
  return x;
}
__device__ __forceinline__ double wave_allsum(double x) {   // result in every lane, fixed order
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}

// ------------------------------------------------------------------------------------------ Jacobian sweep
template <int MODEL>
__device__ __forceinline__ void jac_tile_body(const DevView& v, int cur, double mult, int tile, int lane, double* wl) {
  const int f = v.tile_frame[tile], c = v.tile_cam[tile];
  const int off = v.tile_off[tile], cnt = v.tile_off[tile + 1] - off;
  const double* pose = v.poses[cur] + (size_t)f * kPoseStride;
  const double* cam = v.cams[cur] + (size_t)c * kCamStride;
  TileXf x;
  make_tile_xf(pose, cam, &x);
  double K[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) K[i] = cam[kCamK + i];
  v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
  double cost = 0.0;
  double* mine = wl + lane * kDotStride;
  for (int base = 0; base < cnt; base += 64) {
    const int d = base + lane;
    if (d < cnt) {
      const double2 uv = v.obs_uv[off + d];
      const double* pw = v.points + 3 * (size_t)v.obs_pt[off + d];
      cost += corner_rows<MODEL>(x, K, pw, uv.x, uv.y, mult, mine, mine + 16);
    } else {
#pragma unroll
      for (int i = 0; i < 32; ++i) mine[i] = 0.0;
    }
    wave_lds_sync();
    const int nd = min(64, cnt - base);
    const int nsteps = (nd + 1) >> 1;          // one MFMA covers 2 corners x 2 residual rows (K = 4)
    const double* src = wl + (lane >> 5) * kDotStride + (lane & 31);
    int k = 0;
    for (; k + 1 < nsteps; k += 2) {           // two independent accumulators hide the MFMA dependency latency
      const double u0 = src[2 * k * kDotStride];
      const double u1 = src[(2 * k + 2) * kDotStride];
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(u0, u0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(u1, u1, acc1, 0, 0, 0);
    }
    if (k < nsteps) {
      const double u0 = src[2 * k * kDotStride];
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(u0, u0, acc0, 0, 0, 0);
    }
    wave_lds_sync();
  }
  double* G = v.G + (size_t)tile * kGStride;
#pragma unroll
  for (int i = 0; i < 4; ++i) G[((lane >> 4) + 4 * i) * 16 + (lane & 15)] = acc0[i] + acc1[i];
  cost = wave_sum(cost);
  if (lane == 0) v.tile_cost[tile] = cost;
}

__global__ __launch_bounds__(256) void k_reproj_jac(DevView v) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const Ctrl* ct = v.ctrl;
  if (ct->done || !ct->need_lin) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= v.n_tiles) return;
  double* wl = lds + wave * 64 * kDotStride;
  const int cur = ct->cur;
  const double mult = ct->mult;
  switch (v.cam_model[v.tile_cam[tile]]) {   // wave-uniform
    case kFov: jac_tile_body<kFov>(v, cur, mult, tile, lane, wl); break;
    case kPoly2: jac_tile_body<kPoly2>(v, cur, mult, tile, lane, wl); break;
    case kPoly3: jac_tile_body<kPoly3>(v, cur, mult, tile, lane, wl); break;
    case kKb4: jac_tile_body<kKb4>(v, cur, mult, tile, lane, wl); break;
    default: jac_tile_body<kLinear>(v, cur, mult, tile, lane, wl); break;
  }
}

// ------------------------------------------------------------------------------------------ residual sweeps
template <int MODEL>
__device__ __forceinline__ void res_tile_sweep(const DevView& v, const TileXf& x, const double* K, int off, int cnt, int lane,
                                               double* cost_out, double* sq_out) {
  double cost = 0.0, sq = 0.0;
  for (int d = lane; d < cnt; d += 64) {
    const double2 uv = v.obs_uv[off + d];
    const double* pw = v.points + 3 * (size_t)v.obs_pt[off + d];
    double r[2];
    cost += corner_residual<MODEL>(x, K, pw, uv.x, uv.y, r);
    sq += r[0] * r[0] + r[1] * r[1];
  }
  *cost_out = wave_sum(cost);
  *sq_out = wave_sum(sq);
}
__device__ __forceinline__ void res_tile_dispatch(const DevView& v, int model, const TileXf& x, const double* K, int off, int cnt,
                                                  int lane, double* cost, double* sq) {
  switch (model) {
    case kFov: res_tile_sweep<kFov>(v, x, K, off, cnt, lane, cost, sq); break;
    case kPoly2: res_tile_sweep<kPoly2>(v, x, K, off, cnt, lane, cost, sq); break;
    case kPoly3: res_tile_sweep<kPoly3>(v, x, K, off, cnt, lane, cost, sq); break;
    case kKb4: res_tile_sweep<kKb4>(v, x, K, off, cnt, lane, cost, sq); break;
    default: res_tile_sweep<kLinear>(v, x, K, off, cnt, lane, cost, sq); break;
  }
}
// plain sweep of one state buffer (RMSE, vc_evaluate): tile_trial[t] = {mult * sum rho, sum |r|^2}
__global__ __launch_bounds__(256) void k_reproj_res(DevView v, int state, double mult) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= v.n_tiles) return;
  const int f = v.tile_frame[tile], c = v.tile_cam[tile];
  const double* cam = v.cams[state] + (size_t)c * kCamStride;
  TileXf x;
  make_tile_xf(v.poses[state] + (size_t)f * kPoseStride, cam, &x);
  double K[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) K[i] = cam[kCamK + i];
  double cost, sq;
  res_tile_dispatch(v, v.cam_model[c], x, K, v.tile_off[tile], v.tile_off[tile + 1] - v.tile_off[tile], lane, &cost, &sq);
  if (lane == 0) { v.tile_trial[2 * tile] = mult * cost; v.tile_trial[2 * tile + 1] = sq; }
}

// per-corner outlier mask (RemoveOutliers, vicalibrator.h:859-916): |r| > thresh[cam]
template <int MODEL>
__device__ __forceinline__ void mask_tile_body(const DevView& v, const TileXf& x, const double* K, int off, int cnt, int lane,
                                               double th, unsigned char* mask) {
  for (int d = lane; d < cnt; d += 64) {
    const double2 uv = v.obs_uv[off + d];
    double r[2];
    corner_residual<MODEL>(x, K, v.points + 3 * (size_t)v.obs_pt[off + d], uv.x, uv.y, r);
    mask[off + d] = sqrt(r[0] * r[0] + r[1] * r[1]) > th ? 1 : 0;
  }
}
__global__ __launch_bounds__(256) void k_outlier_mask(DevView v, int state, const double* thresh, unsigned char* mask) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= v.n_tiles) return;
  const int f = v.tile_frame[tile], c = v.tile_cam[tile];
  const int off = v.tile_off[tile], cnt = v.tile_off[tile + 1] - off;
  const double* cam = v.cams[state] + (size_t)c * kCamStride;
  TileXf x;
  make_tile_xf(v.poses[state] + (size_t)f * kPoseStride, cam, &x);
  double K[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) K[i] = cam[kCamK + i];
  const double th = thresh[c];
  switch (v.cam_model[c]) {
    case kFov: mask_tile_body<kFov>(v, x, K, off, cnt, lane, th, mask); break;
    case kPoly2: mask_tile_body<kPoly2>(v, x, K, off, cnt, lane, th, mask); break;
    case kPoly3: mask_tile_body<kPoly3>(v, x, K, off, cnt, lane, th, mask); break;
    case kKb4: mask_tile_body<kKb4>(v, x, K, off, cnt, lane, th, mask); break;
    default: mask_tile_body<kLinear>(v, x, K, off, cnt, lane, th, mask); break;
  }
}

// ------------------------------------------------------------------------------------------ frame elimination
// One wavefront per frame.  Lanes 0..35 own the entries of H_pp, 36..41 those of g_p; every lane then
// factors the 6x6 block redundantly (wave-uniform), and lanes own (tile, column) pairs of Y = L^-1 W.
constexpr int kPrepLds = kMaxCams * 96 + 48;
__global__ __launch_bounds__(256) void k_frame_prep(DevView v) {
  __shared__ double sh[4 * kPrepLds];
  const Ctrl* ct = v.ctrl;
  if (ct->done) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int f = blockIdx.x * 4 + wave;
  if (f >= v.n_frames) return;
  const int cur = ct->cur;
  const int t0 = v.frame_tile_off[f], nt = v.frame_tile_off[f + 1] - t0;
  double* Gs = sh + wave * kPrepLds;
  double* Hs = Gs + kMaxCams * 96;
  const double* cams = v.cams[cur];
  if (nt == 0) {     // frame without observations: nothing to eliminate, keeps its pose
    if (lane == 0) {
      const double* p = v.poses[cur] + (size_t)f * kPoseStride;
      double x2 = 0;
      for (int i = 0; i < 7; ++i) x2 += p[i] * p[i];
      double* o = v.fpart + (size_t)f * kNumScal;
      for (int i = 0; i < kNumScal; ++i) o[i] = 0.0;
      o[kScX2] = x2;
    }
    return;
  }
  for (int i = lane; i < nt * 96; i += 64) Gs[i] = v.G[(size_t)(t0 + i / 96) * kGStride + (i % 96)];
  wave_lds_sync();
  double hval = 0.0;
  if (lane < 42) {
    for (int t = 0; t < nt; ++t) {
      const int c = v.tile_cam[t0 + t];
      double R[9];
      quat_to_R(cams + (size_t)c * kCamStride, R);
      const double* g = Gs + t * 96;
      if (lane < 36) {
        const int i = lane / 6, j = lane % 6, a = i / 3, ii = i % 3, b = j / 3, jj = j % 3;
        double s = 0.0;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int q = 0; q < 3; ++q) s += R[3 * p + ii] * g[(3 * a + p) * 16 + 3 * b + q] * R[3 * q + jj];
        hval += (a == b) ? s : -s;
      } else {
        const int i = lane - 36, a = i / 3, ii = i % 3;
        const int rc = 6 + model_nk(v.cam_model[c]);
        double s = 0.0;
#pragma unroll
        for (int p = 0; p < 3; ++p) s += R[3 * p + ii] * g[(3 * a + p) * 16 + rc];
        hval += (a == 0) ? -s : s;
      }
    }
    Hs[lane] = hval;
  }
  wave_lds_sync();
  double H[36], g6[6], lam[6];
#pragma unroll
  for (int i = 0; i < 36; ++i) H[i] = Hs[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) g6[i] = Hs[36 + i];
  const int init_scale = ct->init_scale, reuse = ct->reuse_diag;
  const double radius = ct->radius;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const double hd = H[i * 6 + i];
    double sc2, dg;
    if (init_scale) { sc2 = jacobi_scale2(hd); if (lane == 0) v.fscale2[(size_t)f * 6 + i] = sc2; }
    else sc2 = v.fscale2[(size_t)f * 6 + i];
    if (!reuse) { dg = lm_clamped_diag(hd, sc2); if (lane == 0) v.fdiag[(size_t)f * 6 + i] = dg; }
    else dg = v.fdiag[(size_t)f * 6 + i];
    lam[i] = dg / (radius * sc2);
    H[i * 6 + i] = hd + lam[i];
  }
  if (!chol_small<6>(H)) {
    if (lane == 0) atomicAdd(&v.flags[0], 1);
#pragma unroll
    for (int i = 0; i < 36; ++i) H[i] = (i % 7 == 0) ? 1.0 : 0.0;
  }
  double z[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) z[i] = g6[i];
  fwd_solve<6>(H, z);
  if (lane == 0) {
    double* fr = v.fr + (size_t)f * kFrStride;
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) fr[kFrL + k++] = H[i * 6 + j];
#pragma unroll
    for (int i = 0; i < 6; ++i) { fr[kFrZ + i] = z[i]; fr[kFrG + i] = g6[i]; fr[kFrLam + i] = lam[i]; }
  }
  // Y columns: lane -> (tile, column)
  for (int idx = lane; idx < nt * 16; idx += 64) {
    const int t = idx >> 4, j = idx & 15;
    const int c = v.tile_cam[t0 + t];
    const int flags = v.cam_flags[c], nk = model_nk(v.cam_model[c]);
    const int nrot = (flags & kCamRotFree) ? 3 : 0, ntr = (flags & kCamTransFree) ? 3 : 0;
    const int nc = nrot + ntr + ((flags & kCamKFree) ? nk : 0);
    double w[6] = {0, 0, 0, 0, 0, 0};
    if (j < nc) {
      double R[9];
      quat_to_R(cams + (size_t)c * kCamStride, R);
      const double* g = Gs + t * 96;
      double u[6];   // column of [Gaa Ea | GaB]
      if (j < nrot) {
#pragma unroll
        for (int r = 0; r < 6; ++r) u[r] = -(g[r * 16 + 3] * R[j] + g[r * 16 + 4] * R[3 + j] + g[r * 16 + 5] * R[6 + j]);
      } else if (j < nrot + ntr) {
        const int jj = j - nrot;
#pragma unroll
        for (int r = 0; r < 6; ++r) u[r] = g[r * 16 + jj];
      } else {
        const int jj = 6 + (j - nrot - ntr);
#pragma unroll
        for (int r = 0; r < 6; ++r) u[r] = g[r * 16 + jj];
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {   // w = Qa^T u, Qa = diag(-R, R)
        w[i] = -(R[i] * u[0] + R[3 + i] * u[1] + R[6 + i] * u[2]);
        w[3 + i] = R[i] * u[3] + R[3 + i] * u[4] + R[6 + i] * u[5];
      }
      fwd_solve<6>(H, w);
    }
    double* Yt = v.Y + (size_t)(t0 + t) * kYStride;
#pragma unroll
    for (int r = 0; r < 6; ++r) Yt[r * kUCols + j] = w[r];
  }
}

// Partial Schur sums of one frame chunk: part = [ sum Y^T Y (upper, D x D) | sum Y^T z (D) | per-camera sum of G (C x 256) ]
constexpr int kSchurBatchTiles = 64;    // tiles staged in LDS per batch (48 KB)
__global__ __launch_bounds__(256) void k_schur_reduce(DevView v) {
  __shared__ double Ys[kSchurBatchTiles * kYStride];
  __shared__ double Zs[kSchurBatchTiles * 6];
  if (v.ctrl->done) return;
  const int chunk = blockIdx.x, tid = threadIdx.x;
  const int f0 = chunk * v.chunk_frames, f1 = min(f0 + v.chunk_frames, v.n_frames);
  const int D = v.D, C = v.n_cams;
  double* part = v.part + (size_t)chunk * v.part_stride;
  const int t0 = v.frame_tile_off[f0], t1 = v.frame_tile_off[f1];
  {
    double gsum[kMaxCams];
#pragma unroll
    for (int c = 0; c < kMaxCams; ++c) gsum[c] = 0.0;
    for (int t = t0; t < t1; ++t) {
      const double gv = v.G[(size_t)t * kGStride + tid];
      const int c = v.tile_cam[t];
#pragma unroll
      for (int k = 0; k < kMaxCams; ++k) gsum[k] += (k == c) ? gv : 0.0;
    }
#pragma unroll
    for (int c = 0; c < kMaxCams; ++c) if (c < C) part[D * D + D + c * kGStride + tid] = gsum[c];
  }
  const int nE = D * D + D;
  // each thread owns entries e = tid, tid + 256, ... (at most 65 for D = 128); accumulate over batches of frames
  double acc[4] = {0, 0, 0, 0};
  const int frames_per_batch = max(1, kSchurBatchTiles / max(C, 1));
  for (int fb = f0; fb < f1; fb += frames_per_batch) {
    const int fe = min(fb + frames_per_batch, f1);
    const int tb = v.frame_tile_off[fb], te = v.frame_tile_off[fe];
    __syncthreads();
    for (int i = tid; i < (te - tb) * kYStride; i += 256) Ys[i] = v.Y[(size_t)tb * kYStride + i];
    for (int i = tid; i < (fe - fb) * 6; i += 256) Zs[i] = v.fr[(size_t)(fb + i / 6) * kFrStride + kFrZ + (i % 6)];
    __syncthreads();
    int slot = 0;
    for (int e = tid; e < nE; e += 256, ++slot) {
      double s = 0.0;
      if (e < D * D) {
        const int ra = e / D, rb = e % D;
        if (rb >= ra) {
          const int ca = v.col_cam[ra], cb = v.col_cam[rb];
          if (ca >= 0 && cb >= 0) {
            const int la = v.col_local[ra], lb = v.col_local[rb];
            for (int f = fb; f < fe; ++f) {
              const int ta = v.frame_cam_tile[f * C + ca], tb2 = v.frame_cam_tile[f * C + cb];
              if (ta < 0 || tb2 < 0) continue;
              const double* ya = Ys + (ta - tb) * kYStride + la;
              const double* yb = Ys + (tb2 - tb) * kYStride + lb;
#pragma unroll
              for (int k = 0; k < 6; ++k) s += ya[k * kUCols] * yb[k * kUCols];
            }
          }
        }
      } else {
        const int ra = e - D * D;
        const int ca = v.col_cam[ra];
        if (ca >= 0) {
          const int la = v.col_local[ra];
          for (int f = fb; f < fe; ++f) {
            const int ta = v.frame_cam_tile[f * C + ca];
            if (ta < 0) continue;
            const double* ya = Ys + (ta - tb) * kYStride + la;
            const double* z = Zs + (f - fb) * 6;
#pragma unroll
            for (int k = 0; k < 6; ++k) s += ya[k * kUCols] * z[k];
          }
        }
      }
      if (slot < 4) acc[slot] += s; else part[e] = (fb == f0 ? 0.0 : part[e]) + s;
    }
  }
  int slot = 0;
  for (int e = tid; e < nE && slot < 4; e += 256, ++slot) part[e] = acc[slot];
}

// Sbuf = [ S = H_ss - sum Y^T Y (full symmetric, undamped) | g_red | diag(H_ss) | g_s | cost, 0 ]
__global__ __launch_bounds__(256) void k_schur_final(DevView v) {
  __shared__ double gsum[kMaxCams * kGStride];
  __shared__ double P[16 * 16];
  __shared__ double T1[16 * 16];
  __shared__ double red[256];
  const Ctrl* ct = v.ctrl;
  if (ct->done) return;
  const int tid = threadIdx.x, D = v.D, C = v.n_cams;
  double* S = v.Sbuf;
  double* gred = S + D * D;
  double* hd = gred + D;
  double* gs = hd + D;
  double* sc = gs + D;
  for (int e = tid; e < D * D + D; e += 256) {
    double s = 0.0;
    for (int k = 0; k < v.n_chunks; ++k) s += v.part[(size_t)k * v.part_stride + e];
    if (e < D * D) S[e] = -s; else gred[e - D * D] = -s;
  }
  for (int c = 0; c < C; ++c) {
    double s = 0.0;
    for (int k = 0; k < v.n_chunks; ++k) s += v.part[(size_t)k * v.part_stride + D * D + D + c * kGStride + tid];
    gsum[c * kGStride + tid] = s;
  }
  {
    double s = 0.0;
    for (int t = tid; t < v.n_tiles; t += 256) s += v.tile_cost[t];
    red[tid] = s;
  }
  for (int i = tid; i < D; i += 256) { hd[i] = 0.0; gs[i] = 0.0; }
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  if (tid == 0) { sc[0] = 0.5 * red[0]; sc[1] = 0.0; }
  // camera blocks H_cc = P^T G P, g_c = P^T G[:, r]   (P: u-columns -> shared columns of the camera)
  for (int c = 0; c < C; ++c) {
    const int flags = v.cam_flags[c], nk = model_nk(v.cam_model[c]);
    const int nu = 6 + nk, nc = cam_ncols(flags, nk), c0 = v.cam_col0[c];
    const int nrot = (flags & kCamRotFree) ? 3 : 0, ntr = (flags & kCamTransFree) ? 3 : 0;
    const double* G = gsum + c * kGStride;
    __syncthreads();
    {
      const int i = tid >> 4, a = tid & 15;     // P[i][a]
      double R[9];
      quat_to_R(v.cams[ct->cur] + (size_t)c * kCamStride, R);
      double pv = 0.0;
      if (i < nu && a < nc) {
        if (a < nrot) { if (i >= 3 && i < 6) pv = -R[3 * (i - 3) + a]; }
        else if (a < nrot + ntr) { if (i == a - nrot) pv = 1.0; }
        else { if (i == 6 + (a - nrot - ntr)) pv = 1.0; }
      }
      P[tid] = pv;
    }
    __syncthreads();
    {
      const int i = tid >> 4, a = tid & 15;     // T1[i][a] = sum_k G[i][k] P[k][a]; row 15 := g_c
      double s = 0.0;
      if (i < nu && a < nc) for (int k = 0; k < nu; ++k) s += G[i * 16 + k] * P[k * 16 + a];
      if (i == 15 && a < nc) { s = 0.0; for (int k = 0; k < nu; ++k) s += P[k * 16 + a] * G[k * 16 + nu]; }
      T1[tid] = s;
    }
    __syncthreads();
    {
      const int b = tid >> 4, a = tid & 15;     // Hcc[b][a] = sum_i P[i][b] T1[i][a]
      if (a < nc && b < nc && a >= b) {
        double s = 0.0;
        for (int i = 0; i < nu; ++i) s += P[i * 16 + b] * T1[i * 16 + a];
        S[(c0 + b) * D + c0 + a] += s;
        if (a == b) hd[c0 + a] = s;
      }
      if (b == 15 && a < nc) { gred[c0 + a] += T1[15 * 16 + a]; gs[c0 + a] = T1[15 * 16 + a]; }
    }
  }
  __syncthreads();
  for (int e = tid; e < D * D; e += 256) {
    const int i = e / D, j = e % D;
    if (j < i) S[e] = S[j * D + i];
  }
}

// ------------------------------------------------------------------------------------------ reduced solve
// One workgroup.  Damped Cholesky of the augmented matrix [S + Lambda, g; g^T, .] in LDS (the forward
// substitution rides along as the extra row), back substitution, then the trial state of the shared
// parameters (cams[1-cur] <- Plus(cams[cur], delta_s)) and their scalar terms scal[8..15].
__global__ __launch_bounds__(256) void k_reduced_solve(DevView v) {
  extern __shared__ __attribute__((aligned(16))) double M[];
  __shared__ double red[6 * 256];
  const Ctrl* ct = v.ctrl;
  if (ct->done) return;
  const int tid = threadIdx.x, D = v.D, ld = D + 1, cur = ct->cur;
  double* col = M + (size_t)(D + 1) * ld;   // scaled pivot column
  double* x = col + D + 1;
  const double* S = v.Sbuf;
  const double* gred = S + D * D;
  const double* hd = gred + D;
  const double* gs = hd + D;
  for (int e = tid; e < D * D; e += 256) M[(e / D) * ld + (e % D)] = S[e];
  for (int i = tid; i < D; i += 256) M[D * ld + i] = gred[i];     // augmented row
  __syncthreads();
  for (int i = tid; i < D; i += 256) {
    double sc2, dg;
    if (ct->init_scale) { sc2 = jacobi_scale2(hd[i]); v.sscale2[i] = sc2; } else sc2 = v.sscale2[i];
    if (!ct->reuse_diag) { dg = lm_clamped_diag(hd[i], sc2); v.sdiag[i] = dg; } else dg = v.sdiag[i];
    const double lam = dg / (ct->radius * sc2);
    v.slam[i] = lam;
    M[i * ld + i] += lam;
  }
  __syncthreads();
  for (int j = 0; j < D; ++j) {
    double d = M[j * ld + j];
    const bool bad = !(d > 0.0);
    if (bad) d = 1.0;
    const double piv = sqrt(d);
    for (int i = j + 1 + tid; i <= D; i += 256) col[i] = M[i * ld + j] / piv;
    __syncthreads();
    if (tid == 0) { M[j * ld + j] = piv; if (bad) v.flags[1] = 1; }   // nobody reads M[j][j] again before the next barrier
    const int n = D - j;                        // rows j+1 .. D (incl. the augmented row)
    for (int idx = tid; idx < n * n; idx += 256) {
      const int i = j + 1 + idx / n, k = j + 1 + idx % n;
      if (k <= i && k < D) M[i * ld + k] -= col[i] * col[k];
    }
    for (int i = j + 1 + tid; i <= D; i += 256) M[i * ld + j] = col[i];
    __syncthreads();
  }
  // y = L^-1 g sits in row D; delta_s = -L^-T y
  for (int i = tid; i < D; i += 256) x[i] = -M[D * ld + i];
  __syncthreads();
  if (D <= 48) {
    if (tid == 0) for (int j = D - 1; j >= 0; --j) { double s = x[j]; for (int k = j + 1; k < D; ++k) s -= M[k * ld + j] * x[k]; x[j] = s / M[j * ld + j]; }
    __syncthreads();
  } else {
    for (int j = D - 1; j >= 0; --j) {
      if (tid == 0) x[j] /= M[j * ld + j];
      __syncthreads();
      const double xj = x[j];
      for (int i = tid; i < j; i += 256) x[i] -= M[j * ld + i] * xj;
      __syncthreads();
    }
  }
  double gd = 0, dld = 0, step2 = 0, x2 = 0, g2 = 0, gmax = 0;
  for (int i = tid; i < D; i += 256) {
    const double d = x[i], g = gs[i];
    v.delta_s[i] = d;
    gd += g * d; dld += v.slam[i] * d * d; g2 += g * g; gmax = fmax(gmax, fabs(g));
  }
  if (tid < v.n_cams) {
    const int c = tid;
    const double* cin = v.cams[cur] + (size_t)c * kCamStride;
    double* cout = v.cams[1 - cur] + (size_t)c * kCamStride;
    for (int i = 0; i < kCamStride; ++i) cout[i] = cin[i];
    const int flags = v.cam_flags[c], nk = model_nk(v.cam_model[c]);
    int cc = v.cam_col0[c];
    if (flags & kCamRotFree) {
      double q[4], w[3] = {x[cc], x[cc + 1], x[cc + 2]}, qi[4] = {cin[0], cin[1], cin[2], cin[3]};
      so3_plus(qi, w, q);
      for (int i = 0; i < 4; ++i) { cout[i] = q[i]; const double e = q[i] - cin[i]; step2 += e * e; x2 += cin[i] * cin[i]; }
      cc += 3;
    }
    if (flags & kCamTransFree) {
      for (int i = 0; i < 3; ++i) { const double d = x[cc + i]; cout[4 + i] = cin[4 + i] + d; step2 += d * d; x2 += cin[4 + i] * cin[4 + i]; }
      cc += 3;
    }
    if (flags & kCamKFree) {
      for (int i = 0; i < nk; ++i) { const double d = x[cc + i]; cout[kCamK + i] = cin[kCamK + i] + d; step2 += d * d; x2 += cin[kCamK + i] * cin[kCamK + i]; }
    }
  }
  red[tid] = gd; red[256 + tid] = dld; red[512 + tid] = step2; red[768 + tid] = x2; red[1024 + tid] = g2; red[1280 + tid] = gmax;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
      for (int k = 0; k < 5; ++k) red[k * 256 + tid] += red[k * 256 + tid + o];
      red[1280 + tid] = fmax(red[1280 + tid], red[1280 + tid + o]);
    }
    __syncthreads();
  }
  if (tid == 0) {
    double* h = v.scal + kNumScal;
    h[kScGd] = red[0]; h[kScDld] = red[256]; h[kScStep2] = red[512]; h[kScX2] = red[768]; h[kScG2] = red[1024];
    h[kScCost] = 0.0; h[kScGmax] = red[1280]; h[kScSq] = 0.0;
  }
}

// ------------------------------------------------------------------------------------------ trial point
// One wavefront per tile: delta_p = -L^-T (z + sum_tiles Y delta_s) (lanes = (tile, column), butterfly sum),
// T_trial = T exp(delta_p), residual sweep of the tile at the trial state.  The first tile of a frame
// also publishes the frame's trial pose and its step terms.
__global__ __launch_bounds__(256) void k_trial(DevView v) {
  const Ctrl* ct = v.ctrl;
  if (ct->done) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= v.n_tiles) return;
  const int cur = ct->cur;
  const int f = v.tile_frame[tile], c = v.tile_cam[tile];
  const int t0 = v.frame_tile_off[f], nt = v.frame_tile_off[f + 1] - t0;
  const double* fr = v.fr + (size_t)f * kFrStride;
  double y[6] = {0, 0, 0, 0, 0, 0};
  for (int idx = lane; idx < nt * 16; idx += 64) {
    const int t = idx >> 4, j = idx & 15;
    const int cc = v.tile_cam[t0 + t];
    const int nc = cam_ncols(v.cam_flags[cc], model_nk(v.cam_model[cc]));
    if (j < nc) {
      const double dj = v.delta_s[v.cam_col0[cc] + j];
      const double* Yt = v.Y + (size_t)(t0 + t) * kYStride + j;
#pragma unroll
      for (int k = 0; k < 6; ++k) y[k] += Yt[k * kUCols] * dj;
    }
  }
  double L[36];
  {
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) L[i * 6 + j] = (j <= i) ? fr[kFrL + (k++)] : 0.0;
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) y[k] = wave_allsum(y[k]) + fr[kFrZ + k];
  bwd_solve<6>(L, y);
  double d[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) d[i] = -y[i];
  const double* pin = v.poses[cur] + (size_t)f * kPoseStride;
  double Tin[7], Tout[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) Tin[i] = pin[i];
  se3_plus(Tin, d, Tout);
  const double* cam = v.cams[1 - cur] + (size_t)c * kCamStride;
  TileXf x;
  make_tile_xf(Tout, cam, &x);
  double K[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) K[i] = cam[kCamK + i];
  double cost, sq;
  res_tile_dispatch(v, v.cam_model[c], x, K, v.tile_off[tile], v.tile_off[tile + 1] - v.tile_off[tile], lane, &cost, &sq);
  if (lane == 0) {
    v.tile_trial[2 * tile] = ct->mult * cost;
    v.tile_trial[2 * tile + 1] = sq;
    if (tile == t0) {
      double* pout = v.poses[1 - cur] + (size_t)f * kPoseStride;
      double gd = 0, dld = 0, step2 = 0, x2 = 0, g2 = 0, gmax = 0;
#pragma unroll
      for (int i = 0; i < 7; ++i) { pout[i] = Tout[i]; const double e = Tout[i] - Tin[i]; step2 += e * e; x2 += Tin[i] * Tin[i]; }
      pout[7] = 0.0;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const double gi = fr[kFrG + i];
        gd += gi * d[i]; dld += fr[kFrLam + i] * d[i] * d[i]; g2 += gi * gi; gmax = fmax(gmax, fabs(gi));
      }
      double* o = v.fpart + (size_t)f * kNumScal;
      o[kScGd] = gd; o[kScDld] = dld; o[kScStep2] = step2; o[kScX2] = x2; o[kScG2] = g2; o[kScCost] = 0.0; o[kScGmax] = gmax; o[kScSq] = 0.0;
    }
  }
}

// ------------------------------------------------------------------------------------------ decision
// The Ceres trust-region bookkeeping (TrustRegionMinimizer + LevenbergMarquardtStrategy) and the
// reference's iteration callback (vicalibrator.h:690-721), run by one thread after every pass.
__device__ void trace_push(const DevView& v, Ctrl* c, const double* rec) {
  if (c->trace_len < c->trace_cap) {
    double* o = v.trace + (size_t)c->trace_len * kTraceCols;
    for (int i = 0; i < kTraceCols; ++i) o[i] = rec[i];
  }
  c->trace_len += 1;
  c->last_gnorm = rec[4];
}
__device__ void lm_decide(const DevView& v) {
  Ctrl* c = v.ctrl;
  const int D = v.D;
  const double* s = v.scal;
  const double* t = v.scal + kNumScal;
  const double R_cost = v.Sbuf[(size_t)D * D + 3 * D];
  const double R_gd = s[kScGd] + t[kScGd], R_dld = s[kScDld] + t[kScDld];
  const double R_step2 = s[kScStep2] + t[kScStep2], R_x2 = s[kScX2] + t[kScX2];
  const double R_gnorm = sqrt(s[kScG2] + t[kScG2]), R_gmax = fmax(s[kScGmax], t[kScGmax]);
  const double R_new_cost = s[kScCost];
  const bool fail = (v.flags[0] != 0) || (v.flags[1] != 0);
  v.flags[0] = 0; v.flags[1] = 0;
  c->passes += 1;
  c->res_sweeps += 1;
  if (c->need_lin) c->jac_sweeps += 1;
  if (c->hold) { c->cost = R_cost; c->gmax = R_gmax; c->gnorm = R_gnorm; return; }
  if (c->pending) {            // the pass linearised at the newly accepted point
    c->cost = R_cost; c->gmax = R_gmax; c->gnorm = R_gnorm;
    c->pend[1] = R_cost; c->pend[3] = R_gmax; c->pend[4] = R_gnorm;
    trace_push(v, c, c->pend);
    c->pending = 0;
    if (R_gmax <= c->gtol) { c->done = kDoneConvergence; return; }
  } else if (c->first) {
    c->cost = R_cost; c->gmax = R_gmax; c->gnorm = R_gnorm;
    const double rec[kTraceCols] = {0.0, R_cost, 0.0, R_gmax, R_gnorm, 0.0, 0.0, c->radius, 1.0, (double)c->stage};
    trace_push(v, c, rec);
    c->first = 0; c->init_scale = 0;
    if (R_gmax <= c->gtol) { c->done = kDoneConvergence; return; }
  }
  c->init_scale = 0;
  // iteration callback (vicalibrator.h:690-721): ++num_iterations_, stop if 0 < |g| < 1e-9
  c->num_callbacks += 1;
  if (c->last_gnorm > 0.0 && c->last_gnorm < 1e-9) { c->done = kDoneUserSuccess; return; }
  if (c->iter >= c->max_iters) { c->done = kDoneNoConvergence; return; }
  c->iter += 1;
  double rec[kTraceCols] = {(double)c->iter, c->cost, 0.0, c->gmax, c->gnorm, 0.0, 0.0, c->radius, 0.0, (double)c->stage};
  const double model_change = -0.5 * R_gd + 0.5 * R_dld;
  if (fail || !(model_change > 0.0)) {
    c->invalid += 1;
    if (c->invalid >= 5) { trace_push(v, c, rec); c->done = kDoneFailure; return; }
    c->radius *= 0.5; rec[7] = c->radius;
    trace_push(v, c, rec);
    c->need_lin = 0; c->reuse_diag = 1;
    return;
  }
  c->invalid = 0;
  rec[5] = sqrt(R_step2);
  const double xnorm = sqrt(R_x2);
  if (rec[5] <= c->ptol * (xnorm + c->ptol)) { trace_push(v, c, rec); c->done = kDoneConvergence; return; }
  rec[2] = c->cost - R_new_cost;
  if (fabs(rec[2]) < c->ftol * c->cost) { trace_push(v, c, rec); c->done = kDoneConvergence; return; }
  rec[6] = rec[2] / model_change;
  if (rec[6] > 1e-3) {
    c->cur = 1 - c->cur;
    const double q = 2.0 * rec[6] - 1.0;
    c->radius = fmin(1e16, c->radius / fmax(1.0 / 3.0, 1.0 - q * q * q));
    c->decrease_factor = 2.0;
    rec[7] = c->radius; rec[8] = 1.0;
    for (int i = 0; i < kTraceCols; ++i) c->pend[i] = rec[i];
    c->pending = 1; c->need_lin = 1; c->reuse_diag = 0;
  } else {
    c->radius = c->radius / c->decrease_factor; c->decrease_factor *= 2.0;
    rec[7] = c->radius;
    trace_push(v, c, rec);
    if (c->radius < 1e-32) { c->done = kDoneConvergence; return; }
    c->need_lin = 0; c->reuse_diag = 1;
  }
}

// mode 0: reduce + decide, 1: reduce only (an all-reduce follows), 2: decide only
__global__ __launch_bounds__(256) void k_final(DevView v, int mode) {
  __shared__ double red[256 * 7];
  if (v.ctrl->done) return;
  const int tid = threadIdx.x;
  if (mode != 2) {
    double s[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int f = tid; f < v.n_frames; f += 256) {
      const double* p = v.fpart + (size_t)f * kNumScal;
      s[0] += p[kScGd]; s[1] += p[kScDld]; s[2] += p[kScStep2]; s[3] += p[kScX2]; s[4] += p[kScG2];
      s[6] = fmax(s[6], p[kScGmax]);
    }
    for (int t = tid; t < v.n_tiles; t += 256) s[5] += v.tile_trial[2 * t];
    for (int k = 0; k < 7; ++k) red[k * 256 + tid] = s[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) {
        for (int k = 0; k < 6; ++k) red[k * 256 + tid] += red[k * 256 + tid + o];
        red[6 * 256 + tid] = fmax(red[6 * 256 + tid], red[6 * 256 + tid + o]);
      }
      __syncthreads();
    }
    if (tid == 0) {
      double* o = v.scal;
      o[kScGd] = red[0]; o[kScDld] = red[256]; o[kScStep2] = red[512]; o[kScX2] = red[768]; o[kScG2] = red[1024];
      o[kScCost] = 0.5 * red[1280]; o[kScGmax] = red[1536]; o[kScSq] = 0.0;
    }
  }
  if (mode != 1 && tid == 0) lm_decide(v);
}

// out[0] = 1/2 sum tile_trial cost, out[1] = sum of squared residuals
__global__ __launch_bounds__(256) void k_sum_tiles(DevView v, double* out) {
  __shared__ double red[512];
  const int tid = threadIdx.x;
  double a = 0, b = 0;
  for (int t = tid; t < v.n_tiles; t += 256) { a += v.tile_trial[2 * t]; b += v.tile_trial[2 * t + 1]; }
  red[tid] = a; red[256 + tid] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) { red[tid] += red[tid + o]; red[256 + tid] += red[256 + tid + o]; } __syncthreads(); }
  if (tid == 0) { out[0] = 0.5 * red[0]; out[1] = red[256]; }
}
// out[c*2] = sum of squared residuals of camera c, out[c*2+1] = corner count (per-camera RMSE, vicalibrator.h:958-971)
__global__ __launch_bounds__(256) void k_cam_sq(DevView v, double* out) {
  __shared__ double red[512];
  const int tid = threadIdx.x, c = blockIdx.x;
  double a = 0, b = 0;
  for (int t = tid; t < v.n_tiles; t += 256)
    if (v.tile_cam[t] == c) { a += v.tile_trial[2 * t + 1]; b += (double)(v.tile_off[t + 1] - v.tile_off[t]); }
  red[tid] = a; red[256 + tid] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) { red[tid] += red[tid + o]; red[256 + tid] += red[256 + tid + o]; } __syncthreads(); }
  if (tid == 0) { out[c * 2] = red[0]; out[c * 2 + 1] = red[256]; }
}

// ------------------------------------------------------------------------------------------ launchers
static inline int tiles_grid(const DevView& v) { return (v.n_tiles + 3) / 4; }

void launch_reproj_jac(const DevView& v, hipStream_t s) {
  if (v.n_tiles == 0) return;
  const size_t lds = 4 * 64 * kDotStride * sizeof(double);
  hipLaunchKernelGGL(k_reproj_jac, dim3(tiles_grid(v)), dim3(256), lds, s, v);
}
void launch_frame_prep(const DevView& v, hipStream_t s) {
  if (v.n_frames == 0) return;
  hipLaunchKernelGGL(k_frame_prep, dim3((v.n_frames + 3) / 4), dim3(256), 0, s, v);
}
void launch_schur_reduce(const DevView& v, hipStream_t s) {
  hipLaunchKernelGGL(k_schur_reduce, dim3(v.n_chunks), dim3(256), 0, s, v);
  hipLaunchKernelGGL(k_schur_final, dim3(1), dim3(256), 0, s, v);
}
void launch_reduced_solve(const DevView& v, hipStream_t s) {
  const size_t lds = ((size_t)(v.D + 1) * (v.D + 1) + 2 * (v.D + 1)) * sizeof(double);
  hipLaunchKernelGGL(k_reduced_solve, dim3(1), dim3(256), lds, s, v);
}
void launch_trial(const DevView& v, hipStream_t s) {
  if (v.n_tiles == 0) return;
  hipLaunchKernelGGL(k_trial, dim3(tiles_grid(v)), dim3(256), 0, s, v);
}
void launch_final(const DevView& v, int mode, hipStream_t s) {
  hipLaunchKernelGGL(k_final, dim3(1), dim3(256), 0, s, v, mode);
}
void launch_reproj_res(const DevView& v, int state, double mult, hipStream_t s) {
  if (v.n_tiles == 0) return;
  hipLaunchKernelGGL(k_reproj_res, dim3(tiles_grid(v)), dim3(256), 0, s, v, state, mult);
}
void launch_sum_tile_cost(const DevView& v, double* out, hipStream_t s) {
  hipLaunchKernelGGL(k_sum_tiles, dim3(1), dim3(256), 0, s, v, out);
}
void launch_cam_sq(const DevView& v, double* out, hipStream_t s) {
  hipLaunchKernelGGL(k_cam_sq, dim3(v.n_cams), dim3(256), 0, s, v, out);
}
void launch_outlier_mask(const DevView& v, int state, const double* thresh, unsigned char* mask, hipStream_t s) {
  if (v.n_tiles == 0) return;
  hipLaunchKernelGGL(k_outlier_mask, dim3(tiles_grid(v)), dim3(256), 0, s, v, state, thresh, mask);
}

}  // namespace vc
