// vc_kernels.hip -- CDNA4 (gfx950, wave64) kernels of the batched Levenberg-Marquardt engine.
//
// Replaces, on the device, what the reference does per residual block inside ceres::Solve
// (vicalibrator.h:956): AutoDiffCostFunction::Evaluate of ImuReprojectionCostFunctor
// (ceres-cost-functions.h:350-373) + loss correction + block-sparse J^T J, the elimination of the
// per-frame pose blocks (Ceres' sparse Cholesky), the dense solve on the shared parameters, the
// manifold update (local-param-se3.h) and the cost evaluation of the trial point.
//
//  K1 k_reproj_jac      one wavefront per (frame,camera) tile: closed-form unique-column Jacobian rows
//                       per corner (lane = corner), rows staged in wave-private LDS, tile Gram block
//                       G = sum w u^T u accumulated on the matrix pipe (v_mfma_f64_16x16x4_f64)
//  K2 k_reproj_res      residual-only sweep (trial cost, RMSE)
//  K3 k_frame_prep      thread per frame: H_pp, g_p from the tile Gram blocks, damping, 6x6 Cholesky,
//                       Y = L^-1 W per tile
//     k_schur_reduce / k_schur_final   S = H_ss - sum Y^T Y, g_red, in fixed order (bit-stable)
//  K4 k_reduced_solve   one workgroup: damped Cholesky in LDS on the shared parameters
//  K5 k_backsub_update  thread per frame: delta_p, T <- T exp(delta_p) into the trial buffer
//     k_reduce_scalars  model decrease terms, step / state norms, gradient norms, trial cost
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
__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
  return x;
}
__device__ __forceinline__ double wave_max(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmax(x, __shfl_down(x, o, 64));
  return x;
}

// ------------------------------------------------------------------------------------------ K1
template <int MODEL>
__device__ __forceinline__ void jac_tile_body(const DevView& v, const LmArgs& a, int tile, int lane, double* wl) {
  const int f = v.tile_frame[tile], c = v.tile_cam[tile];
  const int off = v.tile_off[tile], cnt = v.tile_off[tile + 1] - off;
  const double* pose = v.poses[a.cur] + (size_t)f * kPoseStride;
  const double* cam = v.cams[a.cur] + (size_t)c * kCamStride;
  TileXf x;
  make_tile_xf(pose, cam, &x);
  double K[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) K[i] = cam[kCamK + i];
  v4d acc = {0.0, 0.0, 0.0, 0.0};
  double cost = 0.0;
  double* mine = wl + lane * kDotStride;
  for (int base = 0; base < cnt; base += 64) {
    const int d = base + lane;
    if (d < cnt) {
      const double2 uv = v.obs_uv[off + d];
      const double* pw = v.points + 3 * (size_t)v.obs_pt[off + d];
      cost += corner_rows<MODEL>(x, K, pw, uv.x, uv.y, a.mult, mine, mine + 16);
    } else {
#pragma unroll
      for (int i = 0; i < 32; ++i) mine[i] = 0.0;
    }
    wave_lds_sync();
    const int nd = min(64, cnt - base);
    const int nsteps = (nd + 1) >> 1;          // one MFMA covers 2 corners x 2 residual rows (K = 4)
    const double* src = wl + (lane >> 5) * kDotStride + (lane & 31);
    for (int k = 0; k < nsteps; ++k) {
      const double u = src[2 * k * kDotStride];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(u, u, acc, 0, 0, 0);
    }
    wave_lds_sync();
  }
  double* G = v.G + (size_t)tile * kGStride;
#pragma unroll
  for (int i = 0; i < 4; ++i) G[((lane >> 4) + 4 * i) * 16 + (lane & 15)] = acc[i];
  cost = wave_sum(cost);
  if (lane == 0) v.tile_cost[tile] = cost;
}

__global__ __launch_bounds__(256) void k_reproj_jac(DevView v, LmArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= v.n_tiles) return;
  double* wl = lds + wave * 64 * kDotStride;
  switch (v.cam_model[v.tile_cam[tile]]) {   // wave-uniform
    case kFov: jac_tile_body<kFov>(v, a, tile, lane, wl); break;
    case kPoly2: jac_tile_body<kPoly2>(v, a, tile, lane, wl); break;
    case kPoly3: jac_tile_body<kPoly3>(v, a, tile, lane, wl); break;
    case kKb4: jac_tile_body<kKb4>(v, a, tile, lane, wl); break;
    default: jac_tile_body<kLinear>(v, a, tile, lane, wl); break;
  }
}

// ------------------------------------------------------------------------------------------ K2
template <int MODEL>
__device__ __forceinline__ void res_tile_body(const DevView& v, int state, double mult, int tile, int lane) {
  const int f = v.tile_frame[tile], c = v.tile_cam[tile];
  const int off = v.tile_off[tile], cnt = v.tile_off[tile + 1] - off;
  const double* pose = v.poses[state] + (size_t)f * kPoseStride;
  const double* cam = v.cams[state] + (size_t)c * kCamStride;
  TileXf x;
  make_tile_xf(pose, cam, &x);
  double K[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) K[i] = cam[kCamK + i];
  double cost = 0.0, sq = 0.0;
  for (int d = lane; d < cnt; d += 64) {
    const double2 uv = v.obs_uv[off + d];
    const double* pw = v.points + 3 * (size_t)v.obs_pt[off + d];
    double r[2];
    cost += corner_residual<MODEL>(x, K, pw, uv.x, uv.y, r);
    sq += r[0] * r[0] + r[1] * r[1];
  }
  cost = wave_sum(cost);
  sq = wave_sum(sq);
  if (lane == 0) { v.tile_cost[tile] = mult * cost; v.tile_sq[tile] = sq; }
}
__global__ __launch_bounds__(256) void k_reproj_res(DevView v, int state, double mult) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= v.n_tiles) return;
  switch (v.cam_model[v.tile_cam[tile]]) {
    case kFov: res_tile_body<kFov>(v, state, mult, tile, lane); break;
    case kPoly2: res_tile_body<kPoly2>(v, state, mult, tile, lane); break;
    case kPoly3: res_tile_body<kPoly3>(v, state, mult, tile, lane); break;
    case kKb4: res_tile_body<kKb4>(v, state, mult, tile, lane); break;
    default: res_tile_body<kLinear>(v, state, mult, tile, lane); break;
  }
}

// per-corner outlier mask (RemoveOutliers, vicalibrator.h:859-916): |r| > thresh[cam]
template <int MODEL>
__device__ __forceinline__ void mask_tile_body(const DevView& v, int state, const double* thresh, unsigned char* mask, int tile, int lane) {
  const int f = v.tile_frame[tile], c = v.tile_cam[tile];
  const int off = v.tile_off[tile], cnt = v.tile_off[tile + 1] - off;
  TileXf x;
  const double* cam = v.cams[state] + (size_t)c * kCamStride;
  make_tile_xf(v.poses[state] + (size_t)f * kPoseStride, cam, &x);
  double K[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) K[i] = cam[kCamK + i];
  const double th = thresh[c];
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
  switch (v.cam_model[v.tile_cam[tile]]) {
    case kFov: mask_tile_body<kFov>(v, state, thresh, mask, tile, lane); break;
    case kPoly2: mask_tile_body<kPoly2>(v, state, thresh, mask, tile, lane); break;
    case kPoly3: mask_tile_body<kPoly3>(v, state, thresh, mask, tile, lane); break;
    case kKb4: mask_tile_body<kKb4>(v, state, thresh, mask, tile, lane); break;
    default: mask_tile_body<kLinear>(v, state, thresh, mask, tile, lane); break;
  }
}

// ------------------------------------------------------------------------------------------ K3
__global__ __launch_bounds__(64) void k_frame_prep(DevView v, LmArgs a) {
  const int f = blockIdx.x * 64 + threadIdx.x;
  if (f >= v.n_frames) return;
  double H[36], g[6];
#pragma unroll
  for (int i = 0; i < 36; ++i) H[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) g[i] = 0.0;
  const int t0 = v.frame_tile_off[f], t1 = v.frame_tile_off[f + 1];
  const double* cams = v.cams[a.cur];
  for (int t = t0; t < t1; ++t) {
    const int c = v.tile_cam[t];
    double R[9];
    quat_to_R(cams + (size_t)c * kCamStride, R);
    tile_to_frame_blocks(v.G + (size_t)t * kGStride, R, model_nk(v.cam_model[c]), v.cam_flags[c], H, g, nullptr);
  }
  double* fr = v.fr + (size_t)f * kFrStride;
  double lam[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const double hd = H[i * 6 + i];
    if (a.init_scale) v.fscale2[(size_t)f * 6 + i] = jacobi_scale2(hd);
    const double sc2 = v.fscale2[(size_t)f * 6 + i];
    if (!a.reuse_diag) v.fdiag[(size_t)f * 6 + i] = lm_clamped_diag(hd, sc2);
    lam[i] = v.fdiag[(size_t)f * 6 + i] / (a.radius * sc2);
    H[i * 6 + i] = hd + lam[i];
  }
  if (!chol_small<6>(H)) {
    atomicAdd(&v.flags[0], 1);
#pragma unroll
    for (int i = 0; i < 36; ++i) H[i] = (i % 7 == 0) ? 1.0 : 0.0;
  }
  double z[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) z[i] = g[i];
  fwd_solve<6>(H, z);
  {
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) fr[kFrL + k++] = H[i * 6 + j];
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) { fr[kFrZ + i] = z[i]; fr[kFrG + i] = g[i]; fr[kFrLam + i] = lam[i]; }
  // second pass: W per tile written straight into Y, then Y <- L^-1 W column by column
  for (int t = t0; t < t1; ++t) {
    const int c = v.tile_cam[t];
    double R[9];
    quat_to_R(cams + (size_t)c * kCamStride, R);
    double* Yt = v.Y + (size_t)t * kYStride;
    tile_to_frame_blocks(v.G + (size_t)t * kGStride, R, model_nk(v.cam_model[c]), v.cam_flags[c], nullptr, nullptr, Yt);
    for (int j = 0; j < kUCols; ++j) {
      double col[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) col[r] = Yt[r * kUCols + j];
      fwd_solve<6>(H, col);
#pragma unroll
      for (int r = 0; r < 6; ++r) Yt[r * kUCols + j] = col[r];
    }
  }
}

// Partial Schur sums of one frame chunk: part = [ sum Y^T Y (upper, D x D) | sum Y^T z (D) | per-camera sum of G (C x 256) ]
__global__ __launch_bounds__(256) void k_schur_reduce(DevView v) {
  const int chunk = blockIdx.x, tid = threadIdx.x;
  const int f0 = chunk * v.chunk_frames, f1 = min(f0 + v.chunk_frames, v.n_frames);
  const int D = v.D, C = v.n_cams;
  double* part = v.part + (size_t)chunk * v.part_stride;
  {
    double gsum[kMaxCams];
#pragma unroll
    for (int c = 0; c < kMaxCams; ++c) gsum[c] = 0.0;
    const int t0 = v.frame_tile_off[f0], t1 = v.frame_tile_off[f1];
    for (int t = t0; t < t1; ++t) {
      const double gv = v.G[(size_t)t * kGStride + tid];
      const int c = v.tile_cam[t];
#pragma unroll
      for (int k = 0; k < kMaxCams; ++k) gsum[k] += (k == c) ? gv : 0.0;
    }
#pragma unroll
    for (int c = 0; c < kMaxCams; ++c) if (c < C) part[D * D + D + c * kGStride + tid] = gsum[c];
  }
  for (int e = tid; e < D * D + D; e += 256) {
    double s = 0.0;
    if (e < D * D) {
      const int ra = e / D, rb = e % D;
      if (rb >= ra) {
        const int ca = v.col_cam[ra], cb = v.col_cam[rb];
        if (ca >= 0 && cb >= 0) {
          const int la = v.col_local[ra], lb = v.col_local[rb];
          for (int f = f0; f < f1; ++f) {
            const int ta = v.frame_cam_tile[f * C + ca], tb = v.frame_cam_tile[f * C + cb];
            if (ta < 0 || tb < 0) continue;
            const double* ya = v.Y + (size_t)ta * kYStride + la;
            const double* yb = v.Y + (size_t)tb * kYStride + lb;
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
        for (int f = f0; f < f1; ++f) {
          const int ta = v.frame_cam_tile[f * C + ca];
          if (ta < 0) continue;
          const double* ya = v.Y + (size_t)ta * kYStride + la;
          const double* z = v.fr + (size_t)f * kFrStride + kFrZ;
#pragma unroll
          for (int k = 0; k < 6; ++k) s += ya[k * kUCols] * z[k];
        }
      }
    }
    part[e] = s;
  }
}

// Sbuf = [ S = H_ss - sum Y^T Y (full symmetric, undamped) | g_red | diag(H_ss) | g_s | cost, 0 ]
__global__ __launch_bounds__(256) void k_schur_final(DevView v, LmArgs a) {
  __shared__ double gsum[kMaxCams * kGStride];
  __shared__ double hcc[kMaxCams * 256];
  __shared__ double gc[kMaxCams * 16];
  __shared__ double red[256];
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
  // cost = 1/2 sum of tile costs, fixed-shape tree
  {
    double s = 0.0;
    for (int t = tid; t < v.n_tiles; t += 256) s += v.tile_cost[t];
    red[tid] = s;
  }
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  if (tid < C) {
    double R[9];
    quat_to_R(v.cams[a.cur] + (size_t)tid * kCamStride, R);
    cam_block_from_gsum(gsum + tid * kGStride, R, model_nk(v.cam_model[tid]), v.cam_flags[tid], hcc + tid * 256, gc + tid * 16);
  }
  __syncthreads();
  if (tid == 0) { sc[0] = 0.5 * red[0]; sc[1] = 0.0; }
  for (int i = tid; i < D; i += 256) { hd[i] = 0.0; gs[i] = 0.0; }
  __syncthreads();
  for (int c = 0; c < C; ++c) {
    const int nc = cam_ncols(v.cam_flags[c], model_nk(v.cam_model[c]));
    const int c0 = v.cam_col0[c];
    for (int e = tid; e < nc * nc; e += 256) {
      const int i = e / nc, j = e % nc;
      if (j >= i) S[(c0 + i) * D + c0 + j] += hcc[c * 256 + i * 16 + j];
      if (i == j) hd[c0 + i] = hcc[c * 256 + i * 16 + i];
    }
    for (int i = tid; i < nc; i += 256) { gred[c0 + i] += gc[c * 16 + i]; gs[c0 + i] = gc[c * 16 + i]; }
  }
  __syncthreads();
  for (int e = tid; e < D * D; e += 256) {
    const int i = e / D, j = e % D;
    if (j < i) S[e] = S[j * D + i];
  }
}

// ------------------------------------------------------------------------------------------ K4
__global__ __launch_bounds__(256) void k_reduced_solve(DevView v, LmArgs a) {
  extern __shared__ __attribute__((aligned(16))) double M[];
  const int tid = threadIdx.x, D = v.D;
  double* x = M + D * D;
  const double* S = v.Sbuf;
  const double* gred = S + D * D;
  const double* hd = gred + D;
  for (int e = tid; e < D * D; e += 256) M[e] = S[e];
  __syncthreads();
  for (int i = tid; i < D; i += 256) {
    if (a.init_scale) v.sscale2[i] = jacobi_scale2(hd[i]);
    const double sc2 = v.sscale2[i];
    if (!a.reuse_diag) v.sdiag[i] = lm_clamped_diag(hd[i], sc2);
    const double lam = v.sdiag[i] / (a.radius * sc2);
    v.slam[i] = lam;
    M[i * D + i] += lam;
    x[i] = -gred[i];
  }
  __syncthreads();
  for (int j = 0; j < D; ++j) {
    if (tid == 0) {
      double d = M[j * D + j];
      if (!(d > 0.0)) { v.flags[1] = 1; d = 1.0; }
      M[j * D + j] = sqrt(d);
    }
    __syncthreads();
    const double piv = M[j * D + j];
    for (int i = j + 1 + tid; i < D; i += 256) M[i * D + j] /= piv;
    __syncthreads();
    const int n = D - j - 1;
    for (int idx = tid; idx < n * n; idx += 256) {
      const int i = j + 1 + idx / n, k = j + 1 + idx % n;
      if (k <= i) M[i * D + k] -= M[i * D + j] * M[k * D + j];
    }
    __syncthreads();
  }
  for (int j = 0; j < D; ++j) {
    if (tid == 0) x[j] /= M[j * D + j];
    __syncthreads();
    const double xj = x[j];
    for (int i = j + 1 + tid; i < D; i += 256) x[i] -= M[i * D + j] * xj;
    __syncthreads();
  }
  for (int j = D - 1; j >= 0; --j) {
    if (tid == 0) x[j] /= M[j * D + j];
    __syncthreads();
    const double xj = x[j];
    for (int i = tid; i < j; i += 256) x[i] -= M[j * D + i] * xj;
    __syncthreads();
  }
  for (int i = tid; i < D; i += 256) v.delta_s[i] = x[i];
}

// ------------------------------------------------------------------------------------------ K5
__global__ __launch_bounds__(64) void k_backsub_update(DevView v, LmArgs a) {
  const int f = blockIdx.x * 64 + threadIdx.x;
  double gd = 0, dld = 0, step2 = 0, x2 = 0, g2 = 0, gmax = 0;
  if (f < v.n_frames) {
    const double* fr = v.fr + (size_t)f * kFrStride;
    double L[36], y[6];
    {
      int k = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) L[i * 6 + j] = (j <= i) ? fr[kFrL + (k++)] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] = fr[kFrZ + i];
    const int t0 = v.frame_tile_off[f], t1 = v.frame_tile_off[f + 1];
    for (int t = t0; t < t1; ++t) {
      const int c = v.tile_cam[t];
      const int nc = cam_ncols(v.cam_flags[c], model_nk(v.cam_model[c]));
      const double* ds = v.delta_s + v.cam_col0[c];
      const double* Yt = v.Y + (size_t)t * kYStride;
      for (int j = 0; j < nc; ++j) {
        const double dj = ds[j];
#pragma unroll
        for (int k = 0; k < 6; ++k) y[k] += Yt[k * kUCols + j] * dj;
      }
    }
    bwd_solve<6>(L, y);
    double d[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) d[i] = -y[i];
    const double* pin = v.poses[a.cur] + (size_t)f * kPoseStride;
    double* pout = v.poses[1 - a.cur] + (size_t)f * kPoseStride;
    double Tin[7], Tout[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) Tin[i] = pin[i];
    se3_plus(Tin, d, Tout);
#pragma unroll
    for (int i = 0; i < 7; ++i) { pout[i] = Tout[i]; const double e = Tout[i] - Tin[i]; step2 += e * e; x2 += Tin[i] * Tin[i]; }
    pout[7] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const double gi = fr[kFrG + i];
      gd += gi * d[i]; dld += fr[kFrLam + i] * d[i] * d[i]; g2 += gi * gi; gmax = fmax(gmax, fabs(gi));
    }
  }
  gd = wave_sum(gd); dld = wave_sum(dld); step2 = wave_sum(step2); x2 = wave_sum(x2); g2 = wave_sum(g2); gmax = wave_max(gmax);
  if (threadIdx.x == 0) {
    double* p = v.fpart + (size_t)blockIdx.x * kNumScal;
    p[kScGd] = gd; p[kScDld] = dld; p[kScStep2] = step2; p[kScX2] = x2; p[kScG2] = g2; p[kScCost] = 0.0; p[kScGmax] = gmax; p[kScSq] = 0.0;
  }
}

// scal[0..7]: sums over this rank's frames (+ trial cost of its tiles); scal[8..15]: shared-parameter terms
// (identical on every rank), written by k_shared_update.
__global__ __launch_bounds__(256) void k_reduce_scalars(DevView v, LmArgs a) {
  __shared__ double red[256 * 7];
  const int tid = threadIdx.x;
  double s[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int b = tid; b < v.n_fblocks; b += 256) {
    const double* p = v.fpart + (size_t)b * kNumScal;
    s[0] += p[kScGd]; s[1] += p[kScDld]; s[2] += p[kScStep2]; s[3] += p[kScX2]; s[4] += p[kScG2];
    s[6] = fmax(s[6], p[kScGmax]);
  }
  for (int t = tid; t < v.n_tiles; t += 256) s[5] += v.tile_cost[t];
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

// Trial state of the shared parameters (cams[1-cur] <- Plus(cams[cur], delta_s)) and their scalar terms
// scal[8..15] (identical on every rank).  Runs before the trial residual sweep.
__global__ void k_shared_update(DevView v, LmArgs a) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  {
    const int D = v.D;
    const double* gs = v.Sbuf + D * D + 2 * D;
    double gd = 0, dld = 0, step2 = 0, x2 = 0, g2 = 0, gmax = 0;
    for (int i = 0; i < D; ++i) {
      const double d = v.delta_s[i], g = gs[i];
      gd += g * d; dld += v.slam[i] * d * d; g2 += g * g; gmax = fmax(gmax, fabs(g));
    }
    for (int c = 0; c < v.n_cams; ++c) {
      const double* cin = v.cams[a.cur] + (size_t)c * kCamStride;
      double* cout = v.cams[1 - a.cur] + (size_t)c * kCamStride;
      for (int i = 0; i < kCamStride; ++i) cout[i] = cin[i];
      const int flags = v.cam_flags[c], nk = model_nk(v.cam_model[c]);
      int col = v.cam_col0[c];
      if (flags & kCamRotFree) {
        double q[4];
        so3_plus(cin, v.delta_s + col, q);
        for (int i = 0; i < 4; ++i) { cout[i] = q[i]; const double e = q[i] - cin[i]; step2 += e * e; x2 += cin[i] * cin[i]; }
        col += 3;
      }
      if (flags & kCamTransFree) {
        for (int i = 0; i < 3; ++i) { const double d = v.delta_s[col + i]; cout[4 + i] = cin[4 + i] + d; step2 += d * d; x2 += cin[4 + i] * cin[4 + i]; }
        col += 3;
      }
      if (flags & kCamKFree) {
        for (int i = 0; i < nk; ++i) { const double d = v.delta_s[col + i]; cout[kCamK + i] = cin[kCamK + i] + d; step2 += d * d; x2 += cin[kCamK + i] * cin[kCamK + i]; }
      }
    }
    double* h = v.scal + kNumScal;
    h[kScGd] = gd; h[kScDld] = dld; h[kScStep2] = step2; h[kScX2] = x2; h[kScG2] = g2; h[kScCost] = 0.0; h[kScGmax] = gmax; h[kScSq] = 0.0;
  }
}

// out[0] = 1/2 sum tile_cost, out[1] = sum tile_sq
__global__ __launch_bounds__(256) void k_sum_tiles(DevView v, double* out) {
  __shared__ double red[512];
  const int tid = threadIdx.x;
  double a = 0, b = 0;
  for (int t = tid; t < v.n_tiles; t += 256) { a += v.tile_cost[t]; b += v.tile_sq[t]; }
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
    if (v.tile_cam[t] == c) { a += v.tile_sq[t]; b += (double)(v.tile_off[t + 1] - v.tile_off[t]); }
  red[tid] = a; red[256 + tid] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) { red[tid] += red[tid + o]; red[256 + tid] += red[256 + tid + o]; } __syncthreads(); }
  if (tid == 0) { out[c * 2] = red[0]; out[c * 2 + 1] = red[256]; }
}

// ------------------------------------------------------------------------------------------ launchers
static inline int tiles_grid(const DevView& v) { return (v.n_tiles + 3) / 4; }

void launch_reproj_jac(const DevView& v, const LmArgs& a, hipStream_t s) {
  if (v.n_tiles == 0) return;
  const size_t lds = 4 * 64 * kDotStride * sizeof(double);
  hipLaunchKernelGGL(k_reproj_jac, dim3(tiles_grid(v)), dim3(256), lds, s, v, a);
}
void launch_reproj_res(const DevView& v, int state, double mult, hipStream_t s) {
  if (v.n_tiles == 0) return;
  hipLaunchKernelGGL(k_reproj_res, dim3(tiles_grid(v)), dim3(256), 0, s, v, state, mult);
}
void launch_frame_prep(const DevView& v, const LmArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_frame_prep, dim3((v.n_frames + 63) / 64), dim3(64), 0, s, v, a);
}
void launch_schur_reduce(const DevView& v, const LmArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_schur_reduce, dim3(v.n_chunks), dim3(256), 0, s, v);
  hipLaunchKernelGGL(k_schur_final, dim3(1), dim3(256), 0, s, v, a);
}
void launch_reduced_solve(const DevView& v, const LmArgs& a, hipStream_t s) {
  if (v.D == 0) return;
  const size_t lds = ((size_t)v.D * v.D + v.D) * sizeof(double);
  hipLaunchKernelGGL(k_reduced_solve, dim3(1), dim3(256), lds, s, v, a);
}
void launch_backsub_update(const DevView& v, const LmArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_backsub_update, dim3(v.n_fblocks), dim3(64), 0, s, v, a);
}
void launch_shared_update(const DevView& v, const LmArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_shared_update, dim3(1), dim3(64), 0, s, v, a);
}
void launch_reduce_scalars(const DevView& v, const LmArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_reduce_scalars, dim3(1), dim3(256), 0, s, v, a);
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
