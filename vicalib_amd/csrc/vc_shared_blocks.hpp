// vc_shared_blocks.hpp -- the reduced system's terms that do not come from the frame elimination: the cameras' blocks H_cc = P^T G P and the
// IMU-parameter block, from the fixed-order sums of the chunk records.  Included by vc_kernels.hip (k_reduced adds them itself on vision-only
// and sharded passes) and by vc_imu_kernels.hip (round 6: on single-process visual-inertial passes they are formed AHEAD of k_reduced -- side
// jobs of the chain's upper-level launches, which leave most of the chip idle: hadd_side_job below -- and k_reduced, one workgroup the whole chip
// waits for, only adds the finished record: DevView::hadd_early).
#pragma once
#include "vc_math.hpp"
#include "vc_device.h"
#include "vc_kutil.hpp"
namespace vc {
// Adds the blocks into S (D x D, both triangles) and g_red, SETS their diagonals in hd and their gradients in gs.  gsum: (C + 1) x kGStride
// sums (the cameras' Gram sums, then the IMU block's); P: C x 16 doubles of scratch (R_ck); camq: C x 4 (the cameras' rotations); cd, ipc: LDS
// copies of DevView::cd / imu_param_col.  nthreads: the workgroup's size (a multiple of 64, at most 256); workgroup barriers inside.
__device__ __forceinline__ void shared_blocks_phase(const DevView& v, int nthreads, const double* gsum, double* P, const double* camq, const CamDesc* cd, const int* ipc,
                                                    double* S, double* gred, double* hd, double* gs) {
  const int tid = threadIdx.x, D = v.D, C = v.n_cams;
  // camera blocks H_cc = P^T G P, g_c = P^T G[:, r]   (P: u-columns -> shared columns of the camera).  A column of P is either a
  // unit vector (translation and intrinsics columns) or -R's column a in rows 3..5 (rotation columns): at most three non-zeros, so
  // thread (b, a) forms its entry directly from at most nine entries of G -- no intermediate product, no barrier between the
  // cameras, every LDS read of the phase independent of the others (round 2's P / G P / P^T (G P) passes: 5.5k cycles per camera).
  if (tid < C) { double R[9]; quat_to_R(camq + 4 * tid, R);
#pragma unroll
    for (int k = 0; k < 9; ++k) P[tid * 16 + k] = R[k]; }
  __syncthreads();
  for (int tt = tid; tt < 256; tt += nthreads) {      // (the 16 x 16 grid of a camera's block: wave-uniform trip count)
    // the cameras' blocks are disjoint: every camera's old entries are requested before the first one is written back (one
    // memory round trip for the phase instead of one per camera)
    const int b = tt >> 4, a = tt & 15;
    double so[kMaxCams], go[kMaxCams];
#pragma unroll
    for (int c = 0; c < kMaxCams; ++c) {
      so[c] = 0.0; go[c] = 0.0;
      if (c < C) {
        const int nc = cam_ncols(cd[c].flags, model_nk(cd[c].model)), c0 = cd[c].col0;
        if (a < nc && b < nc && a >= b) so[c] = S[(c0 + b) * D + c0 + a];
        if (b == 15 && a < nc) go[c] = gred[c0 + a];
      }
    }
#pragma unroll
    for (int c = 0; c < kMaxCams; ++c)
      if (c < C) {
        const int flags = cd[c].flags, nk = model_nk(cd[c].model);
        const int nc = cam_ncols(flags, nk), c0 = cd[c].col0;
        const int nrot = (flags & kCamRotFree) ? 3 : 0, ntr = (flags & kCamTransFree) ? 3 : 0;
        const double* G = gsum + c * kGStride;
        const double* R = P + c * 16;
        // column q of P: rows r0 + {0, 1, 2} with coefficients cf[] (unit columns: one row, the other two coefficients zero and
        // their rows kept in range)
        int ra, rb; double ca[3], cb[3];
        {
          const bool rot = a < nrot;
          const int col = a < nc ? a : 0;
          ra = rot ? 3 : (col < nrot + ntr ? col - nrot : 6 + (col - nrot - ntr));
          ca[0] = rot ? -R[col] : 1.0; ca[1] = rot ? -R[3 + col] : 0.0; ca[2] = rot ? -R[6 + col] : 0.0;
        }
        {
          const bool rot = b < nrot;
          const int col = b < nc ? b : 0;
          rb = rot ? 3 : (col < nrot + ntr ? col - nrot : 6 + (col - nrot - ntr));
          cb[0] = rot ? -R[col] : 1.0; cb[1] = rot ? -R[3 + col] : 0.0; cb[2] = rot ? -R[6 + col] : 0.0;
        }
        const int sa = (a < nrot) ? 1 : 0, sb = (b < nrot) ? 1 : 0;      // row step: 1 for rotation columns, 0 for unit columns
        if (a < nc && b < nc && a >= b) {
          double s = 0.0;
#pragma unroll
          for (int x = 0; x < 3; ++x) {
            double t = 0.0;
#pragma unroll
            for (int y = 0; y < 3; ++y) t += ca[y] * G[(rb + sb * x) * 16 + ra + sa * y];
            s += cb[x] * t;
          }
          S[(c0 + b) * D + c0 + a] = so[c] + s;
          if (a == b) hd[c0 + a] = s; else S[(c0 + a) * D + c0 + b] = so[c] + s;      // (S is symmetric on entry and stays so)
        }
        if (b == 15 && a < nc) {
          double gca = 0.0;
#pragma unroll
          for (int y = 0; y < 3; ++y) gca += ca[y] * gram_grad(G, ra + sa * y, nk);
          gred[c0 + a] = go[c] + gca; gs[c0 + a] = gca;
        }
      }
  }
  __syncthreads();
  if (v.imu_on) {      // shared IMU parameters: sum over the blocks of their 15 x 15 Hessian and gradient
    const double* Hi = gsum + C * kGStride;
    for (int tt = tid; tt < 256; tt += nthreads) {
    const int a = tt >> 4, b = tt & 15;
    if (a < 15) {
      const int ca = ipc[a];
      if (ca >= 0) {
        if (b < 15) {
          const int cb = ipc[b];
          if (cb >= ca) {
            const double sn = S[ca * D + cb] + Hi[a * 16 + b];
            S[ca * D + cb] = sn;
            if (a == b) hd[ca] = Hi[a * 16 + a]; else S[cb * D + ca] = sn;
          }
        } else { gred[ca] += Hi[a * 16 + 15]; gs[ca] = Hi[a * 16 + 15]; }
      }
    }
    }
    __syncthreads();
  }
}

// ---- side jobs of the chain's upper-level launches (single-process visual-inertial passes) ---------------------------------------------------
// (1) the chunk records' entries behind S and g_red -- the cameras' Gram sums, the IMU block, the chunk costs: complete when the bottom level
//     (k_chain_l0 / k_chain_init) is -- summed in fixed order into part_total by extra workgroups of the first launch above the bottom level;
// (2) one extra workgroup of the launch after that forms the record H = [camera blocks + IMU block | their g | diag | g_s | cost] (the layout of
//     Sbuf) in DevView::hadd; k_reduced starts from Sbuf + H.
// nthreads: 128 or 256 (the hosting kernel's workgroup).
__device__ __forceinline__ void part_tail_sum_job(const DevView& v, int block, double* sl /* nthreads doubles */, int nthreads) {
  const int done = v.ctrl->done;
  const int tid = threadIdx.x, ent = tid & 15, ks = tid >> 4, SL = nthreads >> 4;
  const int stride = v.part_stride, n = v.n_part, e = v.D * v.D + v.D + block * 16 + ent;
  const bool live = e < stride;
  double s = 0.0;
  if (live) {
    const double* src = v.part + e;
#pragma unroll 8
    for (int k = ks; k < n; k += SL) s += src[(size_t)k * stride];
  }
  if (done) return;
  sl[tid] = s;
  __syncthreads();
  if (tid < 16 && live) {
    double t = sl[tid];
    for (int q = 1; q < SL; ++q) t += sl[q * 16 + tid];
    v.part_total[e] = t;
  }
}
// (3) flag hand-overs: k_part_sum's own work for the entries of S and g_red -- 16 entries x 32 slices of the chunk records per block, summed in
//     k_part_sum's order (a workgroup of 256 threads takes two slices per thread: Sbuf comes out identical to the bit) -- by workgroups of the
//     top level's launch that wait for the Gram chunks' ready words (DevView::part_ride).  sl: 512 doubles of LDS.
__device__ __forceinline__ void part_sum_ride_job(const DevView& v, int block, double* sl) {
  const int done = v.ctrl->done;
  const int tid = threadIdx.x, ent = tid & 15, ks = tid >> 4;
  const int stride = v.part_stride, D = v.D, DD = D * D, n = v.n_part, e = block * 16 + ent;
  int i = 0, j = 0;
  bool live = e < DD + D;
  if (e < DD) { i = e / D; j = e - i * D; live = (i >> 4) <= (j >> 4); }
  if (!done) {
    for (int c = tid; c < v.n_chunks; c += 256) {      // every chunk's record has been performed (bounded like every flag wait: vc_kutil.hpp)
      long long nspin = 0;
      for (;;) {
        const long long r = __hip_atomic_load(v.part_ready + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long m = sync_marked(v);
        if (r >= v.pass_id || (m != 0 && m <= v.sync_seq)) break;
        if (++nspin > v.sync_bound) { mark_sync_timeout(v, v.sync_seq); break; }
        __builtin_amdgcn_s_sleep(8);
      }
    }
  }
  __syncthreads();
  if (done) return;
  double s0 = 0.0, s1 = 0.0;
  if (live) {
    const double* src = v.part + e;
#pragma unroll 8
    for (int k = ks; k < n; k += 32) s0 += __hip_atomic_load(src + (size_t)k * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll 8
    for (int k = ks + 16; k < n; k += 32) s1 += __hip_atomic_load(src + (size_t)k * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  sl[tid] = s0; sl[256 + tid] = s1;
  __syncthreads();
  if (tid < 16 && live) {
    double t = sl[tid];
#pragma unroll
    for (int q = 1; q < 32; ++q) t += sl[q * 16 + tid];
    double* S = v.Sbuf;
    if (e < DD) { S[e] = -t; if ((i >> 4) < (j >> 4)) S[j * D + i] = -t; }
    else S[e] = -t;        // g_red follows S
  }
}
constexpr int kHaddLds = (kMaxCams + 1) * kGStride + kMaxCams * 16 + kMaxCams * 4 + 16 + 8;      // doubles of LDS for hadd_side_job
__device__ __forceinline__ void hadd_side_job(const DevView& v, double* lds, int nthreads) {
  const int tid = threadIdx.x, D = v.D, C = v.n_cams, stride = v.part_stride, nS = D * D + D;
  const Ctrl* ct = v.ctrl;
  const int cur = ct->cur;
  if (ct->done) return;
  double* gsum = lds;
  double* P = gsum + (kMaxCams + 1) * kGStride;
  double* camq = P + kMaxCams * 16;
  CamDesc* cd = reinterpret_cast<CamDesc*>(camq + kMaxCams * 4);
  int* ipc = reinterpret_cast<int*>(cd + kMaxCams);
  const double* ptot = v.part_total;
  double* H = v.hadd;
  for (int e = nS + tid; e < stride - 2; e += nthreads) gsum[e - nS] = ptot[e];
  if (tid < C * 4) camq[tid] = v.cams[cur][(size_t)(tid >> 2) * kCamStride + (tid & 3)];
  if (tid < kMaxCams) cd[tid] = v.cd[tid];
  if (tid >= 64 && tid < 64 + 15) ipc[tid - 64] = v.imu_param_col[tid - 64];
  for (int e = tid; e < nS + 2 * D; e += nthreads) H[e] = 0.0;
  if (tid == 0) { H[nS + 2 * D] = 0.5 * ptot[stride - 1]; H[nS + 2 * D + 1] = 0.0; }      // chunk costs
  __syncthreads();
  shared_blocks_phase(v, nthreads, gsum, P, camq, cd, ipc, H, H + D * D, H + nS, H + nS + D);
}
}  // namespace vc
