// vc_imu_kernels.hip -- inertial stages of the LM engine (gfx950, wave64).
//
// With IMU blocks in the problem (SetupProblem, vicalibrator.h:607-637, :651-655) consecutive frames are
// coupled, so the per-frame elimination of the vision-only path becomes the factorisation of a
// block-tridiagonal chain (9 x 9 blocks: pose 6 + velocity 3) bordered by the shared parameters.
//
//  k_imu_block     the part of the preintegration that does not depend on any pose, one wavefront per block: eight lanes per dual
//                  direction (gyro biases / scale factors, time offset; accelerometer parameters analytically beside them), lane =
//                  two sample intervals (RK4 steps from the identity state), inclusive DPP scan over the eight lanes
//  k_imu_jac       half a wavefront per IMU block; lane = local column of the block, carried as a dual number through the
//                  application of the block's delta to the start state and the residual's tail -- the lane-parallel form of
//                  ceres::Jet<double,35> (vc_imu.hpp, "delta form"); then Cauchy(100) weight and the 33 x 33 weighted
//                  J^T J / J^T r of the block
//  k_imu_weights   16 lanes per block: UpdateImuWeights (vicalibrator.h:723-799) interval-parallel (vc_imu_weights.hpp)
//  k_chain_init    wavefront per frame: 9 x 9 diagonal block (visual tiles + two IMU blocks), coupling to the
//                  next frame, dense border row W (9 x D) and gradient; damping; per-chunk sums of the
//                  camera Gram blocks and of the IMU shared-parameter block
//  k_chain_fwd     one level of the partitioned chain elimination: a wavefront eliminates the interior frames of a group
//                  of 8 (L, X_s = L^-1 C, X_n = L^-1 B, Y = L^-1 [W | g]); the group's first frame survives to the next level
//  k_chain_gram    sum over all frames of [Y | z]^T [Y | z] on the matrix pipe (v_mfma_f64_16x16x4_f64)
//  k_chain_l0      (round 5) k_chain_init's work and the bottom level of the elimination in one launch: the frames' images stay in LDS
//                  (narrow borders, at most two cameras, single process)
//  k_chain_top_gram  the top level of the elimination and, beside it in the same launch, k_chain_gram's sums of the frames below it
//  k_chain_back    back-substitution, one level per launch; the bottom level also writes the trial poses / velocities
//  k_chain_back_levels  ... all levels below the top in one launch (ready words between the workgroups)
//  k_chain_back_path    (round 5) the whole back-substitution in one launch without hand-overs: a workgroup per bottom-level group
//                  recomputes the levels above it (narrow borders, single process)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include "vc_math.hpp"
#include "vc_imu.hpp"
#include "vc_imu_weights.hpp"
#include "vc_device.h"
#include "vc_kutil.hpp"
#include "vc_reduced_tail.hpp"
#include "vc_shared_blocks.hpp"

namespace vc {

__device__ __forceinline__ ImuView imu_view(const DevView& v) { ImuView b = {v.imu_t, v.imu_w, v.imu_a, v.n_imu, v.imu_avg_dt}; return b; }

// ------------------------------------------------------------------------------------------ IMU Jacobian
constexpr int kImuJacLds = 34 * 9 + 6;            // [34][9]: the block's 33 local columns and, as a 34th, the residual itself
// (row a, column b) of every entry of the compact block record (vc_device.h: kSeg*), b = 33: gradient -- one table look-up per entry
struct SegTab {
  unsigned short v[kSegStride];
  constexpr SegTab() : v() { for (int e = 0; e < kSegLen; ++e) { int a = 0, b = 0; seg_entry(e, &a, &b); v[e] = (unsigned short)(a | (b << 8)); } }
};
__constant__ SegTab d_seg_tab = SegTab();
constexpr int kSegIters = (kSegLen + 31) / 32;      // entries of the record per lane (32 lanes per block)
// ... and the other way round for the matrix-pipe form of the block's J^T J (round 6): the 34 columns (33 local columns + the residual) are
// three column tiles of 16; tile pair (I, J) of v_mfma_f64_16x16x4 leaves entry (a, b) = (16 I + 4 g + lane / 16, 16 J + lane % 16) in
// accumulator register g of `lane` -- v[I * 3 + J][lane][g] is that entry's place in the compact record (0xffff: the record has no such entry)
#ifndef VC_IJ_MFMA
#define VC_IJ_MFMA 1      // (0: the block's J^T J as 25 nine-term dot products per lane out of LDS, the form of rounds 1-5 -- A/B builds)
#endif
struct SegInv {
  unsigned short v[9][64][4];
  constexpr SegInv() : v() {
    for (int t = 0; t < 9; ++t) for (int l = 0; l < 64; ++l) for (int g = 0; g < 4; ++g) v[t][l][g] = 0xffff;
    for (int e = 0; e < kSegLen; ++e) {
      int a = 0, b = 0; seg_entry(e, &a, &b);
      const int I = a / 16, J = b / 16, rt = a % 16;
      v[I * 3 + J][(rt % 4) * 16 + b % 16][rt / 4] = (unsigned short)e;
    }
  }
};
__constant__ SegInv d_seg_inv = SegInv();
// The sweep proper.  Two IMU blocks per wavefront, lane = local column of the block: frame j's pose (6), frame j-1's pose (6) and
// velocity (3), gravity (2), biases (6), scale factors (6), time offset -- 30 lanes put the block's delta on the start state and
// run the residual's tail under one dual direction each (vc_imu.hpp: imu_block_final_direction); the three columns of frame j's
// velocity are -W^T rows 6..8, written directly.  Then Cauchy(100) weight and the 33 x 33 weighted J^T J / J^T r of the block.
// trial = 0: at the accepted state, when the control record asks for a linearisation; trial = 1: at the trial state into buffer
// 1 - cur (cost = the block's trial cost; the blocks are the next linearisation if the step is accepted) -- see k_reproj_jac.
// phase stamps of the first wavefront (profiling builds only, -DVC_IJ_STAMPS): 100 MHz s_memrealtime ticks in dbg[0..7]
#ifdef VC_IJ_STAMPS
#define IJSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) v.dbg[i] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define IJSTAMP(i) do { } while (0)
#endif
__global__ __launch_bounds__(256) void k_imu_jac(DevView v, int wr, int trial) {
  __shared__ double sh[8 * kImuJacLds];
  __shared__ double s_wq[8 * 84];                  // the block's 9 x 9 weight, staged once (round 6: every lane loaded all 81 entries itself)
  // (round 6 also tried requesting everything ahead of the control record -- the two frames' states and the gravity record of BOTH state
  //  buffers, one value per lane, the buffer in use to LDS --: 23.1 us against 21.4 on the same box, reverted)
  IJSTAMP(0);
  const Ctrl* ct = v.ctrl;
  if (ct->done || (!trial && !ct->need_lin)) {
    // (a finished solve: the workgroup still counts itself in the second count -- k_final of this pass waits for it, so that the main
    //  stream never runs ahead of this one into the next solve)
    if (trial && v.final_wait > 0 && threadIdx.x == 0) __hip_atomic_fetch_add(v.sync_flags + 11, 1LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int half = lane >> 5, l = lane & 31;
  const int n_blocks = v.n_frames - 1;
  const int s_raw = (blockIdx.x * 4 + wave) * 2 + half;       // block s couples frames s -> s+1
  const bool exists = s_raw < n_blocks;
  const int s = exists ? s_raw : n_blocks - 1;                 // a half past the end shadows the last block and stores nothing
  const int cur = trial ? 1 - ct->cur : ct->cur, j = s + 1;
  double* Jl = sh + (wave * 2 + half) * kImuJacLds;            // [34][9] local columns: cur9 | prev9 | imu15 | residual
  const double* T2 = v.poses[cur] + (size_t)j * kPoseStride;
  const double* T1 = v.poses[cur] + (size_t)(j - 1) * kPoseStride;
  const double* v2 = v.vel[cur] + (size_t)j * 4;
  const double* v1 = v.vel[cur] + (size_t)(j - 1) * 4;
  const double* wq_g = v.wsqrtb[wr] + (size_t)s * 81;
  double* wq = s_wq + (wave * 2 + half) * 84;
  {
    double t3[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) { const int idx = l + 32 * t; t3[t] = wq_g[idx < 81 ? idx : 0]; }
#pragma unroll
    for (int t = 0; t < 3; ++t) { const int idx = l + 32 * t; if (idx < 81) wq[idx] = t3[t]; }
  }
  const double* brec = v.imu_delta_blk + (size_t)s * kBlockDeltaStride;
  double r[9], dr[9];
  const int col = (l < 6) ? l : (l < 30 ? l + 3 : -1);         // lanes 30, 31 carry the values only
  const bool valid = brec[10] >= 0.0;
#if VC_IJ_MFMA
  unsigned long long seg_at[9];      // where the lane's four accumulator entries of every tile pair go in the record (tile pair (2, 0) has none)
#pragma unroll
  for (int t = 0; t < 9; ++t) seg_at[t] = t == 6 ? ~0ull : *reinterpret_cast<const unsigned long long*>(&d_seg_inv.v[t][lane][0]);
#else
  int seg_ab[kSegIters];
#pragma unroll
  for (int it = 0; it < kSegIters; ++it) { const int e = l + 32 * it; seg_ab[it] = d_seg_tab.v[e < kSegLen ? e : 0]; }
#endif
  wave_lds_sync();
  IJSTAMP(1);
  imu_block_final_direction(valid, brec, wq, v.rotation_only, T2, T1, v2, v1, v.imu_grav + cur * 16, col, r, dr);
  IJSTAMP(2);
  if (col >= 0) {
#pragma unroll
    for (int k = 0; k < 9; ++k) Jl[col * 9 + k] = dr[k];
  }
  if (l < 3) {                                                 // frame j's velocity: r = W^T raw, raw[6 + l] = v_pred - v2
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const bool zeroed = v.rotation_only && (k < 3 || k >= 6);
      Jl[(6 + l) * 9 + k] = (valid && !zeroed) ? -wq[(6 + l) * 9 + k] : 0.0;
    }
  }
  if (l == 31) {                                               // the residual rides as a 34th column: J^T r = its products with the others
#pragma unroll
    for (int k = 0; k < 9; ++k) Jl[33 * 9 + k] = r[k];
  }
  wave_lds_sync();
  double ss = 0.0;
#pragma unroll
  for (int k = 0; k < 9; ++k) ss += r[k] * r[k];
  double rho, rho1;
  loss_cauchy100(ss, &rho, &rho1);
  const double w = ct->imu_mult * rho1;
  if (trial) {                 // the workgroup's share of the trial cost (8 blocks, fixed order); all threads reach this barrier
    __shared__ double s_cost[8];
    if (l == 0) s_cost[wave * 2 + half] = exists ? ct->imu_mult * rho : 0.0;
    __syncthreads();
    if (threadIdx.x == 0) {
      const double wc = ((s_cost[0] + s_cost[1]) + (s_cost[2] + s_cost[3])) + ((s_cost[4] + s_cost[5]) + (s_cost[6] + s_cost[7]));
      if (v.final_wait > 0) {
        // flag hand-overs: k_final (main stream) takes the trial cost as soon as every workgroup has delivered its share -- a
        // device-coherent store and a count, no cache write-back -- and decides while this kernel still writes its records
        // (round 6 tried moving the count behind this thread's record stores -- the store's acknowledgement holds wavefront 0 back by
        //  ~2 us here: the kernel ended 1.2 us earlier and k_final 3.2 us later behind it; reverted)
        __hip_atomic_store(v.wg_imu_trial + blockIdx.x, wc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);          // the store has been performed before the count moves
        __hip_atomic_fetch_add(v.sync_flags + 4, 1LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else v.wg_imu_trial[blockIdx.x] = wc;
    }
  }
  IJSTAMP(3);
  // flag hand-overs, trial sweep: the records go out as device-coherent stores and the workgroup counts itself a second time once they
  // have been performed -- k_final ends on that count (round 4: on a flag a one-thread kernel raised behind this one, 6 us after this
  // kernel's end, which since round 5's leaner k_final was what the pass ended on)
  const bool counted2 = trial && v.final_wait > 0;
#if VC_IJ_MFMA
  // J^T J and J^T r of the block in the compact record, entry e = w * <column a, column b> with the residual as column 33 -- on the matrix
  // pipe (round 6).  As 25 nine-term dot products per lane the phase was 450 LDS reads per lane, 6.2 of the kernel's ~20 us; here the whole
  // wavefront takes its two blocks one after the other: the block's [34][9] image read ONCE as nine operand registers (column tile T, rows
  // 4 ks .. 4 ks + 3: the same register is the A operand of tile pairs (T, .) and the B operand of (., T)), 8 tile pairs x 3 k-steps of
  // v_mfma_f64_16x16x4, and every accumulator entry goes to the place the inverse table names.  (a, b) and (b, a) are the same products in
  // the same order: the record's symmetric blocks stay symmetric to the bit.
  {
    const int i16 = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (!__builtin_amdgcn_readlane((int)exists, 32 * h)) continue;               // (wave-uniform)
      const double wh = readlane_f64(w, 32 * h);
      const int sb = __builtin_amdgcn_readlane(s, 32 * h);
      const double* Jb = sh + (wave * 2 + h) * kImuJacLds;
      double op[3][3];
#pragma unroll
      for (int T = 0; T < 3; ++T)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          const int c = 16 * T + i16, rr = 4 * ks + kq;
          const bool in = c < 34 && rr < 9;
          const double x = Jb[in ? c * 9 + rr : 0];
          op[T][ks] = in ? x : 0.0;
        }
      double* rec = v.segb[cur] + (size_t)sb * kSegStride;
#pragma unroll
      for (int I = 0; I < 3; ++I)
#pragma unroll
        for (int J = 0; J < 3; ++J) {
          if (I == 2 && J == 0) continue;
          v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int ks = 0; ks < 3; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(op[I][ks], op[J][ks], acc, 0, 0, 0);
          const unsigned long long at = seg_at[I * 3 + J];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const unsigned e = (unsigned)(at >> (16 * g)) & 0xffffu;
            if (e != 0xffffu) { if (counted2) __hip_atomic_store(rec + e, wh * acc[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else rec[e] = wh * acc[g]; }
          }
        }
    }
    if (exists && l == 0) { if (counted2) __hip_atomic_store(v.seg_costb[cur] + s, ct->imu_mult * rho, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else v.seg_costb[cur][s] = ct->imu_mult * rho; }
  }
#else
  if (exists) {
  // J^T J and J^T r of the block in the compact record: entry e = w * <column a, column b> with the residual as column 33
  // (statically unrolled, the (row, column) pairs of the lane's 25 entries requested at the kernel's entry: as a loop every entry
  //  waited for its own table look-up -- 25 dependent global round trips, 10 of the kernel's 24 us at cfg3)
  double* rec = v.segb[cur] + (size_t)s * kSegStride;
#pragma unroll
  for (int it = 0; it < kSegIters; ++it) {
    const int e = l + 32 * it;
    const int ab = seg_ab[it], a = ab & 255, bb = ab >> 8;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) acc += Jl[a * 9 + k] * Jl[bb * 9 + k];
    if (e < kSegLen) { if (counted2) __hip_atomic_store(rec + e, w * acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else rec[e] = w * acc; }
  }
  if (l == 0) { if (counted2) __hip_atomic_store(v.seg_costb[cur] + s, ct->imu_mult * rho, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else v.seg_costb[cur][s] = ct->imu_mult * rho; }
  }
#endif
  if (counted2) {
    __builtin_amdgcn_s_waitcnt(0);          // this wavefront's stores have been performed
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(v.sync_flags + 11, 1LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  IJSTAMP(4);
}

// UpdateImuWeights from the accepted state (vicalibrator.h:723-799), interval-parallel (vc_imu_weights.hpp, second half).
// 16 lanes per IMU block, four blocks per wavefront, one wavefront per workgroup.  A round handles 16 sample intervals of every
// block (cfg3: 12 per block, one round), lane i = interval i:
//   1. the interval's RK4 delta from the identity state (values only, vc_imu.hpp);
//   2. inclusive scan of the deltas over the 16 lanes (composition is associative): lane i learns the state its interval starts
//      from, q_start * Q_prefix, without walking the block;
//   3. the interval's maps, formed ONCE -- F (41 structural entries) and Q = G R G^T (55) -- from that quaternion;
//   4. the recurrence Sigma <- F Sigma F^T + Q unrolled: Sigma_end = sum_k P_k Q_k P_k^T with P_k the product of the maps after
//      interval k.  A suffix scan over the lanes gives P_k (the block form of F is closed under products), every lane conjugates
//      its own Q, a butterfly sums the 55 entries: no step of the recurrence waits for the one before it.
// All lane exchanges are DPP row operations (rows of 16 lanes = the groups): no LDS round trip anywhere in 1-4.
// Then the projection through the residual's Jacobian in registers (every lane holds Sigma), J Sigma J^T = L L^T by nine lanes
// (lane = row, pivots and pivot columns by 16-lane shuffles) and the stored factor W = L^-T:  W W^T = (J Sigma J^T)^-1 exactly
// as for the reference's symmetric square root (vicalibrator.h:783-796), and cost, gradient and Gauss-Newton Hessian of the
// block depend on W only through W W^T -- same optimisation, no 9x9 eigen-decomposition.
constexpr int kWLds = 184;                       // per block: the factor's rows for the inverse (81) | frame poses j-1, j (16) | v_{j-1} (4) | IMU parameters (15) | Sigma (55)
constexpr int kWPose = 88, kWVel = 104, kWImu = 108, kWSig = 128;

// DPP row operations on doubles (two 32-bit moves).  row_shr:n -- lane i of a row reads lane i - n; row_shl:n -- lane i + n;
// lanes without a source keep `old`.
template <int CTRL> __device__ __forceinline__ double dpp_f64(double old, double x) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(x), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(x), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int N> __device__ __forceinline__ double row_shr(double old, double x) { return dpp_f64<0x110 + N>(old, x); }
template <int N> __device__ __forceinline__ double row_shl(double old, double x) { return dpp_f64<0x100 + N>(old, x); }
// row permutation of a double (every lane has a source: no `old` operand, no copy)
template <int CTRL> __device__ __forceinline__ double dpp_perm(double x) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// sums over the 16 lanes of a row of N values at once, the same bits in every lane (each level adds the partner's value to the
// lane's own: commutative).  Level by level over all values: the N exchanges of a level are independent instructions.
template <int CTRL, int N> __device__ __forceinline__ void row_add_level(double* x) {
  double y[N];
#pragma unroll
  for (int e = 0; e < N; ++e) y[e] = dpp_perm<CTRL>(x[e]);
#pragma unroll
  for (int e = 0; e < N; ++e) x[e] += y[e];
}
template <int N> __device__ __forceinline__ void row_allsum_n(double* x) {
  row_add_level<0xB1, N>(x);                     // quad_perm [1 0 3 2]
  row_add_level<0x4E, N>(x);                     // quad_perm [2 3 0 1]
  row_add_level<0x141, N>(x);                    // row_half_mirror
  row_add_level<0x140, N>(x);                    // row_mirror
}
__device__ __forceinline__ double shfl16(double x, int src) { return __shfl(x, src, 16); }
__device__ __forceinline__ void delta_identity(DeltaAcc<double>* a) {
#pragma unroll
  for (int k = 0; k < 3; ++k) { a->q[k] = 0.0; a->p[k] = 0.0; a->v[k] = 0.0; }
  a->q[3] = 1.0; a->t = 0.0;
}
// a <- a followed by x
__device__ __forceinline__ void delta_then(DeltaAcc<double>* a, const DeltaAcc<double>& x) {
  PoseV<double> d;
#pragma unroll
  for (int k = 0; k < 4; ++k) d.q[k] = x.q[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) { d.p[k] = x.p[k]; d.v[k] = x.v[k]; }
  imu_delta_append(a, d, x.t);
}
// one level of the inclusive scan: lanes >= N of the row take (lane - N's value) followed by their own
template <int N> __device__ __forceinline__ void delta_scan_level(DeltaAcc<double>* X, int c) {
  DeltaAcc<double> A;
#pragma unroll
  for (int k = 0; k < 4; ++k) A.q[k] = row_shr<N>(X->q[k], X->q[k]);
#pragma unroll
  for (int k = 0; k < 3; ++k) { A.p[k] = row_shr<N>(X->p[k], X->p[k]); A.v[k] = row_shr<N>(X->v[k], X->v[k]); }
  A.t = row_shr<N>(X->t, X->t);
  delta_then(&A, *X);
  if (c >= N) *X = A;
}
// one level of the suffix scan of the maps: (lane + N's map) applied after the lane's own; the last N lanes of the row receive
// the identity, whose product with their own map is that map bit for bit
template <int N> __device__ __forceinline__ void map_scan_level(double* S) {
  double B[kWMapF], O[kWMapF];
  w_map_identity(B);
#pragma unroll
  for (int e = 0; e < kWMapF; ++e) B[e] = row_shl<N>(B[e], S[e]);
  w_map_compose(B, S, O);
#pragma unroll
  for (int e = 0; e < kWMapF; ++e) S[e] = O[e];
}

// imu_range (vc_imu.hpp) with its two look-ups side by side: even lanes find the element of t0, odd lanes that of t1 -- the same
// dependent loads, once instead of twice in a row -- and neighbours swap results.  All lanes of the wavefront must be active.
__device__ __forceinline__ ImuRange imu_range_lanes(const ImuView& b, double t0, double t1, double off, int lane) {
  ImuRange r; r.valid = 0; r.k0 = 0; r.k1 = -1; r.i0 = r.i1 = 0; r.first_end = r.last_end = 0;
  if (b.n < 2) return r;
  const bool odd = lane & 1;
  int idx, endc, le;
  imu_element_index(b, odd ? t1 : t0, off, &idx, &endc, &le);
  const int idx_o = __shfl_xor(idx, 1, 64), endc_o = __shfl_xor(endc, 1, 64), le_o = __shfl_xor(le, 1, 64);
  r.valid = (t0 >= b.t[0] + off && t0 <= b.t[b.n - 1] + off) ? 1 : 0;      // HasElement :122-125
  r.i0 = odd ? idx_o : idx; r.first_end = odd ? endc_o : endc;
  r.i1 = odd ? idx : idx_o; r.last_end = odd ? endc : endc_o;
  r.k0 = r.i0 + 1;
  r.k1 = odd ? le : le_o;
  if (r.k1 < r.i0) r.k1 = r.i0;
  return r;
}

// ------------------------------------------------------------------------------------------ IMU block deltas
// The pose-independent half of the IMU sweep (vc_imu.hpp, "delta form") in one launch, nothing but the block records written: the
// interval deltas never leave the registers (round 3 wrote a 1120-byte record per sample interval with k_imu_delta and read it
// straight back with k_imu_block: 53 of the sweep's 73 MB per pass at BASELINE cfg3, behind a chain of ~12 dependent appends per
// block).  ONE WAVEFRONT PER IMU BLOCK, eight groups of eight lanes:
//   group g = 0..5   dual direction = gyro bias / scale factor g, and beside it -- no dual number needed, the velocity and position
//                    deltas are linear in the accelerometer inputs -- the partials along accelerometer bias / scale factor g;
//   group 6          dual direction = time offset (moves the interpolated end samples and the end intervals' lengths);
//   group 7          spare (shadows group 6, stores nothing);  the values ride in every group, group 0 stores them.
// Lane l of a group takes sample intervals 2 l + 1 and 2 l + 2 of the block (vc_imu.hpp: imu_interval_delta_ga, the RK4 step from
// the identity state written out) and appends the second to the first; an inclusive scan over the group's eight lanes (composition is
// associative; DPP row shifts, three levels, no LDS) leaves the block's delta in lane 7.  Blocks with more than 16 intervals run in
// rounds, lane 7 appending the rounds' totals (kept in LDS between rounds).  Round 4's first cut had one lane per (interval,
// record column) -- 14 x 16 lanes, 3.5 wavefronts per block, every lane repeating the values under its own direction and the scan
// costing as much as the steps: 57 us at cfg3 against 53 for the two kernels it replaced; this form runs a quarter of the wavefronts.
// Depends on the shared IMU parameters only: in a solve it runs behind the reduced solve, next to the chain's back-substitution.
// An empty sample range is flagged by T = -1 in the record.
// an IntervalDeltaT (vc_imu.hpp) as kDtDoubles doubles: values Q P V T | tangent dth dp dv dt | ap | av
__device__ __forceinline__ void delta_t_pack(const IntervalDeltaT& X, double* f) {
#pragma unroll
  for (int k = 0; k < 4; ++k) f[k] = X.d.q[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) { f[4 + k] = X.d.p[k]; f[7 + k] = X.d.v[k]; f[11 + k] = X.dth[k]; f[14 + k] = X.dp[k]; f[17 + k] = X.dv[k]; f[21 + k] = X.ap[k]; f[24 + k] = X.av[k]; }
  f[10] = X.d.t; f[20] = X.dt;
}
__device__ __forceinline__ void delta_t_unpack(const double* f, IntervalDeltaT* X) {
#pragma unroll
  for (int k = 0; k < 4; ++k) X->d.q[k] = f[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) { X->d.p[k] = f[4 + k]; X->d.v[k] = f[7 + k]; X->dth[k] = f[11 + k]; X->dp[k] = f[14 + k]; X->dv[k] = f[17 + k]; X->ap[k] = f[21 + k]; X->av[k] = f[24 + k]; }
  X->d.t = f[10]; X->dt = f[20];
}
template <int N> __device__ __forceinline__ void delta_t_scan_level(IntervalDeltaT* X, int l) {
  double f[kDtDoubles];
  delta_t_pack(*X, f);
#pragma unroll
  for (int k = 0; k < kDtDoubles; ++k) f[k] = row_shr<N>(f[k], f[k]);
  IntervalDeltaT A;
  delta_t_unpack(f, &A);
  imu_delta_t_then(&A, *X);
  if (l >= N) *X = A;                            // (the first N lanes of a group have no partner: what the shift brought them belongs to the group below)
}
// phase stamps of one wavefront in the middle of the grid (profiling builds only, -DVC_IB_STAMPS): 100 MHz s_memrealtime ticks in dbg[0..9]
#ifdef VC_IB_STAMPS
#define IBSTAMP(i) do { if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) v.dbg[i] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define IBSTAMP(i) do { } while (0)
#endif
#ifndef VC_IMU_BLOCK_WAVES
#define VC_IMU_BLOCK_WAVES 2
#endif
#ifndef VC_IMU_BLOCK_PARK
#define VC_IMU_BLOCK_PARK 1
#endif
#ifdef VC_IMU_BLOCK_EU      // (A/B builds: a register budget of 512 / n per wavefront -- n = 3: 168 registers, 268 B of scratch)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(VC_IMU_BLOCK_EU, VC_IMU_BLOCK_EU))) void k_imu_block(DevView v, int trial) {
#else
__global__ __launch_bounds__(256, VC_IMU_BLOCK_WAVES) void k_imu_block(DevView v, int trial) {
#endif
  __shared__ double s_carry[32 * kDtDoubles];
#if VC_IMU_BLOCK_PARK
  // the first interval's delta while the second is formed: tangent and accelerometer partials per lane (16 doubles), the VALUES once per
  // interval -- they are the same numbers in all eight groups, group 0 writes them: 42 KB per workgroup with the carry, not 62: two
  // workgroups fit a CU's LDS beside a workgroup of the back-substitution (68 KB at BASELINE cfg3).  (Its registers then hold the grid to
  // one workgroup per CU there -- 252 + 216 of a SIMD's 512 -- and two rounds; a step under ~144 registers spills: HISTORY round 6)
  __shared__ double s_park[256 * (kDtDoubles - 11)];
  __shared__ double s_parkv[4 * 8 * 11];
#endif
  IBSTAMP(0);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (scalar: everything per block stays out of the vector registers)
  const int g = lane >> 3, l = lane & 7;
  const int n_blocks = v.n_frames - 1;
  const int s_raw = blockIdx.x * 4 + wave;                             // block s couples frames s -> s + 1
  const int s = s_raw < n_blocks ? s_raw : n_blocks - 1;               // (a wavefront past the end shadows the last block and stores nothing)
  // requested together with the control record, not behind it (a wavefront's life is a chain of dependent memory round trips -- control
  // record, time offset, sample range, samples: ~7 of its 14 us; this is one of them): the frame times and the time offset of BOTH buffers
  const double toff0 = v.imus[0][14], toff1 = v.imus[1][14];
  const double t_start = v.frame_time[s], t_end = v.frame_time[s + 1];
  const Ctrl* ct = v.ctrl;
  const int c_done = ct->done, c_need = ct->need_lin, c_cur = ct->cur;
  if (c_done || (!trial && !c_need)) return;
  IBSTAMP(1);
  const int cur = trial ? 1 - c_cur : c_cur;
  const double* im = v.imus[cur];
  const double toff = cur ? toff1 : toff0;
  const ImuView buf = imu_view(v);
  ImuRange rg = imu_range_lanes(buf, t_start, t_end, toff, lane);
  rg.valid = __builtin_amdgcn_readfirstlane(rg.valid); rg.i0 = __builtin_amdgcn_readfirstlane(rg.i0); rg.i1 = __builtin_amdgcn_readfirstlane(rg.i1);
  rg.k0 = __builtin_amdgcn_readfirstlane(rg.k0); rg.k1 = __builtin_amdgcn_readfirstlane(rg.k1);
  rg.first_end = __builtin_amdgcn_readfirstlane(rg.first_end); rg.last_end = __builtin_amdgcn_readfirstlane(rg.last_end);
  double* rec = v.imu_delta_blk + (size_t)s * kBlockDeltaStride;
  IBSTAMP(2);
  if (rg.valid) {                                                      // (wave-uniform)
    const int n_int = (rg.k1 - rg.k0 + 1) + 1;                         // intervals between the n_int + 1 range elements
    const int gsel = g < 7 ? g : 6;
    double* cr = s_carry + (threadIdx.x >> 3) * kDtDoubles;
    IntervalDeltaT X;
#pragma unroll 1
    for (int base = 0; base < n_int; base += 16) {
      {
        // the first interval's delta waits in LDS while the second is formed (27 doubles that would otherwise sit -- or spill --
        // under the second RK4 step); a loop that is not unrolled: one copy of the step, nothing of the second interval scheduled
        // into the first
        IntervalDeltaT Y;
#if VC_IMU_BLOCK_PARK
        double* pk = s_park + threadIdx.x;
        double* pv = s_parkv + (wave * 8 + l) * 11;
#endif
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
          // (the group index made opaque per iteration: otherwise everything the step derives from it -- seed masks, unit vector,
          //  selects -- is hoisted out of the loops and sits in registers across both steps and the scan)
          int gs = gsel;
          asm volatile("" : "+v"(gs));
          imu_interval_delta_t(buf, rg, mk(toff, gs == 6 ? 1.0 : 0.0), t_start, t_end, base + 2 * l + 1 + half, n_int, im + 2, im + 8, gs, &Y);
          IBSTAMP(3 + half);
          if (half == 0) {
#if VC_IMU_BLOCK_PARK
            double f[kDtDoubles];
            delta_t_pack(Y, f);
#pragma unroll
            for (int k = 11; k < kDtDoubles; ++k) pk[(k - 11) * 256] = f[k];
            if (g == 0) {
#pragma unroll
              for (int k = 0; k < 11; ++k) pv[k] = f[k];
            }
#else
            X = Y;
#endif
          }
        }
#if VC_IMU_BLOCK_PARK
        wave_lds_sync_local();               // (group 0's values are read by the other groups' lanes)
        double f[kDtDoubles];
#pragma unroll
        for (int k = 0; k < 11; ++k) f[k] = pv[k];
#pragma unroll
        for (int k = 11; k < kDtDoubles; ++k) f[k] = pk[(k - 11) * 256];
        delta_t_unpack(f, &X);
        wave_lds_sync_local();               // (... before the next round's first interval overwrites them)
#endif
        imu_delta_t_then(&X, Y);
      }
      IBSTAMP(5);
      delta_t_scan_level<1>(&X, l); delta_t_scan_level<2>(&X, l); delta_t_scan_level<4>(&X, l);
      IBSTAMP(6);
      if (l == 7) {                                                    // the round's total; the rounds before it rest in LDS
        if (base > 0) {
          IntervalDeltaT A;
          double f[kDtDoubles];
#pragma unroll
          for (int k = 0; k < kDtDoubles; ++k) f[k] = cr[k];
          delta_t_unpack(f, &A);
          imu_delta_t_then(&A, X);
          X = A;
        }
        if (base + 16 < n_int) {
          double f[kDtDoubles];
          delta_t_pack(X, f);
#pragma unroll
          for (int k = 0; k < kDtDoubles; ++k) cr[k] = f[k];
        }
      }
    }
    if (l == 7 && s_raw < n_blocks) imu_block_record_store(X, g, rec);
    IBSTAMP(7);
  } else if (lane == 0 && s_raw < n_blocks) rec[10] = -1.0;
  // the gravity record of this state for k_imu_jac (one lane of the spare group)
  if (blockIdx.x == 0 && threadIdx.x == 63) imu_gravity_record(im, v.imu_grav + cur * 16);
  // flag hand-overs, trial point: k_imu_jac behind this kernel needs the main stream's trial poses.  One thread of this kernel
  // waits for their flag before the kernel ends -- the kernel boundary then orders k_imu_jac behind it like any other kernel,
  // without a waiting kernel of its own (5 us on this stream's queue, which is the critical one at the end of a small pass)
  if (trial && v.block_wait > 0 && blockIdx.x == 0 && threadIdx.x == 0) spin_until_flag<false>(v, 2, v.block_wait);
}

// phase stamps of the first wavefront (profiling builds only, -DVC_W_STAMPS): 100 MHz s_memrealtime ticks in dbg[0..15]
#ifdef VC_W_STAMPS
#define WSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) v.dbg[i] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define WSTAMP(i) do { } while (0)
#endif
__global__ __launch_bounds__(64) void k_imu_weights(DevView v, int wr) {
  __shared__ __attribute__((aligned(16))) double w_lds[4 * kWLds];
  const Ctrl* ct = v.ctrl;
  const int lane = threadIdx.x;
  const int grp = lane >> 4, c = lane & 15;      // c: interval of the round; row of the projection afterwards
  const int n_blocks = v.n_frames - 1;
  const int s_raw = blockIdx.x * 4 + grp;
  const bool exists = s_raw < n_blocks;
  const int s = exists ? s_raw : n_blocks - 1;   // groups past the end shadow the last block and store nothing
  double* L = w_lds + grp * kWLds;
  // the buffer being written must end up complete: blocks that keep their weight (no samples, singular projection) and passes
  // queued behind a finished solve copy the current one -- at the end, off the path of the blocks that get a new weight
  // A pass that is void after a flag time-out (vc_kutil.hpp), and every pass queued behind it, must leave BOTH buffers alone: the
  // resumed pass linearises with the weights its predecessor left in wsqrtb[wr of the void pass], which is the buffer the next
  // queued pass would write (round 4: a time-out in the middle of a solve resumed with the void pass's own weights there -- same
  // state, one update too far: costs off by 1e-5 .. 5e-4 for a few iterations.  A first-pass time-out hides it: both buffers
  // hold the same numbers).  The host takes the buffer index back from its ring (resume_after_sync_timeout).
  if (v.sync_seq > 0) { const long long m = sync_marked(v); if (m != 0 && m <= v.sync_seq) return; }      // wave-uniform
  if (ct->done == kDoneSyncTimeout) return;
  if (ct->done || !v.weights_on) {               // wave-uniform
    if (exists) for (int e = c; e < 81; e += 16) v.wsqrtb[1 - wr][(size_t)s * 81 + e] = v.wsqrtb[wr][(size_t)s * 81 + e];
    return;
  }
  WSTAMP(0);
  const int j = s + 1;
  // The block's state, one value per lane: frames j - 1 and j are 16 consecutive doubles, the IMU parameters 15; both state buffers
  // are requested before the control record says which one is current (one round of loads less), the current one goes to LDS and
  // every lane reads what it needs when it needs it -- nothing of this sits in registers across the phases below.
  {
    const double p0 = v.poses[0][(size_t)(j - 1) * kPoseStride + c], p1 = v.poses[1][(size_t)(j - 1) * kPoseStride + c];
    const double u0 = v.vel[0][(size_t)(j - 1) * 4 + (c & 3)], u1 = v.vel[1][(size_t)(j - 1) * 4 + (c & 3)];
    const int ci = c < 15 ? c : 14;
    const double i0 = v.imus[0][ci], i1 = v.imus[1][ci];
    const bool one = ct->cur != 0;
    L[kWPose + c] = one ? p1 : p0;
    if (c < 4) L[kWVel + c] = one ? u1 : u0;
    if (c < 15) L[kWImu + c] = one ? i1 : i0;
  }
  const double t_start = v.frame_time[j - 1], t_end = v.frame_time[j];
  wave_lds_sync();
  double b[6], sf[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) { b[i] = L[kWImu + 2 + i]; sf[i] = L[kWImu + 8 + i]; }
  const double toff = L[kWImu + 14];
  const ImuView buf = imu_view(v);
  const ImuRange rg = imu_range_lanes(buf, t_start, t_end, toff, lane);
  const bool live = exists && rg.valid;          // an empty range keeps its current weight (vicalibrator.h:731-733)
  const double sg2 = v.gyro_sigma * v.gyro_sigma, sa2 = v.accel_sigma * v.accel_sigma;
  const int n_int = live ? (rg.k1 - rg.k0 + 1) + 1 : 0;      // intervals between the n_meas = n_int + 1 range elements
  int n_max = n_int;                             // the wave runs to its longest group
#pragma unroll
  for (int o = 32; o >= 16; o >>= 1) n_max = max(n_max, __shfl_xor(n_max, o, 64));
  DeltaAcc<double> carry;                        // the block's delta up to the current round
  delta_identity(&carry);
  // Sigma (packed lower triangle) rests in LDS between the rounds and until the projection: 55 doubles that would otherwise sit
  // in registers (or, as they did, in scratch memory) across the maps of every round
  const double g0[3] = {0.0, 0.0, 0.0};
  WSTAMP(1);
#pragma unroll 1
  for (int base = 0; base < n_max; base += 16) {
    const int m = base + c + 1;                  // interval m runs from range element m - 1 to m
    Meas<double> z0, z1;
#pragma unroll
    for (int k = 0; k < 3; ++k) { z0.w[k] = z0.a[k] = z1.w[k] = z1.a[k] = 0.0; }
    z0.time = z1.time = 0.0;
    if (m <= n_int) { imu_range_get_flat(buf, rg, toff, t_start, t_end, m - 1, &z0); imu_range_get_flat(buf, rg, toff, t_start, t_end, m, &z1); }
    const bool act = z1.time != z0.time;         // a zero-length interval is skipped (types.h:150-152)
    WSTAMP(2);
    // 1. the interval's delta from the identity
    DeltaAcc<double> X;
    {
      PoseV<double> d;
#pragma unroll
      for (int k = 0; k < 3; ++k) { d.q[k] = 0.0; d.p[k] = 0.0; d.v[k] = 0.0; }
      d.q[3] = 1.0;
      imu_rk4_step(&d, z0, z1, b, sf, g0);       // (returns at once for a skipped interval: the identity)
#pragma unroll
      for (int k = 0; k < 4; ++k) X.q[k] = d.q[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) { X.p[k] = d.p[k]; X.v[k] = d.v[k]; }
      X.t = z1.time - z0.time;
    }
    WSTAMP(3);
    // 2. inclusive scan over the group's 16 lanes
    delta_scan_level<1>(&X, c); delta_scan_level<2>(&X, c); delta_scan_level<4>(&X, c); delta_scan_level<8>(&X, c);
    double q_st[4];
    {
      double qe[4], ident[4] = {0.0, 0.0, 0.0, 1.0};      // everything before this lane's interval: the carry, then lanes 0 .. c - 1
#pragma unroll
      for (int k = 0; k < 4; ++k) ident[k] = row_shr<1>(ident[k], X.q[k]);
      quat_mul(carry.q, ident, qe);
      const double q1[4] = {L[kWPose + 0], L[kWPose + 1], L[kWPose + 2], L[kWPose + 3]};
      quat_mul(q1, qe, q_st);
      DeltaAcc<double> Tt;                       // the round's total (lane 15) joins the carry
#pragma unroll
      for (int k = 0; k < 4; ++k) Tt.q[k] = shfl16(X.q[k], 15);
#pragma unroll
      for (int k = 0; k < 3; ++k) { Tt.p[k] = shfl16(X.p[k], 15); Tt.v[k] = shfl16(X.v[k], 15); }
      Tt.t = shfl16(X.t, 15);
      delta_then(&carry, Tt);
    }
    WSTAMP(4);
    // 3. the interval's maps (a skipped interval: the identity map and no noise)
    double F[kWMapF], G[kWMapG];
    if (act) w_interval_maps(q_st, z0, z1, b, sf, F, G);
    else {
      w_map_identity(F);
#pragma unroll
      for (int e = 0; e < kWMapG; ++e) G[e] = 0.0;
    }
    WSTAMP(5);
    // 4. Sigma_out = S_0 Sigma_in S_0^T + sum_k P_k G_k R G_k^T P_k^T  (S_0: all maps of the round, P_k: those after interval k)
    map_scan_level<1>(F); map_scan_level<2>(F); map_scan_level<4>(F); map_scan_level<8>(F);     // F: F_15 ... F_c
    double term[kWMapQ];
    {
      double P[kWMapF];
      w_map_identity(P);
#pragma unroll
      for (int e = 0; e < kWMapF; ++e) P[e] = row_shl<1>(P[e], F[e]);        // the maps after this lane's interval
      WSTAMP(6);
      w_noise_term(P, G, sg2, sa2, term);
    }
    if (base > 0) {                              // a later round: the first lane takes Sigma_in through the whole round
      double Sp[kWMapQ], Sin[kWMapQ];
#pragma unroll
      for (int e = 0; e < kWMapQ; ++e) Sp[e] = L[kWSig + e];
      w_conj(F, Sp, Sin);
#pragma unroll
      for (int e = 0; e < kWMapQ; ++e) term[e] += (c == 0) ? Sin[e] : 0.0;
    }
    WSTAMP(7);
#pragma unroll
    for (int e0 = 0; e0 < kWMapQ; e0 += 11) row_allsum_n<11>(term + e0);
    wave_lds_sync();                             // (a later round has read the previous Sigma above)
    if (c == 0) {
#pragma unroll
      for (int e = 0; e < kWMapQ; ++e) L[kWSig + e] = term[e];
    }
    wave_lds_sync();
  }
  WSTAMP(8);
  // the predicted state from the block's delta (vc_imu.hpp), then J = dLog_dSE3(T_pred T2^-1) dt1t2_dt1(T_pred, T2^-1) with
  // the velocity identity appended (9 x 10); lane i (< 9) forms row i of P = J Sigma J^T
  double a[9];
  {
    double Sp[kWMapQ];
#pragma unroll
    for (int e = 0; e < kWMapQ; ++e) Sp[e] = (n_max > 0) ? L[kWSig + e] : 0.0;
    double T1[7], T2[7], v1[3], gw[3], q_end[4], p_end[3], rp[3];
#pragma unroll
    for (int i = 0; i < 7; ++i) { T1[i] = L[kWPose + i]; T2[i] = L[kWPose + 8 + i]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) v1[i] = L[kWVel + i];
    { const double g2[2] = {L[kWImu + 0], L[kWImu + 1]}; imu_gravity(g2, gw); }
    quat_mul(T1, carry.q, q_end);
    tq_rotate(T1, carry.p, rp);
    const double ht2 = 0.5 * (carry.t * carry.t);
#pragma unroll
    for (int i = 0; i < 3; ++i) p_end[i] = ((T1[4 + i] + v1[i] * carry.t) - gw[i] * ht2) + rp[i];
    double rel[7], t2w[7], dl[42], m34[12], m44[16], Jr[6][7], mine[10];
    w_projection_prepare(q_end, p_end, T2, rel, t2w);
    w_dlog_dse3_lean(rel, dl);
    w_dqx_dq(q_end, t2w + 4, m34);
    w_dq1q2_dq1(t2w, m44);
#pragma unroll
    for (int i = 0; i < 6; ++i) w_projection_row(dl, m34, m44, i, Jr[i]);
#pragma unroll
    for (int q = 0; q < 10; ++q) {
      double x = (q >= 7 && c == q - 1) ? 1.0 : 0.0;            // rows 6..8: the velocity identity
#pragma unroll
      for (int i = 0; i < 6; ++i) x = (q < 7 && c == i) ? Jr[i][q] : x;
      mine[q] = x;
    }
    double JS[10];
#pragma unroll
    for (int q = 0; q < 10; ++q) {
      double acc = 0.0;
#pragma unroll
      for (int r = 0; r < 10; ++r) acc += mine[r] * Sp[w_qidx(r, q)];
      JS[q] = acc;
    }
#pragma unroll
    for (int jj = 0; jj < 6; ++jj) {
      double acc = 0.0;
#pragma unroll
      for (int q = 0; q < 7; ++q) acc += JS[q] * Jr[jj][q];
      a[jj] = acc;
    }
#pragma unroll
    for (int jj = 6; jj < 9; ++jj) a[jj] = JS[jj + 1];
  }
  WSTAMP(9);
  // Cholesky P = L L^T, lane i (< 9) = row i in registers: the pivot and the pivot column travel by 16-lane shuffles
  double idg[9];
  bool ok = true;
#pragma unroll
  for (int cc = 0; cc < 9; ++cc) {
    const double d = shfl16(a[cc], cc);
    ok = ok && (d > 0.0);
    const double id = (d > 0.0) ? fast_rsqrt(d) : 0.0;
    idg[cc] = id;
    const double li = a[cc] * id;                // L[i][cc] for rows i >= cc (rows above: unused)
#pragma unroll
    for (int k = cc + 1; k < 9; ++k) a[k] -= li * shfl16(li, k);
    a[cc] = li;
  }
  if (c < 9) {
#pragma unroll
    for (int k = 0; k < 9; ++k) L[c * 9 + k] = a[k];             // the factor's row c (entries k <= c)
  }
  wave_lds_sync();
  WSTAMP(10);
  if (live && !ok && c == 0) atomicAdd((unsigned long long*)&v.dbg[20], 1ull);     // singular projection: keeps the previous weight (counted)
  if (exists && !(live && ok)) for (int e = c; e < 81; e += 16) v.wsqrtb[1 - wr][(size_t)s * 81 + e] = v.wsqrtb[wr][(size_t)s * 81 + e];
  if (live && ok && c < 9) {                     // column c of X = L^-1
    double x[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) x[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      double acc = (i == c) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < 9; ++k) if (k < i) acc -= L[i * 9 + k] * x[k];
      x[i] = (i >= c) ? acc * idg[i] : 0.0;
    }
    double* w = v.wsqrtb[1 - wr] + (size_t)s * 81;       // W[a][b] = X[b][a]
#pragma unroll
    for (int i = 0; i < 9; ++i) w[c * 9 + i] = x[i];
  }
  WSTAMP(11);
}

// ------------------------------------------------------------------------------------------ chain assembly
// One wavefront per frame, one workgroup per chunk (groups of 4 frames).  The frame's image is written ONCE, column by column
// (lane = image column: border [W | g], then the blocks [C | A | B]): every column gathers what belongs in it -- the camera's
// columns from the frame's tile of that camera, the IMU-parameter columns and the 9 x 9 blocks from the two IMU blocks the frame
// takes part in (compact records, vc_device.h: kSeg*), the right-hand side, the separator couplings of a sharded chain -- instead
// of zero-filling the image and scattering into it (round 2: 22 of 45 MB per launch at BASELINE cfg3 were the zero-fill and the
// unread parts of the 33 x 33 blocks).  The padding columns between D + 1 and ldw and behind the blocks are never read.
#ifdef VC_INIT_STAMPS
#define ISTAMP(i) do { if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0 && (i) < 16) v.dbg[i] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ISTAMP(i) do { } while (0)
#endif
constexpr int kInitPad = 160;            // per wavefront behind the Gram records: Hs 42 | A 81 | g 9 | lambda 9 | tile cameras 8 | pad
// Round 5: the loads of a frame are two dependent round trips instead of four (the frame's tile of every camera comes from
// frame_cam_tile[f][c] -- requested before the control record has arrived; then everything else in one go: the tiles' Gram records, the
// two IMU records, costs, damping state -- where round 4 walked frame_tile_off -> tile_cam -> IMU records -> Gram records), and a
// wavefront that owns several frames of its chunk (large problems: the chunk grows with the frame count so that the whole grid is
// resident at once) requests frame i + 1's data before it forms frame i's columns: a frame costs its arithmetic, not arithmetic + latency.
// Stamps (tools/init_stamps.py) before the change: ~10 us of the ~20 us a frame took at every size were the four round trips.
// CMAX: cameras the instance holds registers for (1, 2, 4, 8).
template <int CMAX>
struct InitLoads {
  double gv[CMAX][3];          // Gram record of the frame's tile of camera c as it lies in HBM (packed upper triangle + side vector: entries lane, lane + 64, lane + 128 of kGPack)
  double pre[9];               // lanes 0..14: couplings of IMU parameter `lane`; lanes 15..23: column lane - 15 of B
  double a_imu[2], g_imu, sc2_in, dg_in, isum_in[4], cost_in;
};
template <int CMAX>
__global__ __launch_bounds__(256, 2) void k_chain_init(DevView v) {     // (two workgroups per CU: the kernel waits on memory)
  // Chunk sums of the camera Gram blocks and of the IMU shared block: per-lane accumulators across the frame loop (instances up to four
  // cameras) or a pass of their own behind it (eight cameras, round 6: CMAX x 3 + 4 accumulators beside CMAX x 3 + 4 doubles of the frame in
  // flight kept 76 registers in scratch there -- 304 B per lane --, and a scratch reload waits for every load requested before it, the next
  // frame's among them: k_chain_init 107 -> 91 us at 6250 frames x 8 cameras; with four cameras the second read of the records costs more than
  // the 84 B of scratch did: 35 -> 37 us at 2500 frames)
  constexpr bool kLoopSums = CMAX <= 4;
  constexpr int CS = kLoopSums ? CMAX : 1;
  extern __shared__ __attribute__((aligned(16))) double sh[];
  __shared__ CamDesc s_cd[kMaxCams];
  __shared__ double s_R[kMaxCams * 9];   // the cameras' rotations R_ck, once per workgroup
  __shared__ int s_ci[256];              // what every image column is: owning camera (255: none) | column inside its block << 8 | "comes from the IMU records" << 16
  const Ctrl* ct = v.ctrl;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int C = v.n_cams, D = v.D, N = v.n_frames, ldw = v.ldw, ldx = v.ldx, nW = D + 1, ncol = nW + 27;
  const int chunk = blockIdx.x;
  const int f0 = chunk * v.chunk_frames, f1 = min(f0 + v.chunk_frames, N);
  // the first frame's tile list does not depend on the control record: requested ahead of it
  auto load_fct = [&](int f) { return (lane < C && f < f1) ? v.frame_cam_tile[(size_t)f * C + lane] : -1; };
  int fct = load_fct(f0 + wave);
#ifdef VC_INIT_STAMPS
  const long long ist0_ = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  if (ct->done) return;
#ifdef VC_INIT_STAMPS
  if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) v.dbg[0] = ist0_;      // (not in passes that exit early)
#endif
  const int cur = ct->cur;
  const int init_scale = ct->init_scale, reuse = ct->reuse_diag;
  const double radius = ct->radius;
  int sp_col = -1;                                 // lanes 0..14: the column of IMU parameter `lane` (or none)
#pragma unroll
  for (int a = 0; a < 15; ++a) sp_col = (lane == a) ? v.imu_param_col[a] : sp_col;
  // ---- everything a frame needs from memory, requested in one go (fct_f: lane c's tile of camera c in frame f, or -1)
  auto issue = [&](int f, int fct_f, InitLoads<CMAX>& R) {
    const bool live = f < f1;
    const bool pin_f = (f == 0 && v.pin_first), pin_l = (f == N - 1 && v.pin_last), pin_self = pin_f || pin_l;
    const bool pin_next = (f + 1 == N - 1 && v.pin_last);
#if defined(VC_INIT_EXP) && (VC_INIT_EXP & 1)
    const double* rc = nullptr; const double* rp = nullptr;      // experiment: no IMU-record loads
#else
    const double* rc = (live && f >= 1) ? v.segb[cur] + (size_t)(f - 1) * kSegStride : nullptr;
    const double* rp = (live && f + 1 < N) ? v.segb[cur] + (size_t)f * kSegStride : nullptr;
#endif
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
#if defined(VC_INIT_EXP) && (VC_INIT_EXP & 4)
      const int tc = -1;                                                          // experiment: no Gram-record loads
#else
      const int tc = __builtin_amdgcn_readfirstlane(__shfl(fct_f, c, 64));      // (wave-uniform: a scalar branch per camera)
#endif
#pragma unroll
      for (int q = 0; q < 3; ++q) R.gv[c][q] = 0.0;
      if (c < C && tc >= 0) {
        const double* g = v.Gb[cur] + (size_t)tc * kGPack;      // (three contiguous loads per tile; expanded to the 16 x 16 form on the way into LDS)
#pragma unroll
        for (int q = 0; q < 3; ++q) { const int e = q * 64 + lane; const double x = g[e < kGPack ? e : 0]; R.gv[c][q] = e < kGPack ? x : 0.0; }
      }
    }
    // Branch-free from here on: every lane loads from a valid address (a lane without an entry reads the start of record 0) and keeps
    // the value or a zero by a select.  Loads under divergent branches are not hoisted out of them: round 4's form -- `if (lane has this
    // entry) x += record[..]`, ~40 masked loads per frame -- ran them as a chain of dependent round trips, 8 of the 17 us a wavefront
    // took at BASELINE cfg3 and 60 of 124 us at cfg5's per-rank size (tools/init_stamps.py with -DVC_INIT_EXP=1: the kernel without them).
    const double* safe = v.cg;                 // (always there; only its first entry is ever read through `safe`, and the value is dropped)
    const double* rcs = rc ? rc : safe;
    const double* rps = rp ? rp : safe;
    const bool has_c = rc != nullptr, has_p = rp != nullptr;      // wave-uniform
    const int oc = has_c ? 1 : 0, op = has_p ? 1 : 0;             // offsets into an absent record collapse to 0
    {
      const bool tile_cost = lane < C && fct_f >= 0, blk_cost = lane == 8 && has_c;      // every block counted once, by its "cur" frame
      const double* pcst = tile_cost ? v.tile_costb[cur] + fct_f : (blk_cost ? v.seg_costb[cur] + (f - 1) : safe);      // (has_c: f >= 1 and a block f - 1 exists)
      const double x = *pcst;
      R.cost_in = (tile_cost || blk_cost) ? x : 0.0;
    }
    {
      // lanes 0..14: IMU parameter `lane`'s couplings with this frame (from both records); lanes 15..23: column lane - 15 of B
      const bool isB = lane >= 15 && lane < 24 && has_p && !pin_self && !pin_next;
      const bool par = lane < 15 && sp_col >= 0;
      const int a = lane < 15 ? lane : 0, jb = lane >= 15 && lane < 24 ? lane - 15 : 0;
      const double* pa = par ? rcs + (kSegWc + a) * oc : (isB ? rps + kSegBpc + jb : safe);
      const int sa = par ? 15 * oc : (isB ? 9 : 0);
      const bool va = (par && has_c) || isB;
      const double* pb = par ? rps + (kSegWp + a) * op : safe;
      const int sb = par ? 15 * op : 0;
      const bool vb = par && has_p;
      double xa[9], xb[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) { xa[i] = pa[i * sa]; xb[i] = pb[i * sb]; }
#pragma unroll
      for (int i = 0; i < 9; ++i) R.pre[i] = (va ? xa[i] : 0.0) + (vb ? xb[i] : 0.0);
    }
    {
      double xc[2], xp[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) { const int e = lane + 64 * q, ee = e < 81 ? e : 0; xc[q] = rcs[(kSegAcc + ee) * oc]; xp[q] = rps[(kSegApp + ee) * op]; }
#pragma unroll
      for (int q = 0; q < 2; ++q) { const bool in = lane + 64 * q < 81; R.a_imu[q] = ((in && has_c) ? xc[q] : 0.0) + ((in && has_p) ? xp[q] : 0.0); }
    }
    {
      const int l9 = lane < 9 ? lane : 0;
      const size_t fo = (size_t)(live ? f : 0) * 9 + l9;
      const double gc = rcs[(kSegGc + l9) * oc], gp = rps[(kSegGp + l9) * op], s2 = v.cscale2[fo], dgv = v.cdiag[fo];
      const bool in = lane < 9 && live;
      R.g_imu = ((in && has_c) ? gc : 0.0) + ((in && has_p) ? gp : 0.0);
      R.sc2_in = (in && !init_scale) ? s2 : 0.0;
      R.dg_in = (in && reuse) ? dgv : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {      // IMU shared block of block f-1 (every block counted once, by its "cur" frame)
      const int e = lane + 64 * q, a = e >> 4, b2 = e & 15;
      const bool in = a < 15;
      if (kLoopSums) {
        const double x = rcs[(in ? ((b2 < 15) ? kSegHii + a * 15 + b2 : kSegGi + a) : 0) * oc];
        R.isum_in[q] = (in && has_c) ? x : 0.0;
      } else R.isum_in[q] = 0.0;
    }
  };
  InitLoads<CMAX> R;
  issue(f0 + wave, fct, R);
  int fct_next = load_fct(f0 + wave + 4);          // the tile list of the frame after this one (one load, a frame ahead of its use)
  if (tid < kMaxCams) s_cd[tid] = v.cd[tid];
  if (tid < C) { double Rm[9]; quat_to_R(v.cams[cur] + (size_t)tid * kCamStride, Rm); for (int k = 0; k < 9; ++k) s_R[tid * 9 + k] = Rm[k]; }
  double* Gw = sh + wave * (C * kGStride + kInitPad);
  double* Hs = Gw + C * kGStride;
  double* As = Hs + 42;                  // the frame's own 9 x 9 block (undamped)
  double* gs = As + 81;                  // its right-hand side
  double* ls = gs + 9;                   // its damping
  int* tcam = reinterpret_cast<int*>(ls + 9);      // cameras of the frame's tiles [8], then the frame's tile of every camera [8]
  int* tinv = tcam + kMaxCams;
  double gsum[CS][3];           // per-camera sums of the chunk's Gram records, packed like the records (kLoopSums)
  // where the lane's packed entries go in the 16 x 16 + 16 record: offset of (r, c), of (c, r), -1: no such entry
  int po[3], pm[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int e = q * 64 + lane;
    int r = 0;
#pragma unroll
    for (int a = 1; a < 16; ++a) r += (e >= a * 16 - (a * (a - 1)) / 2) ? 1 : 0;      // row of packed entry e (< kGPackGrad)
    const int cidx = r + (e - (r * 16 - (r * (r - 1)) / 2));
    po[q] = e < kGPackGrad ? r * 16 + cidx : (e < kGPack ? kGGrad + (e - kGPackGrad) : -1);
    pm[q] = (e < kGPackGrad && cidx != r) ? cidx * 16 + r : -1;
  }
  double isum[4] = {0.0, 0.0, 0.0, 0.0};      // IMU shared block: Hii[a][b] at a*16+b (a,b<15), g_i[a] at a*16+15
  double csum = 0.0;            // cost at the linearisation point: the frames' tiles (lanes 0..7) and IMU blocks (lane 8), summed per chunk here
                                // instead of over all tiles and blocks by the one workgroup of k_reduced (32 of its 214 us at 50 000 tiles)
#pragma unroll
  for (int c = 0; c < CS; ++c)
#pragma unroll
    for (int q = 0; q < 3; ++q) gsum[c][q] = 0.0;
  // what each image column is, once for all frames of the chunk: owning camera and column inside its block; the columns that come
  // from the IMU records -- the (at most 15) IMU-parameter columns and the 9 columns of B -- belong to lanes 0..23 whatever their
  // position in the image, so that their values can be requested before anything else happens
  if (tid < ncol) {
    const int col = tid, e = col - nW;
    int cam = 255, loc = 0, skip = (e >= 18 && e < 27) ? 1 : 0;
    if (col < D) {
      const int cc = v.col_cam[col];
      cam = cc >= 0 ? cc : 255; loc = v.col_local[col];
#pragma unroll
      for (int a = 0; a < 15; ++a) skip |= (v.imu_param_col[a] == col) ? 1 : 0;
    }
    s_ci[col] = cam | (loc << 8) | (skip << 16);
  }
  __syncthreads();
  ISTAMP(1);

  for (int fg = f0; fg < f1; fg += 4) {
    const int f = fg + wave;
    if (f >= f1) continue;
    double* Wf = v.cW + (size_t)f * 9 * ldx;     // the frame's image: [W | g] in columns 0..D, then (from ldw) C (zero) | A | B
    // pinned frames (separator / ghost, see DevView): their rows go to sep_strip, the chain sees an isolated identity block
    const bool pin_f = (f == 0 && v.pin_first), pin_l = (f == N - 1 && v.pin_last), pin_self = pin_f || pin_l;
    const bool pin_prev = (f == 1 && v.pin_first), pin_next = (f + 1 == N - 1 && v.pin_last);
    double* Wt = pin_self ? v.sep_strip + (size_t)(pin_f ? 0 : 1) * 9 * ldw : Wf;
    const int ldt = pin_self ? ldw : ldx;         // row stride of Wt
    const int sep_self = pin_f ? v.sep_col0 : v.sep_col1;
    const double* rc = (f >= 1) ? v.segb[cur] + (size_t)(f - 1) * kSegStride : nullptr;      // (the separator couplings below read them again: L2 hits)
    const double* rp = (f + 1 < N) ? v.segb[cur] + (size_t)f * kSegStride : nullptr;
    // ---- this frame's values out of the load registers: Gram records to LDS and into the chunk's per-camera sums, the rest to locals
    const unsigned long long present = __ballot(lane < C && fct >= 0);
    const int nt = __popcll(present);
    if (lane < kMaxCams) {
      const bool have = lane < C && fct >= 0;
      const int slot = __popcll(present & ((1ull << lane) - 1ull));
      tinv[lane] = have ? slot : -1;                       // the frame's tile slot of every camera
    }
    wave_lds_sync_local();      // (wavefront scope: a workgroup-scope fence would wait for the next frame's loads)
    if (lane < C && fct >= 0) tcam[tinv[lane]] = lane;     // camera of every slot
    csum += R.cost_in;
    if (kLoopSums) {
#pragma unroll
      for (int q = 0; q < 4; ++q) isum[q] += R.isum_in[q];
    }
    int slot_c = 0;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      const bool have = c < C && ((present >> c) & 1ull);      // wave-uniform
      if (have) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          if (po[q] >= 0) Gw[slot_c * kGStride + po[q]] = R.gv[c][q];
          if (pm[q] >= 0) Gw[slot_c * kGStride + pm[q]] = R.gv[c][q];
          if (kLoopSums) gsum[c < CS ? c : 0][q] += R.gv[c][q];
        }
        ++slot_c;
      }
    }
    // the image columns that come straight from the IMU records (lanes 0..23) are written now: their registers are free for the next
    // frame's loads
    if (sp_col >= 0) {
#pragma unroll
      for (int i = 0; i < 9; ++i) Wt[i * ldt + sp_col] = R.pre[i];
      if (pin_self) {
#pragma unroll
        for (int i = 0; i < 9; ++i) Wf[i * ldx + sp_col] = 0.0;
      }
    }
    if (lane >= 15 && lane < 24) {
#pragma unroll
      for (int i = 0; i < 9; ++i) Wf[i * ldx + ldw + 18 + (lane - 15)] = R.pre[i];
    }
    const double a_imu[2] = {R.a_imu[0], R.a_imu[1]};
    const double g_imu = R.g_imu, sc2_in = R.sc2_in, dg_in = R.dg_in;
    wave_lds_sync_local();
    ISTAMP(2);
    // ---- the next frame of this wavefront: its loads go out now, under this frame's arithmetic and stores
    fct = fct_next;
    if (f + 4 < f1) { issue(f + 4, fct, R); fct_next = load_fct(f + 8); }
    ISTAMP(3);
    // ---- visual part of the frame's own block: H_pp (6 x 6) and g_p (6) from the tiles (lanes 0..41)
    if (lane < 42) {
      double hval = 0.0;
      for (int t = 0; t < nt; ++t) {
        const int c = tcam[t];
        const double* Rm = s_R + c * 9;
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
          const int nk = model_nk(s_cd[c].model);
          double s = 0.0;
#pragma unroll
          for (int p = 0; p < 3; ++p) s += Rm[3 * p + ii] * gram_grad(g, 3 * a + p, nk);
          hval += (a == 0) ? -s : s;
        }
      }
      Hs[lane] = hval;
    }
    wave_lds_sync_local();
    ISTAMP(4);
    // ---- own block A (lane e = 9 i + j, two slots), right-hand side, damping
    double aval[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = lane + 64 * q, i = e / 9, j = e % 9;
      aval[q] = a_imu[q] + ((e < 81 && i < 6 && j < 6) ? Hs[i * 6 + j] : 0.0);
      if (e < 81) As[e] = aval[q];
    }
    const double gval = g_imu + ((lane < 6) ? Hs[36 + lane] : 0.0);
    double hd = 0.0;
    // diagonal entry i lives in lane (i*10) % 64, slot (i*10) / 64
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int e = i * 10;
      const double d = readlane_f64(aval[e >> 6], e & 63);
      if (lane == i) hd = d;
    }
    if (lane < 9) {
      // damping of the 9 frame parameters (Jacobi scaling fixed per Solve, diagonal re-used after rejections)
      double sc2 = sc2_in, dg = dg_in;
      if (init_scale) { sc2 = jacobi_scale2(hd); v.cscale2[(size_t)f * 9 + lane] = sc2; }
      if (!reuse) { dg = lm_clamped_diag(hd, sc2); v.cdiag[(size_t)f * 9 + lane] = dg; }
      const double lam = pin_self ? 0.0 : dg / (radius * sc2);
      v.clam[(size_t)f * 9 + lane] = lam;
      v.cg[(size_t)f * 9 + lane] = pin_self ? 0.0 : gval;
      gs[lane] = gval; ls[lane] = lam;
    }
    wave_lds_sync_local();
    ISTAMP(5);
    // ---- the image: every column written once (the columns that come from the IMU records went out above); one column per lane and round
    for (int col = lane; col < ncol; col += 64) {
      const int ci = s_ci[col];
      if (ci >> 16) continue;
      double val[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) val[i] = 0.0;
      const int e = col - nW, jb = (e >= 0) ? e % 9 : 0;
      const int cc = ci & 255, j = (ci >> 8) & 255;
      const int t = (cc < kMaxCams) ? tinv[cc] : -1;
      if (col < D && t >= 0) {                     // a camera's column: from the frame's tile of that camera
        const int flags = s_cd[cc].flags, nrot = (flags & kCamRotFree) ? 3 : 0, ntr = (flags & kCamTransFree) ? 3 : 0;
        const double* Rm = s_R + cc * 9;
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
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          val[i] = -(Rm[i] * u[0] + Rm[3 + i] * u[1] + Rm[6 + i] * u[2]);
          val[3 + i] = Rm[i] * u[3] + Rm[3 + i] * u[4] + Rm[6 + i] * u[5];
        }
      }
      // a separator's columns (sharded chain): the coupling of this frame with the pinned neighbour, from the shared IMU block
      if (col < D && pin_prev && rc && col >= v.sep_col0 && col < v.sep_col0 + 9) {       // rows: this frame (cur), cols: the pinned frame 0 (prev)
#pragma unroll
        for (int i = 0; i < 9; ++i) val[i] = rc[kSegBpc + (col - v.sep_col0) * 9 + i];
      }
      if (col < D && pin_next && rp && col >= v.sep_col1 && col < v.sep_col1 + 9) {       // rows: this frame (prev), cols: the pinned last frame (cur)
#pragma unroll
        for (int i = 0; i < 9; ++i) val[i] = rp[kSegBpc + i * 9 + (col - v.sep_col1)];
      }
      if (col < D && pin_self && col >= sep_self && col < sep_self + 9) {                 // a pinned frame's own block sits in its separator columns
#pragma unroll
        for (int i = 0; i < 9; ++i) val[i] += As[i * 9 + (col - sep_self)];
      }
      if (col == D) {                              // the right-hand side rides as column D
#pragma unroll
        for (int i = 0; i < 9; ++i) val[i] = gs[i];
      }
      if (e >= 9 && e < 18) {                      // A (damped; a pinned frame: the identity)
#pragma unroll
        for (int i = 0; i < 9; ++i) val[i] = pin_self ? ((i == jb) ? 1.0 : 0.0) : As[i * 9 + jb] + ((i == jb) ? ls[i] : 0.0);
      }
#if defined(VC_INIT_EXP) && (VC_INIT_EXP & 2)
      if (val[0] == 123.456)                       // experiment: no image stores
#endif
      if (col < nW) {
#pragma unroll
        for (int i = 0; i < 9; ++i) Wt[i * ldt + col] = val[i];
        if (pin_self) {                            // (wave-uniform) the chain's image of a pinned frame has an empty border
#pragma unroll
          for (int i = 0; i < 9; ++i) Wf[i * ldx + col] = 0.0;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) Wf[i * ldx + ldw + e] = val[i];
      }
    }
    wave_lds_sync_local();
    ISTAMP(6);
  }
  __syncthreads();
  ISTAMP(7);
  double* part = v.part + (size_t)chunk * v.part_stride;
  if constexpr (kLoopSums) {
    // (4 wavefronts combined in fixed order)
    const int slot = C * kGStride + kGStride;
#pragma unroll
    for (int c = 0; c < CS; ++c)
      if (c < C) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          if (po[q] >= 0) sh[wave * slot + c * kGStride + po[q]] = gsum[c][q];
          if (pm[q] >= 0) sh[wave * slot + c * kGStride + pm[q]] = gsum[c][q];
        }
      }
#pragma unroll
    for (int q = 0; q < 4; ++q) sh[wave * slot + C * kGStride + q * 64 + lane] = isum[q];
    __syncthreads();
    for (int e = tid; e < slot; e += 256)
      part[D * D + D + e] = (sh[e] + sh[slot + e]) + (sh[2 * slot + e] + sh[3 * slot + e]);
  } else {
    // second reads of the chunk's records (out of the L2 / the memory-side cache), frame by frame in fixed order
    int* tl = reinterpret_cast<int*>(sh);                 // the chunk's tile table [frame][camera] (the frame loop is done with the LDS)
    const int nf = f1 - f0;
    for (int e = tid; e < nf * C; e += 256) tl[e] = v.frame_cam_tile[(size_t)f0 * C + e];
    __syncthreads();
    const double* Gc = v.Gb[cur];
    for (int e = tid; e < C * kGPack; e += 256) {
      const int c = e / kGPack, pe = e - c * kGPack;
      double acc = 0.0;
      for (int fr = 0; fr < nf; fr += 8) {               // eight frames' loads in flight, added in frame order
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int t = fr + u < nf ? tl[(fr + u) * C + c] : -1; const double y = Gc[t >= 0 ? (size_t)t * kGPack + pe : 0]; x[u] = t >= 0 ? y : 0.0; }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += x[u];
      }
      // where the packed entry goes in the 16 x 16 + 16 record: (r, c) and (c, r), or the side vector
      int r = 0;
#pragma unroll
      for (int a2 = 1; a2 < 16; ++a2) r += (pe >= a2 * 16 - (a2 * (a2 - 1)) / 2) ? 1 : 0;
      const int cidx = r + (pe - (r * 16 - (r * (r - 1)) / 2));
      double* pc = part + D * D + D + c * kGStride;
      if (pe < kGPackGrad) { pc[r * 16 + cidx] = acc; if (cidx != r) pc[cidx * 16 + r] = acc; }
      else pc[kGGrad + (pe - kGPackGrad)] = acc;
    }
    {   // IMU shared block of the blocks f - 1, f in the chunk: Hii[a][b] at a*16+b (a, b < 15), g_i[a] at a*16+15
      const int e = tid, a2 = e >> 4, b2 = e & 15;
      const bool in = a2 < 15;
      const int off = in ? ((b2 < 15) ? kSegHii + a2 * 15 + b2 : kSegGi + a2) : 0;
      double acc = 0.0;
      for (int fr = 0; fr < nf; fr += 8) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int f = f0 + fr + u; const bool has = in && fr + u < nf && f >= 1; const double y = v.segb[cur][has ? (size_t)(f - 1) * kSegStride + off : 0]; x[u] = has ? y : 0.0; }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += x[u];
      }
      part[D * D + D + C * kGStride + e] = acc;
    }
  }
  // the chunk's cost: nine lanes per wavefront hold terms; fixed order (lane, then wavefront)
  __syncthreads();
  if (lane < 9) sh[wave * 9 + lane] = csum;
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int w = 0; w < 4; ++w) { double tw = 0.0; for (int k = 0; k < 9; ++k) tw += sh[w * 9 + k]; t += tw; }
    part[v.part_stride - 1] = t;
  }
#ifdef VC_INIT_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
  ISTAMP(8);
#endif
}

// ------------------------------------------------------------------------------------------ chain elimination
// The frame unknowns (pose 6 + velocity 3) form a block-tridiagonal chain with a dense border (the shared parameters and
// the right-hand side).  Every frame owns one IMAGE of 9 rows (cW, row stride ldx): columns 0..D the border [W | g], then
// from column ldw the three 9 x 9 blocks [C | A | B] -- C the coupling to the left separator (fill-in; zero to start
// with), A the frame's own block, B the coupling to the next active frame (rows = this frame).
// The chain is factored by nested partitioning: at level l the active frames are those with index 0 (mod s), s = m^l;
// they are cut into groups of m -- the first frame of a group is its LEFT SEPARATOR and stays active, the other m - 1 (the
// interior) are eliminated by ONE wavefront, sequentially, left to right, with lane = image column throughout:
//   * factor A = L L^T (nine lanes, lane = row, pivots through v_readlane; the factor goes to LDS);
//   * every lane solves L x = (its column); the solved C and B columns [X_s | X_n] (18 columns) go to LDS, the solved
//     image replaces the frame's image in HBM (the A columns receive L itself): that is all the back-substitution needs;
//   * every lane forms out = [X_s | X_n]^T x: rows 9..17 update the lane's column of the NEXT frame's image (which the
//     lane keeps in registers for the next elimination; A's columns go to LDS for the next factorisation), rows 0..8
//     accumulate the group's update of the left separator.
// The last interior's "next" is the right separator, which belongs to the neighbouring group: its update goes to a side
// image (rX, double-buffered by level parity) and is folded in when that frame is loaded or written at the next level.
// A few levels (N = 2000, m = 8: 2000 -> 250 -> 32 -> 4) and a top level that eliminates what is left replace the
// 2 log2(N) launches of cyclic reduction by log_m(N) + 1, with (m - 1) log_m(N) dependent eliminations.
constexpr int kChainM = 8;          // group size: 7 eliminations per wavefront and level
constexpr int kXsLd = 20;           // row stride of the [X_s | X_n] LDS image
// phase stamps of the first group of level 0 (profiling builds only, -DVC_CHAIN_STAMPS): 100 MHz s_memrealtime ticks in dbg[0..31]
#ifdef VC_CHAIN_STAMPS
#define CSTAMP(i) do { if (group == 0 && lvl == 0 && threadIdx.x == 0 && (i) < 32) v.dbg[i] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CSTAMP(i) do { } while (0)
#endif
// NW > 1: the group's columns are spread over NW wavefronts of one workgroup instead of several columns per lane (column
// c = lane + 64 (wave + NW ci)).  Every wavefront factors A itself (its own copy of L in LDS: nine lanes, a microsecond), the
// solved [X_s | X_n] columns and the next frame's A are exchanged through the shared LDS images behind workgroup barriers.  For
// wide borders (D > 36) this replaces columns per lane, which cost 25 us per elimination at three columns (256 VGPR + AGPR
// copies) against a few us here; the columns-per-lane instances stay for A/B runs (launcher below).
// One group of one level: `group` its index, `wave` the wavefront's index inside the group (0 .. NW - 1), the LDS images
// XS (9 x kXsLd), An (81), Ls_all (NW x 81) the group's own.  NW > 1: all wavefronts of the workgroup belong to the group.
template <int CPL, int NW>          // columns per lane x wavefronts: D + 1 + 27 <= 64 CPL NW
__device__ __forceinline__ void chain_fwd_group(const DevView& v, int s, int m, int top, int lvl, int group, int wave, double* XS, double* An, double* Ls_all) {
  auto group_sync = [&]() { if (NW > 1) __syncthreads(); else wave_lds_sync(); };
  CSTAMP(0);
  // the control record is requested here and looked at after the first frame's image has been requested as well: a finished
  // solve costs a few wasted loads, a running one saves the record's round trip at the head of every level
  const int done = v.ctrl->done;
  if (lvl == 1 && blockIdx.x == 0 && threadIdx.x == 0) signal_started(v, 7);      // the bottom level is complete: this launch runs (the weight update waits for it)
  const int lane = threadIdx.x & 63;
  const int N = v.n_frames, D = v.D, ldw = v.ldw, ldx = v.ldx, nW = D + 1, ncol = nW + 27;
  const long gs = (long)m * s;
  const int a = top ? -1 : (int)(group * gs);
  const int first = top ? 0 : a + s;
  const bool pend = lvl > 0;                                   // right contributions of level lvl - 1 are waiting
  const double* rp = v.rX[(lvl + 1) & 1];
  double* rw = v.rX[lvl & 1];
  const size_t isz = (size_t)9 * ldx;
  // role of each of the lane's columns: 0 border (W | g), 1 C, 2 A, 3 B, 4 none
  int role[CPL], pc[CPL], sub[CPL];
#pragma unroll
  for (int ci = 0; ci < CPL; ++ci) {
    const int c = lane + 64 * (wave + NW * ci), e = c - nW;
    role[ci] = c < nW ? 0 : (e < 9 ? 1 : e < 18 ? 2 : e < 27 ? 3 : 4);
    pc[ci] = c < nW ? c : (c < ncol ? ldw + e : 0);
    sub[ci] = e < 9 ? e : e < 18 ? e - 9 : e - 18;            // column inside the 9 x 9 block
  }
  if (first >= N) {            // a separator without interior frames: only its pending right contribution is folded in
    if (!done && !top && a < N) {
#pragma unroll
      for (int ci = 0; ci < CPL; ++ci) {
        double* img = v.cW + (size_t)a * isz + pc[ci];
        if ((role[ci] == 0 || role[ci] == 2) && pend && a > 0) {
#pragma unroll
          for (int k = 0; k < 9; ++k) img[k * ldx] += rp[(size_t)(a / s) * isz + k * ldx + pc[ci]];
        } else if (role[ci] == 3) {
#pragma unroll
          for (int k = 0; k < 9; ++k) img[k * ldx] = 0.0;
        }
      }
    }
    return;
  }
  const int q = top ? (N - 1) / s + 1 : min(m - 1, (N - 1 - first) / s + 1);      // frames this wavefront eliminates
  double xin[CPL][9], dacc[CPL][9];
#pragma unroll
  for (int ci = 0; ci < CPL; ++ci)
#pragma unroll
    for (int k = 0; k < 9; ++k) dacc[ci][k] = 0.0;
  // ---- the first frame: its image columns to the lanes (C = B_a^T, a row of the separator's B block), A to LDS; the
  // second frame's columns are requested in the same breath (o / op: the next frame's image and its pending update)
  // four columns per lane (D > 164) do not leave registers for the pending values as well: they are read where they are used
  constexpr bool kLateOp = CPL >= 4;
  double o[CPL][9], op[kLateOp ? 1 : CPL][9];
  auto request_next = [&](int e, int i) {
    const int n = e + s;
    const bool load_n = (n < N) && !(!top && (i == m - 2));        // wave-uniform
    if (!load_n) return;                                            // (o / op are not read then)
    const double* img = v.cW + (size_t)n * isz;
    // unconditional loads: idle lanes (role 4) read column 0, their values are never used
#pragma unroll
    for (int ci = 0; ci < CPL; ++ci)
#pragma unroll
      for (int k = 0; k < 9; ++k) o[ci][k] = img[k * ldx + pc[ci]];
    if (pend && !kLateOp) {
      const double* rpn = rp + (size_t)(n / s) * isz;
#pragma unroll
      for (int ci = 0; ci < CPL; ++ci)
#pragma unroll
        for (int k = 0; k < 9; ++k) op[ci][k] = rpn[k * ldx + pc[ci]];
    }
  };
#pragma unroll
  for (int ci = 0; ci < CPL; ++ci)
#pragma unroll
    for (int k = 0; k < 9; ++k) { o[ci][k] = 0.0; if (!kLateOp) op[ci][k] = 0.0; }
  {
    const bool pe = pend && first > 0;
    const double* img = v.cW + (size_t)first * isz;
    const double* rpe = rp + (size_t)(first / s) * isz;
    double x0[CPL][9], xp[CPL][9];
#pragma unroll
    for (int ci = 0; ci < CPL; ++ci) {
      if (role[ci] == 1) {
#pragma unroll
        for (int k = 0; k < 9; ++k) { x0[ci][k] = (a >= 0) ? v.cW[(size_t)a * isz + sub[ci] * ldx + ldw + 18 + k] : 0.0; xp[ci][k] = 0.0; }
      } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) { x0[ci][k] = (role[ci] < 4) ? img[k * ldx + pc[ci]] : 0.0; xp[ci][k] = (pe && role[ci] < 4) ? rpe[k * ldx + pc[ci]] : 0.0; }
      }
    }
    request_next(first, 0);
    if (done) return;
    CSTAMP(1);
#pragma unroll
    for (int ci = 0; ci < CPL; ++ci) {
#pragma unroll
      for (int k = 0; k < 9; ++k) xin[ci][k] = x0[ci][k] + xp[ci][k];
      if (role[ci] == 2) {
#pragma unroll
        for (int k = 0; k < 9; ++k) An[k * 9 + sub[ci]] = xin[ci][k];
      }
    }
  }
  for (int i = 0; i < q; ++i) {
    const int e = first + i * s, n = e + s;
    const bool has_n = n < N;
    const bool n_sep = !top && (i == m - 2);                   // the next frame is the right separator (another group's)
    if (i > 0) request_next(e, i);       // the next frame's columns: requested now, used after the factorisation and the solve
    group_sync();                        // the frame's A block (LDS) is complete
    CSTAMP(2 + 4 * i);
    // ---- A = L L^T in every lane, from registers (see k_chain_fwd2: before, nine lanes factored through v_readlane and the factor
    // went through LDS); the triangular solves take L from registers
    double Lr[45], dinv[9];
    {
#pragma unroll
      for (int ii = 0; ii < 9; ++ii)
#pragma unroll
        for (int j = 0; j <= ii; ++j) Lr[ii * (ii + 1) / 2 + j] = An[ii * 9 + j];
      bool bad = false;
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        double d = Lr[j * (j + 1) / 2 + j];
        const bool ok = d > 0.0;
        bad |= !ok;
        d = ok ? d : 1.0;
        const double ip = fast_rsqrt(d);
        dinv[j] = ip;
        Lr[j * (j + 1) / 2 + j] = d * ip;
#pragma unroll
        for (int ii = j + 1; ii < 9; ++ii) Lr[ii * (ii + 1) / 2 + j] *= ip;
#pragma unroll
        for (int ii = j + 1; ii < 9; ++ii)
#pragma unroll
          for (int k = j + 1; k <= ii; ++k) Lr[ii * (ii + 1) / 2 + k] -= Lr[ii * (ii + 1) / 2 + j] * Lr[k * (k + 1) / 2 + j];
      }
      if (bad) {               // uniform (every lane holds the same numbers): the frame gets an identity block, the pass is flagged
        if (lane == 0 && wave == 0) atomicAdd(&v.flags[4 + 2 * v.par], 1);
#pragma unroll
        for (int ii = 0; ii < 9; ++ii) {
          dinv[ii] = 1.0;
#pragma unroll
          for (int j = 0; j <= ii; ++j) Lr[ii * (ii + 1) / 2 + j] = (ii == j) ? 1.0 : 0.0;
        }
      }
    }
    CSTAMP(3 + 4 * i);
    // ---- forward solves, lane = column; the image of e becomes [Y | z | X_s | L | X_n]
    double x[CPL][9];
#pragma unroll
    for (int ci = 0; ci < CPL; ++ci) {
      // column-oriented: x_k is final after k updates, the updates of one column are independent of each other
#pragma unroll
      for (int r = 0; r < 9; ++r) x[ci][r] = xin[ci][r];
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        x[ci][k] *= dinv[k];
#pragma unroll
        for (int r = k + 1; r < 9; ++r) x[ci][r] -= Lr[r * (r + 1) / 2 + k] * x[ci][k];
      }
      // an A lane has solved L x = A e_sub: x = row `sub` of L; the image wants column `sub` -- the nine lanes (of whichever
      // wavefronts) transpose through the group's LDS image behind the barrier of the [X_s | X_n] exchange
      if (role[ci] == 2) {
#pragma unroll
        for (int k = 0; k < 9; ++k) Ls_all[k * 9 + sub[ci]] = (k <= sub[ci]) ? x[ci][k] : 0.0;
      }
      if (role[ci] == 0 || role[ci] == 1 || role[ci] == 3) {
        double* img = v.cW + (size_t)e * isz + pc[ci];
#pragma unroll
        for (int k = 0; k < 9; ++k) img[k * ldx] = x[ci][k];
      }
      if (role[ci] == 1 || role[ci] == 3) {
        const int xc = sub[ci] + (role[ci] == 3 ? 9 : 0);
#pragma unroll
        for (int k = 0; k < 9; ++k) XS[k * kXsLd + xc] = x[ci][k];
      }
    }
    group_sync();                        // [X_s | X_n] is complete
    CSTAMP(4 + 4 * i);
    // ---- Schur updates: out[r] = sum_k [X_s | X_n][k][r] * (column)[k]; the A columns are driven by X_n's columns
#pragma unroll
    for (int ci = 0; ci < CPL; ++ci) {
      if (role[ci] == 2) {
        double* img = v.cW + (size_t)e * isz + pc[ci];
#pragma unroll
        for (int k = 0; k < 9; ++k) img[k * ldx] = Ls_all[sub[ci] * 9 + k];      // L[k][sub] (zero above the diagonal)
#pragma unroll
        for (int k = 0; k < 9; ++k) x[ci][k] = XS[k * kXsLd + 9 + sub[ci]];
      }
      double out[18];
#pragma unroll
      for (int r = 0; r < 18; ++r) out[r] = 0.0;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const double xk = x[ci][k];
#pragma unroll
        for (int r = 0; r < 18; ++r) out[r] += XS[k * kXsLd + r] * xk;
      }
#pragma unroll
      for (int r = 0; r < 9; ++r) dacc[ci][r] += out[r];
      if (has_n) {
        if (n_sep) {
          if (role[ci] < 4) {
            double* ri = rw + (size_t)(n / gs) * isz + pc[ci];
            const bool keep = role[ci] == 0 || role[ci] == 2;
#pragma unroll
            for (int k = 0; k < 9; ++k) ri[k * ldx] = keep ? -out[9 + k] : 0.0;
          }
          if (role[ci] == 3 && a >= 0) {                      // the separator's coupling to the right separator at the next level
            double* img = v.cW + (size_t)a * isz + pc[ci];
#pragma unroll
            for (int k = 0; k < 9; ++k) img[k * ldx] = -out[k];
          }
        } else {
          if (kLateOp) {
            const double* rpn = rp + (size_t)(n / s) * isz + pc[ci];
#pragma unroll
            for (int k = 0; k < 9; ++k) xin[ci][k] = (o[ci][k] + (pend ? rpn[k * ldx] : 0.0)) - (role[ci] == 3 ? 0.0 : out[9 + k]);
          } else {
#pragma unroll
            for (int k = 0; k < 9; ++k) xin[ci][k] = (o[ci][k] + op[ci][k]) - (role[ci] == 3 ? 0.0 : out[9 + k]);
          }
          if (role[ci] == 2) {
#pragma unroll
            for (int k = 0; k < 9; ++k) An[k * 9 + sub[ci]] = xin[ci][k];
          }
        }
      }
    }
    CSTAMP(5 + 4 * i);
  }
  CSTAMP(30);
  // ---- the left separator absorbs its group (and the right contribution it received one level down)
  if (a >= 0) {
    const bool chain_ends = !(q == m - 1 && first + q * s < N);
    const bool pa = pend && a > 0;
    const double* rpa = rp + (size_t)(a / s) * isz;
#pragma unroll
    for (int ci = 0; ci < CPL; ++ci) {
      if (role[ci] == 0 || role[ci] == 1) {                    // the C lanes carry the update of A's columns
        const int pcd = role[ci] == 1 ? ldw + 9 + sub[ci] : pc[ci];
        double* img = v.cW + (size_t)a * isz + pcd;
#pragma unroll
        for (int k = 0; k < 9; ++k) img[k * ldx] += (pa ? rpa[k * ldx + pcd] : 0.0) - dacc[ci][k];
      } else if (role[ci] == 3 && chain_ends) {
        double* img = v.cW + (size_t)a * isz + pc[ci];
#pragma unroll
        for (int k = 0; k < 9; ++k) img[k * ldx] = 0.0;
      }
    }
  }
}

template <int CPL, int NW>
__global__ __launch_bounds__(64 * NW) void k_chain_fwd(DevView v, int s, int m, int top, int lvl) {
  __shared__ __attribute__((aligned(16))) double XS[9 * kXsLd];
  __shared__ double An[81];
  __shared__ double Ls_all[NW * 81];
  chain_fwd_group<CPL, NW>(v, s, m, top, lvl, (int)blockIdx.x, (int)(threadIdx.x >> 6), XS, An, Ls_all);
}

// ---- one elimination of a two-sided sweep (k_chain_fwd2, k_chain_l0) --------------------------------------------------------------------
// The frame's block A is in An (LDS, complete behind the first synchronisation); lane = image column, x = the lane's column of the
// frame's image (role: 0 border (W | g), 1 C, 2 A, 3 B, 4 none; sub = column inside the 9 x 9 block; pc = the column's position in an
// image row).  A = L L^T in EVERY lane, from registers (round 4; before: nine lanes, one row each, pivots and pivot columns through
// v_readlane -- ~240 cycles per pivot, 0.9 us per frame, and the factor went through LDS to the lanes that solve with it): all lanes
// read the lower triangle (broadcast LDS reads) and run the same scalar factorisation -- per pivot one v_rsq_f64 chain, then independent
// multiplies / FMAs; the triangular solve that follows takes L from registers.  The solved column goes to the frame's image
// [Y | z | X_s | L | X_n] in HBM, [X_s | X_n] to XS (LDS) for everybody, out = [X_s | X_n]^T (column).
// WG_SYNC: the sweep is several wavefronts side by side (k_chain_fwd2<NW > 1>): workgroup barriers instead of wavefront-local ones.
// a double at a BYTE offset from a wave-uniform base: scalar base + 32-bit lane offset (global_load / global_store with an SGPR pair as the
// address and one VGPR as the offset) instead of a 64-bit address per lane and access.  A lone wavefront pays ~4 cycles for EVERY instruction
// it issues, address arithmetic included: per frame of a two-sided sweep the columns' addresses were ~340 instructions, the frame's
// elimination itself ~850 (round 6, tools/isa_classes.py with markers).  Images stay below 4 GB (n_frames x 9 x ldx x 8: 0.9 GB at 50 000 frames).
__device__ __forceinline__ double ld_boff(const double* base, unsigned boff) { return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + boff); }
__device__ __forceinline__ void st_boff(double* base, unsigned boff, double x) { *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + boff) = x; }
struct ElimLds { double* XS; double* An; double* Ls; };
#ifndef VC_ELIM_STORE_BOFF
#define VC_ELIM_STORE_BOFF 0      // (1: the solved image stored through scalar base + byte offsets as well -- same-box A/B: the folded bottom level 28.0 -> 30.0 us, the upper levels unchanged)
#endif
template <bool WG_SYNC>
__device__ __forceinline__ void chain_eliminate(const DevView& v, double* img /* the frame's image */, int ldx, int role, int pc, int sub, bool flag_lane,
                                                const ElimLds& M, double* x, double* out, long long* stamp = nullptr /* profiling builds: phase stamps of this call */) {
  auto gsync = [&]() { if (WG_SYNC) __syncthreads(); else wave_lds_sync_local(); };
#ifdef VC_F2_STAMPS
#define ELSTAMP(i) do { if (stamp) stamp[i] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ELSTAMP(i) do { } while (0)
#endif
  double* XS = M.XS; double* An = M.An; double* Ls = M.Ls;
  ELSTAMP(0);
  gsync();
  ELSTAMP(1);
  double Lr[45], dinv[9];
  {
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) Lr[i * (i + 1) / 2 + j] = An[i * 9 + j];
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      double d = Lr[j * (j + 1) / 2 + j];
      const bool ok = d > 0.0;
      bad |= !ok;
      d = ok ? d : 1.0;
      const double ip = fast_rsqrt(d);
      dinv[j] = ip;
      Lr[j * (j + 1) / 2 + j] = d * ip;
#pragma unroll
      for (int i = j + 1; i < 9; ++i) Lr[i * (i + 1) / 2 + j] *= ip;
#pragma unroll
      for (int i = j + 1; i < 9; ++i)
#pragma unroll
        for (int k = j + 1; k <= i; ++k) Lr[i * (i + 1) / 2 + k] -= Lr[i * (i + 1) / 2 + j] * Lr[k * (k + 1) / 2 + j];
    }
    if (bad) {               // wave-uniform (every lane holds the same numbers): the frame gets an identity block, the pass is flagged
      if (flag_lane) atomicAdd(&v.flags[4 + 2 * v.par], 1);
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        dinv[i] = 1.0;
#pragma unroll
        for (int j = 0; j <= i; ++j) Lr[i * (i + 1) / 2 + j] = (i == j) ? 1.0 : 0.0;
      }
    }
  }
  ELSTAMP(2);
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    x[k] *= dinv[k];
#pragma unroll
    for (int rr = k + 1; rr < 9; ++rr) x[rr] -= Lr[rr * (rr + 1) / 2 + k] * x[k];
  }
  ELSTAMP(3);
  // Hand-over through LDS, WITHOUT a divergent branch per role (round 6: the five `if (role == ..)` regions of this function were ~40 exec-mask
  // and branch instructions per elimination of a wavefront that is alone on its SIMD and pays ~4 cycles for every instruction it issues):
  // every lane stores its nine values through ONE address pattern chosen by its role --
  //   A lanes: row `sub` of L (x = L^T e_sub) as column `sub` of Ls (the image wants the column: the nine lanes transpose here),
  //   C / B lanes: their solved column into [X_s | X_n],   everybody else: the padding column 18 of XS (never read).
  {
    double* dst = role == 2 ? Ls + sub : XS + (role == 1 ? sub : role == 3 ? 9 + sub : 18);
    const int st = role == 2 ? 9 : kXsLd;
#pragma unroll
    for (int k = 0; k < 9; ++k) dst[k * st] = (role == 2 && k > sub) ? 0.0 : x[k];
  }
  ELSTAMP(4);
  gsync();
  ELSTAMP(5);
  // the frame's solved image [Y | z | X_s | L | X_n]: one store region for all lanes that own a column; the A lanes store L's column and
  // carry on with X_n's column `sub` (the A columns of the next frame are driven by X_n's)
  {
    const double* rl = Ls + (role == 2 ? sub * 9 : 0);
    const double* rx = XS + (role == 2 ? 9 + sub : 18);
    double lv[9], xv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { lv[k] = rl[k]; xv[k] = rx[k * kXsLd]; }
    if (role < 4) {
#if VC_ELIM_STORE_BOFF
      const unsigned pcb = 8u * (unsigned)pc, ldxb = 8u * (unsigned)ldx;
#pragma unroll
      for (int k = 0; k < 9; ++k) st_boff(img, pcb + k * ldxb, role == 2 ? lv[k] : x[k]);       // (A: L[k][sub], zero above the diagonal)
#else
      double* ic = img + pc;
#pragma unroll
      for (int k = 0; k < 9; ++k) ic[k * ldx] = role == 2 ? lv[k] : x[k];       // (A: L[k][sub], zero above the diagonal)
#endif
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) x[k] = role == 2 ? xv[k] : x[k];
  }
  ELSTAMP(6);
  // out = [X_s | X_n]^T x in two halves: rows 9..17 (the next frame's update, which the critical path waits for) first
#pragma unroll
  for (int h = 1; h >= 0; --h) {
#pragma unroll
    for (int rr = 0; rr < 9; ++rr) out[9 * h + rr] = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const double xk = x[k];
#pragma unroll
      for (int rr = 0; rr < 9; ++rr) out[9 * h + rr] += XS[k * kXsLd + 9 * h + rr] * xk;
    }
    if (h == 1) __builtin_amdgcn_sched_barrier(0);
  }
  ELSTAMP(7);
}

// ---- two-sided elimination of a group (narrow borders: one column per lane) ------------------------------------------------
// A full group [a | e_1 .. e_{m-1} | r] is eliminated from both ends at once: wavefront 0 sweeps e_1, e_2, .. left to right
// against the left separator a (exactly the one-sided scheme), wavefront 1 sweeps e_{m-1}, e_{m-2}, .. right to left against the
// right separator r -- the same recurrence on the mirrored chain: its "coupling to the next frame" is the transposed B block of the
// frame to its left, its first frame's coupling to the separator is that frame's own B block, its separator updates go to r's side
// image -- and the middle frame, which receives the updates of both sweeps, is eliminated last by wavefront 0 with a on one side and
// r on the other.  (m - 2) / 2 + 1 dependent eliminations per level instead of m - 1: 4 instead of 7 at m = 8.  Needs m >= 4.
// Images of the right sweep's frames hold [Y | z | X_s (coupling to r) | L | X_n (coupling to the frame on the LEFT)]; the
// back-substitution (k_chain_back, two = 1) mirrors the order.  The short group at the chain's end works the same way without r.
#ifdef VC_F2_STAMPS
#define F2STAMP(i) do { if (NW == 1 && blockIdx.x == 0 && lvl == 1 && threadIdx.x == 0 && (i) < 16) v.dbg[(i)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)      // (sweep 0; dbg[16..23]: the phases of its second elimination, chain_eliminate)
#else
#define F2STAMP(i) do { } while (0)
#endif
// NW > 1 (round 4): borders wider than one column per lane -- every sweep is NW wavefronts side by side (column c = lane + 64 u, as in
// chain_fwd_group), the group a workgroup of 2 NW wavefronts; the exchanges inside a sweep go through workgroup barriers, which both
// sweeps then meet in lockstep (a sweep with one frame less runs an empty round).  ~370 registers: one wavefront per SIMD, so NW = 2 fills
// a CU -- used on levels whose groups the chip holds at once (chain_levels).
template <int NW>
#ifndef VC_FWD2_WAVES
#define VC_FWD2_WAVES 1
#endif
__global__ __launch_bounds__(128 * NW, VC_FWD2_WAVES) void k_chain_fwd2(DevView v, int s, int m, int lvl, int n_groups) {
  constexpr int W = 64 * NW;
  __shared__ __attribute__((aligned(16))) double XS2[2][9 * kXsLd];
  __shared__ double An2[2][81], Ls2[2][81];
  __shared__ double MID[9 * W];      // right sweep -> wavefront 0: its update of the middle frame (W, A columns) and the middle's coupling to r
  __shared__ double SEPR[9 * W];     // the right sweep's accumulated update of r
  // (round 6) workgroups behind the level's groups: a side job -- the chunk records' camera / IMU / cost entries summed into part_total while
  // the level leaves most of the chip idle (vc_shared_blocks.hpp; DevView::hadd_early)
  if ((int)blockIdx.x >= n_groups) { part_tail_sum_job(v, (int)blockIdx.x - n_groups, MID, 128 * NW); return; }
  // (the wavefront's index as a scalar: everything that depends on it -- sweep direction, frame count, addresses -- stays uniform)
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wave = wv / NW /* the sweep */, lane = threadIdx.x & 63, group = blockIdx.x;
  auto gsync = [&]() { if (NW > 1) __syncthreads(); else wave_lds_sync_local(); };
  const int N = v.n_frames, D = v.D, ldw = v.ldw, ldx = v.ldx, nW = D + 1, ncol = nW + 27;
#ifdef VC_F2_STAMPS
  const long long f2_t0 = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  const long gs = (long)m * s;
  const int a = (int)((long)group * gs), first = a + s;
  const int done = v.ctrl->done;
  if (lvl == 1 && blockIdx.x == 0 && threadIdx.x == 0) signal_started(v, 7);      // the bottom level is complete: this launch runs (the weight update waits for it)
  const bool pend = lvl > 0;
  const double* rp = v.rX[(lvl + 1) & 1];
  double* rw = v.rX[lvl & 1];
  const size_t isz = (size_t)9 * ldx;
  const int c = lane + 64 * (wv % NW), e0 = c - nW;
  const int role = c < nW ? 0 : (e0 < 9 ? 1 : e0 < 18 ? 2 : e0 < 27 ? 3 : 4);     // 0 border (W | g), 1 C, 2 A, 3 B, 4 none
  const int pc = c < nW ? c : (c < ncol ? ldw + e0 : 0);
  const int sub = e0 < 9 ? e0 : e0 < 18 ? e0 - 9 : e0 - 18;
  if (first >= N) {            // a separator without interior frames: only its pending right contribution is folded in
    if (!done && wave == 0 && a < N) {
      double* img = v.cW + (size_t)a * isz + pc;
      if ((role == 0 || role == 2) && pend && a > 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) img[k * ldx] += rp[(size_t)(a / s) * isz + k * ldx + pc];
      } else if (role == 3) {
#pragma unroll
        for (int k = 0; k < 9; ++k) img[k * ldx] = 0.0;
      }
    }
    return;
  }
  const int q = min(m - 1, (N - 1 - first) / s + 1);          // interior frames e_1 .. e_q; the last group of a level may be short
  const int r = first + q * s;                                // right separator, if the chain goes on
  const bool has_r = q == m - 1 && r < N;
  const int nl = (q - 1) / 2, mid = nl + 1, nr = q - 1 - nl;                     // interior indices 1 .. nl | mid | mid + 1 .. q
  const int dir = wave == 0 ? 1 : -1, cnt = wave == 0 ? nl : nr, i0 = wave == 0 ? 1 : q;
  double* XS = XS2[wave]; double* An = An2[wave]; double* Ls = Ls2[wave];
  // the image columns of frame e as this sweep needs them: own values + the pending right contribution of the level below for the
  // border and A; the coupling to the sweep's separator (first frame only: later frames get it as fill-in); the coupling to the
  // next frame of the sweep
  // (branch-free: every lane reads nine values from ONE address pattern -- base + k stride -- chosen by its role, idle lanes read
  // column 0 of the image and scale it by zero; the pending contribution likewise)
  // (round 6: ONE byte offset pattern per lane -- first column entry c0b, step stb -- relative to the image of the frame the lane reads
  //  from (fsel: the sweep's frame e, the left separator a, or the frame e - s whose B block row is this frame's coupling), wave-uniform
  //  bases: 18 offset additions and 18 loads per frame where the 64-bit addresses per lane and load were ~340 instructions)
  const unsigned ldxb = 8u * (unsigned)ldx, iszb = 9u * ldxb;
  unsigned c0b = 0, stb = ldxb; int fsel = 0;
  if (role == 0 || role == 2) c0b = 8u * (unsigned)pc;
  else if (role == 1) { if (dir > 0) { c0b = 8u * (unsigned)(sub * ldx + ldw + 18); stb = 8; fsel = 1; } else c0b = 8u * (unsigned)(ldw + 18 + sub); }
  else if (role == 3) { if (dir > 0) c0b = 8u * (unsigned)pc; else { c0b = 8u * (unsigned)(sub * ldx + ldw + 18); stb = 8; fsel = 2; } }
  const unsigned ab = (unsigned)a * iszb;
  const int qa = group * m;             // e / s of frame e = a + i s is qa + i (a = group m s)
  auto load_cols = [&](int idx /* frame e = a + idx s, idx >= 1 */, bool first_of_sweep, bool with_image, double* x) {
    const int e = a + idx * s;
    const unsigned eb = (unsigned)e * iszb, emb = (unsigned)(e - s) * iszb;
    const unsigned fb = fsel == 1 ? ab : (fsel == 2 ? emb : eb);
    const unsigned pb = (unsigned)(qa + idx) * iszb;      // (the pending image of frame e in rp: one base for the whole kernel, the frame in the offset)
    const bool live0 = with_image && (role == 0 || role == 2 || role == 3 || (role == 1 && first_of_sweep));
    const bool live1 = with_image && pend && (role == 0 || role == 2);
    double y0[9], y1[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) y0[k] = ld_boff(v.cW, fb + c0b + k * stb);
    if (pend) {
#pragma unroll
      for (int k = 0; k < 9; ++k) y1[k] = ld_boff(rp, pb + c0b + k * stb);
    } else {
#pragma unroll
      for (int k = 0; k < 9; ++k) y1[k] = 0.0;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) x[k] = (live0 ? y0[k] : 0.0) + (live1 ? y1[k] : 0.0);
  };
  double xin[9], o[9], dacc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) { dacc[k] = 0.0; o[k] = 0.0; }
  // what the left separator will absorb into at the end (its own columns + the right contribution it received one level down):
  // requested now -- nobody else writes them during this level -- so that the kernel ends with stores, not with a load round trip
  double sep0[9];
  {
    const bool pa = pend && a > 0;
    const int pcd = role == 1 ? ldw + 9 + sub : pc;
    const double* img = v.cW + (size_t)a * isz + pcd;
    const double* rpa = rp + (size_t)(a / s) * isz + pcd;
    const bool mine = wave == 0 && (role == 0 || role == 1);
#pragma unroll
    for (int k = 0; k < 9; ++k) sep0[k] = mine ? img[k * ldx] + (pa ? rpa[k * ldx] : 0.0) : 0.0;
  }
  // (a side without frames of its own: wavefront 0 starts with the middle frame, wavefront 1 has nothing to load)
  load_cols(cnt > 0 ? i0 : mid, true, cnt > 0 || wave == 0, xin);
  // (round 6 tried requesting a frame's columns TWO eliminations ahead -- they come from HBM, the level below wrote them from other XCDs --:
  //  21.9 / 21.1 us per level against 20.2 / 19.6: nine more doubles per lane through the accumulation registers cost more than the wait)
  if (done) return;
#ifdef VC_F2_STAMPS
  if (NW == 1 && blockIdx.x == 0 && lvl == 1 && threadIdx.x == 0) v.dbg[0] = f2_t0;
#endif
  F2STAMP(1);
  if (role == 2) {
#pragma unroll
    for (int k = 0; k < 9; ++k) An[k * 9 + sub] = xin[k];
  }
  // one elimination: A (in An) = L L^T, all columns solved, image stored, [X_s | X_n] to XS, out = [X_s | X_n]^T (column)
  const ElimLds elds = {XS, An, Ls};
#ifdef VC_F2_STAMPS
  int el_n_ = 0;
  auto eliminate = [&](int e, double* x, double* out) {
    long long* st = (NW == 1 && blockIdx.x == 0 && lvl == 1 && threadIdx.x == 0 && el_n_ == 1) ? v.dbg + 16 : nullptr;      // (the second elimination of sweep 0)
    ++el_n_;
    chain_eliminate<(NW > 1)>(v, v.cW + (size_t)e * isz, ldx, role, pc, sub, c == 0, elds, x, out, st);
  };
#else
  auto eliminate = [&](int e, double* x, double* out) { chain_eliminate<(NW > 1)>(v, v.cW + (size_t)e * isz, ldx, role, pc, sub, c == 0, elds, x, out); };
#endif
  // The sweep's frames, then -- wavefront 0 only, behind the barrier that hands it the right sweep's results -- the middle frame:
  // a on its left (C), r on its right (B: the right sweep's fill-in, or -- no right sweep -- its own B block, already loaded).
  // One loop, one copy of the elimination: wavefront 1 leaves at the barrier.
  for (int j = 0; ; ++j) {
    const bool at_mid = j == cnt;
    F2STAMP(2 + 2 * j);
    if (at_mid) {
      if (NW > 1) for (int jj = cnt; jj < max(nl, nr); ++jj) { gsync(); gsync(); }      // the other sweep's extra elimination: its barriers
      if (wave == 1) {
        if (cnt == 0) {
#pragma unroll
          for (int k = 0; k < 9; ++k) MID[k * W + c] = 0.0;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) SEPR[k * W + c] = dacc[k];
      }
      F2STAMP(12);
      __syncthreads();
      F2STAMP(13);
      if (wave == 1) return;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        if (role == 0 || role == 2) xin[k] += MID[k * W + c];
        else if (role == 3 && nr > 0) xin[k] = MID[k * W + nW + sub];      // (no right sweep: the middle's own B block stays)
      }
      if (role == 2) {
#pragma unroll
        for (int k = 0; k < 9; ++k) An[k * 9 + sub] = xin[k];
      }
    }
    const int e = a + (at_mid ? mid : i0 + dir * j) * s;
#ifdef VC_EXP_NO_O_LOADS      // (timing experiment only -- wrong numbers: what the level costs when the next frame's columns need no memory round trip)
    if (!at_mid) { for (int k = 0; k < 9; ++k) o[k] = xin[k] * 0.999; }
#else
    if (!at_mid)                        // the columns of the frame after this one: requested now, used after the elimination
      load_cols(j + 1 < cnt ? i0 + dir * (j + 1) : mid, false, j + 1 < cnt || wave == 0, o);
#endif
    double x[9], out[18];
#pragma unroll
    for (int k = 0; k < 9; ++k) x[k] = xin[k];
    eliminate(e, x, out);
    F2STAMP(3 + 2 * j);
#pragma unroll
    for (int k = 0; k < 9; ++k) dacc[k] += out[k];
    if (at_mid) {
      // the right separator's side image: the middle's update and the right sweep's (whose C lanes carry the update of A's columns)
      if (has_r && role < 4) {
        double* ri = rw + (size_t)(r / gs) * isz + pc;
        const bool keep = role == 0 || role == 2;
        const int src = role == 2 ? nW + sub : c;
#pragma unroll
        for (int k = 0; k < 9; ++k) ri[k * ldx] = keep ? -out[9 + k] - SEPR[k * W + src] : 0.0;
      }
      if (role == 3) {                 // the separator's coupling to the right separator at the next level (none where the chain ends)
        double* img = v.cW + (size_t)a * isz + pc;
#pragma unroll
        for (int k = 0; k < 9; ++k) img[k * ldx] = has_r ? -out[k] : 0.0;
      }
      break;
    }
    if (wave == 1 && j + 1 == cnt) {    // the next frame is the middle, which wavefront 0 eliminates: hand the update over
#pragma unroll
      for (int k = 0; k < 9; ++k) MID[k * W + c] = -out[9 + k];
    } else {
#pragma unroll
      for (int k = 0; k < 9; ++k) xin[k] = o[k] - (role == 3 ? 0.0 : out[9 + k]);
      if (role == 2) {
#pragma unroll
        for (int k = 0; k < 9; ++k) An[k * 9 + sub] = xin[k];
      }
    }
  }
  // ---- the left separator absorbs its group (and the right contribution it received one level down)
  if (role == 0 || role == 1) {                      // the C lanes carry the update of A's columns
    const int pcd = role == 1 ? ldw + 9 + sub : pc;
    double* img = v.cW + (size_t)a * isz + pcd;
#pragma unroll
    for (int k = 0; k < 9; ++k) img[k * ldx] = sep0[k] - dacc[k];
  }
  F2STAMP(14);
}

// ---- bottom level with the chain assembly folded in (round 5; verdict r3 / r4 item 1a) ---------------------------------------------------
// One workgroup of FOUR wavefronts per group of 8 frames [a | e_1 .. e_7 | (r)]: wavefronts 0 / 1 are k_chain_fwd2<1>'s two sweeps,
// wavefronts 2 / 3 are BUILDERS that do k_chain_init's work for the group's frames in the order the sweeps need them (builder 2: the left
// sweep's frames, then the middle; builder 3: the right sweep's frames, then the separator a) and hand every frame over through LDS: the
// border [W | g] and the block A of an unsolved frame never touch HBM (k_chain_init wrote the image, the bottom level read it back, a kernel
// boundary between them).  The 9 x 9 couplings B -- all an image needs besides -- are read by the sweeps straight from the IMU block records.
// A builder requests its next frame's records before it forms the current frame's columns, so the sweeps wait ~one frame's build at their
// start and the builders stay ahead from there.  The frame's solved image, the damping state and the chunk sums (chunk = group) go to HBM as
// before.  Narrow borders (one image column per lane), at most two cameras, single process (no pinned frames), groups of 8, two-sided.
constexpr int kL0Ld = 48;      // row stride of a frame's LDS image: border (D + 1 <= 37 columns), then the 9 columns of A
template <int CMAX>
__global__ __launch_bounds__(256) void k_chain_l0(DevView v) {
  constexpr int W = 64;
  __shared__ __attribute__((aligned(16))) double XS2[2][9 * kXsLd];
  __shared__ double An2[2][81], Ls2[2][81];
  __shared__ double MID[9 * W], SEPR[9 * W];
  __shared__ double IMG[8][9 * kL0Ld];
  __shared__ int READY[12];              // [slot]: the frame's image is in IMG; [8 + w], w = 0, 1, 3: wavefront w's chunk sums are in SUMW
  __shared__ double BW[4][CMAX * kGStride + kInitPad];
  __shared__ double SUMW[4][CMAX * kGStride + kGStride + 16];
  __shared__ CamDesc s_cd2[4][kMaxCams];      // every wavefront keeps its own small tables: no workgroup barrier between the start and the
  __shared__ double s_R2[4][kMaxCams * 9];    // first frames (the one barrier ahead of everything only publishes READY = 0)
  const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63, group = blockIdx.x;
#ifdef VC_L0_STAMPS
  const long long l0t0_ = (long long)__builtin_amdgcn_s_memrealtime();
  int l0n_ = 1;
#define L0STAMP() do { if (blockIdx.x == gridDim.x / 2 && lane == 0 && l0n_ < 16) { v.dbg[(wv == 0 ? 0 : 16) + l0n_] = (long long)__builtin_amdgcn_s_memrealtime(); ++l0n_; } } while (0)
#define L0B(i) do { if (blockIdx.x == gridDim.x / 2 && lane == 0 && wv == 2 && it == 1) v.dbg[24 + (i)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define L0STAMP() do { } while (0)
#define L0B(i) do { } while (0)
#endif
  const int N = v.n_frames, D = v.D, C = v.n_cams, ldw = v.ldw, ldx = v.ldx, nW = D + 1, ncol = nW + 27;
  const Ctrl* ct = v.ctrl;
  const int a = group * 8, first = a + 1;
  const int q = first < N ? min(7, N - 1 - first + 1) : 0;     // interior frames e_1 .. e_q
  const int r = first + q;
  const bool has_r = q == 7 && r < N;
  const int nl = q > 0 ? (q - 1) / 2 : 0, mid = nl + 1, nr = q > 0 ? q - 1 - nl : 0;
  const size_t isz = (size_t)9 * ldx;
  // Hand-over through LDS only.  A workgroup-scope fence would also wait for the wavefront's outstanding GLOBAL loads and stores -- the
  // builder's requests for its next frames, the sweep's stores of the image it has just solved (1.7 us per frame, measured): here the
  // producer waits for its own LDS stores (lgkmcnt(0), nothing else) before it raises the word, the consumer's LDS reads follow its poll
  // in the wavefront's LDS queue, and wavefront-scope fences keep the compiler from moving LDS accesses across either.
  auto wait_ready = [&](int slot) {
    while (__hip_atomic_load(&READY[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  auto set_ready = [&](int slot) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): this wavefront's LDS stores have been performed
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) __hip_atomic_store(&READY[slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  if (tid < 12) READY[tid] = 0;
  __syncthreads();
  // ---- the builders' first requests go out before anything else: the tile list of their first frame ahead of the control record, the
  // frame's records as soon as that record says which linearisation buffer is current -- under the tables and the barrier below
  // Who builds what (round 5, second cut): the SWEEPS build their own first frame -- they would only wait for it -- while the builders
  // already work on the second ones: wavefront 0 the first frame of the left sweep (the middle if there is none), wavefront 1 the first of
  // the right sweep, builder 2 the rest of the left sweep's frames and then the middle, builder 3 the rest of the right sweep's and then
  // the separator a.  (With the builders alone the sweeps ran at the builders' ~4 us per frame instead of their own 2.8.)
  const int b = wv;                                                // (index of this wavefront's scratch and tables)
  const int nmine = q == 0 ? (wv == 3 ? 1 : 0)
                  : wv == 0 ? 1 : wv == 1 ? (nr > 0 ? 1 : 0) : wv == 2 ? (nl > 0 ? nl : 0) : (nr > 0 ? nr - 1 : 0) + 1;
  auto slot_of = [&](int i) {
    return q == 0 ? 0 : wv == 0 ? (nl > 0 ? 1 : mid) : wv == 1 ? q : wv == 2 ? (i < nl - 1 ? 2 + i : mid) : (i < nr - 1 ? q - 1 - i : 0);
  };
  int sp_col = -1;
#pragma unroll
  for (int a2 = 0; a2 < 15; ++a2) sp_col = (lane == a2) ? v.imu_param_col[a2] : sp_col;
  auto load_fct = [&](int f) { return (lane < C && f < N) ? v.frame_cam_tile[(size_t)f * C + lane] : -1; };
  int fctA = -1, fctB = -1;                                      // tile lists of the builder's first two frames
  if (nmine > 0) fctA = load_fct(a + slot_of(0));
  if (nmine > 1) fctB = load_fct(a + slot_of(1));
  if (ct->done) return;                                          // (uniform over the workgroup)
  const int cur = ct->cur;
  const double* segc = v.segb[cur];
  const int init_scale = ct->init_scale, reuse = ct->reuse_diag;
  const double radius = ct->radius;
  // per-lane offsets into a frame's two IMU records, once for all frames (the records' bases are wave-uniform: scalar base + 32-bit
  // lane offset per load, no 64-bit address arithmetic per lane and load -- issuing a frame's ~45 loads took 1.2 us of the builder's 4)
  const bool par = lane < 15 && sp_col >= 0;
  const int o_wc = par ? kSegWc + lane : 0, o_wp = par ? kSegWp + lane : 0, s_par = par ? 15 : 0;
  const int o_acc0 = kSegAcc + lane, o_acc1 = kSegAcc + (lane + 64 < 81 ? lane + 64 : 0), o_app0 = kSegApp + lane, o_app1 = kSegApp + (lane + 64 < 81 ? lane + 64 : 0);
  const int l9 = lane < 9 ? lane : 0;
  int o_ii[4];
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) { const int e = lane + 64 * qq, a2 = e >> 4, b2 = e & 15; o_ii[qq] = a2 < 15 ? ((b2 < 15) ? kSegHii + a2 * 15 + b2 : kSegGi + a2) : 0; }
  auto issue = [&](int f_in, int fct_f, InitLoads<CMAX>& R) {
    const int f = __builtin_amdgcn_readfirstlane(f_in);
    const bool has_c = f >= 1, has_p = f + 1 < N;                               // wave-uniform
    // (an absent record: the base falls back to record 0 -- always there for N >= 2 -- and every value is dropped by its select)
    const double* rcs = segc + (size_t)(has_c ? f - 1 : 0) * kSegStride;
    const double* rps = segc + (size_t)(has_p ? f : 0) * kSegStride;
#pragma unroll
    for (int cc = 0; cc < CMAX; ++cc) {
      const int tc = __builtin_amdgcn_readfirstlane(__shfl(fct_f, cc, 64));
#pragma unroll
      for (int qq = 0; qq < 3; ++qq) R.gv[cc][qq] = 0.0;
      if (cc < C && tc >= 0) {
        const double* g = v.Gb[cur] + (size_t)tc * kGPack;
#pragma unroll
        for (int qq = 0; qq < 3; ++qq) { const int e = qq * 64 + lane; const double x = g[e < kGPack ? e : 0]; R.gv[cc][qq] = e < kGPack ? x : 0.0; }
      }
    }
    {
      const bool tile_cost = lane < C && fct_f >= 0, blk_cost = lane == 8 && has_c;
      const double* pcst = tile_cost ? v.tile_costb[cur] + fct_f : (blk_cost ? v.seg_costb[cur] + (f - 1) : v.cg);
      const double x = *pcst;
      R.cost_in = (tile_cost || blk_cost) ? x : 0.0;
    }
    double xa[9], xb[9], xc0, xc1, xp0, xp1, xi[4];
#pragma unroll
    for (int i = 0; i < 9; ++i) { xa[i] = rcs[o_wc + i * s_par]; xb[i] = rps[o_wp + i * s_par]; }
    xc0 = rcs[o_acc0]; xc1 = rcs[o_acc1]; xp0 = rps[o_app0]; xp1 = rps[o_app1];
    const double gc = rcs[kSegGc + l9], gp = rps[kSegGp + l9], s2 = v.cscale2[(size_t)f * 9 + l9], dgv = v.cdiag[(size_t)f * 9 + l9];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) xi[qq] = rcs[o_ii[qq]];
#pragma unroll
    for (int i = 0; i < 9; ++i) R.pre[i] = ((par && has_c) ? xa[i] : 0.0) + ((par && has_p) ? xb[i] : 0.0);
    const bool in1 = lane + 64 < 81, in9 = lane < 9;
    R.a_imu[0] = (has_c ? xc0 : 0.0) + (has_p ? xp0 : 0.0);
    R.a_imu[1] = ((in1 && has_c) ? xc1 : 0.0) + ((in1 && has_p) ? xp1 : 0.0);
    R.g_imu = ((in9 && has_c) ? gc : 0.0) + ((in9 && has_p) ? gp : 0.0);
    R.sc2_in = (in9 && !init_scale) ? s2 : 0.0;
    R.dg_in = (in9 && reuse) ? dgv : 0.0;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) R.isum_in[qq] = ((lane + 64 * qq) < 240 && has_c) ? xi[qq] : 0.0;
  };
  // two frames' records in flight (two register sets): a frame's loads are requested BEHIND the hand-over of the frame two before it --
  // issuing them (1.3 us of instructions) no longer sits between a frame's arrival and its hand-over
  InitLoads<CMAX> RA, RB;
  if (nmine > 0) issue(a + slot_of(0), fctA, RA);
  if (nmine > 1) issue(a + slot_of(1), fctB, RB);
  // the builder's own tables (camera descriptors, rotations R_ck) and what its lane's column is -- under its first frame's loads
  int col_ci = 0;
  if (nmine > 0) {                                                 // (wave-uniform)
    if (lane < kMaxCams) s_cd2[b][lane] = v.cd[lane];
    if (lane < C) { double Rm[9]; quat_to_R(v.cams[cur] + (size_t)lane * kCamStride, Rm); for (int k2 = 0; k2 < 9; ++k2) s_R2[b][lane * 9 + k2] = Rm[k2]; }
    if (lane < ncol) {
      const int col = lane, e = col - nW;
      int cam = 255, loc = 0, skip = (e >= 18 && e < 27) ? 1 : 0;
      if (col < D) {
        const int cc = v.col_cam[col];
        cam = cc >= 0 ? cc : 255; loc = v.col_local[col];
#pragma unroll
        for (int a2 = 0; a2 < 15; ++a2) skip |= (v.imu_param_col[a2] == col) ? 1 : 0;
      }
      col_ci = cam | (loc << 8) | (skip << 16);
    }
    wave_lds_sync_local();
  }
  // ============================================================ building (k_chain_init's work, frame by frame; all four wavefronts) ====
  double* Gw = BW[b];
  double* Hs = Gw + C * kGStride;
  double* As = Hs + 42;
  double* gs = As + 81;
  double* ls = gs + 9;
  int* tcam = reinterpret_cast<int*>(ls + 9);
  int* tinv = tcam + kMaxCams;
  double gsum[CMAX][3];
  double isum[4] = {0.0, 0.0, 0.0, 0.0};
  double csum = 0.0;
  int po[3], pm[3];
#pragma unroll
  for (int qq = 0; qq < 3; ++qq) {
    const int e = qq * 64 + lane;
    int rr = 0;
#pragma unroll
    for (int a2 = 1; a2 < 16; ++a2) rr += (e >= a2 * 16 - (a2 * (a2 - 1)) / 2) ? 1 : 0;
    const int cidx = rr + (e - (rr * 16 - (rr * (rr - 1)) / 2));
    po[qq] = e < kGPackGrad ? rr * 16 + cidx : (e < kGPack ? kGGrad + (e - kGPackGrad) : -1);
    pm[qq] = (e < kGPackGrad && cidx != rr) ? cidx * 16 + rr : -1;
  }
#pragma unroll
  for (int cc = 0; cc < CMAX; ++cc)
#pragma unroll
    for (int qq = 0; qq < 3; ++qq) gsum[cc][qq] = 0.0;
  // what this lane's image column is (one column per lane: D + 28 <= 64), once for all frames -- the column phase then runs without a
  // divergent branch: a camera's column is three weighted entries per row of its tile's Gram block (a rotation column: -R's column over
  // entries 3..5; a unit column: one entry with weight 1), A's and g's columns come from the frame's own block; everybody loads from valid
  // addresses and keeps what is his by selects.  (As `if (camera column) { if (rotation) .. else .. } if (g) .. if (A) ..` the phase took
  // 2.2 of the ~4 us a builder needs per frame: five serialised branch regions, each with its own LDS round trips.)
  int ckind = 0, ccam = 0, cjb = 0, cidx0 = 0, cidx1 = 0, cidx2 = 0, cic = 0;
  double cw0 = 0.0, cw1 = 0.0, cw2 = 0.0, cR[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) cR[i] = 0.0;
  if (nmine > 0 && lane < ncol) {
    const int col = lane, ci = col_ci, e = col - nW;
    const int cc = ci & 255, j = (ci >> 8) & 255;
    cic = col < nW ? col : nW + (e >= 0 ? e % 9 : 0);
    cjb = (e >= 0) ? e % 9 : 0;
    if ((ci >> 16) || (e >= 0 && e < 9)) ckind = 0;             // IMU-record columns (written with the records' values), C, B
    else if (col == D) ckind = 2;
    else if (e >= 9 && e < 18) ckind = 3;
    else if (col < D && cc < kMaxCams) {
      ckind = 1; ccam = cc;
      const int flags = s_cd2[b][cc].flags, nrot = (flags & kCamRotFree) ? 3 : 0, ntr = (flags & kCamTransFree) ? 3 : 0;
      const double* Rm = s_R2[b] + cc * 9;
#pragma unroll
      for (int i = 0; i < 9; ++i) cR[i] = Rm[i];
      if (j < nrot) { cidx0 = 3; cidx1 = 4; cidx2 = 5; cw0 = -Rm[j]; cw1 = -Rm[3 + j]; cw2 = -Rm[6 + j]; }
      else { const int jj = (j < nrot + ntr) ? j - nrot : 6 + (j - nrot - ntr); cidx0 = jj; cidx1 = jj; cidx2 = jj; cw0 = 1.0; }
    } else if (col < nW) ckind = 4;                                // a border column nobody owns: zeros
  }
  auto build = [&](int it, InitLoads<CMAX>& R, int& fct) {
    const int slot = slot_of(it), f = a + slot;
    double* im = IMG[slot];
    const int fct2 = (it + 2 < nmine) ? load_fct(a + slot_of(it + 2)) : -1;      // the tile list of the frame that takes this register set next
    L0B(0);
    const unsigned long long present = __ballot(lane < C && fct >= 0);
    const int nt = __popcll(present);
    if (lane < kMaxCams) {
      const bool have = lane < C && fct >= 0;
      tinv[lane] = have ? __popcll(present & ((1ull << lane) - 1ull)) : -1;
    }
    wave_lds_sync_local();
    if (lane < C && fct >= 0) tcam[tinv[lane]] = lane;
    csum += R.cost_in;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) isum[qq] += R.isum_in[qq];
    int slot_c = 0;
#pragma unroll
    for (int cc = 0; cc < CMAX; ++cc) {
      const bool have = cc < C && ((present >> cc) & 1ull);
      if (have) {
#pragma unroll
        for (int qq = 0; qq < 3; ++qq) {
          if (po[qq] >= 0) Gw[slot_c * kGStride + po[qq]] = R.gv[cc][qq];
          if (pm[qq] >= 0) Gw[slot_c * kGStride + pm[qq]] = R.gv[cc][qq];
          gsum[cc][qq] += R.gv[cc][qq];
        }
        ++slot_c;
      }
    }
    if (sp_col >= 0) {          // the IMU-parameter columns of the border, straight from the records
#pragma unroll
      for (int i = 0; i < 9; ++i) im[i * kL0Ld + sp_col] = R.pre[i];
    }
    const double a_imu[2] = {R.a_imu[0], R.a_imu[1]};
    const double g_imu = R.g_imu, sc2_in = R.sc2_in, dg_in = R.dg_in;
    wave_lds_sync_local();
    L0B(1);
    L0B(2);
    if (lane < 42) {
      double hval = 0.0;
      for (int t = 0; t < nt; ++t) {
        const int cc = tcam[t];
        const double* Rm = s_R2[b] + cc * 9;
        const double* g = Gw + t * kGStride;
        if (lane < 36) {
          const int i = lane / 6, j = lane % 6, a2 = i / 3, ii = i % 3, b2 = j / 3, jj = j % 3;
          double sacc = 0.0;
#pragma unroll
          for (int pp = 0; pp < 3; ++pp)
#pragma unroll
            for (int qq = 0; qq < 3; ++qq) sacc += Rm[3 * pp + ii] * g[(3 * a2 + pp) * 16 + 3 * b2 + qq] * Rm[3 * qq + jj];
          hval += (a2 == b2) ? sacc : -sacc;
        } else {
          const int i = lane - 36, a2 = i / 3, ii = i % 3;
          const int nk = model_nk(s_cd2[b][cc].model);
          double sacc = 0.0;
#pragma unroll
          for (int pp = 0; pp < 3; ++pp) sacc += Rm[3 * pp + ii] * gram_grad(g, 3 * a2 + pp, nk);
          hval += (a2 == 0) ? -sacc : sacc;
        }
      }
      Hs[lane] = hval;
    }
    wave_lds_sync_local();
    L0B(3);
    double aval[2];
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      const int e = lane + 64 * qq, i = e / 9, j = e % 9;
      aval[qq] = a_imu[qq] + ((e < 81 && i < 6 && j < 6) ? Hs[i * 6 + j] : 0.0);
      if (e < 81) As[e] = aval[qq];
    }
    const double gval = g_imu + ((lane < 6) ? Hs[36 + lane] : 0.0);
    double hd = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int e = i * 10;
      const double d = readlane_f64(aval[e >> 6], e & 63);
      if (lane == i) hd = d;
    }
    if (lane < 9) {
      double sc2 = sc2_in, dg = dg_in;
      if (init_scale) { sc2 = jacobi_scale2(hd); v.cscale2[(size_t)f * 9 + lane] = sc2; }
      if (!reuse) { dg = lm_clamped_diag(hd, sc2); v.cdiag[(size_t)f * 9 + lane] = dg; }
      const double lam = dg / (radius * sc2);
      v.clam[(size_t)f * 9 + lane] = lam;
      v.cg[(size_t)f * 9 + lane] = gval;
      gs[lane] = gval; ls[lane] = lam;
    }
    wave_lds_sync_local();
    L0B(4);
    // the image's border and A columns -> LDS (branch-free, see the lane constants above)
    {
      const int t = tinv[ccam];
      const double* g = Gw + (t >= 0 ? t : 0) * kGStride;
      double g0[6], g1[6], g2[6], av[9], gv9[9];
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) { g0[rr] = g[rr * 16 + cidx0]; g1[rr] = g[rr * 16 + cidx1]; g2[rr] = g[rr * 16 + cidx2]; }
#pragma unroll
      for (int i = 0; i < 9; ++i) { av[i] = As[i * 9 + cjb]; gv9[i] = gs[i]; }
      const double lj = ls[cjb];
      double u[6];
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) u[rr] = g0[rr] * cw0 + g1[rr] * cw1 + g2[rr] * cw2;
      double val[9];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        val[i] = -(cR[i] * u[0] + cR[3 + i] * u[1] + cR[6 + i] * u[2]);
        val[3 + i] = cR[i] * u[3] + cR[3 + i] * u[4] + cR[6 + i] * u[5];
        val[6 + i] = 0.0;
      }
      const bool cam_live = ckind == 1 && t >= 0;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const double x = cam_live ? val[i] : (ckind == 2 ? gv9[i] : (ckind == 3 ? av[i] + ((i == cjb) ? lj : 0.0) : 0.0));
        if (ckind != 0) im[i * kL0Ld + cic] = x;
      }
    }
    L0B(5);
    set_ready(slot);
    L0B(6);
    if (wv == 2) L0STAMP();                 // builder 2: frame handed over
    fct = fct2;
    if (it + 2 < nmine) issue(a + slot_of(it + 2), fct, R);
  };
  for (int it = 0; it < nmine; it += 2) {
    build(it, RA, fctA);
    if (it + 1 < nmine) build(it + 1, RB, fctB);
  }
#ifdef VC_L0_STAMPS
  if (blockIdx.x == gridDim.x / 2 && lane == 0 && (wv == 0 || wv == 2)) v.dbg[wv == 0 ? 0 : 16] = l0t0_;
  if (wv == 0 || wv == 2) L0STAMP();      // 1: behind the first barrier
#endif
  if (wv < 2) {
    // ======================================================= the two sweeps (k_chain_fwd2<1> at s = 1, level 0) =======================
    // this sweep's own frame's contribution to the chunk sums: to LDS for builder 2, which needs it at the very end -- published where the
    // sweep has time (at the hand-over barrier), not between its frame's build and its first elimination
    auto publish_sums = [&]() {
      const int nsum_ = C * kGStride + kGStride;
#pragma unroll
      for (int cc = 0; cc < CMAX; ++cc)
        if (cc < C) {
#pragma unroll
          for (int qq = 0; qq < 3; ++qq) {
            if (po[qq] >= 0) SUMW[wv][cc * kGStride + po[qq]] = gsum[cc][qq];
            if (pm[qq] >= 0) SUMW[wv][cc * kGStride + pm[qq]] = gsum[cc][qq];
          }
        }
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) SUMW[wv][C * kGStride + qq * 64 + lane] = isum[qq];
      if (lane < 9) SUMW[wv][nsum_ + lane] = csum;
      set_ready(8 + wv);
    };
    const int wave = wv;
    auto gsync = [&]() { wave_lds_sync_local(); };
    const int c = lane, e0 = c - nW;
    const int role = c < nW ? 0 : (e0 < 9 ? 1 : e0 < 18 ? 2 : e0 < 27 ? 3 : 4);     // 0 border (W | g), 1 C, 2 A, 3 B, 4 none
    const int pc = c < nW ? c : (c < ncol ? ldw + e0 : 0);
    const int sub = e0 < 9 ? e0 : e0 < 18 ? e0 - 9 : e0 - 18;
    const int icol = role == 0 ? c : nW + sub;                                       // the lane's column in an LDS image (roles 0, 2; role 1 lanes: A's column `sub`)
    if (q == 0) {              // a separator without interior frames: its image as the builders formed it, no coupling to the right
      if (wave == 0 && a < N) {
        wait_ready(0);
        double* img = v.cW + (size_t)a * isz;
        if (role == 0 || role == 2) {
#pragma unroll
          for (int k = 0; k < 9; ++k) img[k * ldx + pc] = IMG[0][k * kL0Ld + icol];
        } else if (role == 3) {
#pragma unroll
          for (int k = 0; k < 9; ++k) img[k * ldx + pc] = 0.0;
        }
      }
      return;
    }
    const int dir = wave == 0 ? 1 : -1, cnt = wave == 0 ? nl : nr, i0 = wave == 0 ? 1 : q;
    double* XS = XS2[wave]; double* An = An2[wave]; double* Ls = Ls2[wave];
    // the image columns of frame e as this sweep needs them: border and A from the builders' LDS image, the couplings from the IMU block
    // records (B of frame f = the `prev x cur` block of record f, rows = frame f); branch-free, a lane without a value reads a valid
    // address and keeps a zero
    auto load_b = [&](int e, bool first_of_sweep, bool with_image, double* xb) {
      const double* p = segc; int st = 0; bool ok = false;
      if (with_image) {
        if (role == 1 && first_of_sweep) {
          if (dir > 0) { p = segc + (size_t)a * kSegStride + kSegBpc + sub * 9; st = 1; ok = true; }                       // row `sub` of B_a
          else { ok = e + 1 < N; p = ok ? segc + (size_t)e * kSegStride + kSegBpc + sub : segc; st = ok ? 9 : 0; }        // column `sub` of B_e
        } else if (role == 3) {
          if (dir > 0) { ok = e + 1 < N; p = ok ? segc + (size_t)e * kSegStride + kSegBpc + sub : segc; st = ok ? 9 : 0; }
          else { p = segc + (size_t)(e - 1) * kSegStride + kSegBpc + sub * 9; st = 1; ok = true; }                          // row `sub` of B_{e-1}
        }
      }
      double y[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) y[k] = p[k * st];
#pragma unroll
      for (int k = 0; k < 9; ++k) xb[k] = ok ? y[k] : 0.0;
    };
    auto take_img = [&](int e, bool with_image, double* x) {      // border and A columns, once the builder says so; added to the couplings
      if (!with_image) return;                                       // (wave-uniform)
      wait_ready(e - a);
      const double* im = IMG[e - a];
      if (role == 0 || role == 2) {
#pragma unroll
        for (int k = 0; k < 9; ++k) x[k] = im[k * kL0Ld + icol];
      }
    };
    double xin[9], o[9], dacc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { dacc[k] = 0.0; o[k] = 0.0; }
    {
      const int e = a + (cnt > 0 ? i0 : mid);
      const bool wi = cnt > 0 || wave == 0;
      load_b(e, true, wi, xin);
      take_img(e, wi, xin);
    }
    if (role == 2) {
#pragma unroll
      for (int k = 0; k < 9; ++k) An[k * 9 + sub] = xin[k];
    }
    // one elimination: A (in An) = L L^T, all columns solved, image stored, [X_s | X_n] to XS, out = [X_s | X_n]^T (column)
    const ElimLds elds = {XS, An, Ls};
    auto eliminate = [&](int e, double* x, double* out) { chain_eliminate<false>(v, v.cW + (size_t)e * isz, ldx, role, pc, sub, c == 0, elds, x, out); };
    for (int j = 0; ; ++j) {
      const bool at_mid = j == cnt;
      if (at_mid) {
        if (wave == 1) {
          if (cnt == 0) {
#pragma unroll
            for (int k = 0; k < 9; ++k) MID[k * W + c] = 0.0;
          }
#pragma unroll
          for (int k = 0; k < 9; ++k) SEPR[k * W + c] = dacc[k];
        }
        publish_sums();
        __syncthreads();                 // (all four wavefronts: the builders arrive when their frames are built)
        if (wave == 1) return;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          if (role == 0 || role == 2) xin[k] += MID[k * W + c];
          else if (role == 3 && nr > 0) xin[k] = MID[k * W + nW + sub];
        }
        if (role == 2) {
#pragma unroll
          for (int k = 0; k < 9; ++k) An[k * 9 + sub] = xin[k];
        }
      }
      const int e = a + (at_mid ? mid : i0 + dir * j);
      int en = 0; bool wn = false;
      if (!at_mid) {                      // the frame after this one: its couplings requested now, its image taken behind the elimination
        en = a + (j + 1 < cnt ? i0 + dir * (j + 1) : mid); wn = j + 1 < cnt || wave == 0;
        load_b(en, false, wn, o);
      }
      double x[9], out[18];
#pragma unroll
      for (int k = 0; k < 9; ++k) x[k] = xin[k];
      if (wv == 0) L0STAMP();               // sweep 0: frame taken, elimination starts
      eliminate(e, x, out);
      if (wv == 0) L0STAMP();               // ... and is done
#pragma unroll
      for (int k = 0; k < 9; ++k) dacc[k] += out[k];
      if (at_mid) {
        if (has_r && role < 4) {
          double* ri = v.rX[0] + (size_t)(r / 8) * isz + pc;
          const bool keep = role == 0 || role == 2;
          const int src = role == 2 ? nW + sub : c;
#pragma unroll
          for (int k = 0; k < 9; ++k) ri[k * ldx] = keep ? -out[9 + k] - SEPR[k * W + src] : 0.0;
        }
        if (role == 3) {
          double* img = v.cW + (size_t)a * isz + pc;
#pragma unroll
          for (int k = 0; k < 9; ++k) img[k * ldx] = has_r ? -out[k] : 0.0;
        }
        break;
      }
      if (wave == 1 && j + 1 == cnt) {
#pragma unroll
        for (int k = 0; k < 9; ++k) MID[k * W + c] = -out[9 + k];
      } else {
        take_img(en, wn, o);
#pragma unroll
        for (int k = 0; k < 9; ++k) xin[k] = o[k] - (role == 3 ? 0.0 : out[9 + k]);
        if (role == 2) {
#pragma unroll
          for (int k = 0; k < 9; ++k) An[k * 9 + sub] = xin[k];
        }
      }
    }
    // ---- the left separator absorbs its group: its border and A as builder 3 formed them, minus the group's update
    wait_ready(0);
    if (role == 0 || role == 1) {
      const int pcd = role == 1 ? ldw + 9 + sub : pc;
      double* img = v.cW + (size_t)a * isz + pcd;
#pragma unroll
      for (int k = 0; k < 9; ++k) img[k * ldx] = IMG[0][k * kL0Ld + icol] - dacc[k];
    }
#ifdef VC_L0_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
    if (wv == 0) L0STAMP();
#endif
    return;
  }
  // ============================================================ the two builders: chunk sums ==========================================
  // ---- chunk sums (chunk = group): wavefronts 0, 1 (above) and builder 3 hand their sums over through LDS, builder 2 adds them to its own
  // in fixed order (2, 3, 0, 1) and writes the record
  __syncthreads();                       // (the sweeps' hand-over barrier: counts all four wavefronts)
  const int nsum = C * kGStride + kGStride;
  if (wv == 3) {
#pragma unroll
    for (int cc = 0; cc < CMAX; ++cc)
      if (cc < C) {
#pragma unroll
        for (int qq = 0; qq < 3; ++qq) {
          if (po[qq] >= 0) SUMW[3][cc * kGStride + po[qq]] = gsum[cc][qq];
          if (pm[qq] >= 0) SUMW[3][cc * kGStride + pm[qq]] = gsum[cc][qq];
        }
      }
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) SUMW[3][C * kGStride + qq * 64 + lane] = isum[qq];
    if (lane < 9) SUMW[3][nsum + lane] = csum;
    set_ready(8 + 3);
    return;
  }
  // builder 2: its own sums through its (now free) Gram scratch, expanded like the others'
#pragma unroll
  for (int cc = 0; cc < CMAX; ++cc)
    if (cc < C) {
#pragma unroll
      for (int qq = 0; qq < 3; ++qq) {
        if (po[qq] >= 0) Gw[cc * kGStride + po[qq]] = gsum[cc][qq];
        if (pm[qq] >= 0) Gw[cc * kGStride + pm[qq]] = gsum[cc][qq];
      }
    }
  wave_lds_sync_local();
  wait_ready(8 + 3);
  const bool sw = q > 0;                  // (a group without interior frames: the sweeps built nothing and published nothing)
  if (sw) { wait_ready(8 + 0); wait_ready(8 + 1); }
  double* part = v.part + (size_t)group * v.part_stride;
  for (int e = lane; e < C * kGStride; e += 64) part[D * D + D + e] = ((Gw[e] + SUMW[3][e]) + (sw ? SUMW[0][e] : 0.0)) + (sw ? SUMW[1][e] : 0.0);
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {
    const int e = C * kGStride + qq * 64 + lane;
    part[D * D + D + e] = ((isum[qq] + SUMW[3][e]) + (sw ? SUMW[0][e] : 0.0)) + (sw ? SUMW[1][e] : 0.0);
  }
  for (int e = C * kGStride + 256 + lane; e < nsum; e += 64) part[D * D + D + e] = 0.0;
  {
    // fixed order: the nine lanes of builder 2, of builder 3, of wavefront 0, of wavefront 1
    const double t2 = (lane < 9) ? csum : 0.0, t3 = (lane < 9) ? SUMW[3][nsum + lane] : 0.0;
    const double t0 = (lane < 9 && sw) ? SUMW[0][nsum + lane] : 0.0, t1 = (lane < 9 && sw) ? SUMW[1][nsum + lane] : 0.0;
    double u2 = 0.0, u3 = 0.0, u0 = 0.0, u1 = 0.0;
#pragma unroll
    for (int kk = 0; kk < 9; ++kk) { u2 += readlane_f64(t2, kk); u3 += readlane_f64(t3, kk); u0 += readlane_f64(t0, kk); u1 += readlane_f64(t1, kk); }
    if (lane == 0) part[v.part_stride - 1] = ((u2 + u3) + u0) + u1;
  }
}

// Back-substitution of one level: delta_e = -L^-T (z + Y delta_s + X_s delta_a + X_n delta_next), right to left inside
// the group.  Everything that does not depend on the chain (z + Y delta_s + X_s delta_a, the rows of X_n, the columns of L)
// is formed by lane (frame i, row k) beforehand; the dependent part is one 9 x 9 product and a triangular solve per frame,
// exchanged through v_readlane.  At level 0 the wavefront also moves its frames (T <- T exp(delta), v <- v + dv) and
// publishes their step terms.
// phase stamps of one group of k_chain_back per level (profiling builds only: -DVC_BACK_STAMPS, tools/back_stamps.py): dbg[8 lvl + i]
#ifdef VC_BACK_STAMPS
#define BSTAMP(i) do { if (blockIdx.x == (top ? 0 : gridDim.x / 2) && threadIdx.x == 0 && lvl < 4) v.dbg[8 * lvl + (i)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define BSTAMP(i) do { } while (0)
#endif
constexpr int kBackT0Frames = 1;      // frames per extra workgroup of the top-level launch (t0 = z + Y delta_s)
// FUSED: all levels below the top one in ONE launch (round 4).  A group needs the steps of its two separators, which a group of a
// level above produces: instead of a kernel boundary per level (four launches of ~9.5 us at cfg3 whose dependent chain is 2.5-3.6 us;
// the rest is launch, first loads and memory latency) every producer publishes its frames' steps with device-coherent stores and then
// raises the frames' ready words (the pass number), and a consumer -- which has requested everything else it needs beforehand --
// polls the two ready words and takes the steps with device-coherent loads: no fence, no cache write-back (the pattern of the
// cross-stream hand-overs).  Workgroups are laid out top level first: a group only ever waits for workgroups with a
// smaller index, which the dispatcher has started before it.
struct BackLevels { int n; int start[8]; int stride[8]; int m[8]; int two[8]; };
template <bool FUSED>
__device__ __forceinline__ void chain_back_group(const DevView& v, int s, int m, int top, int lvl, int two, int group, double* ds, double* dl) {
  const int done = v.ctrl->done;        // looked at once the level's inputs have been requested (see k_chain_fwd)
  const int lane = threadIdx.x;
#ifdef VC_BACK_STAMPS
  const long long bs0_ = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  const int N = v.n_frames, D = v.D, ldw = v.ldw, ldx = v.ldx;
  const long gs = (long)m * s;
  const int a = top ? -1 : (int)(group * gs);
  const int first = top ? 0 : a + s;
  const size_t isz = (size_t)9 * ldx;
  for (int j = lane; j < D; j += 64) ds[j] = v.delta_s[j];
  const int q = first < N ? (top ? (N - 1) / s + 1 : min(m - 1, (N - 1 - first) / s + 1)) : 0;
  const int r = first + q * s;                                  // right separator (or past the end)
  double da[9], dn[9];
  const bool has_a = a >= 0 && a < N, has_r = !top && q > 0 && r < N;
  if (!FUSED) {
#pragma unroll
    for (int k = 0; k < 9; ++k) { da[k] = has_a ? v.cdelta[(size_t)a * 9 + k] : 0.0; dn[k] = has_r ? v.cdelta[(size_t)r * 9 + k] : 0.0; }
  }
  const int fi = lane / 9, k = lane % 9;
  const bool mine = fi < q;
  const int e = first + (mine ? fi : 0) * s;
  // a group eliminated from both ends (k_chain_fwd2: full groups of a level launched two-sided): frames right of the middle hang
  // on the right separator and on the frame to their left
  const bool two_sided = two && !top && q >= 1;
  const int fmid = (q - 1) / 2;                                  // lane group of the middle frame (interior index fmid + 1)
  double t = 0.0, dinv = 1.0, Qrow[9], Lcol[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) { Qrow[c] = 0.0; Lcol[c] = 0.0; }
  if (top) wave_lds_sync();      // (the levels below need delta_s only in the bottom level's epilogue, behind its own synchronisation)
#ifdef VC_BACK_STAMPS
  const long long bs1_ = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  if (mine) {
    const double* img = v.cW + (size_t)e * isz;
    const double* Wr = img + (size_t)k * ldx;
#pragma unroll
    for (int c = 0; c < 9; ++c) { Qrow[c] = Wr[ldw + 18 + c]; Lcol[c] = (c > k) ? img[c * ldx + ldw + 9 + k] : 0.0; }
    dinv = 1.0 / Wr[ldw + 9 + k];
    double acc;
    if (top) { acc = Wr[D]; for (int j = 0; j < D; ++j) acc += Wr[j] * ds[j]; }      // (its own frames: the extra workgroups run beside it)
    else acc = v.ct0[(size_t)e * 9 + k];
    t = acc;
  }
  double Xs[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) Xs[c] = (mine && a >= 0) ? v.cW[(size_t)e * isz + (size_t)k * ldx + ldw + c] : 0.0;
  if (done) return;
  if (FUSED) {
    // everything else is on its way: now the separators' steps (the levels above publish them, see the kernel's header)
    if (lane == 0) {
      long long n = 0;
      while ((has_a && __hip_atomic_load(v.cready + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v.pass_id) ||
             (has_r && __hip_atomic_load(v.cready + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v.pass_id)) {
        __builtin_amdgcn_s_sleep(2);
        // (never seen: would take a dispatcher that starts workgroups out of order.  Loud and lossless like the cross-stream waits where a
        //  resume exists -- the pass is marked void, the host reports it and repeats the pass with events; a numeric-failure mark otherwise)
        if (++n > 4000000) {
          if (v.sync_seq > 0) mark_sync_timeout(v, v.sync_seq); else atomicAdd(&v.flags[4 + 2 * v.par], 1);
          atomicAdd((unsigned long long*)&v.dbg[21], 1ull);
          break;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    (void)__builtin_amdgcn_readfirstlane(lane);      // (the wavefront goes on together)
#pragma unroll
    for (int k2 = 0; k2 < 9; ++k2) {
      da[k2] = has_a ? __hip_atomic_load(v.cdelta + (size_t)a * 9 + k2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
      dn[k2] = has_r ? __hip_atomic_load(v.cdelta + (size_t)r * 9 + k2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    }
  }
  if (mine && a >= 0) {
    const bool right = two_sided && fi > fmid;
#pragma unroll
    for (int c = 0; c < 9; ++c) t += Xs[c] * (right ? dn[c] : da[c]);       // (dn: still the right separator's step here)
  }
#ifdef VC_BACK_STAMPS
  if (__builtin_amdgcn_readfirstlane(__double2loint(t)) == 0x7fffffff) return;      // (forces t before the stamp)
  if (blockIdx.x == (top ? 0 : gridDim.x / 2) && threadIdx.x == 0 && lvl < 4) { v.dbg[8 * lvl] = bs0_; v.dbg[8 * lvl + 1] = bs1_; }      // (not in passes that exit early)
#endif
  BSTAMP(2);
  double my = 0.0;
  if (two_sided) {
    // the middle first (a on its left, r on its right), then outwards on both sides at once: two independent chains of 9 x 9
    // products and triangular solves, interleaved instruction by instruction
    {
      double y = t;
#pragma unroll
      for (int c = 0; c < 9; ++c) y += Qrow[c] * dn[c];
#pragma unroll
      for (int j = 8; j >= 0; --j) {
        const double xj = readlane_f64(y * dinv, fmid * 9 + j);
        dn[j] = -xj;
        y -= Lcol[j] * xj;
      }
      if (fi == fmid) {
#pragma unroll
        for (int j = 0; j < 9; ++j) my = (j == k) ? dn[j] : my;
      }
    }
    double dl_[9], dr_[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) { dl_[j] = dn[j]; dr_[j] = dn[j]; }
    const int nleft = fmid, nright = q - 1 - fmid;
    for (int st = 1; st <= max(nleft, nright); ++st) {
      const bool hl = st <= nleft, hr = st <= nright;              // wave-uniform
      const int il = hl ? fmid - st : 0, ir = hr ? fmid + st : 0;
      double yl = t, yr = t;
#pragma unroll
      for (int c = 0; c < 9; ++c) { yl += Qrow[c] * dl_[c]; yr += Qrow[c] * dr_[c]; }
#pragma unroll
      for (int j = 8; j >= 0; --j) {
        const double xl = readlane_f64(yl * dinv, il * 9 + j), xr = readlane_f64(yr * dinv, ir * 9 + j);
        if (hl) { dl_[j] = -xl; yl -= Lcol[j] * xl; }
        if (hr) { dr_[j] = -xr; yr -= Lcol[j] * xr; }
      }
      if (hl && fi == il) {
#pragma unroll
        for (int j = 0; j < 9; ++j) my = (j == k) ? dl_[j] : my;
      }
      if (hr && fi == ir) {
#pragma unroll
        for (int j = 0; j < 9; ++j) my = (j == k) ? dr_[j] : my;
      }
    }
  } else
  for (int i = q - 1; i >= 0; --i) {
    double y = t;
#pragma unroll
    for (int c = 0; c < 9; ++c) y += Qrow[c] * dn[c];
#pragma unroll
    for (int j = 8; j >= 0; --j) {
      const double xj = readlane_f64(y * dinv, i * 9 + j);
      dn[j] = -xj;
      y -= Lcol[j] * xj;
    }
    if (fi == i) {
#pragma unroll
      for (int j = 0; j < 9; ++j) my = (j == k) ? dn[j] : my;
    }
  }
#ifdef VC_BACK_STAMPS
  if (__builtin_amdgcn_readfirstlane(__double2loint(my)) == 0x7fffffff) return;
#endif
  BSTAMP(3);
  if (lvl != 0 && v.cready) {
    // frames of this level are separators of the levels below: device-coherent stores, then -- once the wavefront's stores have
    // been performed -- the frames' ready words
    if (mine) __hip_atomic_store(v.cdelta + (size_t)e * 9 + k, my, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (mine && k == 0) __hip_atomic_store(v.cready + e, v.pass_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (mine) v.cdelta[(size_t)e * 9 + k] = my;
  if (lvl != 0) { BSTAMP(4); return; }
  // ---- level 0: trial poses / velocities of the group's frames and their step terms
  if (mine) dl[(fi + 1) * 9 + k] = my;
  if (lane < 9) dl[lane] = da[lane];
  wave_lds_sync();
  const int base = top ? 0 : a, cnt = top ? q : (a < N ? q + 1 : 0), off = top ? 1 : 0;
  double gd = 0, dld = 0, step2 = 0, x2 = 0, g2 = 0, gmax = 0;
  if (lane < cnt) {
    const int f = base + lane;
    const int cur = v.ctrl->cur;
    // pinned frames step with the reduced system's solution; their gradient / damping terms are counted there, and only the
    // owner (not the rank that holds the ghost copy) counts the step and parameter norms
    const bool pin_f = (f == 0 && v.pin_first), pin_l = (f == N - 1 && v.pin_last);
    double d[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] = pin_f ? ds[v.sep_col0 + i] : pin_l ? ds[v.sep_col1 + i] : dl[(lane + off) * 9 + i];
    const double* pin = v.poses[cur] + (size_t)f * kPoseStride;
    double* pout = v.poses[1 - cur] + (size_t)f * kPoseStride;
    double Tin[7], Tout[7], dd[6];
    for (int i = 0; i < 7; ++i) Tin[i] = pin[i];
    for (int i = 0; i < 6; ++i) dd[i] = d[i];
    se3_plus(Tin, dd, Tout);
    for (int i = 0; i < 7; ++i) { pout[i] = Tout[i]; const double ee = Tout[i] - Tin[i]; step2 += ee * ee; x2 += Tin[i] * Tin[i]; }
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
  }
  // the group's step terms in one record (lanes >= cnt hold zeros; fixed summation order)
  gd = wave_sum(gd); dld = wave_sum(dld); step2 = wave_sum(step2); x2 = wave_sum(x2); g2 = wave_sum(g2);
#pragma unroll
  for (int o2 = 32; o2 > 0; o2 >>= 1) gmax = fmax(gmax, __shfl_down(gmax, o2, 64));
  if (lane == 0) {
    double* o = v.grp_part + (size_t)group * kNumScal;
    o[kScGd] = gd; o[kScDld] = dld; o[kScStep2] = step2; o[kScX2] = x2; o[kScG2] = g2; o[kScCost] = 0.0; o[kScGmax] = gmax; o[kScSq] = 0.0;
  }
  BSTAMP(4);
}

__global__ __launch_bounds__(64) void k_chain_back(DevView v, int s, int m, int top, int lvl, int two) {
  __shared__ double ds[192 + 8];
  __shared__ double dl[kChainM * 9];
  const int lane = threadIdx.x;
  const int D = v.D, ldx = v.ldx;
  const size_t isz = (size_t)9 * ldx;
  // The chain-independent part of every right-hand side, t0 = z + Y delta_s, for ALL frames: extra workgroups of the top level's
  // launch (which has one group and an idle chip beside it), lane = column so that a load instruction covers one row segment --
  // the levels below read one value per (frame, row) instead of walking D entries of a row per lane, 63 cache lines per load
  // instruction (2.5 us per level at D = 29, 7-16 us at D = 115; tools/back_stamps.py).
  if (top && blockIdx.x > 0) {
    if (v.ctrl->done) return;
    const int f = (int)blockIdx.x - 1;      // one frame per wavefront: its nine rows' loads go out together
    double dsl[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) { const int j = lane + 64 * u; dsl[u] = j < D ? v.delta_s[j] : (j == D ? 1.0 : 0.0); }      // (column D: z itself)
    double p[9];
#pragma unroll
    for (int kk = 0; kk < 9; ++kk) {
      const double* Wr = v.cW + (size_t)f * isz + (size_t)kk * ldx;
      double acc = 0.0;
#pragma unroll
      for (int w = 0; w < 3; ++w) { const int j = lane + 64 * w; if (j <= D) acc += Wr[j] * dsl[w]; }
      p[kk] = acc;
    }
#pragma unroll
    for (int kk = 0; kk < 9; ++kk) {
      const double t0 = wave_sum(p[kk]);
      if (lane == 0) v.ct0[(size_t)f * 9 + kk] = t0;
    }
    return;
  }
  chain_back_group<false>(v, s, m, top, lvl, two, (int)blockIdx.x, ds, dl);
}
// ... the same sums as a launch of their own, one frame per wavefront: what the one-launch back-substitution reads at wide borders
// (k_chain_back_path below recomputes the upper levels in every bottom group -- with their rows [Y | z] that is 58 KB per wavefront at
// D = 115, 230 MB through the L2 at 6250 frames: the one-launch form then took 218 us against 71 for the level-by-level kernels; with t0
// formed once per frame here it reads nine values per frame)
__global__ __launch_bounds__(256) void k_chain_t0(DevView v) {
  if (v.ctrl->done) return;
  const int lane = threadIdx.x & 63, f = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  if (f >= v.n_frames) return;
  const int D = v.D, ldx = v.ldx;
  double dsl[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) { const int j = lane + 64 * u; dsl[u] = j < D ? v.delta_s[j] : (j == D ? 1.0 : 0.0); }      // (column D: z itself)
  double p[9];
#pragma unroll
  for (int kk = 0; kk < 9; ++kk) {
    const double* Wr = v.cW + ((size_t)f * 9 + kk) * ldx;
    double x[3];
#pragma unroll
    for (int w = 0; w < 3; ++w) { const int j = lane + 64 * w; x[w] = Wr[j <= D ? j : 0]; }
    double acc = 0.0;
#pragma unroll
    for (int w = 0; w < 3; ++w) { const int j = lane + 64 * w; acc += (j <= D) ? x[w] * dsl[w] : 0.0; }
    p[kk] = acc;
  }
#pragma unroll
  for (int kk = 0; kk < 9; ++kk) {
    const double t0 = wave_sum(p[kk]);
    if (lane == 0) v.ct0[(size_t)f * 9 + kk] = t0;
  }
}
__global__ __launch_bounds__(64) void k_chain_back_levels(DevView v, BackLevels L) {
  __shared__ double ds[192 + 8];
  __shared__ double dl[kChainM * 9];
  int li = 0;
#pragma unroll
  for (int i = 1; i < 8; ++i) if (i < L.n && (int)blockIdx.x >= L.start[i]) li = i;
  const int lvl = L.n - 1 - li;                       // (the table runs from the highest level down to level 0)
  chain_back_group<true>(v, L.stride[li], L.m[li], 0, lvl, L.two[li], (int)blockIdx.x - L.start[li], ds, dl);
}

// PATH (round 5): the whole back-substitution in ONE launch without any hand-over between workgroups.  A workgroup per bottom-level
// group; wavefront w works the group of level w that the bottom group hangs under (the last wavefront: the top level) -- the upper
// levels are recomputed by every workgroup below them (8 / 64 / 250 times at cfg3: a few hundred KB more of L2 reads) instead of being
// handed down through device-coherent stores and ready words (~5.5 us per level) or kernel boundaries.  All wavefronts request their
// group's data at once; t0 = z + Y delta_s of the group's frames is formed in the kernel (rows staged through LDS: coalesced loads,
// then one row per lane) -- the extra workgroups of the top level's launch and the ct0 round trip are gone with the launch; the steps
// travel top-down through LDS (positions of a group: 0 = left separator, 1 .. q = interior frames, q + 1 = right separator; the top
// level: its frames from position 0).  The dependent chain is chain_back_group's, instruction for instruction.
// Round 6: any border width and sharded passes.  Borders of more than kPathRowCols columns take t0 from k_chain_t0 (instance CT0: staging
// the rows of the recomputed levels in every bottom group then costs more than a launch -- measured 218 us against 71 for the level-by-level
// kernels at 6250 frames x D = 115, 120 us with t0 from its own launch); a pinned frame (separator / ghost of a sharded chain,
// DevView::pin_first / pin_last) steps with the reduced system's solution in the epilogue.
struct BackPath { int n; int stride[6]; int m[6]; int two[6]; int top_stride; int ldr; int dsw; int tail; };
// (round 6) the launch's LAST workgroup, when BackPath::tail is set, is not a bottom group: it forms what k_reduced left out
// (DevView::tail_deferred) -- the trial cameras and the shared parameters' terms of the step scalars -- beside the back-substitution
// instead of at the end of a one-workgroup kernel the whole chip waits for.  The first 256 threads; LDS: kTailLds doubles.
constexpr int kTailLds = kMaxCams * kCamStride + 16 + 8 + 6 * 256;
__device__ __forceinline__ void reduced_tail_workgroup(const DevView& v, double* lds) {
  const int tid = threadIdx.x, D = v.D;
  if (tid >= 256) return;
  const Ctrl* ct = v.ctrl;
  const int cur = ct->cur, done = ct->done;
  if (done) return;
  double* s_cam = lds;
  CamDesc* s_cd = reinterpret_cast<CamDesc*>(lds + kMaxCams * kCamStride);
  int* s_ipc = reinterpret_cast<int*>(lds + kMaxCams * kCamStride + 16);
  double* red = lds + kMaxCams * kCamStride + 16 + 8;
  for (int i = tid; i < v.n_cams * kCamStride; i += 256) s_cam[i] = v.cams[cur][i];
  if (tid < kMaxCams) s_cd[tid] = v.cd[tid];
  if (tid >= 64 && tid < 64 + 15) s_ipc[tid - 64] = v.imu_param_col[tid - 64];
  double pre_imu = 0.0;
  if (v.imu_on && (tid >> 6) == (D <= 64 ? 0 : 1) && (tid & 63) < 16) pre_imu = v.imus[cur][tid & 63];
  __syncthreads();
  reduced_tail(v, cur, v.delta_s, v.Sbuf + (size_t)D * D + 2 * D, v.slam, s_cam, s_cd, s_ipc, pre_imu, nullptr, red, false, false);
}
constexpr int kPathRowCols = 37;       // widest row [Y | z] the in-kernel staging serves (D + 1 entries)
constexpr int kPathRowLoads = 37;      // 63 rows x kPathRowCols entries over 64 lanes
constexpr int kPathDl = 96;            // doubles per level's step record (10 positions x 9)
template <int NWMAX, bool CT0>      // wavefronts per workgroup at most (levels + 1): up to four leave a whole SIMD's registers to each; CT0: t0 from k_chain_t0
#ifdef VC_BACK_PATH_EU      // (A/B builds, as VC_IMU_BLOCK_EU: n = 3: 168 registers, 212 B of scratch)
__global__ __launch_bounds__(64 * NWMAX) __attribute__((amdgpu_waves_per_eu(VC_BACK_PATH_EU, VC_BACK_PATH_EU))) void k_chain_back_path(DevView v, BackPath P) {
#else
__global__ __launch_bounds__(64 * NWMAX) void k_chain_back_path(DevView v, BackPath P) {
#endif
  extern __shared__ __attribute__((aligned(16))) double bp_lds[];
  if (P.tail && blockIdx.x == gridDim.x - 1) { reduced_tail_workgroup(v, bp_lds); return; }
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int nw = P.n + 1;
  double* DLall = bp_lds;                              // [nw][kPathDl]
  double* DSW = DLall + nw * kPathDl + (size_t)wv * P.dsw;      // this wavefront's copy of delta_s
  int* READY = reinterpret_cast<int*>(DLall + nw * kPathDl + nw * P.dsw);      // [8]
  double* RW = DLall + nw * kPathDl + nw * P.dsw + 4 + (size_t)wv * 63 * P.ldr;      // this wavefront's rows [63][ldr]
  double* DL = DLall + (size_t)wv * kPathDl;
  if (threadIdx.x < 8) READY[threadIdx.x] = 0;
  __syncthreads();
  auto wait_ready = [&](int slot) {
    while (__hip_atomic_load(&READY[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  auto set_ready = [&](int slot) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): this wavefront's LDS stores have been performed
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) __hip_atomic_store(&READY[slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  const int done = v.ctrl->done;
  const int N = v.n_frames, D = v.D, ldw = v.ldw, ldx = v.ldx;
  const size_t isz = (size_t)9 * ldx;
  const bool top = wv == P.n;
  const int lvl = top ? P.n : wv;
  const int s = top ? P.top_stride : P.stride[lvl], m = top ? kChainM : P.m[lvl], two = top ? 0 : P.two[lvl];
  const long gs = (long)m * s;
  const long a0 = (long)blockIdx.x * P.m[0] * P.stride[0];        // the bottom group's left separator
  const int a = top ? -1 : (int)((a0 / gs) * gs);
  const int first = top ? 0 : a + s;
  const int q = first < N ? (top ? (N - 1) / s + 1 : min(m - 1, (N - 1 - first) / s + 1)) : 0;
  const int r = first + q * s;
  const bool has_a = a >= 0 && a < N, has_r = !top && q > 0 && r < N;
  const int fi = lane / 9, k = lane % 9;
  const bool mine = fi < q;
  const int e = first + (mine ? fi : 0) * s;
  const bool two_sided = two && !top && q >= 1;
  const int fmid = (q - 1) / 2;
  // ---- requests: the rows [Y | z] of the group's frames (lane-strided, coalesced; narrow borders only -- CT0: t0 comes finished from
  // k_chain_t0), then this lane's blocks
  const int ncolr = D + 1, nel = q * 9 * ncolr;
  double rv[CT0 ? 1 : kPathRowLoads];
  if (!CT0) {
    const int step_r = 64 / ncolr, step_c = 64 - step_r * ncolr;
    int row = lane / ncolr, col = lane - row * ncolr;
#pragma unroll
    for (int u = 0; u < kPathRowLoads; ++u) {
      const int idx = lane + 64 * u;
      const int f2 = (row * 57) >> 9, k2 = row - 9 * f2;          // (row / 9 for rows below 64)
      const bool in = idx < nel;          // (a lane without an entry reads frame 0 and drops the value)
      const double* src = v.cW + (size_t)(in ? first + f2 * s : 0) * isz + (size_t)(in ? k2 : 0) * ldx + (in ? col : 0);
      const double x = *src;
      rv[u] = in ? x : 0.0;
      row += step_r; col += step_c;
      if (col >= ncolr) { col -= ncolr; ++row; }
    }
  }
  double dsv[3];
#pragma unroll
  for (int u = 0; u < (CT0 ? 3 : 1); ++u) { const int j = lane + 64 * u; dsv[u] = v.delta_s[j < D ? j : 0]; }
  double dinv = 1.0, Qrow[9], Lcol[9], Xs[9], t0_in = 0.0;
  {
    const double* img = v.cW + (size_t)(mine ? e : 0) * isz;
    const double* Wr = img + (size_t)k * ldx;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      const double qv = Wr[ldw + 18 + c], lv = img[c * ldx + ldw + 9 + k], xv = Wr[ldw + c];
      Qrow[c] = mine ? qv : 0.0; Lcol[c] = (mine && c > k) ? lv : 0.0; Xs[c] = (mine && a >= 0) ? xv : 0.0;
    }
    const double dg = Wr[ldw + 9 + k];
    dinv = mine ? 1.0 / dg : 1.0;
    if (CT0) { const double tv = v.ct0[(size_t)(mine ? e : 0) * 9 + k]; t0_in = mine ? tv : 0.0; }      // (the finished sum z + Y delta_s)
  }
  const int base = a, cnt = (a < N) ? q + 1 : 0;
  if (done) return;                      // (uniform over the workgroup)
#pragma unroll
  for (int u = 0; u < (CT0 ? 3 : 1); ++u) { const int j = lane + 64 * u; if (j < D) DSW[j] = dsv[u]; }
  double t = t0_in;
  if (!CT0) {
    int row = lane / ncolr, col = lane - row * ncolr;
    const int step_r = 64 / ncolr, step_c = 64 - step_r * ncolr;
#pragma unroll
    for (int u = 0; u < kPathRowLoads; ++u) {
      const int idx = lane + 64 * u;
      if (idx < nel) RW[row * P.ldr + col] = rv[u];
      row += step_r; col += step_c;
      if (col >= ncolr) { col -= ncolr; ++row; }
    }
    wave_lds_sync_local();
    if (mine) {
      const double* Rr = RW + (size_t)(fi * 9 + k) * P.ldr;
      double acc = Rr[D];
      for (int j = 0; j < D; ++j) acc += Rr[j] * DSW[j];
      t = acc;
    }
  }
  // ---- the separators' steps from the level above
  double da[9], dn[9], dr_in[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) { da[c] = 0.0; dn[c] = 0.0; }
  if (!top) {
    wait_ready(lvl + 1);
    const double* DLp = DLall + (size_t)(lvl + 1) * kPathDl;
    int pos;
    if (lvl + 1 == P.n) pos = a / P.top_stride;                               // the top level keeps its frames from position 0
    else { const long gp = (long)P.m[lvl + 1] * P.stride[lvl + 1]; const int ap = (int)((a0 / gp) * gp); pos = (a - ap) / P.stride[lvl + 1]; }
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      const double x0 = DLp[pos * 9 + c], x1 = DLp[(pos + 1) * 9 + c];
      da[c] = has_a ? x0 : 0.0; dn[c] = has_r ? x1 : 0.0;
    }
  }
#pragma unroll
  for (int c = 0; c < 9; ++c) dr_in[c] = dn[c];
  if (mine && a >= 0) {
    const bool right = two_sided && fi > fmid;
#pragma unroll
    for (int c = 0; c < 9; ++c) t += Xs[c] * (right ? dn[c] : da[c]);       // (dn: still the right separator's step here)
  }
  double my = 0.0;
  if (two_sided) {
    {
      double y = t;
#pragma unroll
      for (int c = 0; c < 9; ++c) y += Qrow[c] * dn[c];
#pragma unroll
      for (int j = 8; j >= 0; --j) {
        const double xj = readlane_f64(y * dinv, fmid * 9 + j);
        dn[j] = -xj;
        y -= Lcol[j] * xj;
      }
      if (fi == fmid) {
#pragma unroll
        for (int j = 0; j < 9; ++j) my = (j == k) ? dn[j] : my;
      }
    }
    double dl_[9], dr_[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) { dl_[j] = dn[j]; dr_[j] = dn[j]; }
    const int nleft = fmid, nright = q - 1 - fmid;
    for (int st = 1; st <= max(nleft, nright); ++st) {
      const bool hl = st <= nleft, hr = st <= nright;              // wave-uniform
      const int il = hl ? fmid - st : 0, ir = hr ? fmid + st : 0;
      double yl = t, yr = t;
#pragma unroll
      for (int c = 0; c < 9; ++c) { yl += Qrow[c] * dl_[c]; yr += Qrow[c] * dr_[c]; }
#pragma unroll
      for (int j = 8; j >= 0; --j) {
        const double xl = readlane_f64(yl * dinv, il * 9 + j), xr = readlane_f64(yr * dinv, ir * 9 + j);
        if (hl) { dl_[j] = -xl; yl -= Lcol[j] * xl; }
        if (hr) { dr_[j] = -xr; yr -= Lcol[j] * xr; }
      }
      if (hl && fi == il) {
#pragma unroll
        for (int j = 0; j < 9; ++j) my = (j == k) ? dl_[j] : my;
      }
      if (hr && fi == ir) {
#pragma unroll
        for (int j = 0; j < 9; ++j) my = (j == k) ? dr_[j] : my;
      }
    }
  } else
  for (int i = q - 1; i >= 0; --i) {
    double y = t;
#pragma unroll
    for (int c = 0; c < 9; ++c) y += Qrow[c] * dn[c];
#pragma unroll
    for (int j = 8; j >= 0; --j) {
      const double xj = readlane_f64(y * dinv, i * 9 + j);
      dn[j] = -xj;
      y -= Lcol[j] * xj;
    }
    if (fi == i) {
#pragma unroll
      for (int j = 0; j < 9; ++j) my = (j == k) ? dn[j] : my;
    }
  }
  // ---- this level's steps for the level below (the bottom level: for its own epilogue)
  if (mine) DL[(fi + (top ? 0 : 1)) * 9 + k] = my;
  if (!top && lane < 9) { DL[lane] = da[lane]; DL[(q + 1) * 9 + lane] = dr_in[lane]; }
  if (lvl != 0 || top) { set_ready(lvl); return; }
  wave_lds_sync_local();
  // ---- level 0: trial poses / velocities of the group's frames and their step terms (chain_back_group's epilogue)
  double gd = 0, dld = 0, step2 = 0, x2 = 0, g2 = 0, gmax = 0;
  if (lane < cnt) {
    const int f = base + lane;
    const int cur = v.ctrl->cur;
    double d[9], Tin[7], vin[3], gi9[9], lam9[9];
    {
      const double* pin = v.poses[cur] + (size_t)f * kPoseStride;
      const double* vi = v.vel[cur] + (size_t)f * 4;
#pragma unroll
      for (int i = 0; i < 7; ++i) Tin[i] = pin[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) vin[i] = vi[i];
#pragma unroll
      for (int i = 0; i < 9; ++i) { gi9[i] = v.cg[(size_t)f * 9 + i]; lam9[i] = v.clam[(size_t)f * 9 + i]; }
    }
    // pinned frames (sharded chain) step with the reduced system's solution; their gradient / damping terms are counted there, and only
    // the owner (not the rank that holds the ghost copy) counts the step and parameter norms -- chain_back_group's rule
    const bool pin_f = (f == 0 && v.pin_first), pin_l = (f == N - 1 && v.pin_last);
    const int sc = pin_f ? v.sep_col0 : (pin_l ? v.sep_col1 : 0);
#pragma unroll
    for (int i = 0; i < 9; ++i) { const double dch = DL[lane * 9 + i], dsp = DSW[sc + i]; d[i] = (pin_f || pin_l) ? dsp : dch; }
    double* pout = v.poses[1 - cur] + (size_t)f * kPoseStride;
    double Tout[7], dd[6];
    for (int i = 0; i < 6; ++i) dd[i] = d[i];
    se3_plus(Tin, dd, Tout);
    for (int i = 0; i < 7; ++i) { pout[i] = Tout[i]; const double ee = Tout[i] - Tin[i]; step2 += ee * ee; x2 += Tin[i] * Tin[i]; }
    pout[7] = 0.0;
    double* vout = v.vel[1 - cur] + (size_t)f * 4;
    for (int i = 0; i < 3; ++i) { const double dv = d[6 + i]; vout[i] = vin[i] + dv; step2 += dv * dv; x2 += vin[i] * vin[i]; }
    vout[3] = 0.0;
    for (int i = 0; i < 9; ++i) {
      const double gi = gi9[i];
      gd += gi * d[i]; dld += lam9[i] * d[i] * d[i]; g2 += gi * gi; gmax = fmax(gmax, fabs(gi));
    }
    if (pin_f || pin_l) { gd = 0; dld = 0; g2 = 0; gmax = 0; }
    if (pin_l) { step2 = 0; x2 = 0; }
  }
  gd = wave_sum(gd); dld = wave_sum(dld); step2 = wave_sum(step2); x2 = wave_sum(x2); g2 = wave_sum(g2);
#pragma unroll
  for (int o2 = 32; o2 > 0; o2 >>= 1) gmax = fmax(gmax, __shfl_down(gmax, o2, 64));
  if (lane == 0) {
    double* o = v.grp_part + (size_t)blockIdx.x * kNumScal;
    o[kScGd] = gd; o[kScDld] = dld; o[kScStep2] = step2; o[kScX2] = x2; o[kScG2] = g2; o[kScCost] = 0.0; o[kScGmax] = gmax; o[kScSq] = 0.0;
  }
}

// sum over all frames of [Y | z]^T [Y | z]: part[chunk] = [ D x D | D ]  (same layout as the vision path)
constexpr int kMaxPairsPerWaveI = 9;
// phase stamps of one workgroup of k_chain_gram (profiling builds only: -DVC_GRAM_STAMPS, tools/gram_stamps.py)
#ifdef VC_GRAM_STAMPS
#define GSTAMP(i) do { if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0 && (i) < 16) v.dbg[i] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GSTAMP(i) do { } while (0)
#endif
// NQ: column-tile pairs per wavefront and batch (accumulators with compile-time indices: the version that picked its accumulator
// with a runtime index spent ~150 v_cndmask per pair on it and ran the nine MFMAs of a pair as one dependent chain -- 7.8 us per
// group of four frames at D = 115; here the NQ chains advance side by side).  Rows of the next group are requested before the
// current group's MFMAs and land in registers meanwhile.
// NL: entries of the 36 x ld row image per thread (36 ld / 256, rounded up).
// mask_stride > 0: frames with index 0 (mod mask_stride) contribute nothing (the chain's top level: still being eliminated while this
// runs beside it; k_reduced adds them from their finished images, see DevView::gram_top_stride)
// gather_stride > 0: the chunk is the top level itself -- frames 0, gather_stride, 2 gather_stride, ... (at most 7), summed into the partial
// record behind the dense chunks' (a launch of its own after the top level, where k_reduced does not add these frames itself)
template <int NQ, int NL>
// publish: the record goes out as device-coherent stores and the chunk's ready word follows them (DevView::part_ride: the partial sums ride in the same launch)
__device__ __forceinline__ void chain_gram_chunk(const DevView& v, int chunk, double* R /* 36 x ld */, unsigned short* s_pair /* 128 */, int mask_stride, int gather_stride = 0, bool publish = false) {
  const Ctrl* ct = v.ctrl;
  if (ct->done) return;
  GSTAMP(0);
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int D = v.D, ld = v.ldw, N = v.n_frames;
  const int nT = (D + 1 + 15) / 16, nPairs = nT * (nT + 1) / 2;
  const int fstep = gather_stride > 0 ? gather_stride : 1;      // frame slot q of a group is frame (fg + q) fstep
  const int f0 = gather_stride > 0 ? 0 : chunk * v.chunk_frames, f1 = gather_stride > 0 ? (N - 1) / gather_stride + 1 : min(f0 + v.chunk_frames, N);
  double* part = v.part + (size_t)chunk * v.part_stride;
  if (tid < nPairs) {
    int I = 0, J = 0;
    for (int k = 0; k < tid; ++k) if (++J == nT) { ++I; J = I; }
    s_pair[tid] = (unsigned short)(I | (J << 8));
  }
  // this thread's entries of the row image: i = tid + 256 u -> (row, col) by stepping (no division per entry)
  const int step_r = 256 / ld, step_c = 256 - step_r * ld, row0 = tid / ld, col0 = tid - row0 * ld;
  [[maybe_unused]] int gs_ = 1;
  // column-tile pairs are processed in batches of 4 NQ (NQ accumulators per wavefront); more pairs than that re-read the chunk's
  // rows once per batch
  for (int pb = 0; pb < nPairs; pb += 4 * NQ) {
    v4d acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = (v4d){0.0, 0.0, 0.0, 0.0};
    double tmp[NL];
    auto request = [&](int fg) {
      const int nrow = min(4, f1 - fg) * 9;
      const double* src = v.cW + (size_t)fg * fstep * 9 * v.ldx;
      const unsigned skip = (unsigned)(fstep - 1) * 9u * (unsigned)v.ldx;      // extra offset per frame slot (0 for consecutive frames)
      unsigned mbits = 0u;                  // bit q: frame fg + q is masked (wave-uniform)
      if (mask_stride > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) mbits |= ((fg + q) % mask_stride == 0) ? (1u << q) : 0u;
      }
      int row = row0, col = col0;
#pragma unroll
      for (int u = 0; u < NL; ++u) {
        const unsigned qf = (unsigned)(row * 57) >> 9;               // (row / 9 for rows below 36)
        const bool masked = (mbits >> qf) & 1u;
        tmp[u] = (row < nrow && !masked) ? src[(unsigned)(row * v.ldx + col) + qf * skip] : 0.0;      // (uniform base + 32-bit offset: one register per entry if hoisted)
        row += step_r; col += step_c;
        if (col >= ld) { col -= ld; ++row; }
      }
    };
    request(f0);
    for (int fg = f0; fg < f1; fg += 4) {
#pragma unroll
      for (int u = 0; u < NL; ++u) { const int i = tid + 256 * u; if (i < 36 * ld) R[i] = tmp[u]; }
      __syncthreads();
      GSTAMP(gs_); ++gs_;
      if (fg + 4 < f1) request(fg + 4);
      const double* rq = R + (lane >> 4) * ld + (lane & 15);
      int oa[NQ], ob[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int p = min(pb + 4 * q + wave, nPairs - 1);      // (past the end: recomputes the last pair, never written)
        const int ij = s_pair[p];
        oa[q] = (ij & 255) * 16; ob[q] = (ij >> 8) * 16;
      }
#pragma unroll
      for (int ks = 0; ks < 9; ++ks)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(rq[ks * 4 * ld + oa[q]], rq[ks * 4 * ld + ob[q]], acc[q], 0, 0, 0);
      __syncthreads();
      GSTAMP(gs_); ++gs_;
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int p = pb + 4 * q + wave;
      if (p < nPairs) {
        const int ij = s_pair[p], I = ij & 255, J = ij >> 8;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = I * 16 + (lane >> 4) + 4 * g, col = J * 16 + (lane & 15);
          if (row < D && col <= D) {
            double* dst = col < D ? part + row * D + col : part + D * D + row;
            if (publish) __hip_atomic_store(dst, acc[q][g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *dst = acc[q][g];
          }
        }
      }
    }
  }
  if (publish) {
    __builtin_amdgcn_s_waitcnt(0);          // this wavefront's stores have been performed
    __syncthreads();
    if (tid == 0) __hip_atomic_store(v.part_ready + chunk, v.pass_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#ifdef VC_GRAM_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
  GSTAMP(15);
#endif
}
// gather = 0: one workgroup per chunk of consecutive frames (the top level's frames masked out where DevView::gram_top_stride is set);
// gather = 1: one workgroup, the top level's frames into the partial record behind the dense chunks'
template <int NQ, int NL>
__global__ __launch_bounds__(256, 2) void k_chain_gram(DevView v, int gather) {
  extern __shared__ __attribute__((aligned(16))) double R[];    // 36 x ld
  __shared__ unsigned short s_pair[128];                         // pair p -> I | J << 8
  if (gather) chain_gram_chunk<NQ, NL>(v, v.n_chunks, R, s_pair, 0, v.gram_top_stride);
  else chain_gram_chunk<NQ, NL>(v, (int)blockIdx.x, R, s_pair, v.gram_top_stride);
}
// The chain's top level (one group of NW wavefronts side by side, 4-7 dependent eliminations: ~13 us at BASELINE cfg3, 25-50 us with wide
// borders, the chip idle beside it) and, in the same launch, the Gram sums of every frame below it -- final since the launch before:
// workgroup 0 eliminates, workgroups 1 .. n_chunks are k_chain_gram's with the top level's frames masked out (DevView::gram_top_stride).
template <int NW, int NQ, int NL>
__global__ __launch_bounds__(256) void k_chain_top_gram(DevView v, int s, int m, int lvl) {
  extern __shared__ __attribute__((aligned(16))) double R[];    // 36 x ld
  __shared__ unsigned short s_pair[128];
  __shared__ __attribute__((aligned(16))) double XS[9 * kXsLd];
  __shared__ double An[81];
  __shared__ double Ls_all[NW * 81];
  if (blockIdx.x == 0) {
    if ((int)threadIdx.x >= 64 * NW) return;      // (a barrier counts the wavefronts that are still there)
    chain_fwd_group<1, NW>(v, s, m, 1, lvl, 0, (int)(threadIdx.x >> 6), XS, An, Ls_all);
    return;
  }
  // (round 6) the workgroup behind the chunks: the camera blocks, the IMU block and the chunk costs of the reduced system, formed here -- beside the
  // top level's dependent eliminations -- instead of inside k_reduced (vc_shared_blocks.hpp; DevView::hadd_early)
  if ((int)blockIdx.x == 1 + v.n_chunks) { hadd_side_job(v, R, 256); return; }
  // ... and behind that one (flag hand-overs, DevView::part_ride): the fixed-order sums of the chunk records, k_part_sum's work without its launch
  if ((int)blockIdx.x > 1 + v.n_chunks) { part_sum_ride_job(v, (int)blockIdx.x - 2 - v.n_chunks, R); return; }
  chain_gram_chunk<NQ, NL>(v, (int)blockIdx.x - 1, R, s_pair, v.gram_top_stride, 0, v.part_ride != 0);
}

// ------------------------------------------------------------------------------------------ launchers
void launch_imu_delta(const DevView& v, hipStream_t s, int trial) {
  if (v.n_frames < 2) return;
  hipLaunchKernelGGL(k_imu_block, dim3((v.n_frames - 1 + 3) / 4), dim3(256), 0, s, v, trial);      // one wavefront per IMU block
}
void launch_imu_jac(const DevView& v, int wr, hipStream_t s, int trial) {
  if (v.n_frames < 2) return;
  hipLaunchKernelGGL(k_imu_jac, dim3((v.n_frames - 1 + 7) / 8), dim3(256), 0, s, v, wr, trial);      // 8 blocks per workgroup
}
void launch_imu_weights(const DevView& v, int wr, hipStream_t s) {
  if (v.n_frames < 2) return;
  hipLaunchKernelGGL(k_imu_weights, dim3((v.n_frames - 1 + 3) / 4), dim3(64), 0, s, v, wr);      // four blocks per wavefront
}
// Level schedule of the partitioned chain elimination: strides 1, m, m^2, ... while more than m - 1 frames are active, then
// the top level (one wavefront eliminates the rest).  forward: bottom-up; backward: top-down.
int chain_group_size() {
  static int m = 0;
  if (!m) { const char* e = std::getenv("VICALIB_AMD_CHAIN_M"); m = e ? std::max(2, std::min(kChainM, std::atoi(e))) : kChainM; }
  return m;
}
// group size above the bottom level (test hook VICALIB_AMD_CHAIN_M_UPPER; default: the same as the bottom level)
int chain_group_size_upper() {
  static int m = 0;
  if (!m) { const char* e = std::getenv("VICALIB_AMD_CHAIN_M_UPPER"); m = e ? std::max(2, std::min(kChainM, std::atoi(e))) : chain_group_size(); }
  return m;
}
void launch_chain_gram(const DevView& v, hipStream_t s);
// Level schedule of the partitioned chain elimination: strides 1, m0, m0 m1, ... while more than m - 1 frames are active, then
// the top level (one wavefront eliminates the rest).  forward: bottom-up; backward: top-down.
static void chain_levels(const DevView& v, hipStream_t s, bool forward) {
  const int N = v.n_frames;
  if (N < 1) return;
  int strides[32], ms[32], nl = 0;
  long st = 1;
  while (true) {
    const int m = nl == 0 ? chain_group_size() : chain_group_size_upper();
    if (!((N - 1) / st + 1 > m - 1)) break;
    strides[nl] = (int)st; ms[nl] = m; ++nl; st *= m;
  }
  const int top_stride = (int)st, m_top = kChainM;      // the top level eliminates whatever is left (fewer than a group)
  const int cpl = (v.D + 1 + 27 + 63) / 64;
  // wide borders: wavefronts side by side (measured in round 2 against several columns per lane: 1.68 -> 1.51 ms per pass at cfg5's per-rank
  // size, 8.74 -> 8.39 ms at its full size; the columns-per-lane instances -- 256 registers + up to 256 accumulation registers, scratch at
  // four columns -- were kept for A/B runs until round 6 and are gone)
  // narrow borders: above the bottom level the groups are eliminated from both ends (k_chain_fwd2: 4 dependent eliminations per
  // level instead of 7; measured 26.3 -> 21.7 us per level at cfg3).  Not at the bottom level: its 250 groups would need 500
  // wavefronts of ~370 registers next to the weight update's 500 on the other stream, and queue behind them (42 vs 32 us).
  // VICALIB_AMD_CHAIN_TWO=0: one-sided throughout
  static const bool two_env = [] { const char* e = std::getenv("VICALIB_AMD_CHAIN_TWO"); return !(e && std::atoi(e) == 0); }();
  const bool two_sided = two_env && cpl <= 2;
  // ... and at the bottom level too once the weight update on the other stream starts behind it (vc_pass.cpp: enqueue_pass;
  // VICALIB_AMD_CHAIN_TWO_BOTTOM=0: one-sided bottom level)
  static const bool two_bottom = [] { const char* e = std::getenv("VICALIB_AMD_CHAIN_TWO_BOTTOM"); return !(e && std::atoi(e) == 0); }();
  // (whatever the hand-over mode: a solve resumed with events after a flag time-out must repeat the withheld passes with the same
  //  arithmetic -- with events the weight update runs beside the bottom level and the pass is ~4 us slower than one-sided would be)
  const int two_from = two_bottom ? 0 : 1;
  // (two columns per lane's worth of border, D <= 100: two wavefronts per sweep, a whole CU per group -- on levels whose groups the chip
  //  holds at once; a function of the frame count and the level only: forward, backward and every hand-over mode agree on it)
  auto two_at = [&](int l) {
    if (!(two_sided && ms[l] >= 4 && l >= two_from)) return false;
    if (cpl <= 1) return true;
    const long groups = ((long)N - 1) / ((long)strides[l] * ms[l]) + 1;
    return groups <= 256;
  };
  auto fwd = [&](int groups, int stride, int m, int top, int lvl) {
    if (!top && two_at(lvl)) {
      // (hadd_early: the first launch above the bottom level carries the sums of the chunk records' entries behind S and g_red)
      const int extra = (v.hadd_early && lvl == 1) ? (v.part_stride - (v.D * v.D + v.D) + 15) / 16 : 0;
      if (cpl <= 1) hipLaunchKernelGGL(k_chain_fwd2<1>, dim3(groups + extra), dim3(128), 0, s, v, stride, m, lvl, groups);
      else hipLaunchKernelGGL(k_chain_fwd2<2>, dim3(groups + extra), dim3(256), 0, s, v, stride, m, lvl, groups);
    }
    else if (cpl <= 1) hipLaunchKernelGGL((k_chain_fwd<1, 1>), dim3(groups), dim3(64), 0, s, v, stride, m, top, lvl);
    else if (cpl <= 2) hipLaunchKernelGGL((k_chain_fwd<1, 2>), dim3(groups), dim3(128), 0, s, v, stride, m, top, lvl);
    else if (cpl <= 3) hipLaunchKernelGGL((k_chain_fwd<1, 3>), dim3(groups), dim3(192), 0, s, v, stride, m, top, lvl);
    else hipLaunchKernelGGL((k_chain_fwd<1, 4>), dim3(groups), dim3(256), 0, s, v, stride, m, top, lvl);
  };
  if (forward) {
    for (int l = 0; l < nl; ++l) {
      const int groups = (int)(((long)N - 1) / ((long)strides[l] * ms[l]) + 1);
      if (l == 0 && v.fold_l0) {      // the chain assembly rides in the bottom level's launch (k_chain_l0)
        if (v.n_cams <= 1) hipLaunchKernelGGL(k_chain_l0<1>, dim3(groups), dim3(256), 0, s, v);
        else hipLaunchKernelGGL(k_chain_l0<2>, dim3(groups), dim3(256), 0, s, v);
      } else fwd(groups, strides[l], ms[l], 0, l);
    }
    if (v.gram_top_stride > 0) {
      // early Gram: the top level's one group and the Gram sums of all frames below it in one launch
      const size_t lds = std::max((size_t)36 * v.ldw, (size_t)(v.hadd_early ? kHaddLds : 0)) * sizeof(double);      // (>= 512 doubles: part_sum_ride_job)
      const int ride_blocks = (v.part_ride && v.hadd_early) ? (v.D * v.D + v.D + 15) / 16 : 0;
      const int nT = (v.D + 1 + 15) / 16, nPairs = nT * (nT + 1) / 2, nq = std::min(kMaxPairsPerWaveI, (nPairs + 3) / 4);
      const int nlr = (36 * v.ldw + 255) / 256;
      auto go = [&](auto kern) {
        if (lds > 40000) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(1 + v.n_chunks + (v.hadd_early ? 1 : 0) + ride_blocks), dim3(256), lds, s, v, top_stride, m_top, nl);
      };
      // (wavefronts of the top group by the border's width, Gram instance by the row image's size: the pairs that occur)
      if (cpl <= 1) { if (nq <= 1) go(k_chain_top_gram<1, 1, 7>); else go(k_chain_top_gram<1, 2, 7>); }
      else if (cpl <= 2) { if (nlr <= 12) go(k_chain_top_gram<2, 4, 12>); else if (nq <= 6) go(k_chain_top_gram<2, 6, 16>); else go(k_chain_top_gram<2, 9, 16>); }
      else if (cpl <= 3) { if (nlr <= 16) go(k_chain_top_gram<3, 9, 16>); else if (nlr <= 21) go(k_chain_top_gram<3, 9, 21>); else go(k_chain_top_gram<3, 9, 25>); }
      else { if (nlr <= 25) go(k_chain_top_gram<4, 9, 25>); else go(k_chain_top_gram<4, 9, 30>); }
    } else {
      fwd(1, top_stride, m_top, 1, nl);
    }
  } else {
    // the whole back-substitution as one launch without hand-overs (k_chain_back_path): any border width, sharded passes included
    if (v.back_path && nl >= 1 && nl <= 5) {
      BackPath P; P.n = nl;
      for (int l = 0; l < 6; ++l) { P.stride[l] = l < nl ? strides[l] : 1; P.m[l] = l < nl ? ms[l] : 2; P.two[l] = (l < nl && two_at(l)) ? 1 : 0; }
      const bool ct0 = v.D + 1 > kPathRowCols;
      P.top_stride = top_stride; P.ldr = ct0 ? 1 : ((v.D + 1) | 1); P.dsw = ((v.D + 63) / 64) * 64;
      if (ct0) hipLaunchKernelGGL(k_chain_t0, dim3((N + 3) / 4), dim3(256), 0, s, v);
      const int groups0 = (int)(((long)N - 1) / ((long)strides[0] * ms[0]) + 1), nw = nl + 1;
      P.tail = (v.tail_deferred && nw >= 4) ? 1 : 0;
      const size_t lds = std::max(((size_t)nw * kPathDl + (size_t)nw * P.dsw + 4 + (ct0 ? 0 : (size_t)nw * 63 * P.ldr)), (size_t)(P.tail ? kTailLds : 0)) * sizeof(double);
      static LdsGrant g4, g6;
      auto go = [&](auto kern, LdsGrant& g) {
        if (lds > 60000 && g.need(lds)) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(groups0 + P.tail), dim3(64 * nw), lds, s, v, P);
      };
      if (nw <= 4) { if (ct0) go(k_chain_back_path<4, true>, g4); else go(k_chain_back_path<4, false>, g4); }
      else { if (ct0) go(k_chain_back_path<6, true>, g6); else go(k_chain_back_path<6, false>, g6); }
      return;
    }
    hipLaunchKernelGGL(k_chain_back, dim3(1 + (nl > 0 ? (N + kBackT0Frames - 1) / kBackT0Frames : 0)), dim3(64), 0, s, v, top_stride, m_top, 1, nl, 0);
    // the levels below: one launch (k_chain_back_levels: ready words instead of kernel boundaries); VICALIB_AMD_BACK_FUSED=0: one launch per level
    static const bool fused = [] { const char* e = std::getenv("VICALIB_AMD_BACK_FUSED"); return !(e && std::atoi(e) == 0); }();
    // (only while every workgroup of the launch can be resident at once -- 119 registers, 4 wavefronts per SIMD, 4096 on the chip; half of
    //  that here: a group that waits for its separators then never keeps a producer from starting, whatever order the dispatcher picks)
    int total_groups = 0;
    for (int l = 0; l < nl; ++l) total_groups += (int)(((long)N - 1) / ((long)strides[l] * ms[l]) + 1);
    // (half of what the device can hold of this kernel -- asked once, not a constant of one chip: the other half is left to whatever the
    //  second stream runs beside it)
    static const int resident_limit = [] {
      int dev = 0, per_cu = 0; hipDeviceProp_t pr;
      if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess ||
          hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_chain_back_levels, 64, 0) != hipSuccess || per_cu <= 0) return 2048;
      return std::max(64, per_cu * pr.multiProcessorCount / 2);
    }();
    if (fused && nl > 0 && nl <= 8 && v.cready && total_groups <= resident_limit) {
      BackLevels L; L.n = nl;
      int at = 0;
      for (int i = 0; i < nl; ++i) {
        const int l = nl - 1 - i;
        L.start[i] = at; L.stride[i] = strides[l]; L.m[i] = ms[l]; L.two[i] = two_at(l) ? 1 : 0;
        at += (int)(((long)N - 1) / ((long)strides[l] * ms[l]) + 1);
      }
      for (int i = nl; i < 8; ++i) { L.start[i] = 1 << 30; L.stride[i] = 1; L.m[i] = 2; L.two[i] = 0; }
      hipLaunchKernelGGL(k_chain_back_levels, dim3(at), dim3(64), 0, s, v, L);
    } else
    for (int l = nl - 1; l >= 0; --l)
      hipLaunchKernelGGL(k_chain_back, dim3((int)(((long)N - 1) / ((long)strides[l] * ms[l]) + 1)), dim3(64), 0, s, v, strides[l], ms[l], 0, l,
                         two_at(l) ? 1 : 0);
  }
}
// k_chain_l0 (the chain assembly folded into the bottom level) serves narrow borders, at most two cameras, groups of 8 eliminated from
// both ends -- and at least one level below the top one
bool chain_fold_supported(int n_frames, int D, int n_cams) {
  static const bool two_env = [] { const char* e = std::getenv("VICALIB_AMD_CHAIN_TWO"); return !(e && std::atoi(e) == 0); }();
  static const bool two_bottom = [] { const char* e = std::getenv("VICALIB_AMD_CHAIN_TWO_BOTTOM"); return !(e && std::atoi(e) == 0); }();
  return two_env && two_bottom && chain_group_size() == kChainM && D + 1 + 27 <= 64 && n_cams <= 2 && (n_frames - 1) + 1 > kChainM - 1;
}
// The forward elimination has the two launches above the bottom level that carry the side jobs of DevView::hadd (vc_shared_blocks.hpp): a
// two-sided level 1 (k_chain_fwd2: the sums of the chunk records' entries behind S and g_red) and the top level with the early Gram sums
// (k_chain_top_gram: the record itself).  Mirrors chain_levels.
bool chain_hadd_early(const DevView& v) {
  if (!v.imu_on || !v.hadd || v.gram_top_stride <= 0 || v.n_frames < 1) return false;
  static const bool two_env = [] { const char* e = std::getenv("VICALIB_AMD_CHAIN_TWO"); return !(e && std::atoi(e) == 0); }();
  const int cpl = (v.D + 1 + 27 + 63) / 64;
  if (!two_env || cpl > 2 || chain_group_size_upper() < 4) return false;
  int nl = 0; long st = 1, st1 = 1;
  while (true) {
    const int m = nl == 0 ? chain_group_size() : chain_group_size_upper();
    if (!((v.n_frames - 1) / st + 1 > m - 1)) break;
    if (nl == 1) st1 = st;
    ++nl; st *= m;
  }
  if (nl < 2) return false;
  const long groups1 = ((long)v.n_frames - 1) / (st1 * chain_group_size_upper()) + 1;
  return cpl <= 1 || groups1 <= 256;      // (two_at(1))
}
// the back-substitution is one launch of k_chain_back_path (chain_levels, backward) whose workgroups have at least 256 threads
bool chain_back_is_path(const DevView& v) {
  if (!v.back_path || v.n_frames < 1) return false;
  int nl = 0; long st = 1;
  while (true) {
    const int m = nl == 0 ? chain_group_size() : chain_group_size_upper();
    if (!((v.n_frames - 1) / st + 1 > m - 1)) break;
    ++nl; st *= m;
  }
  return nl >= 3 && nl <= 5;
}
// stride of the frames the top level eliminates (1: no level below it)
int chain_top_stride(int n_frames) {
  if (n_frames < 1) return 1;
  int nl = 0; long st = 1;
  while (true) {
    const int m = nl == 0 ? chain_group_size() : chain_group_size_upper();
    if (!((n_frames - 1) / st + 1 > m - 1)) break;
    ++nl; st *= m;
  }
  return (int)st;
}
// launches of the forward elimination (levels + the top level)
int chain_forward_launches(const DevView& v) {
  const int N = v.n_frames;
  if (N < 1) return 0;
  int nl = 0; long st = 1;
  while (true) {
    const int m = nl == 0 ? chain_group_size() : chain_group_size_upper();
    if (!((N - 1) / st + 1 > m - 1)) break;
    ++nl; st *= m;
  }
  return nl + 1;
}
void launch_chain_init(const DevView& v, hipStream_t s) {
  const size_t slot = (size_t)v.n_cams * kGStride + kGStride;
  const size_t lds = std::max((size_t)4 * (v.n_cams * kGStride + kInitPad), 4 * slot) * sizeof(double);
  auto go = [&](auto kern) {
    if (lds > 65536) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(v.n_chunks), dim3(256), lds, s, v);
  };
  if (v.n_cams <= 1) go(k_chain_init<1>); else if (v.n_cams <= 2) go(k_chain_init<2>); else if (v.n_cams <= 4) go(k_chain_init<4>); else go(k_chain_init<8>);
}
void launch_chain_fwd(const DevView& v, hipStream_t s) { chain_levels(v, s, true); }
static void chain_gram_go(const DevView& v, hipStream_t s, int gather) {
  const size_t lds = (size_t)36 * v.ldw * sizeof(double);
  const int nT = (v.D + 1 + 15) / 16, nPairs = nT * (nT + 1) / 2, nq = std::min(kMaxPairsPerWaveI, (nPairs + 3) / 4);
  auto go = [&](auto kern) {
    if (lds > 65536) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(gather ? 1 : v.n_chunks), dim3(256), lds, s, v, gather);
  };
  // (pairs per wavefront, row-image entries per thread) by the number of column tiles: ld = 48, 80, 112, 144, 176, 208
  const int nl = (36 * v.ldw + 255) / 256;
  if (nl <= 7) { if (nq <= 1) go(k_chain_gram<1, 7>); else go(k_chain_gram<2, 7>); }
  else if (nl <= 12) go(k_chain_gram<4, 12>);
  else if (nl <= 16) { if (nq <= 6) go(k_chain_gram<6, 16>); else go(k_chain_gram<9, 16>); }
  else if (nl <= 21) go(k_chain_gram<9, 21>);
  else if (nl <= 25) go(k_chain_gram<9, 25>);
  else go(k_chain_gram<9, 30>);
}
void launch_chain_gram(const DevView& v, hipStream_t s) { chain_gram_go(v, s, 0); }
// early Gram where k_reduced does not add the top level's frames itself (D > kEarlyTopD, sharded passes): their sums as one more partial record
void launch_chain_gram_top(const DevView& v, hipStream_t s) { chain_gram_go(v, s, 1); }
void launch_chain_solve_b(const DevView& v, hipStream_t s) { chain_levels(v, s, false); }

}  // namespace vc
