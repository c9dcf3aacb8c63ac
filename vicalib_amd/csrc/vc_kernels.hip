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
//  k_frame_schur     one wavefront per frame: H_pp, g_p from the tile Gram blocks, damping, 6x6 Cholesky,
//                    Y = L^-1 W per tile (lanes = columns); per chunk sum Y^T Y, sum Y^T z on the matrix pipe
//  k_part_sum        fixed-order sum of the chunk partials, spread over many CUs
//  k_reduced         one workgroup: camera blocks -> packed reduced system, damped Cholesky, delta_s, trial
//                    state of the shared parameters
//  k_trial           one wavefront per tile: back-substitution, T <- T exp(delta), trial residual sweep
//  k_final           fixed-order reduction of the step scalars + the accept/reject decision (device Ctrl)
#include <hip/hip_runtime.h>
#include <algorithm>
#include "vc_math.hpp"
#include "vc_device.h"
#include "vc_kutil.hpp"

namespace vc {

__device__ void merged_control(const DevView& v, Ctrl* out, double* red, bool writer);

// ------------------------------------------------------------------------------------------ Jacobian sweep
// One wavefront sweeps one tile.  Per pass of 64 corners: lane = corner computes its two unique-column rows into the
// wave-private LDS image; the Gram block G += u^T u then accumulates on the matrix pipe, four rows (two corners) per step.
//
// The step uses v_mfma_f64_4x4x4_f64 (four independent 4x4x4 blocks per instruction), not the 16x16x4 shape.  Measured on
// MI355X (tools/probe/mfma_f64_probe.hip): the 16x16x4 f64 MFMA has ~300 cycles latency and sustains 49 TF/s, the 4x4x4
// one ~40 cycles and 75 TF/s; f64 MFMA and f64 VALU work do NOT overlap (one shared DP datapath), so what counts is the
// total number of f64 operations -- and the symmetric 16x16 block only needs its upper 4x4 sub-blocks.
// Lane layout of the 4x4x4 instruction (tools/probe/mfma_layout_probe.hip): lane = 16 k + 4 b + i for A_b[i][k],
// 16 k + 4 b + j for B_b[k][j], 16 i + 4 b + j for D_b[i][j].  With k = row of the 4-row group and column
// 4 X_b + (lane & 3) read from that row, block b of an instruction accumulates the sub-block (I_b, J_b) of G:
//   all models : A = pattern (0,1,2,3) for every instruction (one LDS read, also the B operand of the diagonal blocks)
//   <= 12 used columns (fov, linear; 3 block columns): 2 instructions -- diagonal (0,0)(1,1)(2,2)(3,3) and, with
//                A = (0,0,1,3), B = (1,2,2,3): (0,1)(0,2)(1,2)(3,3)           [3 LDS reads per group]
//   otherwise  : 3 instructions -- B = A rotated by 0, 1, 2 blocks: diagonal, (0,1)(1,2)(2,3)(3,0), (0,2)(1,3)(2,0)(3,1)
#ifndef VC_JAC_SPLIT_ROWS
#define VC_JAC_SPLIT_ROWS 1      // poly2 / poly3: the two row types' Gram blocks apart (jac_tile_body_split below; 0: the four-block-column form for A/B runs)
#endif
template <int MODEL>
__device__ __forceinline__ double jac_tile_body(const DevView& v, const double* pose, const double* cam, double mult, int tile, int lane,
                                                double* wl, double* G, int off_in, int cnt_in /* corner range if known (cnt_in >= 0) */) {
  constexpr int nk = MODEL == kFov ? 5 : MODEL == kPoly2 ? 6 : MODEL == kPoly3 ? 7 : MODEL == kKb4 ? 8 : MODEL == kRational6 ? 10 : 4;
  constexpr bool kThreeCols = (7 + nk) <= 12;
  constexpr bool kSideGrad = nk >= 10;         // 16 Jacobian columns: J^T r is accumulated beside the matrix pipe (kGGrad)
  const int off = cnt_in >= 0 ? off_in : __builtin_amdgcn_readfirstlane(v.tile_off[tile]);
  const int cnt = cnt_in >= 0 ? cnt_in : __builtin_amdgcn_readfirstlane(v.tile_off[tile + 1]) - off;
  const int b = (lane >> 2) & 3, i4 = lane & 3;
  // column-block patterns of the three operand reads
  const int xa = b;
  const int xb = kThreeCols ? (b == 0 ? 0 : b == 1 ? 0 : b == 2 ? 1 : 3) : ((b + 1) & 3);
  const int xc = kThreeCols ? (b == 0 ? 1 : b == 1 ? 2 : b == 2 ? 2 : 3) : ((b + 2) & 3);
  double acc[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
  double cost = 0.0;
  double gacc[kSideGrad ? 16 : 1];
#pragma unroll
  for (int i = 0; i < (kSideGrad ? 16 : 1); ++i) gacc[i] = 0.0;
#ifdef VC_JAC_STAMPS
#define JSTAMP(i) do { if (tile == 0 && lane == 0) v.dbg[8 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define JSTAMP(i) do {} while (0)
#endif
  JSTAMP(0);
  if (cnt > 0) {
    TileXf x;
    make_tile_xf(pose, cam, &x);
    double K[nk > 8 ? nk : 8];
#pragma unroll
    for (int i = 0; i < (nk > 8 ? nk : 8); ++i) K[i] = cam[kCamK + i];
    ModelPre pre;
    model_precompute(MODEL, K, &pre);
    double* mine = wl + lane * kDotStride;
    // row (lane >> 4) of a 4-row group: corner (lane >> 5), residual row (lane >> 4) & 1
    const double* rowp = wl + (lane >> 5) * kDotStride + ((lane >> 4) & 1) * 16 + i4;
    const double* pa = rowp + 4 * xa;
    const double* pb = rowp + 4 * xb;
    const double* pc = rowp + 4 * xc;
    // four column blocks: the (b, b + 2) products of TWO steps share one instruction -- blocks 0, 1 take them from the even step
    // (A = block b, B = block b + 2), blocks 2, 3 from the odd step (A = block b - 2, B = block b); see `pair_step`
    const double* pa2 = rowp + (b < 2 ? 4 * b : 4 * (b - 2) + 2 * kDotStride);
    const double* pb2 = rowp + (b < 2 ? 4 * (b + 2) : 4 * b + 2 * kDotStride);
    // software prefetch: the corner of the NEXT pass (detection + target point) is in flight while this pass computes
    double2 uv_n = make_double2(0.0, 0.0);
    double pw_n[3] = {0.0, 0.0, 0.0};
    int id_n = 0;
    if (lane < cnt) {
      uv_n = v.obs_uv[off + lane];
      id_n = v.obs_pt[off + lane];
      const double* pp = v.points + 3 * (size_t)(id_n & kObsPointMask);
      pw_n[0] = pp[0]; pw_n[1] = pp[1]; pw_n[2] = pp[2];
    }
    for (int base = 0; base < cnt; base += 64) {
      const int d = base + lane;
      const double2 uv = uv_n;
      const double pw[3] = {pw_n[0], pw_n[1], pw_n[2]};
      const double mult_d = (id_n & kObsOneLess) ? mult - 1.0 : mult;
      if (d + 64 < cnt) {
        uv_n = v.obs_uv[off + d + 64];
        id_n = v.obs_pt[off + d + 64];
        const double* pp = v.points + 3 * (size_t)(id_n & kObsPointMask);
        pw_n[0] = pp[0]; pw_n[1] = pp[1]; pw_n[2] = pp[2];
      }
      if (base == 0) JSTAMP(1);
      if (d < cnt) {
        if (kSideGrad) {
          double rs[2];
          cost += corner_rows<MODEL>(x, K, pre, pw, uv.x, uv.y, mult_d, mine, mine + 16, rs);
#pragma unroll
          for (int i = 0; i < (kSideGrad ? 16 : 1); ++i) gacc[i] += mine[i] * rs[0] + mine[16 + i] * rs[1];
        } else {
          cost += corner_rows<MODEL>(x, K, pre, pw, uv.x, uv.y, mult_d, mine, mine + 16);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 32; ++i) mine[i] = 0.0;
      }
      wave_lds_sync();
      if (base == 0) JSTAMP(2);
      const int ngroups = (min(64, cnt - base) + 1) >> 1;      // groups of 4 rows (2 corners) that hold data; the rest is zero
      auto step = [&](int s, double a, double bb, double cc) {      // three column blocks: 2 instructions per step
        acc[s][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, a, acc[s][0], 0, 0, 0);
        acc[s][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(bb, cc, acc[s][1], 0, 0, 0);     // A = (0,0,1,3), B = (1,2,2,3)
      };
      // Four column blocks: a step needs the 10 distinct 4x4 blocks of the symmetric 16x16 product -- diagonal (4), (b, b + 1)
      // (4, with (3,0) for (0,3)) and (0,2), (1,3).  The four blocks of an instruction are independent, so the two (b, b + 2)
      // products of two consecutive steps fill ONE instruction: 5 instructions per two steps instead of 6 (the third
      // instruction of a step used to compute (2,0) and (3,1) a second time).  The matrix pipe is what this kernel saturates
      // (profiles/r03_sq_sweep_cfg4_2500.txt), so the instruction count is its time.
      auto pair_step = [&](int s, double a0, double b0, double a1, double b1, double a2, double b2) {
        acc[0][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, a0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, b0, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, a1, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, b1, acc[1][1], 0, 0, 0);
        acc[s][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a2, b2, acc[s][2], 0, 0, 0);
      };
      if (kThreeCols) {
        if (ngroups == 32) {
          // full pass, straight-line: operands of 4 groups at a time, the next chunk's LDS reads in flight under the
          // current chunk's MFMAs
          double ua[2][4], ub[2][4], uc[2][4];
#pragma unroll
          for (int g = 0; g < 4; ++g) { ua[0][g] = pa[g * 2 * kDotStride]; ub[0][g] = pb[g * 2 * kDotStride]; uc[0][g] = pc[g * 2 * kDotStride]; }
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            if (c < 7) {
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const int gg = (c + 1) * 4 + g;
                ua[(c + 1) & 1][g] = pa[gg * 2 * kDotStride]; ub[(c + 1) & 1][g] = pb[gg * 2 * kDotStride]; uc[(c + 1) & 1][g] = pc[gg * 2 * kDotStride];
              }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) step(g & 1, ua[c & 1][g], ub[c & 1][g], uc[c & 1][g]);
          }
        } else {
          // ragged last pass: chunks of 8 groups (rows past the data are zero, so a chunk may run over the end)
          for (int g0 = 0; g0 < ngroups; g0 += 8) {
            double va[8], vb[8], vc8[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) { va[g] = pa[(g0 + g) * 2 * kDotStride]; vb[g] = pb[(g0 + g) * 2 * kDotStride]; vc8[g] = pc[(g0 + g) * 2 * kDotStride]; }
#pragma unroll
            for (int g = 0; g < 8; ++g) step(g & 1, va[g], vb[g], vc8[g]);
          }
        }
      } else {
        if (ngroups == 32) {
          // full pass, straight-line: two pairs of steps at a time, the next chunk's LDS reads in flight under the MFMAs
          double ua[2][4], ub[2][4], u2a[2][2], u2b[2][2];
#pragma unroll
          for (int g = 0; g < 4; ++g) { ua[0][g] = pa[g * 2 * kDotStride]; ub[0][g] = pb[g * 2 * kDotStride]; }
#pragma unroll
          for (int q = 0; q < 2; ++q) { u2a[0][q] = pa2[q * 4 * kDotStride]; u2b[0][q] = pb2[q * 4 * kDotStride]; }
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            if (c < 7) {
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const int gg = (c + 1) * 4 + g;
                ua[(c + 1) & 1][g] = pa[gg * 2 * kDotStride]; ub[(c + 1) & 1][g] = pb[gg * 2 * kDotStride];
              }
#pragma unroll
              for (int q = 0; q < 2; ++q) {
                const int pp = (c + 1) * 2 + q;
                u2a[(c + 1) & 1][q] = pa2[pp * 4 * kDotStride]; u2b[(c + 1) & 1][q] = pb2[pp * 4 * kDotStride];
              }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q)
              pair_step(q, ua[c & 1][2 * q], ub[c & 1][2 * q], ua[c & 1][2 * q + 1], ub[c & 1][2 * q + 1], u2a[c & 1][q], u2b[c & 1][q]);
          }
        } else {
          // ragged last pass: chunks of 4 pairs of steps (rows past the data are zero, so a chunk may run over the end)
          for (int g0 = 0; g0 < ngroups; g0 += 8) {
            double va[8], vb[8], v2a[4], v2b[4];
#pragma unroll
            for (int g = 0; g < 8; ++g) { va[g] = pa[(g0 + g) * 2 * kDotStride]; vb[g] = pb[(g0 + g) * 2 * kDotStride]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) { v2a[q] = pa2[(g0 / 2 + q) * 4 * kDotStride]; v2b[q] = pb2[(g0 / 2 + q) * 4 * kDotStride]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) pair_step(q & 1, va[2 * q], vb[2 * q], va[2 * q + 1], vb[2 * q + 1], v2a[q], v2b[q]);
          }
        }
      }
      wave_lds_sync();
      if (base == 0) JSTAMP(3);
    }
  }
  JSTAMP(4);
  // D lane = 16 i + 4 b + j holds G[4 I_b + i][4 J_b + j]; the packed record takes the upper triangle only (unused blocks stay zero
  // from upload)
  {
    const int i = lane >> 4, j = lane & 3;
    const double d0 = acc[0][0] + acc[1][0], d1 = acc[0][1] + acc[1][1], d2 = acc[0][2] + acc[1][2];
    const int r0 = 4 * b + i, c0 = 4 * b + j;
    if (i <= j) G[g_pack_idx(r0, c0)] = d0;
    const int r1 = 4 * (kThreeCols ? xb : b) + i, c1 = 4 * (kThreeCols ? xc : ((b + 1) & 3)) + j;
    G[g_pack_idx(r1, c1)] = d1;                      // (an off-diagonal block: g_pack_idx orders the pair)
    if (!kThreeCols) {
      // blocks 2, 3 hold the odd steps' share of (0,2), (1,3): added to blocks 0, 1 (eight lanes further on in the row of 16)
      const double d2o = dpp_row_f64<0x128, 0xF>(d2, d2);      // row_ror:8
      if (b < 2) {
        const double d2t = d2 + d2o;
        const int r2 = 4 * b + i, c2 = 4 * (b + 2) + j;
        G[g_pack_idx(r2, c2)] = d2t;
      }
    }
  }
  if (kSideGrad) {
#pragma unroll
    for (int i = 0; i < (kSideGrad ? 16 : 1); ++i) {
      const double t = wave_sum(gacc[i]);
      if (lane == 0) G[kGPackGrad + i] = t;
    }
  }
  JSTAMP(5);
  return wave_sum(cost);          // valid in lane 0
}
// ---- fov / linear / poly2 / poly3 (round 6): the two residual rows of a corner have DIFFERENT structural zeros ------------------------------
// Row u of a radial model has no entry in the columns of fy and cy, row v none in fx and cx: of the 11 .. 14 used columns of
// [A | A x q | B | r] each row type fills 9 .. 12 -- THREE block columns of four (poly2 / poly3: instead of four; fov / linear: three
// before as well, but with both row types in every k-step: 16 sub-block slots per four corners for 12 products).  The Gram block is the sum of the two row
// types' Gram blocks, G = sum u_row^T u_row + sum v_row^T v_row, each over its own compact columns (6 sub-blocks of 4 x 4 instead of 10):
// a group of four corners costs THREE v_mfma_f64_4x4x4 (12 sub-block slots, all used: u-type rows of the four corners as the k dimension
// for six of them, v-type rows for the other six) where the four-block-column form needs five for the same four corners (two k-steps of two
// corners x two rows).  The matrix instructions are this kernel's time: -40 % of them for these models.  The two compact
// 12 x 12 blocks are added into the packed 16 x 16 record once per tile.
//   compact column j of row type t:  0..5 -> A, A x q;  6 -> fx | fy;  7 -> cx | cy;  8.. -> the nk - 4 distortion terms;  then r
template <int MODEL>
__device__ __forceinline__ double jac_tile_body_split(const DevView& v, const double* pose, const double* cam, double mult, int tile, int lane,
                                                      double* wl, double* G, int off_in, int cnt_in) {
  static_assert(MODEL == kPoly2 || MODEL == kPoly3 || MODEL == kFov || MODEL == kLinear, "split rows: the radial models with at most seven intrinsics");
  constexpr int nk = MODEL == kFov ? 5 : MODEL == kPoly2 ? 6 : MODEL == kPoly3 ? 7 : 4;
  constexpr int nd = nk - 4;                       // distortion terms
  constexpr int ncomp = 9 + nd;                    // compact columns in use (11 / 12)
  const int off = cnt_in >= 0 ? off_in : __builtin_amdgcn_readfirstlane(v.tile_off[tile]);
  const int cnt = cnt_in >= 0 ? cnt_in : __builtin_amdgcn_readfirstlane(v.tile_off[tile + 1]) - off;
  const int b = (lane >> 2) & 3, i4 = lane & 3, kq = lane >> 4;
  // (row type, block column of A, of B) of block b in the three instructions of a group
  const int t0 = b == 3 ? 1 : 0, xa0 = b == 3 ? 0 : b, xb0 = xa0;                                   // (0;0,0) (0;1,1) (0;2,2) (1;0,0)
  const int t1 = b == 3 ? 1 : 0, xa1 = b == 0 ? 0 : b == 1 ? 0 : 1, xb1 = b == 0 ? 1 : b == 1 ? 2 : b == 2 ? 2 : 1;      // (0;0,1) (0;0,2) (0;1,2) (1;1,1)
  const int xa2 = b == 0 ? 2 : b == 1 ? 0 : b == 2 ? 0 : 1, xb2 = b == 0 ? 2 : b == 1 ? 1 : 2;      // (1;2,2) (1;0,1) (1;0,2) (1;1,2)
  double acc[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
  double cost = 0.0;
  if (cnt > 0) {
    TileXf x;
    make_tile_xf(pose, cam, &x);
    double K[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) K[i] = cam[kCamK + i];
    ModelPre pre;
    model_precompute(MODEL, K, &pre);
    double* mine = wl + lane * kDotStride;
    const double* rowp = wl + kq * kDotStride + i4;                    // row kq of a group of four corners
    const double* pa0 = rowp + t0 * 16 + 4 * xa0; const double* pb0 = rowp + t0 * 16 + 4 * xb0;
    const double* pa1 = rowp + t1 * 16 + 4 * xa1; const double* pb1 = rowp + t1 * 16 + 4 * xb1;
    const double* pa2 = rowp + 16 + 4 * xa2;      const double* pb2 = rowp + 16 + 4 * xb2;
    double2 uv_n = make_double2(0.0, 0.0);
    double pw_n[3] = {0.0, 0.0, 0.0};
    int id_n = 0;
    if (lane < cnt) {
      uv_n = v.obs_uv[off + lane];
      id_n = v.obs_pt[off + lane];
      const double* pp = v.points + 3 * (size_t)(id_n & kObsPointMask);
      pw_n[0] = pp[0]; pw_n[1] = pp[1]; pw_n[2] = pp[2];
    }
    for (int base = 0; base < cnt; base += 64) {
      const int d = base + lane;
      const double2 uv = uv_n;
      const double pw[3] = {pw_n[0], pw_n[1], pw_n[2]};
      const double mult_d = (id_n & kObsOneLess) ? mult - 1.0 : mult;
      if (d + 64 < cnt) {
        uv_n = v.obs_uv[off + d + 64];
        id_n = v.obs_pt[off + d + 64];
        const double* pp = v.points + 3 * (size_t)(id_n & kObsPointMask);
        pw_n[0] = pp[0]; pw_n[1] = pp[1]; pw_n[2] = pp[2];
      }
      if (d < cnt) {
        double r0[16], r1[16];
        cost += corner_rows<MODEL>(x, K, pre, pw, uv.x, uv.y, mult_d, r0, r1);
        // compact rows: u drops fy (7), cy (9); v drops fx (6), cx (8)
#pragma unroll
        for (int j = 0; j < 6; ++j) { mine[j] = r0[j]; mine[16 + j] = r1[j]; }
        mine[6] = r0[6]; mine[7] = r0[8]; mine[16 + 6] = r1[7]; mine[16 + 7] = r1[9];
#pragma unroll
        for (int j = 0; j < nd + 1; ++j) { mine[8 + j] = r0[10 + j]; mine[16 + 8 + j] = r1[10 + j]; }      // distortion terms, then the residual (column 6 + nk)
#pragma unroll
        for (int j = ncomp; j < 12; ++j) { mine[j] = 0.0; mine[16 + j] = 0.0; }
      } else {
#pragma unroll
        for (int i = 0; i < 12; ++i) { mine[i] = 0.0; mine[16 + i] = 0.0; }
      }
      wave_lds_sync();
      const int ngroups = (min(64, cnt - base) + 3) >> 2;      // groups of four corners that hold data; rows past the data are zero
      auto fetch = [&](int g, double* a3, double* b3) {
        const int o = g * 4 * kDotStride;
        a3[0] = pa0[o]; b3[0] = pb0[o]; a3[1] = pa1[o]; b3[1] = pb1[o]; a3[2] = pa2[o]; b3[2] = pb2[o];
      };
      if (ngroups == 16) {
        // full pass, straight-line: operands of four groups at a time, the next chunk's LDS reads in flight under the current chunk's MFMAs
        double ua[2][4][3], ub[2][4][3];
#pragma unroll
        for (int g = 0; g < 4; ++g) fetch(g, ua[0][g], ub[0][g]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (c < 3) {
#pragma unroll
            for (int g = 0; g < 4; ++g) fetch((c + 1) * 4 + g, ua[(c + 1) & 1][g], ub[(c + 1) & 1][g]);
          }
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[g & 1][n] = __builtin_amdgcn_mfma_f64_4x4x4f64(ua[c & 1][g][n], ub[c & 1][g][n], acc[g & 1][n], 0, 0, 0);
        }
      } else {
        for (int g0 = 0; g0 < ngroups; g0 += 4) {              // ragged last pass: chunks of four groups (a chunk may run over the end into zero rows)
          double ua[4][3], ub[4][3];
#pragma unroll
          for (int g = 0; g < 4; ++g) fetch(g0 + g, ua[g], ub[g]);
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[g & 1][n] = __builtin_amdgcn_mfma_f64_4x4x4f64(ua[g][n], ub[g][n], acc[g & 1][n], 0, 0, 0);
        }
      }
      wave_lds_sync();
    }
  }
  // ---- the two compact blocks -> the packed record.  D lane = 16 i + 4 b + j holds entry (4 XA + i, 4 XB + j) of row type t's block.
  {
    double* GC = wl;                                // [2][12 x 12], upper block triangle written (this wavefront's corner rows are done with)
    const int i = lane >> 4, j = lane & 3;
    const double d0 = acc[0][0] + acc[1][0], d1 = acc[0][1] + acc[1][1], d2 = acc[0][2] + acc[1][2];
    GC[t0 * 144 + (4 * xa0 + i) * 12 + 4 * xb0 + j] = d0;
    GC[t1 * 144 + (4 * xa1 + i) * 12 + 4 * xb1 + j] = d1;
    GC[144 + (4 * xa2 + i) * 12 + 4 * xb2 + j] = d2;
    wave_lds_sync();
    // compact column of full column fc in row type t (-1: structurally zero there / unused)
    auto comp = [&](int fc, int t) {
      return fc < 6 ? fc : fc == 6 ? (t == 0 ? 6 : -1) : fc == 7 ? (t == 1 ? 6 : -1) : fc == 8 ? (t == 0 ? 7 : -1) : fc == 9 ? (t == 1 ? 7 : -1)
           : fc <= 6 + nk ? 8 + (fc - 10) : -1;
    };
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int e = lane + 64 * q;
      if (e < kGPackGrad) {
        int r = 0;
#pragma unroll
        for (int a2 = 1; a2 < 16; ++a2) r += (e >= a2 * 16 - (a2 * (a2 - 1)) / 2) ? 1 : 0;      // row of packed entry e
        const int c = r + (e - (r * 16 - (r * (r - 1)) / 2));
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int jr = comp(r, t), jc = comp(c, t);
          // (the upper block triangle is what was written: entry (jr, jc), jr <= jc, lies in block (jr / 4, jc / 4); inside a diagonal
          //  block both triangles are there)
          const double g = GC[t * 144 + (jr >= 0 ? jr : 0) * 12 + (jc >= 0 ? jc : 0)];
          s += (jr >= 0 && jc >= 0) ? g : 0.0;
        }
        G[e] = s;
      }
    }
  }
  return wave_sum(cost);          // valid in lane 0
}
__device__ __forceinline__ double jac_tile_dispatch(const DevView& v, int model, const double* pose, const double* cam, double mult, int tile,
                                                    int lane, double* wl, double* G, int off = 0, int cnt = -1) {
  switch (model) {   // wave-uniform
    case kFov: return jac_tile_body<kFov>(v, pose, cam, mult, tile, lane, wl, G, off, cnt);
    // (fov / linear through the split form as well -- 48 instead of 64 matrix instructions per pass -- measured SLOWER: 237 against 233.5 us
    //  at cfg5 / 6250 frames, k_trial 16.8 against 15.6 us at cfg2: their tiles are three passes long and the per-tile assembly of the
    //  two compact blocks costs more than the instructions saved; poly3 at cfg4's nine passes per tile: 150 against 161 us)
#if VC_JAC_SPLIT_ROWS
    case kPoly2: return jac_tile_body_split<kPoly2>(v, pose, cam, mult, tile, lane, wl, G, off, cnt);
    case kPoly3: return jac_tile_body_split<kPoly3>(v, pose, cam, mult, tile, lane, wl, G, off, cnt);
#else
    case kPoly2: return jac_tile_body<kPoly2>(v, pose, cam, mult, tile, lane, wl, G, off, cnt);
    case kPoly3: return jac_tile_body<kPoly3>(v, pose, cam, mult, tile, lane, wl, G, off, cnt);
#endif
    case kKb4: return jac_tile_body<kKb4>(v, pose, cam, mult, tile, lane, wl, G, off, cnt);
    case kRational6: return jac_tile_body<kRational6>(v, pose, cam, mult, tile, lane, wl, G, off, cnt);
    default: return jac_tile_body<kLinear>(v, pose, cam, mult, tile, lane, wl, G, off, cnt);
  }
}

// trial = 0: linearisation at the accepted state (only when the control record asks for one); trial = 1: the sweep runs at the
// trial state and fills buffer 1 - cur -- its cost is the trial cost, its Gram blocks are the next linearisation if the step is
// accepted (the decision flips `cur`), and a rejected step leaves buffer cur untouched.
#ifndef VC_JAC_WAVES
#define VC_JAC_WAVES 2
#endif
__global__ __launch_bounds__(256, VC_JAC_WAVES) void k_reproj_jac(DevView v, int trial) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const Ctrl* ct = v.ctrl;
#ifdef VC_JAC_STAMPS
  if (blockIdx.x == 0 && threadIdx.x == 0) v.dbg[7] = (long long)__builtin_readcyclecounter();
#endif
  __shared__ double s_cost[4];
  // the trial sweep follows the back-substitution on the main stream: that this kernel has started says the trial poses are complete
  // and written back -- published for the second stream's k_imu_jac (no event record between the two kernels of the critical path)
  if (trial && blockIdx.x == 0 && threadIdx.x == 0) signal_started(v, 2);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + wave;
  // the tile's header (frame, camera, model, corner range: one 48-byte record) is requested together with the control record, not behind
  // it (round 6: control record -> tile_frame / tile_cam -> model -> tile_off -> corners -> target points were five dependent round
  // trips at the head of a 17 us kernel; now three)
  const TileHdr h = v.tile_hdr[tile < v.n_tiles ? tile : 0];
  const int c_done = ct->done, c_need = ct->need_lin, c_cur = ct->cur;
  const double c_mult = ct->mult;
  if (c_done || (!trial && !c_need)) return;
  double cost = 0.0;
  if (tile < v.n_tiles) {
    double* wl = lds + wave * 64 * kDotStride;
    const int cur = trial ? 1 - c_cur : c_cur;
    const int f = __builtin_amdgcn_readfirstlane(h.frame), c = __builtin_amdgcn_readfirstlane(h.cam);
    cost = jac_tile_dispatch(v, __builtin_amdgcn_readfirstlane(h.model), v.poses[cur] + (size_t)f * kPoseStride, v.cams[cur] + (size_t)c * kCamStride, c_mult,
                             tile, lane, wl, v.Gb[cur] + (size_t)tile * kGPack, __builtin_amdgcn_readfirstlane(h.off), __builtin_amdgcn_readfirstlane(h.cnt));
    if (lane == 0) v.tile_costb[cur][tile] = cost;
  }
  if (trial) {               // the workgroup's share of the trial cost, in fixed order
    if (lane == 0) s_cost[wave] = cost;
    __syncthreads();
    if (threadIdx.x == 0) v.wg_trial[blockIdx.x] = (s_cost[0] + s_cost[1]) + (s_cost[2] + s_cost[3]);
  }
}

// ------------------------------------------------------------------------------------------ residual sweeps
template <int MODEL>
__device__ __forceinline__ void res_tile_sweep(const DevView& v, const TileXf& x, const double* K, int off, int cnt, int lane,
                                               double mult, double* cost_out, double* sq_out) {
  double cost = 0.0, sq = 0.0;
  ModelPre pre;
  model_precompute(MODEL, K, &pre);
  for (int d = lane; d < cnt; d += 64) {
    const double2 uv = v.obs_uv[off + d];
    const int id = v.obs_pt[off + d];
    const double* pw = v.points + 3 * (size_t)(id & kObsPointMask);
    double r[2];
    cost += ((id & kObsOneLess) ? mult - 1.0 : mult) * corner_residual<MODEL>(x, K, pre, pw, uv.x, uv.y, r);
    sq += r[0] * r[0] + r[1] * r[1];
  }
  *cost_out = wave_sum(cost);
  *sq_out = wave_sum(sq);
}
__device__ __forceinline__ void res_tile_dispatch(const DevView& v, int model, const TileXf& x, const double* K, int off, int cnt,
                                                  int lane, double mult, double* cost, double* sq) {
  switch (model) {
    case kFov: res_tile_sweep<kFov>(v, x, K, off, cnt, lane, mult, cost, sq); break;
    case kPoly2: res_tile_sweep<kPoly2>(v, x, K, off, cnt, lane, mult, cost, sq); break;
    case kPoly3: res_tile_sweep<kPoly3>(v, x, K, off, cnt, lane, mult, cost, sq); break;
    case kKb4: res_tile_sweep<kKb4>(v, x, K, off, cnt, lane, mult, cost, sq); break;
    case kRational6: res_tile_sweep<kRational6>(v, x, K, off, cnt, lane, mult, cost, sq); break;
    default: res_tile_sweep<kLinear>(v, x, K, off, cnt, lane, mult, cost, sq); break;
  }
}
// plain sweep of one state buffer (RMSE, vc_evaluate): tile_trial[t] = {mult * sum rho, sum |r|^2}
__global__ __launch_bounds__(256) void k_reproj_res(DevView v, int state, double mult) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= v.n_tiles) return;
  if (state >= 2) {            // 2: accepted buffer, 3: trial buffer of the running solve (multiplicity from Ctrl)
    const Ctrl* ct = v.ctrl;
    if (ct->done) return;
    state = (state == 3) ? 1 - ct->cur : ct->cur;
    mult = ct->mult;
  }
  const int f = v.tile_frame[tile], c = v.tile_cam[tile];
  const double* cam = v.cams[state] + (size_t)c * kCamStride;
  TileXf x;
  make_tile_xf(v.poses[state] + (size_t)f * kPoseStride, cam, &x);
  double K[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) K[i] = cam[kCamK + i];
  double cost, sq;
  res_tile_dispatch(v, v.cd[c].model, x, K, v.tile_off[tile], v.tile_off[tile + 1] - v.tile_off[tile], lane, mult, &cost, &sq);
  if (lane == 0) { v.tile_trial[2 * tile] = cost; v.tile_trial[2 * tile + 1] = sq; }
}

// per-corner outlier mask (RemoveOutliers, vicalibrator.h:859-916): |r| > thresh[cam]
template <int MODEL>
__device__ __forceinline__ void mask_tile_body(const DevView& v, const TileXf& x, const double* K, int off, int cnt, int lane,
                                               double th, unsigned char* mask) {
  ModelPre pre;
  model_precompute(MODEL, K, &pre);
  for (int d = lane; d < cnt; d += 64) {
    const double2 uv = v.obs_uv[off + d];
    double r[2];
    corner_residual<MODEL>(x, K, pre, v.points + 3 * (size_t)(v.obs_pt[off + d] & kObsPointMask), uv.x, uv.y, r);
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
  double K[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) K[i] = cam[kCamK + i];
  const double th = thresh[c];
  switch (v.cd[c].model) {
    case kFov: mask_tile_body<kFov>(v, x, K, off, cnt, lane, th, mask); break;
    case kPoly2: mask_tile_body<kPoly2>(v, x, K, off, cnt, lane, th, mask); break;
    case kPoly3: mask_tile_body<kPoly3>(v, x, K, off, cnt, lane, th, mask); break;
    case kKb4: mask_tile_body<kKb4>(v, x, K, off, cnt, lane, th, mask); break;
    case kRational6: mask_tile_body<kRational6>(v, x, K, off, cnt, lane, th, mask); break;
    default: mask_tile_body<kLinear>(v, x, K, off, cnt, lane, th, mask); break;
  }
}

// ------------------------------------------------------------------------------------------ frame elimination
// k_frame_schur: one workgroup per chunk of frames, one wavefront per frame (groups of 4 frames).
//  per frame   lanes 0..35 own the entries of H_pp, 36..41 those of g_p; every lane then factors the damped
//              6x6 block redundantly (wave-uniform); lanes own (tile, column) pairs of Y = L^-1 W.
//  per group   the 4 x 6 rows [Y_f | z_f] (dense over the shared columns) sit in LDS and their Gram matrix
//              sum Y^T Y, sum Y^T z is accumulated on the matrix pipe (v_mfma_f64_16x16x4_f64), 16x16
//              column-tile pairs distributed over the 4 wavefronts.
//  per chunk   part = [ sum Y^T Y (D x D, upper) | sum Y^T z (D) | per-camera sum of G (C x 256) ]
constexpr int kPrepPad = 48;
constexpr int kMaxPairsPerWave = 9;      // 8 column tiles -> 36 pairs over 4 waves
__device__ __forceinline__ int schur_ld(int D) { const int Dp = ((D + 1 + 15) / 16) * 16; return (Dp % 32 == 0) ? Dp + 16 : Dp; }

template <int MAXC, int MAXP, int MINW>
__global__ __launch_bounds__(256, MINW) void k_frame_schur(DevView v) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  __shared__ Ctrl s_ctrl;
  __shared__ CamDesc s_cd[kMaxCams];     // LDS copy of the kernel-argument table (dynamically indexed below)
  __shared__ unsigned short s_pair[4 * kMaxPairsPerWave];      // column-tile pair p -> I | J << 8
  if (threadIdx.x < kMaxCams) s_cd[threadIdx.x] = v.cd[threadIdx.x];
  {
    const int nT_ = (v.D + 1 + 15) / 16, nP_ = nT_ * (nT_ + 1) / 2;
    if ((int)threadIdx.x < nP_ && threadIdx.x < 4 * kMaxPairsPerWave) {
      int I = 0, J = 0;
      for (int k = 0; k < (int)threadIdx.x; ++k) if (++J == nT_) { ++I; J = I; }
      s_pair[threadIdx.x] = (unsigned short)(I | (J << 8));
    }
  }
  const Ctrl* ct = v.ctrl;
  if (v.merged) { merged_control(v, &s_ctrl, sh, blockIdx.x == 0); ct = &s_ctrl; }
  else __syncthreads();
  if (ct->done) return;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int C = v.n_cams, D = v.D;
  const int nT = (D + 1 + 15) / 16, ld = schur_ld(D), nPairs = min(nT * (nT + 1) / 2, 4 * kMaxPairsPerWave);
  double* Gw = sh + wave * (C * kGStride + kPrepPad);
  double* Hs = Gw + C * kGStride;
  double* R = sh + 4 * (C * kGStride + kPrepPad);
  const int cur = ct->cur;
  const int init_scale = ct->init_scale, reuse = ct->reuse_diag;
  const double radius = ct->radius;
  const double* cams = v.cams[cur];
  const int chunk = blockIdx.x;
  const int f0 = chunk * v.chunk_frames, f1 = min(f0 + v.chunk_frames, v.n_frames);
  v4d acc[MAXP];
#pragma unroll
  for (int i = 0; i < MAXP; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
  double gsum[MAXC][5];      // [4]: the record's side vector (lanes < 16)
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
#pragma unroll
    for (int q = 0; q < 5; ++q) gsum[c][q] = 0.0;
  double csum = 0.0;      // lane t: cost of tile t of this wave's frames (the chunk's cost rides in the partials)
  double x2_noobs = 0.0;  // lane 0: parameter norm of this wave's frames that have no observations

  for (int fg = f0; fg < f1; fg += 4) {
    const int f = fg + wave;
    for (int i = lane; i < 6 * ld; i += 64) R[(wave * 6) * ld + i] = 0.0;
    const int t0 = (f < f1) ? v.frame_tile_off[f] : 0;
    const int nt = (f < f1) ? v.frame_tile_off[f + 1] - t0 : 0;
    if (lane < nt) csum += v.tile_costb[cur][t0 + lane];
    // requested once per frame, ahead of their use: the camera of every tile (lane t) and the frame's damping inputs
    const int my_cam = (lane < nt) ? v.tile_cam[t0 + lane] : 0;
    double pre_sc2[6], pre_dg[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      pre_sc2[i] = (nt > 0 && !init_scale) ? v.fscale2[(size_t)f * 6 + i] : 1.0;
      pre_dg[i] = (nt > 0 && reuse) ? v.fdiag[(size_t)f * 6 + i] : 1.0;
    }
    if (f < f1 && nt == 0 && lane == 0) {     // frame without observations: nothing to eliminate, keeps its pose
      const double* p = v.poses[cur] + (size_t)f * kPoseStride;
      double x2 = 0;
      for (int i = 0; i < 7; ++i) x2 += p[i] * p[i];
      double* o = v.fpart + (size_t)f * kNumScal;
      for (int i = 0; i < kNumScal; ++i) o[i] = 0.0;
      o[kScX2] = x2;
      x2_noobs += x2;
    }
    if (nt > 0) {
      for (int t = 0; t < nt; ++t) {           // full 16x16 Gram block of every tile of the frame
        double val[5];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int e = q * 64 + lane; val[q] = v.Gb[cur][(size_t)(t0 + t) * kGPack + g_pack_idx(e >> 4, e & 15)]; }      // (packed record: expanded on load)
        val[4] = (lane < 16) ? v.Gb[cur][(size_t)(t0 + t) * kGPack + kGPackGrad + lane] : 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) Gw[t * kGStride + q * 64 + lane] = val[q];
        if (lane < 16) Gw[t * kGStride + kGGrad + lane] = val[4];
        const int c = __builtin_amdgcn_readlane(my_cam, t);      // wave-uniform: scalar branch, one camera's accumulators touched
#pragma unroll
        for (int k = 0; k < MAXC; ++k)
          if (k == c) {
#pragma unroll
            for (int q = 0; q < 5; ++q) gsum[k][q] += val[q];
          }
      }
      wave_lds_sync();
      double hval = 0.0;
      if (lane < 42) {
        for (int t = 0; t < nt; ++t) {
          const int c = __builtin_amdgcn_readlane(my_cam, t);
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
            const int nk = model_nk(s_cd[c].model);
            double s = 0.0;
#pragma unroll
            for (int p = 0; p < 3; ++p) s += Rm[3 * p + ii] * gram_grad(g, 3 * a + p, nk);
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
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const double hd = H[i * 6 + i];
        double sc2, dg;
        if (init_scale) { sc2 = jacobi_scale2(hd); if (lane == 0) v.fscale2[(size_t)f * 6 + i] = sc2; }
        else sc2 = pre_sc2[i];
        if (!reuse) { dg = lm_clamped_diag(hd, sc2); if (lane == 0) v.fdiag[(size_t)f * 6 + i] = dg; }
        else dg = pre_dg[i];
        lam[i] = dg / (radius * sc2);
        H[i * 6 + i] = hd + lam[i];
      }
      double dinv[6];
      if (!chol_small<6>(H, dinv)) {
        if (lane == 0) atomicAdd(&v.flags[4 + 2 * v.par], 1);
#pragma unroll
        for (int i = 0; i < 36; ++i) H[i] = (i % 7 == 0) ? 1.0 : 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) dinv[i] = 1.0;
      }
      double z[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) z[i] = g6[i];
      fwd_solve_inv<6>(H, dinv, z);
      if (lane == 0) {
        double* fr = v.fr + (size_t)f * kFrStride;
        int k = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = 0; j <= i; ++j) fr[kFrL + k++] = H[i * 6 + j];
#pragma unroll
        for (int i = 0; i < 6; ++i) { fr[kFrZ + i] = z[i]; fr[kFrG + i] = g6[i]; fr[kFrLam + i] = lam[i]; fr[kFrDinv + i] = dinv[i]; R[(wave * 6 + i) * ld + D] = z[i]; }
      }
      for (int idx = lane; idx < nt * 16; idx += 64) {      // Y columns: lane -> (tile, column)
        const int t = idx >> 4, j = idx & 15;
        const int c = __shfl(my_cam, t, 64);
        const int flags = s_cd[c].flags, nk = model_nk(s_cd[c].model);
        const int nrot = (flags & kCamRotFree) ? 3 : 0, ntr = (flags & kCamTransFree) ? 3 : 0;
        const int nc = nrot + ntr + ((flags & kCamKFree) ? nk : 0);
        double w[6] = {0, 0, 0, 0, 0, 0};
        if (j < nc) {
          double Rm[9];
          quat_to_R(cams + (size_t)c * kCamStride, Rm);
          const double* g = Gw + t * kGStride;
          double u[6];   // column of [Gaa Ea | GaB]
          if (j < nrot) {
#pragma unroll
            for (int r = 0; r < 6; ++r) u[r] = -(g[r * 16 + 3] * Rm[j] + g[r * 16 + 4] * Rm[3 + j] + g[r * 16 + 5] * Rm[6 + j]);
          } else {
            const int jj = (j < nrot + ntr) ? j - nrot : 6 + (j - nrot - ntr);
#pragma unroll
            for (int r = 0; r < 6; ++r) u[r] = g[r * 16 + jj];
          }
#pragma unroll
          for (int i = 0; i < 3; ++i) {   // w = Qa^T u, Qa = diag(-R, R)
            w[i] = -(Rm[i] * u[0] + Rm[3 + i] * u[1] + Rm[6 + i] * u[2]);
            w[3 + i] = Rm[i] * u[3] + Rm[3 + i] * u[4] + Rm[6 + i] * u[5];
          }
          fwd_solve_inv<6>(H, dinv, w);
          const int col = s_cd[c].col0 + j;
#pragma unroll
          for (int r = 0; r < 6; ++r) R[(wave * 6 + r) * ld + col] = w[r];
        }
        double* Yt = v.Y + (size_t)(t0 + t) * kYStride;
#pragma unroll
        for (int r = 0; r < 6; ++r) Yt[r * kUCols + j] = w[r];
      }
    }
    __syncthreads();
    {
      // accumulator q of wavefront w belongs to pair 4 q + w (compile-time index: no select chains, and the MAXP chains of
      // dependent MFMAs advance side by side); past the last pair the last one is recomputed and never written
      const double* rq = R + (lane >> 4) * ld + (lane & 15);
      int oa[MAXP], ob[MAXP];
#pragma unroll
      for (int q = 0; q < MAXP; ++q) {
        const int ij = s_pair[min(4 * q + wave, nPairs - 1)];
        oa[q] = (ij & 255) * 16; ob[q] = (ij >> 8) * 16;
      }
#pragma unroll
      for (int ks = 0; ks < 6; ++ks)
#pragma unroll
        for (int q = 0; q < MAXP; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(rq[ks * 4 * ld + oa[q]], rq[ks * 4 * ld + ob[q]], acc[q], 0, 0, 0);
    }
    __syncthreads();
  }
  double* part = v.part + (size_t)chunk * v.part_stride;
#pragma unroll
  for (int q = 0; q < MAXP; ++q) {
    const int p = 4 * q + wave;
    if (p < nPairs) {
      const int ij = s_pair[p], I = ij & 255, J = ij >> 8;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = I * 16 + (lane >> 4) + 4 * g, col = J * 16 + (lane & 15);
        if (row < D) { if (col < D) part[row * D + col] = acc[q][g]; else if (col == D) part[D * D + row] = acc[q][g]; }
      }
    }
  }
  // per-camera sum of G over the chunk: combine the 4 wavefronts through LDS (fixed order)
  __syncthreads();
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c < C) {
#pragma unroll
      for (int q = 0; q < 4; ++q) sh[wave * (C * kGStride) + c * kGStride + q * 64 + lane] = gsum[c][q];
      if (lane < 16) sh[wave * (C * kGStride) + c * kGStride + kGGrad + lane] = gsum[c][4];
    }
  __syncthreads();
  for (int e = tid; e < C * kGStride; e += 256)
    part[D * D + D + e] = (sh[e] + sh[C * kGStride + e]) + (sh[2 * C * kGStride + e] + sh[3 * C * kGStride + e]);
  // chunk cost (sum of the tiles' robustified costs at the linearisation point), last slot of the partial record
  __syncthreads();
  csum = wave_sum(csum);
  if (lane == 0) { sh[wave] = csum; sh[4 + wave] = x2_noobs; }
  __syncthreads();
  if (tid == 0) { part[v.part_stride - 1] = (sh[0] + sh[1]) + (sh[2] + sh[3]); part[v.part_stride - 2] = (sh[4] + sh[5]) + (sh[6] + sh[7]); }
}

// Fixed-order sum of the chunk partials, spread over many CUs (one CU can only pull ~20-50 GB/s): workgroup x owns 16 entries
// (one 128-byte line per chunk); thread (entry = tid & 15, slice = tid >> 4) sums the partials k = slice (mod 32) -- all of a
// thread's loads are independent, two or three rounds of 16 --, the 32 slices are then added in order.  The result goes where its
// consumer wants it: -sum into Sbuf's S (both triangles: the producers only fill tile pairs I <= J, whose mirror image is written
// here -- entries of the lower block triangle are neither summed nor read) and g_red, the per-camera Gram sums / IMU parameter
// block / chunk scalars into part_total.  k_reduced (one workgroup) used to add slab totals, negate and mirror itself: 0.5 MB
// through one CU plus a D^2 loop of dependent round trips, 24 + 8 us at D = 67.  (A two-level version whose last workgroup per
// column added the slab totals behind a device-scope fence was 14x slower: every release fence writes the L2 back.)
constexpr int kSumEntries = 16, kSumSlices = 32;
template <int SLICES>
__device__ __forceinline__ void part_sum_block(const DevView& v, int block, double* sl /* 16 x SLICES */) {
  // (the control record is requested here and looked at once the partials have been requested as well: a finished solve costs a few
  //  wasted loads, a running one saves the record's round trip ahead of the sum's)
  const int done = v.ctrl->done;
  const int tid = threadIdx.x, e = block * kSumEntries + (tid & (kSumEntries - 1)), ks = tid / kSumEntries;
  const int stride = v.part_stride, D = v.D, DD = D * D, n = v.n_part;
  int i = 0, j = 0;
  bool live = e < stride;
  if (e < DD) { i = e / D; j = e - i * D; live = (i >> 4) <= (j >> 4); }
  double s = 0.0;
  if (live) {
    const double* src = v.part + e;
#pragma unroll 16
    for (int k = ks; k < n; k += SLICES) s += src[(size_t)k * stride];
  }
  if (done) return;
  sl[tid] = s;
  __syncthreads();
  if (tid < kSumEntries && live) {
    double t = sl[tid];
#pragma unroll
    for (int q = 1; q < SLICES; ++q) t += sl[q * kSumEntries + tid];
    double* S = v.Sbuf;
    if (e < DD) { S[e] = -t; if ((i >> 4) < (j >> 4)) S[j * D + i] = -t; }
    else if (e < DD + D) S[e] = -t;        // g_red follows S
    else v.part_total[e] = t;
  }
}
__global__ __launch_bounds__(kSumEntries * kSumSlices) void k_part_sum(DevView v) {
  __shared__ double sl[kSumEntries * kSumSlices];
  part_sum_block<kSumSlices>(v, (int)blockIdx.x, sl);
}

// (kSmallD, vc_device.h: reduced systems up to this width are solved by one wavefront; above it the workgroup-wide LDS factorisation is faster)
// Phase A of the reduced system (one workgroup):
// Sbuf = [ S = H_ss - sum Y^T Y (full symmetric, undamped) | g_red | diag(H_ss) | g_s | cost, 0 ]
struct FinalLds { double gsum[(kMaxCams + 1) * kGStride]; double P[kMaxCams * 256]; double T1[kMaxCams * 256]; double red[256]; double camq[kMaxCams * 4]; double gc[kMaxCams * 16]; };
// shader-clock stamps of k_reduced's phases (tools/dbg_stamps.py): profiling builds only (-DVC_REDUCED_STAMPS) -- each stamp is
// a global store whose acknowledgement the next barrier waits for (~1.5k cycles apiece)
#ifdef VC_REDUCED_STAMPS
// (round 6: the stamps are kept in LDS and copied out at the kernel's end -- as global stores every stamp cost ~1.5k cycles at the next barrier)
__shared__ long long s_rst[32];
#define VC_STAMP(i) do { if (threadIdx.x == 0) s_rst[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define VC_STAMP(i) do { } while (0)
#endif
}  // namespace vc
#include "vc_reduced_tail.hpp"
#include "vc_shared_blocks.hpp"
namespace vc {
// S: v.Sbuf, or (single process, D <= kSmallD) an LDS image of it that this phase fills first -- every read-modify-write of the
// phase and the solve's row loads then stay on chip (three L2 round trips less on the critical path); the kernel writes it back.
// top / top_rows (early Gram, DevView::gram_top_stride): the [Y | z] rows of the chain's top-level frames (LDS, 64 rows x kTopLd, zero
// beyond the rows and beyond column D) -- their Gram sums are subtracted here
constexpr int kTopLd = 34;
// the top-level frames' Gram sums (early Gram, D + 1 <= 32 columns): S -= Y^T Y, g_red -= Y^T z on the matrix pipe.  The (at most 64, zero-padded) rows are
// the k dimension, the column tiles (0,0), (0,1), (1,1) one wavefront each (v_mfma_f64_16x16x4; entries (i, j) and (j, i) are the same products in the same
// order: S stays symmetric to the bit).  (As a scalar loop -- thread per entry, two LDS reads per row and entry -- this took 6 us: 0.7 MB through the LDS pipe.)
__device__ __forceinline__ void top_gram_subtract(double* S, int D, const double* top, int top_rows) {
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  if (wave < 3) {
    const int I = wave == 2 ? 1 : 0, J = wave == 0 ? 0 : 1;
    const double* pa = top + (lane >> 4) * kTopLd + I * 16 + (lane & 15);
    const double* pb = top + (lane >> 4) * kTopLd + J * 16 + (lane & 15);
    // (four accumulators over the 16 k-steps of the 64 padded rows: the instruction's ~300 cycles of latency four times, not sixteen)
    v4d ac[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) ac[u] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 16; ks += 4) {
      if (4 * ks < top_rows) {      // (wave-uniform)
#pragma unroll
        for (int u = 0; u < 4; ++u) ac[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[(ks + u) * 4 * kTopLd], pb[(ks + u) * 4 * kTopLd], ac[u], 0, 0, 0);
      }
    }
    const v4d acc = (ac[0] + ac[1]) + (ac[2] + ac[3]);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int row = I * 16 + (lane >> 4) + 4 * g, col = J * 16 + (lane & 15);
      if (row < D) {
        if (col < D) { S[row * D + col] -= acc[g]; if (I != J) S[col * D + row] -= acc[g]; }
        else if (col == D) S[D * D + row] -= acc[g];
      }
    }
  }
}
// rows of the pinned frames of a sharded chain (separator of this rank, ghost of the next rank's): straight into the reduced system
__device__ __forceinline__ void pinned_rows_phase(const DevView& v, double* S, double* gred, double* hd, double* gs) {
  const int tid = threadIdx.x, D = v.D;
  for (int slot = 0; slot < 2; ++slot) {
    if (!(slot ? v.pin_last : v.pin_first)) continue;
    const int sc = slot ? v.sep_col1 : v.sep_col0;
    const double* st = v.sep_strip + (size_t)slot * 9 * v.ldw;
    for (int e = tid; e < 9 * (D + 1); e += 256) {
      const int i = e / (D + 1), col = e % (D + 1);
      const double val = st[i * v.ldw + col];
      if (col == D) { gred[sc + i] += val; gs[sc + i] += val; }
      else if (col < sc) { const double sn = S[col * D + sc + i] + val; S[col * D + sc + i] = sn; S[(sc + i) * D + col] = sn; }
      else if (col >= sc + i) {            // each pair once, both triangles written
        const double sn = S[(sc + i) * D + col] + val;
        S[(sc + i) * D + col] = sn;
        if (col == sc + i) hd[sc + i] += val; else S[col * D + sc + i] = sn;
      }
    }
    __syncthreads();
  }
}
__device__ __forceinline__ void schur_final_phase(const DevView& v, int cur, FinalLds& L, double* x2_noobs, const CamDesc* cd /* LDS copy of v.cd */, const int* ipc /* ... of v.imu_param_col */,
                                                  double* S, bool s_in_lds, const double* top = nullptr, int top_rows = 0) {
  VC_STAMP(0);
  const int tid = threadIdx.x, D = v.D, C = v.n_cams;
  if (v.hadd_early) {
    // (round 6) the camera blocks, the IMU block and the chunk costs were formed ahead of this launch (DevView::hadd, vc_shared_blocks.hpp): the
    // reduced system is Sbuf (the frames' partial sums, k_part_sum) + that record, minus the top-level frames' Gram sums
    const double* H = v.hadd;
    const int nS = D * D + D, nAll = nS + 2 * D + 2;
    if (s_in_lds) {
      constexpr int kAIter = (kSmallD * kSmallD + 3 * kSmallD + 2 + 255) / 256;
      double s_in[kAIter], h_in[kAIter];
#pragma unroll
      for (int q = 0; q < kAIter; ++q) { const int e = tid + 256 * q; s_in[q] = e < nS ? v.Sbuf[e] : 0.0; h_in[q] = e < nAll ? H[e] : 0.0; }
#pragma unroll
      for (int q = 0; q < kAIter; ++q) { const int e = tid + 256 * q; if (e < nAll) S[e] = s_in[q] + h_in[q]; }
      if (top) { __syncthreads(); top_gram_subtract(S, D, top, top_rows); }
    } else {
      // (eight entries per thread and round: all loads out before the first store -- the compiler cannot tell S and H apart, a plain loop is one
      //  memory round trip per entry: 14.7k cycles at D = 67)
      for (int e0 = tid; e0 < nAll; e0 += 8 * 256) {
        double a[8], h[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int e = e0 + 256 * q; a[q] = e < nS ? S[e] : 0.0; h[q] = e < nAll ? H[e] : 0.0; }
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int e = e0 + 256 * q; if (e < nAll) S[e] = a[q] + h[q]; }
      }
    }
    VC_STAMP(1); VC_STAMP(2); VC_STAMP(3);
    __syncthreads();
    if (v.imu_on) pinned_rows_phase(v, S, S + D * D, S + nS, S + nS + D);
    return;
  }
  // (all of a thread's loads go out before the first is stored to LDS: a plain copy loop is one memory round trip per iteration)
  constexpr int kSIter = (kSmallD * kSmallD + kSmallD + 255) / 256;
  double s_in[kSIter];
  if (s_in_lds) {      // S and g_red as k_part_sum left them
#pragma unroll
    for (int q = 0; q < kSIter; ++q) { const int e = tid + 256 * q; s_in[q] = e < D * D + D ? v.Sbuf[e] : 0.0; }
  }
  double* gred = S + D * D;
  double* hd = gred + D;
  double* gs = hd + D;
  double* sc = gs + D;
  const int stride = v.part_stride;
  // S = -sum of the partials (both triangles) and g_red are in place (k_part_sum's last workgroups); the rest of the summed
  // record -- per-camera Gram sums, IMU parameter block, chunk scalars -- comes from part_total
  const double* ptot = v.part_total;
  // camera rotations: fetched now, under the loads' latency, instead of one dependent global load per camera later
  if (tid < C * 4) L.camq[tid] = v.cams[cur][(size_t)(tid >> 2) * kCamStride + (tid & 3)];
  for (int e0 = D * D + D + tid; e0 < stride - 1; e0 += 4 * 256) {
    double t[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int e = e0 + 256 * q; t[q] = e < stride - 1 ? ptot[e] : 0.0; }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = e0 + 256 * q;
      if (e < stride - 2) L.gsum[e - D * D - D] = t[q];
      else if (e == stride - 2) *x2_noobs = t[q];                 // x2 of observation-less frames
    }
  }
  if (s_in_lds) {
#pragma unroll
    for (int q = 0; q < kSIter; ++q) { const int e = tid + 256 * q; if (e < D * D + D) S[e] = s_in[q]; }
    if (top) { __syncthreads(); top_gram_subtract(S, D, top, top_rows); }
  }
  VC_STAMP(1);
  for (int i = tid; i < D; i += 256) { hd[i] = 0.0; gs[i] = 0.0; }
  if (tid == 0) { sc[0] = 0.5 * ptot[stride - 1]; sc[1] = 0.0; }      // chunk costs (the chain path's k_chain_init fills the same slot)
  __syncthreads();
  VC_STAMP(2);
  shared_blocks_phase(v, 256, L.gsum, L.P, L.camq, cd, ipc, S, gred, hd, gs);
  VC_STAMP(3);
  if (v.imu_on) pinned_rows_phase(v, S, gred, hd, gs);
}

// ------------------------------------------------------------------------------------------ reduced solve
// Phase B.  Damped Cholesky of the augmented matrix [S + Lambda, g; g^T, .] (the forward substitution
// rides along as the extra row), back substitution, then the trial state of the shared parameters
// (cams[1-cur] <- Plus(cams[cur], delta_s)) and their scalar terms scal[8..15].
//   D <= 32: one wavefront, rows in registers, pivots exchanged with v_readlane (no barriers);
//   D  > 32: workgroup-wide in LDS.
// Lane i keeps row i of the augmented matrix in registers (static indices: all loops over columns are
// unrolled to DMAX), pivots and pivot columns travel through v_readlane; the factor goes to LDS (Lt: DMAX x ldt) for the
// back-substitution only.  x: D.
// (the arguments by value: a function that is not inlined and takes the DevView by reference makes its caller keep a copy of the
//  whole record in scratch memory -- 1 KB per lane and a slower launch)
struct SmallSolveArgs { int D; double* sscale2; double* sdiag; double* slam; int* bad_flag; long long* dbg; };
#ifndef VC_SMALL_SOLVE_BCAST
#define VC_SMALL_SOLVE_BCAST 1      // (0: pivot columns through v_readlane, the form of rounds 2-5 -- A/B builds)
#endif
template <int DMAX>
__device__ __forceinline__ void solve_small_wave(const SmallSolveArgs v, const Ctrl* ct, int lane, double* Lt, double* x, double pre_sc2, double pre_dg, const double* S /* v.Sbuf or its LDS image */,
                                                 double* bc /* LDS, 2 x 64: the pivot column's broadcast image */) {
  const int D = v.D;
  constexpr int ldt = DMAX + 2;
  const double* gred = S + D * D;
  const double* hd = gred + D;
#ifdef VC_REDUCED_STAMPS
  if (lane == 0) s_rst[8] = (long long)__builtin_readcyclecounter();
#endif
  double row[DMAX];
  {
    // one base address per lane (row `lane` of S; g_red for lane D and, harmlessly, beyond), the column as an immediate offset
    const double* rp = lane < D ? S + lane * D : gred;
#pragma unroll
    for (int k = 0; k < DMAX; ++k) { double t = 0.0; if (k < D) t = rp[k]; row[k] = lane <= D ? t : 0.0; }
  }
#if VC_SMALL_SOLVE_BCAST
  double diag = lane < D ? S[lane * D + lane] : 0.0;      // row `lane`'s diagonal entry, kept apart (see the factorisation below)
#endif
  double lam = 0.0;
  if (lane < D) {
    double sc2, dg;
    // pre_sc2 / pre_dg: v.sscale2[lane] / v.sdiag[lane], requested at kernel entry (only read when they are valid)
    if (ct->init_scale) { sc2 = jacobi_scale2(hd[lane]); v.sscale2[lane] = sc2; } else sc2 = pre_sc2;
    if (!ct->reuse_diag) { dg = lm_clamped_diag(hd[lane], sc2); v.sdiag[lane] = dg; } else dg = pre_dg;
    lam = dg / (ct->radius * sc2);
    v.slam[lane] = lam;
    // the tail's inputs ride along in LDS behind x: damping and g_s of this column
    x[kSmallD + 1 + lane] = lam;
    x[2 * (kSmallD + 1) + lane] = hd[D + lane];
  }
#pragma unroll
  for (int k = 0; k < DMAX; ++k) row[k] += (k == lane) ? lam : 0.0;
#if VC_SMALL_SOLVE_BCAST
  diag += lam;
#endif
#ifdef VC_REDUCED_STAMPS
#define VC_SS(i) do { __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_sched_barrier(0); if (lane == 0) s_rst[8 + (i)] = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define VC_SS(i) do { } while (0)
#endif
  VC_SS(1);
  // Every loop below is fully unrolled (j, k compile-time): register indices are static, pivots and pivot columns travel
  // as scalars through v_readlane -- no LDS round trip and no select chains inside the dependent chain.  The column loop
  // has no branch and no store (one scheduling region): the trailing updates of column j fill the latency of column
  // j + 1's reciprocal square root.  Columns >= D only ever hold zeros (or, for column D, unused values): their steps run
  // on harmless operands (pivot forced to 1) and are not checked, so the loop needs no bound on j or k.
  bool bad = false;
#if VC_SMALL_SOLVE_BCAST
  // Round 6.  What a pivot costs is its DEPENDENT chain, and on a lone wavefront every link is ~10 cycles (a trip through LDS ~130): the
  // v_readlane form paid 2 v_readlane + SGPR hazard + FMA per entry of the trailing update (~460 cycles per pivot at D = 29), and a first
  // LDS-broadcast form put the LDS round trip of column j + 1 between pivot j and pivot j + 1 (~350).  Now:
  //   * the DIAGONAL entry of row i lives in a register of its own (`diag`): it only ever needs the lane's own l_ij, so the next pivot is
  //     one FMA and one v_readlane behind l_ij, and its reciprocal square root (v_rsq_f64 + two Newton steps; the pivot's sign is tested
  //     beside it, not ahead of it) starts at once;
  //   * column j + 1 -- the only one pivot j + 1 waits for -- takes l_(j+1)j through v_readlane;
  //   * every other column takes the pivot column from an LDS broadcast image (one ds_write_b64 per lane, two entries per wave-uniform read)
  //     and applies it ONE PIVOT LATE, behind the next pivot's chain: the LDS latency is never waited for.
  // A row's updates arrive in another order than in the v_readlane form (column j + 1: pivot j ahead of pivot j - 1); no lane select for the
  // pivot's own row: row j's entry j is the pivot itself (same FMAs as `diag`).
  double ipiv;
  {
    const double d0 = readlane_f64(diag, 0);
    const bool ok = d0 > 0.0;
    bad |= (0 < D) & !ok;
    const double y = fast_rsqrt(d0);
    ipiv = ok ? y : 1.0;
  }
  double lprev = 0.0, bprev[DMAX];
#pragma unroll
  for (int k = 0; k < DMAX; ++k) bprev[k] = 0.0;
#pragma unroll
  for (int j = 0; j < DMAX; ++j) {
    const double lij = row[j] * ipiv;
    row[j] = lij * ipiv;      // what the back-substitution wants: column j of L over its diagonal entry (off the dependent chain)
    if (j + 1 < DMAX) {
      diag -= lij * lij;
      const double dn = readlane_f64(diag, j + 1);
      double bnew[DMAX];
      if (j + 2 < DMAX) {
        double* col = bc + (j & 1) * 64;
        col[lane] = lij;
        wave_lds_sync_local();
#pragma unroll
        for (int k = j + 2; k < DMAX; ++k) bnew[k] = col[k];
      }
      __builtin_amdgcn_sched_barrier(0);      // (the reads are ISSUED here -- the scheduler otherwise sinks them, and the store, to their uses one pivot later)
      const bool ok = dn > 0.0;
      bad |= (j + 1 < D) & !ok;
      const double y = fast_rsqrt(dn);      // (of a pivot <= 0: inf / NaN, dropped by the select)
      const double ipn = ok ? y : 1.0;
      row[j + 1] -= lij * readlane_f64(lij, j + 1);
      // pivot j - 1's column, read one pivot ago: columns j + 1 ..
#pragma unroll
      for (int k = j + 1; k < DMAX; ++k) row[k] -= lprev * bprev[k];
      lprev = lij;
#pragma unroll
      for (int k = j + 2; k < DMAX; ++k) bprev[k] = bnew[k];
      ipiv = ipn;
    }
  }
#else
#pragma unroll
  for (int j = 0; j < DMAX; ++j) {
    const double mine = row[j];
    double d = readlane_f64(mine, j);
    const bool ok = d > 0.0;
    bad |= (j < D) & !ok;
    d = ok ? d : 1.0;
    const double ipiv = fast_rsqrt(d);
    const double lij = (lane == j) ? d * ipiv : mine * ipiv;
    row[j] = lij * ipiv;      // what the back-substitution wants: column j of L over its diagonal entry (off the dependent chain)
#pragma unroll
    for (int k = j + 1; k < DMAX; ++k) row[k] -= lij * readlane_f64(lij, k);
  }
#endif
  VC_SS(2);
  if (bad && lane == 0) *v.bad_flag = 1;
  // column j of L / L_jj, contiguous over the rows (entries above the diagonal / beyond row D are never read; lanes past the
  // padded row length all write the last padding slot)
  {
    const int slot = lane < ldt - 1 ? lane : ldt - 1;
#pragma unroll
    for (int j = 0; j < DMAX; ++j) Lt[j * ldt + slot] = row[j];
  }
  // delta_s = -L^-T y.  Lt row i is column i of L over L_ii: c_ij = L[j][i] / L[i][i] for j > i, and y_i / L[i][i] at index D:
  // x_i = -y_i / L_ii - sum_{j > i} c_ij x_j.  Lane i keeps its running sum s; the unknowns are resolved four at a time: the four
  // lanes' sums are broadcast (v_readlane, independent of each other), every lane solves the 4 x 4 triangle itself (its six
  // coefficients are wave-uniform LDS reads, requested ahead) and applies the four unknowns to its sum -- one scalar round trip
  // per four unknowns on the dependent chain instead of one per unknown.  A lane inside the block ends up with its own unknown:
  // the terms the broadcast value was still missing are exactly the ones the update adds (c_ij = 0 for j <= i).
  wave_lds_sync();
  VC_SS(3);
  double c[DMAX];
#pragma unroll
  for (int j = 0; j < DMAX; ++j) c[j] = (j < D && lane < D && j > lane) ? Lt[lane * ldt + j] : 0.0;
  double s = (lane < D) ? -Lt[lane * ldt + D] : 0.0;
#pragma unroll
  for (int hi = DMAX; hi > 0; hi -= 4) {
    const int lo = hi >= 4 ? hi - 4 : 0, n = hi - lo;
    double t[4], xb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) if (q < n) t[q] = readlane_f64(s, lo + q);
#pragma unroll
    for (int q = 3; q >= 0; --q)
      if (q < n) {
        double a = t[q];
#pragma unroll
        for (int r = 3; r > q; --r) if (r < n) a -= ((lo + r < D) ? Lt[(lo + q) * ldt + lo + r] : 0.0) * xb[r];      // oldest unknown first
        xb[q] = a;
      }
#pragma unroll
    for (int q = 0; q < 4; ++q) if (q < n) s -= c[lo + q] * xb[q];
  }
  if (lane < D) x[lane] = s;
  VC_SS(4);
}

// Workgroup-wide factorisation for D > 32.  Only the lower triangle (plus the right-hand-side row D) is kept, packed:
// row i starts at i (i + 1) / 2 -- 129 KB of LDS at D = 178 (8 cameras + IMU + 7 separators) instead of 256 KB.
__device__ __forceinline__ int tri(int i) { return (i * (i + 1)) >> 1; }
// Factorisation: register-tiled, one barrier per two columns (factor_large_tiled below); back-substitution: panels of 16 columns from the
// bottom -- partial sums over the rows below on the 16 x 16 thread grid, the 16 x 16 diagonal block by one wavefront (v_readlane).
// Same packed storage; extra LDS after x: two column images (2 x 192 doubles each, 16-byte aligned), dinv (D).
// (Rounds 3-6 factored in panels of 16 too: diagonal block on one wavefront -- ~400 cycles per pivot --, rows below one per thread, trailing
//  update on the matrix pipe, three barriers per panel; the register-tiled form is 18-20 % faster end to end at D = 67 and D = 115
//  (profiles/r06_ab_reduced_tiled.txt) and a third of the code.  Also measured there and not kept: the diagonal blocks inverted up front so that a
//  panel of the back-substitution is a product -- the inversion costs what the sixteen-step chains did.)
#ifdef VC_REDUCED_STAMPS
#define VC_PH(i) do { if (threadIdx.x == 0) { const long long now_ = (long long)__builtin_readcyclecounter(); ph_[i] += now_ - t_; t_ = now_; } } while (0)
#else
#define VC_PH(i) do { } while (0)
#endif
// (round 6, last part) Register-tiled right-looking Cholesky of the (D + 1) x (D + 1) system [S + damping; g_red^T] for D > 32.  The 256 threads
// form a 16 x 16 grid (tr, tc); thread (tr, tc) OWNS the entries (tr + 16 a, tc + 16 b), b <= a, of the lower triangle and keeps them in
// registers from the load (straight from Sbuf: one memory round trip for the whole matrix, no LDS image of the unfactored system) to the
// end -- NT (NT + 1) / 2 doubles, NT = ceil((D + 1) / 16) <= 12.  A step (two columns, see below) costs ONE workgroup barrier: the columns'
// owners put their unscaled entries into an LDS column image (two images, alternating), everybody reads the pivot block and its own rows' and
// columns' entries, forms the reciprocals itself and updates its tile -- independent FMAs, the block column jb a compile-time constant, so
// that finished block rows / columns cost nothing.  The scaled columns go into the packed triangle M (LDS) -- the factor the
// back-substitution below reads, in the layout the panel form left it in.  The panel form's critical path was one wavefront's
// 16 x 16 factor per panel (~400 cycles per pivot: that wavefront issues every instruction of the step), a row solve and a trailing update
// behind barriers of their own: 75 k of k_reduced's 110 k cycles at D = 67, 170 k of 218 k at D = 115; this form: 51 k / 99 k -- a pair of
// columns is still ~1300 cycles: LDS write -> barrier -> LDS read -> reciprocals -> FMA is a chain of ~130-cycle hops and ~40-cycle
// dependent fp64 operations that four lone wavefronts cannot hide.
template <int NT> __device__ __forceinline__ constexpr int tix(int a, int b) { return a * (a + 1) / 2 + b; }
#ifdef VC_REDUCED_STAMPS
// (profiling builds: the phases of the step of columns 20 / 21, thread 0 -- every stamp behind a full wait, so that it reads when the phase's results exist)
#define VC_MS(i) do { if (j == 20 && tid == 0) { __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_sched_barrier(0); s_rst[20 + (i)] = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define VC_MSV(i, val) do { if (j == 20 && tid == 0) { const int sink_ = __builtin_amdgcn_readfirstlane(__double2hiint(val)); asm volatile("" :: "s"(sink_)); s_rst[20 + (i)] = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define VC_MS(i) do { } while (0)
#define VC_MSV(i, val) do { } while (0)
#endif
template <int NT>
__device__ __forceinline__ void factor_large_tiled(const DevView& v, const Ctrl* ct, double* M, double* x, double* cb0, double* cb1, double* dinv) {
  const int tid = threadIdx.x, tr = tid >> 4, tc = tid & 15, D = v.D;
  const double* S = v.Sbuf;
  const double* hd = S + D * D + D;
  double T[NT * (NT + 1) / 2];
  // rows 0 .. D - 1: S's lower triangle; row D: g_red (which follows S in Sbuf: entry (D, k) is S[D D + k]); everything else zero
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b <= a; ++b) {
      const int i = tr + 16 * a, k = tc + 16 * b;
      const bool in = i <= D && k <= i && k < D;
      const double t = S[in ? i * D + k : 0];
      T[tix<NT>(a, b)] = in ? t : 0.0;
    }
  for (int i = tid; i < D; i += 256) {
    double sc2, dg;
    if (ct->init_scale) { sc2 = jacobi_scale2(hd[i]); v.sscale2[i] = sc2; } else sc2 = v.sscale2[i];
    if (!ct->reuse_diag) { dg = lm_clamped_diag(hd[i], sc2); v.sdiag[i] = dg; } else dg = v.sdiag[i];
    const double lam = dg / (ct->radius * sc2);
    v.slam[i] = lam;
    x[i] = lam;
  }
  __syncthreads();
  if (tr == tc) {
#pragma unroll
    for (int a = 0; a < NT; ++a) { const int i = tr + 16 * a; const double lam = x[i < D ? i : 0]; T[tix<NT>(a, a)] += i < D ? lam : 0.0; }
  }
#ifdef VC_REDUCED_STAMPS
  if (tid == 0) { const int sink_ = __builtin_amdgcn_readfirstlane(__double2hiint(T[0])); asm volatile("" :: "s"(sink_)); s_rst[14] = (long long)__builtin_readcyclecounter(); }
#endif
  // Column steps, TWO columns per barrier.  No masks: an entry whose row or column is <= j + 1 is never read again -- whatever the step does
  // to it stays in entries nobody uses (the upper halves of the diagonal tiles included).  The owners of columns j and j + 1 put their UNSCALED
  // entries a_i, b_i (b not yet updated by column j) side by side into the column image; everybody reads the 2 x 2 pivot block [p; m, b_q] and
  // its own rows' and columns' pairs (one 16-byte LDS read each) and does the small factorisation itself: r1 = 1 / p, b'_i = b_i - a_i (m r1),
  // q = b_q - m (m r1), r2 = 1 / q, then T(i, k) -= a_i (a_k r1) + b'_i (b'_k r2) -- the FMAs of two single steps behind ONE write -> barrier ->
  // read chain (a single step is ~830 cycles at D = 67 of which the tile's FMAs are a tenth).  1 / sqrt(p), 1 / sqrt(q), which only the stored
  // factor needs, run beside the reciprocal chains; the factor's two columns go out coalesced, one row per thread (threads 0 .. 16 NT - 1),
  // straight from the column image.  An odd last column is one single step of the same form.
  // (first cut: one column per barrier, both operands scaled by 1 / sqrt(c_j) behind selects, the owners storing the factor's column through
  //  tri(i): 62 k cycles for the factorisation at D = 67 (75 k in the panel form); without the selects and with coalesced stores: 56 k)
  bool bad = false;
  const int wi = tid < 16 * NT ? tid : 16 * NT - 1, triw = tri(wi);
  typedef double __attribute__((ext_vector_type(2))) v2d_;
#pragma unroll
  for (int jb = 0; jb < NT; ++jb) {
    const int jn = min(16, D - 16 * jb);            // (columns of this block column; <= 0 behind the matrix)
    const int npair = jn > 0 ? jn >> 1 : 0;
#pragma nounroll
    for (int pc = 0; pc < npair; ++pc) {
      const int j = 16 * jb + 2 * pc;
      double* cb = (pc & 1) ? cb1 : cb0;             // (eight pairs per full block column: the parity carries over)
      VC_MS(0);
      if ((tc >> 1) == pc) {
        const int sl = tc & 1;
#pragma unroll
        for (int a = jb; a < NT; ++a) cb[2 * (tr + 16 * a) + sl] = T[tix<NT>(a, jb)];
      }
      VC_MS(1);
      __syncthreads();
      VC_MS(2);
      const v2d_* cb2 = reinterpret_cast<const v2d_*>(cb);
      const v2d_ pv0 = cb2[j], pv1 = cb2[j + 1], cw = cb2[wi];
      v2d_ ri[NT], ck[NT];
#pragma unroll
      for (int b = jb; b < NT; ++b) ck[b] = cb2[tc + 16 * b];
#pragma unroll
      for (int a = jb; a < NT; ++a) ri[a] = cb2[tr + 16 * a];
      const double p = pv0.x, m = pv1.x, bq = pv1.y;
      const bool ok1 = p > 0.0;
      const double ps = ok1 ? p : 1.0;              // (a pivot that is not positive: identity column, the pass is flagged -- as the panel form)
      VC_MSV(3, ri[NT - 1].x);
      const double r1 = fast_rcp(ps), ip1 = fast_rsqrt(ps);
      const double mr = m * r1;
      const double q = bq - m * mr;
      const bool ok2 = q > 0.0;
      const double qs = ok2 ? q : 1.0;
      const double r2 = fast_rcp(qs), ip2 = fast_rsqrt(qs);
      bad |= !ok1 || !ok2;
      VC_MSV(4, r2);
#pragma unroll
      for (int b = jb; b < NT; ++b) { const double bk = ck[b].y - ck[b].x * mr; ck[b].x *= r1; ck[b].y = bk * r2; }
#pragma unroll
      for (int a = jb; a < NT; ++a) ri[a].y -= ri[a].x * mr;
#pragma unroll
      for (int a = jb; a < NT; ++a)
#pragma unroll
        for (int b = jb; b <= a; ++b) T[tix<NT>(a, b)] -= ri[a].x * ck[b].x + ri[a].y * ck[b].y;
      VC_MSV(5, T[tix<NT>(NT - 1, NT - 1)]);
      // the factor's columns j, j + 1 below the diagonal (row D: the forward-substituted right-hand side) and the two diagonal entries
      if (tid > j && tid <= D) M[triw + j] = cw.x * ip1;
      if (tid > j + 1 && tid <= D) M[triw + j + 1] = (cw.y - cw.x * mr) * ip2;
      if (tid == j) { M[triw + j] = ok1 ? p * ip1 : 1.0; dinv[j] = ip1; }
      if (tid == j + 1) { M[triw + j + 1] = ok2 ? q * ip2 : 1.0; dinv[j + 1] = ip2; }
      VC_MS(6);
    }
    if (jn > 0 && (jn & 1)) {                        // the matrix's last column (D odd): a single step
      const int jc = jn - 1, j = 16 * jb + jc;
      double* cb = (npair & 1) ? cb1 : cb0;
      if (tc == jc) {
#pragma unroll
        for (int a = jb; a < NT; ++a) cb[tr + 16 * a] = T[tix<NT>(a, jb)];
      }
      __syncthreads();
      const double cj = cb[j], cw = cb[wi];
      double ci[NT], ck[NT];
#pragma unroll
      for (int b = jb; b < NT; ++b) ck[b] = cb[tc + 16 * b];
#pragma unroll
      for (int a = jb; a < NT; ++a) ci[a] = cb[tr + 16 * a];
      const bool ok = cj > 0.0;
      bad |= !ok;
      const double cjs = ok ? cj : 1.0;
      const double r = fast_rcp(cjs), ipiv = fast_rsqrt(cjs);
#pragma unroll
      for (int b = jb; b < NT; ++b) ck[b] *= r;
#pragma unroll
      for (int a = jb; a < NT; ++a)
#pragma unroll
        for (int b = jb; b <= a; ++b) T[tix<NT>(a, b)] -= ci[a] * ck[b];
      if (tid > j && tid <= D) M[triw + j] = cw * ipiv;
      if (tid == j) { M[triw + j] = ok ? cj * ipiv : 1.0; dinv[j] = ipiv; }
    }
  }
  if (bad && tid == 0) v.flags[5 + 2 * v.par] = 1;
  __syncthreads();
}
__device__ __forceinline__ void solve_large_tiled(const DevView& v, const Ctrl* ct, double* M, double* x) {
  const int tid = threadIdx.x, lane = tid & 63, D = v.D;
#ifdef VC_REDUCED_STAMPS
  long long ph_[6] = {0, 0, 0, 0, 0, 0}, t_ = (long long)__builtin_readcyclecounter();      // - | load + factorisation | - | - | back-subst sums | back-subst solve
#endif
  double* Lp = x + (D + 1);                      // column image 0 of the factorisation: two columns side by side, 2 x 192 doubles, 16-byte aligned
  if (reinterpret_cast<uintptr_t>(Lp) & 8) ++Lp;
  double* red = Lp + 384;                        // column image 1; the back-substitution's 16 x 16 partial sums
  double* dinv = red + 384;
  {
    const int nT = (D + 1 + 15) >> 4;
    switch (nT) {
      case 3: factor_large_tiled<3>(v, ct, M, x, Lp, red, dinv); break;
      case 4: factor_large_tiled<4>(v, ct, M, x, Lp, red, dinv); break;
      case 5: factor_large_tiled<5>(v, ct, M, x, Lp, red, dinv); break;
      case 6: factor_large_tiled<6>(v, ct, M, x, Lp, red, dinv); break;
      case 7: factor_large_tiled<7>(v, ct, M, x, Lp, red, dinv); break;
      case 8: factor_large_tiled<8>(v, ct, M, x, Lp, red, dinv); break;
      case 9: factor_large_tiled<9>(v, ct, M, x, Lp, red, dinv); break;
      case 10: factor_large_tiled<10>(v, ct, M, x, Lp, red, dinv); break;
      case 11: factor_large_tiled<11>(v, ct, M, x, Lp, red, dinv); break;
      default: factor_large_tiled<12>(v, ct, M, x, Lp, red, dinv); break;
    }
    VC_PH(1);
  }
  // ---- delta_s = -L^-T y (y = row D), panels from the bottom ----------------------------------------------------------
  for (int i = tid; i < D; i += 256) x[i] = -M[tri(D) + i];
  __syncthreads();
  const int last = ((D - 1) / 16) * 16;
  for (int p0 = last; p0 >= 0; p0 -= 16) {
    const int nb = min(16, D - p0), r0 = p0 + nb;
    {   // t_j = x_j - sum_{i >= r0} L[i][p0 + j] x_i : 16 columns x 16 partial sums
      const int jcol = tid & 15, part = tid >> 4;
      double acc = 0.0;
      if (jcol < nb) for (int i = r0 + part; i < D; i += 16) acc += M[tri(i) + p0 + jcol] * x[i];
      red[part * 16 + jcol] = acc;
    }
    __syncthreads();
    VC_PH(4);
    if (tid < 64) {
      double t = 0.0;
      if (lane < nb) {
        t = x[p0 + lane];
#pragma unroll
        for (int q = 0; q < 16; ++q) t -= red[q * 16 + lane];
      }
      const double di = (lane < nb) ? dinv[p0 + lane] : 1.0;
      // the lane's column of the diagonal block, requested before the dependent chain (round 6: as a read inside every step the chain was an
      // LDS round trip per unknown -- 3.7k cycles per panel; now a multiply, two v_readlane and an FMA)
      // (the unknowns four at a time, as solve_small_wave -- every lane solving the 4 x 4 triangle itself from wave-uniform LDS reads -- was
      //  measured here and is slower: 56 LDS reads per panel instead of 33, panels 3.4 k -> 5.3 k cycles)
      double Lc[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) { const bool in = j < nb && lane < j; const double t = M[in ? tri(p0 + j) + p0 + lane : 0]; Lc[j] = in ? t : 0.0; }
      __builtin_amdgcn_sched_barrier(0);      // (the reads are issued here, not sunk to their uses)
      double z = 0.0;
#pragma unroll
      for (int j = 15; j >= 0; --j) {
        if (j < nb) {                                  // wave-uniform
          const double zj = readlane_f64(t * di, j);
          if (lane == j) z = zj;
          t -= Lc[j] * zj;
        }
      }
      if (lane < nb) x[p0 + lane] = z;
    }
    __syncthreads();
    VC_PH(5);
  }
#ifdef VC_REDUCED_STAMPS
  if (threadIdx.x == 0) for (int i = 0; i < 6; ++i) s_rst[8 + i] = ph_[i];
#endif
}

__device__ __forceinline__ void reduced_solve_phase(const DevView& v, const Ctrl* ct, double* dyn, double* red /* 6 x 256 */, const double* s_cam,
                                    double pre_sc2, double pre_dg, const double* x2_noobs, const CamDesc* cd /* LDS copy of v.cd */,
                                    const double* Sb /* v.Sbuf or its LDS image */, double pre_imu, const int* ipc /* LDS copy of v.imu_param_col */) {
  const int tid = threadIdx.x, D = v.D, cur = ct->cur;
  double* x;
  VC_STAMP(4);
  if (D <= kSmallD) {
    x = dyn + (kSmallD + 1) * (kSmallD + 2);
    if (tid < 64 && D > 0) {
      // (every column up to DMAX is eliminated whether the matrix has it or not: instances close to the common widths)
      const SmallSolveArgs a = {D, v.sscale2, v.sdiag, v.slam, v.flags + 5 + 2 * v.par, v.dbg};
      if (D <= 16) solve_small_wave<16>(a, ct, tid, dyn, x, pre_sc2, pre_dg, Sb, red);
      else if (D <= 24) solve_small_wave<24>(a, ct, tid, dyn, x, pre_sc2, pre_dg, Sb, red);
      else if (D <= 28) solve_small_wave<28>(a, ct, tid, dyn, x, pre_sc2, pre_dg, Sb, red);
      else if (D <= 30) solve_small_wave<30>(a, ct, tid, dyn, x, pre_sc2, pre_dg, Sb, red);
      else solve_small_wave<32>(a, ct, tid, dyn, x, pre_sc2, pre_dg, Sb, red);
    }
    __syncthreads();
  } else {
    x = dyn + tri(D + 1) + (D + 1);
    solve_large_tiled(v, ct, dyn, x);
  }
  VC_STAMP(5);
  const bool small = D <= kSmallD;      // the one-wavefront solve left damping and g_s in LDS behind x
  const double* gs = small ? x + 2 * (kSmallD + 1) : v.Sbuf + (size_t)D * D + 2 * D;
  const double* lamv = small ? x + (kSmallD + 1) : v.slam;
  if (v.tail_deferred) {
    // (round 6) the back-substitution's launch carries the rest (reduced_tail, one extra workgroup): this kernel -- one workgroup the whole
    // chip waits for -- ends with the step and the trial IMU parameters (which the second stream's k_imu_block(trial) starts from)
    for (int i = tid; i < D; i += 256) v.delta_s[i] = x[i];
    if (v.imu_on && (tid >> 6) == (D <= 64 ? 0 : 1) && (tid & 63) < 16) {
      const int a = tid & 63, col = a < 15 ? ipc[a] : -1;
      v.imus[1 - cur][a] = pre_imu + (col >= 0 ? x[col] : 0.0);      // (g(2) b(6) sf(6) toff(1): plain additive parameters; entry 15: padding)
    }
    return;
  }
  reduced_tail(v, cur, x, gs, lamv, s_cam, cd, ipc, pre_imu, x2_noobs, red, true, true);
}

// mode 0: phase A + phase B in one launch (single process); 1: phase A only (an all-reduce of Sbuf follows);
// 2: phase B only
__global__ __launch_bounds__(256) void k_reduced(DevView v, int mode) {
  extern __shared__ __attribute__((aligned(16))) double dyn[];   // phase A: FinalLds; phase B: the matrix (the phases do not overlap)
  __shared__ double red[6 * 256];
  __shared__ double s_cam[kMaxCams * kCamStride];     // accepted camera records: requested at kernel entry, used by the tail
  __shared__ double s_x2;
  __shared__ CamDesc s_cd[kMaxCams];     // the kernel-argument table costs a scalar memory round trip per (dynamically indexed) access
  __shared__ int s_ipc[16];              // ... and so does DevView::imu_param_col (as 15 s_load_dword, one per use, it was 3.6k cycles of the tail)
  const Ctrl* ct = v.ctrl;
  __shared__ double s_S[kSmallD * kSmallD + 3 * kSmallD + 2];      // single process, D <= kSmallD: the reduced system stays in LDS between the phases
  // early Gram: [Y | z] rows of the chain's top-level frames, behind phase A's record in the dynamic region (static LDS is at its limit
  // where the packed matrix of a D = 178 system takes 137 KB of the dynamic one)
  double* s_top = dyn + (sizeof(FinalLds) + 7) / 8;
  const bool early = v.gram_top_stride > 0 && mode == 0 && v.D <= kEarlyTopD;
  const int top_rows = early ? 9 * min(7, (v.n_frames - 1) / v.gram_top_stride + 1) : 0;
  // the top-level frames' rows (final since the launch before the partial sums): requested first -- together with the control record,
  // not behind it --, their latency under everything below
  double tin[9];
  const double* top_src = early ? v.cW : v.Sbuf;      // (no chain: every lane reads the always-present first entry of Sbuf and drops it)
#pragma unroll
  for (int u = 0; u < 9; ++u) {
    const int idx = threadIdx.x + 256 * u, r = idx / kTopLd, c = idx - r * kTopLd;
    const bool in = early && idx < 64 * kTopLd && r < top_rows && c <= v.D;
    const double x = top_src[in ? ((size_t)(r / 9) * v.gram_top_stride * 9 + (r % 9)) * v.ldx + c : 0];
    tin[u] = in ? x : 0.0;
  }
  if (ct->done) { if (threadIdx.x == 0 && mode != 1) signal_flag(v, 1); return; }
  if (threadIdx.x < kMaxCams) s_cd[threadIdx.x] = v.cd[threadIdx.x];
  if (threadIdx.x >= 64 && threadIdx.x < 64 + 15) s_ipc[threadIdx.x - 64] = v.imu_param_col[threadIdx.x - 64];
  if (early) {
#pragma unroll
    for (int u = 0; u < 9; ++u) { const int idx = threadIdx.x + 256 * u; if (idx < 64 * kTopLd) s_top[idx] = tin[u]; }
  }
  double pre_sc2 = 1.0, pre_dg = 1.0;      // damping inputs of the small solve: requested now, consumed after phase A
  double pre_imu = 0.0;                    // accepted IMU parameters (lane a of the wavefront whose thread forms the trial ones): the tail's
  if (mode != 1) {
    for (int i = threadIdx.x; i < v.n_cams * kCamStride; i += 256) s_cam[i] = v.cams[ct->cur][i];
    if (v.D <= kSmallD && (int)threadIdx.x < v.D) { pre_sc2 = v.sscale2[threadIdx.x]; pre_dg = v.sdiag[threadIdx.x]; }
    if (v.imu_on && (int)(threadIdx.x >> 6) == (v.D <= 64 ? 0 : 1) && (threadIdx.x & 63) < 16) pre_imu = v.imus[ct->cur][threadIdx.x & 63];
  }
  const bool s_in_lds = mode == 0 && v.D <= kSmallD;
  if (s_in_lds) {
    schur_final_phase(v, ct->cur, *reinterpret_cast<FinalLds*>(dyn), &s_x2, s_cd, s_ipc, s_S, true, early ? s_top : nullptr, top_rows);
    __syncthreads();
    // Sbuf keeps its meaning for k_final (cost slot) and the parity hooks: written back off the critical path
    for (int e = threadIdx.x; e < v.D * v.D + 3 * v.D + 2; e += 256) v.Sbuf[e] = s_S[e];
  } else if (mode != 2) { schur_final_phase(v, ct->cur, *reinterpret_cast<FinalLds*>(dyn), &s_x2, s_cd, s_ipc, v.Sbuf, false); __syncthreads(); }
  if (mode != 1) reduced_solve_phase(v, ct, dyn, red, s_cam, pre_sc2, pre_dg, mode == 0 ? &s_x2 : nullptr, s_cd, s_in_lds ? s_S : v.Sbuf, pre_imu, s_ipc);
  if (mode != 1) { __syncthreads(); if (threadIdx.x == 0) signal_flag(v, 1); }      // the trial IMU parameters exist (stream B's deltas wait for this)
#ifdef VC_REDUCED_STAMPS
  VC_STAMP(7);
  __syncthreads();
  if (threadIdx.x < 32) v.dbg[threadIdx.x] = s_rst[threadIdx.x];
#endif
}

// ------------------------------------------------------------------------------------------ trial point
// One wavefront per tile: delta_p = -L^-T (z + sum_tiles Y delta_s) (lanes = (tile, column), butterfly sum),
// T_trial = T exp(delta_p), residual sweep of the tile at the trial state.  The first tile of a frame
// also publishes the frame's trial pose and its step terms.
__device__ void final_phase(const DevView& v, int mode, double* red /* 7 x 256 */);
// Back-substitution of one frame: delta_p = -L^-T (z + sum_tiles Y delta_s) (lanes = (tile, column), six-value butterfly),
// T_trial = T exp(delta_p).  With `publish` lane 0 stores the trial pose and the frame's step terms (fpart).
__device__ __forceinline__ void backsub_frame(const DevView& v, int cur, int f, int lane, const double* ds_s, double* Tout, bool publish,
                                              double* wsum, const TileHdr* hdr = nullptr /* k_trial: the tile's header (registers) */) {
  const int t0 = hdr ? hdr->t0 : v.frame_tile_off[f], nt = hdr ? hdr->nt : v.frame_tile_off[f + 1] - t0;
  const double* fr = v.fr + (size_t)f * kFrStride;
  const double* pin = v.poses[cur] + (size_t)f * kPoseStride;
  double Lr[21], zr[6], di[6], Tin[7];
#pragma unroll
  for (int i = 0; i < 21; ++i) Lr[i] = fr[kFrL + i];
#pragma unroll
  for (int i = 0; i < 6; ++i) { zr[i] = fr[kFrZ + i]; di[i] = fr[kFrDinv + i]; }
#pragma unroll
  for (int i = 0; i < 7; ++i) Tin[i] = pin[i];
  double y[6] = {0, 0, 0, 0, 0, 0};
  const bool small_d = v.D <= kMaxCams * 16 + 16;
  for (int idx = lane; idx < nt * 16; idx += 64) {
    const int t = idx >> 4, j = idx & 15;
    if (hdr) {
      // columns and widths of the frame's tiles come with the header: the Y loads depend on nothing but t0 and go out
      // together with the frame record (Y's padding columns are zero and take a zero step)
      const int col0 = (int)((hdr->col0 >> (8 * t)) & 0xff), nc = (int)((hdr->ncols >> (8 * t)) & 0xff);
      const double* Yt = v.Y + (size_t)(t0 + t) * kYStride + j;
      double yv[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) yv[k] = Yt[k * kUCols];
      const int col = (j < nc) ? col0 + j : 0;
      double dj = small_d ? ds_s[col] : v.delta_s[col];
      dj = (j < nc) ? dj : 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) y[k] += yv[k] * dj;
    } else {
      const CamDesc cd = v.cd[v.tile_cam[t0 + t]];
      if (j < cd.ncols) {
        const double dj = small_d ? ds_s[cd.col0 + j] : v.delta_s[cd.col0 + j];
        const double* Yt = v.Y + (size_t)(t0 + t) * kYStride + j;
#pragma unroll
        for (int k = 0; k < 6; ++k) y[k] += Yt[k * kUCols] * dj;
      }
    }
  }
  double L[36];
  {
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) L[i * 6 + j] = (j <= i) ? Lr[k++] : 0.0;
  }
  {
    double ys[6];
    wave_sum6(y, ys, lane);
#pragma unroll
    for (int k = 0; k < 6; ++k) y[k] = ys[k] + zr[k];
  }
  bwd_solve_inv<6>(L, di, y);
  double d[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) d[i] = -y[i];
  se3_plus(Tin, d, Tout);
  if (publish && lane == 0) {
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
    if (wsum) { wsum[kScGd] = gd; wsum[kScDld] = dld; wsum[kScStep2] = step2; wsum[kScX2] = x2; wsum[kScG2] = g2; wsum[kScGmax] = gmax; }
  }
}
// Large problems (more tiles than the chip holds waves): the back-substitution runs once per frame here instead of once
// per tile inside k_trial.
__global__ __launch_bounds__(256) void k_backsub(DevView v) {
  __shared__ double ds_s[kMaxCams * 16 + 16];
  const Ctrl* ct = v.ctrl;
  if (ct->done) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < v.D; i += 256) if (i < kMaxCams * 16 + 16) ds_s[i] = v.delta_s[i];
  __syncthreads();
  const int f = blockIdx.x * 4 + wave;
  if (f >= v.n_frames || v.frame_tile_off[f + 1] == v.frame_tile_off[f]) return;
  double Tout[7];
  backsub_frame(v, ct->cur, f, lane, ds_s, Tout, true, nullptr);
}
template <bool FUSED>
__device__ __forceinline__ void trial_tile(const DevView& v, int cur, double mult, TileHdr h, const double* cams_trial /* LDS */, int tile, int wave,
                                           int lane, const double* ds_s, double* lds_rows,
                                           double* wsum /* kNumScal, lane 0: this wave's step scalars */) {
  // the tile's header: one (wave-uniform) record instead of the chain tile -> frame -> frame's tiles -> cameras -> columns
  h.frame = __builtin_amdgcn_readfirstlane(h.frame); h.cam = __builtin_amdgcn_readfirstlane(h.cam);
  h.t0 = __builtin_amdgcn_readfirstlane(h.t0); h.nt = __builtin_amdgcn_readfirstlane(h.nt);
  h.off = __builtin_amdgcn_readfirstlane(h.off); h.cnt = __builtin_amdgcn_readfirstlane(h.cnt);
  h.model = __builtin_amdgcn_readfirstlane(h.model);
  const int f = h.frame, c = h.cam;
  const int t0 = h.t0;
  const double* cam = cams_trial + (size_t)c * kCamStride;
  double camr[kCamK + 10];
#pragma unroll
  for (int i = 0; i < kCamK + 10; ++i) camr[i] = cam[i];
  double Tout[7];
  if (v.pre_backsub) {            // k_backsub has been here: take the frame's trial pose and (first tile) its step terms
    const double* pt = v.poses[1 - cur] + (size_t)f * kPoseStride;
#pragma unroll
    for (int i = 0; i < 7; ++i) Tout[i] = pt[i];
    if (lane == 0 && tile == t0) {
      const double* o = v.fpart + (size_t)f * kNumScal;
      wsum[kScGd] = o[kScGd]; wsum[kScDld] = o[kScDld]; wsum[kScStep2] = o[kScStep2]; wsum[kScX2] = o[kScX2]; wsum[kScG2] = o[kScG2];
      wsum[kScGmax] = o[kScGmax];
    }
  } else {
    backsub_frame(v, cur, f, lane, ds_s, Tout, tile == t0, wsum, &h);
  }
  double cost, sq = 0.0;
  if (FUSED) {
    // Jacobian sweep at the trial point: its cost is the trial cost, its Gram block is the next linearisation if the
    // step is accepted (k_final flips `cur`); a rejected step leaves Gb[cur] untouched
    cost = jac_tile_dispatch(v, h.model, Tout, camr, mult, tile, lane, lds_rows + wave * 64 * kDotStride,
                             v.Gb[1 - cur] + (size_t)tile * kGPack, h.off, h.cnt);
    if (lane == 0) v.tile_costb[1 - cur][tile] = cost;
  } else {
    TileXf x;
    make_tile_xf(Tout, camr, &x);
    double K[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) K[i] = camr[kCamK + i];
    res_tile_dispatch(v, h.model, x, K, h.off, h.cnt, lane, mult, &cost, &sq);
  }
  if (lane == 0) {
    v.tile_trial[2 * tile] = cost;
    v.tile_trial[2 * tile + 1] = sq;
    wsum[kScCost] = cost;
  }
}
// (Tried and dropped: letting the last workgroup to finish run the final phase.  Device-scope release/acquire fences are
// expensive on this part -- every XCD has its own L2, so each fence writes back / invalidates L2 -- and cost 25 us at
// 250 workgroups; the kernel boundary does the same flush once.)
template <bool FUSED>
__global__ __launch_bounds__(256, 2) void k_trial(DevView v) {
  extern __shared__ __attribute__((aligned(16))) double lds_rows[];   // fused Jacobian sweep: 4 x 64 x kDotStride row images
  __shared__ double ds_s[kMaxCams * 16 + 16];     // delta_s of the workgroup's cameras is read many times: keep it in LDS
  __shared__ double s_cams[2 * kMaxCams * kCamStride];   // both camera buffers: requested before the control record says which is the trial one
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + wave;
  // everything that depends on nothing is requested first -- the tile's header, the shared step, the camera records --
  // then the control record; one memory latency instead of four in a row
  TileHdr h = v.tile_hdr[tile < v.n_tiles ? tile : 0];
  for (int i = threadIdx.x; i < v.D; i += 256) if (i < kMaxCams * 16 + 16) ds_s[i] = v.delta_s[i];
  for (int i = threadIdx.x; i < 2 * v.n_cams * kCamStride; i += 256) {
    const int b = i >= v.n_cams * kCamStride, o = i - b * v.n_cams * kCamStride;
    const double* src = b ? v.cams[1] : v.cams[0];
    s_cams[b * kMaxCams * kCamStride + o] = src[o];
  }
  const Ctrl* ct = v.ctrl;
  const int cur = ct->cur;
  const double mult = ct->mult;
  if (ct->done) return;
  __syncthreads();
  double wsum[kNumScal];
#pragma unroll
  for (int k = 0; k < kNumScal; ++k) wsum[k] = 0.0;
  if (tile < v.n_tiles) trial_tile<FUSED>(v, cur, mult, h, s_cams + (1 - cur) * kMaxCams * kCamStride, tile, wave, lane, ds_s, lds_rows, wsum);
  if (v.merged) {      // the workgroup's step scalars in one record: the next pass's decision reads n_tiles / 4 of these
    __shared__ double s_w[4 * kNumScal];
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < kNumScal; ++k) s_w[wave * kNumScal + k] = wsum[k];
    }
    __syncthreads();
    if (threadIdx.x < kNumScal) {
      const int k = threadIdx.x;
      const double a = s_w[k], b = s_w[kNumScal + k], c = s_w[2 * kNumScal + k], d = s_w[3 * kNumScal + k];
      v.wgpart[(size_t)blockIdx.x * kNumScal + k] = (k == kScGmax) ? fmax(fmax(a, b), fmax(c, d)) : (a + b) + (c + d);
    }
  }
}

// ------------------------------------------------------------------------------------------ decision
// The Ceres trust-region bookkeeping (TrustRegionMinimizer + LevenbergMarquardtStrategy) and the
// reference's iteration callback (vicalibrator.h:690-721), run by one thread after every pass.
__device__ void trace_push(const DevView& v, Ctrl* c, const double* rec, bool writer = true) {
  if (writer && c->trace_len < c->trace_cap) {
    double* o = v.trace + (size_t)c->trace_len * kTraceCols;
    for (int i = 0; i < kTraceCols; ++i) o[i] = rec[i];
    if (v.host_progress && v.host_trace && c->trace_len < 64) {        // the host's copy (published by the progress word's store)
      double* ho = v.host_trace + (size_t)c->trace_len * kTraceCols;
      for (int i = 0; i < kTraceCols; ++i) ho[i] = rec[i];
    }
  }
  c->trace_len += 1;
  c->last_gnorm = rec[4];
}
// s: the frame-side step scalars of the judged pass, fail: a factorisation of that pass broke down, writer: this caller owns
// the global side effects (trace rows)
// t: the shared parameters' terms (k_reduced: scal[8..15]), R_cost: cost at the linearisation point (Sbuf's cost slot) -- nullptr / NaN-free
// defaults read them here; k_final's single-process path hands in what it requested at its entry
__device__ void lm_decide_local(const DevView& v, Ctrl* c, const double* s, bool fail, bool writer, const double* t_in = nullptr, const double* cost_in = nullptr) {
  const int D = v.D;
  const double* t = t_in ? t_in : v.scal + kNumScal;
  const double R_cost = cost_in ? *cost_in : v.Sbuf[(size_t)D * D + 3 * D];
  const double R_gd = s[kScGd] + t[kScGd], R_dld = s[kScDld] + t[kScDld];
  const double R_step2 = s[kScStep2] + t[kScStep2], R_x2 = s[kScX2] + t[kScX2];
  const double R_gnorm = sqrt(s[kScG2] + t[kScG2]), R_gmax = fmax(s[kScGmax], t[kScGmax]);
  const double R_new_cost = s[kScCost];
  c->passes += 1;
  c->likely_last = 0;
  c->res_sweeps += v.fused ? 0 : 1;
  c->jac_sweeps += (c->need_lin ? 1 : 0) + (v.fused ? 1 : 0);
  if (c->hold) { c->cost = R_cost; c->gmax = R_gmax; c->gnorm = R_gnorm; return; }
  if (c->pending) {            // the pass linearised at the newly accepted point
    c->cost = R_cost; c->gmax = R_gmax; c->gnorm = R_gnorm;
    c->pend[1] = R_cost; c->pend[3] = R_gmax; c->pend[4] = R_gnorm;
    trace_push(v, c, c->pend, writer);
    c->pending = 0;
    if (R_gmax <= c->gtol) { c->done = kDoneConvergence; return; }
  } else if (c->first) {
    c->cost = R_cost; c->gmax = R_gmax; c->gnorm = R_gnorm;
    const double rec[kTraceCols] = {0.0, R_cost, 0.0, R_gmax, R_gnorm, 0.0, 0.0, c->radius, 1.0, (double)c->stage};
    trace_push(v, c, rec, writer);
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
    if (c->invalid >= 5) { trace_push(v, c, rec, writer); c->done = kDoneFailure; return; }
    c->radius *= 0.5; rec[7] = c->radius;
    trace_push(v, c, rec, writer);
    c->need_lin = 0; c->reuse_diag = 1;
    return;
  }
  c->invalid = 0;
  rec[5] = sqrt(R_step2);
  const double xnorm = sqrt(R_x2);
  if (rec[5] <= c->ptol * (xnorm + c->ptol)) { trace_push(v, c, rec, writer); c->done = kDoneConvergence; return; }
  rec[2] = c->cost - R_new_cost;
  if (fabs(rec[2]) < c->ftol * c->cost) { trace_push(v, c, rec, writer); c->done = kDoneConvergence; return; }
  rec[6] = rec[2] / model_change;
  if (rec[6] > 1e-3) {
    {
      // convergence predictor for the feeding host (vc_solve.cpp: solve_once): quadratic-looking approach to the function tolerance
      const double rel = fabs(rec[2]) / c->cost;
      c->likely_last = (rel < 1e3 * c->ftol && rel < 0.1 * c->last_rel) ? 1 : 0;
      c->last_rel = rel;
    }
    c->cur = 1 - c->cur;
    const double q = 2.0 * rec[6] - 1.0;
    c->radius = fmin(1e16, c->radius / fmax(1.0 / 3.0, 1.0 - q * q * q));
    c->decrease_factor = 2.0;
    rec[7] = c->radius; rec[8] = 1.0;
    for (int i = 0; i < kTraceCols; ++i) c->pend[i] = rec[i];
    c->pending = 1; c->need_lin = v.fused ? 0 : 1; c->reuse_diag = 0;
  } else {
    c->radius = c->radius / c->decrease_factor; c->decrease_factor *= 2.0;
    rec[7] = c->radius;
    trace_push(v, c, rec, writer);
    if (c->radius < 1e-32) { c->done = kDoneConvergence; return; }
    c->need_lin = 0; c->reuse_diag = 1;
  }
}

// the host's view of the loop: one system-scope store per decision (vc_solve.cpp: solve_once feeds passes against it)
__device__ __forceinline__ void publish_progress(const DevView& v, const Ctrl& c) {
  if (v.host_progress) {
    // a finished solve: the record itself goes to the host's page-locked copy, then -- after a system-scope fence -- the progress word
    // says `done`; the host reads the result from there while the passes queued past the end drain (no copy, no synchronisation)
    // (the fence only then: it waits for the stores to host memory to be performed -- a round trip over the host link -- and the host
    //  reads the record and the trace rows only once it has seen `done`; the rows of earlier passes were completed by their kernels' ends)
    if (c.done && v.host_ctrl) { *v.host_ctrl = c; __threadfence_system(); }
    __hip_atomic_store(v.host_progress, ((unsigned long long)(unsigned)c.passes << 32) | (unsigned)c.done | (c.likely_last ? kProgressLikelyLast : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// void_pass: another rank of a sharded solve has marked this pass void (final_phase, mode 2) -- honoured whatever THIS rank's
// hand-over mode: a rank on events must withhold the decision the flag ranks withhold, or the ranks' states and collective
// schedules diverge
// Everything the deciding thread reads that this launch does not produce itself: the control record, the failure marks (set by the
// chain kernels of the main stream, long done), the shared parameters' terms and the linearisation cost (k_reduced).  One burst of loads.
struct DecideInputs { Ctrl c; double t[kNumScal]; double lin_cost; int f4, f5; };
__device__ __forceinline__ void load_decide_inputs(const DevView& v, DecideInputs* in) {
  in->c = *v.ctrl;
  in->f4 = v.flags[4 + 2 * v.par]; in->f5 = v.flags[5 + 2 * v.par];
#pragma unroll
  for (int k = 0; k < kNumScal; ++k) in->t[k] = v.scal[kNumScal + k];
  in->lin_cost = v.Sbuf[(size_t)v.D * v.D + 3 * v.D];
}
// s: the step scalars of the pass (v.scal, or registers of the launch that has just reduced them)
// marked: the sticky time-out word if the caller has read it already (behind its last wait), -1: read here
__device__ void lm_decide_with(const DevView& v, DecideInputs& in, bool void_pass, const double* s, long long marked = -1) {
  Ctrl& local = in.c;
  // device-flag hand-overs: a wait of this pass (or of one before it, noticed after that pass's decision) ran into its bound --
  // what this pass computed cannot be trusted.  No judgement: the record stays as the last valid decision left it, the solve ends
  // with kDoneSyncTimeout and the host resumes it with event hand-overs (vc_kutil.hpp: spin_until_flag)
  if (v.sync_seq > 0 && !void_pass) { const long long m = marked >= 0 ? marked : sync_marked(v); void_pass = m != 0 && m <= v.sync_seq; }
  if (void_pass) {
    local.done = kDoneSyncTimeout; local.abort_seq = (int)(v.pass_id & 0x7fffffff);      // (this pass is the first one without a decision; pass_id == sync_seq where flags are on)
    *v.ctrl = local;
    publish_progress(v, local);
    return;
  }
  const bool fail = (in.f4 != 0) || (in.f5 != 0);
  v.flags[4 + 2 * v.par] = 0; v.flags[5 + 2 * v.par] = 0;
  lm_decide_local(v, &local, s, fail, true, in.t, &in.lin_cost);
  *v.ctrl = local;
  publish_progress(v, local);
}
__device__ void lm_decide(const DevView& v, bool void_pass = false) {
  DecideInputs in;
  load_decide_inputs(v, &in);
  lm_decide_with(v, in, void_pass, v.scal);
}
// ---- merged decision --------------------------------------------------------------------------------------------
// The control record of the current pass from the previous pass's record: judge the pending trial point, or carry a finished
// state forward, or (first pass / after k_final_merged) take the record that is already in place.  All threads call it;
// `out` (LDS) holds the result after the trailing barrier; red: 8 x 4 doubles of LDS.  writer: this workgroup stores the
// record and the trace rows.
__device__ void merged_control(const DevView& v, Ctrl* out, double* red, bool writer) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const Ctrl* cp = v.ctrl_prev;
  // everything the decision may need is requested up front (the per-workgroup scalars, the previous record): one memory
  // latency instead of a chain of dependent ones
  const int nwg = (v.n_tiles + 3) / 4;
  double a[kNumScal];
#pragma unroll
  for (int k = 0; k < kNumScal; ++k) a[k] = 0.0;
  for (int i = tid; i < nwg; i += 256) {
#pragma unroll
    for (int k = 0; k < kNumScal; ++k) {
      const double x = v.wgpart[(size_t)i * kNumScal + k];
      a[k] = (k == kScGmax) ? fmax(a[k], x) : a[k] + x;
    }
  }
  Ctrl c;
  if (tid == 0) c = *cp;
  const int nd = cp->needs_decision, pdone = cp->done;       // wave-uniform
  if (nd) {
    // fixed-order sum of k_trial's per-workgroup scalars: every caller (each workgroup of k_frame_schur, k_final_merged)
    // gets bit-identical sums
#pragma unroll
    for (int k = 0; k < kNumScal; ++k) {
      double x = a[k];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { const double y = __shfl_down(x, o, 64); x = (k == kScGmax) ? fmax(x, y) : x + y; }
      if (lane == 0) red[k * 4 + wave] = x;
    }
    __syncthreads();
    if (tid == 0) {
      double s[kNumScal];
#pragma unroll
      for (int k = 0; k < kNumScal; ++k)
        s[k] = (k == kScGmax) ? fmax(fmax(red[k * 4], red[k * 4 + 1]), fmax(red[k * 4 + 2], red[k * 4 + 3]))
                              : (red[k * 4] + red[k * 4 + 1]) + (red[k * 4 + 2] + red[k * 4 + 3]);
      bool fail;
      if (v.world > 1 || v.shard_src) {
        // sharded: the ranks' scalars (already halved costs, failure flags in slot kScSq) were gathered by the all-reduce
        for (int k = 0; k < kNumScal; ++k) {
          double a2 = 0.0;
          for (int r = 0; r < v.world; ++r) a2 = (k == kScGmax) ? fmax(a2, v.gath[r * kNumScal + k]) : a2 + v.gath[r * kNumScal + k];
          s[k] = a2;
        }
        fail = s[kScSq] > 0.0;
      } else {
        s[kScCost] *= 0.5;
        const int fp = 1 - v.par;                               // the judged pass ran with the other parity
        fail = (v.flags[4 + 2 * fp] != 0) || (v.flags[5 + 2 * fp] != 0);
      }
      lm_decide_local(v, &c, s, fail, writer);
      c.needs_decision = 0;
      *out = c;
      if (writer) { *v.ctrl = c; publish_progress(v, c); }
    }
  } else if (tid == 0) {
    if (pdone) { *out = c; if (writer) *v.ctrl = c; }
    else *out = *v.ctrl;
  }
  __syncthreads();
}
// batch end in merged mode: the last pass's trial point is judged here (the host reads the record this writes)
__global__ __launch_bounds__(256) void k_final_merged(DevView v) {
  __shared__ Ctrl c;
  __shared__ double red[kNumScal * 4];
  // here `ctrl` = record of the NEXT pass (to be written), `ctrl_prev` = record of the pass just run
  merged_control(v, &c, red, true);
  if (threadIdx.x == 0 && v.ctrl_prev->needs_decision) const_cast<Ctrl*>(v.ctrl_prev)->needs_decision = 0;
}
// mode 0: reduce + decide, 1: reduce only (an all-reduce follows), 2: decide only
__device__ void final_phase(const DevView& v, int mode, double* red) {
  const int tid = threadIdx.x;
  if (mode != 2) {
    double s[7] = {0, 0, 0, 0, 0, 0, 0};
    if (v.imu_on) {      // pre-reduced where they are produced (k_chain_back, k_reproj_jac / k_imu_jac in trial mode)
      // (unrolled: the loads of several rounds are in flight together -- one workgroup on the critical path, 49 rounds at 50 000 tiles)
#pragma unroll 4
      for (int g = tid; g < v.n_chain_groups; g += 256) {
        const double* p = v.grp_part + (size_t)g * kNumScal;
        s[0] += p[kScGd]; s[1] += p[kScDld]; s[2] += p[kScStep2]; s[3] += p[kScX2]; s[4] += p[kScG2];
        s[6] = fmax(s[6], p[kScGmax]);
      }
#pragma unroll 16
      for (int t = tid; t < (v.n_tiles + 3) / 4; t += 256) s[5] += v.wg_trial[t];
      if (v.final_wait > 0) {      // delivered by device-coherent stores while k_imu_jac still runs (see k_final)
#pragma unroll 4
        for (int t = tid; t < (v.n_frames - 1 + 7) / 8; t += 256) s[5] += __hip_atomic_load(v.wg_imu_trial + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
#pragma unroll 4
        for (int t = tid; t < (v.n_frames - 1 + 7) / 8; t += 256) s[5] += v.wg_imu_trial[t];
      }
    } else {
#pragma unroll 4
      for (int f = tid; f < v.n_frames; f += 256) {
        const double* p = v.fpart + (size_t)f * kNumScal;
        s[0] += p[kScGd]; s[1] += p[kScDld]; s[2] += p[kScStep2]; s[3] += p[kScX2]; s[4] += p[kScG2];
        s[6] = fmax(s[6], p[kScGmax]);
      }
#pragma unroll 16
      for (int t = tid; t < v.n_tiles; t += 256) s[5] += v.tile_trial[2 * t];
    }
    for (int k = 0; k < 7; ++k) red[k * 256 + tid] = s[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) {
        for (int k = 0; k < 6; ++k) red[k * 256 + tid] += red[k * 256 + tid + o];
        red[6 * 256 + tid] = fmax(red[6 * 256 + tid], red[6 * 256 + tid + o]);
      }
      __syncthreads();
    }
#ifdef VC_FINAL_STAMPS
    if (tid == 0) v.dbg[3] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    if (tid == 0) {
      double* o = v.scal;
      o[kScGd] = red[0]; o[kScDld] = red[256]; o[kScStep2] = red[512]; o[kScX2] = red[768]; o[kScG2] = red[1024];
      o[kScCost] = 0.5 * red[1280]; o[kScGmax] = red[1536]; o[kScSq] = 0.0;
      // sharded with flag hand-overs: a rank whose pass is marked void (vc_kutil.hpp) says so with kShardMark on top of its failure
      // count -- after the all-reduce every rank sees it and none of them judges the pass
      double mark = 0.0;
      if (mode == 1 && v.sync_seq > 0) { const long long m = sync_marked(v); if (m != 0 && m <= v.sync_seq) mark = kShardMark; }
      if (mode == 1) {       // sharded: publish this rank's terms (and its numeric-failure flags) in its slot of the gather table
        for (int r = 0; r < v.world; ++r)
          for (int k = 0; k < kNumScal; ++k) v.gath[r * kNumScal + k] = (r == v.rank) ? (k == kScSq ? (double)(v.flags[4 + 2 * v.par] + v.flags[5 + 2 * v.par]) + mark : o[k]) : 0.0;
      }
    }
  }
  bool void_pass = false;
  if (mode == 2 && tid == 0) {   // after the all-reduce: combine the ranks in fixed order, identically everywhere
    double* o = v.scal;
    for (int k = 0; k < kNumScal; ++k) {
      double a = 0.0;
      for (int r = 0; r < v.world; ++r) a = (k == kScGmax) ? fmax(a, v.gath[r * kNumScal + k]) : a + v.gath[r * kNumScal + k];
      o[k] = a;
    }
    // some rank's pass is void: this rank's is, too -- lm_decide below withholds the decision, on a rank that hands over through
    // events as well (ranks may disagree on the mode: a failed priority stream, a per-rank environment, a rank that fell back earlier)
    void_pass = o[kScSq] >= kShardMark;
    if (void_pass && v.sync_seq > 0) mark_sync_timeout(v, v.sync_seq);
    v.flags[4 + 2 * v.par] = (fmod(o[kScSq], kShardMark) > 0.0) ? 1 : 0; v.flags[5 + 2 * v.par] = 0;
  }
#ifdef VC_FINAL_STAMPS
  if (tid == 0) v.dbg[4] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  if (mode != 1 && tid == 0) lm_decide(v, void_pass);
}
// Single-process visual-inertial pass (k_final, mode 0): the same sums, arranged around the wait.  The main stream's terms -- the
// chain groups' step terms, the vision sweep's trial cost -- are reduced BEFORE the wait for the second stream's workgroup count
// (wave shuffles + one LDS exchange instead of an eight-level tree of barriers), the second stream's trial cost behind it, and the deciding
// thread takes the totals from registers.  Stamps (tools/final_stamps.py, round 4 form): 3.4 us reduction + 4.1 us decision behind the
// count, all of it on the critical path at the end of every pass.
__device__ void final_phase_vi(const DevView& v, double* red /* >= 40 */, bool counted, long long nwg_imu) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // (the deciding thread's inputs: requested first, 2.4 us of round trips that used to follow the last barrier)
  DecideInputs din;
  if (tid == 0) load_decide_inputs(v, &din);
  double s[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll 4
  for (int g = tid; g < v.n_chain_groups; g += 256) {
    const double* p = v.grp_part + (size_t)g * kNumScal;
    s[0] += p[kScGd]; s[1] += p[kScDld]; s[2] += p[kScStep2]; s[3] += p[kScX2]; s[4] += p[kScG2];
    s[6] = fmax(s[6], p[kScGmax]);
  }
#pragma unroll 16
  for (int t = tid; t < (v.n_tiles + 3) / 4; t += 256) s[5] += v.wg_trial[t];
#pragma unroll
  for (int k = 0; k < 6; ++k) s[k] = wave_sum(s[k]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s[6] = fmax(s[6], __shfl_down(s[6], o, 64));
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 7; ++k) red[wave * 8 + k] = s[k];
  }
#ifdef VC_FINAL_STAMPS
  if (tid == 0) v.dbg[8] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  if (counted && tid == 0) {
    // the count is a running one (never reset inside a solve: a reset could race with workgroups still adding); this kernel keeps
    // the count at the end of the last judged pass in sync_flags[5]
    const long long base = __hip_atomic_load(v.sync_flags + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    spin_until_flag(v, 4, base + nwg_imu);
    __hip_atomic_store(v.sync_flags + 5, base + nwg_imu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the next pass counts on from here
  }
  __syncthreads();
  // (the sticky time-out word: behind this launch's last wait that could set it, requested together with the second stream's costs)
  const long long marked = (tid == 0 && v.sync_seq > 0) ? sync_marked(v) : 0;
  double b = 0.0;
  if (v.final_wait > 0) {      // delivered by device-coherent stores while k_imu_jac still runs (see k_final)
#pragma unroll 4
    for (int t = tid; t < (v.n_frames - 1 + 7) / 8; t += 256) b += __hip_atomic_load(v.wg_imu_trial + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
#pragma unroll 4
    for (int t = tid; t < (v.n_frames - 1 + 7) / 8; t += 256) b += v.wg_imu_trial[t];
  }
  b = wave_sum(b);
  if (lane == 0) red[32 + wave] = b;
#ifdef VC_FINAL_STAMPS
  if (tid == 0) { v.dbg[3] = (long long)__builtin_amdgcn_s_memrealtime(); }
#endif
  __syncthreads();
  if (tid == 0) {
#ifdef VC_FINAL_STAMPS
    v.dbg[4] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    double o[kNumScal];
    const int idx[6] = {kScGd, kScDld, kScStep2, kScX2, kScG2, kScCost};
#pragma unroll
    for (int k = 0; k < 6; ++k) o[idx[k]] = (red[k] + red[8 + k]) + (red[16 + k] + red[24 + k]);
    o[kScCost] = 0.5 * (o[kScCost] + ((red[32] + red[33]) + (red[34] + red[35])));
    o[kScGmax] = fmax(fmax(red[6], red[14]), fmax(red[22], red[30])); o[kScSq] = 0.0;
#pragma unroll
    for (int k = 0; k < kNumScal; ++k) v.scal[k] = o[k];      // (parity hooks read them)
#ifdef VC_FINAL_STAMPS
    v.dbg[9] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    lm_decide_with(v, din, false, o, marked);
#ifdef VC_FINAL_STAMPS
    v.dbg[10] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  }
}
// The waiting side as a kernel of its own: one wavefront on the second stream; the kernels behind it in that stream start when it
// returns (vc_kutil.hpp: spin_until_flag).
__global__ __launch_bounds__(64) void k_wait_flag(DevView v, int idx, long long seq) {
  if (threadIdx.x == 0) spin_until_flag(v, idx, seq);
}
// the producing side for kernels of many workgroups: a one-thread kernel behind them in their stream (the kernel boundary has
// written their results back; one fence per workgroup would cost more than this launch)
__global__ __launch_bounds__(64) void k_signal_flag(DevView v, int idx) {
  if (threadIdx.x == 0) signal_flag(v, idx);
}
#ifdef VC_FINAL_STAMPS
#define FSTAMP(i) do { if (threadIdx.x == 0 && !over_) v.dbg[i] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FSTAMP(i) do { } while (0)
#endif
__global__ __launch_bounds__(256) void k_final(DevView v, int mode) {
  __shared__ double red[256 * 7];
#ifdef VC_FINAL_STAMPS
  const long long fs0_ = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  // the second stream's share of the trial cost (k_imu_jac's workgroup sums): wait for its flag here, inside the kernel, instead of
  // behind a cross-stream event (12 us between the second stream's last kernel and this one on the timeline); every wavefront
  // then drops what it may have cached of the other stream's results.  Also when the solve is over: the main stream must not
  // run ahead of the second one.
  //
  // Two waits: (1) for the count of k_imu_jac's workgroups that have delivered their share of the trial cost -- the decision is
  // taken while that kernel still writes its records; (2) at the very end, for the flag behind that kernel (k_signal_flag on the
  // second stream): this kernel must not end before the second stream's results are complete and written back -- the next pass's
  // k_chain_init reads them, and a solve that is over must leave nothing running.
  // Sharded (mode 1: reduce, all-reduce, mode 2: decide): the first wait belongs to the reducing launch, the second wait and the
  // flag to the deciding one.
  const bool over = v.ctrl->done != 0;
  [[maybe_unused]] const bool over_ = over;
#ifdef VC_FINAL_STAMPS
  if (threadIdx.x == 0 && !over) v.dbg[0] = fs0_;
#endif
  FSTAMP(1);
  __shared__ long long s_count_base;
  const long long nwg_imu = (long long)((v.n_frames - 1 + 7) / 8);
  const bool counted = v.final_wait > 0 && !over && v.n_frames > 1 && mode != 2;
  const bool vi_single = mode == 0 && v.imu_on && !over;      // the count is waited for inside final_phase_vi, behind the main stream's sums
  if (counted && !vi_single) {
    // the count is a running one (never reset inside a solve: a reset could race with workgroups still adding); this kernel keeps
    // the count at the end of the last judged pass in sync_flags[5]
    if (threadIdx.x == 0) {
      s_count_base = __hip_atomic_load(v.sync_flags + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      spin_until_flag(v, 4, s_count_base + nwg_imu);
      __hip_atomic_store(v.sync_flags + 5, s_count_base + nwg_imu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the next pass counts on from here
    }
    __syncthreads();
  }
  FSTAMP(2);
  if (vi_single) final_phase_vi(v, red, counted, nwg_imu);
  else if (!over) final_phase(v, mode, red);
  if (mode == 1) return;
  FSTAMP(5);
  __syncthreads();
  if (v.final_wait > 0 && v.n_frames > 1) {
    // the second stream's records are complete: every workgroup of k_imu_jac(trial) has counted itself a second time behind its
    // (device-coherent) stores -- a running count like the first, this kernel's book-keeping in sync_flags[12].  Also when the solve is
    // over (the workgroups count themselves at their exit): the main stream must not run ahead of the second one into the next solve.
    if (threadIdx.x == 0) {
      const long long base = __hip_atomic_load(v.sync_flags + 12, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      spin_until_flag(v, 11, base + nwg_imu);
      __hip_atomic_store(v.sync_flags + 12, base + nwg_imu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
  }
  FSTAMP(6);
  // (a pass that has been marked void still signals: the next pass's waiters return at once anyway, and the host takes over)
  if (threadIdx.x == 0) signal_flag(v, 0);
  FSTAMP(7);
}

// One stage's small uploads (vc_upload.cpp: upload): the host packs them into one page-locked staging image that goes to the
// device with ONE copy; this kernel scatters the image's segments to their buffers (several destinations may share a source: both
// state buffers and the "initial state" copy take the same poses) and zero-fills what a stage starts from zero.  Replaces ~45 small
// pageable copies / fills per stage (4.1 of the 19 ms of a complete cfg3 calibration).
__global__ __launch_bounds__(256) void k_unpack(const UnpackSeg* segs, int n, const unsigned* image) {
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nt = (size_t)gridDim.x * 256;
  for (int k = 0; k < n; ++k) {
    const UnpackSeg sg = segs[k];
    unsigned* dst = (unsigned*)sg.dst;
    const size_t words = sg.bytes >> 2;
    if (sg.src_off == ~0ull) { for (size_t i = tid; i < words; i += nt) dst[i] = 0u; }
    else { const unsigned* src = image + (sg.src_off >> 2); for (size_t i = tid; i < words; i += nt) dst[i] = src[i]; }
  }
}
void launch_unpack(const UnpackSeg* segs, int n, const void* image, size_t total_bytes, hipStream_t s) {
  const int blocks = (int)std::min<size_t>(512, std::max<size_t>(1, total_bytes / (256 * 16)));
  hipLaunchKernelGGL(k_unpack, dim3(blocks), dim3(256), 0, s, segs, n, (const unsigned*)image);
}
// a fresh control record for a solve: record 0 <- the host's (a kernel argument), record 1 blank -- one launch instead of a copy and a fill
__global__ __launch_bounds__(64) void k_set_ctrl(Ctrl* d, Ctrl c) {
  if (threadIdx.x == 0) { d[0] = c; Ctrl z = Ctrl(); d[1] = z; }
}
void launch_set_ctrl(Ctrl* d, const Ctrl& c, hipStream_t s) { hipLaunchKernelGGL(k_set_ctrl, dim3(1), dim3(64), 0, s, d, c); }
// both state buffers <- the uploaded initial state (benchmark restarts), one launch
__global__ __launch_bounds__(256) void k_reset_state(DevView v, const double* pose0, const double* cam0, const double* vel0, const double* imu0) {
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, n = (size_t)gridDim.x * 256;
  for (size_t i = tid; i < (size_t)v.n_frames * kPoseStride; i += n) { const double x = pose0[i]; v.poses[0][i] = x; v.poses[1][i] = x; }
  for (size_t i = tid; i < (size_t)v.n_cams * kCamStride; i += n) { const double x = cam0[i]; v.cams[0][i] = x; v.cams[1][i] = x; }
  for (size_t i = tid; i < (size_t)v.n_frames * 4; i += n) { const double x = vel0[i]; v.vel[0][i] = x; v.vel[1][i] = x; }
  for (size_t i = tid; i < 16; i += n) { const double x = imu0[i]; v.imus[0][i] = x; v.imus[1][i] = x; }
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

void launch_reproj_jac(const DevView& v, hipStream_t s, int trial) {
  if (v.n_tiles == 0) return;
  size_t lds = 4 * 64 * kDotStride * sizeof(double);
  // Visual-inertial trial sweep: k_imu_jac(trial) starts on the second stream a few microseconds behind this kernel.  Both take ~230
  // registers; two workgroups of this kernel per CU (two wavefronts per SIMD) leave no room for a wavefront of the other, which then runs
  // behind this kernel's first round instead of beside it (26 instead of 17 us, and the pass ends with it).  One workgroup per CU here --
  // a larger LDS request is the lever a launch has -- lets the two share every SIMD from the start: k_imu_jac 18 us, this kernel 24 us
  // instead of 16 and now the last to end, with k_final's start-up no longer hidden behind the other stream: 0.2075 against 0.2035 ms per
  // iteration.  Off; VICALIB_AMD_TRIAL_ONE_WG=1 for A/B runs
  static const bool one_wg = [] { const char* e = std::getenv("VICALIB_AMD_TRIAL_ONE_WG"); return e && e[0] == '1'; }();
  if (trial && v.imu_on && one_wg && tiles_grid(v) <= 1024) {
    lds = 88 * 1024;
    static LdsGrant granted;
    if (granted.need(lds)) (void)hipFuncSetAttribute((const void*)k_reproj_jac, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  hipLaunchKernelGGL(k_reproj_jac, dim3(tiles_grid(v)), dim3(256), lds, s, v, trial);
}
void launch_part_sum(const DevView& v, hipStream_t s) {
  // (hadd_early: the entries behind S and g_red were summed ahead of this launch, vc_shared_blocks.hpp -- only the frames' partial sums are left)
  const int entries = v.hadd_early ? v.D * v.D + v.D : v.part_stride;
  hipLaunchKernelGGL(k_part_sum, dim3((entries + kSumEntries - 1) / kSumEntries), dim3(kSumEntries * kSumSlices), 0, s, v);
}
void launch_frame_schur(const DevView& v, hipStream_t s) {
  const int D = v.D;
  const int Dp = ((D + 1 + 15) / 16) * 16, ld = (Dp % 32 == 0) ? Dp + 16 : Dp;
  const size_t lds = ((size_t)4 * (v.n_cams * kGStride + kPrepPad) + (size_t)24 * ld) * sizeof(double);
  const int nT = (D + 1 + 15) / 16;
  auto go = [&](auto kern) {
    if (lds > 65536) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(v.n_chunks), dim3(256), lds, s, v);
  };
  if (v.n_cams <= 2 && nT <= 2) go(k_frame_schur<2, 1, 2>);
  else if (v.n_cams <= 4 && nT <= 4) go(k_frame_schur<4, 3, 2>);
  else go(k_frame_schur<kMaxCams, kMaxPairsPerWave, 2>);
  launch_part_sum(v, s);
}
static inline size_t reduced_lds(const DevView& v) {
  const size_t solve = v.D <= kSmallD ? ((size_t)(kSmallD + 1) * (kSmallD + 2) + 3 * (kSmallD + 1)) * sizeof(double)
                                      : ((size_t)(v.D + 1) * (v.D + 2) / 2 + 3 * (v.D + 1) + 2 + 384 + 384) * sizeof(double);      // (M, row D, x, dinv; two column images)
  const size_t top = (v.gram_top_stride > 0 && v.D <= kEarlyTopD) ? (size_t)64 * kTopLd * sizeof(double) : 0;      // (k_reduced: s_top)
  return std::max(solve, (sizeof(FinalLds) + 7) / 8 * 8 + top);
}
// the reduced system's LDS image fits the workgroup's 160 KB (k_reduced: static tables + the packed triangle): D <= 179 on gfx950 -- BASELINE
// cfg5 over 8 ranks is D = 178.  Asked of the runtime, not assumed: a launch beyond it fails without a word.
bool reduced_fits(const DevView& v) {
  hipFuncAttributes fa;
  int dev = 0, max_lds = 0;
  if (hipFuncGetAttributes(&fa, (const void*)k_reduced) != hipSuccess || hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) { (void)hipGetLastError(); return true; }
  if (max_lds < 163840) max_lds = 163840;      // (gfx950: 160 KB per workgroup once hipFuncAttributeMaxDynamicSharedMemorySize is granted; the attribute reports the default 64 KB)
  return reduced_lds(v) + fa.sharedSizeBytes <= (size_t)max_lds;
}
void launch_reduced(const DevView& v, int mode, hipStream_t s) {
  const size_t lds = reduced_lds(v);
  static LdsGrant granted;
  if (lds > 30000 && granted.need(lds)) (void)hipFuncSetAttribute((const void*)k_reduced, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_reduced, dim3(1), dim3(256), lds, s, v, mode);
}
void launch_trial(const DevView& v, hipStream_t s) {
  if (v.n_tiles == 0) return;
  if (v.pre_backsub) hipLaunchKernelGGL(k_backsub, dim3((v.n_frames + 3) / 4), dim3(256), 0, s, v);
  const size_t lds = 4 * 64 * kDotStride * sizeof(double);
  static LdsGrant granted;
  if (lds > 0 && granted.need(lds + 4096)) (void)hipFuncSetAttribute((const void*)k_trial<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds + 4096));
  hipLaunchKernelGGL(k_trial<true>, dim3(tiles_grid(v)), dim3(256), lds, s, v);      // vision-only passes are always fused
}
void launch_final_merged(const DevView& v, hipStream_t s) { hipLaunchKernelGGL(k_final_merged, dim3(1), dim3(256), 0, s, v); }
void launch_final(const DevView& v, int mode, hipStream_t s) {
  hipLaunchKernelGGL(k_final, dim3(1), dim3(256), 0, s, v, mode);
}
void launch_wait_flag(const DevView& v, int idx, long long seq, hipStream_t s) { hipLaunchKernelGGL(k_wait_flag, dim3(1), dim3(64), 0, s, v, idx, seq); }
void launch_signal_flag(const DevView& v, int idx, hipStream_t s) { hipLaunchKernelGGL(k_signal_flag, dim3(1), dim3(64), 0, s, v, idx); }
void launch_reproj_res(const DevView& v, int state, double mult, hipStream_t s) {
  if (v.n_tiles == 0) return;
  hipLaunchKernelGGL(k_reproj_res, dim3(tiles_grid(v)), dim3(256), 0, s, v, state, mult);
}
void launch_reset_state(const DevView& v, const double* pose0, const double* cam0, const double* vel0, const double* imu0, hipStream_t s) {
  const int blocks = std::max(1, std::min(256, (v.n_frames * kPoseStride + 255) / 256));
  hipLaunchKernelGGL(k_reset_state, dim3(blocks), dim3(256), 0, s, v, pose0, cam0, vel0, imu0);
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
