// ORACLE -- TEST / BENCH INFRASTRUCTURE ONLY.
//
// vco_fast.h: the "best CPU" leg of bench.py's cpu_baseline: the same normal-equation work as the
// oracle, but with the reprojection Jacobians in closed form instead of forward duals (Dual<14 + K>, the work shape of Ceres'
// AutoDiffCostFunction).  The closed forms are the ones the HIP kernels use (vicalib_amd/csrc/vc_math.hpp compiles for the
// host): d r / d[upsilon_wk | omega_wk | omega_ck | t_ck | K] = [-A R_ck | (A [q]x) R_ck | -(A [q]x) R_ck | A | B] with
// A = d Project / d p_c, B = d Project / d K, q = p_c - t_ck.  Because it borrows the product's arithmetic it is NOT used by any
// parity test -- only timed; tests/test_oracle.py checks that it reproduces the dual-number blocks.
#pragma once
#include "../vicalib_amd/csrc/vc_math.hpp"
#include "vco_solver.h"

namespace vco {

inline void reproj_block_closed_form(const Calibrator& c, const Obs& o, double* r, double* Jf, double* Jr, double* Jt, double* Jk) {
  const Camera& cam = c.cams[o.cam];
  const Frame& f = c.frames[o.frame];
  vc::TileXf x;
  vc::make_tile_xf(f.T_wk, cam.T_ck, &x);
  double pc[3], pix[2], A[6], B[20], Rck[9];
  vc::tile_point(x, o.p_w, pc);
  vc::ModelPre pre;
  vc::model_precompute(cam.model, cam.K, &pre);
  vc::project_any<true>(cam.model, pc, cam.K, pre, pix, A, B);
  vc::quat_to_R(cam.T_ck, Rck);
  r[0] = pix[0] - o.z[0]; r[1] = pix[1] - o.z[1];
  const double q[3] = {pc[0] - x.tck[0], pc[1] - x.tck[1], pc[2] - x.tck[2]};
  const int nk = cam.nk;
  for (int i = 0; i < 2; ++i) {
    const double* a = A + 3 * i;
    const double aq[3] = {a[1] * q[2] - a[2] * q[1], a[2] * q[0] - a[0] * q[2], a[0] * q[1] - a[1] * q[0]};      // (A [q]x) row
    for (int j = 0; j < 3; ++j) {
      const double aR = a[0] * Rck[j] + a[1] * Rck[3 + j] + a[2] * Rck[6 + j];
      const double vR = aq[0] * Rck[j] + aq[1] * Rck[3 + j] + aq[2] * Rck[6 + j];
      Jf[i * 6 + j] = -aR; Jf[i * 6 + 3 + j] = vR; Jr[i * 3 + j] = -vR; Jt[i * 3 + j] = a[j];
    }
    for (int k = 0; k < nk; ++k) Jk[i * nk + k] = B[i * nk + k];
  }
}

}  // namespace vco
