// Test-only host build of vicalib_amd/csrc/vc_math.hpp (the arithmetic the HIP kernels call), so
// that the closed-form Jacobians and the tile -> frame/camera block algebra can be checked against
// the oracle on a machine without a GPU.  Emulates what one wavefront of k_reproj_jac and one thread
// of k_frame_prep / k_schur_final compute, serially.  Not part of the product library.
#include <cstring>
#include "../../vicalib_amd/csrc/vc_math.hpp"
using namespace vc;

template <int MODEL>
static double gram(const TileXf& x, const double* K, int n, const double* pw, const double* uv, double mult, double* G) {
  double cost = 0;
  ModelPre pre; model_precompute(MODEL, K, &pre);
  for (int d = 0; d < n; ++d) {
    double r0[16], r1[16];
    cost += corner_rows<MODEL>(x, K, pre, pw + 3 * d, uv[2 * d], uv[2 * d + 1], mult, r0, r1);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) G[i * 16 + j] += r0[i] * r0[j] + r1[i] * r1[j];
  }
  return cost;
}
template <int MODEL>
static double resid(const TileXf& x, const double* K, int n, const double* pw, const double* uv, double* r) {
  double cost = 0;
  ModelPre pre; model_precompute(MODEL, K, &pre);
  for (int d = 0; d < n; ++d) cost += corner_residual<MODEL>(x, K, pre, pw + 3 * d, uv[2 * d], uv[2 * d + 1], r + 2 * d);
  return cost;
}

extern "C" {
void hh_project(int model, const double* pc, const double* K, double* pix, double* A, double* B) { ModelPre pre; model_precompute(model, K, &pre); project_any<true>(model, pc, K, pre, pix, A, B); }
double hh_tile_gram(int model, const double* T_wk, const double* T_ck, const double* K, int n, const double* pw, const double* uv,
                    double mult, double* G) {
  TileXf x; make_tile_xf(T_wk, T_ck, &x);
  std::memset(G, 0, 256 * sizeof(double));
  switch (model) {
    case kFov: return gram<kFov>(x, K, n, pw, uv, mult, G);
    case kPoly2: return gram<kPoly2>(x, K, n, pw, uv, mult, G);
    case kPoly3: return gram<kPoly3>(x, K, n, pw, uv, mult, G);
    case kKb4: return gram<kKb4>(x, K, n, pw, uv, mult, G);
    default: return gram<kLinear>(x, K, n, pw, uv, mult, G);
  }
}
double hh_tile_resid(int model, const double* T_wk, const double* T_ck, const double* K, int n, const double* pw, const double* uv, double* r) {
  TileXf x; make_tile_xf(T_wk, T_ck, &x);
  switch (model) {
    case kFov: return resid<kFov>(x, K, n, pw, uv, r);
    case kPoly2: return resid<kPoly2>(x, K, n, pw, uv, r);
    case kPoly3: return resid<kPoly3>(x, K, n, pw, uv, r);
    case kKb4: return resid<kKb4>(x, K, n, pw, uv, r);
    default: return resid<kLinear>(x, K, n, pw, uv, r);
  }
}
void hh_frame_blocks(const double* G, const double* q_ck, int nk, int flags, double* Hff, double* gf, double* W) {
  double R[9]; quat_to_R(q_ck, R);
  tile_to_frame_blocks(G, R, nk, flags, Hff, gf, W);
}
void hh_cam_block(const double* G, const double* q_ck, int nk, int flags, double* Hcc, double* gc) {
  double R[9]; quat_to_R(q_ck, R);
  cam_block_from_gsum(G, R, nk, flags, Hcc, gc);
}
void hh_se3_plus(const double* T, const double* d, double* o) { se3_plus(T, d, o); }
void hh_so3_plus(const double* q, const double* w, double* o) { so3_plus(q, w, o); }
int hh_chol6(double* M) { return chol_small<6>(M) ? 1 : 0; }
}
