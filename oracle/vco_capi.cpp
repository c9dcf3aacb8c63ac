// ORACLE -- TEST INFRASTRUCTURE ONLY (see vco_math.h header). PARITY UNPINNED.
// C entry points over the oracle for ctypes (tests/, bench.py cpu_baseline leg,
// __graft_entry__.smoke()).  Nothing in vicalib_amd/ links or loads this.
#include <cstring>
#include <vector>
#include <chrono>
#include "vco_solver.h"
#ifdef VCO_WITH_FAST
#include "vco_fast.h"        // libvco_fast.so only (bench.py's closed-form CPU leg): borrows the product's arithmetic
#endif

using namespace vco;

extern "C" {

// ---- primitives ------------------------------------------------------------------
void vco_so3_exp(const double* w, double* q) { so3_exp(w, q); }
void vco_so3_log(const double* q, double* w) { so3_log(q, w); }
void vco_se3_exp(const double* d, double* T) { se3_exp(d, T); }
void vco_se3_log(const double* T, double* d) { se3_log(T, d); }
// the restated dLog_dSE3 (vicalibrator-utils.h:308-434), 6 x 7 row-major, columns [t(3), q(4)], T stored [q, t]
void vco_dlog_dse3(const double* T, double* out42) { const Mat m = dlog_dse3(T); for (int i = 0; i < 42; ++i) out42[i] = m.d[i]; }
void vco_se3_mul(const double* A, const double* B, double* o) { se3_mul(A, B, o); }
void vco_se3_inv(const double* A, double* o) { se3_inv(A, o); }
void vco_plus_se3(const double* x, const double* d, double* o) { plus_se3(x, d, o); }
void vco_plus_so3(const double* x, const double* d, double* o) { plus_so3(x, d, o); }
void vco_local_jac_se3(const double* x, double* J) { local_jac_se3(x, J); }
void vco_local_jac_so3(const double* x, double* J) { local_jac_so3(x, J); }
void vco_gravity_vector(const double* dir, double* out) { gravity_vector(dir, gravity_magnitude(), out); }
int vco_model_num_params(int model) { return model_num_params(model); }
void vco_loss(int kind, double a, double s, double* rho) { if (kind == 0) loss_soft_l1(a, s, rho); else loss_cauchy(a, s, rho); }

// pix[2], d pix / d ray (2x3 row-major), d pix / d params (2 x nk)
void vco_project(int model, const double* ray, const double* k, double* pix, double* dray, double* dk) {
  const int nk = model_num_params(model);
  typedef Dual<13> D;
  D r[3], kk[10], p[2];
  for (int i = 0; i < 3; ++i) r[i] = D::Var(ray[i], i);
  for (int i = 0; i < nk; ++i) kk[i] = D::Var(k[i], 3 + i);
  project<D>(model, r, kk, p);
  for (int row = 0; row < 2; ++row) {
    pix[row] = p[row].a;
    if (dray) for (int j = 0; j < 3; ++j) dray[row * 3 + j] = p[row].v[j];
    if (dk) for (int j = 0; j < nk; ++j) dk[row * nk + j] = p[row].v[3 + j];
  }
}

// ---- calibrator ------------------------------------------------------------------
void* vco_create() { return new Calibrator(); }
void vco_destroy(void* h) { delete static_cast<Calibrator*>(h); }
#define CAL static_cast<Calibrator*>(h)

int vco_add_camera(void* h, int model, const double* params, int width, int height, const double* T_ck) {
  Camera c; std::memset(&c, 0, sizeof(c));
  c.model = model; c.nk = model_num_params(model);
  if (c.nk < 0 || c.nk > 10) return -1;
  std::memcpy(c.K, params, c.nk * sizeof(double));
  std::memcpy(c.T_ck, T_ck, 7 * sizeof(double));
  c.width = width; c.height = height;
  CAL->cams.push_back(c);
  return (int)CAL->cams.size() - 1;
}
int vco_add_frame(void* h, const double* T_wk, double time) {
  Frame f; std::memcpy(f.T_wk, T_wk, 56); f.v[0] = f.v[1] = f.v[2] = 0; f.time = time;
  CAL->frames.push_back(f);
  return (int)CAL->frames.size() - 1;
}
int vco_add_observations(void* h, int frame, int cam, int n, const double* p_w, const double* p_c) {
  if (frame < 0 || frame >= (int)CAL->frames.size() || cam < 0 || cam >= (int)CAL->cams.size()) return -1;
  for (int i = 0; i < n; ++i) {
    Obs o; o.frame = frame; o.cam = cam; o.mult_delta = 0;
    std::memcpy(o.p_w, p_w + 3 * i, 24); std::memcpy(o.z, p_c + 2 * i, 16);
    CAL->obs.push_back(o);
  }
  return 0;
}
int vco_add_imu(void* h, int n, const double* gyro, const double* accel, const double* t) {
  for (int i = 0; i < n; ++i) {
    if (!(t[i] > CAL->imu.end_time)) return -1;   // vicalibrator.h:373-378
    CAL->imu.add(gyro + 3 * i, accel + 3 * i, t[i]);
  }
  return 0;
}
void vco_set_flags(void* h, int bias_active, int inertial_active, int rotation_only, int time_offset) {
  CAL->is_scale_active = bias_active; CAL->is_bias_active = bias_active;   // vicalibrator.h:252-260
  CAL->is_inertial_active = inertial_active; CAL->rotation_only = rotation_only; CAL->optimize_time_offset = time_offset;
}
void vco_set_options(void* h, int max_iters, double function_tolerance, int calibrate_imu, int fix_intrinsics,
                     int remove_outliers, double outlier_threshold, int num_threads, int dense_check) {
  CAL->opt.max_iters = max_iters; CAL->opt.function_tolerance = function_tolerance; CAL->opt.calibrate_imu = calibrate_imu;
  CAL->fix_intrinsics = fix_intrinsics; CAL->opt.remove_outliers = remove_outliers; CAL->opt.outlier_threshold = outlier_threshold;
  CAL->opt.num_threads = num_threads; CAL->opt.dense_check = dense_check;
}
// 0 = done; -1 = this build has no closed-form path (libvco_oracle.so, the checker: it holds no product arithmetic)
int vco_set_closed_form(void* h, int on) {
#ifdef VCO_WITH_FAST
  CAL->opt.closed_form = on != 0; return 0;
#else
  (void)h; return on ? -1 : 0;
#endif
}
void vco_set_tolerances(void* h, double gradient_tolerance, double parameter_tolerance) {
  CAL->opt.gradient_tolerance = gradient_tolerance; CAL->opt.parameter_tolerance = parameter_tolerance;
}
void vco_set_imu_state(void* h, const double* biases, const double* scale, const double* g_dir, double time_offset,
                       double gyro_sigma, double accel_sigma) {
  std::memcpy(CAL->biases, biases, 48); std::memcpy(CAL->scale, scale, 48); std::memcpy(CAL->g_dir, g_dir, 16);
  CAL->time_offset = time_offset; CAL->gyro_sigma = gyro_sigma; CAL->accel_sigma = accel_sigma;
}
void vco_set_frame(void* h, int f, const double* T_wk, const double* v) { std::memcpy(CAL->frames[f].T_wk, T_wk, 56); if (v) std::memcpy(CAL->frames[f].v, v, 24); }
void vco_set_camera(void* h, int c, const double* K, const double* T_ck) { std::memcpy(CAL->cams[c].K, K, CAL->cams[c].nk * 8); std::memcpy(CAL->cams[c].T_ck, T_ck, 56); }
int vco_solve(void* h) { return CAL->solve(); }

int vco_num_frames(void* h) { return (int)CAL->frames.size(); }
int vco_num_cameras(void* h) { return (int)CAL->cams.size(); }
void vco_get_camera(void* h, int c, double* K, double* T_ck) { std::memcpy(K, CAL->cams[c].K, CAL->cams[c].nk * 8); std::memcpy(T_ck, CAL->cams[c].T_ck, 56); }
void vco_get_frame(void* h, int f, double* T_wk, double* v) { std::memcpy(T_wk, CAL->frames[f].T_wk, 56); if (v) std::memcpy(v, CAL->frames[f].v, 24); }
void vco_get_imu_state(void* h, double* biases, double* scale, double* g_dir, double* time_offset) {
  std::memcpy(biases, CAL->biases, 48); std::memcpy(scale, CAL->scale, 48); std::memcpy(g_dir, CAL->g_dir, 16); *time_offset = CAL->time_offset;
}
void vco_get_rmse(void* h, double* out) { for (size_t i = 0; i < CAL->cam_rmse.size(); ++i) out[i] = CAL->cam_rmse[i]; }
double vco_get_mse(void* h) { return CAL->mse; }
unsigned vco_get_num_iterations(void* h) { return CAL->num_iterations; }
int vco_trace_len(void* h) { return (int)CAL->trace.size(); }
// rows of 10 doubles: iteration cost cost_change gmax gnorm step_norm rho radius accepted stage
void vco_get_trace(void* h, double* out) {
  for (size_t i = 0; i < CAL->trace.size(); ++i) {
    const IterRecord& r = CAL->trace[i];
    double* o = out + 10 * i;
    o[0] = r.iteration; o[1] = r.cost; o[2] = r.cost_change; o[3] = r.gradient_max_norm; o[4] = r.gradient_norm;
    o[5] = r.step_norm; o[6] = r.relative_decrease; o[7] = r.radius; o[8] = r.accepted; o[9] = r.stage;
  }
}

// ---- evaluation hooks used for block-level parity with the HIP sweeps ---------------
// Prepares layout/multiplicities like the first SetupProblem of the current flags.
void vco_prepare(void* h, int vis_mult, int imu_mult) {
  CAL->sort_obs(); CAL->vis_mult = vis_mult; CAL->imu_mult = imu_mult; CAL->build_layout();
  const int n = (int)CAL->frames.size();
  if (CAL->imu_w_sqrt.size() != (size_t)std::max(0, n - 1) * 81) {
    CAL->imu_w_sqrt.assign((size_t)std::max(0, n - 1) * 81, 0.0);
    for (int j = 0; j + 1 < n; ++j) for (int i = 0; i < 9; ++i) CAL->imu_w_sqrt[(size_t)j * 81 + i * 10] = 500.0;
  }
}
int vco_layout_D(void* h) { return CAL->L.D; }
int vco_layout_df(void* h) { return CAL->L.df; }
// offsets: per camera rot, trans, k ; then g, b, sf, toff
void vco_layout_offsets(void* h, int* out) {
  const size_t C = CAL->cams.size();
  for (size_t c = 0; c < C; ++c) { out[3 * c] = CAL->L.rot[c]; out[3 * c + 1] = CAL->L.trans[c]; out[3 * c + 2] = CAL->L.kk[c]; }
  out[3 * C] = CAL->L.g; out[3 * C + 1] = CAL->L.b; out[3 * C + 2] = CAL->L.sf; out[3 * C + 3] = CAL->L.toff;
}
double vco_evaluate_cost(void* h) { return CAL->evaluate_cost(); }
double vco_linearize(void* h) { return CAL->linearize(); }
// A: N x 81, Cc: N x 81, W: N x 9 x D, Hss: D x D, gf: N x 9, gs: D
void vco_get_normal(void* h, double* A, double* Cc, double* W, double* Hss, double* gf, double* gs) {
  if (A) std::memcpy(A, CAL->A.data(), CAL->A.size() * 8);
  if (Cc) std::memcpy(Cc, CAL->Cc.data(), CAL->Cc.size() * 8);
  if (W) std::memcpy(W, CAL->W.data(), CAL->W.size() * 8);
  if (Hss) std::memcpy(Hss, CAL->Hss.data(), CAL->Hss.size() * 8);
  if (gf) std::memcpy(gf, CAL->gf.data(), CAL->gf.size() * 8);
  if (gs) std::memcpy(gs, CAL->gs.data(), CAL->gs.size() * 8);
}
// Solve (H + diag(lam)) d = -g on the current normal equations; lam: N*9 + D.
int vco_solve_normal(void* h, const double* lam, int dense, double* dfv, double* dsv) {
  std::vector<double> l(lam, lam + (size_t)CAL->N() * 9 + CAL->L.D), a, b;
  const bool ok = dense ? CAL->solve_dense(l, a, b) : CAL->solve_blocks(l, a, b);
  if (!ok) return -1;
  std::memcpy(dfv, a.data(), a.size() * 8); if (!b.empty()) std::memcpy(dsv, b.data(), b.size() * 8);
  return 0;
}
// Per-observation residuals (sorted order) and the sorted (frame, cam) keys.
int vco_num_obs(void* h) { return (int)CAL->obs.size(); }
void vco_residuals(void* h, double* r, int* frame, int* cam) {
  for (size_t i = 0; i < CAL->obs.size(); ++i) {
    CAL->reproj_value(CAL->obs[i], r + 2 * i);
    if (frame) frame[i] = CAL->obs[i].frame;
    if (cam) cam[i] = CAL->obs[i].cam;
  }
}
// Local Jacobian of one observation i (sorted order): r[2], Jf[12], Jr[6], Jt[6], Jk[2*nk]
void vco_reproj_block(void* h, int i, double* r, double* Jf, double* Jr, double* Jt, double* Jk) {
  CAL->reproj_block_any(CAL->obs[i], r, Jf, Jr, Jt, Jk);
}
// IMU block j (1..N-1): r[9] + J (9 x 33 local: J2(6) J1(6) v2(3) v1(3) g(2) b(6) sf(6) t(1))
void vco_imu_block(void* h, int j, double* r, double* J) {
  Calibrator::ImuJac B;
  CAL->imu_block(j, &B);
  std::memcpy(r, B.r, 72);
  for (int row = 0; row < 9; ++row) {
    double* o = J + row * 33;
    for (int c = 0; c < 6; ++c) { o[c] = B.J2[row * 6 + c]; o[6 + c] = B.J1[row * 6 + c]; o[20 + c] = B.Jb[row * 6 + c]; o[26 + c] = B.Js[row * 6 + c]; }
    for (int c = 0; c < 3; ++c) { o[12 + c] = B.Jv2[row * 3 + c]; o[15 + c] = B.Jv1[row * 3 + c]; }
    o[18] = B.Jg[row * 2]; o[19] = B.Jg[row * 2 + 1]; o[32] = B.Jt[row];
  }
}
void vco_imu_value(void* h, int j, double* r) { CAL->imu_value(j, r); }
// GetRange as doubles: returns count, fills up to cap rows of 7 (w, a, time)
int vco_imu_range(void* h, double t0, double t1, double offset, double* out, int cap) {
  std::vector<ImuMeas<double>> m;
  CAL->imu.range(t0, t1, offset, &m);
  for (size_t i = 0; i < m.size() && (int)i < cap; ++i) { for (int k = 0; k < 3; ++k) { out[7 * i + k] = m[i].w[k]; out[7 * i + 3 + k] = m[i].a[k]; } out[7 * i + 6] = m[i].time; }
  return (int)m.size();
}
void vco_update_imu_weights(void* h) { CAL->update_imu_weights(); }
void vco_get_imu_weights(void* h, double* out) { std::memcpy(out, CAL->imu_w_sqrt.data(), CAL->imu_w_sqrt.size() * 8); }
void vco_set_imu_weights(void* h, const double* in) { std::memcpy(CAL->imu_w_sqrt.data(), in, CAL->imu_w_sqrt.size() * 8); }
void vco_compute_rmse(void* h) { CAL->compute_rmse(); }
void vco_init_gravity(void* h) { CAL->init_gravity(); }

// ---- timing hook for bench.py's cpu_baseline leg ------------------------------------
// Runs `iters` full LM-iteration work units (linearise + block solve + trial cost)
// at the current state and returns seconds.
double vco_time_iterations(void* h, int iters) {
  Calibrator* c = CAL;
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < iters; ++k) {
    c->update_imu_weights();        // the iteration callback's UpdateImuWeights (a no-op unless the inertial terms are fully active)
    c->linearize();
    std::vector<double> hd, lam, a, b;
    c->hdiag(hd);
    lam.resize(hd.size());
    for (size_t i = 0; i < hd.size(); ++i) lam[i] = std::max(hd[i], 1e-6) / 1e4;
    c->solve_blocks(lam, a, b);
    (void)c->evaluate_cost();
  }
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // extern "C"
