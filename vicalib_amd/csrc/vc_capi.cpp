// vc_capi.cpp -- the C entry points of include/vicalib_amd.h over struct vc_calibrator (vc_calibrator.hpp): argument checks, device binding,
// status codes; RCCL communicators; parity / timing hooks; solution covariance; results as text and cameras.xml.
#include "vc_calibrator.hpp"

// =====================================================================================================
extern "C" {

int vc_create(vc_calibrator** out, int device) {
  if (!out) return VC_ERR_BAD_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return VC_ERR_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return VC_ERR_NO_DEVICE;
  vc_calibrator* h = new vc_calibrator();
  h->device = device;
  { const char* e = std::getenv("VICALIB_AMD_GRAPHS"); if (e && e[0] == '1') h->use_graphs = true; }
  { const char* e = std::getenv("VICALIB_AMD_NO_MERGED_DECISION"); if (e && e[0] == '1') h->merged_enabled = false; }
  { const char* e = std::getenv("VICALIB_AMD_OVERLAP_WEIGHTS"); if (e && e[0] == '0') h->serial_weights = true; }
  { const char* e = std::getenv("VICALIB_AMD_JAC_STREAM2"); if (e && e[0] == '0') h->jac_on_stream2 = false; }
  // the hand-over events between the calibrator's two streams order work on ONE device: no system-scope fence at the record
  // (VICALIB_AMD_EVENT_SYSTEM_FENCE=1 restores the default, for A/B measurements)
  unsigned evf = hipEventDisableTiming | hipEventDisableSystemFence;
  { const char* e = std::getenv("VICALIB_AMD_EVENT_SYSTEM_FENCE"); if (e && e[0] == '1') evf = hipEventDisableTiming; }
  // The second stream carries the off-critical-path kernels of a visual-inertial pass (weight update, interval deltas).  It is
  // created with the LOWEST priority: (a) its kernels yield to the chain solve they run beside, and (b) streams of a different
  // priority live in their own pool of hardware queues -- with equal priorities HIP multiplexes all streams of the process onto
  // GPU_MAX_HW_QUEUES (4) queues, and a host program with a few streams of its own (torch with an eagerly created NCCL
  // communicator does it) can land both of ours on ONE queue, which serialises the pass: 0.30 -> 0.46 ms at cfg3, measured.
  int prio_least = 0, prio_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  const char* prio_env = std::getenv("VICALIB_AMD_STREAM2_PRIORITY");      // "default": plain hipStreamCreate (A/B measurements)
  const bool plain2 = prio_env && std::strcmp(prio_env, "default") == 0;
  auto make_stream2 = [&]() -> hipError_t {
    const bool high2 = prio_env && std::strcmp(prio_env, "high") == 0;      // (A/B: the second stream in the HIGHEST priority class instead of the lowest)
    if (!plain2 && prio_least != prio_greatest && hipStreamCreateWithPriority(&h->stream2, hipStreamDefault, high2 ? prio_greatest : prio_least) == hipSuccess) { h->flag_sync = true; return hipSuccess; }
    (void)hipGetLastError();
    return hipStreamCreate(&h->stream2);          // (a runtime without stream priorities: plain stream, same results)
  };
  if (hipStreamCreate(&h->stream) != hipSuccess || make_stream2() != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_state, evf) != hipSuccess || hipEventCreateWithFlags(&h->ev_pre, evf) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_weights, evf) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_imujac, evf) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_reduced, evf) != hipSuccess || hipEventCreateWithFlags(&h->ev_back, evf) != hipSuccess) { delete h; return VC_ERR_NO_DEVICE; }
  // flag hand-overs need the two streams on different hardware queues (a waiting kernel at the head of a shared queue would hold
  // its own producer back): only with the second stream in its own priority class; VICALIB_AMD_FLAG_SYNC=0 keeps the events
  // (=1 forces the flags on whatever the second stream's priority class: the test that puts both streams on one hardware queue)
  { const char* e = std::getenv("VICALIB_AMD_FLAG_SYNC"); if (e && e[0] == '0') h->flag_sync = false; if (e && e[0] == '1') h->flag_sync = true; }
  { const char* e = std::getenv("VICALIB_AMD_SYNC_BOUND"); if (e && std::atoll(e) > 0) h->sync_bound = std::atoll(e); }      // (test hook: a tiny bound forces the time-out path)
  { const char* e = std::getenv("VICALIB_AMD_SYNC_BOUND_FROM_PASS"); if (e) h->sync_bound_from_pass = std::atoi(e); }
  // (the flag words are there whatever the hand-over mode: the counted hand-over inside k_reduced's launch uses two of them)
  if (h->d_sync.alloc(kSyncWords) != hipSuccess || hipMemset(h->d_sync.p, 0, kSyncWords * sizeof(long long)) != hipSuccess) { delete h; return VC_ERR_NO_DEVICE; }
  *out = h;
  return VC_OK;
}
void vc_destroy(vc_calibrator* h) { if (h) (void)hipSetDevice(h->device); delete h; }

int vc_clear(vc_calibrator* h) {
  if (!h) return VC_ERR_BAD_ARG;
  h->stop();
  h->cams.clear(); h->frames.clear(); h->o_frame.clear(); h->o_cam.clear(); h->o_pid.clear(); h->pts.clear(); h->o_pc.clear(); h->o_removed.clear();
  h->mse = 0; h->num_iterations = 0; h->is_bias_active = false; h->is_scale_active = false; h->is_inertial_active = false;
  h->is_visual_active = true; h->rotation_only = true; h->is_finished = false; h->gravity_initialized = false;
  h->outliers_removed = false; h->vis_mult = 0; h->imu_mult = 0; h->wsqrt_frames = 0; h->imu_w.clear(); h->imu_a.clear(); h->imu_t.clear(); h->imu_uploaded = 0; h->imu_end_time = -1.0; h->trace.clear(); h->stage = 0; h->device_dirty = true; h->obs_dirty = true;
  return VC_OK;
}

#define NOT_RUNNING(h) do { if (!(h)) return VC_ERR_BAD_ARG; if ((h)->is_running) return VC_ERR_RUNNING; } while (0)
// every entry point that touches HIP binds the calibrator's device first (several calibrators, one per device, may live in one
// process and be driven from any thread: the CLI's -gpus N)
#define BIND_DEVICE(h) do { if (hipSetDevice((h)->device) != hipSuccess) return VC_ERR_NO_DEVICE; } while (0)

int vc_add_camera(vc_calibrator* h, int model, const double* params, int nparams, int width, int height, const double T_ck[7]) {
  NOT_RUNNING(h);
  const int nk = model_nk(model);
  if (nk < 0 || !params || !T_ck || nparams != nk) return VC_ERR_BAD_ARG;
  if ((int)h->cams.size() >= kMaxCams) return VC_ERR_UNSUPPORTED;
  HostCam c; std::memset(&c, 0, sizeof(c));
  c.model = model; c.nk = nk; c.width = width; c.height = height;
  std::memcpy(c.K, params, nk * 8); std::memcpy(c.T_ck, T_ck, 56);
  h->cams.push_back(c); h->cam_rmse.resize(h->cams.size(), 0.0); h->device_dirty = true;
  return (int)h->cams.size() - 1;
}
int vc_fix_camera_intrinsics(vc_calibrator* h, int should_fix) { NOT_RUNNING(h); h->fix_intrinsics = should_fix != 0; h->device_dirty = true; return VC_OK; }
int vc_add_frame(vc_calibrator* h, const double T_wk[7], double time) {
  NOT_RUNNING(h);
  if (!T_wk) return VC_ERR_BAD_ARG;
  HostFrame f; std::memcpy(f.T, T_wk, 56); f.v[0] = f.v[1] = f.v[2] = 0; f.time = time;
  h->frames.push_back(f); h->device_dirty = true;
  return (int)h->frames.size() - 1;
}
int vc_set_frame_pose(vc_calibrator* h, int frame, const double T_wk[7]) {
  NOT_RUNNING(h);
  if (!T_wk || frame < 0 || frame >= (int)h->frames.size()) return VC_ERR_BAD_ARG;
  std::memcpy(h->frames[frame].T, T_wk, 56); h->device_dirty = true;
  return VC_OK;
}
int vc_pnp_planar(int model, const double* params, int nparams, int n, const double* p_w, const double* p_c, double T_cw[7], double* rms) {
  if (!params || !p_w || !p_c || !T_cw || model_nk(model) < 0 || nparams != model_nk(model)) return VC_ERR_BAD_ARG;
  return pnp_planar(model, params, n, p_w, p_c, T_cw, rms) ? VC_OK : VC_ERR_BAD_ARG;
}
int vc_pnp_planar_ransac(int model, const double* params, int nparams, int n, const double* p_w, const double* p_c, int iterations,
                         double tol_px, double T_cw[7], double* rms, int* n_inliers, char* inlier) {
  if (!params || !p_w || !p_c || !T_cw || model_nk(model) < 0 || nparams != model_nk(model) || iterations < 0 || !(tol_px >= 0.0)) return VC_ERR_BAD_ARG;
  return pnp_planar_ransac(model, params, n, p_w, p_c, iterations, tol_px, T_cw, rms, n_inliers, inlier) ? VC_OK : VC_ERR_BAD_ARG;
}
int vc_target_make_pattern(int rows, int cols, unsigned seed, int* pattern) {
  if (rows < 1 || cols < 1 || !pattern) return VC_ERR_BAD_ARG;
  grid_make_pattern(rows, cols, seed, pattern);
  return VC_OK;
}
int vc_target_find(const double* centres, const double* conics, int n, const int* pattern, int rows, int cols, int* dot_index, int* n_matched) {
  if (!centres || !conics || !pattern || !dot_index || n < 0 || rows < 1 || cols < 1) return VC_ERR_BAD_ARG;
  const int m = grid_find_target(centres, conics, nullptr, n, pattern, rows, cols, dot_index);
  if (n_matched) *n_matched = m;
  return VC_OK;
}
int vc_set_pnp_ransac(vc_calibrator* h, int iterations, double tol_px) {
  NOT_RUNNING(h);
  if (iterations < 0 || !(tol_px >= 0.0)) return VC_ERR_BAD_ARG;
  h->pnp_its = iterations; h->pnp_tol = tol_px;
  return VC_OK;
}
int vc_init_frame_poses_pnp(vc_calibrator* h, int* n_initialised) {
  NOT_RUNNING(h);
  const int N = (int)h->frames.size(), C = (int)h->cams.size();
  // group the observation indices per (frame, camera)
  std::vector<std::vector<int>> view((size_t)N * std::max(C, 1));
  for (size_t i = 0; i < h->o_frame.size(); ++i) view[(size_t)h->o_frame[i] * C + h->o_cam[i]].push_back((int)i);
  int done = 0;
  std::vector<double> pw, pc;
  for (int f = 0; f < N; ++f) {
    bool cam0_good = false, any = false;
    for (int c = 0; c < C; ++c) {
      const std::vector<int>& ids = view[(size_t)f * C + c];
      if (ids.size() < 4) continue;
      if (c != 0 && cam0_good) break;           // `ii == 0 || !tracking_good_[0]` (vicalib-task.cc:341)
      pw.resize(3 * ids.size()); pc.resize(2 * ids.size());
      for (size_t k = 0; k < ids.size(); ++k) {
        std::memcpy(&pw[3 * k], &h->pts.xyz[3 * (size_t)h->o_pid[ids[k]]], 24); std::memcpy(&pc[2 * k], &h->o_pc[2 * (size_t)ids[k]], 16);
      }
      const HostCam& cm = h->cams[c];
      double T_cw[7], rms;
      if (!pnp_planar_ransac(cm.model, cm.K, (int)ids.size(), pw.data(), pc.data(), h->pnp_its, h->pnp_tol, T_cw, &rms, nullptr, nullptr)) continue;
      // T_wk = T_cw^-1 * T_ck  (vicalib-task.cc:344-348)
      const double qi[4] = {-T_cw[0], -T_cw[1], -T_cw[2], T_cw[3]};
      double ti[3], tr[3], T[7];
      const double nt[3] = {-T_cw[4], -T_cw[5], -T_cw[6]};
      quat_rotate(qi, nt, ti);
      quat_mul(qi, cm.T_ck, T);
      quat_rotate(qi, cm.T_ck + 4, tr);
      for (int k = 0; k < 3; ++k) T[4 + k] = ti[k] + tr[k];
      std::memcpy(h->frames[f].T, T, 56);
      any = true;
      if (c == 0) cam0_good = true;
    }
    if (any) ++done;
  }
  if (n_initialised) *n_initialised = done;
  h->device_dirty = true;
  return VC_OK;
}
int vc_add_observations(vc_calibrator* h, int frame, int camera, int n, const double* p_w, const double* p_c) {
  NOT_RUNNING(h);
  if (n < 0 || (n > 0 && (!p_w || !p_c))) return VC_ERR_BAD_ARG;
  if (frame < 0 || frame >= (int)h->frames.size() || camera < 0 || camera >= (int)h->cams.size()) return VC_ERR_BAD_ARG;
  h->o_frame.insert(h->o_frame.end(), n, frame); h->o_cam.insert(h->o_cam.end(), n, camera);
  for (int i = 0; i < n; ++i) h->o_pid.push_back(h->pts.intern(p_w + 3 * (size_t)i));
  h->o_pc.insert(h->o_pc.end(), p_c, p_c + 2 * (size_t)n);
  h->o_removed.insert(h->o_removed.end(), n, 0); h->device_dirty = true; h->obs_dirty = true;
  return VC_OK;
}
int vc_add_observation_tiles(vc_calibrator* h, int n_tiles, const int* tile_frame, const int* tile_cam, const long long* tile_off,
                             const double* points, int n_points, const int* point_id, const double* p_c) {
  NOT_RUNNING(h);
  if (n_tiles < 0 || (n_tiles > 0 && (!tile_frame || !tile_cam || !tile_off || !points || !point_id || !p_c)) || n_points < 0) return VC_ERR_BAD_ARG;
  const int N = (int)h->frames.size(), C = (int)h->cams.size();
  for (int t = 0; t < n_tiles; ++t)
    if (tile_frame[t] < 0 || tile_frame[t] >= N || tile_cam[t] < 0 || tile_cam[t] >= C || tile_off[t + 1] < tile_off[t]) return VC_ERR_BAD_ARG;
  if (n_tiles == 0) return VC_OK;
  if (tile_off[0] < 0) return VC_ERR_BAD_ARG;       // (offsets are monotone: a negative first one would index before the arrays)
  const long long n0 = tile_off[0], n1 = tile_off[n_tiles];
  if ((long long)h->o_frame.size() + (n1 - n0) > 0x7fffffffLL) return VC_ERR_UNSUPPORTED;
  for (long long i = n0; i < n1; ++i) if (point_id[i] < 0 || point_id[i] >= n_points) return VC_ERR_BAD_ARG;
  std::vector<int> remap((size_t)n_points);
  for (int i = 0; i < n_points; ++i) remap[i] = h->pts.intern(points + 3 * (size_t)i);
  const size_t add = (size_t)(n1 - n0), base = h->o_frame.size();
  h->o_frame.resize(base + add); h->o_cam.resize(base + add); h->o_pid.resize(base + add);
  for (int t = 0; t < n_tiles; ++t)
    for (long long i = tile_off[t]; i < tile_off[t + 1]; ++i) {
      const size_t k = base + (size_t)(i - n0);
      h->o_frame[k] = tile_frame[t]; h->o_cam[k] = tile_cam[t]; h->o_pid[k] = remap[point_id[i]];
    }
  h->o_pc.insert(h->o_pc.end(), p_c + 2 * n0, p_c + 2 * n1);
  h->o_removed.insert(h->o_removed.end(), add, 0); h->device_dirty = true; h->obs_dirty = true;
  return VC_OK;
}
int vc_add_imu(vc_calibrator* h, int n, const double* gyro, const double* accel, const double* time) {
  NOT_RUNNING(h);
  if (n < 0 || (n > 0 && (!gyro || !accel || !time))) return VC_ERR_BAD_ARG;
  for (int i = 0; i < n; ++i) {
    if (!(time[i] > h->imu_end_time)) return VC_ERR_TIME_ORDER;
    h->imu_w.insert(h->imu_w.end(), gyro + 3 * i, gyro + 3 * i + 3); h->imu_a.insert(h->imu_a.end(), accel + 3 * i, accel + 3 * i + 3);
    h->imu_t.push_back(time[i]); h->imu_end_time = time[i];
  }
  h->device_dirty = true;
  return VC_OK;
}
int vc_set_sigmas(vc_calibrator* h, double g, double a) { NOT_RUNNING(h); h->gyro_sigma = g; h->accel_sigma = a; h->device_dirty = true; return VC_OK; }
int vc_set_biases(vc_calibrator* h, const double b[6]) { NOT_RUNNING(h); if (!b) return VC_ERR_BAD_ARG; std::memcpy(h->biases, b, 48); h->device_dirty = true; return VC_OK; }
int vc_set_scale_factor(vc_calibrator* h, const double s[6]) { NOT_RUNNING(h); if (!s) return VC_ERR_BAD_ARG; std::memcpy(h->scale, s, 48); h->device_dirty = true; return VC_OK; }
int vc_set_time_offset(vc_calibrator* h, double o) { NOT_RUNNING(h); h->time_offset = o; h->device_dirty = true; return VC_OK; }
int vc_set_function_tolerance(vc_calibrator* h, double t) { NOT_RUNNING(h); h->function_tolerance = t; return VC_OK; }
int vc_set_optimization_flags(vc_calibrator* h, int bias, int inertial, int rot_only, int toff) {
  NOT_RUNNING(h);
  h->is_scale_active = bias != 0; h->is_bias_active = bias != 0; h->is_inertial_active = inertial != 0;
  h->rotation_only = rot_only != 0; h->optimize_time_offset = toff != 0; h->device_dirty = true;
  return VC_OK;
}
int vc_set_gravity(vc_calibrator* h, const double g_dir[2]) {
  NOT_RUNNING(h);
  if (!g_dir) return VC_ERR_BAD_ARG;
  { std::lock_guard<std::mutex> lk(h->result_mutex); h->g_dir[0] = g_dir[0]; h->g_dir[1] = g_dir[1]; }
  h->gravity_initialized = true; h->device_dirty = true;
  return VC_OK;
}
int vc_set_frame_velocities(vc_calibrator* h, const double* v_w, int n) {
  NOT_RUNNING(h);
  if (!v_w || n != (int)h->frames.size()) return VC_ERR_BAD_ARG;
  for (int f = 0; f < n; ++f) std::memcpy(h->frames[f].v, v_w + 3 * (size_t)f, 24);
  h->device_dirty = true;
  return VC_OK;
}
int vc_set_tolerances(vc_calibrator* h, double gradient_tolerance, double parameter_tolerance) {
  NOT_RUNNING(h);
  h->gradient_tolerance = gradient_tolerance; h->parameter_tolerance = parameter_tolerance;
  return VC_OK;
}
int vc_set_max_iters(vc_calibrator* h, int m) { NOT_RUNNING(h); h->max_iters = m; return VC_OK; }
int vc_set_calibrate_imu(vc_calibrator* h, int c) { NOT_RUNNING(h); h->calibrate_imu = c != 0; return VC_OK; }
int vc_set_remove_outliers(vc_calibrator* h, int r, double th) { NOT_RUNNING(h); h->remove_outliers = r != 0; h->outlier_threshold = th; return VC_OK; }

int vc_solve(vc_calibrator* h) {
  NOT_RUNNING(h);
  h->should_run = true; h->is_running = true;
  const int rc = h->solve();
  h->is_running = false;
  return rc;
}
int vc_start(vc_calibrator* h) {
  NOT_RUNNING(h);
  if (h->worker.joinable()) h->worker.join();
  h->should_run = true; h->is_running = true;
  h->worker = std::thread([h]() { (void)h->solve(); h->is_running = false; });
  return VC_OK;
}
int vc_set_stage_limit(vc_calibrator* h, int n) { NOT_RUNNING(h); h->stage_limit = n; return VC_OK; }
int vc_sync_timeouts(const vc_calibrator* h) { return h ? h->sync_timeouts : 0; }
int vc_set_kernel_timing(vc_calibrator* h, int on) {
  NOT_RUNNING(h);
  h->ktime_on = on != 0;
  for (double& t : h->kt_total_ms) t = 0.0;
  for (long& c : h->kt_count) c = 0;
  return VC_OK;
}
int vc_get_kernel_timing(vc_calibrator* h, char* names, int names_len, double* total_ms, long long* count, int max_entries) {
  NOT_RUNNING(h);
  if (!names || !total_ms || !count || names_len <= 0) return VC_ERR_BAD_ARG;
  std::string joined;
  int n = 0;
  for (size_t i = 0; i < h->kt_names.size() && n < max_entries; ++i) {
    if (h->kt_count[i] == 0) continue;
    if (!joined.empty()) joined += ";";
    joined += h->kt_names[i];
    total_ms[n] = h->kt_total_ms[i]; count[n] = h->kt_count[i]; ++n;
  }
  if ((int)joined.size() + 1 > names_len) return VC_ERR_BAD_ARG;
  std::memcpy(names, joined.c_str(), joined.size() + 1);
  return n;
}
int vc_resume(vc_calibrator* h) { NOT_RUNNING(h); h->is_finished = false; return VC_OK; }
int vc_is_running(vc_calibrator* h) { return h && h->is_running && !h->is_finished; }
int vc_stop(vc_calibrator* h) { if (!h) return VC_ERR_BAD_ARG; h->stop(); return VC_OK; }

int vc_num_frames(vc_calibrator* h) { return h ? (int)h->frames.size() : VC_ERR_BAD_ARG; }
int vc_num_cameras(vc_calibrator* h) { return h ? (int)h->cams.size() : VC_ERR_BAD_ARG; }
int vc_get_camera(vc_calibrator* h, int c, double* params, int* nparams, double T_ck[7]) {
  if (!h || c < 0 || c >= (int)h->cams.size()) return VC_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(h->result_mutex);
  if (params) std::memcpy(params, h->cams[c].K, h->cams[c].nk * 8);
  if (nparams) *nparams = h->cams[c].nk;
  if (T_ck) std::memcpy(T_ck, h->cams[c].T_ck, 56);
  return VC_OK;
}
int vc_get_frame(vc_calibrator* h, int f, double T_wk[7], double v_w[3], double* time) {
  if (!h || f < 0 || f >= (int)h->frames.size()) return VC_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(h->result_mutex);
  if (T_wk) std::memcpy(T_wk, h->frames[f].T, 56);
  if (v_w) std::memcpy(v_w, h->frames[f].v, 24);
  if (time) *time = h->frames[f].time;
  return VC_OK;
}
int vc_get_biases(vc_calibrator* h, double b[6]) { if (!h || !b) return VC_ERR_BAD_ARG; std::lock_guard<std::mutex> lk(h->result_mutex); std::memcpy(b, h->biases, 48); return VC_OK; }
int vc_get_scale_factor(vc_calibrator* h, double s[6]) { if (!h || !s) return VC_ERR_BAD_ARG; std::lock_guard<std::mutex> lk(h->result_mutex); std::memcpy(s, h->scale, 48); return VC_OK; }
int vc_get_gravity(vc_calibrator* h, double g[2]) { if (!h || !g) return VC_ERR_BAD_ARG; std::lock_guard<std::mutex> lk(h->result_mutex); std::memcpy(g, h->g_dir, 16); return VC_OK; }
double vc_time_offset(vc_calibrator* h) { if (!h) return 0.0; std::lock_guard<std::mutex> lk(h->result_mutex); return h->time_offset; }
double vc_mean_squared_error(vc_calibrator* h) { if (!h) return 0.0; std::lock_guard<std::mutex> lk(h->result_mutex); return h->mse; }
int vc_get_camera_proj_rmse(vc_calibrator* h, double* rmse) {
  if (!h || !rmse) return VC_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(h->result_mutex);
  for (size_t c = 0; c < h->cams.size(); ++c) rmse[c] = c < h->cam_rmse.size() ? h->cam_rmse[c] : 0.0;
  return VC_OK;
}
unsigned vc_get_num_iterations(vc_calibrator* h) { return h ? h->num_iterations.load() : 0u; }
// imu_buffer() :487 -- the stored measurements, in time order
int vc_num_imu_measurements(vc_calibrator* h) { return h ? (int)h->imu_t.size() : VC_ERR_BAD_ARG; }
int vc_get_imu_measurements(vc_calibrator* h, double* gyro, double* accel, double* time, int max_n) {
  if (!h || max_n < 0) return VC_ERR_BAD_ARG;
  const int n = std::min<int>(max_n, (int)h->imu_t.size());
  if (gyro) std::memcpy(gyro, h->imu_w.data(), (size_t)n * 24);
  if (accel) std::memcpy(accel, h->imu_a.data(), (size_t)n * 24);
  if (time) std::memcpy(time, h->imu_t.data(), (size_t)n * 8);
  return n;
}
// GetIntegrationPoses(id) :508-533: the poses the IMU integration passes through between frame id and frame id + 1 (the GUI draws
// them): the start pose, then one pose per measurement of the range (ceres-cost-functions.h:200-227).  Rows of 11 doubles
// [q(4) t(3) v_w(3) time]; empty unless the inertial terms are fully active (:510).  Host arithmetic (vc_imu.hpp).
int vc_get_integration_poses(vc_calibrator* h, int id, double* poses, int max_poses) {
  if (!h || id < 0 || (max_poses > 0 && !poses)) return VC_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(h->result_mutex);
  if (!(h->is_inertial_active && !h->rotation_only)) return 0;
  if (id + 1 >= (int)h->frames.size()) return 0;
  const ImuView buf = {h->imu_t.data(), h->imu_w.data(), h->imu_a.data(), (int)h->imu_t.size(), imu_average_dt(h->imu_t.data(), (int)h->imu_t.size())};
  const HostFrame& f1 = h->frames[id];
  const HostFrame& f2 = h->frames[id + 1];
  const ImuRange rg = imu_range(buf, f1.time, f2.time, h->time_offset);
  if (!rg.valid) return 0;
  double gw[3];
  imu_gravity<double>(h->g_dir, gw);
  PoseV<double> s;
  for (int i = 0; i < 4; ++i) s.q[i] = f1.T[i];
  for (int i = 0; i < 3; ++i) { s.p[i] = f1.T[4 + i]; s.v[i] = f1.v[i]; }
  const int n_meas = (rg.k1 - rg.k0 + 1) + 2;
  int n = 0;
  auto push = [&](double time) {
    if (n < max_poses) { double* o = poses + 11 * (size_t)n; std::memcpy(o, s.q, 32); std::memcpy(o + 4, s.p, 24); std::memcpy(o + 7, s.v, 24); o[10] = time; }
    ++n;
  };
  push(f1.time);
  Meas<double> z0, z1;
  imu_range_get<double>(buf, rg, h->time_offset, f1.time, f2.time, 0, &z0);
  for (int m = 1; m < n_meas; ++m) {
    imu_range_get<double>(buf, rg, h->time_offset, f1.time, f2.time, m, &z1);
    imu_rk4_step<double>(&s, z0, z1, h->biases, h->scale, gw);
    push(z1.time);
    z0 = z1;
  }
  return n;
}
// PrintResults() :536-544 into a caller's buffer: per camera its parameters and T_ck as a 4 x 4 matrix
int vc_print_results(vc_calibrator* h, char* buf, int len) {
  if (!h || len < 0 || (len > 0 && !buf)) return VC_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(h->result_mutex);
  std::string out = "------------------------------------------\n";
  char line[512];
  for (size_t c = 0; c < h->cams.size(); ++c) {
    const HostCam& cm = h->cams[c];
    std::snprintf(line, sizeof(line), "Camera: %zu\n", c); out += line;
    for (int i = 0; i < cm.nk; ++i) { std::snprintf(line, sizeof(line), "%s%.10g", i ? " " : "", cm.K[i]); out += line; }
    out += "\n";
    double R[9];
    quat_to_R(cm.T_ck, R);
    for (int i = 0; i < 3; ++i) { std::snprintf(line, sizeof(line), "%.10g %.10g %.10g %.10g\n", R[3 * i], R[3 * i + 1], R[3 * i + 2], cm.T_ck[4 + i]); out += line; }
    out += "0 0 0 1\n\n";
  }
  if (len == 0) return (int)out.size();          // length query: the text needs a buffer of this + 1 bytes
  if ((int)out.size() + 1 > len) return VC_ERR_BAD_ARG;
  std::memcpy(buf, out.c_str(), out.size() + 1);
  return (int)out.size();
}

// WriteCameraModels, vicalibrator.h:208-229 + calibu WriteXmlRig layout
int vc_write_camera_models(vc_calibrator* h, const char* filename) {
  if (!h || !filename) return VC_ERR_BAD_ARG;
  FILE* f = std::fopen(filename, "w");
  if (!f) return VC_ERR_BAD_ARG;
  static const char* kType[] = {"calibu_fu_fv_u0_v0_w", "calibu_fu_fv_u0_v0_k1_k2", "calibu_fu_fv_u0_v0_k1_k2_k3",
                                "calibu_fu_fv_u0_v0_kb4", "calibu_fu_fv_u0_v0", "calibu_fu_fv_u0_v0_rational6"};   // vicalib-engine.cc:210-260
  std::lock_guard<std::mutex> lk(h->result_mutex);
  const bool robotics = h->calibrate_imu;     // FLAGS_calibrate_imu selects RdfRobotics (:214-219)
  const double rdf_rob[9] = {0, 1, 0, 0, 0, 1, 1, 0, 0}, rdf_vis[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  const double* rdf = robotics ? rdf_rob : rdf_vis;
  std::fprintf(f, "<rig>\n");
  for (size_t c = 0; c < h->cams.size(); ++c) {
    const HostCam& cm = h->cams[c];
    // pose = T_ck^-1 (* SE3(RdfRobotics^-1, 0))
    double R[9], Rt[9], t[3], M[9];
    quat_to_R(cm.T_ck, R);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rt[3 * i + j] = R[3 * j + i];
    for (int i = 0; i < 3; ++i) t[i] = -(Rt[3 * i] * cm.T_ck[4] + Rt[3 * i + 1] * cm.T_ck[5] + Rt[3 * i + 2] * cm.T_ck[6]);
    if (robotics) {   // Rt * rdf^-1 = Rt * rdf^T
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[3 * i + j] = Rt[3 * i] * rdf[3 * j] + Rt[3 * i + 1] * rdf[3 * j + 1] + Rt[3 * i + 2] * rdf[3 * j + 2];
    } else std::memcpy(M, Rt, sizeof(M));
    std::fprintf(f, "  <camera>\n    <camera_model name=\"\" index=\"%zu\" serialno=\"-1\" type=\"%s\" version=\"8\">\n", c, kType[cm.model]);
    std::fprintf(f, "      <width> %d </width>\n      <height> %d </height>\n", cm.width, cm.height);
    std::fprintf(f, "      <right> [ %g; %g; %g ] </right>\n      <down> [ %g; %g; %g ] </down>\n      <forward> [ %g; %g; %g ] </forward>\n",
                 rdf[0], rdf[1], rdf[2], rdf[3], rdf[4], rdf[5], rdf[6], rdf[7], rdf[8]);
    std::fprintf(f, "      <params> [ ");
    for (int i = 0; i < cm.nk; ++i) std::fprintf(f, "%.17g%s", cm.K[i], i + 1 < cm.nk ? "; " : " ");
    std::fprintf(f, "] </params>\n    </camera_model>\n    <pose>\n      <T_wc> [ %.17g, %.17g, %.17g, %.17g; %.17g, %.17g, %.17g, %.17g; %.17g, %.17g, %.17g, %.17g ] </T_wc>\n    </pose>\n  </camera>\n",
                 M[0], M[1], M[2], t[0], M[3], M[4], M[5], t[1], M[6], M[7], M[8], t[2]);
  }
  std::fprintf(f, "</rig>\n");
  std::fclose(f);
  return VC_OK;
}

// ---- engine-level ------------------------------------------------------------------------------------
int vc_trace_len(vc_calibrator* h) { if (!h) return VC_ERR_BAD_ARG; std::lock_guard<std::mutex> lk(h->result_mutex); return (int)h->trace.size(); }
int vc_get_trace(vc_calibrator* h, double* rows, int max_rows) {
  if (!h || !rows) return VC_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(h->result_mutex);
  const int n = std::min<int>(max_rows, (int)h->trace.size());
  for (int i = 0; i < n; ++i) {
    const IterRecord& r = h->trace[i];
    double* o = rows + 10 * i;
    o[0] = r.iteration; o[1] = r.cost; o[2] = r.cost_change; o[3] = r.gmax; o[4] = r.gnorm; o[5] = r.step_norm; o[6] = r.rho;
    o[7] = r.radius; o[8] = r.accepted; o[9] = r.stage;
  }
  return n;
}
int vc_set_shard(vc_calibrator* h, int rank, int world_size, vc_allreduce_fn fn, void* ctx) {
  NOT_RUNNING(h);
  if (world_size < 1 || rank < 0 || rank >= world_size || (world_size > 1 && !fn)) return VC_ERR_BAD_ARG;
  // a callback replaces the library's own communicator (a caller that falls back after vc_set_shard_rccl succeeded on this rank but
  // failed on another one must end up on the same transport everywhere)
  h->drop_comm();
  h->rank = rank; h->world = world_size; h->allreduce = fn; h->allreduce_ctx = ctx; h->device_dirty = true;
  { const char* e = std::getenv("VICALIB_AMD_FORCE_SHARD_PATH"); h->force_shard_path = e && e[0] == '1'; }
  return VC_OK;
}
int vc_rccl_unique_id(void* out128) {
  if (!out128) return VC_ERR_BAD_ARG;
  if (!g_rccl.load()) return VC_ERR_UNSUPPORTED;
  RcclUniqueId id;
  if (g_rccl.GetUniqueId(&id) != 0) return VC_ERR_NO_DEVICE;
  std::memcpy(out128, &id, sizeof(id));
  return VC_OK;
}
// ncclCommInitRank with the failure text a launcher prints before it falls back (vc_last_error)
static int rccl_comm_init(int device, int rank, int world_size, const void* unique_id128, void** comm) {
  g_last_error.clear();          // (vc_last_error() is about THIS call from here on)
  *comm = nullptr;
  if (world_size < 1 || rank < 0 || rank >= world_size || !unique_id128) return VC_ERR_BAD_ARG;
  if (!g_rccl.load()) {
    const char* de = dlerror();      // (one call: dlerror() clears the message it returns)
    g_last_error = std::string("librccl could not be loaded: ") + (de ? de : "ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy not all found");
    return VC_ERR_UNSUPPORTED;
  }
  if (hipSetDevice(device) != hipSuccess) { g_last_error = "hipSetDevice(" + std::to_string(device) + ") failed"; return VC_ERR_NO_DEVICE; }
  RcclUniqueId id;
  std::memcpy(&id, unique_id128, sizeof(id));
  const int nrc = g_rccl.CommInitRank(comm, world_size, id, rank);
  if (nrc != 0) {
    g_last_error = "ncclCommInitRank(rank " + std::to_string(rank) + " of " + std::to_string(world_size) + ", device " + std::to_string(device) + ") = " + std::to_string(nrc);
    if (g_rccl.GetErrorString) g_last_error += std::string(" (") + g_rccl.GetErrorString(nrc) + ")";
    if (g_rccl.GetLastError) { const char* le = g_rccl.GetLastError(nullptr); if (le && le[0]) g_last_error += std::string(": ") + le; }
    *comm = nullptr; return VC_ERR_NO_DEVICE;
  }
  // what RCCL itself says it built: a communicator of another size or with this process at another rank (a launcher that mixed up its
  // environment) would reduce over the wrong set of shards without any error
  int cnt = -1, ur = -1;
  if (g_rccl.CommCount && g_rccl.CommCount(*comm, &cnt) == 0 && g_rccl.CommUserRank && g_rccl.CommUserRank(*comm, &ur) == 0 && (cnt != world_size || ur != rank)) {
    g_last_error = "ncclCommInitRank built a communicator of " + std::to_string(cnt) + " ranks with this process at rank " + std::to_string(ur) + "; asked for rank " +
                   std::to_string(rank) + " of " + std::to_string(world_size);
    (void)g_rccl.CommDestroy(*comm); *comm = nullptr; return VC_ERR_BAD_ARG;
  }
  return VC_OK;
}
static void attach_rccl(vc_calibrator* h, int rank, int world_size, void* comm, bool owned) {
  h->drop_comm();
  h->rccl_comm = comm; h->rccl_comm_owned = owned;
  h->rank = rank; h->world = world_size; h->allreduce = nullptr; h->allreduce_ctx = nullptr; h->device_dirty = true;
  { const char* e = std::getenv("VICALIB_AMD_FORCE_SHARD_PATH"); h->force_shard_path = e && e[0] == '1'; }
  // an RCCL communicator of several ranks has one device per rank: this process has its device to itself, the cross-stream hand-overs of
  // the pass can go through device flags as in a single-process solve (-25 us per pass and rank; VICALIB_AMD_SHARD_FLAG_SYNC=0 keeps events)
  { const char* e = std::getenv("VICALIB_AMD_SHARD_FLAG_SYNC"); if (world_size > 1 && !(e && e[0] == '0')) h->shard_flag_sync = true; }
}
int vc_set_shard_rccl(vc_calibrator* h, int rank, int world_size, const void* unique_id128) {
  g_last_error.clear();
  NOT_RUNNING(h);
  h->drop_comm();
  void* comm = nullptr;
  const int rc = rccl_comm_init(h->device, rank, world_size, unique_id128, &comm);
  if (rc) return rc;
  attach_rccl(h, rank, world_size, comm, true);
  return VC_OK;
}
// One communicator for all calibrators of a process (a launcher that runs several solves in a row -- bench.py builds three
// calibrators -- pays ONE ncclCommInitRank and one ncclCommDestroy, and their order across the ranks is the launcher's, not that
// of three destructors): created once, lent to calibrators with vc_set_shard_comm, destroyed by the caller after the calibrators.
struct vc_shard_comm { void* comm; int device, rank, world; };
int vc_shard_comm_create(int device, int rank, int world_size, const void* unique_id128, vc_shard_comm** out) {
  if (!out) return VC_ERR_BAD_ARG;
  *out = nullptr;
  void* comm = nullptr;
  const int rc = rccl_comm_init(device, rank, world_size, unique_id128, &comm);
  if (rc) return rc;
  *out = new vc_shard_comm{comm, device, rank, world_size};
  return VC_OK;
}
int vc_set_shard_comm(vc_calibrator* h, vc_shard_comm* c) {
  g_last_error.clear();
  NOT_RUNNING(h);
  if (!c || !c->comm) return VC_ERR_BAD_ARG;
  if (c->device != h->device) { g_last_error = "vc_set_shard_comm: the communicator lives on device " + std::to_string(c->device) + ", the calibrator on " + std::to_string(h->device); return VC_ERR_BAD_ARG; }
  attach_rccl(h, c->rank, c->world, c->comm, false);
  return VC_OK;
}
void vc_shard_comm_destroy(vc_shard_comm* c) {
  if (!c) return;
  if (c->comm && g_rccl.CommDestroy) { (void)hipSetDevice(c->device); (void)hipDeviceSynchronize(); (void)g_rccl.CommDestroy(c->comm); }
  delete c;
}
long long vc_allreduce_calls(vc_calibrator* h) { return h ? h->rccl_calls : 0; }
int vc_shard_info(vc_calibrator* h, int* rank, int* world_size, int* rccl_ranks, int* rccl_rank) {
  if (!h) return VC_ERR_BAD_ARG;
  if (rank) *rank = h->rank;
  if (world_size) *world_size = h->world;
  int cnt = -1, ur = -1;
  if (h->rccl_comm && g_rccl.CommCount && g_rccl.CommUserRank) {
    if (g_rccl.CommCount(h->rccl_comm, &cnt) != 0) cnt = -1;
    if (g_rccl.CommUserRank(h->rccl_comm, &ur) != 0) ur = -1;
  }
  if (rccl_ranks) *rccl_ranks = cnt;
  if (rccl_rank) *rccl_rank = ur;
  return VC_OK;
}
const char* vc_last_error(void) { return g_last_error.c_str(); }
int vc_pass_paths(vc_calibrator* h, int* out6) {
  if (!h || !out6) return VC_ERR_BAD_ARG;
  out6[0] = h->dv.imu_on ? h->dv.fold_l0 : 0; out6[1] = h->dv.imu_on ? h->dv.back_path : 0;
  out6[2] = (h->dv.imu_on && h->dv.gram_top_stride > 0) ? 1 : 0; out6[3] = h->top_gram_launch ? 1 : 0;
  // (decided per pass by enqueue_pass from the same predicates and switches)
  static const bool defer_env = [] { const char* e = std::getenv("VICALIB_AMD_DEFER_TAIL"); return !(e && e[0] == '0'); }();
  static const bool hadd_env = [] { const char* e = std::getenv("VICALIB_AMD_HADD_EARLY"); return !(e && e[0] == '0'); }();
  out6[4] = (h->dv.imu_on && defer_env && chain_back_is_path(h->dv)) ? 1 : 0;
  out6[5] = (h->dv.imu_on && hadd_env && chain_hadd_early(h->dv)) ? 1 : 0;
  return VC_OK;
}
void* vc_get_stream(vc_calibrator* h) { return h ? (void*)h->stream : nullptr; }
int vc_prepare(vc_calibrator* h) {
  NOT_RUNNING(h);
  if (h->vis_mult == 0) h->vis_mult = 1;
  if (h->imu_on() && h->imu_mult == 0) h->imu_mult = 1;
  return h->upload();
}
int vc_shared_dim(vc_calibrator* h) { return h ? h->dv.D : VC_ERR_BAD_ARG; }
int vc_linearize(vc_calibrator* h, double* cost, double* Hpp, double* gp, double* S, double* g_red, double* hss_diag, double* g_s) {
  NOT_RUNNING(h);
  BIND_DEVICE(h);
  if (h->device_dirty) { int rc = vc_prepare(h); if (rc) return rc; }
  double lin_cost = 0;
  // radius = +inf-like: lambda -> ~0 so that L L^T = H_pp to rounding; S is stored undamped anyway
  int rc = h->linearize_hold(1e300, &lin_cost); if (rc) return rc;
  const int N = h->dv.n_frames, D = h->dv.D;
  if (cost) *cost = lin_cost;
  std::vector<double> fr((size_t)N * kFrStride);
  if ((Hpp || gp) && N) {
    if (hipMemcpy(fr.data(), h->dv.fr, fr.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return VC_ERR_NO_DEVICE;
    for (int f = 0; f < N; ++f) {
      const double* p = &fr[(size_t)f * kFrStride];
      if (gp) std::memcpy(gp + 6 * (size_t)f, p + kFrG, 48);
      if (Hpp) {   // H_pp + lambda = L L^T
        double L[36] = {0}; int k = 0;
        for (int i = 0; i < 6; ++i) for (int j = 0; j <= i; ++j) L[i * 6 + j] = p[kFrL + k++];
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) {
          double s = 0; for (int q = 0; q < 6; ++q) s += L[i * 6 + q] * L[j * 6 + q];
          Hpp[36 * (size_t)f + i * 6 + j] = s - (i == j ? p[kFrLam + i] : 0.0);
        }
      }
    }
  }
  std::vector<double> sb((size_t)D * D + 3 * D + 2);
  if (hipMemcpy(sb.data(), h->dv.Sbuf, sb.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return VC_ERR_NO_DEVICE;
  if (S) std::memcpy(S, sb.data(), (size_t)D * D * 8);
  if (g_red) std::memcpy(g_red, sb.data() + (size_t)D * D, D * 8);
  if (hss_diag) std::memcpy(hss_diag, sb.data() + (size_t)D * D + D, D * 8);
  if (g_s) std::memcpy(g_s, sb.data() + (size_t)D * D + 2 * D, D * 8);
  return VC_OK;
}
int vc_evaluate(vc_calibrator* h, double* cost, double* sum_sq) {
  NOT_RUNNING(h);
  BIND_DEVICE(h);
  if (h->device_dirty) { int rc = vc_prepare(h); if (rc) return rc; }
  launch_reproj_res(h->dv, h->cur, (double)h->vis_mult, h->stream);
  launch_sum_tile_cost(h->dv, h->d_tmp.p, h->stream);
  double out[2];
  if (hipMemcpyAsync(out, h->d_tmp.p, 16, hipMemcpyDeviceToHost, h->stream) != hipSuccess) return VC_ERR_NO_DEVICE;
  if (hipStreamSynchronize(h->stream) != hipSuccess) return VC_ERR_NO_DEVICE;
  if (cost) *cost = out[0];
  if (sum_sq) *sum_sq = out[1];
  return VC_OK;
}
int vc_run_iterations(vc_calibrator* h, int iters, int* jac_sweeps, int* res_sweeps) {
  NOT_RUNNING(h);
  BIND_DEVICE(h);
  if (h->device_dirty) { int rc = vc_prepare(h); if (rc) return rc; }
  // Exactly `iters` LM iterations of the real solver: complete solves (termination tests on) run back to
  // back from the uploaded initial state; the last one is cut by max_iters so the total is exact.
  const int mi = h->max_iters;
  const long j0 = h->jac_sweeps, r0 = h->res_sweeps;
  h->should_run = true;
  ++h->solve_epoch;
  int done = 0, rc = VC_OK, guard = 0;
  while (done < iters && guard++ < iters + 4) {
    h->max_iters = std::min(mi, iters - done);
    const auto ts0 = std::chrono::steady_clock::now();
    rc = h->reset_state(); if (rc) break;
    Termination t; double fc; long nr;
    rc = h->solve_once(&t, &fc, &nr);
    if (rc) break;
    const int ran = h->last_iters;
    if (std::getenv("VICALIB_AMD_TIMING"))
      std::fprintf(stderr, "[vicalib_amd]   run_iterations: solve of %d iterations in %.3f ms\n", ran, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts0).count());
    done += std::max(ran, 1);
  }
  h->max_iters = mi;
  if (jac_sweeps) *jac_sweeps = (int)(h->jac_sweeps - j0);
  if (res_sweeps) *res_sweeps = (int)(h->res_sweeps - r0);
  return rc ? rc : done;
}
int vc_time_kernels(vc_calibrator* h, int reps, double* jac_ms, double* res_ms) {
  NOT_RUNNING(h);
  BIND_DEVICE(h);
  if (h->device_dirty) { int rc = vc_prepare(h); if (rc) return rc; }
  EventSet<3> evs;
  if (!evs.create()) return VC_ERR_NO_DEVICE;
  hipEvent_t e0 = evs.e[0], e1 = evs.e[1], e2 = evs.e[2];
  Ctrl c; h->init_ctrl(&c); c.hold = 1; if (c.mult < 1) c.mult = 1;
  if (hipMemcpy(h->d_ctrl.p, &c, sizeof(Ctrl), hipMemcpyHostToDevice) != hipSuccess) return VC_ERR_NO_DEVICE;
  h->dv.merged = 0; h->dv.par = 0; h->dv.ctrl = h->d_ctrl.p; h->dv.ctrl_prev = h->d_ctrl.p + 1;
  const double mult = c.mult;
  launch_reproj_jac(h->dv, h->stream); launch_reproj_res(h->dv, h->cur, mult, h->stream);   // warm
  (void)hipEventRecord(e0, h->stream);
  for (int i = 0; i < reps; ++i) launch_reproj_jac(h->dv, h->stream);
  (void)hipEventRecord(e1, h->stream);
  for (int i = 0; i < reps; ++i) launch_reproj_res(h->dv, h->cur, mult, h->stream);
  (void)hipEventRecord(e2, h->stream);
  if (hipEventSynchronize(e2) != hipSuccess) return VC_ERR_NO_DEVICE;
  float m1 = 0, m2 = 0;
  (void)hipEventElapsedTime(&m1, e0, e1); (void)hipEventElapsedTime(&m2, e1, e2);
  if (jac_ms) *jac_ms = m1 / reps;
  if (res_ms) *res_ms = m2 / reps;
  return VC_OK;
}
// Average ms per launch of every stage of one pass, each launched `reps` times back to back with the
// decision logic on hold (state does not change).  out[0..5]: jac, frame_prep, schur_reduce, reduced, trial, final
int vc_time_stages(vc_calibrator* h, int reps, double* out) {
  NOT_RUNNING(h);
  BIND_DEVICE(h);
  if (!out) return VC_ERR_BAD_ARG;
  if (h->device_dirty) { int rc = vc_prepare(h); if (rc) return rc; }
  Ctrl c; h->init_ctrl(&c); c.hold = 1; c.first = 0; if (c.mult < 1) c.mult = 1;
  if (hipMemcpy(h->d_ctrl.p, &c, sizeof(Ctrl), hipMemcpyHostToDevice) != hipSuccess) return VC_ERR_NO_DEVICE;
  // stage timing uses the stand-alone kernels of the unmerged pipeline (k_final as its own launch)
  const bool was_merged = h->merged_enabled;
  h->merged_enabled = false;
  const int rc_pass = h->enqueue_pass(true, true);
  h->merged_enabled = was_merged;
  if (rc_pass) return VC_ERR_NO_DEVICE;
  h->dv.sync_seq = 0; h->dv.final_wait = 0; h->dv.block_wait = 0;      // the stand-alone launches below neither signal nor wait for the other stream
  h->dv.tail_deferred = 0; h->dv.hadd_early = 0; h->dv.part_ride = 0;                         // ... and k_reduced runs its own tail and forms the shared parameters' blocks itself
  EventSet<7> evs;
  if (!evs.create()) return VC_ERR_NO_DEVICE;
  hipEvent_t* ev = evs.e;
  hipStream_t s = h->stream;
  for (int w = 0; w < 2; ++w) {   // first round warms clocks and caches
    (void)hipEventRecord(ev[0], s); for (int i = 0; i < reps; ++i) launch_reproj_jac(h->dv, s);
    (void)hipEventRecord(ev[1], s); for (int i = 0; i < reps; ++i) launch_frame_schur(h->dv, s);
    (void)hipEventRecord(ev[2], s);
    // (k_reduced adds the camera blocks to the S k_part_sum left in Sbuf: repeated without it, S only grows more positive definite --
    // the timing is unaffected and the held state ignores the step)
    (void)hipEventRecord(ev[3], s); for (int i = 0; i < reps; ++i) launch_reduced(h->dv, 0, s);
    (void)hipEventRecord(ev[4], s); for (int i = 0; i < reps; ++i) launch_trial(h->dv, s);
    (void)hipEventRecord(ev[5], s); for (int i = 0; i < reps; ++i) launch_final(h->dv, 0, s);
    (void)hipEventRecord(ev[6], s);
    if (hipEventSynchronize(ev[6]) != hipSuccess) return VC_ERR_NO_DEVICE;
  }
  for (int i = 0; i < 6; ++i) { float ms = 0; (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]); out[i] = ms / reps; }
  return VC_OK;
}
// Weighted J^T J (33 x 33), J^T r (33) and cost of every IMU block after vc_linearize, columns
// [frame j: pose 6, velocity 3 | frame j-1: pose 6, velocity 3 | g 2, b 6, sf 6, time offset 1]
int vc_get_imu_blocks(vc_calibrator* h, double* H, double* g, double* cost) {
  NOT_RUNNING(h);
  BIND_DEVICE(h);
  if (!h->dv.imu_on) return VC_ERR_BAD_ARG;
  const size_t ns = (size_t)std::max(h->dv.n_frames - 1, 0);
  const int b = h->cur;
  if ((H || g) && ns) {                          // the device keeps the blocks compact (vc_device.h: kSeg*): unfold them
    std::vector<double> rec(ns * kSegStride);
    if (hipMemcpy(rec.data(), h->dv.segb[b], rec.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return VC_ERR_NO_DEVICE;
    for (size_t s = 0; s < ns; ++s)
      for (int e = 0; e < kSegLen; ++e) {
        int a = 0, c = 0;
        seg_entry(e, &a, &c);
        const double val = rec[s * kSegStride + e];
        if (c == 33) { if (g) g[s * 33 + a] = val; }
        else if (H) { H[s * 1089 + a * 33 + c] = val; H[s * 1089 + c * 33 + a] = val; }
      }
  }
  if (cost && hipMemcpy(cost, h->dv.seg_costb[b], ns * 8, hipMemcpyDeviceToHost) != hipSuccess) return VC_ERR_NO_DEVICE;
  return VC_OK;
}
int vc_get_imu_weights(vc_calibrator* h, double* out) {
  NOT_RUNNING(h);
  BIND_DEVICE(h);
  if (!out || !h->dv.imu_on) return VC_ERR_BAD_ARG;
  const size_t n = (size_t)std::max(0, h->dv.n_frames - 1) * 81;
  if (n == 0) return VC_OK;
  if (hipMemcpyAsync(out, h->dv.wsqrtb[h->wcur], n * 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return VC_ERR_NO_DEVICE;
  return VC_OK;
}
// GetSolutionCovariance, vicalibrator.h:802-857.  The blocks are the ones SetupProblem files in covariance_params_
// (:561, :567, :594): per camera q_ck (4), p_ck (3) and, unless the intrinsics are fixed, the model parameters.
// The covariance of the shared parameters with the frames (and, with the IMU, velocities and the other IMU
// parameters) marginalised is the inverse of the undamped reduced system S that every LM pass assembles on the device;
// the D x D inverse is host code and runs once.  As Ceres does, the tangent-space covariance of q_ck is lifted with
// the local parameterisation's Jacobian (local-param-se3.h:121-157) and constant blocks get zeros.
static int covariance_layout(vc_calibrator* h, std::vector<int>* first, std::vector<int>* size) {
  int n = 0;
  for (const HostCam& cm : h->cams) {
    first->push_back(n); size->push_back(4); n += 4;
    first->push_back(n); size->push_back(3); n += 3;
    if (!h->fix_intrinsics) { first->push_back(n); size->push_back(cm.nk); n += cm.nk; }
  }
  return n;
}
int vc_solution_covariance_dim(vc_calibrator* h) {
  if (!h) return VC_ERR_BAD_ARG;
  std::vector<int> first, size;
  return covariance_layout(h, &first, &size);
}
int vc_get_solution_covariance_names(vc_calibrator* h, char* buf, int len) {
  if (!h || !buf || len <= 0) return VC_ERR_BAD_ARG;
  std::string out;
  for (size_t c = 0; c < h->cams.size(); ++c) {       // the strings of vicalibrator.h:563, :569, :596-598
    out += "c[" + std::to_string(c) + "].q_ck:(4) c[" + std::to_string(c) + "].p_ck:(3) ";
    if (!h->fix_intrinsics) out += "c[" + std::to_string(c) + "].params:(" + std::to_string(h->cams[c].nk) + ") ";
  }
  if ((int)out.size() + 1 > len) return VC_ERR_BAD_ARG;
  std::memcpy(buf, out.c_str(), out.size() + 1);
  return VC_OK;
}
int vc_get_solution_covariance(vc_calibrator* h, double* cov, int max_n, int* n_out) {
  NOT_RUNNING(h);
  BIND_DEVICE(h);
  if (!cov) return VC_ERR_BAD_ARG;
  std::vector<int> first, size;
  const int n = covariance_layout(h, &first, &size);
  if (n_out) *n_out = n;
  if (n > max_n) return VC_ERR_BAD_ARG;
  if (h->device_dirty) { int rc = vc_prepare(h); if (rc) return rc; }
  { int rc = h->linearize_hold(1e300, nullptr); if (rc) return rc; }
  const int D = h->dv.D;
  std::vector<double> M((size_t)D * D);
  if (D && hipMemcpy(M.data(), h->dv.Sbuf, M.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return VC_ERR_NO_DEVICE;
  // S = L L^T, then S^-1 = L^-T L^-1 (lower triangle of M holds L, then L^-1)
  for (int j = 0; j < D; ++j) {
    double d = M[(size_t)j * D + j];
    for (int k = 0; k < j; ++k) d -= M[(size_t)j * D + k] * M[(size_t)j * D + k];
    if (!(d > 0.0)) return VC_ERR_NUMERIC;            // rank deficient: Ceres reports "Failed to compute covariance" (:853)
    d = std::sqrt(d); M[(size_t)j * D + j] = d;
    for (int i = j + 1; i < D; ++i) {
      double t = M[(size_t)i * D + j];
      for (int k = 0; k < j; ++k) t -= M[(size_t)i * D + k] * M[(size_t)j * D + k];
      M[(size_t)i * D + j] = t / d;
    }
  }
  std::vector<double> Li((size_t)D * D, 0.0), Ct((size_t)D * D, 0.0);
  for (int j = 0; j < D; ++j) {
    Li[(size_t)j * D + j] = 1.0 / M[(size_t)j * D + j];
    for (int i = j + 1; i < D; ++i) {
      double t = 0.0;
      for (int k = j; k < i; ++k) t -= M[(size_t)i * D + k] * Li[(size_t)k * D + j];
      Li[(size_t)i * D + j] = t / M[(size_t)i * D + i];
    }
  }
  for (int i = 0; i < D; ++i) for (int j = 0; j <= i; ++j) {
    double t = 0.0;
    for (int k = i; k < D; ++k) t += Li[(size_t)k * D + i] * Li[(size_t)k * D + j];
    Ct[(size_t)i * D + j] = t; Ct[(size_t)j * D + i] = t;
  }
  // lift: ambient row r of block b = sum_a P_b[r][a] * tangent column (col_b + a);  P = local Jacobian (q_ck) or identity
  std::vector<double> P((size_t)n * D, 0.0);
  {
    std::lock_guard<std::mutex> lk(h->result_mutex);
    int b = 0;
    for (size_t c = 0; c < h->cams.size(); ++c) {
      const HostCam& cm = h->cams[c];
      const int fl = h->cam_flags[c];
      int col = h->cam_col0[c];
      const double* q = cm.T_ck;       // [x y z w]: d(q * exp(w))/dw at 0 = 1/2 [ w I + [v]x ; -v^T ]
      if (fl & kCamRotFree) {
        const double J[12] = {q[3], -q[2], q[1], q[2], q[3], -q[0], -q[1], q[0], q[3], -q[0], -q[1], -q[2]};
        for (int r = 0; r < 4; ++r) for (int a = 0; a < 3; ++a) P[(size_t)(first[b] + r) * D + col + a] = 0.5 * J[3 * r + a];
        col += 3;
      }
      ++b;
      if (fl & kCamTransFree) { for (int r = 0; r < 3; ++r) P[(size_t)(first[b] + r) * D + col + r] = 1.0; col += 3; }
      ++b;
      if (!h->fix_intrinsics) { for (int r = 0; r < cm.nk; ++r) P[(size_t)(first[b] + r) * D + col + r] = 1.0; ++b; }
    }
  }
  std::vector<double> PC((size_t)n * D, 0.0);
  for (int i = 0; i < n; ++i) for (int k = 0; k < D; ++k) {
    const double p = P[(size_t)i * D + k];
    if (p != 0.0) for (int j = 0; j < D; ++j) PC[(size_t)i * D + j] += p * Ct[(size_t)k * D + j];
  }
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
    double t = 0.0;
    for (int k = 0; k < D; ++k) t += PC[(size_t)i * D + k] * P[(size_t)j * D + k];
    cov[(size_t)i * n + j] = t;
  }
  return VC_OK;
}
int vc_get_debug_stamps(vc_calibrator* h, long long* out) {
  if (!h || !out) return VC_ERR_BAD_ARG;
  return hipMemcpy(out, h->dv.dbg, 32 * 8, hipMemcpyDeviceToHost) == hipSuccess ? VC_OK : VC_ERR_NO_DEVICE;
}
long long vc_num_observations(vc_calibrator* h) { return h ? h->dv.n_obs : 0; }
int vc_num_tiles(vc_calibrator* h) { return h ? h->dv.n_tiles : 0; }

}  // extern "C"
