// vicalib_amd.hpp -- C++ face of libvicalib_amd.so: the public surface of
// visual_inertial_calibration::ViCalibrator (reference include/vicalib/vicalibrator.h:119-544) with the same
// member names, argument order and meaning, over the C ABI of vicalib_amd.h.  Header-only, no dependencies
// (the reference's Sophus/Calibu/Eigen types are replaced by plain aggregates with the same memory layout:
// Se3 = Sophus::SE3d::data() = [qx qy qz qw tx ty tz]).  INTEGRATION.md shows the variant that keeps the
// Calibu / Sophus types for a build inside the vicalib tree.
#pragma once
#include <vicalib_amd.h>

#include <algorithm>
#include <array>
#include <stdexcept>
#include <cstring>
#include <string>
#include <vector>

namespace visual_inertial_calibration {

struct Se3 {
  std::array<double, 7> v{{0, 0, 0, 1, 0, 0, 0}};
  double* data() { return v.data(); }
  const double* data() const { return v.data(); }
};

// CameraAndPose (vicalibrator.h:65-73): the calibu camera becomes (model id, parameter vector, image size)
struct CameraAndPose {
  int model = VC_MODEL_POLY3;
  std::vector<double> params;
  int width = 0, height = 0;
  Se3 T_ck;
};
struct VicalibFrame {          // vicalibrator.h:76-97
  Se3 t_wp_;
  std::array<double, 3> v_w_{{0, 0, 0}};
  double time = 0;
};

// The reference CHECK-aborts on misuse (bad index, setter while running, :333-:391); the wrapper throws instead of
// dropping the status code.
inline int vc_checked(int rc, const char* what) {
  if (rc < 0) throw std::runtime_error(std::string("vicalib_amd: ") + what + " failed (status " + std::to_string(rc) + ")");
  return rc;
}

class ViCalibrator {
 public:
  explicit ViCalibrator(int device = 0) {
    const int rc = vc_create(&h_, device);
    if (rc != VC_OK) throw std::runtime_error(rc == VC_ERR_NO_DEVICE ? "vicalib_amd: no HIP device (there is no CPU fallback)" : "vicalib_amd: vc_create failed");
  }
  ~ViCalibrator() { vc_destroy(h_); }
  ViCalibrator(const ViCalibrator&) = delete;
  ViCalibrator& operator=(const ViCalibrator&) = delete;

  void Clear() { vc_checked(vc_clear(h_), "Clear"); }                                                         // :232
  int AddCamera(const CameraAndPose& c) {                                               // :332
    return vc_checked(vc_add_camera(h_, c.model, c.params.data(), (int)c.params.size(), c.width, c.height, c.T_ck.data()), "AddCamera");
  }
  void FixCameraIntrinsics(bool should_fix = true) { vc_checked(vc_fix_camera_intrinsics(h_, should_fix), "FixCameraIntrinsics"); }   // :346
  int AddFrame(const Se3& t_wk, double time) { return vc_checked(vc_add_frame(h_, t_wk.data(), time), "AddFrame"); }        // :355
  void SetFramePose(int frame, const Se3& t_wk) { vc_checked(vc_set_frame_pose(h_, frame, t_wk.data()), "SetFramePose"); }      // GetFrame(id)->t_wp_ = ...
  // AddObservation(frame, cam, p_w, p_c, time) :385 -- and its bulk form
  void AddObservation(size_t frame, size_t cam, const double p_w[3], const double p_c[2], double /*time*/) {
    vc_checked(vc_add_observations(h_, (int)frame, (int)cam, 1, p_w, p_c), "AddObservation");
  }
  void AddObservations(size_t frame, size_t cam, int n, const double* p_w, const double* p_c) {
    vc_checked(vc_add_observations(h_, (int)frame, (int)cam, n, p_w, p_c), "AddObservations");
  }
  bool AddImuMeasurements(const double gyro[3], const double accel[3], double time) {    // :370
    return vc_add_imu(h_, 1, gyro, accel, &time) == VC_OK;
  }
  int AddImuMeasurements(int n, const double* gyro, const double* accel, const double* time) { return vc_add_imu(h_, n, gyro, accel, time); }
  // robust branch of calibu::PosePnPRansac (iterations = 0: the reference's own call, vicalib-task.cc:323-325)
  void SetPnPRansac(int iterations, double tol_px) { vc_checked(vc_set_pnp_ransac(h_, iterations, tol_px), "SetPnPRansac"); }
  int InitFramePosesPnP() { int n = 0; vc_checked(vc_init_frame_poses_pnp(h_, &n), "InitFramePosesPnP"); return n; }      // vicalib-task.cc:335-348

  void SetOptimizationFlags(bool bias_active, bool inertial_active, bool rotation_only, bool optimize_imu_time_offset) {   // :252
    vc_checked(vc_set_optimization_flags(h_, bias_active, inertial_active, rotation_only, optimize_imu_time_offset), "SetOptimizationFlags");
  }
  void SetFunctionTolerance(double t) { vc_checked(vc_set_function_tolerance(h_, t), "SetFunctionTolerance"); }              // :277
  void SetSigmas(double gyro_sigma, double accel_sigma) { vc_checked(vc_set_sigmas(h_, gyro_sigma, accel_sigma), "SetSigmas"); }   // :290
  void SetTimeOffset(double t) { vc_checked(vc_set_time_offset(h_, t), "SetTimeOffset"); }                            // :296
  void SetBiases(const double b[6]) { vc_checked(vc_set_biases(h_, b), "SetBiases"); }                            // :301
  void SetScaleFactor(const double s[6]) { vc_checked(vc_set_scale_factor(h_, s), "SetScaleFactor"); }                 // :308
  // gflags the reference reads inside the class
  void SetMaxIters(int n) { vc_checked(vc_set_max_iters(h_, n), "SetMaxIters"); }
  void SetCalibrateImu(bool b) { vc_checked(vc_set_calibrate_imu(h_, b), "SetCalibrateImu"); }
  void SetRemoveOutliers(bool b, double threshold) { vc_checked(vc_set_remove_outliers(h_, b, threshold), "SetRemoveOutliers"); }

  void Start() { vc_checked(vc_start(h_), "Start"); }                                                         // :263
  bool IsRunning() { return vc_is_running(h_) > 0; }                                     // :314
  void Stop() { vc_checked(vc_stop(h_), "Stop"); }                                                           // :317
  int Solve() { return vc_solve(h_); }                                                   // Start() + join

  size_t NumFrames() { return (size_t)vc_num_frames(h_); }                               // :471
  size_t NumCameras() { return (size_t)vc_num_cameras(h_); }                             // :484
  double time_offset() { return vc_time_offset(h_); }                                    // :474
  double MeanSquaredError() { return vc_mean_squared_error(h_); }                        // :506
  unsigned GetNumIterations() { return vc_get_num_iterations(h_); }                      // :283
  std::vector<double> GetCameraProjRMSE() { std::vector<double> r(NumCameras()); vc_get_camera_proj_rmse(h_, r.data()); return r; }   // :160
  std::array<double, 6> GetBiases() { std::array<double, 6> b; vc_get_biases(h_, b.data()); return b; }              // :286
  std::array<double, 6> GetScaleFactor() { std::array<double, 6> s; vc_get_scale_factor(h_, s.data()); return s; }   // :306
  std::array<double, 2> GetGravity() { std::array<double, 2> g; vc_get_gravity(h_, g.data()); return g; }
  CameraAndPose GetCamera(size_t id) {                                                   // :492
    CameraAndPose c;
    c.params.resize(16);
    int n = 0;
    if (vc_get_camera(h_, (int)id, c.params.data(), &n, c.T_ck.data()) != VC_OK) throw std::out_of_range("GetCamera");
    c.params.resize(n);
    return c;
  }
  VicalibFrame GetFrame(size_t id) {                                                     // :477
    VicalibFrame f;
    if (vc_get_frame(h_, (int)id, f.t_wp_.data(), f.v_w_.data(), &f.time) != VC_OK) throw std::out_of_range("GetFrame");
    return f;
  }
  // imu_buffer() :487 (a copy: [gyro(3) accel(3) time] per measurement), GetIntegrationPoses(id) :508 (rows of 11: q t v time),
  // PrintResults() :536 (returned instead of logged)
  std::vector<std::array<double, 7>> imu_buffer() {
    const int n = vc_num_imu_measurements(h_);
    std::vector<double> g(3 * (size_t)std::max(n, 0)), a(g.size()), t((size_t)std::max(n, 0));
    std::vector<std::array<double, 7>> out((size_t)std::max(n, 0));
    if (n > 0) vc_checked(vc_get_imu_measurements(h_, g.data(), a.data(), t.data(), n), "imu_buffer");
    for (int i = 0; i < n; ++i) out[i] = {{g[3 * i], g[3 * i + 1], g[3 * i + 2], a[3 * i], a[3 * i + 1], a[3 * i + 2], t[i]}};
    return out;
  }
  std::vector<std::array<double, 11>> GetIntegrationPoses(unsigned id) {
    const int n = vc_checked(vc_get_integration_poses(h_, (int)id, nullptr, 0), "GetIntegrationPoses");      // the count first: all of them
    std::vector<std::array<double, 11>> out((size_t)std::max(n, 0));
    if (n > 0) vc_checked(vc_get_integration_poses(h_, (int)id, out[0].data(), n), "GetIntegrationPoses");
    return out;
  }
  std::string PrintResults() {
    // the length first (any number of cameras); a running Start() worker may lengthen the text between the two calls: slack + retry
    for (int attempt = 0; attempt < 8; ++attempt) {
      const int n = vc_checked(vc_print_results(h_, nullptr, 0), "PrintResults");
      std::string s((size_t)n + 65, '\0');
      if (vc_print_results(h_, &s[0], n + 65) >= 0) { s.resize(std::strlen(s.c_str())); return s; }
    }
    throw std::runtime_error("PrintResults: the text kept growing");
  }
  void WriteCameraModels(const std::string& filename) { vc_checked(vc_write_camera_models(h_, filename.c_str()), "WriteCameraModels"); }   // :208
  // GetSolutionCovariance(problem) :802-857: row-major n x n over the blocks named by covariance_names
  std::vector<double> GetSolutionCovariance(int* n_out = nullptr) {
    const int n = vc_solution_covariance_dim(h_);
    std::vector<double> cov(n > 0 ? (size_t)n * n : 0);
    int m = 0;
    if (n <= 0 || vc_get_solution_covariance(h_, cov.data(), n, &m) != VC_OK) cov.clear();
    if (n_out) *n_out = cov.empty() ? 0 : n;
    return cov;
  }
  std::string covariance_names() {
    std::string s(64 * 8 + 64, '\0');
    if (vc_get_solution_covariance_names(h_, &s[0], (int)s.size()) != VC_OK) return std::string();
    s.resize(s.find('\0'));
    return s;
  }
  vc_calibrator* handle() { return h_; }

 private:
  vc_calibrator* h_ = nullptr;
};

}  // namespace visual_inertial_calibration
