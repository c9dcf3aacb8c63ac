// vicalib -- command-line calibration tool on libvicalib_amd.so, keeping the reference tool's command line
// (flags of src/vicalib-engine.cc:30-104 and src/vicalib-task.cc:19-51) and its outputs (cameras.xml, poses.csv,
// poses.txt).  What the reference does with HAL + Calibu image processing (grab images, find conics, match the
// grid) is outside the solver path; here the sensors are files:
//
//   -cam  detections://cam0.csv[,cam1.csv,...]   one file per camera channel, lines  frame,dot_id,u,v,X,Y,Z[,time]
//                                                (the format the reference prints with -output_conics,
//                                                 vicalib-task.cc:313-317; `time` is an optional 8th column)
//   -cam  file://dir/cam0_*.pgm[,dir/cam1_*.pgm]  (round 4) one glob of 8-bit PGM images per camera channel, sorted by name, one frame per
//                                                image -- HAL's FileReader URI of the reference's own usage line (main.cc:11).  Every
//                                                image goes through the GPU dot detector (vc_detector_find_conics) and the grid
//                                                association (vc_target_find): what VicalibTask::AddImageMeasurements does with
//                                                calibu::ImageProcessing / ConicFinder / TargetGridDot (vicalib-task.cc:263-330).
//                                                The target's large / small pattern comes from -grid_pattern_file (rows of 0 / 1), or
//                                                from -grid_height / -grid_width / -grid_seed through the library's own generator;
//                                                Calibu's presets and MakePattern are not in the reference tree.
//   -imu  csv://dir                              HAL CsvDriver layout: dir/accel.txt, dir/gyro.txt, dir/timestamp.txt
//
// Everything from "AddFrame" on is the reference's flow: start intrinsics per -models (vicalib-engine.cc:203-257) or
// -model_files, PnP seed pose per frame (vicalib-task.cc:335-348), Start(has_initial_guess) (vicalib-task.cc:226-234),
// 30 ms polling loop (vicalib-engine.cc:376-431), WriteCalibration (:353-372), success test (vicalib-task.cc:831-856).
#include <vicalib_amd.hpp>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <glob.h>
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

namespace vic = visual_inertial_calibration;

// ------------------------------------------------------------------------------------------- flags
struct Flag { std::string type, value, help; };
static std::map<std::string, Flag> g_flags;
static void Define(const char* name, const char* type, const char* def, const char* help) { g_flags[name] = Flag{type, def, help}; }
static bool FlagBool(const char* n) { const std::string& v = g_flags.at(n).value; return v == "true" || v == "1" || v == "yes"; }
static double FlagDouble(const char* n) { return std::atof(g_flags.at(n).value.c_str()); }
static long FlagInt(const char* n) { return std::atol(g_flags.at(n).value.c_str()); }
static const std::string& FlagString(const char* n) { return g_flags.at(n).value; }

static void DefineFlags() {
  // vicalib-engine.cc:30-104
  Define("calibrate_imu", "bool", "true", "Calibrate the IMU in addition to the camera.");
  Define("calibrate_intrinsics", "bool", "true", "Calibrate the camera intrinsics as well as the extrinsics.");
  Define("save_poses", "bool", "false", "Save calibrated camera poses when done (poses.csv).");
  Define("exit_vicalib_on_finish", "bool", "true", "Exit when the optimisation finishes.");
  Define("frame_skip", "int32", "0", "Number of frames to skip between constraints.");
  Define("grid_height", "int32", "10", "Height of grid in circles.");
  Define("grid_width", "int32", "19", "Width of grid in circles.");
  Define("grid_spacing", "double", "0.01355", "Distance between circles on grid (m).");
  Define("grid_seed", "int32", "71", "Seed used to generate the grid.");
  Define("has_initial_guess", "bool", "false", "Whether or not the given calibration file has a valid guess.");
  Define("output_conics", "bool", "false", "Echo the detections that were used (frame,dot_id,u,v,X,Y,Z).");
  Define("grid_preset", "string", "", "Which grid preset to use: small, large, letter, medium.");
  Define("grid_pattern_file", "string", "", "(image input) large / small pattern of the target: grid_height rows of grid_width 0 / 1 entries (1 = large dot).");
  Define("max_reprojection_error", "double", "0.15", "Maximum allowed reprojection error (pixels).");
  Define("num_vicalib_frames", "int64", "-1", "Number of frames to process before calibration begins (-1: all).");
  Define("print_poses", "bool", "false", "Output poses to poses.txt");
  Define("print_covariance", "bool", "false", "Print the solution covariance of q_ck / p_ck / params (the reference compiles this in with COMPUTE_VICALIB_COVARIANCE).");
  Define("output", "string", "cameras.xml", "Output XML file to write camera models to.");
  Define("output_log_file", "string", "vicalibrator.log", "Calibration result output log file.");
  Define("cam", "string", "", "Camera URI: detections://cam0.csv[,cam1.csv...] or file://dir/cam0_*.pgm[,dir/cam1_*.pgm...] (8-bit PGM images)");
  Define("imu", "string", "", "IMU URI (if available): csv://directory");
  Define("models", "string", "", "Comma-separated list of camera model types: fov, poly2, poly3, rational6, kb4, linear.");
  Define("model_files", "string", "", "Comma-separated list of camera model files to initialise from.");
  Define("max_iters", "int32", "200", "Max iterations.");
  Define("pnp_ransac_its", "int32", "0", "Minimal-sample iterations of the robust pose seed (0: plain PnP, as the reference calls PosePnPRansac).");
  Define("pnp_ransac_tol", "double", "2.0", "Inlier threshold of the robust pose seed, pixels.");
  Define("gyro_sigma", "double", "5.3088444e-5", "Sigma of gyroscope measurements.");
  Define("accel_sigma", "double", "0.001883649", "Sigma of accel measurements.");
  Define("remove_outliers", "bool", "false", "Remove outliers and re-optimise.");
  Define("outlier_threshold", "double", "2.0", "Outlier threshold (x camera RMSE).");
  Define("paused", "bool", "false", "(GUI) ignored.");
  Define("use_only_when_static", "bool", "false", "(sensor front-end) ignored: the detections file already holds the chosen frames.");
  // flags of the sensor / pattern front-end (vicalib-engine.cc:39, :65-77, :88-92, vicalib-task.cc:19): accepted so that an existing
  // command line keeps working, without effect here -- the detections file is what that front-end would have produced
  Define("device_serial", "string", "-1", "(sensor front-end) ignored: serial number of device.");
  Define("scaled_ir_depth_cal", "bool", "false", "(sensor front-end) ignored: produce ir and depth calibration by rescaling RGB.");
  Define("static_accel_threshold", "double", "0.08", "(sensor front-end) ignored: acceleration below which the device counts as static.");
  Define("static_gyro_threshold", "double", "0.04", "(sensor front-end) ignored: angular velocity below which the device counts as static.");
  Define("static_threshold_preset", "int32", "0", "(sensor front-end) ignored: a visual_inertial_calibration::StaticThresholdPreset.");
  Define("use_static_threshold_preset", "bool", "false", "(sensor front-end) ignored: use one of the predefined static thresholds.");
  Define("output_pattern_file", "string", "", "(pattern front-end) ignored: EPS or SVG file to save the calibration pattern.");
  Define("grid_large_rad", "double", "0.00423", "(pattern front-end) ignored: radius of large dots (m).");
  Define("grid_small_rad", "double", "0.00283", "(pattern front-end) ignored: radius of small dots (m).");
  Define("clip_good", "bool", "false", "(sensor front-end) ignored: output proto file of only good tracked images.");
  // vicalib-task.cc:19-51
  Define("find_time_offset", "bool", "true", "Optimize for the time offset between the IMU and images.");
  Define("function_tolerance", "double", "1e-6", "Convergence criterion for the optimizer.");
  Define("max_fx_diff", "double", "10.0", "Maximum fx difference between calibrations.");
  Define("max_fy_diff", "double", "10.0", "Maximum fy difference between calibrations.");
  Define("max_cx_diff", "double", "10.0", "Maximum cx difference between calibrations.");
  Define("max_cy_diff", "double", "10.0", "Maximum cy difference between calibrations.");
  Define("max_fov_w_diff", "double", "0.3", "Maximum fov distortion difference between calibrations.");
  Define("max_poly3_diff_k1", "double", "0.1", "Maximum poly3 k1 difference between calibrations.");
  Define("max_poly3_diff_k2", "double", "0.1", "Maximum poly3 k2 difference between calibrations.");
  Define("max_poly3_diff_k3", "double", "0.1", "Maximum poly3 k3 difference between calibrations.");
  Define("max_camera_trans_diff", "double", "0.1", "Maximum camera translation difference between calibrations.");
  Define("max_camera_angle_diff", "double", "0.1", "Maximum camera angle difference (rad) between calibrations.");
  Define("max_imu_gyro_diff", "double", "0.1", "Maximum gyroscope bias difference between calibrations.");
  Define("max_imu_accel_diff", "double", "0.1", "Maximum accelrometer bias difference between calibrations.");
  Define("imu_diff_sense", "string", "reference", "IMUCalibrationDiffer with -has_initial_guess: 'reference' = the comparisons as the reference writes them "
         "(vicalib-task.cc:811-826: a bias difference BELOW the limit counts as differing), 'corrected' = above the limit.");
  Define("use_system_time", "bool", "true", "Use the first (system) column of timestamp.txt; otherwise the second (device).");
  // new: what HAL would have told the reference
  Define("image_width", "int32", "640", "Image width of every channel (HAL reports it in the reference).");
  Define("image_height", "int32", "480", "Image height of every channel.");
  Define("frame_rate", "double", "30", "Frame rate used for timestamps when the detections carry no time column.");
  Define("device", "int32", "0", "HIP device ordinal (first device with -gpus N).");
  Define("gpus", "int32", "1", "Number of GPUs: frames are sharded, one calibrator per device, RCCL all-reduce per iteration.");
}

static int Usage(int code) {
  std::printf("vicalib (MI355X solver) -- flags (gflags syntax: -flag value, --flag=value, -noflag)\n");
  for (const auto& kv : g_flags) std::printf("  -%-24s (%s) default: %-12s %s\n", kv.first.c_str(), kv.second.type.c_str(), ("\"" + kv.second.value + "\"").c_str(), kv.second.help.c_str());
  return code;
}

static bool ParseFlags(int argc, char** argv, std::string* err) {
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a == "-h" || a == "-help" || a == "--help") { *err = "help"; return false; }
    if (a.size() < 2 || a[0] != '-') { *err = "unexpected argument '" + a + "'"; return false; }
    a = a.substr(a[1] == '-' ? 2 : 1);
    std::string name = a, value; bool has_value = false;
    const size_t eq = a.find('=');
    if (eq != std::string::npos) { name = a.substr(0, eq); value = a.substr(eq + 1); has_value = true; }
    auto it = g_flags.find(name);
    if (it == g_flags.end() && name.compare(0, 2, "no") == 0) {
      auto it2 = g_flags.find(name.substr(2));
      if (it2 != g_flags.end() && it2->second.type == "bool" && !has_value) { it2->second.value = "false"; continue; }
    }
    if (it == g_flags.end()) { *err = "unknown command line flag '" + name + "'"; return false; }
    if (it->second.type == "bool") {
      if (!has_value) { it->second.value = "true"; continue; }
      it->second.value = (value == "true" || value == "1" || value == "yes" || value == "t" || value == "y") ? "true" : "false";
      continue;
    }
    if (!has_value) {
      if (i + 1 >= argc) { *err = "flag '" + name + "' is missing its argument"; return false; }
      value = argv[++i];
    }
    if (it->second.type != "string") {
      char* end = nullptr; std::strtod(value.c_str(), &end);
      if (end == value.c_str() || *end != 0) { *err = "illegal value '" + value + "' specified for " + it->second.type + " flag '" + name + "'"; return false; }
    }
    it->second.value = value;
  }
  return true;
}

// ------------------------------------------------------------------------------------------- inputs
static std::vector<std::string> Split(const std::string& s, char sep) {
  std::vector<std::string> out; std::stringstream ss(s); std::string item;
  while (std::getline(ss, item, sep)) if (!item.empty()) out.push_back(item);
  return out;
}
static std::string StripScheme(const std::string& uri) {
  const size_t p = uri.find("//");
  return p == std::string::npos ? uri : uri.substr(p + 2);
}
static bool ParseNumbers(const std::string& line, std::vector<double>* v) {
  v->clear();
  const char* p = line.c_str();
  while (*p) {
    while (*p == ' ' || *p == '\t' || *p == ',' || *p == ';' || *p == '\r') ++p;
    if (!*p) break;
    char* end = nullptr;
    const double d = std::strtod(p, &end);
    if (end == p) return false;
    v->push_back(d); p = end;
  }
  return !v->empty();
}

struct Detection { long frame; int dot; double u, v, X, Y, Z; };
struct Channel { std::vector<Detection> det; std::map<long, double> frame_time; };

// ---- image channels (vicalib-task.cc:263-330 with files for sensors) ----------------------------------------------------------------
static bool ReadPgm(const std::string& path, int* w, int* h, std::vector<unsigned char>* px, std::string* err) {
  std::ifstream f(path, std::ios::binary);
  if (!f) { *err = "cannot open image " + path; return false; }
  std::string magic; f >> magic;
  if (magic != "P5") { *err = path + ": not a binary PGM (P5)"; return false; }
  auto next_int = [&](long* v) {
    while (true) { f >> std::ws; if (f.peek() == '#') { std::string c; std::getline(f, c); } else break; }
    return (bool)(f >> *v);
  };
  long W = 0, H = 0, M = 0;
  if (!next_int(&W) || !next_int(&H) || !next_int(&M) || W < 1 || H < 1 || M != 255) { *err = path + ": expected an 8-bit PGM header"; return false; }
  f.get();                                          // the single whitespace byte behind the header
  px->resize((size_t)W * H);
  f.read((char*)px->data(), (std::streamsize)px->size());
  if ((size_t)f.gcount() != px->size()) { *err = path + ": truncated image data"; return false; }
  *w = (int)W; *h = (int)H;
  return true;
}
struct TargetSpec { std::vector<int> pattern; int rows = 0, cols = 0; double spacing = 0.0; };
// one camera channel from a glob of images: frame k = k-th file (sorted); detections carry the target dot's index and position
static bool ReadImages(const std::string& pattern_glob, const TargetSpec& tg, int device, Channel* ch, int* width, int* height, std::string* err) {
  glob_t g; std::memset(&g, 0, sizeof(g));
  if (glob(pattern_glob.c_str(), 0, nullptr, &g) != 0 || g.gl_pathc == 0) { globfree(&g); *err = "no images match " + pattern_glob; return false; }
  std::vector<std::string> files(g.gl_pathv, g.gl_pathv + g.gl_pathc);
  globfree(&g);
  std::sort(files.begin(), files.end());
  vc_detector* det = nullptr;
  std::vector<unsigned char> px;
  const int kMax = 4096;
  std::vector<double> cen(2 * kMax), con(9 * kMax);
  std::vector<int> box(4 * kMax), idx(kMax);
  long placed = 0, skipped = 0;
  for (size_t k = 0; k < files.size(); ++k) {
    int w = 0, h = 0;
    if (!ReadPgm(files[k], &w, &h, &px, err)) { if (det) vc_detector_destroy(det); return false; }
    if (!det) {
      const int rc = vc_detector_create(device, w, h, &det);
      if (rc != VC_OK) { *err = rc == VC_ERR_NO_DEVICE ? "no usable HIP device for the dot detector" : "vc_detector_create failed"; return false; }
      *width = w; *height = h;
    } else if (w != *width || h != *height) { *err = files[k] + ": image size differs from the first image of the channel"; vc_detector_destroy(det); return false; }
    int n = 0, m = 0;
    if (vc_detector_find_conics(det, px.data(), w, cen.data(), con.data(), box.data(), kMax, &n) != VC_OK ||
        vc_target_find(cen.data(), con.data(), n, tg.pattern.data(), tg.rows, tg.cols, idx.data(), &m) != VC_OK) { *err = files[k] + ": detection failed"; vc_detector_destroy(det); return false; }
    if (m == 0) { ++skipped; continue; }                 // "Tracking bad" (vicalib-task.cc:278-281): the frame contributes nothing
    for (int i = 0; i < n; ++i) {
      if (idx[i] < 0) continue;
      const int r = idx[i] / tg.cols, c = idx[i] % tg.cols;
      ch->det.push_back(Detection{(long)k, idx[i], cen[2 * i], cen[2 * i + 1], c * tg.spacing, r * tg.spacing, 0.0});
      ++placed;
    }
  }
  if (det) vc_detector_destroy(det);
  std::fprintf(stderr, "I %s: %zu images, %ld dots associated with the target, %ld images without an unambiguous placement\n", pattern_glob.c_str(), files.size(), placed, skipped);
  return true;
}
static bool ReadDetections(const std::string& path, Channel* ch, std::string* err) {
  std::ifstream f(path);
  if (!f) { *err = "cannot open detections file " + path; return false; }
  std::string line; std::vector<double> v; long ln = 0;
  while (std::getline(f, line)) {
    ++ln;
    if (line.empty() || line[0] == '#' || line[0] == '%') continue;
    if (!ParseNumbers(line, &v)) continue;                    // header / log text between the detection lines
    if (v.size() < 7) { *err = path + ":" + std::to_string(ln) + ": expected frame,dot_id,u,v,X,Y,Z[,time]"; return false; }
    if (v[1] < 0) continue;                                   // unmatched conic (vicalib-task.cc:309-311)
    ch->det.push_back(Detection{(long)v[0], (int)v[1], v[2], v[3], v[4], v[5], v[6]});
    if (v.size() >= 8) ch->frame_time[(long)v[0]] = v[7];
  }
  return true;
}

struct ImuData { std::vector<double> gyro, accel, time; };
static bool ReadColumns(const std::string& path, int min_cols, std::vector<std::vector<double>>* rows, std::string* err) {
  std::ifstream f(path);
  if (!f) { *err = "cannot open " + path; return false; }
  std::string line; std::vector<double> v;
  while (std::getline(f, line)) {
    if (line.empty() || line[0] == '#' || line[0] == '%') continue;
    if (!ParseNumbers(line, &v)) continue;
    if ((int)v.size() < min_cols) { *err = path + ": expected at least " + std::to_string(min_cols) + " columns"; return false; }
    rows->push_back(v);
  }
  return true;
}
static bool ReadImu(const std::string& dir, bool system_time, ImuData* imu, std::string* err) {
  std::vector<std::vector<double>> a, g, t;
  if (!ReadColumns(dir + "/accel.txt", 3, &a, err) || !ReadColumns(dir + "/gyro.txt", 3, &g, err) || !ReadColumns(dir + "/timestamp.txt", 1, &t, err)) return false;
  const size_t n = std::min(a.size(), std::min(g.size(), t.size()));
  double last = -1e300;
  for (size_t i = 0; i < n; ++i) {
    const double ts = (!system_time && t[i].size() > 1) ? t[i][1] : t[i][0];
    if (ts <= last) continue;                                 // the calibrator insists on strictly increasing time (vicalibrator.h:373-378)
    last = ts;
    imu->time.push_back(ts);
    for (int k = 0; k < 3; ++k) { imu->gyro.push_back(g[i][k]); imu->accel.push_back(a[i][k]); }
  }
  return true;
}

static int ModelId(const std::string& type) {     // -models strings (vicalib-engine.cc:203-253) and XML type strings (:210-260)
  if (type == "fov" || type == "calibu_fu_fv_u0_v0_w") return VC_MODEL_FOV;
  if (type == "poly2" || type == "calibu_fu_fv_u0_v0_k1_k2") return VC_MODEL_POLY2;
  if (type == "poly3" || type == "poly" || type == "calibu_fu_fv_u0_v0_k1_k2_k3") return VC_MODEL_POLY3;
  if (type == "kb4" || type == "calibu_fu_fv_u0_v0_kb4") return VC_MODEL_KB4;
  if (type == "linear" || type == "calibu_fu_fv_u0_v0") return VC_MODEL_LINEAR;
  if (type == "rational6" || type == "rational" || type == "calibu_fu_fv_u0_v0_rational6") return VC_MODEL_RATIONAL6;      // vicalib-engine.cc:233-240
  return -1;
}
static const char* ModelName(int id) { static const char* n[] = {"fov", "poly2", "poly3", "kb4", "linear", "rational6"}; return (id >= 0 && id < 6) ? n[id] : "?"; }

static std::string Between(const std::string& s, const std::string& a, const std::string& b, size_t from = 0) {
  const size_t p = s.find(a, from); if (p == std::string::npos) return "";
  const size_t q = s.find(b, p + a.size()); if (q == std::string::npos) return "";
  return s.substr(p + a.size(), q - p - a.size());
}
// first <camera_model> of a calibu rig XML (the reference takes rig->cameras_[0], pose ignored: vicalib-engine.cc:190-197)
static bool ReadModelFile(const std::string& path, vic::CameraAndPose* cam, std::string* err) {
  std::ifstream f(path);
  if (!f) { *err = "cannot open model file " + path; return false; }
  std::stringstream ss; ss << f.rdbuf();
  const std::string s = ss.str();
  const std::string head = Between(s, "<camera_model", ">");
  cam->model = ModelId(Between(head, "type=\"", "\""));
  if (cam->model < 0) { *err = path + ": unsupported camera model type '" + Between(head, "type=\"", "\"") + "'"; return false; }
  std::vector<double> v;
  if (ParseNumbers(Between(s, "<width>", "</width>"), &v)) cam->width = (int)v[0];
  if (ParseNumbers(Between(s, "<height>", "</height>"), &v)) cam->height = (int)v[0];
  std::string p = Between(s, "<params>", "</params>");
  std::replace(p.begin(), p.end(), '[', ' '); std::replace(p.begin(), p.end(), ']', ' ');
  if (!ParseNumbers(p, &cam->params)) { *err = path + ": no <params>"; return false; }
  return true;
}

static void RotationMatrix(const double* q, double* R) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}

// IMUCalibrationDiffer (vicalib-task.cc:807-829).  The reference compares with '<': it reports a difference when a bias moved by
// LESS than the limit in any component -- with start biases of zero (vicalib-task.cc:129) that is nearly every calibration, so
// -has_initial_guess makes its IsSuccessful() fail unless all six biases exceed the limits.  The exit status is part of the tool's
// behaviour: 'reference' reproduces it (and says so), 'corrected' compares the way the message reads.
static bool IMUCalibrationDiffer(const double* last, const double* current, bool reference_sense) {
  double diff[6];
  for (int i = 0; i < 6; ++i) diff[i] = last[i] - current[i];
  const double lg = FlagDouble("max_imu_gyro_diff"), la = FlagDouble("max_imu_accel_diff");
  auto out = [&](double d, double lim) { return reference_sense ? std::fabs(d) < lim : std::fabs(d) > lim; };
  if (out(diff[0], lg) || out(diff[1], lg) || out(diff[2], lg)) {
    std::fprintf(stderr, "E IMU bias(es) for gyroscope differ (%g, %g, %g ) more than expected (%g)%s\n", diff[0], diff[1], diff[2], lg,
                 reference_sense ? " [reference comparison sense: '<', see -imu_diff_sense]" : "");
    return true;
  }
  if (out(diff[3], la) || out(diff[4], la) || out(diff[5], la)) {
    std::fprintf(stderr, "E IMU bias(es) for accelrometer differ (%g, %g, %g ) more than expected (%g)%s\n", diff[3], diff[4], diff[5], la,
                 reference_sense ? " [reference comparison sense: '<', see -imu_diff_sense]" : "");
    return true;
  }
  return false;
}

// CameraCalibrationsDiffer (vicalib-task.cc:722-806)
static bool CameraCalibrationsDiffer(const vic::CameraAndPose& last, const vic::CameraAndPose& cur) {
  const char* names[4] = {"fx", "fy", "cx", "cy"};
  const char* lim[4] = {"max_fx_diff", "max_fy_diff", "max_cx_diff", "max_cy_diff"};
  for (int i = 0; i < 4; ++i)
    if (std::fabs(last.params[i] - cur.params[i]) > FlagDouble(lim[i])) { std::fprintf(stderr, "E %s differs too much (%g)\n", names[i], last.params[i] - cur.params[i]); return true; }
  if (cur.model == VC_MODEL_FOV && std::fabs(last.params[4] - cur.params[4]) > FlagDouble("max_fov_w_diff")) { std::fprintf(stderr, "E fov distortion differs too much\n"); return true; }
  if (cur.model == VC_MODEL_POLY3) {
    const char* l3[3] = {"max_poly3_diff_k1", "max_poly3_diff_k2", "max_poly3_diff_k3"};
    for (int i = 0; i < 3; ++i) if (std::fabs(last.params[4 + i] - cur.params[4 + i]) > FlagDouble(l3[i])) { std::fprintf(stderr, "E poly3 distortion differs too much\n"); return true; }
  }
  double d2 = 0;
  for (int i = 4; i < 7; ++i) d2 += (last.T_ck.v[i] - cur.T_ck.v[i]) * (last.T_ck.v[i] - cur.T_ck.v[i]);
  if (std::sqrt(d2) > FlagDouble("max_camera_trans_diff")) { std::fprintf(stderr, "E position of camera differs by %g\n", std::sqrt(d2)); return true; }
  double Ra[9], Rb[9], M[9];
  RotationMatrix(last.T_ck.data(), Ra); RotationMatrix(cur.T_ck.data(), Rb);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { M[3 * i + j] = 0; for (int k = 0; k < 3; ++k) M[3 * i + j] += Ra[3 * k + i] * Rb[3 * k + j]; }
  const double ax = std::atan2(M[7], M[8]), ay = std::atan2(-M[6], std::sqrt(M[7] * M[7] + M[8] * M[8])), az = std::atan2(M[3], M[0]);
  const double lim_a = FlagDouble("max_camera_angle_diff");
  if (std::fabs(ax) > lim_a || std::fabs(ay) > lim_a || std::fabs(az) > lim_a) { std::fprintf(stderr, "E camera orientations are farther apart than expected\n"); return true; }
  return false;
}

// _T2Cart (vicalib-engine.cc:318-351): x y z roll pitch yaw
static void T2Cart(const double* T, double* c) {
  double R[9]; RotationMatrix(T, R);
  c[0] = T[4]; c[1] = T[5]; c[2] = T[6];
  c[3] = std::atan2(R[7], R[8]);
  const double det = -R[6] * R[6] + 1.0;
  c[4] = det <= 0 ? (R[6] > 0 ? -M_PI / 2 : M_PI / 2) : -std::asin(R[6]);
  c[5] = std::atan2(R[3], R[0]);
}

int main(int argc, char** argv) {
  DefineFlags();
  std::string err;
  if (!ParseFlags(argc, argv, &err)) {
    if (err == "help") return Usage(0);
    std::fprintf(stderr, "ERROR: %s\n", err.c_str());
    return 1;
  }
  if (FlagString("cam").empty()) { std::fprintf(stderr, "F No camera URI given\n"); return 1; }      // vicalib-engine.cc:445
  // ---- grid (vicalib-engine.cc:449-464): the detections already carry X,Y,Z; the preset only bounds the dot ids ----
  int grid_w = (int)FlagInt("grid_width"), grid_h = (int)FlagInt("grid_height");
  const std::string preset = FlagString("grid_preset");
  if (!preset.empty()) {
    if (preset == "small" || preset == "0" || preset == "letter") { grid_w = 19; grid_h = 10; }
    else if (preset == "large" || preset == "1") { grid_w = 36; grid_h = 25; }
    else if (preset == "medium") { grid_w = 1 << 15; grid_h = 1; }     // size not recorded in the reference tree: no bound on dot ids
    else { std::fprintf(stderr, "F Unknown grid preset %s\n", preset.c_str()); return 1; }
  }
  // ---- sensors ------------------------------------------------------------------------------------------------------
  const std::vector<std::string> cam_files = Split(StripScheme(FlagString("cam")), ',');
  std::vector<Channel> channels(cam_files.size());
  const bool from_images = FlagString("cam").compare(0, 7, "file://") == 0;      // HAL's FileReader scheme (main.cc:11)
  int img_w = 0, img_h = 0;
  if (from_images) {
    // the target (vicalib-engine.cc:453-465): its pattern from a file, or generated from -grid_height / -grid_width / -grid_seed
    TargetSpec tg;
    tg.rows = grid_h; tg.cols = grid_w; tg.spacing = FlagDouble("grid_spacing");
    if (!FlagString("grid_pattern_file").empty()) {
      std::ifstream pf(FlagString("grid_pattern_file"));
      if (!pf) { std::fprintf(stderr, "F cannot open grid pattern file %s\n", FlagString("grid_pattern_file").c_str()); return 1; }
      std::string line; std::vector<double> v; tg.rows = 0; tg.cols = 0;
      while (std::getline(pf, line)) {
        if (line.empty() || line[0] == '#') continue;
        if (!ParseNumbers(line, &v) || v.empty()) continue;
        if (tg.cols && (int)v.size() != tg.cols) { std::fprintf(stderr, "F grid pattern file: rows of different length\n"); return 1; }
        tg.cols = (int)v.size(); ++tg.rows;
        for (double x : v) tg.pattern.push_back(x != 0.0 ? 1 : 0);
      }
      if (tg.rows < 2 || tg.cols < 2) { std::fprintf(stderr, "F grid pattern file holds no pattern\n"); return 1; }
      grid_w = tg.cols; grid_h = tg.rows;
    } else if (!preset.empty()) {
      std::fprintf(stderr, "F -grid_preset %s with image input: the presets' large / small patterns live in Calibu, which is not part of the reference tree -- "
                           "pass the printed target's pattern with -grid_pattern_file (or generate a target with -grid_height / -grid_width / -grid_seed)\n", preset.c_str());
      return 1;
    } else {
      tg.pattern.resize((size_t)tg.rows * tg.cols);
      vc_target_make_pattern(tg.rows, tg.cols, (unsigned)FlagInt("grid_seed"), tg.pattern.data());
    }
    for (size_t c = 0; c < cam_files.size(); ++c) {
      int w = 0, h = 0;
      if (!ReadImages(cam_files[c], tg, (int)FlagInt("device"), &channels[c], &w, &h, &err)) { std::fprintf(stderr, "F %s\n", err.c_str()); return err.find("HIP device") != std::string::npos ? 3 : 1; }
      img_w = std::max(img_w, w); img_h = std::max(img_h, h);
    }
  } else
  for (size_t c = 0; c < cam_files.size(); ++c)
    if (!ReadDetections(cam_files[c], &channels[c], &err)) { std::fprintf(stderr, "F %s\n", err.c_str()); return 1; }
  const size_t n_cam = channels.size();
  ImuData imu;
  const bool have_imu = !FlagString("imu").empty();
  if (have_imu && !ReadImu(StripScheme(FlagString("imu")), FlagBool("use_system_time"), &imu, &err)) { std::fprintf(stderr, "F %s\n", err.c_str()); return 1; }
  bool calibrate_imu = FlagBool("calibrate_imu");
  if (calibrate_imu && !have_imu) { std::fprintf(stderr, "W -calibrate_imu without -imu: calibrating the cameras only\n"); calibrate_imu = false; }

  // ---- start cameras (vicalib-engine.cc:162-257) -------------------------------------------------------------------
  std::vector<std::string> models = Split(FlagString("models"), ','), model_files = Split(FlagString("model_files"), ',');
  if (model_files.empty() && models.size() < n_cam) {
    std::fprintf(stderr, "I Only %zu models declared; need one for all the %zu channels; assuming poly3\n", models.size(), n_cam);
    models.resize(n_cam, "poly3");
  }
  std::vector<vic::CameraAndPose> input_cameras;
  const int W = img_w > 0 ? img_w : (int)FlagInt("image_width"), H = img_h > 0 ? img_h : (int)FlagInt("image_height");      // (images carry their size)
  if (!model_files.empty()) {
    for (const std::string& mf : model_files) {
      vic::CameraAndPose cam;
      if (!ReadModelFile(mf, &cam, &err)) { std::fprintf(stderr, "F %s\n", err.c_str()); return 1; }
      if (cam.width == 0) { cam.width = W; cam.height = H; }
      input_cameras.push_back(cam);
    }
  } else {
    for (const std::string& type : models) {
      vic::CameraAndPose cam;
      cam.model = ModelId(type);
      if (cam.model < 0) { std::fprintf(stderr, "F camera model '%s' is not supported by this build (fov, poly2, poly3, rational6, kb4, linear)\n", type.c_str()); return 1; }
      cam.width = W; cam.height = H;
      cam.params = {300, 300, W / 2.0, H / 2.0};
      if (cam.model == VC_MODEL_FOV) cam.params.push_back(0.2);
      else cam.params.resize(cam.model == VC_MODEL_POLY2 ? 6 : cam.model == VC_MODEL_POLY3 ? 7 : cam.model == VC_MODEL_KB4 ? 8 : cam.model == VC_MODEL_RATIONAL6 ? 10 : 4, 0.0);
      input_cameras.push_back(cam);
    }
  }
  if (input_cameras.size() < n_cam) { std::fprintf(stderr, "F %zu camera models for %zu channels\n", input_cameras.size(), n_cam); return 1; }
  input_cameras.resize(n_cam);

  // ---- frames: union of the frame ids, -frame_skip, -num_vicalib_frames (vicalib-engine.cc:540-590) ---------------------
  std::set<long> ids;
  for (const Channel& ch : channels) for (const Detection& d : ch.det) ids.insert(d.frame);
  std::vector<long> frame_ids;
  {
    const long skip = FlagInt("frame_skip"), limit = FlagInt("num_vicalib_frames");
    long k = 0;
    for (long id : ids) {
      if (skip > 0 && (k++ % (skip + 1)) != 0) continue;
      if (limit >= 0 && (long)frame_ids.size() >= limit) break;
      frame_ids.push_back(id);
    }
  }
  if (frame_ids.empty()) { std::fprintf(stderr, "F no usable frames in the detections\n"); return 1; }

  // ---- one calibrator per GPU; frames sharded contiguously, the library's own RCCL communicator does the per-iteration
  // all-reduces (-gpus 1: plain single-device run, no communicator) -----------------------------------------------------
  const int n_gpus = std::max(1, (int)FlagInt("gpus"));
  if ((size_t)n_gpus * 2 > frame_ids.size()) { std::fprintf(stderr, "F -gpus %d needs at least %d frames\n", n_gpus, 2 * n_gpus); return 1; }
  std::vector<std::unique_ptr<vic::ViCalibrator>> cals((size_t)n_gpus);
  for (int r = 0; r < n_gpus; ++r) {
    try { cals[r].reset(new vic::ViCalibrator((int)FlagInt("device") + r)); }
    catch (const std::exception& e) { std::fprintf(stderr, "F %s (device %d)\n", e.what(), (int)FlagInt("device") + r); return 3; }
  }
  const bool guess = FlagBool("has_initial_guess");
  // initial time offset (vicalib-task.cc:638-662): with system time the clocks are already aligned
  double image_time_offset = 0.0;
  {
    double t0f = (double)frame_ids[0] / FlagDouble("frame_rate");
    for (const Channel& ch : channels) { auto it = ch.frame_time.find(frame_ids[0]); if (it != ch.frame_time.end()) { t0f = it->second; break; } }
    if (calibrate_imu && FlagBool("find_time_offset") && !FlagBool("use_system_time") && !imu.time.empty()) image_time_offset = imu.time[0] - t0f;
  }
  long n_obs = 0; int seeded = 0;
  for (int r = 0; r < n_gpus; ++r) {
    vic::ViCalibrator& cal = *cals[r];
    cal.SetSigmas(FlagDouble("gyro_sigma"), FlagDouble("accel_sigma"));                     // vicalib-engine.cc:301-303
    const double zeros[6] = {0, 0, 0, 0, 0, 0}, ones[6] = {1, 1, 1, 1, 1, 1};
    cal.SetBiases(zeros); cal.SetScaleFactor(ones);
    cal.FixCameraIntrinsics(!FlagBool("calibrate_intrinsics"));                            // vicalib-task.cc:128
    for (const vic::CameraAndPose& c : input_cameras)
      if (cal.AddCamera(c) < 0) { std::fprintf(stderr, "F AddCamera failed (model %s, %zu parameters)\n", ModelName(c.model), c.params.size()); return 1; }
    // every shard gets the whole IMU stream: its last block reaches into the next shard's first frame
    if (have_imu && !imu.time.empty() && cal.AddImuMeasurements((int)imu.time.size(), imu.gyro.data(), imu.accel.data(), imu.time.data()) != VC_OK) {
      std::fprintf(stderr, "F IMU measurements rejected\n"); return 1;
    }
    const size_t lo = frame_ids.size() * (size_t)r / (size_t)n_gpus, hi = frame_ids.size() * (size_t)(r + 1) / (size_t)n_gpus;
    std::map<long, int> frame_index;
    vic::Se3 placeholder; placeholder.v = {{0, 0, 0, 1, 0, 0, 1000}};                       // vicalib-task.cc:241-244
    for (size_t k = lo; k < hi; ++k) {
      const long id = frame_ids[k];
      double t = (double)id / FlagDouble("frame_rate");
      for (const Channel& ch : channels) { auto it = ch.frame_time.find(id); if (it != ch.frame_time.end()) { t = it->second; break; } }
      frame_index[id] = cal.AddFrame(placeholder, t + image_time_offset);
    }
    std::vector<double> pw, pc;
    for (size_t c = 0; c < n_cam; ++c) {
      std::map<int, std::vector<const Detection*>> per_frame;
      for (const Detection& d : channels[c].det) {
        auto it = frame_index.find(d.frame);
        if (it == frame_index.end()) continue;
        if (d.dot >= grid_w * grid_h) continue;                 // outside the declared grid (vicalib-task.cc:353-354)
        per_frame[it->second].push_back(&d);
      }
      for (const auto& kv : per_frame) {
        pw.clear(); pc.clear();
        for (const Detection* d : kv.second) {
          pw.insert(pw.end(), {d->X, d->Y, d->Z}); pc.insert(pc.end(), {d->u, d->v});
          if (FlagBool("output_conics")) std::printf("%ld,%d,%.10g,%.10g,%.10g,%.10g,%.10g\n", d->frame, d->dot, d->u, d->v, d->X, d->Y, d->Z);
        }
        cal.AddObservations(kv.first, c, (int)kv.second.size(), pw.data(), pc.data());
        n_obs += (long)kv.second.size();
      }
    }
    cal.SetPnPRansac((int)FlagInt("pnp_ransac_its"), FlagDouble("pnp_ransac_tol"));
    seeded += cal.InitFramePosesPnP();
    // ---- VicalibTask::Start(has_initial_guess) (vicalib-task.cc:226-234) + flags read inside the calibrator ---------
    cal.SetOptimizationFlags(guess, guess && calibrate_imu, !guess, FlagBool("find_time_offset"));
    cal.SetFunctionTolerance(FlagDouble("function_tolerance"));
    cal.SetMaxIters((int)FlagInt("max_iters"));
    cal.SetCalibrateImu(calibrate_imu);
    cal.SetRemoveOutliers(FlagBool("remove_outliers"), FlagDouble("outlier_threshold"));
  }
  std::fprintf(stderr, "I %zu cameras, %zu frames on %d GPU(s) (%d with a PnP seed), %ld corner observations, %zu IMU samples\n", n_cam, frame_ids.size(), n_gpus, seeded, n_obs, imu.time.size());
  if (n_gpus > 1) {
    char id[128];
    if (vc_rccl_unique_id(id) != VC_OK) { std::fprintf(stderr, "F RCCL is not available (librccl.so)\n"); return 3; }
    std::vector<int> rc((size_t)n_gpus, 0);
    std::vector<std::thread> th;
    for (int r = 0; r < n_gpus; ++r) th.emplace_back([&, r] { rc[r] = vc_set_shard_rccl(cals[r]->handle(), r, n_gpus, id); });   // ncclCommInitRank: collective
    for (std::thread& t : th) t.join();
    for (int r = 0; r < n_gpus; ++r) if (rc[r] != VC_OK) { std::fprintf(stderr, "F RCCL communicator setup failed on rank %d (%d)\n", r, rc[r]); return 3; }
  }
  vic::ViCalibrator& cal = *cals[0];              // shared parameters and statistics are identical on every rank
  const auto t0 = std::chrono::steady_clock::now();
  for (auto& c : cals) c->Start();
  unsigned last_iters = ~0u;
  auto any_running = [&] { for (auto& c : cals) if (c->IsRunning()) return true; return false; };
  while (any_running()) {                                         // vicalib-engine.cc:376-431, 30 ms
    const unsigned it = cal.GetNumIterations();
    if (it != last_iters) { std::fprintf(stderr, "I iteration %u  mse %.6g\n", it, cal.MeanSquaredError()); last_iters = it; }
    std::this_thread::sleep_for(std::chrono::milliseconds(30));
  }
  for (auto& c : cals) c->Stop();
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  // frames of all shards, in order
  std::vector<vic::VicalibFrame> all_frames;
  for (auto& c : cals) for (size_t i = 0; i < c->NumFrames(); ++i) all_frames.push_back(c->GetFrame(i));

  // ---- Finish + PrintResults (vicalibrator.h:536-544) ------------------------------------------------------------------
  const std::vector<double> rmse = cal.GetCameraProjRMSE();
  std::printf("------------------------------------------\n");
  for (size_t c = 0; c < n_cam; ++c) {
    const vic::CameraAndPose cam = cal.GetCamera(c);
    std::printf("Camera: %zu (%s)\n ", c, ModelName(input_cameras[c].model));
    for (double p : cam.params) std::printf(" %.10g", p);
    std::printf("\n  T_ck [qx qy qz qw tx ty tz]:");
    for (double p : cam.T_ck.v) std::printf(" %.10g", p);
    std::printf("\n  reprojection RMSE: %.6g px\n", rmse[c]);
  }
  if (calibrate_imu) {
    const auto b = cal.GetBiases(), s = cal.GetScaleFactor(); const auto g = cal.GetGravity();
    std::printf("IMU biases (gyro, accel): %.8g %.8g %.8g  %.8g %.8g %.8g\n", b[0], b[1], b[2], b[3], b[4], b[5]);
    std::printf("IMU scale factors:        %.8g %.8g %.8g  %.8g %.8g %.8g\n", s[0], s[1], s[2], s[3], s[4], s[5]);
    std::printf("gravity direction: %.8g %.8g   time offset: %.9g s\n", g[0], g[1], cal.time_offset());
  }
  std::printf("iterations: %u  mse: %.8g  solve time: %.3f s\n", cal.GetNumIterations(), cal.MeanSquaredError(), secs);
  if (FlagBool("print_covariance")) {       // GetSolutionCovariance + its log lines (vicalibrator.h:802-857, :1004-1013)
    std::vector<std::vector<double>> covs((size_t)n_gpus);
    std::vector<int> dims((size_t)n_gpus, 0);
    std::vector<std::thread> th;            // collective when the frames are sharded: every rank linearises
    for (int r = 0; r < n_gpus; ++r) th.emplace_back([&, r] { covs[r] = cals[r]->GetSolutionCovariance(&dims[r]); });
    for (auto& t : th) t.join();
    if (dims[0] > 0) {
      std::printf("Covariance calculated for blocks: %s\nSolution covariance:\n", cal.covariance_names().c_str());
      for (int i = 0; i < dims[0]; ++i) {
        for (int j = 0; j < dims[0]; ++j) std::printf(" %.6e", covs[0][(size_t)i * dims[0] + j]);
        std::printf("\n");
      }
    } else std::printf("Failed to compute covariance...\n");
  }

  // ---- WriteCalibration (vicalib-engine.cc:353-372) + poses.csv (:407-421) ----------------------------------------------
  cal.WriteCameraModels(FlagString("output"));
  if (FlagBool("print_poses")) {
    if (FILE* f = std::fopen("poses.txt", "w")) {
      for (size_t i = 0; i < all_frames.size(); ++i) { double c[6]; T2Cart(all_frames[i].t_wp_.data(), c); std::fprintf(f, "%f\t%f\t%f\t%f\t%f\t%f\n", c[0], c[1], c[2], c[3], c[4], c[5]); }
      std::fclose(f);
    }
  }
  if (FlagBool("save_poses")) {
    if (FILE* f = std::fopen("poses.csv", "w")) {
      std::fprintf(f, "%% Pose file generated with vicalib.\n%% Each line is the 12 elements from the top 3 rows of a 4x4transformation matrix, printed row major.\n");
      for (size_t i = 0; i < all_frames.size(); ++i) {
        const vic::VicalibFrame& fr = all_frames[i];
        double R[9]; RotationMatrix(fr.t_wp_.data(), R);
        std::fprintf(f, "%.10g %.10g %.10g %.10g     %.10g %.10g %.10g %.10g     %.10g %.10g %.10g %.10g\n", R[0], R[1], R[2], fr.t_wp_.v[4], R[3], R[4], R[5], fr.t_wp_.v[5], R[6], R[7], R[8], fr.t_wp_.v[6]);
      }
      std::fclose(f);
    }
  }
  // ---- IsSuccessful (vicalib-task.cc:831-856) ------------------------------------------------------------------------------
  bool success = true;
  for (size_t c = 0; c < n_cam; ++c)
    if (!(rmse[c] <= FlagDouble("max_reprojection_error"))) {
      std::fprintf(stderr, "W Reprojection error of %g was greater than maximum of %g for camera %zu\n", rmse[c], FlagDouble("max_reprojection_error"), c);
      success = false;
    }
  if (success && guess) for (size_t c = 0; c < n_cam; ++c) {
    vic::CameraAndPose now = cal.GetCamera(c); now.model = input_cameras[c].model;
    if (CameraCalibrationsDiffer(input_cameras[c], now)) { success = false; break; }
  }
  if (success && guess) {                        // vicalib-task.cc:852-853 (input_imu_biases_: the calibrator's biases at construction, :129)
    const double input_imu_biases[6] = {0, 0, 0, 0, 0, 0};
    const auto bias_now = cal.GetBiases();
    if (IMUCalibrationDiffer(input_imu_biases, bias_now.data(), FlagString("imu_diff_sense") != "corrected")) success = false;
  }
  std::printf("calibration %s -> %s\n", success ? "succeeded" : "FAILED", FlagString("output").c_str());
  return success ? 0 : 2;
}
