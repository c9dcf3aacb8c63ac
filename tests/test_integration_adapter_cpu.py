"""INTEGRATION.md's adapter (the class a vicalib maintainer would paste in place of vicalibrator.h) is compiled here against stub
declarations of what it includes -- Calibu's camera classes, Sophus::SE3d, Eigen vectors, glog's LOG(FATAL), the engine's gflags --
and against the REAL include/vicalib_amd.h; a mock of the C entry points records what the adapter passes.  Checks: the text compiles;
every camera type of vicalibrator.h:412-453 maps to its VC_MODEL_* constant (round 3's text sent Rational6Camera to LINEAR);
an unknown type and a camera the library refuses end in LOG(FATAL) (as the reference's CHECK does), never in a camera that exists on
one side only."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUBS = {
    "calibu/Calibu.h": r'''
#pragma once
#include <memory>
#include <string>
#include <vector>
#include <stdexcept>
#include <sstream>
namespace Eigen {
struct VectorXd { std::vector<double> v; double* data() { return v.data(); } };
struct Vector3d { double v[3]; const double* data() const { return v; } };
struct Vector2d { double v[2]; const double* data() const { return v; } };
}
namespace calibu {
template <class S> struct CameraInterface {
  Eigen::VectorXd p; unsigned w = 640, h = 480;
  explicit CameraInterface(int n) { p.v.assign(n, 0.5); }
  virtual ~CameraInterface() {}
  Eigen::VectorXd& GetParams() { return p; }
  unsigned NumParams() const { return (unsigned)p.v.size(); }
  unsigned Width() const { return w; }
  unsigned Height() const { return h; }
  virtual std::string Type() const = 0;
};
#define STUB_CAM(Name, N) template <class S> struct Name : CameraInterface<S> { Name() : CameraInterface<S>(N) {} std::string Type() const override { return #Name; } };
STUB_CAM(FovCamera, 5) STUB_CAM(Poly2Camera, 6) STUB_CAM(Poly3Camera, 7) STUB_CAM(Rational6Camera, 10) STUB_CAM(KannalaBrandtCamera, 8) STUB_CAM(LinearCamera, 4)
STUB_CAM(SomeOtherCamera, 4)
}
// glog / gflags stand-ins
struct FatalStream { std::ostringstream s; template <class T> FatalStream& operator<<(const T& x) { s << x; return *this; } ~FatalStream() noexcept(false) { throw std::runtime_error(s.str()); } };
#define LOG(severity) FatalStream()
static int FLAGS_max_iters = 100; static bool FLAGS_calibrate_imu = true, FLAGS_remove_outliers = false; static double FLAGS_outlier_threshold = 2.0;
''',
    "sophus/se3.hpp": r'''
#pragma once
namespace Sophus { struct SE3d { double d[7] = {0, 0, 0, 1, 0, 0, 0}; const double* data() const { return d; } double* data() { return d; } }; }
''',
}

MAIN = r'''
#include <cstdio>
#include <cstring>
// ---- mock of the C entry points the adapter calls (the real library needs a GPU) ----
static int g_last_model = -99, g_n_cams = 0, g_refuse = 0;
extern "C" {
int vc_create(vc_calibrator** out, int) { *out = (vc_calibrator*)0x1; return VC_OK; }
void vc_destroy(vc_calibrator*) {}
int vc_add_camera(vc_calibrator*, int model, const double*, int nk, int, int, const double*) {
  static const int want[6] = {5, 6, 7, 8, 4, 10};
  g_last_model = model;
  if (g_refuse || model < 0 || model > 5 || nk != want[model]) return VC_ERR_BAD_ARG;
  return g_n_cams++;
}
int vc_add_frame(vc_calibrator*, const double*, double) { return 0; }
int vc_add_observations(vc_calibrator*, int, int, int, const double*, const double*) { return 0; }
int vc_add_imu(vc_calibrator*, int, const double*, const double*, const double*) { return 0; }
int vc_set_optimization_flags(vc_calibrator*, int, int, int, int) { return 0; }
int vc_set_function_tolerance(vc_calibrator*, double) { return 0; }
int vc_set_max_iters(vc_calibrator*, int) { return 0; }
int vc_set_calibrate_imu(vc_calibrator*, int) { return 0; }
int vc_set_remove_outliers(vc_calibrator*, int, double) { return 0; }
int vc_start(vc_calibrator*) { return 0; }
int vc_is_running(vc_calibrator*) { return 0; }
int vc_stop(vc_calibrator*) { return 0; }
int vc_get_camera(vc_calibrator*, int, double*, int* n, double*) { *n = 0; return 0; }
int vc_get_camera_proj_rmse(vc_calibrator*, double*) { return 0; }
double vc_mean_squared_error(vc_calibrator*) { return 0; }
unsigned vc_get_num_iterations(vc_calibrator*) { return 0; }
int vc_write_camera_models(vc_calibrator*, const char*) { return 0; }
}
using namespace visual_inertial_calibration;
template <class C> static int add(ViCalibrator& v) { return v.AddCamera(std::make_shared<C>()); }
int main() {
  ViCalibrator v;
  int ok = 1;
  ok &= add<calibu::FovCamera<double>>(v) == 0 && g_last_model == VC_MODEL_FOV;
  ok &= add<calibu::Poly2Camera<double>>(v) == 1 && g_last_model == VC_MODEL_POLY2;
  ok &= add<calibu::Poly3Camera<double>>(v) == 2 && g_last_model == VC_MODEL_POLY3;
  ok &= add<calibu::Rational6Camera<double>>(v) == 3 && g_last_model == VC_MODEL_RATIONAL6;
  ok &= add<calibu::KannalaBrandtCamera<double>>(v) == 4 && g_last_model == VC_MODEL_KB4;
  ok &= add<calibu::LinearCamera<double>>(v) == 5 && g_last_model == VC_MODEL_LINEAR;
  std::printf("mapping %s\n", ok ? "ok" : "WRONG");
  int fatal_unknown = 0, fatal_refused = 0;
  try { add<calibu::SomeOtherCamera<double>>(v); } catch (const std::runtime_error& e) { fatal_unknown = std::strstr(e.what(), "SomeOtherCamera") != nullptr; }
  g_refuse = 1;
  try { add<calibu::FovCamera<double>>(v); } catch (const std::runtime_error&) { fatal_refused = 1; }
  std::printf("unknown type fatal %d, refused camera fatal %d, cameras kept by the adapter %d\n", fatal_unknown, fatal_refused, (int)v.GetCameraProjRMSE().size());
  return (ok && fatal_unknown && fatal_refused && v.GetCameraProjRMSE().size() == 6) ? 0 : 1;
}
'''


def test_integration_adapter_compiles_and_maps_every_camera_type(tmp_path):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## Adapter a maintainer would add"):]
    adapter = re.search(r"```cpp\n(.*?)```", sec, re.S).group(1)
    assert "Rational6Camera" in adapter and "VC_MODEL_RATIONAL6" in adapter
    for rel, body in STUBS.items():
        path = tmp_path / rel
        path.parent.mkdir(parents=True, exist_ok=True)
        path.write_text(body)
    src = tmp_path / "adapter_main.cpp"
    src.write_text(adapter + MAIN)
    exe = tmp_path / "adapter_main"
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wno-unused-variable", "-I", str(tmp_path), "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mapping ok" in r.stdout
