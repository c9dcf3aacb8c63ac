// vc_synth.cpp -- deterministic synthetic calibration problems, native generator.
//
// Stands in for the reference's sensor front-end (vicalib-task.cc:247-372: conic detection -> PnP ->
// AddFrame / AddObservation, and the IMU handler vicalib-engine.cc:557): it produces what that front-end hands to
// ViCalibrator -- cameras at the engine's start values (vicalib-engine.cc:203-263), frames with an initial pose,
// per-(frame, camera) dot detections of the planar grid p_w = spacing * (i, j, 0) (vicalib-task.cc:357-358) and
// 200 Hz IMU samples generated through the reference's own measurement model (ceres-cost-functions.h:98-102).
//
// Same specification as vicalib_amd/synth.py (the numpy generator the small parity cases use): every random number is
// a counter-based splitmix64 hash of (seed, keys...), so the two produce the same problem -- the integer outputs
// (visible dots per tile) identically, the floating-point ones up to the last bits of libm vs numpy transcendentals
// (tests/test_synth_native.py).  Everything is a pure function of (config, frame index): every rank of a sharded run
// generates its own frame range, BASELINE cfg4 / cfg5 (5e7 corners) take seconds on the host cores.
// Host code only (no HIP); built as vicalib_amd/libvicalib_synth.so; test / bench infrastructure, not the product.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace {

constexpr double kPi = 3.14159265358979323846;
constexpr double kGravity = 9.8007;      // types.h:40-42

inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
struct Hash {
  uint64_t h;
  explicit Hash(uint64_t seed) : h(splitmix64(seed)) {}
  Hash key(uint64_t k) const { Hash o = *this; o.h = splitmix64(h ^ (k * 0xD6E8FEB86659FD93ull)); return o; }
  double uniform() const { return ((double)(h >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
  double normal() const {
    const double u1 = key(0x11).uniform(), u2 = key(0x22).uniform();
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * kPi * u2);
  }
};

struct M3 { double m[9]; };
inline M3 mul(const M3& a, const M3& b) {
  M3 c;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) c.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
  return c;
}
inline M3 transpose(const M3& a) { M3 c; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) c.m[3 * i + j] = a.m[3 * j + i]; return c; }
inline void mulv(const M3& a, const double v[3], double o[3]) { for (int i = 0; i < 3; ++i) o[i] = a.m[3 * i] * v[0] + a.m[3 * i + 1] * v[1] + a.m[3 * i + 2] * v[2]; }
inline void mulTv(const M3& a, const double v[3], double o[3]) { for (int i = 0; i < 3; ++i) o[i] = a.m[i] * v[0] + a.m[3 + i] * v[1] + a.m[6 + i] * v[2]; }
inline M3 identity() { return M3{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }

M3 so3_exp(const double w[3]) {
  const double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]), th2 = th * th;
  const double a = th > 1e-8 ? std::sin(th) / th : 1.0 - th2 / 6.0;
  const double b = th > 1e-8 ? (1.0 - std::cos(th)) / th2 : 0.5 - th2 / 24.0;
  const M3 K{{0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0}};
  const M3 K2 = mul(K, K);
  M3 R = identity();
  for (int i = 0; i < 9; ++i) R.m[i] += a * K.m[i] + b * K2.m[i];
  return R;
}
void quat_from_matrix(const M3& R, double q[4]) {
  const double* m = R.m;
  const double tr = m[0] + m[4] + m[8];
  if (tr > 0) {
    const double s = std::sqrt(tr + 1.0) * 2;
    q[0] = (m[7] - m[5]) / s; q[1] = (m[2] - m[6]) / s; q[2] = (m[3] - m[1]) / s; q[3] = 0.25 * s;
  } else if (m[0] > m[4] && m[0] > m[8]) {
    const double s = std::sqrt(1.0 + m[0] - m[4] - m[8]) * 2;
    q[0] = 0.25 * s; q[1] = (m[1] + m[3]) / s; q[2] = (m[2] + m[6]) / s; q[3] = (m[7] - m[5]) / s;
  } else if (m[4] > m[8]) {
    const double s = std::sqrt(1.0 + m[4] - m[0] - m[8]) * 2;
    q[0] = (m[1] + m[3]) / s; q[1] = 0.25 * s; q[2] = (m[5] + m[7]) / s; q[3] = (m[2] - m[6]) / s;
  } else {
    const double s = std::sqrt(1.0 + m[8] - m[0] - m[4]) * 2;
    q[0] = (m[2] + m[6]) / s; q[1] = (m[5] + m[7]) / s; q[2] = 0.25 * s; q[3] = (m[3] - m[1]) / s;
  }
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; ++i) q[i] /= n;
}
M3 quat_to_matrix(const double q[4]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  return M3{{1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
             2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
             2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)}};
}

// SURVEY 9.1; model ids as in vicalib_amd.h (fov, poly2, poly3, kb4, linear, rational6)
constexpr int kNk[6] = {5, 6, 7, 8, 4, 10};
const double kGtK[6][10] = {
    {330.0, 330.0, 320.0, 240.0, 0.92},
    {400.0, 400.0, 320.0, 240.0, -0.28, 0.09},
    {400.0, 400.0, 320.0, 240.0, -0.28, 0.09, -0.012},
    {260.0, 260.0, 320.0, 240.0, -0.012, 0.004, -0.0015, 0.0002},
    {400.0, 400.0, 320.0, 240.0},
    {400.0, 400.0, 320.0, 240.0, 0.12, 0.05, 0.004, 0.40, -0.04, 0.002}};
void project(int model, const double* K, const double P[3], double pix[2]) {
  const double X = P[0], Y = P[1], Z = P[2];
  if (model == 3) {
    const double rxy = std::sqrt(X * X + Y * Y), th = std::atan2(rxy, Z), th2 = th * th;
    const double r = th * (1 + th2 * (K[4] + th2 * (K[5] + th2 * (K[6] + th2 * K[7]))));
    const double c = rxy > 0 ? X / rxy : 1.0, s = rxy > 0 ? Y / rxy : 0.0;
    pix[0] = K[0] * r * c + K[2]; pix[1] = K[1] * r * s + K[3];
    return;
  }
  const double x = X / Z, y = Y / Z, r2 = x * x + y * y;
  double fac = 1.0;
  if (model == 0) {
    const double r = std::sqrt(r2), w = K[4], m = 2.0 * std::tan(w / 2.0);
    fac = r2 < 1e-5 ? m / w : std::atan(r * m) / (r * w);
  } else if (model == 1) fac = 1 + K[4] * r2 + K[5] * r2 * r2;
  else if (model == 2) fac = 1 + K[4] * r2 + K[5] * r2 * r2 + K[6] * r2 * r2 * r2;
  else if (model == 5) fac = (1 + K[4] * r2 + K[5] * r2 * r2 + K[6] * r2 * r2 * r2) / (1 + K[7] * r2 + K[8] * r2 * r2 + K[9] * r2 * r2 * r2);
  pix[0] = K[0] * x * fac + K[2]; pix[1] = K[1] * y * fac + K[3];
}

struct Config {
  int n_cams; int models[8]; int grid, n_frames, imu; long long seed; int width, height;
  double frame_rate, imu_rate, pixel_sigma, pose_sigma_t, pose_sigma_r; long long first_frame; int threads, extrinsics_prior;
};
struct Grid { int gw, gh; double sp, w, h; };
Grid grid_of(int g) {
  Grid r;
  if (g == 1) { r.gw = 25; r.gh = 36; r.sp = 0.03156; } else { r.gw = 19; r.gh = 10; r.sp = 0.254 / 18.0; }
  r.w = (r.gw - 1) * r.sp; r.h = (r.gh - 1) * r.sp;
  return r;
}
// camera-0 pose in the world at time t: position, R_wc (columns = camera axes in the world)
void trajectory(double t, const Grid& g, double p[3], M3* R) {
  const double cx = 0.5 * g.w, cy = 0.5 * g.h;
  const double d = g.w * (0.625 + 0.275 * std::sin(2 * kPi * t / 6.3 + 1.0));
  p[0] = cx + 0.30 * g.w * std::sin(2 * kPi * t / 3.1);
  p[1] = cy + 0.30 * g.h * std::sin(2 * kPi * t / 4.7 + 0.5);
  p[2] = -d;
  if (!R) return;
  const double tx = cx + 0.25 * g.w * std::sin(2 * kPi * t / 5.3 + 2.0), ty = cy + 0.25 * g.h * std::sin(2 * kPi * t / 3.7 + 0.3);
  double z[3] = {tx - p[0], ty - p[1], 0.0 - p[2]};
  const double zn = std::sqrt(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
  for (int i = 0; i < 3; ++i) z[i] /= zn;
  const double roll = (25.0 * kPi / 180.0) * std::sin(2 * kPi * t / 7.9 + 0.7);
  const double up[3] = {std::sin(roll), std::cos(roll), 0.0};
  double xa[3] = {up[1] * z[2] - up[2] * z[1], up[2] * z[0] - up[0] * z[2], up[0] * z[1] - up[1] * z[0]};
  const double xn = std::sqrt(xa[0] * xa[0] + xa[1] * xa[1] + xa[2] * xa[2]);
  for (int i = 0; i < 3; ++i) xa[i] /= xn;
  const double ya[3] = {z[1] * xa[2] - z[2] * xa[1], z[2] * xa[0] - z[0] * xa[2], z[0] * xa[1] - z[1] * xa[0]};
  for (int i = 0; i < 3; ++i) { R->m[3 * i] = xa[i]; R->m[3 * i + 1] = ya[i]; R->m[3 * i + 2] = z[i]; }
}

}  // namespace

struct vcs_problem {
  Config cfg;
  std::vector<double> grid;                       // M x 3
  std::vector<double> K_gt, K_init;               // C x 10 (padded)
  std::vector<double> T_ck_gt, T_ck_init;         // C x 7
  std::vector<double> frame_time, T_wk_gt, T_wk_init, v_gt;
  std::vector<int> tile_frame, tile_cam;
  std::vector<long long> tile_off;
  std::vector<int> ids;
  std::vector<double> pix;
  std::vector<double> imu_t, imu_gyro, imu_accel;
  double imu_gt[15];                              // bg(3) ba(3) sg(3) sa(3) g_dir(2) time_offset
};

extern "C" {

vcs_problem* vcs_generate(const void* cfg_bytes, int cfg_size) {
  if (!cfg_bytes || cfg_size != (int)sizeof(Config)) return nullptr;
  vcs_problem* P = new vcs_problem();
  std::memcpy(&P->cfg, cfg_bytes, sizeof(Config));
  const Config& cfg = P->cfg;
  const int C = cfg.n_cams, N = cfg.n_frames;
  if (C < 1 || C > 8 || N < 0) { delete P; return nullptr; }
  for (int c = 0; c < C; ++c) if (cfg.models[c] < 0 || cfg.models[c] > 5) { delete P; return nullptr; }
  const Grid g = grid_of(cfg.grid);
  const int M = g.gw * g.gh;
  P->grid.resize((size_t)M * 3);
  for (int i = 0; i < g.gw; ++i) for (int j = 0; j < g.gh; ++j) {
    double* o = &P->grid[3 * (size_t)(i * g.gh + j)];
    o[0] = i * g.sp; o[1] = j * g.sp; o[2] = 0.0;
  }
  const uint64_t seed = (uint64_t)cfg.seed;
  // cameras --------------------------------------------------------------------------------------------
  const M3 rdf{{0, 1, 0, 0, 0, 1, 1, 0, 0}};
  const M3 R_ck0 = cfg.imu ? rdf : identity();
  P->K_gt.assign((size_t)C * 10, 0.0); P->K_init.assign((size_t)C * 10, 0.0);
  P->T_ck_gt.assign((size_t)C * 7, 0.0); P->T_ck_init.assign((size_t)C * 7, 0.0);
  std::vector<M3> R_ck(C);
  std::vector<double> t_ck((size_t)C * 3);
  for (int c = 0; c < C; ++c) {
    const int m = cfg.models[c], nk = kNk[m];
    for (int i = 0; i < nk; ++i) P->K_gt[(size_t)c * 10 + i] = kGtK[m][i] * (1.0 + 0.02 * (2.0 * Hash(seed + 17).key(c).key(i).uniform() - 1.0));
    double* k0 = &P->K_init[(size_t)c * 10];
    k0[0] = 300.0; k0[1] = 300.0; k0[2] = cfg.width / 2.0; k0[3] = cfg.height / 2.0;
    if (m == 0) k0[4] = 0.2;
    double w[3];
    for (int i = 0; i < 3; ++i) w[i] = (3.0 * kPi / 180.0) * (2.0 * Hash(seed + 29).key(c).key(i).uniform() - 1.0) * (c > 0 ? 1.0 : 0.0);
    const M3 R_c_c0 = so3_exp(w);
    const double b[3] = {0.06 * c, 0.0, 0.0};
    double t[3];
    mulv(R_c_c0, b, t);
    const M3 Rck = mul(R_c_c0, R_ck0);
    double* T = &P->T_ck_gt[(size_t)c * 7];
    quat_from_matrix(Rck, T);
    for (int i = 0; i < 3; ++i) T[4 + i] = -t[i];
    P->T_ck_init[(size_t)c * 7 + 3] = 1.0;
    if (cfg.extrinsics_prior && c > 0) {      // a rough prior on the rig instead of the engine's identity (synth.py: extrinsics_prior)
      double dw[3], dtr[3];
      for (int i = 0; i < 3; ++i) {
        dw[i] = (0.5 * kPi / 180.0) * (2.0 * Hash(seed + 31).key(c).key(i).uniform() - 1.0);
        dtr[i] = 0.01 * (2.0 * Hash(seed + 37).key(c).key(i).uniform() - 1.0);
      }
      double* Ti = &P->T_ck_init[(size_t)c * 7];
      quat_from_matrix(mul(so3_exp(dw), R_c_c0), Ti);
      for (int i = 0; i < 3; ++i) Ti[4 + i] = -t[i] + dtr[i];
    }
    R_ck[c] = quat_to_matrix(T);                 // the detections are generated from the stored quaternion, as in synth.py
    for (int i = 0; i < 3; ++i) t_ck[(size_t)c * 3 + i] = T[4 + i];
  }
  // frames + detections, frame ranges in parallel -----------------------------------------------------------
  P->frame_time.resize(N); P->T_wk_gt.resize((size_t)N * 7); P->T_wk_init.resize((size_t)N * 7); P->v_gt.resize((size_t)N * 3);
  int nthreads = cfg.threads > 0 ? cfg.threads : (int)std::thread::hardware_concurrency();
  nthreads = std::max(1, std::min(nthreads, std::max(1, N / 64)));
  struct Part { std::vector<int> tf, tc, cnt, ids; std::vector<double> pix; };
  std::vector<Part> parts(nthreads);
  auto work = [&](int th) {
    Part& part = parts[th];
    const int f0 = (int)((long long)N * th / nthreads), f1 = (int)((long long)N * (th + 1) / nthreads);
    std::vector<int> vis(M);
    std::vector<double> px((size_t)M * 2);
    for (int n = f0; n < f1; ++n) {
      const long long fi = cfg.first_frame + n;
      const double ft = 1.0 + (double)fi / cfg.frame_rate;
      P->frame_time[n] = ft;
      double p[3], pp[3], pm[3];
      M3 Rwc;
      trajectory(ft, g, p, &Rwc);
      const M3 Rwk = mul(Rwc, R_ck0);
      double* Tg = &P->T_wk_gt[(size_t)n * 7];
      quat_from_matrix(Rwk, Tg);
      for (int i = 0; i < 3; ++i) Tg[4 + i] = p[i];
      const double h = 1e-5;
      trajectory(ft + h, g, pp, nullptr); trajectory(ft - h, g, pm, nullptr);
      for (int i = 0; i < 3; ++i) P->v_gt[(size_t)n * 3 + i] = (pp[i] - pm[i]) / (2 * h);
      double dt[3], dr[3];
      for (int i = 0; i < 3; ++i) {
        dt[i] = cfg.pose_sigma_t * Hash(seed + 41).key((uint64_t)fi).key(i).normal();
        dr[i] = cfg.pose_sigma_r * Hash(seed + 43).key((uint64_t)fi).key(i).normal();
      }
      const M3 Rinit = mul(Rwc, so3_exp(dr));
      double* Ti = &P->T_wk_init[(size_t)n * 7];
      quat_from_matrix(Rinit, Ti);
      double rdt[3];
      mulv(Rwc, dt, rdt);
      for (int i = 0; i < 3; ++i) Ti[4 + i] = p[i] + rdt[i];
      for (int c = 0; c < C; ++c) {
        const int m = cfg.models[c];
        const double* K = &P->K_gt[(size_t)c * 10];
        const double max_ang = (m == 3 ? 75.0 : 55.0) * kPi / 180.0;
        int cnt = 0;
        for (int j = 0; j < M; ++j) {
          const double* pw = &P->grid[3 * (size_t)j];
          const double d[3] = {pw[0] - p[0], pw[1] - p[1], pw[2] - p[2]};
          double pk[3], pc[3], pxy[2];
          mulTv(Rwk, d, pk);
          mulv(R_ck[c], pk, pc);
          for (int i = 0; i < 3; ++i) pc[i] += t_ck[(size_t)c * 3 + i];
          project(m, K, pc, pxy);
          bool ok = pc[2] > 1e-3 && pxy[0] >= 5 && pxy[0] <= cfg.width - 5 && pxy[1] >= 5 && pxy[1] <= cfg.height - 5;
          ok = ok && std::atan2(std::sqrt(pc[0] * pc[0] + pc[1] * pc[1]), pc[2]) < max_ang;
          if (!ok) continue;
          const Hash hn = Hash(seed + 5678).key((uint64_t)fi).key(c).key(j);
          px[2 * (size_t)cnt] = pxy[0] + cfg.pixel_sigma * hn.key(0).normal();
          px[2 * (size_t)cnt + 1] = pxy[1] + cfg.pixel_sigma * hn.key(1).normal();
          vis[cnt++] = j;
        }
        if (cnt >= 4) {
          part.tf.push_back(n); part.tc.push_back(c); part.cnt.push_back(cnt);
          part.ids.insert(part.ids.end(), vis.begin(), vis.begin() + cnt);
          part.pix.insert(part.pix.end(), px.begin(), px.begin() + 2 * (size_t)cnt);
        }
      }
    }
  };
  {
    std::vector<std::thread> pool;
    for (int th = 1; th < nthreads; ++th) pool.emplace_back(work, th);
    work(0);
    for (auto& t : pool) t.join();
  }
  size_t nt = 0, no = 0;
  for (const Part& p : parts) { nt += p.tf.size(); no += p.ids.size(); }
  P->tile_frame.reserve(nt); P->tile_cam.reserve(nt); P->tile_off.reserve(nt + 1); P->ids.reserve(no); P->pix.reserve(2 * no);
  P->tile_off.push_back(0);
  for (Part& p : parts) {
    P->tile_frame.insert(P->tile_frame.end(), p.tf.begin(), p.tf.end());
    P->tile_cam.insert(P->tile_cam.end(), p.tc.begin(), p.tc.end());
    for (int c : p.cnt) P->tile_off.push_back(P->tile_off.back() + c);
    P->ids.insert(P->ids.end(), p.ids.begin(), p.ids.end());
    P->pix.insert(P->pix.end(), p.pix.begin(), p.pix.end());
    Part().tf.swap(p.tf); std::vector<int>().swap(p.ids); std::vector<double>().swap(p.pix);
  }
  // IMU ------------------------------------------------------------------------------------------------
  if (cfg.imu && N > 0) {
    const double bg[3] = {0.002, -0.001, 0.0015}, ba[3] = {0.03, -0.02, 0.05}, sg[3] = {1.01, 0.99, 1.005}, sa[3] = {0.995, 1.01, 0.99};
    const double gd[2] = {0.03, -0.02}, toff = 0.003;
    std::memcpy(P->imu_gt, bg, 24); std::memcpy(P->imu_gt + 3, ba, 24); std::memcpy(P->imu_gt + 6, sg, 24); std::memcpy(P->imu_gt + 9, sa, 24);
    P->imu_gt[12] = gd[0]; P->imu_gt[13] = gd[1]; P->imu_gt[14] = toff;
    const double t0 = P->frame_time.front() - 0.1, t1 = P->frame_time.back() + 0.1;
    const long long k0 = (long long)std::floor(t0 * cfg.imu_rate), k1 = (long long)std::ceil(t1 * cfg.imu_rate);
    const size_t S = (size_t)(k1 - k0 + 1);
    P->imu_t.resize(S); P->imu_gyro.resize(S * 3); P->imu_accel.resize(S * 3);
    const double gw[3] = {-kGravity * std::cos(gd[0]) * std::sin(gd[1]), kGravity * std::sin(gd[0]), -kGravity * std::cos(gd[0]) * std::cos(gd[1])};
    int nth = std::max(1, std::min(cfg.threads > 0 ? cfg.threads : (int)std::thread::hardware_concurrency(), (int)(S / 1024) + 1));
    auto imu_work = [&](int th) {
      const size_t s0 = S * th / nth, s1 = S * (th + 1) / nth;
      for (size_t s = s0; s < s1; ++s) {
        const long long k = k0 + (long long)s;
        const double ti = (double)k / cfg.imu_rate, tb = ti + toff, h = 1e-4;
        double p0[3], pp[3], pm[3];
        M3 R0, Rp, Rm;
        trajectory(tb, g, p0, &R0); trajectory(tb + h, g, pp, &Rp); trajectory(tb - h, g, pm, &Rm);
        double aw[3];
        for (int i = 0; i < 3; ++i) aw[i] = (pp[i] - 2 * p0[i] + pm[i]) / (h * h);
        const M3 Rk0 = mul(R0, R_ck0), Rkp = mul(Rp, R_ck0), Rkm = mul(Rm, R_ck0);
        M3 dR;
        for (int i = 0; i < 9; ++i) dR.m[i] = (Rkp.m[i] - Rkm.m[i]) / (2 * h);
        const M3 Wx = mul(dR, transpose(Rk0));
        const double ww[3] = {(Wx.m[7] - Wx.m[5]) * 0.5, (Wx.m[2] - Wx.m[6]) * 0.5, (Wx.m[3] - Wx.m[1]) * 0.5};
        double wb[3], ab[3];
        const double ag[3] = {aw[0] + gw[0], aw[1] + gw[1], aw[2] + gw[2]};
        mulTv(Rk0, ww, wb); mulTv(Rk0, ag, ab);
        P->imu_t[s] = ti;
        for (int i = 0; i < 3; ++i) {
          P->imu_gyro[3 * s + i] = (wb[i] - bg[i]) / sg[i] + 5.3088444e-5 * Hash(seed + 9001).key((uint64_t)k).key(i).normal();
          P->imu_accel[3 * s + i] = (ab[i] - ba[i]) / sa[i] + 0.001883649 * Hash(seed + 9002).key((uint64_t)k).key(i).normal();
        }
      }
    };
    std::vector<std::thread> pool;
    for (int th = 1; th < nth; ++th) pool.emplace_back(imu_work, th);
    imu_work(0);
    for (auto& t : pool) t.join();
  }
  return P;
}

void vcs_free(vcs_problem* p) { delete p; }

// what: 0 grid points, 1 tiles, 2 observations, 3 IMU samples, 4 frames, 5 cameras
long long vcs_count(const vcs_problem* p, int what) {
  if (!p) return -1;
  switch (what) {
    case 0: return (long long)(p->grid.size() / 3);
    case 1: return (long long)p->tile_frame.size();
    case 2: return (long long)p->ids.size();
    case 3: return (long long)p->imu_t.size();
    case 4: return (long long)p->frame_time.size();
    case 5: return p->cfg.n_cams;
    default: return -1;
  }
}
// what: 0 grid (double), 1 K_gt, 2 K_init (C x 10), 3 T_ck_gt, 4 T_ck_init (C x 7), 5 frame_time, 6 T_wk_gt, 7 T_wk_init,
// 8 v_gt, 9 tile_frame (int), 10 tile_cam (int), 11 tile_off (long long), 12 ids (int), 13 pix (double), 14 imu_t,
// 15 imu_gyro, 16 imu_accel, 17 imu_gt (15 doubles)
const void* vcs_array(const vcs_problem* p, int what) {
  if (!p) return nullptr;
  switch (what) {
    case 0: return p->grid.data(); case 1: return p->K_gt.data(); case 2: return p->K_init.data();
    case 3: return p->T_ck_gt.data(); case 4: return p->T_ck_init.data(); case 5: return p->frame_time.data();
    case 6: return p->T_wk_gt.data(); case 7: return p->T_wk_init.data(); case 8: return p->v_gt.data();
    case 9: return p->tile_frame.data(); case 10: return p->tile_cam.data(); case 11: return p->tile_off.data();
    case 12: return p->ids.data(); case 13: return p->pix.data(); case 14: return p->imu_t.data();
    case 15: return p->imu_gyro.data(); case 16: return p->imu_accel.data(); case 17: return p->imu_gt;
    default: return nullptr;
  }
}
int vcs_config_size(void) { return (int)sizeof(Config); }

}  // extern "C"
