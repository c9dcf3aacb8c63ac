// vc_calibrator.cpp -- host driver behind the C ABI (include/vicalib_amd.h).
//
// Mirrors visual_inertial_calibration::ViCalibrator (include/vicalib/vicalibrator.h): the problem
// container (AddCamera :332, AddFrame :355, AddObservation :385, AddImuMeasurements :370), the
// constancy rules and residual multiplicities of SetupProblem (:548-679), the iteration callback
// (:690-721), per-camera RMSE (:958-971), RemoveOutliers (:859-916) and the SolveThread stage machine
// (:919-1040).  Where the reference hands a ceres::Problem to ceres::Solve (:956) this driver runs a
// trust-region Levenberg-Marquardt loop (the Ceres algorithm: Jacobi scaling, diagonal clamp, step
// quality, radius update) whose every O(observations) and O(frames) step is a HIP kernel
// (vc_kernels.hip).  The host only takes the accept/reject decision from a handful of scalars.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/vicalib_amd.h"
#include "vc_device.h"
#include "vc_math.hpp"
#include "vc_pnp.hpp"
#include "vc_grid.hpp"
#include "vc_imu.hpp"

using namespace vc;

#define HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { last_hip_error = e_; if (std::getenv("VC_DEBUG")) std::fprintf(stderr, "[vicalib_amd] %s:%d %s -> %s\n", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); return VC_ERR_NO_DEVICE; } } while (0)

namespace {

template <class T> struct DBuf {
  T* p = nullptr; size_t n = 0;
  ~DBuf() { release(); }
  void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
  hipError_t alloc(size_t count) {
    if (count <= n && p) return hipSuccess;
    release();
    if (count == 0) count = 1;
    hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
    if (e == hipSuccess) n = count;
    return e;
  }
  hipError_t upload(const std::vector<T>& h, hipStream_t s) {
    hipError_t e = alloc(h.size());
    if (e != hipSuccess || h.empty()) return e;
    return hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s);
  }
};

// One stage's small uploads, packed: arrays are appended to a page-locked staging image (grow-only), every destination is a
// segment; flush() sends image + segment table with one copy and scatters them with one kernel (launch_unpack).  A source may
// serve several destinations; zero() adds a fill.  Everything is ordered on the calibrator's stream like the copies it replaces.
struct Packer {
  char* host = nullptr; size_t cap = 0, used = 0;
  DBuf<char> dev;
  std::vector<UnpackSeg> segs;
  ~Packer() { if (host) (void)hipHostFree(host); }
  hipError_t reserve(size_t need) {
    if (need <= cap) return hipSuccess;
    const size_t ncap = std::max(need, cap * 2 + (1u << 16));
    char* nh = nullptr;
    hipError_t e = hipHostMalloc((void**)&nh, ncap, hipHostMallocDefault);
    if (e != hipSuccess) return e;
    if (host) { std::memcpy(nh, host, used); (void)hipHostFree(host); }
    host = nh; cap = ncap;
    return hipSuccess;
  }
  void begin() { used = 0; segs.clear(); }
  // appends the bytes, returns their offset (16-byte aligned) or ~0 on failure
  size_t put(const void* p, size_t bytes) {
    const size_t off = (used + 15) & ~(size_t)15;
    if (reserve(off + bytes + 16) != hipSuccess) return ~(size_t)0;
    if (bytes) std::memcpy(host + off, p, bytes);
    used = off + bytes;
    return off;
  }
  template <class T> hipError_t add(DBuf<T>& d, const std::vector<T>& h) {
    static_assert(sizeof(T) % 4 == 0, "segments are copied in 32-bit words");
    hipError_t e = d.alloc(h.size());
    if (e != hipSuccess || h.empty()) return e;
    const size_t off = put(h.data(), h.size() * sizeof(T));
    if (off == ~(size_t)0) return hipErrorOutOfMemory;
    segs.push_back({(unsigned long long)(uintptr_t)d.p, (unsigned long long)off, (unsigned long long)(h.size() * sizeof(T))});
    return hipSuccess;
  }
  // another destination for the array added last
  template <class T> hipError_t also(DBuf<T>& d, size_t count) {
    hipError_t e = d.alloc(count);
    if (e != hipSuccess || count == 0 || segs.empty()) return e;
    UnpackSeg sg = segs.back(); sg.dst = (unsigned long long)(uintptr_t)d.p;
    segs.push_back(sg);
    return hipSuccess;
  }
  // room for `bytes` inside the image, to be filled in place (valid until the next put / slot call grows the image: reserve first)
  void* slot(size_t bytes, size_t* off) {
    *off = (used + 15) & ~(size_t)15;
    if (reserve(*off + bytes + 16) != hipSuccess) return nullptr;
    used = *off + bytes;
    return host + *off;
  }
  void seg(void* dst, size_t off, size_t bytes) { if (bytes) segs.push_back({(unsigned long long)(uintptr_t)dst, (unsigned long long)off, (unsigned long long)bytes}); }
  void zero(void* p, size_t bytes) { if (p && bytes) segs.push_back({(unsigned long long)(uintptr_t)p, ~0ull, (unsigned long long)bytes}); }
  hipError_t flush(hipStream_t s) {
    if (segs.empty()) return hipSuccess;
    const size_t img = used;
    const size_t tab = put(segs.data(), segs.size() * sizeof(UnpackSeg));
    if (tab == ~(size_t)0) return hipErrorOutOfMemory;
    hipError_t e = dev.alloc(used + 16);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(dev.p, host, used, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    launch_unpack((const UnpackSeg*)(dev.p + tab), (int)segs.size(), dev.p, img, s);
    return hipGetLastError();
  }
};

// a fixed set of timing events, destroyed on every exit path
template <int N> struct EventSet {
  hipEvent_t e[N] = {};
  bool create() { for (int i = 0; i < N; ++i) if (hipEventCreate(&e[i]) != hipSuccess) return false; return true; }
  ~EventSet() { for (int i = 0; i < N; ++i) if (e[i]) (void)hipEventDestroy(e[i]); }
};

struct HostCam { int model, nk, width, height; double K[10]; double T_ck[7]; };
struct HostFrame { double T[7]; double v[3]; double time; };
struct IterRecord { int iteration; double cost, cost_change, gmax, gnorm, step_norm, rho, radius; int accepted, stage; };
enum Termination { kConvergence = 0, kNoConvergence = 1, kUserSuccess = 2, kFailure = 3 };

struct PointKey {
  double x, y, z;
  bool operator==(const PointKey& o) const { return std::memcmp(this, &o, sizeof(PointKey)) == 0; }
};
struct PointHash {
  size_t operator()(const PointKey& k) const {
    uint64_t b[3]; std::memcpy(b, &k, 24);
    uint64_t h = 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < 3; ++i) { h ^= b[i] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); h *= 0xBF58476D1CE4E5B9ull; }
    return (size_t)h;
  }
};

// Distinct target points in the order they were first seen: open-addressing table over the bit pattern of (x, y, z).
struct PointTable {
  std::vector<double> xyz;          // 3 per point
  std::vector<int> slot;            // power-of-two sized, -1 = empty
  int size() const { return (int)(xyz.size() / 3); }
  void clear() { xyz.clear(); slot.clear(); }
  void grow() {
    const size_t cap = slot.empty() ? 1024 : slot.size() * 2;
    slot.assign(cap, -1);
    for (int i = 0; i < size(); ++i) {
      const PointKey k{xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2]};
      size_t h = PointHash()(k) & (cap - 1);
      while (slot[h] >= 0) h = (h + 1) & (cap - 1);
      slot[h] = i;
    }
  }
  int intern(const double* p) {
    if ((size_t)size() * 2 >= slot.size()) grow();
    const PointKey k{p[0], p[1], p[2]};
    const size_t mask = slot.size() - 1;
    size_t h = PointHash()(k) & mask;
    while (slot[h] >= 0) {
      const double* q = &xyz[3 * (size_t)slot[h]];
      if (std::memcmp(q, p, 24) == 0) return slot[h];
      h = (h + 1) & mask;
    }
    slot[h] = size();
    xyz.insert(xyz.end(), p, p + 3);
    return slot[h];
  }
};

}  // namespace

// ---- RCCL, bound at run time (the library is already in the process when the host is PyTorch; a plain C++ host gets
// /opt/rocm/lib/librccl.so).  Only what the per-iteration all-reduce needs.
struct RcclUniqueId { char internal[128]; };
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(RcclUniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, RcclUniqueId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;      // optional: diagnostics only
  const char* (*GetLastError)(void*) = nullptr;
  bool load() {
    if (AllReduce) return true;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) { lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (lib) break; }
    if (!lib) for (const char* n : names) { lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
    if (!lib) return false;
    GetUniqueId = (int (*)(RcclUniqueId*))dlsym(lib, "ncclGetUniqueId");
    CommInitRank = (int (*)(void**, int, RcclUniqueId, int))dlsym(lib, "ncclCommInitRank");
    AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(lib, "ncclAllReduce");
    CommDestroy = (int (*)(void*))dlsym(lib, "ncclCommDestroy");
    GetErrorString = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
    GetLastError = (const char* (*)(void*))dlsym(lib, "ncclGetLastError");
    if (!GetUniqueId || !CommInitRank || !AllReduce || !CommDestroy) { AllReduce = nullptr; return false; }
    return true;
  }
};
static RcclApi g_rccl;

// roctx ranges around the stages, solves and passes (tracing hooks), bound at run time and only on request
// (VICALIB_AMD_ROCTX=1): `rocprofv3 --marker-trace --kernel-trace` then shows which kernels belong to which LM pass of which stage.
struct RoctxApi {
  int (*Push)(const char*) = nullptr;
  int (*Pop)() = nullptr;
  bool tried = false, on = false;
  bool load() {
    if (tried) return on;
    tried = true;
    const char* e = std::getenv("VICALIB_AMD_ROCTX");
    if (!e || e[0] != '1') return false;
    void* lib = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return false;
    Push = (int (*)(const char*))dlsym(lib, "roctxRangePushA");
    Pop = (int (*)())dlsym(lib, "roctxRangePop");
    on = Push && Pop;
    return on;
  }
};
static RoctxApi g_roctx;
struct RoctxRange {
  bool live;
  explicit RoctxRange(const char* name) : live(g_roctx.load()) { if (live) (void)g_roctx.Push(name); }
  ~RoctxRange() { if (live) (void)g_roctx.Pop(); }
};
// text of the last failure of an entry point that has more to say than its status code (vc_last_error; per thread)
static thread_local std::string g_last_error;
constexpr int kNcclDouble = 8, kNcclSum = 0, kNcclMax = 2;     // ncclDataType_t / ncclRedOp_t values of nccl.h

struct vc_calibrator {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;        // the IMU weight update of a pass runs here, under the pass's Jacobian sweeps and chain solve
  hipEvent_t ev_state = nullptr, ev_weights = nullptr, ev_imujac = nullptr, ev_reduced = nullptr, ev_back = nullptr, ev_pre = nullptr;
  bool top_gram_launch = false;         // early Gram: the chain's top-level frames need a Gram launch of their own (upload)
  bool pre_weights_pending = false;     // solve_once has recorded ev_pre ahead of the weight update that precedes a solve
  bool pre_weights_fresh = false;       // ... and nothing has moved the state since: the first pass's own update would repeat it
  int wcur = 0;                         // weight buffer holding the current weight_sqrt_
  Packer pack;                          // staging image of a stage's small uploads
  bool flag_sync = false;               // hand-overs to the second stream through device flags instead of event records (set at creation)
  bool weights_behind_l0 = !(std::getenv("VICALIB_AMD_WEIGHTS_BEHIND_L0") && std::getenv("VICALIB_AMD_WEIGHTS_BEHIND_L0")[0] == '0');
  bool shard_flag_sync = std::getenv("VICALIB_AMD_SHARD_FLAG_SYNC") && std::getenv("VICALIB_AMD_SHARD_FLAG_SYNC")[0] == '1';
  long long sync_bound = 400000;        // polls before a flag wait gives up (~0.2 s); VICALIB_AMD_SYNC_BOUND (test hook: a tiny bound forces the time-out path)
  int sync_bound_from_pass = 0;         // VICALIB_AMD_SYNC_BOUND_FROM_PASS (test hook): the tiny bound only from this pass of the calibrator on -- a time-out in the middle of a solve
  int wr_ring[16] = {0};                // weight buffer read by pass (pass_seq & 15)
  int sync_timeouts = 0;                // flag hand-overs that ran into their bound (each one reported on stderr, the solve resumed with events)
  long long pass_seq = 0;               // passes enqueued (the value the flags carry)
  bool prev_pass_signals = false;       // the previous pass of this solve was enqueued with signalling kernels
  DBuf<long long> d_sync;
  bool jac_on_stream2 = true;           // the trial point's k_imu_jac beside the vision sweep (VICALIB_AMD_JAC_STREAM2=0: after it, main stream)
  bool serial_weights = false;          // false: IMU Jacobians + weight update on the second stream (VICALIB_AMD_OVERLAP_WEIGHTS=0: in line); was: VICALIB_AMD_OVERLAP_WEIGHTS=1 moves it to a second
                                        // stream under the Jacobian sweeps / chain solve (measured: the two latency-bound kernels then
                                        // share the CUs and the pass gets 4 % slower on cfg3)
  hipGraphExec_t pass_graph[2] = {nullptr, nullptr};   // one captured LM pass per weight-buffer parity (single process)
  bool use_graphs = false;      // measured slower on ROCm 7.2 (cfg2: 65 vs 62 us / pass, instantiation ~10 ms per stage): opt-in via VICALIB_AMD_GRAPHS=1
  hipError_t last_hip_error = hipSuccess;
  // ---- problem (host copy) ---------------------------------------------------------------
  std::vector<HostCam> cams;
  std::vector<HostFrame> frames;
  std::vector<int> o_frame, o_cam, o_pid;     // o_pid: index into the table of distinct target points
  std::vector<double> o_pc;
  PointTable pts;                             // exact-bit de-duplication of the p_w the caller passes, done once at AddObservation
  std::vector<signed char> o_removed;       // RemoveOutliers: 1 = no copy left (dropped), 2 = one copy fewer than vis_mult (kObsOneLess)
  long n_one_less = 0;
  bool obs_dirty = true;          // the observation set (or its multiplicity bits) changed since the tile layout was built
  int n_points_dev = 0;
  std::vector<double> imu_w, imu_a, imu_t;
  double imu_end_time = -1.0;
  double g_dir[2] = {0, 0}, time_offset = 0, biases[6] = {0, 0, 0, 0, 0, 0}, scale[6] = {1, 1, 1, 1, 1, 1};
  double gyro_sigma = 5.3088444e-5, accel_sigma = 0.001883649;   // types.h:34-35
  // ---- flags: Clear() defaults, vicalibrator.h:232-249 -----------------------------------
  bool fix_intrinsics = false, is_bias_active = false, is_scale_active = false, is_inertial_active = false,
       is_visual_active = true, rotation_only = true, optimize_time_offset = true, is_finished = false,
       gravity_initialized = false, outliers_removed = false;
  int pnp_its = 0; double pnp_tol = 0.0;     // PosePnPRansac(..., robust_3pt_its = 0, robust_3pt_tol = 0, ...) at vicalib-task.cc:323-325
  int max_iters = 200;                       // FLAGS_max_iters
  double function_tolerance = 1e-6;          // vicalibrator.h:149
  double gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;   // Ceres defaults
  bool calibrate_imu = true, remove_outliers = false;
  double outlier_threshold = 2.0;
  int vis_mult = 0, imu_mult = 0;
  // ---- results ----------------------------------------------------------------------------
  std::vector<double> cam_rmse;
  double mse = 0;
  std::atomic<unsigned> num_iterations{0};
  std::vector<IterRecord> trace;
  int stage = 0;
  long jac_sweeps = 0, res_sweeps = 0;
  // ---- threading --------------------------------------------------------------------------
  std::thread worker;
  std::atomic<bool> is_running{false}, should_run{false};
  std::mutex result_mutex;
  // ---- sharding ---------------------------------------------------------------------------
  int rank = 0, world = 1;
  bool force_shard_path = false;   // VICALIB_AMD_FORCE_SHARD_PATH=1: run the sharded code path (split kernels + callbacks) with one rank (test hook)
  bool sharded() const { return world > 1 || (force_shard_path && (allreduce || rccl_comm)); }
  DBuf<double> d_halo, d_sep_strip, d_gath;
  long global_first = 0, global_total = 0;     // this rank's frame range in the sharded problem (known after gather_shard_info)
  vc_allreduce_fn allreduce = nullptr;
  void* allreduce_ctx = nullptr;
  void* rccl_comm = nullptr;        // RCCL communicator (vc_set_shard_rccl: own; vc_set_shard_comm: borrowed): all-reduces go straight onto `stream`
  bool rccl_comm_owned = false;     // this calibrator created it and destroys it
  void drop_comm() {
    if (rccl_comm && rccl_comm_owned && g_rccl.CommDestroy) { (void)hipStreamSynchronize(stream); (void)g_rccl.CommDestroy(rccl_comm); }
    rccl_comm = nullptr; rccl_comm_owned = false;
  }
  long rccl_calls = 0;
  // ---- device -----------------------------------------------------------------------------
  bool device_dirty = true;      // host problem changed since the last upload
  DevView dv{};
  int cur = 0;
  DBuf<double2> d_uv; DBuf<unsigned short> d_pt; DBuf<double> d_points;
  DBuf<TileHdr> d_tile_hdr;
  DBuf<int> d_tile_frame, d_tile_cam, d_tile_off, d_frame_tile_off, d_frame_cam_tile, d_cam_model, d_cam_flags, d_cam_col0,
      d_col_cam, d_col_local, d_flags;
  DBuf<double> d_wgpart;
  int kpass = 0;                  // passes enqueued since init_ctrl (merged mode: selects the control record and flag parity)
  DBuf<double> d_pose[2], d_cam[2], d_G[2], d_tile_cost[2], d_Y, d_fr, d_fdiag, d_fscale2, d_part, d_Sbuf, d_sdiag,
      d_sscale2, d_slam, d_delta_s, d_fpart, d_scal, d_tmp, d_pose_init, d_cam_init, d_tile_trial, d_trace, d_part_total;
  DBuf<Ctrl> d_ctrl;
  DBuf<double> d_vel[2], d_imus[2], d_imu_t, d_imu_w, d_imu_a, d_frame_time, d_wsqrt[2], d_seg[2], d_seg_cost[2],
      d_cW, d_cdelta, d_ct0, d_cg, d_clam, d_cdiag, d_cscale2, d_vel_init, d_imus_init, d_rX[2], d_grp_part, d_wg_trial, d_wg_imu_trial,
      d_imu_delta_blk, d_imu_grav;
  DBuf<long long> d_cready;
  size_t wsqrt_frames = 0;       // number of frames the device weight_sqrt_ array was initialised for
  size_t imu_uploaded = 0; double imu_uploaded_last = 0.0;      // sample count / last time stamp of the device copy of the IMU samples
  int trace_cap = 0;
  struct Pinned { Ctrl up; Ctrl down; Ctrl dev; double trace[64 * kTraceCols]; unsigned long long progress; };      // dev / trace / progress: written by the device
  Pinned* pin = nullptr;        // page-locked staging (async copies without a bounce buffer)
  long nres_global_cached = -1; int nres_mult_cached[2] = {-1, -1};      // sharded: the all-reduced residual count and the multiplicities it was formed with
  long solve_epoch = 0, nres_epoch_cached = -1;     // ... and the public solve call it was formed in (bumped by every rank at the same entry points)
  int expected_passes = 8;       // passes the previous solve needed: size of the first batch of the next one (batched schedule)
  int feed_ahead = 1;            // passes kept queued beyond the last decision seen (grows when the host is found late)
  bool feed_passes = std::getenv("VICALIB_AMD_BATCHED") == nullptr;   // single process: feed passes against the device's progress word
  DBuf<unsigned char> d_mask;
  std::vector<int> h_tile_frame, h_tile_cam, h_tile_off, h_obs_index;   // h_obs_index: device corner -> host observation
  std::vector<int> cam_flags, cam_col0;

  ~vc_calibrator() {
    stop();
    drop_comm();
    drop_graphs();
    kt_free();
    if (stream2) (void)hipStreamDestroy(stream2);
    if (ev_state) (void)hipEventDestroy(ev_state);
    if (ev_pre) (void)hipEventDestroy(ev_pre);
    if (ev_weights) (void)hipEventDestroy(ev_weights);
    if (ev_imujac) (void)hipEventDestroy(ev_imujac);
    if (ev_reduced) (void)hipEventDestroy(ev_reduced);
    if (ev_back) (void)hipEventDestroy(ev_back);
    if (stream) (void)hipStreamDestroy(stream);
    if (pin) (void)hipHostFree(pin);
  }
  void stop() {
    should_run = false;
    if (worker.joinable()) worker.join();
  }

  // ---- layout of the shared (non-frame) parameters: SetupProblem constancy rules -------------
  int build_layout(std::vector<int>& col_cam, std::vector<int>& col_local) {
    const int C = (int)cams.size();
    cam_flags.assign(C, 0); cam_col0.assign(C, 0);
    int o = 0;
    col_cam.clear(); col_local.clear();
    for (int c = 0; c < C; ++c) {
      bool rf = true, tf = true;
      if (c == 0) {                               // vicalibrator.h:572-587
        if (!is_inertial_active) { rf = false; tf = false; } else { rf = true; tf = !rotation_only; }
      }
      int fl = 0;
      if (rf) fl |= kCamRotFree;
      if (tf) fl |= kCamTransFree;
      if (!fix_intrinsics) fl |= kCamKFree;       // :591-593
      cam_flags[c] = fl; cam_col0[c] = o;
      const int nc = cam_ncols(fl, cams[c].nk);
      for (int i = 0; i < nc; ++i) { col_cam.push_back(c); col_local.push_back(i); }
      o += nc;
    }
    for (int a = 0; a < 15; ++a) imu_param_col[a] = -1;
    if (imu_on()) {
      auto add = [&](int first, int n) { for (int i = 0; i < n; ++i) { imu_param_col[first + i] = o++; col_cam.push_back(-1); col_local.push_back(0); } };
      if (!rotation_only) add(0, 2);            // gravity: constant while rotation-only (vicalibrator.h:657-660, :986)
      if (is_bias_active) add(2, 6);            // :663-666, :990
      if (is_scale_active) add(8, 6);           // :668-671, :994
      if (optimize_time_offset) add(14, 1);     // :673-676
    }
    return o;
  }
  bool imu_on() const { return calibrate_imu && is_inertial_active; }
  int imu_param_col[15];

  // sum a small host vector over the ranks through the caller's device all-reduce
  int host_allreduce_sum(std::vector<double>& v) {
    if (!sharded()) return VC_OK;
    HIP_OK(d_halo.upload(v, stream));
    int rc = do_allreduce(d_halo.p, (int)v.size(), 0); if (rc) return rc;
    HIP_OK(hipMemcpyAsync(v.data(), d_halo.p, v.size() * 8, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    return VC_OK;
  }
  // every rank's first frame (pose, velocity, time) and frame count: slot r of the gathered table
  int gather_shard_info(std::vector<double>* table) {
    table->assign((size_t)world * 12, 0.0);
    if (!frames.empty()) {
      double* o = table->data() + (size_t)rank * 12;
      std::memcpy(o, frames[0].T, 56); std::memcpy(o + 7, frames[0].v, 24); o[10] = frames[0].time;
    }
    (*table)[(size_t)rank * 12 + 11] = (double)frames.size();
    int rc = host_allreduce_sum(*table); if (rc) return rc;
    global_first = 0; global_total = 0;
    for (int r = 0; r < world; ++r) { if (r < rank) global_first += (long)(*table)[(size_t)r * 12 + 11]; global_total += (long)(*table)[(size_t)r * 12 + 11]; }
    return VC_OK;
  }

  int upload() {
    RoctxRange rr("vicalib_amd: upload (SetupProblem of a stage)");
    HIP_OK(hipSetDevice(device));
    drop_graphs();
    pre_weights_fresh = false; pre_weights_pending = false;      // (the state is about to change under them)
    const bool up_timing = std::getenv("VICALIB_AMD_TIMING") != nullptr;
    const auto up_t0 = std::chrono::steady_clock::now();
    auto up_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - up_t0).count(); };
    double up_a = 0, up_b = 0, up_c = 0;
    const int Nown = (int)frames.size(), C = (int)cams.size();
    if (C > kMaxCams) return VC_ERR_UNSUPPORTED;
    // ---- frame-sharded IMU chain: this rank's first frame is a separator of the reduced system (rank > 0) and the
    // next rank's separator is kept here as a ghost frame (the IMU block that ends in it is ours)
    const bool shard_imu = world > 1 && imu_on();
    bool ghost = false;
    HostFrame ghost_frame{};
    if (shard_imu) {
      if (Nown < 2) return VC_ERR_BAD_ARG;
      std::vector<double> table;
      int rc = gather_shard_info(&table); if (rc) return rc;
      if (rank + 1 < world) {
        const double* o = table.data() + (size_t)(rank + 1) * 12;
        std::memcpy(ghost_frame.T, o, 56); std::memcpy(ghost_frame.v, o + 7, 24); ghost_frame.time = o[10];
        ghost = true;
      }
    }
    const int N = Nown + (ghost ? 1 : 0);
    auto frame_at = [&](int f) -> const HostFrame& { return f < Nown ? frames[f] : ghost_frame; };
    // ---- tiles: sort the active observations by (frame, camera) -- only when the observation set changed (the stage
    // machine re-uploads state and layout four times per calibration, the 10 ms sort / de-dup / 7 MB copy happen once)
    if (obs_dirty) {
      const size_t n_all = o_frame.size();
      std::vector<int> idx; idx.reserve(n_all);
      for (size_t i = 0; i < n_all; ++i) if (o_removed[i] != 1) idx.push_back((int)i);
      bool in_order = true;             // the usual caller adds frame by frame, camera by camera: nothing to sort then
      for (size_t k = 1; k < idx.size() && in_order; ++k) {
        const int a = idx[k - 1], b = idx[k];
        in_order = o_frame[a] != o_frame[b] ? o_frame[a] < o_frame[b] : o_cam[a] <= o_cam[b];
      }
      if (!in_order) std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) {
        return o_frame[a] != o_frame[b] ? o_frame[a] < o_frame[b] : o_cam[a] < o_cam[b]; });
      h_obs_index = idx;
      h_tile_frame.clear(); h_tile_cam.clear(); h_tile_off.clear();
      if (pts.size() > kObsPointMask + 1) return VC_ERR_TOO_MANY_POINTS;
      const size_t n_o = idx.size();
      if (n_o * sizeof(double2) > ((size_t)64 << 20)) {
        // (very large observation sets keep the streaming pageable copies: a page-locked image of their size costs more to allocate
        //  than it saves)
        std::vector<double2> uv(n_o);
        std::vector<unsigned short> pt(n_o);
        n_one_less = 0;
        for (size_t k = 0; k < n_o; ++k) {
          const int i = idx[k];
          if (k == 0 || o_frame[i] != o_frame[idx[k - 1]] || o_cam[i] != o_cam[idx[k - 1]]) {
            h_tile_frame.push_back(o_frame[i]); h_tile_cam.push_back(o_cam[i]); h_tile_off.push_back((int)k);
          }
          pt[k] = (unsigned short)(o_pid[i] | (o_removed[i] == 2 ? kObsOneLess : 0));
          if (o_removed[i] == 2) ++n_one_less;
          uv[k] = make_double2(o_pc[2 * (size_t)i], o_pc[2 * (size_t)i + 1]);
        }
        h_tile_off.push_back((int)n_o);
        n_points_dev = pts.size();
        HIP_OK(d_uv.upload(uv, stream)); HIP_OK(d_pt.upload(pt, stream)); HIP_OK(d_points.upload(pts.xyz, stream));
        HIP_OK(d_tile_frame.upload(h_tile_frame, stream)); HIP_OK(d_tile_cam.upload(h_tile_cam, stream));
        HIP_OK(d_tile_off.upload(h_tile_off, stream));
        HIP_OK(d_mask.alloc(std::max<size_t>(n_o, 1)));
        HIP_OK(hipStreamSynchronize(stream));        // the staging vectors go out of scope
      } else {
      // detections and point indices are written straight into the page-locked staging image (no pageable vector in between: the 6 MB
      // copy of cfg3's detections was 1.5 ms of the first stage's upload)
      pack.begin();
      const size_t pt_bytes = ((n_o * sizeof(unsigned short) + 3) / 4) * 4;
      if (pack.reserve(n_o * sizeof(double2) + pt_bytes + pts.xyz.size() * 8 + (h_tile_frame.capacity() + n_o / 8 + 64) * 16 + 4096) != hipSuccess) return VC_ERR_NO_DEVICE;
      size_t uv_off = 0, pt_off = 0;
      double2* uv = (double2*)pack.slot(n_o * sizeof(double2), &uv_off);
      unsigned short* pt = (unsigned short*)pack.slot(pt_bytes, &pt_off);
      if (!uv || !pt) return VC_ERR_NO_DEVICE;
      if (pt_bytes > n_o * sizeof(unsigned short)) pt[n_o] = 0;      // (padding word)
      n_one_less = 0;
      for (size_t k = 0; k < idx.size(); ++k) {
        const int i = idx[k];
        if (k == 0 || o_frame[i] != o_frame[idx[k - 1]] || o_cam[i] != o_cam[idx[k - 1]]) {
          h_tile_frame.push_back(o_frame[i]); h_tile_cam.push_back(o_cam[i]); h_tile_off.push_back((int)k);
        }
        pt[k] = (unsigned short)(o_pid[i] | (o_removed[i] == 2 ? kObsOneLess : 0));
        if (o_removed[i] == 2) ++n_one_less;
        uv[k] = make_double2(o_pc[2 * (size_t)i], o_pc[2 * (size_t)i + 1]);
      }
      h_tile_off.push_back((int)idx.size());
      n_points_dev = pts.size();
      HIP_OK(d_uv.alloc(std::max<size_t>(n_o, 1))); HIP_OK(d_pt.alloc(pt_bytes / sizeof(unsigned short) + 2));
      pack.seg(d_uv.p, uv_off, n_o * sizeof(double2)); pack.seg(d_pt.p, pt_off, pt_bytes);
      HIP_OK(pack.add(d_points, pts.xyz));
      HIP_OK(pack.add(d_tile_frame, h_tile_frame)); HIP_OK(pack.add(d_tile_cam, h_tile_cam));
      HIP_OK(pack.add(d_tile_off, h_tile_off));
      HIP_OK(d_mask.alloc(std::max<size_t>(idx.size(), 1)));
      HIP_OK(pack.flush(stream));
      HIP_OK(hipStreamSynchronize(stream));        // the staging image is re-used below
      }
      obs_dirty = false;
    }
    up_a = up_ms();
    const size_t n_active = h_obs_index.size();
    const int T = (int)h_tile_frame.size();
    std::vector<int> frame_tile_off(N + 1, T), frame_cam_tile((size_t)N * std::max(C, 1), -1);
    {
      int t = 0;
      for (int f = 0; f <= N; ++f) { while (t < T && h_tile_frame[t] < f) ++t; frame_tile_off[f] = t; }
      for (int t2 = 0; t2 < T; ++t2) frame_cam_tile[(size_t)h_tile_frame[t2] * C + h_tile_cam[t2]] = t2;
    }
    std::vector<int> col_cam, col_local, cam_model(C);
    const int D0 = build_layout(col_cam, col_local);
    const int D = D0 + (shard_imu ? 9 * (world - 1) : 0);          // + one 9-column separator per shard boundary
    col_cam.resize(D, -1); col_local.resize(D, 0);
    // the widest column layout this problem can reach (every camera and IMU parameter free): the stage machine only widens the
    // layout, and growing a multi-megabyte buffer is a free + malloc of a few hundred microseconds -- the big ones get their
    // final capacity at the first upload
    int Dmax = (imu_on() ? 15 : 0) + (shard_imu ? 9 * (world - 1) : 0);
    for (int c = 0; c < C; ++c) Dmax += 6 + cams[c].nk;
    Dmax = std::max(Dmax, D);
    // the reduced solve lives in LDS (packed lower triangle + 12 KB of staging), the chain Gram handles 12 column tiles
    if (((size_t)(D + 1) * (D + 2) / 2 + 3 * (D + 1) + 528) * sizeof(double) + 13 * 1024 > 160 * 1024 || D + 1 > 12 * 16) return VC_ERR_UNSUPPORTED;
    // vision-only passes: k_frame_schur keeps at most 36 column-tile pairs (8 column tiles: D <= 127; eight 16-column cameras
    // with the first one's extrinsics fixed are 122)
    if (!imu_on() && D + 1 > 8 * 16) return VC_ERR_UNSUPPORTED;
    for (int c = 0; c < C; ++c) cam_model[c] = cams[c].model;
    up_b = up_ms();
    // ---- upload ---------------------------------------------------------------------------------
    // Everything small goes through ONE page-locked staging image, one copy and one scatter kernel (Packer above; round 3: ~30
    // pageable copies and ~8 fills per stage, 4.1 of the 19 ms of a complete cfg3 calibration).  Arrays above 4 MB keep their own copy.
    pack.begin();
    auto up = [&](auto& d, const auto& h) -> hipError_t {
      if (h.size() * sizeof(h[0]) > (size_t)4 << 20) return d.upload(h, stream);
      return pack.add(d, h);
    };
    // the same array into a second / third buffer (state double buffer, initial-state copy)
    auto up_also = [&](auto& d, const auto& h) -> hipError_t {
      if (h.size() * sizeof(h[0]) > (size_t)4 << 20) return d.upload(h, stream);
      return pack.also(d, h.size());
    };
    {   // tile headers: depend on the tile layout and on this stage's column layout
      std::vector<TileHdr> hdr((size_t)T);
      for (int t = 0; t < T; ++t) {
        TileHdr& h = hdr[(size_t)t];
        const int f = h_tile_frame[t];
        h.frame = f; h.cam = h_tile_cam[t]; h.t0 = frame_tile_off[f]; h.nt = frame_tile_off[f + 1] - frame_tile_off[f];
        h.off = h_tile_off[t]; h.cnt = h_tile_off[t + 1] - h_tile_off[t]; h.model = cams[h.cam].model; h.pad = 0;
        h.col0 = 0; h.ncols = 0;
        for (int k = 0; k < h.nt && k < kMaxCams; ++k) {
          const int c = h_tile_cam[h.t0 + k];
          h.col0 |= (unsigned long long)(cam_col0[c] & 0xff) << (8 * k);
          h.ncols |= (unsigned long long)(cam_ncols(cam_flags[c], cams[c].nk) & 0xff) << (8 * k);
        }
      }
      HIP_OK(up(d_tile_hdr, hdr));
    }
    HIP_OK(up(d_frame_tile_off, frame_tile_off));
    HIP_OK(up(d_frame_cam_tile, frame_cam_tile)); HIP_OK(up(d_cam_model, cam_model));
    HIP_OK(up(d_cam_flags, cam_flags)); HIP_OK(up(d_cam_col0, cam_col0));
    HIP_OK(up(d_col_cam, col_cam)); HIP_OK(up(d_col_local, col_local));
    std::vector<double> poses((size_t)N * kPoseStride, 0.0), camrec((size_t)C * kCamStride, 0.0);
    for (int f = 0; f < N; ++f) std::memcpy(&poses[(size_t)f * kPoseStride], frame_at(f).T, 56);
    for (int c = 0; c < C; ++c) {
      std::memcpy(&camrec[(size_t)c * kCamStride], cams[c].T_ck, 56);
      std::memcpy(&camrec[(size_t)c * kCamStride + kCamK], cams[c].K, cams[c].nk * 8);
    }
    cur = 0;
    HIP_OK(up(d_pose[0], poses)); HIP_OK(up_also(d_pose[1], poses)); HIP_OK(up_also(d_pose_init, poses));
    HIP_OK(up(d_cam[0], camrec)); HIP_OK(up_also(d_cam[1], camrec)); HIP_OK(up_also(d_cam_init, camrec));
    // one wavefront per frame, 4 frames per group: up to 2048 chunks (= partial sums) before chunks grow
    int chunk_frames = std::max(4, (((N + 2047) / 2048) + 3) / 4 * 4);
    // wide borders: a chunk's partial record is D^2 doubles -- written once and read once per pass; keep all of them under ~64 MB
    // (8 cameras, 6250 frames per rank: 197 MB at 4 frames per chunk, k_part_sum 47 -> 19 us at 16)
    while (chunk_frames < 16 && (double)((N + chunk_frames - 1) / chunk_frames) * ((double)D * D + D + (C + 1) * kGStride) * 8.0 > 64e6) chunk_frames *= 2;
    { const char* e = std::getenv("VICALIB_AMD_CHUNK_FRAMES"); if (e && std::atoi(e) >= 4) chunk_frames = std::atoi(e) / 4 * 4; }      // (A/B hook)
    const int n_chunks = std::max(1, (N + chunk_frames - 1) / chunk_frames);
    const int part_stride = D * D + D + C * kGStride + (imu_on() ? kGStride : 0) + 2;     // ... + [x2 of observation-less frames, chunk cost] (vision path)
    for (int b = 0; b < 2; ++b) {
      HIP_OK(d_G[b].alloc((size_t)std::max(T, 1) * kGPack)); HIP_OK(d_tile_cost[b].alloc(std::max(T, 1)));
      HIP_OK(hipMemsetAsync(d_G[b].p, 0, (size_t)std::max(T, 1) * kGPack * sizeof(double), stream));   // sub-blocks a model never writes stay 0
    } HIP_OK(d_tile_trial.alloc((size_t)std::max(T, 1) * 2));
    HIP_OK(d_Y.alloc((size_t)std::max(T, 1) * kYStride)); HIP_OK(d_fr.alloc((size_t)std::max(N, 1) * kFrStride));
    HIP_OK(d_fdiag.alloc((size_t)std::max(N, 1) * 6)); HIP_OK(d_fscale2.alloc((size_t)std::max(N, 1) * 6));
    HIP_OK(d_part.alloc((size_t)(n_chunks + 1) * ((size_t)Dmax * Dmax + Dmax + C * kGStride + kGStride + 2)));      // (+ 1: the chain's top level, early Gram)
    HIP_OK(d_part.alloc((size_t)(n_chunks + 1) * part_stride)); HIP_OK(d_part_total.alloc((size_t)part_stride)); HIP_OK(d_Sbuf.alloc((size_t)D * D + 3 * D + 2));
    HIP_OK(d_sdiag.alloc(D)); HIP_OK(d_sscale2.alloc(D)); HIP_OK(d_slam.alloc(D)); HIP_OK(d_delta_s.alloc(D));
    HIP_OK(d_fpart.alloc((size_t)std::max(N, 1) * kNumScal));
    trace_cap = max_iters + 8; HIP_OK(d_trace.alloc((size_t)trace_cap * kTraceCols)); HIP_OK(d_ctrl.alloc(2));
    HIP_OK(hipMemsetAsync(d_part.p, 0, (size_t)(n_chunks + 1) * part_stride * sizeof(double), stream));
    pack.zero(d_fpart.p, (size_t)std::max(N, 1) * kNumScal * sizeof(double));
    HIP_OK(d_scal.alloc(2 * kNumScal)); HIP_OK(d_flags.alloc(8));
    HIP_OK(d_wgpart.alloc((size_t)std::max(1, (T + 3) / 4) * kNumScal));
    pack.zero(d_scal.p, 2 * kNumScal * sizeof(double));
    HIP_OK(d_tmp.alloc(128));       // [0,16) per-camera sums, [32,40) outlier thresholds, [64,96) profiling stamps
    pack.zero(d_tmp.p, 128 * sizeof(double));
    pack.zero(d_flags.p, 8 * sizeof(int));
    pack.zero(d_ctrl.p, 2 * sizeof(Ctrl));
    pack.zero(d_delta_s.p, std::max(D, 1) * sizeof(double));
    for (int c = 0; c < kMaxCams; ++c) { dv.cd[c].model = 0; dv.cd[c].flags = 0; dv.cd[c].col0 = 0; dv.cd[c].ncols = 0; }
    for (int c = 0; c < C; ++c) { dv.cd[c].model = cams[c].model; dv.cd[c].flags = cam_flags[c]; dv.cd[c].col0 = cam_col0[c]; dv.cd[c].ncols = cam_ncols(cam_flags[c], cams[c].nk); }
    dv.n_frames = N; dv.n_cams = C; dv.n_tiles = T; dv.n_points = n_points_dev; dv.D = D;
    dv.n_chunks = n_chunks; dv.n_part = n_chunks; dv.chunk_frames = chunk_frames; dv.n_obs = (long long)n_active;
    dv.obs_uv = d_uv.p; dv.obs_pt = d_pt.p; dv.points = d_points.p;
    dv.tile_hdr = d_tile_hdr.p;
    dv.tile_frame = d_tile_frame.p; dv.tile_cam = d_tile_cam.p; dv.tile_off = d_tile_off.p;
    dv.frame_tile_off = d_frame_tile_off.p; dv.frame_cam_tile = d_frame_cam_tile.p;
    dv.cam_model = d_cam_model.p; dv.cam_flags = d_cam_flags.p; dv.cam_col0 = d_cam_col0.p;
    dv.col_cam = d_col_cam.p; dv.col_local = d_col_local.p;
    dv.poses[0] = d_pose[0].p; dv.poses[1] = d_pose[1].p; dv.cams[0] = d_cam[0].p; dv.cams[1] = d_cam[1].p;
    for (int b = 0; b < 2; ++b) { dv.Gb[b] = d_G[b].p; dv.tile_costb[b] = d_tile_cost[b].p; }
    dv.fused = 1;
 dv.tile_trial = d_tile_trial.p; dv.Y = d_Y.p; dv.fr = d_fr.p;
    dv.fdiag = d_fdiag.p; dv.fscale2 = d_fscale2.p; dv.part = d_part.p; dv.part_total = d_part_total.p; dv.Sbuf = d_Sbuf.p;
    dv.sdiag = d_sdiag.p; dv.sscale2 = d_sscale2.p; dv.slam = d_slam.p; dv.delta_s = d_delta_s.p;
    dv.fpart = d_fpart.p; dv.scal = d_scal.p; dv.flags = d_flags.p;
    dv.pre_backsub = (T > 2048) ? 1 : 0;
    { const char* e = std::getenv("VICALIB_AMD_PRE_BACKSUB"); if (e && (e[0] == '0' || e[0] == '1')) dv.pre_backsub = e[0] - '0'; }   // test hook     // 1024 SIMDs x 2 resident waves: beyond that the per-tile repeat of the back-substitution is pure cost
    {
      // bottom-level groups of the chain elimination (launch_chain_solve_*: groups of 8 while more than 7 frames are active)
      const int cm = chain_group_size();
      const int groups = (N > cm - 1) ? (N - 1) / cm + 1 : 1;
      HIP_OK(d_grp_part.alloc((size_t)groups * kNumScal)); HIP_OK(d_wg_trial.alloc((size_t)std::max(1, (T + 3) / 4)));
      HIP_OK(d_wg_imu_trial.alloc((size_t)std::max(1, (N + 6) / 8)));
      dv.grp_part = d_grp_part.p; dv.wg_trial = d_wg_trial.p; dv.wg_imu_trial = d_wg_imu_trial.p; dv.n_chain_groups = groups;
    }
    dv.wgpart = d_wgpart.p; dv.merged = 0; dv.par = 0; dv.ctrl_prev = d_ctrl.p + 1;
    dv.part_stride = part_stride; dv.ctrl = d_ctrl.p; dv.trace = d_trace.p; dv.dbg = (long long*)(d_tmp.p + 64);
    // ---- inertial terms ------------------------------------------------------------------------------
    std::vector<double> vels((size_t)std::max(N, 1) * 4, 0.0), imus(16, 0.0), ftime(std::max(N, 1), 0.0);
    for (int f = 0; f < N; ++f) { std::memcpy(&vels[(size_t)f * 4], frame_at(f).v, 24); ftime[f] = frame_at(f).time; }
    imus[0] = g_dir[0]; imus[1] = g_dir[1];
    for (int i = 0; i < 6; ++i) { imus[2 + i] = biases[i]; imus[8 + i] = scale[i]; }
    imus[14] = time_offset;
    HIP_OK(up(d_vel[0], vels)); HIP_OK(up_also(d_vel[1], vels)); HIP_OK(up_also(d_vel_init, vels));
    HIP_OK(up(d_imus[0], imus)); HIP_OK(up_also(d_imus[1], imus)); HIP_OK(up_also(d_imus_init, imus));
    HIP_OK(up(d_frame_time, ftime));
    dv.imu_on = imu_on() ? 1 : 0; dv.rotation_only = rotation_only ? 1 : 0;
    dv.weights_on = (is_inertial_active && !rotation_only) ? 1 : 0;
    dv.n_imu = (int)imu_t.size();
    dv.imu_avg_dt = imu_average_dt(imu_t.data(), (int)imu_t.size());
    dv.gyro_sigma = gyro_sigma; dv.accel_sigma = accel_sigma;
    for (int a = 0; a < 15; ++a) dv.imu_param_col[a] = imu_param_col[a];
    dv.ldw = (((D + 1 + 15) / 16) * 16 % 32 == 0) ? ((D + 1 + 15) / 16) * 16 + 16 : ((D + 1 + 15) / 16) * 16;
    dv.ldx = dv.ldw + 32;
    {
      // early Gram (vc_device.h): narrow reduced systems of a single process; a function of the problem only, never of the hand-over
      // mode -- a solve resumed with events after a flag time-out must repeat the withheld passes with the same arithmetic
      static const bool early_env = [] { const char* e = std::getenv("VICALIB_AMD_EARLY_GRAM"); return !(e && e[0] == '0'); }();
      // (where it pays: the top level's one group must outlast the Gram sums beside it -- at 6250 frames x 8 cameras, D = 115, the top
      //  level is two frames and the sums take 50 us: 0.906 -> 0.938 ms per pass with them in its launch; at 2500 frames, D = 67: -4.5 us)
      const bool narrow = D + 1 + 27 <= 128;      // at most two image columns per lane
      dv.gram_top_stride = (early_env && dv.imu_on && N >= 1 && N <= 4096 && narrow) ? chain_top_stride(N) : 0;
      // the top level's own frames: added by k_reduced (single process, narrow system) or a partial record of their own
      top_gram_launch = dv.gram_top_stride > 0 && !(D <= kSmallD && !sharded());
      dv.n_part = dv.n_chunks + (top_gram_launch ? 1 : 0);
    }
    dv.pin_first = (shard_imu && rank > 0) ? 1 : 0; dv.pin_last = ghost ? 1 : 0;
    dv.sep_col0 = D0 + 9 * (rank - 1); dv.sep_col1 = D0 + 9 * rank;
    HIP_OK(d_sep_strip.alloc((size_t)2 * 9 * dv.ldw)); dv.sep_strip = d_sep_strip.p;
    HIP_OK(d_gath.alloc((size_t)world * kNumScal)); dv.gath = d_gath.p; dv.rank = rank; dv.world = world;
    if (dv.imu_on) {
      if (imu_uploaded != imu_t.size() || imu_uploaded_last != (imu_t.empty() ? 0.0 : imu_t.back())) {      // the samples do not change from stage to stage
        HIP_OK(d_imu_t.upload(imu_t, stream)); HIP_OK(d_imu_w.upload(imu_w, stream)); HIP_OK(d_imu_a.upload(imu_a, stream));
        imu_uploaded = imu_t.size(); imu_uploaded_last = imu_t.empty() ? 0.0 : imu_t.back();
      }
      const size_t ns = (size_t)std::max(N - 1, 1);
      if (wsqrt_frames != (size_t)N) {          // initial weight 500 * I (vicalibrator.h:616); later stages keep the current weights
        std::vector<double> w(ns * 81, 0.0);
        for (size_t k = 0; k < ns; ++k) for (int i = 0; i < 9; ++i) w[k * 81 + i * 10] = 500.0;
        HIP_OK(d_wsqrt[0].upload(w, stream)); HIP_OK(d_wsqrt[1].upload(w, stream)); wsqrt_frames = (size_t)N; wcur = 0;
        HIP_OK(hipStreamSynchronize(stream));
      }
      for (int b = 0; b < 2; ++b) { HIP_OK(d_seg[b].alloc(ns * kSegStride)); HIP_OK(d_seg_cost[b].alloc(ns)); }
      HIP_OK(d_imu_delta_blk.alloc(ns * kBlockDeltaStride)); HIP_OK(d_imu_grav.alloc(32));
      const size_t nf = (size_t)std::max(N, 1);
      {
        const int ldw_max = (((Dmax + 1 + 15) / 16) * 16 % 32 == 0) ? ((Dmax + 1 + 15) / 16) * 16 + 16 : ((Dmax + 1 + 15) / 16) * 16;
        HIP_OK(d_cW.alloc(nf * 9 * (ldw_max + 32)));
        for (int b = 0; b < 2; ++b) HIP_OK(d_rX[b].alloc((nf / std::min(chain_group_size(), chain_group_size_upper()) + 2) * 9 * (ldw_max + 32)));
      }
      HIP_OK(d_cW.alloc(nf * 9 * dv.ldx)); HIP_OK(d_cdelta.alloc(nf * 9)); HIP_OK(d_ct0.alloc(nf * 9)); HIP_OK(d_cg.alloc(nf * 9)); HIP_OK(d_clam.alloc(nf * 9));
      HIP_OK(d_cdiag.alloc(nf * 9)); HIP_OK(d_cscale2.alloc(nf * 9));
      HIP_OK(d_cready.alloc(nf)); pack.zero(d_cready.p, nf * sizeof(long long));
      for (int b = 0; b < 2; ++b) HIP_OK(d_rX[b].alloc((nf / std::min(chain_group_size(), chain_group_size_upper()) + 2) * 9 * dv.ldx));
    }
    dv.imu_t = d_imu_t.p; dv.imu_w = d_imu_w.p; dv.imu_a = d_imu_a.p; dv.frame_time = d_frame_time.p;
    dv.vel[0] = d_vel[0].p; dv.vel[1] = d_vel[1].p; dv.imus[0] = d_imus[0].p; dv.imus[1] = d_imus[1].p;
    dv.wsqrtb[0] = d_wsqrt[0].p; dv.wsqrtb[1] = d_wsqrt[1].p;
    dv.imu_delta_blk = d_imu_delta_blk.p; dv.imu_grav = d_imu_grav.p;
    for (int b = 0; b < 2; ++b) { dv.segb[b] = d_seg[b].p; dv.seg_costb[b] = d_seg_cost[b].p; }
    dv.cW = d_cW.p; dv.cdelta = d_cdelta.p; dv.ct0 = d_ct0.p; dv.cg = d_cg.p;
    dv.clam = d_clam.p; dv.cdiag = d_cdiag.p; dv.cscale2 = d_cscale2.p; dv.cready = d_cready.p;
    for (int b = 0; b < 2; ++b) dv.rX[b] = d_rX[b].p;
    HIP_OK(pack.flush(stream));
    up_c = up_ms();
    HIP_OK(hipStreamSynchronize(stream));   // the staging vectors above go out of scope
    if (up_timing) std::fprintf(stderr, "[vicalib_amd]   upload: observations %.3f, layout %.3f, copies + allocations %.3f, drain %.3f ms\n", up_a, up_b - up_a, up_c - up_b, up_ms() - up_c);
    device_dirty = false;
    return VC_OK;
  }
  // device-to-device reset of the state to what upload() put there (benchmark restarts)
  int reset_state() {
    cur = 0;
    launch_reset_state(dv, d_pose_init.p, d_cam_init.p, d_vel_init.p, d_imus_init.p, stream);
    return VC_OK;
  }
  // copy the accepted device state back into the host problem
  int download_state() {
    const int N = (int)frames.size(), C = (int)cams.size();
    std::vector<double> poses((size_t)N * kPoseStride), camrec((size_t)C * kCamStride);
    if (N) HIP_OK(hipMemcpyAsync(poses.data(), dv.poses[cur], poses.size() * 8, hipMemcpyDeviceToHost, stream));
    if (C) HIP_OK(hipMemcpyAsync(camrec.data(), dv.cams[cur], camrec.size() * 8, hipMemcpyDeviceToHost, stream));
    std::vector<double> vels((size_t)std::max(N, 1) * 4, 0.0), imus(16, 0.0);
    if (N) HIP_OK(hipMemcpyAsync(vels.data(), dv.vel[cur], (size_t)N * 4 * 8, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(imus.data(), dv.imus[cur], 16 * 8, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    std::lock_guard<std::mutex> lk(result_mutex);
    for (int f = 0; f < N; ++f) { std::memcpy(frames[f].T, &poses[(size_t)f * kPoseStride], 56); std::memcpy(frames[f].v, &vels[(size_t)f * 4], 24); }
    g_dir[0] = imus[0]; g_dir[1] = imus[1];
    for (int i = 0; i < 6; ++i) { biases[i] = imus[2 + i]; scale[i] = imus[8 + i]; }
    time_offset = imus[14];
    for (int c = 0; c < C; ++c) {
      std::memcpy(cams[c].T_ck, &camrec[(size_t)c * kCamStride], 56);
      std::memcpy(cams[c].K, &camrec[(size_t)c * kCamStride + kCamK], cams[c].nk * 8);
    }
    return VC_OK;
  }

  void drop_graphs() {
    for (int i = 0; i < 2; ++i) if (pass_graph[i]) { (void)hipGraphExecDestroy(pass_graph[i]); pass_graph[i] = nullptr; }
  }
  // A pass is a fixed sequence of launches (up to ~45 with the IMU chain): captured once per upload and replayed, the host
  // pays one graph launch per pass instead of one call per kernel.  Sharded runs keep direct launches (host callbacks).
  int launch_pass_graph() {
    const bool flips = dv.imu_on && dv.weights_on;
    const int par = flips ? wcur : 0;
    if (!pass_graph[par]) {
      hipGraph_t g = nullptr;
      const int w0 = wcur;
      if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { use_graphs = false; return enqueue_pass(false); }
      const int rc = enqueue_pass(false);
      const hipError_t e = hipStreamEndCapture(stream, &g);
      wcur = w0;
      if (rc != VC_OK || e != hipSuccess || !g || hipGraphInstantiate(&pass_graph[par], g, nullptr, nullptr, 0) != hipSuccess) {
        if (g) (void)hipGraphDestroy(g);
        pass_graph[par] = nullptr; use_graphs = false;
        (void)hipGetLastError();
        return enqueue_pass(false);
      }
      (void)hipGraphDestroy(g);
    }
    HIP_OK(hipGraphLaunch(pass_graph[par], stream));
    if (flips) wcur = 1 - wcur;
    return VC_OK;
  }

  // ---- in-loop kernel timing (vc_set_kernel_timing): every launch group of a pass bracketed by a pair of events on the
  // calibrator's stream; durations are read back after the solve.  Off by default (an event record costs ~1 us of stream time).
  bool ktime_on = false;
  std::vector<hipEvent_t> kt_ev;            // pool: 2 per bracket
  std::vector<int> kt_label;                // label of bracket i
  size_t kt_used = 0;
  std::vector<std::string> kt_names;
  std::vector<double> kt_total_ms; std::vector<long> kt_count;
  int kt_label_id(const char* name) {
    for (size_t i = 0; i < kt_names.size(); ++i) if (kt_names[i] == name) return (int)i;
    kt_names.push_back(name); kt_total_ms.push_back(0.0); kt_count.push_back(0);
    return (int)kt_names.size() - 1;
  }
  hipStream_t kt_stream = nullptr;
  void kt_begin(const char* name, hipStream_t strm = nullptr) {
    kt_stream = strm ? strm : stream;
    if (kt_ev.size() < 2 * (kt_used + 1)) {
      hipEvent_t a = nullptr, b = nullptr;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { ktime_on = false; return; }
      kt_ev.push_back(a); kt_ev.push_back(b);
    }
    kt_label.resize(kt_used + 1); kt_label[kt_used] = kt_label_id(name);
    (void)hipEventRecord(kt_ev[2 * kt_used], kt_stream);
  }
  void kt_end() { (void)hipEventRecord(kt_ev[2 * kt_used + 1], kt_stream); ++kt_used; }
  void kt_collect() {          // after a stream synchronisation
    for (size_t i = 0; i < kt_used; ++i) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, kt_ev[2 * i], kt_ev[2 * i + 1]) == hipSuccess) { kt_total_ms[kt_label[i]] += ms; kt_count[kt_label[i]] += 1; }
    }
    kt_used = 0;
  }
  void kt_free() { for (hipEvent_t e : kt_ev) (void)hipEventDestroy(e); kt_ev.clear(); kt_used = 0; }
#define KT(name, call) do { if (ktime_on) kt_begin(name); call; if (ktime_on) kt_end(); } while (0)
#define KT2(name, call) do { if (ktime_on) kt_begin(name, stream2); call; if (ktime_on) kt_end(); } while (0)

  // ---- one pass of the device pipeline (all asynchronous; the decision is taken on the device) ------
  int do_allreduce(double* p, int n, int op) {
    if (sharded() && rccl_comm) {
      ++rccl_calls;
      if (g_rccl.AllReduce(p, p, (size_t)n, kNcclDouble, op == 1 ? kNcclMax : kNcclSum, rccl_comm, stream) != 0) return VC_ERR_NO_DEVICE;
    } else if (sharded() && allreduce) { if (allreduce(allreduce_ctx, p, n, op) != 0) return VC_ERR_NO_DEVICE; }
    return VC_OK;
  }
  // first_pass: the pass right after init_ctrl (the only one that needs k_reproj_jac when k_trial carries the sweep)
  // events_only: a stand-alone pass outside a solve (parity hooks, timing): nobody would resume it after a flag time-out
  int enqueue_pass(bool first_pass = true, bool events_only = false) {
    RoctxRange rr(first_pass ? "vicalib_amd: LM pass (first of a solve: + linearisation)" : "vicalib_amd: LM pass");
    const int D = dv.D;
    dv.merged = 0; dv.par = 0; dv.ctrl = d_ctrl.p; dv.ctrl_prev = d_ctrl.p + 1;
    if (dv.imu_on) {
      // UpdateImuWeights of the iteration callback (vicalibrator.h:691): linearise with the current weights, evaluate the
      // trial point with the updated ones.  The pass is a small graph over two streams: the weight update (which only needs the
      // accepted state and writes the other weight buffer) and the interval / block deltas of the trial point (which only need
      // the trial IMU parameters) run on the second, low-priority stream next to the chain solve -- every one of these kernels
      // is a few hundred latency-bound wavefronts, far from filling the chip on its own.  Everything on the critical path --
      // chain, both trial sweeps, decision -- stays on the main stream: kernels of one stream follow each other without a gap,
      // an event hand-over costs 5-13 us.
      // (the first pass of a solve: UpdateImuWeights() has just run on this very state (solve_once) -- the pass's own update would write the
      //  same numbers into the other buffer, 30 us on the second stream beside the first linearisation and the bottom chain level: the pass
      //  evaluates its trial point with the buffer it linearises with, and the buffers do not swap)
      const bool upd = dv.weights_on != 0 && !(first_pass && pre_weights_fresh && !events_only);      // (a stand-alone pass always updates)
      if (first_pass) pre_weights_fresh = false;
      // (sharded solves: flags when every rank has a device of its own -- vc_set_shard_rccl with more than one rank, or
      //  VICALIB_AMD_SHARD_FLAG_SYNC=1; the one-GPU gloo tests keep the events: several processes' waiting kernels would burn each
      //  other's time slices.  A time-out is lossless there too: the mark travels with the step scalars' all-reduce, all ranks resume)
      const bool fs = flag_sync && !serial_weights && (!sharded() || shard_flag_sync) && !use_graphs && !events_only;      // (a captured pass has fixed arguments and needs the events to fork the capture)
      ++pass_seq;
      dv.pass_id = pass_seq;
      wr_ring[pass_seq & 15] = wcur;              // (what a resume after a flag time-out restores: the weight buffer this pass reads)
      dv.sync_flags = d_sync.p; dv.sync_seq = fs ? pass_seq : 0; dv.final_wait = 0; dv.block_wait = 0; dv.sync_bound = (pass_seq >= sync_bound_from_pass) ? sync_bound : 400000;
      const bool fs_trial = fs && jac_on_stream2 && dv.n_tiles > 0;      // (no tiles: no trial sweep to publish the back-substitution's end)
      dv.final_wait = fs_trial ? pass_seq : 0;      // (k_imu_jac(trial) and k_final both look at it)
      dv.block_wait = fs_trial ? pass_seq : 0;      // (k_imu_block(trial))
      // The Jacobian sweeps at the head of the pass only run when the control record asks for a linearisation: the first pass
      // of a solve.  Afterwards the trial point is evaluated by the same sweeps in trial mode (below), which leave the next
      // linearisation behind if the step is accepted; after a rejected step the old one is still in place.
      if (!serial_weights) {
        bool block_done = false;
        if (first_pass && pre_weights_pending) {
          // the weight update that precedes a solve (solve_once) is still running on the main stream: the block deltas need the IMU
          // parameters only, not the weights -- they start from the event recorded ahead of it (35 us less per solve at cfg3)
          HIP_OK(hipStreamWaitEvent(stream2, ev_pre, 0));
          KT2("k_imu_block", launch_imu_delta(dv, stream2, 0));
          block_done = true;
        }
        pre_weights_pending = false;
        if (fs && !first_pass && prev_pass_signals) launch_wait_flag(dv, 0, pass_seq - 1, stream2);      // the previous pass's k_final
        else {
          HIP_OK(hipEventRecord(ev_state, stream));
          // the vision linearisation of a first pass depends on nothing the second stream does: it goes out before that stream's
          // launches (five runtime calls: the main stream would sit idle behind the preceding weight update for as long as they take)
          if (first_pass) KT("k_reproj_jac", launch_reproj_jac(dv, stream, 0));
          HIP_OK(hipStreamWaitEvent(stream2, ev_state, 0));
        }
        if (first_pass) {
          if (!block_done) KT2("k_imu_block", launch_imu_delta(dv, stream2, 0));
          KT2("k_imu_jac", launch_imu_jac(dv, wcur, stream2, 0));
          HIP_OK(hipEventRecord(ev_imujac, stream2));       // ahead of the weight update: the chain does not read the weights
        }
        // flag hand-overs: the weight update (500 wavefronts that take a SIMD's whole register file each) starts behind the bottom level of
        // the chain elimination, whose two-sided form needs the chip to itself; it is not needed before k_imu_jac(trial)
        if (upd && fs && !first_pass && weights_behind_l0 && chain_forward_launches(dv) >= 2) launch_wait_flag(dv, 7, pass_seq, stream2);
        if (upd) KT2("k_imu_weights", launch_imu_weights(dv, wcur, stream2));
        if (first_pass) HIP_OK(hipStreamWaitEvent(stream, ev_imujac, 0));
      } else {
        if (upd) KT("k_imu_weights", launch_imu_weights(dv, wcur, stream));
        if (first_pass) {
          KT("k_reproj_jac", launch_reproj_jac(dv, stream, 0));
          KT("k_imu_block", launch_imu_delta(dv, stream, 0)); KT("k_imu_jac", launch_imu_jac(dv, wcur, stream, 0));
        }
      }
      KT("k_chain_init", launch_chain_init(dv, stream));
      KT("k_chain_fwd", launch_chain_fwd(dv, stream));
      if (dv.gram_top_stride == 0) KT("k_chain_gram", launch_chain_gram(dv, stream));      // (early Gram: the sums ride in the top level's launch)
      else if (top_gram_launch) KT("k_chain_gram(top)", launch_chain_gram_top(dv, stream));
      KT("k_part_sum", launch_part_sum(dv, stream));
      int rc = VC_OK;
      if (sharded()) {
        KT("k_reduced(assemble)", launch_reduced(dv, 1, stream));
        KT("allreduce(S)", rc = do_allreduce(dv.Sbuf, D * D + 3 * D + 2, 0)); if (rc) return rc;
        KT("k_reduced(solve)", launch_reduced(dv, 2, stream));
      } else {
        KT("k_reduced", launch_reduced(dv, 0, stream));
      }
      // the trial IMU parameters exist: the interval deltas of the trial point run on the second stream next to the chain's
      // back-substitution (they depend on no pose)
      if (!serial_weights) {
        if (fs) launch_wait_flag(dv, 1, pass_seq, stream2);      // this pass's k_reduced
        else {
          HIP_OK(hipEventRecord(ev_reduced, stream));
          HIP_OK(hipStreamWaitEvent(stream2, ev_reduced, 0));
        }
        KT2("k_imu_block(trial)", launch_imu_delta(dv, stream2, 1));
      }
      {
        // (a captured pass freezes its arguments, pass_id among them: from the second replay on the ready words of the fused
        //  back-substitution would already hold a number >= it and its consumers would not wait -- one launch per level there)
        long long* const ready = dv.cready;
        if (use_graphs) dv.cready = nullptr;
        KT("k_chain_back", launch_chain_solve_b(dv, stream));
        dv.cready = ready;
      }
      // trial point: both sweeps in trial mode on the main stream, the IMU blocks with the weights this pass has just updated
      // (second stream: weight update, then the deltas -- ev_weights covers both); the decision follows without another
      // cross-stream hop (each costs 6-13 us on the device's timeline)
      if (upd) wcur = 1 - wcur;
      if (!serial_weights && jac_on_stream2) {
        // the IMU blocks' final stage (needs the trial poses) beside the vision sweep: the second stream is already past its
        // deltas when the back-substitution ends
        // (flag hand-overs: k_imu_block(trial), the kernel before it on the second stream, has waited for the flag the first
        // workgroup of k_reproj_jac(trial) sets -- one thread, before the kernel ended.  Letting k_imu_jac's own workgroups wait
        // at their entry was tried: 250 workgroups each invalidating the L2 under the running vision sweep, both kernels 2.3x
        // slower; a waiting kernel of its own costs 5 us on this stream's queue)
        if (!fs_trial) {
          HIP_OK(hipEventRecord(ev_back, stream));
          HIP_OK(hipStreamWaitEvent(stream2, ev_back, 0));
        }
        KT2("k_imu_jac(trial)", launch_imu_jac(dv, wcur, stream2, 1));
        if (!fs_trial) HIP_OK(hipEventRecord(ev_weights, stream2));      // (flag hand-overs: k_final ends on the second count of k_imu_jac's workgroups)
        KT("k_reproj_jac(trial)", launch_reproj_jac(dv, stream, 1));
        if (!fs_trial) HIP_OK(hipStreamWaitEvent(stream, ev_weights, 0));      // (flag hand-overs: k_final waits for the second stream itself)
      } else {
        if (!serial_weights) HIP_OK(hipEventRecord(ev_weights, stream2));
        KT("k_reproj_jac(trial)", launch_reproj_jac(dv, stream, 1));
        if (!serial_weights) HIP_OK(hipStreamWaitEvent(stream, ev_weights, 0));
        else KT("k_imu_block(trial)", launch_imu_delta(dv, stream, 1));
        KT("k_imu_jac(trial)", launch_imu_jac(dv, wcur, stream, 1));
      }
      if (sharded()) {
        KT("k_final(reduce)", launch_final(dv, 1, stream));
        KT("allreduce(step scalars)", rc = do_allreduce(dv.gath, world * kNumScal, 0)); if (rc) return rc;
        KT("k_final(decide)", launch_final(dv, 2, stream));
      } else {
        KT("k_final", launch_final(dv, 0, stream));
      }
      prev_pass_signals = fs;
      return VC_OK;
    }
    dv.sync_seq = 0; dv.final_wait = 0; dv.block_wait = 0;
    // merged decision (single process): control records alternate, pass k judges pass k-1 at the head of k_frame_schur
    const bool merged = merged_enabled && !use_graphs;      // (a captured graph has fixed kernel arguments)
    dv.shard_src = sharded() ? 1 : 0;
    if (merged) {
      if (first_pass) kpass = 0;
      dv.merged = 1; dv.par = kpass & 1; dv.ctrl = d_ctrl.p + (kpass & 1); dv.ctrl_prev = d_ctrl.p + ((kpass + 1) & 1);
      ++kpass;
    }
    if (first_pass || !dv.fused) KT("k_reproj_jac", launch_reproj_jac(dv, stream));
    KT("k_frame_schur+k_part_sum", launch_frame_schur(dv, stream));
    int rc = VC_OK;
    if (sharded()) {
      launch_reduced(dv, 1, stream);
      rc = do_allreduce(dv.Sbuf, D * D + 3 * D + 2, 0); if (rc) return rc;
      launch_reduced(dv, 2, stream);
    } else {
      KT("k_reduced", launch_reduced(dv, 0, stream));
    }
    KT("k_trial", launch_trial(dv, stream));
    if (sharded()) {
      launch_final(dv, 1, stream);
      rc = do_allreduce(dv.gath, world * kNumScal, 0); if (rc) return rc;
      if (!merged) launch_final(dv, 2, stream);      // merged: the next pass's frame elimination combines the ranks and decides
    } else if (!merged) {
      KT("k_final", launch_final(dv, 0, stream));
    }
    return VC_OK;
  }
  // merged mode: judge the last enqueued pass; afterwards `ctrl_result()` is the record to read back
  void finish_batch() {
    if (!dv.merged) return;
    dv.par = kpass & 1; dv.ctrl = d_ctrl.p + (kpass & 1); dv.ctrl_prev = d_ctrl.p + ((kpass + 1) & 1);
    launch_final_merged(dv, stream);
  }
  const Ctrl* ctrl_result() const { return dv.merged ? d_ctrl.p + (kpass & 1) : d_ctrl.p; }
  bool merged_enabled = true;
  // a fresh control record goes to buffer 0; buffer 1 (the "previous pass" of the first pass in merged mode) is blanked
  int upload_ctrl(const Ctrl* c) {
    launch_set_ctrl(d_ctrl.p, *c, stream);
    kpass = 0;
    return VC_OK;
  }
  void init_ctrl(Ctrl* c) {
    std::memset(c, 0, sizeof(Ctrl));
    c->radius = 1e4; c->decrease_factor = 2.0;
    c->ftol = function_tolerance; c->gtol = gradient_tolerance; c->ptol = parameter_tolerance; c->mult = (double)vis_mult;
    c->imu_mult = (double)imu_mult;
    c->cur = cur; c->reuse_diag = 0; c->need_lin = 1; c->init_scale = 1; c->max_iters = max_iters;
    c->first = 1; c->trace_cap = trace_cap; c->stage = stage;
  }

  // A device-flag hand-over ran into its bound (vc_kutil.hpp: spin_until_flag): the device has withheld the decision of pass
  // `abort_seq` and of everything queued behind it, the accepted state and the control record are those of the last valid decision.
  // Say so, switch this calibrator to event hand-overs for good, and put the solve back on its feet: both streams idle, flags
  // cleared, the weight buffer that pass was reading current again, `done` cleared and a fresh linearisation at the accepted state
  // requested (what the trial sweeps of the last judged pass left behind may be incomplete).  The passes that follow repeat the
  // withheld ones with events: the same iterates as a run that never used the flags.
  int resume_after_sync_timeout(const Ctrl& c) {
    HIP_OK(hipStreamSynchronize(stream));
    if (stream2) HIP_OK(hipStreamSynchronize(stream2));
    ++sync_timeouts;
    flag_sync = false;
    std::fprintf(stderr, "vicalib_amd: a device-flag hand-over between the two streams ran into its bound in LM pass %d (the streams share a hardware "
                         "queue, a tool serialises the queues, or several processes share the device); no step was taken on its data -- resuming the "
                         "solve with event hand-overs, which this calibrator keeps from now on (VICALIB_AMD_FLAG_SYNC=0 selects them from the start)\n",
                 c.passes + 1);
    wcur = wr_ring[c.abort_seq & 15];
    if (d_sync.p) HIP_OK(hipMemsetAsync(d_sync.p, 0, kSyncWords * sizeof(long long), stream));
    HIP_OK(hipMemsetAsync(dv.flags + 4, 0, 4 * sizeof(int), stream));      // numeric-failure marks of the void passes
    Ctrl r = c;
    // need_lin stays as the last decision left it: after a rejected step the linearisation in place is the one made when the state was
    // accepted, with the IMU weights of THAT pass -- the weights have moved on since (they are updated every pass), so linearising
    // again here would not reproduce it (costs of the following rejected steps off by 1e-7 relative: the time-out test in the middle of
    // a rejected streak); the void passes only wrote the trial-side buffers, the accepted state's records are intact
    r.done = 0; r.abort_seq = 0;
    HIP_OK(hipMemcpyAsync(d_ctrl.p, &r, sizeof(Ctrl), hipMemcpyHostToDevice, stream));
    HIP_OK(hipStreamSynchronize(stream));
    pin->down.done = 0;
    prev_pass_signals = false;
    return VC_OK;
  }
  // The trust-region loop (ceres::Solve :956 with LEVENBERG_MARQUARDT).  The loop itself runs
  // on the device (lm_decide in vc_kernels.hip); the host enqueues passes in batches and polls Ctrl::done.
  int solve_once(Termination* term, double* final_cost, long* nres) {
    RoctxRange rr("vicalib_amd: solve (ceres::Solve of one stage)");
    if (device_dirty) { int rc = upload(); if (rc) return rc; }
    *nres = 2L * ((long)dv.n_obs * vis_mult - n_one_less) + (dv.imu_on ? 9L * imu_mult * std::max(0, dv.n_frames - 1) : 0L);
    if (sharded()) {
      // The global residual count changes with the observation set and with the multiplicities.  The test for a fresh collective
      // must not depend on anything rank-local (a rank that skipped it while another entered would hang the job) -- not on
      // device_dirty, which a mutator called on one rank only would set there alone: the key is the public solve call (every rank
      // enters Solve() / vc_run_iterations together, they contain collectives anyway) and the multiplicities, which all ranks
      // bump together, outlier removal included.  One collective per stage of a solve.
      if (nres_epoch_cached != solve_epoch || nres_mult_cached[0] != vis_mult || nres_mult_cached[1] != imu_mult) {
        std::vector<double> v = {(double)*nres};
        int rc = host_allreduce_sum(v); if (rc) return rc;
        nres_global_cached = (long)v[0]; nres_mult_cached[0] = vis_mult; nres_mult_cached[1] = imu_mult; nres_epoch_cached = solve_epoch;
      }
      *nres = nres_global_cached;
    }
    if (trace_cap < max_iters + 8) { trace_cap = max_iters + 8; HIP_OK(d_trace.alloc((size_t)trace_cap * kTraceCols)); dv.trace = d_trace.p; }
    // the progress word is written by the device and polled by the host: coherent (fine-grained), mapped memory whatever
    // HIP_HOST_COHERENT says -- with a non-coherent allocation the host would never see the device's stores
    if (!pin) HIP_OK(hipHostMalloc((void**)&pin, sizeof(Pinned), hipHostMallocCoherent | hipHostMallocMapped));
    init_ctrl(&pin->up);
    { int rcu = upload_ctrl(&pin->up); if (rcu) return rcu; }
    // a wait that ran into its bound in a pass queued past the end of the previous solve (nobody judged it, nobody reported it) must
    // not void this solve's first pass: the sticky word starts every solve clear
    if (d_sync.p) HIP_OK(hipMemsetAsync(d_sync.p + 6, 0, sizeof(long long), stream));
    pre_weights_fresh = false; pre_weights_pending = false;
    if (dv.imu_on && dv.weights_on) {     // UpdateImuWeights() before ceres::Solve (vicalibrator.h:955)
      if (!serial_weights && stream2) { HIP_OK(hipEventRecord(ev_pre, stream)); pre_weights_pending = true; }
      dv.sync_seq = 0;      // (not a pass: a sticky time-out mark left by the previous solve's last pass must not make this update skip itself)
      launch_imu_weights(dv, wcur, stream); wcur = 1 - wcur;
      pre_weights_fresh = true;
    }
    const size_t trace_bytes = (size_t)std::min(trace_cap, 64) * kTraceCols * 8;
    int guard = 0, n_enq = 0;
    bool first_enq = true;              // the next pass enqueued linearises at the accepted state (start of the solve, resume after a flag time-out)
    double enqueue_ms = 0.0, wait_ms = 0.0;
    const auto tso0 = std::chrono::steady_clock::now();
    for (;;) {                          // (one round, unless a device-flag hand-over runs into its bound: then a second one, with events)
    const bool feed = !sharded() && !use_graphs && feed_passes && dv.imu_on;
    if (feed) {
      // Single process, visual-inertial passes (18 launches at cfg3, ~270 us): the deciding thread publishes (decisions << 32 | done) to a page-locked word after every decision and
      // the host keeps kAhead passes queued beyond the last decision it has seen -- no stream synchronisation inside the
      // solve (each one drains the queue: ~40 us of idle device), at most kAhead passes enqueued past the end (they return at
      // their first instruction).  Enqueueing a pass takes the host a fraction of the pass's run time.
      // kAhead starts at 1 and grows (up to 4, kept for the calibrator's lifetime) whenever the host finds every enqueued pass
      // already decided -- it came back late (a busy host: one box of the pool ran cfg3 at 0.40 instead of 0.30 ms per pass, with
      // the launch-ahead schedules unaffected) and the device has been idle; a longer queue rides such gaps out.
      int& kAhead = feed_ahead;
      volatile unsigned long long* prog = &pin->progress;
      *prog = (unsigned long long)(unsigned)n_enq << 32;      // (0 at the start of a solve; the decisions taken so far when a solve is resumed)
      dv.host_progress = &pin->progress; dv.host_ctrl = &pin->dev; dv.host_trace = pin->trace;
      bool done_seen = false;
      auto t_seen = std::chrono::steady_clock::now();
      unsigned long long last = 0ull;
      while (should_run) {
        const unsigned long long f = *prog;
        if ((unsigned)(f & 0xffffffffull & ~(unsigned long long)kProgressLikelyLast) != 0u) { done_seen = true; break; }     // Ctrl::done
        if (f != last) { last = f; t_seen = std::chrono::steady_clock::now(); }
        const int decided = (int)(f >> 32);
        if (n_enq >= max_iters + 8) break;
        if (!first_enq && decided >= n_enq && kAhead < 4 && !(f & kProgressLikelyLast)) ++kAhead;
        // the device expects the pass after the last decision to end the solve (lm_decide_local: likely_last): nothing is queued
        // past it -- a pass queued past the end costs ~90 us of empty launches at cfg3 before the stream is free again, a wrong guess
        // one host round trip
        const int ahead = (f & kProgressLikelyLast) ? 0 : kAhead;
        if (n_enq - decided <= ahead) {
          int rc = enqueue_pass(first_enq); if (rc) { dv.host_progress = nullptr; return rc; }
          first_enq = false;
          ++n_enq; t_seen = std::chrono::steady_clock::now();
        } else {
          __builtin_ia32_pause();
          // the queue is full: nothing to do until the device decides a pass (~0.3 ms); past ~50 us without news, yield the core
          const auto idle = std::chrono::steady_clock::now() - t_seen;
          if (idle > std::chrono::microseconds(50)) std::this_thread::yield();
          if (idle > std::chrono::seconds(5)) {          // a stuck device (or a progress word the host cannot see): say so, then
            std::fprintf(stderr, "vicalib_amd: no progress from the device for 5 s (%d passes queued, %d decided) -- falling back to a "
                                 "synchronising read\n", n_enq, decided);                     // fall through to the synchronising read
            break;
          }
        }
      }
      dv.host_progress = nullptr; dv.host_ctrl = nullptr; dv.host_trace = nullptr;
      finish_batch();
      std::atomic_thread_fence(std::memory_order_acquire);
      if (done_seen && !ktime_on && pin->dev.trace_len <= 64 && pin->dev.done != kDoneSyncTimeout) {
        // the deciding thread has left the record and the trace rows in page-locked memory before it said `done`: nothing to
        // copy and nothing to wait for -- the passes queued past the end (they return at their first instruction) drain while the
        // host goes on; whatever the caller enqueues next is ordered behind them by the streams
        pin->down = pin->dev;
      } else {
        HIP_OK(hipMemcpyAsync(&pin->down, ctrl_result(), sizeof(Ctrl), hipMemcpyDeviceToHost, stream));
        HIP_OK(hipMemcpyAsync(pin->trace, d_trace.p, trace_bytes, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        if (ktime_on) kt_collect();
      }
    }
    // Sharded (every rank must run the same schedule: the passes contain collectives), graph replay and the vision-only path
    // (four launches of ~13 us per pass, a handful of passes per solve: measured 60 vs 63 us per iteration at cfg2 -- the
    // passes fed past the end cost more there than one synchronisation): batches.  First batch =
    // what the previous solve needed (repeated solves of similar problems: no wasted launches, one host sync per solve); then
    // small top-up batches until the device reports `done`.
    // (visual-inertial: a first batch of at most 16 passes: a long previous solve -- stage C's 38 iterations ahead of stage D's 12 -- must not
    //  queue dozens of passes past the end; then top-ups of 8: one synchronisation per ~2 ms of device work)
    int batch = std::max(1, std::min(dv.imu_on ? std::min(expected_passes, 16) : expected_passes, max_iters + 1));
    while (!feed || (!pin->down.done && should_run && n_enq < max_iters + 8)) {
      const auto tq0 = std::chrono::steady_clock::now();
      for (int b = 0; b < batch; ++b) {
        const bool first = first_enq; first_enq = false; ++n_enq;
        int rc = (first || sharded() || !use_graphs) ? enqueue_pass(first) : launch_pass_graph();
        if (rc) return rc;
      }
      finish_batch();
      enqueue_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tq0).count();
      HIP_OK(hipMemcpyAsync(&pin->down, ctrl_result(), sizeof(Ctrl), hipMemcpyDeviceToHost, stream));
      HIP_OK(hipMemcpyAsync(pin->trace, d_trace.p, trace_bytes, hipMemcpyDeviceToHost, stream));
      const auto tw0 = std::chrono::steady_clock::now();
      HIP_OK(hipStreamSynchronize(stream));
      wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count();
      if (ktime_on) kt_collect();
      // Stop() is a collective decision when the frames are sharded: a rank that left its enqueue loop alone would leave
      // its peers waiting in the next all-reduce (every rank runs the same batch schedule, so the counts line up)
      // (every rank holds the same control record: a finished solve ends here on all of them without another collective)
      if (pin->down.done) break;
      if (sharded()) {
        std::vector<double> v = {should_run ? 0.0 : 1.0};
        int rc = host_allreduce_sum(v); if (rc) return rc;
        if (v[0] > 0.0) should_run = false;
      }
      if (!should_run || ++guard > max_iters + 8) break;
      batch = dv.imu_on ? 8 : 2;        // (vision-only passes are 50 us: a synchronisation every two of them was the better trade there)
    }
    if (pin->down.done != kDoneSyncTimeout) break;
    { int rc = resume_after_sync_timeout(pin->down); if (rc) return rc; }
    n_enq = pin->down.passes; first_enq = true; guard = 0;
    }
    const Ctrl c = pin->down;
    pre_weights_fresh = false; pre_weights_pending = false;      // (a solve that queued no pass must not leave them to a later stand-alone pass)
    if (std::getenv("VICALIB_AMD_TIMING") && enqueue_ms > 0.0)
      std::fprintf(stderr, "[vicalib_amd]   solve: %d passes enqueued in %.3f ms of host time (batched schedule), %d decided; waited %.3f ms for the device, %.3f ms in all\n",
                   n_enq, enqueue_ms, c.passes, wait_ms, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tso0).count());
    expected_passes = std::max(1, c.passes);
    const int n = std::min(c.trace_len, trace_cap);
    std::vector<double> rows((size_t)std::max(n, 1) * kTraceCols);
    if (n <= 64) std::memcpy(rows.data(), pin->trace, (size_t)n * kTraceCols * 8);
    else HIP_OK(hipMemcpy(rows.data(), d_trace.p, (size_t)n * kTraceCols * 8, hipMemcpyDeviceToHost));
    {
      std::lock_guard<std::mutex> lk(result_mutex);
      for (int i = 0; i < n; ++i) {
        const double* r = &rows[(size_t)i * kTraceCols];
        IterRecord rec = {(int)r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], (int)r[8], (int)r[9]};
        trace.push_back(rec);
      }
    }
    cur = c.cur;
    num_iterations += (unsigned)c.num_callbacks;
    jac_sweeps += c.jac_sweeps; res_sweeps += c.res_sweeps;
    last_iters = c.iter;
    *final_cost = c.cost;
    switch (c.done) {
      case kDoneConvergence: *term = kConvergence; break;
      case kDoneUserSuccess: *term = kUserSuccess; break;
      case kDoneFailure: *term = kFailure; break;
      default: *term = kNoConvergence; break;
    }
    return VC_OK;
  }
  int last_iters = 0;
  // one pass with the decision logic on hold (parity hooks): linearise at the accepted state
  int linearize_hold(double radius, double* cost) {
    Ctrl c;
    init_ctrl(&c);
    c.hold = 1; c.radius = radius; c.first = 0;
    { int rcu = upload_ctrl(&c); if (rcu) return rcu; }
    int rc = enqueue_pass(true, true); if (rc) return rc;
    finish_batch();
    HIP_OK(hipMemcpyAsync(&c, ctrl_result(), sizeof(Ctrl), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    if (cost) *cost = c.cost;
    return VC_OK;
  }

  // per-camera RMSE, vicalibrator.h:958-971 (unrobustified, latest copy)
  int compute_rmse() {
    const int C = (int)cams.size();
    launch_reproj_res(dv, cur, 1.0, stream);
    launch_cam_sq(dv, d_tmp.p, stream);
    int rc = do_allreduce(d_tmp.p, 2 * C, 0); if (rc) return rc;
    double h[2 * kMaxCams];
    HIP_OK(hipMemcpyAsync(h, d_tmp.p, sizeof(double) * 2 * C, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    std::lock_guard<std::mutex> lk(result_mutex);
    cam_rmse.assign(C, 0.0);
    for (int c = 0; c < C; ++c) cam_rmse[c] = std::sqrt(0.5 * h[2 * c] / h[2 * c + 1]);
    return VC_OK;
  }
  // RemoveOutliers, vicalibrator.h:859-916
  int remove_outliers_pass() {
    const int C = (int)cams.size();
    std::vector<double> th(C);
    for (int c = 0; c < C; ++c) th[c] = outlier_threshold * cam_rmse[c];
    HIP_OK(hipMemcpyAsync(d_tmp.p + 32, th.data(), C * 8, hipMemcpyHostToDevice, stream));
    launch_outlier_mask(dv, cur, d_tmp.p + 32, d_mask.p, stream);
    std::vector<unsigned char> mask((size_t)dv.n_obs);
    if (!mask.empty()) HIP_OK(hipMemcpyAsync(mask.data(), d_mask.p, mask.size(), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    int rc = download_state(); if (rc) return rc;
    // The reference removes the blocks of the LATEST copy only (:911-914).  Vision-only: there is one copy, the
    // corner is gone.  With the IMU the stage loop re-adds every block right after (SetupProblem :641-649), so
    // an outlier ends up with one copy fewer than the inliers.
    const signed char mark = calibrate_imu ? 2 : 1;
    for (size_t k = 0; k < mask.size(); ++k) if (mask[k] && o_removed[h_obs_index[k]] == 0) o_removed[h_obs_index[k]] = mark;
    device_dirty = true; obs_dirty = true;
    return VC_OK;
  }

  // Gravity initialisation, vicalibrator.h:927-949: accel at the middle frame's time (offset 0), rotated into the world
  int init_gravity() {
    const int N = (int)frames.size(), n = (int)imu_t.size();
    gravity_initialized = true;
    // the middle frame of the WHOLE problem; when the frames are sharded its owner computes, everybody receives
    long mid = N / 2; bool mine = true;
    if (world > 1) {
      std::vector<double> table;
      int rc = gather_shard_info(&table); if (rc) return rc;
      mid = global_total / 2 - global_first;
      mine = mid >= 0 && mid < N;
    }
    double g[2] = {0.0, 0.0};
    if (mine && N > 0 && n > 0) {
      const HostFrame& fr = frames[mid];
      double a[3];
      const double time = fr.time;
      if (imu_t[0] > time) { for (int k = 0; k < 3; ++k) a[k] = imu_a[k]; }
      else if (imu_t[n - 1] <= time || n < 2) { for (int k = 0; k < 3; ++k) a[k] = imu_a[3 * (size_t)(n - 1) + k]; }
      else {
        int lo = 0, hi = n - 1;
        while (hi - lo > 1) { const int m2 = (lo + hi) >> 1; if (imu_t[m2] <= time) lo = m2; else hi = m2; }
        const double f = (time - imu_t[lo]) / (imu_t[lo + 1] - imu_t[lo]);
        for (int k = 0; k < 3; ++k) a[k] = imu_a[3 * (size_t)lo + k] * (1.0 - f) + imu_a[3 * (size_t)(lo + 1) + k] * f;
      }
      const double nrm = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
      const double gb[3] = {a[0] / nrm, a[1] / nrm, a[2] / nrm};
      double gw[3];
      quat_rotate(fr.T, gb, gw);
      g[0] = std::asin(gw[1]);
      g[1] = std::asin(-gw[0] / std::cos(g[0]));
    }
    if (world > 1) {
      std::vector<double> v = {g[0], g[1]};
      int rc = host_allreduce_sum(v); if (rc) return rc;
      g[0] = v[0]; g[1] = v[1];
    } else if (!(N > 0 && n > 0)) return VC_OK;
    { std::lock_guard<std::mutex> lk(result_mutex); g_dir[0] = g[0]; g_dir[1] = g[1]; }
    return VC_OK;
  }

  // SolveThread, vicalibrator.h:919-1040
  static constexpr int kMaxRepeats = 64;
  int stage_limit = -1;
  int solve() {
    // the worker thread of Start() (and any caller's thread) starts on device 0: bind this calibrator's device first
    HIP_OK(hipSetDevice(device));
    ++solve_epoch;
    // is_finished_ is sticky until Clear() (vicalibrator.h:246, :922): a finished calibrator's Start()/Solve() returns at once
    int status = VC_OK;
    int stages_done = 0;
    while (should_run && !is_finished) {
      if (is_visual_active) vis_mult += 1;                      // SetupProblem re-adds every block (:641-649)
      if (calibrate_imu && is_inertial_active) imu_mult += 1;   // :651-655
      if (is_inertial_active && !rotation_only && !gravity_initialized) { status = init_gravity(); if (status) break; }   // :927-949
      device_dirty = true;                                      // constancy flags may have changed
      if (stage_limit >= 0 && stages_done++ >= stage_limit) break;   // bench / test hook: the next stage is set up, not run
      bool stage_done = false;
      int inner = 0;
      // "Crank optimization" (:952): the same problem is solved again while Ceres reports NO_CONVERGENCE.  The reference
      // loops without bound; here a run of kMaxRepeats unconverged solves returns VC_ERR_NO_CONVERGENCE with the
      // multiplicities untouched (the outer loop is NOT re-entered, which would re-add every block).
      while (!stage_done && should_run && !is_finished) {
        if (inner++ >= kMaxRepeats) { status = VC_ERR_NO_CONVERGENCE; break; }
        if (o_frame.empty()) { is_finished = true; break; }
        Termination t; double fc = 0; long nr = 1;
        const bool timing = std::getenv("VICALIB_AMD_TIMING") != nullptr;
        auto now = []() { return std::chrono::steady_clock::now(); };
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        const auto t0 = now();
        if (device_dirty) { status = upload(); if (status) break; }
        const auto t1 = now();
        status = solve_once(&t, &fc, &nr); if (status) break;
        const auto t2 = now();
        status = compute_rmse(); if (status) break;
        status = download_state(); if (status) break;
        if (timing) std::fprintf(stderr, "[vicalib_amd] stage %d: upload %.3f ms, solve %.3f ms (%d iterations), rmse + download %.3f ms\n", stage, ms(t0, t1), ms(t1, t2), last_iters, ms(t2, now()));
        { std::lock_guard<std::mutex> lk(result_mutex); mse = fc / (double)std::max(1L, nr); }
        ++stage;
        if (t != kNoConvergence && calibrate_imu) {
          if (!is_inertial_active) is_inertial_active = true;                          // :978-981
          else if (rotation_only) { rotation_only = false; is_bias_active = true; }   // :982-990
          else if (!is_scale_active) is_scale_active = true;                           // :991-994
          else if (remove_outliers && !outliers_removed) { status = remove_outliers_pass(); outliers_removed = true; }
          else is_finished = true;
          stage_done = true;
        } else if (t != kNoConvergence) {
          if (remove_outliers && !outliers_removed) { status = remove_outliers_pass(); outliers_removed = true; }
          else is_finished = true;
        }
        if (status) break;
      }
      if (status) break;
    }
    if (!device_dirty) { int rc = download_state(); if (!status) status = rc; }
    return status;
  }
};

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
const char* vc_last_error(void) { return g_last_error.c_str(); }
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
