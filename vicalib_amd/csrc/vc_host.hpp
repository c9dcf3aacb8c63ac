#pragma once
// vc_host.hpp -- what the host driver's translation units share: includes, the HIP_OK macro, device buffers, the packed upload, the
// target-point table, the run-time bindings of RCCL and roctx.  (The driver itself: vc_calibrator.hpp.)
//
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

namespace vch {

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

}  // namespace vch
using namespace vch;

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
  int (*CommCount)(void*, int*) = nullptr;           // optional: what the communicator itself says about its size and this rank
  int (*CommUserRank)(void*, int*) = nullptr;        // (checked against the caller's numbers at creation; reported by vc_shard_info)
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
    CommCount = (int (*)(void*, int*))dlsym(lib, "ncclCommCount");
    CommUserRank = (int (*)(void*, int*))dlsym(lib, "ncclCommUserRank");
    if (!GetUniqueId || !CommInitRank || !AllReduce || !CommDestroy) { AllReduce = nullptr; return false; }
    return true;
  }
};
inline RcclApi g_rccl;

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
inline RoctxApi g_roctx;
struct RoctxRange {
  bool live;
  explicit RoctxRange(const char* name) : live(g_roctx.load()) { if (live) (void)g_roctx.Push(name); }
  ~RoctxRange() { if (live) (void)g_roctx.Pop(); }
};
// text of the last failure of an entry point that has more to say than its status code (vc_last_error; per thread)
inline thread_local std::string g_last_error;
constexpr int kNcclDouble = 8, kNcclSum = 0, kNcclMax = 2;     // ncclDataType_t / ncclRedOp_t values of nccl.h

