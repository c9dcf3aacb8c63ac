// vc_kutil.hpp -- small device helpers shared by the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include "vc_device.h"

namespace vc {

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int kDotStride = 34;   // doubles per corner in LDS: 2 rows x 16 + 2 pad (272 B: conflict-free b128 stores)

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// The same exchange between the lanes of ONE wavefront inside a workgroup of several: wavefront-scope fences.  (With workgroup
// scope every such point waits for the wavefront's outstanding GLOBAL loads and stores as well -- the prefetch of the next frame, the
// stores of the solved image -- once the workgroup has more than one wavefront.)  LDS operations of a wavefront execute in order.
__device__ __forceinline__ void wave_lds_sync_local() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double readlane_f64(double x, int lane /* wave-uniform */) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
  return __hiloint2double(hi, lo);
}
// DPP move of a double within rows of 16 lanes (two 32-bit moves): lanes of the banks (groups of four lanes) selected by BANK
// receive `src` through the row operation CTRL, the others keep `old`.  row_ror:n = 0x120 + n: lane i reads lane (i - n) mod 16.
template <int CTRL, int BANK>
__device__ __forceinline__ double dpp_row_f64(double old, double src) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xF, BANK, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xF, BANK, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double x) {      // result valid in lane 0
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
  return x;
}
// Sums of six per-lane values over the wavefront, all six results in every lane.  Instead of six butterflies (36 exchanges)
// the lanes first split the six values among themselves: after the xor-1, xor-2 and xor-4 exchanges (3 + 2 + 1 values
// travel) every lane owns ONE of the sums, partially reduced; three more exchanges finish it and six v_readlane pick the
// owners.  The summation order is fixed.
__device__ __forceinline__ void wave_sum6(const double* a, double* out, int lane) {
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
  double b[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double r = __shfl_xor(b0 ? a[i] : a[3 + i], 1, 64);       // even lanes keep sums 0..2, odd lanes 3..5
    b[i] = (b0 ? a[3 + i] : a[i]) + r;
  }
  const double r0 = __shfl_xor(b1 ? b[0] : b[2], 2, 64);            // bit1 = 0 keeps the first two, bit1 = 1 the third
  const double r1 = __shfl_xor(b1 ? b[1] : 0.0, 2, 64);
  const double c0 = (b1 ? b[2] : b[0]) + r0, c1 = b1 ? 0.0 : b[1] + r1;
  const double r2 = __shfl_xor(b2 ? c0 : c1, 4, 64);                // bit2 picks between the two
  double d = (b2 ? c1 : c0) + r2;
  d += __shfl_xor(d, 8, 64);
  d += __shfl_xor(d, 16, 64);
  d += __shfl_xor(d, 32, 64);
  // owner lanes: sum 0 -> lane 0, 1 -> lane 4, 2 -> lane 2, 3 -> lane 1, 4 -> lane 5, 5 -> lane 3
  out[0] = readlane_f64(d, 0); out[1] = readlane_f64(d, 4); out[2] = readlane_f64(d, 2);
  out[3] = readlane_f64(d, 1); out[4] = readlane_f64(d, 5); out[5] = readlane_f64(d, 3);
}
__device__ __forceinline__ double wave_allsum(double x) {   // result in every lane, fixed order
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}


// hipFuncAttributeMaxDynamicSharedMemorySize is a property of a function ON ONE DEVICE: what has been granted is remembered per device
// (a process-wide flag skipped the attribute on a second device and the > 64 KB launch there failed silently -- advice r5).
struct LdsGrant {
  size_t got[16] = {};
  bool need(size_t lds) {
    int dev = 0; (void)hipGetDevice(&dev); dev = (dev >= 0 && dev < 16) ? dev : 0;
    if (lds <= got[dev]) return false;
    got[dev] = lds; return true;
  }
};

// ---- cross-stream hand-overs through device flags (DevView::sync_flags) --------------------------------------------------
// Publishing "this kernel of pass sync_seq is done" to the other stream (one thread of a single-workgroup kernel, behind a
// workgroup barrier: everybody's stores have reached the L2, the device-scope fence writes them back before the flag moves).
// An event record on the main stream would cost it 5 us per hand-over (the record's barrier packet sits between two kernels
// of the critical path); the flag costs the producer one fence at its very end.
__device__ __forceinline__ void signal_flag(const DevView& v, int idx) {
  if (v.sync_seq > 0) {
    __threadfence();
    __hip_atomic_store(v.sync_flags + idx, v.sync_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// ... for a kernel that only announces that it has STARTED (what it publishes are its predecessor's results, which that kernel's end
// has written back already): no fence -- the device-scope fence of signal_flag writes back the XCD's L2, 3-6 us right behind a
// kernel that has just dirtied it, on the signalling thread and on everybody waiting for the flag
__device__ __forceinline__ void signal_started(const DevView& v, int idx) {
  if (v.sync_seq > 0) __hip_atomic_store(v.sync_flags + idx, v.sync_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The waiting side.  Bounded: a flag that never comes (the two streams sharing one hardware queue would do it: the producer
// queued behind the waiting kernel; a tool that serialises the kernels of all queues; several processes time-sliced on one
// device) must not hang the device -- and must not change a result either.  A wait that runs into its bound (~0.2 s) marks the
// pass it belongs to in a STICKY word, sync_flags[6] (the smallest pass number wins), and lets its kernel continue on whatever
// data there is: everything that can be wrong from here on lives in the trial-side buffers of that pass and the passes behind
// it, and the deciding thread of a marked pass (lm_decide) does not judge -- it ends the solve with kDoneSyncTimeout, the accepted
// state and the control record exactly as the last valid decision left them.  The host reports the time-out, switches to
// event hand-overs and resumes from there (vc_solve.cpp: solve_once): same iterates as a run without flags.  Once a pass is
// marked every later wait returns at once (no cascade of 0.2 s bounds through the passes already queued).
__device__ __forceinline__ long long sync_marked(const DevView& v) {
  return __hip_atomic_load(v.sync_flags + 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// marks pass `seq` (and everything queued behind it) void: the sticky word keeps the smallest pass number
__device__ __forceinline__ void mark_sync_timeout(const DevView& v, long long seq) {
  long long expect = 0;
  if (!__hip_atomic_compare_exchange_strong(v.sync_flags + 6, &expect, seq, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    __hip_atomic_fetch_min(v.sync_flags + 6, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (relaxed polls, ONE acquire at the end: an acquire load per poll invalidates the CU's L1 every time round -- 2-3x slower per hop; the
//  flag and the sticky word are requested together: one memory round trip per poll.)  ACQ = false: the waiting thread reads nothing of
//  what the flag publishes itself (k_imu_block's thread, which only holds its kernel's end back: the next kernel's start acquires).
template <bool ACQ = true>
__device__ __forceinline__ void spin_until_flag(const DevView& v, int idx, long long seq) {      // one thread
  long long n = 0;
  for (;;) {
    const long long f = __hip_atomic_load(v.sync_flags + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long m = sync_marked(v);
    if (f >= seq) break;
    if (m != 0 && m <= v.sync_seq) return;             // this pass is void already
    if (++n > v.sync_bound) { mark_sync_timeout(v, v.sync_seq); return; }
    __builtin_amdgcn_s_sleep(8);
  }
  if (ACQ) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
// ... and for a whole workgroup at its entry: thread 0 waits, then every wavefront drops what it may hold of the other stream's
// results (device-scope acquire).  For single-workgroup kernels: a grid of workgroups doing this invalidates the L2 under whatever
// runs beside it (k_imu_jac with 250 workgroups: itself and the vision sweep 2.3x slower)
__device__ __forceinline__ void workgroup_wait_flag(const DevView& v, int idx, long long seq) {
  if (threadIdx.x == 0) spin_until_flag(v, idx, seq);
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // (acquire only: nothing of this workgroup's needs writing back here)
}

}  // namespace vc
