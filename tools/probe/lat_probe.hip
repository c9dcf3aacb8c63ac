// Dependent-chain latencies of a lone wavefront on gfx950 (cycles per link), tools/probe/lat_probe.hip
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/lat_probe tools/probe/lat_probe.hip && tools/probe/lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N 256
__device__ __forceinline__ double readlane_f64(double x, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane), hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
  return __hiloint2double(hi, lo);
}
template <int MODE>
__global__ void k(double* out, long long* cyc, double seed) {
  __shared__ double sh[128];
  const int lane = threadIdx.x;
  double x = seed + lane * 1e-3, y = seed * 0.5, acc = 0.0;
  sh[lane] = x; sh[64 + lane] = y;
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(x), "+v"(y), "+v"(acc));
  __builtin_amdgcn_sched_barrier(0);
  const long long t0 = __builtin_readcyclecounter();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (MODE == 0) x = __builtin_fma(x, y, y);                                   // dependent v_fma_f64
    else if (MODE == 1) x = x * y;                                               // dependent v_mul_f64
    else if (MODE == 2) x = __builtin_amdgcn_rsq(x) + 1.0;                       // v_rsq_f64 + v_add_f64
    else if (MODE == 3) x = __builtin_fma(readlane_f64(x, i & 31), y, x);       // 2 v_readlane -> SGPR operand of a v_fma_f64
    else if (MODE == 4) { const double d = readlane_f64(x, i & 31); x = d > 0.0 ? x * y : y; }      // uniform compare -> SALU -> select
    else if (MODE == 5) { sh[lane] = x; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); x = __builtin_fma(sh[(i & 31)], y, x); }   // LDS write -> broadcast read
    else if (MODE == 6) { const int lo = __builtin_amdgcn_ds_bpermute(4 * ((lane + i) & 63), __double2loint(x)), hi = __builtin_amdgcn_ds_bpermute(4 * ((lane + i) & 63), __double2hiint(x)); x = __builtin_fma(__hiloint2double(hi, lo), y, x); }
    else if (MODE == 7) { const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x150 + 3, 0xF, 0xF, false), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x150 + 3, 0xF, 0xF, false); x = __builtin_fma(__hiloint2double(hi, lo), y, x); }   // row_newbcast:3
    else if (MODE == 8) { x = __builtin_fma(x, y, y); acc = __builtin_fma(acc, y, x); }      // two chains interleaved (issue-bound?)
    else if (MODE == 9) { double r = __builtin_amdgcn_rsq(x); r = r * (1.5 - 0.5 * x * r * r); r = r * (1.5 - 0.5 * x * r * r); x = r + 1.0; }      // fast_rsqrt chain
    else if (MODE == 10) { const float f = (float)x; x = (double)__builtin_amdgcn_rsqf(f) + 1.0; }   // f32 rsq seed path
    else if (MODE == 11) { x = __builtin_fmaf((float)x, 1.0f, 1.0f); }          // cvt + f32 fma + cvt
    else if (MODE == 12) { x = __builtin_fma(x, y, y); x = x > 0.0 ? x : y; }   // v_fma_f64 + per-lane compare + 2 v_cndmask
    else if (MODE == 13) { x = __builtin_fma(x, sh[64 + (i & 31)], y); }       // v_fma_f64 whose operand is a fresh broadcast LDS read (independent of the chain)
    else if (MODE == 14) { x = __builtin_amdgcn_rcp(x) + 1.0; }                 // v_rcp_f64 + v_add_f64
    else if (MODE == 15) { x = __builtin_fma(x, y, y); acc = __builtin_fma(acc, y, y); seed = __builtin_fma(seed, y, y); }   // three chains
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(x), "+v"(acc));
  __builtin_amdgcn_sched_barrier(0);
  const long long t1 = __builtin_readcyclecounter();
  __builtin_amdgcn_sched_barrier(0);
  out[lane] = x + acc + seed;
  if (lane == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char* name, double* d_out, long long* d_cyc) {
  long long c = 0;
  for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, d_out, d_cyc, 1.0001); hipDeviceSynchronize(); }
  hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
  printf("%-64s %7.1f cycles per link\n", name, (double)c / N);
}
int main() {
  double* d_out; long long* d_cyc;
  hipMalloc(&d_out, 64 * 8); hipMalloc(&d_cyc, 8);
  run<0>("dependent v_fma_f64", d_out, d_cyc);
  run<1>("dependent v_mul_f64", d_out, d_cyc);
  run<2>("v_rsq_f64 + v_add_f64", d_out, d_cyc);
  run<3>("2 v_readlane -> v_fma_f64 with SGPR operand", d_out, d_cyc);
  run<4>("2 v_readlane -> uniform compare -> select -> v_mul_f64", d_out, d_cyc);
  run<5>("ds_write_b64 -> broadcast ds_read_b64 -> v_fma_f64", d_out, d_cyc);
  run<6>("2 ds_bpermute_b32 -> v_fma_f64", d_out, d_cyc);
  run<7>("2 v_mov_b32 dpp row_newbcast -> v_fma_f64", d_out, d_cyc);
  run<8>("two independent v_fma_f64 chains (per pair)", d_out, d_cyc);
  run<9>("fast_rsqrt (v_rsq_f64 + two Newton steps) + v_add_f64", d_out, d_cyc);
  run<10>("cvt f64->f32, v_rsq_f32, cvt, v_add_f64", d_out, d_cyc);
  run<11>("cvt, v_fma_f32, cvt", d_out, d_cyc);
  run<12>("v_fma_f64 + per-lane v_cmp + 2 v_cndmask", d_out, d_cyc);
  run<13>("v_fma_f64 with an independent broadcast ds_read_b64 operand", d_out, d_cyc);
  run<14>("v_rcp_f64 + v_add_f64", d_out, d_cyc);
  run<15>("three independent v_fma_f64 chains (per triple)", d_out, d_cyc);
  return 0;
}
