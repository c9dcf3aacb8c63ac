// Cold-code cost of a lone wavefront on gfx950: N dependent v_fma_f64 unrolled (8 bytes each), executed twice by a loop that is not unrolled --
// cycles of the first pass (instruction cache cold) against the second.  tools/probe/icache_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N>
__global__ void k(double* out, long long* cyc, double seed) {
  double x = seed + threadIdx.x * 1e-3, y = seed * 0.5;
  long long t[3];
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(x), "+v"(y));
    __builtin_amdgcn_sched_barrier(0);
    t[pass] = __builtin_readcyclecounter();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_fma(x, y, y);
    asm volatile("" : "+v"(x));
    __builtin_amdgcn_sched_barrier(0);
    if (pass == 1) t[2] = __builtin_readcyclecounter();
    else { long long tt = __builtin_readcyclecounter(); t[2] = tt; cyc[2] = tt - t[0]; }
  }
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) { cyc[1] = t[2] - t[1]; }
}
template <int N> void run(double* d_out, long long* d_cyc) {
  long long c[3] = {0, 0, 0};
  hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, d_out, d_cyc, 1.0001); hipDeviceSynchronize();
  hipMemcpy(c, d_cyc, 24, hipMemcpyDeviceToHost);
  printf("%6d instructions (%4d KB): first pass %8lld cycles (%.1f per instruction), second pass %8lld (%.1f)\n", N, N * 8 / 1024, c[2], (double)c[2] / N, c[1], (double)c[1] / N);
}
int main() {
  double* d_out; long long* d_cyc;
  hipMalloc(&d_out, 64 * 8); hipMalloc(&d_cyc, 24);
  run<256>(d_out, d_cyc); run<1024>(d_out, d_cyc); run<4096>(d_out, d_cyc); run<8192>(d_out, d_cyc);
  run<256>(d_out, d_cyc); run<4096>(d_out, d_cyc);
  return 0;
}
