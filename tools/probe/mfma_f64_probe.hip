// Micro-probe (MI355X): (1) lane layout of v_mfma_f64_4x4x4_f64, (2) cycles of the two f64 MFMA shapes,
// (3) does f64 VALU work overlap with f64 MFMA work inside one wavefront / across two wavefronts of a SIMD?
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe tools/probe/mfma_f64_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void k_layout(const double* a, const double* b, double* d) {
  const int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
template <int MODE>   // 0: 16x16x4 only, 1: 4x4x4 only, 2: VALU fma only, 3: 16x16x4 + VALU interleaved, 4: 4x4x4 + VALU interleaved
__global__ __launch_bounds__(256) void k_time(double* out, int iters, double seed) {
  v4d acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  double c0 = 0, c1 = 0;
  double x0 = seed + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  const double m = 0.999999, a = 1e-9;
  const double u = seed * 0.5 + threadIdx.x;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 3) {
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(u, u, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(u, u, acc1, 0, 0, 0);
    }
    if (MODE == 1 || MODE == 4) {
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(u, u, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(u, u, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(u, u, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(u, u, c1, 0, 0, 0);
    }
    if (MODE >= 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {     // 32 independent-ish f64 FMAs = 128 cycles of DP VALU
        x0 = x0 * m + a; x1 = x1 * m + a; x2 = x2 * m + a; x3 = x3 * m + a; x4 = x4 * m + a; x5 = x5 * m + a; x6 = x6 * m + a; x7 = x7 * m + a;
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[1] + c0 + c1 + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (double)(t1 - t0);
}
template <int MODE> double run(int blocks, int iters, double* d_out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_time<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.0);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_time<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double cyc; hipMemcpy(&cyc, d_out + (1 << 20), 8, hipMemcpyDeviceToHost);
  printf("mode %d blocks %d: %.3f ms, wave0 cycles/iter %.1f\n", MODE, blocks, ms, cyc / iters);
  return ms;
}
int main() {
  double *da, *db, *dd, *d_out;
  hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512); hipMalloc(&d_out, ((1 << 20) + 8) * 8);
  std::vector<double> a(64), b(64), d(64);
  srand(1);
  for (int i = 0; i < 64; ++i) { a[i] = (rand() % 1000) / 100.0; b[i] = (rand() % 1000) / 100.0; }
  hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, da, db, dd);
  hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost);
  // hypotheses: lane = 16*blk + 4*p + q ; A: (p,q) = (k,i) or (i,k) ; B: (k,j) or (j,k) ; D: (i,j) or (j,i)
  for (int ha = 0; ha < 2; ++ha) for (int hb = 0; hb < 2; ++hb) for (int hd = 0; hd < 2; ++hd) {
    double err = 0;
    for (int blk = 0; blk < 4; ++blk) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
      double s = 0;
      for (int k = 0; k < 4; ++k) {
        const double av = a[16 * blk + (ha ? 4 * i + k : 4 * k + i)], bv = b[16 * blk + (hb ? 4 * j + k : 4 * k + j)];
        s += av * bv;
      }
      err += fabs(s - d[16 * blk + (hd ? 4 * j + i : 4 * i + j)]);
    }
    printf("layout hypothesis A:%s B:%s D:%s  err %.3g\n", ha ? "lane=16b+4i+k" : "lane=16b+4k+i", hb ? "lane=16b+4j+k" : "lane=16b+4k+j", hd ? "lane=16b+4j+i" : "lane=16b+4i+j", err);
  }
  const int iters = 2000;
  for (int blocks : {256, 512, 1024, 2048}) {      // 1, 2, 4, 8 wavefronts per SIMD (256 CUs x 4 SIMDs)
    printf("--- %d workgroups of 4 waves\n", blocks);
    run<0>(blocks, iters, d_out); run<1>(blocks, iters, d_out); run<2>(blocks, iters, d_out); run<3>(blocks, iters, d_out); run<4>(blocks, iters, d_out);
  }
  return 0;
}
