// Which (A lane, B lane) pairs feed which D lane in v_mfma_f64_4x4x4_f64?  One-hot A, B[l] = l + 1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const double* a, const double* b, double* d) {
  const int l = threadIdx.x, e = blockIdx.x;
  d[e * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[e * 64 + l], b[l], 0.0, 0, 0, 0);
}
int main() {
  std::vector<double> a(64 * 64, 0.0), b(64), d(64 * 64);
  for (int e = 0; e < 64; ++e) a[e * 64 + e] = 1.0;
  for (int l = 0; l < 64; ++l) b[l] = l + 1;
  double *da, *db, *dd;
  hipMalloc(&da, a.size() * 8); hipMalloc(&db, 512); hipMalloc(&dd, d.size() * 8);
  hipMemcpy(da, a.data(), a.size() * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, 0, da, db, dd);
  hipMemcpy(d.data(), dd, d.size() * 8, hipMemcpyDeviceToHost);
  for (int la = 0; la < 64; ++la) {
    printf("A lane %2d ->", la);
    for (int ld = 0; ld < 64; ++ld) if (d[la * 64 + ld] != 0.0) printf(" D%d<-B%d", ld, (int)d[la * 64 + ld] - 1);
    printf("\n");
  }
  return 0;
}
