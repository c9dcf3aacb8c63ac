// vc_kutil.hpp -- small device helpers shared by the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>

namespace vc {

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int kDotStride = 34;   // doubles per corner in LDS: 2 rows x 16 + 2 pad (272 B: conflict-free b128 stores)

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ double readlane_f64(double x, int lane /* wave-uniform */) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double x) {      // result valid in lane 0
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
  return x;
}
__device__ __forceinline__ double wave_allsum(double x) {   // result in every lane, fixed order
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}


}  // namespace vc
