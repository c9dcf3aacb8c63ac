#!/bin/bash
# profiling build of the library with the phase stamps of k_reduced compiled in (tools/reduced_stamps.py)
cd "$(dirname "$0")/.." && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function -DVC_REDUCED_STAMPS \
  -o tools/probe/libvicalib_amd_rstamps.so vicalib_amd/csrc/vc_kernels.hip vicalib_amd/csrc/vc_imu_kernels.hip vicalib_amd/csrc/vc_detect.hip vicalib_amd/csrc/vc_calibrator.cpp 2>&1 | grep -i "error"
