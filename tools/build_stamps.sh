#!/bin/bash
# profiling builds of the library, one per stamped kernel (loaded through VICALIB_AMD_LIB by the stamp tools):
#   -DVC_W_STAMPS        k_imu_weights   tools/weights_stamps.py
#   -DVC_F2_STAMPS       k_chain_fwd2    tools/f2_stamps.py
#   -DVC_REDUCED_STAMPS  k_reduced       tools/reduced_stamps.py
#   -DVC_GRAM_STAMPS     k_chain_gram    tools/gram_stamps.py
#   -DVC_INIT_STAMPS     k_chain_init    tools/init_stamps.py      (round 5)
#   -DVC_FINAL_STAMPS    k_final         tools/final_stamps.py     (round 5)
#   -DVC_L0_STAMPS       k_chain_l0      tools/l0_stamps.py        (round 5)
#   -DVC_IB_STAMPS       k_imu_block     tools/ib_stamps.py        (round 6)
cd "$(dirname "$0")/.."
for spec in "W_STAMPS wstamps" "F2_STAMPS f2stamps" "REDUCED_STAMPS rstamps" "GRAM_STAMPS gstamps" "BACK_STAMPS bstamps" "JAC_STAMPS jstamps" "INIT_STAMPS istamps" "FINAL_STAMPS fstamps" "L0_STAMPS l0stamps" "IB_STAMPS ibstamps"; do
  set -- $spec
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function -DVC_$1 \
    -o tools/probe/libvicalib_amd_$2.so vicalib_amd/csrc/vc_kernels.hip vicalib_amd/csrc/vc_imu_kernels.hip vicalib_amd/csrc/vc_detect.hip vicalib_amd/csrc/vc_upload.cpp vicalib_amd/csrc/vc_pass.cpp vicalib_amd/csrc/vc_solve.cpp vicalib_amd/csrc/vc_capi.cpp 2>&1 | grep -i "error" &
done
wait
