#!/bin/bash
# Builds the C-ABI library for gfx950 in-tree: vicalib_amd/libvicalib_amd.so
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT="$HERE/../libvicalib_amd.so"
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function \
  -o "$OUT" "$HERE/vc_kernels.hip" "$HERE/vc_imu_kernels.hip" "$HERE/vc_detect.hip" "$HERE/vc_upload.cpp" "$HERE/vc_pass.cpp" "$HERE/vc_solve.cpp" "$HERE/vc_capi.cpp" "$@"
echo "built $OUT"
# the synthetic-problem generator (host only; test / bench infrastructure): vicalib_amd/libvicalib_synth.so
${CXX:-g++} -O2 -std=c++17 -fPIC -shared -Wall -pthread -o "$HERE/../libvicalib_synth.so" "$HERE/vc_synth.cpp"
echo "built $HERE/../libvicalib_synth.so"
# the command-line tool (host C++ over the C ABI only): vicalib_amd/vicalib
CXX=${CXX:-g++}
$CXX -O2 -std=c++17 -Wall -I"$HERE/../../include" -o "$HERE/../vicalib" "$HERE/../../apps/vicalib.cpp" \
  -L"$HERE/.." -lvicalib_amd -Wl,-rpath,'$ORIGIN' -Wl,-rpath-link,/opt/rocm/lib -pthread
echo "built $HERE/../vicalib"
